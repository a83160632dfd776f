"""Small-batch latency of the eval forward: eager launches against the captured HIP graph (GMMReg.capture_graph)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg

dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
m = GMMReg(512, 16, cfg); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
for B in (1, 4, 16, 64):
    N = 1024
    src, tgt, _, _ = synth.make_batch(0, B, N)
    starts = synth.fps_starts_for(0, B, N)
    src, tgt, starts_d = src.to(dev), tgt.to(dev), starts.to(dev)
    run = m.capture_graph(B, N)
    with torch.no_grad():
        e = m(src, tgt, fps_starts=starts)
        e = [t.clone() for t in e]
        g = run(src, tgt, starts_d)
        same = all(torch.equal(a, b) for a, b in zip(e, g))
        res = {}
        for name, fn in (("eager", lambda: m(src, tgt, fps_starts=starts)), ("graph", lambda: run(src, tgt, starts_d))):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30): fn()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 30 * 1e3
    print("B=%-3d eager %7.3f ms  graph %7.3f ms  (%.0f -> %.0f pairs/s)  outputs identical: %s" % (B, res["eager"], res["graph"], B / res["eager"] * 1e3, B / res["graph"] * 1e3, same))
