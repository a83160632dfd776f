"""How long does the host take to ENQUEUE one eval forward (B = 64), against the GPU's time to run it?  usage (GPU box): python3 tools/host_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(model.state_dict()); model = model.to(dev).eval()
src, tgt, _, _ = synth.make_batch(0, 64, 1024, "partial"); starts = synth.fps_starts_for(0, 64, 1024)
src, tgt = src.to(dev), tgt.to(dev)
with torch.no_grad():
    for _ in range(5): model(src, tgt, fps_starts=starts)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): model(src, tgt, fps_starts=starts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.2f ms per forward, total %.2f ms per forward (GPU drained %.2f ms after the last enqueue)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
    run = model.capture_graph(64, 1024)
    for _ in range(3): run(src, tgt, fps_starts=starts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): run(src, tgt, fps_starts=starts)
    torch.cuda.synchronize()
    print("graph replay: %.2f ms per forward" % ((time.perf_counter() - t0) / n * 1e3))
