"""Soak of the pipelined head (GMMReg.pipeline_head): 600 eval forwards over four resident batches of two shapes, enqueued back to back; every output of every forward
must equal the serial reference of its batch bit for bit; reports throughput per 100 forwards and device-memory growth.
usage (GPU box): python3 tools/soak_pipelined.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
m = GMMReg(512, 16, cfg); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
batches = []
for i, (B, N) in enumerate(((64, 1024), (6, 1024), (64, 1024), (4, 717))):
    s, t, _, _ = synth.make_batch(10 * i, B, N, "partial")
    batches.append((s.to(dev), t.to(dev), synth.fps_starts_for(10 * i, B, N)))
with torch.no_grad():
    m.pipeline_head = False
    ref = []
    for s, t, st in batches:
        ref.append([x.clone() for x in m(s, t, fps_starts=st)]); torch.cuda.synchronize()
    m.pipeline_head = True
    bad = 0
    for rnd in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = []
        for j in range(100):
            s, t, st = batches[j % 4]
            outs.append((j % 4, [x.clone() for x in m(s, t, fps_starts=st)]))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        for b, o in outs:
            bad += int(not all(torch.equal(x, y) for x, y in zip(o, ref[b])))
        print("round %d: %.1f forwards/s, allocated %.0f MiB reserved %.0f MiB, forwards differing from the serial reference so far: %d" % (
            rnd, 100 / dt, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, bad))
assert bad == 0 and not m.fp16_overflowed()
print("SOAK OK: 600 pipelined forwards bit-identical to the serial reference")
