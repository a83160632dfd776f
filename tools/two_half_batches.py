"""Experiment: the batch of 64 pairs as two half batches whose forwards run concurrently on two streams (latency-/VALU-bound kernels of
one half next to the GEMMs of the other) against one forward over the whole batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg

dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
B, N, J = 64, 1024, 16
src, tgt, _, _ = synth.make_batch(0, B, N)
starts = synth.fps_starts_for(0, B, N)
src, tgt = src.to(dev), tgt.to(dev)
models = []
for i in range(2):
    m = GMMReg(512, J, cfg); synth.fill_state_dict(m.state_dict()); models.append(m.to(dev).eval())
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
halves = [(src[:B // 2].contiguous(), tgt[:B // 2].contiguous(), starts[:, :B // 2].contiguous()),
          (src[B // 2:].contiguous(), tgt[B // 2:].contiguous(), starts[:, B // 2:].contiguous())]

def whole():
    return models[0](src, tgt, fps_starts=starts)

def split():
    outs = []
    for i in range(2):
        streams[i].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[i]):
            outs.append(models[i](*halves[i][:2], fps_starts=halves[i][2]))
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    return outs

with torch.no_grad():
    for name, fn in (("whole batch, one stream", whole), ("two half batches, two streams", split), ("whole batch, one stream", whole)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%-32s %7.2f ms/step  %7.0f pairs/s" % (name, dt * 1e3, B / dt))
    a = whole(); b = split()
    print("R diff", float((a[0] - torch.cat([b[0][0], b[1][0]])).abs().max()))
