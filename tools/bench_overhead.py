import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from argparse import Namespace
from ogmm_amd import synth, ops
from ogmm_amd.gmmreg import GMMReg
dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
m = GMMReg(512, 16, cfg); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
src, tgt, _, _ = synth.make_batch(0, 64, 1024); src, tgt = src.to(dev), tgt.to(dev)
starts = synth.fps_starts_for(0, 64, 1024)
def run(steps, timeline, use_starts=True):
    with torch.no_grad():
        for _ in range(3): m(src, tgt, fps_starts=starts if use_starts else None)
        if timeline:
            ops.GEMM_TIMELINE, ops.GEMM_TIMELINE_ONLY = [], {"f16x3"}
            m(src, tgt, fps_starts=starts)
            per = len(ops.GEMM_TIMELINE); ops.recycle_timing_events(ops.GEMM_TIMELINE)
            ops._EVENT_POOL.extend(torch.cuda.Event(enable_timing=True) for _ in range(2 * per * steps))
            ops.GEMM_TIMELINE = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): m(src, tgt, fps_starts=starts if use_starts else None)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        ops.GEMM_TIMELINE, ops.GEMM_TIMELINE_ONLY = None, None
    return 64 * steps / el
for rep in range(2):
    print("steps 20 timeline %.0f   steps 20 plain %.0f   steps 100 timeline %.0f   steps 100 plain %.0f   steps 100 plain random starts %.0f" %
          (run(20, True), run(20, False), run(100, True), run(100, False), run(100, False, False)))
