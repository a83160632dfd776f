"""Consecutive eval forwards (B = 64, N = 1024, J = 16) on ONE stream against the same forwards alternating between TWO streams (two forwards in flight: the
kernel-boundary bubbles of one are filled by the other), with and without the pipelined head.  usage (GPU box): python3 tools/two_streams_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
dev = torch.device("cuda", 0)
m = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
src, tgt, _, _ = synth.make_batch(0, 64, 1024, "partial"); src, tgt = src.to(dev), tgt.to(dev)
starts = synth.fps_starts_for(0, 64, 1024)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n_streams, K=24):
    outs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(K):
            with torch.cuda.stream(streams[i % n_streams]):
                outs.append(m(src, tgt, fps_starts=starts))
    torch.cuda.synchronize()
    return 64 * K / (time.perf_counter() - t0), outs
for ph in (False, True):
    m.pipeline_head = ph
    run(1, 6); run(2, 6)
    for rep in range(2):
        a, o1 = run(1); b, o2 = run(2)
        same = all(torch.equal(x, y) for x, y in zip(o1[-1], o2[-1]))
        print("pipeline_head=%s: one stream %.0f pairs/s, two streams alternating %.0f pairs/s (%+.1f %%), outputs identical: %s" % (ph, a, b, 100 * (b / a - 1), same))
