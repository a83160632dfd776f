import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
m = GMMReg(512, 16, cfg); synth.fill_state_dict(m.state_dict()); m = m.cuda().eval()
src, tgt, _, _ = synth.make_batch(0, 64, 1024, "partial"); st = synth.fps_starts_for(0, 64, 1024)
src, tgt = src.cuda(), tgt.cuda()
for chunk in (64, 32, 16, 8, 64):
    def run():
        for b0 in range(0, 64, chunk):
            m(src[b0:b0 + chunk], tgt[b0:b0 + chunk], fps_starts=st[:, b0:b0 + chunk])
    with torch.no_grad():
        run(); run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
    print("chunk %3d: %.2f ms per 64 pairs" % (chunk, (time.perf_counter() - t0) / 5 * 1e3))
