import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
dev = torch.device("cuda", 0)
m = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
src, tgt, _, _ = synth.make_batch(0, 64, 1024, "partial"); src, tgt = src.to(dev), tgt.to(dev)
starts = synth.fps_starts_for(0, 64, 1024)
with torch.no_grad():
    for _ in range(3): m(src, tgt, fps_starts=starts)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m(src, tgt, fps_starts=starts); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
print("host time of ONE forward call on an idle GPU (no back-pressure): %s ms" % ["%.2f" % (t * 1e3) for t in ts])
