"""Resident E/M kernel against the launch sequence (OGMM_EM_RESIDENT=0 in a child process would be needed for bitwise A/B; here: both via the C ABI env read once,
so this script is run twice by the caller and compares saved outputs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
out = sys.argv[1]
res = {}
for (C, N, J) in ((128, 2048, 64), (96, 2048, 64), (100, 2048, 64), (6, 2048, 64), (3, 1500, 32), (2, 717, 64), (5, 300, 20)):
    torch.manual_seed(C * 1000 + N + J)
    xyz = torch.randn(C, N, 3, device="cuda") * 0.5
    o = torch.rand(C, N, device="cuda")
    ids = ops.fps(xyz, J, None)
    g, pi, mu = ops.gmm_em(xyz, o, ids, engine="multi")
    torch.cuda.synchronize()
    res[(C, N, J)] = (g.cpu(), pi.cpu(), mu.cpu())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.gmm_em(xyz, o, ids, engine="multi")
    e1.record(); torch.cuda.synchronize()
    print("C=%d N=%d J=%d  %.1f us  finite=%s" % (C, N, J, e0.elapsed_time(e1) / 3 * 1e3, bool(torch.isfinite(g).all() and torch.isfinite(mu).all())))
if os.path.exists(out):
    ref = torch.load(out)
    for key, (g, pi, mu) in res.items():
        rg, rpi, rmu = ref[key]
        print(key, "max |dgamma| %.2e  |dpi| %.2e  |dmu| %.2e" % ((g - rg).abs().max(), (pi - rpi).abs().max(), (mu - rmu).abs().max()))
        bad = [(c, float((mu[c] - rmu[c]).abs().max())) for c in range(mu.shape[0]) if not torch.equal(mu[c], rmu[c])]
        if bad: print("   clouds that differ:", bad[:40])
else:
    torch.save(res, out)
