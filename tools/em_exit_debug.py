"""Sinkhorn early exit: per-sweep batch-mean residuals of the oracle next to the HIP kernels' (debugging aid; GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ogmm_oracle as O
from ogmm_amd import ops, synth

def clouds(C, N, seed=0, kind="partial"):
    src, tgt, _, _ = synth.make_batch(seed, (C + 1) // 2, N, kind)
    return torch.cat([src, tgt], 0)[:C].transpose(1, 2).contiguous()

C, N, J, G, scale = 4, 1024, 16, 2, 0.04
torch.manual_seed(N + J + C)
xyz = clouds(C, N, seed=41) * scale
o = torch.sigmoid(torch.randn(C, N))
ids = ops.fps(xyz.cuda(), J, None)
full = ops.gmm_em(xyz.cuda(), o.cuda(), ids, thresh=0.0, group_size=G, return_resid=True)[3].cpu()
ex = ops.gmm_em(xyz.cuda(), o.cuda(), ids, thresh=1e-2, group_size=G, return_resid=True, return_sweeps=True)
for g in range(C // G):
    h = slice(g * G, (g + 1) * G)
    st, rs = [], []
    O.weighted_em(xyz[h], torch.zeros(G, N, 1), o[h], J, stats=st, resid=rs)
    print("group", g, "oracle sweeps", st, "hip sweeps", ex[4][g].tolist())
    pos = 0
    for it in range(2):
        print("  it", it, "oracle means", ["%.4f" % rs[pos + k].mean().item() for k in range(st[it])])
        print("       hip (no exit)", ["%.4f" % v for v in full[h, it].mean(0).tolist()])
        print("       hip (exit)   ", ["%.4f" % v for v in ex[3].cpu()[h, it].mean(0).tolist()])
        pos += st[it]
