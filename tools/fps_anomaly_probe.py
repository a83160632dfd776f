"""Developer aid (GPU box, OGMM_FPS_BEHIND_KNN=1): what do the FPS chains look like when they differ?  The in-forward chains against a standalone, synchronised
recomputation of ops.fps on the same clouds and starts: per differing (set, cloud) the first differing position, how many positions differ, whether the
difference runs to the end of the chain (a diverged chain) or is a patch (an overwritten output), and whether the entries are valid point indices."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
model = GMMReg(512, 16, cfg); synth.fill_state_dict(model.state_dict()); model = model.cuda().eval()
for B, N in ((4, 717), (6, 1024), (64, 1024)):
    src, tgt, _, _ = synth.make_batch(40, B, N, "partial"); st = synth.fps_starts_for(40, B, N)
    src, tgt = src.cuda(), tgt.cuda()
    xyz = ops.pack_clouds(src, tgt)
    starts = st.reshape(3, 2 * B).to(torch.int32).cuda()
    ref = ops.fps(xyz, 128, starts).clone()
    torch.cuda.synchronize()
    n_bad = 0
    for rep in range(12):
        with torch.no_grad():
            model(src, tgt, fps_starts=st, capture=True)
        torch.cuda.synchronize()
        got = model.last_intermediates["fps_anchor"]
        d = got != ref
        if d.any():
            n_bad += 1
            for s_, c_ in sorted(set(map(tuple, d.nonzero()[:, :2].tolist())))[:4]:
                row = d[s_, c_]
                pos = row.nonzero().flatten()
                first, cnt = int(pos[0]), int(row.sum())
                vals = got[s_, c_]
                # is the in-forward chain a VALID FPS chain from its point of divergence?  (recompute: given the first `first` picks, the next pick must be the farthest point)
                print("  B=%d N=%d rep %d (set %d, cloud %d): first differing position %d, %d of %d positions differ, to the end: %s, all entries valid indices: %s, distinct: %s" % (
                    B, N, rep, s_, c_, first, cnt, 128, bool(row[first:].all()), bool(((vals >= 0) & (vals < N)).all()), int(vals.unique().numel())))
    print("B=%d N=%d: %d of 12 forwards with differing chains" % (B, N, n_bad), flush=True)
