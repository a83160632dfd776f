"""Developer aid (GPU box): run-to-run determinism of consecutive eval forwards, with and without a synchronisation between them, compared output by output
and stage by stage (which intermediate is the first to differ)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
model = GMMReg(512, 16, cfg); synth.fill_state_dict(model.state_dict()); model = model.cuda().eval()
batches = []
for i, (B, N) in enumerate(((6, 1024), (4, 717), (6, 1024), (4, 717), (64, 1024), (64, 1024))):
    src, tgt, _, _ = synth.make_batch(40 + 10 * i, B, N, "partial")
    batches.append((src.cuda(), tgt.cuda(), synth.fps_starts_for(40 + 10 * i, B, N)))
torch.cuda.synchronize()
keys = ("knn_idx", "fps_anchor", "fps_J", "emb", "x0", "ft", "f", "o", "f2", "gamma", "mu", "muf", "near")
def run(flag, sync, capture=False):
    res = []
    with torch.no_grad():
        for s, t, st in batches:
            out = model(s, t, fps_starts=st, capture=capture)
            res.append([x.clone() for x in out] + ([model.last_intermediates[k].clone() for k in keys] if capture else []))
            if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return res
ref = run(False, True, True)
bad = 0
for rep in range(3):
    for tag, flag, sync, capt in (("sync capture", False, True, True), ("nosync", False, False, False), ("nosync capture", False, False, True)):
        got = run(flag, sync, capt)
        same = [all(torch.equal(x, y) for x, y in zip(a, b)) for a, b in zip(ref, got)]
        bad += sum(not v for v in same)
        if not all(same):
            for bi, (a, b) in enumerate(zip(ref, got)):
                if not same[bi]:
                    names = ["R", "t", "so", "to", "loss"] + list(keys)
                    print("  ", tag, "batch", bi, "differs in:", [n for n, x, y in zip(names, a, b) if not torch.equal(x, y)])
        print(rep, tag, "identical" if all(same) else same)
print("DETERMINISM", "OK" if bad == 0 else "FAILED (%d batch results differ)" % bad)
