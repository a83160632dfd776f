"""The pairs at the tail of the parity distribution, one by one (GPU box): is a distance above 1e-5 this path's arithmetic or the pair's conditioning?

    python tools/parity_outliers.py cfg1:128,188 cfg2:2060 n717:413,334,365

Per pair: the HIP forward with the default term budget, with three terms everywhere and on the exact-fp32 engine, each against the CPU oracle in fp32 (the
reference's arithmetic, 16 threads) -- and, as yard-sticks of the pair itself, the oracle against ITSELF: fp32 at 1 thread against 16 threads (the
reference's own run-to-run reproducibility) and fp32 against an fp64 evaluation with the fp32 run's kNN graph pinned (how well the reference's own
fp32 result is defined).  R distances in rad.  (One pair per forward: at that size the small-tile engines run, which always issue three terms -- the
"budget" column equals the "3 terms" one; the batch-level figure with the budget is the distribution's own.)"""
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import ogmm_oracle as O  # noqa: E402
from ogmm_amd import synth  # noqa: E402
from ogmm_amd.gmmreg import GMMReg  # noqa: E402

WORK = {"cfg1": (1024, 16, "partial"), "cfg2": (2048, 64, "partial"), "cfg3": (2048, 64, "room"), "n717": (717, 128, "partial")}


def main():
    print("%-6s %6s | %-32s | %-21s | %s" % ("", "pair", "HIP vs oracle fp32: budget / 3 terms / f32", "oracle 1 vs 16 threads", "oracle fp32 vs fp64   HIP(budget) vs fp64"))
    for spec in sys.argv[1:]:
        name, ids = spec.split(":")
        N, J, kind = WORK[name]
        cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
        model = GMMReg(512, J, cfg)
        synth.fill_state_dict(model.state_dict())
        P = {k: v.clone() for k, v in model.state_dict().items()}
        P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
        model = model.cuda().eval()
        for pid in (int(v) for v in ids.split(",")):
            src, tgt, _, _ = synth.make_batch(pid, 1, N, kind)
            starts = synth.fps_starts_for(pid, 1, N)
            hip = {}
            for tag, prec, budget in (("budget", "f16x3", None), ("x3", "f16x3", {}), ("f32", "f32", None)):
                model.precision = prec
                model.term_budget = dict(GMMReg(512, J, cfg).term_budget) if budget is None else budget
                with torch.no_grad():
                    hip[tag] = model(src.cuda(), tgt.cuda(), fps_starts=starts)[0].cpu()
            with torch.no_grad():
                torch.set_num_threads(16)
                cap = {}
                r16 = O.forward(P, cfg, src, tgt, starts, cap=cap)[0]
                torch.set_num_threads(1)
                r1 = O.forward(P, cfg, src, tgt, starts)[0]
                torch.set_num_threads(16)
                inj = {k: cap[k] for k in ("knn_idx_src", "knn_idx_tgt") if k in cap}
                r64 = O.forward(P64, cfg, src.double(), tgt.double(), starts, inject=inj or None)[0]
            d = lambda a, b: O.rotation_error_rad(a.double(), b.double()).max().item()  # noqa: E731
            print("%-6s %6d | %.2e / %.2e / %.2e       | %.2e              | %.2e              %.2e" % (
                name, pid, d(hip["budget"], r16), d(hip["x3"], r16), d(hip["f32"], r16), d(r1, r16), d(r16, r64), d(hip["budget"], r64)), flush=True)


if __name__ == "__main__":
    main()
