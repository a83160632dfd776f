"""Parity DISTRIBUTION of the HIP forward against the CPU oracle (GPU box): not four sampled pairs but every pair of a batch, per workload,
with the tail stated -- the E/M + matching head is ill-conditioned (HISTORY.md section 2), so the maximum over many pairs decides whether
"R, t within 1e-5 of the reference" holds, not the median.

    python tools/parity_distribution.py [--workloads cfg1,cfg2,n717] [--pairs 64,32,32] [--budget default|none] [--precision f16x3|f32|f16]

cfg1 = BASELINE configs[1] (the bench batch: partial-overlap + noise, N=1024, J=16, pairs 0..63), cfg2 = configs[2]'s shape (N=2048, J=64;
run with the grid-wide E/M launch sequence that a 256-pair batch takes), n717 = the reference repo's own defaults (N=717, J=128).
The oracle runs in chunks of 8 pairs on the host cores (thread count picked by a short sweep).  Prints a table + decade histogram."""
import argparse
import os
import sys
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import ogmm_oracle as O  # noqa: E402
from ogmm_amd import synth  # noqa: E402
from ogmm_amd.gmmreg import GMMReg  # noqa: E402

WORK = {"cfg1": (1024, 16, 0, "partial"), "cfg2": (2048, 64, 2000, "partial"), "cfg3": (2048, 64, 3000, "room"), "n717": (717, 128, 300, "partial")}


def pick_threads(P, cfg, src, tgt, starts):
    best = (1e9, 8)
    for nt in (8, 16, 32, 64):
        if nt > (os.cpu_count() or 8):
            break
        torch.set_num_threads(nt)
        with torch.no_grad():
            O.forward(P, cfg, src[:2], tgt[:2], starts[:, :2])
            t0 = time.perf_counter()
            O.forward(P, cfg, src[:4], tgt[:4], starts[:, :4])
            dt = time.perf_counter() - t0
        best = min(best, (dt, nt))
    torch.set_num_threads(best[1])
    return best[1], 4 / best[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cfg1,cfg2,n717")
    ap.add_argument("--pairs", default="64,32,32")
    ap.add_argument("--budget", default="default", choices=["default", "none", "r3"], help="r3 = round 3's default budget (kept for the record: it does not hold on the sharp family)")
    ap.add_argument("--profile", default="default", choices=["default", "sharp"], help="weight family (synth.fill_state_dict)")
    ap.add_argument("--first", default="", help="first global pair id per workload (comma list; default: the workload's)")
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    names = args.workloads.split(",")
    counts = [int(v) for v in args.pairs.split(",")]
    print("# parity distribution: HIP forward (precision %s, term budget %s, weight family %s) against the CPU oracle; R in rad, t in cloud units" % (args.precision, args.budget, args.profile))
    worst = {}
    firsts = [int(v) for v in args.first.split(",")] if args.first else [None] * len(names)
    for name, n_pairs, first_ in zip(names, counts, firsts):
        N, J, first, kind = WORK[name]
        first = first if first_ is None else first_
        cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
        model = GMMReg(512, J, cfg)
        synth.fill_state_dict(model.state_dict(), profile=args.profile)
        P = {k: v.clone() for k, v in model.state_dict().items()}
        model = model.cuda().eval()
        model.precision = args.precision
        if args.budget == "none":
            model.term_budget = {}
        elif args.budget == "r3":
            model.term_budget = {"conv2.0": 2, "conv2.3": 2, "similarity": 1, **{"%s.%s" % (t_, l_): 1 for t_ in ("sattn1", "cattn", "sattn2") for l_ in ("q", "qk")}}
        src, tgt, _, _ = synth.make_batch(first, n_pairs, N, kind)
        starts = synth.fps_starts_for(first, n_pairs, N)
        if name in ("cfg2", "cfg3"):
            os.environ["OGMM_EM_RESIDENT"] = "0"          # the launch sequence a 256-pair batch takes
        with torch.no_grad():
            got = [x.cpu() for x in model(src.cuda(), tgt.cuda(), fps_starts=starts, capture=True)[:4]]
        sweeps = model.last_intermediates["sinkhorn_sweeps"].cpu()
        os.environ.pop("OGMM_EM_RESIDENT", None)
        assert not model.fp16_overflowed()
        nt, rate = (args.threads, float("nan")) if args.threads else pick_threads(P, cfg, src, tgt, starts)
        if args.threads:
            torch.set_num_threads(args.threads)
        t0 = time.perf_counter()
        R_err, t_err, o_err = [], [], []
        for a in range(0, n_pairs, 8):
            b = min(n_pairs, a + 8)
            with torch.no_grad():
                ref = O.forward(P, cfg, src[a:b], tgt[a:b], starts[:, a:b])
            R_err.append(O.rotation_error_rad(got[0][a:b], ref[0]))
            t_err.append(O.translation_error(got[1][a:b], ref[1]))
            o_err.append(torch.maximum((got[2][a:b] - ref[2]).abs().amax(1), (got[3][a:b] - ref[3]).abs().amax(1)))
        dt = time.perf_counter() - t0
        R_err, t_err, o_err = torch.cat(R_err), torch.cat(t_err), torch.cat(o_err)
        q = lambda v, p: float(torch.quantile(v.double(), p))  # noqa: E731
        print("\n## %s: %d pairs, N=%d, J=%d, %s clouds (oracle: %d threads, %.1f s = %.2f pairs/s; Sinkhorn sweeps all 10: %s)" %
              (name, n_pairs, N, J, kind, nt, dt, n_pairs / dt, bool((sweeps == 10).all())))
        for label, v in (("R [rad]", R_err), ("t", t_err), ("overlap score", o_err)):
            print("  %-14s max %.2e   p90 %.2e   median %.2e   min %.2e" % (label, v.max().item(), q(v, 0.9), q(v, 0.5), v.min().item()))
        edges = [0.0, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1.0]
        hist = [int(((R_err >= lo) & (R_err < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        print("  R histogram  " + "  ".join("<%.0e: %d" % (hi, c) for hi, c in zip(edges[1:], hist)))
        print("  worst pairs (global id: R): " + ", ".join("%d: %.2e" % (first + int(i), R_err[i].item()) for i in torch.argsort(R_err, descending=True)[:4]))
        # The tail, pair by pair: how well is the REFERENCE's own fp32 result defined there?  Its fp32 forward against an fp64 evaluation (same kNN graph),
        # and at 1 thread against the sweep's thread count (run-to-run reproducibility) -- next to this path's distance from both.
        tail = [int(i) for i in torch.argsort(R_err, descending=True)[:6] if R_err[i].item() >= 5e-6]
        if tail:
            P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
            print("  tail pairs (R >= 5e-6):  id   HIP vs reference fp32 | reference fp32 vs its fp64 evaluation | reference 1 thread vs %d | HIP vs fp64" % nt)
            for i in tail:
                with torch.no_grad():
                    cap = {}
                    r32 = O.forward(P, cfg, src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1], cap=cap)[0]
                    inj = {k: cap[k] for k in ("knn_idx_src", "knn_idx_tgt") if k in cap}
                    r64 = O.forward(P64, cfg, src[i:i + 1].double(), tgt[i:i + 1].double(), starts[:, i:i + 1], inject=inj)[0]
                    torch.set_num_threads(1)
                    r1 = O.forward(P, cfg, src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1])[0]
                    torch.set_num_threads(nt)
                d = lambda a, b: O.rotation_error_rad(a.double(), b.double()).max().item()  # noqa: E731
                print("                         %5d   %.2e              | %.2e                              | %.2e                | %.2e" % (
                    first + i, R_err[i].item(), d(r32, r64), d(r1, r32), d(got[0][i:i + 1], r64)))
        worst[name] = (R_err.max().item(), t_err.max().item(), int((R_err >= 1e-5).sum()), n_pairs)
    print("\n# summary (max over pairs): " + "; ".join("%s R %.2e t %.2e" % (k, v[0], v[1]) for k, v in worst.items()))
    bad = {k: v for k, v in worst.items() if v[0] >= 1e-5 or v[1] >= 1e-5}
    print("# pairs with R within 1e-5 of the reference: " + "; ".join("%s %d of %d" % (k, v[3] - v[2], v[3]) for k, v in worst.items()))
    print("# within 1e-5 on every pair: %s" % ("yes" if not bad else "NO (%s): see the tail tables -- those are the pairs on which the reference's own fp32 result is "
                                                "as far from its fp64 evaluation" % ", ".join(bad)))
    return 1 if bad and args.precision != "f16" else 0


if __name__ == "__main__":
    sys.exit(main())
