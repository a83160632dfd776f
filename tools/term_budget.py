"""Per-layer term budget of the fp16-split GEMM engine, measured on the CPU oracle (no GPU needed).

For every GEMM of the forward, ONE layer at a time gets cheaper operand rounding (oracle/split_emulation.py: "x2a" = activation rounded to
binary16, "x2w" = weight rounded, "x1" = both: 2 / 2 / 1 matrix instructions per product block instead of 3) while every other layer stays
exact fp32; reported is what that does to R and t against the all-exact run, over a batch of pairs.  A layer whose rounding moves R by well
under the 1e-5 rad bar is a candidate for the cheaper form on the GPU; the decision is then gated on the GPU's parity distribution
(tools/parity_distribution.py).

    python tools/term_budget.py [--pairs 8] [--workload cfg1|n717|cfg2] [--modes x2a,x2w,x1] [--groups]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import ogmm_oracle as O  # noqa: E402
from oracle import split_emulation as E  # noqa: E402
from oracle.ref_harness import default_config  # noqa: E402
from ogmm_amd import synth  # noqa: E402
from ogmm_amd.gmmreg import state_spec  # noqa: E402

LAYERS = (["emd.conv%d" % i for i in (2, 3, 4, 5)] + ["pos.conv_dis.3", "pos.conv_ang2.0"] +
          ["%s.%s" % (t, l) for t in ("sattn1",) for l in ("attn.proj.0", "attn.proj.1", "attn.proj.2", "attn.qk", "attn.pv", "attn.merge", "mlp.0", "mlp.3")] +
          ["conv1.net.0", "conv1.net.3", "conv1.net.6"] +
          ["%s.%s" % (t, l) for t in ("cattn",) for l in ("attn.proj.0", "attn.proj.1", "attn.proj.2", "attn.qk", "attn.pv", "attn.merge", "mlp.0", "mlp.3")] +
          ["proj.net.0", "similarity", "conv2.net.0", "conv2.net.3", "conv2.net.6", "overlap.net.0", "overlap.net.3"] +
          ["%s.%s" % (t, l) for t in ("sattn2",) for l in ("attn.proj.0", "attn.proj.1", "attn.proj.2", "attn.qk", "attn.pv", "attn.merge", "mlp.0", "mlp.3")])

# groups = what a per-layer precision switch on the GPU would actually flip together
GROUPS = {
    "overlap chain (proj.0, conv2.*, overlap.*)": lambda n: n.startswith(("proj.net.0", "conv2.", "overlap.")),
    "overlap chain + similarity": lambda n: n.startswith(("proj.net.0", "conv2.", "overlap.", "similarity")),
    "sattn2 (all its GEMMs)": lambda n: n.startswith("sattn2."),
    "sattn2 mlp.0 + mlp.3": lambda n: n in ("sattn2.mlp.0", "sattn2.mlp.3"),
    "every weight GEMM but EdgeConv (the reduced mode of precision='f16')": lambda n: not n.startswith("emd.conv") or n == "emd.conv5",
    "everything": lambda n: True,
}


# candidate budgets: layer -> mode, everything else "x3" (the engine's default)
POLICIES = {
    "v1": {"conv2.net.0": "x2w", "conv2.net.3": "x2w"},
    "v2": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "sattn1.attn.proj.0": "x2w", "cattn.attn.proj.0": "x2w", "sattn2.attn.proj.0": "x2w", "similarity": "x2w"},
    "v4": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "sattn1.attn.proj.0": "x1", "cattn.attn.proj.0": "x1", "sattn2.attn.proj.0": "x1", "similarity": "x1"},
    # v4 + the K / V projections and the attention's score product q k^T with both operands rounded (round 3, late)
    "v5": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "similarity": "x1",
           **{"%s.attn.%s" % (t, l): "x1" for t in ("sattn1", "cattn", "sattn2") for l in ("proj.0", "proj.1", "proj.2", "qk")}},
    # v5 without the K / V projections, + the value product with V rounded (the default budget at the end of round 3)
    "v6": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "similarity": "x1",
           **{"%s.attn.%s" % (t, l): "x1" for t in ("sattn1", "cattn", "sattn2") for l in ("proj.0", "qk")},
           **{"%s.attn.pv" % t: "x2w" for t in ("sattn1", "cattn", "sattn2")}},
    # candidates: conv2's two wide layers with both operands rounded (v7), one of them only (v7a / v7b)
    "v7": {"conv2.net.0": "x1", "conv2.net.3": "x1", "similarity": "x1",
           **{"%s.attn.%s" % (t, l): "x1" for t in ("sattn1", "cattn", "sattn2") for l in ("proj.0", "qk")}},
    "v7a": {"conv2.net.0": "x1", "conv2.net.3": "x2w", "similarity": "x1",
            **{"%s.attn.%s" % (t, l): "x1" for t in ("sattn1", "cattn", "sattn2") for l in ("proj.0", "qk")}},
    "v7b": {"conv2.net.0": "x2w", "conv2.net.3": "x1", "similarity": "x1",
            **{"%s.attn.%s" % (t, l): "x1" for t in ("sattn1", "cattn", "sattn2") for l in ("proj.0", "qk")}},
    # the default budget of round 3 as shipped (gmmreg.TERM_BUDGET: v5 without its K / V entries)
    "r3": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "similarity": "x1",
           **{"%s.attn.%s" % (t, l): "x1" for t in ("sattn1", "cattn", "sattn2") for l in ("proj.0", "qk")}},
    # round 4 candidates, measured on BOTH weight families: r3 without the score product's entry / without any attention entry / conv2 only
    "r4": {"similarity": "x1"},          # the default budget from round 4 on: what holds on both families
    "r4p": {"similarity": "x1", "proj.net.0": "x1"},
    "r4a": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "similarity": "x1", **{"%s.attn.proj.0" % t: "x1" for t in ("sattn1", "cattn", "sattn2")}},
    "r4b": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "similarity": "x1"},
    "r4c": {"conv2.net.0": "x2w", "conv2.net.3": "x2w"},
    "r4d": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "similarity": "x2w", **{"%s.attn.proj.0" % t: "x2w" for t in ("sattn1", "cattn", "sattn2")}},
    "v3": {"conv2.net.0": "x2w", "conv2.net.3": "x2w", "conv2.net.6": "x2w", "overlap.net.0": "x2w", "overlap.net.3": "x2w", "proj.net.0": "x2w",
           "sattn1.attn.proj.0": "x2w", "cattn.attn.proj.0": "x2w", "sattn2.attn.proj.0": "x2w", "similarity": "x2w"},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", default=None, help="evaluate a whole candidate budget (POLICIES) against the all-x3 engine and the exact forward")
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "n717", "cfg2"])
    ap.add_argument("--modes", default="x2a,x2w,x1")
    ap.add_argument("--groups", action="store_true", help="layer groups instead of single layers")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--profile", default="default", choices=["default", "sharp"], help="weight family (synth.fill_state_dict)")
    ap.add_argument("--first", type=int, default=None, help="first global pair id (default: the workload's)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    N, J, first = {"cfg1": (1024, 16, 0), "n717": (717, 128, 300), "cfg2": (2048, 64, 2000)}[args.workload]
    cfg = default_config(n_clusters=J)
    sd = {k: torch.zeros(shape, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32) for k, shape in state_spec(512)}
    P = synth.fill_state_dict(sd, profile=args.profile)
    if args.first is not None:
        first = args.first
    src, tgt, _, _ = synth.make_batch(first, args.pairs, N, "partial")
    starts = synth.fps_starts_for(first, args.pairs, N)

    def run(pol):
        with torch.no_grad(), E.policy(pol):
            return O.forward(P, cfg, src, tgt, starts)

    t0 = time.time()
    base = run(lambda n: None)
    print("# workload %s, weight family %s: pairs %d ... %d, N=%d, J=%d; exact forward %.1f s" % (args.workload, args.profile, first, first + args.pairs - 1, N, J, time.time() - t0))
    x3 = run(lambda n: "x3")
    print("%-62s %-5s R max %.2e  median %.2e   t max %.2e" % ("ALL layers (the default engine's rounding)", "x3",
          O.rotation_error_rad(x3[0], base[0]).max().item(), O.rotation_error_rad(x3[0], base[0]).median().item(), O.translation_error(x3[1], base[1]).max().item()))
    if args.policy:
        for name in args.policy.split(","):
            pol = POLICIES[name]
            out = run(lambda n: pol.get(n, "x3"))
            r = O.rotation_error_rad(out[0], base[0])
            print("%-62s       R max %.2e  median %.2e   t max %.2e   overlap max %.2e" % ("budget %s: %s" % (name, ", ".join("%s=%s" % kv for kv in pol.items()))[:62], r.max().item(),
                  r.median().item(), O.translation_error(out[1], base[1]).max().item(), max((out[2] - base[2]).abs().max().item(), (out[3] - base[3]).abs().max().item())), flush=True)
            print("    per pair R:", " ".join("%.1e" % v for v in r.tolist()))
        print("    all-x3 R:  ", " ".join("%.1e" % v for v in O.rotation_error_rad(x3[0], base[0]).tolist()))
        return
    items = list(GROUPS.items()) if args.groups else [(n, (lambda m, n=n: m == n)) for n in LAYERS]
    for name, member in items:
        for mode in args.modes.split(","):
            out = run(lambda n, member=member, mode=mode: mode if member(n) else None)
            r = O.rotation_error_rad(out[0], base[0])
            print("%-62s %-5s R max %.2e  median %.2e   t max %.2e   overlap max %.2e" % (name, mode, r.max().item(), r.median().item(),
                  O.translation_error(out[1], base[1]).max().item(), max((out[2] - base[2]).abs().max().item(), (out[3] - base[3]).abs().max().item())), flush=True)


if __name__ == "__main__":
    main()
