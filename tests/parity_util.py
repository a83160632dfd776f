"""Shared helpers of the parity-distribution tests (tests/test_hip_parity_tail.py, tests/test_hip_forward.py) and of tools/parity_distribution.py:
every pair of a batch against the CPU oracle, and -- for the pairs in the tail -- how well the REFERENCE's own fp32 result is defined.

Why the second part exists.  The E/M + soft matching + weighted SVD head (lib/utils.py:269-291, models/dgcnn.py:96-115, lib/se3.py:256-289) is
ill-conditioned on some pairs: the reference's own (R, t) then moves by 1e-5 ... 5e-5 rad when nothing but its summation order changes (1 host thread
against several: MKL blocks the GEMMs differently) or when the same algorithm is evaluated in fp64.  "Within 1e-5 of the reference" is not a defined
quantity on such a pair -- any second fp32 implementation, the reference at another thread count included, lands that far away.  The tests therefore
state the bar as: within 1e-5 on every pair, EXCEPT pairs that are demonstrably ill-conditioned in the reference itself, where this path must stay
within a small multiple of the reference's own spread."""
import itertools
import os

import torch

from oracle import ogmm_oracle as O
from oracle import split_emulation as E
from ogmm_amd import synth

SPREAD_THREADS = (1, 4, 16)          # MKL's 1-thread GEMM sums in another order than its threaded one; threaded runs differ among themselves by less
# (Round 6: the one-ulp probes of round 5 were ONE-SIDED -- 1 + 2^-24 is 1 in fp32, so only half of the elements moved, all downwards;
# oracle/split_emulation.one_ulp now goes to the neighbouring fp32 value in either direction.  The probe set itself is unchanged.)
# Round 5: the thread probes are a property of the HOST (sharp configs[1] pair 75: 4e-6 between 1 / 4 / 16 threads on the GPU box's CPU, 1.6e-5 between 1 / 8
# threads on the build container's), so a pair could pass or fail the "ill-conditioned" test by where it ran.  Added: three evaluations of the reference's
# algorithm in its own fp32 arithmetic with every GEMM's activation operand moved by ONE UNIT IN THE LAST PLACE, random sign per element
# (oracle/split_emulation.py "ulp:<seed>": the stochastic-arithmetic conditioning estimate of CESTAC / CADNA) -- deterministic and host-independent.
# Ordinary pairs move by 0.3e-6 ... 1.7e-6 under it on both weight families; pair 75 by 1.0e-5, pair 84 by 2.7e-5, pair 112 by 1.9e-4.
JITTER_SEEDS = (1, 2, 3)
# Round 5, late: three more host-independent probes, the reference's arithmetic with ANOTHER ORDER OF ADDITIONS in every contraction (oracle/split_emulation.py
# "sum:<seed>": four interleaved partial contractions added in a seeded order).  The one-ulp jitter perturbs what goes into the sums, not how they are summed,
# and on some pairs only the latter matters: sharp configs[1] pair 298 moves by 2e-6 under the jitter and by 2e-5 between two summation orders of one engine
# (profiles/round5_parity_extended.txt).  Ten evaluations per tail pair now: 1 / 4 / 16 threads, fp64, three jitters, three summation orders.
SUM_SEEDS = (1, 2, 3)
# ... and two with the reference's softmax / exp results moved by one ulp ("ew:<seed>"): what a second implementation's exponentials and softmax sums do.
EW_SEEDS = (1, 2)
TAIL_FACTOR = 2.0                    # a tail pair's HIP distance may be at most this multiple of the reference's own spread on that pair (round 4: 4, round 5: 3; largest measured in round 6: 1.50)
ILL_CONDITIONED = 5e-6               # ... and the pair must be visibly ill-conditioned: ordinary pairs spread by 0.3e-6 ... 3e-6


def distribution(model, P, cfg, first, B, N, kind, threads=16, chunk=64, label=None):
    """HIP forward over pairs [first, first + B) in chunks of `chunk` pairs, each pair against the oracle (run 8 pairs at a time on the host):
    per-pair R [rad], t, overlap-score errors + the inputs (for the tail probes)."""
    src, tgt, _, _ = synth.make_batch(first, B, N, kind)
    starts = synth.fps_starts_for(first, B, N)
    got = [[], [], [], []]
    with torch.no_grad():
        for a in range(0, B, chunk):
            e = min(B, a + chunk)
            out = model(src[a:e].cuda(), tgt[a:e].cuda(), fps_starts=starts[:, a:e])
            for lst, x in zip(got, out[:4]):
                lst.append(x.cpu())
    got = [torch.cat(g) for g in got]
    old = torch.get_num_threads()
    torch.set_num_threads(min(threads, os.cpu_count() or threads))
    r, t, o = [], [], []
    try:
        for a in range(0, B, 8):
            e = min(B, a + 8)
            with torch.no_grad():
                ref = O.forward(P, cfg, src[a:e], tgt[a:e], starts[:, a:e])
            r.append(O.rotation_error_rad(got[0][a:e], ref[0]))
            t.append(O.translation_error(got[1][a:e], ref[1]))
            o.append(torch.maximum((got[2][a:e] - ref[2]).abs().amax(1), (got[3][a:e] - ref[3]).abs().amax(1)))
    finally:
        torch.set_num_threads(old)
    r, t, o = torch.cat(r), torch.cat(t), torch.cat(o)
    if label:
        edges = [0.0, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1.0]
        hist = " ".join("<%.0e:%d" % (hi, int(((r >= lo) & (r < hi)).sum())) for lo, hi in zip(edges[:-1], edges[1:]))
        print("PARITY-DISTRIBUTION %s: %d pairs  R max %.2e median %.2e [%s]  t max %.2e  overlap max %.2e" % (label, B, r.max(), r.median(), hist, t.max(), o.max()))
    return r, t, o, (src, tgt, starts)


def reference_spread(P, cfg, src1, tgt1, starts1, threads=SPREAD_THREADS):
    """How far the reference's own (R, t) of ONE pair is defined: the largest distance between evaluations of the same algorithm on the same inputs that
    differ only in summation order or working precision -- the fp32 oracle (bit-identical to the reference on its fixtures) at several host thread
    counts, and an fp64 evaluation on the same kNN graph.  Returns (spread_R [rad], spread_t, {probe: R distance to the first probe})."""
    old = torch.get_num_threads()
    outs = {}
    try:
        cap = {}
        for nt in threads:
            torch.set_num_threads(max(1, min(nt, os.cpu_count() or nt)))
            with torch.no_grad():
                cap = {}
                outs["t%d" % nt] = O.forward(P, cfg, src1, tgt1, starts1, cap=cap)[:2]
        P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
        inj = {k: cap[k] for k in ("knn_idx_src", "knn_idx_tgt")}
        with torch.no_grad():
            outs["f64"] = O.forward(P64, cfg, src1.double(), tgt1.double(), starts1, inject=inj)[:2]
            for seed in JITTER_SEEDS:
                with E.policy(lambda name, seed=seed: "ulp:%d" % seed):
                    outs["ulp%d" % seed] = O.forward(P, cfg, src1, tgt1, starts1)[:2]
            for seed in SUM_SEEDS:
                with E.policy(lambda name, seed=seed: "sum:%d" % seed):
                    outs["sum%d" % seed] = O.forward(P, cfg, src1, tgt1, starts1)[:2]
            for seed in EW_SEEDS:
                with E.policy(lambda name, seed=seed: "ew:%d" % seed):
                    outs["ew%d" % seed] = O.forward(P, cfg, src1, tgt1, starts1)[:2]
    finally:
        torch.set_num_threads(old)
    dr = {(a, b): O.rotation_error_rad(outs[a][0].double(), outs[b][0].double()).max().item() for a, b in itertools.combinations(outs, 2)}
    dt = {(a, b): O.translation_error(outs[a][1].double(), outs[b][1].double()).max().item() for a, b in itertools.combinations(outs, 2)}
    first = next(iter(outs))
    return max(dr.values()), max(dt.values()), {k[1]: v for k, v in dr.items() if k[0] == first}


def check_tail(label, r, t, inputs, P, cfg, first, min_within, bar=1e-5):
    """The parity statement as an assertion.  (i) at least `min_within` of the pairs are within `bar` in R and t; (ii) EVERY pair beyond it is one on which
    the reference itself is ill-conditioned -- its own spread (reference_spread) is >= ILL_CONDITIONED -- and this path's distance is within TAIL_FACTOR of
    that spread.  Returns the tail table for printing.  (Round 5 had a strict=False escape for one window; round 6 removed it with the cause: module docstring of
    tests/test_hip_parity_tail.py.)"""
    src, tgt, starts = inputs
    n = r.numel()
    bad = [int(i) for i in torch.nonzero((r >= bar) | (t >= bar)).flatten()]
    rows = []
    for i in bad:
        sr, st, probes = reference_spread(P, cfg, src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1])
        rows.append((first + i, r[i].item(), t[i].item(), sr, st, probes))
        print("PARITY-TAIL %s pair %d: HIP R %.2e t %.2e | reference's own spread R %.2e t %.2e, HIP / spread %.2f (%s)" % (
            label, first + i, r[i].item(), t[i].item(), sr, st, r[i].item() / max(sr, 1e-12), " ".join("%s %.1e" % kv for kv in probes.items())))
    within = n - len(bad)
    print("PARITY-TAIL %s: %d of %d pairs within %.0e; %d beyond, all characterised" % (label, within, n, bar, len(bad)))
    assert within >= min_within, "%s: only %d of %d pairs within %.0e (stated floor: %d)" % (label, within, n, bar, min_within)
    for pid, ri, ti, sr, st, _ in rows:
        assert sr >= ILL_CONDITIONED, "%s pair %d is %.2e from the reference although the reference is well defined there (spread %.2e)" % (label, pid, ri, sr)
        assert ri <= TAIL_FACTOR * sr and ti <= TAIL_FACTOR * max(st, sr), "%s pair %d: %.2e / %.2e, beyond %g x the reference's own spread %.2e / %.2e" % (label, pid, ri, ti, TAIL_FACTOR, sr, st)
    return rows
