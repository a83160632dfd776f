"""Operand-rounding emulation for the CPU oracle -- TEST INFRASTRUCTURE ONLY (same rules as oracle/ogmm_oracle.py).

The HIP path's GEMM engine multiplies fp32 operands as sums of binary16 terms on the f16 matrix cores with fp32 accumulation
(HISTORY.md section 2):   x = hi + lo,  hi = rn16(x),  lo = rn16(x - hi)   (22 significand bits)
    "x3"   a b ~ hi hi + hi lo + lo hi          the default engine (OGMM_PREC_F16X3*): fp32-class
    "x2a"  a b ~ hi (hi + lo)                   activation rounded to binary16, weight kept: 2 MFMAs per product block
    "x2w"  a b ~ (hi + lo) hi                   weight rounded to binary16, activation kept: 2 MFMAs
    "x1"   a b ~ hi hi                          both rounded (precision = "f16", the labelled reduced mode): 1 MFMA
    "bf16" both operands rounded to bfloat16    what BASELINE configs[2] literally names; shown for comparison
    "f32" / None  exact fp32 (the reference's arithmetic)
    "ulp:<seed>"  NOT an engine mode -- a conditioning probe (round 5): exact fp32 arithmetic on operands whose activation side was moved by ONE UNIT IN THE
                  LAST PLACE with a random sign per element (each element to its upper or lower fp32 neighbour, seeded per layer): the classical stochastic-arithmetic estimate (CESTAC / CADNA)
                  of how far a result is defined.  A pair whose (R, t) moves by >= 5e-6 under it is ill-conditioned whatever host evaluates the reference.
    "sum:<seed>"  NOT an engine mode -- the second conditioning probe (round 5, late): exact fp32 products, ANOTHER ORDER OF ADDITIONS.  Every contraction is
                  evaluated as four partial contractions over an interleaved split of its index (k = seed-shifted residues mod 4) whose fp32 results are added
                  in a seeded order: what any tiled or multi-threaded fp32 GEMM does differently from the next one.  One-ulp input jitter perturbs what goes
                  INTO the sums; this perturbs HOW they are summed, which is the difference between two fp32 implementations that the jitter can miss
                  (sharp configs[1] pair 298: jitter 2e-6, two summation orders of the same engine 2e-5 apart; profiles/round5_parity_extended.txt).
    "ew:<seed>"   NOT an engine mode -- the third conditioning probe (round 5, late): exact fp32 GEMMs, but the results of the reference's TRANSCENDENTAL steps -- the
                  attention softmax, the overlap block's two softmaxes, the matching softmax, the E/M's final exp -- moved by one unit in the last place (ew()
                  below, called by the oracle at those sites; the identity without a policy).  A second implementation differs from the reference there too (v_exp_f32,
                  another order in the softmax sums), which the two GEMM probes do not model.
This module restates those roundings on the CPU so that the oracle can answer, per layer, "what does rounding THIS layer's operands do to
(R, t)?" -- the measurement behind the engine's per-layer term budget (tools/term_budget.py) and behind the stated tolerance of the reduced
precision mode (tests/test_hip_forward.py::test_reduced_precision_mode_against_the_emulating_oracle).  It emulates the operand roundings, not
the engine's summation order: products are exact in fp32 either way and both accumulate in fp32, so what is left is the order of additions,
which the reference's own MKL GEMM does not pin either.

Use:   with split_emulation.policy(lambda name: "x1" if name.startswith("conv2.") else None):  O.forward(...)
Layer names are the reference's state_dict prefixes ("emd.conv2", "sattn1.mlp.0", "conv2.net.3", ...) plus the weight-free contractions
"similarity" (models/gmmreg.py:75) and "<transformer>.attn.qk" / "<transformer>.attn.pv" (models/attn.py:79-81).
"""
import contextlib
import math

import torch
import torch.nn.functional as F

MODES = ("f32", "x3", "x2a", "x2w", "x1", "bf16")
_POLICY = None          # callable(name) -> mode or None


@contextlib.contextmanager
def policy(fn):
    """Installs `fn(layer_name) -> mode` for the duration of the block (None / "f32": exact)."""
    global _POLICY
    prev, _POLICY = _POLICY, fn
    try:
        yield
    finally:
        _POLICY = prev


def mode_of(name):
    if _POLICY is None:
        return None
    m = _POLICY(name)
    if m is not None and (m.startswith("ulp:") or m.startswith("sum:")):
        return m
    if m is not None and m.startswith("ew:"):
        return None          # (the GEMMs of an "ew" evaluation are exact)
    if m is not None and m not in MODES:
        raise ValueError("unknown emulation mode %r for %s" % (m, name))
    return None if m == "f32" else m


def rn16(x):
    """round to nearest binary16 (overflow clamps to +-65504, as the engine's staging does)"""
    return x.clamp(-65504.0, 65504.0).half().float()


def split16(x, scale_pow2=False):
    """x (fp32) -> (hi, lo, inv_scale): x * 2^e = hi + lo (+ 2^-22 relative); the engine scales WEIGHTS by a per-tensor power of two so that
    max|W| 2^e is in [2^11, 2^12) and `lo` stays a normal binary16 number (ogmm_amd/ops.py split_f16); activations are split unscaled."""
    inv = 1.0
    if scale_pow2:
        amax = float(x.abs().max())
        if amax > 0 and math.isfinite(amax):
            e = max(-24, min(24, 11 - math.floor(math.log2(amax))))
            x = x * (2.0 ** e)
            inv = 2.0 ** (-e)
    hi = rn16(x)
    return hi, rn16(x - hi), inv


def one_ulp(a, sign):
    """every element of `a` moved to its NEIGHBOURING fp32 value, away from zero where sign = +1, towards zero where sign = -1 (zeros stay).
    (Round 5 built this as a * (1 + sign * 2^-24) with the factor held in fp32: 1 + 2^-24 rounds to 1, so only the sign = -1 half ever moved --
    ADVICE.md round 5; tests/test_oracle_golden.py::test_one_ulp_probe_moves_both_ways keeps it two-sided.)"""
    return torch.nextafter(a, a + sign * a)


def _jitter(a, mode, salt):
    """a moved by one unit in the last place, random sign per element; deterministic in (seed of the mode, salt = the operand's shape)"""
    g = torch.Generator().manual_seed((int(mode[4:]) * 1000003 + salt) % (2 ** 31))
    sign = torch.randint(0, 2, a.shape, generator=g, dtype=torch.int8).to(a.dtype) * 2 - 1
    return one_ulp(a, sign) if a.dtype == torch.float32 else a


def ew(x, site):
    """the oracle's hook at its softmax / exp sites: x unchanged unless an "ew:<seed>" policy is installed, then every element moved to a neighbouring fp32 value, random direction per element"""
    if _POLICY is None or x.dtype != torch.float32:
        return x
    m = _POLICY(site)
    if m is None or not m.startswith("ew:"):
        return x
    g = torch.Generator().manual_seed((int(m[3:]) * 1000003 + sum(ord(c) for c in site) * 7919 + x.numel()) % (2 ** 31))
    sign = torch.randint(0, 2, x.shape, generator=g, dtype=torch.int8).to(x.dtype) * 2 - 1
    return one_ulp(x, sign)


def _resummed(parts, mode):
    """the partial contractions added in the order the mode's seed picks (fp32, left to right)"""
    seed = int(mode[4:])
    order = [(seed + j * (1 + 2 * (seed % 2))) % len(parts) for j in range(len(parts))]          # a rotation, forwards or backwards
    y = parts[order[0]]
    for j in order[1:]:
        y = y + parts[j]
    return y


def _terms(a, w, mode, contract):
    """a: activation-side operand, w: weight-side operand, contract(a', w') -> the fp32 contraction"""
    if mode.startswith("ulp:"):
        return contract(_jitter(a, mode, a.numel() + 7 * w.numel()), w)
    if mode.startswith("sum:"):          # conv: the contraction index is dim 1 of both operands
        K = a.shape[1]
        if K < 8:
            return contract(a, w)
        return _resummed([contract(a[:, r::4].contiguous(), w[:, r::4].contiguous()) for r in range(4)], mode)
    if mode == "bf16":
        return contract(a.bfloat16().float(), w.bfloat16().float())
    ah, al, _ = split16(a)
    wh, wl, inv = split16(w, scale_pow2=True)
    y = contract(ah, wh)
    if mode in ("x3", "x2a"):
        y = y + contract(ah, wl)
    if mode in ("x3", "x2w"):
        y = y + contract(al, wh)
    return y * inv


def conv(x, w, b, mode):
    """1x1 convolution (conv1d / conv2d by the weight's rank) with the operands rounded per `mode`; bias added in fp32"""
    f = F.conv2d if w.dim() == 4 else F.conv1d
    y = _terms(x, w, mode, lambda a_, w_: f(a_, w_))
    if b is not None:
        y = y + b.view(1, -1, *([1] * (y.dim() - 2)))
    return y


def einsum(eq, a, b, mode):
    """weight-free contraction (similarity, attention scores / values): both operands are activations, `b` plays the B-operand role"""
    if mode == "bf16":
        return torch.einsum(eq, a.bfloat16().float(), b.bfloat16().float())
    if mode.startswith("ulp:"):
        return torch.einsum(eq, _jitter(a, mode, a.numel() + 7 * b.numel()), b)
    if mode.startswith("sum:"):
        ins, out = eq.split("->")
        ia, ib = ins.split(",")
        con = [c for c in ia if c in ib and c not in out]
        if len(con) != 1 or a.shape[ia.index(con[0])] < 8:
            return torch.einsum(eq, a, b)
        da, db = ia.index(con[0]), ib.index(con[0])
        parts = []
        for r in range(4):
            sa = [slice(None)] * a.dim(); sa[da] = slice(r, None, 4)
            sb = [slice(None)] * b.dim(); sb[db] = slice(r, None, 4)
            parts.append(torch.einsum(eq, a[tuple(sa)].contiguous(), b[tuple(sb)].contiguous()))
        return _resummed(parts, mode)
    ah, al, _ = split16(a)
    bh, bl, _ = split16(b)
    y = torch.einsum(eq, ah, bh)
    if mode in ("x3", "x2a"):
        y = y + torch.einsum(eq, ah, bl)
    if mode in ("x3", "x2w"):
        y = y + torch.einsum(eq, al, bh)
    return y
