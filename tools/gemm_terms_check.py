"""The two-term form of the LDS-DMA GEMM engine (struct ogmm_gemm.terms = 2: the weight rounded to binary16, (a_hi + a_lo) w_hi) on the GPU box:
results against fp64 products with the rounded weight, then interleaved timings against the three-term form on the forward's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops

torch.manual_seed(0)
dev = "cuda"


def run(A, W, sp, out, terms, A2=None, variant=None, **kw):
    m, k1 = A.shape
    n = W.shape[0]
    k2 = A2.shape[1] if A2 is not None else 0
    sp = dict(sp)
    if variant is not None:
        sp["variant"] = variant
    ops.gemm_nt(A, A.stride(0), k1, W, k1 + k2, m, n, C=out, ldc=out.stride(0), A2=A2, lda2=(A2.stride(0) if A2 is not None else 0), K2=k2, split=sp, terms=terms, **kw)


for (m, n, k1, k2, has_res, act) in [(65536, 512, 512, 0, True, 1), (70000 // 256 * 256, 1024, 1024, 0, False, 2), (65536, 1024, 512, 32, False, 1), (131072, 512, 96, 0, False, 0),
                                     (131072, 256, 1024, 0, False, 1), (65536, 256, 512, 0, True, 2)]:          # (N = 256: the 8-wave engine)
    A = torch.randn(m, k1, device=dev) * (0.5 + 3 * torch.rand(m, 1, device=dev))          # (O(1) rows: below ~0.1 the activation's lo term is a binary16 subnormal and the split keeps absolute, not relative, precision)
    A2 = torch.randn(m, k2, device=dev) if k2 else None
    W = torch.randn(n, k1 + k2, device=dev) * 0.05
    W[1::7] *= 17.0
    sp = ops.split_f16(W, frag=True, k1=(k1 if k2 else None))
    res = torch.randn(m, n, device=dev) if has_res else None
    scale = (torch.rand(n, device=dev) + 0.5) if act else None
    shift = torch.randn(n, device=dev) if act else None
    kw = dict(res=res, ldr=(n if has_res else 0), scale=scale, shift=shift, act=act)
    out3 = torch.full((m, n), float("nan"), device=dev)
    out2 = torch.full((m, n), float("nan"), device=dev)
    out1 = torch.full((m, n), float("nan"), device=dev)
    run(A, W, sp, out3, 0, A2=A2, **kw)
    run(A, W, sp, out2, 2, A2=A2, **kw)
    run(A, W, sp, out1, 1, A2=A2, **kw)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.randint(0, m, (700,), device=dev), torch.arange(m - 300, m, device=dev)])
    Af = (A[rows] if A2 is None else torch.cat([A[rows], A2[rows]], 1)).double()
    e = 11 - torch.floor(torch.log2(W.abs().max())).item()
    Wh = (W * 2.0 ** e).half().double() * 2.0 ** (-e)          # the weight's leading binary16 term (the image's power-of-two scale: ops.split_f16)

    def finish(ref):
        if act:
            ref = ref * scale.double() + shift.double()
            ref = torch.relu(ref) if act == 1 else torch.where(ref > 0, ref, 0.2 * ref)
        return ref + res[rows].double() if has_res else ref
    ref2, ref3 = finish(Af @ Wh.t()), finish(Af @ W.double().t())
    mag = (Af.abs() @ W.double().abs().t()).clamp_min(1e-30) * (scale.double() if act else 1.0) + ref3.abs()          # (the stored fp32 value's own rounding)
    e2 = ((out2[rows].double() - ref2).abs() / mag).max().item()
    e3 = ((out3[rows].double() - ref3).abs() / mag).max().item()
    e23 = ((out2[rows].double() - ref3).abs() / mag).max().item()
    print("M=%6d N=%4d K=%4d+%2d res=%d act=%d: two-term vs fp64 with rounded weight %.2e | three-term vs fp64 %.2e | two-term vs exact weight %.2e (expected ~2^-12 = 2.4e-4 worst case) nan %d"
          % (m, n, k1, k2, has_res, act, e2, e3, e23, torch.isnan(out2).sum().item()))
    assert e2 < 2e-6 and not torch.isnan(out2).any()
    if n >= 512:          # one term: both operands rounded (the 4-wave engine only; N = 256 runs all three)
        ref1 = finish(Af.half().double() @ Wh.t())
        e1 = ((out1[rows].double() - ref1).abs() / mag).max().item()
        print("        one-term vs fp64 with both operands rounded %.2e | vs exact operands %.2e" % (e1, ((out1[rows].double() - ref3).abs() / mag).max().item()))
        assert e1 < 2e-6 and not torch.isnan(out1).any()
    else:
        assert torch.equal(out1, out3)

print("timing (interleaved, ms): three terms / two terms / one term")
for (m, n, k) in [(131072, 1024, 1024), (131072, 1024, 512), (131072, 512, 1024), (131072, 512, 512), (131072, 256, 1024), (131072, 256, 512), (131072, 256, 256)]:
    A = torch.randn(m, k, device=dev)
    W = torch.randn(n, k, device=dev) * 0.05
    sp = ops.split_f16(W, frag=True)
    out = torch.empty((m, n), device=dev)
    ts = {0: [], 2: [], 1: []}
    for rep in range(6):
        for terms in (0, 2, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(A, W, sp, out, terms)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts[terms].append(e0.elapsed_time(e1) / 5)
    t3, t2, t1 = sorted(ts[0])[len(ts[0]) // 2], sorted(ts[2])[len(ts[2]) // 2], sorted(ts[1])[len(ts[1]) // 2]
    print("  %6d x %4d x %4d: %.3f / %.3f / %.3f ms  (x%.2f, x%.2f)   %.0f / %.0f / %.0f TF-alg" % (m, n, k, t3, t2, t1, t2 / t3, t1 / t3, 2.0 * m * n * k / t3 / 1e9, 2.0 * m * n * k / t2 / 1e9, 2.0 * m * n * k / t1 / 1e9))
