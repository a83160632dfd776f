"""LDS-DMA GEMM engine (gemm_f16x3_v6.hip) on the GPU box: results against an fp64 product and against the register-staged engine (v4),
then interleaved timings of the forward's shapes.  usage: gemm_v6_check.py [--time-only] [variant codes to time ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
time_only = "--time-only" in sys.argv
variants = [int(v) for v in args] or [23, 60]
torch.manual_seed(0)
dev = "cuda"


import probe          # tools/probe.py: every GEMM descriptor of this tool goes to libogmm_probe.so (the engines' build WITH ablation / clock-probe codes)
probe.install()


def run(v, A, k1, W, m, n, out, A2=None, k2=0, split=None, **kw):
    split = dict(split); split["variant"] = v
    ops.gemm_nt(A, A.stride(0), k1, W, k1 + k2, m, n, C=out, ldc=out.stride(0), A2=A2, lda2=(A2.stride(0) if A2 is not None else 0), K2=k2, split=split, **kw)


if not time_only:
    # (M, N, K1, K2, residual, scale/shift + act, stats)
    cases = [(65536, 256, 512, 0, False, 0, False), (65536, 512, 512, 0, True, 1, False), (70000, 512, 1024, 0, False, 2, False), (65536, 1024, 512, 512, False, 1, False),
             (65536 + 100, 1024, 512, 32, True, 0, False), (131072, 1024, 1024, 0, False, 1, True), (65536, 256, 64, 0, False, 1, False), (65536, 768, 512, 0, False, 0, False), (65536, 256, 32, 0, False, 0, False), (65536, 512, 96, 0, True, 1, False), (66000, 256, 64, 32, False, 2, False)]
    worst = 0.0
    for (m, n, k1, k2, has_res, act, stats) in cases:
        A = torch.randn(m, k1, device=dev) * torch.rand(m, 1, device=dev) * 3
        A = torch.relu(A) + 0.01 * torch.randn_like(A)
        A2 = torch.randn(m, k2, device=dev) if k2 else None
        W = torch.randn(n, k1 + k2, device=dev) * 0.05
        W[1::7] *= 17.0
        sp = ops.split_f16(W, frag=True, k1=(k1 if k2 else None))
        res = torch.randn(m, n, device=dev) if has_res else None
        scale = (torch.rand(n, device=dev) + 0.5) if act else None
        shift = torch.randn(n, device=dev) if act else None
        kw = dict(res=res, ldr=(n if has_res else 0), scale=scale, shift=shift, act=act)
        outs = {}
        for v in (23, 60, 100, 110):
            out = torch.full((m, n), float("nan"), device=dev)
            st = torch.zeros(((m + 1023) // 1024, n, 2), dtype=torch.float64, device=dev) if stats else None
            run(v, A, k1, W, m, n, out, A2=A2, k2=k2, split=sp, col_stats=st, group_rows=(1024 if stats else 0), **kw)
            torch.cuda.synchronize()
            outs[v] = (out, st)
        # fp64 reference on a row sample (the full product in fp64 is slow)
        rows = torch.cat([torch.arange(0, 300, device=dev), torch.randint(0, m, (700,), device=dev), torch.arange(m - 300, m, device=dev)])
        Af = A[rows].double() if A2 is None else torch.cat([A[rows], A2[rows]], 1).double()
        ref = Af @ W.double().t()
        if act:
            ref = ref * scale.double() + shift.double()
            ref = torch.relu(ref) if act == 1 else torch.where(ref > 0, ref, 0.2 * ref)
        if has_res:
            ref = ref + res[rows].double()
        mag = (Af.abs() @ W.double().abs().t()).clamp_min(1e-30) * (scale.double() if act else 1.0)
        e6 = ((outs[60][0][rows].double() - ref).abs() / mag).max().item()
        e4 = ((outs[23][0][rows].double() - ref).abs() / mag).max().item()
        same = torch.equal(outs[60][0], outs[23][0])
        dmax = (outs[60][0] - outs[23][0]).abs().max().item()
        nan6 = torch.isnan(outs[60][0]).sum().item()
        sdiff = 0.0
        if stats:
            sdiff = ((outs[60][1] - outs[23][1]).abs() / outs[23][1].abs().clamp_min(1e-9)).max().item()
        print("M=%6d N=%4d K=%4d+%3d res=%d act=%d stats=%d : v6 err %.2e  v4 err %.2e (relative to sum|a||w|)  v6==v4 bitwise %s (max diff %.2e)  nan %d  stats rel diff %.1e" %
              (m, n, k1, k2, has_res, act, stats, e6, e4, same, dmax, nan6, sdiff))
        same8 = torch.equal(outs[100][0], outs[23][0])
        print("        v8==v4 bitwise %s (max diff %.2e, nan %d)%s" % (same8, (outs[100][0] - outs[23][0]).abs().max().item(), torch.isnan(outs[100][0]).sum().item(),
              ("  stats rel diff %.1e" % ((outs[100][1] - outs[23][1]).abs() / outs[23][1].abs().clamp_min(1e-9)).max().item()) if stats else ""))
        assert same8, "v8 differs from v4"
        same10 = torch.equal(outs[110][0], outs[23][0])
        print("        v10==v4 bitwise %s (max diff %.2e, nan %d)%s" % (same10, (outs[110][0] - outs[23][0]).abs().max().item(), torch.isnan(outs[110][0]).sum().item(),
              ("  stats rel diff %.1e" % ((outs[110][1] - outs[23][1]).abs() / outs[23][1].abs().clamp_min(1e-9)).max().item()) if stats else ""))
        assert same10, "v10 differs from v4"
        if stats:
            assert ((outs[110][1] - outs[23][1]).abs() / outs[23][1].abs().clamp_min(1e-9)).max().item() < 1e-5, "v10 column statistics differ from v4"
        worst = max(worst, e6)
        assert nan6 == 0 and (same or e6 < max(2e-6, 1.2 * e4)), "v6 result off"
    # fused InstanceNorm on the A side (a_scale / a_shift per (row group, k)): v8 against v4
    for (m, n, k1, k2, relu) in [(65536, 512, 1024, 0, True), (66560, 256, 512, 512, False)]:
        A = torch.randn(m, k1, device=dev); A2 = torch.randn(m, k2, device=dev) if k2 else None
        W = torch.randn(n, k1 + k2, device=dev) * 0.05
        sp = ops.split_f16(W, frag=True, k1=(k1 if k2 else None))
        G = m // 1024
        asc = torch.rand(G, k1 + k2, device=dev) + 0.5; ash = torch.randn(G, k1 + k2, device=dev)
        res = torch.randn(m, n, device=dev)
        outs = {}
        for v in (23, 100, 110):
            out = torch.full((m, n), float("nan"), device=dev)
            run(v, A, k1, W, m, n, out, A2=A2, k2=k2, split=sp, a_affine=(asc, ash, relu), group_rows=1024, res=res, ldr=n)
            torch.cuda.synchronize(); outs[v] = out
        assert torch.equal(outs[23], outs[110]), "v10 AFF differs from v4 (max diff %.2e)" % (outs[23] - outs[110]).abs().max().item()
        same = torch.equal(outs[23], outs[100])
        print("A-side InstanceNorm M=%d N=%d K=%d+%d relu=%d: v8==v4 bitwise %s (max diff %.2e, nan %d)" % (m, n, k1, k2, relu, same, (outs[23] - outs[100]).abs().max().item(), torch.isnan(outs[100]).sum().item()))
        assert same, "v8 AFF differs from v4"
    print("v6 correctness OK, worst relative error %.2e" % worst)

if "--grid-sweep" in sys.argv:          # the same tile work on 1/4, 1/2 and all of the chip's CUs, one tile per workgroup: is the operand path a per-CU or a chip-wide limit?
    for m in (4096, 8192, 16384, 32768, 131072):
        n, k1 = 1024, 1024
        A = torch.relu(torch.randn(m, k1, device=dev)); W = torch.randn(n, k1, device=dev) * 0.03
        out = torch.empty(m, n, device=dev); sp = ops.split_f16(W, frag=True)
        row = "M=%6d (%4d tiles)" % (m, m // 256 * 4)
        for v in variants:
            sp2 = dict(sp); sp2["variant"] = v
            import ctypes
            os.environ["OGMM_V4_MIN_TILES"] = "1"
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run(v, A, k1, W, m, n, out, split=sp)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            rounds = max(1, (m // 256 * 4 + 255) // 256)
            row += "  v%-2d %7.1f us (%5.1f us per round of tiles)" % (v, best * 1e3, best * 1e3 / rounds)
        print(row, flush=True)
    sys.exit(0)

if "--clock" in sys.argv:          # in-kernel clock probes (variants 80..86): the shader clock each ablation actually ran at, outside the profiler
    import ctypes
    from ogmm_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    buf = (ctypes.c_ulonglong * 3)()
    m, n, k1 = 131072, 1024, 1024
    A = torch.relu(torch.randn(m, k1, device=dev)); W = torch.randn(n, k1, device=dev) * 0.03
    out = torch.empty(m, n, device=dev); sp = ops.split_f16(W, frag=True)
    for rnd in range(2):
        for v in variants:
            for _ in range(3): run(v, A, k1, W, m, n, out, split=sp)
            probe = lambda b_, v=v: __import__('probe').clock_probe(v, b_)  # noqa: E731
            torch.cuda.synchronize(); probe(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(v, A, k1, W, m, n, out, split=sp)
            e1.record(); torch.cuda.synchronize(); probe(buf)
            ms = e0.elapsed_time(e1) / 10
            cyc, wall, wg = buf[0], buf[1], buf[2]
            print("v%-3d %.3f ms/launch  %5.1f TF-alg   per workgroup: %8.0f shader cycles, %6.2f us -> shader clock %.3f GHz   (MFMA pipe needs 98304 cycles per tile: busy %.1f %%)" %
                  (v, ms, 2.0 * m * n * k1 / ms / 1e9, cyc / max(wg, 1), wall / max(wg, 1) / 100.0, cyc / max(wall, 1) * 0.1, 98304.0 / max(cyc / max(wg, 1), 1.0) * 100), flush=True)
    sys.exit(0)

M = 131072
shapes = [("mlp0 1024x(512+512)", M, 1024, 512, 512), ("conv.3 1024x1024", M, 1024, 1024, 0), ("mlp3 512x1024", M, 512, 1024, 0), ("conv.0 1024x512", M, 1024, 512, 0),
          ("q/merge 512x512", M, 512, 512, 0), ("proj 256x512", M, 256, 512, 0)]
for name, m, n, k1, k2 in shapes:
    A = torch.relu(torch.randn(m, k1, device=dev))
    A2 = torch.randn(m, k2, device=dev) if k2 else None
    W = torch.randn(n, k1 + k2, device=dev) * 0.03
    out = torch.empty(m, n, device=dev)
    sp = ops.split_f16(W, frag=True, k1=(k1 if k2 else None))
    best = {v: 1e9 for v in variants}
    for v in variants:
        run(v, A, k1, W, m, n, out, A2=A2, k2=k2, split=sp)
    torch.cuda.synchronize()
    for rnd in range(5):          # interleaved rounds: A/B inside one process
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(v, A, k1, W, m, n, out, A2=A2, k2=k2, split=sp)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 5)
    row = "%-22s" % name
    for v in variants:
        row += "  v%-2d %6.1f TF (%6.3f ms)" % (v, 2.0 * m * n * (k1 + k2) / best[v] / 1e9, best[v])
    print(row, flush=True)
# the InstanceNorm-fusing consumer (mlp3: 512 x 1024 with the A transform)
m, n, k1 = M, 512, 1024
A = torch.randn(m, k1, device=dev); W = torch.randn(n, k1, device=dev) * 0.03; out = torch.empty(m, n, device=dev); sp = ops.split_f16(W, frag=True)
asc = torch.rand(m // 1024, k1, device=dev) + 0.5; ash = torch.randn(m // 1024, k1, device=dev)
res = torch.randn(m, n, device=dev)
for label, kw in (("mlp3 + A transform", {}), ("mlp3 + A tr. + residual", dict(res=res, ldr=n))):
    row = "%-22s" % label
    for v in [x for x in variants if x in (23, 100, 110)]:
        best_v = 1e9
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(v, A, k1, W, m, n, out, split=sp, a_affine=(asc, ash, True), group_rows=1024, **kw)
            e1.record(); torch.cuda.synchronize()
            best_v = min(best_v, e0.elapsed_time(e1) / 5)
        row += "  v%-2d %6.1f TF (%6.3f ms)" % (v, 2.0 * m * n * k1 / best_v / 1e9, best_v)
    print(row, flush=True)
