"""One eval forward (B = 64, N = 1024, J = 16) with every GEMM launch timed: shape, duration, algorithmic TFLOP/s -- to find the launches the engines
serve badly -- and (round 5) the per-tile arithmetic that explains them: cycles the matrix pipe needs for a 256 x 256 tile (terms x K x 32: one
v_mfma_f32_32x32x16_f16 occupies a SIMD for 32 cycles, 16 of them per k16 block and term), cycles the CU's vector-memory path needs to take the tile's operands
in (64 KiB per K step of 32 with three terms, 48 KiB with fewer -- the lo plane of the weights is not fetched --, at the measured ~32 B/clk of
global_load_lds) and to put its output out (256 KiB at the measured ~16 B/clk of global_store, + 256 KiB of residual in at 32 B/clk), and a model of the
round time: t = T0 + (mfma + out) / f  with the stores NOT overlapped (the wave stores from the accumulators it would need for the next tile), f = the shader
clock the part holds under this load (1.60 GHz, profiles/round4_gemm_engine.txt clock probes) and T0 = 7 us (workgroup dispatch + first DMA round trip +
accumulator clear).  `in / mfma` close to or above 1 means the operand stream alone keeps the memory path as busy as the matrix pipe.
usage (GPU box): python3 tools/gemm_launch_table.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg

dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG)
synth.fill_state_dict(model.state_dict())
model = model.to(dev).eval()
src, tgt, _, _ = synth.make_batch(0, 64, 1024, "partial")
starts = synth.fps_starts_for(0, 64, 1024)
src, tgt = src.to(dev), tgt.to(dev)

shapes = []
real = ops.gemm_nt
def spy(A, lda, K1, B, ldb, M, N, *a, **kw):
    shapes.append((M, N, K1, kw.get("K2", 0), kw.get("batch", (1, 1)), kw.get("res") is not None, kw.get("col_stats") is not None, kw.get("a_affine") is not None, kw.get("pool_k", 0)))
    return real(A, lda, K1, B, ldb, M, N, *a, **kw)

with torch.no_grad():
    for _ in range(3):
        model(src, tgt, fps_starts=starts)
    ops.gemm_nt = spy
    for mod in list(sys.modules.values()):          # modules that did `from .ops import gemm_nt`
        if getattr(mod, "__name__", "").startswith("ogmm_amd") and getattr(mod, "gemm_nt", None) is real:
            mod.gemm_nt = spy
    best = None
    for rep in range(5):
        shapes.clear()
        ops.GEMM_TIMELINE, ops.GEMM_TIMELINE_ONLY = [], None
        model(src, tgt, fps_starts=starts)
        torch.cuda.synchronize()
        tl = [(e0.elapsed_time(e1) * 1e3, f) for e0, e1, f, *_ in ops.GEMM_TIMELINE]
        ops.GEMM_TIMELINE = None
        best = tl if best is None else [(min(a[0], b[0]), a[1]) for a, b in zip(best, tl)]
tot = 0.0
F_GHZ, T0_US, CUS = 1.60, 7.0, 256
print("%8s %5s %5s %4s %8s  res stat aff pool   %8s %7s | %6s %9s %8s %8s %8s %7s | %9s %9s" % (
    "M", "N", "K1", "K2", "batch", "us", "TF-alg", "rounds", "us/round", "mfma kc", "in kc", "out kc", "in/mfma", "model us", "meas/mod"))
for (m, n, k1, k2, bt, r, st, af, pk), (us, f) in zip(shapes, best):
    tot += us
    K = k1 + k2
    tiles = ((m + 255) // 256) * ((n + 255) // 256) * bt[0] * bt[1]
    rounds = (tiles + CUS - 1) // CUS
    flop_alg = 2.0 * m * n * K * bt[0] * bt[1]
    terms = 1 if (bt[0] > 1 and n == m) else 3          # the batched N x N similarity is the one single-term layer of the default budget
    mfma = terms * K * 32
    vin = (K / 32.0) * ((32768 + (32768 if terms == 3 else 16384)) / 32.0)
    vout = 262144 / 16.0 + (262144 / 32.0 if r else 0.0)
    if bt[0] > 1 and n == m:
        vout = 2 * 256 * 12 / 16.0          # the similarity's epilogue writes two partial softmax-dot triples per row / column, not the tile
    model = T0_US + (mfma + vout) / (F_GHZ * 1e3)
    print("%8d %5d %5d %4d %8s   %d    %d   %d  %3d   %8.1f %7.1f | %6d %9.1f %8.1f %8.1f %8.1f %7.2f | %9.1f %9.2f" % (
        m, n, k1, k2, "%dx%d" % bt, r, st, af, pk, us, flop_alg / us / 1e6, rounds, us / rounds, mfma / 1e3, vin / 1e3, vout / 1e3, vin / mfma, model * rounds, us / (model * rounds)))
print("GEMM launches: %d, total %.1f us (each launch timed alone between two events: includes launch gaps)" % (len(best), tot))
