"""Component timing of ops.weight_grad (dW = dY^T X on the engine) against the library GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, _lib

dev = "cuda:0"
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for R, n, k in ((131072, 1024, 1024), (131072, 512, 1024), (131072, 1024, 512), (131072, 512, 512), (131072, 256, 512)):
    dy = torch.randn(R, n, device=dev); x = torch.randn(R, k, device=dev)
    ref = dy.t().double() @ x.double() if R * n * k < 2e14 else None
    got = ops.weight_grad(dy, [x])
    err = float((got.double() - ref).norm() / ref.norm())
    # components
    tiles_mn = ((n + 255) // 256) * ((k + 255) // 256)
    S = max(1, min((512 + tiles_mn - 1) // tiles_mn, (R + 255) // 256)); chunk = ((R + S - 1) // S + 63) // 64 * 64; S = (R + chunk - 1) // chunk; pitch = chunk + 64
    dyt = torch.empty((S, n, pitch), device=dev); n_pad = (k + 255) // 256 * 256
    hi = torch.empty(S * n_pad * pitch, dtype=torch.float16, device=dev); lo = torch.empty_like(hi); part = torch.empty((S, n, k), device=dev)
    split = {"W_hi": hi, "W_lo": lo, "inv_scale": 1.0, "variant": ops.PREC_F16X3_FRAG, "ldb_h": pitch, "sB": n_pad * pitch}
    t_tr = t(lambda: _lib.call("ogmm_transpose_pad", ops._p(dy), dy.stride(0), R, n, chunk, pitch, S, ops._p(dyt), None, 1, ops._stream()))
    t_pk = t(lambda: _lib.call("ogmm_pack_frag_t", ops._p(x), x.stride(0), R, k, chunk, pitch, S, n_pad, ops._p(hi), ops._p(lo), None, ops._stream()))
    t_mm = t(lambda: ops.gemm_nt(dyt, pitch, chunk, None, 0, n, k, C=part, ldc=k, batch=(S, 1), sA=(n * pitch, 0), sC=(n * k, 0), split=split))
    t_sum = t(lambda: part.sum(dim=0))
    t_all = t(lambda: ops.weight_grad(dy, [x]))
    t_lib = t(lambda: dy.t() @ x)
    fl = 2.0 * R * n * k
    print("R=%d n=%d k=%d S=%d chunk=%d err=%.1e | transpose %.3f pack %.3f gemm %.3f (%.0f TF) sum %.3f | total %.3f ms vs library %.3f ms (%.0f TF)" % (
        R, n, k, S, chunk, err, t_tr, t_pk, t_mm, fl / t_mm / 1e9, t_sum, t_all, t_lib, fl / t_lib / 1e9))
