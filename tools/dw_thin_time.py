"""Thin weight gradient (kernel T9) per shape of the training step at B = 128 (5.2 M per-edge rows): exact fp32 against the fp16x3 form (round 5).
usage (GPU box): python3 tools/dw_thin_time.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
dev = torch.device("cuda", 0)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best

print("%-28s %12s %12s %10s %10s %12s %12s" % ("dY [R, n] x X [R, k]", "fp32 us", "fp16x3 us", "GB", "GFLOP", "fp32 err", "fp16x3 err"))
for R, n, k in ((5242880, 256, 128), (5242880, 128, 64), (5242880, 64, 64), (5242880, 64, 6), (262144, 512, 2), (1310720, 64, 1)):
    dy = torch.randn(R, n, device=dev); x = torch.randn(R, k, device=dev)
    if not ops.weight_grad_thin_supported(dy, x):
        print("%-28s unsupported" % ("%d x %d, %d" % (R, n, k))); continue
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    t32 = timed(lambda: ops.weight_grad_thin(dy, x))
    t16 = timed(lambda: ops.weight_grad_thin(dy, x, split=True, overflow=ovf))
    want = (dy[:524288].double().t() @ x[:524288].double())
    e32 = float((ops.weight_grad_thin(dy[:524288], x[:524288]).double() - want).norm() / want.norm())
    e16 = float((ops.weight_grad_thin(dy[:524288], x[:524288], split=True, overflow=ovf).double() - want).norm() / want.norm())
    print("%-28s %12.1f %12.1f %10.2f %10.1f %12.1e %12.1e" % ("%d x %d, %d" % (R, n, k), t32, t16, R * (n + k) * 4 / 1e9, 2.0 * R * n * k / 1e9, e32, e16))
    del dy, x
