"""The per-edge dense layers of the training step (5.2 M rows at 128 pairs; K, N <= 128) on the GEMM engine: time and HBM rate per shape, forward (Y = X W^T) and dX (dY W).
usage (GPU box): python3 tools/thin_linear_time.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, train_ops
dev = torch.device("cuda", 0)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
R = 5242880
print("%-30s %10s %8s %8s" % ("layer", "us", "GB", "TB/s"))
for k, n in ((64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (256, 128)):
    x = torch.randn(R, k, device=dev); W = torch.randn(n, k, device=dev) * 0.1
    layer = {"W": W.contiguous()}
    layer["split"] = ops.split_f16_training(layer["W"], n, frag=True, k1=k)
    layer["scale"] = layer["split"]["col_scale"]
    t = timed(lambda: ops.conv1x1(x, layer, ops.ACT_NONE, split=True, overflow=ovf))
    gb = R * (k + n) * 4 / 1e9
    print("%-30s %10.1f %8.2f %8.2f" % ("[%d x %d] -> %d" % (R, k, n), t, gb, gb / t * 1e3))
    del x
