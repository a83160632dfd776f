"""Per-shape time and HBM rate of the normalisation backward kernels of a training step (B = 128, N = 1024, k = 20): reduction and apply, dense and
pool-routed upstream gradients.  Algorithmic bytes: reduce reads x + dy (8 B / element; routed: x + arg / k + dpool / k), apply reads the same and writes dx.
usage (GPU box): python3 tools/norm_bwd_time.py"""
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

print("%-34s %10s %10s %10s" % ("map", "us", "GB", "TB/s"))
for rows, cols, group, k in ((262144, 512, 1024, 0), (262144, 512, 131072, 0), (262144, 1024, 1024, 0), (262144, 256, 131072, 0), (5242880, 64, 2621440, 20), (5242880, 128, 2621440, 20), (5242880, 256, 2621440, 20), (5242880, 64, 2621440, 0)):
    x = torch.randn(rows, cols, device=dev)
    G = rows // group
    scale, shift, mean, rstd = (torch.rand(G, cols, device=dev) + 0.5 for _ in range(4))
    if k:
        P = rows // k
        dpool = torch.randn(P, cols, device=dev); arg = torch.randint(0, k, (P, cols), device=dev, dtype=torch.uint8); dy = None
        gb_in = (rows * cols * 4 + P * cols * 5) / 1e9
    else:
        dy = torch.randn(rows, cols, device=dev); dpool = arg = None
        gb_in = rows * cols * 8 / 1e9
    t = timed(lambda: ops.norm_bwd(x, dy, group, scale, shift, mean, rstd, ops.ACT_RELU, dpool=dpool, arg=arg, k=k))
    gb = 2 * gb_in + rows * cols * 4 / 1e9
    print("%-34s %10.1f %10.2f %10.2f   (reduce + apply)" % ("%d x %d g=%d k=%d" % (rows, cols, group, k), t, gb, gb / t * 1e3))
    if not k:
        sums = torch.zeros(G, cols, 2, dtype=torch.float64, device=dev)
        t = timed(lambda: ops.norm_bwd_apply(x, dy, group, scale, shift, mean, rstd, sums))
        gb = rows * cols * 12 / 1e9
        print("%-34s %10.1f %10.2f %10.2f   (apply alone)" % ("", t, gb, gb / t * 1e3))
    del x, dy, dpool, arg
