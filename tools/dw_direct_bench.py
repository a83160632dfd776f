"""A/B of the weight gradient's two forms (round 4): dY read as it lies by the engine's transposing fragment reads (struct ogmm_gemm.a_trans) against the
materialised dY^T (ogmm_transpose_pad) -- time per call incl. every relayout, and torch.equal of the results.
usage: python3 tools/dw_direct_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ogmm_amd import ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    dev = torch.device("cuda", 0)
    print("# dW = dY^T X, R = %d rows; ms per call (relayouts + engine + partial sums); TF-alg = 2 R n k / time" % R)
    print("%6s %6s %6s  %10s %10s  %8s %8s  %s" % ("n", "k", "bias", "copy ms", "direct ms", "copy TF", "direct TF", "equal"))
    for n, k in ((1024, 1024), (512, 1024), (1024, 512), (512, 512), (256, 512), (512, 256), (1024, 256)):
        for colsum in (False, True):
            dy = torch.randn(R, n, device=dev)
            x = torch.randn(R, k, device=dev)
            res = {}
            for tag, flag in (("copy", False), ("direct", True)):
                ops.DW_TRANSPOSED_A = flag
                out = ops.weight_grad(dy, [x], colsum=colsum)
                res[tag] = (out[0] if colsum else out, timed(lambda: ops.weight_grad(dy, [x], colsum=colsum)))
            fl = 2.0 * R * n * k
            print("%6d %6d %6s  %10.3f %10.3f  %8.1f %8.1f  %s" % (n, k, colsum, res["copy"][1], res["direct"][1], fl / res["copy"][1] / 1e9, fl / res["direct"][1] / 1e9,
                                                                 torch.equal(res["copy"][0], res["direct"][0])))
            del dy, x, res
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
