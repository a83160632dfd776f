"""GEMM-engine microbenchmark on the forward's shapes (GPU box).  usage: gemm_bench.py [variant codes...]
   0 = exact-fp32 engine, 1 = fp16x3 row-major B (128x128), 2 = fp16x3 default dispatch, 21 = v2 128x128, 22 = v2 128x256 (2 WG/CU),
   23 = large-shape engine (v4), ablations of it: 26 = MFMA + barrier only, 29 = + A path only, 18 = no output stores, 19 = MFMA only, no stores, 24 = the default pipeline,
   25 = default without the epilogue, 40 = default with every tile stored to one 256 x 256 patch (epilogue instructions without HBM write traffic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
import probe          # tools/probe.py: the ablation codes live in libogmm_probe.so
probe.install()

variants = [int(v) for v in sys.argv[1:]] or [0, 1]
M = 131072
shapes = [("mlp0 1024x(512+512)", M, 1024, 512, 512), ("conv.3 1024x1024", M, 1024, 1024, 0), ("conv2.0 1024x(512+4)", M, 1024, 512, 4), ("mlp3 512x1024", M, 512, 1024, 0), ("conv.0 1024x512", M, 1024, 512, 0),
          ("q/merge 512x512", M, 512, 512, 0), ("proj 256x512", M, 256, 512, 0), ("pos 256x64", M, 256, 64, 0)]
torch.manual_seed(0)
for name, m, n, k1, k2 in shapes:
    A = torch.randn(m, k1, device="cuda")
    A2 = torch.randn(m, k2, device="cuda") if k2 else None
    W = torch.randn(n, k1 + k2, device="cuda") * 0.03
    out = torch.empty(m, n, device="cuda")
    row = "%-22s" % name
    ref = None
    for v in variants:
        split = None
        if v != 0:
            split = ops.split_f16(W, frag=(v == 2 or v >= 18)); split["variant"] = v
        def run():
            ops.gemm_nt(A, k1, k1, W, k1 + k2, m, n, C=out, ldc=n, A2=A2, lda2=k2, K2=k2, split=split)
        run(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        err = (out - ref).abs().max().item()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        ms = best
        row += "  v%-2d %7.1f TF (%6.3f ms, d=%.1e)" % (v, 2.0 * m * n * (k1 + k2) / ms / 1e9, ms, err)
    print(row)
