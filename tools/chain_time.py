"""The CONV stack of models/dgcnn.py:19-28 (conv1: 512 -> 1024 -> 1024 -> 512) as three launches of the GEMM engine against ONE chain launch (ogmm_gemm_chain):
outputs (intermediates included) must be identical; time per stack at the headline shape.  usage (GPU box): python3 tools/chain_time.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(model.state_dict()); model = model.to(dev).eval()
L = model._layers()
eng = ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev))
x = torch.randn(rows, 512, device=dev)
x2 = torch.rand(rows, 32, device=dev)


def stack(S, chain, x2=None):
    ops.GEMM_CHAIN = chain
    outs = []
    with ops.gemm_chain():
        h = ops.conv1x1(x, S["0"], ops.ACT_RELU, x2=x2, eng=eng); outs.append(h)
        h = ops.conv1x1(h, S["3"], ops.ACT_RELU, eng=eng); outs.append(h)
        h = ops.conv1x1(h, S["6"], eng=eng); outs.append(h)
    return outs


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, S, xx in (("conv1", L["conv1"], None), ("conv2 (two A pieces in its first layer)", L["conv2"], x2)):
    a, b = stack(S, False, xx), stack(S, True, xx)
    torch.cuda.synchronize()
    print("%s, %d rows: identical %s; three launches %.1f us, chain %.1f us" % (
        name, rows, [bool(torch.equal(p, q)) for p, q in zip(a, b)], timed(lambda: stack(S, False, xx)), timed(lambda: stack(S, True, xx))))
