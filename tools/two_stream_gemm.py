"""Does running two half-size GEMM chains on two streams beat one full-size chain? (GPU box experiment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
torch.manual_seed(0)
M = 131072
def mk(n, k): 
    W = torch.randn(n, k, device="cuda") * 0.03
    return {"W": W, "shift": torch.zeros(n, device="cuda"), "split": ops.split_f16(W, frag=True)}
L = [mk(1024, 512), mk(1024, 1024), mk(512, 1024), mk(512, 512), mk(1024, 512), mk(512, 1024)]
x = torch.randn(M, 512, device="cuda")
def chain(inp):
    h = inp
    for l in L:
        h = ops.conv1x1(h, l, act=ops.ACT_RELU)
    return h
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    return chain(x)
def two():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): a = chain(x[:M // 2])
    with torch.cuda.stream(s2): b = chain(x[M // 2:])
    main.wait_stream(s1); main.wait_stream(s2)
    return a, b
for name, fn in (("one stream ", one), ("two streams", two), ("one stream ", one), ("two streams", two)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.3f ms" % (e0.elapsed_time(e1) / 5))
