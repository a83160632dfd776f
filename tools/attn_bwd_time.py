"""Times the attention backward kernel (T9) against the library path on the training shard's shape (C = 256 clouds, N = 1024, M = 128, H = 4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops, train_ops

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, M, H, D = 1024, 128, 4, 512
dev = "cuda:0"
torch.manual_seed(0)
q, k, v, g = (torch.randn(r_, D, device=dev) for r_ in (C * N, C * M, C * M, C * N))

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

t_k = timed(lambda: ops.attention_bwd(q, k, v, g, C, N, M, H))
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
t_s = timed(lambda: ops.attention_bwd(q, k, v, g, C, N, M, H, split=1, overflow=ovf))
t_a = timed(lambda: ops.attention_bwd(q, k, v, g, C, N, M, H, split=2, overflow=ovf))
def lib():
    a, b, c_ = (t_.detach().requires_grad_(True) for t_ in (q, k, v))
    o = train_ops._attention_torch(a, b, c_, C, N, M, H)
    torch.autograd.grad(o, (a, b, c_), g)
t_l = timed(lib)
flops = 5 * 2.0 * C * N * M * D
print("attention backward C=%d: kernel %.3f ms (%.1f TFLOP/s fp32 MFMA), library path %.3f ms" % (C, t_k, flops / t_k / 1e9, t_l))
print("attention backward C=%d, S and dP on the fp16x3 arithmetic (round 5): %.3f ms (%.1f TFLOP/s algorithmic)" % (C, t_s, flops / t_s / 1e9))
print("attention backward C=%d, all five products on the fp16x3 arithmetic (round 5): %.3f ms (%.1f TFLOP/s algorithmic)" % (C, t_a, flops / t_a / 1e9))
