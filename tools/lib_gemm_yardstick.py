"""Library yardstick for the GEMM engine: hipBLASLt (torch.matmul) on the forward's dominant shape in binary16, bf16 and fp32."""
import torch
M, N, K = 131072, 1024, 1024
for dt in (torch.float16, torch.bfloat16, torch.float32):
    a = torch.randn(M, K, device="cuda", dtype=dt); b = torch.randn(N, K, device="cuda", dtype=dt)
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-16s %7.3f ms  %7.1f TFLOP/s" % (str(dt), ms, 2.0 * M * N * K / ms / 1e9))
