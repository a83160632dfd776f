"""The fused attention kernel alone at the headline shape (C = 128 clouds, N = 1024 queries, M = 128 anchors, H = 4 heads), three inputs: time per call, achieved
HBM rate on the algorithmic bytes (Q in, O out, K / V in), and a checksum of the output bits (to A/B bit-identical forms across processes).
    OGMM_ATTN_GX=n python tools/attention_time.py       # query tiles per workgroup grid dimension
Round 6 used it for a null: the second product as O = P V instead of O^T = V^T P^T -- the same fragments with their roles swapped, bit-identical,
lane = channel so that a store writes two whole 128-byte row segments instead of sixty-four 16-byte pieces -- ran 147.1 / 148.7 us against 139.8 at N = 1024, 312 against 323 at
N = 2048, 119-123 against 126 at N = 717: the output stores are not what bounds this kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ogmm_amd import ops  # noqa: E402

torch.manual_seed(0)
C, N, M, H, D = 128, 1024, 128, 4, 512
for N in (1024, 717, 2048):
    q = torch.randn(C * N, D, device="cuda")
    k = torch.randn(C * M, D, device="cuda")
    v = torch.randn(C * M, D, device="cuda")
    out = torch.full_like(q, float("nan"))
    for _ in range(3):
        ops.attention(q, k, v, C, N, M, H, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention(q, k, v, C, N, M, H, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nbytes = 4.0 * (2 * C * N * D + 2 * C * M * D)
    bits = out.view(torch.int32).to(torch.int64)
    print("N=%d gx=%s  %.1f us  %.0f GB/s = %.2f of 8 TB/s   checksum %d  nan %d" % (
        N, os.environ.get("OGMM_ATTN_GX", "auto"), us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000.0,
        int((bits * (torch.arange(bits.numel(), device="cuda").view_as(bits) % 1000003 + 1)).sum().item()), int(torch.isnan(out).sum())))
