import sys, time, torch
sys.path.insert(0, '/root/repo')
from ogmm_amd import losses, ops
B, N = 128, 1024
torch.manual_seed(0)
src = torch.randn(B, N, 3, device="cuda"); tgt = torch.randn(B, N, 3, device="cuda")
R = torch.eye(3, device="cuda").repeat(B, 1, 1).requires_grad_(True); t = torch.zeros(B, 3, device="cuda", requires_grad=True)
so, to = torch.rand(B, N, device="cuda"), torch.rand(B, N, device="cuda")
def run_new():
    l = losses.welsch_loss(src, tgt, R, t, so, to, 10.0, 512); l.backward(); return l
def run_old():
    moved = torch.bmm(src, R.transpose(1, 2)) + t.reshape(-1, 1, 3)
    s_ids = torch.topk(so, k=512, dim=-1)[1]; t_ids = torch.topk(to, k=512, dim=-1)[1]
    take = lambda p, ids: torch.gather(p, 1, ids[:, :, None].expand(-1, -1, 3))
    z1 = torch.cdist(take(moved, s_ids), tgt).min(dim=-1)[0]; z2 = torch.cdist(take(tgt, t_ids), moved).min(dim=-1)[0]
    l = (2.0 - torch.exp(-0.5 * z1 * z1 / 100.0) - torch.exp(-0.5 * z2 * z2 / 100.0)).sum(dim=1).mean(); l.backward(); return l
for name, fn in (("cdist", run_old), ("nearest_point", run_new)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): l = fn()
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t0) * 100), float(l))
