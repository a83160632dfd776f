"""Developer aid (GPU box): per-stage max-abs difference between the HIP forward and the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from oracle import ogmm_oracle as O
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg

B, N, J = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 16
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
m = GMMReg(512, J, cfg); synth.fill_state_dict(m.state_dict())
P = {k: v.clone() for k, v in m.state_dict().items()}
m = m.cuda().eval()
src, tgt, _, _ = synth.make_batch(0, B, N, "partial"); starts = synth.fps_starts_for(0, B, N)
cap = {}
with torch.no_grad():
    t0 = time.time(); ref = O.forward(P, cfg, src, tgt, starts, cap); print("oracle %.2fs" % (time.time() - t0))
    out = m(src.cuda(), tgt.cuda(), fps_starts=starts, capture=True); torch.cuda.synchronize()
g = m.last_intermediates
def both(key): return torch.cat([cap[key + "_src"], cap[key + "_tgt"]], 0)
print("knn equal", torch.equal(g["knn_idx"].cpu().long(), both("knn_idx")))
for st in range(3): print("fps%d equal" % st, torch.equal(g["fps_anchor"][st].cpu().long(), both("fps%d" % st)))
print("fpsJ equal", torch.equal(g["fps_J"].cpu().long(), both("fpsJ")))
def feat(key, ref_key=None):
    r = both(ref_key or key).transpose(1, 2).reshape(2 * B * N, -1)
    d = (g[key].cpu() - r).abs().max().item(); print("%-6s max|diff| %.3e  (|ref| max %.3e)" % (key, d, r.abs().max().item()))
feat("emb"); x0 = both("emb") + both("pos"); print("x0     max|diff| %.3e" % (g["x0"].cpu() - x0.transpose(1, 2).reshape(2 * B * N, -1)).abs().max().item())
feat("ft"); feat("f"); feat("f2")
print("wo     %.3e" % (g["wo"].cpu() - both("wo").reshape(-1)).abs().max().item())
print("o      %.3e" % (g["o"].cpu() - both("o")).abs().max().item())
print("gamma  %.3e  pi %.3e  mu %.3e  muf %.3e" % ((g["gamma"].cpu() - both("gamma")).abs().max().item(), (g["pi"].cpu() - both("pi")).abs().max().item(),
      (g["mu"].cpu() - both("mu")).abs().max().item(), (g["muf"].cpu() - both("muf")).abs().max().item()))
print("near equal", torch.equal(g["near"].cpu().long(), both("near")))
print("R err", O.rotation_error_rad(out[0].cpu(), ref[0]).tolist(), "t err", O.translation_error(out[1].cpu(), ref[1]).tolist())
print("loss", out[4].item(), ref[4].item())
