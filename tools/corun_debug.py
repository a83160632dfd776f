import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg
dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(model.state_dict()); model = model.to(dev).eval()
L = model._layers()
B, N = 6, 1024
C = 2 * B
src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
starts = synth.fps_starts_for(0, B, N).reshape(3, C).to(torch.int32).to(dev)
xyz = ops.pack_clouds(src.to(dev), tgt.to(dev))
eng16 = ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev))
x = torch.randn(C * N, 512, device=dev)
gamma = torch.softmax(torch.randn(C, N, 16, device=dev), -1); pi = gamma.mean(1)
keep = {"x": x.clone(), "gamma": gamma.clone(), "pi": pi.clone(), "xyz": xyz.clone(), "starts": starts.clone()}
ref_fm = ops.gmm_feat_mean(gamma, pi, x, C, N).clone()
ref_fps = ops.fps(xyz, 128, starts).clone()
torch.cuda.synchronize()
other = torch.cuda.Stream()
# 1. does the load change any input?
with torch.cuda.stream(other):
    for _ in range(50):
        y = ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng16)
torch.cuda.synchronize()
print("inputs unchanged after 50 small GEMMs:", {k: bool(torch.equal(v, {"x": x, "gamma": gamma, "pi": pi, "xyz": xyz, "starts": starts}[k])) for k, v in keep.items()})
print("victims after the loads have finished (no concurrency): feat_mean", bool(torch.equal(ops.gmm_feat_mean(gamma, pi, x, C, N), ref_fm)), " fps", bool(torch.equal(ops.fps(xyz, 128, starts), ref_fps)))
# 2. concurrent, with the GEMM writing into a PRE-ALLOCATED output (no allocator traffic on the other stream)
out = torch.empty((C * N, 512), device=dev)
torch.cuda.synchronize()
bad_fm = bad_fps = 0
for rep in range(30):
    with torch.cuda.stream(other):
        for _ in range(8):
            ops.conv1x1(x, L["emd5"], ops.ACT_RELU, out=out, eng=eng16)
    g1 = ops.gmm_feat_mean(gamma, pi, x, C, N)
    g2 = ops.fps(xyz, 128, starts)
    torch.cuda.synchronize()
    bad_fm += int(not torch.equal(g1, ref_fm)); bad_fps += int(not torch.equal(g2, ref_fps))
print("concurrent, pre-allocated GEMM output: feat_mean differs %d / 30, fps differs %d / 30" % (bad_fm, bad_fps))
# 3. how far off is feat_mean when it differs?
with torch.cuda.stream(other):
    for _ in range(8):
        ops.conv1x1(x, L["emd5"], ops.ACT_RELU, out=out, eng=eng16)
g1 = ops.gmm_feat_mean(gamma, pi, x, C, N)
torch.cuda.synchronize()
d = (g1 - ref_fm).abs()
print("feat_mean difference: max %.3e, entries differing %d of %d, where (cloud, j) : %s" % (d.max().item(), int((d > 0).sum()), d.numel(), sorted(set(map(tuple, (d > 0).nonzero()[:, :2].tolist())))[:10]))
# 4. the victim's OUTPUT buffer pre-allocated too? (ops.gmm_feat_mean allocates its output with torch.empty on the default stream)
