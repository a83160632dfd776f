"""Developer aid (GPU box): are the FPS chains reproducible while another kernel family shares the chip?  ops.fps on one stream, repeated, while a second
stream runs (a) nothing, (b) the persistent EdgeConv kernel, (c) a large GEMM of the fp16x3 engine, (d) the kNN head kernel, (e) the E/M kernel."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg

dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(model.state_dict()); model = model.to(dev).eval()
L = model._layers()
B, N, M = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 128
REPS, LOADS = (int(sys.argv[3]) if len(sys.argv) > 3 else 12), (int(sys.argv[4]) if len(sys.argv) > 4 else 3)
src, tgt, _, _ = synth.make_batch(0, B, N, "partial"); starts = synth.fps_starts_for(0, B, N).reshape(3, 2 * B).to(torch.int32).to(dev)
xyz = ops.pack_clouds(src.to(dev), tgt.to(dev))
C = 2 * B
idx = ops.knn(xyz, 20)
ref = ops.fps(xyz, M, starts).clone()
refj = ops.fps(xyz, 16, None).clone()
torch.cuda.synchronize()
other = torch.cuda.Stream()
eng = ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev))
xcat = torch.empty((C * N, 512), dtype=torch.float32, device=dev)
x = torch.randn(C * N, 512, device=dev)
o = torch.rand(C, N, device=dev)
emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
loads = {
    "nothing": lambda: None,
    "edgeconv": lambda: ops.edgeconv_fused(xyz, idx, emd, xcat),
    "gemm 512x512": lambda: ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng),
    "pos GEMM (K=64)": lambda: ops.conv1x1(x[:, :64], L["pos_dis2"], ops.ACT_LEAKY02, eng=eng),
    "attention chain": lambda: ops.attention(x, x[:C * 128], x[:C * 128], C, N, 128, 4),
    "gemm 1024 (conv1.0)": lambda: ops.conv1x1(x, L["conv1"]["0"], ops.ACT_RELU, eng=eng),
    "knn head": lambda: ops.knn_pos_head(xyz, 20, L["pos"]),
    "E/M": lambda: ops.gmm_em(xyz, o, refj, iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=1e-2, group_size=B),
}
for name, load in loads.items():
    bad = badj = 0
    for rep in range(REPS):
        with torch.cuda.stream(other):
            for _ in range(LOADS):
                load()
        got = ops.fps(xyz, M, starts)
        gotj = ops.fps(xyz, 16, None)
        torch.cuda.synchronize()
        bad += int(not torch.equal(got, ref))
        badj += int(not torch.equal(gotj, refj))
    print("FPS beside %-22s: %2d / %d runs differ (random starts), %2d / %d (centre start)" % (name, bad, REPS, badj, REPS), flush=True)
