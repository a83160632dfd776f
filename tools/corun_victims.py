"""Developer aid (GPU box): which kernels' results change when a small-tile GEMM shares the chip?  Each victim (FPS chains, the kNN head, the E/M, the nearest-point
search, the cluster means) runs on the default stream while a second stream runs a loop of small GEMMs; results against the victim's own solo result."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg

dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(model.state_dict()); model = model.to(dev).eval()
L = model._layers()
B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 6), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
C = 2 * B
src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
starts = synth.fps_starts_for(0, B, N).reshape(3, C).to(torch.int32).to(dev)
xyz = ops.pack_clouds(src.to(dev), tgt.to(dev))
eng = ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev))
x = torch.randn(C * N, 512, device=dev)
o = torch.rand(C, N, device=dev)
ids_j = ops.fps(xyz, 16, None)
gamma, pi, mu = ops.gmm_em(xyz, o, ids_j, iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=1e-2, group_size=B)
victims = {
    "fps (random starts)": lambda: ops.fps(xyz, 128, starts),
    "knn head": lambda: torch.cat([t.float().flatten() for t in ops.knn_pos_head(xyz, 20, L["pos"])]),
    "gmm_em": lambda: torch.cat([t.flatten() for t in ops.gmm_em(xyz, o, ids_j, iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=1e-2, group_size=B)]),
    "nearest_point": lambda: ops.nearest_point(xyz, mu),
    "gmm_feat_mean": lambda: ops.gmm_feat_mean(gamma, pi, x, C, N),
    "attention": lambda: ops.attention(x, x[:C * 128], x[:C * 128], C, N, 128, 4),
}
loads = {
    "nothing": lambda: None,
    "small GEMM 512 -> 512": lambda: ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng),
    "small GEMM 512 -> 1024": lambda: ops.conv1x1(x, L["conv1"]["0"], ops.ACT_RELU, eng=eng),
}
other = torch.cuda.Stream()
for vname, v in victims.items():
    ref = v().clone()
    torch.cuda.synchronize()
    for lname, load in loads.items():
        bad = 0
        for rep in range(REPS):
            with torch.cuda.stream(other):
                for _ in range(8):
                    load()
            got = v()
            torch.cuda.synchronize()
            bad += int(not torch.equal(got, ref))
        print("%-22s beside %-24s: %2d / %d runs differ" % (vname, lname, bad, REPS), flush=True)
# and the other way round: is the GEMM's own result stable while FPS chains run beside it?
ref = ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng).clone()
torch.cuda.synchronize()
bad = 0
for rep in range(REPS):
    with torch.cuda.stream(other):
        for _ in range(4):
            ops.fps(xyz, 128, starts)
    got = ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng)
    torch.cuda.synchronize()
    bad += int(not torch.equal(got, ref))
print("small GEMM 512 -> 512 beside FPS chains: %d / %d runs differ" % (bad, REPS))
