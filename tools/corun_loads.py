"""Developer aid (GPU box): WHICH co-running kernels change the FPS chains / the cluster means?  Victims on the default stream, a loop of one load kernel on a second stream."""
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
eng16 = ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev))
eng32 = ops.Engine("f32", None)
x = torch.randn(C * N, 512, device=dev)
xh = x.half()
w = torch.randn(512, 512, device=dev)
wh = w.half()
gamma = torch.softmax(torch.randn(C, N, 16, device=dev), -1); pi = gamma.mean(1)
victims = {"fps": lambda: ops.fps(xyz, 128, starts), "gmm_feat_mean": lambda: ops.gmm_feat_mean(gamma, pi, x, C, N)}
loads = {
    "fp16x3 small-tile GEMM (v2)": lambda: ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng16),
    "exact-fp32 engine GEMM": lambda: ops.conv1x1(x, L["emd5"], ops.ACT_RELU, eng=eng32),
    "hipBLASLt fp32 matmul": lambda: x @ w.t(),
    "hipBLASLt f16 matmul": lambda: xh @ wh.t(),
    "l2norm_rows (pointwise)": lambda: ops.l2norm_rows(x),
    "torch elementwise (x * 2)": lambda: x * 2.0,
    "torch softmax over rows": lambda: torch.softmax(x, -1),
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
        print("%-14s beside %-30s: %2d / %d runs differ" % (vname, lname, bad, REPS), flush=True)
