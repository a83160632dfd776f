"""GPU: the side-stream kernels give the same bits beside the fp16x3 GEMMs as alone.

Round 5 traced the run-to-run differences of FPS queued beside the GEMMs to the hardware, not to a race in any kernel: a packed-fp32
instruction with non-default operand selects (the compiler's form of "pair (op) broadcast scalar") computes lanes 48-63 with the default selects
when a wave of another kernel on the same SIMD issues an f16 matrix instruction beside it (tools/pk_mfma_hazard.hip: no product code;
HISTORY.md section 4).  The library is built without packed-fp32 instructions (tests/test_isa_hazards.py); this is the behavioural check on the
part itself: the kernels the forward runs on its side streams -- FPS, the cluster-feature means, the GMM E/M, the nearest-point search -- beside
a stream that keeps launching the small-tile fp16x3 GEMM (the engine whose workgroups share compute units with them), 30 times each."""
from argparse import Namespace

import pytest
import torch

from ogmm_amd import synth

pytestmark = pytest.mark.gpu
CFG = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
REPS, LOADS = 30, 8


@pytest.fixture(scope="module")
def rig():
    assert torch.cuda.is_available()
    from ogmm_amd import ops
    from ogmm_amd.gmmreg import GMMReg
    dev = torch.device("cuda", 0)
    model = GMMReg(512, 16, CFG)
    synth.fill_state_dict(model.state_dict())
    model = model.to(dev).eval()
    L = model._layers()
    B, N = 6, 1024
    C = 2 * B
    src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
    xyz = ops.pack_clouds(src.to(dev), tgt.to(dev))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(C * N, 512, generator=g).to(dev)
    gamma = torch.softmax(torch.randn(C, N, 16, generator=g), -1).to(dev)
    o = torch.rand(C, N, generator=g).to(dev)
    return Namespace(ops=ops, dev=dev, L=L, B=B, N=N, C=C, xyz=xyz, x=x, gamma=gamma, pi=gamma.mean(1).contiguous(), o=o,
                     starts=synth.fps_starts_for(0, B, N).reshape(3, C).to(torch.int32).to(dev),
                     eng=ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev)), out=torch.empty((C * N, 512), device=dev),
                     other=torch.cuda.Stream())


def _beside_gemms(rig, fn):
    """fn() alone, then REPS times with LOADS small-tile GEMMs queued on another stream just before it: every result must equal the first."""
    ops = rig.ops
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else (ref,))]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(REPS):
        with torch.cuda.stream(rig.other):
            for _ in range(LOADS):
                ops.conv1x1(rig.x, rig.L["emd5"], ops.ACT_RELU, out=rig.out, eng=rig.eng)
        got = fn()
        torch.cuda.synchronize()
        got = got if isinstance(got, (tuple, list)) else (got,)
        bad += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
    return bad


def test_fps_beside_small_tile_gemms(rig):
    assert _beside_gemms(rig, lambda: rig.ops.fps(rig.xyz, 128, rig.starts)) == 0


def test_cluster_feature_means_beside_small_tile_gemms(rig):
    assert _beside_gemms(rig, lambda: rig.ops.gmm_feat_mean(rig.gamma, rig.pi, rig.x, rig.C, rig.N)) == 0


def test_gmm_em_beside_small_tile_gemms(rig):
    ids = rig.ops.fps(rig.xyz, 16, None)
    assert _beside_gemms(rig, lambda: rig.ops.gmm_em(rig.xyz, rig.o, ids, thresh=0.0)[:3]) == 0


def test_nearest_point_and_infonce_beside_small_tile_gemms(rig):
    mu = rig.xyz[:, :16, :].contiguous()
    muf = rig.ops.gmm_feat_mean(rig.gamma, rig.pi, rig.x, rig.C, rig.N)
    assert _beside_gemms(rig, lambda: rig.ops.clu_infonce(rig.xyz, mu, rig.x, muf, rig.C, rig.N, 0.1)) == 0
