"""`torch.ops.ogmm.*`: the path's operator surface as registered PyTorch custom ops (ogmm_amd/torch_ops.py; SURVEY.md section 8b's list).
CPU: the schemas exist and say what section 8b says, fake kernels give the right shapes, CPU tensors are refused (no fallback).
GPU: torch.library.opcheck on the differentiable head and a selection op, a whole forward assembled from the registered ops alone against
the model, and two models of different precision in one process."""
from argparse import Namespace

import pytest
import torch

from ogmm_amd import synth
from ogmm_amd import torch_ops as T
from ogmm_amd.gmmreg import GMMReg

CFG = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)


def test_every_op_of_the_survey_list_is_registered_with_its_schema():
    assert set(T.SCHEMAS) == {"knn_idx", "edgeconv_dgcnn", "fps", "pos_encoding", "anchor_transformer", "conv_mlp", "overlap_cross", "gmm_em",
                              "gmm_feat_mean", "match_kabsch", "kabsch", "clu_infonce"}          # SURVEY.md 8b "C-ABI / op surface to export"
    for name, want in T.SCHEMAS.items():
        op = getattr(torch.ops.ogmm, name)
        assert str(op.default._schema) == want, name


def test_fake_kernels_trace_shapes_without_a_gpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        xyz = torch.empty(4, 100, 3)
        assert torch.ops.ogmm.knn_idx(xyz, 7).shape == (4, 100, 7)
        assert torch.ops.ogmm.fps(xyz, 16, None, True).shape == (4, 16)
        assert torch.ops.ogmm.fps(xyz, 16, torch.empty(3, 4, dtype=torch.int32), False).shape == (3, 4, 16)
        g, pi, mu = torch.ops.ogmm.gmm_em(xyz, torch.empty(4, 100), torch.empty(4, 8, dtype=torch.int32), 10, 10, 1e-2, 1e-2, 1.0, 2)
        assert g.shape == (4, 100, 8) and pi.shape == (4, 8) and mu.shape == (4, 8, 3)
        assert torch.ops.ogmm.gmm_feat_mean(g, pi, torch.empty(400, 64)).shape == (4, 8, 64)
        R, t = torch.ops.ogmm.match_kabsch(mu[:2], mu[2:], torch.empty(2, 8, 64), torch.empty(2, 8, 64), 0.05)
        assert R.shape == (2, 3, 3) and t.shape == (2, 3)
        R, t = torch.ops.ogmm.kabsch(torch.empty(2, 3, 8), torch.empty(2, 3, 8), torch.empty(2, 1, 8))
        assert R.shape == (2, 3, 3) and t.shape == (2, 3, 1)
        a, b = torch.ops.ogmm.overlap_cross(torch.empty(2, 100, 64), torch.empty(2, 100, 64), torch.empty(2, 100), torch.empty(2, 100), 1, None)
        assert a.shape == (2, 100) and b.shape == (2, 100)


def test_cpu_tensors_are_refused():
    with pytest.raises((RuntimeError, NotImplementedError)):          # no CPU kernel is registered: the product path has no fallback
        torch.ops.ogmm.knn_idx(torch.zeros(1, 10, 3), 3)


# ---------------------------------------------------------------------------------------------------------------------- GPU
def _model(precision="f16x3", J=16):
    m = GMMReg(512, J, Namespace(**{**vars(CFG), "n_clusters": J}))
    m.precision = precision
    synth.fill_state_dict(m.state_dict())
    return m.cuda().eval()


@pytest.mark.gpu
def test_opcheck_on_three_ops():
    from torch.library import opcheck
    torch.manual_seed(0)
    B, J, D = 3, 16, 64
    src = torch.randn(B, 3, J, device="cuda", requires_grad=True)
    corr = torch.randn(B, 3, J, device="cuda", requires_grad=True)
    w = torch.rand(B, 1, J, device="cuda", requires_grad=True)
    opcheck(torch.ops.ogmm.kabsch, (src, corr, w))
    mu_s, mu_t = torch.randn(B, J, 3, device="cuda", requires_grad=True), torch.randn(B, J, 3, device="cuda", requires_grad=True)
    f_s, f_t = torch.randn(B, J, D, device="cuda", requires_grad=True), torch.randn(B, J, D, device="cuda", requires_grad=True)
    opcheck(torch.ops.ogmm.match_kabsch, (mu_s, mu_t, f_s, f_t, 0.05))
    gamma = torch.softmax(torch.randn(2, 128, J, device="cuda"), -1)
    feats = torch.randn(2 * 128, D, device="cuda", requires_grad=True)
    opcheck(torch.ops.ogmm.gmm_feat_mean, (gamma, gamma.mean(1), feats))
    opcheck(torch.ops.ogmm.knn_idx, (torch.randn(2, 200, 3, device="cuda"), 8))
    # an op that writes the engines' fp16-range flag: declared as mutated, which opcheck's schema test verifies against what the
    # kernel actually touches
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    fs, ft = torch.randn(2, 256, 64, device="cuda"), torch.randn(2, 256, 64, device="cuda")
    opcheck(torch.ops.ogmm.overlap_cross, (fs, ft, torch.randn(2, 256, device="cuda"), torch.randn(2, 256, device="cuda"), 1, ovf))


@pytest.mark.gpu
def test_match_kabsch_gradients_against_autograd_of_the_oracle():
    from oracle import ogmm_oracle as O
    torch.manual_seed(1)
    B, J, D = 4, 16, 96
    args = [torch.randn(B, J, 3), torch.randn(B, J, 3), torch.randn(B, J, D), torch.randn(B, J, D)]
    args[3] = args[2][:, torch.randperm(J)] + 0.2 * torch.randn(B, J, D)
    gR, gt = torch.randn(B, 3, 3), torch.randn(B, 3)
    ref_in = [a.double().requires_grad_(True) for a in args]
    R, t, _ = O.match_and_solve(*ref_in)
    ((R * gR.double()).sum() + (t * gt.double()).sum()).backward()
    got_in = [a.cuda().requires_grad_(True) for a in args]
    R2, t2 = torch.ops.ogmm.match_kabsch(*got_in, 0.05)
    ((R2 * gR.cuda()).sum() + (t2 * gt.cuda()).sum()).backward()
    for a, b in zip(ref_in, got_in):
        scale = a.grad.abs().max().item()
        assert (a.grad - b.grad.cpu().double()).abs().max().item() < 2e-4 * max(scale, 1.0)


@pytest.mark.gpu
def test_a_forward_assembled_from_the_registered_ops_alone_equals_the_model():
    """models/gmmreg.py:50-119 written against torch.ops.ogmm.* only (packed weights from the model): the op surface is complete, and what it
    computes is what GMMReg.forward computes (same kernels: R, t, overlap scores and loss agree to the last bits that kernel-fusion choices move)."""
    B, N, J = 2, 1024, 16
    model = _model()
    src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
    starts = synth.fps_starts_for(0, B, N)
    with torch.no_grad():
        want = model(src.cuda(), tgt.cuda(), fps_starts=starts)
    L = model._layers()
    P = T.pack_layers
    o = torch.ops.ogmm
    C, D, H, k, M = 2 * B, 512, 4, 20, 128
    dev = torch.device("cuda")
    xyz = torch.cat([src, tgt]).cuda().transpose(1, 2).contiguous()
    st = starts.reshape(3, 2 * B).to(device=dev, dtype=torch.int32)
    swap = torch.cat([torch.arange(B, C), torch.arange(0, B)]).to(device=dev, dtype=torch.int32)
    with torch.no_grad():
        idx, idx5 = o.knn_idx(xyz, k), o.knn_idx(xyz, 5)
        ids_a, ids_j = o.fps(xyz, M, st, False), o.fps(xyz, J, None, True)
        emb = o.edgeconv_dgcnn(xyz, idx, *P([L["emd%d" % i] for i in range(1, 6)]), 1, None)
        pos = o.pos_encoding(xyz, idx5, [L["pos"][key] for key in ("w_dis", "s_dis", "t_dis", "w_ang", "s_ang", "t_ang")], *P([L["pos_dis2"], L["pos_ang2"]]), 1, None)
        x0 = emb + pos

        def tr(name, x, feats, ids, cmap=None):
            T_ = L[name]
            return o.anchor_transformer(x, feats, ids, cmap, N, H, *P([T_["q"], T_["kv"], T_["mlp0_folded"], T_["mlp3"]]), 1, None)
        t1 = tr("sattn1", x0, emb, ids_a[0])
        ft = o.conv_mlp(t1, None, *P([L["conv1"]["0"], L["conv1"]["3"], L["conv1"]["6"]]), [T.ACT_RELU, T.ACT_RELU, T.ACT_NONE], 1, None, None)
        # the engines' fp16-range flag through the dispatcher (declared as a mutated argument): an input beyond +-65504 must raise it
        ovf = torch.zeros(1, dtype=torch.int32, device=dev)
        o.conv_mlp(t1 * 1e7, None, *P([L["conv1"]["0"], L["conv1"]["3"], L["conv1"]["6"]]), [T.ACT_RELU, T.ACT_RELU, T.ACT_NONE], 1, ovf, None)
        assert int(ovf.item()) != 0
        f = tr("cattn", ft, ft, ids_a[1], swap)
        head = {"W": L["proj"]["3"]["w"].view(1, -1).contiguous(), "shift": L["proj"]["3"]["b"]}
        ol = o.conv_mlp(f, None, *P([L["proj"]["0"], head]), [T.ACT_RELU, T.ACT_NONE], 1, None, None).view(C, N)
        wo_s, wo_t = o.overlap_cross(f[:B * N].view(B, N, D), f[B * N:].view(B, N, D), ol[:B], ol[B:], 1, None)
        XW = L["conv2"]["0"]["W"].shape[1] - D
        extra = torch.zeros((C * N, XW), device=dev)
        extra[:, 0], extra[:, 1] = torch.cat([wo_s, wo_t]).view(-1), ol.view(-1)
        head2 = {"W": L["overlap"]["6"]["w"].view(1, -1).contiguous(), "shift": L["overlap"]["6"]["b"]}
        ov = o.conv_mlp(f, extra, *P([L["conv2"]["0"], L["conv2"]["3"], L["conv2_6_overlap_0"], L["overlap"]["3"], head2]),
                        [T.ACT_RELU, T.ACT_RELU, T.ACT_RELU, T.ACT_RELU, T.ACT_SIGMOID], 1, None, None).view(C, N)
        f2 = tr("sattn2", f, f, ids_a[2])
        gamma, pi, mu = o.gmm_em(xyz, ov.contiguous(), ids_j, 10, 10, 1e-2, 1e-2, 1.0, B)
        muf = o.gmm_feat_mean(gamma, pi, f2)
        R, t = o.match_kabsch(mu[:B], mu[B:], muf[:B], muf[B:], 0.05)
        row_loss, _ = o.clu_infonce(xyz, mu, f2, muf, 0.1)
        loss = row_loss.mean()
    from oracle import ogmm_oracle as O
    assert O.rotation_error_rad(R.cpu(), want[0].cpu()).max().item() < 2e-6 and O.translation_error(t.cpu(), want[1].cpu()).max().item() < 2e-6
    assert (ov[:B] - want[2]).abs().max().item() < 2e-6 and (ov[B:] - want[3]).abs().max().item() < 2e-6
    assert abs(loss.item() - want[4].item()) < 1e-5


@pytest.mark.gpu
def test_two_models_of_different_precision_in_one_process_keep_their_own_engines():
    """The engine choice travels with each model's calls (ops.Engine), not through module state: interleaved forwards of an fp16x3, an exact-fp32 and a
    reduced-precision model give what each gives alone."""
    B, N = 2, 512
    src, tgt, _, _ = synth.make_batch(40, B, N, "partial")
    starts = synth.fps_starts_for(40, B, N)
    src, tgt = src.cuda(), tgt.cuda()
    models = {p: _model(p) for p in ("f16x3", "f32", "f16")}
    with torch.no_grad():
        alone = {p: [x.clone() for x in m(src, tgt, fps_starts=starts)] for p, m in models.items()}
        for order in (("f16", "f32", "f16x3"), ("f32", "f16x3", "f16"), ("f16x3", "f16", "f32")):
            for p in order:
                got = models[p](src, tgt, fps_starts=starts)
                for a, b in zip(got, alone[p]):
                    assert torch.equal(a, b), p
    assert not torch.equal(alone["f32"][0], alone["f16x3"][0])          # (the engines do differ; at this small shape "f16" runs the three-term small-tile kernels)
