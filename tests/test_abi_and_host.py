"""CPU: the C-ABI library loads and exports every symbol include/ogmm_hip.h declares; host logic
(state_dict surface, weight packing, error behaviour) without any compute call."""
import ctypes
import os
import re
from argparse import Namespace

import pytest
import torch

from ogmm_amd import _lib, synth
from ogmm_amd.gmmreg import GMMReg, pack_weights, state_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.isfile(_lib.LIB_PATH):
        g.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "ogmm_hip.h")).read()
    declared = set(re.findall(r"\b(ogmm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.PROTOTYPES), "binding table and header disagree: %s" % (declared ^ set(_lib.PROTOTYPES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ogmm_abi_version() == _lib.ABI_VERSION


def test_gemm_desc_matches_header_layout():
    header = open(os.path.join(ROOT, "include", "ogmm_hip.h")).read()
    body = header[header.index("typedef struct ogmm_gemm {"):header.index("} ogmm_gemm;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"[\*\s]([A-Za-z_0-9]+)\s*[;,]", body)
    assert fields == [f[0] for f in _lib.GemmDesc._fields_]


def test_state_dict_surface_matches_reference_keys():
    spec = state_spec(512)
    assert len(spec) == 153
    m = GMMReg(512, 16, CFG)
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in spec]
    for k, shape in spec:
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in m.parameters()) == 13022210
    assert sd["conv2.net.0.weight"].shape == (1024, 514, 1)


def test_pack_weights_head_permutation_and_bn_fold():
    m = GMMReg(512, 16, CFG)
    synth.fill_state_dict(m.state_dict())
    sd = m.state_dict()
    L = pack_weights(sd, 512, 4)
    # reference channel c = d*H + h  <->  packed channel c' = h*dh + d
    wq = sd["sattn1.attn.proj.0.weight"][:, :, 0]
    for (h, d) in ((0, 0), (1, 5), (3, 127)):
        assert torch.equal(L["sattn1"]["q"]["W"][h * 128 + d], wq[d * 4 + h])
        assert torch.equal(L["sattn1"]["merge"]["W"][:, h * 128 + d], sd["sattn1.attn.merge.weight"][:, d * 4 + h, 0])
    assert L["conv2"]["0"]["W"].shape == (1024, 544) and float(L["conv2"]["0"]["W"][:, 514:].abs().max()) == 0.0
    # folded eval-BN == F.batch_norm
    x = torch.randn(7, 256)
    ref = torch.nn.functional.batch_norm(x + sd["proj.net.0.bias"], sd["proj.net.1.running_mean"], sd["proj.net.1.running_var"],
                                         sd["proj.net.1.weight"], sd["proj.net.1.bias"], False, 0.1, 1e-5)
    got = x * L["proj"]["0"]["scale"] + L["proj"]["0"]["shift"]
    assert torch.allclose(ref, got, atol=2e-6)


def test_packed_weight_cache_follows_every_kind_of_edit():
    """The eval forward runs on folded / split weights cached per model.  The cache must notice optimizer-style in-place edits (tensor
    _version), edits through `.data` (no _version bump: content fingerprint, explicit invalidate), load_state_dict and train()/eval()."""
    m = GMMReg(512, 16, CFG).eval()
    synth.fill_state_dict(m.state_dict())
    L0 = m._layers()
    assert m._layers() is L0                                                  # unchanged weights: cached
    hi0 = L0["proj"]["0"]["split"]["W_hi"].clone()                            # (the fp32 "W" entries may alias the parameters: compare the split images)
    with torch.no_grad():
        m.state_dict()["proj.net.0.weight"].mul_(1.5)                          # bumps _version
    L1 = m._layers()
    assert L1 is not L0 and not torch.equal(L1["proj"]["0"]["split"]["W_hi"], hi0)
    m.fingerprint_every = 1
    p = dict(m.named_parameters())["conv1.net.0.weight"]
    c1 = L1["conv1"]["0"]["W"].clone()
    p.data.mul_(2.0)                                                          # invisible to (data_ptr, _version)
    L2 = m._layers()
    assert L2 is not L1 and torch.equal(L2["conv1"]["0"]["W"], 2.0 * c1)
    m.fingerprint_every = 0                                                   # checks off: only the explicit call helps
    p.data.mul_(0.5)
    assert m._layers() is L2
    m.invalidate_packed()
    L3 = m._layers()
    assert L3 is not L2 and torch.equal(L3["conv1"]["0"]["W"], c1)
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert m._packed is None                                                  # load_state_dict invalidates
    m._layers(); m.train(); assert m._packed is None
    m.eval(); m._layers(); assert m._packed is not None
    assert "conv2_6_overlap_0" in m._layers() and m._layers()["conv2_6_overlap_0"]["W"].shape == (256, 1024)


def test_product_path_refuses_cpu_and_train_mode():
    m = GMMReg(512, 16, CFG).eval()
    x = torch.zeros(1, 3, 128)
    with pytest.raises(_lib.OgmmError):
        m(x, x)                                   # CPU tensors: no fallback


def test_reference_checkpoint_roundtrip(tmp_path):
    """state_dict written by one instance loads into another (same contract as lib/metric.py:293-297 / train.py:219-225)."""
    a, b = GMMReg(512, 16, CFG), GMMReg(512, 16, CFG)
    synth.fill_state_dict(a.state_dict())
    path = str(tmp_path / "optim_model.pt")
    torch.save(a.state_dict(), path)
    b.load_state_dict(torch.load(path), strict=True)
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k])


def test_hot_kernels_do_not_spill():
    """The default GEMM engines, the EdgeConv kernel and the attention kernel keep everything in registers (code-object metadata of the build; a
    run-time branch added to one of them once cost 87 spilled registers without any other visible sign)."""
    import importlib.util, shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = os.path.join(root, "ogmm_amd", "csrc", "build")
    if not os.path.isdir(build) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None:
        pytest.skip("no build directory / LLVM tools here")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.main()
    # (the fused-head instantiation <0, false, true> spills in its four-way epilogue only -- behind the K loop -- and is not in this list)
    hot = ["gemm_f16x3_v8_kernel<0, false, false, 3>", "gemm_f16x3_v8_kernel<0, false, false, 2>", "gemm_f16x3_v8_kernel<0, true, false, 3>",
           "gemm_f16x3_v10_kernel<0, false, false, 3, false, 0>", "gemm_f16x3_v10_kernel<0, false, false, 2, false, 0>",
           "gemm_f16x3_v10_kernel<0, true, false, 3, false, 0>",
           "gemm_f16x3_v10_kernel<0, false, false, 3, true, 0>",          # (the normalisation-backward epilogue of the training step: its own instantiation)
           "gemm_f16x3_v10_kernel<0, false, false, 3, false, 1>",         # (the transposed-A form of the weight gradient, and with the bias gradient's column sums)
           "gemm_f16x3_v10_kernel<0, false, false, 3, false, 2>",
           "edgeconv_fused_kernel<20, false>",
           "attention_t_kernelILi4ELb1E",
           "knn_kernel<21>", "knn2_kernel<21>", "gmm_em_cached_kernel<16, true>"]
    for h in hot:
        found = [r for r in rows if h in r[1]]
        assert found, "kernel %s not found in the build" % h
        for r in found:
            assert r[4] == 0 and r[5] == 0, "%s spills: %d registers, %d bytes of scratch" % (r[1], r[4], r[5])
    # ogmm_knn_pos_head_supported bounds dynamic + STATIC LDS of the head kernel by 64 KiB; the constant it adds for the static part
    # (KNN_HEAD_STATIC_LDS = 8 KiB, knn_fps.hip) must cover what the build really declares
    head = [r for r in rows if "knn4_kernel<" in r[1]]
    assert head and max(r[6] for r in head) <= 8 * 1024, [(r[1][:60], r[6]) for r in head]


def test_knn_head_fit_counts_static_lds(lib):
    """N * 16 B of points + the candidate lists (2 * 20 * 256 int16 at k <= 20, 2 * 32 * 256 beyond) + 8 KiB of static arrays <= 64 KiB: N = 2304 is the last
    cloud size the fused head takes at k = 20 (it used to answer "supported" up to 2816 for a 72 KiB workgroup), 1536 at k = 32; the forward falls back to the
    three-kernel head beyond (tests/test_hip_ops.py::test_knn_head_boundary runs both sides on the GPU)."""
    assert lib.ogmm_knn_pos_head_supported(2304, 20) == 1 and lib.ogmm_knn_pos_head_supported(2305, 20) == 0
    assert lib.ogmm_knn_pos_head_supported(2816, 20) == 0
    assert lib.ogmm_knn_pos_head_supported(1536, 32) == 1 and lib.ogmm_knn_pos_head_supported(1537, 32) == 0
    assert lib.ogmm_knn_pos_head_supported(1024, 20) == 1 and lib.ogmm_knn_pos_head_supported(2048, 20) == 1
