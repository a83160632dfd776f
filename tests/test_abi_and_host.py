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
