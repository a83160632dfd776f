"""CPU: the re-statement of libstdc++'s heap-select / introselect in ogmm_amd/csrc/torch_topk_select.h keeps exactly
the candidates torch.topk(largest=False) keeps, including on rows full of exact ties (both of PyTorch's code paths:
k*64 <= n -> partial_sort, else nth_element)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("select") / "libselect_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(HERE, "host", "select_host.cpp"), "-o", out], check=True)
    lib = ctypes.CDLL(out)
    lib.ogmm_test_topk_set.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def run(lib, vals, k):
    rows, n = vals.shape
    out = np.empty((rows, k), dtype=np.int32)
    v = np.ascontiguousarray(vals, dtype=np.float32)
    lib.ogmm_test_topk_set(v.ctypes.data, rows, n, k, out.ctypes.data)
    return out


@pytest.mark.parametrize("n,k", [(1024, 20), (717, 20), (2048, 20), (1024, 5), (200, 12), (200, 5), (4096, 20), (1279, 20), (1280, 20),
                                 (64, 32), (33, 1), (2048, 64), (20, 20), (21, 20), (5000, 3)])
@pytest.mark.parametrize("levels", [3, 17, 400, 0])
def test_same_set_as_torch_topk(host_lib, n, k, levels):
    g = torch.Generator().manual_seed(n * 131 + k * 7 + levels)
    rows = 64
    if levels:
        vals = torch.randint(0, levels, (rows, n), generator=g).float() / levels          # heavy exact ties
    else:
        vals = torch.rand(rows, n, generator=g)
    ref_v, ref_i = torch.topk(vals, k, dim=-1, largest=False, sorted=True)
    got = torch.from_numpy(run(host_lib, vals.numpy(), k)).long()
    assert torch.equal(torch.gather(vals, 1, got), ref_v)
    assert torch.equal(got.sort(-1)[0], ref_i.sort(-1)[0]), "kept set differs from torch.topk"


def test_real_knn_rows_with_boundary_ties(host_lib):
    """Rows of real expanded-formula distances (room-like clouds have many exact ties)."""
    from ogmm_amd import synth
    from oracle import ogmm_oracle as O
    src, _, _, _ = synth.make_batch(400, 1, 2048, "room")
    xyz = src.transpose(1, 2).contiguous()
    d = O.sq_dist_expanded(xyz, xyz)[0]
    for k in (20, 5):
        ref = torch.topk(d, k, dim=-1, largest=False, sorted=True)[1]
        got = torch.from_numpy(run(host_lib, d.numpy(), k)).long()
        assert torch.equal(got.sort(-1)[0], ref.sort(-1)[0])
