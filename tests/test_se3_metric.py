"""a21 / a23: SE(3) pack/unpack and the parity metrics, against the reference when it is mounted
(build container) and against closed-form cases everywhere."""
import math

import numpy as np
import pytest
import torch

from ogmm_amd import metric, se3
from oracle import ref_harness


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * K @ K


def test_pack_unpack_round_trip():
    g = torch.Generator().manual_seed(3)
    R = torch.randn(5, 3, 3, generator=g)
    t = torch.randn(5, 3, 1, generator=g)
    T = se3.integrate_trans(R, t)
    assert T.shape == (5, 4, 4) and T.dtype == torch.float32
    assert torch.equal(T[:, 3], torch.tensor([0.0, 0, 0, 1]).expand(5, 4))
    R2, t2 = se3.decompose_trans(T)
    assert torch.equal(R2, R) and torch.equal(t2, t)
    T1 = se3.integrate_trans(R[0], t[0])
    assert torch.equal(T1, T[0])
    Tn = se3.integrate_trans(R.numpy(), t.numpy())
    assert Tn.dtype == np.float64 and np.array_equal(Tn.astype(np.float32), T.numpy())
    Rn, tn = se3.decompose_trans(Tn[0])
    assert Rn.shape == (3, 3) and tn.shape == (3, 1)
    assert torch.equal(se3.integrate_trans(R, t[:, :, 0]), T)          # [B,3] translations as GMMReg returns them
    with pytest.raises(ValueError):
        se3.decompose_trans(torch.zeros(3, 3))


def test_metrics_closed_form():
    angles = [0.0, 1e-6, 1e-4, 0.3, 3.0]
    R1 = torch.tensor(np.stack([_rot([1, 2, 3], 0.7)] * len(angles)))
    R2 = torch.tensor(np.stack([_rot([1, 2, 3], 0.7) @ _rot([0.2, -1, 0.5], a) for a in angles]))
    rad = metric.rotation_error_rad(R1.float(), R2.float())
    for a, r in zip(angles, rad.tolist()):
        assert abs(r - a) < 2e-7 + 1e-6 * a
    deg = metric.rotation_error(R1.float(), R2.float())
    assert abs(deg[3].item() - math.degrees(0.3)) < 1e-3
    t1 = torch.tensor([[0.0, 0, 0], [1, 2, 3]])
    t2 = torch.tensor([[3.0, 4, 0], [1, 2, 3]])
    assert metric.translation_error(t1, t2).tolist() == [5.0, 0.0]
    with pytest.raises(ValueError):
        metric.rotation_error(R1[:2], R2[:3])


@pytest.mark.skipif(not ref_harness.reference_available(), reason="reference not mounted (GPU box)")
def test_against_reference():
    ref_harness.import_reference()
    import lib.metric as ref_metric
    import lib.se3 as ref_se3
    g = torch.Generator().manual_seed(11)
    R1 = torch.linalg.qr(torch.randn(6, 3, 3, generator=g))[0]
    R2 = torch.linalg.qr(torch.randn(6, 3, 3, generator=g))[0]
    t1, t2 = torch.randn(6, 3, generator=g), torch.randn(6, 3, generator=g)
    assert torch.equal(metric.rotation_error(R1, R2), ref_metric.rotation_error(R1, R2))
    assert torch.equal(metric.translation_error(t1, t2), ref_metric.translation_error(t1, t2))
    T = se3.integrate_trans(R1, t1[:, :, None])
    assert torch.equal(T, ref_se3.integrate_trans(R1, t1[:, :, None]))
    assert torch.equal(se3.integrate_trans(R1[0], t1[0, :, None]), ref_se3.integrate_trans(R1[0], t1[0, :, None]))
    for a, b in zip(se3.decompose_trans(T), ref_se3.decompose_trans(T)):
        assert torch.equal(a, b)


def test_euler_zyx_matches_scipy():
    from scipy.spatial.transform import Rotation
    R = Rotation.random(200, random_state=4).as_matrix()
    want = Rotation.from_matrix(R).as_euler("zyx", degrees=True)
    got = metric.euler_zyx_deg(torch.from_numpy(R)).numpy()
    assert np.abs(got - want).max() < 1e-9


def test_summarize_metrics_matches_reference_fixture():
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_b4_n300.npz"))
    m = {k[2:]: torch.from_numpy(np.asarray(fx[k])) for k in fx.files if k.startswith("m/")}
    got = metric.summarize_metrics(m)
    want = {k[2:]: float(fx[k]) for k in fx.files if k.startswith("s/")}
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), k
