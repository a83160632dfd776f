"""oracle/icp_oracle.py (numpy fp64 restatement of open3d's published point-to-point ICP; parity unpinned because open3d
is absent) validated the only way available: it must recover the ground-truth motion on clean pairs."""
import numpy as np

from ogmm_amd import synth
from oracle import icp_oracle


def _perturbed(R, t, angle, shift, seed):
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    dR = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = dR @ R, t + shift * rng.normal(size=3)
    return T


def test_recovers_ground_truth_on_clean_pairs():
    for pair in (0, 1, 2):
        src, tgt, R, t = synth.make_pair(pair, 400, "clean")
        T0 = _perturbed(R.astype(np.float64), t.astype(np.float64), 0.05, 0.01, pair)
        T, fit, rmse, it = icp_oracle.icp_point_to_point(src.T, tgt.T, T0, 0.07)
        assert fit == 1.0 and rmse < 1e-6 and it < 30
        assert np.abs(T[:3, :3] - R).max() < 1e-6 and np.abs(T[:3, 3] - t).max() < 1e-6


def test_no_correspondence_returns_initial_motion():
    src, tgt, R, t = synth.make_pair(5, 100, "clean")
    T0 = np.eye(4)
    T0[:3, 3] = 50.0
    T, fit, rmse, it = icp_oracle.icp_point_to_point(src.T, tgt.T, T0, 0.07)
    assert fit == 0.0 and rmse == 0.0 and np.array_equal(T, T0) and it == 1
