"""Shared helpers of the training-step parity tests (oracle vs fixtures, HIP path vs fixtures)."""
import os
from argparse import Namespace

import numpy as np
import torch

from ogmm_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN_CASES = ["train_b2_n512_j16", "train_b3_n320_j8_k12"]
SAMPLE = 97


def sample_idx(numel):
    """the strided entries of a flattened gradient that tests/golden/make_golden_train.py stored"""
    return np.unique(np.linspace(0, numel - 1, min(numel, SAMPLE)).astype(np.int64))


def load_train_case(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    B, N, J, k, M, D, H, top_k = (int(v) for v in fx["meta"])
    cfg = Namespace(gnn_k=k, num_heads=H, km_clusters=M, overlap_radius=0.035, n_clusters=J)
    return fx, cfg, (B, N, J, D, top_k)


def filled_state(module_or_spec):
    """closed-form weights of ogmm_amd/synth.py as a fresh dict of tensors"""
    return synth.fill_state_dict(module_or_spec)


def check_grads(fx, grads, factor=4.0, floor=3e-4, report=None):
    """grads: {key: tensor or None}.  Per parameter, the candidate's distance from the fp64 truth (relative L2 over the
    stored strided sample, and of the norm) must be <= factor * max(the reference's own fp32 distance for that parameter,
    the median of the reference's distances over all parameters) -- the reference's per-parameter distance is a single
    draw of rounding noise (median 4e-4..6e-4, max 5e-3 on the fixtures), so the median keeps a lucky draw from becoming
    an unreachable bar -- and never below `floor`.
    Parameters whose reference gradient is structurally zero (norm < 1e-6 of the total) must stay below 1e-5 of the
    total; parameters the forward never touches must have no gradient.  Returns the worst ratio error/allowed."""
    total = float(fx["gnorm_total"])
    worst = 0.0
    live = [float(fx[f]) for f in fx.files if f.startswith("gerr/") and float(fx["gnorm/" + f[5:]]) >= 1e-6 * total]
    typical = float(np.median(live))
    for key in (f[len("gnorm/"):] for f in fx.files if f.startswith("gnorm/")):
        ref_norm = float(fx["gnorm/" + key])
        g = grads.get(key)
        if ref_norm < 0:                                      # parameter the forward never touches
            assert g is None or float(g.abs().max()) == 0.0, key
            continue
        assert g is not None, "no gradient for " + key
        g = g.detach().cpu().reshape(-1).double().numpy()
        if ref_norm < 1e-6 * total:
            assert np.linalg.norm(g) < 1e-5 * total, (key, np.linalg.norm(g))
            continue
        truth, tnorm = fx["gsamp64/" + key], float(fx["gnorm64/" + key])
        got = g[sample_idx(g.size)]
        scale = max(np.linalg.norm(truth), tnorm * np.sqrt(len(truth) / g.size))
        err = max(np.linalg.norm(got - truth) / scale, abs(np.linalg.norm(g) - tnorm) / tnorm)
        allowed = max(factor * float(fx["gerr/" + key]), factor * typical, floor)
        if report is not None:
            report[key] = (err, allowed)
        worst = max(worst, err / allowed)
        assert err <= allowed, "%s: gradient error vs fp64 truth %.3e > allowed %.3e (reference's own: %.3e)" % (
            key, err, allowed, float(fx["gerr/" + key]))
    return worst
