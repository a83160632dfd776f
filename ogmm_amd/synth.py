"""Synthetic "ModelNet-shaped" registration pairs and a closed-form, RNG-free parameter fill.

There is no dataset and no checkpoint on the GPU box, so the bench, the smoke test and the parity
tests all use (a) point-cloud pairs generated here the way the reference's loaders shape theirs and
(b) GMMReg weights produced by an integer hash of (tensor name, element index), scaled like
PyTorch's default Kaiming-uniform init, with non-trivial BatchNorm running statistics so that
eval-mode BN is exercised.

Reference shapes being mimicked:
  * clouds are unit-sphere normalised shapes (datasets/datautils.py:146-159);
  * tgt = R_gt * P + t_gt with per-axis euler angles <= rot_mag (45 deg) and t in U(-0.5, 0.5)^3
    (datasets/transforms.py:152-190, configs/cfgs.py:18-19);
  * "clean": the same points in independent orders (datasets/modelnet.py:45-56);
  * "partial": an independent half-space crop keeping 70 % of each cloud
    (datasets/transforms.py:428-453), resampling to N, jitter clip(N(0, 0.01), +-0.05)
    (datasets/transforms.py:402-415), shuffle.
"""
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


# ------------------------------------------------------------------ point clouds
def _patch_cloud(rng, n):
    """n points on a random mixture of 4-8 planar / cylindrical patches, centred, unit sphere."""
    n_patch = int(rng.integers(4, 9))
    counts = rng.multinomial(n, rng.dirichlet(np.full(n_patch, 2.0)))
    out = []
    for c in counts:
        centre = rng.uniform(-0.5, 0.5, 3)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if rng.integers(0, 2) == 0:
            uv = rng.uniform(-1.0, 1.0, (c, 2)) * rng.uniform(0.2, 0.6, 2)
            p = centre + uv[:, :1] * q[:, 0] + uv[:, 1:] * q[:, 1]
        else:
            r, h = rng.uniform(0.1, 0.35), rng.uniform(0.2, 0.7)
            th = rng.uniform(0.0, 2.0 * np.pi * rng.uniform(0.3, 1.0), c)
            z = rng.uniform(-h, h, c)
            p = centre + (r * np.cos(th))[:, None] * q[:, 0] + (r * np.sin(th))[:, None] * q[:, 1] + z[:, None] * q[:, 2]
        out.append(p)
    p = np.concatenate(out, 0)
    p = p - p.mean(0)
    return p / np.linalg.norm(p, axis=1).max()


def _room_cloud(rng, n):
    """ICL-NUIM-like: n points on 3-6 large axis-aligned room planes (datasets/realdata.py:152-178)."""
    n_plane = int(rng.integers(3, 7))
    counts = rng.multinomial(n, np.full(n_plane, 1.0 / n_plane))
    out = []
    for i, c in enumerate(counts):
        ax = i % 3
        p = rng.uniform(-1.0, 1.0, (c, 3))
        p[:, ax] = rng.choice([-1.0, 1.0]) * rng.uniform(0.6, 1.0)
        out.append(p)
    p = np.concatenate(out, 0)
    p = p - p.mean(0)
    return p / np.linalg.norm(p, axis=1).max()


def _euler_rotation(rng, max_deg):
    a = np.deg2rad(rng.uniform(-max_deg, max_deg, 3))
    cx, cy, cz = np.cos(a)
    sx, sy, sz = np.sin(a)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _crop(rng, p, keep):
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    proj = p @ d
    return p[proj >= np.quantile(proj, 1.0 - keep)]


def _resample(rng, p, n):
    if p.shape[0] >= n:
        return p[rng.choice(p.shape[0], n, replace=False)]
    extra = rng.choice(p.shape[0], n - p.shape[0], replace=True)
    return np.concatenate([p, p[extra]], 0)


def make_pair(pair_idx, n_points, kind="partial", seed=1234, rot_mag=45.0, trans_mag=0.5):
    """One pair, a pure function of (seed + pair_idx): src, tgt [3,N] float32, R_gt [3,3], t_gt [3]."""
    rng = np.random.Generator(np.random.PCG64(seed + int(pair_idx)))
    if kind == "clean":
        base = _patch_cloud(rng, n_points)
        src, tgt = base[rng.permutation(n_points)], base[rng.permutation(n_points)]
    elif kind in ("partial", "room"):
        n_base = int(np.ceil(n_points / 0.7)) + 8
        base = _patch_cloud(rng, n_base) if kind == "partial" else _room_cloud(rng, n_base)
        src = _resample(rng, _crop(rng, base, 0.7), n_points)
        tgt = _resample(rng, _crop(rng, base, 0.7), n_points)
        src = src + np.clip(rng.normal(0.0, 0.01, src.shape), -0.05, 0.05)
        tgt = tgt + np.clip(rng.normal(0.0, 0.01, tgt.shape), -0.05, 0.05)
        src, tgt = src[rng.permutation(n_points)], tgt[rng.permutation(n_points)]
    else:
        raise ValueError("unknown pair kind %r" % kind)
    R = _euler_rotation(rng, rot_mag)
    t = rng.uniform(-trans_mag, trans_mag, 3)
    tgt = tgt @ R.T + t
    return (src.T.astype(np.float32), tgt.T.astype(np.float32), R.astype(np.float32), t.astype(np.float32))


def make_batch(first_pair, batch, n_points, kind="partial", seed=1234):
    """Pairs [first_pair, first_pair+batch) as torch tensors: src, tgt [B,3,N], R_gt [B,3,3], t_gt [B,3].
    Indexing by GLOBAL pair id keeps a sharded run bit-identical to the single-GPU run."""
    parts = [make_pair(first_pair + i, n_points, kind, seed) for i in range(batch)]
    return tuple(torch.from_numpy(np.stack([p[j] for p in parts])) for j in range(4))


def overlap_labels(src, tgt, R, t, thresh=0.05):
    """Ground-truth overlap labels of one pair the way the reference's loader derives them (lib/o3dutils.py:217-226 <-
    datasets/modelnet.py:212): a src point is 1 when some tgt point lies within `thresh` of it after the ground-truth
    motion, and symmetrically for tgt.  src, tgt [3,N] -> two float32 [N] arrays.  Brute force in fp64 (N <= a few k)."""
    moved = src.T.astype(np.float64) @ np.asarray(R, np.float64).T + np.asarray(t, np.float64)
    d2 = ((moved[:, None, :] - tgt.T.astype(np.float64)[None, :, :]) ** 2).sum(-1)
    hit = d2 < thresh * thresh
    return hit.any(1).astype(np.float32), hit.any(0).astype(np.float32)


def make_train_batch(first_pair, batch, n_points, kind="partial", seed=1234, thresh=0.05):
    """Training inputs of train.py:38-56 for pairs [first_pair, first_pair+batch): src, tgt [B,3,N], transform_gt
    [B,4,4], src_overlap, tgt_overlap [B,N]."""
    src, tgt, R, t = make_batch(first_pair, batch, n_points, kind, seed)
    T = torch.eye(4).repeat(batch, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    labels = [overlap_labels(src[i].numpy(), tgt[i].numpy(), R[i].numpy(), t[i].numpy(), thresh) for i in range(batch)]
    so = torch.from_numpy(np.stack([l[0] for l in labels]))
    to = torch.from_numpy(np.stack([l[1] for l in labels]))
    return src, tgt, T, so, to


def fps_starts_for(first_pair, batch, n_points, seed=1234):
    """The six random FPS start indices per pair (lib/utils.py:190 draws them from the global
    generator; here they are a pure function of the global pair id) -> int64 [6,B]."""
    out = np.empty((6, batch), dtype=np.int64)
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64(seed * 7919 + first_pair + i))
        out[:, i] = rng.integers(0, n_points, 6)
    return torch.from_numpy(out)


# ------------------------------------------------------------------ parameters
def _hash_uniform(name, numel):
    """numel floats in [0,1): splitmix64 of (crc32(name) << 32 | i), top 24 bits."""
    with np.errstate(over="ignore"):
        x = (np.uint64(zlib.crc32(name.encode())) << np.uint64(32)) + np.arange(numel, dtype=np.uint64)
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


# profile "sharp" (round 4): the default fill leaves the network in a degenerate regime -- attention logits within +-0.03 of each other (a uniform softmax
# over the 128 anchors), overlap scores within 0.493 ... 0.500 -- in which rounding of the Q / K / score operands cannot matter.  A trained checkpoint is
# not like that.  The sharp family keeps the closed form and multiplies chosen tensors by constants (found by experiment on the CPU oracle:
# tools/weight_profile_stats.py prints what they achieve) so that (i) every transformer's logits span >= +-10 (peaked softmax: mean max probability
# 0.3 ... 0.8), (ii) the overlap scores span (0.01, 0.99), (iii) BatchNorm running statistics and affine parameters vary by 8x in variance.
SHARP_GAINS = {}        # filled below, after the function that uses it (kept next to the numbers' provenance)


def fill_state_dict(state_dict, profile="default"):
    """Overwrites every entry of a GMMReg state_dict (ours or the reference's: same 153 keys)
    in place with the closed-form fill.  conv weights/biases: U(-b, b), b = 1/sqrt(fan_in);
    BN weight U(0.5,1.5), bias U(-0.2,0.2), running_mean U(-0.2,0.2), running_var U(0.5,1.5).
    profile="sharp": the same hash with wider BatchNorm ranges (weight U(0.4,2.0), bias U(-0.5,0.5), running_mean U(-0.5,0.5), running_var
    U(0.25,2.0)) and the per-tensor gains of SHARP_GAINS (peaked attention, saturated overlap scores)."""
    if profile not in ("default", "sharp", "mid"):
        raise ValueError("unknown weight profile %r" % profile)
    sharp = profile in ("sharp", "mid")
    for name, t in state_dict.items():
        if name.endswith("num_batches_tracked"):
            t.zero_()
            continue
        u = _hash_uniform(name, t.numel()).reshape(tuple(t.shape))
        leaf = name.rsplit(".", 1)[1]
        is_norm = t.dim() == 1 and (name + "x").replace(leaf + "x", "running_var") in state_dict
        if is_norm and sharp:
            v = {"weight": 0.4 + 1.6 * u, "bias": u - 0.5, "running_mean": u - 0.5, "running_var": 0.25 + 1.75 * u}[leaf]
        elif is_norm:
            v = {"weight": 0.5 + u, "bias": 0.4 * u - 0.2, "running_mean": 0.4 * u - 0.2, "running_var": 0.5 + u}[leaf]
        else:
            if t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
            else:  # conv bias: fan_in of the sibling weight
                w = state_dict[name.rsplit(".", 1)[0] + ".weight"]
                fan_in = int(np.prod(w.shape[1:]))
            b = 1.0 / np.sqrt(fan_in)
            v = (2.0 * u - 1.0) * b
        if sharp:
            g = SHARP_GAINS.get(name)
            if g is not None:
                # "mid": the square root of every gain and a quarter of the offset -- half-way (in the logarithm) between the two families; the family of
                # the TRAIN-mode fixture: with batch statistics the full gains saturate every overlap score to 0 / 1 and the reference's own train-mode
                # forward is then only reproducible to 1e-4 ... 1e-3 (tests/golden/make_golden_train.py records that noise)
                v = v * g[0] + g[1] if profile == "sharp" else v * (g[0] ** 0.5) + 0.25 * g[1]
        t.copy_(torch.from_numpy(np.ascontiguousarray(v)).to(t.dtype))
    return state_dict


def _sharp_gains():
    g = {}
    for tr, gq in (("sattn1", 25.0), ("cattn", 23.0), ("sattn2", 8.5)):
        for i in (0, 1):                       # Q and K projections (models/attn.py:96-97): logits scale with the product of the two gains
            g["%s.attn.proj.%d.weight" % (tr, i)] = (gq, 0.0)
            g["%s.attn.proj.%d.bias" % (tr, i)] = (gq, 0.0)
    # the chain behind the overlap scores (models/gmmreg.py:83-89): BatchNorm gains of overlap.net and its last convolution, bias re-centred
    g["overlap.net.1.weight"] = (3.0, 0.0)
    g["overlap.net.4.weight"] = (3.0, 0.0)
    g["overlap.net.6.weight"] = (20.0, 0.0)
    g["overlap.net.6.bias"] = (1.0, 7.0)
    return g


SHARP_GAINS.update(_sharp_gains())
