"""Device-side version of the reference's training-sample pipeline: the "crop" transform chain of
datasets/modelnet.py:73-80 -- SplitSourceRef, RandomCrop, RandomTransformSE3_euler, Resampler, RandomJitter, ShufflePoints
(datasets/transforms.py) -- for a whole batch at once, on the GPU, plus the radius-based overlap labels of
lib/o3dutils.py:217-226 on the nearest-distance kernel.  The reference runs this per sample in six DataLoader workers (numpy,
KD-trees); at kHz pair rates that starves the GPUs.

Every step takes its random numbers as explicit tensors (`draw(...)` makes them from a torch generator), so each step can be
checked against the reference's own transform with the same numbers (tests/test_augment.py); whole-pipeline draws cannot be
numpy-identical (numpy's global MT19937 stream), only identically distributed.
"""
import math

import torch


def draw(batch, n_points, n_out, generator=None, device="cpu", rot_mag=45.0, trans_mag=0.5, jitter_scale=0.01):
    """The random numbers of one batch: crop directions (uniform on the sphere: datasets/datautils.py:93-114), Euler angles
    U(0,1) * pi * rot_mag / 180 per axis and translation U(-trans_mag, trans_mag)^3 (datasets/transforms.py:152-190; the angles
    are non-negative, as there), resampling / shuffling keys, jitter noise N(0, scale) (datasets/transforms.py:402-415)."""
    def rand(*shape):
        return torch.rand(*shape, generator=generator, device=device)
    d = {}
    for name in ("crop_dir_src", "crop_dir_ref"):
        phi, cos_t = rand(batch) * 2 * math.pi, rand(batch) * 2 - 1
        sin_t = torch.sqrt((1 - cos_t * cos_t).clamp_min(0))
        d[name] = torch.stack([sin_t * torch.cos(phi), sin_t * torch.sin(phi), cos_t], dim=1)
    d["euler_xyz"] = rand(batch, 3) * math.pi * rot_mag / 180.0
    d["translation"] = (rand(batch, 3) * 2 - 1) * trans_mag
    for name in ("resample_key_src", "resample_key_ref"):
        d[name] = rand(batch, n_points)
        d[name.replace("key", "extra")] = rand(batch, n_out)
    for name in ("jitter_src", "jitter_ref"):
        d[name] = torch.randn(batch, n_out, 3, generator=generator, device=device) * jitter_scale
    for name in ("shuffle_key_src", "shuffle_key_ref"):
        d[name] = rand(batch, n_out)
    return d


def euler_to_matrix(angles_xyz):
    """R = Rx @ Ry @ Rz (datasets/transforms.py:166-186).  angles [B,3] -> [B,3,3]"""
    ax, ay, az = angles_xyz[:, 0], angles_xyz[:, 1], angles_xyz[:, 2]
    cx, cy, cz, sx, sy, sz = torch.cos(ax), torch.cos(ay), torch.cos(az), torch.sin(ax), torch.sin(ay), torch.sin(az)
    one, zero = torch.ones_like(ax), torch.zeros_like(ax)
    Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).view(-1, 3, 3)
    Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).view(-1, 3, 3)
    Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).view(-1, 3, 3)
    return Rx @ Ry @ Rz


def crop_mask(points, direction, p_keep=0.7):
    """`RandomCrop.crop` (datasets/transforms.py:441-453): keep the points whose offset from the centroid, projected on `direction`,
    exceeds the (1 - p_keep) percentile (linear interpolation, as numpy); p_keep == 0.5 keeps the positive half-space.
    points [B,P,3], direction [B,3] -> bool [B,P]"""
    centred = points - points.mean(dim=1, keepdim=True)
    dist = (centred * direction[:, None, :]).sum(dim=2)
    if p_keep == 0.5:
        return dist > 0
    thr = torch.quantile(dist, 1.0 - p_keep, dim=1, keepdim=True, interpolation="linear")
    return dist > thr


def resample_indices(mask, n_out, key, extra):
    """`Resampler._resample` (datasets/transforms.py:310-327) on the masked points of every cloud: n_out indices into [0,P);
    a cloud with at least n_out kept points is sampled without repetition, a smaller one contributes every kept point once and
    the rest with repetition.  mask bool [B,P]; key [B,P], extra [B,n_out] uniform(0,1) -> int64 [B,n_out]"""
    B, P = mask.shape
    order = torch.argsort(torch.where(mask, key, key + 2.0), dim=1)          # kept points first, in random order
    n_kept = mask.sum(dim=1, keepdim=True)                                    # [B,1]
    pos = torch.arange(n_out, device=mask.device)[None, :].expand(B, -1)
    wrapped = (extra * n_kept).long().clamp_max(P - 1)                        # a uniformly random kept point, for positions >= n_kept
    pos = torch.where(pos < n_kept, pos, torch.minimum(wrapped, n_kept - 1))
    return torch.gather(order, 1, pos)


def crop_pipeline(points, draws, n_out=717, p_keep=0.7, jitter_clip=0.05):
    """The reference's "crop" training chain (datasets/modelnet.py:75-80) for a batch.  points [B,P,3] (the raw cloud; both copies
    start from it).  n_out = 717: the reference resamples both crops to 717 points whatever `num_points` says
    (datasets/transforms.py:343-345, kept "to be consistent with Predator").
    Returns dict: src_xyz, tgt_xyz [B,n_out,3], transform_gt [B,4,4] (maps src onto tgt), src_overlap, tgt_overlap [B,n_out]
    (1 where the point also survived the other cloud's crop: datasets/transforms.py:472-481), and the index maps into the raw cloud."""
    B, P, _ = points.shape
    m_src = crop_mask(points, draws["crop_dir_src"], p_keep)
    m_ref = crop_mask(points, draws["crop_dir_ref"], p_keep)
    # RandomTransformSE3_euler: the SOURCE is moved by T, transform_gt = T^-1 takes it back onto the reference (transforms.py:113-148)
    R = euler_to_matrix(draws["euler_xyz"])
    t = draws["translation"]
    i_src = resample_indices(m_src, n_out, draws["resample_key_src"], draws["resample_extra_src"])
    i_ref = resample_indices(m_ref, n_out, draws["resample_key_ref"], draws["resample_extra_ref"])
    take = lambda x, idx: torch.gather(x, 1, idx[:, :, None].expand(-1, -1, 3))          # noqa: E731
    src = take(points, i_src) @ R.transpose(1, 2) + t[:, None, :]
    ref = take(points, i_ref)
    src_overlap = torch.gather(m_ref, 1, i_src).float()          # the same raw point is also in the reference crop
    ref_overlap = torch.gather(m_src, 1, i_ref).float()
    src = src + draws["jitter_src"].clamp(-jitter_clip, jitter_clip)
    ref = ref + draws["jitter_ref"].clamp(-jitter_clip, jitter_clip)
    p_src, p_ref = torch.argsort(draws["shuffle_key_src"], dim=1), torch.argsort(draws["shuffle_key_ref"], dim=1)
    src, ref = take(src, p_src), take(ref, p_ref)
    src_overlap, ref_overlap = torch.gather(src_overlap, 1, p_src), torch.gather(ref_overlap, 1, p_ref)
    T = torch.eye(4, device=points.device, dtype=points.dtype).repeat(B, 1, 1)
    T[:, :3, :3] = R.transpose(1, 2)
    T[:, :3, 3] = -(R.transpose(1, 2) @ t[:, :, None])[:, :, 0]
    return {"src_xyz": src, "tgt_xyz": ref, "transform_gt": T, "src_overlap": src_overlap, "tgt_overlap": ref_overlap,
            "src_index": torch.gather(i_src, 1, p_src), "tgt_index": torch.gather(i_ref, 1, p_ref)}


def overlap_labels(src, tgt, transform_gt, thresh=0.05):
    """Radius-based labels (lib/o3dutils.py:217-226 <- datasets/modelnet.py:212): 1 where the other cloud has a point within `thresh`
    after moving src by transform_gt.  src, tgt [B,N,3] on the GPU -> two float [B,N] (nearest-distance kernel K21)."""
    from . import ops
    moved = torch.baddbmm(transform_gt[:, None, :3, 3], src, transform_gt[:, :3, :3].transpose(1, 2))
    t2 = thresh * thresh
    return (ops.min_sqdist(moved, tgt.contiguous()) < t2).float(), (ops.min_sqdist(tgt.contiguous(), moved) < t2).float()
