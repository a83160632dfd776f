"""Device-side sample pipeline (ogmm_amd/augment.py): each step against the reference's own transform with the same
random numbers (build container only, the reference is imported), the whole chain through its defining properties."""
import numpy as np
import pytest
import torch

from ogmm_amd import augment, synth
from oracle import ref_harness

needs_ref = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference not mounted (GPU box)")


def _cloud(B, P, seed=0):
    return torch.stack([torch.from_numpy(synth._patch_cloud(np.random.Generator(np.random.PCG64(seed + i)), P)).float() for i in range(B)])


@needs_ref
def test_euler_transform_matches_reference():
    ref_harness.import_reference()
    import datasets.transforms as T
    tr = T.RandomTransformSE3_euler(rot_mag=45.0, trans_mag=0.5)
    rng = np.random.default_rng(2)
    u = rng.uniform(size=(5, 3))
    tvec = rng.uniform(-0.5, 0.5, size=(5, 3))
    real_uniform = np.random.uniform
    for b in range(5):
        seq = iter([u[b, 0], u[b, 1], u[b, 2]])
        np.random.uniform = lambda *a, **k: next(seq) if not a else tvec[b]          # 3 scalar draws, then the translation vector
        try:
            M = tr.generate_transform()
        finally:
            np.random.uniform = real_uniform
        R = augment.euler_to_matrix(torch.from_numpy(u[b:b + 1] * np.pi * 45.0 / 180.0))[0].numpy()
        assert np.abs(R - M[:, :3]).max() < 1e-6 and np.abs(tvec[b] - M[:, 3]).max() < 1e-6


@needs_ref
def test_crop_mask_matches_reference():
    ref_harness.import_reference()
    import datasets.transforms as T
    pts = _cloud(4, 1024)
    d = augment.draw(4, 1024, 717, torch.Generator().manual_seed(1))
    real = T.uniform2sphere
    for b in range(4):
        T.uniform2sphere = lambda num=None: d["crop_dir_src"][b].double().numpy()
        try:
            kept, mask = T.RandomCrop.crop(pts[b].double().numpy(), 0.7)
        finally:
            T.uniform2sphere = real
        mine = augment.crop_mask(pts[b:b + 1].double(), d["crop_dir_src"][b:b + 1].double(), 0.7)[0].numpy()
        assert np.array_equal(mine, mask) and kept.shape[0] == mask.sum()


def test_pipeline_properties():
    B, P, n_out = 6, 1024, 717
    pts = _cloud(B, P, seed=40)
    d = augment.draw(B, P, n_out, torch.Generator().manual_seed(7))
    out = augment.crop_pipeline(pts, d, n_out=n_out)
    m_src, m_ref = augment.crop_mask(pts, d["crop_dir_src"]), augment.crop_mask(pts, d["crop_dir_ref"])
    R, t = augment.euler_to_matrix(d["euler_xyz"]), d["translation"]
    for b in range(B):
        si, ri = out["src_index"][b], out["tgt_index"][b]
        assert m_src[b][si].all() and m_ref[b][ri].all()                      # only points of the kept half-space
        n_s, n_r = int(m_src[b].sum()), int(m_ref[b].sum())
        assert abs(n_s - 0.7 * P) <= 2 and abs(n_r - 0.7 * P) <= 2
        assert len(set(si.tolist())) == min(n_out, n_s)                        # no repetition unless the crop is smaller than n_out
        if n_s < n_out:
            assert set(si.tolist()) == set(torch.nonzero(m_src[b]).flatten().tolist())      # then every kept point appears
        # undo jitter bound: the moved-back source lies within the jitter clip of its raw point
        back = (out["src_xyz"][b] @ out["transform_gt"][b, :3, :3].T + out["transform_gt"][b, :3, 3])
        assert (back - pts[b][si]).abs().max() < 0.05 * 3 ** 0.5 + 1e-5
        assert (out["tgt_xyz"][b] - pts[b][ri]).abs().max() <= 0.05 + 1e-6
        assert torch.equal(out["src_overlap"][b], m_ref[b][si].float()) and torch.equal(out["tgt_overlap"][b], m_src[b][ri].float())
        Tm = torch.eye(4)
        Tm[:3, :3], Tm[:3, 3] = R[b], t[b]
        assert torch.allclose(out["transform_gt"][b] @ Tm, torch.eye(4), atol=1e-5)
    assert (d["euler_xyz"] >= 0).all() and (d["euler_xyz"] <= np.pi / 4 + 1e-6).all() and d["translation"].abs().max() <= 0.5
    assert torch.allclose(d["crop_dir_src"].norm(dim=1), torch.ones(B), atol=1e-5)


def test_resample_small_crops_repeat_every_point_first():
    mask = torch.zeros(2, 50, dtype=torch.bool)
    mask[0, :10] = True
    mask[1, 5:45] = True
    g = torch.Generator().manual_seed(3)
    idx = augment.resample_indices(mask, 20, torch.rand(2, 50, generator=g), torch.rand(2, 20, generator=g))
    assert set(idx[0, :10].tolist()) == set(range(10)) and all(0 <= i < 10 for i in idx[0].tolist())
    assert len(set(idx[1].tolist())) == 20 and all(5 <= i < 45 for i in idx[1].tolist())


@pytest.mark.gpu
def test_overlap_labels_on_the_gpu_match_brute_force():
    B, N = 3, 600
    src, tgt, T, so, to = synth.make_train_batch(70, B, N, "partial")
    a, b = augment.overlap_labels(src.transpose(1, 2).contiguous().cuda(), tgt.transpose(1, 2).contiguous().cuda(), T.cuda(), 0.05)
    # fp32 distances against the fp64 brute force of synth.overlap_labels: only points within rounding of the radius may differ
    assert (a.cpu() != so).float().mean() < 2e-3 and (b.cpu() != to).float().mean() < 2e-3


def _chain_fixture():
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_crop_chain.npz"))
    for g in range(3):
        raw = torch.from_numpy(fx["g%d/raw" % g])
        draws = {k.split("/")[-1]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("g%d/draw/" % g)}
        want = {k.split("/")[-1]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("g%d/out/" % g)}
        yield g, raw, draws, want


def _check_chain(device):
    """The whole chain against what the reference's own transforms made of the same clouds and the same random numbers
    (tests/golden/make_golden_augment.py): labels and point identities exact, coordinates to fp32 rounding."""
    for g, raw, draws, want in _chain_fixture():
        out = augment.crop_pipeline(raw.double().to(device), {k: v.double().to(device) for k, v in draws.items()}, n_out=717)
        out = {k: v.cpu() for k, v in out.items()}
        assert torch.equal(out["src_overlap"], want["src_overlap"]) and torch.equal(out["tgt_overlap"], want["tgt_overlap"]), g
        assert (out["src_xyz"] - want["src_xyz"]).abs().max().item() < 1e-6, g
        assert (out["tgt_xyz"] - want["tgt_xyz"]).abs().max().item() < 1e-6, g
        assert (out["transform_gt"][:, :3] - want["transform_gt"]).abs().max().item() < 1e-6, g
        # every output point is its raw point moved and jittered: identities follow from the coordinates being equal, and the
        # small-crop group repeats points exactly where the reference does
        if g == 1:
            assert all(len(set(out["src_index"][b].tolist())) == int(want["n_kept"][b, 0]) for b in range(2))


def test_whole_chain_matches_the_reference_transforms():
    _check_chain("cpu")


@pytest.mark.gpu
def test_whole_chain_matches_the_reference_transforms_on_the_gpu():
    _check_chain("cuda")
