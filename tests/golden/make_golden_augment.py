"""Generates tests/golden/augment_crop_chain.npz: the reference's own "crop" sample chain (datasets/modelnet.py:75-80 --
SplitSourceRef, RandomCrop, RandomTransformSE3_euler, Resampler, RandomJitter, ShufflePoints of datasets/transforms.py) run
here on CPU with numpy's generator seeded, every random number it draws recorded on the way, and the recorded numbers
re-expressed as the explicit draw tensors ogmm_amd/augment.py takes (sorting keys instead of index lists).  The fixture is
data: raw clouds, draws, and what the reference's chain made of them (points, overlap labels, transform_gt).

Three groups of two clouds: 1024 raw points (the dataset's size: the 70 % crop leaves 716-717 points for the 717-point
resampler), 900 (crop smaller than 717: every point once + repeats, datasets/transforms.py:324-325) and 1200 (crop larger:
sampling without repetition, :320)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_harness import import_reference  # noqa: E402
from ogmm_amd import synth                       # noqa: E402

N_OUT = 717


class Recorder:
    """Wraps one numpy.random function: calls through and keeps (args, result)."""

    def __init__(self, owner, name):
        self.owner, self.name, self.real, self.calls = owner, name, getattr(owner, name), []

    def __enter__(self):
        def wrapped(*a, **k):
            out = self.real(*a, **k)
            self.calls.append((a, k, np.array(out, copy=True)))
            return out
        setattr(self.owner, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.owner, self.name, self.real)


def keys_from_order(order, size):
    """Sorting keys in (0, 1) whose argsort starts with `order` (the remaining positions follow in index order)."""
    key = np.empty(size, dtype=np.float64)
    rest = np.setdiff1d(np.arange(size), order)
    key[np.concatenate([order, rest])] = (np.arange(size) + 0.5) / size
    return key


def one_sample(T, raw, seed):
    """The reference's chain on one raw cloud; returns (draws of this sample as augment.py names them, outputs)."""
    chain = [T.SplitSourceRef(), T.RandomCrop([0.7, 0.7]), T.RandomTransformSE3_euler(rot_mag=45.0, trans_mag=0.5), T.Resampler(1024),
             T.RandomJitter(), T.ShufflePoints()]
    P = raw.shape[0]
    np.random.seed(seed)
    sample = {"points": raw.copy(), "idx": np.array(seed, dtype=np.int32)}
    with Recorder(T, "uniform2sphere") as sph, Recorder(np.random, "uniform") as uni, Recorder(np.random, "choice") as cho, \
            Recorder(np.random, "normal") as nor, Recorder(np.random, "permutation") as per:
        masks = []
        real_crop = T.RandomCrop.crop

        def crop(points, p_keep):
            out, mask = real_crop(points, p_keep)
            masks.append(mask.copy())
            return out, mask
        T.RandomCrop.crop = staticmethod(crop)
        try:
            for tr in chain:
                sample = tr(sample)
        finally:
            T.RandomCrop.crop = staticmethod(real_crop)
    assert len(sph.calls) == 2 and len(nor.calls) == 2 and len(per.calls) == 2 and len(masks) == 2
    # uniform: 2 x (phi, cos theta) inside uniform2sphere, three Euler draws, one translation vector
    assert len(uni.calls) == 8 and uni.calls[7][2].shape == (3,)
    d = {"crop_dir_src": sph.calls[0][2], "crop_dir_ref": sph.calls[1][2],
         "euler_xyz": np.array([uni.calls[i][2] for i in (4, 5, 6)], dtype=np.float64) * np.pi * 45.0 / 180.0,
         "translation": uni.calls[7][2].astype(np.float64),
         "jitter_src": nor.calls[0][2], "jitter_ref": nor.calls[1][2]}
    # Resampler: source first, then reference; each either one draw without repetition or (a permutation, repeats)
    calls = list(cho.calls)
    for side, mask in (("src", masks[0]), ("ref", masks[1])):
        kept = np.nonzero(mask)[0]                                  # cropped index -> raw index
        n = kept.shape[0]
        first = calls.pop(0)
        assert first[0][0] == n and first[1].get("replace") is False
        if N_OUT <= n:
            assert first[0][1] == N_OUT
            order, extra = first[2], np.full(N_OUT, 0.5)
        else:
            assert first[0][1] == n
            rep = calls.pop(0)
            assert rep[0] == (n, N_OUT - n) and rep[1].get("replace") is True
            order = first[2]
            where = np.empty(n, dtype=np.int64)
            where[order] = np.arange(n)                             # position of a cropped index in the random order
            extra = np.full(N_OUT, 0.5)
            extra[n:] = (where[rep[2]] + 0.5) / n                   # augment.resample_indices: floor(extra * n) = that position
        d["resample_key_" + side] = keys_from_order(kept[order], P)
        d["resample_extra_" + side] = extra
    assert not calls
    # ShufflePoints draws the reference's permutation first (datasets/transforms.py:507-508)
    for side, call in (("ref", per.calls[0]), ("src", per.calls[1])):
        d["shuffle_key_" + side] = keys_from_order(call[2], N_OUT)
    out = {"src_xyz": sample["points_src"][:, :3], "tgt_xyz": sample["points_ref"][:, :3], "transform_gt": sample["transform_gt"],
           "src_overlap": sample["src_overlap"].astype(np.float32), "tgt_overlap": sample["ref_overlap"].astype(np.float32),
           "n_kept": np.array([masks[0].sum(), masks[1].sum()])}
    assert out["src_xyz"].shape == (N_OUT, 3) and out["tgt_xyz"].shape == (N_OUT, 3)
    return d, out


def main():
    import_reference()
    import datasets.transforms as T
    fx = {}
    for g, (P, seed0) in enumerate(((1024, 500), (900, 510), (1200, 520))):
        raws, draws, outs = [], [], []
        for b in range(2):
            raw = synth._patch_cloud(np.random.Generator(np.random.PCG64(seed0 + b)), P).astype(np.float32)
            d, o = one_sample(T, raw, seed0 + b)
            raws.append(raw), draws.append(d), outs.append(o)
        fx["g%d/raw" % g] = np.stack(raws)
        for k in draws[0]:
            fx["g%d/draw/%s" % (g, k)] = np.stack([d[k] for d in draws])
        for k in outs[0]:
            fx["g%d/out/%s" % (g, k)] = np.stack([o[k] for o in outs])
        print("group %d: P = %d, kept %s" % (g, P, fx["g%d/out/n_kept" % g].tolist()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "augment_crop_chain.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
