"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *real* reference (gfmei/ogmm, mounted read-only at /root/reference) on CPU so that
the restatement in oracle/ogmm_oracle.py can be pinned against it and golden vectors can be
generated (tests/golden/make_golden.py).  Only usable inside the build container: the reference
does not travel to the GPU box in any form.

Two third-party modules that the reference imports at module scope but never touches on the
hot path (`open3d` via lib/o3dutils.py:11, `transforms3d` via lib/se3.py:10) are absent from
this image; empty stub modules are registered for them.  `h5py` (file loader only,
datasets/datautils.py:15) is stubbed the same way so that lib/metric.py and lib/loss.py import.
"""
import os
import sys
import types
from argparse import Namespace

REFERENCE_ROOT = os.environ.get("OGMM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "gmmreg.py"))


def import_reference():
    """Returns the reference's `models.gmmreg` module (and makes `lib.*` importable)."""
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    for name in ("open3d", "transforms3d", "transforms3d.quaternions", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["transforms3d"].quaternions = sys.modules["transforms3d.quaternions"]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models.gmmreg as ref_gmmreg  # noqa: E402
    return ref_gmmreg


def default_config(**kw):
    """The four attributes GMMReg reads from `config` (models/gmmreg.py:35,44-46,55,117),
    with the defaults of configs/cfgs.py:24,33,36,39."""
    cfg = dict(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
    cfg.update(kw)
    return Namespace(**cfg)
