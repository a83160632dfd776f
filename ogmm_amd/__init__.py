"""ogmm_amd: MI355X-native (gfx950) hot path of gfmei/ogmm -- drop-in `GMMReg` on hand-written HIP kernels."""
from .gmmreg import GMMReg  # noqa: F401
