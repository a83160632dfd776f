"""Tools-only: route `ogmm_gemm_nt` descriptors to libogmm_probe.so -- the second build of the GEMM engine sources that carries the ablation / clock-probe
instantiations (precision codes 12..40, 100..121), the row-major-planes engine (code 1) and the retired first LDS-DMA engine (codes 60..89) -- none of
which are in the product library.  `install()` swaps the entry under ogmm_amd.ops for the rest of the process; the product path never imports this."""
import ctypes
import os

from ogmm_amd import _lib, ops

_PROBE = None


def lib():
    global _PROBE
    if _PROBE is None:
        _lib.load()
        _PROBE = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libogmm_probe.so"))
        for fn in (_PROBE.ogmm_probe_gemm_nt, _PROBE.ogmm_probe_gemm_v6):
            fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int
    return _PROBE


def install():
    real = ops._lib.call
    if getattr(real, "_probe_routed", False):
        return

    def routed(name, *a):
        if name != "ogmm_gemm_nt":
            return real(name, *a)
        code = getattr(a[0], "_obj", a[0]).precision          # a[0] = ctypes.byref(GemmDesc)
        fn = lib().ogmm_probe_gemm_v6 if 60 <= code < 90 else lib().ogmm_probe_gemm_nt
        if fn(*a) != 0:
            raise RuntimeError("libogmm_probe.so: %s" % _lib.load().ogmm_last_error().decode(errors="replace"))
    routed._probe_routed = True
    ops._lib.call = routed


def clock_probe(code, buf):
    """read-and-clear of the engine's in-kernel clock probe {shader cycles, 100 MHz wall ticks, workgroups} for ablation code `code`"""
    L = lib()
    fn = L.ogmm_debug_v10_probe if code >= 110 else (L.ogmm_debug_v8_probe if code >= 100 else L.ogmm_debug_v6_probe)
    return fn(buf)
