"""ogmm_amd: MI355X-native (gfx950) hot path of gfmei/ogmm -- drop-in `GMMReg` on hand-written HIP kernels."""
import os as _os

import torch as _torch

# HIP graph replay (GMMReg.capture_graph, Trainer(graph=True)) on ROCm 7.2: with the runtime's "graph packet capture" path on (its default), a replayed
# graph faults on GPU memory once a few thousand ordinary kernel launches have gone through the same device between two replays -- found with the
# ~1800-node training step (tools/train_capture_debug.py launch30000 / twin), gone with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.  The runtime reads the flag
# when it initialises, so it is set here, at import; `graph_replay_safe()` tells whether that was early enough.
def _runtime_untouched():
    """True if nothing in this process has initialised the HIP runtime yet.  torch.cuda.is_initialized() only tracks torch's own lazy initialisation;
    torch.cuda.is_available() (hipGetDeviceCount), a ctypes HIP call or a profiler's preloaded library initialise CLR -- which reads its flags then --
    without torch knowing, and torch caches nothing that would tell (`_cached_device_count` is only set after torch's initialisation).  What every such
    path leaves behind is the ROCr runtime's open handle on the compute driver: a file descriptor on /dev/kfd (amdsmi's device count, which
    torch.cuda.device_count() uses on ROCm, does not open it)."""
    if _torch.cuda.is_initialized():
        return False
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return False
            except OSError:
                continue
    except OSError:
        pass          # no procfs: fall back on torch's word
    return True


_SET_BEFORE_INIT = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" or _runtime_untouched()
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def graph_replay_safe():
    """True if this process's HIP runtime runs (or will run) with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: replayed graphs may then be mixed with other work on
    the device.  Safe means: the variable was ALREADY '0' in the environment when ogmm_amd was imported (export it before launch: the documented way,
    and the only one that also covers profilers and other libraries that touch the GPU first), or nothing had touched the runtime by then (neither
    torch.cuda's initialisation nor a device-count query).  False otherwise, or when the flag was set to something else."""
    return _SET_BEFORE_INIT and _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"


from .gmmreg import GMMReg  # noqa: F401,E402
