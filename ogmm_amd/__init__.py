"""ogmm_amd: MI355X-native (gfx950) hot path of gfmei/ogmm -- drop-in `GMMReg` on hand-written HIP kernels."""
import os as _os

import torch as _torch

# HIP graph replay (GMMReg.capture_graph, Trainer(graph=True)) on ROCm 7.2: with the runtime's "graph packet capture" path on (its default), a replayed
# graph faults on GPU memory once a few thousand ordinary kernel launches have gone through the same device between two replays -- found with the
# ~1800-node training step (tools/train_capture_debug.py launch30000 / twin), gone with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.  The runtime reads the flag
# when it initialises, so it is set here, at import; `graph_replay_safe()` tells whether that was early enough.
_SET_BEFORE_INIT = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" or not _torch.cuda.is_initialized()
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def graph_replay_safe():
    """True if this process's HIP runtime runs (or will run) with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: replayed graphs may then be mixed with other work on
    the device.  False: the runtime was already initialised without the flag when ogmm_amd was imported, or the flag was set to something else."""
    return _SET_BEFORE_INIT and _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"


from .gmmreg import GMMReg  # noqa: F401,E402
