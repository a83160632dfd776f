"""The fused EdgeConv kernel alone at B=64 / N=1024 / k=20.  (The phase-by-phase ablation of HISTORY.md section 4 was done with temporary
switches in the kernel -- pooling / plane writes / MFMAs / layer 1 / flushes off one at a time -- read through OGMM_EDGE_ABL.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth, ops
from ogmm_amd.gmmreg import GMMReg
dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
m = GMMReg(512, 16, cfg); synth.fill_state_dict(m.state_dict()); m = m.to(dev).eval()
src, tgt, R, t = synth.make_batch(0, 64, 1024, "partial")
with torch.no_grad():
    m(src.to(dev), tgt.to(dev), fps_starts=synth.fps_starts_for(0, 64, 1024))          # packs the weights
L = m._layers()
xyz = torch.cat([src, tgt], 0).transpose(1, 2).contiguous().to(dev)
idx = ops.knn(xyz, 20)
xcat = torch.empty((xyz.shape[0] * 1024, 512), device=dev)
emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
outs = {}
for pc in (False, True):
    ops.EDGECONV_PC = pc
    xc = torch.full_like(xcat, float("nan"))
    for _ in range(2): ops.edgeconv_fused(xyz, idx, emd, xc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.edgeconv_fused(xyz, idx, emd, xc)
    e1.record(); torch.cuda.synchronize()
    outs[pc] = xc
    print("edgeconv %s %.1f us   nan %d" % ("producer/consumer" if pc else "barrier-phased   ", e0.elapsed_time(e1) / 5 * 1e3, int(torch.isnan(xc).sum())))
print("bit-identical:", torch.equal(outs[False], outs[True]), " max diff %.3e" % (outs[False] - outs[True]).abs().max().item())
ops.EDGECONV_PC = False

if os.environ.get("OGMM_EDGECONV_PROBE") == "1":
    import ctypes
    from ogmm_amd import _lib
    buf = (ctypes.c_ulonglong * 8)()
    _lib.call("ogmm_debug_edgeconv_probe", ctypes.cast(buf, ctypes.c_void_p))          # clear
    ops.edgeconv_fused(xyz, idx, emd, xcat)
    torch.cuda.synchronize()
    _lib.call("ogmm_debug_edgeconv_probe", ctypes.cast(buf, ctypes.c_void_p))
    v = list(buf)
    tiles = max(1, v[6])
    names = ["setup -> barrier 1", "layer 1", "layer 2", "layer 3", "layer 4", "flush + turn-around"]
    tot = sum(v[:5])
    print("phase probe: %d tiles; shader cycles per tile (thread 0 of each workgroup):" % tiles)
    for n_, c in zip(names, v[:5]):
        print("   %-22s %8.0f  (%4.1f %%)" % (n_, c / tiles, 100.0 * c / tot))
    print("   total %.0f cycles per tile; MFMA-only: 10560 (1320 matrix instructions of 32 cycles on 4 SIMDs)" % (tot / tiles))

    ops.EDGECONV_PC = True
    _lib.call("ogmm_debug_edgeconv_pc_probe", ctypes.cast(buf, ctypes.c_void_p))
    ops.edgeconv_fused(xyz, idx, emd, xcat)
    torch.cuda.synchronize()
    _lib.call("ogmm_debug_edgeconv_pc_probe", ctypes.cast(buf, ctypes.c_void_p))
    v = list(buf)
    blocks = max(1, v[7])
    print("producer / consumer kernel, %d blocks: shader cycles (lane 0) per block of the role" % blocks)
    for n_, c, per in (("producer L1", v[0], blocks), ("producer L2", v[1], blocks), ("producer L3 MFMAs", v[2], blocks), ("producer slot wait", v[3], blocks),
                       ("producer L3 epilogue", v[4], blocks), ("consumer wait (mean of 4)", v[5], 4 * blocks), ("consumer compute (mean of 4)", v[6], 4 * blocks)):
        print("   %-30s %8.0f" % (n_, c / per))
