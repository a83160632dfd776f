"""E/M clustering alone (default B=64: 128 clouds, N=1024, J=16; or `em_time.py C N J`): on-chip and grid-wide engines."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ogmm_amd import ops
torch.manual_seed(0)
C, N, J = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 1024, 16)
xyz = torch.randn(C, N, 3, device="cuda") * 0.5
o = torch.rand(C, N, device="cuda")
ids = ops.fps(xyz, J, None)
for eng in (None, "multi"):
    for thresh in (0.0, 1e-2):          # early exit off / on (two call groups of C / 2 clouds, as GMMReg.forward calls it)
        kw = dict(engine=eng, thresh=thresh, group_size=C // 2 if C % 2 == 0 else C)
        for _ in range(2): ops.gmm_em(xyz, o, ids, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gmm_em(xyz, o, ids, **kw)
        e1.record(); torch.cuda.synchronize()
        print("C=%d N=%d J=%d engine=%s thresh=%g  %.1f us" % (C, N, J, eng, thresh, e0.elapsed_time(e1) / 5 * 1e3))
