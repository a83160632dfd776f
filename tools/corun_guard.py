"""Co-run investigation, step 4 (HISTORY.md section 4, round 5): a guard kernel (tools/lds_guard.hip) that fills its LDS and registers with a pattern and
keeps re-reading them runs on one stream while the small-tile fp16x3 GEMM runs on another.  Says whether a neighbour's LDS words, registers or
global loads are what changes, and to what.
usage (GPU box): python3 tools/corun_guard.py        (tools/lds_guard.so is built first if it is missing: hipcc is on the box)"""
import sys, os, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ogmm_amd import ops, synth
from ogmm_amd.gmmreg import GMMReg

so = os.path.join(ROOT, "tools", "lds_guard.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "lds_guard.hip"), "-o", so])
lib = ctypes.CDLL(so)
P, I = ctypes.c_void_p, ctypes.c_int
lib.lds_guard.argtypes = [P, P, I, I, I, I, I, P]
lib.load_guard_fill.argtypes = [P, I, I, P]
lib.load_guard.argtypes = [P, I, I, P, P, I, I, P]
lib.pk_guard.argtypes = [P, P, I, I, I, I, P]

dev = torch.device("cuda", 0)
model = GMMReg(512, 16, bench.CFG); synth.fill_state_dict(model.state_dict()); model = model.to(dev).eval()
L = model._layers()
C, N = 12, 1024
eng16 = ops.Engine("f16x3", torch.zeros(1, dtype=torch.int32, device=dev))
eng32 = ops.Engine("f32", None)
x = torch.randn(C * N, 512, device=dev)
out = torch.empty((C * N, 512), device=dev)
other = torch.cuda.Stream()
MAXR = 4096
report = torch.zeros(MAXR * 8, dtype=torch.int32, device=dev)
count = torch.zeros(1, dtype=torch.int32, device=dev)
words = 8192
pool = torch.zeros(512 * words, dtype=torch.int32, device=dev)
lib.load_guard_fill(pool.data_ptr(), 512, words, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()

def loads(kind, n):
    with torch.cuda.stream(other):
        for _ in range(n):
            if kind == "v2":
                ops.conv1x1(x, L["emd5"], ops.ACT_RELU, out=out, eng=eng16)
            elif kind == "f32":
                ops.conv1x1(x, L["emd5"], ops.ACT_RELU, out=out, eng=eng32)

def show(tag):
    torch.cuda.synchronize()
    n = int(count.item())
    print("%-58s mismatches %d" % (tag, n))
    if n:
        r = report.view(-1, 8)[:min(n, MAXR)].cpu().numpy().astype("uint32")
        kinds = {0: "lds", 1: "reg", 2: "load", 3: "pk"}
        import collections
        print("    by kind:", dict(collections.Counter(kinds[int(k)] for k in r[:, 0])))
        print("    workgroups hit: %d, iterations: %d..%d" % (len(set(r[:, 1].tolist())), r[:, 5].min(), r[:, 5].max()))
        for row in r[:12]:
            print("    %s wg %4d word %6d (byte %6d) saw %08x want %08x it %d hw %08x xcc %x" % (kinds[int(row[0])], row[1], row[2], row[2] * 4, row[3], row[4], row[5], row[6], row[7]))
        pk = r[r[:, 0] == 3]
        if len(pk):
            lanes = pk[:, 2] & 63
            print("    pk: lanes 0-15/16-31/32-47/48-63: %s   accumulator j even/odd: %d/%d   low only/high only/both: %s" % (
                [int(((lanes >> 4) == q).sum()) for q in range(4)], int((pk[:, 6] % 2 == 0).sum()), int((pk[:, 6] % 2 == 1).sum()), [int((pk[:, 7] == h).sum()) for h in (1, 2, 3)]))
        idx = sorted(set(r[r[:, 0] == 0][:, 2].tolist()))
        if idx:
            print("    lds words hit: min %d max %d count %d; first 24: %s" % (idx[0], idx[-1], len(idx), idx[:24]))
    count.zero_(); report.zero_()
    torch.cuda.synchronize()

s0 = torch.cuda.current_stream().cuda_stream
for kind in ("none", "f32", "v2"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.pk_guard(report.data_ptr(), count.data_ptr(), 768, 20000, MAXR, 0, s0)
    e1.record()
    loads(kind, 60)
    torch.cuda.synchronize()
    print("    (guard ran %.2f ms)" % e0.elapsed_time(e1))
    show("load %-4s  packed-fp32 guard (v_pk_fma_f32 against scalar fma)" % kind)
for kind in ("none", "f32", "v2"):
    for lds_bytes in (12288, 32768, 40960, 65536):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.lds_guard(report.data_ptr(), count.data_ptr(), 768, lds_bytes, 600, MAXR, 4, s0)
        e1.record()
        loads(kind, 40)
        torch.cuda.synchronize()
        print("    (guard ran %.2f ms)" % e0.elapsed_time(e1))
        show("load %-4s  lds guard %6d B x 768 workgroups" % (kind, lds_bytes))
    lib.load_guard(pool.data_ptr(), 512, words, report.data_ptr(), count.data_ptr(), 200, MAXR, s0)
    loads(kind, 40)
    show("load %-4s  global-load guard" % kind)

# the victims themselves, in this process, and WHERE their results differ
gamma = torch.softmax(torch.randn(C, N, 16, device=dev), -1); pi = gamma.mean(1)
ref = ops.gmm_feat_mean(gamma, pi, x, C, N).clone()
torch.cuda.synchronize()
for rep in range(6):
    loads("v2", 8)
    g1 = ops.gmm_feat_mean(gamma, pi, x, C, N)
    torch.cuda.synchronize()
    d = (g1 != ref)
    if not bool(d.any()):
        print("feat_mean rep %d: identical" % rep); continue
    nz = d.nonzero()
    print("feat_mean rep %d: %d entries differ" % (rep, nz.shape[0]))
    seen = {}
    for c, j, ch in nz.tolist():
        seen.setdefault((c, j), []).append(ch)
    for (c, j), chs in list(seen.items())[:8]:
        runs, a = [], chs[0]
        for u, v in zip(chs, chs[1:] + [None]):
            if v != u + 1:
                runs.append((a, u)); a = v
        print("    cloud %d j %2d: %4d channels, runs %s   got/ref at first: %.6f / %.6f" % (c, j, len(chs), runs[:6], g1[c, j, chs[0]].item(), ref[c, j, chs[0]].item()))
