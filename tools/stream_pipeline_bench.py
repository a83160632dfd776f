"""Throughput of consecutive FULL forwards issued round-robin on S streams (S = 1: the plain loop): step i + 1's latency-bound front end and the dependent-launch
gaps of one forward are filled by the other forward's kernels.  usage: stream_pipeline_bench.py [B] [N] [J] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
J = int(sys.argv[3]) if len(sys.argv) > 3 else 16
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
dev = torch.device("cuda", 0)
src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
starts = synth.fps_starts_for(0, B, N)
src, tgt = src.to(dev), tgt.to(dev)


def make():
    m = GMMReg(512, J, cfg)
    synth.fill_state_dict(m.state_dict())
    return m.to(dev).eval()


for S, shared in ((1, True), (2, True), (2, False), (3, True), (1, True), (2, True)):
    models = [make()] if shared else [make() for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    with torch.no_grad():
        for i in range(6):
            with torch.cuda.stream(streams[i % S]):
                ref = models[i % len(models)](src, tgt, fps_starts=starts)
        torch.cuda.synchronize()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                with torch.cuda.stream(streams[i % S]):
                    out = models[i % len(models)](src, tgt, fps_starts=starts)
            torch.cuda.synchronize()
            best = max(best, B * steps / (time.perf_counter() - t0))
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print("streams %d (%s): %.0f pairs/s  (%.3f ms per forward of %d pairs); outputs identical to the warm-up's: %s" % (S, "one model" if shared else "a model per stream", best, 1e3 * B / best, B, same), flush=True)

# ---- the same question without the host in the way: every forward a HIP-graph replay (GMMReg.capture_graph: one launch per forward), replayed round-robin
# on S streams from S separately captured graphs (own static buffers)
if "--graphs" in sys.argv:
    for S in (1, 2, 1, 2):
        models = [make() for _ in range(S)]
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        runs = []
        for m, st in zip(models, streams):
            with torch.cuda.stream(st):
                runs.append(m.capture_graph(B, N))
        torch.cuda.synchronize()
        sdev = starts.to(dev)
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                with torch.cuda.stream(streams[i % S]):
                    out = runs[i % S](src, tgt, sdev)
            torch.cuda.synchronize()
            best = max(best, B * steps / (time.perf_counter() - t0))
        print("graph replays on %d stream(s): %.0f pairs/s (%.3f ms per forward)" % (S, best, 1e3 * B / best), flush=True)
