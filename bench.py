#!/usr/bin/env python
"""Headline benchmark: point-cloud pairs/sec of GMMReg.forward (eval, is_test=False) on BASELINE.json configs[1]
(ModelNet40-shaped partial-overlap + noise, N=1024 points, J=16 mixtures, batch 64 per GPU), fp32-class arithmetic.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without a launcher: this process starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one forward over one batch of 64 synthetic pairs already resident in HBM.  One process per GPU; pairs are
independent, so ranks shard the global pair ids with no data-path collective (weak scaling); the only collectives are
the barriers bracketing the timed region and a max over ranks of the elapsed time.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is the weight-GEMM engine (~90 % of the path's flops).  Default engine: fp16x3 (gemm_f16x3_v10_kernel /
gemm_f16x3_v8_kernel: fp32 operands split into two binary16 terms, 3 v_mfma_f32_32x32x16_f16 per product block, fp32 accumulate: fp32-class
accuracy, parity-tested on every pair of the timed batch); `--precision f32` selects the exact-fp32 engine (v_mfma_f32_32x32x2_f32).  Its launches
are timed live with events on the launch stream inside the timed region: achieved = sum of ALGORITHMIC flops 2*M*N*K over the launches / sum of their
durations, against the dense MFMA peak of the issued dtype (f16: 2500 TFLOP/s; f32: 157.3 TFLOP/s).  `issued_frac` counts the matrix instructions
actually issued (3 per product; 1 on the layers of the term budget).  `path_frac` prices the whole forward (52.82 GFLOP/pair) against the
same peak.  `roofline_other`: the runner-up kernels (EdgeConv: MFMA; attention: HBM), same brackets.
cpu_baseline: the CPU oracle (a plain-PyTorch port of the reference, bit-identical to it) timed on this host's cores (thread sweep, B = 1 and 8; rank 0,
N=1 only); the parity block checks EVERY pair of the timed batch against it.
secondary (N=1 only, behind the headline's timed region): the headline workload on the SHARP weight family (directly behind the headline, with a repeat of the
default family behind it: an A/B in the record), the headline workload with --precision f32 (the strict same-arithmetic figure), and short legs of the other
BASELINE configs -- configs[2] shape (B=256, N=2048, J=64), configs[3] shape per GPU (room clouds) and configs[4] (the training step, 128 pairs per GPU) -- each
with its throughput, its engine roofline fraction and a parity sample against the oracle, so that the driver's record observes them too.
"""
import argparse
import json
import os
import statistics
import sys
import time
from argparse import Namespace

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required by RCCL on this pool's driver (multi-process runs)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # replayed HIP graphs mixed with other launches fault on ROCm 7.2 otherwise (ogmm_amd/__init__.py)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16": 2500.0}   # MI355X_MICROARCH.md chip table: fp32-matrix / dense f16 MFMA
EVENT_EVERY = 4          # the dominant kernel's launches are bracketed by HIP events in every 4th timed step (see eval_leg)
PMC_FILE = "profiles/round6_pmc_counters.txt"
# workload -> (pairs per GPU and step, points, mixtures, algorithmic GFLOP per pair (FlopCounterMode on the reference), cloud kind, first pair id)
WORKLOADS = {"cfg1": (64, 1024, 16, 52.82, "partial", 0), "cfg2": (256, 2048, 64, 107.50, "partial", 0), "cfg3": (64, 2048, 64, 107.50, "room", 0)}


BARE_MFMA_STREAM_TFLOPS = 1710.0          # v_mfma_f32_32x32x16_f16 stream of the engine with nothing else in the loop, real data, whole chip (the part clocks to its power budget: 2495 on zeros)


def make_cfg(J):
    return Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)


CFG = make_cfg(16)          # (tools import it)


def arithmetic_label(precision, term_budget):
    """the arithmetic of the timed path as ONE string (the driver's parsed record keeps strings, not nested dicts)"""
    if precision == "f32":
        return "f32 (exact: v_mfma_f32_32x32x2_f32)"
    if precision == "f16":
        return "f16 (REDUCED: one binary16 term per operand in the large GEMMs, fp32 accumulate)"
    reduced = ", ".join("%s=%d" % kv for kv in sorted(term_budget.items())) or "none"
    return "f32-class via f16x3 split (x = hi + lo in binary16; 3 v_mfma_f32_32x32x16_f16 per product, fp32 accumulate; reduced-term layers: %s)" % reduced


def workload_text(workload, precision):
    return {"cfg1": "BASELINE configs[1]: ModelNet40-shaped partial-overlap+noise pairs, N=1024 points, J=16 mixtures, "
                    "batch 64 per GPU, GMMReg.forward eval (D=512, k=20, M=128, H=4), closed-form weights",
            "cfg2": "BASELINE configs[2] shape: unseen-category-like pairs, N=2048, J=64, batch 256 per GPU (" +
                    ("single-term binary16 GEMMs: REDUCED precision with the tolerance of tests/test_hip_forward.py::"
                     "test_reduced_precision_mode_against_the_emulating_oracle; the config is quoted in bf16)" if precision == "f16"
                     else "run in fp32-class arithmetic, not bf16: bf16 / single-term operands put R at 1e-4...1e-3 rad (SURVEY section 7), "
                          "the parity bar is 1e-5; --precision f16 is the labelled reduced mode)"),
            "cfg3": "BASELINE configs[3] shape per GPU: ICL-NUIM-like room pairs, N=2048, J=64, batch 64 per GPU"}[workload]


def pmc_traffic_bytes(kernel_substr, path=os.path.join(ROOT, PMC_FILE)):
    """HBM bytes per launch of `kernel_substr` from the committed PMC summary: FETCH_SIZE (KiB, doubled: the gfx950 correction of
    MI355X_MICROARCH.md) + WRITE_SIZE (KiB); None when the file or the kernel is absent."""
    try:
        tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        calls = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        section = None
        for line in open(path):
            if line.startswith("## pass:"):
                section = line.split(":", 1)[1].split()
            elif any(k in line for k in ((kernel_substr,) if isinstance(kernel_substr, str) else kernel_substr)) and section in (["FETCH_SIZE"], ["WRITE_SIZE"]):
                tok = line.split()          # <kernel name ...> <launches> <avg us> <counter mean per launch>
                n, val = float(tok[-3]), float(tok[-1])
                tot[section[0]] += n * val          # every template instance of the kernel, weighted by its launches
                calls[section[0]] += n
        if not calls["FETCH_SIZE"] or not calls["WRITE_SIZE"]:
            return None
        return (2.0 * tot["FETCH_SIZE"] / calls["FETCH_SIZE"] + tot["WRITE_SIZE"] / calls["WRITE_SIZE"]) * 1024.0
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=64, help="pairs of the timed batch checked against the CPU oracle (64 = all; 0 = skip the CPU leg)")
    ap.add_argument("--precision", choices=["f16x3", "f32", "f16"], default="f16x3",
                    help="f16 = reduced precision (single binary16 term in the large GEMMs): only meaningful for --workload cfg2, which BASELINE quotes in bf16")
    ap.add_argument("--workload", choices=["cfg1", "cfg2", "cfg3", "train"], default="cfg1",
                    help="cfg1 = BASELINE configs[1] (headline); cfg2 = configs[2] shape (N=2048, J=64, B=256); cfg3 = configs[3] shape per GPU "
                         "(room clouds, N=2048, J=64, B=64); train = configs[4]: full training step, N=1024, J=16, 128 pairs per GPU (global 1024 on 8)")
    ap.add_argument("--train-batch", type=int, default=128, help="pairs per GPU and step for --workload train")
    ap.add_argument("--weights", choices=["default", "sharp"], default="default",
                    help="weight family of the synthetic model (ogmm_amd/synth.py): sharp = peaked attention, overlap scores spanning (0, 1); for A/B runs -- the "
                         "headline is quoted on the default family, the sharp family is a secondary leg")
    ap.add_argument("--pipeline-head", type=int, default=1, help="1 (default): GMMReg.pipeline_head -- the head of a forward (kNN graph, FPS chains: it depends on the resident inputs "
                                                                 "alone) is queued without waiting for the previous forward's tail; 0: every forward strictly behind the previous one")
    ap.add_argument("--secondary", type=int, default=1, help="1 (default): at N=1 the headline run also times short legs of cfg2 / cfg3 / train and attaches them as "
                                                             "`secondary`; 0: headline only")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)          # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks and never touches a GPU

    from ogmm_amd import dist as odist
    rank, local_rank, world = odist.env_rank_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # Test seam (tests/test_dist_gloo.py): OGMM_BENCH_STUB=1 runs THIS control flow -- the world-size guard, the sharding, the barriers around the timed
    # region, the max over ranks, the rank-0-only JSON line -- on CPU over gloo with a stand-in for the forward.  Never set on a GPU box.
    stub = os.environ.get("OGMM_BENCH_STUB") == "1"
    if stub and os.environ.get("OGMM_BENCH_FAIL_RANK") == str(rank):          # test seam: a rank that dies before the rendezvous
        raise SystemExit(3)
    if not stub:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    dist = odist.init("gloo" if stub else "nccl", rank, world, dev)
    ctx = Namespace(rank=rank, world=world, dist=dist, dev=dev, stub=stub)

    if args.workload == "train":
        result = train_leg(args, ctx, args.train_batch, args.steps, args.warmup, cpu_check=args.cpu_sample > 0)
    else:
        result, keep = eval_leg(args, ctx, args.workload, args.steps, args.warmup, profile=args.weights)
        if args.weights != "default":
            result["config"]["workload"] += "; weight family: " + args.weights
        if rank == 0 and world == 1 and not stub:
            result["roofline"]["library_gemm_same_box"] = library_yardstick(dev, result["config"]["pairs_per_gpu_step"] * 2 * result["config"]["n_points"])
        if rank == 0 and world == 1 and args.cpu_sample > 0 and not stub:
            result.update(cpu_leg(args, keep))
        result["fp16_split_overflowed"] = bool(keep.model.fp16_overflowed())      # |activation| > 65504 clamped anywhere in the run?
        del keep
        if args.workload == "cfg1" and world == 1 and not stub and args.secondary:
            torch.cuda.empty_cache()
            result["secondary"] = secondary_legs(args, ctx, result["value"])
            serial = [l for l in result["secondary"] if "SERIAL forwards" in l.get("workload", "") and "value" in l]
            if serial and result["config"]["consecutive_forwards"].startswith("pipelined"):          # the opt-in mode's gain, stated beside the headline it is part of
                result["config"]["serial_pairs_per_s"] = serial[0]["value"]
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def eval_leg(args, ctx, workload, steps, warmup, precision=None, profile="default", pipeline_head=None):
    """One eval workload: W untimed forwards, exactly K timed ones bracketed by barrier + synchronize, MAX over ranks -> the bench line's fields (dict)
    and what the CPU leg needs (model, inputs, last outputs)."""
    from ogmm_amd import dist as odist, ops, synth
    from ogmm_amd.gmmreg import GMMReg
    precision = precision or args.precision
    b_per_gpu, n_points, J, gflop_per_pair, kind, first0 = WORKLOADS[workload]
    cfg = make_cfg(J)
    rank, world, dist, dev, stub = ctx.rank, ctx.world, ctx.dist, ctx.dev, ctx.stub
    model = GMMReg(512, J, cfg)
    synth.fill_state_dict(model.state_dict(), profile=profile)          # "sharp": the second, non-degenerate weight family of the parity suite (secondary leg)
    params_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model.precision = precision
    if stub:
        b_per_gpu, n_points = 2, 64
        real, model = model, _StubForward(rank)
        model.precision, model.term_budget, model.sinkhorn_thresh = real.precision, real.term_budget, real.sinkhorn_thresh
    else:
        model = model.to(dev).eval()
        # the inputs are resident in HBM before the timed region starts (the contract's own premise), so consecutive forwards may be pipelined: the head of
        # step i + 1 (cloud stacking, kNN graph, FPS chains -- it depends on the inputs alone) is queued on its own streams and runs under the tail of step i.
        # Every step's whole work is inside the timed region (barrier + synchronize on both sides).  --pipeline-head 0 switches it off.
        model.pipeline_head = bool(getattr(args, "pipeline_head", 1) if pipeline_head is None else pipeline_head)

    first, _ = odist.shard_pairs(rank, world, b_per_gpu, first0)         # global pair ids of this rank's shard
    src, tgt, _, _ = synth.make_batch(first, b_per_gpu, n_points, kind)
    starts = synth.fps_starts_for(first, b_per_gpu, n_points)
    src, tgt = src.to(dev), tgt.to(dev)

    def barrier():
        odist.barrier(dist, cuda=not stub)

    dom_tag = "f16x3" if precision == "f16" else precision
    with torch.no_grad():
        for _ in range(warmup):
            out = model(src, tgt, fps_starts=starts)
        # The GEMM launches and the two runner-up kernels (EdgeConv, attention) are bracketed by HIP events on their launch stream inside the timed
        # region -- with events made beforehand (creating two events per launch in the loop costs the host more than the launch: 0.3 ms per step)
        # and only in every EVENT_EVERY-th step: each bracket costs the GPU an extra signal packet on both sides of the launch (tools/bench_overhead.py:
        # 6700 pairs/s without brackets, 6375-6600 with all of them).  Average launch durations come from the sampled steps, the throughput from all.
        ops.GEMM_TIMELINE, ops.GEMM_TIMELINE_ONLY, ops.KERNEL_TIMELINE = [], None, []
        out = model(src, tgt, fps_starts=starts)          # one more warm-up forward: counts the bracketed launches
        per_step = len(ops.GEMM_TIMELINE) + len(ops.KERNEL_TIMELINE)
        ops.recycle_timing_events(ops.GEMM_TIMELINE)
        ops.recycle_timing_events(ops.KERNEL_TIMELINE)
        sampled = [i % EVENT_EVERY == 0 for i in range(steps)]
        if not stub:
            ops._EVENT_POOL.extend(torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step * sum(sampled)))
            torch.cuda.synchronize()
        barrier()
        ops.GEMM_TIMELINE, ops.KERNEL_TIMELINE = [], None
        ktl = []
        t0 = time.perf_counter()
        for i in range(steps):
            ops.GEMM_TIMELINE_ONLY = None if sampled[i] else set()
            ops.KERNEL_TIMELINE = ktl if sampled[i] else None
            out = model(src, tgt, fps_starts=starts)
        barrier()
        elapsed = time.perf_counter() - t0
        timeline, ops.GEMM_TIMELINE, ops.GEMM_TIMELINE_ONLY, ops.KERNEL_TIMELINE = ops.GEMM_TIMELINE, None, None, None
    n_sampled = sum(sampled)
    elapsed = odist.max_over_ranks(dist, elapsed, dev)

    pairs = b_per_gpu * world * steps
    value = pairs / elapsed
    step_ms = 1e3 * elapsed / steps
    all_gemm_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in timeline)
    all_gemm_flop = sum(f for _, _, f, *_ in timeline)
    dom = [(e0.elapsed_time(e1), f, iss) for e0, e1, f, v, _, iss in timeline if v == dom_tag]
    gemm_bytes = sum(b for _, _, _, v, b, _ in timeline if v == dom_tag)     # un-pooled, N > 64 launches of the engine
    gemm_ms, gemm_flop = sum(d for d, _, _ in dom), sum(f for _, f, _ in dom)
    issued_flop = sum(f * iss for _, f, iss in dom)
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    peak = PEAK_TFLOPS[precision]
    kernel = {"f16x3": "gemm_f16x3_v10_kernel / gemm_f16x3_v8_kernel (LDS-DMA engine: 256x256x32 tiles, both operands by global_load_lds, 3 v_mfma_f32_32x32x16_f16 per "
                       "product block -- 1 on the similarity, the one layer of the term budget that holds on both weight families; v10 = 4 waves of 64x256 for N >= 512, "
                       "v8 = 8 waves of 32x256 for N = 256; gemm_f16x3_v2 <2,2,2,2> for shapes under 256 tiles)",
              "f16": "gemm_f16x3_v4_kernel in single-term mode on the large shapes (1 v_mfma_f32_32x32x16_f16 per block; REDUCED precision), the fp16x3 kernels elsewhere",
              "f32": "gemm_nt_kernel<2,2,2,2,false> (v_mfma_f32_32x32x2_f32)"}[precision]

    # HBM traffic of the dominant kernel: NOT measured in this run (rocprofv3 counters need their own process and passes) -- replayed from the
    # committed PMC summary of the same command, with its provenance spelled out; null when there is none for this workload / precision
    traffic = pmc_traffic_bytes(("gemm_f16x3_v10_kernel", "gemm_f16x3_v8_kernel")) if precision == "f16x3" and workload == "cfg1" else None

    # runner-up kernels, as bracketed in the same sampled steps
    others = []
    for name, bound, pk, unit in (("edgeconv_fused_kernel", "mfma", PEAK_TFLOPS["f16x3"], "TFLOP/s"), ("attention_t_kernel", "hbm", 8000.0, "GB/s")):
        rows = [(e0.elapsed_time(e1), fl, by) for e0, e1, nm, fl, by in ktl if nm == name]
        if rows:
            ms = sum(r[0] for r in rows)
            ach = (sum(r[1] for r in rows) / (ms * 1e-3) / 1e12) if bound == "mfma" else (sum(r[2] for r in rows) / (ms * 1e-3) / 1e9)
            others.append({"kernel": name, "bound": bound, "achieved": ach, "peak": pk, "unit": unit, "frac": ach / pk, "launches": len(rows),
                           "avg_launch_us": 1e3 * ms / len(rows), "share_of_step": ms / n_sampled / step_ms,
                           "note": "algorithmic flops of the four EdgeConv layers (fp16x3: 3x issued)" if bound == "mfma" else "algorithmic bytes: Q in, O out, K / V in"})

    budget = dict(model.term_budget) if precision == "f16x3" else {}
    label = arithmetic_label(precision, budget)
    result = {
        "metric": "pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (reduced)" if precision == "f16" else ("f32" if precision == "f32" else "f32 (f16x3 split: %s)" % (", ".join("%s=%d" % kv for kv in sorted(budget.items())) or "3 terms everywhere")),
        "data": "synthetic", "engine": precision,
        "config": {"workload": workload_text(workload, precision) + "; arithmetic: " + label,
                   "arithmetic": label,
                   "pairs_per_gpu_step": b_per_gpu, "n_points": n_points, "n_clusters": J, "parallelism": "pairs sharded x%d, no data-path collective" % world,
                   "term_budget": budget, "sinkhorn_thresh": model.sinkhorn_thresh,
                   "consecutive_forwards": ("pipelined: the head of step i+1 (kNN graph, FPS chains; depends on the resident inputs alone) is queued on its own streams and overlaps "
                                            "the tail of step i (GMMReg.pipeline_head; --pipeline-head 0: serial)" if getattr(model, "pipeline_head", False) else "serial: every forward behind the previous one")},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": traffic,
                     "traffic_measured_in_this_run": False,
                     "traffic_source": None if traffic is None else PMC_FILE + " (rocprofv3 --pmc passes of `python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --secondary 0`, collected by "
                                       "tools/collect_profiles.sh; its header names the commit): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, launch-weighted mean",
                     "algorithmic_bytes_per_launch": gemm_bytes / max(1, len(dom)),
                     "kernel": kernel, "launches": len(dom), "avg_launch_us": 1e3 * gemm_ms / max(1, len(dom)),
                     "issued_frac": (issued_flop / (gemm_ms * 1e-3) / 1e12 / peak) if gemm_ms > 0 else 0.0,      # matrix instructions actually issued: 3 (1 under the term budget) per product
                     "bracketed_steps": "%d of the %d timed steps (every %d-th)" % (n_sampled, steps, EVENT_EVERY),
                     "kernel_share_of_step": gemm_ms / n_sampled / step_ms,
                     "all_gemm_share_of_step": all_gemm_ms / n_sampled / step_ms,          # every GEMM launch of the sampled steps, small-tile and fp32 ones included
                     "all_gemm_gflop_per_pair": all_gemm_flop / (b_per_gpu * n_sampled) / 1e9,
                     "path_frac": value / world * gflop_per_pair / 1e3 / peak,
                     # what this design could reach at most on this part: every matrix instruction the engine ISSUES (3 per product block, 1 under the term budget) at the
                     # rate its bare MFMA stream sustains on real data (BARE_MFMA_STREAM_TFLOPS: MFMA + barrier only, profiles/round5_gemm_engine.txt v84 / v118), plus
                     # the step's time outside the GEMM launches as measured here (EdgeConv, attention, kNN / FPS head, E/M + matching tail, launch gaps)
                     "ceiling_pairs_per_s": (world * b_per_gpu / (sum(f * iss for e0, e1, f, v, _, iss in timeline) / n_sampled / (BARE_MFMA_STREAM_TFLOPS * 1e12)
                                                                + max(0.0, step_ms - all_gemm_ms / n_sampled) * 1e-3)) if (precision == "f16x3" and not stub and all_gemm_ms > 0) else None,
                     "ceiling_model": "pairs per step / (issued matrix flops of the step / %.0f TFLOP/s bare f16 MFMA stream on real data + this run's step time outside GEMM launches)" % BARE_MFMA_STREAM_TFLOPS},
        "roofline_other": others,
    }
    keep = Namespace(model=model, cfg=cfg, params_cpu=params_cpu, src=src, tgt=tgt, starts=starts, out=out)
    return result, keep


def library_yardstick(dev, rows):
    """the vendor library (hipBLASLt through torch.matmul) on the dominant GEMM shape, measured here and now"""
    lib = {}
    for tag, dt in (("fp32", torch.float32), ("f16", torch.float16)):
        a_ = torch.randn(rows, 1024, device=dev, dtype=dt)
        b_ = torch.randn(1024, 1024, device=dev, dtype=dt)
        for _ in range(2):
            a_ @ b_.t()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            a_ @ b_.t()
        e1.record()
        torch.cuda.synchronize()
        lib[tag + "_tflops"] = 2.0 * a_.shape[0] * 1024 * 1024 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        del a_, b_
    lib["shape"] = "%dx1024x1024" % rows
    return lib


def parity_sample(keep, ids, threads):
    """pairs `ids` of the leg's LAST TIMED forward against the CPU oracle -> dict (max R / t / overlap-score error)"""
    from oracle import ogmm_oracle as O
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    r, t, o = [], [], []
    try:
        with torch.no_grad():
            for i in ids:
                ref = O.forward(keep.params_cpu, keep.cfg, keep.src[i:i + 1].cpu(), keep.tgt[i:i + 1].cpu(), keep.starts[:, i:i + 1])
                r.append(O.rotation_error_rad(keep.out[0][i:i + 1].cpu(), ref[0]).item())
                t.append(O.translation_error(keep.out[1][i:i + 1].cpu(), ref[1]).item())
                o.append(max((keep.out[2][i:i + 1].cpu() - ref[2]).abs().max().item(), (keep.out[3][i:i + 1].cpu() - ref[3]).abs().max().item()))
    finally:
        torch.set_num_threads(old)
    return {"R_err_rad_max": max(r), "t_err_max": max(t), "overlap_err_max": max(o), "pairs_checked": len(ids),
            "of": "pairs %s of the leg's last timed forward" % list(ids), "against": "CPU oracle"}


def reduced_mode_policy(name):
    """oracle/split_emulation.py policy of precision = "f16": one binary16 term per operand ("x1") on the layers the large-shape engine multiplies; the EdgeConv kernel,
    the attention kernel and the small anchor-side projections keep their split terms ("x3")"""
    if name.startswith("emd.conv") and name != "emd.conv5":
        return "x3"
    if name.endswith((".attn.qk", ".attn.pv", ".attn.proj.1", ".attn.proj.2")):
        return "x3"
    return "x1"


def reduced_parity_sample(keep, ids, threads):
    """pairs `ids` of a reduced-precision leg's last timed forward: distance to the exact CPU oracle, beside the distance of the oracle evaluated with the same
    operand rounding (what the labelled arithmetic does to the reference's own algorithm) -- the HIP path is held to the latter's scale, not to 1e-5"""
    from oracle import ogmm_oracle as O
    from oracle import split_emulation as E
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    r_hip, r_emu, r_he = [], [], []
    try:
        with torch.no_grad():
            for i in ids:
                a = (keep.params_cpu, keep.cfg, keep.src[i:i + 1].cpu(), keep.tgt[i:i + 1].cpu(), keep.starts[:, i:i + 1])
                exact = O.forward(*a)
                with E.policy(reduced_mode_policy):
                    emu = O.forward(*a)
                got = keep.out[0][i:i + 1].cpu()
                r_hip.append(O.rotation_error_rad(got, exact[0]).item())
                r_emu.append(O.rotation_error_rad(emu[0], exact[0]).item())
                r_he.append(O.rotation_error_rad(got, emu[0]).item())
    finally:
        torch.set_num_threads(old)
    return {"R_err_rad_max_vs_exact_oracle": max(r_hip), "emulating_oracle_R_err_rad_max_vs_exact_oracle": max(r_emu), "R_err_rad_max_vs_emulating_oracle": max(r_he),
            "pairs_checked": len(ids), "of": "pairs %s of the leg's last timed forward" % list(ids),
            "against": "CPU oracle, exact and with the reduced mode's operand rounding emulated (oracle/split_emulation.py 'x1')",
            "bar": "reduced precision cannot meet 1e-5; the asserted bar (tests/test_hip_forward.py) is <= 4 x the emulating oracle's own deviation, max and median over 16 pairs"}


def secondary_legs(args, ctx, headline_value=None):
    """BASELINE configs[2], [3] (shapes per GPU) and [4] (training step) as short legs behind the headline's timed region: the same code paths as
    `--workload cfg2 | cfg3 | train`, fewer steps; each with a parity sample.  A leg that fails reports its error instead of taking the headline down."""
    legs = []
    threads = min(16, os.cpu_count() or 1)
    roof_keys = ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us", "kernel_share_of_step")
    try:
        # the headline workload once more on the SHARP weight family (peaked attention, overlap scores spanning (0, 1)): the same kernels, so the same
        # throughput -- what the leg adds to the record is the parity of a timed forward on non-degenerate weights (16 of its 64 pairs against the oracle).
        # Run DIRECTLY behind the headline (round 4's record had it behind cfg2 + cfg3 and 5.5 % low: leg order / clocks, or data dependence?), and followed by
        # a repeat of the default family: `ab_with_headline` = default (headline) -> sharp -> default again, same box, same minute.
        res, keep = eval_leg(args, ctx, "cfg1", 20, 5, profile="sharp")          # (the headline's own step count: short legs read low -- two steps carry the event brackets)
        leg = {"workload": res["config"]["workload"].replace("closed-form weights", "closed-form weights of the SHARP family (synth.fill_state_dict(profile='sharp'))"),
               "metric": "pairs_per_sec", "value": res["value"], "unit": "pairs/s", "ms_per_step": res["ms_per_step"], "steps": 20, "warmup": 5,
               "roofline": {k: res["roofline"][k] for k in roof_keys},
               "parity": parity_sample(keep, tuple(range(0, 64, 4)), threads) if args.cpu_sample > 0 else None,
               "fp16_split_overflowed": bool(keep.model.fp16_overflowed())}
        del keep
        torch.cuda.empty_cache()
        res2, keep = eval_leg(args, ctx, "cfg1", 20, 5)
        del keep
        leg["ab_with_headline"] = {"default_before_pairs_per_s": headline_value, "sharp_pairs_per_s": res["value"], "default_after_pairs_per_s": res2["value"],
                                   "note": "three consecutive legs of 20 timed steps on one box: the headline, this leg, the headline's workload again"}
        legs.append(leg)
    except Exception as e:          # noqa: BLE001
        legs.append({"workload": "cfg1 on sharp weights", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.empty_cache()
    try:
        # the headline workload with every forward strictly behind the previous one (GMMReg.pipeline_head = False: what a drop-in caller gets by default)
        res, keep = eval_leg(args, ctx, "cfg1", 20, 5, pipeline_head=0)
        legs.append({"workload": res["config"]["workload"] + "; SERIAL forwards (--pipeline-head 0)", "metric": "pairs_per_sec", "value": res["value"], "unit": "pairs/s",
                     "ms_per_step": res["ms_per_step"], "steps": 20, "warmup": 5, "consecutive_forwards": res["config"]["consecutive_forwards"],
                     "roofline": {k: res["roofline"][k] for k in roof_keys}})
        del keep
    except Exception as e:          # noqa: BLE001
        legs.append({"workload": "cfg1 --pipeline-head 0", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.empty_cache()
    try:
        # the strict same-arithmetic figure: every GEMM on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s peak)
        res, keep = eval_leg(args, ctx, "cfg1", 5, 2, precision="f32")
        legs.append({"workload": res["config"]["workload"], "metric": "pairs_per_sec", "value": res["value"], "unit": "pairs/s", "ms_per_step": res["ms_per_step"], "steps": 5, "warmup": 2,
                     "dtype": res["dtype"], "roofline": {k: res["roofline"][k] for k in roof_keys},
                     "parity": parity_sample(keep, (0, 21, 42, 63), threads) if args.cpu_sample > 0 else None})
        del keep
    except Exception as e:          # noqa: BLE001
        legs.append({"workload": "cfg1 --precision f32", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.empty_cache()
    for wl in ("cfg2", "cfg3"):
        try:
            res, keep = eval_leg(args, ctx, wl, 5, 2)
            b = res["config"]["pairs_per_gpu_step"]
            leg = {"workload": res["config"]["workload"], "metric": "pairs_per_sec", "value": res["value"], "unit": "pairs/s", "ms_per_step": res["ms_per_step"], "steps": 5, "warmup": 2,
                   "roofline": {k: res["roofline"][k] for k in roof_keys},
                   "parity": parity_sample(keep, (0, b // 2, b - 1), threads) if args.cpu_sample > 0 else None,
                   "fp16_split_overflowed": bool(keep.model.fp16_overflowed())}
            del keep
        except Exception as e:          # noqa: BLE001
            leg = {"workload": wl, "error": "%s: %s" % (type(e).__name__, e)}
        legs.append(leg)
        torch.cuda.empty_cache()
    torch.cuda.empty_cache()
    try:
        # BASELINE configs[2] AS QUOTED: reduced precision.  precision = "f16": one binary16 term per operand (11 significand bits >= bf16's 8) in the large GEMMs,
        # fp32 accumulation.  It cannot meet 1e-5; its parity is stated against the CPU oracle evaluated WITH THE SAME OPERAND ROUNDING
        # (oracle/split_emulation.py, the policy of tests/test_hip_forward.py::test_reduced_precision_mode_against_the_emulating_oracle) next to the exact oracle.
        res, keep = eval_leg(args, ctx, "cfg2", 5, 2, precision="f16")
        b = res["config"]["pairs_per_gpu_step"]
        legs.append({"workload": res["config"]["workload"], "metric": "pairs_per_sec", "value": res["value"], "unit": "pairs/s", "ms_per_step": res["ms_per_step"], "steps": 5, "warmup": 2,
                     "dtype": res["dtype"], "roofline": {k: res["roofline"][k] for k in roof_keys},
                     "parity": reduced_parity_sample(keep, (0, b // 2, b - 1), threads) if args.cpu_sample > 0 else None,
                     "fp16_split_overflowed": bool(keep.model.fp16_overflowed())})
        del keep
    except Exception as e:          # noqa: BLE001
        legs.append({"workload": "cfg2 --precision f16", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.empty_cache()
    try:
        res = train_leg(args, ctx, args.train_batch, 5, 2, cpu_check=args.cpu_sample > 0)
        legs.append({"workload": res["config"]["workload"], "metric": res["metric"], "value": res["value"], "unit": "pairs/s", "ms_per_step": res["ms_per_step"], "steps": 5, "warmup": 2,
                     "roofline": {k: res["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "kernel_share_of_step")},
                     "parity": res.get("parity"), "step_launch": res["step_launch"], "final_loss": res["final_loss"], "fp16_split_overflowed": res["fp16_split_overflowed"]})
    except Exception as e:          # noqa: BLE001
        legs.append({"workload": "train", "error": "%s: %s" % (type(e).__name__, e)})
    torch.cuda.empty_cache()
    return legs


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start N fresh child processes of this same command, one rank
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR = 127.0.0.1 / a free MASTER_PORT in their environment -- what `python -m
    torch.distributed.run --nproc-per-node N` would set), pass rank 0's stdout (the ONE JSON line) through, everything else to stderr, and exit with the
    worst child status.  The reference's own multi-GPU mechanism is one process driving all GPUs (`nn.DataParallel`, train.py:190-192); here it is one
    process per GPU over RCCL, and this parent only waits: it makes no HIP call (a process that has initialised the GPU must not fork / exec workers)."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # like torchrun: a rank that dies takes the job down (its peers would sit in the rendezvous / a collective until their timeouts otherwise); the children
    # are ended by their exact PIDs
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.05)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    reader.join(timeout=10)
    sys.stdout.write(out0[0] if out0 else "")
    sys.stdout.flush()
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        print("[bench] rank exit codes: %s" % rcs, file=sys.stderr)
        raise SystemExit(worst if worst < 256 else 1)


class _StubForward:
    """stand-in for the model under OGMM_BENCH_STUB=1 (CPU test of main()'s multi-process control flow): rank r 'takes' (r + 1) ms per forward"""

    def __init__(self, rank):
        self.rank = rank

    def __call__(self, src, tgt, fps_starts=None):
        time.sleep(1e-3 * (self.rank + 1))
        B, _, N = src.shape
        return (torch.eye(3).expand(B, 3, 3), torch.zeros(B, 3), torch.full((B, N), 0.5), torch.full((B, N), 0.5), torch.zeros(()))

    def fp16_overflowed(self):
        return False


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; None when the file does not say"""
    try:
        seen, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def cpu_leg(args, keep):
    """cpu_baseline (the CPU oracle -- a plain-PyTorch port of the reference, bit-identical to it on the generating machine -- on this host's
    cores, B = 1 and B = 8, thread sweep, 3 warm-up + >= 10 timed at the best setting) and the parity of the TIMED forward (every pair of its batch)."""
    from oracle import ogmm_oracle as O
    host = os.cpu_count() or 1
    cfg, params_cpu, starts, out = keep.cfg, keep.params_cpu, keep.starts, keep.out
    s_cpu, t_cpu = keep.src.cpu(), keep.tgt.cpu()
    budget_t0 = time.perf_counter()

    def timed(nt, b, warm, reps):
        torch.set_num_threads(nt)
        ts = []
        with torch.no_grad():
            for i in range(warm + reps):
                c0 = time.perf_counter()
                O.forward(params_cpu, cfg, s_cpu[:b], t_cpu[:b], starts[:, :b])
                dt = time.perf_counter() - c0
                print("[bench cpu leg] %d threads, B=%d: %.2f s" % (nt, b, dt), file=sys.stderr, flush=True)
                if i >= warm:
                    ts.append(dt)
                if dt > 8.0 * b:          # hopelessly oversubscribed setting (more threads than the container has cores): one sample is enough
                    return b / dt
        return b / statistics.median(ts)

    # thread counts: the full sweep at 1, 8, 16 and -- B = 1 only, to stay inside the command's time budget -- 32, 64, 128 where the host has them.  The box's
    # containers see all of the host's hardware threads but may run on a fraction of them, and an OpenMP team larger than that spins in every parallel region
    # (round 2's 128-thread figure was 5x slower than 16 threads; round 3's sweep had 32 and 64 threads at 0.45x and 0.2x of 16): the large teams are in the
    # record as numbers (cpu_baseline.sweep_pairs_per_s), the reported value is the best setting
    sweep = {}
    for nt in (1, 8, 16, 32, 64, 128):
        if nt > host or time.perf_counter() - budget_t0 > (30.0 if nt <= 16 else 50.0):
            continue
        if nt > 16:          # (round 6: VERDICT asked for the large teams' numbers in the record; one warm-up + two timed B = 1 forwards each, all physical cores included)
            sweep[nt] = {"B1": timed(nt, 1, 1, 2)}
            continue
        sweep[nt] = {"B1": timed(nt, 1, 1, 3)}
        if nt > 1 and sweep[nt]["B1"] > 0.5:
            sweep[nt]["B8"] = timed(nt, 8, 1, 2)
    best_nt, best_b = max(((nt, b) for nt, d in sweep.items() for b in d), key=lambda k: sweep[k[0]][k[1]])
    best = timed(best_nt, 8 if best_b == "B8" else 1, 3, 10)
    # parity on the TIMED path: every pair of the last timed step's outputs (eval-mode pairs are independent and the anchor draws are pinned per pair
    # id, so pair i of the 64-pair forward is the same computation as the oracle's pair i) -- a separate small forward would run the small-shape
    # GEMM engines instead of the ones this benchmark measures
    n = min(args.cpu_sample, s_cpu.shape[0]) if args.cpu_sample < 64 else s_cpu.shape[0]
    torch.set_num_threads(best_nt)
    got = [t_[:n].cpu() for t_ in out[:4]]
    r_all, t_all, o_all = [], [], []
    with torch.no_grad():
        for a in range(0, n, 8):
            e = min(n, a + 8)
            ref = O.forward(params_cpu, cfg, s_cpu[a:e], t_cpu[a:e], starts[:, a:e])
            r_all.append(O.rotation_error_rad(got[0][a:e], ref[0])); t_all.append(O.translation_error(got[1][a:e], ref[1]))
            o_all.append(torch.maximum((got[2][a:e] - ref[2]).abs().amax(1), (got[3][a:e] - ref[3]).abs().amax(1)))
    r_all, t_all, o_all = torch.cat(r_all), torch.cat(t_all), torch.cat(o_all)
    return {
        "cpu_baseline": {"value": best, "unit": "pairs/s", "cores": best_nt, "kind": "port", "host_threads": host,
                         "cores_semantics": "threads used by the BEST setting of the sweep {1, 8, 16, 32, 64, 128} (every team size up to the host's hardware threads is in "
                                            "sweep_pairs_per_s) -- NOT the host's core count (physical_cores)",
                         "physical_cores": physical_cores(),
                         "one_thread_pairs_per_s": sweep.get(1, {}).get("B1"),
                         "sweep_pairs_per_s": {str(nt): {k: round(v, 3) for k, v in d.items()} for nt, d in sweep.items()},
                         "sample": "best of sweep: CPU oracle forward on pairs of the same batch, thread sweep {1, 8, 16} at B = 1 and B = 8 and {32, 64, 128} at B = 1 (1 warm-up + 2-3 timed each), "
                                   "then the best setting (%d threads, %s) with 3 warm-up + 10 timed forwards, median" % (best_nt, best_b)},
        "parity": {"R_err_rad_max": r_all.max().item(), "R_err_rad_median": r_all.median().item(), "t_err_max": t_all.max().item(),
                   "overlap_err_max": o_all.max().item(), "pairs_checked": n, "pairs_over_1e-5": int(((r_all >= 1e-5) | (t_all >= 1e-5)).sum()),
                   "of": "the timed %d-pair forward itself (%s)" % (s_cpu.shape[0], "every pair" if n == s_cpu.shape[0] else "its first %d pairs" % n),
                   "against": "CPU oracle (bit-identical to the reference on its golden fixtures)"},
    }


def train_leg(args, ctx, B, steps, warmup, cpu_check=True):
    """BASELINE configs[4]: one step = forward (train-mode BatchNorm) + the loss of train.py:54-72 + backward + one flat
    gradient all-reduce (RCCL; skipped at N=1) + Adam + BatchNorm-buffer broadcast, on 128 synthetic pairs per GPU."""
    from ogmm_amd import dist as odist, ops, synth, train_ops
    from ogmm_amd.gmmreg import GMMReg
    from ogmm_amd.trainer import Trainer
    N, J_ = 1024, 16
    cfg = make_cfg(J_)
    rank, world, dist, dev = ctx.rank, ctx.world, ctx.dist, ctx.dev
    stub = getattr(ctx, "stub", False)
    if stub:
        # Test seam (tests/test_train_dist_gloo.py::test_bench_train_leg_under_two_ranks): THIS function's control flow -- the sharding of the global pair
        # ids, the Trainer's all-reduce / buffer broadcast, the barriers, the max over ranks, the result's fields -- on CPU over gloo, with the training graph on
        # the plain-PyTorch operation set of tests/train_ref.py at a toy size.  Never set on a GPU box.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from train_ref import RefTrainOps
        N, J_, B = 96, 8, 1
        cfg = Namespace(gnn_k=8, num_heads=4, km_clusters=16, overlap_radius=0.035, n_clusters=J_)
    model = GMMReg(512, J_, cfg)
    synth.fill_state_dict(model.state_dict())
    params_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    model.precision = args.precision if args.precision != "f16" else "f16x3"
    if stub:
        model._train_ops = RefTrainOps()
    first, _ = odist.shard_pairs(rank, world, B)
    batch = [t.to(dev) for t in synth.make_train_batch(first, B, N, "partial")]
    starts = synth.fps_starts_for(first, B, N)
    # forward + loss + backward replayed from a HIP graph (Trainer(graph=True): ~2500 launches per step, whose enqueueing takes the host as long as the GPU
    # needs to run them); OGMM_TRAIN_GRAPH=0 times the eager step.  The first steps are eager, the next one records: all inside the warm-up.
    use_graph = os.environ.get("OGMM_TRAIN_GRAPH", "1") != "0" and not stub
    trainer = Trainer(model, dist=dist, world=world, graph=use_graph, **({"welsch_top_k": 48} if stub else {}))
    for _ in range(max(warmup, trainer.graph_warmup + 2) if use_graph else warmup):
        info = trainer.step(*batch, fps_starts=starts)
    odist.barrier(dist, cuda=not stub)
    t0 = time.perf_counter()
    for i in range(steps):
        info = trainer.step(*batch, fps_starts=starts)
    odist.barrier(dist, cuda=not stub)
    elapsed = time.perf_counter() - t0
    final_loss, loss_parts = float(info["loss"]), {k: float(v) for k, v in info["parts"].items()}
    # the engine's launches are bracketed in separate EAGER steps behind the timed region (events cannot sit inside a replayed graph, and they cost GPU time)
    trainer.graph = False
    ops.GEMM_TIMELINE = []
    sampled = [True] * (1 if use_graph else max(1, steps // EVENT_EVERY))
    for _ in sampled:
        trainer.step(*batch, fps_starts=starts)
    if not stub:
        torch.cuda.synchronize(dev)
    timeline, ops.GEMM_TIMELINE, ops.GEMM_TIMELINE_ONLY = ops.GEMM_TIMELINE, None, None
    n_sampled = sum(sampled)
    elapsed = odist.max_over_ranks(dist, elapsed, dev)
    value = B * world * steps / elapsed
    dom = [(e0.elapsed_time(e1), f) for e0, e1, f, v, *_ in timeline if v == model.precision]
    gemm_ms, gemm_flop = sum(d for d, _ in dom), sum(f for _, f in dom)
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    peak = PEAK_TFLOPS[model.precision]
    result = {
        "metric": "train_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if model.precision == "f32" else "f32 (f16x3 split: forward and dX three terms, dW %s; loss scale 2^16)" % (
            "two terms (activation operand rounded to binary16)" if train_ops.BWD_TERMS_DW == 2 else "three terms"), "data": "synthetic", "engine": model.precision,
        "config": {"workload": "BASELINE configs[4]: end-to-end training step (forward in train mode, loss of train.py, backward, gradient "
                               "all-reduce, Adam), ModelNet40-shaped partial-overlap pairs, N=1024, J=16, %d pairs per GPU" % B,
                   "pairs_per_gpu_step": B, "n_points": N, "n_clusters": J_,
                   "parallelism": "data parallel x%d: per-rank BatchNorm statistics, one 52 MB gradient all-reduce per step" % world},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                     "kernel": "GEMM engine launches of the training step: forward layers, dX = dY W and the split-K dW = dY^T X (all on the fp16x3 engine)", "launches": len(dom),
                     "bracketed_steps": "%d eager step(s) behind the %d timed ones" % (n_sampled, steps),
                     "kernel_share_of_step": gemm_ms / n_sampled / (1e3 * elapsed / steps)},
        "step_launch": "HIP graph replay of forward + loss + backward; all-reduce, un-scaling and Adam eager" if use_graph else "eager",
        "final_loss": final_loss, "loss_parts": loss_parts,
    }
    if rank == 0 and world == 1 and cpu_check and not stub:
        # parity sample + CPU baseline of the training step: the first 2 pairs as their own training batch (train-mode BatchNorm statistics are per batch, so
        # a sub-batch of the timed one is a different computation) through a fresh HIP model and through the oracle: loss parts, and the oracle's time
        from oracle import ogmm_oracle as O
        from ogmm_amd import losses
        n = 2
        P = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in params_cpu.items()}
        cb = [t[:n].cpu() for t in batch]
        times = []
        for _ in range(2):
            for v in P.values():
                v.grad = None
            Pc = {k: (v if v.requires_grad else v.clone()) for k, v in P.items()}          # (the train-mode oracle updates running statistics in place)
            c0 = time.perf_counter()
            out_o = O.forward(Pc, cfg, cb[0], cb[1], starts[:, :n], train=True)
            loss_o = O.training_loss(out_o, cb[0], cb[1], cb[2], cb[3], cb[4], 10.0, 512)
            loss_o.backward()
            times.append(time.perf_counter() - c0)
        result["cpu_baseline"] = {"value": n / times[-1], "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": "forward+loss+backward of the CPU oracle on the first %d pairs (no optimizer step), second of two runs" % n}
        m2 = GMMReg(512, J_, cfg)
        m2.load_state_dict(params_cpu)
        m2 = m2.to(dev).train()
        m2.precision = model.precision
        out_h = m2(batch[0][:n], batch[1][:n], fps_starts=starts[:, :n])
        loss_h, _ = losses.training_loss(out_h, *[t[:n] for t in batch], 10.0, 512)
        result["parity"] = {"train_loss_rel_err": abs(float(loss_h) - float(loss_o)) / abs(float(loss_o)),
                            "R_err_rad_max": O.rotation_error_rad(out_h[0].detach().cpu(), out_o[0].detach()).max().item(),
                            "t_err_max": O.translation_error(out_h[1].detach().cpu(), out_o[1].detach()).max().item(), "pairs_checked": n,
                            "of": "a train-mode forward + loss of the first %d pairs as their own batch" % n, "against": "CPU oracle, train mode"}
        del m2, out_h, loss_h
    result["fp16_split_overflowed"] = bool(model.fp16_overflowed())
    if stub:
        # every rank must leave the steps with the same parameters and BatchNorm buffers (gradient all-reduce + rank-0 buffer broadcast)
        chk = torch.stack([v.double().abs().sum() for v in model.state_dict().values() if v.is_floating_point()]).sum().reshape(1)
        if dist is not None:
            got = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(got, chk)
            result["stub_ranks_agree"] = bool(all(torch.equal(g, got[0]) for g in got))
        result["stub_state_checksum"] = float(chk)
    return result


if __name__ == "__main__":
    main()
