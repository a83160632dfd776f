#!/bin/bash
# Runs on the GPU box (gpurun): everything the round's profiles/ files are made of, written under gpurun_out/collect/.
#   usage: /usr/local/graft/bin/gpurun --timeout 2400 -- "bash tools/collect_profiles.sh round3 $(git rev-parse --short HEAD)"
# rocprofv3 is given the program itself after `--` (python3 bench.py ...); counters are collected in their own passes
# (--kernel-trace --pmc only), as MI355X_MICROARCH.md prescribes.
tag=${1:-round}
commit=${2:-unknown}          # the tree's commit (the box has no .git): goes into every summary's header
out=gpurun_out/collect
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0          # before anything (rocprofv3's preloaded library included) initialises the HIP runtime: ogmm_amd.graph_replay_safe()
BENCH="python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --secondary 0"

# 1. headline bench line (with the CPU baseline and the parity sample) and the other workloads
timeout 600 python3 bench.py --steps 20 --warmup 5 2> $out/bench_n1.err | tail -1 > $out/${tag}_bench_n1.json
for w in cfg2 cfg3; do timeout 400 python3 bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 0 --secondary 0 2>/dev/null | tail -1 > $out/${tag}_bench_$w.json; done
timeout 500 python3 bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/${tag}_train_bench_b128.json
OGMM_TRAIN_GRAPH=0 timeout 500 python3 bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/${tag}_train_bench_b128_eager.json

# 2. kernel trace + stats of the headline command
rocprofv3 --kernel-trace --stats -d $out/trace -o r --output-format rocpd -- $BENCH > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
{ echo "# commit $commit"; echo "# rocprofv3 --kernel-trace --stats -- $BENCH   (9 forwards: 2 warm-up + 1 counting + 5 timed + ...; the first one also packs the weights)"; python3 tools/rocpd_stats.py $db; } > $out/${tag}_kernel_stats.txt
{ echo "# one eval step (B=64, N=1024, J=16) as dispatched: start, gap to the previous kernel's end (negative: overlapped with a side stream), duration, grid"; python3 tools/rocpd_timeline.py $db "pack_clouds_kernel" | head -70; } > $out/${tag}_step_timeline.txt

# 3. PMC passes (separate runs)
{
echo "# commit $commit"
echo "# rocprofv3 --kernel-trace --pmc <counters> -- $BENCH   (separate passes per counter set; per-dispatch means, summed over the XCD instances rocprofv3 reports)"
echo "# FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> bench.py doubles it."
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    d=$out/pmc_$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass -d $d -o r --output-format rocpd -- $BENCH > $d.log 2>&1
    f=$(find $d -name "*.db" | head -1)
    echo "## pass: $pass"
    python3 tools/rocpd_pmc.py $f | head -14
done
} > $out/${tag}_pmc_counters.txt

# 4. GEMM engine: variants and ablations with the in-kernel clock probes (cycles per tile, shader clock)
export OGMM_V6_MIN_TILES=1 OGMM_V4_MIN_TILES=1 OGMM_V8_MIN_TILES=1 OGMM_V10_MIN_TILES=1
{
echo "# tools/gemm_v6_check.py: register-staged engine (v23 = gemm_f16x3_v4), LDS-DMA engines: v60 = gemm_f16x3_v6 (4x2 waves), v100 = gemm_f16x3_v8 (8x1 waves), v110 = gemm_f16x3_v10 (4 waves of 64 x 256, the default from N = 512); x1 = without output stores"
timeout 300 python3 tools/gemm_v6_check.py --time-only 23 60 100 110 101 111 2>&1 | grep TF
echo "# clock probes (131072 x 1024 x 1024): v6 ablations 80 full, 83 MFMA + barrier, 84 MFMA only, 85 MFMA only on zeros; v8: 102 full, 103 no stores, 104 no DMA; v10: 112 full, 113 no stores, 114 no DMA,"
echo "#   115 no split arithmetic, 116 no weight-fragment reads, 117 DMA + MFMA + barrier, 118 MFMA + barrier, 119 fragment reads + MFMA"
timeout 300 python3 tools/gemm_v6_check.py --time-only --clock 80 83 84 85 102 103 104 112 113 114 115 116 117 118 119 2>&1 | tail -15
echo "# one tile per workgroup on 64 / 128 / 256 CUs and whole rounds beyond (v10 full, v10 without stores, v8 full, v8 without stores)"
timeout 200 python3 tools/gemm_v6_check.py --time-only --grid-sweep 110 111 100 101 2>&1 | tail -5
echo "# every GEMM launch of one eval forward (tools/gemm_launch_table.py)"
timeout 200 python3 tools/gemm_launch_table.py 2>&1 | tail -27
} > $out/${tag}_gemm_engine.txt
unset OGMM_V6_MIN_TILES OGMM_V4_MIN_TILES OGMM_V8_MIN_TILES OGMM_V10_MIN_TILES

# 4b. the training step (BASELINE configs[4], 128 pairs per GPU): kernel statistics, per-operation breakdown, attention backward alone
rocprofv3 --kernel-trace --stats -d $out/trace_train -o r --output-format rocpd -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0 > $out/trace_train.log 2>&1
dbt=$(find $out/trace_train -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0   (training steps of 128 pairs: 2 eager + the recording one + 1 warm-up replay + 3 timed replays + 1 eager bracketed step)"; python3 tools/rocpd_stats.py $dbt | head -80; } > $out/${tag}_train_kernel_stats.txt
{ echo "# tools/train_breakdown.py 128: forward / backward of the autograd functions of one training step (events)"; timeout 300 python3 tools/train_breakdown.py 128 2>&1 | grep -v amdgpu.ids;
  echo "# tools/attn_bwd_time.py"; timeout 200 python3 tools/attn_bwd_time.py 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_train_breakdown.txt
rm -rf $out/trace_train

# 5. parity: the distribution over every pair of a batch per workload on both weight families, then the PARITY lines of the GPU tests
{ echo "# commit $commit"; timeout 1800 python3 tools/parity_distribution.py --workloads cfg1,cfg2,cfg3,n717 --pairs 256,64,32,128 2>&1 | grep -v amdgpu.ids;
  echo; echo "# PARITY lines of pytest -m gpu (tests/test_hip_forward.py, test_hip_parity_tail.py, test_hip_deepgmr.py, test_hip_icp.py)";
  timeout 1500 python3 -m pytest tests/test_hip_forward.py tests/test_hip_parity_tail.py tests/test_hip_deepgmr.py tests/test_hip_icp.py -m gpu -q -s 2>&1 | grep -E "PARITY|TRAINED|passed|failed"; } > $out/${tag}_parity.txt
{ echo "# commit $commit"; echo "# weight family sharp (synth.fill_state_dict(profile='sharp')): the shipped budget on the full windows, then three terms everywhere and the exact-fp32 engine on the SAME windows (round 5: configs[1] pairs 0..127 incl. 75 / 84 / 112, N = 717 pairs 300..427), round 3's budget for the record (NOT parity-safe)";
  timeout 900 python3 tools/parity_distribution.py --profile sharp --workloads cfg1,cfg2,n717 --pairs 128,32,128 2>&1 | grep -v amdgpu.ids;
  timeout 900 python3 tools/parity_distribution.py --profile sharp --budget none --workloads cfg1,n717 --pairs 128,128 2>&1 | grep -v amdgpu.ids;
  timeout 900 python3 tools/parity_distribution.py --profile sharp --precision f32 --workloads cfg1,n717 --pairs 128,128 2>&1 | grep -v amdgpu.ids;
  timeout 600 python3 tools/parity_distribution.py --profile sharp --budget r3 --workloads cfg1 --pairs 64 2>&1 | grep -v amdgpu.ids;
  echo; echo "# tools/parity_probe.py: pairs 64..127 in one batch, four arithmetics, the reference's own probes, stage by stage";
  timeout 900 python3 tools/parity_probe.py --first 64 --pairs 64 --ids 75,84,112,99,124 2>&1 | grep -v amdgpu.ids;
  echo; echo "# room clouds (configs[3] shape), both weight families, 32 pairs each";
  timeout 900 python3 tools/parity_distribution.py --workloads cfg3 --pairs 32 2>&1 | grep -v amdgpu.ids;
  timeout 900 python3 tools/parity_distribution.py --profile sharp --workloads cfg3 --pairs 32 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_parity_sharp.txt
# 6. a weight family that has left the initial regime: 5000 training steps, regime report, every-pair parity (tools/parity_trained.py)
{ echo "# commit $commit"; timeout 1500 python3 tools/parity_trained.py --steps 5000 --pairs 128 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_parity_trained.txt
# 7. run-to-run reproducibility of consecutive forwards; the kNN head and the cluster-mean kernel alone
{ echo "# commit $commit"; timeout 300 python3 tools/determinism_check.py 2>&1 | grep -v amdgpu.ids; timeout 300 python3 tools/fps_corun.py 2>&1 | grep -v amdgpu.ids;
  timeout 200 python3 tools/knn_time.py 2>&1 | grep -v amdgpu.ids; timeout 100 python3 tools/featmean_time.py 2>&1 | grep -v amdgpu.ids; timeout 100 python3 tools/host_time.py 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_head_and_determinism.txt
# 8. (round 5) the packed-fp32 / f16-matrix co-run hazard: the standalone reproducer, the guard kernels beside the product GEMMs, the product kernels beside them,
#    the thin weight gradient and the normalisation backward per shape
{ echo "# commit $commit"; echo "== tools/pk_mfma_hazard.hip (no product code)";
  hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-slp-vectorize tools/pk_mfma_hazard.hip -o /tmp/pk_mfma_hazard 2>/dev/null && timeout 300 /tmp/pk_mfma_hazard;
  echo; echo "== tools/corun_guard.py (tools/lds_guard.hip beside the product GEMMs)"; timeout 300 python3 tools/corun_guard.py 2>&1 | grep -v amdgpu.ids | head -24;
  echo; echo "== tools/corun_debug.py (library as built)"; timeout 300 python3 tools/corun_debug.py 2>&1 | grep -v amdgpu.ids;
  echo; echo "== tools/corun_victims.py (library as built)"; timeout 300 python3 tools/corun_victims.py 2>&1 | grep -v amdgpu.ids | tail -20; } > $out/${tag}_pk_mfma_hazard_rerun.txt
{ echo "# commit $commit"; timeout 300 python3 tools/dw_thin_time.py 2>&1 | grep -v amdgpu.ids; timeout 300 python3 tools/norm_bwd_time.py 2>&1 | grep -v amdgpu.ids;
  timeout 300 python3 tools/train_call_census.py 128 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_train_kernels.txt
rm -rf $out/trace $out/pmc_*          # the rocpd databases exceed what gpurun copies back; the summaries above are what gets committed
ls -la $out | head -40
