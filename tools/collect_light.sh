#!/bin/bash
# The cheap half of tools/collect_profiles.sh (bench lines, kernel trace, step timeline, PMC passes, head / determinism tools) at the current tree:
#   /usr/local/graft/bin/gpurun --timeout 1800 -- "bash tools/collect_light.sh round5 $(git rev-parse --short HEAD)"
tag=${1:-round}
commit=${2:-unknown}
out=gpurun_out/collect
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
BENCH="python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --secondary 0"
timeout 900 python3 bench.py --steps 20 --warmup 5 2> $out/bench_n1.err | tail -1 > $out/${tag}_bench_n1.json
rocprofv3 --kernel-trace --stats -d $out/trace -o r --output-format rocpd -- $BENCH > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
{ echo "# commit $commit"; echo "# rocprofv3 --kernel-trace --stats -- $BENCH   (9 forwards: 2 warm-up + 1 counting + 5 timed + ...; the first one also packs the weights)"; python3 tools/rocpd_stats.py $db; } > $out/${tag}_kernel_stats.txt
# (the timeline from a SERIAL run: with the bench's pipelined head the kernels of consecutive forwards interleave)
rocprofv3 --kernel-trace --stats -d $out/trace_s -o r --output-format rocpd -- $BENCH --pipeline-head 0 > $out/trace_s.log 2>&1
dbs=$(find $out/trace_s -name "*.db" | head -1)
{ echo "# commit $commit"; echo "# one SERIAL eval step (B=64, N=1024, J=16; $BENCH --pipeline-head 0) as dispatched: start, gap to the previous kernel's end (negative: overlapped with a side stream), duration, grid"; python3 tools/rocpd_timeline.py $dbs "pack_clouds_kernel" | head -52; } > $out/${tag}_step_timeline.txt
rm -rf $out/trace_s
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
{ echo "# commit $commit"; timeout 300 python3 tools/determinism_check.py 2>&1 | grep -v amdgpu.ids; timeout 300 python3 tools/fps_corun.py 2>&1 | grep -v amdgpu.ids;
  timeout 200 python3 tools/knn_time.py 2>&1 | grep -v amdgpu.ids; timeout 100 python3 tools/featmean_time.py 2>&1 | grep -v amdgpu.ids; timeout 100 python3 tools/host_time.py 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_head_and_determinism.txt
# the other workloads and the training step (kernel statistics, per-operation breakdown, the training kernels alone)
for w in cfg2 cfg3; do timeout 400 python3 bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 0 --secondary 0 2>/dev/null | tail -1 > $out/${tag}_bench_$w.json; done
timeout 500 python3 bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/${tag}_train_bench_b128.json
OGMM_TRAIN_GRAPH=0 timeout 500 python3 bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/${tag}_train_bench_b128_eager.json
rocprofv3 --kernel-trace --stats -d $out/trace_train -o r --output-format rocpd -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0 > $out/trace_train.log 2>&1
dbt=$(find $out/trace_train -name "*.db" | head -1)
{ echo "# commit $commit"; echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0   (training steps of 128 pairs: 2 eager + the recording one + 1 warm-up replay + 3 timed replays + 1 eager bracketed step)"; python3 tools/rocpd_stats.py $dbt; } > $out/${tag}_train_kernel_stats.txt
{ echo "# commit $commit"; echo "# tools/train_breakdown.py 128: forward / backward of the autograd functions of one training step (events)"; timeout 300 python3 tools/train_breakdown.py 128 2>&1 | grep -v amdgpu.ids;
  echo "# tools/attn_bwd_time.py"; timeout 200 python3 tools/attn_bwd_time.py 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_train_breakdown.txt
{ echo "# commit $commit"; timeout 300 python3 tools/dw_thin_time.py 2>&1 | grep -v amdgpu.ids; timeout 300 python3 tools/norm_bwd_time.py 2>&1 | grep -v amdgpu.ids;
  timeout 300 python3 tools/train_call_census.py 128 2>&1 | grep -v amdgpu.ids; } > $out/${tag}_train_kernels.txt
rm -rf $out/trace $out/pmc_* $out/trace_train
ls -la $out | grep $tag
