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
{ echo "# commit $commit"; echo "# one eval step (B=64, N=1024, J=16) as dispatched: start, gap to the previous kernel's end (negative: overlapped with a side stream), duration, grid"; python3 tools/rocpd_timeline.py $db "pack_clouds_kernel" | head -70; } > $out/${tag}_step_timeline.txt
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
rm -rf $out/trace $out/pmc_*
ls -la $out | grep $tag
