out=gpurun_out/r5
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
BENCH="python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --secondary 0"
rocprofv3 --kernel-trace --stats -d $out/trace -o r --output-format rocpd -- $BENCH > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python3 tools/rocpd_stats.py $db > $out/kernel_stats_a.txt
python3 tools/rocpd_timeline.py $db "pack_clouds_kernel" | head -75 > $out/step_timeline_a.txt
rm -rf $out/trace
