set -x
out=gpurun_out/r6e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_hip_train.py -x -q -m gpu > $out/t_train.log 2>&1; tail -3 $out/t_train.log
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "edgeconv or attention_backward" > $out/t_ops.log 2>&1; tail -2 $out/t_ops.log
python bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/train_bench.json; tail -c 300 $out/train_bench.json
rocprofv3 --kernel-trace --stats -d $out/trace_train -o r --output-format rocpd -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0 > $out/trace_train.log 2>&1
dbt=$(find $out/trace_train -name "*.db" | head -1)
python3 tools/rocpd_stats.py $dbt > $out/train_kernel_stats.txt
rm -rf $out/trace_train
grep -c "Cijk\|rocprim\|indexFunc\|indexSelect" $out/train_kernel_stats.txt
grep "Cijk\|rocprim\|indexFunc\|indexSelect\|small_bmm\|scatter_add" $out/train_kernel_stats.txt | cut -c1-150
head -30 $out/train_kernel_stats.txt | cut -c1-150
