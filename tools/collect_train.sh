out=gpurun_out/collect
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 500 python3 bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/round3_train_bench_b128.json
OGMM_TRAIN_GRAPH=0 timeout 500 python3 bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $out/round3_train_bench_b128_eager.json
rocprofv3 --kernel-trace --stats -d $out/trace_train -o r --output-format rocpd -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0 > $out/trace_train.log 2>&1
dbt=$(find $out/trace_train -name "*.db" | head -1)
{ echo "# commit $1"; echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --steps 3 --warmup 2 --cpu-sample 0   (training steps of 128 pairs: 2 eager + 1 recording + 1 warm-up replay + 3 timed replays + 1 eager bracketed step)"; python3 tools/rocpd_stats.py $dbt | head -80; } > $out/round3_train_kernel_stats.txt
{ echo "# tools/train_breakdown.py 128: forward / backward of the autograd functions of one training step (events)"; timeout 300 python3 tools/train_breakdown.py 128 2>&1 | grep -v amdgpu.ids;
  echo "# tools/attn_bwd_time.py"; timeout 200 python3 tools/attn_bwd_time.py 2>&1 | grep -v amdgpu.ids; } > $out/round3_train_breakdown.txt
rm -rf $out/trace_train
cut -c1-160 $out/round3_train_bench_b128.json; cut -c1-160 $out/round3_train_bench_b128_eager.json; head -6 $out/round3_train_kernel_stats.txt | cut -c1-140
