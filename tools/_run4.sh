set -x
mkdir -p gpurun_out/r6d
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "edgeconv or attention_backward_kernel" > gpurun_out/r6d/t_ops.log 2>&1; tail -3 gpurun_out/r6d/t_ops.log
OGMM_EDGECONV_PROBE=1 python tools/edgeconv_time.py > gpurun_out/r6d/edgeconv_probe.txt 2>&1; grep -v amdgpu.ids gpurun_out/r6d/edgeconv_probe.txt
python tools/edgeconv_time.py > gpurun_out/r6d/edgeconv_time.txt 2>&1; grep "edgeconv \|bit-id" gpurun_out/r6d/edgeconv_time.txt
python -m pytest tests/test_hip_train.py -x -q -s -m gpu > gpurun_out/r6d/t_train.log 2>&1; tail -3 gpurun_out/r6d/t_train.log
python tools/train_op_census.py 8 > gpurun_out/r6d/census.txt 2>&1
python bench.py --cpu-sample 0 --secondary 0 > gpurun_out/r6d/bench_eval.json 2> gpurun_out/r6d/bench_eval.err; tail -c 300 gpurun_out/r6d/bench_eval.json
python bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r6d/bench_train.json 2> gpurun_out/r6d/bench_train.err; tail -c 400 gpurun_out/r6d/bench_train.json
