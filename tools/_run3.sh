set -x
mkdir -p gpurun_out/r6c
python -m pytest tests/test_hip_train.py -x -q -m gpu -k "small_bmm or weight_grad_thin" > gpurun_out/r6c/t_small.log 2>&1; tail -3 gpurun_out/r6c/t_small.log
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "attention_backward_kernel or knn_head_boundary" > gpurun_out/r6c/t_ops.log 2>&1; tail -3 gpurun_out/r6c/t_ops.log
python -m pytest tests/test_hip_forward.py -x -q -s -m gpu -k "configs3_at_its_full or graph_replay" > gpurun_out/r6c/t_fwd.log 2>&1; tail -3 gpurun_out/r6c/t_fwd.log
python -m pytest tests/test_hip_train.py -x -q -s -m gpu -k "training_step_matches_reference or feat_mean or match or thin_linear or loss" > gpurun_out/r6c/t_train.log 2>&1; tail -3 gpurun_out/r6c/t_train.log
python tools/train_op_census.py 8 > gpurun_out/r6c/census.txt 2>&1
OGMM_EDGECONV_PROBE=1 python tools/edgeconv_time.py > gpurun_out/r6c/edgeconv_probe.txt 2>&1
python bench.py --workload train --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r6c/bench_train.json 2> gpurun_out/r6c/bench_train.err; tail -c 400 gpurun_out/r6c/bench_train.json
