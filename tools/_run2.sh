set -x
mkdir -p gpurun_out/r6b
python tools/tail_bisect.py --first 256 --pairs 64 > gpurun_out/r6b/bisect_sharp_cfg1_256.txt 2>&1
python tools/tail_bisect.py --first 128 --pairs 64 > gpurun_out/r6b/bisect_sharp_cfg1_128.txt 2>&1
python tools/tail_bisect.py --workload n717 --first 300 --pairs 64 > gpurun_out/r6b/bisect_sharp_n717_300.txt 2>&1
python tools/tail_bisect.py --workload n717 --first 364 --pairs 64 > gpurun_out/r6b/bisect_sharp_n717_364.txt 2>&1
python -m pytest tests/test_hip_parity_tail.py -q -s -m gpu > gpurun_out/r6b/parity_tail.log 2>&1
python -m pytest tests/test_hip_forward.py tests/test_hip_ops.py tests/test_hip_train.py -x -q -m gpu -k "graph_replay or knn_head_boundary or attention_backward_kernel or weight_grad_thin or fused_instance_norm or knn_identical" > gpurun_out/r6b/new_tests.log 2>&1
python bench.py > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err
tail -3 gpurun_out/r6b/parity_tail.log; tail -3 gpurun_out/r6b/new_tests.log
