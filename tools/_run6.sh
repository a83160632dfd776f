mkdir -p gpurun_out/r6f
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r6f/gpu_suite.log 2>&1; tail -5 gpurun_out/r6f/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6f/smoke.log 2>&1; tail -2 gpurun_out/r6f/smoke.log
