mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4g/gpu_tests.log 2>&1
tail -30 gpurun_out/r4g/gpu_tests.log
