mkdir -p gpurun_out/r4h
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/r4h/gpu_tests.log 2>&1
tail -30 gpurun_out/r4h/gpu_tests.log
