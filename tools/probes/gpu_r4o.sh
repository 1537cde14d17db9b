mkdir -p gpurun_out/r4o
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r4o/gpu_tests.log 2>&1
tail -5 gpurun_out/r4o/gpu_tests.log
timeout 3000 bash tools/collect_profiles.sh r04 > gpurun_out/r4o/collect.log 2>&1
tail -60 gpurun_out/r4o/collect.log
