mkdir -p gpurun_out/r4k
timeout 1200 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_block_hooks.py tests/test_gpu_parity.py tests/test_gpu_conv3h.py tests/test_gpu_swinv2.py -q -m gpu -x > gpurun_out/r4k/tests.log 2>&1
tail -15 gpurun_out/r4k/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err
tail -c 6000 gpurun_out/r4k/bench.json
