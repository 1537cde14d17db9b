mkdir -p gpurun_out/r4p
timeout 600 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_conv3h.py -q -m gpu -x > gpurun_out/r4p/tests.log 2>&1
tail -4 gpurun_out/r4p/tests.log
timeout 400 python tests/precision_budget/measure_on_gpu.py --labels bf16 fp16 "mixed (shipped)" "mixed + fusion_in x3 (whole decoder)" --out gpurun_out/r4p/budget_short.json 2>&1 | grep -v amdgpu
