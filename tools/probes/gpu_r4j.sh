mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_conv3h.py -x -q -m gpu > gpurun_out/r4j/tests.log 2>&1
timeout 400 python tests/precision_budget/measure_on_gpu.py --labels bf16 fp16 "mixed (shipped)" "mixed + fusion_in x3 (whole decoder)" "mixed, no compensation" bf16x3 --out gpurun_out/r4j/budget_short.json > gpurun_out/r4j/budget.log 2>&1
tail -15 gpurun_out/r4j/tests.log
cat gpurun_out/r4j/budget.log
