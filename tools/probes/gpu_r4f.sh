mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_precision_modes.py -x -q -m gpu > gpurun_out/r4f/prec_tests.log 2>&1
timeout 300 python tests/precision_budget/measure_on_gpu.py --labels bf16 fp16 "fp16, no compensation" "mixed (shipped)" "mixed, no compensation" --out gpurun_out/r4f/budget_short.json > gpurun_out/r4f/budget.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f/prof -- python tests/precision_budget/measure_on_gpu.py --labels fp16 --no-split --steps 5 > gpurun_out/r4f/prof_run.log 2>&1
tail -25 gpurun_out/r4f/prec_tests.log
cat gpurun_out/r4f/budget.log
find gpurun_out/r4f/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-170
