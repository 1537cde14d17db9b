#!/bin/bash
# round 5, step a: two-pass (activation-split) classes - kernel tests, precision-mode tests, measured error / throughput table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05a
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_conv3h.py tests/test_gpu_precision_modes.py -x -q -m gpu 2>&1 | tail -15 > "$OUT/pytest.txt"
timeout 1200 python tests/precision_budget/measure_on_gpu.py --only "bf16" "fp16 (" "mixed" "fp16x3" --out "$OUT/precision_budget.json" > "$OUT/precision_budget.log" 2>&1
cat "$OUT/pytest.txt"; grep -v amdgpu "$OUT/precision_budget.log"
