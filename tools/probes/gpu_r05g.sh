#!/bin/bash
# round 5, step g: whole GPU suite (export / host, prepare_image dtype, grid cache, LayerNorm-with-sums), mixed rows, batch-1 legs
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05g
mkdir -p "$OUT"
cd "$R"
timeout 1800 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -30 > "$OUT/pytest.txt"
timeout 600 python tests/precision_budget/measure_on_gpu.py --only "bf16" "mixed (shipped)" "mixed, no comp" "fp16 (" --out "$OUT/precision_budget.json" > "$OUT/precision_budget.log" 2>&1
cat "$OUT/pytest.txt"; grep -v amdgpu "$OUT/precision_budget.log"
