#!/bin/bash
# round 5, step f: whole GPU suite after the two-pass classes / head_tail2 / LayerNorm-with-means, mixed rows of the budget table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05f
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -30 > "$OUT/pytest.txt"
timeout 600 python tests/precision_budget/measure_on_gpu.py --only "bf16" "mixed (shipped)" "mixed of round" "mixed, no comp" --out "$OUT/precision_budget.json" > "$OUT/precision_budget.log" 2>&1
cat "$OUT/pytest.txt"; grep -v amdgpu "$OUT/precision_budget.log"
