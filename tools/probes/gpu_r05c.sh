#!/bin/bash
# round 5, step c: the two-plane head tail kernel - tests, mixed-mode table rows, kernel shares
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05c
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_precision_modes.py -x -q -m gpu 2>&1 | tail -25 > "$OUT/pytest.txt"
timeout 1200 python tests/precision_budget/measure_on_gpu.py --only "bf16" "mixed (shipped)" "mixed of round" "head_tail" "mixed, head =" "mixed, fusion =" --out "$OUT/precision_budget.json" > "$OUT/precision_budget.log" 2>&1
python tools/probes/gpu_kernel_share_any.py vitl 504 32 mixed 2>&1 | grep -v amdgpu > "$OUT/kernel_share_mixed.txt"
cat "$OUT/pytest.txt"; grep -v amdgpu "$OUT/precision_budget.log"; head -14 "$OUT/kernel_share_mixed.txt"
