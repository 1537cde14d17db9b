#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05d
mkdir -p "$OUT"
cd "$R"
python tools/probes/gpu_family_class_budget.py beitl swinl 2>&1 | grep -v amdgpu > "$OUT/family_class_budget.txt"
cat "$OUT/family_class_budget.txt"
