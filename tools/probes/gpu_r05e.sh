#!/bin/bash
# round 5, step e: phase stamps of the two-plane head tail kernel (debug build on the box only)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05e
mkdir -p "$OUT"
cd "$R"
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES python -c "from muggled_dpt_amd import native; native.build(force=True)" > "$OUT/build.log" 2>&1
python tools/probes/gpu_head_tail_phases.py 16 mixed 2>&1 | grep -v amdgpu > "$OUT/head_tail2_phases.txt"
python tools/probes/gpu_head_tail_phases.py 16 bf16 2>&1 | grep -v amdgpu >> "$OUT/head_tail2_phases.txt"
cat "$OUT/head_tail2_phases.txt"
