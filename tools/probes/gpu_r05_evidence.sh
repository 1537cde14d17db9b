#!/bin/bash
# Runs ON the MI355X box (gpurun): everything under profiles/r05_* in ONE pass at ONE source hash.
#   1. the GPU test suite (parity report -> gpurun_out/parity_report.json)
#   2. tools/collect_profiles.sh r05: rocprofv3 kernel stats (split / no split / x3 / fp16 / mixed), forward timeline, PMC passes (HBM traffic, SQ,
#      clock), plain bench lines of every configuration, precision budget table, batch-1 sweeps, SwinV2-L / BEiT-L stats + SwinV2-L SQ counters,
#      determinism screens, round-5 probes (mixed kernel shares, per-family class budget, fp16 weight scale, fp16-vs-bf16 MFMA power)
#   3. (debug build, LAST: it replaces the library on the box) per-phase stamps of head_tail_kernel / head_tail2_kernel
# Usage: gpurun --timeout 5400 -- 'bash tools/probes/gpu_r05_evidence.sh [notests]'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
mkdir -p gpurun_out/profiles_r05
python -c "from muggled_dpt_amd import native; print('source hash', native.source_hash())" > gpurun_out/profiles_r05/source_hash.txt 2>&1
if [ "${1:-}" != "notests" ]; then
  timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > gpurun_out/profiles_r05/pytest_gpu.txt
fi
bash tools/collect_profiles.sh r05 > gpurun_out/profiles_r05/collect.log 2>&1
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES python -c "from muggled_dpt_amd import native; native.build(force=True)" > gpurun_out/profiles_r05/debug_build.log 2>&1
{ python tools/probes/gpu_head_tail_phases.py 16 mixed; python tools/probes/gpu_head_tail_phases.py 16 bf16; } 2>&1 | grep -v amdgpu > gpurun_out/profiles_r05/head_tail_phases.txt
cat gpurun_out/profiles_r05/source_hash.txt gpurun_out/profiles_r05/pytest_gpu.txt 2>/dev/null; ls gpurun_out/profiles_r05
