#!/bin/bash
# round 5, step b: kernel tests of the two-pass form, kernel shares of the mixed mode (batch split off)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05b
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_conv3h.py tests/test_gpu_precision_modes.py tests/test_gpu_gemm_fuzz.py -x -q -m gpu 2>&1 | tail -15 > "$OUT/pytest.txt"
python tools/probes/gpu_kernel_share_any.py vitl 504 32 mixed 2>&1 | grep -v amdgpu > "$OUT/kernel_share_mixed.txt"
python tools/probes/gpu_kernel_share_any.py vitl 504 32 fp16 2>&1 | grep -v amdgpu > "$OUT/kernel_share_fp16.txt"
cat "$OUT/pytest.txt" "$OUT/kernel_share_mixed.txt"
