#!/bin/bash
# round 5, step h: power-of-two weight scale of the fp16 build (tests), kernel shares of the mixed mode with the LayerNorm-with-sums launch
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05h
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_gemm_fuzz.py tests/test_gpu_latency_mode.py tests/test_gpu_beit.py -q -m gpu -x 2>&1 | tail -30 > "$OUT/pytest.txt"
python tools/probes/gpu_kernel_share_any.py vitl 504 32 mixed 2>&1 | grep -v amdgpu > "$OUT/kernel_share_mixed.txt"
cat "$OUT/pytest.txt"; head -30 "$OUT/kernel_share_mixed.txt"
