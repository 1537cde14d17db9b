#!/bin/bash
# Round 6, last session: A/B of the fc1 epilogue's erf (A = Abramowitz-Stegun 7.1.26 with rcp + exp2, B = 1 - 2^(-|v| q(|v|)) with one exp2) on ONE box.
# tools/probes/_bin/libmdpt_{A,B}.so are the two builds. Leaves B in the tree and runs the GELU-touching parity tests with it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; mkdir -p gpurun_out
{
bash tools/probes/ab_libs.sh 2 python tools/probes/gpu_gelu_cost.py
bash tools/probes/ab_libs.sh 2 python tools/probes/bench_value.py --steps 20 --warmup 3
for v in A B; do
  cp tools/probes/_bin/libmdpt_$v.so muggled_dpt_amd/csrc/libmdpt.so
  echo "== kernel share $v"; python tools/probes/gpu_kernel_share_any.py vitl 504 32 2>&1 | grep -v amdgpu | head -7
done
} > gpurun_out/gelu_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_gemm_fuzz.py tests/test_gpu_parity.py tests/test_gpu_precision_modes.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/gelu_ab_tests.txt
cat gpurun_out/gelu_ab.txt gpurun_out/gelu_ab_tests.txt
