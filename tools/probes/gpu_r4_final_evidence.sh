#!/bin/bash
# Runs ON the MI355X box: evidence that needs no source change - rocprofv3 kernel stats of the configs[4] models (batch split off), the run-to-run
# determinism screens of the default path (ViT-L batch 32 / 16, both modes, with and without the two-stream split) and of latency mode, the SwinV2-L
# kernel shares after the K split of fc2. Output: gpurun_out/final_evidence/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/final_evidence
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for m in swinl beitl; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$m" -- python "$R/bench.py" --model $m --no-split --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_${m}_nosplit_under_rocprof.json" 2> "$OUT/$m.log"
  f=$(find "$OUT/$m" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_${m}_nosplit.csv"
  rm -rf "$OUT/$m"
done
cd "$R"
python tools/probes/gpu_determinism_stress.py 60 2>&1 | grep -v amdgpu > "$OUT/determinism_default.txt"
python tools/probes/gpu_ksplit_determinism.py 300 2>&1 | grep -v amdgpu > "$OUT/determinism_latency.txt"
python tools/probes/gpu_kernel_share_any.py swinl 384 16 2>&1 | grep -v amdgpu > "$OUT/kernel_share_swinl.txt"
ls -la "$OUT"
