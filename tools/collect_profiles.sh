#!/bin/bash
# Runs ON the MI355X box (through gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   1. kernel trace + stats of the default bench command (two-stream batch split), of `--no-split`, and of the bf16x3 (fp32-class) mode
#   2. PMC passes (separate runs, --kernel-trace only): FETCH_SIZE, WRITE_SIZE -> HBM traffic per launch; SQ wave-state counters;
#      GRBM_GUI_ACTIVE -> effective clock per kernel
#      plus one forward as a launch-by-launch timeline (tools/forward_timeline.py)
#   3. the driver-runnable secondary configurations (1036x1036, BEiT-L, SwinV2-L) as plain bench lines
# Output: gpurun_out/profiles_<tag>/ ; copy what is to be judged into profiles/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/split" -- python "$R/bench.py" --steps 10 --warmup 3 --no-secondary > "$OUT/bench_under_rocprof.json" 2> "$OUT/split.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/nosplit" -- python "$R/bench.py" --no-split --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_nosplit_under_rocprof.json" 2> "$OUT/nosplit.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/x3" -- python "$R/bench.py" --precision bf16x3 --no-split --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_x3_nosplit_under_rocprof.json" 2> "$OUT/x3.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fp16" -- python "$R/bench.py" --precision fp16 --no-split --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_fp16_nosplit_under_rocprof.json" 2> "$OUT/fp16.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mixed" -- python "$R/bench.py" --precision mixed --no-split --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_mixed_nosplit_under_rocprof.json" 2> "$OUT/mixed.log"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/timeline" -- python "$R/bench.py" --no-split --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> "$OUT/timeline.log"
python "$R/tools/forward_timeline.py" "$OUT/timeline" "$OUT/forward_timeline.md" > /dev/null 2>> "$OUT/timeline.log"
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python "$R/bench.py" --no-split --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> "$OUT/pmc_$C.log"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT/pmc_SQ" -- python "$R/bench.py" --no-split --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> "$OUT/pmc_SQ.log"
python "$R/tools/summarize_sq.py" "$OUT/pmc_SQ" "$OUT/sq_counters.md" > "$OUT/sq_counters.log" 2>&1
python "$R/tools/summarize_clock.py" "$OUT/pmc_GRBM_GUI_ACTIVE" "$OUT/effective_clock.md" > "$OUT/effective_clock.log" 2>&1
for d in split nosplit x3 fp16 mixed; do
  f=$(find "$OUT/$d" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${d}_kernel_stats.csv"
done
python "$R/tools/summarize_pmc.py" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/hbm_traffic" > "$OUT/hbm_traffic.log" 2>&1
# the raw per-dispatch traces are large; keep the summaries only
rm -rf "$OUT/split" "$OUT/nosplit" "$OUT/x3" "$OUT/fp16" "$OUT/mixed" "$OUT/timeline" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_SQ" "$OUT/pmc_GRBM_GUI_ACTIVE"
cd "$R"
python bench.py --steps 20 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python bench.py --precision bf16x3 --steps 20 --warmup 3 > "$OUT/bench_x3.json" 2> "$OUT/bench_x3.err"
python bench.py --precision mixed --steps 20 --warmup 3 > "$OUT/bench_mixed.json" 2> "$OUT/bench_mixed.err"
python bench.py --precision fp16 --steps 20 --warmup 3 > "$OUT/bench_fp16.json" 2> "$OUT/bench_fp16.err"
# round 4: measured error / throughput table of the arithmetic modes, batch-1 tile sweep, small-batch conv dispatch sweep, batch-1 kernel shares
python tests/precision_budget/measure_on_gpu.py --out "$OUT/precision_budget.json" > "$OUT/precision_budget.log" 2>&1
python tests/precision_budget/measure_on_gpu.py --render "$OUT/precision_budget.json" > "$OUT/precision_budget.md" 2>> "$OUT/precision_budget.log"
python tools/probes/gpu_b1_tile_sweep.py 2>&1 | grep -v amdgpu.ids > "$OUT/b1_tile_sweep.txt"
python tools/probes/gpu_conv3h_small_batch.py 2>&1 | grep -v amdgpu.ids > "$OUT/conv3h_small_batch.txt"
{ python tools/probes/gpu_kernel_share_any.py vitl 504 1; python tools/probes/gpu_kernel_share_any.py vits 504 1; } 2>&1 | grep -v amdgpu.ids > "$OUT/kernel_share_b1.txt"
python bench.py --size 1036 --steps 10 --warmup 2 > "$OUT/bench_1036.json" 2> "$OUT/bench_1036.err"
python bench.py --model beitl --steps 10 --warmup 2 > "$OUT/bench_beitl.json" 2> "$OUT/bench_beitl.err"
python bench.py --model swinl --steps 10 --warmup 2 > "$OUT/bench_swinl.json" 2> "$OUT/bench_swinl.err"
# SwinV2-L / BEiT-L kernel stats (batch split off), SQ counters of the SwinV2-L forward (window attention), determinism screens, kernel shares
cd /tmp
for m in swinl beitl; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$m" -- python "$R/bench.py" --model $m --no-split --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_${m}_nosplit_under_rocprof.json" 2> "$OUT/$m.log"
  f=$(find "$OUT/$m" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_${m}_nosplit.csv"
  rm -rf "$OUT/$m"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT/pmc_SQ_swinl" -- python "$R/bench.py" --model swinl --no-split --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> "$OUT/pmc_SQ_swinl.log"
python "$R/tools/summarize_sq.py" "$OUT/pmc_SQ_swinl" "$OUT/sq_counters_swinl.md" > "$OUT/sq_counters_swinl.log" 2>&1
rm -rf "$OUT/pmc_SQ_swinl"
cd "$R"
python tools/probes/gpu_determinism_stress.py 60 2>&1 | grep -v amdgpu > "$OUT/determinism_default.txt"
python tools/probes/gpu_ksplit_determinism.py 300 2>&1 | grep -v amdgpu > "$OUT/determinism_latency.txt"
python tools/probes/gpu_kernel_share_any.py swinl 384 16 2>&1 | grep -v amdgpu > "$OUT/kernel_share_swinl.txt"
# round 5: kernel shares of the mixed mode, per-family class budget of the mixed table, fp16 weight-scale check, fp16-vs-bf16 MFMA power probe
python tools/probes/gpu_kernel_share_any.py vitl 504 32 mixed 2>&1 | grep -v amdgpu > "$OUT/kernel_share_mixed.txt"
python tools/probes/gpu_family_class_budget.py beitl swinl 2>&1 | grep -v amdgpu > "$OUT/family_class_budget.txt"
python tools/probes/gpu_wscale_check.py 2>&1 | grep -v amdgpu > "$OUT/wscale_check.txt"
if [ -x tools/probes/_bin/mfma_power ] && [ -x tools/probes/_bin/mfma_power_f16 ]; then
  { echo "== bf16"; tools/probes/_bin/mfma_power | head -3; echo "== fp16"; tools/probes/_bin/mfma_power_f16 | head -3; } > "$OUT/mfma_power_f16_vs_bf16.txt" 2>&1
fi
# round 6: fp8 cross-term forms (instruction probes, per-class table under both roundings of the fp16 weight scale), what the token-mean compensation
# costs under the two-stream split, the fused inference path
python tools/probes/gpu_family_class_budget.py beitl swinl 2>&1 | grep -v amdgpu > "$OUT/family_class_budget_default_rows.txt"
MDPT_BUDGET_R06=1 python tools/probes/gpu_family_class_budget.py beitl 2>&1 | grep -v amdgpu > "$OUT/beitl_class_budget_both_roundings.txt"
python tools/probes/gpu_wrc_cost.py 2>&1 | grep -v amdgpu > "$OUT/wrc_cost.txt"
python tools/probes/gpu_kernel_share_any.py vitl 504 32 fp16 2>&1 | grep -v amdgpu > "$OUT/kernel_share_fp16.txt"
python tools/probes/gpu_kernel_share_any.py beitl 384 16 2>&1 | grep -v amdgpu > "$OUT/kernel_share_beitl.txt"
[ -x tools/probes/_bin/f8_cross_probe ] && tools/probes/_bin/f8_cross_probe > "$OUT/f8_cross_probe.txt" 2>&1
[ -x tools/probes/_bin/f8_shape_equiv_probe ] && tools/probes/_bin/f8_shape_equiv_probe > "$OUT/f8_shape_equiv_probe.txt" 2>&1
# side stream: hardware-queue collisions (first stream handed out vs probed), reassembly branches beside the encoder at batch 1
python tools/probes/gpu_side_stream_queue.py 2>&1 | grep -v amdgpu > "$OUT/side_stream_queue.txt"
python tools/probes/b1_overlap_ab.py 2>&1 | grep -v amdgpu > "$OUT/b1_overlap_ab.txt"
[ -x tools/probes/_bin/exp_throughput ] && tools/probes/_bin/exp_throughput > "$OUT/exp_throughput.txt" 2>&1
# non-finite propagation (mdpt_set_nonfinite_propagation): cost of its memset + launch
python tools/probes/gpu_nonfinite_cost.py 2>&1 | grep -v amdgpu > "$OUT/nonfinite_cost.txt"
ls -la "$OUT"
