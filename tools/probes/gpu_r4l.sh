mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_gpu_swinv2.py tests/test_gpu_attention_maps.py tests/test_gpu_model_fuzz.py -q -m gpu -x > gpurun_out/r4l/tests.log 2>&1
tail -6 gpurun_out/r4l/tests.log
timeout 300 python bench.py --model swinl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4l/bench_swinl.json 2> gpurun_out/r4l/bench_swinl.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4l/bench_swinl.json'))
print(d['value'], d['ms_per_step'])
for k,v in list(d['kernel_time_share'].items())[:8]: print(f"  {v:.3f} {k}")
PY
