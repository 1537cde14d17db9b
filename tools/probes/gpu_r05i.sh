#!/bin/bash
# round 5, step i: fp16 weight scale tests, then the default bench line (new legs: inference_b1, 1036 mixed + error)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/r05i
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_latency_mode.py tests/test_gpu_beit.py tests/test_gpu_swinv2.py -q -m gpu -x 2>&1 | tail -30 > "$OUT/pytest.txt"
timeout 900 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cat "$OUT/pytest.txt"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05i/bench_n1.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "path", d["roofline"].get("path_frac"))
for k in ("mixed_mode","fp16_mode","fp32_class_mode"):
    print(k, d[k]["value"], d[k]["error_vs_cpu_fp32"])
for k,v in d["secondary"].items():
    print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","latency_mode","inference_b1","mixed_mode","error_vs_cpu_fp32","leg_seconds","error")})
PY
