mkdir -p gpurun_out/r4v
bash tools/probes/ab_libs.sh 2 python tools/probes/b1_time_latency.py 2>&1 | tee gpurun_out/r4v/ab_b1_splitkv_threshold.txt
