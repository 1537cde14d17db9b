mkdir -p gpurun_out/r4n
bash tools/probes/ab_libs.sh 3 python tools/probes/bench_value.py --model swinl --steps 20 --warmup 3 --no-profile 2>&1 | tee gpurun_out/r4n/ab_lnres.txt
