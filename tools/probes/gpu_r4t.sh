mkdir -p gpurun_out/r4t
bash tools/probes/ab_libs.sh 3 python tools/probes/b1_time.py 2>&1 | tee gpurun_out/r4t/ab_b1_conv_ring3.txt
