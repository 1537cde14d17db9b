mkdir -p gpurun_out/r4i
timeout 300 python tools/probes/gpu_kernel_share_any.py vitl 504 1 > gpurun_out/r4i/vitl_b1.txt 2>&1
timeout 300 python tools/probes/gpu_kernel_share_any.py vits 504 1 > gpurun_out/r4i/vits_b1.txt 2>&1
cat gpurun_out/r4i/vitl_b1.txt gpurun_out/r4i/vits_b1.txt
