mkdir -p gpurun_out/r4u
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES python -c "from muggled_dpt_amd import native; native.build(force=True)" > gpurun_out/r4u/build.log 2>&1
tail -3 gpurun_out/r4u/build.log
MDPT_SWEEP_TILE7=1 python tools/probes/gpu_b1_tile_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4u/b1_sweep_tile7.txt
