mkdir -p gpurun_out/r4q
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES python -c "from muggled_dpt_amd import native; native.build(force=True)" > gpurun_out/r4q/build.log 2>&1
for r in 1 2 3; do
  for v in 0 1 2 3; do
    echo "== MDPT_ATTN_PRIO=$v round $r"
    MDPT_ATTN_PRIO=$v python tools/probes/gpu_attn_kernel_ab.py 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r4q/attn_prio_ab.txt
