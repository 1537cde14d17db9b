#!/usr/bin/env python3
"""Whole ViT-L batch-32 forwards with the in-library HIP-event profile on (batch split off): forward time and the average duration of the
kernels whose name contains one of the given substrings.   python tools/probes/gpu_kernel_share.py layernorm attn"""
import ctypes, json, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
lib = native.load()
_, model = make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("vitl", 0))
model = model.to("cuda", torch.bfloat16)
x = torch.randn(32, 3, 504, 504, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
native.check(lib, lib.mdpt_set_batch_split(model._get_engine().handle, 0))
native.check(lib, lib.mdpt_debug_set_reassemble_overlap(model._get_engine().handle, 0))  # nothing on the side stream: every kernel alone
if os.environ.get("X3"):
    model = model.to(torch.float32); x = x.float()
    native.check(lib, lib.mdpt_set_batch_split(model._get_engine().handle, 0))
    native.check(lib, lib.mdpt_debug_set_reassemble_overlap(model._get_engine().handle, 0))  # nothing on the side stream: every kernel alone
if os.environ.get("TILE"):
    model.set_gemm_tile(int(os.environ["TILE"]))  # force one tile variant for every GEMM (MDPT_TILE_*)
with torch.inference_mode():
    for _ in range(2): model(x)
    torch.cuda.synchronize()
    lib.mdpt_profile_enable(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): model(x)
    e1.record(); torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.mdpt_profile_report(buf, len(buf)); lib.mdpt_profile_enable(0)
pr = json.loads(buf.value.decode())
out = [f"forward {e0.elapsed_time(e1)/5:7.3f} ms"]
if "ALL" in sys.argv[1:]:  # the whole table: per-forward milliseconds and fraction of the dense bf16 MFMA peak of every kernel
    print(out[0])
    for k in pr["kernels"]:
        print(f"  {k['name']:52s} {k['launches'] / 5:5.1f} launches/fwd  avg {k['avg_us']:8.1f} us  {k['total_ms'] / 5:7.3f} ms/fwd  "
              f"{(k['tflops'] / 2500 if k['gflop'] > 0 else float('nan')):6.3f} of peak")
    sys.exit(0)
for pat in sys.argv[1:]:
    for k in pr["kernels"]:
        if pat in k["name"]:
            out.append(f"{k['name']} {k['avg_us']:.1f} us x {k['launches']}")
print(" | ".join(out), flush=True)
