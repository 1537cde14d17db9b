#!/usr/bin/env python3
"""Attention kernel at the ViT-L encoder shape (32 x 16 heads x 1297 tokens x 64) as it runs inside the model: whole forwards with the
in-library HIP-event profile on, with and without the two-stream batch split; prints the forward time and the attention launches' average."""
import ctypes, json, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
lib = native.load()
_, model = make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("vitl", 0))
model = model.to("cuda", torch.bfloat16)
x = torch.randn(32, 3, 504, 504, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
for split in (0, 8):
    native.check(lib, lib.mdpt_set_batch_split(model._get_engine().handle, split))
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
    att = [k for k in pr["kernels"] if k["name"].startswith("attn")][0]
    print(f"split={split}: forward {e0.elapsed_time(e1)/5:7.3f} ms   attention avg {att['avg_us']:7.1f} us x {att['launches']}  ({att['tflops']:.0f} TF)", flush=True)
