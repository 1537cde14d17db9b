#!/usr/bin/env python3
"""Achieved bandwidth of the two full-size upsample launches of a ViT-L B=32 forward (HBM-bound helper)."""
import ctypes, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
native.load()
lib = ctypes.CDLL(native.LIB_PATH)
fn = getattr(lib, "_Z20mdpt_launch_upsamplePKfPDF16bS1_PfiiiiiiP12ihipStream_t")
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
fn.restype = ctypes.c_int
stream = torch.cuda.current_stream().cuda_stream
for (B, Hi, Wi, Ho, Wo, C, tag) in [(32, 144, 144, 288, 288, 256, "fusion x2"), (32, 288, 288, 504, 504, 128, "head x1.75"), (32, 144, 144, 252, 252, 64, "vits head"), (3, 37, 50, 74, 101, 256, "ragged x2")]:
    src = torch.randn(B, Hi, Wi, C, device="cuda")
    out = torch.empty(B, Ho, Wo, C, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        fn(src.data_ptr(), out.data_ptr(), None, None, B, Hi, Wi, Ho, Wo, C, stream)
    torch.cuda.synchronize()
    # the direct kernel (taken when an fp32 copy is requested) must give the same bits as the LDS-tiled one
    ref, f32 = torch.empty_like(out), torch.empty(B, Ho, Wo, C, device="cuda")
    fn(src.data_ptr(), ref.data_ptr(), None, f32.data_ptr(), B, Hi, Wi, Ho, Wo, C, stream)
    torch.cuda.synchronize()
    print(f"{tag:12s}: tiled == direct bitwise: {torch.equal(out.view(torch.int16), ref.view(torch.int16))}", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(src.data_ptr(), out.data_ptr(), None, None, B, Hi, Wi, Ho, Wo, C, stream)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = (src.numel() * 4 + out.numel() * 2) / 1e9
    print(f"{tag:12s}: {us:8.1f} us  {gb:5.2f} GB algorithmic -> {gb / us * 1e6 / 1e3:5.2f} TB/s", flush=True)
