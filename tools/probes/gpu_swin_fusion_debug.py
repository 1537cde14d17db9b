#!/usr/bin/env python3
"""Fused Q/K epilogue vs swin_qk_prep: max abs difference of the depth maps per (dtype, tile) - run twice, with and without
MDPT_SWIN_NO_QK_FUSION=1 (library built with MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES), to separate the fusion from the tile variants."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests.test_gpu_swinv2 import _build
from tests.helpers import seeded_input

name = sys.argv[1] if len(sys.argv) > 1 else "swin2_base_384"
for dtype in (torch.bfloat16, torch.float32):
    model, cfg, w = _build(name, 3, dtype)
    x = seeded_input((2, 3, 384, 384), 9).to("cuda", dtype)
    model.set_gemm_tile(1)
    y1 = model(x).float()
    for tile in (5, 0, 2):
        model.set_gemm_tile(tile)
        y = model(x).float()
        print(f"{name} {dtype} tile {tile} vs tile 1: max |diff| {float((y - y1).abs().max()):.3e}  (max |y| {float(y1.abs().max()):.3f})", flush=True)
