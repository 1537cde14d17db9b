#!/usr/bin/env python3
"""Headless single-image run (SURVEY §8(b) "who calls it" / §8(d) config 1): load a checkpoint (or build a seeded synthetic one),
run DPTModel.inference on one BGR uint8 image and print load ms, inference ms and the output shape, like the reference's
run_image.py:204-208 does before it opens its window. No display, no OpenCV: images come from a .npy file (HxWx3 uint8, BGR) or
are synthesised; the 8-bit depth map can be saved as .npy (device-side post-processing, muggled_dpt_amd.postprocess).

  python tools/mdpt_run_image.py --synthetic vits --size 518 --fp32
  python tools/mdpt_run_image.py -m model_weights/depth_anything_v2_vitl.pth -i image.npy -o depth_u8.npy
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--model_path", default=None, help="checkpoint file (any of the four supported families)")
    ap.add_argument("--synthetic", default="vits", help="config name for seeded synthetic weights when no checkpoint is given")
    ap.add_argument("-i", "--image_path", default=None, help=".npy file holding an HxWx3 uint8 BGR image")
    ap.add_argument("-s", "--size", type=int, default=None, help="max side length (default: the model's base size)")
    ap.add_argument("-a", "--use_aspect_ratio", action="store_true", help="keep the image aspect ratio (default: square sizing)")
    ap.add_argument("--fp32", action="store_true", help="float32 model = split-bf16 fp32-class arithmetic (default: bfloat16)")
    ap.add_argument("-o", "--output", default=None, help="save the 8-bit depth map (.npy)")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("mdpt_run_image needs an MI355X: no GPU visible (there is no CPU fallback)")

    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, make_dpt_from_state_dict
    from muggled_dpt_amd.postprocess import convert_to_uint8
    t0 = time.perf_counter()
    if args.model_path:
        cfg, model = make_dpt_from_state_dict(args.model_path)
    else:
        from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
        cfg, model = make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict(args.synthetic, 0))
    model.to("cuda", torch.float32 if args.fp32 else torch.bfloat16)
    if args.image_path:
        img = np.load(args.image_path)
    else:
        img = np.random.default_rng(1).integers(0, 256, (518, 518, 3), dtype=np.uint8)
    model.inference(img, args.size, not args.use_aspect_ratio)  # first call builds the engine (weight repack)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"Loading model & first call: {round(1000 * (t1 - t0))} ms", flush=True)
    depth = model.inference(img, args.size, not args.use_aspect_ratio)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"Inference: {round(1000 * (t2 - t1), 2)} ms", f"Prediction shape: {tuple(depth.shape)}, dtype {depth.dtype}, device {depth.device}", sep="\n")
    if args.output:
        np.save(args.output, convert_to_uint8(depth).squeeze(0).cpu().numpy())
        print("saved", args.output)


if __name__ == "__main__":
    main()
