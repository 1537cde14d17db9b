#!/usr/bin/env python3
"""Inspect / write / check `.mdpt` model files (muggled_dpt_amd/export.py: the deployment artefact a torch-free host runs through the C ABI).

    python tools/mdpt_model_file.py info model.mdpt                     header, arithmetic mode, parameter inventory
    python tools/mdpt_model_file.py export checkpoint.pth model.mdpt [--dtype bf16|fp16|fp32] [--precision mixed ...]
                                                                        any checkpoint make_dpt_from_state_dict understands -> model file
    python tools/mdpt_model_file.py synth vitl model.mdpt [--dtype ...]  seeded synthetic checkpoint of a named configuration (no network here)
    python tools/mdpt_model_file.py roundtrip model.mdpt                reload it and compare every tensor with the file (CPU only)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muggled_dpt_amd import export as mexport  # noqa: E402
from muggled_dpt_amd import native  # noqa: E402

DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def info(path):
    rec = mexport.read_model_file(path)
    c = rec["config"]
    names = {v: k for k, v in native.PRECISIONS.items()}
    fam = {native.FAMILY_DAV2: "Depth-Anything V2", native.FAMILY_DAV1: "Depth-Anything V1", native.FAMILY_BEIT: "MiDaS v3.1 BEiT", native.FAMILY_SWINV2: "MiDaS v3.1 SwinV2"}
    nbytes = sum(t.numel() * t.element_size() for t in rec["tensors"].values())
    print(f"{path}: ABI {rec['abi_version']}, {fam[c.family]}, F={c.features_per_token} heads={c.num_heads} blocks={c.num_blocks} patch={c.patch_size_px} "
          f"fusion={c.fusion_channels}, precision {names[c.precision]}, class passes {rec['class_passes'] or 'mode default'}, "
          f"compensation {rec['wrc']}, latency mode {rec['latency_mode']}")
    print(f"  preprocessing: mean {rec['rgb_mean']}, std {rec['rgb_std']}, sides snap to {rec['tiling_size']} px, default side {rec['default_side']} px")
    print(f"  {len(rec['tensors'])} tensors, {nbytes / 1e6:.1f} MB, dtype {rec['meta'].get('param_dtype')}")
    for name, t in list(rec["tensors"].items())[:8]:
        print(f"    {name:64s} {tuple(t.shape)}")
    if len(rec["tensors"]) > 8:
        print(f"    ... {len(rec['tensors']) - 8} more")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("cmd", choices=["info", "export", "synth", "roundtrip"])
    ap.add_argument("src")
    ap.add_argument("dst", nargs="?")
    ap.add_argument("--dtype", choices=sorted(DTYPES), default=None)
    ap.add_argument("--precision", choices=sorted(native.PRECISIONS), default=None)
    a = ap.parse_args()
    if a.cmd == "info":
        return info(a.src)
    if a.cmd == "roundtrip":
        rec = mexport.read_model_file(a.src)
        _, model = mexport.load_exported(a.src)
        sd = {f"{comp}.{k}": v for comp in mexport.COMPONENTS for k, v in getattr(model, comp).state_dict().items()}
        assert sd.keys() == rec["tensors"].keys()
        assert all(torch.equal(sd[k].view(torch.int16) if sd[k].dtype == torch.bfloat16 else sd[k], v.view(torch.int16) if v.dtype == torch.bfloat16 else v)
                   for k, v in rec["tensors"].items())
        print(f"{a.src}: {len(sd)} tensors reload bit for bit")
        return
    if a.cmd == "export":
        from muggled_dpt_amd import make_dpt_from_state_dict
        _, model = make_dpt_from_state_dict(a.src)
    else:
        import bench
        model, _ = bench.make_model_and_weights(a.src)
    if a.dtype:
        model = model.to(DTYPES[a.dtype])
    if a.precision:
        model.set_precision(a.precision)
    print(model.export(a.dst))


if __name__ == "__main__":
    main()
