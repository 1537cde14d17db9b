"""Deployment artefact of a DPTModel: one `.mdpt` file that a host WITHOUT PyTorch can run through the C ABI (include/mdpt.h).

What it replaces: the reference hands a model to another runtime with `experiments/export_onnx.py:119-148` (torch.onnx.export of the PyTorch
graph with dynamic H / W axes). A model on this path is not a traceable graph - it is a configuration, a set of named parameters and a C handle -
so the native analogue of that export is the file a torch-free host feeds to mdpt_create / mdpt_bind_weight / mdpt_finalize: it keeps the
dynamic axes for free (one handle serves any legal (B, H, W)) and carries the arithmetic mode the model was validated in.
tests/c_host/host_main.cpp is such a host (plain C++ + HIP runtime); tools/mdpt_model_file.py inspects / converts / reloads files.

File layout (little endian; every block starts on a 16-byte boundary):

    char     magic[8]            "MDPTMDL1"
    int32    abi_version         MDPT_ABI_VERSION the file was written for (the layout of mdpt_config below)
    int32    config_bytes        sizeof(mdpt_config)
    byte     config[...]         the mdpt_config struct (family, precision and the reference's config keys; include/mdpt.h)
    int32    class_passes[16]    per MDPT_CLASS_*: 0 = the precision mode's default, else 1 / 2 / 3 (mdpt_set_class_passes); unused entries 0
    int32    wrc                 -1 = mode default, 0 / 1 / class mask = mdpt_set_weight_rounding_compensation
    int32    latency_mode        mdpt_set_latency_mode
    float    rgb_mean[3], rgb_std[3]     PatchEmbed.prepare_image normalisation (patch_embed.py:38-39)
    int32    tiling_size, default_side   its size rule (patch_embed.py:69,116-130): sides snap to multiples of tiling_size
    int32    json_bytes
    byte     json[...]           utf-8 JSON {"family", "config": the Python config dict, "precision"}: what the Python loader rebuilds the DPTModel from
    int32    n_tensors
    n x {  int32 name_bytes; char name[...] ("<component>.<reference new-format key>", mdpt_weight_name);
           int32 dtype (MDPT_DTYPE_*); int32 ndim; int64 shape[ndim]; int64 data_bytes;  (pad to 16)  byte data[...]  (pad to 16) }
"""
from __future__ import annotations

import json
import struct

import numpy as np
import torch

from . import native
from .state_dict_conversion import COMPONENTS

MAGIC = b"MDPTMDL1"
_NP_OF = {native.DTYPE_F32: np.float32, native.DTYPE_F16: np.float16, native.DTYPE_BF16: np.uint16}
_TORCH_OF = {native.DTYPE_F32: torch.float32, native.DTYPE_F16: torch.float16, native.DTYPE_BF16: torch.bfloat16}


def _pad16(f) -> None:
    f.write(b"\0" * ((-f.tell()) % 16))


def _jsonable(v):
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, torch.Tensor):
        return v.tolist()
    return v


def export_model(model, path: str, dtype: torch.dtype | None = None) -> dict:
    """Write `model` (a DPTModel on any device) to `path`. Parameters are stored in `dtype` (default: as the model holds them - a bfloat16 model
    exports 2-byte tensors, which is what its engine binds too, so a host that binds them the same way reproduces its bits). Returns a summary."""
    from .dpt_model import model_precision_code, native_config
    p0 = next(model.parameters())
    store = dtype or (p0.dtype if p0.dtype in _TORCH_OF.values() else torch.float32)
    # the arithmetic is that of the STORED dtype (what both readers of the file bind): an fp32 model exported with dtype=bfloat16 is a bfloat16 model
    c = native_config(model.config, model.family, model_precision_code(model, store))
    passes = [0] * 16
    for name, n in (model.__dict__.get("_class_passes") or {}).items():
        passes[native.OP_CLASSES.index(name)] = int(n)
    wrc = model.__dict__.get("_wrc")
    pe = model.patch_embed
    params = {f"{comp}.{k}": v for comp in COMPONENTS for k, v in getattr(model, comp).state_dict().items()}
    meta = {"family": model.family, "config": {k: _jsonable(v) for k, v in model.config.items()},
            "precision": model.__dict__.get("_precision"), "param_dtype": str(store).replace("torch.", "")}
    blob = json.dumps(meta).encode()
    total = 0
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<2i", native.ABI_VERSION, len(bytes(c))))
        f.write(bytes(c))
        f.write(struct.pack("<16i", *passes))
        f.write(struct.pack("<2i", -1 if wrc is None else int(wrc), int(bool(model.__dict__.get("_latency_mode", False)))))
        f.write(struct.pack("<6f", *pe.rgb_offset, *pe.rgb_stdev))
        f.write(struct.pack("<2i", int(pe._tiling_size), int(pe._default_size_px)))
        f.write(struct.pack("<i", len(blob)) + blob)
        f.write(struct.pack("<i", len(params)))
        for name, t in params.items():
            t = t.detach().to("cpu", store).contiguous()
            raw = t.view(torch.int16).numpy().tobytes() if store == torch.bfloat16 else t.numpy().tobytes()
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb)
            f.write(struct.pack("<2i", native.dtype_code(store), t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(struct.pack("<q", len(raw)))
            _pad16(f)
            f.write(raw)
            _pad16(f)
            total += len(raw)
    return {"path": path, "tensors": len(params), "parameter_bytes": total, "family": model.family, "precision": c.precision}


def read_model_file(path: str) -> dict:
    """Parse a `.mdpt` file: {"abi_version", "config" (MdptConfig), "class_passes", "wrc", "latency_mode", "rgb_mean", "rgb_std", "tiling_size",
    "default_side", "meta" (the JSON block), "tensors": {name: torch.Tensor (CPU, stored dtype)}}."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != MAGIC:
        raise ValueError(f"{path}: not an mdpt model file (bad magic)")
    off = 8
    abi, cfg_bytes = struct.unpack_from("<2i", data, off); off += 8
    if abi != native.ABI_VERSION or cfg_bytes != len(bytes(native.MdptConfig())):
        raise ValueError(f"{path}: written for ABI version {abi} (mdpt_config of {cfg_bytes} bytes), this library is version {native.ABI_VERSION}")
    cfg = native.MdptConfig.from_buffer_copy(data[off:off + cfg_bytes]); off += cfg_bytes
    passes = struct.unpack_from("<16i", data, off); off += 64
    wrc, latency = struct.unpack_from("<2i", data, off); off += 8
    norm = struct.unpack_from("<6f", data, off); off += 24
    tiling, default_side = struct.unpack_from("<2i", data, off); off += 8
    (jn,) = struct.unpack_from("<i", data, off); off += 4
    meta = json.loads(data[off:off + jn].decode()); off += jn
    (n,) = struct.unpack_from("<i", data, off); off += 4
    tensors = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<i", data, off); off += 4
        name = data[off:off + ln].decode(); off += ln
        dt, ndim = struct.unpack_from("<2i", data, off); off += 8
        shape = struct.unpack_from(f"<{ndim}q", data, off); off += 8 * ndim
        (nb,) = struct.unpack_from("<q", data, off); off += 8
        off += (-off) % 16
        arr = np.frombuffer(data, dtype=_NP_OF[dt], count=nb // np.dtype(_NP_OF[dt]).itemsize, offset=off).copy()
        t = torch.from_numpy(arr.view(np.int16)).view(torch.bfloat16) if dt == native.DTYPE_BF16 else torch.from_numpy(arr)
        tensors[name] = t.reshape(tuple(shape))
        off += nb
        off += (-off) % 16
    return {"abi_version": abi, "config": cfg, "class_passes": {native.OP_CLASSES[i]: v for i, v in enumerate(passes[:len(native.OP_CLASSES)]) if v},
            "wrc": None if wrc < 0 else (bool(wrc) if wrc <= 1 else tuple(n for i, n in enumerate(native.OP_CLASSES) if (wrc >> i) & 1)), "latency_mode": bool(latency), "rgb_mean": norm[:3], "rgb_std": norm[3:], "tiling_size": tiling,
            "default_side": default_side, "meta": meta, "tensors": tensors}


def load_exported(path: str):
    """(config dict, DPTModel on the CPU) rebuilt from a `.mdpt` file - the Python-side reader of the artefact (a torch-free host reads the
    same file: tests/c_host/host_main.cpp). The model keeps the stored parameter dtype and the arithmetic settings it was exported with."""
    from .dpt_model import DPTModel
    rec = read_model_file(path)
    meta = rec["meta"]
    cfg = dict(meta["config"])
    for k in ("base_patch_grid_hw", "window_size_hw"):
        if k in cfg and cfg[k] is not None:
            cfg[k] = tuple(cfg[k])
    model = DPTModel(cfg, meta["family"])
    dt = next(iter(rec["tensors"].values())).dtype
    model = model.to(dt)
    for comp in COMPONENTS:
        sd = {k[len(comp) + 1:]: v for k, v in rec["tensors"].items() if k.startswith(comp + ".")}
        getattr(model, comp).load_state_dict(sd, strict=True)
    if meta.get("precision"):
        model.set_precision(meta["precision"])
    if rec["class_passes"]:
        model.set_class_passes(rec["class_passes"])
    if rec["wrc"] is not None:
        model.set_weight_rounding_compensation(rec["wrc"])
    if rec["latency_mode"]:
        model.set_latency_mode(True)
    return cfg, model
