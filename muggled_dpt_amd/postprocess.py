"""Depth post-processing on the GPU: drop-in for the tensor helpers of the reference's muggled_dpt/demo_helpers/postprocess.py
(scale_prediction :22-29, normalize_01 :63-74, convert_to_uint8 :79-91) plus the 24-bit packing step of run_3dviewer.py:576-590.

Every function takes the CUDA tensor the model returned and launches HIP kernels (libmdpt: mdpt_post_*) on the current torch
stream; results stay on the device (the reference's convert_to_uint8 does the same, postprocess.py:85-87). min / max never visit
the host. There is no CPU implementation here: host arrays raise (numpy callers should keep using numpy).
"""

from __future__ import annotations

import torch
from torch import Tensor

from . import native


def _dev_f32(t, what: str) -> Tensor:
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise RuntimeError(f"{what}: expected a CUDA tensor (muggled_dpt_amd post-processing runs on the MI355X only, no CPU fallback)")
    return t.detach().to(torch.float32).contiguous()


def _launch(dev, fn, *args):
    lib = native.load()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        native.check(lib, getattr(lib, fn)(*args, stream))


def scale_prediction(prediction_tensor: Tensor, target_wh: tuple[int, int], interpolation: str = "bilinear") -> Tensor:
    """BxHxW -> Bx(target_h)x(target_w), F.interpolate(mode="bilinear", align_corners=False) (postprocess.py:22-29)."""
    if interpolation != "bilinear":
        raise NotImplementedError(f"interpolation '{interpolation}' is not built on the MI355X path (bilinear only)")
    x = _dev_f32(prediction_tensor, "scale_prediction")
    if x.dim() != 3:
        raise RuntimeError(f"scale_prediction expects BxHxW, got {tuple(x.shape)}")
    b, h, w = x.shape
    oh, ow = int(target_wh[1]), int(target_wh[0])
    out = torch.empty((b, oh, ow), device=x.device, dtype=torch.float32)
    _launch(x.device, "mdpt_post_scale_prediction", x.data_ptr(), b, h, w, out.data_ptr(), oh, ow, None, None)
    return out.to(prediction_tensor.dtype)


def _minmax(x: Tensor) -> Tensor:
    mm = torch.empty(2, device=x.device, dtype=torch.float32)
    scratch = torch.empty(2, device=x.device, dtype=torch.int32)
    _launch(x.device, "mdpt_post_minmax", x.data_ptr(), x.numel(), mm.data_ptr(), scratch.data_ptr())
    return mm


def normalize_01(data: Tensor) -> Tensor:
    """(data - min) / (max - min) (postprocess.py:63-74); result in the input dtype."""
    x = _dev_f32(data, "normalize_01")
    out = torch.empty_like(x)
    mm = _minmax(x)
    _launch(x.device, "mdpt_post_normalize", x.data_ptr(), x.numel(), mm.data_ptr(), out.data_ptr(), native.POST_F32, 0)
    return out.to(data.dtype)


def convert_to_uint8(depth_prediction_tensor: Tensor) -> Tensor:
    """(255 * normalize_01(x)).byte(), still on the device (postprocess.py:79-91)."""
    x = _dev_f32(depth_prediction_tensor, "convert_to_uint8")
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    mm = _minmax(x)
    _launch(x.device, "mdpt_post_normalize", x.data_ptr(), x.numel(), mm.data_ptr(), out.data_ptr(), native.POST_U8, 0)
    return out


def scale_and_convert_to_uint8(prediction_tensor: Tensor, target_wh: tuple[int, int]) -> Tensor:
    """convert_to_uint8(scale_prediction(x, target_wh)) as the video loop does (run_video.py:348-349): the resize pass also
    reduces min / max, so the display-size fp32 map is written once and read once."""
    x = _dev_f32(prediction_tensor, "scale_and_convert_to_uint8")
    b, h, w = x.shape
    oh, ow = int(target_wh[1]), int(target_wh[0])
    scaled = torch.empty((b, oh, ow), device=x.device, dtype=torch.float32)
    mm = torch.empty(2, device=x.device, dtype=torch.float32)
    scratch = torch.empty(2, device=x.device, dtype=torch.int32)
    _launch(x.device, "mdpt_post_scale_prediction", x.data_ptr(), b, h, w, scaled.data_ptr(), oh, ow, mm.data_ptr(), scratch.data_ptr())
    out = torch.empty((b, oh, ow), device=x.device, dtype=torch.uint8)
    _launch(x.device, "mdpt_post_normalize", scaled.data_ptr(), scaled.numel(), mm.data_ptr(), out.data_ptr(), native.POST_U8, 0)
    return out


def pack_depth_u24(depth_prediction: Tensor, is_metric: bool = False, lossy: bool = False) -> Tensor:
    """[1,H,W] (or [H,W]) depth -> uint8 [H,W,4] BGRA carrying round(16777215 * normalize_01(depth)) in B (low), G, R (high);
    alpha is zero for the caller's mask (run_3dviewer.py:576-593). is_metric skips the normalisation, lossy keeps the top byte."""
    x = _dev_f32(depth_prediction, "pack_depth_u24").squeeze()
    if x.dim() != 2:
        raise RuntimeError(f"pack_depth_u24 expects one depth map, got {tuple(depth_prediction.shape)}")
    out = torch.empty((x.shape[0], x.shape[1], 4), device=x.device, dtype=torch.uint8)
    mm = None if is_metric else _minmax(x)
    _launch(x.device, "mdpt_post_normalize", x.data_ptr(), x.numel(), None if mm is None else mm.data_ptr(), out.data_ptr(),
            native.POST_U24, int(bool(lossy)))
    return out


def remove_inf_tensor(data: Tensor, inf_replacement_value: float = 0.0, in_place: bool = True) -> Tensor:
    """postprocess.py:34-40 (plain torch indexing on whatever device the tensor lives on; not a kernel of ours)."""
    data = data if in_place else data.clone()
    data[data.isinf()] = inf_replacement_value
    return data
