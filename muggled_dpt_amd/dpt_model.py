"""DPTModel facade over libmdpt (MI355X-native). Mirrors the reference's public surface:

    DPTModel.forward / inference / prepare_image_bgr / verify_input      reference muggled_dpt/dpt_model.py:61,87,113,133
    model.patch_embed / imgencoder / reassemble / fusion / head           dpt_model.py:50-54 (callable one by one,
                                                                           simple_examples/internal_features.py:39-45)
    nn.Module plumbing: .to(device, dtype), .parameters(), .state_dict(), sub-module load_state_dict with the
    reference's converted key names (make_depthanythingv2_dpt.py:55-59)

The five sub-modules hold the parameters (fp32/bf16 nn.Parameters named exactly like the reference's) and forward to
the stage entry points of the C ABI. All compute happens in hand-written HIP kernels; there is NO CPU or PyTorch
fallback: calling the model on a CPU tensor raises.

dtype semantics (the C ABI's `precision`):
    model dtype float32   -> MDPT_PREC_BF16X3 (split-bf16 MFMA, fp32-class accuracy; the parity configuration)
    model dtype bfloat16  -> MDPT_PREC_BF16   (single-pass bf16 MFMA; the reference's GPU default, misc.py:73-77)
    model dtype float16   -> MDPT_PREC_BF16 arithmetic with fp16 tensors at the boundary
Outputs have the model dtype and live on the model device; nothing synchronises (callers do `.cpu()` themselves).
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import native
from .state_dict_conversion import COMPONENTS, expected_new_keys

RGB_MEAN = (0.485, 0.456, 0.406)  # reference v2_depthanything/patch_embed.py:38
RGB_STD = (0.229, 0.224, 0.225)   # :39
BEIT_RGB_MEAN = (0.5, 0.5, 0.5)   # reference v31_beit/patch_embed.py:39-40
BEIT_RGB_STD = (0.5, 0.5, 0.5)


# Bumped whenever a Parameter object is (re)assigned anywhere in a model tree of this module: DPTModel caches its parameter list and
# re-collects it only when this moved (a module-tree walk per forward cost several hundred microseconds at batch 1). The counter is process-wide
# on purpose (a node does not know its DPTModel), so it only ever says "walk your tree again": what decides whether an ENGINE is stale is the
# identity of the Parameter objects found and their version counters - building a second model never re-packs the first one's weights.
_TREE_GENERATION = [0]
_RECHECK_EVERY = 256  # forwards between two unconditional re-walks of the tree (writes to module._parameters[...] that bypass the hooks below)


class _ParamWatch:
    """Mixin: attribute assignment / registration of a Parameter invalidates the cached parameter lists."""

    def __setattr__(self, name, value):
        if isinstance(value, nn.Parameter) or name in self.__dict__.get("_parameters", ()):
            _TREE_GENERATION[0] += 1
        super().__setattr__(name, value)

    def register_parameter(self, name, param):
        _TREE_GENERATION[0] += 1
        super().register_parameter(name, param)

    def __delattr__(self, name):
        if name in self.__dict__.get("_parameters", ()):
            _TREE_GENERATION[0] += 1
        super().__delattr__(name)

    def _apply(self, fn, *args, **kwargs):
        # sub_module.to(...) / .half() / .to_empty() may replace Parameter objects by writing module._parameters[...] directly
        # (torch.__future__.set_overwrite_module_params_on_conversion, swap_tensors): no __setattr__ is seen
        out = super()._apply(fn, *args, **kwargs)
        _TREE_GENERATION[0] += 1
        return out


class _Node(_ParamWatch, nn.Module):
    """Anonymous container so dotted reference keys ("stages.0.blocks.1.attn.qkv.weight") become real module paths."""

    def forward(self, *args, **kwargs):  # pragma: no cover - containers are not callable
        raise RuntimeError("parameter container, not a callable layer")

    # numeric children behave like the reference's nn.ModuleList / nn.Sequential (model.fusion.blocks[2], len(...), iteration)
    def __getitem__(self, idx):
        if isinstance(idx, slice):  # blocks[:-4] etc. -> a list of the selected children, like slicing a ModuleList and iterating it
            return [self[i] for i in range(*idx.indices(len(self)))]
        return getattr(self, str(int(idx) if int(idx) >= 0 else len(self) + int(idx)))

    def __len__(self):
        return sum(1 for name in self._modules if name.isdigit())

    def __bool__(self):  # a container without numeric children (len 0) is still a module, not "empty"
        return True

    def __iter__(self):
        return iter(getattr(self, str(i)) for i in range(len(self)))


class _SoftmaxProbe(nn.Softmax):
    """Stand-in for the nn.Softmax module of the reference's non-optimised attention (transformer_block.py:101): the fused HIP
    kernel never materialises the softmax matrix, but when a forward hook is registered here (the way
    demo_helpers/model_capture.py:54-59 captures attention maps) the encoder dumps this block's [B, heads, N, N] weights with a
    dedicated kernel (mdpt_encoder_probe) and passes them through this module so the hook fires with them as output."""

    def forward(self, attention_weights: Tensor) -> Tensor:
        return attention_weights


class _BlockProbe(_Node):
    """A transformer block of the encoder. The whole encoder is one C call, so the block is never *called* on the hot path; when a
    forward hook is registered on it (demo_helpers/model_capture.py:54-59 does that for experiments/block_norm_visualization.py:282)
    the encoder also writes this block's output tokens (mdpt_encoder_probe_blocks) and passes them through this module so the hook
    fires with them as the output, like a hook on the reference's TransformerBlock (transformer_block.py:41-62)."""

    def forward(self, block_output_tokens: Tensor) -> Tensor:
        return block_output_tokens


class TransformerBlock(_BlockProbe):
    """Public name of the encoder blocks of the ViT families (Depth-Anything V1 / V2, BEiT): what the reference's tooling selects with
    isinstance(module, TransformerBlock) (demo_helpers/model_capture.py:54-59, experiments/block_norm_visualization.py:263-282)."""


class SwinTransformerBlock(_BlockProbe):
    """Public name of the SwinV2 encoder blocks (reference v31_swinv2/image_encoder_model.py:164-225)."""


def _register_tree(root: nn.Module, shapes: dict[str, tuple]) -> None:
    for key, shape in shapes.items():
        parts = key.split(".")
        mod = root
        for name in parts[:-1]:
            if not hasattr(mod, name):
                mod.add_module(name, _Node())
            mod = getattr(mod, name)
        mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(tuple(shape)), requires_grad=False))


class _Stage(_ParamWatch, nn.Module):
    """Base of the five sub-modules: owns its parameters, forwards to the engine of the parent DPTModel."""

    def __init__(self, component: str, shapes: dict[str, tuple]):
        super().__init__()
        self._component = component
        _register_tree(self, shapes)
        self.__dict__["_owner"] = None  # set by DPTModel (kept out of the module tree to avoid cycles)

    def _engine(self):
        owner = self.__dict__["_owner"]
        if owner is None:
            raise RuntimeError("sub-module is not attached to a DPTModel")
        return owner._get_engine()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        owner = self.__dict__["_owner"]
        if owner is not None:
            owner._invalidate()


class PatchEmbed(_Stage):
    """Image -> patch tokens. reference v2_depthanything/patch_embed.py:23-165."""

    def __init__(self, shapes, patch_size_px: int, default_image_size: int, rgb_mean=RGB_MEAN, rgb_std=RGB_STD,
                 tiling_patches: int = 2):
        super().__init__("patch_embed", shapes)
        self.patch_size_px = patch_size_px
        self._default_size_px = round(default_image_size)
        self._tiling_size = round(tiling_patches * patch_size_px)  # patch_embed.py:69 (SwinV2: 8 patches, v31_swinv2/patch_embed.py:68)
        self.rgb_offset, self.rgb_stdev = tuple(rgb_mean), tuple(rgb_std)

    def forward(self, image_tensor_bchw: Tensor) -> tuple[Tensor, tuple[int, int]]:
        eng = self._engine()
        x = eng.as_input(image_tensor_bchw, 4)
        b, _, h, w = x.shape
        gh, gw = h // self.patch_size_px, w // self.patch_size_px
        out = torch.empty((b, gh * gw, eng.F), device=x.device, dtype=torch.float32)
        # PatchEmbed alone accepts odd patch grids (the reference only fails later, in fusion): the workspace is planned on the even-rounded size
        tile = self._tiling_size if eng.swin else 2 * self.patch_size_px
        eng.call("mdpt_patch_embed", x, b, h, w, out, size_hw=(-(-h // tile) * tile, -(-w // tile) * tile), batch=b)
        return eng.as_output(out), (gh, gw)

    def _prepare_plan(self, image_bgr: np.ndarray, max_side_length: int | None, use_square_sizing: bool, interpolation_mode: str):
        """Size rule and argument checks of prepare_image (patch_embed.py:103-130): -> (model tensor (H, W), MDPT_INTERP_*, parameter, image dtype)."""
        if max_side_length is None:
            max_side_length = self._default_size_px
        img_h, img_w = image_bgr.shape[0:2]
        largest_side = max(img_h, img_w)
        scale = max_side_length / largest_side
        targ_hw = (largest_side, largest_side) if use_square_sizing else (img_h, img_w)
        scaled_hw = [max(1, round(side * scale / self._tiling_size)) * self._tiling_size for side in targ_hw]
        p = next(self.parameters())
        # one HIP kernel (antialiased resize + BGR->RGB + normalisation), mdpt_prepare_image. No torch fallback: what the kernel does not
        # cover raises, exactly where torch's own F.interpolate(antialias=True) would (it supports bilinear and bicubic only).
        interp = {"bilinear": native.INTERP_BILINEAR, "bicubic": native.INTERP_BICUBIC}.get(interpolation_mode)
        if interp is None:
            raise ValueError(f"Anti-alias option is restricted to bilinear and bicubic modes (got interpolation_mode={interpolation_mode!r})")
        if p.device.type != "cuda":
            raise RuntimeError("prepare_image runs on the GPU only (no CPU fallback): move the model to a cuda device first")
        if not (isinstance(image_bgr, np.ndarray) and image_bgr.dtype == np.uint8 and image_bgr.ndim == 3 and image_bgr.shape[2] == 3):
            raise TypeError("prepare_image expects an OpenCV-style uint8 HxWx3 BGR image (cv2.imread output)")
        out_dtype = p.dtype if p.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32
        return scaled_hw, interp, p, out_dtype

    def prepare_image(self, image_bgr: np.ndarray, max_side_length: int | None = None, use_square_sizing: bool = True,
                      interpolation_mode: str = "bilinear") -> Tensor:
        """uint8 HxWx3 BGR -> normalised [1,3,H',W'] on the model device/dtype (patch_embed.py:103-145).
        Sides snap to multiples of 2*patch (so a 518x518 image is processed at 504x504)."""
        scaled_hw, interp, p, out_dtype = self._prepare_plan(image_bgr, max_side_length, use_square_sizing, interpolation_mode)
        img_h, img_w = image_bgr.shape[0:2]
        lib = native.load()
        with torch.cuda.device(p.device):
            src = self._stage_host_image(image_bgr, p.device)
            out = torch.empty((1, 3, scaled_hw[0], scaled_hw[1]), device=p.device, dtype=out_dtype)
            stream = torch.cuda.current_stream(p.device).cuda_stream
            mean3, std3 = self._norm_constants()
            # the kernel writes the model's dtype (dtype-tagged like mdpt_forward's tensors): nothing but this launch between the copy and the forward
            native.check(lib, lib.mdpt_prepare_image(src.data_ptr(), img_h, img_w, out.data_ptr(), native.dtype_code(out_dtype), scaled_hw[0], scaled_hw[1],
                                                     mean3, std3, interp, stream))
        return out if out.dtype == p.dtype else out.to(p.dtype)

    def _norm_constants(self):
        c = self.__dict__.get("_norm_c")
        if c is None:
            c = self.__dict__["_norm_c"] = ((ctypes.c_float * 3)(*self.rgb_offset), (ctypes.c_float * 3)(*self.rgb_stdev))
        return c

    def _stage_host_image(self, image_bgr: np.ndarray, device: torch.device) -> Tensor:
        """uint8 host image -> device through a reusable PINNED staging buffer (an asynchronous copy from pageable memory is a synchronous
        one in disguise): numpy -> pinned (host memcpy) -> device (non-blocking on the current stream). One (pinned, device) pair per
        PatchEmbed, grown on demand; the pinned buffer is rewritten only after the previous call's copy has drained (event)."""
        n = int(image_bgr.size)
        stages = self.__dict__.setdefault("_host_stage", {})
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)  # per stream: the device buffer is reused in stream order
        st = stages.get(key)
        if st is None or st["pinned"].numel() < n:
            if len(stages) >= 4:
                torch.cuda.synchronize(device)
                stages.clear()
            st = {"pinned": torch.empty(max(n, 1 << 20), dtype=torch.uint8).pin_memory(), "dev": torch.empty(max(n, 1 << 20), dtype=torch.uint8, device=device),
                  "event": None}
            stages[key] = st
        if st["event"] is not None:
            st["event"].synchronize()
        pinned = st["pinned"][:n]
        pinned.view(image_bgr.shape).numpy()[...] = image_bgr  # (handles non-contiguous views: numpy does the strided copy)
        dev = st["dev"][:n]
        dev.copy_(pinned, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        st["event"] = ev
        return dev

    def verify_input(self, image_tensor_bchw: Tensor) -> bool:
        _, c, h, w = image_tensor_bchw.shape
        assert c == 3, f"Bad channel count! Expected 3 got {c}"
        s = self.patch_size_px
        assert h % s == 0, f"Bad height! Image must have height ({h}) divisble by {s}"
        assert w % s == 0, f"Bad width! Image must have width ({w}) divisble by {s}"
        return True


class ImageEncoder(_Stage):
    """DINOv2 encoder with 4 taps. reference v2_depthanything/image_encoder_model.py:23-147."""

    def forward(self, patch_tokens: Tensor, patch_grid_hw: tuple[int, int]):
        eng = self._engine()
        x = eng.as_input(patch_tokens, 3)
        gh, gw = int(patch_grid_hw[0]), int(patch_grid_hw[1])
        b = x.shape[0]
        assert x.shape[1] == gh * gw and x.shape[2] == eng.F, f"tokens {tuple(x.shape)} do not match grid {gh}x{gw}, F={eng.F}"
        probes = self.__dict__.get("_softmax_probes") or []
        blocks = self.__dict__.get("_block_probes") or []
        hooked = [i for i, pr in enumerate(probes) if len(pr._forward_hooks) > 0]
        hooked_blocks = [i for i, (node, _) in enumerate(blocks) if len(node._forward_hooks) > 0]
        if hooked or hooked_blocks:  # hooks on softmax modules (enable_optimizations=False; SwinV2: always there) / on blocks: same pass + dumps
            if eng.swin:
                outs = [torch.empty((b, (gh >> s) * (gw >> s), eng.stage_features[s]), device=x.device, dtype=torch.float32) for s in range(4)]
                size_hw = (gh * eng.P, gw * eng.P)
            else:
                outs = [torch.empty((b, gh * gw + 1, eng.F), device=x.device, dtype=torch.float32) for _ in range(4)]
                size_hw = ((gh + gh % 2) * eng.P, (gw + gw % 2) * eng.P)
            dumps = {}
            for i in hooked:  # [B, heads, N, N]; SwinV2: [B * windows, heads of the stage, Nw, Nw] (mdpt_attn_probe_shape)
                shp = (ctypes.c_int64 * 4)()
                native.check(eng.lib, eng.lib.mdpt_attn_probe_shape(eng.handle, b, gh, gw, i, shp))
                dumps[i] = torch.empty(tuple(shp), device=x.device, dtype=torch.float32)
            arr = (ctypes.c_void_p * len(probes))(*[dumps[i].data_ptr() if i in dumps else None for i in range(len(probes))]) if probes else None
            bdumps = {}
            for i in hooked_blocks:  # block output tokens: [B, 1 + gh*gw, F]; SwinV2: [B, tokens of the block's stage, its features]
                st = blocks[i][1]
                shape = (b, (gh >> st) * (gw >> st), eng.stage_features[st]) if eng.swin else (b, gh * gw + 1, eng.F)
                bdumps[i] = torch.empty(shape, device=x.device, dtype=torch.float32)
            barr = (ctypes.c_void_p * len(blocks))(*[bdumps[i].data_ptr() if i in bdumps else None for i in range(len(blocks))]) if blocks else None
            eng.call_checked("mdpt_encoder_probe_blocks", x, b, gh, gw, eng.ptr_array(outs), arr, barr, size_hw=size_hw, batch=b)
            for i in hooked:
                probes[i](eng.as_output(dumps[i]))  # fires the registered forward hooks with the weights as module output
            for i in hooked_blocks:
                blocks[i][0](eng.as_output(bdumps[i]))  # ... with the block's output tokens
            return tuple(eng.as_output(o) for o in outs)
        if eng.swin:  # stage s: [B, (gh >> s) * (gw >> s), F_s] (v31_swinv2/image_encoder_model.py:77-98)
            outs = [torch.empty((b, (gh >> s) * (gw >> s), eng.stage_features[s]), device=x.device, dtype=torch.float32) for s in range(4)]
            eng.call_checked("mdpt_encoder", x, b, gh, gw, eng.ptr_array(outs), size_hw=(gh * eng.P, gw * eng.P), batch=b)
            return tuple(eng.as_output(o) for o in outs)
        outs = [torch.empty((b, gh * gw + 1, eng.F), device=x.device, dtype=torch.float32) for _ in range(4)]
        eng.call("mdpt_encoder", x, b, gh, gw, eng.ptr_array(outs), size_hw=((gh + gh % 2) * eng.P, (gw + gw % 2) * eng.P), batch=b)
        return tuple(eng.as_output(o) for o in outs)


class ReassembleModel(_Stage):
    """4 token stages -> 4 image-like maps. reference v2_depthanything/reassembly_model.py:21-310."""

    def forward(self, stage_1_tokens, stage_2_tokens, stage_3_tokens, stage_4_tokens, patch_grid_hw):
        eng = self._engine()
        gh, gw = int(patch_grid_hw[0]), int(patch_grid_hw[1])
        xs = [eng.as_input(t, 3) for t in (stage_1_tokens, stage_2_tokens, stage_3_tokens, stage_4_tokens)]
        b = xs[0].shape[0]
        c = eng.C
        sizes = [(4 * gh, 4 * gw), (2 * gh, 2 * gw), (gh, gw), (gh // 2, gw // 2)]
        if eng.swin:  # patch_grid_hw is the stage-0 grid; maps at 1, 1/2, 1/4, 1/8 of it (v31_swinv2/reassembly_model.py:113-122)
            sizes = [(gh >> s, gw >> s) for s in range(4)]
        outs = [torch.empty((b, c, sh, sw), device=xs[0].device, dtype=torch.float32) for sh, sw in sizes]
        eng.call_checked("mdpt_reassemble", eng.ptr_array(xs), b, gh, gw, eng.ptr_array(outs), size_hw=(gh * eng.P, gw * eng.P), batch=b)
        return tuple(eng.as_output(o) for o in outs)


class _FusionBlock(_Node):
    """One fusion block, callable on its own like the reference's FusionBlock / TopMostFusionBlock (fusion_model.py:89-154):
    blocks[3](downx2_map) ; blocks[i](reassembly_map, previous_fusion_map) -> map at twice the resolution."""

    def forward(self, reassembly_feature_map: Tensor, previous_fusion_feature_map: Tensor | None = None) -> Tensor:
        fusion = self.__dict__["_fusion"]
        idx = self.__dict__["_idx"]
        eng = fusion._engine()
        x = eng.as_input(reassembly_feature_map, 4)
        b, _, sh, sw = x.shape
        if (previous_fusion_feature_map is None) != (idx == 3):
            raise TypeError(f"fusion block {idx} takes {'one input' if idx == 3 else 'two inputs'}")
        prior = None if previous_fusion_feature_map is None else eng.as_input(previous_fusion_feature_map, 4)
        if prior is not None and prior.shape != x.shape:
            raise RuntimeError(f"The size of tensor a {tuple(x.shape)} must match the size of tensor b {tuple(prior.shape)}")
        out = torch.empty((b, eng.C, 2 * sh, 2 * sw), device=x.device, dtype=torch.float32)
        gh, gw = (2 * sh, 2 * sw) if idx == 3 else (sh // (4 >> idx), sw // (4 >> idx))
        eng.call_checked("mdpt_fusion_block", idx, x, prior, b, sh, sw, out,
                         size_hw=(gh * eng.Pdec, gw * eng.Pdec), batch=b)
        return eng.as_output(out)


class FusionModel(_Stage):
    """RefineNet-style coarse-to-fine fusion. reference v2_depthanything/fusion_model.py:20-220."""

    def __init__(self, component, shapes):
        super().__init__(component, shapes)
        for i in range(4):  # make blocks[i] callable: same parameters, same names, class swapped on the container node
            blk = getattr(self.blocks, str(i))
            blk.__class__ = _FusionBlock
            blk.__dict__["_idx"] = i
            blk.__dict__["_fusion"] = self

    def forward(self, upx4_featuremap, upx2_featuremap, noscale_featuremap, downx2_featuremap):
        eng = self._engine()
        xs = [eng.as_input(t, 4) for t in (upx4_featuremap, upx2_featuremap, noscale_featuremap, downx2_featuremap)]
        b, _, gh, gw = xs[2].shape
        if xs[3].shape[2] * 2 != gh or xs[3].shape[3] * 2 != gw or xs[0].shape[2] != 4 * gh or xs[1].shape[2] != 2 * gh:
            # same failure class as the reference (size mismatch at fusion_model.py:151)
            raise RuntimeError(f"fusion inputs are not scaled x2 relative to each other: {[tuple(x.shape) for x in xs]}")
        out = torch.empty((b, eng.C, 8 * gh, 8 * gw), device=xs[0].device, dtype=torch.float32)
        eng.call_checked("mdpt_fusion", eng.ptr_array(xs), b, gh, gw, out, size_hw=(gh * eng.Pdec, gw * eng.Pdec), batch=b)
        return eng.as_output(out)


class MonocularDepthHead(_Stage):
    """Feature map -> inverse depth. reference v2_depthanything/head_model.py:20-106."""

    def forward(self, imagelike_bchw: Tensor) -> Tensor:
        eng = self._engine()
        x = eng.as_input(imagelike_bchw, 4)
        b, _, fh, fw = x.shape
        if fh % 8 or fw % 8:
            # the fused map of a DPT forward is always 8x the patch grid; the kernels behind mdpt_head are planned on that grid (the
            # reference's standalone head would take any size: say so instead of silently flooring)
            raise RuntimeError(f"mdpt_head expects a fused feature map of 8x the patch grid, got {fh}x{fw} (not divisible by 8)")
        gh, gw = fh // 8, fw // 8
        out = torch.empty((b, gh * eng.Pdec, gw * eng.Pdec), device=x.device, dtype=torch.float32)
        eng.call_checked("mdpt_head", x, b, gh, gw, out, size_hw=(gh * eng.Pdec, gw * eng.Pdec), batch=b)
        return eng.as_output(out)


def model_precision_code(model: "DPTModel", dtype: torch.dtype) -> int:
    """MDPT_PREC_* of a model: DPTModel.set_precision's override, else the default arithmetic of the model dtype: fp32 -> split-bf16 x3
    (fp32 class), bf16 -> bf16 operands, fp16 -> fp16 operands (what each dtype means in the reference: demo_helpers/misc.py:61-77)."""
    default_prec = {torch.float32: native.PREC_BF16X3, torch.bfloat16: native.PREC_BF16, torch.float16: native.PREC_FP16}
    override = model.__dict__.get("_precision")
    return native.PRECISIONS[override] if override else default_prec.get(dtype, native.PREC_BF16)


def native_config(cfg: dict, family: str, precision: int) -> "native.MdptConfig":
    """The reference's config dict (make_*_dpt keyword arguments) as the C ABI's mdpt_config (include/mdpt.h)."""
    c = native.MdptConfig()
    if family == "swinv2":
        feats = [int(v) for v in cfg["features_per_stage"]]
        c.features_per_token, c.num_heads, c.num_blocks = feats[0], int(cfg["heads_per_stage"][0]), int(sum(cfg["layers_per_stage"]))
        for i in range(4):
            c.reassembly_features[i] = feats[i]
            c.swin_heads[i], c.swin_layers[i] = int(cfg["heads_per_stage"][i]), int(cfg["layers_per_stage"][i])
            pre = cfg["pretrained_window_sizes_per_stage"][i]
            c.swin_pretrained_window[i] = 0 if pre is None else int(pre)
        c.swin_window_h, c.swin_window_w = (int(v) for v in cfg["window_size_hw"])
    else:
        c.features_per_token, c.num_heads, c.num_blocks = cfg["features_per_token"], cfg["num_heads"], cfg["num_blocks"]
        for i in range(4):
            c.reassembly_features[i] = int(cfg["reassembly_features_list"][i])
    c.base_patch_grid_h, c.base_patch_grid_w = (int(v) for v in cfg["base_patch_grid_hw"])
    c.fusion_channels, c.patch_size_px = cfg["fusion_channels"], cfg["patch_size_px"]
    c.is_giant, c.is_metric = int(bool(cfg.get("is_giant", False))), int(bool(cfg.get("is_metric", False)))
    c.precision = precision
    c.family = {"v2": native.FAMILY_DAV2, "v1": native.FAMILY_DAV1, "beit": native.FAMILY_BEIT, "swinv2": native.FAMILY_SWINV2}[family]
    return c


class _Engine:
    """Owns the C handle, the packed-weight buffer and cached workspaces for one (device, dtype) of a DPTModel."""

    def __init__(self, model: "DPTModel", device: torch.device, dtype: torch.dtype):
        if device.type != "cuda":
            raise RuntimeError(
                f"muggled_dpt_amd runs on MI355X GPUs only (model is on '{device}'); there is no CPU fallback. "
                "Move the model with .to('cuda').")
        self.lib = native.load()
        self.device, self.dtype = device, dtype
        cfg = model.config
        self.swin = model.family == "swinv2"
        if self.swin:
            self.stage_features = [int(v) for v in cfg["features_per_stage"]]
            self.F, self.C, self.P, self.Pdec = self.stage_features[0], cfg["fusion_channels"], cfg["patch_size_px"], 4 * cfg["patch_size_px"]
        else:
            self.F, self.C, self.P = cfg["features_per_token"], cfg["fusion_channels"], cfg["patch_size_px"]
            self.Pdec = self.P
        c = native_config(cfg, model.family, model_precision_code(model, dtype))
        self.precision = c.precision
        handle = ctypes.c_void_p()
        native.check(self.lib, self.lib.mdpt_create(ctypes.byref(c), ctypes.byref(handle)))
        self.handle = handle
        for cls_name, passes in (model.__dict__.get("_class_passes") or {}).items():
            native.check(self.lib, self.lib.mdpt_set_class_passes(self.handle, native.OP_CLASSES.index(cls_name), int(passes)))
        if model.__dict__.get("_dbg_wscale_all", False):  # test hook: the other valid rounding of the fp16 build's weight scale (mdpt_debug_set_wscale_policy)
            native.check(self.lib, self.lib.mdpt_debug_set_wscale_policy(self.handle, 1))
        wrc = model.__dict__.get("_wrc")
        if wrc is not None:
            native.check(self.lib, self.lib.mdpt_set_weight_rounding_compensation(self.handle, int(wrc)))
        self._workspaces: dict[tuple, torch.Tensor] = {}
        tile = model.__dict__.get("_gemm_tile", 0)
        if tile:
            native.check(self.lib, self.lib.mdpt_set_gemm_tile(self.handle, tile))
        if model.__dict__.get("_latency_mode", False):
            native.check(self.lib, self.lib.mdpt_set_latency_mode(self.handle, 1))
        if not model.__dict__.get("_nonfinite_propagation", True):
            native.check(self.lib, self.lib.mdpt_set_nonfinite_propagation(self.handle, 0))
        self._grid_cache_on = bool(model.config.get("enable_cache", False))
        if self._grid_cache_on:  # the reference's make_*_dpt(..., enable_cache=True): per-grid constants computed once per (workspace, shape)
            native.check(self.lib, self.lib.mdpt_set_grid_cache(self.handle, 1))
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            keep = []
            params = {f"{comp}.{k}": v for comp in COMPONENTS for k, v in getattr(model, comp).state_dict().items()}
            for i in range(self.lib.mdpt_num_weights(self.handle)):
                name = self.lib.mdpt_weight_name(self.handle, i).decode()
                if name not in params:
                    raise RuntimeError(f"Missing key(s) in state_dict: \"{name}\"")
                t = params[name].detach().to(device=device)
                if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
                    t = t.to(torch.float32)
                t = t.contiguous()  # bound in the parameter's own dtype (MDPT_DTYPE_*): the pack kernels read it as it is
                keep.append(t)
                shape = (ctypes.c_int64 * t.dim())(*t.shape)
                native.check(self.lib, self.lib.mdpt_bind_weight(self.handle, name.encode(), t.data_ptr(), native.dtype_code(t.dtype), t.dim(), shape))
            nbytes = ctypes.c_size_t()
            native.check(self.lib, self.lib.mdpt_packed_bytes(self.handle, ctypes.byref(nbytes)))
            self.packed = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=device)
            self._packed_ptr = (self.packed.data_ptr() + 255) & ~255
            native.check(self.lib, self.lib.mdpt_finalize(self.handle, self._packed_ptr, nbytes.value, stream))
            torch.cuda.current_stream(device).synchronize()  # fp32 sources in `keep` may be freed after this
            del keep

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mdpt_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- tensor plumbing
    def as_input(self, t: Tensor, ndim: int, keep_dtype: bool = False) -> Tensor:
        if not isinstance(t, torch.Tensor):
            raise TypeError("expected a torch.Tensor")
        if t.dim() != ndim:
            raise RuntimeError(f"expected a {ndim}-D tensor, got shape {tuple(t.shape)}")
        if t.device != self.device:
            raise RuntimeError(f"Expected all tensors to be on the same device, model is on {self.device} but input is on {t.device}")
        if keep_dtype and t.dtype in (torch.float32, torch.bfloat16, torch.float16):  # mdpt_forward takes dtype-tagged tensors
            return t.detach().contiguous()
        return t.detach().to(dtype=torch.float32).contiguous()

    def as_output(self, t: Tensor) -> Tensor:
        return t if self.dtype == torch.float32 else t.to(self.dtype)

    def ptr_array(self, tensors):
        arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in tensors])
        arr._keep = tensors
        return arr

    def workspace(self, batch: int, size_hw: tuple[int, int]) -> tuple[int, int]:
        # one scratch buffer per (shape, stream): work queued on two different streams never shares activations (the library orders
        # everything on the stream it is given and keeps no other state per call)
        key = (batch, int(size_hw[0]), int(size_hw[1]), torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._workspaces.get(key)
        nbytes = ctypes.c_size_t()  # asked every call (host-side arithmetic only): settings such as the batch split change the need
        native.check(self.lib, self.lib.mdpt_workspace_bytes(self.handle, batch, key[1], key[2], ctypes.byref(nbytes)))
        if ws is None or ws.numel() < nbytes.value + 256:
            if len(self._workspaces) >= 4 and not torch.cuda.is_current_stream_capturing():  # (no device-wide wait inside a graph capture: keep all)
                torch.cuda.synchronize(self.device)  # buffers of other streams may still be in use
                self._workspaces.clear()
            ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
            self._workspaces[key] = ws
            # the grid cache (mdpt_set_grid_cache) is keyed on the workspace ADDRESS and trusts the bytes behind it: a fresh allocation may land where a
            # freed workspace was - drop the slots whenever memory is (re)allocated
            if self._grid_cache_on:
                native.check(self.lib, self.lib.mdpt_set_grid_cache(self.handle, 1))
        ptr = (ws.data_ptr() + 255) & ~255
        return ptr, ws.numel() - (ptr - ws.data_ptr())

    def call(self, fn_name: str, *args, size_hw, batch):
        """Invoke a stage entry point: (handle, *args, workspace, ws_bytes, stream) on the current torch stream."""
        with torch.cuda.device(self.device):
            ws_ptr, ws_bytes = self.workspace(batch, size_hw)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            cargs = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
            native.check(self.lib, getattr(self.lib, fn_name)(self.handle, *cargs, ws_ptr, ws_bytes, stream))

    def call_checked(self, fn_name: str, *args, size_hw, batch):
        """call(), with the library's grid-shape error surfaced as the RuntimeError the reference raises for such inputs."""
        try:
            self.call(fn_name, *args, size_hw=size_hw, batch=batch)
        except native.MdptError as e:
            if e.code == native.E_GRID:
                raise RuntimeError(str(e)) from None
            raise

    def export_tap(self, which: int, out: Tensor, batch: int, size_hw):
        with torch.cuda.device(self.device):
            ws_ptr, ws_bytes = self.workspace(batch, size_hw)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            native.check(self.lib, self.lib.mdpt_export_tap(self.handle, which, out.data_ptr(), ws_ptr, ws_bytes, stream))


class DPTModel(nn.Module):
    """Drop-in for the reference's DPTModel (muggled_dpt/dpt_model.py:21-168) running on libmdpt."""

    def __init__(self, config: dict, family: str = "v2"):
        super().__init__()
        self.config = dict(config)
        # "v2": Depth-Anything V2 (taps after each quarter of the blocks); "v1": taps after the last 4 blocks; "beit": MiDaS v3.1 BEiT
        self.family = family
        if family == "beit":
            from .state_dict_conversion_beit import expected_new_keys as beit_keys
            keys = beit_keys(self.config)
        elif family == "swinv2":
            from .state_dict_conversion_swinv2 import expected_new_keys as swin_keys
            keys = swin_keys(self.config)
        else:
            keys = expected_new_keys(self.config, family)
        shapes = _new_key_shapes(self.config, family)
        per = {comp: {k: shapes[f"{comp}.{k}"] for k in keys[comp]} for comp in COMPONENTS}
        default_px = self.config["base_patch_grid_hw"][0] * self.config["patch_size_px"]
        if family == "beit":
            self.patch_embed = PatchEmbed(per["patch_embed"], self.config["patch_size_px"], default_px, BEIT_RGB_MEAN, BEIT_RGB_STD)
        elif family == "swinv2":  # same normalisation as BEiT (v31_swinv2/patch_embed.py:39-40), sizes snap to 8 patches
            self.patch_embed = PatchEmbed(per["patch_embed"], self.config["patch_size_px"], default_px, BEIT_RGB_MEAN, BEIT_RGB_STD, 8)
        else:
            self.patch_embed = PatchEmbed(per["patch_embed"], self.config["patch_size_px"], default_px)
        self.imgencoder = ImageEncoder("imgencoder", per["imgencoder"])
        self.reassemble = ReassembleModel("reassemble", per["reassemble"])
        self.fusion = FusionModel("fusion", per["fusion"])
        self.head = MonocularDepthHead("head", per["head"])
        for comp in COMPONENTS:
            getattr(self, comp).__dict__["_owner"] = self
        # enable_optimizations=False: every block gets an `attn.softmax` module that hooks can observe (reference Attention class,
        # components/transformer_block.py:79-136). The arithmetic path is the same fused kernel either way.
        if self.config.get("enable_optimizations", True) is False and family in ("v2", "v1", "beit"):
            probes = []
            bps = max(1, self.config["num_blocks"] // 4)
            for blk in range(self.config["num_blocks"]):
                node = self.imgencoder.blocks[blk] if family == "v1" else self.imgencoder.stages[blk // bps].blocks[blk % bps]
                node.attn.add_module("softmax", _SoftmaxProbe(dim=-1))
                probes.append(node.attn.softmax)
            self.imgencoder.__dict__["_softmax_probes"] = probes
        if family == "swinv2":  # the window attention always runs through a hookable nn.Softmax (windowed_attention.py:60-61,119)
            probes = []
            for s, nl in enumerate(self.config["layers_per_stage"]):
                for l in range(int(nl)):
                    node = self.imgencoder.stages[s].blocks[l]
                    node.attn.add_module("softmax", _SoftmaxProbe(dim=-1))
                    probes.append(node.attn.softmax)
            self.imgencoder.__dict__["_softmax_probes"] = probes
        # every transformer block is hookable (output tokens), in block order; SwinV2 stage-major, with the stage of each block
        block_nodes = []
        if family == "swinv2":
            for s, nl in enumerate(self.config["layers_per_stage"]):
                block_nodes += [(self.imgencoder.stages[s].blocks[l], s) for l in range(int(nl))]
        elif family == "v1":
            block_nodes = [(self.imgencoder.blocks[i], 0) for i in range(self.config["num_blocks"])]
        else:
            bps = max(1, self.config["num_blocks"] // 4)
            block_nodes = [(self.imgencoder.stages[i // bps].blocks[i % bps], i // bps) for i in range(self.config["num_blocks"])]
        for node, _ in block_nodes:
            node.__class__ = SwinTransformerBlock if family == "swinv2" else TransformerBlock
        self.imgencoder.__dict__["_block_probes"] = block_nodes
        self.__dict__["_engine_obj"] = None
        self.__dict__["_param_cache"] = None
        self.__dict__["_gemm_tile"] = 0
        self.eval()  # inference only (dpt_model.py:57)

    # ---- engine lifetime
    def _invalidate(self):
        self.__dict__["_engine_obj"] = None
        self.__dict__["_param_cache"] = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    def _param_versions(self) -> tuple:
        """(identity of every Parameter object, version counter of every parameter), in a fixed order: what an engine's packed snapshot is
        compared against. The parameter LIST is cached (the cache holds the Parameter objects, so an id can not be reused behind its back)
        and re-collected when a Parameter was assigned / registered / deleted / converted somewhere in ANY model tree (_TREE_GENERATION -
        process-wide, so it only triggers the walk; it is not part of the key), the model was moved or cast (_apply), a state dict was
        loaded, and unconditionally every _RECHECK_EVERY calls (raw writes to module._parameters[...] by third-party utilities bypass
        every hook: DPTModel.refresh_weights() is the documented way to pick those up at once). Inference tensors (a model built / loaded /
        moved under torch.inference_mode()) have no version counter and can not be modified in place either: they contribute a constant."""
        cache = self.__dict__.get("_param_cache")
        calls = self.__dict__.get("_param_calls", 0) + 1
        self.__dict__["_param_calls"] = calls
        if cache is None or cache[0] != _TREE_GENERATION[0] or calls % _RECHECK_EVERY == 0:
            plist = list(self.parameters())
            versioned = [p for p in plist if not p.is_inference()]
            cache = (_TREE_GENERATION[0], plist, versioned, tuple([id(p) for p in plist]))
            self.__dict__["_param_cache"] = cache
        return (cache[3], tuple([p._version for p in cache[2]]))

    def _get_engine(self) -> _Engine:
        """The engine holds a PACKED SNAPSHOT of the weights (bf16 hi[/lo] panels, layer scales folded in). It is rebuilt when the model
        moves / changes dtype (_apply), when a state dict is loaded, and when any parameter was modified in place since the snapshot
        (tensor version counters: p.data.copy_(), nn.init.*, p.mul_() under no_grad ... - the reference always reads the live tensors)."""
        p = next(self.parameters())
        eng = self.__dict__["_engine_obj"]
        versions = self._param_versions()
        if eng is None or eng.device != p.device or eng.dtype != p.dtype or eng.param_versions != versions:
            eng = _Engine(self, p.device, p.dtype)
            eng.param_versions = versions
            self.__dict__["_engine_obj"] = eng
        return eng

    def refresh_weights(self) -> None:
        """Force a re-pack of the weights on the next call (for edits the version counters cannot see, e.g. writes through raw pointers)."""
        self._invalidate()

    def set_precision(self, mode: str | None) -> None:
        """Arithmetic mode of the MFMA operands, independent of the tensor dtype at the boundary: "bf16", "bf16x3", "fp16", "fp16x3",
        "mixed" (fp16 operands, three passes for the op classes that dominate the error: <= 1e-3 of the fp32 reference at ~3/4 of the
        single-pass throughput), or None = the default of the model's dtype (fp32 -> bf16x3, bf16 -> bf16, fp16 -> fp16)."""
        if mode is not None and mode not in native.PRECISIONS:
            raise ValueError(f"unknown precision mode {mode!r}: one of {sorted(native.PRECISIONS)} or None")
        self.__dict__["_precision"] = mode
        self._invalidate()

    def set_class_passes(self, passes: dict[str, int] | None) -> None:
        """Per op class (native.OP_CLASSES: "patch", "qkv", "attn", "proj", "fc1", "fc2", "reasm", "fusion", "fusion_in", "fusion_proj", "head",
        "head_tail") MFMA pass count on top of the precision mode (mdpt_set_class_passes): 1, 3 (both operands split) or 2 (activations split,
        weights one plane; not for "attn"): the knob the error-budget study turns. native.PASSES_2F8 (4) / PASSES_3F8 (5), for the classes in
        native.F8_CLASSES: the same two / three products with the cross terms on fp8 operands (1.5 / 2 pass-equivalents; fp16 operand modes)."""
        for k, v in (passes or {}).items():
            ok = k in native.OP_CLASSES and (int(v) in (1, 2, 3) or (int(v) in (native.PASSES_2F8, native.PASSES_3F8) and k in native.F8_CLASSES))
            if not ok or (k == "attn" and int(v) == 2):
                raise ValueError(f"bad class pass entry {k!r}: {v!r}")
        self.__dict__["_class_passes"] = dict(passes) if passes else None
        self._invalidate()

    def _debug_set_wscale_policy(self, scale_all: bool) -> None:
        """Test hook (mdpt_debug_set_wscale_policy): the fp16 build's power-of-two weight scale applied to EVERY layer-scale-folded matrix instead of
        only those whose largest entry is below 2^-5 - an equally valid rounding of the same weights, under which the mixed mode's tolerance is asserted too."""
        self.__dict__["_dbg_wscale_all"] = bool(scale_all)
        self._invalidate()

    def set_weight_rounding_compensation(self, on: bool | None) -> None:
        """Token-mean compensation of the weight rounding in the fp16 operand modes (mdpt_set_weight_rounding_compensation): None = the
        mode's default (on for "mixed"; off for single-pass "fp16", where the decoder's own rounding dominates and the map does not improve),
        True / False force it; a collection of class names ("qkv", "proj", "fc1", "fc2") compensates those classes only."""
        if on is not None and not isinstance(on, bool):  # a collection of class names: compensate those only
            names = tuple(on)
            if not names or any(n not in ("qkv", "proj", "fc1", "fc2") for n in names):
                raise ValueError(f"compensated classes are qkv, proj, fc1, fc2: {names!r}")
            on = sum(1 << native.OP_CLASSES.index(n) for n in set(names))
        self.__dict__["_wrc"] = on
        self._invalidate()

    def set_gemm_tile(self, tile: int) -> None:
        """Benchmark knob: 0 auto, 1 = 128x128, 2 = 256x256 GEMM tiles."""
        self.__dict__["_gemm_tile"] = int(tile)
        self._invalidate()

    def set_latency_mode(self, on: bool = True) -> None:
        """Off (default): an image's result does not depend on the batch it is part of, bit for bit. On: launches too small to fill the
        GPU (batch 1 of the small models) may use faster forms whose summation order differs (mdpt_set_latency_mode)."""
        self.__dict__["_latency_mode"] = bool(on)
        eng = self.__dict__.get("_engine_obj")
        if eng is not None:
            native.check(eng.lib, eng.lib.mdpt_set_latency_mode(eng.handle, int(bool(on))))

    def set_nonfinite_propagation(self, on: bool = True) -> None:
        """On (default): an image tensor that holds a NaN / inf gives an all-NaN depth map for that image, in every arithmetic mode - what the
        reference's forward returns (dpt_model.py:61-83). Off: the saturating converts / v_max ReLUs of the kernels decide (a finite, meaningless
        map); saves one small launch and one memset per forward (mdpt_set_nonfinite_propagation)."""
        self.__dict__["_nonfinite_propagation"] = bool(on)
        eng = self.__dict__.get("_engine_obj")
        if eng is not None:
            native.check(eng.lib, eng.lib.mdpt_set_nonfinite_propagation(eng.handle, int(bool(on))))

    def export(self, path: str, dtype: torch.dtype | None = None) -> dict:
        """Write the deployment artefact of this model: a `.mdpt` file (configuration incl. family, arithmetic mode and per-class passes,
        preprocessing constants, every parameter under its reference key) that a host without PyTorch runs through the C ABI at any legal
        image size - the native analogue of the reference's ONNX export with dynamic axes (experiments/export_onnx.py:119-148). Format and
        readers: muggled_dpt_amd/export.py, tests/c_host/host_main.cpp, tools/mdpt_model_file.py."""
        from .export import export_model
        return export_model(self, path, dtype)

    # ---- reference API
    def forward(self, image_rgb_normalized_bchw: Tensor) -> Tensor:
        """[B,3,H,W] normalised RGB -> inverse depth [B,H,W] (dpt_model.py:61-83), one fused C-ABI call."""
        probes = self.imgencoder.__dict__.get("_softmax_probes") or []
        blocks = self.imgencoder.__dict__.get("_block_probes") or []
        if any(len(pr._forward_hooks) > 0 for pr in probes) or any(len(node._forward_hooks) > 0 for node, _ in blocks):
            # somebody is listening on attention softmax modules or on blocks: go stage by stage so the encoder can dump what they see
            tokens, hw = self.patch_embed(image_rgb_normalized_bchw)
            return self.head(self.fusion(*self.reassemble(*self.imgencoder(tokens, hw), hw)))
        eng = self._get_engine()
        x = eng.as_input(image_rgb_normalized_bchw, 4, keep_dtype=True)
        b, c, h, w = x.shape
        if c != 3:
            raise RuntimeError(f"expected 3 input channels, got {c}")
        # image in, depth out in their own dtypes (dtype-tagged pointers at the C ABI): no cast kernels around the call; the output
        # dtype is the model's, like the reference's (dpt_model.py:105-107)
        out = torch.empty((b, h, w), device=x.device, dtype=eng.dtype)
        try:
            eng.call("mdpt_forward", x, native.dtype_code(x.dtype), b, h, w, out, native.dtype_code(out.dtype), size_hw=(h, w), batch=b)
        except native.MdptError as e:
            if e.code == native.E_GRID:
                raise RuntimeError(str(e)) from None  # the reference raises RuntimeError for odd grids too
            raise
        return out

    def inference(self, image_bgr: np.ndarray, max_side_length: int | None = None, use_square_sizing: bool = True) -> Tensor:
        """prepare_image + forward under inference_mode -> [1,H,W] (dpt_model.py:87-109)."""
        with torch.inference_mode():
            probes = self.imgencoder.__dict__.get("_softmax_probes") or []
            blocks = self.imgencoder.__dict__.get("_block_probes") or []
            pe = self.patch_embed
            scaled_hw, interp, p, img_dtype = pe._prepare_plan(image_bgr, max_side_length, use_square_sizing, "bilinear")
            if img_dtype != p.dtype or any(len(pr._forward_hooks) > 0 for pr in probes) or any(len(node._forward_hooks) > 0 for node, _ in blocks):
                return self(pe.prepare_image(image_bgr, max_side_length, use_square_sizing))  # (hooks listening: the stage-by-stage route of forward())
            # mdpt_forward_bgr: the im2col kernel of the patch embedding reads the uint8 image itself (prepare_image fused into patchify) - same bits as
            # prepare_image + forward, one launch and one trip of the image through memory fewer
            eng = self._get_engine()
            h, w = scaled_hw
            with torch.cuda.device(p.device):
                src = pe._stage_host_image(image_bgr, p.device)
                out = torch.empty((1, h, w), device=p.device, dtype=eng.dtype)
                mean3, std3 = pe._norm_constants()
                eng.call_checked("mdpt_forward_bgr", src, image_bgr.shape[0], image_bgr.shape[1], native.dtype_code(img_dtype), h, w, mean3, std3, interp,
                                 out, native.dtype_code(out.dtype), size_hw=(h, w), batch=1)
            return out

    def prepare_image_bgr(self, image_bgr: np.ndarray, max_side_length: int | None = None, use_square_sizing: bool = True,
                          interpolation_mode: str = "bilinear") -> Tensor:
        return self.patch_embed.prepare_image(image_bgr, max_side_length, use_square_sizing, interpolation_mode)

    def verify_input(self, image_rgb_normalized_bchw: Tensor) -> bool:
        """Same assertions as dpt_model.py:133-166."""
        assert isinstance(image_rgb_normalized_bchw, torch.Tensor), "Image must be provided as a tensor!"
        p = next(self.parameters())
        img = image_rgb_normalized_bchw
        assert img.device == p.device, f"Device mismatch! Image: {img.device}, model: {p.device}"
        assert img.dtype == p.dtype, f"Data type mismatch! Image: {img.dtype}, model: {p.dtype}"
        shape_str = "x".join(str(v) for v in img.shape)
        assert len(img.shape) == 4, f"Bad image shape! Image ({shape_str}) should have a shape of BxCXHxW"
        return self.patch_embed.verify_input(img)

    # ---- extras (not in the reference): stage boundaries of the last forward, for parity tests
    def debug_taps(self, batch: int, size_hw: tuple[int, int]) -> dict:
        eng = self._get_engine()
        h, w = size_hw
        dev = eng.device
        out = {"stages": [], "reasm": []}
        if eng.swin:
            g0h, g0w = h // eng.P, w // eng.P
            for s in range(4):
                t = torch.empty((batch, (g0h >> s) * (g0w >> s), eng.stage_features[s]), device=dev, dtype=torch.float32)
                eng.export_tap(s, t, batch, size_hw)
                out["stages"].append(t)
            for s in range(4):
                t = torch.empty((batch, eng.C, g0h >> s, g0w >> s), device=dev, dtype=torch.float32)
                eng.export_tap(4 + s, t, batch, size_hw)
                out["reasm"].append(t)
            t = torch.empty((batch, eng.C, 2 * g0h, 2 * g0w), device=dev, dtype=torch.float32)
            eng.export_tap(8, t, batch, size_hw)
            out["fused"] = t
            return out
        gh, gw = h // eng.P, w // eng.P
        n = gh * gw + 1
        for i in range(4):
            t = torch.empty((batch, n, eng.F), device=dev, dtype=torch.float32)
            eng.export_tap(i, t, batch, size_hw)
            out["stages"].append(t)
        for i, (sh, sw) in enumerate([(4 * gh, 4 * gw), (2 * gh, 2 * gw), (gh, gw), (gh // 2, gw // 2)]):
            t = torch.empty((batch, eng.C, sh, sw), device=dev, dtype=torch.float32)
            eng.export_tap(4 + i, t, batch, size_hw)
            out["reasm"].append(t)
        t = torch.empty((batch, eng.C, 8 * gh, 8 * gw), device=dev, dtype=torch.float32)
        eng.export_tap(8, t, batch, size_hw)
        out["fused"] = t
        return out


def _new_key_shapes(cfg: dict, family: str = "v2") -> dict[str, tuple]:
    """Shapes of every new-format parameter, derived from the config (same inventory as the C side)."""
    from .synthetic import original_state_dict_shapes
    from .state_dict_conversion import original_to_new_key_table

    if family == "swinv2":
        from .state_dict_conversion_swinv2 import original_to_new_key_table as swin_table
        from .synthetic import swinv2_original_state_dict_shapes
        orig = swinv2_original_state_dict_shapes(cfg)
        shapes = {}
        for old, (comp, new) in swin_table(cfg).items():
            shp = tuple(orig[old])
            if new.endswith("q_bias") or new.endswith("v_bias"):
                heads = cfg["heads_per_stage"][int(new.split(".")[1])]
                shp = (1, heads, 1, shp[0] // heads)
            shapes[f"{comp}.{new}"] = shp
        return shapes
    if family == "beit":
        from .state_dict_conversion_beit import original_to_new_key_table as beit_table
        from .synthetic import beit_original_state_dict_shapes
        orig = beit_original_state_dict_shapes(cfg)
        shapes = {}
        for old, (comp, new) in beit_table(cfg).items():
            shp = tuple(orig[old])
            if new.endswith("q_bias") or new.endswith("v_bias"):
                shp = (1, cfg["num_heads"], 1, shp[0] // cfg["num_heads"])
            shapes[f"{comp}.{new}"] = shp
        return shapes
    orig = original_state_dict_shapes(cfg)
    table = original_to_new_key_table(cfg, family)
    shapes = {}
    for old, (comp, new) in table.items():
        shapes[f"{comp}.{new}"] = tuple(orig[old])
    pe = orig["pretrained.pos_embed"]
    shapes["imgencoder.posenc.cls_embedding"] = (1, 1, pe[2])
    shapes["imgencoder.posenc.base_patch_embedding"] = (1, pe[1] - 1, pe[2])
    return shapes
