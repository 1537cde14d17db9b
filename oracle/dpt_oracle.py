"""CPU ORACLE (TEST INFRASTRUCTURE - NOT PRODUCT CODE) for the Depth-Anything-V2 DPT hot path.

A plain-torch, fp32, functional restatement of the reference's forward path:

    patch-embed -> DINOv2 encoder (4 taps) -> reassemble -> RefineNet fusion -> depth head

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
module, and only as the *checker*. The product path (muggled_dpt_amd) never routes through it.

Parity pinning: the reference ships no tests or golden vectors ("parity unpinned" by the
reference itself). This oracle is instead pinned to the *imported reference*
(torch 2.10 CPU fp32): `tests/golden/gen_golden.py` (run in the build container, where
/root/reference exists) asserts oracle == reference to <= 2e-5 at every stage boundary and
writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` re-checks the oracle against those
committed fixtures on any machine.

Weights are a flat dict keyed by "<component>.<new-format key>" (e.g.
"imgencoder.stages.0.blocks.1.attn.qkv.weight"), fp32 CPU tensors.
Every function cites the reference lines it follows (paths relative to /root/reference/muggled_dpt).
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

RGB_MEAN = (0.485, 0.456, 0.406)  # v2_depthanything/patch_embed.py:38
RGB_STD = (0.229, 0.224, 0.225)  # v2_depthanything/patch_embed.py:39
LN_EPS = 1e-6  # components/misc_helpers.py:190-210 (LayerNormEPS6)


# ---------------------------------------------------------------------------------------------------------------------
# pre-processing


def prepared_size(img_h: int, img_w: int, max_side_length: int | None, use_square_sizing: bool,
                  default_size_px: int = 518, tiling_px: int = 28) -> tuple[int, int]:
    """Model tensor size for an input image: each side = max(1, round(side*scale/28))*28.

    patch_embed.py:121-130. Python's banker's rounding matters: 518/28 = 18.5 -> 18 -> 504.
    """
    if max_side_length is None:
        max_side_length = default_size_px
    largest = max(img_h, img_w)
    scale = max_side_length / largest
    targ = (largest, largest) if use_square_sizing else (img_h, img_w)
    return tuple(max(1, round(side * scale / tiling_px)) * tiling_px for side in targ)


def prepare_image(image_bgr, max_side_length: int | None = None, use_square_sizing: bool = True,
                  interpolation_mode: str = "bilinear", default_size_px: int = 518, tiling_px: int = 28,
                  rgb_mean=RGB_MEAN, rgb_std=RGB_STD) -> torch.Tensor:
    """uint8 HxWx3 BGR ndarray -> normalised fp32 [1,3,H',W'] (patch_embed.py:103-145; BEiT uses mean = std = 0.5,
    v31_beit/patch_embed.py:39-40)."""
    h, w = image_bgr.shape[0:2]
    size_hw = prepared_size(h, w, max_side_length, use_square_sizing, default_size_px, tiling_px)
    rgb = torch.from_numpy(image_bgr[:, :, ::-1].copy()).permute(2, 0, 1).to(torch.float32)  # :134-135
    x = F.interpolate(rgb.unsqueeze(0), size=size_hw, align_corners=False, antialias=True, mode=interpolation_mode)
    mean = torch.tensor(rgb_mean).view(1, 3, 1, 1)
    inv_std = 1.0 / torch.tensor(rgb_std).view(1, 3, 1, 1)
    return ((x / 255.0) - mean) * inv_std  # :145


# ---------------------------------------------------------------------------------------------------------------------
# stages


def patch_embed(w: dict, image_bchw: torch.Tensor) -> tuple[torch.Tensor, tuple[int, int]]:
    """Conv k=s=patch, then BFHW -> BNF (patch_embed.py:77-99)."""
    pw = w["patch_embed.proj.weight"]
    y = F.conv2d(image_bchw, pw, w["patch_embed.proj.bias"], stride=pw.shape[-1])
    grid_hw = (int(y.shape[2]), int(y.shape[3]))
    tokens = y.flatten(2).transpose(1, 2)
    if "patch_embed.norm.weight" in w:  # SwinV2: LayerNorm (default eps 1e-5) on the patch tokens (v31_swinv2/patch_embed.py:76-94)
        tokens = F.layer_norm(tokens, (tokens.shape[-1],), w["patch_embed.norm.weight"], w["patch_embed.norm.bias"], 1e-5)
    return tokens, grid_hw


def position_embedding(w: dict, grid_hw: tuple[int, int]) -> torch.Tensor:
    """Learned [1,Gh*Gw,F] grid resized to the patch grid with bicubic (A=-0.75), align_corners=False,
    no antialias (components/position_encoder.py:108-143)."""
    base = w["imgencoder.posenc.base_patch_embedding"].float()
    n_base, feat = base.shape[1], base.shape[2]
    g = int(math.isqrt(n_base))
    img = base.reshape(1, g, g, feat).permute(0, 3, 1, 2)
    img = F.interpolate(img, size=grid_hw, mode="bicubic", antialias=False)
    return img.permute(0, 2, 3, 1).reshape(1, -1, feat)


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), weight, bias, LN_EPS)


def attention(w: dict, pre: str, x: torch.Tensor, num_heads: int, capture: list | None = None) -> torch.Tensor:
    """qkv Linear -> per-head softmax(q k^T / sqrt(d)) v -> proj Linear
    (components/transformer_block.py:105-136 / :154-170; both forms are the same math). `capture` collects the
    [B, heads, N, N] softmax output, i.e. what a forward hook on the non-optimised form's nn.Softmax (:101, :131) sees."""
    b, n, c = x.shape
    d = c // num_heads
    qkv = F.linear(x, w[f"{pre}.qkv.weight"], w[f"{pre}.qkv.bias"]).reshape(b, n, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q * d**-0.5) @ k.transpose(-2, -1), dim=-1)
    if capture is not None:
        capture.append(att)
    y = (att @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(y, w[f"{pre}.proj.weight"], w[f"{pre}.proj.bias"])


def mlp(w: dict, pre: str, x: torch.Tensor) -> torch.Tensor:
    """Linear(F->4F) -> exact (erf) GELU -> Linear(4F->F) (components/misc_helpers.py:88-120); ViT-G: SwiGLU FFN,
    Linear(F->2h) split in halves (a, b) -> silu(a) * b -> Linear(h->F) (:170-185)."""
    if f"{pre}.inner_linear_doubled.weight" in w:
        a, b = F.linear(x, w[f"{pre}.inner_linear_doubled.weight"], w[f"{pre}.inner_linear_doubled.bias"]).chunk(2, dim=-1)
        return F.linear(F.silu(a) * b, w[f"{pre}.outer_linear.weight"], w[f"{pre}.outer_linear.bias"])
    h = F.gelu(F.linear(x, w[f"{pre}.layers.0.weight"], w[f"{pre}.layers.0.bias"]))
    return F.linear(h, w[f"{pre}.layers.2.weight"], w[f"{pre}.layers.2.bias"])


def transformer_block(w: dict, pre: str, x: torch.Tensor, num_heads: int, capture: list | None = None) -> torch.Tensor:
    """Pre-norm block with layer-scale: t + g1*attn(LN1 t); u + g2*mlp(LN2 u) (transformer_block.py:53-65)."""
    a = attention(w, f"{pre}.attn", layernorm(x, w[f"{pre}.norm1.weight"], w[f"{pre}.norm1.bias"]), num_heads, capture)
    x = x + w[f"{pre}.scale_attn"] * a
    m = mlp(w, f"{pre}.mlp", layernorm(x, w[f"{pre}.norm2.weight"], w[f"{pre}.norm2.bias"]))
    return x + w[f"{pre}.scale_mlp"] * m


def image_encoder(w: dict, cfg: dict, patch_tokens: torch.Tensor, grid_hw: tuple[int, int], capture: list | None = None,
                  block_outputs: list | None = None) -> list[torch.Tensor]:
    """+pos-embed, prepend cls(+cls_embedding), 4 stages of round(num_blocks/4) blocks, shared out-norm on
    each stage output (image_encoder_model.py:80-94, :69, :136-147; position_encoder.py:55-76)."""
    b = patch_tokens.shape[0]
    cls = w["imgencoder.cls_token"] + w["imgencoder.posenc.cls_embedding"]
    tokens = torch.cat((cls.expand(b, -1, -1), patch_tokens + position_embedding(w, grid_hw)), dim=1)
    taps = []
    if "imgencoder.blocks.0.norm1.weight" in w:
        # Depth-Anything V1: flat block list, tapped after each of the LAST FOUR blocks
        # (v1_depthanything/image_encoder_model.py:55-61)
        n = cfg["num_blocks"]
        for i in range(n):
            tokens = transformer_block(w, f"imgencoder.blocks.{i}", tokens, cfg["num_heads"], capture)
            if block_outputs is not None:
                block_outputs.append(tokens)  # what a forward hook on the block sees (tests of mdpt_encoder_probe_blocks)
            if i >= n - 4:
                taps.append(tokens)
    else:
        per_stage = int(round(cfg["num_blocks"] / 4))
        for s in range(4):
            for i in range(per_stage):
                tokens = transformer_block(w, f"imgencoder.stages.{s}.blocks.{i}", tokens, cfg["num_heads"], capture)
                if block_outputs is not None:
                    block_outputs.append(tokens)
            taps.append(tokens)
    return [layernorm(t, w["imgencoder.outnorm.weight"], w["imgencoder.outnorm.bias"]) for t in taps]


_REASM = ("spatial_upx4", "spatial_upx2", "spatial_noscale", "spatial_downx2")


# ---------------------------------------------------------------------------------------------------------------------
# MiDaS v3.1 BEiT family (reference muggled_dpt/v31_beit/*)


def is_beit(w: dict) -> bool:
    return "imgencoder.stages.0.blocks.0.attn.q_bias" in w


def beit_relative_position_index(grid_hw: tuple[int, int]) -> torch.Tensor:
    """[N,N] int64 index into the (resized table ++ 3 cls entries) LUT; token i attends token j with
    idx = (yi - yj + gh - 1) * (2 gw - 1) + (xi - xj + gw - 1); row 0 / column 0 / [0,0] take the three cls entries
    (components/relative_positional_encoder.py:126-186)."""
    gh, gw = grid_hw
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    rel = (ys[:, None] - ys[None, :] + gh - 1) * (2 * gw - 1) + (xs[:, None] - xs[None, :] + gw - 1)
    num_rel = (2 * gh - 1) * (2 * gw - 1)
    idx = torch.zeros((gh * gw + 1, gh * gw + 1), dtype=torch.int64)
    idx[1:, 1:] = rel
    idx[0, :] = num_rel
    idx[:, 0] = num_rel + 1
    idx[0, 0] = num_rel + 2
    return idx


def beit_relpos_bias(lut: torch.Tensor, base_grid_hw: tuple[int, int], grid_hw: tuple[int, int]) -> torch.Tensor:
    """[1,H,N,N] additive attention bias: the learned [(2G-1)^2 + 3, H] table is resized (bilinear, align_corners=False,
    no antialias) to (2gh-1)x(2gw-1), the 3 cls entries are appended and the result is gathered by
    beit_relative_position_index (relative_positional_encoder.py:190-229)."""
    heads = lut.shape[1]
    rh, rw = 2 * base_grid_hw[0] - 1, 2 * base_grid_hw[1] - 1
    table = lut[: rh * rw].reshape(1, rh, rw, heads).permute(0, 3, 1, 2)
    nh, nw = 2 * grid_hw[0] - 1, 2 * grid_hw[1] - 1
    table = F.interpolate(table, size=(nh, nw), mode="bilinear")
    full = torch.cat([table.permute(0, 2, 3, 1).reshape(nh * nw, heads), lut[rh * rw:]])
    idx = beit_relative_position_index(grid_hw)
    n = idx.shape[0]
    return full[idx.reshape(-1)].reshape(n, n, heads).permute(2, 0, 1).unsqueeze(0)


def beit_attention(w: dict, pre: str, x: torch.Tensor, cfg: dict, grid_hw: tuple[int, int], capture: list | None = None) -> torch.Tensor:
    """qkv Linear without bias, +q_bias / +v_bias (k has none), softmax(q k^T / sqrt(d) + relpos) v, proj
    (image_encoder_model.py:331-356)."""
    b, n, c = x.shape
    heads = cfg["num_heads"]
    qkv = F.linear(x, w[f"{pre}.qkv.weight"]).reshape(b, n, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] + w[f"{pre}.q_bias"], qkv[1], qkv[2] + w[f"{pre}.v_bias"]
    att = (q * (c // heads) ** -0.5) @ k.transpose(-2, -1)
    att = att + beit_relpos_bias(w[f"{pre}.relpos_enc.ref_bias_lut"], cfg["base_patch_grid_hw"], grid_hw)
    att = torch.softmax(att, dim=-1)
    if capture is not None:
        capture.append(att)
    y = (att @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(y, w[f"{pre}.proj.weight"], w[f"{pre}.proj.bias"])


def beit_image_encoder(w: dict, cfg: dict, patch_tokens: torch.Tensor, grid_hw: tuple[int, int], capture: list | None = None,
                       block_outputs: list | None = None) -> list[torch.Tensor]:
    """cls ++ patch tokens (no absolute position embedding), 4 stages, raw stage outputs are the taps (no out-norm)
    (image_encoder_model.py:80-99, block :241-251)."""
    tokens = torch.cat((w["imgencoder.cls_token"].expand(patch_tokens.shape[0], -1, -1), patch_tokens), dim=1)
    per_stage = int(round(cfg["num_blocks"] / 4))
    taps = []
    for s in range(4):
        for i in range(per_stage):
            pre = f"imgencoder.stages.{s}.blocks.{i}"
            a = beit_attention(w, f"{pre}.attn", layernorm(tokens, w[f"{pre}.norm1.weight"], w[f"{pre}.norm1.bias"]), cfg, grid_hw, capture)
            tokens = tokens + w[f"{pre}.scale_attn"] * a
            m = mlp(w, f"{pre}.mlp", layernorm(tokens, w[f"{pre}.norm2.weight"], w[f"{pre}.norm2.bias"]))
            tokens = tokens + w[f"{pre}.scale_mlp"] * m
            if block_outputs is not None:
                block_outputs.append(tokens)
        taps.append(tokens)
    return taps


def beit_readout(w: dict, pre: str, tokens: torch.Tensor) -> torch.Tensor:
    """GELU(Linear(concat(patch token, cls token))) -> [B, N-1, F] (components/readout_projection.py:41-79)."""
    cls, img = tokens[:, :1], tokens[:, 1:]
    cat = torch.cat((img, cls.expand_as(img)), dim=-1)
    return F.gelu(F.linear(cat, w[f"{pre}.readout_proj.1.weight"], w[f"{pre}.readout_proj.1.bias"]))


def beit_reassemble(w: dict, stage_tokens: list[torch.Tensor], grid_hw: tuple[int, int]) -> list[torch.Tensor]:
    """readout projection, tokens->BCHW, then the same project/resample/fuse_proj chain as Depth-Anything
    (v31_beit/reassembly_model.py:114-127)."""
    outs = []
    for name, tok in zip(_REASM, stage_tokens):
        p = f"reassemble.{name}"
        x = beit_readout(w, p, tok).transpose(1, 2).unflatten(2, grid_hw)
        x = F.conv2d(x, w[f"{p}.resample.0.weight"], w[f"{p}.resample.0.bias"])
        if name == "spatial_upx4":
            x = F.conv_transpose2d(x, w[f"{p}.resample.1.weight"], w[f"{p}.resample.1.bias"], stride=4)
        elif name == "spatial_upx2":
            x = F.conv_transpose2d(x, w[f"{p}.resample.1.weight"], w[f"{p}.resample.1.bias"], stride=2)
        elif name == "spatial_downx2":
            x = F.conv2d(x, w[f"{p}.resample.1.weight"], w[f"{p}.resample.1.bias"], stride=2, padding=1)
        outs.append(F.conv2d(x, w[f"{p}.fuse_proj.weight"], None, padding=1))
    return outs


# ---------------------------------------------------------------------------------------------------------------------
# MiDaS v3.1 SwinV2 family (reference muggled_dpt/v31_swinv2/*)

_SWIN_REASM = ("spatial_noscale", "spatial_downx2", "spatial_downx4", "spatial_downx8")


def is_swin(w: dict) -> bool:
    return "imgencoder.patch_merge_layers.0.reduction.weight" in w


def swin_window_and_shift(grid_hw: tuple[int, int], target_hw: tuple[int, int]) -> tuple[tuple[int, int], tuple[int, int]]:
    """Window / shift sizes for a patch grid: min(target, grid) if it tiles the grid, else the divisor of the grid side in
    [win/2, 2 win) closest to the grid side; shift = win // 2 unless one window covers the side
    (components/windowed_attention.py:345-388)."""
    out_win, out_shift = [], []
    for patch, targ in zip(grid_hw, target_hw):
        win = min(targ, patch)
        if patch % win:
            divisors = [d for d in range(win // 2, 2 * win) if patch % d == 0]
            win = min(divisors, key=lambda d: abs(patch - d))
        out_win.append(win)
        out_shift.append(0 if patch <= win else win // 2)
    return tuple(out_win), tuple(out_shift)


def swin_partition(x_bhwc: torch.Tensor, win_hw: tuple[int, int]) -> torch.Tensor:
    """[B,H,W,C] -> [B*nWy*nWx, wh*ww, C], windows row-major, tokens row-major inside a window (:262-289)."""
    b, h, w, c = x_bhwc.shape
    wh, ww = win_hw
    x = x_bhwc.reshape(b, h // wh, wh, w // ww, ww, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, wh * ww, c)


def swin_unpartition(win_tokens: torch.Tensor, win_hw: tuple[int, int], bhwc: tuple[int, int, int, int]) -> torch.Tensor:
    b, h, w, c = bhwc
    wh, ww = win_hw
    x = win_tokens.reshape(b, h // wh, w // ww, wh, ww, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(b, h, w, c)


def swin_shift_mask(grid_hw, win_hw, shift_hw) -> torch.Tensor:
    """[nW, 1, wa, wa] additive mask (0 / -100) separating the regions a cyclic shift glues together (:394-439). Built with
    the same Python slices as the reference: with a zero shift in one dimension, slice(-0, None) covers the whole side."""
    gh, gw = grid_hw
    wh, ww = win_hw
    sh, sw = shift_hw
    img = torch.zeros((1, gh, gw, 1))
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    ids = swin_partition(img, win_hw).reshape(-1, wh * ww)
    diff = ids.unsqueeze(1) - ids.unsqueeze(2)
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).unsqueeze(1)


def swin_relative_position_index(win_hw) -> torch.Tensor:
    """[wa, wa] index into the (2wh-1)(2ww-1) offset table: (yi - yj + wh - 1) * (2ww - 1) + (xi - xj + ww - 1)
    (components/relative_positional_encoder.py:193-283; worked 2x3 example at :270-275)."""
    wh, ww = win_hw
    iy, ix = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    iy, ix = iy.flatten(), ix.flatten()
    return (iy[:, None] - iy[None, :] + wh - 1) * (2 * ww - 1) + (ix[:, None] - ix[None, :] + ww - 1)


def swin_cpb_bias(w: dict, pre: str, win_hw, pretrained_window, heads: int) -> torch.Tensor:
    """[1, heads, wa, wa] continuous position bias: 16 * sigmoid(MLP(2->512->heads)(log-spaced relative offsets)), gathered by
    the relative position index (components/relative_positional_encoder.py:60-93, :122-188)."""
    wh, ww = win_hw
    ys = torch.arange(-(wh - 1), wh, dtype=torch.float32)
    xs = torch.arange(-(ww - 1), ww, dtype=torch.float32)
    table = torch.stack(torch.meshgrid([ys, xs], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    table[..., 0] /= max((wh if pretrained_window is None else pretrained_window) - 1, 1)
    table[..., 1] /= max((ww if pretrained_window is None else pretrained_window) - 1, 1)
    table = torch.sign(table) * torch.log2(torch.abs(table * 8) + 1.0) / torch.log2(torch.tensor(8.0))
    hid = F.relu(F.linear(table, w[f"{pre}.bias_mlp.0.weight"], w[f"{pre}.bias_mlp.0.bias"]))
    lut = F.linear(hid, w[f"{pre}.bias_mlp.2.weight"]).reshape(-1, heads)
    idx = swin_relative_position_index(win_hw)
    bias = 16 * torch.sigmoid(lut[idx.reshape(-1)])
    return bias.reshape(wh * ww, wh * ww, heads).permute(2, 0, 1).unsqueeze(0)


def swin_window_attention(w: dict, pre: str, tokens: torch.Tensor, grid_hw, cfg: dict, stage: int, is_shift_block: bool,
                          capture: list | None = None) -> torch.Tensor:
    """Roll (shift blocks) -> windows -> cosine attention with logit scale, position bias and shift mask -> un-window -> roll
    back (windowed_attention.py:65-123, :171-260)."""
    b, n, c = tokens.shape
    gh, gw = grid_hw
    heads = cfg["heads_per_stage"][stage]
    win_hw, shift_hw = swin_window_and_shift(grid_hw, cfg["window_size_hw"])
    need_shift = is_shift_block and (shift_hw[0] > 0 or shift_hw[1] > 0)
    img = tokens.reshape(b, gh, gw, c)
    if need_shift:
        img = torch.roll(img, shifts=(-shift_hw[0], -shift_hw[1]), dims=(1, 2))
    win = swin_partition(img, win_hw)
    p, wa, _ = win.shape
    qkv = F.linear(win, w[f"{pre}.qkv.weight"]).reshape(p, wa, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] + w[f"{pre}.q_bias"], qkv[1], qkv[2] + w[f"{pre}.v_bias"]
    att = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    att = att * w[f"{pre}.logit_scale"]
    att = att + swin_cpb_bias(w, f"{pre}.relpos_enc", win_hw, cfg["pretrained_window_sizes_per_stage"][stage], heads)
    if need_shift:
        att = att + swin_shift_mask(grid_hw, win_hw, shift_hw).repeat(b, 1, 1, 1)
    weights = torch.softmax(att, dim=-1)  # the reference's hookable self.softmax (windowed_attention.py:60-61,119): [B*nW, heads, Nw, Nw]
    if capture is not None:
        capture.append(weights)
    out = (weights @ v).transpose(1, 2).reshape(p, wa, c)
    out = F.linear(out, w[f"{pre}.proj.weight"], w[f"{pre}.proj.bias"])
    img = swin_unpartition(out, win_hw, (b, gh, gw, c))
    if need_shift:
        img = torch.roll(img, shifts=shift_hw, dims=(1, 2))
    return img.reshape(b, n, c)


def swin_patch_merge(w: dict, pre: str, tokens: torch.Tensor, grid_hw):
    """cat(TL, BL, TR, BR) -> Linear(4C -> 2C, no bias) -> LayerNorm(eps 1e-5) (components/patch_merge.py:49-103)."""
    b, n, c = tokens.shape
    img = tokens.reshape(b, grid_hw[0], grid_hw[1], c)
    cat = torch.cat([img[:, 0::2, 0::2], img[:, 1::2, 0::2], img[:, 0::2, 1::2], img[:, 1::2, 1::2]], dim=-1)
    out_hw = (cat.shape[1], cat.shape[2])
    x = F.linear(cat.reshape(b, n // 4, 4 * c), w[f"{pre}.reduction.weight"])
    return F.layer_norm(x, (x.shape[-1],), w[f"{pre}.norm.weight"], w[f"{pre}.norm.bias"], 1e-5), out_hw


def swin_image_encoder(w: dict, cfg: dict, patch_tokens: torch.Tensor, grid_hw, capture: list | None = None,
                       block_outputs: list | None = None) -> list[torch.Tensor]:
    """4 stages of (plain, shifted) post-norm block pairs with a patch merge between stages; taps = stage outputs
    (image_encoder_model.py:77-98, :155-161, :213-225)."""
    tokens, hw, taps = patch_tokens, tuple(grid_hw), []
    for s in range(4):
        for l in range(cfg["layers_per_stage"][s]):
            pre = f"imgencoder.stages.{s}.blocks.{l}"
            a = swin_window_attention(w, f"{pre}.attn", tokens, hw, cfg, s, is_shift_block=bool(l % 2), capture=capture)
            tokens = tokens + F.layer_norm(a, (a.shape[-1],), w[f"{pre}.norm1.weight"], w[f"{pre}.norm1.bias"], 1e-5)
            m = mlp(w, f"{pre}.mlp", tokens)
            tokens = tokens + F.layer_norm(m, (m.shape[-1],), w[f"{pre}.norm2.weight"], w[f"{pre}.norm2.bias"], 1e-5)
            if block_outputs is not None:
                block_outputs.append(tokens)
        taps.append(tokens)
        if s < 3:
            tokens, hw = swin_patch_merge(w, f"imgencoder.patch_merge_layers.{s}", tokens, hw)
    return taps


def swin_reassemble(w: dict, stage_tokens: list[torch.Tensor], grid_hw) -> list[torch.Tensor]:
    """tokens -> BCHW at 1, 1/2, 1/4, 1/8 of the patch grid, 3x3 conv (no bias) to the fusion width
    (v31_swinv2/reassembly_model.py:113-122)."""
    outs = []
    for s, (name, tok) in enumerate(zip(_SWIN_REASM, stage_tokens)):
        hw = (grid_hw[0] // (2 ** s), grid_hw[1] // (2 ** s))
        x = tok.transpose(1, 2).unflatten(2, hw)
        outs.append(F.conv2d(x, w[f"reassemble.{name}.fuse_proj.weight"], None, padding=1))
    return outs


def beit_fusion(w: dict, reasm: list[torch.Tensor]) -> torch.Tensor:
    """Same dataflow as fusion(); parameter names differ (conv_seq / proj_seq, v31_beit/fusion_model.py:95-164)."""

    def rcu(pre, x):
        y = F.conv2d(F.relu(x), w[f"{pre}.conv_seq.1.weight"], w[f"{pre}.conv_seq.1.bias"], padding=1)
        y = F.conv2d(F.relu(y), w[f"{pre}.conv_seq.3.weight"], w[f"{pre}.conv_seq.3.bias"], padding=1)
        return y + x

    f = None
    for idx in (3, 2, 1, 0):
        p = f"fusion.blocks.{idx}"
        x = reasm[idx] if f is None else rcu(f"{p}.conv_reassembly", reasm[idx]) + f
        x = upsample_bilinear_ac(rcu(f"{p}.proj_seq.0", x), 2)
        f = F.conv2d(x, w[f"{p}.proj_seq.2.weight"], w[f"{p}.proj_seq.2.bias"])
    return f


def reassemble(w: dict, stage_tokens: list[torch.Tensor], grid_hw: tuple[int, int]) -> list[torch.Tensor]:
    """Per stage: drop cls, tokens->BCHW, 1x1 conv, {convT k4s4 | convT k2s2 | none | conv3x3 s2 p1},
    3x3 conv (no bias) to fusion channels (reassembly_model.py:61-94, :139-149, :208-211, :238-310)."""
    outs = []
    for name, tok in zip(_REASM, stage_tokens):
        p = f"reassemble.{name}"
        x = tok[:, 1:, :].transpose(1, 2).unflatten(2, grid_hw)
        x = F.conv2d(x, w[f"{p}.resample.0.weight"], w[f"{p}.resample.0.bias"])
        if name == "spatial_upx4":
            x = F.conv_transpose2d(x, w[f"{p}.resample.1.weight"], w[f"{p}.resample.1.bias"], stride=4)
        elif name == "spatial_upx2":
            x = F.conv_transpose2d(x, w[f"{p}.resample.1.weight"], w[f"{p}.resample.1.bias"], stride=2)
        elif name == "spatial_downx2":
            x = F.conv2d(x, w[f"{p}.resample.1.weight"], w[f"{p}.resample.1.bias"], stride=2, padding=1)
        outs.append(F.conv2d(x, w[f"{p}.fuse_proj.weight"], None, padding=1))
    return outs


def residual_conv_unit(w: dict, pre: str, x: torch.Tensor) -> torch.Tensor:
    """conv3x3(relu(conv3x3(relu(x)))) + x, both convs biased (fusion_model.py:205-220)."""
    y = F.conv2d(F.relu(x), w[f"{pre}.resconv_seq.1.weight"], w[f"{pre}.resconv_seq.1.bias"], padding=1)
    y = F.conv2d(F.relu(y), w[f"{pre}.resconv_seq.3.weight"], w[f"{pre}.resconv_seq.3.bias"], padding=1)
    return y + x


def upsample_bilinear_ac(x: torch.Tensor, scale: float) -> torch.Tensor:
    """F.interpolate(scale_factor, bilinear, align_corners=True) (components/misc_helpers.py:39-42)."""
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)


def fusion_block(w: dict, idx: int, reasm: torch.Tensor, prior: torch.Tensor | None) -> torch.Tensor:
    """blocks[idx] of the fusion model (fusion_model.py:89-114 top-most, :148-154 regular, :159-182)."""
    p = f"fusion.blocks.{idx}"
    x = reasm if prior is None else residual_conv_unit(w, f"{p}.conv_reassembly", reasm) + prior
    x = residual_conv_unit(w, f"{p}.scale_proj_seq.0", x)
    x = upsample_bilinear_ac(x, 2)
    return F.conv2d(x, w[f"{p}.scale_proj_seq.2.weight"], w[f"{p}.scale_proj_seq.2.bias"])


def fusion(w: dict, reasm: list[torch.Tensor]) -> torch.Tensor:
    """Coarse-to-fine: blocks[3](r4) -> blocks[2](r3, .) -> blocks[1](r2, .) -> blocks[0](r1, .)
    (fusion_model.py:55-80). Requires an even patch grid (odd grids mismatch at :151)."""
    f = fusion_block(w, 3, reasm[3], None)
    for idx in (2, 1, 0):
        f = fusion_block(w, idx, reasm[idx], f)
    return f


def head(w: dict, cfg: dict, fused: torch.Tensor) -> torch.Tensor:
    """conv3x3(C->C/2) -> bilinear x(patch/8) align_corners -> conv3x3(->32)+ReLU -> conv1x1(->1)+ReLU|Sigmoid
    -> squeeze (head_model.py:67-85, :89-106)."""
    x = F.conv2d(fused, w["head.spatial_upsampler.0.weight"], w["head.spatial_upsampler.0.bias"], padding=1)
    x = upsample_bilinear_ac(x, 2 if is_swin(w) else cfg["patch_size_px"] / 8)  # v31_swinv2/head_model.py:43
    x = F.relu(F.conv2d(x, w["head.proj_1ch.0.weight"], w["head.proj_1ch.0.bias"], padding=1))
    x = F.conv2d(x, w["head.proj_1ch.2.weight"], w["head.proj_1ch.2.bias"])
    x = torch.sigmoid(x) if cfg.get("is_metric", False) else F.relu(x)
    return x.squeeze(1)


def forward(w: dict, cfg: dict, image_bchw: torch.Tensor, return_stages: bool = False):
    """DPTModel.forward (dpt_model.py:61-83). With return_stages, also returns every stage boundary."""
    with torch.inference_mode():
        tokens, grid_hw = patch_embed(w, image_bchw)
        if grid_hw[0] % 2 or grid_hw[1] % 2:
            # the reference crashes in fusion (tensor size mismatch, fusion_model.py:151)
            raise RuntimeError(f"patch grid {grid_hw} must be even in both dimensions")
        if is_swin(w):
            if grid_hw[0] % 8 or grid_hw[1] % 8:
                # the reference crashes in patch_merge.py:91 (odd grids) or fusion_model.py (sizes not x2 apart)
                raise RuntimeError(f"patch grid {grid_hw} must be divisible by 8 in both dimensions")
            taps = swin_image_encoder(w, cfg, tokens, grid_hw)
            reasm = swin_reassemble(w, taps, grid_hw)
            fused = beit_fusion(w, reasm)
        elif is_beit(w):
            taps = beit_image_encoder(w, cfg, tokens, grid_hw)
            reasm = beit_reassemble(w, taps, grid_hw)
            fused = beit_fusion(w, reasm)
        else:
            taps = image_encoder(w, cfg, tokens, grid_hw)
            reasm = reassemble(w, taps, grid_hw)
            fused = fusion(w, reasm)
        depth = head(w, cfg, fused)
    if return_stages:
        return depth, {"patch_tokens": tokens, "grid_hw": grid_hw, "stages": taps, "reasm": reasm, "fused": fused}
    return depth


def inference(w: dict, cfg: dict, image_bgr, max_side_length=None, use_square_sizing=True) -> torch.Tensor:
    """DPTModel.inference (dpt_model.py:87-109): prepare_image + forward -> [1,H,W]."""
    default_px = cfg["base_patch_grid_hw"][0] * cfg["patch_size_px"]
    norm = dict(rgb_mean=(0.5, 0.5, 0.5), rgb_std=(0.5, 0.5, 0.5)) if (is_beit(w) or is_swin(w)) else {}
    tiling = (8 if is_swin(w) else 2) * cfg["patch_size_px"]  # v31_swinv2/patch_embed.py:68
    x = prepare_image(image_bgr, max_side_length, use_square_sizing, default_size_px=default_px, tiling_px=tiling, **norm)
    return forward(w, cfg, x)


# ---------------------------------------------------------------------------------------------------------------------
# depth post-processing (reference muggled_dpt/demo_helpers/postprocess.py, run_3dviewer.py:576-590)


def scale_prediction(prediction: torch.Tensor, target_wh: tuple[int, int]) -> torch.Tensor:
    """BxHxW -> Bx(h')x(w') bilinear, align_corners=False, no antialias (postprocess.py:22-29)."""
    return F.interpolate(prediction.unsqueeze(1), size=(int(target_wh[1]), int(target_wh[0])), mode="bilinear").squeeze(1)


def normalize_01(data: torch.Tensor) -> torch.Tensor:
    """(data - min) / (max - min) (postprocess.py:63-74)."""
    lo, hi = data.min(), data.max()
    return (data - lo) / (hi - lo)


def convert_to_uint8(depth: torch.Tensor) -> torch.Tensor:
    """(255 * normalize_01(depth)).byte(): truncation toward zero (postprocess.py:79-91)."""
    return (255.0 * normalize_01(depth)).byte()


def pack_depth_u24(depth: torch.Tensor, is_metric: bool = False, lossy: bool = False) -> torch.Tensor:
    """round(16777215 * normalize_01(depth)) split over B (low), G (mid), R (high) of a BGRA uint8 image; alpha is filled by the
    caller's mask (run_3dviewer.py:576-593, MAX_UINT24 :516)."""
    d = depth if is_metric else normalize_01(depth)
    q = torch.round(16777215 * d).to(torch.int32).squeeze()
    out = torch.zeros((*q.shape, 4), dtype=torch.uint8)
    out[..., 2] = ((q >> 16) & 255).to(torch.uint8)
    if not lossy:
        out[..., 1] = ((q >> 8) & 255).to(torch.uint8)
        out[..., 0] = (q & 255).to(torch.uint8)
    return out
