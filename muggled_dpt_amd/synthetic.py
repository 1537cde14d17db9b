"""Seeded random-init Depth-Anything-V2 checkpoints in the *original upstream key format*.

No real weights ship with the reference (model_weights/ holds only a README) and there is no
network, so benchmarks, smoke tests and golden fixtures all run on synthetic weights. Plain
PyTorch default init gives an all-zero depth map (the final ReLU, reference
v2_depthanything/head_model.py:84, clips everything), so the recipe below is
tuned to produce non-degenerate outputs. It is pure CPU torch RNG => reproducible anywhere.
"""

from __future__ import annotations

import torch

STANDARD_CONFIGS = {
    # reference make_depthanythingv2_dpt.py:88-122
    "vits": dict(features_per_token=384, num_heads=6, num_blocks=12, reassembly_features_list=[48, 96, 192, 384],
                 base_patch_grid_hw=(37, 37), fusion_channels=64, patch_size_px=14),
    "vitb": dict(features_per_token=768, num_heads=12, num_blocks=12, reassembly_features_list=[96, 192, 384, 768],
                 base_patch_grid_hw=(37, 37), fusion_channels=128, patch_size_px=14),
    "vitl": dict(features_per_token=1024, num_heads=16, num_blocks=24, reassembly_features_list=[256, 512, 1024, 1024],
                 base_patch_grid_hw=(37, 37), fusion_channels=256, patch_size_px=14),
    "vitg": dict(features_per_token=1536, num_heads=24, num_blocks=40, reassembly_features_list=[1536, 1536, 1536, 1536],
                 base_patch_grid_hw=(37, 37), fusion_channels=384, patch_size_px=14, is_giant=True),
    # not a real model: small enough for full-tensor golden fixtures (KBs) and fast CPU tests
    "tiny_giant": dict(features_per_token=128, num_heads=2, num_blocks=4, reassembly_features_list=[16, 32, 64, 64],
                       base_patch_grid_hw=(5, 5), fusion_channels=32, patch_size_px=14, is_giant=True),  # SwiGLU hidden 344 (not /64)
    "tiny": dict(features_per_token=64, num_heads=1, num_blocks=4, reassembly_features_list=[16, 32, 64, 64],
                 base_patch_grid_hw=(5, 5), fusion_channels=32, patch_size_px=14),
}


BEIT_CONFIGS = {
    # reference make_beit_dpt.py:86-113
    "beit_large_384": dict(features_per_token=1024, num_heads=16, num_blocks=24, reassembly_features_list=[256, 512, 1024, 1024],
                           base_patch_grid_hw=(24, 24), fusion_channels=256, patch_size_px=16),
    "beit_large_512": dict(features_per_token=1024, num_heads=16, num_blocks=24, reassembly_features_list=[256, 512, 1024, 1024],
                           base_patch_grid_hw=(32, 32), fusion_channels=256, patch_size_px=16),
    "beit_base_384": dict(features_per_token=768, num_heads=12, num_blocks=12, reassembly_features_list=[96, 192, 384, 768],
                          base_patch_grid_hw=(24, 24), fusion_channels=256, patch_size_px=16),
    "beit_tiny": dict(features_per_token=128, num_heads=2, num_blocks=4, reassembly_features_list=[16, 32, 64, 64],
                      base_patch_grid_hw=(4, 4), fusion_channels=32, patch_size_px=16),
}


def beit_original_state_dict_shapes(cfg: dict) -> dict[str, tuple]:
    """Every tensor of a MiDaS v3.1 BEiT DPT checkpoint (the keys the reference's converter consumes), with its shape."""
    F, P, C, H = cfg["features_per_token"], cfg["patch_size_px"], cfg["fusion_channels"], cfg["num_heads"]
    gh, gw = cfg["base_patch_grid_hw"]
    hid = cfg["reassembly_features_list"]
    nlut = (2 * gh - 1) * (2 * gw - 1) + 3
    s: dict[str, tuple] = {}
    s["pretrained.model.cls_token"] = (1, 1, F)
    s["pretrained.model.patch_embed.proj.weight"] = (F, 3, P, P)
    s["pretrained.model.patch_embed.proj.bias"] = (F,)
    for i in range(cfg["num_blocks"]):
        b = f"pretrained.model.blocks.{i}"
        s[f"{b}.gamma_1"] = (F,)
        s[f"{b}.gamma_2"] = (F,)
        s[f"{b}.norm1.weight"] = (F,)
        s[f"{b}.norm1.bias"] = (F,)
        s[f"{b}.attn.q_bias"] = (F,)
        s[f"{b}.attn.v_bias"] = (F,)
        s[f"{b}.attn.relative_position_bias_table"] = (nlut, H)
        s[f"{b}.attn.qkv.weight"] = (3 * F, F)
        s[f"{b}.attn.proj.weight"] = (F, F)
        s[f"{b}.attn.proj.bias"] = (F,)
        s[f"{b}.norm2.weight"] = (F,)
        s[f"{b}.norm2.bias"] = (F,)
        s[f"{b}.mlp.fc1.weight"] = (4 * F, F)
        s[f"{b}.mlp.fc1.bias"] = (4 * F,)
        s[f"{b}.mlp.fc2.weight"] = (F, 4 * F)
        s[f"{b}.mlp.fc2.bias"] = (F,)
    for i in range(4):
        a = f"pretrained.act_postprocess{i + 1}"
        s[f"{a}.0.project.0.weight"] = (F, 2 * F)
        s[f"{a}.0.project.0.bias"] = (F,)
        s[f"{a}.3.weight"] = (hid[i], F, 1, 1)
        s[f"{a}.3.bias"] = (hid[i],)
    s["pretrained.act_postprocess1.4.weight"] = (hid[0], hid[0], 4, 4)
    s["pretrained.act_postprocess1.4.bias"] = (hid[0],)
    s["pretrained.act_postprocess2.4.weight"] = (hid[1], hid[1], 2, 2)
    s["pretrained.act_postprocess2.4.bias"] = (hid[1],)
    s["pretrained.act_postprocess4.4.weight"] = (hid[3], hid[3], 3, 3)
    s["pretrained.act_postprocess4.4.bias"] = (hid[3],)
    for i in range(4):
        s[f"scratch.layer{i + 1}_rn.weight"] = (C, hid[i], 3, 3)
    for n in (1, 2, 3, 4):
        r = f"scratch.refinenet{n}"
        s[f"{r}.out_conv.weight"] = (C, C, 1, 1)
        s[f"{r}.out_conv.bias"] = (C,)
        for unit in ("resConfUnit1", "resConfUnit2"):
            for conv in ("conv1", "conv2"):
                s[f"{r}.{unit}.{conv}.weight"] = (C, C, 3, 3)
                s[f"{r}.{unit}.{conv}.bias"] = (C,)
    s["scratch.output_conv.0.weight"] = (C // 2, C, 3, 3)
    s["scratch.output_conv.0.bias"] = (C // 2,)
    s["scratch.output_conv.2.weight"] = (32, C // 2, 3, 3)
    s["scratch.output_conv.2.bias"] = (32,)
    s["scratch.output_conv.4.weight"] = (1, 32, 1, 1)
    s["scratch.output_conv.4.bias"] = (1,)
    return s


def make_synthetic_beit_state_dict(cfg: dict | str, seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded fp32 MiDaS-format BEiT checkpoint (every parameter explicitly initialised: the reference creates several
    with torch.empty). Same recipe as the Depth-Anything one; relative-position tables ~ 0.5 N."""
    if isinstance(cfg, str):
        cfg = BEIT_CONFIGS[cfg]
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    sd: dict[str, torch.Tensor] = {}
    for key, shape in beit_original_state_dict_shapes(cfg).items():
        if "gamma_" in key:
            t = 0.5 + 0.5 * torch.rand(shape, generator=gen)
        elif ".norm" in key and key.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen)
        elif key.endswith("relative_position_bias_table"):
            t = 0.5 * torch.randn(shape, generator=gen)
        elif key == "scratch.output_conv.4.bias":
            t = torch.full(shape, 0.5)
        elif len(shape) >= 2 and key.endswith("weight"):
            if key.endswith("act_postprocess1.4.weight") or key.endswith("act_postprocess2.4.weight"):
                fan_in = shape[0]
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
            t = torch.randn(shape, generator=gen) * (float(fan_in) ** -0.5)
        elif key == "pretrained.model.cls_token":
            t = 0.5 * torch.randn(shape, generator=gen)
        else:
            t = 0.1 * torch.randn(shape, generator=gen)
        sd[key] = t.to(torch.float32).contiguous()
    return sd


SWINV2_CONFIGS = {
    # reference make_swinv2_dpt.py:87-118
    "swin2_large_384": dict(features_per_stage=[192, 384, 768, 1536], heads_per_stage=[6, 12, 24, 48], layers_per_stage=[2, 2, 18, 2],
                            base_patch_grid_hw=(96, 96), window_size_hw=(24, 24), pretrained_window_sizes_per_stage=[12, 12, 12, 6],
                            fusion_channels=256, patch_size_px=4),
    "swin2_base_384": dict(features_per_stage=[128, 256, 512, 1024], heads_per_stage=[4, 8, 16, 32], layers_per_stage=[2, 2, 18, 2],
                           base_patch_grid_hw=(96, 96), window_size_hw=(24, 24), pretrained_window_sizes_per_stage=[12, 12, 12, 6],
                           fusion_channels=256, patch_size_px=4),
    # the smallest MiDaS v3.1 SwinV2: a 96-wide first stage (3 heads of 32) - operand planes padded to 128 columns on the device
    "swin2_tiny_256": dict(features_per_stage=[96, 192, 384, 768], heads_per_stage=[3, 6, 12, 24], layers_per_stage=[2, 2, 6, 2],
                           base_patch_grid_hw=(64, 64), window_size_hw=(16, 16), pretrained_window_sizes_per_stage=[16, 16, 16, 8],
                           fusion_channels=256, patch_size_px=4),
    # not a real model: 64x64 px base image (grid 16 -> 8 -> 4 -> 2), window 4: stages 0/1 shift, stage 2 is one window,
    # stage 3 shrinks the window to 2x2
    "swin2_tiny": dict(features_per_stage=[64, 128, 256, 512], heads_per_stage=[2, 4, 8, 16], layers_per_stage=[2, 2, 4, 2],
                       base_patch_grid_hw=(16, 16), window_size_hw=(4, 4), pretrained_window_sizes_per_stage=[None] * 4,
                       fusion_channels=32, patch_size_px=4),
}


def swinv2_original_state_dict_shapes(cfg: dict) -> dict[str, tuple]:
    """Every tensor of a MiDaS v3.1 SwinV2 DPT checkpoint that the reference's converter consumes (+ the attn_mask buffers its
    config sniffing needs), with its shape."""
    feats, heads, layers = cfg["features_per_stage"], cfg["heads_per_stage"], cfg["layers_per_stage"]
    P, C = cfg["patch_size_px"], cfg["fusion_channels"]
    gh, gw = cfg["base_patch_grid_hw"]
    wh, ww = cfg["window_size_hw"]
    s: dict[str, tuple] = {}
    s["pretrained.model.patch_embed.proj.weight"] = (feats[0], 3, P, P)
    s["pretrained.model.patch_embed.proj.bias"] = (feats[0],)
    s["pretrained.model.patch_embed.norm.weight"] = (feats[0],)
    s["pretrained.model.patch_embed.norm.bias"] = (feats[0],)
    for st in range(4):
        F, H = feats[st], heads[st]
        for l in range(layers[st]):
            b = f"pretrained.model.layers.{st}.blocks.{l}"
            if st == 0 and l == 1:  # timm registers the shift mask as a buffer: [num_windows, window_area, window_area]
                s[f"{b}.attn_mask"] = ((gh // wh) * (gw // ww), wh * ww, wh * ww)
            s[f"{b}.attn.logit_scale"] = (H, 1, 1)
            s[f"{b}.attn.q_bias"] = (F,)
            s[f"{b}.attn.v_bias"] = (F,)
            s[f"{b}.attn.cpb_mlp.0.weight"] = (512, 2)
            s[f"{b}.attn.cpb_mlp.0.bias"] = (512,)
            s[f"{b}.attn.cpb_mlp.2.weight"] = (H, 512)
            s[f"{b}.attn.qkv.weight"] = (3 * F, F)
            s[f"{b}.attn.proj.weight"] = (F, F)
            s[f"{b}.attn.proj.bias"] = (F,)
            s[f"{b}.norm1.weight"] = (F,)
            s[f"{b}.norm1.bias"] = (F,)
            s[f"{b}.mlp.fc1.weight"] = (4 * F, F)
            s[f"{b}.mlp.fc1.bias"] = (4 * F,)
            s[f"{b}.mlp.fc2.weight"] = (F, 4 * F)
            s[f"{b}.mlp.fc2.bias"] = (F,)
            s[f"{b}.norm2.weight"] = (F,)
            s[f"{b}.norm2.bias"] = (F,)
        if st < 3:
            d = f"pretrained.model.layers.{st}.downsample"
            s[f"{d}.reduction.weight"] = (feats[st + 1], 4 * F)
            s[f"{d}.norm.weight"] = (feats[st + 1],)
            s[f"{d}.norm.bias"] = (feats[st + 1],)
    for i in range(4):
        s[f"scratch.layer{i + 1}_rn.weight"] = (C, feats[i], 3, 3)
    for n in (1, 2, 3, 4):
        r = f"scratch.refinenet{n}"
        s[f"{r}.out_conv.weight"] = (C, C, 1, 1)
        s[f"{r}.out_conv.bias"] = (C,)
        for unit in ("resConfUnit1", "resConfUnit2"):
            for conv in ("conv1", "conv2"):
                s[f"{r}.{unit}.{conv}.weight"] = (C, C, 3, 3)
                s[f"{r}.{unit}.{conv}.bias"] = (C,)
    s["scratch.output_conv.0.weight"] = (C // 2, C, 3, 3)
    s["scratch.output_conv.0.bias"] = (C // 2,)
    s["scratch.output_conv.2.weight"] = (32, C // 2, 3, 3)
    s["scratch.output_conv.2.bias"] = (32,)
    s["scratch.output_conv.4.weight"] = (1, 32, 1, 1)
    s["scratch.output_conv.4.bias"] = (1,)
    return s


def make_synthetic_swinv2_state_dict(cfg: dict | str, seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded fp32 MiDaS-format SwinV2 checkpoint. logit_scale is stored pre-exp as in timm (log of 5..15; the loader clamps at
    log(100) and exponentiates); the post-norm LayerNorm weights are ~0.3 so the residual stream stays O(1) over 24 blocks."""
    if isinstance(cfg, str):
        cfg = SWINV2_CONFIGS[cfg]
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    sd: dict[str, torch.Tensor] = {}
    for key, shape in swinv2_original_state_dict_shapes(cfg).items():
        if key.endswith("attn_mask"):
            t = torch.zeros(shape)
        elif key.endswith("logit_scale"):
            t = torch.log(5.0 + 10.0 * torch.rand(shape, generator=gen))
        elif ".blocks." in key and ".norm" in key and key.endswith("weight"):
            t = 0.3 + 0.05 * torch.randn(shape, generator=gen)
        elif ".norm" in key and key.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen)
        elif key.endswith("cpb_mlp.0.weight"):
            t = torch.randn(shape, generator=gen)
        elif key.endswith("cpb_mlp.2.weight"):
            t = 0.1 * torch.randn(shape, generator=gen)
        elif key == "scratch.output_conv.4.bias":
            t = torch.full(shape, 0.5)
        elif len(shape) >= 2 and key.endswith("weight"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=gen) * (float(fan_in) ** -0.5)
        else:
            t = 0.1 * torch.randn(shape, generator=gen)
        sd[key] = t.to(torch.float32).contiguous()
    return sd


def swiglu_hidden(features_per_token: int, ratio: float = 4) -> int:
    """Hidden width of the ViT-G SwiGLU FFN: 2/3 of the MLP width rounded up to 8 (components/misc_helpers.py:164-165)."""
    return 8 * ((int(int(ratio * features_per_token) * 2 / 3) + 7) // 8)


def original_state_dict_shapes(cfg: dict) -> dict[str, tuple]:
    """Every tensor of an upstream DA-V2 (non-giant) checkpoint, in upstream order, with its shape."""
    F = cfg["features_per_token"]
    P = cfg["patch_size_px"]
    gh, gw = cfg["base_patch_grid_hw"]
    C = cfg["fusion_channels"]
    hid = cfg["reassembly_features_list"]
    shapes: dict[str, tuple] = {}
    shapes["pretrained.cls_token"] = (1, 1, F)
    shapes["pretrained.pos_embed"] = (1, 1 + gh * gw, F)
    shapes["pretrained.mask_token"] = (1, F)
    shapes["pretrained.patch_embed.proj.weight"] = (F, 3, P, P)
    shapes["pretrained.patch_embed.proj.bias"] = (F,)
    for i in range(cfg["num_blocks"]):
        b = f"pretrained.blocks.{i}"
        shapes[f"{b}.norm1.weight"] = (F,)
        shapes[f"{b}.norm1.bias"] = (F,)
        shapes[f"{b}.attn.qkv.weight"] = (3 * F, F)
        shapes[f"{b}.attn.qkv.bias"] = (3 * F,)
        shapes[f"{b}.attn.proj.weight"] = (F, F)
        shapes[f"{b}.attn.proj.bias"] = (F,)
        shapes[f"{b}.ls1.gamma"] = (F,)
        shapes[f"{b}.norm2.weight"] = (F,)
        shapes[f"{b}.norm2.bias"] = (F,)
        if cfg.get("is_giant", False):  # SwiGLU FFN (reference components/misc_helpers.py:162-168)
            shid = swiglu_hidden(F)
            shapes[f"{b}.mlp.w12.weight"] = (2 * shid, F)
            shapes[f"{b}.mlp.w12.bias"] = (2 * shid,)
            shapes[f"{b}.mlp.w3.weight"] = (F, shid)
            shapes[f"{b}.mlp.w3.bias"] = (F,)
        else:
            shapes[f"{b}.mlp.fc1.weight"] = (4 * F, F)
            shapes[f"{b}.mlp.fc1.bias"] = (4 * F,)
            shapes[f"{b}.mlp.fc2.weight"] = (F, 4 * F)
            shapes[f"{b}.mlp.fc2.bias"] = (F,)
        shapes[f"{b}.ls2.gamma"] = (F,)
    shapes["pretrained.norm.weight"] = (F,)
    shapes["pretrained.norm.bias"] = (F,)
    for i in range(4):
        shapes[f"depth_head.projects.{i}.weight"] = (hid[i], F, 1, 1)
        shapes[f"depth_head.projects.{i}.bias"] = (hid[i],)
    shapes["depth_head.resize_layers.0.weight"] = (hid[0], hid[0], 4, 4)  # ConvTranspose2d k4 s4: (Cin, Cout, kh, kw)
    shapes["depth_head.resize_layers.0.bias"] = (hid[0],)
    shapes["depth_head.resize_layers.1.weight"] = (hid[1], hid[1], 2, 2)  # ConvTranspose2d k2 s2
    shapes["depth_head.resize_layers.1.bias"] = (hid[1],)
    shapes["depth_head.resize_layers.3.weight"] = (hid[3], hid[3], 3, 3)  # Conv2d k3 s2 p1
    shapes["depth_head.resize_layers.3.bias"] = (hid[3],)
    for i in range(4):
        shapes[f"depth_head.scratch.layer{i + 1}_rn.weight"] = (C, hid[i], 3, 3)
    for n in (1, 2, 3, 4):
        r = f"depth_head.scratch.refinenet{n}"
        shapes[f"{r}.out_conv.weight"] = (C, C, 1, 1)
        shapes[f"{r}.out_conv.bias"] = (C,)
        for unit in ("resConfUnit1", "resConfUnit2"):
            for conv in ("conv1", "conv2"):
                shapes[f"{r}.{unit}.{conv}.weight"] = (C, C, 3, 3)
                shapes[f"{r}.{unit}.{conv}.bias"] = (C,)
    shapes["depth_head.scratch.output_conv1.weight"] = (C // 2, C, 3, 3)
    shapes["depth_head.scratch.output_conv1.bias"] = (C // 2,)
    shapes["depth_head.scratch.output_conv2.0.weight"] = (32, C // 2, 3, 3)
    shapes["depth_head.scratch.output_conv2.0.bias"] = (32,)
    shapes["depth_head.scratch.output_conv2.2.weight"] = (1, 32, 1, 1)
    shapes["depth_head.scratch.output_conv2.2.bias"] = (1,)
    return shapes


def make_synthetic_original_state_dict(cfg: dict | str, seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded fp32 CPU checkpoint with upstream key names (loads through the normal factory path).

    Recipe: matrices/kernels ~ N(0, 1/fan_in); norm weights 1 + 0.1 N; layer-scale gammas ~ U(0.5, 1);
    other vectors 0.1 N; position embedding 0.5 N; final 1-channel bias 0.5 (keeps the ReLU alive).
    Tensors are drawn in `original_state_dict_shapes` order from one generator, so (cfg, seed)
    fully determines the checkpoint.
    """
    if isinstance(cfg, str):
        cfg = STANDARD_CONFIGS[cfg]
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    sd: dict[str, torch.Tensor] = {}
    for key, shape in original_state_dict_shapes(cfg).items():
        if key.endswith("gamma"):
            t = 0.5 + 0.5 * torch.rand(shape, generator=gen)
        elif ".norm" in key and key.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=gen)
        elif key == "pretrained.pos_embed":
            t = 0.5 * torch.randn(shape, generator=gen)
        elif key == "depth_head.scratch.output_conv2.2.bias":
            t = torch.full(shape, 0.5)
        elif len(shape) >= 2 and key.endswith("weight"):
            if "resize_layers.0" in key or "resize_layers.1" in key:
                fan_in = shape[0]  # transposed conv with k == s: every output pixel sees Cin inputs
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
            t = torch.randn(shape, generator=gen) * (float(fan_in) ** -0.5)
        else:
            t = 0.1 * torch.randn(shape, generator=gen)
        sd[key] = t.to(torch.float32).contiguous()
    return sd


def realistic_statistics(osd: dict[str, torch.Tensor], seed: int = 0, gamma_lo: float = 1e-4, gamma_hi: float = 1.0) -> dict[str, torch.Tensor]:
    """A synthetic Depth-Anything checkpoint (upstream key names) perturbed toward the statistics real DINOv2 weights are known for - the stand-in
    for the real file that cannot be fetched here (tools/probes/gpu_realistic_stats_check.py, tests/test_gpu_precision_modes.py):
    per-channel log-uniform layer-scale gammas (gamma_lo ... gamma_hi: 1e-4 ... 1 by default, 2 % at 1e-5), log-normal LayerNorm weights with a few x8 / x0.05 channels, encoder
    Linears with row scales exp(N(0, 0.3)) and one entry in 1000 six times larger, and two "massive activation" channels of the residual stream
    (fc2 bias of block 4: +150 / -90; position embedding: +40 / -25)."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1000 + int(seed))
    out = {}
    for k, v in osd.items():
        t = v.clone()
        enc = k.startswith("pretrained.blocks.")
        if enc and k.endswith("gamma"):
            import math
            t = gamma_hi * torch.exp(torch.rand(t.shape, generator=gen) * math.log(gamma_lo / gamma_hi))  # log-uniform gamma_lo ... gamma_hi
            t[torch.rand(t.shape, generator=gen) < 0.02] = 1e-5
        elif enc and ".norm" in k and k.endswith("weight"):
            t = torch.exp(0.5 * torch.randn(t.shape, generator=gen))
            r = torch.rand(t.shape, generator=gen)
            t[r < 0.005] *= 8.0
            t[r > 0.995] *= 0.05
        elif enc and k.endswith("weight") and t.dim() == 2:
            t = t * torch.exp(0.3 * torch.randn(t.shape[0], 1, generator=gen))
            t = t * (1.0 + 5.0 * (torch.rand(t.shape, generator=gen) < 1e-3).float())
        elif k == "pretrained.blocks.4.mlp.fc2.bias":
            t[7] += 150.0
            t[t.numel() // 2 + 9] -= 90.0
        elif k == "pretrained.pos_embed":
            t[:, 1:, 7] += 40.0
            t[:, 1:, t.shape[-1] // 2 + 9] -= 25.0
        out[k] = t.to(torch.float32).contiguous()
    g4 = out.get("pretrained.blocks.4.ls2.gamma")
    if g4 is not None:  # the bias reaches the stream through the layer scale: those two channels pass it on unscaled
        g4[7] = 1.0
        g4[g4.numel() // 2 + 9] = 1.0
    return out
