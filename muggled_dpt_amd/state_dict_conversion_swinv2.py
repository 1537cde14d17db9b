"""MiDaS v3.1 SwinV2 checkpoint -> (config, per-component weight dicts).

Restates the contract of the reference loader (paths relative to /root/reference/muggled_dpt/v31_swinv2/state_dict_conversion):
  * config sniffing ............ config_from_midas_state_dict.py:17-45 (heads from logit_scale :50-66, layers from block keys
                                 :71-89, window / base grid from the first attn_mask buffer [nW, wa, wa] :94-141,
                                 pretrained window sizes from a fixed table {16: [16,16,16,8], 24: [12,12,12,6]} :146-160)
  * key renaming ............... convert_midas_state_dict_keys.py:15-352 (cpb_mlp -> relpos_enc.bias_mlp, mlp.fc1/fc2 ->
                                 mlp.layers.0/2, layers.S.downsample -> patch_merge_layers.S, layer{i}_rn -> fuse_proj,
                                 attn_mask and refinenet4.resConfUnit1 dropped)
  * logit_scale ................ clamp(max=log(100)).exp() at load time (:115-131)
  * q_bias / v_bias reshape .... [F] -> [1, heads, 1, F/heads] (:137-161)
"""

from __future__ import annotations

import math

import torch

from .state_dict_conversion import COMPONENTS

_STAGE_NAMES = ("spatial_noscale", "spatial_downx2", "spatial_downx4", "spatial_downx8")
_PRETRAINED_WINDOW_LUT = {16: [16, 16, 16, 8], 24: [12, 12, 12, 6]}


def get_model_config_from_state_dict(state_dict: dict, enable_cache: bool = False, enable_optimizations: bool = True) -> dict:
    def need(key: str):
        assert key in state_dict, f"Error reading model config! Couldn't find {key} key"
        return state_dict[key]

    patch_w = need("pretrained.model.patch_embed.proj.weight")
    heads, layers = {}, {}
    for key in state_dict:
        parts = str(key).split(".")
        if key.startswith("pretrained.model.layers.") and len(parts) > 5 and parts[4] == "blocks":
            s, l = int(parts[3]), int(parts[5])
            layers[s] = max(layers.get(s, 0), l + 1)
            if key.endswith("logit_scale"):
                heads[s] = int(state_dict[key].shape[0])
    assert len(heads) == 4, f"Expecting 4 stages in swinv2 dpt, got: {len(heads)}"
    assert len(layers) == 4, f"Expecting 4 stages in swinv2 dpt, got: {len(layers)}"
    mask_keys = sorted(k for k in state_dict if str(k).endswith("attn_mask"))
    assert mask_keys, "Error, couldn't find attn_mask key, can't determine window size!"
    num_windows, window_area = (int(v) for v in state_dict[mask_keys[0]].shape[0:2])
    win = int(math.sqrt(window_area))
    grid = int(math.sqrt(num_windows * window_area))
    return {
        "features_per_stage": [int(patch_w.shape[0]) * (2 ** i) for i in range(4)],
        "heads_per_stage": [heads[s] for s in sorted(heads)],
        "layers_per_stage": [layers[s] for s in sorted(layers)],
        "base_patch_grid_hw": (grid, grid),
        "window_size_hw": (win, win),
        "pretrained_window_sizes_per_stage": list(_PRETRAINED_WINDOW_LUT.get(win, [None] * 4)),
        "fusion_channels": int(need("scratch.layer1_rn.weight").shape[0]),
        "patch_size_px": int(patch_w.shape[3]),
        "enable_cache": enable_cache,
        "enable_optimizations": enable_optimizations,
    }


def original_to_new_key_table(config: dict) -> dict[str, tuple[str, str]]:
    table: dict[str, tuple[str, str]] = {}
    wb = ("weight", "bias")
    for s in wb:
        table[f"pretrained.model.patch_embed.proj.{s}"] = ("patch_embed", f"proj.{s}")
        table[f"pretrained.model.patch_embed.norm.{s}"] = ("patch_embed", f"norm.{s}")
    for st in range(4):
        for l in range(config["layers_per_stage"][st]):
            old, new = f"pretrained.model.layers.{st}.blocks.{l}", f"stages.{st}.blocks.{l}"
            for name in ("attn.q_bias", "attn.v_bias", "attn.qkv.weight", "attn.logit_scale"):
                table[f"{old}.{name}"] = ("imgencoder", f"{new}.{name}")
            table[f"{old}.attn.cpb_mlp.0.weight"] = ("imgencoder", f"{new}.attn.relpos_enc.bias_mlp.0.weight")
            table[f"{old}.attn.cpb_mlp.0.bias"] = ("imgencoder", f"{new}.attn.relpos_enc.bias_mlp.0.bias")
            table[f"{old}.attn.cpb_mlp.2.weight"] = ("imgencoder", f"{new}.attn.relpos_enc.bias_mlp.2.weight")
            for s in wb:
                table[f"{old}.attn.proj.{s}"] = ("imgencoder", f"{new}.attn.proj.{s}")
                table[f"{old}.norm1.{s}"] = ("imgencoder", f"{new}.norm1.{s}")
                table[f"{old}.norm2.{s}"] = ("imgencoder", f"{new}.norm2.{s}")
                table[f"{old}.mlp.fc1.{s}"] = ("imgencoder", f"{new}.mlp.layers.0.{s}")
                table[f"{old}.mlp.fc2.{s}"] = ("imgencoder", f"{new}.mlp.layers.2.{s}")
        if st < 3:
            old, new = f"pretrained.model.layers.{st}.downsample", f"patch_merge_layers.{st}"
            table[f"{old}.reduction.weight"] = ("imgencoder", f"{new}.reduction.weight")
            for s in wb:
                table[f"{old}.norm.{s}"] = ("imgencoder", f"{new}.norm.{s}")
        table[f"scratch.layer{st + 1}_rn.weight"] = ("reassemble", f"{_STAGE_NAMES[st]}.fuse_proj.weight")
    for n in (1, 2, 3, 4):
        blk, old = f"blocks.{n - 1}", f"scratch.refinenet{n}"
        for s in wb:
            table[f"{old}.out_conv.{s}"] = ("fusion", f"{blk}.proj_seq.2.{s}")
            for conv, seq in (("conv1", "conv_seq.1"), ("conv2", "conv_seq.3")):
                table[f"{old}.resConfUnit2.{conv}.{s}"] = ("fusion", f"{blk}.proj_seq.0.{seq}.{s}")
                if n != 4:
                    table[f"{old}.resConfUnit1.{conv}.{s}"] = ("fusion", f"{blk}.conv_reassembly.{seq}.{s}")
    for s in wb:
        table[f"scratch.output_conv.0.{s}"] = ("head", f"spatial_upsampler.0.{s}")
        table[f"scratch.output_conv.2.{s}"] = ("head", f"proj_1ch.0.{s}")
        table[f"scratch.output_conv.4.{s}"] = ("head", f"proj_1ch.2.{s}")
    return table


def convert_state_dict_keys(config: dict, midas_state_dict: dict) -> dict[str, dict]:
    table = original_to_new_key_table(config)
    out: dict[str, dict] = {name: {} for name in COMPONENTS}
    for key, data in midas_state_dict.items():
        hit = table.get(str(key))
        if hit is None:
            continue  # attn_mask buffers, refinenet4.resConfUnit1.* (dropped by the reference too)
        comp, new = hit
        if new.endswith("logit_scale"):
            data = torch.clamp(data, max=math.log(1.0 / 0.01)).exp()
        elif new.endswith("q_bias") or new.endswith("v_bias"):
            stage = int(new.split(".")[1])
            data = data.reshape(1, config["heads_per_stage"][stage], 1, -1)
        out[comp][new] = data
    return out


def expected_new_keys(config: dict) -> dict[str, list[str]]:
    keys: dict[str, list[str]] = {name: [] for name in COMPONENTS}
    for comp, new_key in original_to_new_key_table(config).values():
        keys[comp].append(new_key)
    return keys
