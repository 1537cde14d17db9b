"""MiDaS v3.1 BEiT checkpoint -> (config, per-component weight dicts).

Restates the contract of the reference loader (paths relative to /root/reference/muggled_dpt/v31_beit/state_dict_conversion):
  * config sniffing ............ config_from_midas_state_dict.py:17-249 (heads from the bias-table width :70-84, base grid from
                                 its length L = (2g-1)^2 + 3 :215-249)
  * key renaming ............... convert_midas_state_dict_keys.py:15-328 (gamma_1/2 -> scale_attn/mlp, relative_position_bias_table
                                 -> relpos_enc.ref_bias_lut, act_postprocess{i}.{0.project.0|3|4} -> readout_proj.1 / resample.0 /
                                 resample.1, output_conv.{0,2,4} -> head layers; relative_position_index and refinenet4.resConfUnit1 dropped)
  * q_bias / v_bias reshape .... [F] -> [1, heads, 1, F/heads] (:137-161)
"""

from __future__ import annotations

import math

from .state_dict_conversion import COMPONENTS

_STAGE_NAMES = ("spatial_upx4", "spatial_upx2", "spatial_noscale", "spatial_downx2")


def get_model_config_from_state_dict(state_dict: dict, enable_cache: bool = False, enable_optimizations: bool = True) -> dict:
    def need(key: str):
        assert key in state_dict, f"Error reading model config! Couldn't find {key} key"
        return state_dict[key]

    patch_w = need("pretrained.model.patch_embed.proj.weight")
    table = need("pretrained.model.blocks.0.attn.relative_position_bias_table")
    block_ids = [int(k.split(".")[3]) for k in state_dict if k.startswith("pretrained.model.blocks.")]
    assert len(block_ids) > 0 and max(block_ids) > 0, "Error determining number of transformer blocks!"
    num_rel = int(table.shape[0]) - 3
    side = math.isqrt(num_rel)
    if side * side != num_rel or (side + 1) % 2:
        raise ValueError("Error calculating base grid size. Got non-integer results, base grid is not square?")
    grid = (side + 1) // 2
    return {
        "features_per_token": int(patch_w.shape[0]),
        "num_blocks": 1 + max(block_ids),
        "num_heads": int(table.shape[1]),
        "reassembly_features_list": [int(need(f"scratch.layer{i}_rn.weight").shape[1]) for i in (1, 2, 3, 4)],
        "fusion_channels": int(need("scratch.layer1_rn.weight").shape[0]),
        "patch_size_px": int(patch_w.shape[3]),
        "base_patch_grid_hw": (grid, grid),
        "enable_cache": enable_cache,
        "enable_optimizations": enable_optimizations,
    }


def original_to_new_key_table(config: dict) -> dict[str, tuple[str, str]]:
    table: dict[str, tuple[str, str]] = {}
    wb = ("weight", "bias")
    for s in wb:
        table[f"pretrained.model.patch_embed.proj.{s}"] = ("patch_embed", f"proj.{s}")
    table["pretrained.model.cls_token"] = ("imgencoder", "cls_token")
    bps = config["num_blocks"] // 4
    for i in range(config["num_blocks"]):
        old, new = f"pretrained.model.blocks.{i}", f"stages.{i // bps}.blocks.{i % bps}"
        for s in wb:
            table[f"{old}.norm1.{s}"] = ("imgencoder", f"{new}.norm1.{s}")
            table[f"{old}.norm2.{s}"] = ("imgencoder", f"{new}.norm2.{s}")
            table[f"{old}.attn.proj.{s}"] = ("imgencoder", f"{new}.attn.proj.{s}")
            table[f"{old}.mlp.fc1.{s}"] = ("imgencoder", f"{new}.mlp.layers.0.{s}")
            table[f"{old}.mlp.fc2.{s}"] = ("imgencoder", f"{new}.mlp.layers.2.{s}")
        table[f"{old}.attn.qkv.weight"] = ("imgencoder", f"{new}.attn.qkv.weight")
        table[f"{old}.attn.q_bias"] = ("imgencoder", f"{new}.attn.q_bias")
        table[f"{old}.attn.v_bias"] = ("imgencoder", f"{new}.attn.v_bias")
        table[f"{old}.attn.relative_position_bias_table"] = ("imgencoder", f"{new}.attn.relpos_enc.ref_bias_lut")
        table[f"{old}.gamma_1"] = ("imgencoder", f"{new}.scale_attn")
        table[f"{old}.gamma_2"] = ("imgencoder", f"{new}.scale_mlp")
    for i, name in enumerate(_STAGE_NAMES):
        for s in wb:
            table[f"pretrained.act_postprocess{i + 1}.0.project.0.{s}"] = ("reassemble", f"{name}.readout_proj.1.{s}")
            table[f"pretrained.act_postprocess{i + 1}.3.{s}"] = ("reassemble", f"{name}.resample.0.{s}")
            if i != 2:
                table[f"pretrained.act_postprocess{i + 1}.4.{s}"] = ("reassemble", f"{name}.resample.1.{s}")
        table[f"scratch.layer{i + 1}_rn.weight"] = ("reassemble", f"{name}.fuse_proj.weight")
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
    heads = config["num_heads"]
    for key, data in midas_state_dict.items():
        hit = table.get(str(key))
        if hit is None:
            continue  # relative_position_index, refinenet4.resConfUnit1.*, ... (dropped by the reference too)
        if hit[1].endswith("q_bias") or hit[1].endswith("v_bias"):
            data = data.reshape(1, heads, 1, -1)
        out[hit[0]][hit[1]] = data
    return out


def expected_new_keys(config: dict) -> dict[str, list[str]]:
    keys: dict[str, list[str]] = {name: [] for name in COMPONENTS}
    for comp, new_key in original_to_new_key_table(config).values():
        keys[comp].append(new_key)
    return keys
