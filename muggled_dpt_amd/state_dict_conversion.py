"""Checkpoint -> (config, per-component weight dicts) for Depth-Anything-V2 DPT models.

This is bookkeeping only (no arithmetic): it restates the *contract* of the reference's
loader so original upstream ``depth_anything_v2_vit*.pth`` files load unchanged.

Reference behaviour followed here (paths relative to /root/reference/muggled_dpt):
  * config sniffing from tensor shapes ......... v2_depthanything/state_dict_conversion/config_from_original_state_dict.py:17-259
  * key renaming into 5 component dicts ........ v2_depthanything/state_dict_conversion/convert_original_state_dict_keys.py:15-86
  * pos_embed split into cls / patch parts ..... convert_original_state_dict_keys.py:295-317
  * dropped keys: mask_token, refinenet4.resConfUnit1 ... :131-132, :232-233

Unlike the reference (regex helpers walking every key), the mapping is generated from the
config as an explicit old-name -> (component, new-name) table, which also gives us the
complete list of expected tensors for strict loading.
"""

from __future__ import annotations

import math

COMPONENTS = ("patch_embed", "imgencoder", "reassemble", "fusion", "head")

_STAGE_NAMES = ("spatial_upx4", "spatial_upx2", "spatial_noscale", "spatial_downx2")


# ---------------------------------------------------------------------------------------------------------------------
# config


def get_model_config_from_state_dict(state_dict: dict, enable_cache: bool = False, enable_optimizations: bool = True,
                                     family: str = "v2") -> dict:
    """Infer the 11-key model config from an *original-format* DA-V2 state dict.

    Mirrors config_from_original_state_dict.py:17-43 (same keys, same derivations:
    heads = F // 64 (:90), grid = isqrt(len(pos_embed) - 1) (:230-240), etc.).
    Missing keys raise AssertionError like the reference's sniffers do.
    """

    def need(key: str):
        assert key in state_dict, f"Error reading model config! Couldn't find {key} key"
        return state_dict[key]

    patch_w = need("pretrained.patch_embed.proj.weight")
    features_per_token = int(patch_w.shape[0])
    patch_size_px = int(patch_w.shape[3])

    block_ids = [int(k.split(".")[2]) for k in state_dict if k.startswith("pretrained.blocks.")]
    assert len(block_ids) > 0 and max(block_ids) > 0, "Error determining number of transformer blocks!"
    num_blocks = 1 + max(block_ids)

    reassembly_features_list = [int(need(f"depth_head.scratch.layer{i}_rn.weight").shape[1]) for i in (1, 2, 3, 4)]
    fusion_channels = int(need("depth_head.scratch.layer1_rn.weight").shape[0])

    num_pos = int(need("pretrained.pos_embed").shape[1]) - 1
    base_grid = int(math.isqrt(num_pos))

    cfg = {
        "features_per_token": features_per_token,
        "num_blocks": num_blocks,
        "num_heads": features_per_token // 64,
        "reassembly_features_list": reassembly_features_list,
        "fusion_channels": fusion_channels,
        "patch_size_px": patch_size_px,
        "base_patch_grid_hw": (base_grid, base_grid),
    }
    if family == "v2":  # the Depth-Anything V1 sniffer returns 9 keys (v1_depthanything/.../config_from_original_state_dict.py:17-36)
        cfg["is_giant"] = "pretrained.blocks.0.mlp.w12.weight" in state_dict
        cfg["is_metric"] = "is_metric" in state_dict
    cfg["enable_cache"] = enable_cache
    cfg["enable_optimizations"] = enable_optimizations
    return cfg


# ---------------------------------------------------------------------------------------------------------------------
# key table


def original_to_new_key_table(config: dict, family: str = "v2") -> dict[str, tuple[str, str]]:
    """Explicit {original_key: (component, new_key)} table for a given config.

    New-format names are the reference's module attribute paths, e.g.
    ``imgencoder.stages.0.blocks.0.attn.qkv.weight``, ``reassemble.spatial_upx4.resample.1.weight``,
    ``fusion.blocks.0.conv_reassembly.resconv_seq.1.weight``, ``head.proj_1ch.2.weight``.
    ``pretrained.pos_embed`` is handled separately (split in two).
    """
    table: dict[str, tuple[str, str]] = {}
    wb = ("weight", "bias")

    for s in wb:
        table[f"pretrained.patch_embed.proj.{s}"] = ("patch_embed", f"proj.{s}")

    # encoder (convert_original_state_dict_keys.py:110-170). blocks_per_stage = num_blocks // 4 (:29)
    blocks_per_stage = config["num_blocks"] // 4
    table["pretrained.cls_token"] = ("imgencoder", "cls_token")
    for s in wb:
        table[f"pretrained.norm.{s}"] = ("imgencoder", f"outnorm.{s}")
    for i in range(config["num_blocks"]):
        old = f"pretrained.blocks.{i}"
        # V1 keeps a flat block list (v1_depthanything/.../convert_original_state_dict_keys.py:139-140)
        new = f"blocks.{i}" if family == "v1" else f"stages.{i // blocks_per_stage}.blocks.{i % blocks_per_stage}"
        for s in wb:
            table[f"{old}.norm1.{s}"] = ("imgencoder", f"{new}.norm1.{s}")
            table[f"{old}.norm2.{s}"] = ("imgencoder", f"{new}.norm2.{s}")
            table[f"{old}.attn.qkv.{s}"] = ("imgencoder", f"{new}.attn.qkv.{s}")
            table[f"{old}.attn.proj.{s}"] = ("imgencoder", f"{new}.attn.proj.{s}")
            if config.get("is_giant", False):
                table[f"{old}.mlp.w12.{s}"] = ("imgencoder", f"{new}.mlp.inner_linear_doubled.{s}")
                table[f"{old}.mlp.w3.{s}"] = ("imgencoder", f"{new}.mlp.outer_linear.{s}")
            else:
                table[f"{old}.mlp.fc1.{s}"] = ("imgencoder", f"{new}.mlp.layers.0.{s}")
                table[f"{old}.mlp.fc2.{s}"] = ("imgencoder", f"{new}.mlp.layers.2.{s}")
        table[f"{old}.ls1.gamma"] = ("imgencoder", f"{new}.scale_attn")
        table[f"{old}.ls2.gamma"] = ("imgencoder", f"{new}.scale_mlp")

    # reassembly (:175-215): projects.i -> resample.0 ; resize_layers.i -> resample.1 ; layer{i+1}_rn -> fuse_proj
    for i, name in enumerate(_STAGE_NAMES):
        for s in wb:
            table[f"depth_head.projects.{i}.{s}"] = ("reassemble", f"{name}.resample.0.{s}")
            if i != 2:  # stage 3 ("noscale") is an Identity upstream: no parameters
                table[f"depth_head.resize_layers.{i}.{s}"] = ("reassemble", f"{name}.resample.1.{s}")
        table[f"depth_head.scratch.layer{i + 1}_rn.weight"] = ("reassemble", f"{name}.fuse_proj.weight")

    # fusion (:220-275): refinenet{n} -> blocks.{n-1}
    for n in (1, 2, 3, 4):
        blk = f"blocks.{n - 1}"
        old = f"depth_head.scratch.refinenet{n}"
        for s in wb:
            table[f"{old}.out_conv.{s}"] = ("fusion", f"{blk}.scale_proj_seq.2.{s}")
            for conv, seq in (("conv1", "resconv_seq.1"), ("conv2", "resconv_seq.3")):
                table[f"{old}.resConfUnit2.{conv}.{s}"] = ("fusion", f"{blk}.scale_proj_seq.0.{seq}.{s}")
                if n != 4:  # top-most block has no reassembly RCU (dropped, :232-233)
                    table[f"{old}.resConfUnit1.{conv}.{s}"] = ("fusion", f"{blk}.conv_reassembly.{seq}.{s}")

    # head (:280-292)
    for s in wb:
        table[f"depth_head.scratch.output_conv1.{s}"] = ("head", f"spatial_upsampler.0.{s}")
        table[f"depth_head.scratch.output_conv2.0.{s}"] = ("head", f"proj_1ch.0.{s}")
        table[f"depth_head.scratch.output_conv2.2.{s}"] = ("head", f"proj_1ch.2.{s}")
    return table


_IGNORED_ORIGINAL_PREFIXES = (
    "pretrained.mask_token",
    "depth_head.scratch.refinenet4.resConfUnit1",
    "is_metric",
)


def convert_state_dict_keys(config: dict, original_state_dict: dict, family: str = "v2") -> dict[str, dict]:
    """Original upstream state dict -> {"patch_embed":{}, "imgencoder":{}, "reassemble":{}, "fusion":{}, "head":{}}.

    Same output contract as convert_original_state_dict_keys.py:15-86 (unknown keys are silently
    skipped there too; strictness is enforced later by load_state_dict).
    """
    table = original_to_new_key_table(config, family)
    out: dict[str, dict] = {name: {} for name in COMPONENTS}
    for key, data in original_state_dict.items():
        key = str(key)
        if key == "pretrained.pos_embed":
            # [1, 1+G*G, F] -> cls part + patch-grid part (:295-317)
            out["imgencoder"]["posenc.cls_embedding"] = data[:, :1, :]
            out["imgencoder"]["posenc.base_patch_embedding"] = data[:, 1:, :]
            continue
        hit = table.get(key)
        if hit is not None:
            out[hit[0]][hit[1]] = data
    return out


def expected_new_keys(config: dict, family: str = "v2") -> dict[str, list[str]]:
    """All new-format keys a complete model must have (used for strict loading)."""
    keys: dict[str, list[str]] = {name: [] for name in COMPONENTS}
    for comp, new_key in original_to_new_key_table(config, family).values():
        keys[comp].append(new_key)
    keys["imgencoder"] += ["posenc.cls_embedding", "posenc.base_patch_embedding"]
    return keys


def is_converted_state_dict(state_dict: dict) -> bool:
    """True if `state_dict` is already in the 5-component form."""
    return all(name in state_dict and isinstance(state_dict[name], dict) for name in COMPONENTS)


def flatten_components(component_dicts: dict[str, dict]) -> dict:
    """{"head": {"proj_1ch.2.bias": t}} -> {"head.proj_1ch.2.bias": t} (the names the C ABI binds by)."""
    flat = {}
    for comp in COMPONENTS:
        for k, v in component_dicts[comp].items():
            flat[f"{comp}.{k}"] = v
    return flat
