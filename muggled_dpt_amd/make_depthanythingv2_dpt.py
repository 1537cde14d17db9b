"""Depth-Anything-V2 factory. Same names / kwargs / returns as the reference's
muggled_dpt/make_depthanythingv2_dpt.py (:24-61 and :67-138), building the libmdpt-backed DPTModel."""

from __future__ import annotations


from .dpt_model import DPTModel
from .state_dict_conversion import (COMPONENTS, convert_state_dict_keys, get_model_config_from_state_dict,
                                    is_converted_state_dict)


def make_depthanythingv2_dpt_from_original_state_dict(
    state_dict: dict,
    enable_cache: bool = False,
    enable_optimizations: bool = True,
    strict_load: bool = True,
) -> tuple[dict, DPTModel]:
    """Original upstream Depth-Anything-V2 checkpoint (or an already converted 5-component dict plus "config")
    -> (config_dict, DPTModel). Mirrors make_depthanythingv2_dpt.py:24-61."""
    if not strict_load:
        print("", "WARNING:", "  Loading model weights without 'strict' mode enabled!",
              "  Some weights may be missing or unused!", sep="\n", flush=True)
    if is_converted_state_dict(state_dict):
        if "config" not in state_dict:
            raise KeyError("a converted (5-component) state dict must carry its model config under the key 'config'")
        config_dict = dict(state_dict["config"])
        config_dict["enable_cache"], config_dict["enable_optimizations"] = enable_cache, enable_optimizations
        new_state_dict = state_dict
    else:
        config_dict = get_model_config_from_state_dict(state_dict, enable_cache, enable_optimizations)
        new_state_dict = convert_state_dict_keys(config_dict, state_dict)
    dpt_model = make_depthanythingv2_dpt(**config_dict)
    for comp in COMPONENTS:
        getattr(dpt_model, comp).load_state_dict(new_state_dict[comp], strict_load)
    return config_dict, dpt_model


def make_depthanythingv2_dpt(
    features_per_token: int,
    num_heads: int,
    num_blocks: int,
    reassembly_features_list: tuple[int, int, int, int],
    base_patch_grid_hw: tuple[int, int],
    fusion_channels: int = 256,
    patch_size_px: int = 14,
    is_giant: bool = False,
    is_metric: bool = False,
    enable_cache: bool = False,
    enable_optimizations: bool = True,
) -> DPTModel:
    """Build an (uninitialised) model from explicit sizes; see muggled_dpt_amd.synthetic.STANDARD_CONFIGS for the
    vit-small/base/large numbers (reference make_depthanythingv2_dpt.py:88-122)."""
    # enable_optimizations=False: every block grows an `attn.softmax` module; forward hooks on it receive the [B, heads, N, N]
    # attention weights (dumped by mdpt_encoder_probe), like the reference's non-optimised Attention (transformer_block.py:101)
    # enable_cache (reference position_encoder.py:152-227 GridCache): the resized position embedding and the zero pads of the operand planes stay in
    # the workspace between forwards of the same shape (mdpt_set_grid_cache) instead of being recomputed by ~20 us of small launches per call
    config = {
        "features_per_token": int(features_per_token),
        "num_heads": int(num_heads),
        "num_blocks": int(num_blocks),
        "reassembly_features_list": [int(v) for v in reassembly_features_list],
        "base_patch_grid_hw": tuple(int(v) for v in base_patch_grid_hw),
        "fusion_channels": int(fusion_channels),
        "patch_size_px": int(patch_size_px),
        "is_giant": bool(is_giant),
        "is_metric": bool(is_metric),
        "enable_cache": bool(enable_cache),
        "enable_optimizations": bool(enable_optimizations),
    }
    return DPTModel(config)
