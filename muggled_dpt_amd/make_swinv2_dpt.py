"""MiDaS v3.1 SwinV2 factory. Same names / kwargs / returns as the reference's muggled_dpt/make_swinv2_dpt.py (:24-61, :67-125)."""

from __future__ import annotations

from .dpt_model import DPTModel
from .state_dict_conversion import COMPONENTS
from .state_dict_conversion_swinv2 import convert_state_dict_keys, get_model_config_from_state_dict


def make_swinv2_dpt_from_midas_v31_state_dict(
    midas_v31_state_dict: dict,
    enable_cache: bool = False,
    enable_optimizations: bool = True,
    strict_load: bool = True,
) -> tuple[dict, DPTModel]:
    if not strict_load:
        print("", "WARNING:", "  Loading model weights without 'strict' mode enabled!",
              "  Some weights may be missing or unused!", sep="\n", flush=True)
    config_dict = get_model_config_from_state_dict(midas_v31_state_dict, enable_cache, enable_optimizations)
    new_state_dict = convert_state_dict_keys(config_dict, midas_v31_state_dict)
    dpt_model = make_swinv2_dpt(**config_dict)
    for comp in COMPONENTS:
        getattr(dpt_model, comp).load_state_dict(new_state_dict[comp], strict_load)
    return config_dict, dpt_model


def make_swinv2_dpt(
    features_per_stage: tuple[int, int, int, int],
    heads_per_stage: tuple[int, int, int, int],
    layers_per_stage: tuple[int, int, int, int],
    base_patch_grid_hw: tuple[int, int],
    window_size_hw: tuple[int, int],
    pretrained_window_sizes_per_stage: tuple,
    fusion_channels: int = 256,
    patch_size_px: int = 4,
    enable_cache: bool = True,
    **unused_kwargs,
) -> DPTModel:
    """Standard sizes: muggled_dpt_amd.synthetic.SWINV2_CONFIGS (reference make_swinv2_dpt.py:87-118). `enable_cache` (default True
    like the reference's) keeps the continuous-position-bias tables of all blocks - 16 sigmoid(MLP(log-coords)) per head, weights and
    window geometry only - in the workspace between forwards of the same shape (mdpt_set_grid_cache: one 165 us launch per grid instead
    of per forward); no [heads, wa, wa] bias or [nW, wa, wa] mask tensor is ever materialised."""
    config = {
        "features_per_stage": [int(v) for v in features_per_stage],
        "heads_per_stage": [int(v) for v in heads_per_stage],
        "layers_per_stage": [int(v) for v in layers_per_stage],
        "base_patch_grid_hw": tuple(int(v) for v in base_patch_grid_hw),
        "window_size_hw": tuple(int(v) for v in window_size_hw),
        "pretrained_window_sizes_per_stage": [None if v is None else int(v) for v in pretrained_window_sizes_per_stage],
        "fusion_channels": int(fusion_channels),
        "patch_size_px": int(patch_size_px),
        "enable_cache": bool(enable_cache),
    }
    return DPTModel(config, family="swinv2")
