"""MiDaS v3.1 BEiT factory. Same names / kwargs / returns as the reference's muggled_dpt/make_beit_dpt.py (:24-61, :67-125)."""

from __future__ import annotations

from .dpt_model import DPTModel
from .state_dict_conversion import COMPONENTS
from .state_dict_conversion_beit import convert_state_dict_keys, get_model_config_from_state_dict


def make_beit_dpt_from_midas_v31_state_dict(
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
    dpt_model = make_beit_dpt(**config_dict)
    for comp in COMPONENTS:
        getattr(dpt_model, comp).load_state_dict(new_state_dict[comp], strict_load)
    return config_dict, dpt_model


def make_beit_dpt(
    features_per_token: int,
    num_heads: int,
    num_blocks: int,
    reassembly_features_list: tuple[int, int, int, int],
    base_patch_grid_hw: tuple[int, int],
    fusion_channels: int = 256,
    patch_size_px: int = 16,
    enable_cache: bool = False,
    enable_optimizations: bool = True,
    **unused_kwargs,
) -> DPTModel:
    """Standard sizes: muggled_dpt_amd.synthetic.BEIT_CONFIGS (reference make_beit_dpt.py:86-113). `enable_cache=True` (what the
    reference's video demo asks for, run_video.py:144; v31_beit/components/README.md:91) keeps the relative-position tables of all blocks,
    resized to the current grid, in the workspace between forwards of the same shape (mdpt_set_grid_cache); there is no [heads, N, N] bias
    tensor - the tables are per-head LUTs that live in LDS during attention."""
    config = {
        "features_per_token": int(features_per_token),
        "num_heads": int(num_heads),
        "num_blocks": int(num_blocks),
        "reassembly_features_list": [int(v) for v in reassembly_features_list],
        "base_patch_grid_hw": tuple(int(v) for v in base_patch_grid_hw),
        "fusion_channels": int(fusion_channels),
        "patch_size_px": int(patch_size_px),
        "enable_cache": bool(enable_cache),
        "enable_optimizations": bool(enable_optimizations),  # False: hookable attn.softmax modules (see DPTModel.__init__)
    }
    if int(features_per_token) != 64 * int(num_heads):
        raise NotImplementedError("the MI355X attention kernel supports head dim 64 only")
    return DPTModel(config, family="beit")
