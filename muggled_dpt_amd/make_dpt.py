"""Generic checkpoint -> model entry point; same signature/returns as the reference's muggled_dpt/make_dpt.py:21-72."""

from __future__ import annotations

import os.path as osp

import torch


def make_dpt_from_state_dict(
    path_to_state_dict: str,
    enable_cache: bool = False,
    enable_optimizations: bool = True,
    strict_load: bool = True,
    model_type: str | None = None,
) -> tuple[dict, torch.nn.Module]:
    """Load a checkpoint file, work out which model family it is and build the model.
    Returns (model_config_dict, DPTModel). Unknown / not-yet-supported families raise NotImplementedError
    (reference make_dpt.py:48-51)."""
    try:
        state_dict = torch.load(path_to_state_dict)
    except RuntimeError:
        state_dict = torch.load(path_to_state_dict, map_location="cpu")  # make_dpt.py:38-41

    if model_type is None:
        model_type = determine_model_type_from_state_dict(path_to_state_dict, state_dict)
    known_model_types = ["swinv2", "beit", "depthanythingv1", "depthanythingv2"]
    if model_type not in known_model_types:
        print("Accepted model types:", *known_model_types, sep="\n")
        raise NotImplementedError(f"Bad model type: {model_type}, no support for this yet!")
    if model_type == "swinv2":
        from .make_swinv2_dpt import make_swinv2_dpt_from_midas_v31_state_dict as make_swin

        return make_swin(state_dict, enable_cache, enable_optimizations, strict_load)

    if model_type == "beit":
        from .make_beit_dpt import make_beit_dpt_from_midas_v31_state_dict as make_beit

        return make_beit(state_dict, enable_cache, enable_optimizations, strict_load)

    if model_type == "depthanythingv1":
        from .make_depthanythingv1_dpt import make_depthanythingv1_dpt_from_original_state_dict as make_v1

        return make_v1(state_dict, enable_cache, enable_optimizations, strict_load)

    if "metric" in path_to_state_dict:  # make_dpt.py:56-66 (file-name based metric-head switch)
        state_dict["is_metric"] = torch.tensor((1), dtype=torch.float32)
        print("", "Warning: Metric Depth-Anything V2 model detected!", "  These models are not officially supported,",
              "  model outputs may be incorrect...", sep="\n", flush=True)

    from .make_depthanythingv2_dpt import make_depthanythingv2_dpt_from_original_state_dict as make_dpt_func

    return make_dpt_func(state_dict, enable_cache, enable_optimizations, strict_load)


def determine_model_type_from_state_dict(model_path: str, state_dict: dict) -> str:
    """Key sniffing, same rules as make_dpt.py:78-116."""
    keys = state_dict.keys()
    if all(k in keys for k in ("patch_embed", "imgencoder", "reassemble", "fusion", "head")):
        return "depthanythingv2"  # already-converted form (ours)
    if "pretrained.model.layers.0.blocks.0.attn.logit_scale" in keys:
        return "swinv2"
    if "pretrained.model.blocks.0.attn.relative_position_bias_table" in keys:
        return "beit"
    if "pretrained.blocks.0.ls1.gamma" in keys:
        name = osp.basename(model_path).lower()
        is_v2 = "v2" in name
        is_v1 = (not is_v2) and (("anything_vit" in name) or ("v1" in name))
        if (not is_v1) and (not is_v2):
            print("", "WARNING: Unable to determine DepthAnything model version!", "-> Will assume v2",
                  "-> Will use v1 if the file name contains 'v1'", sep="\n")
        return "depthanythingv1" if is_v1 else "depthanythingv2"
    return "unknown"
