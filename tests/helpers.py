"""Shared test helpers (CPU side)."""
import numpy as np
import torch

from muggled_dpt_amd.state_dict_conversion import (convert_state_dict_keys, flatten_components,
                                                   get_model_config_from_state_dict)
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict

_CACHE = {}


def synthetic_model(name: str, seed: int = 0):
    """(original_state_dict, config, flat new-format weight dict) for a named synthetic config."""
    key = (name, seed)
    if key not in _CACHE:
        osd = make_synthetic_original_state_dict(name, seed)
        cfg = get_model_config_from_state_dict(osd)
        w = flatten_components(convert_state_dict_keys(cfg, osd))
        _CACHE.clear()  # keep at most one (ViT-L is 1.3 GB)
        _CACHE[key] = (osd, cfg, w)
    return _CACHE[key]


def seeded_input(shape, seed=1):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def rel_err(y: torch.Tensor, ref: torch.Tensor) -> float:
    """The north-star error metric: max|y - ref| / max|ref|."""
    return float((y.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30))


def stats(t: torch.Tensor) -> np.ndarray:
    t = t.double()
    return np.array([t.min(), t.max(), t.mean(), t.pow(2).sum().sqrt()], dtype=np.float64)
