"""Shared test helpers (CPU side)."""
import json
import os

import numpy as np
import torch

from muggled_dpt_amd.state_dict_conversion import (convert_state_dict_keys, flatten_components,
                                                   get_model_config_from_state_dict)
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict

# ---- parity tolerances (rel = max|y - ref| / max|ref|, the north-star metric) ---------------------------
# float32 model  = MDPT_PREC_BF16X3 (hi/lo split bf16 MFMA operands, fp32 accumulate). North-star bar 1e-3; measured on the
#                  MI355X 1.5e-5 ... 3.5e-5 on every full-size model (ViT-L 504 / 1036, BEiT-L, SwinV2-L): asserted at 1e-4.
# bfloat16 model = MDPT_PREC_BF16 (single-pass bf16 MFMA operands, fp32 accumulate). Measured 0.9e-2 ... 1.4e-2 on the full-size
#                  models (PyTorch's own bf16 CPU path is 1.9e-2 off its fp32 path on the same weights, BASELINE.md §2 - a pure
#                  bf16 pipeline cannot meet 1e-3): asserted at 2e-2. Where a test needs more, it says so next to the measured figure.
REL_TOL_X3 = 1e-4
REL_TOL_BF16 = 2e-2
# The 64-feature / 4-block TOY configurations ("tiny", "beit_tiny", "swin2_tiny": no averaging over wide features, a 32-channel toy head)
# amplify the bf16 rounding noise: measured 0.6e-2 ... 2.5e-2 at the stage boundaries and depth (gpurun_out/parity_report.json of the
# round-2 run: tiny_v1 2.51e-2, tiny_rect 2.09e-2), up to 3.4e-2 on the depth of beit_tiny's 6x2 grid (tests that need it say so).
REL_TOL_BF16_TOY = 3e-2

_CACHE = {}
_RECORDS = []  # (pytest node id, error): every rel_err a test computes, dumped by conftest.py at session end


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


def record_err(err: float, note: str = "") -> float:
    """Remember a measured parity error under the running test's id (tolerances are set from these reports)."""
    _RECORDS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], note, float(err)))
    return err


def rel_err(y: torch.Tensor, ref: torch.Tensor) -> float:
    """The north-star error metric: max|y - ref| / max|ref|."""
    return record_err(float((y.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)))


def dump_records(path: str) -> None:
    if not _RECORDS:
        return
    worst = {}
    for test, note, err in _RECORDS:
        key = test + (f" [{note}]" if note else "")
        worst[key] = max(worst.get(key, 0.0), err)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        json.dump(dict(sorted(worst.items())), fh, indent=1)


def stats(t: torch.Tensor) -> np.ndarray:
    t = t.double()
    return np.array([t.min(), t.max(), t.mean(), t.pow(2).sum().sqrt()], dtype=np.float64)
