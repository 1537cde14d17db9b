"""muggled_dpt_amd: MI355X-native (gfx950 HIP) Depth-Anything-V2 DPT inference behind muggled_dpt's Python API."""

from .dpt_model import DPTModel  # noqa: F401
from .make_depthanythingv2_dpt import make_depthanythingv2_dpt, make_depthanythingv2_dpt_from_original_state_dict  # noqa: F401
from .make_depthanythingv1_dpt import make_depthanythingv1_dpt, make_depthanythingv1_dpt_from_original_state_dict  # noqa: F401
from .make_beit_dpt import make_beit_dpt, make_beit_dpt_from_midas_v31_state_dict  # noqa: F401
from .make_dpt import make_dpt_from_state_dict  # noqa: F401
from .make_swinv2_dpt import make_swinv2_dpt, make_swinv2_dpt_from_midas_v31_state_dict  # noqa: F401
from .export import export_model, load_exported  # noqa: F401

__all__ = ["DPTModel", "make_dpt_from_state_dict", "make_depthanythingv2_dpt", "make_depthanythingv2_dpt_from_original_state_dict",
           "make_depthanythingv1_dpt", "make_depthanythingv1_dpt_from_original_state_dict",
           "make_beit_dpt", "make_beit_dpt_from_midas_v31_state_dict",
           "make_swinv2_dpt", "make_swinv2_dpt_from_midas_v31_state_dict", "export_model", "load_exported"]
