"""Build + ctypes binding of libmdpt.so (the C ABI declared in include/mdpt.h).

The shared library is built IN-TREE (muggled_dpt_amd/csrc/libmdpt.so) with hipcc for gfx950 only;
it links against the HIP runtime and nothing else (no torch types cross the boundary).
There is no fallback: if the library is missing and cannot be built, importing callers get a loud
RuntimeError - the product never silently runs on a CPU/PyTorch path.
"""

from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(CSRC, "libmdpt.so")
# kernel files that touch MFMA operand planes are compiled twice, once per operand format (csrc/op_types.h): bf16 and, with
# -DMDPT_OP_F16, fp16; the launcher symbols carry the format as a suffix and the host side (mdpt_internal.h: OPL) picks per handle
OPERAND_SOURCES = ("gemm.hip", "conv3h.hip", "attention.hip", "elementwise.hip", "swin.hip", "head.hip")
PLAIN_SOURCES = ("postprocess.hip", "stream_probe.hip", "mdpt_api.cpp", "mdpt_inventory.cpp", "mdpt_stages.cpp", "mdpt_debug.cpp", "mdpt_prof.cpp")
SOURCES = OPERAND_SOURCES + PLAIN_SOURCES
HEADERS = ("gemm_common.inc", "gemm_epilogue_strip.inc", "gemm_lockstep.inc", "gemm8_epilogues.inc", "gemm8.inc", "mdpt_kernels.h", "mdpt_launchers.inc", "op_types.h", "mdpt_prof.h", "mdpt_internal.h", "mdpt_swin_plan.inc", "mdpt_swin_stages.inc", "ln_row.h",
           "up_bf16.h",
           os.path.join(REPO, "include", "mdpt.h"))
# (source, extra flags, object stem)
UNITS = tuple((s, (), os.path.splitext(s)[0]) for s in SOURCES) + tuple((s, ("-DMDPT_OP_F16",), os.path.splitext(s)[0] + "_f16")
                                                                        for s in OPERAND_SOURCES)

ABI_VERSION = 6  # MDPT_ABI_VERSION in include/mdpt.h
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
PREC_BF16 = 0
PREC_BF16X3 = 1
PREC_FP16 = 2
PREC_FP16X3 = 3
PREC_MIXED = 4
PRECISIONS = {"bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "fp16": PREC_FP16, "fp16x3": PREC_FP16X3, "mixed": PREC_MIXED}
PASSES_2F8, PASSES_3F8 = 4, 5  # MDPT_PASSES_2F8 / _3F8: two / three products with the cross terms on fp8 planes (csrc/f8_cross.h)
F8_CLASSES = ("reasm", "fusion", "fusion_in", "fusion_proj", "head")  # the classes that have the fp8 form
OP_CLASSES = ("patch", "qkv", "attn", "proj", "fc1", "fc2", "reasm", "fusion", "head", "fusion_in", "head_tail", "fusion_proj")  # MDPT_CLASS_* of include/mdpt.h
FAMILY_DAV2 = 0
FAMILY_DAV1 = 1
FAMILY_BEIT = 2
FAMILY_SWINV2 = 3
E_GRID = -7
POST_F32, POST_U8, POST_U24 = 0, 1, 2
INTERP_BILINEAR, INTERP_BICUBIC = 0, 1


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmdpt.so cannot be built (set HIPCC or install ROCm)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel / host sources and headers libmdpt.so is built from. Measurements that are only
    valid for one version of the kernels (profiles/*_hbm_traffic.json) are stamped with it and ignored when it differs."""
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [x if os.path.isabs(x) else os.path.join(CSRC, x) for x in HEADERS]:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into csrc/libmdpt.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"), "-I", CSRC,
             "-Wno-unused-result"]
    # A/B builds (tools/probes): e.g. MDPT_EXTRA_HIPCC_FLAGS="-DMDPT_DEBUG_SWITCHES" compiles the environment switches of the kernel
    # launchers in; release builds read no environment variable on the launch path
    flags += os.environ.get("MDPT_EXTRA_HIPCC_FLAGS", "").split()
    _sweep_stale_temporaries()

    # objects and the linked library go through pid-unique names and an atomic rename: several processes (one per GPU) may find
    # the library stale at the same time
    tag = f".{os.getpid()}"
    made = []

    def compile_one(unit) -> str:
        src, extra, stem = unit
        obj = os.path.join(CSRC, stem + tag + ".o")
        made.append(obj)
        cmd = [hipcc, *flags, *extra, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src} {' '.join(extra)}:\n{r.stdout}\n{r.stderr}")
        return obj

    tmp = LIB_PATH + tag + ".tmp"
    try:
        with ThreadPoolExecutor(max_workers=min(len(UNITS), max(2, (os.cpu_count() or 8)))) as ex:
            objs = list(ex.map(compile_one, UNITS))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp, "-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        for obj in objs:
            os.replace(obj, obj[: -len(tag + ".o")] + ".o")  # keep the objects (incremental relinks by hand, disassembly)
        os.replace(tmp, LIB_PATH)
    finally:  # an interrupted / failed build leaves nothing pid-tagged behind (they used to ship to the GPU box with every push)
        for path in made + [tmp]:
            if os.path.exists(path):
                os.remove(path)
    return LIB_PATH


def _sweep_stale_temporaries(max_age_s: float = 3600.0) -> None:
    """Remove pid-tagged objects / libraries of builds that died (older than an hour: not a concurrent build of another rank)."""
    import re
    import time
    now = time.time()
    for name in os.listdir(CSRC):
        if re.search(r"\.\d+\.(o|tmp)$", name):
            path = os.path.join(CSRC, name)
            try:
                if now - os.path.getmtime(path) > max_age_s:
                    os.remove(path)
            except OSError:
                pass


class MdptConfig(ctypes.Structure):
    _fields_ = [
        ("features_per_token", ctypes.c_int32),
        ("num_heads", ctypes.c_int32),
        ("num_blocks", ctypes.c_int32),
        ("reassembly_features", ctypes.c_int32 * 4),
        ("base_patch_grid_h", ctypes.c_int32),
        ("base_patch_grid_w", ctypes.c_int32),
        ("fusion_channels", ctypes.c_int32),
        ("patch_size_px", ctypes.c_int32),
        ("is_giant", ctypes.c_int32),
        ("is_metric", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("family", ctypes.c_int32),
        ("swin_heads", ctypes.c_int32 * 4),
        ("swin_layers", ctypes.c_int32 * 4),
        ("swin_window_h", ctypes.c_int32),
        ("swin_window_w", ctypes.c_int32),
        ("swin_pretrained_window", ctypes.c_int32 * 4),
    ]


# every symbol include/mdpt.h declares: (restype, argtypes)
_VP = ctypes.c_void_p
_SZ = ctypes.c_size_t
_I = ctypes.c_int32
_VP4 = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = {
    "mdpt_abi_version": (ctypes.c_int, []),
    "mdpt_last_error": (ctypes.c_char_p, []),
    "mdpt_create": (ctypes.c_int, [ctypes.POINTER(MdptConfig), ctypes.POINTER(_VP)]),
    "mdpt_destroy": (None, [_VP]),
    "mdpt_num_weights": (ctypes.c_int, [_VP]),
    "mdpt_weight_name": (ctypes.c_char_p, [_VP, ctypes.c_int]),
    "mdpt_weight_shape": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_int64)]),
    "mdpt_bind_weight": (ctypes.c_int, [_VP, ctypes.c_char_p, _VP, _I, _I, ctypes.POINTER(ctypes.c_int64)]),
    "mdpt_packed_bytes": (ctypes.c_int, [_VP, ctypes.POINTER(_SZ)]),
    "mdpt_finalize": (ctypes.c_int, [_VP, _VP, _SZ, _VP]),
    "mdpt_workspace_bytes": (ctypes.c_int, [_VP, _I, _I, _I, ctypes.POINTER(_SZ)]),
    "mdpt_forward": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _I, _VP, _I, _VP, _SZ, _VP]),
    "mdpt_patch_embed": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "mdpt_encoder": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _VP4, _VP, _SZ, _VP]),
    "mdpt_reassemble": (ctypes.c_int, [_VP, _VP4, _I, _I, _I, _VP4, _VP, _SZ, _VP]),
    "mdpt_fusion": (ctypes.c_int, [_VP, _VP4, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "mdpt_encoder_probe": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _VP4, _VP4, _VP, _SZ, _VP]),
    "mdpt_attn_probe_shape": (ctypes.c_int, [_VP, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_int64)]),
    "mdpt_encoder_probe_blocks": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _VP4, _VP, _VP, _VP, _SZ, _VP]),
    "mdpt_fusion_block": (ctypes.c_int, [_VP, _I, _VP, _VP, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "mdpt_head": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "mdpt_prepare_image": (ctypes.c_int, [_VP, _I, _I, _VP, _I, _I, _I, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _I, _VP]),
    "mdpt_forward_bgr": (ctypes.c_int, [_VP, _VP, _I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _I, _VP, _I, _VP, _SZ, _VP]),
    "mdpt_post_minmax": (ctypes.c_int, [_VP, _SZ, _VP, _VP, _VP]),
    "mdpt_post_scale_prediction": (ctypes.c_int, [_VP, _I, _I, _I, _VP, _I, _I, _VP, _VP, _VP]),
    "mdpt_post_normalize": (ctypes.c_int, [_VP, _SZ, _VP, _VP, _I, _I, _VP]),
    "mdpt_export_tap": (ctypes.c_int, [_VP, _I, _VP, _VP, _SZ, _VP]),
    "mdpt_set_gemm_tile": (ctypes.c_int, [_VP, _I]),
    "mdpt_set_batch_split": (ctypes.c_int, [_VP, _I]),
    "mdpt_set_latency_mode": (ctypes.c_int, [_VP, _I]),
    "mdpt_set_nonfinite_propagation": (ctypes.c_int, [_VP, _I]),
    "mdpt_debug_gemm": (ctypes.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _VP]),
    "mdpt_debug_attention": (ctypes.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    "mdpt_debug_conv3": (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "mdpt_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "mdpt_profile_report": (ctypes.c_int, [ctypes.c_char_p, _SZ]),
    "mdpt_debug_set_stop": (ctypes.c_int, [_VP, _I, _I]),
    "mdpt_debug_set_ksplit_min": (ctypes.c_int, [_VP, _I, _I]),
    "mdpt_debug_set_reassemble_overlap": (ctypes.c_int, [_VP, _I]),
    "mdpt_debug_set_wscale_policy": (ctypes.c_int, [_VP, _I]),
    "mdpt_debug_set_side_stream_priority": (ctypes.c_int, [_VP, _I]),
    "mdpt_debug_set_side_stream_probe": (ctypes.c_int, [_VP, _I]),
    "mdpt_debug_side_stream_info": (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "mdpt_debug_set_operand_format": (ctypes.c_int, [_I]),
    "mdpt_set_grid_cache": (ctypes.c_int, [_VP, _I]),
    "mdpt_set_class_passes": (ctypes.c_int, [_VP, _I, _I]),
    "mdpt_get_class_passes": (ctypes.c_int, [_VP, _I, ctypes.POINTER(_I)]),
    "mdpt_get_class_f8": (ctypes.c_int, [_VP, _I, ctypes.POINTER(_I)]),
    "mdpt_default_mixed_passes_r05": (None, [_I, ctypes.POINTER(_I)]),
    "mdpt_default_mixed_passes": (None, [ctypes.POINTER(_I)]),
    "mdpt_default_mixed_passes_for": (None, [_I, ctypes.POINTER(_I)]),
    "mdpt_set_weight_rounding_compensation": (ctypes.c_int, [_VP, _I]),
    "mdpt_debug_read": (ctypes.c_int, [_VP, ctypes.c_char_p, _VP, _SZ, _VP, _SZ, _VP]),
    "mdpt_allgather": (ctypes.c_int, [_VP, _VP, _VP, _SZ, _I, _VP]),
}

_LIB = None


def load(auto_build: bool = True) -> ctypes.CDLL:
    """dlopen csrc/libmdpt.so (building it first if needed) and type every entry point."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if auto_build and _stale():
        build()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift between mdpt.h and the .so
        fn.restype = res
        fn.argtypes = args
    if lib.mdpt_abi_version() != ABI_VERSION:
        raise RuntimeError("libmdpt ABI version mismatch")
    _LIB = lib
    return lib


def dtype_code(torch_dtype) -> int:
    """MDPT_DTYPE_* of a torch dtype (fp32 / bf16 / fp16 are the tensor types the C ABI takes)."""
    import torch
    try:
        return {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16}[torch_dtype]
    except KeyError:
        raise TypeError(f"libmdpt takes float32, bfloat16 or float16 tensors, not {torch_dtype}") from None


class MdptError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmdpt error {code}: {message}")
        self.code = code


def check(lib: ctypes.CDLL, rc: int) -> None:
    if rc != 0:
        raise MdptError(rc, lib.mdpt_last_error().decode("utf-8", "replace"))
