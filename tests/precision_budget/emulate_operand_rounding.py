"""CPU emulation of the MFMA operand-rounding modes of libmdpt (TEST INFRASTRUCTURE: imports the oracle).

Every contraction of the path (Linear / conv / transposed conv / q k^T / p v) is computed in fp32 on operands that were first rounded the
way the GPU kernels round them (bf16, fp16, or the hi + lo split of either), per OP CLASS. That reproduces the error of a precision mode
against the fp32 oracle without a GPU (the accumulation order differs, which is an fp32-level effect), so the per-class error budget that
`MDPT_PREC_MIXED` is derived from can be rebuilt anywhere:

    python tests/precision_budget/emulate_operand_rounding.py --model vitl --images 0 7 --study budget

The GPU check of the same table is tests/test_gpu_precision_modes.py (the emulated and the measured errors are compared there).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import dpt_oracle as oracle  # noqa: E402

CLASSES = ("patch", "qkv", "attn", "proj", "fc1", "fc2", "reasm", "fusion", "head")
F16_MAX = 65504.0


def rnd(x: torch.Tensor, mode: str) -> torch.Tensor:
    if mode == "f32":
        return x
    if mode == "bf16":
        return x.bfloat16().float()
    if mode == "f16":
        return x.clamp(-F16_MAX, F16_MAX).half().float()
    if mode == "bf16x3":
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    if mode == "f16x3":
        hi = x.clamp(-F16_MAX, F16_MAX).half().float()
        return hi + (x - hi).half().float()
    if mode == "f16x2a":  # activations split (2 planes), weights one fp16 plane: a_hi*w + a_lo*w
        raise ValueError("asymmetric modes are handled by rnd_a / rnd_w")
    raise ValueError(mode)


def rnd_a(x, mode):
    if mode == "f16x2a":
        return rnd(x, "f16x3")
    if mode == "f16x2w":
        return rnd(x, "f16")
    return rnd(x, mode)


def rnd_w(x, mode):
    if mode == "f16x2a":
        return rnd(x, "f16")
    if mode == "f16x2w":
        return rnd(x, "f16x3")
    return rnd(x, mode)


# ---- round-6 study: the CROSS TERMS of a split product (A_lo W_hi, A_hi W_lo) on block-scaled low-precision operands (gfx950's
# v_mfma_scale_f32_32x32x64_f8f6f4: OCP MX formats, one power-of-two scale per 32 consecutive K elements; fp8 at 2x, fp6 / fp4 at 4x the fp16 MFMA
# rate). Mode "f16x2a@mxfp6" = A_hi W_hi in fp16 + Q(A_lo) Q(W_hi) with Q = MX e2m3; "f16x3@..." adds Q(A_hi) Q(W_lo).
MX = {"mxfp8": (4, 3, -6, 8, 448.0), "mxfp6": (2, 3, 0, 2, 7.5), "mxbf6": (3, 2, -2, 4, 28.0), "mxfp4": (2, 1, 0, 2, 6.0)}  # (E, M, emin, emax, max)


def mx_quant(x: torch.Tensor, fmt: str, dim: int) -> torch.Tensor:
    """OCP MX quantisation of x along `dim` in blocks of 32 (zero padded): shared scale 2^(floor(log2 max|v|) - emax), elements rounded to
    nearest (ties to even) in the element format, saturating."""
    _, m, emin, emax, vmax = MX[fmt]
    xt = x.movedim(dim, -1)
    k = xt.shape[-1]
    pad = (-k) % 32
    if pad:
        xt = torch.nn.functional.pad(xt, (0, pad))
    blk = xt.reshape(*xt.shape[:-1], -1, 32).double()
    amax = blk.abs().amax(dim=-1, keepdim=True)
    scale = torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-300))) - emax)
    v = blk / scale
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-300))).clamp(emin, emax)
    q = torch.exp2(e - m)
    r = (torch.round(v / q) * q).clamp(-vmax, vmax)  # torch.round: half to even
    out = torch.where(amax > 0, r * scale, torch.zeros_like(r)).reshape(*xt.shape).float()
    if pad:
        out = out[..., :k]
    return out.movedim(-1, dim)


# ---- the form round 6 BUILT (csrc/f8_cross.h): fp8 cross terms with STATIC scales - no per-block scale arithmetic in any kernel.
#   activations: E5M2 (fp16's exponent range, 2 significand bits). lo plane = e5m2((A - A_hi) * 2^16) with the constant E8M0 scale 2^-16 in
#                the MFMA's scale operand (A_lo <= 2^-11 |A|: the shift keeps the residue of every normal fp16 value inside e5m2's NORMAL range);
#                the A_hi W_lo term of the 3-pass classes reads a8 = e5m2(A_hi), scale 1. Saturating.
#   weights:     E4M3 with ONE power-of-two scale per output row (row maximum into [256, 448]): a float format keeps 3 significand bits
#                over 15 binades below the row maximum, so the MX block scale buys nothing here; per-row = a register constant of the kernel.
def _fp_quant(v: torch.Tensor, m: int, emin: int, vmax: float) -> torch.Tensor:
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-300))).clamp_min(emin)
    q = torch.exp2(e - m)
    return (torch.round(v / q) * q).clamp(-vmax, vmax)


def sf8_act(x: torch.Tensor, shift: int) -> torch.Tensor:
    """e5m2(x * 2^shift) * 2^-shift, round to nearest even, saturating at 57344"""
    return (_fp_quant(x.double() * 2.0 ** shift, 2, -14, 57344.0) * 2.0 ** -shift).float()


def sf8_weight(w: torch.Tensor, rows_dim: int) -> torch.Tensor:
    """e4m3 with one power-of-two scale per output row (`rows_dim` = the output-channel dimension of the tensor)"""
    wt = w.movedim(rows_dim, 0).double()
    flat = wt.reshape(wt.shape[0], -1)
    amax = flat.abs().amax(dim=1, keepdim=True)
    scale = torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-300))) - 7)  # row maximum into [128, 256): pack_weight_f8_kernel (e4m3 ends at 448)
    out = torch.where(amax > 0, _fp_quant(flat / scale, 3, -6, 448.0) * scale, torch.zeros_like(flat))
    return out.reshape(wt.shape).movedim(0, rows_dim).float()


def split_modes(mode: str):
    """"f16x2a@mxfp6" -> ("f16x2a", "mxfp6"); plain modes -> (mode, None)"""
    return tuple(mode.split("@")) if "@" in mode else (mode, None)


FINE = False  # --fine: split the decoder classes by layer (study only; the library switches whole classes)
FINE_CLASSES = ("reasm_1x1", "reasm_resample", "reasm_fuse3x3", "fusion_rcu_a", "fusion_rcu_b", "fusion_proj1x1", "head_conv1", "head_conv2")


def fine_class(key: str) -> str | None:
    if key.startswith("reassemble."):
        return "reasm_1x1" if ".resample.0." in key else ("reasm_resample" if ".resample.1." in key else "reasm_fuse3x3")
    if key.startswith("fusion."):
        if "conv_reassembly" in key:
            return "fusion_rcu_a"
        return "fusion_proj1x1" if key.endswith("_seq.2.weight") or key.endswith("_seq.2.bias") else "fusion_rcu_b"
    if key.startswith("head.spatial_upsampler"):
        return "head_conv1"
    if key.startswith("head.proj_1ch.0"):
        return "head_conv2"
    return None


def weight_class(key: str) -> str | None:
    if FINE and fine_class(key):
        return fine_class(key)
    if key.startswith("patch_embed.proj"):
        return "patch"
    if ".attn.qkv." in key:
        return "qkv"
    if ".attn.proj." in key:
        return "proj"
    if ".mlp.layers.0." in key or "inner_linear_doubled" in key:
        return "fc1"
    if ".mlp.layers.2." in key or "outer_linear" in key:
        return "fc2"
    if key.startswith("reassemble."):
        return "reasm"
    if key.startswith("fusion."):
        return "fusion"
    if key.startswith("head.proj_1ch.2"):
        return None  # the 32 -> 1 projection runs in fp32 registers on the GPU
    if key.startswith("head."):
        return "head"
    return None


class _FProxy:
    """torch.nn.functional with operand rounding in front of the three contraction entry points the oracle uses."""

    def __init__(self, policy: dict, idmap: dict):
        self.policy, self.idmap = policy, idmap

    def __getattr__(self, name):
        return getattr(TF, name)

    def _mode(self, weight):
        cls = self.idmap.get(id(weight))
        return "f32" if cls is None else self.policy[cls]

    def _contract(self, fn, x, weight, bias, m, kdim_x, kdim_w, **kw):
        """split product with its cross terms on MX operands (see MX above)"""
        base, fmt = split_modes(m)
        xh, wh = rnd(x, "f16"), rnd(weight, "f16")
        if fmt == "sf8":  # static-scale fp8 (the built form): rows of a Linear / conv weight = dim 0, of a transposed-conv weight = dim 1
            rows = 1 if fn is TF.conv_transpose2d else 0
            y = fn(xh, wh, bias, **kw) + fn(sf8_act(x - xh, 16), sf8_weight(wh, rows), None, **kw)
            if base == "f16x3":
                y = y + fn(sf8_act(xh, 0), sf8_weight(weight - wh, rows), None, **kw)
            return y
        y = fn(xh, wh, bias, **kw) + fn(mx_quant(x - xh, fmt, kdim_x), mx_quant(wh, fmt, kdim_w), None, **kw)
        if base == "f16x3":
            y = y + fn(mx_quant(xh, fmt, kdim_x), mx_quant(weight - wh, fmt, kdim_w), None, **kw)
        return y

    def linear(self, x, weight, bias=None):
        m = self._mode(weight)
        if "@" in m:
            return self._contract(TF.linear, x, weight, bias, m, -1, -1)
        if m == "f16c":  # single fp16 pass + the token-mean compensation of the weight rounding (mdpt_stages.cpp wrc_bias)
            xr, wr = rnd(x, "f16"), rnd(weight, "f16")
            wlo = rnd(weight - wr, "f16")
            n = xr.shape[-2]
            step = 8 if n >= 1024 else (4 if n >= 256 else 1)
            mean = rnd(xr[..., ::step, :].mean(dim=-2, keepdim=True), "f16")
            return TF.linear(xr, wr, bias) + TF.linear(mean, wlo)
        return TF.linear(rnd_a(x, m), rnd_w(weight, m), bias)

    def conv2d(self, x, weight, bias=None, **kw):
        m = self._mode(weight)
        if "@" in m:
            return self._contract(TF.conv2d, x, weight, bias, m, 1, 1, **kw)
        return TF.conv2d(rnd_a(x, m), rnd_w(weight, m), bias, **kw)

    def conv_transpose2d(self, x, weight, bias=None, **kw):
        m = self._mode(weight)
        if "@" in m:
            return self._contract(TF.conv_transpose2d, x, weight, bias, m, 1, 0, **kw)
        return TF.conv_transpose2d(rnd_a(x, m), rnd_w(weight, m), bias, **kw)


def emulated_call(fn, w: dict, policy: dict, *args, **kw):
    """Any stage function of the oracle (reassemble, fusion_block, fusion, head ...) with the contraction operands of each class rounded per
    `policy`: tests/test_gpu_f8_cross.py checks the stage-level entry points of the fp8 cross-term forms against this."""
    idmap = {id(t): weight_class(k) for k, t in w.items()}
    saved = oracle.F
    oracle.F = _FProxy(policy, idmap)
    try:
        return fn(*args, **kw)
    finally:
        oracle.F = saved


def emulated_forward(w: dict, cfg: dict, x: torch.Tensor, policy: dict, fold_layer_scale: bool = True) -> torch.Tensor:
    """oracle.forward with the contraction operands of each class rounded per `policy` ({class: mode})."""
    idmap = {id(t): weight_class(k) for k, t in w.items()}
    proxy = _FProxy(policy, idmap)
    am = policy["attn"]

    def attention(wd, pre, t, num_heads, capture=None):
        b, n, c = t.shape
        d = c // num_heads
        qkv = proxy.linear(t, wd[f"{pre}.qkv.weight"], wd[f"{pre}.qkv.bias"]).reshape(b, n, 3, num_heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = rnd_a(qkv[0] * d**-0.5, am), rnd_w(qkv[1], am), rnd_w(qkv[2], am)
        s = q @ k.transpose(-2, -1)
        p = torch.exp(s - s.amax(dim=-1, keepdim=True))  # the kernel rounds the un-normalised probabilities and divides O by the fp32 row sum
        y = (rnd_a(p, am) @ v) / p.sum(dim=-1, keepdim=True)
        y = y.transpose(1, 2).reshape(b, n, c)
        return proxy.linear(y, wd[f"{pre}.proj.weight"], wd[f"{pre}.proj.bias"])

    saved = (oracle.F, oracle.attention)
    oracle.F, oracle.attention = proxy, attention
    try:
        return oracle.forward(w, cfg, x)
    finally:
        oracle.F, oracle.attention = saved


def rel_err(y, ref):
    return float((y.double() - ref.double()).abs().max() / ref.double().abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vitl")
    ap.add_argument("--size", type=int, default=504)
    ap.add_argument("--images", type=int, nargs="+", default=[0])
    ap.add_argument("--study", default="budget", choices=["budget", "modes", "policy", "twopass", "lowlo"])
    ap.add_argument("--policy", default="", help="study=policy: comma list class=mode, others take --base")
    ap.add_argument("--base", default="f16")
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--fine", action="store_true", help="decoder classes split by layer (policy keys: " + ", ".join(FINE_CLASSES) + ")")
    args = ap.parse_args()
    global FINE
    FINE = args.fine
    torch.set_num_threads(args.threads)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from helpers import seeded_input, synthetic_model
    _, cfg, w = synthetic_model(args.model)
    xs = seeded_input((32, 3, args.size, args.size), 1)
    report = {"model": args.model, "size": args.size, "images": args.images, "rows": []}

    def run(label, policy):
        errs = []
        for i in args.images:
            x = xs[i:i + 1]
            key = ("ref", i)
            if key not in run.cache:
                run.cache[key] = oracle.forward(w, cfg, x)
            t0 = time.time()
            y = emulated_forward(w, cfg, x, policy)
            errs.append(rel_err(y, run.cache[key]))
        report["rows"].append({"label": label, "policy": policy, "rel_err": errs})
        print(f"{label:34s} " + " ".join(f"{e:.3e}" for e in errs) + f"   ({time.time() - t0:.1f} s)", flush=True)

    run.cache = {}
    uniform = lambda m: {c: m for c in CLASSES + (FINE_CLASSES if FINE else ())}
    if args.study == "modes":
        for m in ("bf16", "f16", "bf16x3", "f16x3"):
            run(f"all {m}", uniform(m))
    elif args.study == "budget":
        for base in ("bf16", "f16"):
            run(f"all {base}", uniform(base))
            for c in CLASSES:  # one class alone at 1 pass, everything else exact: that class's own contribution
                p = uniform("f32")
                p[c] = base
                run(f"only {c} = {base}", p)
            for c in CLASSES:  # one class lifted to 3 passes: what lifting it buys
                p = uniform(base)
                p[c] = base + "x3"
                run(f"all {base}, {c} = {base}x3", p)
    elif args.study == "twopass":
        # the shipped mixed assignment (encoder: one fp16 pass + compensation; decoder at 3 passes except the fusion blocks' conv_reassembly
        # units), then every decoder layer group alone at 2 passes: weights split (x2w = A_hi W_hi + A_hi W_lo) or activations split (x2a)
        assert FINE, "--study twopass needs --fine"
        def mixed():
            p = uniform("f16x3")
            for c in ("qkv", "proj", "fc1", "fc2"):
                p[c] = "f16c"
            p["attn"] = "f16"
            p["fusion_rcu_a"] = "f16"
            return p
        run("mixed (shipped)", mixed())
        dec = ("reasm_1x1", "reasm_resample", "reasm_fuse3x3", "fusion_rcu_b", "fusion_proj1x1", "head_conv1", "head_conv2")
        for two in ("f16x2w", "f16x2a", "f16"):
            for c in dec:
                p = mixed()
                p[c] = two
                run(f"mixed, {c} = {two}", p)
            p = mixed()
            for c in dec:
                p[c] = two
            run(f"mixed, decoder = {two}", p)
    elif args.study == "lowlo":
        # the shipped mixed table (Depth-Anything families: reasm / 1x1 fusion projections 3 passes, fusion 3x3 convs and both head convs 2 passes
        # with the activations split, conv_reassembly units 1 pass, encoder 1 pass + compensation), then the SAME table with every cross term on
        # block-scaled low-precision operands
        assert FINE, "--study lowlo needs --fine"
        def table(fmt):
            at = "" if fmt is None else "@" + fmt
            p = uniform("f16x3")
            for c in ("qkv", "proj", "fc1", "fc2"):
                p[c] = "f16c"
            p["attn"] = "f16"
            p["fusion_rcu_a"] = "f16"
            for c in ("reasm_1x1", "reasm_resample", "reasm_fuse3x3", "fusion_proj1x1"):
                p[c] = "f16x3" + at
            for c in ("fusion_rcu_b", "head_conv1", "head_conv2"):
                p[c] = "f16x2a" + at
            if fmt is not None:
                p["patch"] = "f16x3" + at
            return p
        run("mixed (shipped table, fp16 cross terms)", table(None))
        for fmt in ("sf8", "mxfp8", "mxfp6", "mxbf6", "mxfp4"):
            run(f"mixed, cross terms in {fmt}", table(fmt))
        p = table(None)
        for c in ("reasm_1x1", "reasm_resample", "reasm_fuse3x3", "fusion_proj1x1", "fusion_rcu_b", "head_conv1", "head_conv2", "patch"):
            p[c] = "f16"
        run("(decoder + patch at one fp16 pass)", p)
    else:
        p = uniform(args.base)
        for item in filter(None, args.policy.split(",")):
            c, m = item.split("=")
            p[c] = m
        run(f"{args.base} + {args.policy}", p)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
