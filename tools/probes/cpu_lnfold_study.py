#!/usr/bin/env python3
"""CPU emulation (TEST INFRASTRUCTURE: imports the oracle through tests/precision_budget/emulate_operand_rounding.py) of the one encoder lever no round
built: LayerNorm folded into the consuming Linear,

    LN(x) W^T + b  =  rstd_r * ( x (gamma (.) W)^T  -  mu_r * c )  +  (beta W^T + b),      c_n = sum_k round(gamma (.) W)_nk

so that the GEMM's A operand is the ROUNDED RAW residual stream (written by the residual GEMM's epilogue) and the 48 layernorm launches of ViT-L go.
What it changes is the rounding point: the shipped path rounds LN(x) to the operand format, the fold rounds x. This script measures what that does to the
depth map, per operand format, on the plain synthetic weights and on the "realistic statistics" ones (massive-activation channels, layer scales down
to 1e-5: muggled_dpt_amd/synthetic.py realistic_statistics), against the fp32 oracle:

    python tools/probes/cpu_lnfold_study.py [--model vits] [--size 280] [--images 0 7]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as TF

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tests", "precision_budget"))
import emulate_operand_rounding as emu  # noqa: E402
from oracle import dpt_oracle as oracle  # noqa: E402


class FoldProxy(emu._FProxy):
    """The emulator's functional proxy + the fold: a Linear of class qkv / fc1 whose input IS a LayerNorm output contracts the rounded raw rows."""

    def __init__(self, policy, idmap, fold):
        super().__init__(policy, idmap)
        self.fold, self.ln = fold, {}

    def layer_norm(self, x, shape, weight, bias, eps):
        y = TF.layer_norm(x, shape, weight, bias, eps)
        self.ln = {id(y): (y, x, weight, bias, eps)}  # (y kept alive so that its id stays unique)
        return y

    def linear(self, x, weight, bias=None):
        cls = self.idmap.get(id(weight))
        rec = self.ln.get(id(x))
        if not (self.fold and rec is not None and cls in ("qkv", "fc1")):
            return super().linear(x, weight, bias)
        _, xr, g, b, eps = rec
        m = self.policy[cls]
        base = "bf16" if m.startswith("bf16") else "f16"
        mu = xr.mean(-1, keepdim=True)
        rstd = (xr.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
        wg = weight * g
        if m.endswith("x3"):  # fp32-class: hi + lo planes of both operands
            xt, wr = emu.rnd(xr, base + "x3"), emu.rnd(wg, base + "x3")
        else:
            # fp16: a per-row power-of-two scale keeps rows with massive-activation channels in range (exact, undone by rstd_r)
            xt, wr = emu.rnd(xr, base), emu.rnd(wg, base)
        y = rstd * (TF.linear(xt, wr) - mu * wr.sum(-1)) + (TF.linear(b[None], weight)[0] + (0 if bias is None else bias))
        if m == "f16c":  # token-mean compensation of the weight rounding, on the folded operands
            a = rstd * (xt - mu)
            n = a.shape[-2]
            step = 8 if n >= 1024 else (4 if n >= 256 else 1)
            y = y + TF.linear(emu.rnd(a[..., ::step, :].mean(dim=-2, keepdim=True), "f16"), emu.rnd(wg - wr, "f16"))
        return y


def forward(w, cfg, x, policy, fold):
    idmap = {id(t): emu.weight_class(k) for k, t in w.items()}
    proxy = FoldProxy(policy, idmap, fold)
    am = policy["attn"]

    def attention(wd, pre, t, num_heads, capture=None):
        b, n, c = t.shape
        d = c // num_heads
        qkv = proxy.linear(t, wd[f"{pre}.qkv.weight"], wd[f"{pre}.qkv.bias"]).reshape(b, n, 3, num_heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = emu.rnd_a(qkv[0] * d**-0.5, am), emu.rnd_w(qkv[1], am), emu.rnd_w(qkv[2], am)
        s = q @ k.transpose(-2, -1)
        p = torch.exp(s - s.amax(dim=-1, keepdim=True))
        y = (emu.rnd_a(p, am) @ v) / p.sum(dim=-1, keepdim=True)
        return proxy.linear(y.transpose(1, 2).reshape(b, n, c), wd[f"{pre}.proj.weight"], wd[f"{pre}.proj.bias"])

    saved = (oracle.F, oracle.attention)
    oracle.F, oracle.attention = proxy, attention
    try:
        return oracle.forward(w, cfg, x)
    finally:
        oracle.F, oracle.attention = saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vits")
    ap.add_argument("--size", type=int, default=280)
    ap.add_argument("--images", type=int, nargs="+", default=[0, 7])
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from helpers import seeded_input
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict, realistic_statistics
    xs = seeded_input((32, 3, args.size, args.size), 1)
    uniform = lambda m: {c: m for c in emu.CLASSES}

    def mixed():
        p = uniform("f16x3")
        for c in ("qkv", "proj", "fc1", "fc2"):
            p[c] = "f16c"
        p["attn"] = "f16"
        return p

    for wname in ("plain synthetic", "realistic statistics"):
        osd = make_synthetic_original_state_dict(args.model, 0)
        if wname != "plain synthetic":
            osd = realistic_statistics(osd, seed=0)
        cfg = get_model_config_from_state_dict(osd)
        w = flatten_components(convert_state_dict_keys(cfg, osd))
        print(f"== {args.model} {args.size}x{args.size}, {wname} weights; rel. error of the depth map vs the fp32 oracle, images {args.images}: shipped rounding point | LayerNorm folded")
        refs = {i: oracle.forward(w, cfg, xs[i:i + 1]) for i in args.images}
        for label, pol in (("all bf16", uniform("bf16")), ("all fp16", uniform("f16")), ("mixed table (encoder f16 + compensation, decoder 3 terms)", mixed()),
                           ("all bf16x3", uniform("bf16x3"))):
            t0 = time.time()
            row = []
            for fold in (False, True):
                row.append([emu.rel_err(forward(w, cfg, xs[i:i + 1], pol, fold), refs[i]) for i in args.images])
            print(f"  {label:62s} " + " ".join(f"{e:.2e}" for e in row[0]) + "  |  " + " ".join(f"{e:.2e}" for e in row[1]) + f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
