#!/usr/bin/env python3
"""One-shot GPU diagnosis (run on the MI355X box): checks every kernel step of transformer block 0, every stage
entry point and the fused forward against the CPU oracle, in both arithmetic modes, and writes
gpurun_out/diag.json. Not a pytest file: it never stops at the first mismatch, so one GPU trip localises a bug.

usage: python tools/probes/gpu_diagnose.py [--skip-vitl] [--only tiny]
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native  # noqa: E402
from oracle import dpt_oracle as O  # noqa: E402
from tests.helpers import rel_err, seeded_input, synthetic_model  # noqa: E402

REPORT = {}


def note(key, value):
    REPORT[key] = value
    print(f"{key:60s} {value}", flush=True)


def err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    if a.shape != b.shape:
        return f"SHAPE {tuple(a.shape)} vs {tuple(b.shape)}"
    d = (a - b).abs()
    return {"max_abs": float(d.max()), "rel_to_max": float(d.max() / b.abs().max().clamp_min(1e-30)), "nan": int(torch.isnan(a).sum())}


def dbg_read(model, name, numel, batch, size_hw):
    eng = model._get_engine()
    out = torch.empty(numel, device=eng.device, dtype=torch.float32)
    ws_ptr, ws_bytes = eng.workspace(batch, size_hw)
    stream = torch.cuda.current_stream().cuda_stream
    native.check(eng.lib, eng.lib.mdpt_debug_read(eng.handle, name.encode(), out.data_ptr(), numel, ws_ptr, ws_bytes, stream))
    torch.cuda.synchronize()
    return out.cpu()


def set_stop(model, block, step):
    eng = model._get_engine()
    native.check(eng.lib, eng.lib.mdpt_debug_set_stop(eng.handle, block, step))


def block0_steps(tag, model, w, cfg, x):
    """Per-kernel check of transformer block 0 (steps LN1, QKV, attention, proj, LN2, fc1, fc2)."""
    B, _, H, W = x.shape
    F, heads = cfg["features_per_token"], cfg["num_heads"]
    tok, hw = O.patch_embed(w, x)
    N = tok.shape[1] + 1
    npad, npadv = (N + 7) // 8 * 8, (N + 63) // 64 * 64
    cls = w["imgencoder.cls_token"] + w["imgencoder.posenc.cls_embedding"]
    t0 = torch.cat((cls.expand(B, -1, -1), tok + O.position_embedding(w, hw)), dim=1)
    pre = "imgencoder.stages.0.blocks.0"
    xn1 = O.layernorm(t0, w[f"{pre}.norm1.weight"], w[f"{pre}.norm1.bias"])
    qkv = torch.nn.functional.linear(xn1, w[f"{pre}.attn.qkv.weight"], w[f"{pre}.attn.qkv.bias"]).reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * 0.125, qkv[1], qkv[2]
    att = (torch.softmax(q @ k.transpose(-2, -1), dim=-1) @ v).transpose(1, 2).reshape(B, N, F)
    r1 = t0 + w[f"{pre}.scale_attn"] * torch.nn.functional.linear(att, w[f"{pre}.attn.proj.weight"], w[f"{pre}.attn.proj.bias"])
    xn2 = O.layernorm(r1, w[f"{pre}.norm2.weight"], w[f"{pre}.norm2.bias"])
    hb = torch.nn.functional.gelu(torch.nn.functional.linear(xn2, w[f"{pre}.mlp.layers.0.weight"], w[f"{pre}.mlp.layers.0.bias"]))
    r2 = r1 + w[f"{pre}.scale_mlp"] * torch.nn.functional.linear(hb, w[f"{pre}.mlp.layers.2.weight"], w[f"{pre}.mlp.layers.2.bias"])
    xg = x.to("cuda").to(next(model.parameters()).dtype)
    size = (H, W)

    def run(step):
        set_stop(model, 0, step)
        model(xg)
        torch.cuda.synchronize()

    run(0)
    note(f"{tag}/im2col_vs_unfold", err(dbg_read(model, "im2col", B * (N - 1) * ((3 * 196 + 63) // 64 * 64), B, size).view(B, N - 1, -1)[:, :, :588],
                                          torch.nn.functional.unfold(x, 14, stride=14).transpose(1, 2)))
    note(f"{tag}/pos", err(dbg_read(model, "pos", (N - 1) * F, B, size).view(1, N - 1, F), O.position_embedding(w, hw)))
    note(f"{tag}/step0_LN1(xn)", err(dbg_read(model, "xn", B * npad * F, B, size).view(B, npad, F)[:, :N], xn1))
    run(1)
    note(f"{tag}/step1_Q", err(dbg_read(model, "q", B * heads * npad * 64, B, size).view(B, heads, npad, 64)[:, :, :N], q))
    note(f"{tag}/step1_K", err(dbg_read(model, "k", B * heads * npad * 64, B, size).view(B, heads, npad, 64)[:, :, :N], k))
    vt = dbg_read(model, "vt", B * heads * 64 * npadv, B, size).view(B, heads, 64, npadv)
    note(f"{tag}/step1_Vt", err(vt[:, :, :, :N], v.transpose(-2, -1)))
    note(f"{tag}/step1_Vt_pad_is_zero", float(vt[:, :, :, npad:].abs().max()) if npadv > npad else 0.0)
    run(2)
    note(f"{tag}/step2_attention", err(dbg_read(model, "att", B * npad * F, B, size).view(B, npad, F)[:, :N], att))
    run(3)
    note(f"{tag}/step3_proj+resid", err(dbg_read(model, "resid", B * npad * F, B, size).view(B, npad, F)[:, :N], r1))
    run(4)
    note(f"{tag}/step4_LN2(xn)", err(dbg_read(model, "xn", B * npad * F, B, size).view(B, npad, F)[:, :N], xn2))
    run(5)
    note(f"{tag}/step5_fc1+gelu", err(dbg_read(model, "hbuf", B * npad * 4 * F, B, size).view(B, npad, 4 * F)[:, :N], hb))
    run(6)
    note(f"{tag}/step6_fc2+resid", err(dbg_read(model, "resid", B * npad * F, B, size).view(B, npad, F)[:, :N], r2))
    set_stop(model, -1, -1)


def full_and_stages(tag, model, w, cfg, x, time_it=False):
    dt = next(model.parameters()).dtype
    xg = x.to("cuda").to(dt)
    depth_ref, st = O.forward(w, cfg, x, return_stages=True)
    B, _, H, W = x.shape
    y = model(xg)
    torch.cuda.synchronize()
    note(f"{tag}/forward_depth", err(y.float(), depth_ref))
    taps = model.debug_taps(B, (H, W))
    torch.cuda.synchronize()
    for i in range(4):
        note(f"{tag}/fwd_tap{i}", err(taps["stages"][i], st["stages"][i]))
    for i in range(4):
        note(f"{tag}/fwd_reasm{i}", err(taps["reasm"][i], st["reasm"][i]))
    note(f"{tag}/fwd_fused", err(taps["fused"], st["fused"]))
    # stage entry points fed with ORACLE inputs (isolates each stage)
    tok, hw = model.patch_embed(xg)
    note(f"{tag}/stage_patch_embed", err(tok.float(), st["patch_tokens"]))
    note(f"{tag}/stage_patch_grid", f"{tuple(hw)} vs {tuple(st['grid_hw'])}")
    enc = model.imgencoder(st["patch_tokens"].to("cuda").to(dt), st["grid_hw"])
    for i in range(4):
        note(f"{tag}/stage_encoder_tap{i}", err(enc[i].float(), st["stages"][i]))
    rs = model.reassemble(*[t.to("cuda").to(dt) for t in st["stages"]], st["grid_hw"])
    for i in range(4):
        note(f"{tag}/stage_reassemble{i}", err(rs[i].float(), st["reasm"][i]))
    fu = model.fusion(*[t.to("cuda").to(dt) for t in st["reasm"]])
    note(f"{tag}/stage_fusion", err(fu.float(), st["fused"]))
    hd = model.head(st["fused"].to("cuda").to(dt))
    note(f"{tag}/stage_head", err(hd.float(), depth_ref))
    if time_it:
        for _ in range(3):
            model(xg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            model(xg)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        note(f"{tag}/ms_per_forward(B={B})", round(ms, 3))
        note(f"{tag}/maps_per_s", round(B / ms * 1e3, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-vitl", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    note("device", torch.cuda.get_device_name(0))
    try:
        # ---------------- tiny: full step-by-step
        osd, cfg, w = synthetic_model("tiny", 0)
        x = seeded_input((2, 3, 56, 56), 1)
        for dtype, tag in ((torch.float32, "tiny/x3"), (torch.bfloat16, "tiny/bf16")):
            _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
            model = model.to("cuda", dtype)
            try:
                block0_steps(tag, model, w, cfg, x)
            except Exception as e:  # keep going: the stage checks may still tell us something
                note(f"{tag}/block0_steps_EXCEPTION", repr(e))
            try:
                full_and_stages(tag, model, w, cfg, x)
            except Exception as e:
                note(f"{tag}/full_EXCEPTION", repr(e))
            # rectangular grid (2x6)
            try:
                x2 = seeded_input((1, 3, 28, 84), 2)
                y2 = model(x2.to("cuda").to(dtype))
                note(f"{tag}/rect_2x6_depth", err(y2.float(), O.forward(w, cfg, x2)))
            except Exception as e:
                note(f"{tag}/rect_EXCEPTION", repr(e))
            del model
        if args.only == "tiny":
            return
        # ---------------- ViT-S @504
        osd, cfg, w = synthetic_model("vits", 0)
        x = seeded_input((1, 3, 504, 504), 1)
        for dtype, tag in ((torch.float32, "vits504/x3"), (torch.bfloat16, "vits504/bf16")):
            _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
            model = model.to("cuda", dtype)
            try:
                block0_steps(tag, model, w, cfg, x)
                full_and_stages(tag, model, w, cfg, x, time_it=True)
            except Exception as e:
                note(f"{tag}/EXCEPTION", repr(e))
            del model
        # ---------------- ViT-L @504
        if not args.skip_vitl:
            osd, cfg, w = synthetic_model("vitl", 0)
            x = seeded_input((1, 3, 504, 504), 1)
            t0 = time.perf_counter()
            depth_ref = O.forward(w, cfg, x)
            note("vitl504/oracle_cpu_seconds(B=1)", round(time.perf_counter() - t0, 2))
            for dtype, tag in ((torch.float32, "vitl504/x3"), (torch.bfloat16, "vitl504/bf16")):
                _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
                model = model.to("cuda", dtype)
                try:
                    y = model(x.to("cuda").to(dtype))
                    torch.cuda.synchronize()
                    note(f"{tag}/forward_depth", err(y.float(), depth_ref))
                    for bsz in (1, 8, 32):
                        xb = torch.randn(bsz, 3, 504, 504, device="cuda", dtype=dtype)
                        for tile in (1, 2):
                            model.set_gemm_tile(tile)
                            for _ in range(2):
                                model(xb)
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            n = 5
                            for _ in range(n):
                                model(xb)
                            torch.cuda.synchronize()
                            ms = (time.perf_counter() - t0) / n * 1e3
                            note(f"{tag}/B={bsz}/tile={tile}/ms", round(ms, 2))
                            note(f"{tag}/B={bsz}/tile={tile}/maps_per_s", round(bsz / ms * 1e3, 2))
                        model.set_gemm_tile(0)
                except Exception as e:
                    note(f"{tag}/EXCEPTION", repr(e))
                del model
                torch.cuda.empty_cache()
    finally:
        with open(os.path.join(REPO, "gpurun_out", "diag.json"), "w") as f:
            json.dump(REPORT, f, indent=1, default=str)


if __name__ == "__main__":
    main()
