#!/usr/bin/env python3
"""Which product terms does the GPU's fp8 form carry? One fusion block (stage-level entry point) against CPU emulations that keep the main term and
(a) both fp8 cross terms, (b) only A_lo W_hi, (c) only A_hi W_lo, (d) none. Debug probe for tests/test_gpu_f8_cross.py.
   python tools/probes/gpu_f8_term_check.py [batch size]"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "precision_budget"))
import emulate_operand_rounding as emu
from oracle import dpt_oracle as orc
from tests.test_gpu_f8_cross import _model, _policy
from muggled_dpt_amd import native
import torch.nn.functional as TF
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 56
which = [int(a) for a in sys.argv[3:]] or [3]
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
orig = emu._FProxy._contract
def make(term1, term2):
    def _contract(self, fn, x, weight, bias, m, kdim_x, kdim_w, **kw):
        base, fmt = emu.split_modes(m)
        xh, wh = emu.rnd(x, "f16"), emu.rnd(weight, "f16")
        rows = 1 if fn is TF.conv_transpose2d else 0
        y = fn(xh, wh, bias, **kw)
        if term1: y = y + fn(emu.sf8_act(x - xh, 16), emu.sf8_weight(wh, rows), None, **kw)
        if term2: y = y + fn(emu.sf8_act(xh, 0), emu.sf8_weight(weight - wh, rows), None, **kw)
        return y
    return _contract
for classes in ({"fusion": 5, "fusion_in": 5, "fusion_proj": 5}, {"fusion": 4, "fusion_in": 4, "fusion_proj": 4}):
    model, cfg, w = _model(classes)
    g = size // 14
    sizes = [4 * g, 2 * g, g, g // 2]
    gen = torch.Generator().manual_seed(5)
    reasm = [torch.randn(batch, 256, s, s, generator=gen) * 2.0 for s in sizes]
    for i in which:
        prior = None if i == 3 else torch.randn(batch, 256, sizes[i], sizes[i], generator=gen)
        dev = (reasm[i].cuda(),) if prior is None else (reasm[i].cuda(), prior.cuda())
        y8 = model.fusion.blocks[i](*dev).cpu()[:1]
        pol = _policy(emu, fusion="f16x3@sf8")
        print(f"classes {classes} block {i} batch {batch} size {size}:")
        for name, t1, t2 in (("both cross terms", 1, 1), ("A_lo W_hi only", 1, 0), ("A_hi W_lo only", 0, 1), ("no cross term", 0, 0)):
            emu._FProxy._contract = make(t1, t2)
            ref = emu.emulated_call(orc.fusion_block, w, pol, w, i, reasm[i][:1], None if prior is None else prior[:1])
            print(f"   GPU vs emulation with {name:18s}: {rel(y8, ref):.2e}")
        emu._FProxy._contract = orig
