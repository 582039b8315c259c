#!/usr/bin/env python3
"""Probe insv2v_ffn_fused with structured weights to localise a layout error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch
import torch.nn.functional as F
from insv2v import ops
from insv2v.fused import pack_ffn_stream
dev = torch.device("cuda:0")
C, NH, M = 320, 1280, 128
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).half()

def run(w1f, b1f, w2, b2, what):
    st = pack_ffn_stream(w1f, b1f, w2, b2).to(dev)
    out = ops.ffn_fused(x.to(dev), st, NH).float().cpu()
    xf = x.float()
    xn = ((xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()).half().float()
    y = xn @ w1f.t() + b1f
    h, gg = y.chunk(2, -1)
    ref = (h * F.gelu(gg)).half().float() @ w2.t() + b2 + xf
    err = (out - ref).abs()
    print(f"{what}: max err {err.max():.4g} (ref max {ref.abs().max():.3g}); bad rows {int((err.max(1).values > 0.05).sum())}/{M}, bad cols {int((err.max(0).values > 0.05).sum())}/{C}")
    if err.max() > 0.05:
        r, c = divmod(int(err.argmax()), C)
        print("   worst at row", r, "col", c, "out", out[r, c].item(), "ref", ref[r, c].item())
        print("   per-col max err (first 48):", [round(v, 2) for v in err.max(0).values[:48].tolist()])
        print("   per-row max err (first 40):", [round(v, 2) for v in err.max(1).values[:40].tolist()])
    return out, ref

Z1, zb1, Z2, zb2 = torch.zeros(2 * NH, C), torch.zeros(2 * NH), torch.zeros(C, NH), torch.zeros(C)
run(Z1, zb1, Z2, zb2, "all zero (out = x)")
run(Z1, zb1, Z2, torch.arange(C).float() * 0.01, "b2 only")
# h = 1 (bias), g = 3 (bias): P = gelu(3) everywhere; W2 = one-hot rows -> out[c] = x + gelu(3) * sum_h W2[c][h]
b1 = torch.cat([torch.ones(NH), torch.full((NH,), 3.0)])
w2 = torch.zeros(C, NH); w2[torch.arange(C), torch.arange(C) * 4] = 1.0
run(Z1, b1, w2.half().float(), zb2, "bias-only hidden, one-hot W2")
b1v = torch.cat([torch.arange(NH).float() * 0.001, torch.full((NH,), 3.0)])
run(Z1, b1v, w2.half().float(), zb2, "hidden = ramp, one-hot W2")
w1 = torch.zeros(2 * NH, C); w1[torch.arange(NH), torch.arange(NH) % C] = 1.0
run(w1.half().float(), torch.cat([torch.zeros(NH), torch.full((NH,), 3.0)]), w2.half().float(), zb2, "one-hot W1 (h = xn[c]), one-hot W2")
w1r = (torch.randn(2 * NH, C, generator=g) * C ** -0.5).half().float()
run(w1r, zb1, w2.half().float(), zb2, "random W1, one-hot W2")
w2r = (torch.randn(C, NH, generator=g) * NH ** -0.5).half().float()
run(w1r, torch.randn(2 * NH, generator=g) * 0.3, w2r, torch.randn(C, generator=g) * 0.3, "all random")
