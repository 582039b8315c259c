#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) go / no-go prototype for the stride-1 3x3 convolutions with K >= 5 760 (resnet.py:143,159).

MEASUREMENT / ANALYSIS ONLY - nothing here is part of the product.  Two legs:
  numerics (CPU or GPU, torch): the algorithm with the storage the product would use - activations, transformed input V = B^T d B,
      transformed weights U = G g G^T and the 16 product matrices M_k in fp16, fp32 accumulation inside each GEMM - against fp32
      F.conv2d on the same fp16-rounded operands; the single-kernel tolerance of this build is 2e-3 of max|ref|.
  timing (GPU, --time): the 16 GEMMs as ONE launch of the product's engine with 4 x M rows (what a grouped launch with per-row-group
      weights costs; same tiles, same epilogue traffic) beside the direct implicit-GEMM convolution, plus the bytes the two transform
      passes move (priced at the rate the norm passes reach on the box).
usage: winograd_proto.py [--time] [--m-fp32]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv(x, w, m_dtype=torch.float16, dev="cpu"):
    """x [N, C, H, W] fp16 values, w [O, C, 3, 3] fp16 values; returns [N, O, H, W] fp32."""
    N, C, H, W = x.shape
    O = w.shape[0]
    xp = F.pad(x.float(), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [N, C, H/2, W/2, 4, 4]
    V = torch.einsum("ai,nchwij,bj->nchwab", BT.to(dev), d, BT.to(dev)).half()   # stored fp16
    U = torch.einsum("ai,ocij,bj->ocab", G.to(dev), w.float(), G.to(dev)).half()  # stored fp16
    M = torch.einsum("nchwab,ocab->nohwab", V.float(), U.float()).to(m_dtype)     # fp32 accumulation, stored in m_dtype
    Y = torch.einsum("ia,nohwab,jb->nohwij", AT.to(dev), M.float(), AT.to(dev))   # [N, O, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, O, H, W)


def numerics(m_dtype, dev):
    from insv2v import synth
    print(f"numerics: V, U fp16; M {'fp32' if m_dtype == torch.float32 else 'fp16'}; fp32 accumulation; reference fp32 F.conv2d on the same fp16 operands")
    for name, (N, C, O, H, W) in {"(92160,1280,11520)-like: 1280->1280 @ 8x12": (4, 1280, 1280, 8, 12),
                                   "(368640,640,5760)-like: 640->640 @ 16x24": (2, 640, 640, 16, 24),
                                   "(1474560,320,5760)-like: 640->320 @ 32x48": (1, 640, 320, 32, 48)}.items():
        # activations as the convolution sees them: GroupNorm + SiLU of a normal tensor; weights fan-in scaled (insv2v.synth)
        x = F.silu(synth.synth_input("wino.x." + name, (N, C, H, W))).half().to(dev)
        w = (synth.synth_input("wino.w." + name, (O, C, 3, 3)) * (9 * C) ** -0.5).half().to(dev)
        ref = F.conv2d(x.float(), w.float(), padding=1)
        direct = F.conv2d(x.float(), w.float(), padding=1).half().float()     # what a direct fp16-output convolution loses: output rounding only
        got = winograd_conv(x, w, m_dtype, dev).half().float()
        mx = ref.abs().max().item()
        for what, y in (("direct (fp16 output)", direct), ("winograd", got)):
            err = (y - ref).abs().max().item()
            rms = ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            print(f"  {name:44s} {what:22s} max err {err:.3e} = {err / mx:.2e} of max|ref| ({mx:.2f}), rel-RMS {rms:.2e}")


def timing():
    from insv2v import ops
    dev = "cuda:0"
    print("timing (B = 60 stack shapes): direct convolution vs the 16 GEMMs as one launch of 4 x M rows")
    for name, (NB, H, W, Cin, Cout) in {"conv (92160,1280,11520)": (960, 8, 12, 1280, 1280), "conv (368640,640,5760)": (960, 16, 24, 640, 640),
                                         "conv (92160,1280,23040)": (960, 8, 12, 2560, 1280), "conv (1474560,320,5760)": (960, 32, 48, 640, 320),
                                         "conv (23040,1280,11520)": (960, 4, 6, 1280, 1280), "conv (368640,640,11520)": (960, 16, 24, 1280, 640)}.items():
        M = NB * H * W
        x = torch.randn(M, Cin, device=dev).half()
        w = (torch.randn(Cout, 9 * Cin, device=dev) * (9 * Cin) ** -0.5).half()
        b = torch.zeros(Cout, device=dev)
        res = torch.randn(M, Cout, device=dev).half()
        v = torch.randn(4 * M, Cin, device=dev).half()
        u = (torch.randn(Cout, Cin, device=dev) * Cin ** -0.5).half()

        def t(fn, n=10):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        t_direct = t(lambda: ops.conv3x3(x, (NB, H, W), w, b, residual=res))
        t_gemm = t(lambda: ops.gemm(v, u))
        gb_in = 2.0 * M * Cin * (4 - 1) / 1e9            # V is 4 x the tensor the norm pass writes anyway: 3 x extra write
        gb_out = 2.0 * M * Cout * (4 + 1 + 1) / 1e9      # output transform: read M (4 x), read residual, write y
        t_xf = (gb_in + gb_out) / 5.0e3 * 1e6            # at 5 TB/s (what gn_apply reaches)
        print(f"  {name:26s} direct {t_direct:8.1f} us | grouped GEMM {t_gemm:8.1f} us ({2.0 * 4 * M * Cin * Cout / t_gemm * 1e-6:6.0f} TF/s) + transforms "
              f"{gb_in + gb_out:5.2f} GB ~ {t_xf:6.1f} us = {t_gemm + t_xf:8.1f} us  ({(t_gemm + t_xf) / t_direct:.2f} of direct)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--m-fp32", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    numerics(torch.float32 if a.m_fp32 else torch.float16, dev)
    if a.m_fp32 is False:
        numerics(torch.float32, dev)
    if a.time:
        timing()
