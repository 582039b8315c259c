#!/usr/bin/env python3
"""Go / no-go for Winograd output tiles larger than 2x2 on the product's 3x3 convolutions (F(3x3, 3x3): 3.24x fewer MACs,
F(4x4, 3x3): 4x fewer, against the 2.25x of the F(2x2, 3x3) form the product runs; resnet.py:143,159).

MEASUREMENT / ANALYSIS ONLY (CPU, torch) - nothing here is part of the product.  The algorithm is run with the storage the product uses
- activations, V = B^T d B, U = G g G^T and the product matrices M in fp16, fp32 accumulation inside each GEMM, fp32 transform arithmetic -
against fp32 F.conv2d on the same fp16-rounded operands.  Transform matrices come from the Cook-Toom construction over the given
interpolation points (+ infinity); B^T is solved for numerically so that A^T [(G g) . (B^T d)] is exact in fp64.  The bar is the
single-kernel tolerance of this build: 2e-3 of max|ref| (tests/test_kernels_gpu.py::test_winograd_conv3x3_vs_fp32).
usage: winograd_larger_tiles.py [threads]"""
import os
import sys
from fractions import Fraction as Fr

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]


def cook_toom(points, m, r=3):
    """(A^T [m, a], G [a, r], B^T [a, a]) of F(m, r) over a - 1 = m + r - 2 finite points and the point at infinity."""
    a = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == a - 1
    at = [[(pts[j] ** i if j < a - 1 else int(i == m - 1)) for j in range(a)] for i in range(m)]

    def norm(j):
        v = Fr(1)
        for l in range(a - 1):
            if l != j:
                v *= pts[j] - pts[l]
        return v
    g = [[(pts[j] ** k / norm(j) if j < a - 1 else int(k == r - 1)) for k in range(r)] for j in range(a)]
    atf, gf = np.array(at, dtype=np.float64), np.array(g, dtype=np.float64)
    lhs = np.array([[atf[i, j] * gf[j, k] for j in range(a)] for i in range(m) for k in range(r)])
    bt = np.zeros((a, a))
    for n in range(a):   # column n of B^T: sum_j A^T[i, j] G[j, k] B^T[j, n] = [n == i + k]
        bt[:, n] = np.linalg.lstsq(lhs, np.array([float(n == i + k) for i in range(m) for k in range(r)]), rcond=None)[0]
    return tuple(torch.tensor(x, dtype=torch.float32) for x in (atf, gf, bt))


def winograd(x, w, at, g, bt, m, m_dtype=torch.float16):
    n, c, h, wd = x.shape
    a = m + 2
    d = F.pad(x.float(), (1, 1, 1, 1)).unfold(2, a, m).unfold(3, a, m)
    v = torch.einsum("ai,nchwij,bj->nchwab", bt, d, bt).half()
    u = torch.einsum("ai,ocij,bj->ocab", g, w.float(), g).half()
    mm = torch.einsum("nchwab,ocab->nohwab", v.float(), u.float()).to(m_dtype)
    y = torch.einsum("ia,nohwab,jb->nohwij", at, mm.float(), at)
    return y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], h, wd), v.abs().max().item()


def main():
    from insv2v import synth
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 16)
    name = "(92160,1280,11520)-like: 1280->1280 @ 8x12"
    w = (synth.synth_input("wino.w." + name, (320, 1280, 3, 3)) * (9 * 1280) ** -0.5).half()   # 320 of the layer's 1280 output channels
    for m, shape, point_sets in ((2, (2, 1280, 8, 12), ([0, 1, -1],)),
                                 (3, (2, 1280, 9, 12), ([0, 1, -1, 2], [0, 1, -1, "1/2"], [0, 1, -1, "-1/2"])),
                                 (4, (2, 1280, 8, 12), ([0, 1, -1, 2, -2], [0, 1, -1, "1/2", "-1/2"], [0, 1, -1, "1/2", -2], [0, 1, -1, 2, "-1/2"]))):
        x = F.silu(synth.synth_input("wino.x." + name, shape)).half()   # activations as the convolution sees them: SiLU of a normalised tensor
        ref = F.conv2d(x.float(), w.float(), padding=1)
        mx = ref.abs().max().item()

        def report(tag, y):
            y = y.half().float()
            err = (y - ref).abs().max().item()
            rms = ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
            print(f"  {tag:58s} max err {err:.3e} = {err / mx:.2e} of max|ref|  rel-RMS {rms:.2e}  {'ok' if err / mx < 2e-3 else 'OUT of 2e-3'}")
        print(f"F({m}x{m}, 3x3), {9 * m * m / (m + 2) ** 2:.2f}x fewer MACs, V = {(m + 2) ** 2 / m ** 2:.2f}x the input bytes; x {shape}")
        report("direct convolution (fp16 output rounding only)", ref)
        for pts in point_sets:
            at, g, bt = cook_toom(pts, m)
            for md, tag in ((torch.float16, "M fp16"), (torch.float32, "M fp32")):
                y, vmax = winograd(x, w, at, g, bt, m, md)
                report(f"points {pts} + inf, {tag} (max|V| {vmax:.0f})", y)


if __name__ == "__main__":
    main()
