#!/usr/bin/env python3
"""Level-0 GEGLU feed-forward (C = 320): the fused register-resident kernel (insv2v_ffn_fused) vs the three-launch path
(row statistics + FF1/GEGLU GEMM + FF2/residual GEMM), same operands, interleaved rounds in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.fused import pack_ffn_stream  # noqa: E402
from insv2v.unet import fold_layernorm, interleave32  # noqa: E402

dev = torch.device("cuda:0")
C, NH = 320, 1280
g = torch.Generator().manual_seed(0)
w1, b1 = torch.randn(2 * NH, C, generator=g) * C ** -0.5, torch.randn(2 * NH, generator=g) * 0.3
w2, b2 = (torch.randn(C, NH, generator=g) * NH ** -0.5).half(), torch.randn(C, generator=g) * 0.3
wf, col, bf = fold_layernorm(w1, 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g), b1)
stream = pack_ffn_stream(wf.float(), bf, w2.float(), b2).to(dev)
w1i, b1i, csi, w2d, b2d = interleave32(wf).to(dev), interleave32(bf).to(dev), interleave32(col).to(dev), w2.to(dev), b2.to(dev)


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for M in (73728, 294912, 368640):
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).half().to(dev)
    out = torch.empty_like(x)

    def fused():
        ops.ffn_fused(x, stream, NH, out=out)

    def split():
        h = ops.gemm(x, w1i, b1i, act=ops.ACT_GEGLU, row_stats=ops.layernorm_stats(x), col_sum=csi)
        ops.gemm(h, w2d, b2d, residual=x)

    flops = 2.0 * M * C * 3 * NH
    for r in range(3):
        tf, ts = timeit(fused), timeit(split)
        print(f"M={M:7d} round {r}: fused {tf:8.1f} us = {flops / tf * 1e-6:7.1f} TF/s | 3 launches {ts:8.1f} us = {flops / ts * 1e-6:7.1f} TF/s", flush=True)
