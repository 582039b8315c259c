#!/usr/bin/env python3
"""Spatial self-attention launches of the UNet at the B = 60 stack (levels 0-2), for A/Bs of attention-kernel changes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for BF, HW, heads, hd in ((960, 1536, 8, 40), (960, 384, 8, 80), (960, 96, 8, 160), (960, 24, 8, 160)):
    C = heads * hd
    qkv = torch.randn(BF * HW, 3 * C, device=dev).half()
    out = torch.empty((BF * HW, C), device=dev, dtype=torch.float16)
    p = qkv.data_ptr()
    f = lambda: ops.attention(p, p + 2 * C, p + 4 * C, out, batch=BF, heads=heads, head_dim=hd, seq_q=HW, seq_k=HW, scale=hd ** -0.5, q_rs=3 * C, k_rs=3 * C,
                              v_rs=3 * C, o_rs=C, q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0), o_addr=(1, HW * C, 0))
    us = min(timeit(f), timeit(f))
    print(f"self BF={BF} HW={HW} d={hd}: {us:8.1f} us  {4.0 * BF * heads * HW * HW * hd / us / 1e6:7.1f} TF/s  {(qkv.numel() + out.numel()) * 2 / us / 1e6:.2f} TB/s")
    del qkv, out
