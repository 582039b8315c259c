#!/usr/bin/env python3
"""The temporal self-attention launches of the motion modules at the B = 60 stack (and nothing else), for rocprofv3 --pmc passes and for
timing: CASE = t1280 (levels 2: 5760 pixels x 8 heads x d 160 x 16 frames, attn_short with the positional-encoding bias), t1280s (level 3:
1440 pixels), c5_320 (C5's level 0: 55296 pixels x 8 x 40 x 24 frames, generic kernel).  Addressing exactly as unet.py issues it: rows
(sample, frame, pixel) of the fused [tokens, 3 C] q/k/v tensor, one problem per (sample, pixel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402

dev = torch.device("cuda:0")
CASES = {"t1280": (60, 96, 16, 8, 160, True), "t1280s": (60, 24, 16, 8, 160, True), "c5_320": (18, 3072, 24, 8, 40, False)}
for case in os.environ.get("CASE", "t1280").split(","):
    B, HW, F, heads, hd, bias = CASES[case]
    C = heads * hd
    qkv = torch.randn(B * F * HW, 3 * C, device=dev).half()
    out = torch.empty((B * F * HW, C), device=dev, dtype=torch.float16)
    pe = (0.1 * torch.randn(F, 3 * C, device=dev)).half() if (bias and not os.environ.get("NOBIAS")) else None
    p = qkv.data_ptr()
    addr = (HW, F * HW * 3 * C, 3 * C)

    def run():
        ops.attention(p, p + 2 * C, p + 4 * C, out, batch=B * HW, heads=heads, head_dim=hd, seq_q=F, seq_k=F, scale=hd ** -0.5, q_rs=HW * 3 * C,
                      k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, q_addr=addr, kv_addr=addr, o_addr=(HW, F * HW * C, C), qkv_bias=pe)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(os.environ.get("ITERS", "5"))
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byt = (qkv.numel() + out.numel()) * 2
    print(f"{case}: B={B} HW={HW} F={F} heads={heads} d={hd}: {us:8.1f} us  {byt / 1e6:.0f} MB algorithmic (q, k, v in, o out) = {byt / us / 1e6:.2f} TB/s")
