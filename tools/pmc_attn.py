#!/usr/bin/env python3
"""The level-0 spatial self-attention launch (d = 40, 1536 tokens, 48 frames x 8 heads) for rocprofv3 --pmc passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402

dev = torch.device("cuda:0")
for BF, HW, heads, hd in ((48, 1536, 8, 40), (48, 384, 8, 80)):
    C = heads * hd
    qkv = torch.randn(BF * HW, 3 * C, device=dev).half()
    out = torch.empty((BF * HW, C), device=dev, dtype=torch.float16)
    p = qkv.data_ptr()
    for _ in range(4):
        ops.attention(p, p + 2 * C, p + 4 * C, out, batch=BF, heads=heads, head_dim=hd, seq_q=HW, seq_k=HW, scale=hd ** -0.5, q_rs=3 * C, k_rs=3 * C,
                      v_rs=3 * C, o_rs=C, q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0), o_addr=(1, HW * C, 0))
torch.cuda.synchronize()
