#!/usr/bin/env python3
"""The 96-token attention launches of the 8x12 level (d = 160) at 20 stacked clips: spatial self-attention and text cross-attention."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


BF, HW, heads, hd = 960, 96, 8, 160
C = heads * hd
qkv = torch.randn(BF * HW, 3 * C, device=dev).half()
out = torch.empty((BF * HW, C), device=dev, dtype=torch.float16)
p = qkv.data_ptr()
ms = timeit(lambda: ops.attention(p, p + 2 * C, p + 4 * C, out, batch=BF, heads=heads, head_dim=hd, seq_q=HW, seq_k=HW, scale=hd ** -0.5, q_rs=3 * C, k_rs=3 * C,
                                  v_rs=3 * C, o_rs=C, q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0), o_addr=(1, HW * C, 0)))
print(f"self  BF={BF} HW={HW} d={hd}: {ms * 1e3:8.1f} us  ({(qkv.numel() + out.numel()) * 2 / ms / 1e9:.2f} TB/s of q,k,v + out)")
kv = torch.randn(60 * 77, 2 * C, device=dev).half()
q = torch.randn(BF * HW, C, device=dev).half()
kp = kv.data_ptr()
ms = timeit(lambda: ops.attention(q.data_ptr(), kp, kp + 2 * C, out, batch=BF, heads=heads, head_dim=hd, seq_q=HW, seq_k=77, scale=hd ** -0.5, q_rs=C, k_rs=2 * C,
                                  v_rs=2 * C, o_rs=C, q_addr=(1, HW * C, 0), kv_addr=(16, 77 * 2 * C, 0), o_addr=(1, HW * C, 0)))
print(f"cross BF={BF} HW={HW} d={hd}: {ms * 1e3:8.1f} us  ({(q.numel() + out.numel()) * 2 / ms / 1e9:.2f} TB/s of q + out)")
