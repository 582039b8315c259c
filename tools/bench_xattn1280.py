#!/usr/bin/env python3
"""Text cross-attention of the C = 1280 levels at the B = 60 stack: one problem per (sample, frame) (96 / 24 queries, the form unet.py issued through
round 6) against one problem per SAMPLE (the frames' queries are consecutive rows and share the sample's text K / V: 1 536 / 384 queries)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, F, heads, hd, L = int(os.environ.get("NB", "60")), 16, 8, 160, 77
C = heads * hd
for HW in (96, 24):
    q = torch.randn(B * F * HW, C, device=dev).half()
    kv = torch.randn(B * L, 2 * C, device=dev).half()
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    kp = kv.data_ptr()
    kw = dict(heads=heads, head_dim=hd, seq_k=L, scale=hd ** -0.5, q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C)
    f1 = lambda: ops.attention(q.data_ptr(), kp, kp + 2 * C, o1, batch=B * F, seq_q=HW, q_addr=(1, HW * C, 0), kv_addr=(F, L * 2 * C, 0), o_addr=(1, HW * C, 0), **kw)
    f2 = lambda: ops.attention(q.data_ptr(), kp, kp + 2 * C, o2, batch=B, seq_q=F * HW, q_addr=(1, F * HW * C, 0), kv_addr=(1, L * 2 * C, 0), o_addr=(1, F * HW * C, 0), **kw)
    t1, t2 = timeit(f1), timeit(f2)
    t1b, t2b = timeit(f1), timeit(f2)
    err = (o1.float() - o2.float()).abs().max().item()
    print(f"HW={HW}: per (sample, frame) {min(t1, t1b):7.1f} us   per sample {min(t2, t2b):7.1f} us   max |diff| {err:.2e}  ({q.numel() * 4 / 1e6:.0f} MB q + o)")
