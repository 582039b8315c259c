#!/usr/bin/env python3
"""The per-frame GroupNorm in front of the C = 1280 spatial transformers at the B = 60 stack (960 frames x 96 / 24 tokens x 1280 channels):
time per launch; INSV2V_GN_FRAME_SPLIT = workgroups per sample (channel split of gn_frame_kernel; 1 = one workgroup per sample)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402

dev = torch.device("cuda:0")
for ns, rows, C in ((960, 96, 1280), (960, 24, 1280), (960, 96, 2560)):
    x = torch.randn(ns * rows, C, device=dev).half()
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    f = lambda: ops.groupnorm(x, ns, rows, g, b, 32, 1e-6)
    y = f()
    xf = x.float().reshape(ns, rows, 32, C // 32)
    ref = ((xf - xf.mean((1, 3), keepdim=True)) * torch.rsqrt(xf.var((1, 3), unbiased=False, keepdim=True) + 1e-6)).reshape(ns * rows, C) * g + b
    err = (y.float() - ref).abs().max().item()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"gn ({ns}, {rows}, {C}): {best:7.1f} us  {x.numel() * 4 / best / 1e6:.2f} TB/s (one read + one write)  max |err| vs fp32 {err:.2e}")
