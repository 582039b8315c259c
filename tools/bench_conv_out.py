#!/usr/bin/env python3
"""The UNet's conv_out (320 -> 4 channels, 3x3) at the stacked clip counts: implicit-GEMM form (ops.conv3x3) vs ops.conv3x3_narrow
(one GEMM over the input channels for the nine taps' partial outputs + insv2v_tap_gather)."""
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


cin, cout, H, W = 320, 4, 32, 48
w = (torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5).half()
b = (torch.randn(cout) * 0.3).to(dev)
wk = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(dev)
wt = ops.tap_weights(w).to(dev)
for nb in (3, 6, 12, 30, 60):
    NB = nb * 16
    x = torch.randn(NB * H * W, cin, device=dev).half()
    f1 = lambda: ops.conv3x3(x, (NB, H, W), wk, b, out_fp32=True)[0]
    f2 = lambda: ops.conv3x3_narrow(x, (NB, H, W), wt, b, cout)
    err = (f1() - f2()).abs().max().item()
    t1, t2 = min(timeit(f1), timeit(f1)), min(timeit(f2), timeit(f2))
    print(f"B={nb:2d} M={NB * H * W:8d}: implicit GEMM {t1:7.1f} us   narrow (GEMM + gather) {t2:7.1f} us   max |diff| {err:.2e}")
    del x
