#!/usr/bin/env python3
"""A few representative insv2v_gemm launches for rocprofv3 --pmc passes (see profiles/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.unet import prep_conv3x3  # noqa: E402

dev = torch.device("cuda:0")
tile = int(os.environ.get("TILE", "0"))
for M, N, K, act in ((8192, 8192, 8192, 0), (73728, 2560, 320, 2), (18432, 640, 2560, 0)):
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev)
    for _ in range(3):
        ops.gemm(a, w, b, act=act, tile=tile)
for nb, h, w_, cin, cout in ((48, 16, 24, 640, 640), (48, 32, 48, 320, 320), (48, 8, 12, 1280, 1280)):
    x = torch.randn(nb * h * w_, cin, device=dev).half()
    wk, bk = prep_conv3x3({"c.weight": torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, "c.bias": torch.zeros(cout)}, "c", dev)
    for _ in range(3):
        ops.conv3x3(x, (nb, h, w_), wk, bk, tile=tile)
torch.cuda.synchronize()
