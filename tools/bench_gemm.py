#!/usr/bin/env python3
"""Micro-benchmark of insv2v_gemm on the UNet's shapes: every tile config, HIP-event timed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.unet import prep_conv3x3  # noqa: E402

dev = torch.device("cuda:0")
LIN = [(73728, 320, 320, 0, True), (73728, 2560, 320, 2, False), (73728, 960, 320, 0, False), (73728, 320, 1280, 0, True),
       (18432, 640, 640, 0, True), (18432, 5120, 640, 2, False), (18432, 640, 2560, 0, True), (4608, 1280, 1280, 0, True),
       (4608, 10240, 1280, 2, False), (4608, 1280, 5120, 0, True), (1152, 1280, 1280, 0, True), (1152, 3840, 1280, 0, False),
       (8192, 8192, 8192, 0, False)]
CONV = [(48, 32, 48, 320, 320), (48, 32, 48, 640, 320), (48, 16, 24, 640, 640), (48, 8, 12, 1280, 1280), (48, 4, 6, 1280, 1280),
        (48, 4, 6, 2560, 1280)]


TILES = [int(t) for t in os.environ.get("TILES", "11,21,31,14,24,34").split(",")]
print("tile codes (pipe*10+shape; pipe 1=reg-staged 2=DMA2 3=DMA3; shape 1=128x128 4=64x64):", TILES)


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M, N, K, act, res in LIN:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N // 2 if act == 2 else N, device=dev).half() if res else None
    out = []
    for tile in TILES:
        if act == 2 and tile % 10 in (3, 4, 7, 9):
            out.append("       -        ")
            continue
        ms = timeit(lambda: ops.gemm(a, w, b, act=act, residual=r, tile=tile))
        out.append(f"{ms * 1e3:7.1f}us {2.0 * M * N * K / ms / 1e9:6.0f}TF")
    print(f"lin  M={M:6d} N={N:5d} K={K:5d} act={act} res={int(res)} | " + " | ".join(out), flush=True)
for nb, h, w_, cin, cout in CONV:
    x = torch.randn(nb * h * w_, cin, device=dev).half()
    wt = torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5
    wk, bk = prep_conv3x3({"c.weight": wt, "c.bias": torch.zeros(cout)}, "c", dev)
    out = []
    for tile in TILES:
        ms = timeit(lambda: ops.conv3x3(x, (nb, h, w_), wk, bk, tile=tile))
        out.append(f"{ms * 1e3:7.1f}us {2.0 * nb * h * w_ * cout * 9 * cin / ms / 1e9:6.0f}TF")
    print(f"conv M={nb * h * w_:6d} N={cout:5d} K={9 * cin:5d}            | " + " | ".join(out), flush=True)
for rows, C in ((73728, 320), (18432, 640), (4608, 1280)):
    x = torch.randn(rows, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.layernorm(x, g, b))
    print(f"layernorm {rows}x{C}: {ms * 1e3:.1f} us  {rows * C * 4 / ms / 1e9:.2f} TB/s")
if os.environ.get("HALO"):
    print("halo (tile 100) vs auto-gathered (tile 8) conv:")
    for nb, h, w_, cin, cout, ups in [(48, 32, 48, 320, 320, False), (48, 32, 48, 640, 320, False), (48, 32, 48, 960, 320, False),
                                      (48, 16, 24, 640, 640, False), (48, 16, 24, 1280, 640, False), (48, 16, 24, 640, 640, True),
                                      (16, 256, 384, 128, 128, False), (16, 128, 192, 256, 256, False)]:
        x = torch.randn(nb * h * w_, cin, device=dev).half()
        wt = torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5
        wk, bk = prep_conv3x3({"c.weight": wt, "c.bias": torch.zeros(cout)}, "c", dev)
        up = 4 if ups else 1
        fl = 2.0 * nb * h * w_ * up * cout * 9 * cin
        res = []
        for tile in (5, 100, 5, 100):
            ms = timeit(lambda: ops.conv3x3(x, (nb, h, w_), wk, bk, upsample=ups, tile=tile))
            res.append(f"{ms * 1e3:7.1f}us {fl / ms / 1e9:6.0f}TF")
        print(f"conv nb={nb} {h}x{w_} cin={cin} cout={cout} up={int(ups)} | " + " | ".join(res), flush=True)
