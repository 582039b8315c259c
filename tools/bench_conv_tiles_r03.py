#!/usr/bin/env python3
"""Round 3: 3x3 convolutions at the stacked-clip sizes (B = 15 -> 240 images): patch-tiled kernel (auto) vs the persistent GEMM kernels
(tile 200 = 256x256 8-phase, 210 / 211 = two 4-wave workgroups per CU) vs the gathered 128x128 tile (5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch
from insv2v import ops, _lib
from insv2v.unet import prep_conv3x3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

NB = int(os.environ.get("NB", 240))
for name, h, w, cin, cout, res in (("L0 320->320 +res", 32, 48, 320, 320, True), ("L0 640->320", 32, 48, 640, 320, False), ("L1 640->640 +res", 16, 24, 640, 640, True),
                                   ("L1 1280->640", 16, 24, 1280, 640, False), ("L2 1280->1280 +res", 8, 12, 1280, 1280, True), ("L2 2560->1280", 8, 12, 2560, 1280, False),
                                   ("L3 1280->1280 +res", 4, 6, 1280, 1280, True)):
    M = NB * h * w
    x = torch.randn(M, cin, generator=g).half().to(dev)
    wk, bk = prep_conv3x3({"c.weight": torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5, "c.bias": torch.zeros(cout)}, "c", dev)
    r = torch.randn(M, cout, generator=g).half().to(dev) if res else None
    line = f"{name:20s} M={M:6d}:"
    ref = None
    tiles = [int(t) for t in os.environ.get("TILES", "0,5,200,210,211").split(",")]
    best = {}
    for rnd_ in range(3):                       # interleaved rounds: the minimum over rounds per tile (order / clock effects cancel)
        for tile in (tiles if rnd_ % 2 == 0 else tiles[::-1]):
            try:
                t = timeit(lambda: ops.conv3x3(x, (NB, h, w), wk, bk, residual=r, tile=tile))
                best[tile] = min(best.get(tile, 1e30), t)
                if rnd_ == 0:
                    out, _ = ops.conv3x3(x, (NB, h, w), wk, bk, residual=r, tile=tile)
                    if ref is None: ref = out
                    if (out.float() - ref.float()).abs().max().item() > 0.05: best[tile] = float("nan")
            except _lib.HipKernelError:
                best[tile] = None
    for tile in tiles:
        t = best[tile]
        line += f"  t{tile}: unsupported" if t is None else f"  t{tile}: {t:7.1f} us {2.0 * M * cout * 9 * cin / t * 1e-6:5.0f} TF"
    print(line, flush=True)
