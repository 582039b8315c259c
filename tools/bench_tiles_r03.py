#!/usr/bin/env python3
"""Round 3: at the stacked-clip token counts (B = 15), do the persistent GEMMs (tile 200 = 256x256 8-phase, 210 / 211 = two 4-wave
workgroups per CU) beat the 128x128 tile on the linear shapes that still run on it (residual GEMMs with long K, level-2 q/k/v)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch
from insv2v import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

GEGLU = [("FF1 L1 geglu", 92160, 5120, 640), ("FF1 L2 geglu", 23040, 10240, 1280), ("FF1 L3 geglu", 5760, 10240, 1280)]
if os.environ.get("GEGLU"):
    from insv2v.unet import interleave32
    for name, M, N, K in GEGLU:
        M *= int(os.environ.get("MSCALE", 1))
        a = torch.randn(M, K, generator=g).half().to(dev)
        w, b = interleave32((torch.randn(N, K, generator=g) * K ** -0.5)).half().to(dev), interleave32(torch.randn(N, generator=g)).to(dev)
        st, cs = ops.layernorm_stats(a), w.float().sum(1).contiguous()
        line = f"{name:16s} {M:6d}x{N:5d}x{K:4d}:"
        best = {}
        for rd in range(3):
            for tile in ((0, 5, 200, 210) if rd % 2 == 0 else (210, 200, 5, 0)):
                try:
                    best[tile] = min(best.get(tile, 1e30), timeit(lambda: ops.gemm(a, w, b, act=ops.ACT_GEGLU, row_stats=st, col_sum=cs, tile=tile)))
                except _lib.HipKernelError:
                    best[tile] = None
        for tile in (0, 5, 200, 210):
            line += f"  t{tile}: unsupported" if best[tile] is None else f"  t{tile}: {best[tile]:7.1f} us {2.0 * M * N * K / best[tile] * 1e-6:6.0f} TF"
        print(line, flush=True)
    sys.exit(0)
shapes = [("FF2 L1", 92160, 640, 2560, True, False), ("FF2 L2", 23040, 1280, 5120, True, False), ("N=C L2 +res", 23040, 1280, 1280, True, False),
          ("N=C L2", 23040, 1280, 1280, False, False), ("qkv L2 (LN)", 23040, 3840, 1280, False, True), ("FF2 L3", 5760, 1280, 5120, True, False),
          ("N=C L3 +res", 5760, 1280, 1280, True, False), ("qkv L3 (LN)", 5760, 3840, 1280, False, True), ("shortcut L0 cat", 368640, 320, 960, False, False),
          ("FF2 L1 B=3", 18432, 640, 2560, True, False), ("FF2 L2 B=3", 4608, 1280, 5120, True, False)]
MS = int(os.environ.get("MSCALE", 1))   # 2 = the token counts of 10 stacked clips (B = 30)
for name, M, N, K, res, ln in shapes:
    M *= MS
    a = (torch.randn(M, K, generator=g)).half().to(dev)
    w, b = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev), torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).half().to(dev) if res else None
    kw = {}
    if ln:
        kw = dict(row_stats=ops.layernorm_stats(a), col_sum=w.float().sum(1).contiguous())
    out = torch.empty((M, N), device=dev, dtype=torch.float16)
    line = f"{name:16s} {M:6d}x{N:4d}x{K:4d}:"
    ref = None
    for tile in (0, 5, 200, 210, 211):
        try:
            t = timeit(lambda: ops.gemm(a, w, b, residual=r, out=out, tile=tile, **kw))
            if ref is None: ref = out.clone()
            err = (out.float() - ref.float()).abs().max().item()
            line += f"  t{tile}: {t:7.1f} us {2.0 * M * N * K / t * 1e-6:6.0f} TF" + ("" if err < 0.05 else f" ERR {err:.2g}")
        except _lib.HipKernelError:
            line += f"  t{tile}: unsupported"
    print(line, flush=True)
