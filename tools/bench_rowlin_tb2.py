#!/usr/bin/env python3
"""K = 640 row Linears: one vs two 32-token blocks per wave (INSV2V_ROWLIN_TB2 = bit mask over LN << 2 | FRAME << 1 | RES, read once per
process: run this script once per mask).  Prints one line per form of the level-1 transformer blocks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch
from insv2v import ops
from insv2v.fused import pack_linear_stream
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
K, HW, F_ = 640, 384, 16

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print("INSV2V_ROWLIN_TB2 =", os.environ.get("INSV2V_ROWLIN_TB2", "(default)"))
for samples in (15, 30):
    M = samples * F_ * HW
    x = (torch.randn(M, K, generator=g) * 1.3 + 0.2).half().to(dev)
    for N, ln, frame, res, name in ((K, False, False, False, "proj_in"), (K, False, False, True, "out-proj + residual"), (K, True, False, False, "q (LN)"),
                                    (3 * K, True, False, False, "q/k/v (LN)"), (3 * K, True, True, False, "temporal q/k/v (LN, frame bias)")):
        w, b = (torch.randn(N, K, generator=g) * K ** -0.5).half(), torch.randn(N, generator=g) * 0.3
        table = torch.randn(F_, N, generator=g) * 0.3
        st = (pack_linear_stream(w.float(), None, table) if frame else pack_linear_stream(w.float(), b)).to(dev)
        r = torch.randn(M, N, generator=g).half().to(dev) if res else None
        out = torch.empty((M, N), device=dev, dtype=torch.float16)
        fn = lambda: ops.rowlin(x, st, N, layernorm=ln, residual=r, out=out, frames=F_ if frame else 0, rows_per_frame=HW if frame else 0)
        t = min(timeit(fn) for _ in range(3))
        print(f"M={M:7d} N={N:4d} {name:34s} {t:7.1f} us = {2.0 * M * N * K / t * 1e-6:6.1f} TF/s", flush=True)
