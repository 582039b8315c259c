#!/usr/bin/env python3
"""K = 640 Linear layers of UNet level 1 at the B = 60 stack (M = 368 640): insv2v_rowlin (register-resident) vs insv2v_gemm (the 256 x 320
ping-pong engine), interleaved in one process - the forms the transformer blocks issue: to_out + residual (with / without LayerNorm
statistics of the output), plain."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.fused import pack_linear_stream  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M, K, N = int(os.environ.get("M", 368640)), 640, 640
x = (torch.randn(M, K, generator=g) * 1.3 + 0.2).half().to(dev)
w, b = (torch.randn(N, K, generator=g) * K ** -0.5).half(), torch.randn(N, generator=g) * 0.3
st = pack_linear_stream(w.float(), b).to(dev)
wd, bd = w.to(dev), b.to(dev)
r = torch.randn(M, N, generator=g).half().to(dev)
out = torch.empty((M, N), device=dev, dtype=torch.float16)
flops = 2.0 * M * N * K
for name, res, stats in (("to_out + residual", True, False), ("to_out + residual + LN statistics of the output", True, True), ("plain", False, False)):
    new = lambda: ops.rowlin(x, st, N, residual=r if res else None, emit_stats=stats) if stats else ops.rowlin(x, st, N, residual=r if res else None, out=out)
    old = lambda: ops.gemm(x, wd, bd, residual=r if res else None, emit_stats=stats) if stats else ops.gemm(x, wd, bd, residual=r if res else None, out=out)
    for rd in range(2):
        tn, to = timeit(new), timeit(old)
        print(f"M={M} K={K} N={N} {name:50s} round {rd}: rowlin {tn:7.1f} us = {flops / tn * 1e-6:6.1f} TF/s | gemm {to:7.1f} us = {flops / to * 1e-6:6.1f} TF/s", flush=True)
