#!/usr/bin/env python3
"""K = 320 Linear layers of UNet level 0: insv2v_rowlin (register-resident) vs insv2v_gemm, interleaved rounds in one process."""
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


for M, K in ((73728, 320), (294912, 320), (18432, 640), (73728, 640)):
    x = (torch.randn(M, K, generator=g) * 1.3 + 0.2).half().to(dev)
    for N, ln, res, name in ((K, False, True, "out-proj + residual"), (K, False, False, "proj_in"), (K, True, False, "q (LayerNorm)"),
                             (3 * K, True, False, "q/k/v (LayerNorm)")):
        w, b = (torch.randn(N, K, generator=g) * K ** -0.5).half(), torch.randn(N, generator=g) * 0.3
        st = pack_linear_stream(w.float(), b).to(dev)
        wd, bd, cs = w.to(dev), b.to(dev), w.float().sum(1).to(dev)
        r = (torch.randn(M, N, generator=g)).half().to(dev) if res else None
        out = torch.empty((M, N), device=dev, dtype=torch.float16)

        def new():
            ops.rowlin(x, st, N, layernorm=ln, residual=r, out=out)

        def old():
            if ln:
                ops.gemm(x, wd, bd, row_stats=ops.layernorm_stats(x), col_sum=cs, out=out)
            else:
                ops.gemm(x, wd, bd, residual=r, out=out)

        flops = 2.0 * M * N * K
        for rd in range(2):
            tn, to = timeit(new), timeit(old)
            print(f"M={M:7d} K={K} N={N:4d} {name:22s} round {rd}: rowlin {tn:7.1f} us = {flops / tn * 1e-6:6.1f} TF/s | gemm{' + ln_stats' if ln else ''} {to:7.1f} us = {flops / to * 1e-6:6.1f} TF/s", flush=True)
