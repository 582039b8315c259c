#!/usr/bin/env python3
"""External reference for the GEMM engine's ceiling: torch.matmul (= hipBLASLt / rocBLAS behind PyTorch-ROCm) on the
same box, same operands (uniform random [-1, 1), both matrices K-major: C = A x B^T) and sizes as tools/gemm_guide_8phase and
gemm_check --set big.  MEASUREMENT ONLY: nothing here is linked into or imported by the product.
usage: engine_ceiling_matmul.py [f16|bf16] [n] [seconds]"""
import sys
import time

import torch

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
g = torch.Generator(device="cuda").manual_seed(0)
a = (torch.rand(n, n, device="cuda", generator=g) * 2 - 1).to(dt)
b = (torch.rand(n, n, device="cuda", generator=g) * 2 - 1).to(dt)
c = torch.empty(n, n, device="cuda", dtype=dt)
for _ in range(5):
    torch.matmul(a, b.t(), out=c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
for p in range(2):
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b.t(), out=c)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    if p == 0:
        iters = max(10, int(seconds * 1e6 / us))
print(f"torch.matmul (hipBLASLt) {'bf16' if dt == torch.bfloat16 else 'f16'} n={n}: {us:.1f} us  {2.0 * n ** 3 / us * 1e-6:.1f} TFLOP/s  ({iters} launches)", flush=True)
