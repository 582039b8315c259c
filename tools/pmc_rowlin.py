#!/usr/bin/env python3
"""A few launches of the row Linear at the B = 60 shapes (and nothing else) for rocprofv3 --pmc passes: CASE = qkv320 | res640 | gn320."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.fused import pack_linear_stream  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
case = os.environ.get("CASE", "qkv320")
B = 60
M0, M1 = B * 16 * 1536, B * 16 * 384
if case == "qkv320":
    x = (R(M0, 320) * 1.3 + 0.2).half().to(dev)
    st = pack_linear_stream(R(960, 320, scale=320 ** -0.5).half().float(), R(960) * 0.3).to(dev)
    fn = lambda: ops.rowlin(x, st, 960, layernorm=True)
elif case == "gn320":
    x = (R(M0, 320) * 1.3 + 0.2).half().to(dev)
    st = pack_linear_stream(R(320, 320, scale=320 ** -0.5).half().float(), R(320) * 0.3).to(dev)
    ab = (R(B * 16, 320, 2) * 0.5 + 1.0).float().to(dev)
    fn = lambda: ops.rowlin(x, st, 320, gn_ab=ab, gn_rows=1536)
else:
    x = (R(M1, 640) * 1.3).half().to(dev)
    r = R(M1, 640).half().to(dev)
    st = pack_linear_stream(R(640, 640, scale=640 ** -0.5).half().float(), R(640) * 0.3).to(dev)
    fn = lambda: ops.rowlin(x, st, 640, residual=r, emit_stats=True)
for _ in range(6):
    fn()
torch.cuda.synchronize()
