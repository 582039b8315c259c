#!/usr/bin/env python3
"""A few launches of the fused feed-forward (and nothing else) for rocprofv3 --pmc passes; INSV2V_FFN_DBG selects the variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.fused import pack_ffn_stream  # noqa: E402
from insv2v.unet import fold_layernorm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
M, C, NH = int(os.environ.get("M", 294912)), 320, 1280
x = (R(M, C) * 1.3 + 0.2).half().to(dev)
wf, col, bf = fold_layernorm(R(2 * NH, C, scale=C ** -0.5), 1 + 0.1 * R(C), 0.1 * R(C), R(2 * NH) * 0.3)
ffn = pack_ffn_stream(wf.float(), bf, R(C, NH, scale=NH ** -0.5).half().float(), R(C) * 0.3).to(dev)
for _ in range(6):
    ops.ffn_fused(x, ffn, NH)
torch.cuda.synchronize()
