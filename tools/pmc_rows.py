#!/usr/bin/env python3
"""Representative launches of the round-3 kernels and of the round-2 kernels VERDICT asked counters for, for rocprofv3 --pmc passes
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, LDS bank conflicts): fused feed-forward, row-linear (K = 320 / 640), fused
temporal attention, the persistent GEMMs (p8: GEGLU FF1 K = 640; w4: q/k/v N = 1920) and the patch-tiled convolution."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.fused import pack_ffn_stream, pack_linear_stream, pack_tattn_stream, pack_tattn_qkv_stream, pack_xattn_stream, pack_xattn_kv  # noqa: E402
from insv2v.unet import prep_conv3x3, fold_layernorm, interleave32  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
M, C, NH = 294912, 320, 1280
x = (R(M, C) * 1.3 + 0.2).half().to(dev)
wf, col, bf = fold_layernorm(R(2 * NH, C, scale=C ** -0.5), 1 + 0.1 * R(C), 0.1 * R(C), R(2 * NH) * 0.3)
ffn = pack_ffn_stream(wf.float(), bf, R(C, NH, scale=NH ** -0.5).half().float(), R(C) * 0.3).to(dev)
lin = pack_linear_stream(R(960, C, scale=C ** -0.5).half().float(), R(960) * 0.3).to(dev)
linr = pack_linear_stream(R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3).to(dev)
ta = pack_tattn_stream(R(3 * C, C, scale=C ** -0.5).half().float(), R(16, 3 * C) * 0.3, R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3).to(dev)
xa = pack_xattn_stream(R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3, R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3).to(dev)
xkv = pack_xattn_kv((R(12 * 77, 2 * C) * 1.5).half().to(dev), 12, 77, C, 8)
x6 = (R(73728, 640) * 1.3).half().to(dev)
ta6 = pack_tattn_qkv_stream(R(1920, 640, scale=640 ** -0.5).half().float(), R(16, 1920) * 0.3).to(dev)
lin6 = pack_linear_stream(R(1920, 640, scale=640 ** -0.5).half().float(), R(1920) * 0.3).to(dev)
w1 = R(5120, 640, scale=640 ** -0.5).half().to(dev); b1 = R(5120).to(dev)
wq = R(1920, 640, scale=640 ** -0.5).half().to(dev)
xc = (R(48 * 32 * 48, 320)).half().to(dev)
wk, bk = prep_conv3x3({"c.weight": R(320, 320, 3, 3, scale=(9 * 320) ** -0.5), "c.bias": torch.zeros(320)}, "c", dev)
for _ in range(3):
    ops.ffn_fused(x, ffn, NH)
    ops.rowlin(x, lin, 960, layernorm=True)
    ops.rowlin(x, linr, C, residual=x)
    ops.tattn_fused(x, ta, 12, 1536, 8, 16)
    ops.xattn_fused(x, xa, xkv, 16 * 1536, 8, 77)
    ops.tattn_attn(x6, ta6, 12, 384, 8, 16)
    ops.rowlin(x6, lin6, 1920, layernorm=True)
    ops.gemm(x6, w1, b1, act=ops.ACT_GEGLU)          # gemm_p8
    ops.gemm(x6, wq)                                 # gemm_w4
    ops.conv3x3(xc, (48, 32, 48), wk, bk)            # conv_halo
# ---- round 4: the 8-wave ping-pong kernels at the benched (B = 30) token counts
M0, M1, M2 = 737280, 184320, 46080
xc0 = R(480 * 32 * 48, 320).half().to(dev)                                   # conv L0 320 -> 320: gemm_r8 conv
xl1 = (R(M1, 640) * 1.3).half().to(dev)
w2 = R(640, 2560, scale=2560 ** -0.5).half().to(dev); h1 = R(M1, 2560).half().to(dev); r1 = R(M1, 640).half().to(dev)
xl2 = R(M2, 1280).half().to(dev)
w1l2 = R(10240, 1280, scale=1280 ** -0.5).half().to(dev); b1l2 = R(10240).to(dev)
wq2 = R(3840, 1280, scale=1280 ** -0.5).half().to(dev)
xc2 = R(480 * 8 * 12, 1280).half().to(dev)
wk2, bk2 = prep_conv3x3({"c.weight": R(1280, 1280, 3, 3, scale=(9 * 1280) ** -0.5), "c.bias": torch.zeros(1280)}, "c", dev)
big = R(8192, 8192).half().to(dev)
for _ in range(3):
    ops.conv3x3(xc0, (480, 32, 48), wk, bk)                       # gemm_r8 CONV (N = 320, one column tile)
    ops.conv3x3(xc0, (480, 32, 48), wk, bk, residual=xc0)         # gemm_r8 CONV + residual
    ops.gemm(h1, w2, None, residual=r1)                           # gemm_r8 LINEAR + residual (FF2 of level 1)
    ops.gemm(xl1, w1, b1, act=ops.ACT_GEGLU)                      # gemm_q8 GEGLU (FF1 of level 1 at M = 184 320)
    ops.gemm(xl2, w1l2, b1l2, act=ops.ACT_GEGLU)                  # gemm_q8 GEGLU (FF1 of level 2)
    ops.gemm(xl2, wq2)                                            # q/k/v of level 2: gemm_r8 / gemm_q8 by the tile-count rule
    ops.conv3x3(xc2, (480, 8, 12), wk2, bk2)                      # conv level 2 (N = 1280)
    ops.gemm(big, big, tile=230)                                  # gemm_q8 at 8192^3
    ops.gemm(big, big, tile=230)                                  # gemm_q8 at 8192^3
# ---- round 6: a Winograd convolution (input transform, grouped GEMM on the 16x16x32 engine, output transform) and the upsample form
U2 = ops.winograd_weights(R(1280, 1280, 3, 3, scale=(9 * 1280) ** -0.5), dev)
Uu = ops.winograd_weights(R(1280, 1280, 3, 3, scale=(9 * 1280) ** -0.5), dev, upsample=True)
ab = torch.stack([1.0 + 0.1 * R(30, 1280), 0.1 * R(30, 1280)], -1).contiguous().to(dev)
for _ in range(3):
    ops.winograd_conv3x3(xc2, (480, 8, 12), U2, bk2, gn_ab=ab, gn_images_per_sample=16, gn_silu=True, residual=xc2)
    ops.winograd_conv3x3(xc2, (480, 8, 12), Uu, bk2, upsample=True)
torch.cuda.synchronize()
