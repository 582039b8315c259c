#!/usr/bin/env python3
"""The register-resident row kernels at the token counts of a stacked forward (default B = 60), one line per launch shape: for A/B builds
of csrc/fused_rows.hip (INSV2V_LIB=path/to/other/libinsv2v_hip.so python tools/bench_rows_ab.py) on one box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import ops  # noqa: E402
from insv2v.fused import pack_ffn_stream, pack_linear_stream, pack_tattn_stream, pack_tattn_qkv_stream, pack_xattn_stream, pack_xattn_kv  # noqa: E402
from insv2v.unet import fold_layernorm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
B = int(os.environ.get("NB", 60))
M0, M1, C, NH = B * 16 * 1536, B * 16 * 384, 320, 1280


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


x0 = (R(M0, C) * 1.3 + 0.2).half().to(dev)
r0 = R(M0, C).half().to(dev)
x1 = (R(M1, 640) * 1.3).half().to(dev)
r1 = R(M1, 640).half().to(dev)
wf, col, bf = fold_layernorm(R(2 * NH, C, scale=C ** -0.5), 1 + 0.1 * R(C), 0.1 * R(C), R(2 * NH) * 0.3)
w2, b2 = R(C, NH, scale=NH ** -0.5).half().float(), R(C) * 0.3
ffn = pack_ffn_stream(wf.float(), bf, w2, b2).to(dev)
lin320 = pack_linear_stream(R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3).to(dev)
lin960 = pack_linear_stream(R(960, C, scale=C ** -0.5).half().float(), R(960) * 0.3).to(dev)
lin640 = pack_linear_stream(R(640, 640, scale=640 ** -0.5).half().float(), R(640) * 0.3).to(dev)
lin1920 = pack_linear_stream(R(1920, 640, scale=640 ** -0.5).half().float(), R(1920) * 0.3).to(dev)
ta = pack_tattn_stream(R(3 * C, C, scale=C ** -0.5).half().float(), R(16, 3 * C) * 0.3, R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3).to(dev)
xa = pack_xattn_stream(R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3, R(C, C, scale=C ** -0.5).half().float(), R(C) * 0.3).to(dev)
xkv = pack_xattn_kv((R(B * 77, 2 * C) * 1.5).half().to(dev), B, 77, C, 8)
ta6 = pack_tattn_qkv_stream(R(1920, 640, scale=640 ** -0.5).half().float(), R(16, 1920) * 0.3).to(dev)
o0, o1 = torch.empty_like(x0), torch.empty_like(x1)
o960, o1920 = torch.empty((M0, 960), device=dev, dtype=torch.float16), torch.empty((M1, 1920), device=dev, dtype=torch.float16)
gnab = (R(B * 16, C, 2) * 0.5 + 1.0).float().to(dev)
cases = [
    ("ffn_fused            M0 x 320 x 1280", lambda: ops.ffn_fused(x0, ffn, NH, out=o0), 2.0 * M0 * C * 3 * NH),
    ("rowlin M0 320->320 +res +stats      ", lambda: ops.rowlin(x0, lin320, C, residual=r0, out=o0, emit_stats=True), 2.0 * M0 * C * C),
    ("rowlin M0 320->320                  ", lambda: ops.rowlin(x0, lin320, C, out=o0), 2.0 * M0 * C * C),
    ("rowlin M0 320->320 GroupNorm on load", lambda: ops.rowlin(x0, lin320, C, out=o0, gn_ab=gnab, gn_rows=1536), 2.0 * M0 * C * C),
    ("rowlin M0 320->960 LN               ", lambda: ops.rowlin(x0, lin960, 960, layernorm=True, out=o960), 2.0 * M0 * C * 960),
    ("rowlin M1 640->640 +res +stats      ", lambda: ops.rowlin(x1, lin640, 640, residual=r1, out=o1, emit_stats=True), 2.0 * M1 * 640 * 640),
    ("rowlin M1 640->640                  ", lambda: ops.rowlin(x1, lin640, 640, out=o1), 2.0 * M1 * 640 * 640),
    ("rowlin M1 640->1920 LN              ", lambda: ops.rowlin(x1, lin1920, 1920, layernorm=True, out=o1920), 2.0 * M1 * 640 * 1920),
    ("tattn_fused M0                      ", lambda: ops.tattn_fused(x0, ta, B, 1536, 8, 16, out=o0), 2.0 * M0 * C * 4 * C + 4.0 * M0 * 16 * C),
    ("xattn_fused M0                      ", lambda: ops.xattn_fused(x0, xa, xkv, 16 * 1536, 8, 77, out=o0), 4.0 * M0 * C * C + 4.0 * M0 * 77 * C),
    ("tattn_attn  M1 (C = 640)            ", lambda: ops.tattn_attn(x1, ta6, B, 384, 8, 16, out=o1), 2.0 * M1 * 640 * 3 * 640 + 4.0 * M1 * 16 * 640),
]
for rd in range(2):
    for name, fn, fl in cases:
        t = timeit(fn)
        if rd:
            print(f"{name} {t:8.1f} us  {fl / t * 1e-6:7.1f} TF/s", flush=True)
