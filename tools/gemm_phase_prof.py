#!/usr/bin/env python3
"""Where does a GEMM workgroup spend its life?  Builds a DEBUG copy of the library with -DINSV2V_GEMM_PROF
(per-workgroup wall-clock marks, 100 MHz) and prints mean microseconds per phase for the UNet's shapes.
Phases: wait first slice | K loop | epilogue math | barrier | copy-out (until stores retire)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instruct-video-to-video_amd")
out = os.path.join(PKG, "build", "libinsv2v_prof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(PKG, "csrc", f) for f in ("gemm.hip", "gemm_p8.hip", "gemm_w4.hip", "norm.hip", "attention.hip", "elementwise.hip")]
if not (os.environ.get("PROF_SKIP_BUILD") and os.path.exists(out)):  # build here (no GPU needed), run on the GPU box with PROF_SKIP_BUILD=1
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                           "-I" + os.path.join(ROOT, "include"), "-DINSV2V_GEMM_PROF", *srcs, "-o", out])
if os.environ.get("PROF_BUILD_ONLY"):
    sys.exit(0)
os.environ["INSV2V_LIB"] = out
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402
from insv2v import ops, _lib  # noqa: E402
from insv2v.unet import prep_conv3x3  # noqa: E402

lib = _lib.load()
lib.insv2v_prof_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda:0")


def read(reset=True):
    buf = (ctypes.c_ulonglong * 8)()
    assert lib.insv2v_prof_read(buf, int(reset)) == 0
    return list(buf)


def report(name, fn, iters=5):
    fn()
    torch.cuda.synchronize()
    read()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = read()
    n = max(t[5], 1)
    us = [x / n / 100.0 for x in t[:5]]
    print(f"{name:46s} {e0.elapsed_time(e1) / iters * 1e3:8.1f} us/launch  wg={n // iters:6d}  first-slice {us[0]:6.2f}  loop {us[1]:6.2f}  "
          f"epi {us[2]:6.2f}  barrier {us[3]:6.2f}  copy-out {us[4]:6.2f}   (us per workgroup)", flush=True)


tiles = [int(t) for t in os.environ.get("TILES", "0").split(",")]
LIN = [(73728, 320, 320, 0, True), (24576, 320, 320, 0, True), (18432, 640, 640, 0, True), (6144, 640, 640, 0, True), (4608, 1280, 1280, 0, True),
       (1536, 1280, 1280, 0, True), (73728, 320, 1280, 0, True), (24576, 320, 1280, 0, True), (8192, 8192, 8192, 0, False)]
for M, N, K, act, res in LIN:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N // 2 if act == 2 else N, device=dev).half() if res else None
    for tile in tiles:
        report(f"lin {M}x{N}x{K} act{act} res{int(res)} tile{tile}", lambda: ops.gemm(a, w, b, act=act, residual=r, tile=tile))
for nb, h, w_, cin, cout in [(48, 32, 48, 320, 320), (48, 16, 24, 640, 640), (48, 8, 12, 1280, 1280)]:
    x = torch.randn(nb * h * w_, cin, device=dev).half()
    wt = torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5
    wk, bk = prep_conv3x3({"c.weight": wt, "c.bias": torch.zeros(cout)}, "c", dev)
    for tile in tiles:
        report(f"conv {nb * h * w_}x{cout}x{9 * cin} tile{tile}", lambda: ops.conv3x3(x, (nb, h, w_), wk, bk, tile=tile))
