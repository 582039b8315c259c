#!/usr/bin/env python3
"""Build libinsv2v_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the
library is a plain C-ABI shared object (include/insv2v_hip.h) loaded through ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "insv2v", "libinsv2v_hip.so")
SOURCES = ["gemm.hip", "gemm_q8.hip", "gemm_r8.hip", "gemm_w4.hip", "fused_rows.hip", "norm.hip", "attention.hip", "elementwise.hip", "raft.hip", "winograd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# The persistent GEMM epilogues must not be SLP-vectorised: the v_pk_fma_f32 chains hipcc forms from the per-element
# fp32 epilogue arithmetic gave wrong values in lanes 12-15 of every 16 on gfx950 (profiles/r02_gemm_debug.md), and packed
# fp32 VALU is slower than scalar next to MFMAs anyway (cdna_hip_programming.md).
EXTRA_FLAGS = {"gemm_q8.hip": ["-fno-slp-vectorize"], "gemm_r8.hip": ["-fno-slp-vectorize"], "gemm_w4.hip": ["-fno-slp-vectorize"], "fused_rows.hip": ["-fno-slp-vectorize"]}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_dma.h"), os.path.join(HERE, "..", "include", "insv2v_hip.h")]
    jobs = []
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(OUT) or any(_newer(o, OUT) for o in objs):   # (an object compiled by hand must not leave a stale library)
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
