#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03z; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xattn" > $O/pytest_xattn.txt 2>&1; tail -5 $O/pytest_xattn.txt
timeout 600 python tools/bench_xattn.py > $O/xattn_pre_microbench.txt 2>&1; grep -v amdgpu $O/xattn_pre_microbench.txt
