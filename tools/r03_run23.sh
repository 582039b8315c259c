#!/bin/bash
# C = 640 fused LayerNorm + q/k/v + temporal attention: parity, microbench, end-to-end A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03w; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "tattn" > $O/pytest_tattn.txt 2>&1; tail -5 $O/pytest_tattn.txt
timeout 600 python tools/bench_tattn640.py > $O/tattn640_microbench.txt 2>&1; grep -v amdgpu $O/tattn640_microbench.txt
