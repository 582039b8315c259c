#!/bin/bash
# Row-kernel epilogue on v_fma_mix / v_dot2 (store_tile, finish_tile): numerics tests, then the B = 60 launch shapes before / after
# (instruct-video-to-video_amd/build/lib_before_mix.so = the previous build), alternating, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_rows_epi; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "ffn or rowlin or tattn or xattn or fused or stat" 2>&1 | tail -3 | tee -a $O/log.txt
for rep in 1 2; do
  echo "== before (rep $rep)" | tee -a $O/log.txt
  INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_before_mix.so timeout 600 python tools/bench_rows_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt
  echo "== after (rep $rep)" | tee -a $O/log.txt
  timeout 600 python tools/bench_rows_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt
done
