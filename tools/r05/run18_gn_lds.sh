#!/bin/bash
# Round 5: GroupNorm-on-load row Linear with the sample's (scale, shift) table staged in LDS: -DROWLIN_GN_LDS=0 (lib_G0), 1 (lib_G1),
# 1 + next-tile row prefetch for this form (lib_G2); numerics tests, then alternating, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gn_lds; mkdir -p $O
cd $R
for v in G1 G2; do INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_$v.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rowlin" 2>&1 | tail -1 | tee -a $O/log.txt; done
for rep in 1 2; do for v in G0 G1 G2; do
  echo "== lib_$v (rep $rep)" | tee -a $O/log.txt
  INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_$v.so timeout 600 python tools/bench_rows_ab.py 2>&1 | grep "GroupNorm" | tee -a $O/log.txt
done; done
