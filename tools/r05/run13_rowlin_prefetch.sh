#!/bin/bash
# Round 5: row Linear requesting its next tile's rows before the last epilogue: A = off, B = on without the GroupNorm-on-load form,
# C = on for all one-block forms (three builds of fused_rows.hip linked into build/lib_pf_{A,B,C}.so), alternating, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_rowlin_pf; mkdir -p $O
cd $R
for v in B C; do INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_pf_$v.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rowlin" 2>&1 | tail -1 | tee -a $O/log.txt; done
for rep in 1 2; do for v in A B C; do
  echo "== lib_pf_$v (rep $rep)" | tee -a $O/log.txt
  INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_pf_$v.so timeout 600 python tools/bench_rows_ab.py 2>&1 | grep rowlin | tee -a $O/log.txt
done; done
