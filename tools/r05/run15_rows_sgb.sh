#!/bin/bash
# Round 5 experiment: scheduler pipelines (one MFMA : N VALU) in every fragment group of the attention-block row kernels
# (fused_rows.hip -DROWS_SGB=3 / 5 -> build/lib_S3.so / lib_S5.so) against the default build, alternating, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_rows_sgb; mkdir -p $O
cd $R
for v in S3 S5; do INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_$v.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "tattn or xattn" 2>&1 | tail -1 | tee -a $O/log.txt; done
for rep in 1 2; do for v in default S3 S5; do
  echo "== $v (rep $rep)" | tee -a $O/log.txt
  L=$R/instruct-video-to-video_amd/build/lib_$v.so; [ $v = default ] && L=$R/instruct-video-to-video_amd/insv2v/libinsv2v_hip.so
  INSV2V_LIB=$L timeout 600 python tools/bench_rows_ab.py 2>&1 | grep attn | tee -a $O/log.txt
done; done
