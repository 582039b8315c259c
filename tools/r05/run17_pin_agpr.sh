#!/bin/bash
# Round 5: row fragments of the two-token-block row Linear pinned in accumulator registers (MFMA B operands read from AGPRs directly
# instead of four v_accvgpr_read per use): -DROWS_PIN_AGPR=0 / 1 builds (build/lib_P0.so, lib_P1.so), alternating, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_pin_agpr; mkdir -p $O
cd $R
INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_P1.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rowlin or wide_store or stat" 2>&1 | tail -1 | tee -a $O/log.txt
for rep in 1 2; do for v in P0 P1; do
  echo "== lib_$v (rep $rep)" | tee -a $O/log.txt
  INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_$v.so timeout 600 python tools/bench_rows_ab.py 2>&1 | grep "rowlin M1" | tee -a $O/log.txt
done; done
