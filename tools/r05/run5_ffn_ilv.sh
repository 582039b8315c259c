#!/bin/bash
# Round 5: feed-forward row kernel with the GEGLU arithmetic spread between the MFMAs (INSV2V_FFN_DBG: 0 = product: one pair of hidden
# units per fragment group, two P buffers, explicit fragment wait; 32 = round-3 form, GEGLU in one lump; 64 = without the explicit wait;
# 128 = S values pinned per pair): row-kernel numerics tests + microbench per variant, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_ffn_ilv; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "ffn or rowlin or tattn or xattn or fused" 2>&1 | tail -3 | tee -a $O/log.txt
for v in ${VARIANTS:-32 0 64 128}; do
  echo "== INSV2V_FFN_DBG=$v" | tee -a $O/log.txt
  INSV2V_FFN_DBG=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "ffn" 2>&1 | tail -1 | tee -a $O/log.txt
  INSV2V_FFN_DBG=$v timeout 600 python tools/bench_ffn.py 2>&1 | grep -v "round 0" | tee -a $O/log.txt
done
