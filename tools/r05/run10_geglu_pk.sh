#!/bin/bash
# Round 5: GEGLU epilogue of gemm_r8 in packed fp16 arithmetic (INSV2V_R8_GEGLU_PK=1, default) vs the fp32 form (=0): GEGLU tests under
# both, per-shape profile at B = 60 under both (alternating), one box.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_geglu_pk; mkdir -p $O
cd $R
for v in 0 1; do INSV2V_R8_GEGLU_PK=$v timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "geglu or gemm" 2>&1 | tail -2 | tee -a $O/log.txt; done
for v in 0 1 0 1; do
  echo "== INSV2V_R8_GEGLU_PK=$v" | tee -a $O/log.txt
  INSV2V_R8_GEGLU_PK=$v NB=60 timeout 600 python tools/profile_unet.py > $O/per_shape_pk$v.txt 2>&1
  grep -E "^B=|'lin', [0-9]+, (5120|10240)" $O/per_shape_pk$v.txt | tee -a $O/log.txt
done
