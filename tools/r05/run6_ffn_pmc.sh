#!/bin/bash
# PMC passes over the fused feed-forward alone: matrix-pipe busy cycles, wave cycles, wait / active issue cycles, per variant.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_ffn_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-0 32}; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    INSV2V_FFN_DBG=$v timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/v${v}_$tag -o t -- python $R/tools/pmc_ffn.py > $O/v${v}_$tag.log 2>&1
    DB=$(find $O/v${v}_$tag -name "*.db" | head -1)
    echo "== variant $v: $pass" >> $O/report.txt
    [ -n "$DB" ] && python $R/tools/pmc_report.py $DB | grep ffn >> $O/report.txt
  done
done
find $O -name "*.db" -delete
cat $O/report.txt
