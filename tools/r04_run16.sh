#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r04run16; mkdir -p $O
for f in 1 2; do
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    INSV2V_ATTN_FOLD=$f timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/f${f}_$i -o p -- python $R/tools/pmc_attn.py > $O/f${f}_$i.log 2>&1
    DB=$(find $O/f${f}_$i -name "*.db" | head -1)
    [ -n "$DB" ] && python $R/tools/pmc_report.py $DB 2>&1 | grep "attn_kernel" | sed "s/^/fold $f: /"
  done
done | tee $O/summary.txt
rm -rf $O/f*_*/
