#!/bin/bash
# PMC passes over the row Linear alone at three B = 60 shapes: where its waves spend their cycles.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_rowlin_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in qkv320 gn320 res640; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    CASE=$c timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/${c}_$tag -o t -- python $R/tools/pmc_rowlin.py > $O/${c}_$tag.log 2>&1
    DB=$(find $O/${c}_$tag -name "*.db" | head -1)
    echo "== $c: $pass" >> $O/report.txt
    [ -n "$DB" ] && python $R/tools/pmc_report.py $DB | grep rowlin >> $O/report.txt
  done
done
find $O -name "*.db" -delete
cat $O/report.txt
