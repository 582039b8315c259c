#!/bin/bash
# What bounds the temporal attention launches (tools/pmc_attn_short.py): timing, then rocprofv3 --pmc passes (counters only, beside --kernel-trace):
# L1 -> L2 requests, L2 hits / misses, fabric-side requests.  Output: gpurun_out/${TAG}_attn_short/
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-r06}; O=$R/gpurun_out/${TAG}_attn_short; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CASE=t1280,t1280s,c5_320 ITERS=20 python $R/tools/pmc_attn_short.py 2>&1 | grep -v amdgpu.ids > $O/timing.txt; cat $O/timing.txt
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i + 1))
  CASE=t1280,c5_320 ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p -- python $R/tools/pmc_attn_short.py > $O/p$i.log 2>&1
  DB=$(find $O/p$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_report.py $DB 2>&1 | grep -i "attn" > $O/pmc_$i.txt || tail -5 $O/p$i.log
  cat $O/pmc_$i.txt 2>/dev/null
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
