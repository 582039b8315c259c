#!/bin/bash
# Matrix-pipe utilisation, LDS bank conflicts and wait shares of the product's kernels (two rocprofv3 --pmc passes over tools/pmc_rows.py,
# counters only - no --stats, no trace domains beside --kernel-trace).  Output: gpurun_out/${TAG}_pmc/pmc_pipe_utilisation.txt
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-r06}; O=$R/gpurun_out/${TAG}_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d $O/pmc_mfma -o m -- python $R/tools/pmc_rows.py > $O/pmc_mfma.log 2>&1
DBM=$(find $O/pmc_mfma -name "*.db" | head -1); [ -n "$DBM" ] && python $R/tools/pmc_report.py $DBM > $O/pmc_rows_mfma.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_lds -o l -- python $R/tools/pmc_rows.py > $O/pmc_lds.log 2>&1
DBL=$(find $O/pmc_lds -name "*.db" | head -1); [ -n "$DBL" ] && python $R/tools/pmc_report.py $DBL > $O/pmc_rows_lds.txt 2>&1
python $R/tools/pmc_pipe_summary.py $O/pmc_rows_mfma.txt $O/pmc_rows_lds.txt > $O/pmc_pipe_utilisation.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
cat $O/pmc_pipe_utilisation.txt
