#!/bin/bash
# PMC split of wave cycles (issuing / parked in waits / issue-stalled) and matrix-pipe busy cycles of every row kernel (tools/pmc_rows.py).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_rows_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O/$tag -o t -- python $R/tools/pmc_rows.py > $O/$tag.log 2>&1
  DB=$(find $O/$tag -name "*.db" | head -1)
  echo "== $pass" >> $O/report.txt
  [ -n "$DB" ] && python $R/tools/pmc_report.py $DB | grep -E "ffn_fused|rowlin|tattn|xattn" >> $O/report.txt
done
find $O -name "*.db" -delete
python - $O/report.txt <<'PY'
import re, sys, collections
d = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    m = re.match(r'(.{44}) grid\s+(\d+) (\S+)\s+([0-9.e+]+) n=(\d+) dur_us=([0-9.]+)', l)
    if m: d[(m.group(1).strip()[-34:], m.group(2))][m.group(3)] = (float(m.group(4)), float(m.group(6)))
for k, v in d.items():
    if 'SQ_WAVE_CYCLES' not in v or 'SQ_WAIT_ANY' not in v: continue
    wc, waves = v['SQ_WAVE_CYCLES'][0], v['SQ_WAVES'][0]
    simds = min(waves, 1024)
    print(f"{k[0]:34s} grid {k[1]:>8s} dur {v['SQ_WAVE_CYCLES'][1]:7.1f} us  pipe busy {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (wc * 4 / waves * simds):5.1f} %  issuing {100 * v['SQ_ACTIVE_INST_ANY'][0] / wc:4.1f} %  parked {100 * v['SQ_WAIT_ANY'][0] / wc:4.1f} %  issue-stalled {100 * v['SQ_WAIT_INST_ANY'][0] / wc:4.1f} %")
PY
