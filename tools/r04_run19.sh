#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r04run19; mkdir -p $O
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p -- python $R/tools/pmc_rows.py > $O/p$i.log 2>&1
  DB=$(find $O/p$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_report.py $DB > $O/rows$i.txt 2>&1
done
rm -rf $O/p1 $O/p2
python3 - <<PY
import re,collections
d=collections.defaultdict(dict)
for f in ("$O/rows1.txt","$O/rows2.txt"):
    for l in open(f):
        m=re.match(r"(.{44}) grid\s+(\d+) (\S+)\s+([\d.e+]+) n=(\d+) dur_us=([\d.]+)", l)
        if m: d[(m.group(1).strip(), int(m.group(2)), round(float(m.group(6)),-1))][m.group(3)]=float(m.group(4)); d[(m.group(1).strip(), int(m.group(2)), round(float(m.group(6)),-1))]["dur"]=float(m.group(6))
for k,c in sorted(d.items()):
    if "SQ_BUSY_CYCLES" not in c: continue
    cyc=c["SQ_BUSY_CYCLES"]/32
    valu=c.get("SQ_ACTIVE_INST_VALU",0)*4/1024/cyc
    print(f"{k[0][:44]:44s} {c['dur']:8.1f} us  VALU issue busy {100*valu:5.1f} %  VALU insts/MFMA {c.get('SQ_INSTS_VALU',0)/max(c.get('SQ_INSTS_MFMA',1),1):5.2f}  cycles/VALU inst {c.get('SQ_ACTIVE_INST_VALU',0)*4/max(c.get('SQ_INSTS_VALU',1),1):4.2f}")
PY
