#!/usr/bin/env python3
"""Matrix-pipe utilisation and LDS conflict share per kernel from the two tools/pmc_report.py summaries of tools/pmc_rows.py
(usage: pmc_pipe_summary.py pmc_rows_mfma.txt pmc_rows_lds.txt).  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs
(32 per v_mfma_f32_32x32x16_f16), SQ_BUSY_CYCLES sums 32 shader engines: utilisation = MFMA_BUSY / (1024 x SQ_BUSY / 32)."""
import re
import sys
from collections import defaultdict


def parse(path):
    d = defaultdict(dict)
    for l in open(path):
        m = re.match(r"(.{44}) grid\s+(\d+) (\S+)\s+([\d.e+]+) n=(\d+) dur_us=([\d.]+)", l)
        if m:
            key = (m.group(1).strip(), int(m.group(2)), float(m.group(6)))
            d[key][m.group(3)] = float(m.group(4))
    return d


mf, ld = parse(sys.argv[1]), parse(sys.argv[2])
for (name, grid, dur), c in sorted(mf.items(), key=lambda kv: kv[0][0]):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "SQ_BUSY_CYCLES" not in c:
        continue
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["SQ_BUSY_CYCLES"] / 32)
    clk = c["SQ_BUSY_CYCLES"] / 32 / dur / 1e3
    print(f"{name[:44]:44s} grid {grid:7d} {dur:8.1f} us  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES']:.4g}  SQ busy {c['SQ_BUSY_CYCLES']:.4g}  -> MFMA utilisation {100 * util:5.1f} %  (busy clock ~{clk:.2f} GHz; waves {int(c.get('SQ_WAVES', 0))})")
for (name, grid, dur), c in sorted(ld.items(), key=lambda kv: kv[0][0]):
    if "SQ_LDS_IDX_ACTIVE" in c:
        print(f"{name[:44]:44s} grid {grid:7d} {dur:8.1f} us  LDS bank-conflict cycles {c.get('SQ_LDS_BANK_CONFLICT', 0):.4g} of {c['SQ_LDS_IDX_ACTIVE']:.4g} active = "
              f"{100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c['SQ_LDS_IDX_ACTIVE'], 1):.1f} %;  SQ_WAIT_INST_ANY {c.get('SQ_WAIT_INST_ANY', 0):.4g}  SQ_WAIT_ANY {c.get('SQ_WAIT_ANY', 0):.4g}")
