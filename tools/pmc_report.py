#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite outputs: per (kernel, grid) mean counter values over dispatches."""
import sqlite3
import sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    print("==", path)
    rows = c.execute("select kernel_name, grid_size, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                     "group by kernel_name, grid_size, counter_name order by kernel_name, grid_size, counter_name").fetchall()
    for r in rows:
        if any(t in r[0] for t in ("gemm", "conv_halo", "attn", "norm", "splitk", "ffn_fused", "rowlin", "tattn", "wino")):
            print(f"{r[0][:44]:44s} grid {r[1]:9d} {r[2]:26s} {r[3]:14.5g} n={r[4]} dur_us={r[5] / 1e3 if r[5] else 0:.1f}")
