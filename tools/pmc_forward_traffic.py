#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE per kernel family from a rocprofv3 --pmc csv of tools/profile_unet.py (2 UNet forwards).
usage: pmc_forward_traffic.py <counter_collection.csv>   (FETCH_SIZE / WRITE_SIZE are KiB; FETCH under-reports wide reads 2x on gfx950)"""
import csv
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
seen = set()
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"]
    fam = "gemm/conv" if ("gemm_kernel" in name or "conv_halo" in name or "splitk" in name) else \
          "attention" if "attn_kernel" in name else "norm" if ("gn_" in name or "ln_" in name or "layernorm" in name) else "other"
    tot[fam][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"], fam)
    if key not in seen:
        seen.add(key)
        n[fam] += 1
for fam in tot:
    f, w = tot[fam].get("FETCH_SIZE", 0.0), tot[fam].get("WRITE_SIZE", 0.0)
    print(f"{fam:10s} dispatches(2 forwards)={n[fam]:5d}  per forward: FETCH_SIZE {f / 2 / 2**20:7.2f} GiB (x2 corrected {f / 2**20:7.2f} GiB)  "
          f"WRITE_SIZE {w / 2 / 2**20:7.2f} GiB")
