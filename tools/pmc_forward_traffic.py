#!/usr/bin/env python3
"""HBM-side traffic of one eager UNet forward per kernel family, from a rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run of tools/profile_unet.py (2 forwards; sqlite output).  Read requests are 128 B for the wide streaming loads of these
kernels (gfx950 tallies them at 64 B in FETCH_SIZE, MI355X_MICROARCH.md), write requests 64 B (checked: a GEMM's WRREQ x 64 B
equals its output size exactly)."""
import json
import os
import sqlite3
import sys
from collections import defaultdict

# usage: pmc_forward_traffic.py results.db [out.json [B F h w]]   (the JSON is what bench.py loads as roofline.traffic)
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
fam = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for name, counter, val, n in rows:
    f = "gemm/conv" if any(t in name for t in ("gemm_kernel", "conv_halo", "splitk", "gemm_p8", "gemm_q8", "gemm_r8", "gemm_w4", "ffn_fused", "rowlin", "tattn_fused", "tattn640", "xattn640", "xattn_fused", "ln_finalize")) else "attention" if ("attn_kernel" in name or "attn_short" in name) else \
        "norm" if any(t in name for t in ("gn_", "ln_stats", "layernorm", "wino_input", "wino_output")) else "other (incl. weight init)"
    fam[f][counter] += val
    if counter.startswith("TCC_EA0_RD"):
        cnt[f] += n
for f, d in fam.items():
    rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0) * 128 / 2, d.get("TCC_EA0_WRREQ_sum", 0.0) * 64 / 2
    n = max(cnt[f] // 2, 1)
    print(f"{f:26s} launches/forward {n:5d}  read {rd / 2**30:7.2f} GiB  write {wr / 2**30:7.2f} GiB  per launch: {(rd + wr) / n / 2**20:8.1f} MiB")

if len(sys.argv) > 2:
    shape = [int(v) for v in sys.argv[3:7]] if len(sys.argv) >= 7 else [3, 16, 32, 48]
    out = {"shape": shape, "source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- python tools/profile_unet.py "
                                     "(read requests x 128 B, write requests x 64 B, 2 forwards averaged); " + os.path.basename(sys.argv[1]),
           "families": {}}
    for f, d in fam.items():
        rd, wr = d.get("TCC_EA0_RDREQ_sum", 0.0) * 128 / 2, d.get("TCC_EA0_WRREQ_sum", 0.0) * 64 / 2
        out["families"][f] = {"launches_per_forward": max(cnt[f] // 2, 1), "read_bytes_per_forward": rd, "write_bytes_per_forward": wr,
                              "bytes_per_forward": rd + wr}
    # one file holds every measured batch: bench.py picks the entry whose shape equals the batch its timed region launches
    path, shapes = sys.argv[2], []
    if os.path.exists(path):
        old = json.load(open(path))
        shapes = [e for e in (old.get("shapes") or ([{k: old[k] for k in ("shape", "source", "families")}] if "shape" in old else [])) if e["shape"] != shape]
    shapes.append(out)
    json.dump({"shapes": sorted(shapes, key=lambda e: e["shape"])}, open(path, "w"), indent=1)
    print("wrote", path, [e["shape"] for e in shapes])
