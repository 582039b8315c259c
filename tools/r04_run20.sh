#!/bin/bash
# GroupNorm in sample groups (statistics pass + apply pass per group: does the apply pass hit the memory-side cache?)
O=gpurun_out/r04run20; mkdir -p $O
for g in 0 96 0 48 192; do
  INSV2V_GN_GROUP_MB=$g timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_g$g.json 2> $O/bench_g$g.err || tail -5 $O/bench_g$g.err
  python - <<PY
import json
d=json.loads(open("$O/bench_g$g.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("GN group MB $g:", round(d["value"],3), "frames/s", {k:v["ms"] for k,v in r["families"].items()}, "rms", d["config"].get("stacked_vs_single_rel_rms"))
PY
done 2>&1 | tee $O/summary.txt
