#!/bin/bash
O=gpurun_out/r04run23; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm" 2>&1 | tail -8 | tee $O/pytest.txt
for f in 0 1 0 1; do
  INSV2V_GN_FRAME=$f timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f$f.json 2> $O/bench_f$f.err || tail -5 $O/bench_f$f.err
  python - <<PY
import json
d=json.loads(open("$O/bench_f$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("gn frame $f:", round(d["value"],3), "frames/s", {k:v["ms"] for k,v in r["families"].items()}, "rms", d["config"].get("stacked_vs_single_rel_rms"))
PY
done 2>&1 | tee $O/summary.txt
