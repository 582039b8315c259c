#!/bin/bash
O=gpurun_out/r04run22; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "geglu or q8 or co_residency" 2>&1 | tail -8 | tee $O/pytest.txt
for f in 0 1 0 1; do
  INSV2V_R8_GEGLU=$f timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_g$f.json 2> $O/bench_g$f.err || tail -5 $O/bench_g$f.err
  python - <<PY
import json
d=json.loads(open("$O/bench_g$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("r8 geglu $f:", round(d["value"],3), "frames/s frac", round(r["frac"],4), {k:v["ms"] for k,v in r["families"].items()}, "rms", d["config"].get("stacked_vs_single_rel_rms"))
PY
done 2>&1 | tee $O/summary.txt
