#!/bin/bash
# folded softmax A/B end to end (same box, alternating) + model-level tests
O=gpurun_out/r04run14; mkdir -p $O
python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest.txt
for f in 0 1 0 1; do
  INSV2V_ATTN_FOLD=$f timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_fold$f.json 2> $O/bench_fold$f.err || tail -5 $O/bench_fold$f.err
  python - <<PY
import json
d=json.loads(open("$O/bench_fold$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("fold $f:", round(d["value"],3), "frames/s", d["config"].get("clip_groups"), "frac", round(r["frac"],4), "rms", d["config"].get("stacked_vs_single_rel_rms"), {k:v["ms"] for k,v in r["families"].items()})
PY
done 2>&1 | tee $O/summary.txt
