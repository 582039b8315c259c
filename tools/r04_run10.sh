#!/bin/bash
# 20 stacked clips (B = 60) vs 10 (B = 30), same box; rowlin with operands beyond 2 GiB
O=gpurun_out/r04run10; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "rowlin" 2>&1 | tail -5 > $O/pytest_rowlin.txt
cat $O/pytest_rowlin.txt
for c in 10 20 10 20; do
  INSV2V_MAX_CLIPS=$c timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c$c.json 2> $O/bench_c$c.err || tail -5 $O/bench_c$c.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_c$c.json").read().strip().splitlines()[-1])
    print("max clips $c:", round(d["value"],3), "frames/s", d["config"].get("clip_groups"), "frac", round(d["roofline"]["frac"],4), "rms", d["config"].get("stacked_vs_single_rel_rms"), "fwd batch", d["roofline"].get("unet_batch"))
except Exception as e:
    print("max clips $c: FAILED", e)
PY
done 2>&1 | tee $O/summary.txt
