#!/bin/bash
O=gpurun_out/r04run12; mkdir -p $O
python -m pytest tests/test_full_size_gpu.py -q -m gpu -k "stacked" 2>&1 | tail -8 | tee $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
timeout 900 python bench.py --frames 24 --height 384 --width 512 --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err || tail -5 $O/bench_c5.err
for f in default c5; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f:", round(d["value"],3), "frames/s", d["config"].get("clip_groups"), "frac", round(d["roofline"]["frac"],4), "rms", d["config"].get("stacked_vs_single_rel_rms"))
PY
done 2>&1 | tee $O/summary.txt
