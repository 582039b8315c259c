#!/bin/bash
# Round-3 run 1: GPU test suite on the round's first changes + clip-mode A/B (stacked B = 3n vs interleaved streams) + per-shape profiles.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
for mode in stacked streams; do
  timeout 600 python bench.py --no-cpu-baseline --clip-mode $mode > $O/bench_$mode.json 2> $O/bench_$mode.err; python - $O/bench_$mode.json <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r["value"],3), "frames/s", round(r["ms_per_step"],1), "ms/clip frac", round(r["roofline"]["frac"],4), r["config"]["clip_groups"])
except Exception as e: print("bench failed", e)
PY
done
for cc in 2 3 4; do
  timeout 600 python bench.py --no-cpu-baseline --clip-mode stacked --steps $((cc*1)) --concurrent-clips $cc > $O/bench_stacked_cc$cc.json 2> $O/bench_stacked_cc$cc.err
  python -c "
import json,sys
r=json.loads(open('$O/bench_stacked_cc$cc.json').read().strip().splitlines()[-1]); print('stacked cc=$cc', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4))" 2>&1 | tail -1
done
NB=3 timeout 300 python tools/profile_unet.py > $O/unet_per_shape_B3.txt 2>&1; head -2 $O/unet_per_shape_B3.txt
NB=12 timeout 300 python tools/profile_unet.py > $O/unet_per_shape_B12.txt 2>&1; head -2 $O/unet_per_shape_B12.txt
