#!/bin/bash
# final state of the round: full GPU suite, smoke, both headline bench commands
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03bb; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for f in bench_steps20 bench; do python -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline']['operator_launches_per_unet_forward'])"; done
