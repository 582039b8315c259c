#!/bin/bash
# final state check: full GPU suite, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03v; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; python -c "
import json
r=json.loads(open('$O/bench_steps20.json').read().strip().splitlines()[-1]); print(round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'cpu', r['cpu_baseline']['value'])"
