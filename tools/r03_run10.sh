#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for f in 1 0; do
  INSV2V_GEMM_PERSISTENT=$f timeout 600 python bench.py --no-cpu-baseline > $O/bench_persist$f.json 2> $O/bench_persist$f.err
  python -c "
import json
r=json.loads(open('$O/bench_persist$f.json').read().strip().splitlines()[-1]); print('GEMM_PERSISTENT=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
