#!/bin/bash
# segfault check of the runner-cache finalizer + two-token-block row Linear (K = 640) parity and A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
for m in 0 255; do
  INSV2V_ROWLIN_TB2=$m timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rowlin or co_residency" > $O/pytest_rowlin_tb$m.txt 2>&1; tail -3 $O/pytest_rowlin_tb$m.txt
done
for m in 0 255 0 255; do INSV2V_ROWLIN_TB2=$m timeout 600 python tools/bench_rowlin_tb2.py 2>&1 | grep -v amdgpu >> $O/rowlin_tb2.txt; done
cat $O/rowlin_tb2.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for m in 0 17 51 255; do
  INSV2V_ROWLIN_TB2=$m timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tb$m.json 2> $O/bench_tb$m.err
  python -c "
import json
r=json.loads(open('$O/bench_tb$m.json').read().strip().splitlines()[-1]); print('ROWLIN_TB2=$m', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
