#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03aa; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest_model.txt 2>&1; tail -2 $O/pytest_model.txt
for f in 1 0 1 0; do
  INSV2V_FUSE_XATTN_PRE=$f timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_p$f.json 2> $O/bench_p$f.err
  python -c "
import json
r=json.loads(open('$O/bench_p$f.json').read().strip().splitlines()[-1]); print('FUSE_XATTN_PRE=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
