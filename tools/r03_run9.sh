#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn_fused or rowlin or tattn" > $O/pytest_new.txt 2>&1; tail -12 $O/pytest_new.txt
if grep -q "failed\|error" $O/pytest_new.txt; then exit 0; fi
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  INSV2V_FUSE_FFN_POST=$1 INSV2V_ROWLIN_GN=$2 timeout 600 python bench.py --no-cpu-baseline > $O/bench_post$1_gn$2.json 2> $O/bench_post$1_gn$2.err
  python -c "
import json
r=json.loads(open('$O/bench_post$1_gn$2.json').read().strip().splitlines()[-1]); print('FFN_POST=$1 ROWLIN_GN=$2', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
