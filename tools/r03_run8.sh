#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "tattn" > $O/pytest_new.txt 2>&1; tail -12 $O/pytest_new.txt
if grep -q "failed\|error" $O/pytest_new.txt; then exit 0; fi
timeout 300 python tools/bench_tattn.py 2>&1 | grep -v amdgpu > $O/bench_tattn.txt; cat $O/bench_tattn.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for f in 1 0; do
  INSV2V_FUSE_TATTN=$f timeout 600 python bench.py --no-cpu-baseline > $O/bench_tattn$f.json 2> $O/bench_tattn$f.err
  python -c "
import json
r=json.loads(open('$O/bench_tattn$f.json').read().strip().splitlines()[-1]); print('FUSE_TATTN=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
