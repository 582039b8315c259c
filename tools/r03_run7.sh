#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn_fused or rowlin or co_residency" > $O/pytest_new.txt 2>&1; tail -12 $O/pytest_new.txt
if grep -q "failed" $O/pytest_new.txt; then exit 0; fi
timeout 300 python tools/bench_rowlin.py 2>&1 | grep -v amdgpu > $O/bench_rowlin.txt; cat $O/bench_rowlin.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for f in 1 0; do
  INSV2V_ROWLIN_640=$f timeout 600 python bench.py --no-cpu-baseline > $O/bench_rowlin640_$f.json 2> $O/bench_rowlin640_$f.err
  python -c "
import json
r=json.loads(open('$O/bench_rowlin640_$f.json').read().strip().splitlines()[-1]); print('ROWLIN_640=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
NB=15 timeout 300 python tools/profile_unet.py > $O/unet_per_shape_B15.txt 2>&1; head -40 $O/unet_per_shape_B15.txt
bash tools/sclk_probe.sh $O/sclk > $O/sclk_probe.txt 2>&1; head -30 $O/sclk_probe.txt
