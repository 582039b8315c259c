#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn_fused or rowlin" > $O/pytest_new.txt 2>&1; tail -12 $O/pytest_new.txt
if grep -q "failed" $O/pytest_new.txt; then exit 0; fi
for d in 0 8 16 24; do echo "== INSV2V_FFN_DBG=$d"; INSV2V_FFN_DBG=$d timeout 300 python tools/bench_ffn.py 2>&1 | grep "M= 294912 round [12]"; done | tee $O/bench_ffn_variants.txt
timeout 300 python tools/bench_rowlin.py 2>&1 | grep -v amdgpu > $O/bench_rowlin.txt; cat $O/bench_rowlin.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for f in 1 0; do
  INSV2V_ROWLIN=$f timeout 600 python bench.py --no-cpu-baseline > $O/bench_rowlin$f.json 2> $O/bench_rowlin$f.err
  python -c "
import json
r=json.loads(open('$O/bench_rowlin$f.json').read().strip().splitlines()[-1]); print('ROWLIN=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
