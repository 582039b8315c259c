#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
python tools/debug_ffn.py 2>&1 | grep -v amdgpu.ids | grep -v "per-" > $O/debug_ffn.txt; cat $O/debug_ffn.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn_fused" > $O/pytest_ffn.txt 2>&1; tail -4 $O/pytest_ffn.txt
if grep -q "failed" $O/pytest_ffn.txt; then exit 0; fi
for d in 0 1 2 4 7; do echo "== INSV2V_FFN_DBG=$d"; INSV2V_FFN_DBG=$d timeout 300 python tools/bench_ffn.py 2>&1 | grep "M= 294912 round 2"; done | tee $O/bench_ffn_ablation.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for f in 1 0; do
  INSV2V_FUSE_FFN=$f timeout 600 python bench.py --no-cpu-baseline > $O/bench_ffn$f.json 2> $O/bench_ffn$f.err
  python -c "
import json
r=json.loads(open('$O/bench_ffn$f.json').read().strip().splitlines()[-1]); print('FUSE_FFN=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
