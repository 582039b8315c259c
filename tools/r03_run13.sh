#!/bin/bash
# fused cross-attention block: parity, microbench, end-to-end A/B at the driver's command line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "xattn or conv" > $O/pytest_xattn.txt 2>&1; tail -15 $O/pytest_xattn.txt
timeout 600 python tools/bench_xattn.py > $O/xattn_microbench.txt 2>&1; cat $O/xattn_microbench.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest_model.txt 2>&1; tail -5 $O/pytest_model.txt
for f in 1 0; do
  INSV2V_FUSE_XATTN=$f timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_xattn$f.json 2> $O/bench_xattn$f.err
  python -c "
import json
r=json.loads(open('$O/bench_xattn$f.json').read().strip().splitlines()[-1]); print('FUSE_XATTN=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'), 'traffic', r['roofline']['traffic'])" 2>&1 | tail -1
done
