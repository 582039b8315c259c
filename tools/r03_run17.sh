#!/bin/bash
# positional-encoding bias inside the <= 16-row attention kernel (temporal q/k/v GEMM at C = 1280 without row bias -> gemm_p8): parity + A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > $O/pytest_attn.txt 2>&1; tail -3 $O/pytest_attn.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
for f in 1 0 1 0; do
  INSV2V_ATTN_PE_BIAS=$f timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pe$f.json 2> $O/bench_pe$f.err
  python -c "
import json
r=json.loads(open('$O/bench_pe$f.json').read().strip().splitlines()[-1]); print('ATTN_PE_BIAS=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
