#!/bin/bash
# N = 640 persistent GEMM split into 512 + 128 columns: parity + A/B
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $O/pytest_gemm.txt 2>&1; tail -3 $O/pytest_gemm.txt
for f in 1 0 1 0; do
  INSV2V_GEMM_SPLIT_N=$f timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_split$f.json 2> $O/bench_split$f.err
  python -c "
import json
r=json.loads(open('$O/bench_split$f.json').read().strip().splitlines()[-1]); print('GEMM_SPLIT_N=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4))" 2>&1 | tail -1
done
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest_model.txt 2>&1; tail -2 $O/pytest_model.txt
