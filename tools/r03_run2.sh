#!/bin/bash
# Round-3 run 2: producer-side LN statistics + fused FFN kernel: kernel tests, microbench, model tests, bench A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn_fused or emits_layernorm or gemm_plain or gemm_epilogues" > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
timeout 300 python tools/bench_ffn.py > $O/bench_ffn.txt 2>&1; cat $O/bench_ffn.txt
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
for f in 1 0; do
  INSV2V_FUSE_FFN=$f timeout 600 python bench.py --no-cpu-baseline > $O/bench_ffn$f.json 2> $O/bench_ffn$f.err
  python -c "
import json
r=json.loads(open('$O/bench_ffn$f.json').read().strip().splitlines()[-1]); print('FUSE_FFN=$f', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'ops', r['roofline'].get('operator_launches_per_unet_forward'))" 2>&1 | tail -1
done
