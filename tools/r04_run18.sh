#!/bin/bash
O=gpurun_out/r04run18; mkdir -p $O
INSV2V_ATTN_FOLD=2 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | tail -6 | tee $O/pytest_fold2.txt
for f in 1 2 1 2 1 2; do echo "INSV2V_ATTN_FOLD=$f"; INSV2V_ATTN_FOLD=$f python tools/bench_attn.py 2>&1 | grep "self" | head -2; done | tee $O/bench_attn.txt
