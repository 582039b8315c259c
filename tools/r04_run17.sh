#!/bin/bash
O=gpurun_out/r04run17; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | tail -12 | tee $O/pytest.txt
for f in 0 1 0 1; do echo "INSV2V_ATTN_FOLD=$f"; INSV2V_ATTN_FOLD=$f python tools/bench_attn.py 2>&1 | grep "self" | head -2; done | tee $O/bench_attn.txt
