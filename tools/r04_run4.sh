#!/bin/bash
# round 4 run 4: full GPU test suite + the driver's bench command line with gemm_q8 dispatched (and gemm_p8 for A/B on the same box)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_run4_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_run4_pytest.txt
tail -5 gpurun_out/r04_run4_pytest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_run4_bench_q8.json 2> gpurun_out/r04_run4_bench_q8.err; tail -c 600 gpurun_out/r04_run4_bench_q8.json
INSV2V_GEMM_P8=1 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_run4_bench_p8.json 2> gpurun_out/r04_run4_bench_p8.err; tail -c 600 gpurun_out/r04_run4_bench_p8.json
