#!/bin/bash
# round 4 run 6: kernel tests + full-size tests + the driver's bench with gemm_r8 / gemm_q8 dispatched, and with INSV2V_GEMM_R8=0 (q8 only) on the same box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_run6_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_run6_pytest.txt
tail -4 gpurun_out/r04_run6_pytest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_run6_bench_r8.json 2> gpurun_out/r04_run6_bench_r8.err; tail -c 300 gpurun_out/r04_run6_bench_r8.json
INSV2V_GEMM_R8=0 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_run6_bench_q8only.json 2> gpurun_out/r04_run6_bench_q8only.err; tail -c 300 gpurun_out/r04_run6_bench_q8only.json
