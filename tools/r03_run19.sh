#!/bin/bash
# retired graphs parked instead of destroyed: both test orders + the bench line of the final build
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > $O/pytest_reordered.txt 2>&1; tail -2 $O/pytest_reordered.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 600 $O/bench_steps20.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
