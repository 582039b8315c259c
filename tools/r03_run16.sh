#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
timeout 600 python bench.py --long-video --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_long_video.json 2> $O/bench_long.err; tail -c 400 $O/bench_long_video.json; tail -3 $O/bench_long.err
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
