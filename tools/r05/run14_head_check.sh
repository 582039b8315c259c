#!/bin/bash
# Round 5: the final evidence run was made one kernel change before HEAD (row Linear prefetch): GPU suite, bench line and B = 60 profile of HEAD.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_head; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench.err; tail -c 400 $O/bench_steps20.json
NB=60 timeout 600 python tools/profile_unet.py > $O/per_shape_B60.txt 2>&1; head -2 $O/per_shape_B60.txt | tail -1
