#!/bin/bash
# Round 5, GPU call 4: the optical-flow pipe with its own estimator, RAFT timing per window, C3 with RAFT inside the timed region.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_run4; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_raft_gpu.py -q -x -s -k "pipe" > $O/pytest_raft_pipe.txt 2>&1; tail -8 $O/pytest_raft_pipe.txt
timeout 600 python tools/bench_raft.py > $O/bench_raft.txt 2>&1; cat $O/bench_raft.txt
timeout 900 python bench.py --flow-correction --raft --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_raft.json 2> $O/bench_c3_raft.err; tail -c 500 $O/bench_c3_raft.json; tail -3 $O/bench_c3_raft.err
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c2_steps10.json 2> $O/bench_c2.err; tail -c 300 $O/bench_c2_steps10.json
