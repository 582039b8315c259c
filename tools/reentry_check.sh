#!/bin/bash
# HEAD check in ONE gpurun call (after a container re-creation / before the round ends): the GPU suite, smoke(), the driver's bench command
# line, and the C5 line at the largest stack its operand window allows (7 clips: whole rounds of 256-row tiles at levels 0-2).
# Outputs: gpurun_out/${TAG}/
cd ${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r06_reentry}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 400 $O/bench_steps20.json
timeout 900 python bench.py --frames 24 --height 384 --width 512 --steps 7 --warmup 2 --no-cpu-baseline > $O/bench_c5_steps7.json 2> $O/bench_c5.err; head -c 400 $O/bench_c5_steps7.json
timeout 900 python bench.py --frames 24 --height 384 --width 512 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_steps6.json 2> $O/bench_c5_6.err; head -c 400 $O/bench_c5_steps6.json
