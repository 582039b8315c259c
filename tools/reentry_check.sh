set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_reentry; mkdir -p $O
export TMPDIR=/tmp
TAG=r06b MODE=lds SEC=6 bash tools/engine_ceiling.sh > $O/ceiling_lds.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 600 $O/bench_steps20.json
