#!/bin/bash
# halo-conv dead-wave skip A/B + VAE per-shape profile + clips-in-flight sweep at the driver's --steps 20 --warmup 5
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv" > $O/pytest_conv.txt 2>&1; tail -3 $O/pytest_conv.txt
for m in 1 0 1 0; do
  echo "== INSV2V_HALO_LEGACY_MAP=$m" >> $O/halo_ab.txt
  INSV2V_HALO_LEGACY_MAP=$m TILES=0 timeout 600 python tools/bench_conv_tiles_r03.py 2>&1 | grep -E "^L[01]" >> $O/halo_ab.txt
done
cat $O/halo_ab.txt
timeout 600 python tools/profile_vae.py > $O/vae_per_shape.txt 2> $O/vae.err; head -40 $O/vae_per_shape.txt
for c in 4 5 7 10; do
  timeout 900 python bench.py --steps 20 --warmup 5 --concurrent-clips $c --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err
  python -c "
import json
r=json.loads(open('$O/bench_c$c.json').read().strip().splitlines()[-1]); print('clips in flight $c', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'groups', r['config']['clip_groups'])" 2>&1 | tail -1
done
INSV2V_HALO_LEGACY_MAP=1 timeout 900 python bench.py --steps 20 --warmup 5 --concurrent-clips 5 --no-cpu-baseline > $O/bench_c5_legacy.json 2> $O/bench_c5_legacy.err
python -c "
import json
r=json.loads(open('$O/bench_c5_legacy.json').read().strip().splitlines()[-1]); print('clips in flight 5, legacy halo map', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4))" 2>&1 | tail -1
