#!/bin/bash
# Round 5: shared CFG prefix computed once (branch-major stack, unet.forward_cl(cfg_clips)): tests, per-shape profile with / without, bench A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_dedup; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py -q -m gpu -x -k "branch_major or build_unet or cfg_step or dedup or stacked or second_clip or c4 or ddpm or graph or edit" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
CFG_CLIPS=0 NB=60 timeout 600 python tools/profile_unet.py > $O/per_shape_B60_nodedup.txt 2>&1; head -2 $O/per_shape_B60_nodedup.txt | tail -1
NB=60 timeout 600 python tools/profile_unet.py > $O/per_shape_B60_dedup.txt 2>&1; head -2 $O/per_shape_B60_dedup.txt | tail -1
for v in 0 1 0 1; do
  INSV2V_DEDUP_CFG=$v timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dedup$v.json 2> $O/bench$v.err
  python - $O/bench_dedup$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print('DEDUP', sys.argv[2], d['value'], d['ms_per_step'], d['config'].get('stacked_vs_single_rel_rms'))
PY
done
