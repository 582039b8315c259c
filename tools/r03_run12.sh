#!/bin/bash
# driver command line with the new grouping (20 steps = [10, 10], B = 30) + per-shape / PMC traffic / tile sweeps at B = 30
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err
python -c "
import json
r=json.loads(open('$O/bench_s20.json').read().strip().splitlines()[-1]); print('steps 20:', round(r['value'],3), 'frames/s frac', round(r['roofline']['frac'],4), 'groups', r['config']['clip_groups'], 'ops', r['roofline']['operator_launches_per_unet_forward'], 'traffic', r['roofline']['traffic'])" 2>&1 | tail -1
NB=30 timeout 600 python tools/profile_unet.py > $O/unet_per_shape_B30.txt 2>&1; head -3 $O/unet_per_shape_B30.txt
cd /tmp && export TMPDIR=/tmp
NB=30 timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc -o t -- python $R/tools/profile_unet.py > $O/pmc.log 2>&1
DB=$(find $O/pmc -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/pmc_forward_traffic.py $DB $O/pmc_forward_traffic_b30.json 30 16 32 48 > $O/pmc_forward_traffic_b30.txt 2>&1; cat $O/pmc_forward_traffic_b30.txt
rm -rf $O/pmc
cd $R
MSCALE=2 timeout 600 python tools/bench_tiles_r03.py > $O/tiles_b30.txt 2>&1; cat $O/tiles_b30.txt
NB=480 timeout 900 python tools/bench_conv_tiles_r03.py > $O/conv_tiles_b30.txt 2>&1; cat $O/conv_tiles_b30.txt
