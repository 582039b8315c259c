#!/bin/bash
# End-to-end check of the round-5 row-kernel work: per-shape eager profile at B = 60 with the round-3 feed-forward schedule
# (INSV2V_FFN_DBG=32, previous library) and with the current build, then the bench line at the driver's command line.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_e2e; mkdir -p $O
cd $R
INSV2V_FFN_DBG=32 INSV2V_LIB=$R/instruct-video-to-video_amd/build/lib_before_mix.so NB=60 timeout 600 python tools/profile_unet.py > $O/per_shape_B60_before.txt 2>&1; head -12 $O/per_shape_B60_before.txt
NB=60 timeout 600 python tools/profile_unet.py > $O/per_shape_B60_after.txt 2>&1; head -12 $O/per_shape_B60_after.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench.err; tail -c 1500 $O/bench_steps20.json | head -c 600
