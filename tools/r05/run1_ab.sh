#!/bin/bash
# Round 5, GPU call 1: same-process A/B of every fusion / row-kernel switch against the round-4 GEMM engine at the benched stack
# (B = 60) and at B = 30 (where the unfused feed-forward's hidden tensor still fits the 2 GiB operand window), plus the default
# configuration's per-shape profile at the launch shapes the product meets (B = 3, 6, 12).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_run1; mkdir -p $O
cd $R
export PER_SHAPE=$O/per_shape
NB=60 timeout 1500 python tools/ab_switches.py default FUSE_FFN=0 FUSE_FFN_POST=0 ROWLIN_640=0 FUSE_TATTN=0 FUSE_TATTN_640=0 FUSE_XATTN=0 FUSE_XATTN_640=0 FUSE_XATTN_PRE=0 ROWLIN_GN=0 ROWLIN=0 default > $O/ab_B60.txt 2>&1
cat $O/ab_B60.txt
NB=30 timeout 900 python tools/ab_switches.py default FUSE_FFN=0 ROWLIN_640=0 FUSE_TATTN=0 FUSE_TATTN_640=0 ROWLIN=0 > $O/ab_B30.txt 2>&1
cat $O/ab_B30.txt
for nb in 3 6 12; do NB=$nb timeout 600 python tools/ab_switches.py default ROWLIN_640=0 FUSE_TATTN_640=0 FUSE_XATTN_640=0 FUSE_FFN=0 > $O/ab_B$nb.txt 2>&1; cat $O/ab_B$nb.txt; done
