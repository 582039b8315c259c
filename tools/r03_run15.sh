#!/bin/bash
# patch-tiled conv with 4 waves x 64x64 wave tiles (tile 102) vs the shipped 8 waves x 32x64 (tile 0 / 100)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
for nb in 240 480 48; do
  echo "== NB=$nb" >> $O/halo_102.txt
  NB=$nb TILES=0,102,101 timeout 900 python tools/bench_conv_tiles_r03.py 2>&1 | grep -E "^L[01]" >> $O/halo_102.txt
done
cat $O/halo_102.txt
