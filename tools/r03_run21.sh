#!/bin/bash
# 16-wave / 256-pixel-patch / deep-weight-ring form of the patch-tiled conv (tile 103): parity + microbench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3x3" > $O/pytest_conv.txt 2>&1; tail -8 $O/pytest_conv.txt
for nb in 240 480 48; do
  echo "== NB=$nb" >> $O/halo_103.txt
  NB=$nb TILES=0,103 timeout 900 python tools/bench_conv_tiles_r03.py 2>&1 | grep -E "^L0" >> $O/halo_103.txt
done
cat $O/halo_103.txt
