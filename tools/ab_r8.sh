#!/bin/bash
# A/B of two builds of the library on one box (round 6: 16x16x32 vs 32x32x16 MFMAs in the GEMM engine): the stand-alone harness linked against
# the current library (gemm_check) and against build/libinsv2v_hip_old.so (gemm_check_old), alternating; correctness of the new one first.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-r06}_ab; mkdir -p $O; cd $R
B=$R/instruct-video-to-video_amd/build
TILES=${TILES:-240}
{ echo "== correctness, new build: edge320 / unet60 (forced tile $TILES) / big320"; $B/gemm_check --set edge320 --tiles $TILES --iters 2; $B/gemm_check --set unet60 --tiles $TILES --iters 3; $B/gemm_check --set big320 --tiles $TILES --iters 3 --uniform
  for rep in 1 2; do
    echo "== timing rep $rep OLD build"; $B/gemm_check_old --set unet60 --tiles $TILES --iters 10 --nocheck; $B/gemm_check_old --set big320 --tiles $TILES --iters 20 --nocheck --uniform
    echo "== timing rep $rep NEW build"; $B/gemm_check --set unet60 --tiles $TILES --iters 10 --nocheck; $B/gemm_check --set big320 --tiles $TILES --iters 20 --nocheck --uniform
  done; } > $O/ab_tiles$TILES.txt 2>&1
cat $O/ab_tiles$TILES.txt
