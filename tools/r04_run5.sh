#!/bin/bash
# round 4 run 5: gemm_r8 (256x320 tile; 240 product, 242 no epilogue) - correctness on the edge set, then the B = 30 shapes against the
# dispatch (0 = conv_halo / 128x128 / gemm_q8) and gemm_q8 (230)
cd "$GRAFT_REPO_ROOT"
G=instruct-video-to-video_amd/build/gemm_check
{
echo "== edge320 (correctness)"; timeout 120 $G --set edge320 --tiles 5,240 --iters 3
echo "== unet30"; timeout 600 $G --set unet30 --tiles 0,230,240,242 --iters 5
} > gpurun_out/r04_run5_gemm_r8.txt 2>&1
cat gpurun_out/r04_run5_gemm_r8.txt
