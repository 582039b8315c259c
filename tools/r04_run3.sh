#!/bin/bash
# round 4 run 3: gemm_q8 product schedule (230 = LDS-DMA inside the MFMA segments) vs 231 (in the load segments), 233 (serial epilogues), 200 (gemm_p8), 0 (dispatch)
cd "$GRAFT_REPO_ROOT"
G=instruct-video-to-video_amd/build/gemm_check
{
echo "== edge (correctness)"; $G --set edge --tiles 230,231 --iters 3
echo "== unet B=3"; $G --set unet --tiles 0,200,230,231,233 --iters 10
echo "== big"; $G --set big --tiles 230,231,200 --iters 10
} > gpurun_out/r04_run3_gemm_check.txt 2>&1
cat gpurun_out/r04_run3_gemm_check.txt
