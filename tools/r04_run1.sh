#!/bin/bash
# round 4 run 1: gemm_q8 (interleaved half-tile ownership, vmcnt(6), concurrent epilogues) vs gemm_p8
cd "$GRAFT_REPO_ROOT"
G=instruct-video-to-video_amd/build/gemm_check
mkdir -p gpurun_out
{
echo "== edge (correctness)"; $G --set edge --tiles 200,230 --iters 3
echo "== big"; $G --set big --tiles 200,230,203,232 --iters 10
echo "== big again (interleaved order)"; $G --set big --tiles 230,200,232,203 --iters 10
echo "== unet B=3"; $G --set unet --tiles 0,200,230,233 --iters 10
} > gpurun_out/r04_run1_gemm_check.txt 2>&1
tail -50 gpurun_out/r04_run1_gemm_check.txt
