#!/bin/bash
# round 4 run 2: where does a barrier interval of the 8-phase loop go?  s_memtime stamps (tiles 234 / 235) and the
# LDS-DMA-inside-the-MFMA-segment variant (236 no epilogue, 237 product) against 230 / 232
cd "$GRAFT_REPO_ROOT"
G=instruct-video-to-video_amd/build/gemm_check
mkdir -p gpurun_out
{
echo "== big, timing"; $G --set big --tiles 230,237,232,236,200 --iters 10
echo "== big, timing (other order)"; $G --set big --tiles 236,232,237,230 --iters 10
echo "== stamps VAR 0"; $G --set big --only 8192 --tiles 234 --iters 2 --nocheck --stamps
echo "== stamps VAR 1"; $G --set big --only 8192 --tiles 235 --iters 2 --nocheck --stamps
} > gpurun_out/r04_run2_gemm_stamps.txt 2>&1
cat gpurun_out/r04_run2_gemm_stamps.txt | head -150
