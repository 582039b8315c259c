#!/bin/bash
# round 4 run 7: (a) startup stagger of gemm_r8's workgroups (INSV2V_R8_STAGGER, shader cycles) on the B = 30 shapes; (b) full GPU test suite
cd "$GRAFT_REPO_ROOT"
G=instruct-video-to-video_amd/build/gemm_check
{
for st in 0 40000 80000 160000; do
  echo "== INSV2V_R8_STAGGER=$st"
  INSV2V_R8_STAGGER=$st timeout 300 $G --set unet30 --only "conv L0 320->320" --tiles 240 --iters 5 --nocheck
  INSV2V_R8_STAGGER=$st timeout 300 $G --set unet30 --only "conv L1 640->640" --tiles 240 --iters 5 --nocheck
  INSV2V_R8_STAGGER=$st timeout 300 $G --set unet30 --only "lin L1 184320x640" --tiles 240 --iters 5 --nocheck
  INSV2V_R8_STAGGER=$st timeout 300 $G --set unet30 --only "lin L0 737280x320x640" --tiles 240 --iters 5 --nocheck
done
} > gpurun_out/r04_run7_stagger.txt 2>&1
cat gpurun_out/r04_run7_stagger.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04_run7_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_run7_pytest.txt
tail -6 gpurun_out/r04_run7_pytest.txt
