#!/bin/bash
G=instruct-video-to-video_amd/build/gemm_check
for pad in 0 16384; do for sp in 3 5; do echo "== ldspad $pad split $sp"; INSV2V_AS_LDSPAD=$pad INSV2V_AS_SPLIT=$sp timeout 120 $G --set unet --only "L0 73728x960x320 ln (sp" --tiles 230,231 | head -3; done; done
