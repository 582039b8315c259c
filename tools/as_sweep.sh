#!/bin/bash
G=instruct-video-to-video_amd/build/gemm_check
timeout 200 $G --set unet --only "L0 73728x960x320 ln (sp" --tiles 232,233
timeout 200 $G --set unet --only "q   L0" --tiles 232,233
