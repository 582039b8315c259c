#!/bin/bash
G=instruct-video-to-video_amd/build/gemm_check
timeout 300 $G --set unet --tiles 0
timeout 300 $G --set unet1 --only "+res" --tiles 0
timeout 300 $G --set edge --tiles 0,5,4,2
