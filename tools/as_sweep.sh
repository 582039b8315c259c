#!/bin/bash
G=instruct-video-to-video_amd/build/gemm_check
timeout 300 $G --set unet1 --tiles 0
