cd $GRAFT_REPO_ROOT
for v in st8 st16 st8 st16; do cp tools/_tmp_lib_$v.so instruct-video-to-video_amd/insv2v/libinsv2v_hip.so; echo $v; python tools/bench_xattn1280.py 2>&1 | grep -v amdgpu | head -1;  python tools/bench_attn_b60.py 2>&1 | grep -v amdgpu | tail -2 | head -1; done
