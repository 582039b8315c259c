#!/bin/bash
# Round 5, GPU call 3: the new paths - RAFT estimator tests, the C4-unit / DDPM / stacked-C3 full-size golden tests, the hipGraph
# destroy / recapture repros (stand-alone HIP, torch, and the product's own test order with INSV2V_GRAPH_PURGE=destroy), C3 stacked bench.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_run3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_raft_gpu.py -q -x -s > $O/pytest_raft.txt 2>&1; tail -25 $O/pytest_raft.txt
timeout 1800 python -m pytest tests/test_full_size_gpu.py -q -x -s -k "c4_unit or ddpm or c3" > $O/pytest_full_new.txt 2>&1; grep "parity\|passed\|failed\|Error" $O/pytest_full_new.txt | tail -25
B=instruct-video-to-video_amd/build/graph_recapture
for m in 0 1 2; do echo "== graph_recapture mode $m"; timeout 120 $B $m 4 2>&1 | tail -2; echo "rc=$?"; done > $O/graph_recapture_hip.txt 2>&1
for m in keep destroy destroy_empty; do echo "== torch mode $m"; timeout 300 python tools/repro/graph_recapture_torch.py $m 4 2>&1 | tail -2; echo "rc=${PIPESTATUS[0]}"; done > $O/graph_recapture_torch.txt 2>&1
cat $O/graph_recapture_hip.txt $O/graph_recapture_torch.txt
( INSV2V_GRAPH_PURGE=destroy timeout 1500 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py -q -x -k "c2_unet_forward or c1_exact or graph or pipeline or ddim" > $O/pytest_purge_destroy.txt 2>&1; echo "rc=$?" >> $O/pytest_purge_destroy.txt ); tail -5 $O/pytest_purge_destroy.txt
timeout 900 python bench.py --flow-correction --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_stacked.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3_stacked.json
