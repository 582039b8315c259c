#!/bin/bash
# Round 5, GPU call 2: channel-block-major K order of gemm_r8's stride-1 convolutions (INSV2V_R8_CIM=0/1, same box): correctness against the
# fp32 reference of tools/gemm_check, per-shape timing at B = 30, cycle stamps, the conv / gemm_r8 kernel tests, the new stacked-C3 test,
# the forward's total at B = 60 and its fabric-side traffic.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_run2; mkdir -p $O
cd $R
G=instruct-video-to-video_amd/build/gemm_check
{ for c in 0 1 0 1; do echo "== INSV2V_R8_CIM=$c"; INSV2V_R8_CIM=$c timeout 600 $G --set unet30 --only conv --tiles 240 --iters 5; done
  echo "== edge320 (CIM=1)"; timeout 300 $G --set edge320 --tiles 240 --iters 2
  for c in 0 1; do echo "== stamps INSV2V_R8_CIM=$c"; INSV2V_R8_CIM=$c timeout 300 $G --set unet30 --only "conv L0 320->320" --tiles 244 --iters 2 --stamps; done
} > $O/gemm_check_cim.txt 2>&1
# slack-aware de-phasing (INSV2V_R8_DEPHASE = permille of a tile time over which the workgroups with a tile to spare start late)
{ for d in 0 500 800 1000 0 800; do echo "== INSV2V_R8_DEPHASE=$d"; INSV2V_R8_DEPHASE=$d timeout 600 $G --set unet60 --tiles 240 --iters 5 --nocheck; done
} > $O/gemm_check_dephase.txt 2>&1
cat $O/gemm_check_dephase.txt
cat $O/gemm_check_cim.txt | grep -v "^  block\|^     " | head -80
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv or r8 or q8" > $O/pytest_conv.txt 2>&1; tail -3 $O/pytest_conv.txt
timeout 1500 python -m pytest tests/test_full_size_gpu.py -q -x -k "c3 or c2_unet_forward or stacked_forward" > $O/pytest_full.txt 2>&1; tail -5 $O/pytest_full.txt
for c in "0 0" "1 0" "1 800" "0 0" "1 0" "1 800"; do set -- $c; echo "== INSV2V_R8_CIM=$1 INSV2V_R8_DEPHASE=$2"; INSV2V_R8_CIM=$1 INSV2V_R8_DEPHASE=$2 NB=60 timeout 600 python tools/ab_switches.py default; done > $O/ab_cim_B60.txt 2>&1
grep -v amdgpu $O/ab_cim_B60.txt
cd /tmp; export TMPDIR=/tmp
for c in 0 1; do
  INSV2V_R8_CIM=$c NB=60 timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc$c -o t -- python $R/tools/profile_unet.py > $O/pmc$c.log 2>&1
  DB=$(find $O/pmc$c -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/pmc_forward_traffic.py $DB > $O/pmc_traffic_cim$c.txt 2>&1; cat $O/pmc_traffic_cim$c.txt
  rm -rf $O/pmc$c
done
