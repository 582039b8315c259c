#!/bin/bash
# Round 3, VERDICT item 7(iv): reconcile the matrix-pipe ceiling measured here (1.84 PFLOP/s) with the guide's 2 495 TF figure by
# logging the shader clock while tools/mfma_rate runs nothing but independent v_mfma_f32_32x32x16_f16 chains on ZERO and on
# non-zero operands.  Output: gpurun_out/<dir>/sclk_probe.txt (rate lines interleaved with rocm-smi clock samples).
R=${GRAFT_REPO_ROOT:-$PWD}; O=${1:-$R/gpurun_out/r03_sclk}; mkdir -p $O
B=$R/instruct-video-to-video_amd/build/mfma_rate
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/mfma_rate.hip -o $B
( while true; do echo "smi $(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | tr -s ' ' | tr '\n' ';')"; sleep 0.5; done ) > $O/smi.log 2>&1 &
SMI=$!
sleep 1
{ echo "== idle"; sleep 2; echo "== zero operands $(date +%s.%N | cut -c1-14)"; $B zero 6; echo "== non-zero operands $(date +%s.%N | cut -c1-14)"; $B rand 6; echo "== end $(date +%s.%N | cut -c1-14)"; } > $O/rate.log 2>&1
kill $SMI
cat $O/rate.log; grep -c smi $O/smi.log; awk 'NR%2==1' $O/smi.log | head -40
