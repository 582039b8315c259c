#!/bin/bash
# Round-4 evidence run (one gpurun call): GPU test suite, the bench lines (driver's command line, default, driver mode, single clip, C3, C5,
# long video), rocprofv3 kernel stats of the bench command, PMC traffic pass of the forward at the benched batches, PMC pipe-utilisation
# counters of the round-4 kernels, per-shape eager profiles, stand-alone GEMM harness tables.  Outputs land in gpurun_out/r04final/ and are
# copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt )
for nb in 15 60; do   # HBM-side traffic at both benched batches (default run: 5 clips, B = 15; driver's command line: 20 clips, B = 60) FIRST: bench.py loads the JSON it writes
  NB=$nb timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc$nb -o t -- python $R/tools/profile_unet.py > $O/pmc$nb.log 2>&1
  DB=$(find $O/pmc$nb -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/pmc_forward_traffic.py $DB $O/pmc_forward_traffic.json $nb 16 32 48 > $O/pmc_forward_traffic_B$nb.txt 2>&1; cat $O/pmc_forward_traffic_B$nb.txt
done
cp $O/pmc_forward_traffic.json $R/profiles/pmc_forward_traffic.json
( cd $R && timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 700 $O/bench_steps20.json )   # the driver's command line
( cd $R && timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json )
( cd $R && timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --driver-mode > $O/bench_driver_mode.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver_mode.json )
( cd $R && timeout 600 python bench.py --no-cpu-baseline --concurrent-clips 1 --steps 2 > $O/bench_single_clip.json 2> $O/bench_single.err; tail -c 300 $O/bench_single_clip.json )
( cd $R && timeout 600 python bench.py --long-video --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_long_video.json 2> $O/bench_long.err; tail -c 300 $O/bench_long_video.json )
( cd $R && timeout 900 python bench.py --long-video --driver-mode --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_long_video_driver.json 2> $O/bench_long_driver.err; tail -c 300 $O/bench_long_video_driver.json )
( cd $R && timeout 900 python bench.py --frames 24 --height 384 --width 512 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.json )
( cd $R && timeout 600 python bench.py --flow-correction --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json )
( cd $R && NB=60 timeout 600 python tools/profile_unet.py > $O/unet_forward_per_shape_B60.txt 2>&1; head -3 $O/unet_forward_per_shape_B60.txt )
( cd $R && NB=30 timeout 600 python tools/profile_unet.py > $O/unet_forward_per_shape_B30.txt 2>&1; head -3 $O/unet_forward_per_shape_B30.txt )
( cd $R && NB=15 timeout 600 python tools/profile_unet.py > $O/unet_forward_per_shape_B15.txt 2>&1; head -3 $O/unet_forward_per_shape_B15.txt )
( cd $R && NB=3 timeout 600 python tools/profile_unet.py > $O/unet_forward_per_shape_B3.txt 2>&1; head -3 $O/unet_forward_per_shape_B3.txt )
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o r04f -- python $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d $O/pmc_mfma -o m -- python $R/tools/pmc_rows.py > $O/pmc_mfma.log 2>&1
DBM=$(find $O/pmc_mfma -name "*.db" | head -1); [ -n "$DBM" ] && python $R/tools/pmc_report.py $DBM > $O/pmc_rows_mfma.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_lds -o l -- python $R/tools/pmc_rows.py > $O/pmc_lds.log 2>&1
DBL=$(find $O/pmc_lds -name "*.db" | head -1); [ -n "$DBL" ] && python $R/tools/pmc_report.py $DBL > $O/pmc_rows_lds.txt 2>&1
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do   # the spatial self-attention launch
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_attn -o a -- python $R/tools/pmc_attn.py > $O/pmc_attn.log 2>&1
  DBA=$(find $O/pmc_attn -name "*.db" | head -1); [ -n "$DBA" ] && python $R/tools/pmc_report.py $DBA | grep attn_kernel >> $O/pmc_attn.txt; rm -rf $O/pmc_attn
done
cat $O/pmc_attn.txt
( cd $R && python tools/bench_attn.py > $O/bench_attn.txt 2>&1; head -4 $O/bench_attn.txt )
python $R/tools/pmc_pipe_summary.py $O/pmc_rows_mfma.txt $O/pmc_rows_lds.txt > $O/pmc_pipe_utilisation.txt 2>&1; head -60 $O/pmc_pipe_utilisation.txt
G=$R/instruct-video-to-video_amd/build/gemm_check
{ echo "== big (8192^3, 4096^3): 230 gemm_q8, 231 gemm_q8 with the requests in the load segments, 200 gemm_p8 (round 3), 232 / 203 without epilogue"; $G --set big --tiles 230,231,200,232,203 --iters 10
  echo "== stride"; $G --set stride --tiles 230,232,200 --iters 10 --nocheck
  echo "== unet (B = 3): 0 dispatch, 200 gemm_p8, 230 gemm_q8"; $G --set unet --tiles 0,200,230 --iters 10
  echo "== unet30 (B = 30): 0 dispatch, 5 128x128 tile, 100 patch-tiled conv, 230 gemm_q8, 240 gemm_r8, 242 gemm_r8 without epilogue"; $G --set unet30 --tiles 0,5,100,230,240,242 --iters 5
  echo "== edge / edge320"; $G --set edge --tiles 230 --iters 2; $G --set edge320 --tiles 240 --iters 2; } > $O/gemm_check.txt 2>&1
tail -30 $O/gemm_check.txt
$R/instruct-video-to-video_amd/build/phase_rate 2000 > $O/phase_rate.txt 2>&1
$R/instruct-video-to-video_amd/build/store_rate > $O/store_rate.txt 2>&1
DBS=$(find $O/stats -name "*.db" | head -1)
[ -n "$DBS" ] && python - "$DBS" "$O/kernel_stats.csv" <<'PY'
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
PY
find $O -name "*.db" -delete; find $O/stats -name "*kernel_trace.csv" -delete 2>/dev/null
ls -la $O | head -60
