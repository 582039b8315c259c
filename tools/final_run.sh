#!/bin/bash
# End-of-round evidence run (one gpurun call; TAG=r06 tools/final_run.sh): GPU test suite, the bench lines (driver's command line, default, driver mode, single clip, the
# product's small stacks B = 6 / 12, C3 stacked, C5, long video), rocprofv3 kernel stats of the bench command, PMC traffic pass of the
# forward at the benched batches, per-shape eager profiles, stand-alone GEMM harness tables.  Outputs land in gpurun_out/${TAG}final/ and
# are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-r06}; O=$R/gpurun_out/${TAG}final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt )
( cd $R && INSV2V_GRAPH_PURGE=destroy timeout 2400 python -m pytest tests -q -m gpu -k "not raft" > $O/pytest_gpu_graph_purge_destroy.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu_graph_purge_destroy.txt; tail -3 $O/pytest_gpu_graph_purge_destroy.txt )
for nb in 15 60; do   # fabric-side traffic at both benched batches FIRST: bench.py loads the JSON it writes (REPS=1: one warm + one counted forward pair)
  REPS=1 NB=$nb timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc$nb -o t -- python $R/tools/profile_unet.py > $O/pmc$nb.log 2>&1
  DB=$(find $O/pmc$nb -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/pmc_forward_traffic.py $DB $O/pmc_forward_traffic.json $nb 16 32 48 > $O/pmc_forward_traffic_B$nb.txt 2>&1; cat $O/pmc_forward_traffic_B$nb.txt
done
cp $O/pmc_forward_traffic.json $R/profiles/pmc_forward_traffic.json
# the driver's command line, with the board's power / shader clock sampled beside it (the kernels run power-limited)
( while true; do echo "smi $(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \t' ' ' | tr '\n' ';')"; sleep 0.5; done ) > $O/smi_bench_steps20.log 2>&1 &
SMI=$!
( cd $R && timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 700 $O/bench_steps20.json )   # the driver's command line
kill $SMI
python - $O/smi_bench_steps20.log > $O/smi_bench_steps20_summary.txt <<'PY'
import re, sys
pw, ck = [], []
for l in open(sys.argv[1]):
    m = re.search(r'Power[^:]*:\s*([0-9.]+)', l); c = re.search(r'sclk[^(]*\((\d+)Mhz\)', l)
    if m and c: pw.append(float(m.group(1))); ck.append(int(c.group(1)))
busy = [(p, c) for p, c in zip(pw, ck) if p > 0.6 * max(pw)] if pw else []
print(f"{len(pw)} samples; while busy ({len(busy)} samples): power mean {sum(p for p, _ in busy) / max(len(busy), 1):.0f} W (max {max(pw) if pw else 0:.0f} W), sclk mean {sum(c for _, c in busy) / max(len(busy), 1):.0f} MHz (min {min((c for _, c in busy), default=0)}, max {max((c for _, c in busy), default=0)})")
PY
cat $O/smi_bench_steps20_summary.txt
( cd $R && timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json )
( cd $R && timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --driver-mode > $O/bench_driver_mode.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver_mode.json )
( cd $R && timeout 600 python bench.py --no-cpu-baseline --concurrent-clips 1 --steps 2 > $O/bench_single_clip.json 2> $O/bench_single.err; tail -c 300 $O/bench_single_clip.json )
# the stacks the product meets below the bench's: C4's two units per GPU (B = 6), one video's four prompts (B = 12)
( cd $R && timeout 600 python bench.py --no-cpu-baseline --concurrent-clips 2 --steps 4 --warmup 2 > $O/bench_b6.json 2> $O/bench_b6.err; tail -c 300 $O/bench_b6.json )
( cd $R && timeout 600 python bench.py --no-cpu-baseline --concurrent-clips 4 --steps 8 --warmup 2 > $O/bench_b12.json 2> $O/bench_b12.err; tail -c 300 $O/bench_b12.json )
( cd $R && timeout 900 python bench.py --flow-correction --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json )
( cd $R && timeout 600 python bench.py --long-video --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_long_video.json 2> $O/bench_long.err; tail -c 300 $O/bench_long_video.json )
( cd $R && timeout 900 python bench.py --long-video --driver-mode --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_long_video_driver.json 2> $O/bench_long_driver.err; tail -c 300 $O/bench_long_video_driver.json )
( cd $R && timeout 900 python bench.py --frames 24 --height 384 --width 512 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.json )
for nb in 60 30 12 6 3; do ( cd $R && NB=$nb timeout 600 python tools/profile_unet.py > $O/unet_forward_per_shape_B$nb.txt 2>&1; head -2 $O/unet_forward_per_shape_B$nb.txt | tail -1 ); done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o ${TAG}f -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
G=$R/instruct-video-to-video_amd/build/gemm_check
{ echo "== big (8192^3, 4096^3): 230 gemm_q8, 232 without epilogue"; $G --set big --tiles 230,232 --iters 10
  echo "== unet60 (B = 60): 0 dispatch, 240 gemm_r8, 242 gemm_r8 without epilogue"; $G --set unet60 --tiles 0,240,242 --iters 5
  echo "== unet30 (B = 30): 0 dispatch, 5 128x128 tile, 230 gemm_q8, 240 gemm_r8"; $G --set unet30 --tiles 0,5,230,240 --iters 5
  echo "== edge / edge320"; $G --set edge --tiles 230 --iters 2; $G --set edge320 --tiles 240 --iters 2; } > $O/gemm_check.txt 2>&1
tail -30 $O/gemm_check.txt
( cd $R && python tools/bench_attn.py > $O/bench_attn.txt 2>&1; head -4 $O/bench_attn.txt )
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
