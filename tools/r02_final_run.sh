#!/bin/bash
# Round-2 evidence run (one gpurun call): GPU test suite, bench line, rocprofv3 kernel stats of the same command, PMC traffic pass,
# per-shape eager profile, long-video bench.  Outputs land in gpurun_out/r02f/ and are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02f; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt )
( cd $R && timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json )
( cd $R && timeout 600 python tools/profile_unet.py > $O/unet_forward_per_shape.txt 2>&1; head -3 $O/unet_forward_per_shape.txt )
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o r02f -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc -o t -- python $R/tools/profile_unet.py > $O/pmc.log 2>&1
DB=$(find $O/pmc -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/pmc_forward_traffic.py $DB $O/pmc_forward_traffic.json 3 16 32 48 > $O/pmc_forward_traffic.txt 2>&1; cat $O/pmc_forward_traffic.txt
( cd $R && timeout 600 python bench.py --long-video --steps 1 --warmup 1 > $O/bench_long_video.json 2> $O/bench_long.err; tail -c 400 $O/bench_long_video.json )
( cd $R && timeout 300 python tools/time_unet_streams.py > $O/unet_step_streams_vs_batched.txt 2>&1; cat $O/unet_step_streams_vs_batched.txt )
# keep the merged output small: export the per-kernel summary, then drop the sqlite files (tens of MB)
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
find $O -name "*.db" -delete; find $O/stats -name "*kernel_trace.csv" -delete
ls -la $O $O/stats/* | head -30
