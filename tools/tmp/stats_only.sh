R=${GRAFT_REPO_ROOT:-$PWD}; TAG=r06; O=$R/gpurun_out/${TAG}final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o ${TAG}f -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
echo "rc=$?"
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
ls $O/stats | head; head -5 $O/kernel_stats.csv | cut -c1-200; tail -c 300 $O/bench_under_rocprof.json
cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
