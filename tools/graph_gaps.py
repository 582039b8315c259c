#!/usr/bin/env python3
"""Sum of kernel durations vs wall time of the captured UNet graph's batched replays, from a rocprofv3 --kernel-trace sqlite file of
tools/time_unet_streams.py (second half of the run = batched single-stream graph replays).  Under the profiler the dispatches are
serialised back to back (gaps read 0); the useful number is the SUM OF DURATIONS per step (28.8 ms on the run of round 2) against the
un-profiled step time the same script prints (31.05 ms on that box): the difference is launch / dependency overhead.
    rocprofv3 --kernel-trace -d out -o t -- python tools/time_unet_streams.py ; python tools/graph_gaps.py out/.../t_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, name, stream_id from kernels order by start").fetchall()
print(len(rows), "dispatches")
# the batched replays are the tail of the run: take the last 5 x 618 kernels
n = 618 * 5
seg = rows[-n:]
gaps = sorted(seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1))
busy = sum(e - s for s, e, *_ in seg)
wall = seg[-1][1] - seg[0][0]
print(f"last {n} kernels: wall {wall / 1e6:.2f} ms, sum of durations {busy / 1e6:.2f} ms, idle {100 * (wall - busy) / wall:.1f} %")
print("gap ns: median %d  p25 %d  p75 %d  p90 %d  mean %.0f" % (gaps[len(gaps) // 2], gaps[len(gaps) // 4], gaps[3 * len(gaps) // 4], gaps[int(len(gaps) * .9)],
                                                           sum(gaps) / len(gaps)))
