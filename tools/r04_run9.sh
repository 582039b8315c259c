#!/bin/bash
# round 4 run 9: the other bench modes with the round-4 kernels (C5, C3, single clip, default 5 steps)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
b() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > $O/r04_run9_$name.json 2> $O/r04_run9_$name.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r04_run9_$name.json") if l.startswith("{")][-1])
    print("$name", round(d["value"],3), "frames/s", d["config"].get("clip_groups"), "frac", round(d["roofline"]["frac"],3))
except Exception as e:
    print("$name ERR", e); print(open("$O/r04_run9_$name.err").read()[-600:])
PY
}
b c5 --frames 24 --height 384 --width 512 --steps 6 --warmup 2
b c3 --flow-correction --steps 4 --warmup 2
b single --concurrent-clips 1 --steps 4 --warmup 2
b default5
