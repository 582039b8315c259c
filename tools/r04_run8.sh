#!/bin/bash
# round 4 run 8: concurrent stacks experiment (run_stacked(stacks=...)) and the driver-mode bench lines, one box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
b() { name=$1; shift; python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/r04_run8_$name.json 2> $O/r04_run8_$name.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/r04_run8_$name.json") if l.startswith("{")][-1])
    print("$name", round(d["value"],3), "frames/s", d["config"].get("clip_groups"), d["config"].get("stacked_vs_single_rel_rms"))
except Exception as e:
    print("$name ERR", e); print(open("$O/r04_run8_$name.err").read()[-800:])
PY
}
b base
b stacks2x10 --concurrent-clips 20 --stacks 2
b stacks2x5 --stacks 2
b stacks4x5 --concurrent-clips 20 --stacks 4
b driver --driver-mode
b base2
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --long-video --driver-mode > $O/r04_run8_longvideo_driver.json 2> $O/r04_run8_longvideo_driver.err; tail -c 400 $O/r04_run8_longvideo_driver.json | head -c 300; echo
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --long-video > $O/r04_run8_longvideo.json 2> $O/r04_run8_longvideo.err; python - <<PY
import json
for n in ("longvideo_driver","longvideo"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04_run8_%s.json"%n) if l.startswith("{")][-1]); print(n, round(d["value"],3))
    except Exception as e: print(n,"ERR",e)
PY
