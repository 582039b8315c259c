#!/bin/bash
# Board power / shader clock (rocm-smi, 0.5 s samples) beside (a) bare MFMA chains on zero and on non-zero operands
# (tools/mfma_rate), (b) the GEMM engine alone (gemm_q8 at 8192^3 without / with non-power-of-two K, gemm_r8 on a level-0 convolution),
# (c) the fused feed-forward, (d) GroupNorm (HBM-bound), (e) eager B = 60 forwards: energy per FLOP of each.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-r06}_power; mkdir -p $O
cd $R
B=$R/instruct-video-to-video_amd/build/mfma_rate
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/mfma_rate.hip -o $B
G=$R/instruct-video-to-video_amd/build/gemm_check
( while true; do echo "smi $(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \t' ' ' | tr '\n' ';')"; sleep 0.5; done ) > $O/smi.log 2>&1 &
SMI=$!
stamp() { echo "== $1 $(date +%s.%N | cut -c1-14)"; }
{
  sleep 2
  if [ -z "$GEMM_ONLY" ]; then
  stamp "mfma_zero";  $B zero 8
  stamp "idle1"; sleep 3
  stamp "mfma_rand";  $B rand 8
  stamp "idle2"; sleep 3
  fi
  stamp "gemm_q8_8192";  $G --set big --tiles 230 --iters 6000 --nocheck --only 8256
  stamp "idle3"; sleep 3
  stamp "conv_r8_L0";  $G --set unet60 --tiles 240 --iters 1800 --nocheck --only 640-
  stamp "idle4"; sleep 3
  if [ -z "$GEMM_ONLY" ]; then
  stamp "ffn";  NB=60 python tools/bench_rows_ab.py 2>/dev/null | grep -E "ffn|rowlin M0 320->320   "
  stamp "idle5"; sleep 3
  stamp "forward";  NB=60 REPS=12 python tools/profile_unet.py | head -2 | tail -1
  fi
  stamp "end"
} > $O/run.log 2>&1
kill $SMI
python - $O/run.log $O/smi.log > $O/summary.txt <<'PY'
import re, sys
marks = [(m.group(1), float(m.group(2))) for m in re.finditer(r'== (\S+) ([0-9.]+)', open(sys.argv[1]).read())]
smp = []
for l in open(sys.argv[2]):
    t = re.match(r'smi ([0-9.]+)', l); p = re.search(r'Power[^:]*:\s*([0-9.]+)', l); c = re.search(r'sclk[^(]*\((\d+)Mhz\)', l)
    if t and p and c: smp.append((float(t.group(1)), float(p.group(1)), int(c.group(1))))
for (name, t0), (_, t1) in zip(marks, marks[1:]):
    w = [(p, c) for t, p, c in smp if t0 + 1.0 <= t <= t1 - 0.3]
    if not w: print(f"{name:14s} no samples"); continue
    hi = [x for x in w if x[0] >= 0.8 * max(p for p, _ in w)]
    print(f"{name:14s} {len(w):3d} samples  power mean {sum(p for p, _ in w) / len(w):6.0f} W  max {max(p for p, _ in w):6.0f} W | top-20%-power samples: {sum(p for p, _ in hi) / len(hi):6.0f} W at sclk {sum(c for _, c in hi) / len(hi):5.0f} MHz")
PY
cat $O/run.log | grep -vE "amdgpu.ids"; cat $O/summary.txt
