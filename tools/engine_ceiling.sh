#!/bin/bash
# the GEMM engine's ceiling on ONE box - the guide's 256^2 8-phase + st_16x32 template (tools/gemm_guide_8phase.hip),
# torch.matmul (hipBLASLt) and the product's gemm_q8 / gemm_r8, same uniform [-1,1) operands, 8192^3 and 4096^3, rocm-smi power / shader
# clock sampled beside each (0.5 s samples, idle socket power subtracted for pJ/FLOP).  Output: gpurun_out/${TAG}_ceiling/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${TAG:-r06}; O=$R/gpurun_out/${TAG}_ceiling${MODE:+_$MODE}; mkdir -p $O
cd $R
SEC=${SEC:-6}
B=$R/instruct-video-to-video_amd/build
[ -x $B/gemm_guide_8phase ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/gemm_guide_8phase.hip -o $B/gemm_guide_8phase
G=$B/gemm_check
( while true; do echo "smi $(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \t' ' ' | tr '\n' ';')"; sleep 0.5; done ) > $O/smi.log 2>&1 &
SMI=$!
stamp() { echo "== $1 $(date +%s.%N | cut -c1-14)"; }
{
  python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null   # page torch in before anything is timed
  stamp "idle0"; sleep 4
  if [ "$MODE" = "shape" ]; then
    # round-6 A/B of the MFMA shape: bare chains of both shapes (one and two waves per SIMD), and the SAME 8-phase structure with either shape
    for w in 1 2; do
      stamp "bare_32x32x16_w$w"; $B/mfma_rate shape32 $SEC $w
      stamp "idle"; sleep 3
      stamp "bare_16x16x32_w$w"; $B/mfma_rate shape16 $SEC $w
      stamp "idle"; sleep 3
    done
    for rep in 1 2; do
      stamp "guide_f16_8192_rep$rep"; $B/gemm_guide_8phase f16 8192 $SEC 1
      stamp "idle"; sleep 3
      stamp "guide_f16x32_8192_rep$rep"; $B/gemm_guide_8phase f16x32 8192 $SEC 1
      stamp "idle"; sleep 3
    done
    stamp "gemm_q8_f16_8192"; $G --set big --tiles 230 --only "8192" --uniform --nocheck --iters $((SEC * 1000000 / 950))
    stamp "idle"; sleep 3
  elif [ "$MODE" = "lds" ]; then
    # round-6 ablation: what the LDS fragment reads cost in the template's structure (24 / 16 / 8 ds_read_b128 per wave and K tile, same MFMAs;
    # 16 = the traffic per MFMA of a 4-wave workgroup with 128 x 128 wave tiles), beside hipBLASLt on the same box
    for rep in 1 2; do
      for v in f16 f16lds1 f16lds2; do
        stamp "guide_${v}_8192_rep$rep"; $B/gemm_guide_8phase $v 8192 $SEC 1
        stamp "idle"; sleep 3
      done
    done
    stamp "hipblaslt_f16_8192"; python tools/engine_ceiling_matmul.py f16 8192 $SEC 2>/dev/null
    stamp "idle"; sleep 3
  elif [ "$MODE" = "engine" ]; then
    # the product's engine after the switch to 16x16x32: gemm_q8 (230 product schedule, 231 requests in the load segments, 232 no epilogue),
    # gemm_r8 (240, 242 no epilogue) and the guide template, 8192^3-class problems
    for t in 230 231 232; do
      stamp "gemm_q8_t${t}_8192"; $G --set big --tiles $t --only "8192" --uniform --nocheck --iters $((SEC * 1000000 / 900))
      stamp "idle"; sleep 3
    done
    for t in 240 242; do
      stamp "gemm_r8_t${t}_8192x8320"; $G --set big320 --tiles $t --only "8192" --uniform --nocheck --iters $((SEC * 1000000 / 950))
      stamp "idle"; sleep 3
    done
    stamp "guide_f16_8192"; $B/gemm_guide_8phase f16 8192 $SEC 1
    stamp "idle"; sleep 3
  else
  for n in 8192 4096; do
    for dt in f16 bf16; do
      stamp "guide_${dt}_$n"; $B/gemm_guide_8phase $dt $n $SEC 1
      stamp "idle"; sleep 3
      stamp "hipblaslt_${dt}_$n"; python tools/engine_ceiling_matmul.py $dt $n $SEC 2>/dev/null
      stamp "idle"; sleep 3
    done
    stamp "gemm_q8_f16_$n"; $G --set big --tiles 230 --only "$n" --uniform --nocheck --iters $((SEC * 1000000 / (n == 8192 ? 950 : 130)))
    stamp "idle"; sleep 3
    stamp "gemm_q8_noepi_f16_$n"; $G --set big --tiles 232 --only "$n" --uniform --nocheck --iters $((2 * 1000000 / (n == 8192 ? 950 : 130)))
    stamp "idle"; sleep 3
    stamp "gemm_r8_f16_$n"; $G --set big320 --tiles 240 --only "$n" --uniform --nocheck --iters $((SEC * 1000000 / (n == 8192 ? 950 : 130)))
    stamp "idle"; sleep 3
  done
  fi
  stamp "end"
} > $O/run.log 2>&1
kill $SMI
python - $O/run.log $O/smi.log > $O/summary.txt <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
marks = [(m.group(1), float(m.group(2)), m.end()) for m in re.finditer(r'== (\S+) ([0-9.]+)', txt)]
smp = []
for l in open(sys.argv[2]):
    t = re.match(r'smi ([0-9.]+)', l); p = re.search(r'Power[^:]*:\s*([0-9.]+)', l); c = re.search(r'sclk[^(]*\((\d+)Mhz\)', l)
    if t and p and c: smp.append((float(t.group(1)), float(p.group(1)), int(c.group(1))))
idle = [p for t, p, c in smp if marks and marks[0][1] + 0.5 <= t <= marks[1][1] - 0.5] if len(marks) > 1 else []
idle_w = sum(idle) / len(idle) if idle else 250.0
print(f"idle socket power {idle_w:.0f} W (subtracted for the dynamic energy per FLOP)")
print(f"{'load':28s} {'TFLOP/s':>9s} {'power W':>8s} {'max W':>7s} {'sclk MHz':>9s} {'pJ/FLOP dyn':>12s}")
for (name, t0, e0), (_, t1, _e) in zip(marks, marks[1:]):
    if name.startswith("idle"): continue
    seg = txt[e0:txt.find("==", e0)] if txt.find("==", e0) > 0 else txt[e0:]
    tf = [float(x) for x in re.findall(r'([0-9.]+) TFLOP/s', seg)] or [float(x) for x in re.findall(r'([0-9.]+)TF', seg)]
    w = [(p, c) for t, p, c in smp if t0 + 1.5 <= t <= t1 - 0.5]
    if not w or not tf: print(f"{name:28s} no samples / no rate: {seg.strip()[:120]}"); continue
    hi = [x for x in w if x[0] >= 0.9 * max(p for p, _ in w)]
    pw = sum(p for p, _ in hi) / len(hi); ck = sum(c for _, c in hi) / len(hi)
    print(f"{name:28s} {tf[0]:9.1f} {pw:8.0f} {max(p for p, _ in w):7.0f} {ck:9.0f} {(pw - idle_w) / tf[0]:12.3f}")
PY
grep -vE "amdgpu.ids" $O/run.log; cat $O/summary.txt
