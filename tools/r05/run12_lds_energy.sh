#!/bin/bash
# Round 5: what LDS operand reads cost at the power cap: sustained 32x32x16 MFMA chains (two waves per SIMD, 160 accumulators per wave) with
# R fresh ds_read_b128 fragments per ten MFMAs (tools/mfma_rate lds R), rocm-smi power / clock beside them.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_lds_energy; mkdir -p $O
B=$R/instruct-video-to-video_amd/build/mfma_rate
( while true; do echo "smi $(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \t' ' ' | tr '\n' ';')"; sleep 0.5; done ) > $O/smi.log 2>&1 &
SMI=$!
{ sleep 2
  for r in 0 4 7 10 0 5 7; do echo "== lds_R$r $(date +%s.%N | cut -c1-14)"; $B lds $r 7 | awk 'NR%3==0'; echo "== idle $(date +%s.%N | cut -c1-14)"; sleep 3; done
  echo "== end $(date +%s.%N | cut -c1-14)"; } > $O/run.log 2>&1
kill $SMI
python - $O/run.log $O/smi.log > $O/summary.txt <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
marks = [(m.group(1), float(m.group(2))) for m in re.finditer(r'== (\S+) ([0-9.]+)', txt)]
smp = []
for l in open(sys.argv[2]):
    t = re.match(r'smi ([0-9.]+)', l); p = re.search(r'Power[^:]*:\s*([0-9.]+)', l); c = re.search(r'sclk[^(]*\((\d+)Mhz\)', l)
    if t and p and c: smp.append((float(t.group(1)), float(p.group(1)), int(c.group(1))))
for (name, t0), (_, t1) in zip(marks, marks[1:]):
    if name == "idle": continue
    w = [(p, c) for t, p, c in smp if t0 + 1.5 <= t <= t1 - 0.3]
    seg = txt[txt.index(f"== {name} {t0:.3f}"[:len(name) + 8]):]
    rates = [float(x) for x in re.findall(r'([0-9.]+) TF/s', seg.split("== idle")[0])]
    tf = sum(rates[1:]) / max(len(rates[1:]), 1) if len(rates) > 1 else (rates[0] if rates else 0)
    if w:
        pw = sum(p for p, _ in w) / len(w); ck = sum(c for _, c in w) / len(w)
        print(f"{name:9s} {tf:7.1f} TF/s  power {pw:6.0f} W  sclk {ck:5.0f} MHz  dynamic {(pw - 250) / tf:5.3f} pJ/FLOP")
PY
cat $O/summary.txt
