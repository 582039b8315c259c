#!/usr/bin/env python3
"""Cross-check of bench.py's roofline timing (HIP events around every launch, on the launch stream) against rocprofv3's kernel durations.

    rocprofv3 --kernel-trace --stats -d DIR -o e -- env NB=60 python tools/profile_unet.py > per_shape.txt
    python tools/rocprof_vs_events.py DIR/<...>_results.db per_shape.txt [forwards = 1 warm + REPS]

tools/profile_unet.py runs the same eager stacked forward bench.py's roofline leg records (1 untimed + REPS recorded), so every kernel of the
trace belongs to one of those forwards: per family, rocprofv3's (sum of durations, calls) per forward stand beside the HIP-event (sum, launches)
of the fastest recorded forward.  Families as in bench.py: GEMM family = gemm_* / conv_halo / splitk_reduce / ffn_fused / rowlin / tattn* /
xattn*; attention = attn_*; norms + Winograd transforms = gn_* / ln_* / layernorm / wino_*."""
import ast
import re
import sqlite3
import sys

FAMILIES = (("gemm_kernel", ("gemm_", "conv_halo", "splitk_reduce", "ffn_fused", "rowlin", "tattn", "xattn")),
            ("attn_kernel", ("attn_kernel", "attn_short")),
            ("groupnorm", ("gn_", "wino_", "ln_", "layernorm")))
EVENT_KINDS = {"gemm_kernel": ("lin", "conv", "ffn", "rowlin", "wino_gemm", "tattn", "tattn_attn", "xattn", "xattn_attn"),
               "attn_kernel": ("attn",), "groupnorm": ("gn", "gnstats", "wino_in", "wino_out", "ln", "lnstats")}


def main():
    db, per_shape = sys.argv[1], sys.argv[2]
    forwards = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows = sqlite3.connect(db).execute("select name,total_calls,total_duration from top_kernels").fetchall()
    ev = {}
    for line in open(per_shape):
        m = re.match(r"\s*([\d.]+) ms\s+[\d.]+%\s+n=\s*(\d+)\s+.*?(\(.*\))\s*$", line)
        if m:
            kind = ast.literal_eval(m.group(3))[0]
            e = ev.setdefault(kind, [0.0, 0])
            e[0] += float(m.group(1)); e[1] += int(m.group(2))
    print(f"{'family':12s} | rocprofv3 kernel durations, per forward ({forwards} in the trace)  | HIP events, fastest of the recorded forwards")
    for fam, keys in FAMILIES:
        dur = sum(r[2] for r in rows if any(k in r[0] for k in keys) and not (fam == "gemm_kernel" and "attn_kernel" in r[0])) / 1e3 / forwards   # top_kernels reports microseconds
        calls = sum(r[1] for r in rows if any(k in r[0] for k in keys) and not (fam == "gemm_kernel" and "attn_kernel" in r[0])) / forwards
        ems = sum(ev.get(k, [0, 0])[0] for k in EVENT_KINDS[fam]); en = sum(ev.get(k, [0, 0])[1] for k in EVENT_KINDS[fam])
        print(f"{fam:12s} | {dur:8.2f} ms over {calls:6.1f} kernels = {dur / max(calls, 1) * 1e3:7.1f} us each | {ems:8.2f} ms over {en:4d} launches = {ems / max(en, 1) * 1e3:7.1f} us each"
              f" | events / rocprof = {ems / dur:.3f}")


if __name__ == "__main__":
    main()
