#!/usr/bin/env python3
"""Same-box, same-process A/B of the host-side kernel-choice switches of insv2v/unet.py at a given stacked batch.

For every configuration the UNet is rebuilt from ONE synthetic state dict with the switches set (they are module constants read when the
weights are packed), one eager forward warms up, the next is recorded with HIP events around every launch (ops.set_launch_recorder) and
the sum of the launch times is printed next to the per-family sums.  In the stacked-clip mode the captured graph's wall time equals this
sum, so the totals rank the configurations the way the bench would.

usage: NB=60 python tools/ab_switches.py default FUSE_FFN=0 ROWLIN_640=0 FUSE_TATTN=0,FUSE_TATTN_640=0 ...
       (a configuration = comma-separated NAME=0/1 pairs of the module constants of insv2v/unet.py; "default" = none)
       PER_SHAPE=dir writes the per-shape table of every configuration into dir/<config>.txt
"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import synth, shapes, ops, unet as unet_mod  # noqa: E402
from insv2v.inference import GraphedUNet  # noqa: E402

NB = int(os.environ.get("NB", 60))
F, h, w = int(os.environ.get("F", 16)), int(os.environ.get("LH", 32)), int(os.environ.get("LW", 48))
REPS = int(os.environ.get("REPS", 2))
OUT = os.environ.get("PER_SHAPE")
DEFAULTS = {k: getattr(unet_mod, k) for k in dir(unet_mod) if k.isupper() and isinstance(getattr(unet_mod, k), bool)}


def family(tag):
    k = tag[0]
    if k in ("lin", "conv", "wino_gemm"):
        return "gemm/conv"
    if k in ("ffn", "rowlin", "tattn", "tattn_attn", "xattn", "xattn_attn"):
        return "row kernels"
    if k == "attn":
        return "attention"
    if k in ("gn", "gnstats", "lnstats", "ln", "copy", "wino_in", "wino_out"):
        return "norm"
    return "other"


def run(name, sd):
    for k, v in DEFAULTS.items():
        setattr(unet_mod, k, v)
    if name != "default":
        for kv in name.split(","):
            k, v = kv.split("=")
            if k not in DEFAULTS:
                raise SystemExit(f"unknown switch {k}; known: {sorted(DEFAULTS)}")
            setattr(unet_mod, k, v != "0")
    net = unet_mod.UNet3DConditionModel(**synth.UNET_FULL, device="cuda:0").load_state_dict(sd)
    r = GraphedUNet(net, NB, F, h, w, 77, use_graph=False)
    r.set_context(synth.synth_input("p.ctx", (NB, 77, 768)))
    r.x_in.normal_()
    r.t.fill_(500.0)
    r.run()
    torch.cuda.synchronize()
    totals = []
    for _ in range(REPS):
        rec = []
        ops.set_launch_recorder(rec)
        r.run()
        torch.cuda.synchronize()
        ops.set_launch_recorder(None)
        groups, fams = {}, {}
        for _, work, e0, e1, tag in rec:
            ms = e0.elapsed_time(e1)
            g = groups.setdefault(tag, [0, 0.0, 0.0])
            g[0] += 1
            g[1] += ms
            g[2] += work
            fams[family(tag)] = fams.get(family(tag), 0.0) + ms
        totals.append((sum(g[1] for g in groups.values()), len(rec), groups, fams))
    tot, n, groups, fams = min(totals, key=lambda t: t[0])
    print(f"{name:44s} B={NB:3d} total {tot:8.2f} ms  {n:4d} launches  " + "  ".join(f"{k} {v:7.2f}" for k, v in sorted(fams.items())), flush=True)
    if OUT:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, f"B{NB}_{name.replace('=', '').replace(',', '_')}.txt"), "w") as f:
            f.write(f"B={NB} F={F} {h}x{w} [{name}]: total {tot:.2f} ms over {n} launches\n")
            for tag, (c, ms, work) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{ms:8.3f} ms {100 * ms / tot:5.1f}%  n={c:3d}  {ms / c * 1e3:8.1f} us/launch  {work / ms / 1e9 if ms else 0:8.1f} TF/s  {tag}\n")
    del r, net
    gc.collect()
    torch.cuda.empty_cache()
    return tot


if __name__ == "__main__":
    sd = synth.synth_state_dict(shapes.unet_shapes(**synth.UNET_FULL))
    configs = sys.argv[1:] or ["default"]
    base = None
    for c in configs:
        try:
            t = run(c, sd)
        except Exception as e:   # a configuration the kernels refuse at this batch (operand window) is reported, not fatal
            print(f"{c:44s} B={NB:3d} FAILED: {type(e).__name__}: {str(e)[:200]}", flush=True)
            gc.collect()
            torch.cuda.empty_cache()
            continue
        if base is None:
            base = t
        else:
            print(f"    -> {100 * (t / base - 1):+.2f} % vs {configs[0]}", flush=True)
