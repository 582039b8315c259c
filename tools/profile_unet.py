#!/usr/bin/env python3
"""Per-launch breakdown of one eager full-size UNet forward at the bench shape (HIP events around
every launch on the launch stream).  Writes a table grouped by kernel family + shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import synth, shapes, ops  # noqa: E402
from insv2v.unet import UNet3DConditionModel  # noqa: E402
from insv2v.inference import GraphedUNet  # noqa: E402

NB = int(os.environ.get("NB", 3))
F, h, w = int(os.environ.get("F", 16)), int(os.environ.get("LH", 32)), int(os.environ.get("LW", 48))
unet = UNet3DConditionModel(**synth.UNET_FULL, device="cuda:0").load_state_dict(synth.synth_state_dict(shapes.unet_shapes(**synth.UNET_FULL)))
# CFG_CLIPS = n: the product's stacked layout (3 n branch-major samples, common prefix of branches 1 and 2 computed once); default: as the
# product stacks NB / 3 clips (0 = every sample on its own)
CFG_CLIPS = int(os.environ.get("CFG_CLIPS", NB // 3 if (NB % 3 == 0 and NB > 3) else 0))
r = GraphedUNet(unet, NB, F, h, w, 77, use_graph=False, cfg_clips=CFG_CLIPS)
r.set_context(synth.synth_input("p.ctx", (NB, 77, 768)))
r.x_in.normal_()
r.t.fill_(500.0)
# Three recorded forwards after one untimed one; per shape the FASTEST forward's time is reported.  An eager launch's event pair also
# covers host work that happens while the queue is empty (a first-touch hipMalloc of the caching allocator: 5 launches of 12 ms in round
# 4's B = 3 table) - the captured graph never sees those, and neither should this table.
r.run()
torch.cuda.synchronize()
runs = []
for it in range(int(os.environ.get("REPS", 3))):
    rec = []
    ops.set_launch_recorder(rec)
    r.run()
    torch.cuda.synchronize()
    ops.set_launch_recorder(None)
    groups = {}
    for name, work, e0, e1, tag in rec:
        g = groups.setdefault(tag, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += e0.elapsed_time(e1)
        g[2] += work
    runs.append((groups, len(rec)))
groups = {tag: min((run[0][tag] for run in runs), key=lambda g: g[1]) for tag in runs[0][0]}
tot = sum(g[1] for g in groups.values())
print(f"B={NB} F={F} {h}x{w}: total {tot:.2f} ms over {runs[0][1]} launches (per shape: fastest of {len(runs)} eager forwards)")
for tag, (n, ms, work) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms:8.3f} ms {100 * ms / tot:5.1f}%  n={n:3d}  {ms / n * 1e3:8.1f} us/launch  {work / ms / 1e9 if ms else 0:8.1f} TF/s  {tag}")
