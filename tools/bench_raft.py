#!/usr/bin/env python3
"""Time of the optical-flow estimator on the HIP kernels as the optical-flow pipe calls it for one window of config C3: R = 4 reference
frames x 12 query frames at 256x384 (inference.py:303-311) - per-query calls (B = 4, the reference's loop) against the batched form
(obtain_flow_batched: 24 pairs per call), with a per-op breakdown of one batched call from the launch recorder."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import synth, shapes, ops  # noqa: E402
from insv2v.raft import RAFTFlow  # noqa: E402

dev = "cuda:0"
H, W, R, Q = int(os.environ.get("H", 256)), int(os.environ.get("W", 384)), 4, 12
est = RAFTFlow(dev, synth.synth_raft_state_dict(shapes.raft_shapes()))
refs = synth.synth_input("br.refs", (R, 3, H, W), kind="uniform").to(dev)
qs = synth.synth_input("br.q", (Q, 3, H, W), kind="uniform").to(dev)


def per_query():
    return [est(q.unsqueeze(0).repeat(R, 1, 1, 1), refs) for q in qs]


def batched():
    per = est.max_pairs // R
    out = []
    for i in range(0, Q, per):
        b = qs[i:i + per]
        f = est(b.repeat_interleave(R, dim=0), refs.repeat(len(b), 1, 1, 1))
        out += list(f.reshape(len(b), R, *f.shape[1:]))
    return out


for name, fn in (("per query (12 calls, B = 4)", per_query), ("batched (obtain_flow_batched)", batched)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        o = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / 3 * 1e3:8.1f} ms per window ({R} refs x {Q} queries at {H}x{W})")
a, b = per_query(), batched()
print("batched == per-query:", all(torch.equal(x, y) for x, y in zip(a, b)), " max |diff|", max((x - y).abs().max().item() for x, y in zip(a, b)))
rec = []
ops.set_launch_recorder(rec)
batched()
torch.cuda.synchronize()
ops.set_launch_recorder(None)
groups = {}
for r in rec:
    tag = r[4][0] if len(r) > 4 else r[0]
    g = groups.setdefault(tag, [0, 0.0])
    g[0] += 1
    g[1] += r[2].elapsed_time(r[3])
tot = sum(g[1] for g in groups.values())
print(f"recorded launches of one window (GEMMs, im2col, instance norm, look-ups): {tot:.1f} ms")
for tag, (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"  {ms:8.2f} ms  n={n:4d}  {tag}")
