#!/usr/bin/env python3
"""Wall time of the bench's UNet step (hipGraph, 3 CFG-branch streams, C2 shape) - for A/B runs of library variants
selected with INSV2V_LIB (outputs of debug variants may be garbage; only the timing is used)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch  # noqa: E402
from insv2v import synth, shapes  # noqa: E402
from insv2v.unet import UNet3DConditionModel  # noqa: E402
from insv2v.inference import GraphedUNet  # noqa: E402

unet = UNet3DConditionModel(**synth.UNET_FULL, device="cuda:0").load_state_dict(synth.synth_state_dict(shapes.unet_shapes(**synth.UNET_FULL)))
for streams in (True, False):
    r = GraphedUNet(unet, 3, 16, 32, 48, 77, use_graph=True, branch_streams=streams)
    r.set_context(synth.synth_input("p.ctx", (3, 77, 768)))
    r.x_in.normal_()
    r.t.fill_(500.0)
    for _ in range(3):
        r.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        r.run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{'3 streams' if streams else 'batched  '}: {e0.elapsed_time(e1) / 20:.2f} ms per UNet step", flush=True)
