#!/usr/bin/env python3
"""Pins the ORACLE (oracle/pipelines.py, oracle/unet3d.py: fp32 CPU restatement) against the full-width goldens the unmodified reference
produced (tests/golden/c4_unit_full.npz: the driver-level carry of a 32-frame unit; c2_ddpm4_full.npz: the shipped DDPM sampler).  ~30 min
of host time, so it is a build-container tool like tools/gen_golden.py, not part of the CPU test suite; needs no /root/reference.
usage: python tools/check_oracle_full.py [c4unit] [c2ddpm]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from insv2v import synth  # noqa: E402
import oracle.unet3d as o_unet  # noqa: E402
import oracle.pipelines as o_pipe  # noqa: E402

torch.set_grad_enabled(False)
GOLD = os.path.join(ROOT, "tests", "golden")
parts = sys.argv[1:] or ["c2ddpm", "c4unit"]
unet = o_unet.UNet3DConditionModel(**synth.UNET_FULL)
unet.load_state_dict({k: synth.synth_tensor(k, v) for k, v in unet.state_dict().items()})
unet.eval()


def cmp(tag, got, want, tol=2e-4):
    want = torch.from_numpy(want)
    err, scale = (got - want).abs().max().item(), want.abs().max().item()
    print(f"{tag}: max|oracle - reference golden| = {err:.3e} (golden max {scale:.3f})", flush=True)
    assert err <= tol * max(1.0, scale), tag


if "c2ddpm" in parts:
    g = np.load(os.path.join(GOLD, "c2_ddpm4_full.npz"))
    p = o_pipe.InferenceIP2PVideo(unet, scheduler="ddpm", num_ddim_steps=4)
    p.variance_noises = [torch.from_numpy(g[f"noise{k}"]) for k in range(3)] + [None]
    t0 = time.time()
    r = p(synth.synth_input("c2.latent", (1, 16, 4, 32, 48)), synth.synth_input("c2.text_cond", (1, 77, 768)),
          synth.synth_input("c2.text_uncond", (1, 77, 768)), synth.synth_input("c2.cond", (1, 16, 4, 32, 48)), text_cfg=7.5, img_cfg=1.5)
    print(f"oracle DDPM 4 steps {time.time() - t0:.0f}s")
    cmp("c2_ddpm4_full.latent", r["latent"], g["latent"])

if "c4unit" in parts:
    g = np.load(os.path.join(GOLD, "c4_unit_full.npz"))
    cond = synth.synth_input("c4.cond", (1, 32, 4, 32, 48))

    class Vae:   # the oracle's edit_video encodes frames itself: hand it the golden's conditioning latent, keep the decode out of it
        def encode(self, x, noise=None):
            return cond[0]

        def decode(self, z):
            return z

    p = o_pipe.InferenceIP2PVideo(unet, scheduler="ddim", num_ddim_steps=4)
    t0 = time.time()
    _, lat = o_pipe.edit_video(p, Vae(), torch.zeros(1, 32, 3, 8, 8), synth.synth_input("c4.text_cond", (1, 77, 768)),
                               synth.synth_input("c4.text_uncond", (1, 77, 768)), 7.5, 1.8, [torch.from_numpy(g[f"noise{k}"]) for k in range(3)])
    print(f"oracle C4 unit {time.time() - t0:.0f}s")
    cmp("c4_unit_full.latent", lat, g["latent"])
print("oracle == reference goldens")
