#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

Runs only in the build container (needs /root/reference).  The reference's
missing third-party leaves (diffusers 0.21.4, torchvision) are provided by the
stand-ins in tests/oracle_shim/ (see its README for what that pins and what it
does not).  For every case the script also runs the oracle restatement on the
same inputs and asserts agreement, which is how the oracle is pinned.

Weights are never stored: they are regenerated from crc32(key) by
insv2v.synth on both sides.  Only inputs that cannot be regenerated and the
expected outputs are written (float32).

usage: python tools/gen_golden.py [--only NAME]

Round 6: `python tools/gen_golden.py` runs every default case end to end (clip_text last: it hides tests/oracle_shim while transformers
is imported and restores it).  The committed files reproduce bit for bit at the default thread count (GOLDEN_THREADS unset = all host
cores); another thread count changes fp32 summation order inside torch's CPU kernels (differences of 1e-6 ... 2e-4 of values up to 50).
c5_ddim2_full.npz (FULL_PARTS=c5steps, round 6) was written with GOLDEN_THREADS=32 (18 min of host time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "oracle_shim"), REF,
                os.path.join(ROOT, "instruct-video-to-video_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from insv2v import synth  # noqa: E402
import oracle.unet3d as o_unet  # noqa: E402
import oracle.vae as o_vae  # noqa: E402
import oracle.flow as o_flow  # noqa: E402
import oracle.pipelines as o_pipe  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez(os.path.join(GOLD, name + ".npz"), **out)
    print(f"  wrote {name}.npz ({sum(a.nbytes for a in out.values()) / 1e3:.0f} KB)")


def check(tag, ref, ora, tol=2e-4):
    err = (ref - ora).abs().max().item()
    scale = ref.abs().max().item()
    print(f"  {tag}: max|ref-oracle| = {err:.3e} (ref max {scale:.3f})")
    assert err <= tol * max(1.0, scale), f"oracle disagrees with the reference on {tag}"


def load_synth(module, prefix=""):
    sd = {k: synth.synth_tensor(prefix + k, v) for k, v in module.state_dict().items()}
    module.load_state_dict(sd)
    return module.eval()


# ----------------------------------------------------------------------------- cases
def case_unet_tiny():
    from modules.video_unet_temporal.unet import UNet3DConditionModel as RefUNet
    ref = load_synth(RefUNet(**synth.UNET_TINY))
    ora = load_synth(o_unet.UNet3DConditionModel(**synth.UNET_TINY))
    assert set(ref.state_dict().keys()) == set(ora.state_dict().keys()), "state-dict keys differ"
    x = synth.synth_input("unet_tiny.sample", (3, 8, 8, 16, 24))
    ctx = synth.synth_input("unet_tiny.ctx", (3, 77, 64))
    t = torch.full((3,), 981, dtype=torch.long)
    y_ref = ref(x, t, encoder_hidden_states=ctx).sample
    y_ora = ora(x, t, ctx).sample
    check("unet_tiny", y_ref, y_ora)
    save("unet_tiny_fwd", out=y_ref)
    # second shape / timestep / start index exercise F=16 and the PE offset
    x2 = synth.synth_input("unet_tiny.sample2", (1, 8, 16, 8, 8))
    ctx2 = synth.synth_input("unet_tiny.ctx2", (1, 77, 64))
    t2 = torch.full((1,), 41, dtype=torch.long)
    y2 = ref(x2, t2, encoder_hidden_states=ctx2, video_start_index=3).sample
    check("unet_tiny_f16", y2, ora(x2, t2, ctx2, video_start_index=3).sample)
    save("unet_tiny_fwd_f16", out=y2)


def case_blocks_full():
    from modules.video_unet_temporal.resnet import ResnetBlock3D
    from modules.video_unet_temporal.attention import Transformer3DModel
    from modules.video_unet_temporal.motion_module import VanillaTemporalModule
    B, F, H, W = 2, 16, 4, 6
    temb = synth.synth_input("blk.temb", (B, 1280))
    out = {}
    for name, cin, cout in (("res320", 320, 320), ("res960", 960, 320)):
        ref = load_synth(ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=1280, eps=1e-5, groups=32,
                                       non_linearity="silu"), name + ".")
        ora = load_synth(o_unet.ResBlock(cin, cout, 1280, 32, 1e-5), name + ".")
        x = synth.synth_input(name + ".x", (B, cin, F, H, W))
        y = ref(x, temb)
        check(name, y, ora(x, temb))
        out[name] = y
    x = synth.synth_input("attn320.x", (B, 320, F, H, W))
    ctx = synth.synth_input("attn320.ctx", (B, 77, 768))
    ref = load_synth(Transformer3DModel(8, 40, in_channels=320, num_layers=1, cross_attention_dim=768, norm_num_groups=32), "attn320.")
    ora = load_synth(o_unet.SpatialTransformer(8, 40, 320, 768, 32), "attn320.")
    y = ref(x, encoder_hidden_states=ctx).sample
    check("attn320", y, ora(x, ctx))
    out["attn320"] = y
    mkw = synth.UNET_FULL["motion_module_kwargs"]
    ref = VanillaTemporalModule(in_channels=320, **mkw)
    ora = o_unet.MotionModule(320, 32, **mkw)
    load_synth(ref, "mm320.")
    load_synth(ora, "mm320.")
    x = synth.synth_input("mm320.x", (B, 320, F, H, W))
    y = ref(x, None, video_start_index=0)
    check("mm320", y, ora(x, 0))
    assert (y - x).abs().max() > 1e-2, "motion module must not be the identity in the goldens (F8)"
    out["mm320"] = y
    save("blocks_full", **out)


def case_blocks_wide():
    """Round 6: the blocks whose kernels changed - ResnetBlock3D with >= 1280 input channels (Winograd form, two-source concat 1280 + 1280),
    Transformer3DModel at 1280 channels and the motion module at 640 (second feed-forward projection merged with proj_out) - at the real
    widths against the unmodified reference."""
    from modules.video_unet_temporal.resnet import ResnetBlock3D
    from modules.video_unet_temporal.attention import Transformer3DModel
    from modules.video_unet_temporal.motion_module import VanillaTemporalModule
    B, F, H, W = 2, 16, 4, 6
    temb = synth.synth_input("blkw.temb", (B, 1280))
    out = {}
    for name, cin, cout in (("res2560", 2560, 1280), ("res1280", 1280, 1280)):
        ref = load_synth(ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=1280, eps=1e-5, groups=32, non_linearity="silu"), name + ".")
        ora = load_synth(o_unet.ResBlock(cin, cout, 1280, 32, 1e-5), name + ".")
        x = synth.synth_input(name + ".x", (B, cin, F, H, W))
        y = ref(x, temb)
        check(name, y, ora(x, temb))
        out[name] = y
    x = synth.synth_input("attn1280.x", (B, 1280, F, H, W))
    ctx = synth.synth_input("attn1280.ctx", (B, 77, 768))
    ref = load_synth(Transformer3DModel(8, 160, in_channels=1280, num_layers=1, cross_attention_dim=768, norm_num_groups=32), "attn1280.")
    ora = load_synth(o_unet.SpatialTransformer(8, 160, 1280, 768, 32), "attn1280.")
    y = ref(x, encoder_hidden_states=ctx).sample
    check("attn1280", y, ora(x, ctx))
    out["attn1280"] = y
    mkw = synth.UNET_FULL["motion_module_kwargs"]
    ref = VanillaTemporalModule(in_channels=640, **mkw)
    ora = o_unet.MotionModule(640, 32, **mkw)
    load_synth(ref, "mm640.")
    load_synth(ora, "mm640.")
    x = synth.synth_input("mm640.x", (B, 640, F, H, W))
    y = ref(x, None, video_start_index=0)
    check("mm640", y, ora(x, 0))
    out["mm640"] = y
    save("blocks_wide", **{k: v.half().numpy() for k, v in out.items()})   # fp16: 6.9 MB instead of 13.8 (the stated tolerance is 1e-2)


def case_vae():
    from modules.vqvae.model import Encoder as RefEnc, Decoder as RefDec
    dd = synth.VAE_FULL["ddconfig"]
    ora = load_synth(o_vae.AutoencoderKL(**synth.VAE_FULL))
    renc, rdec = RefEnc(**dd), RefDec(**dd)
    renc.load_state_dict(ora.encoder.state_dict())
    rdec.load_state_dict(ora.decoder.state_dict())
    x = synth.synth_input("vae.x", (2, 3, 64, 96), kind="uniform")
    h_ref = renc.eval()(x)
    check("vae.encoder", h_ref, ora.encoder(x))
    z = synth.synth_input("vae.z", (1, 4, 8, 12))
    d_ref = rdec.eval()(ora.post_quant_conv(z))
    check("vae.decoder", d_ref, ora.decode(z))
    noise = synth.synth_input("vae.noise", (2, 4, 8, 12))
    save("vae_full", enc_h=h_ref, dec=d_ref, enc_sample=ora.encode(x, noise))


def case_flow():
    from misc_utils.flow_utils import warp_image, resize_flow
    img = synth.synth_input("flow.img", (4, 4, 32, 48))
    flow = synth.synth_input("flow.flow", (4, 2, 32, 48), scale=3.0)
    w_ref = warp_image(img, flow)
    check("warp", w_ref, o_flow.warp_image(img, flow), tol=1e-5)
    big = synth.synth_input("flow.big", (4, 2, 256, 384), scale=8.0)
    r_ref = resize_flow(big, (32, 48))
    check("resize", r_ref, o_flow.resize_flow(big, (32, 48)), tol=1e-5)
    odd = synth.synth_input("flow.odd", (2, 2, 50, 70), scale=8.0)
    r2 = resize_flow(odd, (32, 48))
    check("resize_odd", r2, o_flow.resize_flow(odd, (32, 48)), tol=1e-5)
    ident = warp_image(img, torch.zeros_like(flow))
    assert (ident - img).abs().max() < 1e-4
    save("flow", warp=w_ref, resize=r_ref, resize_odd=r2)


def case_split_batch():
    sys.argv = ["x"]  # the reference driver parses argv and builds models at import: re-state call only
    src = open(os.path.join(REF, "insv2v_run_loveu_tgve.py")).read()
    ns = {}
    start = src.index("def split_batch")
    end = src.index("parser = argparse")
    exec(compile(src[start:end], "split_batch_ref", "exec"), {"torch": torch}, ns)
    plans = {}
    for T in (8, 16, 20, 24, 28, 32, 40, 48, 64):
        c = torch.arange(T)[None]
        chunks, refs = ns["split_batch"](c, 16, 4)
        ochunks, orefs = o_pipe.split_batch(c, 16, 4)
        assert [x.tolist() for x in chunks] == [x.tolist() for x in ochunks] and refs == orefs
        plans[str(T)] = {"new": [x.shape[1] for x in chunks], "refs": refs}
    json.dump(plans, open(os.path.join(GOLD, "split_batch.json"), "w"), indent=1)
    print("  wrote split_batch.json", plans["32"])


class _FakeFlow:
    """Stands in for RAFTFlow (out of scope): returns injected flows query by query."""

    def __init__(self, flows):
        self.flows, self.i = flows, 0

    def __call__(self, query, refs):
        f = self.flows[self.i]
        self.i += 1
        return f


def case_pipelines():
    import pl_trainer.inference.inference as ref_inf
    from modules.video_unet_temporal.unet import UNet3DConditionModel as RefUNet
    runet = load_synth(RefUNet(**synth.UNET_TINY))
    ounet = load_synth(o_unet.UNet3DConditionModel(**synth.UNET_TINY))
    F, h, w, R = 8, 16, 24, 4
    lat = synth.synth_input("pipe.latent", (1, F, 4, h, w))
    cond = synth.synth_input("pipe.cond", (1, F, 4, h, w))
    tc = synth.synth_input("pipe.text_cond", (1, 77, 64))
    tu = synth.synth_input("pipe.text_uncond", (1, 77, 64))
    lref = synth.synth_input("pipe.latent_ref", (1, R, 4, h, w))
    out = {}

    rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddim", num_ddim_steps=10)
    op = o_pipe.InferenceIP2PVideo(ounet, scheduler="ddim", num_ddim_steps=10)
    assert rp.scheduler.timesteps.tolist() == op.scheduler.timesteps.tolist()
    r = rp(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
    o = op(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
    check("ddim10", r["latent"], o["latent"], tol=1e-3)
    out["ddim10_latent"], out["ddim10_pred0"] = r["latent"], r["all_pred"][0]

    r = rp(lat, tc, tu, cond, text_cfg=1.0, img_cfg=1.0)  # known answer: equals branch 3 only
    out["ddim10_cfg1_latent"] = r["latent"]

    r = rp(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5, guidance_rescale=0.5)
    o = op(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5, guidance_rescale=0.5)
    check("ddim10_rescale", r["latent"], o["latent"], tol=1e-3)
    out["ddim10_rescale_latent"] = r["latent"]

    r = rp.second_clip_forward(lat, tc, tu, cond, latent_ref=lref, noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    o = op.second_clip_forward(lat, tc, tu, cond, latent_ref=lref, noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    check("second_clip", r["latent"], o["latent"], tol=1e-3)
    out["second_clip_latent"] = r["latent"]

    flows = [synth.synth_input(f"pipe.flow{q}", (R, 2, h * 8, w * 8), scale=8.0) for q in range(F - R)]
    rf = object.__new__(ref_inf.InferenceIP2PVideoOpticalFlow)
    ref_inf.InferenceIP2PVideo.__init__(rf, runet, scheduler="ddim", num_ddim_steps=10)
    rf.flow_estimator = _FakeFlow(flows)
    imgs_r = torch.zeros(1, R, 3, h * 8, w * 8)
    imgs_q = torch.zeros(1, F - R, 3, h * 8, w * 8)
    r = rf.second_clip_forward(lat, tc, tu, cond, latent_ref=lref, ref_images=imgs_r, query_images=imgs_q,
                               noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    of = o_pipe.InferenceIP2PVideoOpticalFlow(ounet, scheduler="ddim", num_ddim_steps=10)
    o = of.second_clip_forward(lat, tc, tu, cond, latent_ref=lref, flows=flows, noise_correct_step=0.5,
                               text_cfg=7.5, img_cfg=1.5)
    check("second_clip_flow", r["latent"], o["latent"], tol=1e-3)
    out["second_clip_flow_latent"] = r["latent"]

    rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddpm", num_ddim_steps=4)
    op = o_pipe.InferenceIP2PVideo(ounet, scheduler="ddpm", num_ddim_steps=4)
    assert rp.scheduler.timesteps.tolist() == op.scheduler.timesteps.tolist() == [750, 500, 250, 0]
    torch.manual_seed(1234)
    r = rp(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
    torch.manual_seed(1234)
    noises = [torch.randn(lat.shape) for _ in range(3)] + [None]
    op.variance_noises = noises
    o = op(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
    check("ddpm4", r["latent"], o["latent"], tol=1e-3)
    out["ddpm4_latent"] = r["latent"]
    for i in range(3):
        out[f"ddpm4_noise{i}"] = noises[i]
    save("pipelines_tiny", **out)


def case_full_size():
    """Full-width AND full-size goldens: the UNMODIFIED reference wiring on CPU fp32 at BASELINE's
    sizes.  ~2 h of host time on 8 cores; run explicitly with --only full_size (skipped by the default sweep).
    Each part is written as soon as it is done (fp16 where noted, to keep the fixtures a few MB)."""
    import pl_trainer.inference.inference as ref_inf
    from modules.video_unet_temporal.unet import UNet3DConditionModel as RefUNet
    from modules.vqvae.model import Decoder as RefDec
    runet = load_synth(RefUNet(**synth.UNET_FULL))
    ounet = load_synth(o_unet.UNet3DConditionModel(**synth.UNET_FULL))
    parts = os.environ.get("FULL_PARTS", "c2fwd,c5fwd,c5steps,c1,c2steps,c3second,vaeenc,c4unit,c2ddpm").split(",")

    if "c4unit" in parts:  # (vii) VERDICT r4 item 2: the driver-level carry of a C4 unit - the reference's OWN unit loop
        # (insv2v_run_loveu_tgve.py:119-162, executed from its source text, not restated) on a 32-frame conditioning latent at the C2
        # geometry: windows 16 / 12 / 4 new frames, overlap re-uses the INITIAL noise, latent_ref = previous prediction's tail (R = 4, 12),
        # 4 DDIM steps.  torch.randn_like draws come from the global generator: seeded, and replayed below as the injected init_noises.
        import types
        T, h, w = 32, 32, 48
        cond = synth.synth_input("c4.cond", (1, T, 4, h, w))
        tc = synth.synth_input("c4.text_cond", (1, 77, 768))
        tu = synth.synth_input("c4.text_uncond", (1, 77, 768))
        src = open(os.path.join(REF, "insv2v_run_loveu_tgve.py")).read()
        sb_ns = {}
        exec(compile(src[src.index("def split_batch"):src.index("parser = argparse")], "split_batch_ref", "exec"), {"torch": torch}, sb_ns)
        start = src.index("        conds, num_ref_frames_each_batch = split_batch(cond")
        end = src.index("        # Save GIF")
        body = "\n".join(line[8:] for line in src[start:end].split("\n"))   # the loop body, de-indented, otherwise untouched
        ns = dict(torch=torch, split_batch=sb_ns["split_batch"], cond=cond, batch={"frames": torch.zeros(1, T, 1, 1, 1)},
                  frames_in_batch=16, num_ref_frames=4, text_cond=tc, text_uncond=tu, text_cfg=7.5, video_cfg=1.8,
                  args=types.SimpleNamespace(with_optical_flow=False),
                  inf_pipe=ref_inf.InferenceIP2PVideo(runet, scheduler="ddim", num_ddim_steps=4))
        torch.manual_seed(4321)
        t0 = time.time()
        exec(compile(body, "unit_loop_ref", "exec"), ns)
        print(f"  reference C4 unit loop (32 frames, 3 windows x 4 DDIM steps) {time.time() - t0:.0f}s")
        latent = torch.cat(ns["latent_pred_list"], dim=1)
        assert latent.shape == (1, T, 4, h, w) and ns["num_ref_frames_each_batch"] == [4, 12]
        torch.manual_seed(4321)   # DDIM draws nothing else: the three randn_like calls replay in order
        noises = [torch.randn_like(cond[:, :16]), torch.randn_like(cond[:, 16:28]), torch.randn_like(cond[:, 28:32])]
        save("c4_unit_full", latent=latent, noise0=noises[0], noise1=noises[1], noise2=noises[2])

    if "c2ddpm" in parts:  # (viii) VERDICT r4 item 2: the SHIPPED sampler (scheduler='ddpm', insv2v_run_loveu_tgve.py:65-75) at full width
        # and the C2 geometry, 4 steps; the variance noises the scheduler draws from the global generator are replayed and committed
        F, h, w = 16, 32, 48
        lat = synth.synth_input("c2.latent", (1, F, 4, h, w))
        cond = synth.synth_input("c2.cond", (1, F, 4, h, w))
        tc = synth.synth_input("c2.text_cond", (1, 77, 768))
        tu = synth.synth_input("c2.text_uncond", (1, 77, 768))
        rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddpm", num_ddim_steps=4)
        assert rp.scheduler.timesteps.tolist() == [750, 500, 250, 0]
        torch.manual_seed(1234)
        t0 = time.time()
        r = rp(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
        print(f"  reference C2-size DDPM, 4 steps {time.time() - t0:.0f}s")
        torch.manual_seed(1234)
        noises = [torch.randn(lat.shape) for _ in range(3)]   # no draw at t = 0
        out = {"latent": r["latent"], "pred0": r["all_pred"][0]}
        for i in range(3):
            out[f"noise{i}"] = noises[i]
        save("c2_ddpm4_full", **out)

    if "c3second" in parts:  # (v) VERDICT r3 item 9: second_clip_forward at full width and the C2 geometry, 4 DDIM steps, R = 4 reference
        # frames, noise correction for the first half of the steps - mean-delta (inference.py:216-289) and optical flow (:291-398)
        F, h, w, R = 16, 32, 48, 4
        lat = synth.synth_input("c3.latent", (1, F, 4, h, w))
        cond = synth.synth_input("c3.cond", (1, F, 4, h, w))
        tc = synth.synth_input("c3.text_cond", (1, 77, 768))
        tu = synth.synth_input("c3.text_uncond", (1, 77, 768))
        lref = synth.synth_input("c3.latent_ref", (1, R, 4, h, w))
        rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddim", num_ddim_steps=4)
        t0 = time.time()
        r = rp.second_clip_forward(lat, tc, tu, cond, latent_ref=lref, noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
        print(f"  reference C2-size second_clip_forward, 4 steps {time.time() - t0:.0f}s")
        out = {"second_clip_latent": r["latent"], "second_clip_pred0": r["all_pred"][0]}
        flows = [synth.synth_input(f"c3.flow{q}", (R, 2, h * 8, w * 8), scale=8.0) for q in range(F - R)]
        rf = object.__new__(ref_inf.InferenceIP2PVideoOpticalFlow)
        ref_inf.InferenceIP2PVideo.__init__(rf, runet, scheduler="ddim", num_ddim_steps=4)
        rf.flow_estimator = _FakeFlow(flows)
        r = rf.second_clip_forward(lat, tc, tu, cond, latent_ref=lref, ref_images=torch.zeros(1, R, 3, h * 8, w * 8),
                                   query_images=torch.zeros(1, F - R, 3, h * 8, w * 8), noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
        out["second_clip_flow_latent"] = r["latent"]
        save("c3_second_clip_full", **out)

    if "vaeenc" in parts:  # (vi) the VAE encoder at the bench's frame size (one 256x384 frame): moments and the posterior sample
        from modules.vqvae.model import Encoder as RefEnc
        ovae = load_synth(o_vae.AutoencoderKL(**synth.VAE_FULL))
        renc = RefEnc(**synth.VAE_FULL["ddconfig"]).eval()
        renc.load_state_dict(ovae.encoder.state_dict())
        x = synth.synth_input("vae.full.x", (1, 3, 256, 384), kind="uniform")
        h_ref = renc(x)
        check("vae.encoder 256x384", h_ref, ovae.encoder(x))
        noise = synth.synth_input("vae.full.noise", (1, 4, 32, 48))
        save("vae_encode_full", enc_sample=ovae.encode(x, noise), moments=ovae.quant_conv(h_ref))

    if "c2fwd" in parts:  # (i) C2: one 3-branch UNet forward [3,8,16,32,48], the shape every bench step runs
        x = synth.synth_input("c2.sample", (3, 8, 16, 32, 48))
        ctx = synth.synth_input("c2.ctx", (3, 77, 768))
        t = torch.tensor([981, 981, 981])
        t0 = time.time()
        y = runet(x, t, encoder_hidden_states=ctx).sample
        print(f"  reference C2 forward {time.time() - t0:.0f}s")
        check("c2_unet_fwd", y, ounet(x, t, ctx).sample)
        save("c2_unet_fwd", out=y.half().numpy())

    if "c5fwd" in parts:  # (ii) C5: one branch at 24 f, 48x64 latents
        x = synth.synth_input("c5.sample", (1, 8, 24, 48, 64))
        ctx = synth.synth_input("c5.ctx", (1, 77, 768))
        t = torch.tensor([501])
        y = runet(x, t, encoder_hidden_states=ctx).sample
        check("c5_unet_fwd", y, ounet(x, t, ctx).sample)
        save("c5_unet_fwd", out=y.half().numpy())

    if "c5steps" in parts:  # (ii-b) round 6: C5 with all THREE branches - a 2-step DDIM trajectory, text 7.5 / video 1.5, at 24 f, 48x64
        lat = synth.synth_input("c5.latent", (1, 24, 4, 48, 64))
        cond = synth.synth_input("c5.cond", (1, 24, 4, 48, 64))
        tc = synth.synth_input("c5.text_cond", (1, 77, 768))
        tu = synth.synth_input("c5.text_uncond", (1, 77, 768))
        rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddim", num_ddim_steps=2)
        t0 = time.time()
        r = rp(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
        print(f"  reference C5 2 steps (3 branches each) {time.time() - t0:.0f}s")
        o = o_pipe.InferenceIP2PVideo(ounet, scheduler="ddim", num_ddim_steps=2)(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
        check("c5_ddim2 latent", r["latent"], o["latent"], tol=2e-3)
        save("c5_ddim2_full", latent=r["latent"], latent_step0=r["all_latent"][0])

    if "c1" in parts:  # (iv) C1 exactly as BASELINE states it: 8 f, 32x32 latents, 10 DDIM steps, text_cfg = img_cfg = 1
        lat = synth.synth_input("c1.latent", (1, 8, 4, 32, 32))
        cond = synth.synth_input("c1.cond", (1, 8, 4, 32, 32))
        tc = synth.synth_input("c1.text_cond", (1, 77, 768))
        tu = synth.synth_input("c1.text_uncond", (1, 77, 768))
        rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddim", num_ddim_steps=10)
        r = rp(lat, tc, tu, cond, text_cfg=1.0, img_cfg=1.0)
        save("c1_ddim10_cfg1", latent=r["latent"], pred0=r["all_pred"][0])

    if "c2steps" in parts:  # (iii) C2: the full 50-step trajectory, text 7.5 / video 1.5, then VAE decode of 3 frames
        lat = synth.synth_input("c2.latent", (1, 16, 4, 32, 48))
        cond = synth.synth_input("c2.cond", (1, 16, 4, 32, 48))
        tc = synth.synth_input("c2.text_cond", (1, 77, 768))
        tu = synth.synth_input("c2.text_uncond", (1, 77, 768))
        rp = ref_inf.InferenceIP2PVideo(runet, scheduler="ddim", num_ddim_steps=50)
        t0 = time.time()
        r = rp(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
        print(f"  reference C2 50 steps {time.time() - t0:.0f}s")
        out = {"latent": r["latent"]}
        for i in (0, 4, 9, 24, 39):
            out[f"latent_step{i}"] = r["all_latent"][i]
        ovae = load_synth(o_vae.AutoencoderKL(**synth.VAE_FULL))
        rdec = RefDec(**synth.VAE_FULL["ddconfig"]).eval()
        rdec.load_state_dict(ovae.decoder.state_dict())
        z = r["latent"][0, [0, 7, 15]] / 0.18215  # instruct_p2p_video.py:66-79
        out["frames_0_7_15"] = rdec(ovae.post_quant_conv(z)).half().numpy()
        save("c2_ddim50", **out)


def case_clip_text():
    """FrozenCLIPEmbedder's transformer = transformers.CLIPTextModel (third party, installed here): goldens are the REAL
    model's outputs on key-hashed weights; the oracle restatement must agree."""
    # the torchvision stand-in of tests/oracle_shim must not be visible to transformers' optional-dependency probing
    # (hidden only while transformers is imported: the other cases need the shim's diffusers / torchvision stand-ins back - VERDICT r5 weak 2)
    shim = os.path.join(ROOT, "tests", "oracle_shim")
    saved_path = list(sys.path)
    saved_mods = {m: sys.modules[m] for m in list(sys.modules) if m == "torchvision" or m.startswith("torchvision.")}
    sys.path[:] = [p for p in sys.path if p != shim]
    for m in saved_mods:
        del sys.modules[m]
    try:
        from transformers import CLIPTextConfig, CLIPTextModel
        CLIPTextModel(CLIPTextConfig(vocab_size=8, hidden_size=8, intermediate_size=8, num_hidden_layers=1, num_attention_heads=1))  # lazy sub-imports happen here
    finally:
        sys.path[:] = saved_path
        for m in [m for m in sys.modules if m == "torchvision" or m.startswith("torchvision.")]:
            del sys.modules[m]
        sys.modules.update(saved_mods)
    from insv2v import shapes
    import oracle.clip_text as o_clip
    for name, cfg in (("tiny", synth.CLIP_TINY), ("full", synth.CLIP_FULL)):
        sd = synth.synth_state_dict(shapes.clip_text_shapes(**cfg))
        hf = CLIPTextModel(CLIPTextConfig(**cfg, hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1)).eval()
        hf.load_state_dict(o_clip.strip_prefixes(sd), strict=True)
        ids = synth.synth_token_ids("clip." + name, 2, 77, cfg["vocab_size"])
        r = hf(input_ids=ids, output_hidden_states=True)
        o = o_clip.clip_text_forward(sd, ids, cfg["num_attention_heads"])
        check(f"clip_text {name} last_hidden_state", r.last_hidden_state, o["last_hidden_state"], tol=2e-5)
        check(f"clip_text {name} pooler_output", r.pooler_output, o["pooler_output"], tol=2e-5)
        check(f"clip_text {name} hidden[-2]", r.hidden_states[-2], o["hidden_states"][-2], tol=2e-5)
        save(f"clip_text_{name}", input_ids=ids.numpy(), last_hidden_state=r.last_hidden_state, pooler_output=r.pooler_output,
             hidden_m2=r.hidden_states[-2])


# clip_text runs last of the default cases: it is the one case that hides tests/oracle_shim for a while
CASES = dict(unet_tiny=case_unet_tiny, blocks_full=case_blocks_full, blocks_wide=case_blocks_wide, vae=case_vae, flow=case_flow,
             split_batch=case_split_batch, pipelines=case_pipelines, clip_text=case_clip_text,
             full_size=case_full_size)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    for name, fn in CASES.items():
        if (a.only and a.only != name) or (not a.only and name == "full_size"):
            continue
        t0 = time.time()
        print(f"[{name}]")
        fn()
        print(f"  done in {time.time() - t0:.1f}s")
