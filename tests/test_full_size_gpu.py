"""Value-level parity at FULL width AND FULL size (BASELINE.json configs C1, C2, C5).

The goldens are outputs of the UNMODIFIED reference wiring (tools/gen_golden.py --only full_size: fp32 on the build
container's CPU, ~2 h of host time) on crc32(key)-hashed weights and seeded inputs; nothing here needs the oracle or
/root/reference at run time.  These tests pin, in one go, what the kernel-level tests pin separately: the persistent
256x256 / 128x256 GEMMs, split-K, the patch-tiled convolutions, the 8-wave tiles at M = 73 728, the hipGraph with three
CFG-branch streams - at the sizes bench.py runs.

Stated fp16 tolerance (fp16 weights / activations, fp32 accumulation and statistics, vs the reference's fp32 path):
  one UNet forward                       rel-RMS <= 1e-2
  10-step DDIM trajectory                rel-RMS <= 3e-2
  50-step DDIM trajectory (C2, CFG 7.5/1.5): rel-RMS <= 1e-2 on the final latent, <= 1.5e-2 on decoded frames
Measured on MI355X (round 2): forward 1.2e-3 (C2, C5), C1 10 steps 5.7e-4, C2 50 steps 1.6e-3 (latent) / 2.0e-3 (frames);
the error does not grow along the trajectory (8.5e-4 after step 1, 1.6e-3 from step 10 on): DESIGN.md section 4.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def report(out, ref, what, rms_tol, max_tol):
    out, ref = out.detach().float().cpu(), torch.as_tensor(np.asarray(ref)).float().cpu()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    rms = ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    mx = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"[parity] {what}: rel-rms {rms:.3e}  max-abs/max-ref {mx:.3e}")
    assert math.isfinite(rms) and rms <= rms_tol and mx <= max_tol, f"{what}: rel-rms {rms:.3e} (tol {rms_tol}), max {mx:.3e} (tol {max_tol})"
    return rms


def _gold(name):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (tools/gen_golden.py --only full_size)")
    return np.load(path)


@pytest.fixture(scope="module")
def full_unet():
    from insv2v import synth, shapes
    from insv2v.unet import UNet3DConditionModel
    sd = synth.synth_state_dict(shapes.unet_shapes(**synth.UNET_FULL))
    return UNet3DConditionModel(**synth.UNET_FULL, device=DEV).load_state_dict(sd)


def test_c2_unet_forward_vs_reference_golden(full_unet):
    """One 3-branch UNet forward at the headline size [3, 8, 16, 32, 48] (reference: unet.py:296-434), through the eager
    batched path AND through the bench's hipGraph with one stream per CFG branch."""
    from insv2v import synth, ops
    from insv2v.inference import GraphedUNet
    g = _gold("c2_unet_fwd")["out"]
    x = synth.synth_input("c2.sample", (3, 8, 16, 32, 48))
    ctx = synth.synth_input("c2.ctx", (3, 77, 768))
    t = torch.tensor([981, 981, 981])
    out = full_unet(x, t, encoder_hidden_states=ctx).sample
    report(out, g, "C2 full-size UNet forward, batched eager (reference golden)", 1e-2, 4e-2)
    B, F, H, W = 3, 16, 32, 48
    r = GraphedUNet(full_unet, B, F, H, W, 77, use_graph=True, branch_streams=True)
    r.set_context(ctx)
    xin = ops.nchw_to_nhwc_f16(x.to(DEV).permute(0, 2, 1, 3, 4).reshape(B * F, 8, H, W).contiguous(), r.x_in.shape[-1])
    r.x_in.copy_(xin)
    r.t.fill_(981.0)
    eps = r.run()
    out2 = ops.nhwc_to_nchw_f32(eps, B * F, 4, H, W).reshape(B, F, 4, H, W).permute(0, 2, 1, 3, 4)
    report(out2, g, "C2 full-size UNet forward, hipGraph + 3 branch streams (reference golden)", 1e-2, 4e-2)


def test_c5_unet_forward_vs_reference_golden(full_unet):
    """BASELINE config C5's geometry at full width: one branch, 24 frames, 48x64 latents (M = 73 728 tokens at level 0)."""
    from insv2v import synth
    g = _gold("c5_unet_fwd")["out"]
    x = synth.synth_input("c5.sample", (1, 8, 24, 48, 64))
    ctx = synth.synth_input("c5.ctx", (1, 77, 768))
    out = full_unet(x, torch.tensor([501]), encoder_hidden_states=ctx).sample
    report(out, g, "C5 full-size UNet forward (reference golden)", 1e-2, 4e-2)


def test_c5_three_branch_trajectory_vs_reference_golden(full_unet):
    """BASELINE config C5 with all THREE CFG branches, by value: a 2-step DDIM trajectory at text 7.5 / video 1.5 on the
    24-frame 48x64 latents - golden of the unmodified reference (tools/gen_golden.py FULL_PARTS=c5steps).  The 24-frame temporal path
    (per-frame-table GEMM, unfused temporal attention, banded Winograd staging at the 24x32 level) through the single-clip pipe and, by
    value again, inside a stack of two clips (B = 6)."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideo
    g = _gold("c5_ddim2_full")
    kw = dict(latent=synth.synth_input("c5.latent", (1, 24, 4, 48, 64)), img_cond=synth.synth_input("c5.cond", (1, 24, 4, 48, 64)),
              text_cond=synth.synth_input("c5.text_cond", (1, 77, 768)), text_uncond=synth.synth_input("c5.text_uncond", (1, 77, 768)),
              text_cfg=7.5, img_cfg=1.5)
    pipe = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=2)
    r = pipe(**kw)
    report(r["all_latent"][0], g["latent_step0"], "C5 3-branch DDIM step 0 (reference golden)", 1e-2, 5e-2)
    report(r["latent"], g["latent"], "C5 3-branch 2-step trajectory (reference golden)", 1e-2, 5e-2)
    rs = pipe.run_stacked([kw, kw])
    assert torch.equal(rs[0]["latent"], rs[1]["latent"])
    report(rs[0]["latent"], g["latent"], "C5 3-branch 2-step trajectory inside a stack of two (reference golden)", 1e-2, 5e-2)


def test_c1_exact_baseline_config_vs_reference_golden(full_unet):
    """BASELINE config C1 exactly as stated: 8 frames, 256x256 (32x32 latents), 10 DDIM steps, text_cfg = img_cfg = 1
    (the reference still runs all three branches, inference.py:183-203; with both scales 1 the result is branch 3)."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideo
    g = _gold("c1_ddim10_cfg1")
    lat = synth.synth_input("c1.latent", (1, 8, 4, 32, 32))
    cond = synth.synth_input("c1.cond", (1, 8, 4, 32, 32))
    tc = synth.synth_input("c1.text_cond", (1, 77, 768))
    tu = synth.synth_input("c1.text_uncond", (1, 77, 768))
    out = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=10)(lat, tc, tu, cond, text_cfg=1.0, img_cfg=1.0)
    report(out["all_pred"][0], g["pred0"], "C1: first x0 prediction (reference golden)", 1e-2, 4e-2)
    report(out["latent"], g["latent"], "C1: 10-step DDIM latent, CFG off (reference golden)", 3e-2, 1e-1)


def test_c2_50_step_trajectory_vs_reference_golden(full_unet):
    """The headline workload end to end: 50 DDIM steps at text 7.5 / video 1.5 on [1, 16, 4, 32, 48] latents, then the VAE
    decode of frames 0, 7, 15.  The measured rel-RMS of the final latent is the stated 50-step bound of DESIGN.md."""
    from insv2v import synth, shapes
    from insv2v.inference import InferenceIP2PVideo
    from insv2v.vae import AutoencoderKL
    g = _gold("c2_ddim50")
    lat = synth.synth_input("c2.latent", (1, 16, 4, 32, 48))
    cond = synth.synth_input("c2.cond", (1, 16, 4, 32, 48))
    tc = synth.synth_input("c2.text_cond", (1, 77, 768))
    tu = synth.synth_input("c2.text_uncond", (1, 77, 768))
    out = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=50)(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
    for i in (0, 4, 9, 24, 39):
        report(out["all_latent"][i], g[f"latent_step{i}"], f"C2: latent after step {i + 1} (reference golden)", 1e-2, 5e-2)
    report(out["latent"], g["latent"], "C2: 50-step DDIM latent (reference golden)", 1e-2, 5e-2)
    vae = AutoencoderKL(**synth.VAE_FULL, device=DEV).load_state_dict(synth.synth_state_dict(shapes.vae_shapes(**synth.VAE_FULL)))
    z = out["latent"][0, [0, 7, 15]].to(DEV) / 0.18215
    report(vae.decode(z), g["frames_0_7_15"], "C2: decoded frames 0/7/15 after 50 steps (reference golden)", 1.5e-2, 8e-2)


def test_c5_three_branch_forward_as_benched(full_unet):
    """BASELINE config C5 as bench.py runs it (--frames 24 --height 384 --width 512): THREE branches at [3, 8, 24, 48, 64]
    (221 184 tokens at level 0; 24 frames: the temporal blocks take the per-frame-table GEMM path, the spatial ones the
    register-resident row kernels).  Branch 0 carries the inputs of the one-branch reference golden, so it is pinned by value;
    the captured hipGraph must replay the eager result bit for bit, and three branch streams must agree with the batch."""
    from insv2v import synth, ops
    from insv2v.inference import GraphedUNet
    g = _gold("c5_unet_fwd")["out"]
    B, F, H, W = 3, 24, 48, 64
    x = torch.cat([synth.synth_input("c5.sample", (1, 8, F, H, W)), synth.synth_input("c5.sample.b", (2, 8, F, H, W))], 0)
    ctx = torch.cat([synth.synth_input("c5.ctx", (1, 77, 768)), synth.synth_input("c5.ctx.b", (2, 77, 768))], 0)
    out = full_unet(x, torch.tensor([501, 501, 501]), encoder_hidden_states=ctx).sample
    assert torch.isfinite(out).all()
    report(out[:1], g, "C5 three-branch forward, branch 0 (reference golden)", 1e-2, 4e-2)
    outs = []
    for streams in (False, True):
        r = GraphedUNet(full_unet, B, F, H, W, 77, use_graph=True, branch_streams=streams)
        r.set_context(ctx)
        r.x_in.copy_(ops.nchw_to_nhwc_f16(x.to(DEV).permute(0, 2, 1, 3, 4).reshape(B * F, 8, H, W).contiguous(), r.x_in.shape[-1]))
        r.t.fill_(501.0)
        e1 = r.run().clone()
        assert torch.equal(e1, r.run()), "graph replay is not deterministic"
        outs.append(ops.nhwc_to_nchw_f32(e1, B * F, 4, H, W).reshape(B, F, 4, H, W).permute(0, 2, 1, 3, 4))
    assert torch.equal(outs[0].cpu(), out.cpu()), "captured graph (batched) differs from the eager forward"
    report(outs[1], out.cpu(), "C5 three-branch forward: 3 branch streams vs batched", 5e-3, 2e-2)


def test_c5_vae_24_frames_384x512_natural_chunking():
    """C5's VAE leg as benched: 24 frames at 384x512 exceed the 2 GiB operand window of the LDS-DMA loads, so encode / decode
    chunk the frames (_frames_per_call = 21 -> 21 + 3).  The chunked result must equal frame-by-frame work (frames are
    independent: instruct_p2p_video.py:57-79 decodes one at a time) up to the fp16 noise of other tile choices."""
    from insv2v import synth, shapes
    from insv2v.vae import AutoencoderKL
    vae = AutoencoderKL(**synth.VAE_FULL, device=DEV).load_state_dict(synth.synth_state_dict(shapes.vae_shapes(**synth.VAE_FULL)))
    N, H, W = 24, 384, 512
    assert vae._frames_per_call(H, W) < N
    x = synth.synth_input("vae.c5.x", (N, 3, H, W), kind="uniform")
    noise = synth.synth_input("vae.c5.noise", (N, 4, H // 8, W // 8))
    z = vae.encode(x, noise)
    assert z.shape == (N, 4, H // 8, W // 8) and torch.isfinite(z).all()
    img = vae.decode(z)
    assert img.shape == (N, 3, H, W) and torch.isfinite(img).all()
    for i in (0, 20, 21, 23):   # both chunks, both sides of the chunk boundary
        zi = vae.encode(x[i:i + 1], noise[i:i + 1])
        assert (zi - z[i:i + 1]).abs().max() <= 5e-3 * z.abs().max(), f"encode frame {i}"
        ii = vae.decode(z[i:i + 1])
        assert (ii - img[i:i + 1]).abs().max() <= 5e-3 * img.abs().max(), f"decode frame {i}"


def test_c4_full_width_long_video_unit(full_unet):
    """BASELINE config C4's unit at full width: a 32-frame 256x384 clip edited as 3 overlapping windows (16 + 12 + 4 new frames,
    4 / 12 reference frames, mean-delta noise correction for the first half of the steps: insv2v_run_loveu_tgve.py:119-165) through
    edit_video, with a short schedule (4 DDIM steps) to bound the time.  Checked: finite, deterministic, window plan, and the
    reference frames of every later window are pinned to the previous window's result while the correction is active
    (the noise correction makes x_t of the reference frames follow latent_ref exactly when noise_correct_step covers the step)."""
    from insv2v import synth, shapes
    from insv2v.model import create_model
    from insv2v.inference import InferenceIP2PVideo
    from insv2v.run_loveu_tgve import edit_video, split_batch
    model = create_model({"unet": {"params": synth.UNET_FULL}, "vae": {"params": synth.VAE_FULL}}, device=DEV)
    model.unet = full_unet
    model.vae.load_state_dict(synth.synth_state_dict(shapes.vae_shapes(**synth.VAE_FULL)))
    T, H, W = 32, 256, 384
    frames = synth.synth_input("c4.frames", (1, T, 3, H, W), kind="uniform")
    tc, tu = synth.synth_input("c4.tc", (1, 77, 768)), synth.synth_input("c4.tu", (1, 77, 768))
    news, refs = split_batch(torch.zeros(1, T, 1), 16, 4)
    assert [c.shape[1] for c in news] == [16, 12, 4] and refs == [4, 12]
    noises = [synth.synth_input(f"c4.noise.{k}", (1, c.shape[1], 4, H // 8, W // 8)) for k, c in enumerate(news)]
    enc = synth.synth_input("c4.enc", (1, T, 4, H // 8, W // 8))
    pipe = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=4)
    out1, lat1 = edit_video(model, pipe, frames, tc, tu, 7.5, 1.5, init_noises=noises, enc_noise=enc, return_latent=True)
    out2, lat2 = edit_video(model, pipe, frames, tc, tu, 7.5, 1.5, init_noises=noises, enc_noise=enc, return_latent=True)
    assert out1.shape == (1, T, 3, H, W) and lat1.shape == (1, T, 4, H // 8, W // 8)
    assert torch.isfinite(out1).all() and out1.abs().max() <= 1.0
    assert torch.equal(out1, out2) and torch.equal(lat1, lat2), "edit_video is not deterministic"
    # a different prompt must change the result (the pipeline is not ignoring its text input)
    out3 = edit_video(model, pipe, frames, synth.synth_input("c4.tc2", (1, 77, 768)), tu, 7.5, 1.5, init_noises=noises, enc_noise=enc)
    assert (out3 - out1).abs().max() > 1e-2


def test_c2_stacked_forward_vs_reference_golden(full_unet):
    """The bench's stacked-clip mode at full width: FOUR clips' CFG triples in one forward (B = 12, M = 294 912 tokens at level 0) - the
    token counts at which the dispatch switches kernels (persistent 256x256 GEMMs for the long-K linears and the 8x12-level
    convolutions, the register-resident row kernels, fused feed-forward / temporal attention at 12 samples).  Every triple carries
    the inputs of the C2 reference golden, so each is pinned by value; samples are independent, so the four results must also agree
    with each other bit for bit."""
    from insv2v import synth
    g = _gold("c2_unet_fwd")["out"]
    x = synth.synth_input("c2.sample", (3, 8, 16, 32, 48)).repeat(4, 1, 1, 1, 1)
    ctx = synth.synth_input("c2.ctx", (3, 77, 768)).repeat(4, 1, 1)
    out = full_unet(x, torch.full((12,), 981, dtype=torch.long), encoder_hidden_states=ctx).sample
    assert out.shape == (12, 4, 16, 32, 48) and torch.isfinite(out).all()
    for i in range(4):
        report(out[3 * i:3 * i + 3], g, f"C2 stacked forward (B = 12), clip {i} (reference golden)", 1e-2, 4e-2)
    for i in range(1, 4):
        assert torch.equal(out[:3], out[3 * i:3 * i + 3]), f"clip {i} differs from clip 0: samples are not independent"


@pytest.mark.parametrize("n", [20])   # (B = 30 is pinned by the 4-step run below through the bench's own comparison; round 6 cut the suite's time)
def test_c2_stacked_forward_B30_vs_reference_golden(full_unet, n):
    """The launch shapes bench.py times at the driver's command line: TEN or TWENTY clips' CFG triples in one forward (B = 30 / 60,
    M = 737 280 / 1 474 560 tokens at level 0) - where the dispatch moves the level-0/1 convolutions and the N = 640 / 960 / 1920 linears
    to the 256x320 ping-pong kernel, the 4x6-level convolutions to the 256x256 one, and the K = 640 row Linear to two token blocks per
    wave; at B = 60 the fused q/k/v rows of level 0 are a 2.8 GB operand (per-tile descriptor bases in insv2v_rowlin) and the normalised
    960-channel input of the first level-0 up block is convolved in two image-aligned parts (ops.conv3x3).
    Every triple carries the C2 reference golden's inputs: each is pinned by value and all must agree bit for bit."""
    from insv2v import synth
    g = _gold("c2_unet_fwd")["out"]
    x = synth.synth_input("c2.sample", (3, 8, 16, 32, 48)).repeat(n, 1, 1, 1, 1)
    ctx = synth.synth_input("c2.ctx", (3, 77, 768)).repeat(n, 1, 1)
    out = full_unet(x, torch.full((3 * n,), 981, dtype=torch.long), encoder_hidden_states=ctx).sample
    assert out.shape == (3 * n, 4, 16, 32, 48) and torch.isfinite(out).all()
    for i in range(n):
        report(out[3 * i:3 * i + 3], g, f"C2 stacked forward (B = {3 * n}), clip {i} (reference golden)", 1e-2, 4e-2)
    for i in range(1, n):
        assert torch.equal(out[:3], out[3 * i:3 * i + 3]), f"clip {i} differs from clip 0: samples are not independent"
    del out
    torch.cuda.empty_cache()


def test_cfg_prefix_dedup_matches_full_compute(full_unet):
    """Round 5: in a branch-major stack of CFG triples the samples of branches 1 and 2 - (no text, video) and (text, video),
    inference.py:183-194 - carry identical UNet inputs; forward_cl(cfg_clips=n) computes the first ResnetBlock3D, GroupNorm, proj_in,
    q/k/v and spatial self-attention once for the two and copies the rows.  Against the same forward with every sample computed on its
    own: the same values up to the kernels the two batch sizes pick (stated fp16 tolerance), and fewer launched FLOPs."""
    from insv2v import synth, ops
    from insv2v.inference import GraphedUNet
    n, F, H, W = 4, 16, 32, 48
    rows1 = F * H * W
    ctx = torch.cat([synth.synth_input("dd.tu", (1, 77, 768)).repeat(2 * n, 1, 1)] + [synth.synth_input(f"dd.tc.{c}", (1, 77, 768)) for c in range(n)], 0)
    outs, flops = [], []
    for clips in (0, n):
        r = GraphedUNet(full_unet, 3 * n, F, H, W, 77, use_graph=False, cfg_clips=clips)
        assert r.cfg_clips == clips
        r.set_context(ctx)
        for c in range(n):
            lat, cond = synth.synth_input(f"dd.lat.{c}", (F, 4, H, W)).to(DEV), synth.synth_input(f"dd.cond.{c}", (F, 4, H, W)).to(DEV)
            ops.build_unet_input(lat, cond, r.x_in[c * rows1:], r.t[c:], 981, 3, branch_rows=n * rows1, t_stride=n)
        rec = []
        ops.set_launch_recorder(rec)
        try:
            outs.append(r.run().clone())
            torch.cuda.synchronize()
        finally:
            ops.set_launch_recorder(None)
        flops.append(sum(x[1] for x in rec))
    report(outs[1], outs[0].cpu(), "UNet forward with the shared CFG prefix computed once vs every sample on its own (B = 12)", 5e-3, 2e-2)
    assert flops[1] < 0.995 * flops[0], flops


@pytest.mark.parametrize("n", [20])
def test_run_stacked_10_clips_full_width_vs_sequential(full_unet, n):
    """run_stacked as benched (10 or 20 clips = the stack cap, B = 30 / 60, captured graph) against the same clips run one at a time
    (3 branch streams), 4 DDIM steps at text 7.5 / video 1.5 on the C2 geometry: different kernels serve the two launch shapes, so the
    comparison is by the stated fp16 tolerance, not bit for bit; identical clips inside the stack must agree exactly."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideo, max_clips_in_flight
    assert max_clips_in_flight(16, 32, 48) >= n, "the stack would be split: this test pins ONE launch chain of n clips"
    pipe = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=4)
    calls = []
    for k in range(n):
        j = k % 3   # three distinct clips, repeated: equal inputs must give equal outputs inside one stack
        calls.append(dict(latent=synth.synth_input(f"st.lat.{j}", (1, 16, 4, 32, 48)), img_cond=synth.synth_input(f"st.cond.{j}", (1, 16, 4, 32, 48)),
                          text_cond=synth.synth_input(f"st.tc.{j}", (1, 77, 768)), text_uncond=synth.synth_input("st.tu", (1, 77, 768)),
                          text_cfg=7.5, img_cfg=1.5))
    res = pipe.run_stacked(calls)
    assert len(res) == n
    for k in range(3, n):
        assert torch.equal(res[k]["latent"], res[k % 3]["latent"]), f"stacked clip {k} differs from its twin {k % 3}"
    for j in range(3):
        one = pipe(**calls[j])
        report(res[j]["latent"], one["latent"].cpu(), f"run_stacked ({n} clips) clip {j} vs the single-clip run, 4 steps", 1e-2, 5e-2)


def test_c3_second_clip_full_width_vs_reference_golden(full_unet):
    """BASELINE config C3's call at full width and the C2 geometry, by value: second_clip_forward with R = 4 reference
    frames, noise correction for the first half of 4 DDIM steps - the mean-delta form (inference.py:216-289) and the optical-flow form
    (:291-398, flows injected as in the golden's fake estimator) - against goldens of the UNMODIFIED reference (tools/gen_golden.py,
    FULL_PARTS=c3second).  Stated tolerance of a short trajectory: rel-RMS <= 1e-2."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideo, InferenceIP2PVideoOpticalFlow
    g = _gold("c3_second_clip_full")
    F, h, w, R = 16, 32, 48, 4
    lat = synth.synth_input("c3.latent", (1, F, 4, h, w))
    cond = synth.synth_input("c3.cond", (1, F, 4, h, w))
    tc = synth.synth_input("c3.text_cond", (1, 77, 768))
    tu = synth.synth_input("c3.text_uncond", (1, 77, 768))
    lref = synth.synth_input("c3.latent_ref", (1, R, 4, h, w))
    r = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=4).second_clip_forward(
        lat, tc, tu, cond, latent_ref=lref, noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    report(r["all_pred"][0], g["second_clip_pred0"], "C3 full width: first x0 prediction of second_clip_forward (reference golden)", 1e-2, 5e-2)
    report(r["latent"], g["second_clip_latent"], "C3 full width: second_clip_forward, mean-delta correction, 4 steps (reference golden)", 1e-2, 5e-2)
    flows = [synth.synth_input(f"c3.flow{q}", (R, 2, h * 8, w * 8), scale=8.0) for q in range(F - R)]
    r = InferenceIP2PVideoOpticalFlow(full_unet, scheduler="ddim", num_ddim_steps=4).second_clip_forward(
        lat, tc, tu, cond, latent_ref=lref, flows=flows, noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    report(r["latent"], g["second_clip_flow_latent"], "C3 full width: second_clip_forward, optical-flow correction, 4 steps (reference golden)", 1e-2, 5e-2)


def test_c3_stacked_two_clips_vs_reference_golden(full_unet):
    """the flow-corrected window inside a STACK (run_stacked, B = 6): clip 0 carries the golden's flows, clip 1 the
    mean-delta correction of the same inputs - each against ITS reference golden, from one shared UNet launch chain; then the batched
    call surface (b = 2 through second_clip_forward, inference.py:183-187) with the flows per batch entry."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideoOpticalFlow
    g = _gold("c3_second_clip_full")
    F, h, w, R = 16, 32, 48, 4
    base = dict(latent=synth.synth_input("c3.latent", (1, F, 4, h, w)), img_cond=synth.synth_input("c3.cond", (1, F, 4, h, w)),
                text_cond=synth.synth_input("c3.text_cond", (1, 77, 768)), text_uncond=synth.synth_input("c3.text_uncond", (1, 77, 768)),
                latent_ref=synth.synth_input("c3.latent_ref", (1, R, 4, h, w)), noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    flows = [synth.synth_input(f"c3.flow{q}", (R, 2, h * 8, w * 8), scale=8.0) for q in range(F - R)]
    pipe = InferenceIP2PVideoOpticalFlow(full_unet, scheduler="ddim", num_ddim_steps=4)
    res = pipe.run_stacked([dict(base, flows=flows), dict(base)])
    report(res[0]["latent"], g["second_clip_flow_latent"], "C3 stacked (B = 6): optical-flow clip (reference golden)", 1e-2, 5e-2)
    report(res[1]["latent"], g["second_clip_latent"], "C3 stacked (B = 6): mean-delta clip of the same stack (reference golden)", 1e-2, 5e-2)
    res2 = pipe.run_stacked([dict(base, flows=flows), dict(base, flows=flows)])
    assert torch.equal(res2[0]["latent"], res2[1]["latent"]) and torch.equal(res2[0]["latent"], res[0]["latent"])


def test_c4_unit_carry_vs_reference_golden(full_unet):
    """the DRIVER-level carry of a C4 unit by value at full width - 32 frames = windows of 16 / 12 / 4 new frames,
    the overlap re-uses the INITIAL noise, latent_ref = the previous prediction's last R frames (R = 4, then 12), mean-delta correction for
    the first half of 4 DDIM steps - against a golden produced by executing the reference's OWN loop text (insv2v_run_loveu_tgve.py:119-162,
    tools/gen_golden.py FULL_PARTS=c4unit).  Through edit_video (one unit) AND edit_videos (two units stacked per window)."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideo
    from insv2v.run_loveu_tgve import edit_video, edit_videos
    g = _gold("c4_unit_full")
    T, h, w = 32, 32, 48

    class Model:   # the VAE legs are pinned elsewhere: the conditioning latent is handed over, the decode is the identity on latents
        scale_factor = 0.18215
        unet = full_unet

        def decode_latent_to_image(self, lat):
            return lat

    cond = synth.synth_input("c4.cond", (1, T, 4, h, w))
    tc = synth.synth_input("c4.text_cond", (1, 77, 768))
    tu = synth.synth_input("c4.text_uncond", (1, 77, 768))
    noises = [torch.from_numpy(g[f"noise{k}"]) for k in range(3)]
    pipe = InferenceIP2PVideo(full_unet, scheduler="ddim", num_ddim_steps=4)
    frames = torch.zeros(1, T, 3, 8, 8)
    _, lat = edit_video(Model(), pipe, frames, tc, tu, 7.5, 1.8, init_noises=noises, cond=cond, return_latent=True)
    report(lat, g["latent"], "C4 unit (32 frames, windows 16/12/4): edit_video latent (reference golden)", 1e-2, 5e-2)
    for k, (a, b) in enumerate(((0, 16), (16, 28), (28, 32))):
        report(lat[:, a:b], g["latent"][:, a:b], f"C4 unit: window {k} frames {a}-{b - 1}", 1e-2, 5e-2)
    unit = dict(frames=frames, text_cond=tc, text_uncond=tu, text_cfg=7.5, video_cfg=1.8, init_noises=noises, cond=cond)
    outs = edit_videos(Model(), pipe, [dict(unit), dict(unit)], return_latent=True)
    assert torch.equal(outs[0][1], outs[1][1])
    report(outs[0][1], g["latent"], "C4 unit: edit_videos, two units stacked per window (reference golden)", 1e-2, 5e-2)


def test_c2_ddpm_shipped_sampler_vs_reference_golden(full_unet):
    """the SHIPPED sampler (scheduler='ddpm', insv2v_run_loveu_tgve.py:65-75) at full width and the C2 geometry by
    value: 4 ancestral steps (t = 750, 500, 250, 0) with the variance noises the reference drew (replayed from its seeded generator and
    committed), text 7.5 / video 1.5 - the stochastic branch of cfg_step at full size; single clip and a 2-clip stack."""
    from insv2v import synth
    from insv2v.inference import InferenceIP2PVideo
    g = _gold("c2_ddpm4_full")
    lat = synth.synth_input("c2.latent", (1, 16, 4, 32, 48))
    cond = synth.synth_input("c2.cond", (1, 16, 4, 32, 48))
    tc = synth.synth_input("c2.text_cond", (1, 77, 768))
    tu = synth.synth_input("c2.text_uncond", (1, 77, 768))
    pipe = InferenceIP2PVideo(full_unet, scheduler="ddpm", num_ddim_steps=4)
    assert [int(t) for t in pipe.scheduler.timesteps] == [750, 500, 250, 0]
    pipe.variance_noises = [torch.from_numpy(g[f"noise{k}"]) for k in range(3)] + [None]
    r = pipe(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5)
    report(r["all_pred"][0], g["pred0"], "C2 DDPM: first x0 prediction (reference golden)", 1e-2, 5e-2)
    report(r["latent"], g["latent"], "C2 DDPM: 4 ancestral steps, injected variance noises (reference golden)", 1e-2, 5e-2)
    pipe.variance_noises = [torch.cat([torch.from_numpy(g[f"noise{k}"])] * 2, 0) for k in range(3)] + [None]
    rb = pipe(torch.cat([lat, lat], 0), torch.cat([tc, tc], 0), torch.cat([tu, tu], 0), torch.cat([cond, cond], 0), text_cfg=7.5, img_cfg=1.5)
    assert torch.equal(rb["latent"][0], rb["latent"][1])
    report(rb["latent"][:1], g["latent"], "C2 DDPM: batch of 2 through the stacked path (reference golden)", 1e-2, 5e-2)


def test_vae_encode_full_size_vs_reference_golden():
    """The VAE encoder at the bench's frame size (one 256x384 frame): moments and posterior sample against the reference's Encoder
    (modules/vqvae/model.py:277-302 + kl_autoencoder/autoencoder.py:10-23,89-95) - the full-size encode golden round 3 lacked."""
    from insv2v import synth, shapes
    from insv2v.vae import AutoencoderKL
    g = _gold("vae_encode_full")
    vae = AutoencoderKL(**synth.VAE_FULL, device=DEV).load_state_dict(synth.synth_state_dict(shapes.vae_shapes(**synth.VAE_FULL)))
    x = synth.synth_input("vae.full.x", (1, 3, 256, 384), kind="uniform")
    noise = synth.synth_input("vae.full.noise", (1, 4, 32, 48))
    report(vae.encode(x, noise), g["enc_sample"], "VAE encode 256x384: posterior sample (reference golden)", 1e-2, 4e-2)
