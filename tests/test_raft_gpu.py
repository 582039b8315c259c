"""Optical-flow estimator on the HIP kernels (RAFTFlow, misc_utils/flow_utils.py:134-189) against the CPU fp32
restatement of torchvision's raft_large (oracle/raft.py).  PARITY UNPINNED for the third-party arithmetic: torchvision is absent from the
reference tree and from this image, so these tests pin HIP == oracle, and the oracle is the build's reading of the published network.

Stated fp16 tolerance: single kernels <= 2e-3 of max|ref| (4e-3 for the norm); the whole estimator (12 recurrent updates, fp16
activations, fp32 correspondences) rel-RMS <= 2e-2 on the final flow, and a mean end-point error below 0.25 px at 256x384.
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(out, ref, tol, what):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    err = (out - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    print(f"[parity] {what}: max err {err:.3e} ({err / scale:.3e} of max|ref|)")
    assert math.isfinite(err) and err <= tol * scale, f"{what}: {err / scale:.3e} of max|ref| (tol {tol})"


@pytest.mark.parametrize("cfg", [dict(C=8, kh=7, kw=7, stride=2), dict(C=64, kh=3, kw=3, stride=1), dict(C=96, kh=3, kw=3, stride=2),
                                 dict(C=64, kh=1, kw=1, stride=2), dict(C=384, kh=1, kw=5, stride=1), dict(C=384, kh=5, kw=1, stride=1, split=128)])
def test_im2col_gemm_is_conv2d(cfg):
    """insv2v_im2col + insv2v_gemm == F.conv2d for every kernel geometry of the estimator (7x7 s2, 3x3 s1 / s2, 1x1 s2, 1x5, 5x1), with the
    two-source channel concat of the ConvGRU."""
    from insv2v import ops
    torch.manual_seed(0)
    N, H, W, C, kh, kw, st = 2, 20, 28, cfg["C"], cfg["kh"], cfg["kw"], cfg["stride"]
    x = torch.randn(N, C, H, W).half().float()
    w = (torch.randn(40, C, kh, kw) * (C * kh * kw) ** -0.5).half().float()
    b = torch.randn(40)
    ref = F.relu(F.conv2d(x, w, b, stride=st, padding=((kh - 1) // 2, (kw - 1) // 2)))
    rows = x.permute(0, 2, 3, 1).reshape(N * H * W, C).half().to(DEV).contiguous()
    split = cfg.get("split")
    if split:
        cols, g = ops.im2col(rows[:, :split].contiguous(), (N, H, W), C, kh, kw, st, ((kh - 1) // 2, (kw - 1) // 2), x2=rows[:, split:].contiguous())
    else:
        cols, g = ops.im2col(rows, (N, H, W), C, kh, kw, st, ((kh - 1) // 2, (kw - 1) // 2))
    wk = w.permute(0, 2, 3, 1).reshape(40, -1).half().to(DEV).contiguous()
    out = ops.gemm(cols, wk, b.to(DEV), act=ops.ACT_RELU)
    assert g == (N, ref.shape[2], ref.shape[3])
    close(out.reshape(N, g[1], g[2], 40).permute(0, 3, 1, 2), ref, 2e-3, f"im2col + gemm vs conv2d {cfg}")


@pytest.mark.parametrize("act,fn", [(5, torch.sigmoid), (6, torch.tanh), (4, F.relu)])
def test_gemm_raft_activations(act, fn):
    from insv2v import ops
    torch.manual_seed(1)
    a, w, b = torch.randn(300, 72).half(), (torch.randn(48, 72) * 0.3).half(), torch.randn(48) * 3
    out = ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), act=act)
    close(out, fn(a.float() @ w.float().T + b), 2e-3, f"gemm act {act}")


@pytest.mark.parametrize("C,relu", [(64, True), (96, True), (128, False), (256, True)])
def test_instance_norm(C, relu):
    from insv2v import ops
    torch.manual_seed(2)
    N, HW = 3, 1200
    x = (torch.randn(N, HW, C) * 2 + 5 * torch.randn(1, 1, C)).half()
    ref = F.instance_norm(x.float().permute(0, 2, 1).reshape(N, C, HW, 1)).reshape(N, C, HW).permute(0, 2, 1)
    if relu:
        ref = F.relu(ref)
    out = ops.instance_norm(x.reshape(N * HW, C).to(DEV), N, HW, relu=relu)
    close(out.reshape(N, HW, C), ref, 4e-3, f"instance norm C={C}")


def test_elementwise_ops_and_column_slices():
    from insv2v import ops
    torch.manual_seed(3)
    a, b, c = torch.randn(500, 128).half(), torch.randn(500, 128).half(), torch.rand(500, 128).half()
    buf = torch.zeros(500, 384, dtype=torch.float16, device=DEV)
    ops.ew(ops.EW_TANH, a.to(DEV), out=buf[:, 128:256])
    close(buf[:, 128:256], torch.tanh(a.float()), 2e-3, "tanh into a column slice")
    assert float(buf[:, :128].abs().max()) == 0 and float(buf[:, 256:].abs().max()) == 0
    close(ops.ew(ops.EW_RELU, a.to(DEV)), F.relu(a.float()), 1e-3, "relu")
    close(ops.ew(ops.EW_ADD_RELU, a.to(DEV), b.to(DEV)), F.relu(a.float() + b.float()), 2e-3, "add + relu")
    close(ops.ew(ops.EW_GRU_RH, a.to(DEV), b.to(DEV)), a.float() * b.float(), 2e-3, "r * h")
    close(ops.ew(ops.EW_GRU_OUT, a.to(DEV), b.to(DEV), c.to(DEV)), (1 - c.float()) * b.float() + c.float() * a.float(), 2e-3, "GRU blend")


def test_correlation_pyramid_lookup_and_upsample_vs_oracle():
    """CorrBlock (all-pairs volume, pooled pyramid, 4 x 81 bilinear look-ups incl. out-of-range zeros) and upsample_flow, each against
    oracle/raft.py on the same operands."""
    from insv2v import ops
    from oracle.raft import CorrBlock, upsample_flow, coords_grid
    torch.manual_seed(4)
    B, C, h, w = 2, 64, 16, 24
    f1, f2 = torch.randn(B, C, h, w).half().float(), torch.randn(B, C, h, w).half().float()
    cb = CorrBlock(4, 4)
    cb.build_pyramid(f1, f2)
    coords = coords_grid(B, h, w) + torch.randn(B, 2, h, w) * 6      # large displacements: many look-ups fall outside
    ref = cb.index_pyramid(coords)
    r1 = f1.permute(0, 2, 3, 1).reshape(B * h * w, C).half().to(DEV).contiguous()
    r2 = f2.permute(0, 2, 3, 1).reshape(B * h * w, C).half().to(DEV).contiguous()
    rows = h * w
    corr = torch.empty((B, rows, rows), device=DEV, dtype=torch.float32)
    ops.gemm(r1, r2, out=corr, out_fp32=True, alpha=C ** -0.5, batch=B, M=rows, N=rows, K=C, a_bs=rows * C, w_bs=rows * C, c_bs=rows * rows,
             lda=C, ldw=C, ldc=rows)
    close(corr.reshape(B * rows, 1, h, w), cb.pyramid[0], 1e-3, "all-pairs correlation volume")
    pyr = [corr.reshape(B * rows, h, w)]
    for lv in range(1, 4):
        pyr.append(ops.avgpool2x2(pyr[-1], B * rows, h >> (lv - 1), w >> (lv - 1)))
        close(pyr[-1], cb.pyramid[lv][:, 0], 1e-3, f"pyramid level {lv}")
    out = ops.corr_lookup(pyr, coords.to(DEV).contiguous(), B, h, w, 4, 328)
    close(out[:, :324].reshape(B, h, w, 324).permute(0, 3, 1, 2), ref, 2e-3, "correlation look-up (4 levels x 81)")
    assert float(out[:, 324:].abs().max()) == 0
    mask = torch.randn(B, 576, h, w).half().float()
    up_ref = upsample_flow(coords - coords_grid(B, h, w), mask)
    up = ops.convex_upsample(coords.to(DEV).contiguous(), mask.permute(0, 2, 3, 1).reshape(B * rows, 576).half().to(DEV).contiguous(), B, h, w)
    close(up, up_ref, 1e-3, "convex upsampling")
    c1 = coords.to(DEV).contiguous()
    delta = torch.randn(B * rows, 8, device=DEV)
    fr = torch.full((B * rows, 8), 7.0, device=DEV, dtype=torch.float16)
    ops.raft_flow_rows(c1, delta, fr, B, h, w)
    want = coords + delta[:, :2].cpu().reshape(B, h, w, 2).permute(0, 3, 1, 2)
    close(c1, want, 1e-6, "coords1 += delta")
    close(fr[:, :2].float().reshape(B, h, w, 2).permute(0, 3, 1, 2), want - coords_grid(B, h, w), 1e-3, "flow rows")
    assert float(fr[:, 2:].abs().max()) == 0


@pytest.mark.parametrize("B,H,W", [(2, 128, 192), (4, 256, 384)])
def test_raftflow_vs_oracle(B, H, W):
    """The whole estimator as the optical-flow pipe calls it (inference.py:303-311: query frame repeated against the R reference frames):
    key-hashed weights under torchvision's key names, frames in [-1, 1], 12 updates, the last prediction."""
    from insv2v import shapes, synth
    from insv2v.raft import RAFTFlow
    from oracle.raft import RAFTFlow as OracleFlow
    sd = synth.synth_raft_state_dict(shapes.raft_shapes())
    ora = OracleFlow()
    ora.model.load_state_dict(sd)
    q = synth.synth_input("raft.query", (1, 3, H, W), kind="uniform").repeat(B, 1, 1, 1)
    refs = synth.synth_input("raft.refs", (B, 3, H, W), kind="uniform")
    torch.set_num_threads(min(16, os.cpu_count()))   # torch's CPU kernels regress beyond ~16 threads on the 256-core hosts (tools/cpu_threads_probe.py)
    want = ora(q, refs)
    got = RAFTFlow(DEV, sd)(q, refs)
    assert tuple(got.shape) == (B, 2, H, W)
    rms = ((got.cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    epe = (got.cpu() - want).pow(2).sum(1).sqrt().mean().item()
    print(f"[parity] RAFTFlow {B}x{H}x{W}: rel-rms {rms:.3e}, mean end-point error {epe:.4f} px (|flow| mean {want.abs().mean().item():.2f} px)")
    assert math.isfinite(rms) and rms <= 2e-2 and epe <= 0.25, (rms, epe)
    # deterministic
    assert torch.equal(got, RAFTFlow(DEV, sd)(q, refs))


def test_raftflow_load_state_dict_from_checkpoint_file_with_real_batchnorm_statistics(tmp_path):
    """RAFTFlow().load_state_dict on a CHECKPOINT FILE with torchvision's exact key set and dtypes: fp32 tensors,
    int64 ``num_batches_tracked``, the optional ``model.`` prefix of the reference's wrapper module (flow_utils.py:157), and BatchNorm
    running statistics far from the identity (mean up to +-2, variance 0.25 .. 4, gammas -1 .. 1.5) - the folding of the context
    encoder's BatchNorm into its convolutions is exercised with the statistics a trained checkpoint carries, HIP against the oracle."""
    from insv2v import shapes, synth
    from insv2v.raft import RAFTFlow
    from oracle.raft import RAFTFlow as OracleFlow
    sd = synth.synth_raft_state_dict(shapes.raft_shapes())
    g = torch.Generator().manual_seed(7)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = 2.0 * (torch.rand(sd[k].shape, generator=g) * 2 - 1)
        elif k.endswith("running_var"):
            sd[k] = torch.exp(torch.rand(sd[k].shape, generator=g) * 2.77 - 1.386)       # 0.25 .. 4
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(123456, dtype=torch.long)
        elif ".1.weight" in k and "context_encoder" in k:                                # BatchNorm gamma
            sd[k] = (torch.rand(sd[k].shape, generator=g) * 2.5 - 1)                      # -1 .. 1.5
        elif ".1.bias" in k and "context_encoder" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.5
    assert set(sd) == set(shapes.raft_shapes()) and all(v.dtype == (torch.long if k.endswith("num_batches_tracked") else torch.float32) for k, v in sd.items())
    path = tmp_path / "raft_large_synth.pth"
    torch.save({"model." + k: v for k, v in sd.items()}, path)
    loaded = torch.load(path, map_location="cpu")
    est = RAFTFlow(DEV).load_state_dict(loaded)
    ora = OracleFlow()
    ora.model.load_state_dict(sd)
    B, H, W = 2, 128, 192
    q = synth.synth_input("raftbn.query", (B, 3, H, W), kind="uniform")
    r = synth.synth_input("raftbn.refs", (B, 3, H, W), kind="uniform")
    want = ora(q, r)
    got = est(q, r)
    rms = ((got.cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    epe = (got.cpu() - want).pow(2).sum(1).sqrt().mean().item()
    print(f"[parity] RAFTFlow from a checkpoint file, non-identity BatchNorm statistics: rel-rms {rms:.3e}, mean end-point error {epe:.4f} px (|flow| mean {want.abs().mean().item():.2f} px)")
    assert math.isfinite(rms) and rms <= 2e-2 and epe <= 0.25, (rms, epe)
    # a missing key is an error, as with torch's strict loading
    bad = dict(loaded)
    bad.pop("model.context_encoder.convnormrelu.1.running_var")
    with pytest.raises((KeyError, RuntimeError)):
        RAFTFlow(DEV).load_state_dict(bad)


def test_optical_flow_pipe_builds_the_estimator_like_the_reference():
    """InferenceIP2PVideoOpticalFlow(raft_state_dict=...) builds RAFTFlow itself (inference.py:294) and second_clip_forward(ref_images=,
    query_images=) estimates the flows (:303-311, :337) - same result as handing the estimated flows over; the batched estimator calls
    (several query frames per call) agree with the reference's one-query-per-call loop within fp16 GEMM tile effects."""
    from insv2v import shapes, synth
    from insv2v.unet import UNet3DConditionModel
    from insv2v.inference import InferenceIP2PVideoOpticalFlow
    unet = UNet3DConditionModel(**synth.UNET_TINY, device=DEV).load_state_dict(synth.synth_state_dict(shapes.unet_shapes(**synth.UNET_TINY)))
    rsd = synth.synth_raft_state_dict(shapes.raft_shapes())
    pipe = InferenceIP2PVideoOpticalFlow(unet, scheduler="ddim", num_ddim_steps=2, raft_state_dict=rsd)
    F_, h, w, R = 8, 16, 24, 4
    lat, cond = synth.synth_input("rp.lat", (1, F_, 4, h, w)), synth.synth_input("rp.cond", (1, F_, 4, h, w))
    tc, tu = synth.synth_input("rp.tc", (1, 77, 64)), synth.synth_input("rp.tu", (1, 77, 64))
    lref = synth.synth_input("rp.lref", (1, R, 4, h, w))
    imgs_r = synth.synth_input("rp.ir", (1, R, 3, 8 * h, 8 * w), kind="uniform").to(DEV)
    imgs_q = synth.synth_input("rp.iq", (1, F_ - R, 3, 8 * h, 8 * w), kind="uniform").to(DEV)
    kw = dict(latent_ref=lref, noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5)
    r1 = pipe.second_clip_forward(lat, tc, tu, cond, ref_images=imgs_r, query_images=imgs_q, **kw)
    flows = pipe.obtain_flow_batched(imgs_r[0], imgs_q[0])
    assert len(flows) == F_ - R and tuple(flows[0].shape) == (R, 2, 8 * h, 8 * w)
    r2 = pipe.second_clip_forward(lat, tc, tu, cond, flows=flows, **kw)
    assert torch.equal(r1["latent"], r2["latent"])
    r0 = pipe.second_clip_forward(lat, tc, tu, cond, flows=[torch.zeros_like(f) for f in flows], **kw)
    assert (r0["latent"] - r1["latent"]).abs().max() > 1e-3, "the estimated flows must matter"
    loop = [pipe.flow_estimator(q.unsqueeze(0).repeat(R, 1, 1, 1), imgs_r[0]) for q in imgs_q[0]]
    for a, b in zip(flows, loop):
        close(a, b, 2e-3, "batched estimator call vs one query per call")
    # a stack of two clips estimates per clip
    rs = pipe.run_stacked([dict(latent=lat, text_cond=tc, text_uncond=tu, img_cond=cond, ref_images=imgs_r, query_images=imgs_q, **kw)] * 2)
    assert torch.equal(rs[0]["latent"], rs[1]["latent"])
    close(rs[0]["latent"], r1["latent"], 2e-2, "stacked optical-flow clip vs single")
