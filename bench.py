#!/usr/bin/env python3
"""Headline benchmark: denoised video-frames/sec (16 f @ 256x384, 50 DDIM steps) on N MI355X.

One "step" = the whole hot path for one synthetic clip (unit): VAE encode of 16 frames -> 50 DDIM
steps of the 3-way-CFG UNet -> VAE decode.  Inputs are resident in HBM before the timed region.
Units are independent, so up to --concurrent-clips of the K timed steps are stacked into every UNet launch
(auto: as few and as even groups as possible of at most inference.max_clips_in_flight clips - 20 at C2 -, B = 3 x clips,
branch-major with the shared CFG prefix computed once); K steps are timed in total.
N > 1: one process per GPU, every rank edits its own clips (weak scaling, no data-path collective) and the edited
frames are collected with ONE all_gather (RCCL) inside the timed region.  Launched under torch.distributed.run the
ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; a plain `python bench.py --gpus N` launches them itself
(self_launch_command).  Prints one JSON line on rank 0 (contract in the task statement).

roofline: the dominant kernel family is the MFMA GEMM / implicit-conv / row-kernel family: its
algorithmic FLOPs (2*M*N*K per launch, unpadded) over its summed launch durations, measured with
HIP events on the launch stream during one instrumented UNet forward of the same workload.
cpu_baseline: the fp32 CPU oracle timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_MFMA_F16_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0         # HBM3E peak, MI355X_MICROARCH.md
# HBM-side traffic per kernel family for ONE eager UNet forward comes from a separate rocprofv3 --pmc pass
# (tools/pmc_forward_traffic.py writes this file; collected and corrected as MI355X_MICROARCH.md prescribes).  It is
# only reported when the file's shape key matches the benchmarked workload, otherwise `traffic` is null.
PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_forward_traffic.json")


def algorithmic_bytes(tag):
    """Compulsory HBM bytes of one launch from its recorded shape tag (ops._timed): every operand and result moved
    exactly once at its stored width (fp16 activations / weights, fp32 statistics).  This is the per-launch figure
    `roofline.achieved` / `algorithmic_bytes_per_launch` are computed from; DESIGN.md section 5 states the same model."""
    kind = tag[0]
    if kind == "lin":
        _, M, N, K, batch, act, res = tag
        n_out = N // 2 if act == 2 else N
        return batch * 2.0 * (M * K + N * K + M * n_out * (2 if res else 1))
    if kind == "ffn":      # fused LayerNorm + GEGLU feed-forward + residual: x in, out out, weights once
        _, M, C, hidden = tag
        return 2.0 * (2 * M * C + 3 * C * hidden)
    if kind == "tattn":    # fused temporal attention sub-block: x in, out out, q/k/v/o weights once
        _, M, C, heads, frames = tag
        return 2.0 * (2 * M * C + 4 * C * C)
    if kind == "tattn_attn":   # fused LayerNorm + q/k/v + temporal attention (C = 640): x in, attention output out, q/k/v weights once
        _, M, C, heads, frames = tag
        return 2.0 * (2 * M * C + 3 * C * C)
    if kind == "xattn_attn":   # fused LayerNorm + q + text cross-attention (C = 640): x in, attention output out, q weights once
        _, M, C, heads, L = tag
        return 2.0 * (2 * M * C + C * C)
    if kind == "xattn":    # fused text cross-attention sub-block: x in, out out, q / o weights once (the text K / V are a few hundred KB)
        _, M, C, heads, L, pre = tag
        return 2.0 * ((3 if pre else 2) * M * C + (3 if pre else 2) * C * C)
    if kind == "rowlin":   # register-resident Linear: x, W, out (+ residual)
        _, M, N, K, ln, res = tag
        return 2.0 * (M * K + N * K + M * N * (2 if res else 1))
    if kind == "conv":
        _, M, N, K, stride, up, res = tag
        cin = K // 9
        m_in = M * stride * stride if not up else M / 4.0
        return 2.0 * (m_in * cin + N * K + M * N * (2 if res else 1))
    if kind == "wino_gemm":   # the 16 transformed-tap GEMMs of a Winograd convolution as one grouped launch: V in, U once, M out
        _, rows, N, K = tag
        return 2.0 * (rows * K + 9 * N * K + rows * N)
    if kind == "wino_in":     # Winograd input transform (GroupNorm apply + SiLU + B^T d B): one read, 4 x write
        _, M, C = tag
        return 2.0 * M * C * 5
    if kind == "wino_out":    # Winograd output transform: 4 x read, one write
        _, M, C = tag
        return 2.0 * M * C * 5
    if kind == "attn":
        _, batch, heads, d, sq, sk = tag
        return 2.0 * batch * heads * d * (2 * sq + 2 * sk)          # q, o, k, v once
    if kind == "gn":
        _, ns, rows, C = tag
        return 2.0 * ns * rows * C * 2                              # one read + one write (statistics fused ideally)
    if kind == "gnstats":
        _, ns, rows, C = tag
        return 2.0 * ns * rows * C                                  # one read, no write
    if kind in ("lnstats", "ln"):
        _, rows, C = tag
        return 2.0 * rows * C * (1 if kind == "lnstats" else 2)
    if kind == "copy":     # duplicated CFG-branch rows (unet.forward_cl(cfg_clips=...)): one read + one write
        _, rows, C = tag
        return 2.0 * rows * C * 2
    return 0.0


def top_shapes(records, family="gemm_kernel", n=3):
    """The launch shapes of one kernel family that take the most time in the recorded forward: [(tag, launches, seconds, flops)] -> the
    `roofline.top_shapes` list (what a per-kernel reading of the family figure needs: the dominant SINGLE shapes with their own rate).
    `records` = (family name, flops, seconds, tag) per launch."""
    agg = {}
    for name, flops, sec, tag in records:
        if name != family or tag is None:
            continue
        d = agg.setdefault(tuple(tag), [0, 0.0, 0.0])
        d[0] += 1; d[1] += sec; d[2] += flops
    out = []
    for tag, (cnt, sec, flops) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
        ach = flops / max(sec, 1e-12) / 1e12
        out.append({"shape": [x if isinstance(x, (str, bool)) else int(x) for x in tag], "launches": cnt, "ms": round(1e3 * sec, 3),
                    "avg_launch_us": round(1e6 * sec / cnt, 1), "achieved": ach, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F16_TFLOPS,
                    "algorithmic_gbytes_per_launch": round(algorithmic_bytes(tag) / 1e9, 4)})
    return out


MAX_CLIPS_IN_FLIGHT = 20


def max_clips_in_flight(frames=16, h=32, w=48):
    """insv2v.inference.max_clips_in_flight (the product's own cap, applied by run_stacked): clips per launch chain."""
    from insv2v.inference import max_clips_in_flight as f
    return f(frames, h, w)


def clip_groups(steps, concurrent, plain=True, cap=MAX_CLIPS_IN_FLIGHT):
    """How the K timed steps are scheduled on one GPU: returns (clips in flight, group sizes).  concurrent <= 0 = auto: as few, as
    large and as even groups as possible with at most `cap` clips stacked into a launch - 5 -> [5], 12 -> [12],
    20 -> [20], 25 -> [13, 12] (measured at --steps 20 on one box: 4 in flight 12.17, 5: 12.65, 10: 12.83 frames/s,
    profiles/r03_clips_in_flight.txt; round 4: 10 -> 20 in flight 14.57 -> 14.70, profiles/r04_clips20.txt); fewer than 3 steps, or a mode without a concurrent form (flow correction, long video): one
    clip at a time."""
    if concurrent <= 0:
        ng = max(1, -(-steps // max(1, cap)))
        concurrent = -(-steps // ng) if (plain and steps >= 3) else 1
    if not plain:
        concurrent = 1
    cc = max(1, concurrent)
    ngroups = max(1, -(-steps // cc))
    sizes = [steps // ngroups + (1 if g < steps % ngroups else 0) for g in range(ngroups)]   # as even as possible, each <= cc
    return cc, sizes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--driver-mode", action="store_true",
                    help="time the units through the product driver's code (insv2v.run_loveu_tgve.edit_videos) instead of the bench's own stacking")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-branch-streams", action="store_true", help="batch the 3 CFG branches in every launch instead of 3 HIP streams")
    ap.add_argument("--branch-streams", action="store_true", help="force one HIP stream per CFG branch (default for fewer than 3 concurrent clips)")
    ap.add_argument("--concurrent-clips", type=int, default=0,
                    help="independent clips (timed steps) in flight on one GPU at once; 0 = auto: as few, as even groups as possible of at most "
                         "inference.max_clips_in_flight clips (20 at C2: the default 5 steps run as one group of 5, 20 steps as one group "
                         "of 20, 25 as 13 + 12), fewer than 3 steps: 1")
    ap.add_argument("--clip-mode", choices=["stacked", "streams"], default="stacked",
                    help="how a group of clips shares the GPU: 'stacked' = ONE UNet launch chain with B = 3 x clips (run_stacked: weights read once "
                         "per group, every launch chip-filling); 'streams' = one HIP stream + captured graph per clip, DDIM loops interleaved "
                         "(run_concurrent, the round-2 mode).  A/B: profiles/r03_clip_mode_ab.txt")
    ap.add_argument("--tiny", action="store_true", help="reduced-width model (plumbing check, not a valid bench)")
    ap.add_argument("--flow-correction", action="store_true",
                    help="config C3: second_clip_forward with optical-flow noise correction (R=4 reference frames, synthetic flows, "
                         "noise_correct_step 0.5) instead of the plain loop; not the headline metric")
    ap.add_argument("--raft", action="store_true",
                    help="with --flow-correction: estimate the flows inside the timed region with the RAFT network on the HIP kernels "
                         "(key-hashed weights; R = 4 reference frames x 12 query frames per clip) instead of handing over synthetic flows")
    ap.add_argument("--long-video", action="store_true",
                    help="config C4's unit: a 32-frame clip edited as 3 overlapping 16-frame windows (16 + 12 + 4 new frames, 4 / 12 "
                         "reference frames, mean-delta noise correction) through run_loveu_tgve.edit_video; not the headline metric")
    return ap.parse_args()


def self_launch_command(gpus, argv, port=None):
    """The command a plain `python bench.py --gpus N` (N > 1, no RANK in the environment) re-executes itself as: one rank per
    GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 (the form the driver itself uses for N > 1)."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:   # launched plainly: start the ranks ourselves
        if not torch.cuda.is_available() or torch.cuda.device_count() < a.gpus:
            sys.exit(f"bench.py --gpus {a.gpus}: {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible on this node (no GPU fallback exists)")
        import subprocess
        sys.exit(subprocess.call(self_launch_command(a.gpus, sys.argv[1:]), env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus"
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    # under torch.distributed.run (the driver's launch for N > 1; tests/test_dist_gpu.py at N = 1) the barrier and the max over ranks go
    # through RCCL whatever the world size; a plain `python bench.py` needs no process group
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    from insv2v import synth, shapes, ops
    from insv2v.model import create_model
    from insv2v.inference import InferenceIP2PVideo

    ucfg, vcfg = (synth.UNET_TINY, synth.VAE_TINY) if a.tiny else (synth.UNET_FULL, synth.VAE_FULL)
    ctx_dim = ucfg["cross_attention_dim"]
    model = create_model({"unet": {"params": ucfg}, "vae": {"params": vcfg}}, device=str(dev))
    usd = synth.synth_state_dict(shapes.unet_shapes(**ucfg))
    model.unet.load_state_dict(usd)
    if not (rank == 0 and world == 1 and not a.no_cpu_baseline):
        del usd
    model.vae.load_state_dict(synth.synth_state_dict(shapes.vae_shapes(**vcfg)))
    if a.flow_correction:
        from insv2v.inference import InferenceIP2PVideoOpticalFlow as PipeCls
    else:
        PipeCls = InferenceIP2PVideo
    # Throughput mode: units are independent (insv2v_run_loveu_tgve.py:83,101), so several can be in flight on one GPU.  With >= 3 clips
    # interleaved, every launch carries all 3 CFG branches (chip-filling kernels) and the OTHER clips fill its launch gaps and tails;
    # with one clip the three branch streams do that job.
    # (round 5: the optical-flow correction is per-clip elementwise work behind the shared UNet launch - C3 stacks like C2)
    plain = not a.long_video or a.driver_mode   # the driver's edit_videos stacks long-video units too
    a.concurrent_clips, sizes = clip_groups(a.steps, a.concurrent_clips, plain, max_clips_in_flight(a.frames, a.height // 8, a.width // 8))
    if a.concurrent_clips >= 3 and not a.branch_streams:
        a.no_branch_streams = True
    pkw = {}
    if a.raft:
        assert a.flow_correction, "--raft is a mode of --flow-correction"
        pkw["raft_state_dict"] = synth.synth_raft_state_dict(shapes.raft_shapes())
    pipe = PipeCls(model.unet, scheduler="ddim", num_ddim_steps=a.ddim_steps, use_graph=not a.no_graph, branch_streams=not a.no_branch_streams, **pkw)

    if a.long_video:
        a.frames = 32
    F, H, W = a.frames, a.height, a.width
    h, w = H // 8, W // 8
    n_units = a.steps + a.warmup
    frames = [synth.synth_input(f"bench.frames.{rank}.{i}", (1, F, 3, H, W), kind="uniform").to(dev) for i in range(min(n_units, 2))]
    text_cond = synth.synth_input("bench.text_cond", (1, 77, ctx_dim)).to(dev)
    text_uncond = synth.synth_input("bench.text_uncond", (1, 77, ctx_dim)).to(dev)
    init = synth.synth_input(f"bench.init.{rank}", (1, F, 4, h, w)).to(dev)
    enc_noise = synth.synth_input(f"bench.enc.{rank}", (1, F, 4, h, w)).to(dev)

    R = 4
    if a.flow_correction:  # flows ~ N(0, 8 px) at image resolution, one [R,2,H,W] set per query frame
        lref = synth.synth_input(f"bench.lref.{rank}", (1, R, 4, h, w)).to(dev)
        flows = [synth.synth_input(f"bench.flow.{rank}.{q}", (R, 2, H, W), scale=8.0).to(dev) for q in range(F - R)]
        ref_imgs = synth.synth_input(f"bench.refimg.{rank}", (1, R, 3, H, W), kind="uniform").to(dev)   # the previous window's last R frames
    breakdown = {}

    if a.long_video:
        from insv2v.run_loveu_tgve import edit_video, split_batch
        news, _ = split_batch(torch.zeros(1, F, 1), 16, 4)
        lv_noises = [synth.synth_input(f"bench.lv.{rank}.{k}", (1, c.shape[1], 4, h, w)).to(dev) for k, c in enumerate(news)]

    def one_unit(i, timed=False):
        fr = frames[i % len(frames)]
        if a.long_video:
            return edit_video(model, pipe, fr, text_cond, text_uncond, 7.5, 1.5, init_noises=lv_noises, enc_noise=enc_noise)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            ev[0].record()
        cond = model.encode_image_to_latent(fr, enc_noise) / model.scale_factor
        if timed:
            ev[1].record()
        if a.flow_correction:
            fkw = dict(ref_images=ref_imgs, query_images=fr[:, R:]) if a.raft else dict(flows=flows)
            lat = pipe.second_clip_forward(latent=init, text_cond=text_cond, text_uncond=text_uncond, img_cond=cond, latent_ref=lref,
                                           noise_correct_step=0.5, text_cfg=7.5, img_cfg=1.5, **fkw)["latent"]
        else:
            lat = pipe(latent=init, text_cond=text_cond, text_uncond=text_uncond, img_cond=cond, text_cfg=7.5, img_cfg=1.5)["latent"]
        if timed:
            ev[2].record()
        img = model.decode_latent_to_image(lat).clip(-1, 1)
        if timed:
            ev[3].record()
            torch.cuda.synchronize()
            breakdown.update(vae_encode_ms=ev[0].elapsed_time(ev[1]), ddim_loop_ms=ev[1].elapsed_time(ev[2]),
                             vae_decode_ms=ev[2].elapsed_time(ev[3]))
        return img

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_unit(i, timed=(i == a.warmup - 1 and i > 0))  # stage breakdown from the last (already warm) warm-up unit
    def units(idx):
        """Edit len(idx) clips; their sampling loops are interleaved (independent units, clip-parallel on one GPU)."""
        if a.driver_mode:   # the product's own unit loop (run_loveu_tgve.edit_videos: VAE encode, stacked windows, VAE decode)
            from insv2v.run_loveu_tgve import edit_videos
            return edit_videos(model, pipe, [dict(frames=frames[i % len(frames)], text_cond=text_cond, text_uncond=text_uncond, text_cfg=7.5,
                                                  video_cfg=1.5, init_noises=lv_noises if a.long_video else [init], enc_noise=enc_noise) for i in idx])
        if len(idx) == 1:
            return [one_unit(idx[0])]
        conds = [model.encode_image_to_latent(frames[i % len(frames)], enc_noise) / model.scale_factor for i in idx]
        run = pipe.run_stacked if a.clip_mode == "stacked" else pipe.run_concurrent
        extra = [dict(latent_ref=lref, noise_correct_step=0.5, **(dict(ref_images=ref_imgs, query_images=frames[i % len(frames)][:, R:]) if a.raft else dict(flows=flows)))
                 if a.flow_correction else {} for i in idx]
        res = run([dict(latent=init, text_cond=text_cond, text_uncond=text_uncond, img_cond=c, text_cfg=7.5, img_cfg=1.5, **e) for c, e in zip(conds, extra)])
        return [model.decode_latent_to_image(r["latent"]).clip(-1, 1) for r in res]

    cc = a.concurrent_clips
    for n in sorted(set(sizes), reverse=True):  # capture the graph of every group size outside the timed region
        if n > 1:
            units(list(range(n)))
    sync()
    t0 = time.perf_counter()
    outs, done = [], 0
    for n in sizes:
        outs += [o.half() for o in units([a.warmup + done + j for j in range(n)])]
        done += n
    local_out = torch.cat(outs, 0)
    if world > 1:  # the single exchange of the path: collect every rank's edited frames (the gloo-tested helper of the drivers)
        from insv2v.clip_parallel import gather_frames
        gathered = gather_frames(local_out, world * local_out.shape[0], item_shape=tuple(local_out.shape[1:]))
        assert gathered.shape[0] == world * local_out.shape[0]
    sync()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    assert torch.isfinite(local_out.float()).all(), "non-finite output frames"
    # outside the timed region: the first timed clip of a stacked group against the same clip edited alone (one launch chain per clip,
    # other kernels at that launch shape) - the value-level check of what the timed region computed
    stacked_vs_single = None
    if rank == 0 and plain and not a.long_video and sizes and sizes[0] > 1 and a.clip_mode == "stacked":
        # the FIRST and the LAST clip of the first (largest) stack: the last one's rows lie beyond 2 GiB in the 960-wide operands of a
        # 20-clip stack, exactly where a 32-bit offset slip would show
        stacked_vs_single = 0.0
        for j in sorted({0, sizes[0] - 1}):
            alone = one_unit(a.warmup + j).half().float()
            got = outs[j].float()
            rel = ((got - alone).pow(2).mean().sqrt() / alone.pow(2).mean().sqrt()).item()
            assert rel <= 2e-2, f"stacked clip {j} differs from the single-clip run: rel-RMS {rel:.3e}"
            stacked_vs_single = max(stacked_vs_single, rel)

    result = None
    if rank == 0:
        total_frames = world * a.steps * F
        result = {
            "metric": "denoised video-frames/sec (16f@256x384, 50 DDIM steps)", "value": total_frames / elapsed,
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16", "data": "synthetic",
            "config": {"workload": f"{'C4 unit (32-frame clip = 3 windows 16/12/4 new frames, overlap stitching): ' if a.long_video else ''}{('C3 (C2 + optical-flow noise correction, R=4' + (', flows estimated by RAFT on the HIP kernels inside the timed region)' if a.raft else ', synthetic flows handed over)')) if a.flow_correction else ('C2' if (F, H, W) in ((16, 256, 384), (32, 256, 384)) else 'C5' if (F, H, W) == (24, 384, 512) else 'custom geometry')}: 1 clip/step = VAE-encode + {a.ddim_steps} DDIM steps (3-way CFG, text 7.5 / video 1.5) + VAE-decode, "
                                   f"{F} frames @ {H}x{W}, random-init {'TINY (invalid)' if a.tiny else 'full-width'} UNet+VAE"
                                   + (f"; {a.concurrent_clips} independent clips in flight per GPU ("
                                      + ("stacked into every UNet launch, B = 3 x clips" if a.clip_mode == "stacked" else "DDIM loops interleaved on one stream each, CFG branches batched")
                                      + ")" if a.concurrent_clips > 1 else ""),
                       "frames": F, "height": H, "width": W, "ddim_steps": a.ddim_steps, "clips_per_gpu": a.steps,
                       "parallelism": f"clip-parallel x{world}, one all_gather", "hip_graph": not a.no_graph, "cfg_branch_streams": not a.no_branch_streams, "concurrent_clips": a.concurrent_clips, "clip_mode": a.clip_mode if a.concurrent_clips > 1 else "single", "clip_groups": sizes,
                       "stage_breakdown_note": "one clip alone (last warm-up unit: latency mode with the CFG branches batched), not the stacked groups",
                       "stacked_vs_single_rel_rms": stacked_vs_single, "driver_mode": bool(a.driver_mode),
                       "single_clip_latency": ({"ms_per_clip": round(sum(breakdown.values()), 1), "frames_per_s": round(F / max(sum(breakdown.values()), 1e-9) * 1e3, 3)}
                                               if breakdown else None),
                       "stage_breakdown_ms": {k: round(v, 2) for k, v in breakdown.items()}},
        }
        if not a.tiny:
            result["config"]["stage_breakdown_ms"]["text_encode_2_prompts_ms (outside the metric)"] = round(text_encode_ms(dev), 2)
        nb = 3 * a.concurrent_clips if (a.concurrent_clips > 1 and a.clip_mode == "stacked") else 3
        result["roofline"] = roofline(model, pipe, 16 if a.long_video else F, h, w, text_cond, text_uncond, dev, a, nb)   # long video: the 16-frame windows it launches
        result["cpu_baseline"] = None
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ucfg, vcfg, usd, F, H, W, a.ddim_steps)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


def text_encode_ms(dev):
    """The per-unit text stage (2 prompts through the HIP CLIP ViT-L/14 text tower, random-init weights, synthetic token ids);
    reported beside the metric, not part of it (SURVEY.md 8d defines the unit as VAE-encode + DDIM loop + VAE-decode)."""
    from insv2v import synth, shapes
    from insv2v.clip_text import CLIPTextTransformer
    enc = CLIPTextTransformer(device=dev, **synth.CLIP_FULL).load_state_dict(synth.synth_state_dict(shapes.clip_text_shapes(**synth.CLIP_FULL)))
    ids = synth.synth_token_ids("bench.clip", 2, 77, synth.CLIP_FULL["vocab_size"])
    enc(ids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        enc(ids)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


def roofline(model, pipe, F, h, w, text_cond, text_uncond, dev, a, nb=3):
    """Per-launch HIP-event timing of every kernel family during one eager UNet forward of the bench workload, at the batch the
    timed region launches (nb = 3 CFG branches x the clips stacked into one launch chain).
    The top-level fields describe the dominant family (fp16 MFMA GEMM / implicit-GEMM conv: flops over summed launch
    durations against the dense MFMA peak); `families` carries the same for attention (MFMA) and the norm kernels (HBM)."""
    from insv2v import ops
    from insv2v.inference import shared_runner
    # (the stacked mode's runner: branch-major CFG triples whose common prefix the UNet computes once - the launches recorded here are the captured graph's)
    runner = shared_runner(pipe.unet, nb, F, h, w, text_cond.shape[1], 0, pipe.use_graph, False, cfg_clips=nb // 3) if nb != 3 else pipe._runner(3, F, h, w, text_cond.shape[1])
    if runner.kvs is None:   # a shape the timed region did not launch: give it a valid CFG stack, built exactly as the pipe builds one
        n = nb // 3
        if nb != 3:   # branch-major (inference._stacked_gen): (no text) x n, (no text) x n, (text) x n
            runner.set_context(torch.cat([text_uncond] * (2 * n) + [text_cond] * n, 0))
        else:         # one clip: branch order of InferenceIP2PVideo._step_inputs (uncond, uncond, cond)
            runner.set_context(torch.cat([text_uncond, text_uncond, text_cond], 0))
        lat = torch.randn(F, 4, h, w, device=dev)
        rows1 = F * h * w
        for c in range(n):
            ops.build_unet_input(lat, lat, runner.x_in[c * rows1:], runner.t[c:], 981, 3, branch_rows=n * rows1, t_stride=n)
    rec = []
    ops.set_launch_recorder(rec)
    try:
        with ops.workspace(runner._ws[0]):
            runner.unet.forward_cl(runner.x_in, runner.t, runner.kvs, text_cond.shape[1], nb, F, h, w, cfg_clips=runner.cfg_clips)
        torch.cuda.synchronize()
    finally:
        ops.set_launch_recorder(None)
    fam = {}
    for name, flops, e0, e1, *tag in rec:
        d = fam.setdefault(name, {"flops": 0.0, "s": 0.0, "n": 0, "bytes": 0.0})
        d["flops"] += flops
        d["s"] += e0.elapsed_time(e1) * 1e-3
        d["n"] += 1
        d["bytes"] += algorithmic_bytes(tag[0]) if tag else 0.0
    try:   # the family's dominant single launch shapes (reported beside the family figure; never allowed to break the bench line)
        tops = top_shapes([(name, flops, e0.elapsed_time(e1) * 1e-3, tag[0] if tag else None) for name, flops, e0, e1, *tag in rec])
    except Exception as exc:   # noqa: BLE001
        tops = [{"error": repr(exc)}]
    g = fam.get("gemm_kernel", {"flops": 0.0, "s": 1.0, "n": 0, "bytes": 0.0})
    total_t = sum(v["s"] for v in fam.values())
    ach = g["flops"] / g["s"] / 1e12
    traffic, traffic_src = None, None
    if os.path.exists(PMC_TRAFFIC_JSON) and not a.tiny:
        pmc = json.load(open(PMC_TRAFFIC_JSON))
        for e in pmc.get("shapes", [pmc]):   # one entry per measured batch (tools/pmc_forward_traffic.py)
            if e.get("shape") == [nb, F, h, w] and "gemm/conv" in e.get("families", {}):
                traffic = e["families"]["gemm/conv"]["bytes_per_forward"] / max(g["n"], 1)
                traffic_src = e.get("source")
    families = {}
    for name, v in sorted(fam.items()):
        e = {"launches": v["n"], "ms": round(1e3 * v["s"], 3), "algorithmic_gbytes": round(v["bytes"] / 1e9, 3)}
        if name in ("gemm_kernel", "attn_kernel"):
            e.update(bound="mfma", achieved=v["flops"] / v["s"] / 1e12, peak=PEAK_MFMA_F16_TFLOPS, unit="TFLOP/s",
                     frac=v["flops"] / v["s"] / 1e12 / PEAK_MFMA_F16_TFLOPS)
        else:
            e.update(bound="hbm", achieved=v["bytes"] / v["s"] / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s",
                     frac=v["bytes"] / v["s"] / 1e9 / PEAK_HBM_GBPS)
        families[name] = e
    # Winograd convolutions (round 6): `achieved` counts a convolution's ALGORITHMIC work (2 * pixels * Cout * 9 * Cin) whatever form
    # computes it; the grouped GEMM of the Winograd form launches 4 / 9 (upsample form: 1 / 4) of those multiply-adds.  Both figures are reported.
    wino = [(flops, tag[0]) for name, flops, e0, e1, *tag in rec if tag and tag[0][0] == "wino_gemm"]
    launched = g["flops"] - sum(f for f, _ in wino) + sum(2.0 * t[1] * t[2] * t[3] for _, t in wino)
    return {"bound": "mfma", "kernel": "fp16 MFMA GEMM family (v_mfma_f32_16x16x32_f16 engine gemm_q8 / gemm_r8: Linear, implicit-GEMM conv3x3 and the grouped "
                                       "transformed-tap GEMMs of the Winograd convolutions; 128x128 gemm_kernel / conv_halo; register-resident row kernels ffn_fused, "
                                       "rowlin, tattn*, xattn*)",
            "achieved": ach, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F16_TFLOPS,
            "traffic": traffic, "traffic_unit": "HBM-side bytes per launch", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": g["bytes"] / max(g["n"], 1),
            "algorithmic_gbytes_per_unet_forward": g["bytes"] / 1e9,
            "unet_batch": nb, "launches_per_unet_forward": g["n"], "operator_launches_per_unet_forward": len(rec), "avg_launch_us": 1e6 * g["s"] / max(g["n"], 1),
            "algorithmic_tflop_per_unet_forward": g["flops"] / 1e12, "launched_tflop_per_unet_forward": launched / 1e12,
            "launched_tflops": launched / g["s"] / 1e12, "launched_frac": launched / g["s"] / 1e12 / PEAK_MFMA_F16_TFLOPS, "share_of_unet_forward_time": g["s"] / max(total_t, 1e-9),
            "top_shapes": tops, "families": families}


def cpu_baseline(ucfg, vcfg, usd, F, H, W, ddim_steps):
    """fp32 CPU oracle ("port") on the host cores, as SURVEY.md 8d prescribes: TWO full DDIM steps of the 3-way-CFG
    pipeline at the bench shape (the oracle's own sampling loop: 3-branch UNet forward + CFG combine + scheduler step)
    + VAE encode/decode of one frame, extrapolated linearly to the unit (stated in `sample`)."""
    import oracle.unet3d as ou
    import oracle.vae as ov
    import oracle.pipelines as op
    from insv2v import synth
    # PyTorch CPU kernels on this path stop scaling (and regress) beyond ~16 threads on the 256-core host
    # (tools/cpu_threads_probe.py: 16 thr 8.2 s, 32 thr 9.5 s, 64 thr 13.9 s, 256 thr 317 s for one branch),
    # so the baseline uses the best setting and reports the threads actually used.
    cores = min(os.cpu_count(), 16)
    torch.set_num_threads(cores)
    h, w = H // 8, W // 8
    n_steps = 2
    with torch.no_grad():
        unet = ou.UNet3DConditionModel(**ucfg).eval()
        unet.load_state_dict(usd)
        pipe = op.InferenceIP2PVideo(unet, scheduler="ddim", num_ddim_steps=ddim_steps)
        lat = synth.synth_input("cpu.lat", (1, F, 4, h, w))
        cond = synth.synth_input("cpu.cond", (1, F, 4, h, w))
        tc = synth.synth_input("cpu.tc", (1, 77, ucfg["cross_attention_dim"]))
        tu = synth.synth_input("cpu.tu", (1, 77, ucfg["cross_attention_dim"]))
        t0 = time.perf_counter()
        pipe(lat, tc, tu, cond, text_cfg=7.5, img_cfg=1.5, start_time=ddim_steps - n_steps)
        t_steps = time.perf_counter() - t0
        del unet, pipe
        vae = ov.AutoencoderKL(**vcfg).eval()
        vae.load_state_dict(synth.synth_state_dict(vae))
        img = synth.synth_input("cpu.img", (1, 3, H, W), kind="uniform")
        t0 = time.perf_counter()
        z = vae.encode(img, torch.zeros(1, 4, h, w))
        vae.decode(z)
        t_vae = time.perf_counter() - t0
    unit = ddim_steps * t_steps / n_steps + F * t_vae
    return {"value": F / unit, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"fp32 torch-CPU oracle: {n_steps} full DDIM steps (3-branch UNet + CFG + scheduler; {F}f, {h}x{w} latents) = "
                      f"{t_steps:.1f}s, VAE enc+dec of 1 frame = {t_vae:.1f}s; unit time extrapolated as "
                      f"{ddim_steps}/{n_steps} * steps + {F} * vae = {unit:.0f}s"}


if __name__ == "__main__":
    main()
