"""Long-video editing driver: counterpart of insv2v_run_loveu_tgve.py.

  split_batch   insv2v_run_loveu_tgve.py:12-29  (window plan: 16-frame windows, 4-frame overlap)
  edit_video    insv2v_run_loveu_tgve.py:98, :119-165 for one (video, prompt) unit
  main          same CLI flags as :31-45; units are sharded clip-parallel over ranks (one process per GPU)

``main`` has three sources of work: ``--synthetic`` clips (random-init weights), a ``--units`` .pt file holding
{"frames": [n,T,3,H,W], "text_cond": [n,77,768], "text_uncond": [1,77,768]}, or - as the reference - the LOVEU-TGVE
dataset under ``--data-dir`` (``insv2v/video_io.py``: CSV + frames, prompts from ``--edit-prompt-file``, CLIP text tower on
the HIP kernels with the BPE vocabulary from ``--tokenizer-dir``), writing the reference's GIF / JPG result tree.
"""
import argparse
import os
from itertools import product

import torch


def split_batch(cond, frames_in_batch=16, num_ref_frames=4):
    """Split [b, T, ...] along frames: first window = frames_in_batch frames, later windows add
    (frames_in_batch - num_ref_frames) NEW frames each (the last one whatever remains).  Returns the
    new-frame chunks and, per later window, how many frames of the previous window it re-uses."""
    total = cond.shape[1]
    chunks = [cond[:, :frames_in_batch]]
    refs = []
    ptr = frames_in_batch
    while ptr < total:
        left = total - ptr
        new = left if left < frames_in_batch else frames_in_batch - num_ref_frames
        chunks.append(cond[:, ptr:ptr + new])
        refs.append(frames_in_batch - new)
        ptr += new
    return chunks, refs


@torch.no_grad()
def edit_video(model, inf_pipe, frames, text_cond, text_uncond, text_cfg=7.5, video_cfg=1.8, frames_in_batch=16,
               num_ref_frames=4, init_noises=None, enc_noise=None, flows_per_window=None, return_latent=False, cond=None):
    """frames [1,T,3,H,W] in [-1,1] -> edited frames [1,T,3,H,W] clipped to [-1,1].

    ``init_noises[k]`` / ``enc_noise`` optionally inject the random draws the reference takes from the
    global RNG (randn_like at :125,:139; the VAE posterior noise) so runs are reproducible.
    ``flows_per_window[k]`` (list over query frames of [R,2,H,W] flows) feeds the optical-flow variant precomputed
    flows; without it an optical-flow pipe gets the frames of the previous / current window (``ref_images`` /
    ``query_images``, insv2v_run_loveu_tgve.py:141-160) and runs its injected ``flow_estimator``.
    ``cond`` = an already encoded conditioning latent: the reference encodes a video ONCE and shares the posterior
    sample across its four prompts (:98)."""
    dev = model.unet.device
    if cond is None:
        cond = model.encode_image_to_latent(frames, enc_noise) / model.scale_factor
    conds, refs = split_batch(cond, frames_in_batch, num_ref_frames)
    frame_chunks, _ = split_batch(frames, frames_in_batch, num_ref_frames)
    wants_flow = hasattr(inf_pipe, "obtain_flow_batched")
    if wants_flow and flows_per_window is None and getattr(inf_pipe, "flow_estimator", None) is None and len(conds) > 1:
        raise RuntimeError("optical-flow pipeline without a flow source: pass flows_per_window= or build the pipe with flow_estimator=")

    def draw(k, like):
        if init_noises is not None:
            return init_noises[k].to(device=dev, dtype=torch.float32)
        return torch.randn(like.shape, device=dev, dtype=torch.float32)

    init = draw(0, conds[0])
    pred = inf_pipe(latent=init, text_cond=text_cond, text_uncond=text_uncond, img_cond=conds[0],
                    text_cfg=text_cfg, img_cfg=video_cfg)["latent"]
    preds = [pred]
    for k, (prev_cond, cond_k, R) in enumerate(zip(conds[:-1], conds[1:], refs)):
        init = torch.cat([init[:, -R:], draw(k + 1, cond_k)], dim=1)  # overlap re-uses the INITIAL noise (:139)
        cond_k = torch.cat([prev_cond[:, -R:], cond_k], dim=1)
        kw = {}
        if flows_per_window is not None:
            kw["flows"] = flows_per_window[k]
        elif wants_flow:
            prev_frames = torch.cat(frame_chunks[:k + 1], dim=1)
            kw["ref_images"], kw["query_images"] = prev_frames[:, -R:], frame_chunks[k + 1]
        pred = inf_pipe.second_clip_forward(latent=init, text_cond=text_cond, text_uncond=text_uncond, img_cond=cond_k,
                                            latent_ref=pred[:, -R:], noise_correct_step=0.5, text_cfg=text_cfg,
                                            img_cfg=video_cfg, **kw)["latent"]
        preds.append(pred[:, R:])
    latent = torch.cat(preds, dim=1)
    image = model.decode_latent_to_image(latent).clip(-1, 1)
    return (image, latent) if return_latent else image


@torch.no_grad()
def edit_videos(model, inf_pipe, units, frames_in_batch=16, num_ref_frames=4, return_latent=False):
    """Several independent units at once - the throughput form of ``edit_video`` (the reference's unit loop offers them naturally: four
    prompts per video, insv2v_run_loveu_tgve.py:83,101): window k of ALL units runs as one stacked launch chain
    (``InferenceIP2PVideo.run_stacked``: B = 3 x units in every UNet launch, weights read once, every launch fills the chip), windows
    stay sequential inside a unit (latent_ref / initial-noise carry, :139-161).  ``units``: list of dicts with ``frames`` [1,T,3,H,W],
    ``text_cond``, ``text_uncond`` and optionally ``text_cfg`` (7.5), ``video_cfg`` (1.8), ``init_noises``, ``enc_noise``, ``cond`` - the
    arguments of ``edit_video`` (optical-flow pipes also ``flows_per_window``; without it the pipe's ``flow_estimator`` sees the frames
    of the previous / current window, :141-160).  All units must share T, H, W.  Returns the list of edited frames (and latents).  A
    single unit takes ``edit_video`` (one clip per launch chain, three branch streams).  The flow-warped correction is per-unit
    elementwise work behind the shared UNet launch, so optical-flow units stack like the others (round 5)."""
    if len(units) == 0:
        return []
    wants_flow = hasattr(inf_pipe, "obtain_flow_batched")
    if len(units) == 1:
        keys = ("text_cfg", "video_cfg", "init_noises", "enc_noise", "cond", "flows_per_window")
        return [edit_video(model, inf_pipe, u["frames"], u["text_cond"], u["text_uncond"], frames_in_batch=frames_in_batch,
                           num_ref_frames=num_ref_frames, return_latent=return_latent, **{k: u[k] for k in keys if k in u}) for u in units]
    dev = model.unet.device
    shape = tuple(units[0]["frames"].shape)
    st = []
    for u in units:
        if tuple(u["frames"].shape) != shape:
            raise ValueError("edit_videos: all units must share [1,T,3,H,W]")
        cond = u.get("cond")
        if cond is None:
            cond = model.encode_image_to_latent(u["frames"], u.get("enc_noise")) / model.scale_factor
        conds, refs = split_batch(cond, frames_in_batch, num_ref_frames)
        if wants_flow and u.get("flows_per_window") is None and getattr(inf_pipe, "flow_estimator", None) is None and len(conds) > 1:
            raise RuntimeError("optical-flow pipeline without a flow source: pass flows_per_window= or build the pipe with flow_estimator=")
        st.append(dict(u=u, conds=conds, refs=refs, preds=[], init=None, pred=None,
                       frame_chunks=split_batch(u["frames"], frames_in_batch, num_ref_frames)[0] if wants_flow else None))

    def draw(s, k, like):
        noises = s["u"].get("init_noises")
        if noises is not None:
            return noises[k].to(device=dev, dtype=torch.float32)
        return torch.randn(like.shape, device=dev, dtype=torch.float32)

    def common(s):
        u = s["u"]
        return dict(text_cond=u["text_cond"], text_uncond=u["text_uncond"], text_cfg=u.get("text_cfg", 7.5), img_cfg=u.get("video_cfg", 1.8))

    calls = []
    for s in st:
        s["init"] = draw(s, 0, s["conds"][0])
        calls.append(dict(common(s), latent=s["init"], img_cond=s["conds"][0]))
    for s, r in zip(st, inf_pipe.run_stacked(calls)):
        s["pred"] = r["latent"]
        s["preds"].append(s["pred"])
    for k, R in enumerate(st[0]["refs"]):
        calls = []
        for s in st:
            s["init"] = torch.cat([s["init"][:, -R:], draw(s, k + 1, s["conds"][k + 1])], dim=1)  # overlap re-uses the INITIAL noise (:139)
            cond_k = torch.cat([s["conds"][k][:, -R:], s["conds"][k + 1]], dim=1)
            call = dict(common(s), latent=s["init"], img_cond=cond_k, latent_ref=s["pred"][:, -R:], noise_correct_step=0.5)
            if s["u"].get("flows_per_window") is not None:
                call["flows"] = s["u"]["flows_per_window"][k]
            elif wants_flow:   # ref_images = the last R frames before this window, query_images = its new frames (:141-147)
                prev_frames = torch.cat(s["frame_chunks"][:k + 1], dim=1)
                call["ref_images"], call["query_images"] = prev_frames[:, -R:], s["frame_chunks"][k + 1]
            calls.append(call)
        for s, r in zip(st, inf_pipe.run_stacked(calls)):
            s["pred"] = r["latent"]
            s["preds"].append(s["pred"][:, R:])
    outs = []
    for s in st:
        latent = torch.cat(s["preds"], dim=1)
        image = model.decode_latent_to_image(latent).clip(-1, 1)
        outs.append((image, latent) if return_latent else image)
    return outs


def build_parser():
    p = argparse.ArgumentParser(description="InsV2V LOVEU-TGVE editing on MI355X")
    p.add_argument("--text-cfg", nargs="+", type=float, default=[7.5], help="Text configuration parameter")
    p.add_argument("--video-cfg", nargs="+", type=float, default=[1.8], help="Image configuration parameter")
    p.add_argument("--num-frames", nargs="+", type=int, default=[32], help="Number of frames")
    p.add_argument("--image-size", nargs="+", type=int, default=[384], help="Image size")
    p.add_argument("--prompt-source", type=str, default="edit", help="Prompt source")
    p.add_argument("--ckpt-path", type=str, help="Path to checkpoint")
    p.add_argument("--config-path", type=str, default="configs/instruct_v2v.yaml", help="Path to config file")
    p.add_argument("--data-dir", type=str, default="loveu-tgve-2023", help="Path to LOVEU dataset")
    p.add_argument("--with_optical_flow", action="store_true", help="Use motion compensation")
    # additions of this build
    p.add_argument("--edit-prompt-file", type=str, default="dataset/loveu_tgve_edit_prompt_dict.json")
    p.add_argument("--tokenizer-dir", type=str, default=None, help="directory with the CLIP vocab.json / merges.txt")
    p.add_argument("--units", type=str, default=None, help=".pt file with pre-decoded frames and text embeddings")
    p.add_argument("--synthetic", type=int, default=0, help="run on N synthetic clips with random-init weights")
    p.add_argument("--out", type=str, default="v2v_results/edited.pt")
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--scheduler", type=str, default="ddpm")
    p.add_argument("--no-stack", action="store_true",
                   help="edit one unit at a time (one clip per UNet launch chain) instead of stacking a rank's units / a video's four prompts")
    p.add_argument("--raft-ckpt", type=str, default=None,
                   help="torchvision raft_large checkpoint (state dict) for --with_optical_flow: the estimator runs on the HIP kernels")
    p.add_argument("--flows", type=str, default=None,
                   help=".pt file with precomputed optical flows for --with_optical_flow: flows[unit][window][query] = [R,2,H,W]")
    return p


def check_args(args):
    """Fail at parse time, not after the first window has been sampled.  The optical-flow variant needs a flow source: the RAFT
    estimator runs on the HIP kernels (insv2v/raft.py) but its pretrained weights are not bundled (the reference downloads them,
    flow_utils.py:157): ``--raft-ckpt`` names torchvision's raft_large checkpoint, ``--synthetic`` runs use key-hashed weights, and
    ``--flows`` supplies precomputed flows instead."""
    if args.with_optical_flow and not (args.flows or args.raft_ckpt or args.synthetic):
        raise SystemExit("--with_optical_flow needs --raft-ckpt FILE (torchvision raft_large weights) or --flows FILE (precomputed flows)")
    if args.flows and args.units is None and not args.synthetic:
        raise SystemExit("--flows is supported with --units / --synthetic (flows are indexed by unit)")
    return args


def optical_flow_pipe_kwargs(args):
    """Constructor arguments of InferenceIP2PVideoOpticalFlow for the CLI's flow source (none when --flows supplies them)."""
    if not args.with_optical_flow or args.flows:
        return {}
    if args.raft_ckpt:
        return {"raft_state_dict": torch.load(args.raft_ckpt, map_location="cpu")}
    from . import synth, shapes
    return {"raft_state_dict": synth.synth_raft_state_dict(shapes.raft_shapes())}


def run_dataset(args, model, pipe, rank=0, world=1):
    """The reference's main loop (insv2v_run_loveu_tgve.py:83-170): every (video, cfg, size) x 4 prompt kinds is one
    independent unit; units are dealt round-robin to ranks, each rank writes its own result files."""
    import json
    from .video_io import LoveuTgveVideoDataset, save_tensor_to_gif, save_tensor_to_images, output_paths
    if model.text_model is None or model.text_model.tokenizer is None:
        raise SystemExit("dataset mode needs the CLIP tokenizer: pass --tokenizer-dir DIR (vocab.json + merges.txt)")
    prompts = json.load(open(args.edit_prompt_file, "r"))
    combos = list(product(range(len(prompts)), args.text_cfg, args.video_cfg, args.num_frames, args.image_size))
    for ci, (video_id, text_cfg, video_cfg, num_frames, image_size) in enumerate(combos):
        if ci % world != rank:
            continue
        batch = LoveuTgveVideoDataset(root_dir=args.data_dir, image_size=(image_size, image_size))[video_id]
        n = len(batch["frames"])
        skip = n // num_frames if n > num_frames else 1
        frames = batch["frames"][::skip].to(model.unet.device)[None]
        text_uncond = model.encode_text([""])
        cond = model.encode_image_to_latent(frames) / model.scale_factor  # once per video, shared by the four prompts (:98)
        todo = []
        for key in ("style", "object", "background", "multiple"):
            prompt = prompts[batch["video_name"]]["edit_" + key] if args.prompt_source == "edit" else batch[key]
            gif_path, image_dir = output_paths(args.prompt_source, image_size, video_id, video_cfg, text_cfg, num_frames,
                                               batch["video_name"], key, batch[key])
            if os.path.exists(gif_path):
                print(f"File {gif_path} exists, skip")
                continue
            todo.append((gif_path, image_dir, dict(frames=frames, text_cond=model.encode_text([prompt]), text_uncond=text_uncond,
                                                   text_cfg=text_cfg, video_cfg=video_cfg, cond=cond)))
        # the four prompts of a video share the conditioning latent and the window plan: one stacked launch chain per window (B = 12)
        if getattr(args, "no_stack", False):
            edits = [edit_videos(model, pipe, [u])[0] for _, _, u in todo]
        else:
            edits = edit_videos(model, pipe, [u for _, _, u in todo])
        for (gif_path, image_dir, _), edited in zip(todo, edits):
            save_tensor_to_gif(torch.cat([frames.float().cpu(), edited.float().cpu()], dim=4), gif_path, fps=5)
            save_tensor_to_images(edited.float().cpu(), image_dir)


def main(argv=None):
    import torch.distributed as dist
    from . import synth, shapes
    from .model import create_model
    from .inference import InferenceIP2PVideo, InferenceIP2PVideoOpticalFlow
    from .clip_parallel import shard_units, gather_frames

    args = check_args(build_parser().parse_args(argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")
    if args.synthetic:
        conf = {"unet": {"params": synth.UNET_FULL}, "vae": {"params": synth.VAE_FULL}}
        model = create_model(conf, device=f"cuda:{local}")
        model.unet.load_state_dict(synth.synth_state_dict(shapes.unet_shapes(**synth.UNET_FULL)))
        model.vae.load_state_dict(synth.synth_state_dict(shapes.vae_shapes(**synth.VAE_FULL)))
    else:
        tok = None
        if args.tokenizer_dir:
            from transformers import CLIPTokenizer
            tok = CLIPTokenizer.from_pretrained(args.tokenizer_dir, local_files_only=True)
        model = create_model(args.config_path, device=f"cuda:{local}", tokenizer=tok)
        ckpt = torch.load(args.ckpt_path, map_location="cpu")
        model.load_state_dict(ckpt, strict=False)
    if args.synthetic:
        g = torch.Generator().manual_seed(0)
        T, S = args.num_frames[0], args.image_size[0]
        data = {"frames": torch.rand((args.synthetic, T, 3, S, S), generator=g) * 2 - 1,
                "text_cond": torch.randn((args.synthetic, 77, 768), generator=g),
                "text_uncond": torch.randn((1, 77, 768), generator=g)}
    elif args.units is None:
        cls = InferenceIP2PVideoOpticalFlow if args.with_optical_flow else InferenceIP2PVideo
        run_dataset(args, model, cls(unet=model.unet, num_ddim_steps=args.steps, scheduler=args.scheduler, **optical_flow_pipe_kwargs(args)), rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    else:
        data = torch.load(args.units, map_location="cpu")
    cls = InferenceIP2PVideoOpticalFlow if args.with_optical_flow else InferenceIP2PVideo
    pipe = cls(unet=model.unet, num_ddim_steps=args.steps, scheduler=args.scheduler, **optical_flow_pipe_kwargs(args))
    n = data["frames"].shape[0]
    flows = torch.load(args.flows, map_location="cpu") if args.flows else None
    item_shape = tuple(data["frames"].shape[1:])  # a rank without units still takes part in the all_gather
    outs = []
    for text_cfg, video_cfg in product(args.text_cfg, args.video_cfg):
        mine = shard_units(n, rank, world)
        if args.no_stack:
            local_out = [edit_video(model, pipe, data["frames"][i:i + 1], data["text_cond"][i:i + 1], data["text_uncond"],
                                    text_cfg, video_cfg, flows_per_window=flows[i] if flows is not None else None) for i in mine]
        else:   # a rank's units as stacked launch chains (run_stacked caps the stack at what the kernels' operand window allows)
            from .inference import max_clips_in_flight
            T, S = data["frames"].shape[1], data["frames"].shape[-1]
            cap = max_clips_in_flight(min(T, 16), data["frames"].shape[-2] // 8, S // 8)
            local_out = []
            for g in range(0, len(mine), cap):
                local_out += edit_videos(model, pipe, [dict(frames=data["frames"][i:i + 1], text_cond=data["text_cond"][i:i + 1],
                                                            text_uncond=data["text_uncond"], text_cfg=text_cfg, video_cfg=video_cfg,
                                                            **({"flows_per_window": flows[i]} if flows is not None else {}))
                                                       for i in mine[g:g + cap]])
        local_out = torch.cat(local_out, 0).half() if local_out else torch.zeros((0, *item_shape), device=model.unet.device).half()
        outs.append(gather_frames(local_out, n, item_shape=item_shape))
    if rank == 0:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        torch.save(torch.stack(outs, 0).cpu(), args.out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
