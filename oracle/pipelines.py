"""CPU restatement of the sampling loops (test oracle).

pl_trainer/inference/inference.py:13-24 (rescale_noise_cfg), :26-51 (ctor),
:159-218 (InferenceIP2PVideo.__call__), :220-289 (second_clip_forward),
:291-398 (optical-flow variant; RAFT is out of scope, so flows are INJECTED:
``flows[q]`` is the [R,2,H,W] image-resolution flow query_q -> each ref frame
that the reference would obtain from RAFT at :306-309),
insv2v_run_loveu_tgve.py:12-29 (split_batch), :119-165 (long-video driver),
pl_trainer/instruct_p2p_video.py:57-79 + pl_trainer/diffusion.py:242-249 (VAE wrappers).
"""
import torch

from .schedulers import DDIMScheduler, DDPMScheduler
from .flow import warp_image, resize_flow


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def split_batch(cond, frames_in_batch=16, num_ref_frames=4):
    chunks = [cond[:, :frames_in_batch]]
    ptr, refs = frames_in_batch, []
    total = cond.shape[1]
    while ptr < total:
        remaining = total - ptr
        new = remaining if remaining < frames_in_batch else frames_in_batch - num_ref_frames
        chunks.append(cond[:, ptr:ptr + new])
        refs.append(frames_in_batch - new)
        ptr += new
    return chunks, refs


class InferenceIP2PVideo:
    def __init__(self, unet, scheduler="ddim", beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", num_ddim_steps=20, guidance_scale=5):
        self.unet = unet
        cls = {"ddim": DDIMScheduler, "ddpm": DDPMScheduler}.get(scheduler)
        if cls is None:
            raise NotImplementedError()
        self.scheduler = cls(beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule)
        self.scheduler.set_timesteps(num_ddim_steps)
        self.num_ddim_steps = num_ddim_steps
        self.guidance_scale = guidance_scale
        self.variance_noises = None  # optional injected DDPM noises (list, one per step)

    def _eps(self, latent, t, text_cond, text_uncond, img_cond, text_cfg, img_cfg, guidance_rescale):
        l1 = torch.cat([latent, torch.zeros_like(img_cond)], dim=2)
        l2 = torch.cat([latent, img_cond], dim=2)
        x = torch.cat([l1, l2, l2.clone()], dim=0).permute(0, 2, 1, 3, 4)
        ctx = torch.cat([text_uncond, text_uncond, text_cond], dim=0)
        n = self.unet(x, torch.full((3,), t, dtype=torch.long), encoder_hidden_states=ctx).sample
        n1, n2, n3 = n.permute(0, 2, 1, 3, 4).chunk(3, dim=0)
        noise = n1 + img_cfg * (n2 - n1) + text_cfg * (n3 - n2)
        if guidance_rescale > 0:
            noise = rescale_noise_cfg(noise, n1, guidance_rescale)
        return noise

    def _step(self, i, noise, t, latent):
        if isinstance(self.scheduler, DDPMScheduler) and self.variance_noises is not None:
            return self.scheduler.step(noise, t, latent, variance_noise=self.variance_noises[i])
        return self.scheduler.step(noise, t, latent)

    @torch.no_grad()
    def __call__(self, latent, text_cond, text_uncond, img_cond, text_cfg=7.5, img_cfg=1.2,
                 start_time=0, guidance_rescale=0.0):
        all_latent, all_pred = [], []
        for i, t in enumerate(self.scheduler.timesteps[start_time:]):
            t = int(t)
            noise = self._eps(latent, t, text_cond, text_uncond, img_cond, text_cfg, img_cfg, guidance_rescale)
            out = self._step(i, noise, t, latent)
            latent = out.prev_sample
            all_latent.append(latent)
            all_pred.append(out.pred_original_sample)
        return {"latent": latent, "all_latent": all_latent, "all_pred": all_pred}

    def _correct(self, noise, latent, latent_ref, t, R):
        a = self.scheduler.alphas_cumprod[t]
        noise_ref = (latent[:, :R] - (a ** 0.5) * latent_ref) / ((1 - a) ** 0.5)
        delta = noise_ref - noise[:, :R]
        return delta

    @torch.no_grad()
    def second_clip_forward(self, latent, text_cond, text_uncond, img_cond, latent_ref,
                            noise_correct_step=1.0, text_cfg=7.5, img_cfg=1.2, start_time=0,
                            guidance_rescale=0.0):
        R = latent_ref.shape[1]
        all_latent, all_pred = [], []
        for i, t in enumerate(self.scheduler.timesteps[start_time:]):
            t = int(t)
            noise = self._eps(latent, t, text_cond, text_uncond, img_cond, text_cfg, img_cfg, guidance_rescale)
            if noise_correct_step * self.num_ddim_steps > i:
                delta = self._correct(noise, latent, latent_ref, t, R)
                noise = torch.cat([noise[:, :R] + delta, noise[:, R:] + delta.mean(dim=1, keepdim=True)], dim=1)
            out = self._step(i, noise, t, latent)
            latent = out.prev_sample
            all_latent.append(latent)
            all_pred.append(out.pred_original_sample)
        return {"latent": latent, "all_latent": all_latent, "all_pred": all_pred}


class InferenceIP2PVideoOpticalFlow(InferenceIP2PVideo):
    """inference.py:291-398 with the RAFT estimator replaced by injected flows."""

    @torch.no_grad()
    def second_clip_forward(self, latent, text_cond, text_uncond, img_cond, latent_ref, flows,
                            noise_correct_step=1.0, text_cfg=7.5, img_cfg=1.2, start_time=0,
                            guidance_rescale=0.0):
        assert latent.shape[0] == 1, "only support batch size 1"
        R = latent_ref.shape[1]
        all_latent, all_pred = [], []
        for i, t in enumerate(self.scheduler.timesteps[start_time:]):
            t = int(t)
            noise = self._eps(latent, t, text_cond, text_uncond, img_cond, text_cfg, img_cfg, guidance_rescale)
            if noise_correct_step * self.num_ddim_steps > i:
                delta = self._correct(noise, latent, latent_ref, t, R)
                noise = noise.clone()
                noise[:, :R] = noise[:, :R] + delta
                for q, flow in zip(range(R, noise.shape[1]), flows):
                    f = resize_flow(flow, delta.shape[3:])
                    warped = warp_image(delta[0], f)
                    mask = warp_image(torch.ones_like(delta[0])[:, :1], f)
                    msum = mask[None].sum(dim=1, keepdim=True)
                    upd = torch.where(msum > 0.5, warped[None].sum(dim=1, keepdim=True) / msum, torch.zeros(()))
                    noise[:, q:q + 1] += torch.where(msum > 0.5, upd, torch.zeros(()))
            out = self._step(i, noise, t, latent)
            latent = out.prev_sample
            all_latent.append(latent)
            all_pred.append(out.pred_original_sample)
        return {"latent": latent, "all_latent": all_latent, "all_pred": all_pred}


SCALE_FACTOR = 0.18215


def encode_image_to_latent(vae, frames, noise=None):
    """instruct_p2p_video.py:57-64 -> diffusion.py:242-244 (frames [b,f,3,H,W] in [-1,1])."""
    b, f = frames.shape[:2]
    z = vae.encode(frames.reshape(b * f, *frames.shape[2:]), noise) * SCALE_FACTOR
    return z.reshape(b, f, *z.shape[1:])


def decode_latent_to_image(vae, latent):
    """instruct_p2p_video.py:66-79 -> diffusion.py:246-249 (one frame at a time)."""
    b, f = latent.shape[:2]
    flat = latent.reshape(b * f, *latent.shape[2:])
    imgs = [vae.decode(z[None] / SCALE_FACTOR) for z in flat]
    img = torch.cat(imgs, dim=0)
    return img.reshape(b, f, *img.shape[1:])


@torch.no_grad()
def edit_video(pipe, vae, frames, text_cond, text_uncond, text_cfg, video_cfg, init_noises,
               enc_noise=None, frames_in_batch=16, num_ref_frames=4, flows_per_window=None):
    """insv2v_run_loveu_tgve.py:98, :119-165 for one (video, prompt) unit.
    ``init_noises[k]`` is the randn_like draw for window k (new frames only)."""
    cond = encode_image_to_latent(vae, frames, enc_noise) / SCALE_FACTOR
    conds, refs = split_batch(cond, frames_in_batch, num_ref_frames)
    init = init_noises[0]
    pred = pipe(latent=init, text_cond=text_cond, text_uncond=text_uncond, img_cond=conds[0],
                text_cfg=text_cfg, img_cfg=video_cfg)["latent"]
    preds = [pred]
    for k, (prev_cond, cond_k, R) in enumerate(zip(conds[:-1], conds[1:], refs)):
        init = torch.cat([init[:, -R:], init_noises[k + 1]], dim=1)
        cond_k = torch.cat([prev_cond[:, -R:], cond_k], dim=1)
        kw = {}
        if flows_per_window is not None:
            kw["flows"] = flows_per_window[k]
        pred = pipe.second_clip_forward(latent=init, text_cond=text_cond, text_uncond=text_uncond,
                                        img_cond=cond_k, latent_ref=pred[:, -R:], noise_correct_step=0.5,
                                        text_cfg=text_cfg, img_cfg=video_cfg, **kw)["latent"]
        preds.append(pred[:, R:])
    latent = torch.cat(preds, dim=1)
    return decode_latent_to_image(vae, latent).clip(-1, 1), latent
