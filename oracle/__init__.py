"""CPU oracle for the InsV2V denoising hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch fp32 CPU
restatement of the reference algorithm (amazon-science/instruct-video-to-video,
files cited per function as ``file:line`` relative to the reference root).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / reported
baseline.  The product path (``instruct-video-to-video_amd/insv2v``) never
imports it and has no CPU fallback.

Parity pinning status
---------------------
* VAE Encoder/Decoder (``modules/vqvae/model.py``), ``warp_image`` /
  ``resize_flow`` (``misc_utils/flow_utils.py``), ``split_batch`` and the
  sampling loops of ``pl_trainer/inference/inference.py`` are pinned against the
  reference itself, imported in the build container by
  ``tools/gen_golden.py`` (golden vectors under ``tests/golden/``).
* The UNet block wiring (``modules/video_unet_temporal/*.py``) is pinned the
  same way, but the leaf arithmetic it calls lives in the third-party
  ``diffusers`` package (pinned 0.21.4 in the reference's THIRD-PARTY file),
  which is absent from the container and has no golden vectors in the
  reference.  Those leaves (Attention, FeedForward/GEGLU, Timesteps,
  TimestepEmbedding, DDIM/DDPM scheduler step) are restated from the published
  0.21.4 algorithm in ``oracle/leaves.py`` / ``oracle/schedulers.py``:
  **parity unpinned for the diffusers leaves** (the reference ships no tests).
* The CLIP text tower behind ``FrozenCLIPEmbedder`` (``modules/openclip/modules.py``)
  is third-party ``transformers.CLIPTextModel``; unlike diffusers it IS installed
  (5.15.0), so ``oracle/clip_text.py`` is pinned against the real implementation
  (goldens ``tests/golden/clip_text_*.npz`` + a live comparison in the CPU tests).
"""
