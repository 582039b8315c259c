"""Model facade with the call surface the reference drivers use on ``diffusion_model``
(insv2v_run_loveu_tgve.py:58-62,98-99,118,165):

  encode_image_to_latent   pl_trainer/instruct_p2p_video.py:57-64 -> pl_trainer/diffusion.py:242-244
  decode_latent_to_image   pl_trainer/instruct_p2p_video.py:66-79 -> pl_trainer/diffusion.py:246-249
  encode_text              pl_trainer/diffusion.py:286-290 -> FrozenCLIPEmbedder.encode (insv2v/clip_text.py, HIP)
  load_state_dict          flat checkpoint with ``unet.`` / ``vae.`` / ``text_model.`` prefixes
  create_model             misc_utils/train_utils.py:74-80 (unit_test_create_model) + model_utils.py:6-17
"""
import yaml
import torch

from .unet import UNet3DConditionModel
from .vae import AutoencoderKL


class InstructP2PVideoModel:
    def __init__(self, unet, vae, text_model=None, scale_factor=0.18215, **unused):
        self.unet, self.vae, self.text_model, self.scale_factor = unet, vae, text_model, scale_factor

    @torch.no_grad()
    def encode_image_to_latent(self, image, noise=None):
        """image [b,f,3,H,W] in [-1,1] -> latent [b,f,4,H/8,W/8] (posterior sample x scale_factor)."""
        b, f = image.shape[:2]
        z = self.vae.encode(image.reshape(b * f, *image.shape[2:]),
                            None if noise is None else noise.reshape(b * f, *noise.shape[2:]), scale=self.scale_factor)
        return z.reshape(b, f, *z.shape[1:])

    @torch.no_grad()
    def decode_latent_to_image(self, latent):
        """latent [b,f,4,h,w] -> image [b,f,3,8h,8w]."""
        b, f = latent.shape[:2]
        img = self.vae.decode(latent.reshape(b * f, *latent.shape[2:]), scale=1.0 / self.scale_factor)
        return img.reshape(b, f, *img.shape[1:])

    @torch.no_grad()
    def encode_text(self, x):
        if self.text_model is None:
            raise RuntimeError("no text model attached: build the model from a config with a text_model block, pass "
                               "text_model=, or feed [n,77,768] embeddings directly")
        if isinstance(x, tuple):
            x = list(x)
        return self.text_model.encode(x)

    def load_state_dict(self, ckpt, strict=False):
        ckpt = {k.replace("_forward_module.", ""): v for k, v in ckpt.items()}
        for name, mod in (("unet", self.unet), ("vae", self.vae)):
            sub = {k[len(name) + 1:]: v for k, v in ckpt.items() if k.startswith(name + ".")}
            if sub:
                mod.load_state_dict(sub)
            elif strict:
                raise KeyError(f"no '{name}.*' keys in checkpoint")
        if self.text_model is not None and hasattr(self.text_model, "load_state_dict"):
            sub = {k[len("text_model."):]: v for k, v in ckpt.items() if k.startswith("text_model.")}
            if sub:
                self.text_model.load_state_dict(sub, strict=False)
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self


def load_config(path):
    """YAML with {target, params} blocks (configs/instruct_v2v_inference.yaml).  Plain PyYAML parses
    ``norm_eps: 1e-05`` as a string (OmegaConf does not), so numeric strings are coerced."""
    conf = yaml.safe_load(open(path))

    def fix(o):
        if isinstance(o, dict):
            return {k: fix(v) for k, v in o.items()}
        if isinstance(o, list):
            return [fix(v) for v in o]
        if isinstance(o, str):
            try:
                return float(o)
            except ValueError:
                return o
        return o
    return fix(conf)


def create_model(config, device="cuda", text_model=None, tokenizer=None):
    """``config`` is a YAML path or an already-loaded dict with ``unet.params`` / ``vae.params`` (and optionally the
    ``text_model`` block of configs/instruct_v2v_inference.yaml:90-93, built as the HIP FrozenCLIPEmbedder; ``tokenizer`` or
    a local ``version`` directory supplies the CLIP BPE vocabulary, which cannot be downloaded offline)."""
    conf = load_config(config) if isinstance(config, str) else config
    unet = UNet3DConditionModel(**conf["unet"]["params"], device=device)
    vae = AutoencoderKL(**conf["vae"]["params"], device=device)
    if text_model is None and "text_model" in conf:
        from .clip_text import FrozenCLIPEmbedder
        tp = dict(conf["text_model"].get("params") or {})
        tp.pop("device", None)
        text_model = FrozenCLIPEmbedder(device=device, tokenizer=tokenizer, **tp)
    dp = conf.get("diffusion", {}).get("params", {})
    return InstructP2PVideoModel(unet, vae, text_model, scale_factor=dp.get("scale_factor", 0.18215))


unit_test_create_model = create_model  # reference name (misc_utils/train_utils.py:74)
