"""DDIM / DDPM scheduler restatement (diffusers 0.21.4; PARITY UNPINNED, see
oracle/__init__.py).  Configured exactly as the reference does in
pl_trainer/inference/inference.py:26-51."""
import numpy as np
import torch


class _Base:
    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 num_train_timesteps=1000):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(beta_schedule)
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_inference_steps = None
        self.timesteps = None


class StepOutput:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMScheduler(_Base):
    """set_alpha_to_one=False, steps_offset=1, clip_sample=False, eta=0, 'leading' spacing."""

    def __init__(self, set_alpha_to_one=False, steps_offset=1, clip_sample=False, **kw):
        super().__init__(**kw)
        assert not clip_sample
        self.steps_offset = steps_offset
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def coefficients(self, t):
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output, t, sample):
        a_t, a_prev = self.coefficients(int(t))
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * model_output
        return StepOutput(prev, x0)


class DDPMScheduler(_Base):
    """clip_sample=False, variance_type='fixed_small', 'leading' spacing, steps_offset=0."""

    def __init__(self, clip_sample=False, **kw):
        super().__init__(**kw)
        assert not clip_sample

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, t, sample, variance_noise=None):
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        out = (a_prev ** 0.5 * cur_b) / b_t * x0 + cur_a ** 0.5 * b_prev / b_t * sample
        if t > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, dtype=model_output.dtype)
            var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
            out = out + var ** 0.5 * variance_noise
        return StepOutput(out, x0)
