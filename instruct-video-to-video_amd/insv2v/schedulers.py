"""DDIM / DDPM schedules as the reference configures them (pl_trainer/inference/inference.py:26-51;
diffusers 0.21.4 DDIMScheduler(set_alpha_to_one=False, steps_offset=1, clip_sample=False) and
DDPMScheduler(clip_sample=False), 'leading' spacing, scaled_linear betas).

Only the per-step SCALAR coefficients are computed on the host (fp32, same operation order as
diffusers); the tensor update runs in insv2v_cfg_step:
    x0   = (x_t - sqrt(1-a_t) eps) / sqrt(a_t)
    prev = c_x0 * x0 + c_eps * eps + c_xt * x_t + c_noise * noise
"""
import numpy as np
import torch


class _Schedule:
    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000, **unused):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(beta_schedule)
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.num_inference_steps = None
        self.timesteps = None

    def _leading(self, n, offset):
        ratio = self.num_train_timesteps // n
        return torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + offset)


class DDIMScheduler(_Schedule):
    stochastic = False

    def __init__(self, set_alpha_to_one=False, steps_offset=1, clip_sample=False, **kw):
        super().__init__(**kw)
        if clip_sample:
            raise NotImplementedError("clip_sample")
        self.steps_offset = steps_offset
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, n):
        self.num_inference_steps = n
        self.timesteps = self._leading(n, self.steps_offset)

    def coefficients(self, t):
        """-> dict(sqrt_a, sqrt_1ma, coef=(c_x0, c_eps, c_xt, c_noise)) as python floats."""
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return dict(sqrt_a=float(a_t ** 0.5), sqrt_1ma=float((1 - a_t) ** 0.5),
                    coef=(float(a_prev ** 0.5), float((1 - a_prev) ** 0.5), 0.0, 0.0))


class DDPMScheduler(_Schedule):
    stochastic = True

    def __init__(self, clip_sample=False, **kw):
        super().__init__(**kw)
        if clip_sample:
            raise NotImplementedError("clip_sample")

    def set_timesteps(self, n):
        self.num_inference_steps = n
        self.timesteps = self._leading(n, 0)

    def coefficients(self, t):
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else torch.tensor(1.0)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c_noise = 0.0
        if t > 0:
            c_noise = float(torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5)
        return dict(sqrt_a=float(a_t ** 0.5), sqrt_1ma=float(b_t ** 0.5),
                    coef=(float((a_prev ** 0.5 * cur_b) / b_t), 0.0, float(cur_a ** 0.5 * b_prev / b_t), c_noise))
