import torch.nn as nn
from oracle.leaves import TimestepEmbedding, timestep_sinusoid  # noqa: F401


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.n, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, t):
        return timestep_sinusoid(t, self.n, self.flip, self.shift)
