"""Restatement of the diffusers 0.21.4 leaf modules the reference UNet calls.

diffusers is a third-party dependency (reference THIRD-PARTY:31 pins 0.21.4),
not vendored under /root/reference and not installed here: PARITY UNPINNED for
this file (no reference test / golden vector exists at this boundary).
Call sites in the reference: modules/video_unet_temporal/attention.py:160-190,
motion_module.py:200,245-331, unet.py:95-98,358-364.

State-dict key names follow diffusers so a real ``insv2v.pth`` loads.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention (default processor).

    to_q/to_k/to_v have no bias, to_out[0] has bias, heads are contiguous
    channel slices, out = softmax(q k^T * dim_head**-0.5) v.
    """

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False):
        super().__init__()
        inner = heads * dim_head
        kv_dim = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])

    def split_heads(self, x):
        b, s, _ = x.shape
        return x.reshape(b, s, self.heads, self.dim_head).permute(0, 2, 1, 3)

    def attend(self, q, k, v):
        q, k, v = self.split_heads(q), self.split_heads(k), self.split_heads(v)
        w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        o = torch.matmul(w, v)
        b, h, s, d = o.shape
        return o.permute(0, 2, 1, 3).reshape(b, s, h * d)

    def forward(self, hidden_states, encoder_hidden_states=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        o = self.attend(self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx))
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)  # erf GELU


class FeedForward(nn.Module):
    """diffusers FeedForward(dim, activation_fn='geglu', mult=4)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


def timestep_sinusoid(timesteps, dim, flip_sin_to_cos=True, shift=0.0, max_period=10000):
    """diffusers.models.embeddings.get_timestep_embedding (scale=1)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))
