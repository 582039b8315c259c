"""diffusers-API-shaped Attention for the reference's VersatileAttention subclass."""
import torch
import torch.nn as nn
from oracle.leaves import FeedForward as _FF


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0,
                 bias=False, upcast_attention=False, **kw):
        super().__init__()
        inner = heads * dim_head
        kv = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads, self.scale = heads, dim_head ** -0.5
        self.group_norm, self.added_kv_proj_dim = None, None
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def head_to_batch_dim(self, t):
        b, s, c = t.shape
        return t.reshape(b, s, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, s, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, s, d).permute(0, 2, 1, 3).reshape(b, s, self.heads * d)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = self.head_to_batch_dim(self.to_q(hidden_states))
        k = self.head_to_batch_dim(self.to_k(ctx))
        v = self.head_to_batch_dim(self.to_v(ctx))
        w = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * self.scale, dim=-1)
        return self.to_out[1](self.to_out[0](self.batch_to_head_dim(torch.bmm(w, v))))


class FeedForward(_FF):
    def __init__(self, dim, dropout=0.0, activation_fn="geglu", **kw):
        assert activation_fn == "geglu"
        super().__init__(dim)


class AdaLayerNorm(nn.Module):  # imported by the reference, never used by the config
    pass
