"""fp32 CPU restatement of the InsV2V 3D UNet (test oracle, see oracle/__init__.py).

Follows the reference's wiring, with identical state-dict key names:
  modules/video_unet_temporal/unet.py:37-225 (ctor), :296-434 (forward)
  modules/video_unet_temporal/unet_blocks.py (block containers)
  modules/video_unet_temporal/resnet.py:10-18, 21-107, 110-204
  modules/video_unet_temporal/attention.py:33-138, 141-270
  modules/video_unet_temporal/motion_module.py:42-351
Tensors are kept in the reference's (b, c, f, h, w) layout throughout.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .leaves import Attention, FeedForward, TimestepEmbedding, timestep_sinusoid


def _frames_to_batch(x):
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), f


def _batch_to_frames(x, f):
    bf, c, h, w = x.shape
    return x.reshape(bf // f, f, c, h, w).permute(0, 2, 1, 3, 4)


class FrameConv(nn.Conv2d):
    """resnet.py:10-18 InflatedConv3d: a Conv2d applied to every frame."""

    def forward(self, x):
        y, f = _frames_to_batch(x)
        return _batch_to_frames(super().forward(y), f)


class ResBlock(nn.Module):
    """resnet.py:110-204 (time_embedding_norm='default', output_scale_factor=1).
    GroupNorm runs on the 5-D tensor, i.e. statistics span all frames."""

    def __init__(self, cin, cout, temb_ch, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = FrameConv(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = FrameConv(cout, cout, 3, padding=1)
        self.conv_shortcut = FrameConv(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Down(nn.Module):
    """resnet.py:76-107: 3x3 stride-2 pad-1 conv per frame."""

    def __init__(self, ch):
        super().__init__()
        self.conv = FrameConv(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Up(nn.Module):
    """resnet.py:21-73: nearest x(1,2,2) then 3x3 conv."""

    def __init__(self, ch):
        super().__init__()
        self.conv = FrameConv(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest"))


class SpatialBlock(nn.Module):
    """attention.py:141-270 BasicTransformerBlock (self-attn, text cross-attn, GEGLU FF)."""

    def __init__(self, dim, heads, dim_head, ctx_dim):
        super().__init__()
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, ctx):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), ctx) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    """attention.py:33-138 Transformer3DModel, use_linear_projection=False (1x1 convs),
    per-frame GroupNorm eps 1e-6, text context repeated per frame (:96)."""

    def __init__(self, heads, dim_head, ch, ctx_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, heads * dim_head, 1)
        self.transformer_blocks = nn.ModuleList([SpatialBlock(heads * dim_head, heads, dim_head, ctx_dim)])
        self.proj_out = nn.Conv2d(heads * dim_head, ch, 1)

    def forward(self, x, ctx):
        y, f = _frames_to_batch(x)
        ctx = ctx.repeat_interleave(f, dim=0)
        bf, c, h, w = y.shape
        t = self.proj_in(self.norm(y)).permute(0, 2, 3, 1).reshape(bf, h * w, -1)
        for blk in self.transformer_blocks:
            t = blk(t, ctx)
        t = t.reshape(bf, h, w, -1).permute(0, 3, 1, 2)
        return _batch_to_frames(self.proj_out(t) + y, f)


class PosEnc(nn.Module):
    """motion_module.py:220-242: interleaved sin/cos table, added before q/k/v."""

    def __init__(self, d_model, max_len):
        super().__init__()
        pos = torch.arange(max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(pos * div)
        pe[0, :, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe)

    def forward(self, x, start):
        if start + x.size(1) > self.pe.size(1):
            start = start - self.pe.size(1)
        if start < 0:
            raise ValueError(f"start_index must be non-negative, but got {start}")
        return x + self.pe[:, start:start + x.size(1)]


class TemporalAttention(Attention):
    """motion_module.py:245-336 VersatileAttention, 'Temporal' self-attention over frames."""

    def __init__(self, dim, heads, dim_head, max_len):
        super().__init__(dim, None, heads, dim_head)
        self.pos_encoder = PosEnc(dim, max_len)

    def forward(self, x, video_length, start):
        bf, d, c = x.shape
        b = bf // video_length
        t = x.reshape(b, video_length, d, c).permute(0, 2, 1, 3).reshape(b * d, video_length, c)
        t = self.pos_encoder(t, start)
        o = self.to_out[0](self.attend(self.to_q(t), self.to_k(t), self.to_v(t)))
        return o.reshape(b, d, video_length, c).permute(0, 2, 1, 3).reshape(bf, d, c)


class TemporalBlock(nn.Module):
    """motion_module.py:155-217 TemporalTransformerBlock."""

    def __init__(self, dim, heads, dim_head, n_attn, max_len):
        super().__init__()
        self.attention_blocks = nn.ModuleList([TemporalAttention(dim, heads, dim_head, max_len) for _ in range(n_attn)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(n_attn)])
        self.ff = FeedForward(dim)
        self.ff_norm = nn.LayerNorm(dim)

    def forward(self, x, video_length, start):
        for attn, norm in zip(self.attention_blocks, self.norms):
            x = attn(norm(x), video_length, start) + x
        return self.ff(self.ff_norm(x)) + x


class TemporalTransformer(nn.Module):
    """motion_module.py:79-152 TemporalTransformer3DModel (Linear proj_in/out)."""

    def __init__(self, ch, heads, dim_head, n_layers, n_attn, max_len, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, inner)
        self.transformer_blocks = nn.ModuleList(
            [TemporalBlock(inner, heads, dim_head, n_attn, max_len) for _ in range(n_layers)])
        self.proj_out = nn.Linear(inner, ch)

    def forward(self, x, start):
        y, f = _frames_to_batch(x)
        bf, c, h, w = y.shape
        t = self.proj_in(self.norm(y).permute(0, 2, 3, 1).reshape(bf, h * w, c))
        for blk in self.transformer_blocks:
            t = blk(t, f, start)
        t = self.proj_out(t).reshape(bf, h, w, c).permute(0, 3, 1, 2)
        return _batch_to_frames(t + y, f)


class MotionModule(nn.Module):
    """motion_module.py:42-76 VanillaTemporalModule (proj_out zero-initialised :68-69)."""

    def __init__(self, ch, groups, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self", "Temporal_Self"),
                 temporal_position_encoding=True, temporal_position_encoding_max_len=24,
                 temporal_attention_dim_div=1, zero_initialize=True, **unused):
        super().__init__()
        assert temporal_position_encoding and all(t == "Temporal_Self" for t in attention_block_types)
        self.temporal_transformer = TemporalTransformer(
            ch, num_attention_heads, ch // num_attention_heads // temporal_attention_dim_div,
            num_transformer_block, len(attention_block_types), temporal_position_encoding_max_len, groups)
        if zero_initialize:
            nn.init.zeros_(self.temporal_transformer.proj_out.weight)
            nn.init.zeros_(self.temporal_transformer.proj_out.bias)

    def forward(self, x, start=0):
        return self.temporal_transformer(x, start)


class DownBlock(nn.Module):
    """unet_blocks.py:239-364 CrossAttnDownBlock3D / :367-458 DownBlock3D."""

    def __init__(self, cin, cout, temb_ch, n_layers, groups, eps, heads, ctx_dim, cross, downsample, motion, mkw):
        super().__init__()
        self.resnets = nn.ModuleList([ResBlock(cin if i == 0 else cout, cout, temb_ch, groups, eps) for i in range(n_layers)])
        if cross:
            self.attentions = nn.ModuleList([SpatialTransformer(heads, cout // heads, cout, ctx_dim, groups) for _ in range(n_layers)])
        self.cross = cross
        self.motion_modules = nn.ModuleList([MotionModule(cout, groups, **mkw) if motion else None for _ in range(n_layers)])
        self.downsamplers = nn.ModuleList([Down(cout)]) if downsample else None

    def forward(self, x, temb, ctx, start):
        outs = []
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.cross:
                x = self.attentions[i](x, ctx)
            if self.motion_modules[i] is not None:
                x = self.motion_modules[i](x, start)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    """unet_blocks.py:142-236 UNetMidBlock3DCrossAttn."""

    def __init__(self, ch, temb_ch, groups, eps, heads, ctx_dim, motion, mkw):
        super().__init__()
        self.resnets = nn.ModuleList([ResBlock(ch, ch, temb_ch, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([SpatialTransformer(heads, ch // heads, ch, ctx_dim, groups)])
        self.motion_modules = nn.ModuleList([MotionModule(ch, groups, **mkw) if motion else None])

    def forward(self, x, temb, ctx, start):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        if self.motion_modules[0] is not None:
            x = self.motion_modules[0](x, start)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    """unet_blocks.py:461-591 CrossAttnUpBlock3D / :594-680 UpBlock3D."""

    def __init__(self, cin, cout, prev, temb_ch, n_layers, groups, eps, heads, ctx_dim, cross, upsample, motion, mkw):
        super().__init__()
        res = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResBlock(rin + skip, cout, temb_ch, groups, eps))
        self.resnets = nn.ModuleList(res)
        if cross:
            self.attentions = nn.ModuleList([SpatialTransformer(heads, cout // heads, cout, ctx_dim, groups) for _ in range(n_layers)])
        self.cross = cross
        self.motion_modules = nn.ModuleList([MotionModule(cout, groups, **mkw) if motion else None for _ in range(n_layers)])
        self.upsamplers = nn.ModuleList([Up(cout)]) if upsample else None

    def forward(self, x, skips, temb, ctx, start):
        for i, res in enumerate(self.resnets):
            x = res(torch.cat([x, skips.pop()], dim=1), temb)
            if self.cross:
                x = self.attentions[i](x, ctx)
            if self.motion_modules[i] is not None:
                x = self.motion_modules[i](x, start)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet3DConditionModel(nn.Module):
    """unet.py:37-434.  Accepts the reference YAML's ``unet.params`` as kwargs."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 down_block_types=("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",),
                 up_block_types=("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3,
                 layers_per_block=2, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 attention_head_dim=8, flip_sin_to_cos=True, freq_shift=0,
                 use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
                 motion_module_mid_block=True, motion_module_decoder_only=False,
                 motion_module_type="Vanilla", motion_module_kwargs=None, **unused):
        super().__init__()
        norm_eps = float(norm_eps)
        mkw = dict(motion_module_kwargs or {})
        ch = list(block_out_channels)
        temb_ch = ch[0] * 4
        heads = attention_head_dim  # used as the head COUNT (unet_blocks.py:293-294)
        self.flip, self.shift, self.ch0 = flip_sin_to_cos, freq_shift, ch[0]
        self.conv_in = FrameConv(in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb_ch)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, typ in enumerate(down_block_types):
            cin, out = out, ch[i]
            mot = use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only
            self.down_blocks.append(DownBlock(cin, out, temb_ch, layers_per_block, norm_num_groups, norm_eps, heads,
                                              cross_attention_dim, typ.startswith("CrossAttn"), i != len(ch) - 1, mot, mkw))
        self.mid_block = MidBlock(ch[-1], temb_ch, norm_num_groups, norm_eps, heads, cross_attention_dim,
                                  use_motion_module and motion_module_mid_block, mkw)
        self.up_blocks = nn.ModuleList()
        rev = ch[::-1]
        out = rev[0]
        for i, typ in enumerate(up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            mot = use_motion_module and (2 ** (3 - i) in motion_module_resolutions)
            self.up_blocks.append(UpBlock(cin, out, prev, temb_ch, layers_per_block + 1, norm_num_groups, norm_eps, heads,
                                          cross_attention_dim, typ.startswith("CrossAttn"), i != len(ch) - 1, mot, mkw))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch[0], eps=norm_eps)
        self.conv_out = FrameConv(ch[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, video_start_index=0):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.int64)
        t = timestep.reshape(-1).expand(sample.shape[0])
        temb = self.time_embedding(timestep_sinusoid(t, self.ch0, self.flip, self.shift).to(sample.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states, video_start_index)
            skips += outs
        x = self.mid_block(x, temb, encoder_hidden_states, video_start_index)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states, video_start_index)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return UNetOutput(x)
