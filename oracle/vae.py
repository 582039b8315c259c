"""fp32 CPU restatement of the KL-VAE encode/decode (test oracle).

Follows modules/vqvae/model.py:35-136 (Upsample/Downsample/ResnetBlock),
:145-197 (AttnBlock), :211-302 (Encoder), :305-411 (Decoder) and
modules/kl_autoencoder/autoencoder.py:10-23, 89-100 (posterior sample, encode,
decode).  State-dict key names match the reference (encoder.down.N.block.M...,
mid.block_1, mid.attn_1, nin_shortcut, ...).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _norm(ch):
    return nn.GroupNorm(32, ch, eps=1e-6)


def swish(x):
    return x * torch.sigmoid(x)


class VResBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1, self.conv1 = _norm(cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = _norm(cout), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)
        self.cin, self.cout = cin, cout

    def forward(self, x):
        h = self.conv1(swish(self.norm1(x)))
        h = self.conv2(swish(self.norm2(h)))
        return (self.nin_shortcut(x) if self.cin != self.cout else x) + h


class VAttn(nn.Module):
    """model.py:145-197 single-head attention over h*w with 1x1-conv q/k/v/proj_out."""

    def __init__(self, ch):
        super().__init__()
        self.norm = _norm(ch)
        self.q, self.k, self.v = nn.Conv2d(ch, ch, 1), nn.Conv2d(ch, ch, 1), nn.Conv2d(ch, ch, 1)
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        n = self.norm(x)
        q = self.q(n).reshape(b, c, h * w).transpose(1, 2)
        k = self.k(n).reshape(b, c, h * w)
        v = self.v(n).reshape(b, c, h * w)
        a = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
        o = torch.bmm(v, a.transpose(1, 2)).reshape(b, c, h, w)
        return x + self.proj_out(o)


class VDown(nn.Module):
    """model.py:56-74: pad (0,1,0,1) then 3x3 stride-2 pad-0 conv."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class VUp(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(x.dtype))


class _Level(nn.Module):
    pass


class Encoder(nn.Module):
    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4,
                 double_z=True, attn_resolutions=(), **unused):
        super().__init__()
        assert len(attn_resolutions) == 0
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        self.down = nn.ModuleList()
        mult = (1,) + tuple(ch_mult)
        cur = ch
        for lvl in range(len(ch_mult)):
            L = _Level()
            L.block = nn.ModuleList()
            L.attn = nn.ModuleList()
            cur = ch * mult[lvl]
            for _ in range(num_res_blocks):
                L.block.append(VResBlock(cur, ch * ch_mult[lvl]))
                cur = ch * ch_mult[lvl]
            if lvl != len(ch_mult) - 1:
                L.downsample = VDown(cur)
            self.down.append(L)
        self.mid = _Level()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = VResBlock(cur, cur), VAttn(cur), VResBlock(cur, cur)
        self.norm_out = _norm(cur)
        self.conv_out = nn.Conv2d(cur, 2 * z_channels if double_z else z_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for L in self.down:
            for blk in L.block:
                h = blk(h)
            if hasattr(L, "downsample"):
                h = L.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(swish(self.norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4,
                 attn_resolutions=(), **unused):
        super().__init__()
        assert len(attn_resolutions) == 0
        cur = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, cur, 3, padding=1)
        self.mid = _Level()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = VResBlock(cur, cur), VAttn(cur), VResBlock(cur, cur)
        ups = []
        for lvl in reversed(range(len(ch_mult))):
            L = _Level()
            L.block = nn.ModuleList()
            L.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                L.block.append(VResBlock(cur, ch * ch_mult[lvl]))
                cur = ch * ch_mult[lvl]
            if lvl != 0:
                L.upsample = VUp(cur)
            ups.insert(0, L)
        self.up = nn.ModuleList(ups)
        self.norm_out = _norm(cur)
        self.conv_out = nn.Conv2d(cur, out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for lvl in reversed(range(len(self.up))):
            L = self.up[lvl]
            for blk in L.block:
                h = blk(h)
            if hasattr(L, "upsample"):
                h = L.upsample(h)
        return self.conv_out(swish(self.norm_out(h)))


class AutoencoderKL(nn.Module):
    """autoencoder.py:50-100.  ``encode`` returns a SAMPLE of the posterior; the
    Gaussian noise is drawn on the CPU (autoencoder.py:22) so it can be injected."""

    def __init__(self, ddconfig, embed_dim=4, **unused):
        super().__init__()
        self.encoder, self.decoder = Encoder(**ddconfig), Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def encode(self, x, noise=None):
        mean, logvar = torch.chunk(self.moments(x), 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        if noise is None:
            noise = torch.randn(mean.shape)
        return mean + std * noise.to(mean.device)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
