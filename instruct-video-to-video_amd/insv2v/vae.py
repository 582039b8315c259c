"""KL-VAE encode/decode on the HIP kernels (drop-in for modules/kl_autoencoder/autoencoder.py:89-100
over modules/vqvae/model.py Encoder/Decoder).  Same ``ddconfig``/``embed_dim`` kwargs and state-dict
keys (encoder.*, decoder.*, quant_conv.*, post_quant_conv.*).

Channels-last fp16 throughout; frames are batched (the reference decodes one frame at a time,
instruct_p2p_video.py:73-76 -- same values, fewer launches).  The mid AttnBlock (model.py:145-197,
single head, C=512) is three batched GEMMs + a row softmax: S = q k^T / sqrt(C), P = softmax(S),
O = P v, with v^T produced directly by a GEMM so no transpose kernel is needed.
"""
import torch

from . import ops
from .unet import prep_conv3x3, prep_linear, prep_norm, _dev

EPS = 1e-6
GROUPS = 32


class VResBlock:
    def __init__(self, sd, key, cin, cout, dev):
        self.n1, self.n2 = prep_norm(sd, key + ".norm1", dev), prep_norm(sd, key + ".norm2", dev)
        self.c1, self.c2 = prep_conv3x3(sd, key + ".conv1", dev), prep_conv3x3(sd, key + ".conv2", dev)
        self.sc = prep_linear(sd, key + ".nin_shortcut", dev) if cin != cout else None

    def __call__(self, x, geom):
        N, H, W = geom
        n = ops.groupnorm(x, N, H * W, *self.n1, GROUPS, EPS, silu=True)
        h, _ = ops.conv3x3(n, geom, *self.c1)
        n = ops.groupnorm(h, N, H * W, *self.n2, GROUPS, EPS, silu=True)
        res = ops.gemm(x, *self.sc) if self.sc is not None else x
        out, _ = ops.conv3x3(n, geom, *self.c2, residual=res)
        return out


class VAttn:
    def __init__(self, sd, key, ch, dev):
        self.ch = ch
        self.norm = prep_norm(sd, key + ".norm", dev)
        self.wqk = _dev(torch.cat([sd[f"{key}.{n}.weight"].reshape(ch, ch).float() for n in "qk"], 0), torch.float16, dev)
        self.bqk = _dev(torch.cat([sd[f"{key}.{n}.bias"].float() for n in "qk"], 0), torch.float32, dev)
        self.wv, self.bv = prep_linear(sd, key + ".v", dev)
        self.proj = prep_linear(sd, key + ".proj_out", dev)

    def __call__(self, x, geom):
        N, H, W = geom
        C, HW = self.ch, H * W
        if HW % 8:
            raise NotImplementedError("VAE attention needs h*w to be a multiple of 8")
        n = ops.groupnorm(x, N, HW, *self.norm, GROUPS, EPS)
        qk = ops.gemm(n, self.wqk, self.bqk)  # [N*HW, 2C]
        s = torch.empty((N, HW, HW), device=x.device, dtype=torch.float16)
        # S_f = q_f k_f^T * C^-0.5 (alpha applied in the epilogue keeps fp16 in range)
        ops.gemm(qk, qk[:, C:], out=s, alpha=float(C) ** -0.5, batch=N, M=HW, N=HW, K=C, lda=2 * C, ldw=2 * C, ldc=HW,
                 a_bs=HW * 2 * C, w_bs=HW * 2 * C, c_bs=HW * HW)
        ops.softmax_rows(s)
        # v^T_f [C, HW] = Wv n_f^T ; the v bias is added after P.v (softmax rows sum to 1)
        vt = torch.empty((N, C, HW), device=x.device, dtype=torch.float16)
        ops.gemm(self.wv, n, out=vt, batch=N, M=C, N=HW, K=C, lda=C, ldw=C, ldc=HW, a_bs=0, w_bs=HW * C, c_bs=C * HW)
        o = torch.empty((N * HW, C), device=x.device, dtype=torch.float16)
        ops.gemm(s.reshape(N * HW, HW), vt.reshape(N * C, HW), self.bv, out=o, batch=N, M=HW, N=C, K=HW, lda=HW, ldw=HW,
                 ldc=C, a_bs=HW * HW, w_bs=C * HW, c_bs=HW * C)
        return ops.gemm(o, *self.proj, residual=x)


class AutoencoderKL:
    def __init__(self, ddconfig, embed_dim=4, device="cuda", **unused):
        self.dd = dict(ddconfig)
        self.embed_dim = embed_dim
        self.device = torch.device(device)
        if len(self.dd.get("attn_resolutions", [])):
            raise NotImplementedError("attn_resolutions is empty in the InsV2V config")
        self.loaded = False

    def load_state_dict(self, sd, strict=True):
        dd, dev = self.dd, self.device
        ch, mult, nres = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
        # encoder
        e = "encoder"
        self.e_in = prep_conv3x3(sd, e + ".conv_in", dev)
        self.e_in_pad = self.e_in[0].shape[1] // 9
        self.e_down = []
        cur = ch
        in_mult = [1] + mult
        for lvl in range(len(mult)):
            blocks = []
            cur = ch * in_mult[lvl]
            for j in range(nres):
                blocks.append(VResBlock(sd, f"{e}.down.{lvl}.block.{j}", cur, ch * mult[lvl], dev))
                cur = ch * mult[lvl]
            ds = prep_conv3x3(sd, f"{e}.down.{lvl}.downsample.conv", dev) if lvl != len(mult) - 1 else None
            self.e_down.append((blocks, ds))
        self.e_mid = (VResBlock(sd, e + ".mid.block_1", cur, cur, dev), VAttn(sd, e + ".mid.attn_1", cur, dev),
                      VResBlock(sd, e + ".mid.block_2", cur, cur, dev))
        self.e_norm = prep_norm(sd, e + ".norm_out", dev)
        self.e_out = prep_conv3x3(sd, e + ".conv_out", dev)
        self.quant = prep_linear(sd, "quant_conv", dev)
        # decoder
        d = "decoder"
        self.post_quant = prep_linear(sd, "post_quant_conv", dev)
        cur = ch * mult[-1]
        self.d_in = prep_conv3x3(sd, d + ".conv_in", dev)
        self.d_in_pad = self.d_in[0].shape[1] // 9
        self.d_mid = (VResBlock(sd, d + ".mid.block_1", cur, cur, dev), VAttn(sd, d + ".mid.attn_1", cur, dev),
                      VResBlock(sd, d + ".mid.block_2", cur, cur, dev))
        self.d_up = []
        for lvl in reversed(range(len(mult))):
            blocks = []
            for j in range(nres + 1):
                blocks.append(VResBlock(sd, f"{d}.up.{lvl}.block.{j}", cur, ch * mult[lvl], dev))
                cur = ch * mult[lvl]
            us = prep_conv3x3(sd, f"{d}.up.{lvl}.upsample.conv", dev) if lvl != 0 else None
            self.d_up.append((blocks, us))
        self.d_norm = prep_norm(sd, d + ".norm_out", dev)
        self.d_out = prep_conv3x3(sd, d + ".conv_out", dev)
        self.loaded = True
        return self

    # ---------------------------------------------------------------------------------------------
    def _frames_per_call(self, H, W):
        """Frames whose largest activation ([N*H*W, C] fp16 at image resolution) stays inside the 2 GiB descriptor
        window of the LDS-DMA operand loads (insv2v_gemm rejects larger operands): e.g. 21 frames at 384x512."""
        cmax = self.dd["ch"] * max(self.dd["ch_mult"][:2])  # widest feature map held at full / half resolution
        return max(1, int((2 ** 31 - 2 ** 24) // (H * W * cmax * 2)))

    @torch.no_grad()
    def moments(self, x):
        """x [N,3,H,W] float -> channels-last fp32 moments [N*h*w, 2*embed_dim], (N,h,w)."""
        N, C, H, W = x.shape
        step = self._frames_per_call(H, W)
        if N > step:  # frames are independent: chunk so every operand fits the addressing window
            parts = [self.moments(x[i:i + step]) for i in range(0, N, step)]
            return torch.cat([p[0] for p in parts], 0), (N, *parts[0][1][1:])
        t = ops.nchw_to_nhwc_f16(x.to(device=self.device, dtype=torch.float32), self.e_in_pad)
        geom = (N, H, W)
        t, geom = ops.conv3x3(t, geom, *self.e_in)
        for blocks, ds in self.e_down:
            for b in blocks:
                t = b(t, geom)
            if ds is not None:
                t, geom = ops.conv3x3(t, geom, *ds, stride=2, pad=(0, 0))  # F.pad (0,1,0,1) + stride 2 (model.py:67-71)
        r1, at, r2 = self.e_mid
        t = r2(at(r1(t, geom), geom), geom)
        n = ops.groupnorm(t, geom[0], geom[1] * geom[2], *self.e_norm, GROUPS, EPS, silu=True)
        t, _ = ops.conv3x3(n, geom, *self.e_out)
        return ops.gemm(t, *self.quant, out_fp32=True), geom

    @torch.no_grad()
    def encode(self, x, noise=None, scale=1.0):
        """Posterior SAMPLE (autoencoder.py:89-95); ``noise`` [N,4,h,w] defaults to a CPU randn like the
        reference (autoencoder.py:22).  Returns fp32 [N,4,h,w] times ``scale``."""
        mom, (N, h, w) = self.moments(x)
        if noise is None:
            noise = torch.randn((N, self.embed_dim, h, w))
        noise = noise.to(device=self.device, dtype=torch.float32)
        return ops.posterior_sample(mom, noise, N, h, w, scale)

    @torch.no_grad()
    def decode(self, z, scale=1.0):
        """z [N,4,h,w] float -> image [N,3,8h,8w] fp32 (autoencoder.py:97-100); z is multiplied by ``scale`` first."""
        N, C, h, w = z.shape
        step = self._frames_per_call(8 * h, 8 * w)
        if N > step:
            return torch.cat([self.decode(z[i:i + step], scale) for i in range(0, N, step)], 0)
        t = ops.nchw_to_nhwc_f16(z.to(device=self.device, dtype=torch.float32), 8, scale)  # 4 latent ch + zero pad
        wp, bp = self._post_quant_padded()
        t = ops.gemm(t, wp, bp)  # [N*h*w, d_in_pad]; channels >= z_channels are exactly zero
        geom = (N, h, w)
        t, geom = ops.conv3x3(t, geom, *self.d_in)
        r1, at, r2 = self.d_mid
        t = r2(at(r1(t, geom), geom), geom)
        for blocks, us in self.d_up:
            for b in blocks:
                t = b(t, geom)
            if us is not None:
                t, geom = ops.conv3x3(t, geom, *us, upsample=True)
        n = ops.groupnorm(t, geom[0], geom[1] * geom[2], *self.d_norm, GROUPS, EPS, silu=True)
        t, _ = ops.conv3x3(n, geom, *self.d_out, out_fp32=True)
        return ops.nhwc_to_nchw_f32(t, geom[0], self.dd["out_ch"], geom[1], geom[2])

    def _post_quant_padded(self):
        """post_quant_conv weight [z, embed] zero-padded to K=8 (GEMM K granularity) and to d_in_pad rows,
        so its output is directly the zero-padded channels-last input of decoder.conv_in."""
        if not hasattr(self, "_pq"):
            w, b = self.post_quant
            wp = torch.zeros((self.d_in_pad, 8), device=w.device, dtype=torch.float16)
            wp[:w.shape[0], :w.shape[1]] = w
            bp = torch.zeros((self.d_in_pad,), device=w.device, dtype=torch.float32)
            bp[:b.shape[0]] = b
            self._pq = (wp, bp)
        return self._pq

    def to(self, *a, **k):
        return self

    def eval(self):
        return self
