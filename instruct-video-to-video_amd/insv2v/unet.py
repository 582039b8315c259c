"""UNet3DConditionModel on the HIP kernels (drop-in for modules/video_unet_temporal/unet.py).

Same constructor kwargs (the ``unet.params`` block of configs/instruct_v2v_inference.yaml), same
state-dict key names, same call signature
``unet(sample[b,c,f,h,w], timestep[b], encoder_hidden_states=[b,77,768]).sample`` as the
reference (unet.py:296-434).  Internally everything is channels-last fp16: a video tensor is the
token matrix [b*f*h*w, c], so InflatedConv3d (resnet.py:10-18), the (b f) c h w <-> b c f h w
rearranges and the NCHW <-> token permutes of attention.py:95-134 / motion_module.py:131-150
cost nothing, and torch.cat of skip connections (unet_blocks.py:561,659) is replaced by
two-source kernels.

Host code only sequences kernel launches; all arithmetic is in libinsv2v_hip.so.
"""
import math
import os

import torch

from . import ops
from . import fused
from .fused import pack_ffn_stream, pack_linear_stream, pack_tattn_stream, pack_tattn_qkv_stream

CPAD = 64  # implicit-GEMM K slices are 64 channels wide: conv inputs are zero-padded to this
# GroupNorm+SiLU applied inside the patch-tiled conv (conv3x3(gn_ab=...)): parity-green, but measured SLOWER end to end on
# MI355X (UNet step 33.15 vs 32.41 ms with 3 streams, 33.77 vs 32.90 ms batched, profiles/r02_groupnorm_fusion_ab.txt): the
# in-LDS normalise + SiLU pass lengthens every tap of the convolution by more than the removed apply pass cost.  Off by default.
FUSE_GN = os.environ.get("INSV2V_FUSE_GN", "0") != "0"
# Register-resident fused feed-forward at C = 320 (csrc/fused_rows.hip); INSV2V_FUSE_FFN=0 restores the two-GEMM path for A/B runs.
FUSE_FFN = os.environ.get("INSV2V_FUSE_FFN", "1") != "0"
FUSE_FFN_POST = os.environ.get("INSV2V_FUSE_FFN_POST", "1") != "0"   # + the trailing proj_out Linear and its residual in the same launch
# Round 5: in a stack of CFG triples the branches (no text, video) and (text, video) share everything in front of the first text
# cross-attention; forward_cl(cfg_clips=n) computes it once for the two (INSV2V_DEDUP_CFG=0: every sample on its own, for A/B runs).
DEDUP_CFG = os.environ.get("INSV2V_DEDUP_CFG", "1") != "0"
# Round 6: Winograd F(2x2, 3x3) form of the ResnetBlock3D convolutions with >= WINOGRAD_MIN_CIN input channels on even latents of at most
# 512 pixels (UNet levels 1-3): 2.25 x fewer MACs on an engine that runs at the board's power cap; 0.60 - 0.78 of the direct convolution's
# time at the B = 60 stack for Cin >= 1280, 0.96 at Cin = 640, a loss at level 0 (profiles/r06_winograd_proto.txt).  INSV2V_WINOGRAD=0: direct.
WINOGRAD = os.environ.get("INSV2V_WINOGRAD", "1") != "0"
WINOGRAD_MIN_CIN = int(os.environ.get("INSV2V_WINOGRAD_MIN_CIN", "1280"))
WINOGRAD_MIN_ROWS = int(os.environ.get("INSV2V_WINOGRAD_MIN_ROWS", "4608"))
WINOGRAD_UP_MIN_CIN = int(os.environ.get("INSV2V_WINOGRAD_UP_MIN_CIN", "640"))   # the upsampler convolutions (4 x fewer MACs): all three levels
# Round 6: where the feed-forward is two GEMMs (C >= 640) its second projection and the module's trailing proj_out + residual
# (attention.py:89,134 / motion_module.py:146-150) are ONE two-source GEMM: proj_out(x + W2 g + b2) + x0 = [Wp W2 | Wp] [g | x] + (Wp b2 + bp) + x0
# - the same FLOPs, one launch and one [M, C] round trip fewer per transformer module.  INSV2V_MERGE_FF2_POST=0: separate launches.
MERGE_FF2_POST = os.environ.get("INSV2V_MERGE_FF2_POST", "1") != "0"
# conv_out (320 -> 4 channels) of a stack as ops.conv3x3_narrow (one GEMM over the input channels for the nine taps' partial outputs + a shifted
# sum): 1.6 - 2 x faster than the implicit-GEMM form on its 64-wide output tile from one clip's three branches up (profiles/r06_conv_out_taps.txt);
# the reduced-width test models stay on the implicit-GEMM form
NARROW_CONV_OUT = os.environ.get("INSV2V_NARROW_CONV_OUT", "1") != "0"
NARROW_CONV_MIN_ROWS = int(os.environ.get("INSV2V_NARROW_CONV_MIN_ROWS", "16384"))
# Register-resident Linear kernel for the K = 320 layers (insv2v_rowlin); INSV2V_ROWLIN=0 restores insv2v_gemm for A/B runs.
ROWLIN = os.environ.get("INSV2V_ROWLIN", "1") != "0"
ROWLIN_640 = os.environ.get("INSV2V_ROWLIN_640", "1") != "0"   # the K = 640 (level 1) form separately, for A/B runs
# Round 5 (profiles/r05_fusion_switch_ab.txt): below ~9 stacked CFG triples the C = 640 row kernels (row Linear, fused temporal / cross
# attention) lose to the GEMM engine + generic attention (-1.0 % of a forward at B = 6, -0.06 % at B = 3; +1.3 % at B = 12, +2.0 % at
# B = 60), so the level-1 blocks pick per call by their token count (B = 6: 36 864 rows, B = 12: 73 728)
# (one branch per stream - the single-clip latency mode, 6 144 rows - stays on the row kernels: the engine needs M >= 8 192)
ROWLIN_640_MIN_ROWS = int(os.environ.get("INSV2V_ROWLIN_640_MIN_ROWS", "55296"))


def _rowlin_640_pays(rows):
    return rows >= ROWLIN_640_MIN_ROWS or rows < 12288
# Temporal attention sub-block (LayerNorm -> q/k/v -> attention over 16 frames -> to_out -> + residual) as one register-resident launch
# at C = 320 (insv2v_tattn_fused); INSV2V_FUSE_TATTN=0 restores row-linear + attention + row-linear for A/B runs.
FUSE_TATTN = os.environ.get("INSV2V_FUSE_TATTN", "1") != "0"
# C = 640: LayerNorm -> q/k/v -> attention as one launch (insv2v_tattn_attn), to_out + residual as a row Linear; =0 for A/B runs
FUSE_TATTN_640 = os.environ.get("INSV2V_FUSE_TATTN_640", "1") != "0"
# GroupNorm of the transformer blocks applied inside the proj_in row kernel (statistics pass only, no normalised copy)
ROWLIN_GN = os.environ.get("INSV2V_ROWLIN_GN", "1") != "0"
# text cross-attention sub-block (LayerNorm -> q -> attention over the text tokens -> out-proj + residual) as ONE launch at C = 320
# (insv2v_xattn_fused; the text K / V become a per-sample fragment stream); INSV2V_FUSE_XATTN=0 restores the three launches for A/B runs.
FUSE_XATTN = os.environ.get("INSV2V_FUSE_XATTN", "1") != "0"
FUSE_XATTN_640 = os.environ.get("INSV2V_FUSE_XATTN_640", "1") != "0"
FUSE_XATTN_PRE = os.environ.get("INSV2V_FUSE_XATTN_PRE", "1") != "0"   # C = 320: + the self-attention's output projection in the same launch   # the C = 640 form (no output projection), for A/B runs
# temporal blocks the row kernels do not reach (C = 1280): the per-frame positional-encoding bias of q/k/v is added by the attention kernel
# as it loads the rows, so the projection in front carries no row bias and may run on the persistent 256x256 GEMM (INSV2V_ATTN_PE_BIAS=0: in the GEMM epilogue)
ATTN_PE_BIAS = os.environ.get("INSV2V_ATTN_PE_BIAS", "1") != "0"


def rowlin_stream(w, bias, device, table=None):
    """fp16 fragment stream of a Linear for insv2v_rowlin, or None where that kernel does not apply (K != 320, N % 64)."""
    w = w.reshape(w.shape[0], -1)
    if not (ROWLIN and ops.rowlin_supported(w.shape[0], w.shape[1])) or (w.shape[1] == 640 and not ROWLIN_640):
        return None
    return _dev(pack_linear_stream(w.detach().half().float(), bias, table), torch.float16, device)


class Act:
    """Channels-last activation: t is [B*F*H*W, C] fp16."""
    __slots__ = ("t", "B", "F", "H", "W")

    def __init__(self, t, B, F, H, W):
        self.t, self.B, self.F, self.H, self.W = t, B, F, H, W

    @property
    def hw(self):
        return self.H * self.W

    def like(self, t, H=None, W=None):
        return Act(t, self.B, self.F, self.H if H is None else H, self.W if W is None else W)


# ----------------------------------------------------------------------------- weight preparation
def _dev(t, dtype, device):
    return t.detach().to(device=device, dtype=dtype).contiguous()


def prep_linear(sd, key, device, bias=True):
    w = _dev(sd[key + ".weight"].reshape(sd[key + ".weight"].shape[0], -1), torch.float16, device)
    b = _dev(sd[key + ".bias"], torch.float32, device) if bias and (key + ".bias") in sd else None
    return w, b


def prep_conv3x3(sd, key, device, cin_pad=None):
    w = sd[key + ".weight"].detach().float()  # [Cout, Cin, 3, 3]
    cout, cin = w.shape[:2]
    cp = cin_pad or ((cin + CPAD - 1) // CPAD * CPAD)
    wk = torch.zeros(cout, 3, 3, cp)
    wk[..., :cin] = w.permute(0, 2, 3, 1)
    return _dev(wk.reshape(cout, 9 * cp), torch.float16, device), _dev(sd[key + ".bias"], torch.float32, device)


def prep_norm(sd, key, device):
    return _dev(sd[key + ".weight"], torch.float32, device), _dev(sd[key + ".bias"], torch.float32, device)


def interleave32(t):
    """[h; g] (each n rows) -> [h0..31, g0..31, h32..63, g32..63, ...] for the fused GEGLU epilogue."""
    n = t.shape[0] // 2
    h, g = t[:n], t[n:]
    rest = t.shape[1:]
    return torch.stack([h.reshape(n // 32, 32, *rest), g.reshape(n // 32, 32, *rest)], dim=1).reshape(2 * n, *rest)


def fold_layernorm(w, gamma, beta, bias=None):
    """Fold a preceding LayerNorm into a Linear: LayerNorm(x) @ w.T + bias ==
    rstd * (x @ wf.T - mean * col) + b  with wf = w * gamma (rounded to fp16 FIRST, so that col = sum_k wf
    cancels the mean component exactly as the MFMA sees it), b = w @ beta + bias.  Consumed by the
    folded-LayerNorm epilogue of insv2v_gemm together with per-token (mean, rstd)."""
    w = w.detach().float()
    wf = (w * gamma.detach().float()[None, :]).half()
    col = wf.float().sum(1)
    b = w @ beta.detach().float()
    if bias is not None:
        b = b + bias.detach().float()
    return wf, col, b


class FeedForwardW:
    """diffusers FeedForward(geglu) with its preceding LayerNorm folded into the first projection."""

    def __init__(self, sd, key, device, norm_key, post_key=None):
        wf, col, b = fold_layernorm(sd[key + ".net.0.proj.weight"], sd[norm_key + ".weight"], sd[norm_key + ".bias"],
                                    sd[key + ".net.0.proj.bias"])
        self.w1 = _dev(interleave32(wf), torch.float16, device)
        self.cs1 = _dev(interleave32(col), torch.float32, device)
        self.b1 = _dev(interleave32(b), torch.float32, device)
        self.w2, self.b2 = prep_linear(sd, key + ".net.2", device)
        # C = 320 (UNet level 0): LayerNorm + both projections + GEGLU + residual as ONE register-resident kernel
        # (csrc/fused_rows.hip); its weights are a second, fragment-ordered copy (2.6 MB per layer)
        self.hidden, self.stream, self.stream_post = self.w2.shape[1], None, None
        self.w2p = self.b2p = None   # [Wp W2 | Wp] and Wp b2 + bp: the second projection merged with the module's proj_out (MERGE_FF2_POST)
        if FUSE_FFN and ops.ffn_fused_supported(self.w2.shape[0], self.hidden):
            w2f, b2f = sd[key + ".net.2.weight"].half().float(), sd[key + ".net.2.bias"]
            self.stream = _dev(pack_ffn_stream(wf.float(), b, w2f, b2f), torch.float16, device)
            if post_key is not None and FUSE_FFN_POST:   # + the module's trailing Linear (proj_out): insv2v_ffn_fused(post=1)
                self.stream_post = _dev(pack_ffn_stream(wf.float(), b, w2f, b2f, post=(sd[post_key + ".weight"].half().float(), sd[post_key + ".bias"])),
                                        torch.float16, device)

        if self.stream is None and post_key is not None and MERGE_FF2_POST and self.hidden % 64 == 0:   # (k_split in whole 64-wide K tiles)
            wp = sd[post_key + ".weight"].detach().float().reshape(sd[post_key + ".weight"].shape[0], -1)
            w2f, b2f = sd[key + ".net.2.weight"].detach().float(), sd[key + ".net.2.bias"].detach().float()
            self.w2p = _dev(torch.cat([wp @ w2f, wp], 1), torch.float16, device)
            self.b2p = _dev(wp @ b2f + sd[post_key + ".bias"].detach().float(), torch.float32, device)

    def with_proj_out_gemm(self, x, stats, module_input):
        """proj_out(x + FF(LN(x))) + module_input with the two-GEMM feed-forward: GEGLU projection, then ONE two-source GEMM (w2p)."""
        g = ops.gemm(x, self.w1, self.b1, act=ops.ACT_GEGLU, row_stats=stats if stats is not None else ops.layernorm_stats(x), col_sum=self.cs1)
        return ops.gemm(g, self.w2p, self.b2p, a2=x, residual=module_input)

    def with_proj_out(self, x, module_input):
        """proj_out(x + FF(LN(x))) + module_input in ONE launch (only where stream_post exists)."""
        return ops.ffn_fused(x, self.stream_post, self.hidden, post_residual=module_input)

    def __call__(self, x, residual, stats=None):
        """x: the un-normalised tokens (the LayerNorm runs inside the GEMM epilogue); stats: their row statistics if the producer
        of x emitted them (ops.gemm(emit_stats=True)), else a statistics pass reads x."""
        if self.stream is not None and (residual is x or residual is None) and x.is_contiguous():
            return ops.ffn_fused(x, self.stream, self.hidden)
        if residual is None:
            residual = x
        g = ops.gemm(x, self.w1, self.b1, act=ops.ACT_GEGLU, row_stats=stats if stats is not None else ops.layernorm_stats(x), col_sum=self.cs1)
        return ops.gemm(g, self.w2, self.b2, residual=residual)


# ----------------------------------------------------------------------------- blocks
class ResBlock:
    """ResnetBlock3D (resnet.py:110-204)."""

    def __init__(self, sd, key, cin, cout, groups, eps, device, temb_slice):
        self.cin, self.cout, self.groups, self.eps = cin, cout, groups, eps
        self.n1 = prep_norm(sd, key + ".norm1", device)
        self.n2 = prep_norm(sd, key + ".norm2", device)
        self.w1, self.b1 = prep_conv3x3(sd, key + ".conv1", device)
        self.w2, self.b2 = prep_conv3x3(sd, key + ".conv2", device)
        # transformed taps U = G g G^T of the convolutions wide enough for the Winograd form (16 / 9 of the weight bytes, kept beside the direct ones)
        self.u1 = ops.winograd_weights(sd[key + ".conv1.weight"].detach(), device) if (WINOGRAD and cin >= WINOGRAD_MIN_CIN and cin % 64 == 0) else None
        self.u2 = ops.winograd_weights(sd[key + ".conv2.weight"].detach(), device) if (WINOGRAD and cout >= WINOGRAD_MIN_CIN and cout % 64 == 0) else None
        self.sc = prep_linear(sd, key + ".conv_shortcut", device) if (key + ".conv_shortcut.weight") in sd else None
        self.temb_slice = temb_slice  # (start, stop) columns of the batched time_emb_proj output

    def __call__(self, x, temb_all, skip=None, out=None):
        rows = x.F * x.hw
        geom = (x.B * x.F, x.H, x.W)
        x2 = skip.t if skip is not None else None
        c1 = x.t.shape[1]
        tb = temb_all[:, self.temb_slice[0]:self.temb_slice[1]]
        # Where the convolution runs on the patch-tiled kernel (levels 0-1) GroupNorm + SiLU are applied to its input
        # patch in LDS: only the statistics pass reads the tensor, the normalised copy never exists (resnet.py:177-194).
        wino = x.t.shape[0] >= WINOGRAD_MIN_ROWS
        if self.u1 is not None and wino and ops.winograd_ok(geom, self.cin, c1 if x2 is not None else 0):
            # GroupNorm statistics (one read), then scale / shift + SiLU applied by the Winograd input transform: no normalised copy
            ab = ops.groupnorm_stats(x.t, x.B, rows, *self.n1, self.groups, self.eps, x2=x2)
            h = ops.winograd_conv3x3(x.t, geom, self.u1, self.b1, x2=x2, gn_ab=ab, gn_images_per_sample=x.F, gn_silu=True, row_bias=tb, rows_per_group=rows)
        elif FUSE_GN and ops.conv3x3_fuses_groupnorm(geom, self.cin, self.cout, c1 if x2 is not None else 0):
            ab = ops.groupnorm_stats(x.t, x.B, rows, *self.n1, self.groups, self.eps, x2=x2)
            h, _ = ops.conv3x3(x.t, geom, self.w1, self.b1, x2=x2, row_bias=tb, rows_per_group=rows,
                               gn_ab=ab, gn_images_per_sample=x.F, gn_silu=True)
        else:
            n = ops.groupnorm(x.t, x.B, rows, *self.n1, self.groups, self.eps, silu=True, x2=x2)
            h, _ = ops.conv3x3(n, geom, self.w1, self.b1, row_bias=tb, rows_per_group=rows)
        if self.sc is not None:
            res = ops.gemm(x.t, self.sc[0], self.sc[1], a2=x2)
        else:
            res = x.t
        if self.u2 is not None and wino and ops.winograd_ok(geom, self.cout):
            ab = ops.groupnorm_stats(h, x.B, rows, *self.n2, self.groups, self.eps)
            out = ops.winograd_conv3x3(h, geom, self.u2, self.b2, gn_ab=ab, gn_images_per_sample=x.F, gn_silu=True, residual=res, out=out)
        elif FUSE_GN and ops.conv3x3_fuses_groupnorm(geom, self.cout, self.cout):
            ab = ops.groupnorm_stats(h, x.B, rows, *self.n2, self.groups, self.eps)
            out, _ = ops.conv3x3(h, geom, self.w2, self.b2, residual=res, gn_ab=ab, gn_images_per_sample=x.F, gn_silu=True, out=out)
        else:
            n = ops.groupnorm(h, x.B, rows, *self.n2, self.groups, self.eps, silu=True)
            out, _ = ops.conv3x3(n, geom, self.w2, self.b2, residual=res, out=out)
        return x.like(out)


class SpatialTransformer:
    """Transformer3DModel + BasicTransformerBlock (attention.py:33-270)."""

    def __init__(self, sd, key, ch, heads, groups, device):
        self.ch, self.heads, self.groups = ch, heads, groups
        self.norm = prep_norm(sd, key + ".norm", device)
        self.proj_in = prep_linear(sd, key + ".proj_in", device)
        self.proj_out = prep_linear(sd, key + ".proj_out", device)
        b = key + ".transformer_blocks.0"
        # the three LayerNorms (attention.py:236,249,259) are folded into the GEMMs that consume them
        wf, col, bb = fold_layernorm(torch.cat([sd[f"{b}.attn1.to_{n}.weight"].float() for n in "qkv"], 0),
                                     sd[f"{b}.norm1.weight"], sd[f"{b}.norm1.bias"])
        self.wqkv, self.qkv_cs, self.qkv_b = _dev(wf, torch.float16, device), _dev(col, torch.float32, device), _dev(bb, torch.float32, device)
        self.wo1 = prep_linear(sd, f"{b}.attn1.to_out.0", device)
        wf, col, bb = fold_layernorm(sd[f"{b}.attn2.to_q.weight"], sd[f"{b}.norm2.weight"], sd[f"{b}.norm2.bias"])
        self.wq2, self.q2_cs, self.q2_b = _dev(wf, torch.float16, device), _dev(col, torch.float32, device), _dev(bb, torch.float32, device)
        self.wkv2 = _dev(torch.cat([sd[f"{b}.attn2.to_{n}.weight"].float() for n in "kv"], 0), torch.float16, device)
        self.wo2 = prep_linear(sd, f"{b}.attn2.to_out.0", device)
        self.ff = FeedForwardW(sd, b + ".ff", device, b + ".norm3", post_key=key + ".proj_out")
        # K = 320: every Linear of the block on the register-resident kernel (LayerNorm in registers, no statistics at all)
        self.rl = None
        if ROWLIN and ops.rowlin_supported(ch, ch) and (ch != 640 or ROWLIN_640):
            lin = lambda k: (sd[k + ".weight"], sd[k + ".bias"])
            self.rl = dict(proj_in=rowlin_stream(*lin(key + ".proj_in"), device), proj_out=rowlin_stream(*lin(key + ".proj_out"), device),
                           qkv=rowlin_stream(self.wqkv.float(), self.qkv_b, device), wo1=rowlin_stream(*lin(f"{b}.attn1.to_out.0"), device),
                           q2=rowlin_stream(self.wq2.float(), self.q2_b, device), wo2=rowlin_stream(*lin(f"{b}.attn2.to_out.0"), device))
        self.xa_stream, self.xa640_stream, self.xa_pre_stream = None, None, None
        if self.rl is not None and FUSE_XATTN and ops.xattn_fused_supported(ch, heads, 77, 128):
            self.xa_stream = fused.pack_xattn_stream(self.wq2.float(), self.q2_b, *lin(f"{b}.attn2.to_out.0")).to(device)
            if FUSE_XATTN_PRE:
                self.xa_pre_stream = fused.pack_xattn_stream(self.wq2.float(), self.q2_b, *lin(f"{b}.attn2.to_out.0"),
                                                             pre=lin(f"{b}.attn1.to_out.0")).to(device)
        elif self.rl is not None and FUSE_XATTN_640 and ops.xattn_attn_supported(ch, heads, 77, 128):
            # C = 640: LayerNorm -> q -> attention in one launch (insv2v_xattn_attn); to_out + residual stay a row Linear
            self.xa640_stream = fused.pack_xattn_q_stream(self.wq2.float(), self.q2_b).to(device)

    def project_context(self, ctx2d, ctx_len=0):
        """K/V of the text tokens: loop-invariant over the DDIM steps.  [B*L, 2C]; where the fused cross-attention
        kernel can run, a tuple (kv, per-sample fragment streams [B, n]) - the kernel's second weight stream."""
        kv = ops.gemm(ctx2d, self.wkv2)
        if self.xa_stream is not None and ctx_len and 64 < ctx_len <= 96:
            return kv, fused.pack_xattn_kv(kv, kv.shape[0] // ctx_len, ctx_len, self.ch, self.heads)
        if self.xa640_stream is not None and ctx_len and 64 < ctx_len <= 96:
            return kv, fused.pack_xattn640_kv(kv, kv.shape[0] // ctx_len, ctx_len, self.ch, self.heads)
        return kv

    def __call__(self, x, kv, ctx_len, shared=0):
        """shared = n > 0: the batch is 3 n samples laid out branch-major and samples [n, 2n) and [2n, 3n) hold IDENTICAL rows of x (the
        CFG branches (no text, video) and (text, video) up to the first text cross-attention, inference.py:183-194): GroupNorm,
        proj_in, q/k/v and the spatial self-attention run on the first 2 n samples and their results are copied for the last n."""
        C, hd, BF, HW = self.ch, self.ch // self.heads, x.B * x.F, x.hw
        scale = hd ** -0.5
        rl = self.rl if (C != 640 or _rowlin_640_pays(x.t.shape[0])) else None
        a = None
        if rl is not None and ROWLIN_GN and HW % 32 == 0 and shared > 0 and x.B == 3 * shared:
            r1, r2 = shared * x.F * HW, 2 * shared * x.F * HW
            ab = ops.groupnorm_stats(x.t[:r2], 2 * shared * x.F, HW, *self.norm, self.groups, 1e-6)
            h, st = torch.empty((x.t.shape[0], C), device=x.t.device, dtype=torch.float16), None
            ops.rowlin(x.t[:r2], rl["proj_in"], C, gn_ab=ab, gn_rows=HW, out=h[:r2])
            qkv = ops.rowlin(h[:r2], rl["qkv"], 3 * C, layernorm=True)
            a = torch.empty((x.t.shape[0], C), device=h.device, dtype=torch.float16)
            p = qkv.data_ptr()
            ops.attention(p, p + 2 * C, p + 4 * C, a[:r2], batch=2 * shared * x.F, heads=self.heads, head_dim=hd, seq_q=HW, seq_k=HW,
                          scale=scale, q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C,
                          q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0), o_addr=(1, HW * C, 0))
            ops.copy_rows(h[r2:], h[r1:r2])
            ops.copy_rows(a[r2:], a[r1:r2])
        elif rl is not None and ROWLIN_GN and HW % 32 == 0:
            ab = ops.groupnorm_stats(x.t, BF, HW, *self.norm, self.groups, 1e-6)
            h, st = ops.rowlin(x.t, rl["proj_in"], C, gn_ab=ab, gn_rows=HW), None
            qkv = ops.rowlin(h, rl["qkv"], 3 * C, layernorm=True)
        elif rl is not None:
            n = ops.groupnorm(x.t, BF, HW, *self.norm, self.groups, 1e-6)
            h, st = ops.rowlin(n, rl["proj_in"], C), None
            qkv = ops.rowlin(h, rl["qkv"], 3 * C, layernorm=True)
        else:
            n = ops.groupnorm(x.t, BF, HW, *self.norm, self.groups, 1e-6)
            # every LayerNorm input is the output of an N = C GEMM: its epilogue emits the row statistics (emit_stats), nothing re-reads h
            h, st = ops.gemm(n, *self.proj_in, emit_stats=True)
            # self attention over the h*w tokens of each frame
            qkv = ops.gemm(h, self.wqkv, self.qkv_b, row_stats=st, col_sum=self.qkv_cs)
        if a is None:
            a = torch.empty((x.t.shape[0], C), device=h.device, dtype=torch.float16)
            p = qkv.data_ptr()
            ops.attention(p, p + 2 * C, p + 4 * C, a, batch=BF, heads=self.heads, head_dim=hd, seq_q=HW, seq_k=HW,
                          scale=scale, q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C,
                          q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0), o_addr=(1, HW * C, 0))
        kv, kv_frag = kv if isinstance(kv, tuple) else (kv, None)
        if rl is None:
            kv_frag = None
        xa = kv_frag is not None and ops.xattn_fused_supported(C, self.heads, ctx_len, x.F * HW)
        if xa:   # out-proj of the self-attention + the whole cross-attention sub-block in one launch (x1 = h + to_out(a) stays in registers)
            if self.xa_pre_stream is not None and h.is_contiguous():
                h2 = ops.xattn_fused(a, self.xa_pre_stream, kv_frag, x.F * HW, self.heads, ctx_len, pre_residual=h)
            else:
                h = ops.rowlin(a, rl["wo1"], C, residual=h)
                h2 = ops.xattn_fused(h, self.xa_stream, kv_frag, x.F * HW, self.heads, ctx_len)
            if self.ff.stream_post is not None:
                return x.like(self.ff.with_proj_out(h2, x.t))
            h = self.ff(h2, None) if self.ff.stream is not None else self.ff(h2, h2, ops.layernorm_stats(h2))
            return x.like(ops.rowlin(h, rl["proj_out"], C, residual=x.t))
        if kv_frag is not None and self.xa640_stream is not None and ops.xattn_attn_supported(C, self.heads, ctx_len, x.F * HW):
            h = ops.rowlin(a, rl["wo1"], C, residual=h)
            a = ops.xattn_attn(h, self.xa640_stream, kv_frag, x.F * HW, self.heads, ctx_len)
            if self.ff.stream is not None:
                h = self.ff(ops.rowlin(a, rl["wo2"], C, residual=h), None)
            else:
                h, st = ops.rowlin(a, rl["wo2"], C, residual=h, emit_stats=True)
                if self.ff.w2p is not None:
                    return x.like(self.ff.with_proj_out_gemm(h, st, x.t))
                h = self.ff(h, h, st)
            return x.like(ops.rowlin(h, rl["proj_out"], C, residual=x.t))
        if rl is not None:
            h = ops.rowlin(a, rl["wo1"], C, residual=h)
            q = ops.rowlin(h, rl["q2"], C, layernorm=True)
        else:
            h, st = ops.gemm(a, *self.wo1, residual=h, emit_stats=True)
            # cross attention to the text tokens of the frame's sample
            q = ops.gemm(h, self.wq2, self.q2_b, row_stats=st, col_sum=self.q2_cs)
        a = torch.empty_like(a)
        kp = kv.data_ptr()
        ops.attention(q.data_ptr(), kp, kp + 2 * C, a, batch=BF, heads=self.heads, head_dim=hd, seq_q=HW, seq_k=ctx_len,
                      scale=scale, q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C,
                      q_addr=(1, HW * C, 0), kv_addr=(x.F, ctx_len * 2 * C, 0), o_addr=(1, HW * C, 0))
        if rl is not None:
            if self.ff.stream_post is not None:
                return x.like(self.ff.with_proj_out(ops.rowlin(a, rl["wo2"], C, residual=h), x.t))
            if self.ff.stream is not None:
                h = self.ff(ops.rowlin(a, rl["wo2"], C, residual=h), None)
            else:   # the two-GEMM feed-forward consumes finished row statistics: the row kernel emits them with its stores
                h, st = ops.rowlin(a, rl["wo2"], C, residual=h, emit_stats=True)
                if self.ff.w2p is not None:
                    return x.like(self.ff.with_proj_out_gemm(h, st, x.t))
                h = self.ff(h, h, st)
            return x.like(ops.rowlin(h, rl["proj_out"], C, residual=x.t))
        h, st = ops.gemm(a, *self.wo2, residual=h, emit_stats=True)
        if self.ff.w2p is not None:
            return x.like(self.ff.with_proj_out_gemm(h, st, x.t))
        h = self.ff(h, h, st)
        return x.like(ops.gemm(h, *self.proj_out, residual=x.t))


def sinusoid_table(d_model, max_len):
    """motion_module.py:229-233."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class MotionModule:
    """VanillaTemporalModule -> TemporalTransformer3DModel (motion_module.py:42-351)."""

    def __init__(self, sd, key, ch, groups, device, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), temporal_position_encoding=True,
                 temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, **unused):
        if not temporal_position_encoding or any(t != "Temporal_Self" for t in attention_block_types):
            raise NotImplementedError("only Temporal_Self attention with positional encoding is used by InsV2V")
        k = key + ".temporal_transformer"
        self.ch, self.heads, self.groups = ch, num_attention_heads, groups
        self.max_len = temporal_position_encoding_max_len
        self.norm = prep_norm(sd, k + ".norm", device)
        self.proj_in = prep_linear(sd, k + ".proj_in", device)
        self.proj_out = prep_linear(sd, k + ".proj_out", device)
        self.blocks = []
        for bi in range(num_transformer_block):
            b = f"{k}.transformer_blocks.{bi}"
            attns = []
            for ai in range(len(attention_block_types)):
                ab = f"{b}.attention_blocks.{ai}"
                wraw = torch.cat([sd[f"{ab}.to_{n}.weight"].float() for n in "qkv"], 0)
                pe_key = f"{ab}.pos_encoder.pe"
                pe = sd[pe_key].reshape(-1, ch).float() if pe_key in sd else sinusoid_table(ch, self.max_len)
                # LayerNorm (motion_module.py:206) folded into the QKV GEMM; the positional encoding added AFTER the
                # norm (motion_module.py:277-278, so it reaches q, k AND v) becomes the per-frame bias table pe @ W^T
                wf, col, bb = fold_layernorm(wraw, sd[f"{b}.norms.{ai}.weight"], sd[f"{b}.norms.{ai}.bias"])
                attns.append(dict(wqkv=_dev(wf, torch.float16, device), cs=_dev(col, torch.float32, device),
                                  b=_dev(bb, torch.float32, device), wo=prep_linear(sd, f"{ab}.to_out.0", device),
                                  pe_bias=_dev(pe @ wraw.t(), torch.float32, device),
                                  # K = 320: register-resident kernel; the q/k/v stream carries the per-frame table and is built per
                                  # (start, frames) on first use (rl_qkv); wf / bb / the table stay on the host for that
                                  rl_wo=rowlin_stream(sd[f"{ab}.to_out.0.weight"], sd[f"{ab}.to_out.0.bias"], device),
                                  host=(wf.float(), bb.float(), (pe @ wraw.t()).float()), rl_qkv={}, rl_tattn={}, pe_half={},
                                  host_o=(sd[f"{ab}.to_out.0.weight"].detach().half().float(), sd[f"{ab}.to_out.0.bias"].detach().float())))
            last = bi == num_transformer_block - 1
            self.blocks.append(dict(attns=attns, ff=FeedForwardW(sd, b + ".ff", device, b + ".ff_norm", post_key=(k + ".proj_out") if last else None)))
        self.device = device
        self.rl = None
        if ROWLIN and ops.rowlin_supported(ch, ch) and (ch != 640 or ROWLIN_640):
            self.rl = dict(proj_in=rowlin_stream(sd[k + ".proj_in.weight"], sd[k + ".proj_in.bias"], device),
                           proj_out=rowlin_stream(sd[k + ".proj_out.weight"], sd[k + ".proj_out.bias"], device))

    def _tattn_stream(self, at, start, F):
        """Stream of the whole attention sub-block for insv2v_tattn_fused (same per-frame table as _qkv_stream + the output projection)."""
        st = at["rl_tattn"].get((start, F))
        if st is None:
            wf, bb, pe_bias = at["host"]
            wo, bo = at["host_o"]
            st = at["rl_tattn"][(start, F)] = _dev(pack_tattn_stream(wf, pe_bias[start:start + F] + bb[None, :], wo, bo), torch.float16, self.device)
        return st

    def _tattn_qkv_stream(self, at, start, F):
        """Stream of LayerNorm -> q/k/v -> attention for insv2v_tattn_attn (C = 640: the output projection stays a row Linear)."""
        st = at["rl_tattn"].get((start, F, "qkv"))
        if st is None:
            wf, bb, pe_bias = at["host"]
            st = at["rl_tattn"][(start, F, "qkv")] = _dev(pack_tattn_qkv_stream(wf, pe_bias[start:start + F] + bb[None, :]), torch.float16, self.device)
        return st

    def _qkv_stream(self, at, start, F):
        """Stream of the fused q/k/v projection with the positional-encoding rows start .. start+F-1 folded into a per-frame bias."""
        st = at["rl_qkv"].get((start, F))
        if st is None:
            wf, bb, pe_bias = at["host"]
            st = at["rl_qkv"][(start, F)] = rowlin_stream(wf, None, self.device, table=pe_bias[start:start + F] + bb[None, :])
        return st

    def __call__(self, x, start=0):
        C, hd, HW, F = self.ch, self.ch // self.heads, x.hw, x.F
        if start + F > self.max_len:  # motion_module.py:236-241
            start -= self.max_len
        if start < 0:
            raise ValueError(f"start_index must be non-negative, but got {start}")
        # The per-frame bias step of insv2v_rowlin (and the fused attention blocks) hold 16 frames.  Longer windows (BASELINE config C5: 24
        # frames) keep the row kernels for everything that needs no frame table - GroupNorm + proj_in, the output projections with their
        # residuals, the fused feed-forward, proj_out - and take the folded-LayerNorm GEMM with a per-frame row bias + the generic
        # attention kernel for q/k/v only (round 4; round 3 sent the whole module down the generic path).
        rl = self.rl if (C != 640 or _rowlin_640_pays(x.t.shape[0])) else None
        rl_frames = rl is not None and F <= 16
        if rl is not None and ROWLIN_GN and HW % 32 == 0:
            ab = ops.groupnorm_stats(x.t, x.B * F, HW, *self.norm, self.groups, 1e-6)
            h, st = ops.rowlin(x.t, rl["proj_in"], C, gn_ab=ab, gn_rows=HW), None
        elif rl is not None:
            h, st = ops.rowlin(ops.groupnorm(x.t, x.B * F, HW, *self.norm, self.groups, 1e-6), rl["proj_in"], C), None
        else:
            h, st = ops.gemm(ops.groupnorm(x.t, x.B * F, HW, *self.norm, self.groups, 1e-6), *self.proj_in, emit_stats=True)
        if rl is not None and not rl_frames:
            st = ops.layernorm_stats(h)   # the q/k/v GEMM folds the LayerNorm: statistics of the proj_in rows
        fused_attn = rl_frames and FUSE_TATTN and ops.tattn_fused_supported(C, self.heads, F)
        for bi, blk in enumerate(self.blocks):
            for at in blk["attns"]:
                if fused_attn:   # the whole sub-block in one launch: q / k / v never exist in memory
                    h = ops.tattn_fused(h, self._tattn_stream(at, start, F), x.B, HW, self.heads, F)
                    continue
                pe_half = None
                if rl_frames and FUSE_TATTN_640 and ops.tattn_attn_supported(C, self.heads, F) and h.is_contiguous():
                    # C = 640: LayerNorm -> q/k/v -> attention in one launch (q, k, v never exist in memory), then to_out + residual
                    a = ops.tattn_attn(h, self._tattn_qkv_stream(at, start, F), x.B, HW, self.heads, F)
                    if blk["ff"].stream is None and at is blk["attns"][-1]:
                        h, st = ops.rowlin(a, at["rl_wo"], C, residual=h, emit_stats=True)
                    else:
                        h = ops.rowlin(a, at["rl_wo"], C, residual=h)
                    continue
                if rl_frames:
                    qkv = ops.rowlin(h, self._qkv_stream(at, start, F), 3 * C, layernorm=True, frames=F, rows_per_frame=HW)
                elif ATTN_PE_BIAS and ops.attention_short_supported(self.heads, hd, F):
                    pe_half = at["pe_half"].get((start, F))
                    if pe_half is None:
                        pe_half = at["pe_half"][(start, F)] = at["pe_bias"][start:start + F].half().contiguous()
                    qkv = ops.gemm(h, at["wqkv"], at["b"], row_stats=st, col_sum=at["cs"])
                else:
                    qkv = ops.gemm(h, at["wqkv"], at["b"], row_stats=st, col_sum=at["cs"],
                                   row_bias=at["pe_bias"][start:start + F], rows_per_group=HW, rb_mod=F)
                a = torch.empty((x.t.shape[0], C), device=h.device, dtype=torch.float16)
                p = qkv.data_ptr()
                addr = (HW, F * HW * 3 * C, 3 * C)
                ops.attention(p, p + 2 * C, p + 4 * C, a, batch=x.B * HW, heads=self.heads, head_dim=hd, seq_q=F, seq_k=F,
                              scale=hd ** -0.5, q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C,
                              q_addr=addr, kv_addr=addr, o_addr=(HW, F * HW * C, C), qkv_bias=pe_half)
                if rl is not None:
                    # finished row statistics ride with the stores where a folded-LayerNorm GEMM consumes this output: the two-GEMM
                    # feed-forward, or (windows longer than 16 frames) the next attention's q/k/v GEMM
                    if (blk["ff"].stream is None and at is blk["attns"][-1]) or (not rl_frames and at is not blk["attns"][-1]):
                        h, st = ops.rowlin(a, at["rl_wo"], C, residual=h, emit_stats=True)
                    else:
                        h, st = ops.rowlin(a, at["rl_wo"], C, residual=h), None   # (st described the rows this call replaced)
                else:
                    h, st = ops.gemm(a, *at["wo"], residual=h, emit_stats=True)
            if rl is not None and bi + 1 == len(self.blocks) and blk["ff"].stream_post is not None and h.is_contiguous():
                return x.like(blk["ff"].with_proj_out(h, x.t))
            if bi + 1 == len(self.blocks) and blk["ff"].w2p is not None:
                return x.like(blk["ff"].with_proj_out_gemm(h, st, x.t))
            h = blk["ff"](h, h, st)
            if bi + 1 < len(self.blocks) and not rl_frames:  # a further transformer block starts from a statistics pass over the FF output
                st = ops.layernorm_stats(h)
        if rl is not None:
            return x.like(ops.rowlin(h, rl["proj_out"], C, residual=x.t))
        return x.like(ops.gemm(h, *self.proj_out, residual=x.t))


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet3DConditionModel:
    """See module docstring.  ``load_state_dict`` takes a CPU/any-device state dict with the
    reference key names and stages fp16 weights on ``device``."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 down_block_types=("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",),
                 up_block_types=("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3,
                 layers_per_block=2, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 attention_head_dim=8, flip_sin_to_cos=True, freq_shift=0, use_motion_module=True,
                 motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=True,
                 motion_module_decoder_only=False, motion_module_type="Vanilla", motion_module_kwargs=None,
                 device="cuda", **unused):
        if not flip_sin_to_cos:
            raise NotImplementedError("flip_sin_to_cos=False")
        if motion_module_type != "Vanilla":
            raise ValueError
        self.cfg = dict(in_channels=in_channels, out_channels=out_channels, ch=list(block_out_channels),
                        down=list(down_block_types), up=list(up_block_types), layers=layers_per_block,
                        groups=norm_num_groups, eps=float(norm_eps), ctx_dim=cross_attention_dim,
                        heads=attention_head_dim, shift=float(freq_shift), motion=use_motion_module,
                        mres=list(motion_module_resolutions), mmid=motion_module_mid_block,
                        mdec=motion_module_decoder_only, mkw=dict(motion_module_kwargs or {}))
        self.device = torch.device(device)
        self.loaded = False
        self.weights_version = 0  # bumped by load_state_dict: captured hipGraphs hold the OLD weight pointers

    # -- structure -----------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        c, dev = self.cfg, self.device
        ch, groups, eps, heads = c["ch"], c["groups"], c["eps"], c["heads"]
        self.conv_in = prep_conv3x3(sd, "conv_in", dev)
        self.in_pad = self.conv_in[0].shape[1] // 9
        self.te1 = prep_linear(sd, "time_embedding.linear_1", dev)
        self.te2 = prep_linear(sd, "time_embedding.linear_2", dev)
        temb_w, temb_b, self._temb_off = [], [], 0

        def res(key, cin, cout):
            w, b = sd[key + ".time_emb_proj.weight"], sd[key + ".time_emb_proj.bias"]
            sl = (self._temb_off, self._temb_off + cout)
            self._temb_off += cout
            temb_w.append(w.float()), temb_b.append(b.float())
            return ResBlock(sd, key, cin, cout, groups, eps, dev, sl)

        def motion(key, chn, on):
            return MotionModule(sd, key, chn, groups, dev, **c["mkw"]) if on else None

        self.down = []
        out = ch[0]
        for i, typ in enumerate(c["down"]):
            cin, out = out, ch[i]
            cross = typ.startswith("CrossAttn")
            mot = c["motion"] and (2 ** i in c["mres"]) and not c["mdec"]
            k = f"down_blocks.{i}"
            blk = dict(res=[], attn=[], mot=[], down=None)
            for j in range(c["layers"]):
                blk["res"].append(res(f"{k}.resnets.{j}", cin if j == 0 else out, out))
                blk["attn"].append(SpatialTransformer(sd, f"{k}.attentions.{j}", out, heads, groups, dev) if cross else None)
                blk["mot"].append(motion(f"{k}.motion_modules.{j}", out, mot))
            if i != len(ch) - 1:
                blk["down"] = prep_conv3x3(sd, f"{k}.downsamplers.0.conv", dev)
            self.down.append(blk)
        k = "mid_block"
        self.mid = dict(res=[res(f"{k}.resnets.0", ch[-1], ch[-1]), res(f"{k}.resnets.1", ch[-1], ch[-1])],
                        attn=SpatialTransformer(sd, f"{k}.attentions.0", ch[-1], heads, groups, dev),
                        mot=motion(f"{k}.motion_modules.0", ch[-1], c["motion"] and c["mmid"]))
        self.up = []
        rev = ch[::-1]
        out = rev[0]
        for i, typ in enumerate(c["up"]):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            cross = typ.startswith("CrossAttn")
            mot = c["motion"] and (2 ** (3 - i) in c["mres"])
            k = f"up_blocks.{i}"
            blk = dict(res=[], attn=[], mot=[], up=None)
            n_layers = c["layers"] + 1
            for j in range(n_layers):
                skip = cin if j == n_layers - 1 else out
                rin = prev if j == 0 else out
                blk["res"].append(res(f"{k}.resnets.{j}", rin + skip, out))
                blk["attn"].append(SpatialTransformer(sd, f"{k}.attentions.{j}", out, heads, groups, dev) if cross else None)
                blk["mot"].append(motion(f"{k}.motion_modules.{j}", out, mot))
            if i != len(ch) - 1:
                blk["up"] = prep_conv3x3(sd, f"{k}.upsamplers.0.conv", dev)
                # Upsample3D (nearest x2 + 3x3 convolution) in Winograd form: 9 transformed taps per INPUT pixel, 4 x fewer MACs (csrc/winograd.hip)
                blk["up_u"] = (ops.winograd_weights(sd[f"{k}.upsamplers.0.conv.weight"].detach(), dev, upsample=True)
                               if (WINOGRAD and out >= WINOGRAD_UP_MIN_CIN and out % 64 == 0) else None)
            self.up.append(blk)
        self.norm_out = prep_norm(sd, "conv_norm_out", dev)
        self.conv_out = prep_conv3x3(sd, "conv_out", dev)
        # stacked clips: conv_out (320 -> 4) as a plain GEMM of per-tap partial outputs + a shifted sum (ops.conv3x3_narrow) instead of the
        # implicit-GEMM form on a 64-wide output tile
        self.conv_out_taps = _dev(ops.tap_weights(sd["conv_out.weight"]), torch.float16, dev)
        # all 22 time_emb_proj Linears as ONE GEMM (resnet.py:183)
        self.temb_w = _dev(torch.cat(temb_w, 0), torch.float16, dev)
        self.temb_b = _dev(torch.cat(temb_b, 0), torch.float32, dev)
        self.loaded = True
        self.weights_version += 1
        return self

    def spatial_transformers(self):
        for blk in self.down:
            yield from (a for a in blk["attn"] if a is not None)
        yield self.mid["attn"]
        for blk in self.up:
            yield from (a for a in blk["attn"] if a is not None)

    # -- channels-last forward ---------------------------------------------------------------------
    def project_context(self, ctx):
        """ctx [B, L, ctx_dim] (any float dtype) -> per-layer text K/V; hoisted out of the step loop."""
        ctx2d = ctx.to(device=self.device, dtype=torch.float16).reshape(-1, ctx.shape[-1]).contiguous()
        return [st.project_context(ctx2d, ctx.shape[1]) for st in self.spatial_transformers()], ctx.shape[1]

    def forward_cl(self, x_in, t_dev, kvs, ctx_len, B, F, H, W, start=0, cfg_clips=0):
        """x_in: [B*F*H*W, in_pad] fp16 channels-last (zero padded channels); t_dev: [B] fp32 on device;
        kvs: project_context() output.  Returns eps [B*F*H*W, out_channels] fp32.
        cfg_clips = n > 0 (B = 3 n): the caller's promise that the samples are the three classifier-free-guidance branches of n clips,
        branch-major, and that samples [n, 2n) and [2n, 3n) - (no text, video) and (text, video), inference.py:183-194 - have identical
        x_in and t: everything in front of the first text cross-attention (the first ResnetBlock3D and the first spatial
        self-attention) is computed once for the two and copied (DEDUP_CFG)."""
        c = self.cfg
        emb = ops.timestep_embedding(t_dev, c["ch"][0], c["shift"])
        emb = ops.gemm(emb, *self.te1, act=ops.ACT_SILU)
        semb = ops.gemm(emb, *self.te2, act=ops.ACT_SILU)  # silu(emb): the only form the resnets consume
        temb_all = ops.gemm(semb, self.temb_w, self.temb_b, out_fp32=True)
        kv_iter = iter(kvs)
        t, geom = ops.conv3x3(x_in, (B * F, H, W), *self.conv_in)
        x = Act(t, B, F, H, W)
        skips = [x]
        shared = cfg_clips if (DEDUP_CFG and cfg_clips > 0 and B == 3 * cfg_clips) else 0
        for blk in self.down:
            for r, a, m in zip(blk["res"], blk["attn"], blk["mot"]):
                if shared and a is not None:
                    r1, r2 = shared * F * x.hw, 2 * shared * F * x.hw
                    full = torch.empty((x.t.shape[0], r.cout), device=x.t.device, dtype=torch.float16)
                    r(Act(x.t[:r2], 2 * shared, F, x.H, x.W), temb_all[:2 * shared], out=full[:r2])
                    ops.copy_rows(full[r2:], full[r1:r2])
                    x = a(x.like(full), next(kv_iter), ctx_len, shared=shared)
                    shared = 0
                    if m is not None:
                        x = m(x, start)
                    skips.append(x)
                    continue
                shared = 0
                x = r(x, temb_all)
                if a is not None:
                    x = a(x, next(kv_iter), ctx_len)
                if m is not None:
                    x = m(x, start)
                skips.append(x)
            if blk["down"] is not None:
                t, (_, oh, ow) = ops.conv3x3(x.t, (B * F, x.H, x.W), *blk["down"], stride=2)
                x = x.like(t, oh, ow)
                skips.append(x)
        x = self.mid["res"][0](x, temb_all)
        x = self.mid["attn"](x, next(kv_iter), ctx_len)
        if self.mid["mot"] is not None:
            x = self.mid["mot"](x, start)
        x = self.mid["res"][1](x, temb_all)
        for blk in self.up:
            for r, a, m in zip(blk["res"], blk["attn"], blk["mot"]):
                x = r(x, temb_all, skip=skips.pop())
                if a is not None:
                    x = a(x, next(kv_iter), ctx_len)
                if m is not None:
                    x = m(x, start)
            if blk["up"] is not None:
                if blk.get("up_u") is not None and x.t.shape[0] >= WINOGRAD_MIN_ROWS // 4 and ops.winograd_ok((B * F, x.H, x.W), x.t.shape[1], upsample=True):
                    t, oh, ow = ops.winograd_conv3x3(x.t, (B * F, x.H, x.W), blk["up_u"], blk["up"][1], upsample=True), 2 * x.H, 2 * x.W
                else:
                    t, (_, oh, ow) = ops.conv3x3(x.t, (B * F, x.H, x.W), *blk["up"], upsample=True)
                x = x.like(t, oh, ow)
        n = ops.groupnorm(x.t, B, F * x.hw, *self.norm_out, c["groups"], c["eps"], silu=True)
        if NARROW_CONV_OUT and n.shape[0] >= NARROW_CONV_MIN_ROWS:
            return ops.conv3x3_narrow(n, (B * F, x.H, x.W), self.conv_out_taps, self.conv_out[1], self.conv_out[0].shape[0])
        eps, _ = ops.conv3x3(n, (B * F, x.H, x.W), *self.conv_out, out_fp32=True)
        return eps

    # -- reference-compatible call -----------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states, video_start_index=0, **unused):
        if not self.loaded:
            raise RuntimeError("UNet3DConditionModel: load_state_dict() has not been called")
        B, Cin, F, H, W = sample.shape
        if any(s % 8 for s in (H, W)):
            raise NotImplementedError("latent height/width must be multiples of 8 (three stride-2 stages)")
        dev = self.device
        x = sample.to(device=dev, dtype=torch.float32).permute(0, 2, 1, 3, 4).reshape(B * F, Cin, H, W)
        x_in = ops.nchw_to_nhwc_f16(x, self.in_pad)
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep])
        t_dev = timestep.reshape(-1).to(device=dev, dtype=torch.float32).expand(B).contiguous()
        kvs, L = self.project_context(encoder_hidden_states)
        eps = self.forward_cl(x_in, t_dev, kvs, L, B, F, H, W, start=video_start_index)
        co = self.cfg["out_channels"]
        out = ops.nhwc_to_nchw_f32(eps, B * F, co, H, W).reshape(B, F, co, H, W).permute(0, 2, 1, 3, 4)
        return UNetOutput(out)

    forward = __call__

    def to(self, *a, **k):
        return self

    def eval(self):
        return self
