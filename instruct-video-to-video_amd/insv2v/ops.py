"""Thin tensor-level wrappers over the C ABI (one Python call = one kernel family launch).

PyTorch is plumbing here: device memory (torch.empty), the current HIP stream, and nothing
else.  No op has a torch/CPU fallback; non-CUDA tensors are rejected.
"""
import ctypes as C

import os
import torch

from . import _lib
from ._lib import GemmDesc, GroupNormDesc, LayerNormDesc, AttentionDesc, StepDesc, FfnDesc, RowLinDesc, TattnDesc, XattnDesc, WinogradInDesc, WinogradOutDesc, check

ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_QUICK_GELU = 0, 1, 2, 3
ACT_RELU, ACT_SIGMOID, ACT_TANH = 4, 5, 6   # the optical-flow network's (insv2v/raft.py)
EW_RELU, EW_ADD_RELU, EW_TANH, EW_GRU_RH, EW_GRU_OUT = 1, 2, 3, 4, 5
_byref = C.byref


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing (bench.py roofline leg): list of (kernel family, algorithmic flops, ev0, ev1)
# with HIP events recorded on the launch stream.  None = no overhead.
_REC = None


def set_launch_recorder(rec):
    global _REC
    _REC = rec


class _timed:
    __slots__ = ("name", "work", "e0", "tag")

    def __init__(self, name, work, tag=None):
        self.name, self.work, self.tag = name, work, tag

    def __enter__(self):
        if _REC is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _REC is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _REC.append((self.name, self.work, self.e0, e1) if self.tag is None else (self.name, self.work, self.e0, e1, self.tag))
        return False


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _req(t, dtype, name):
    if not torch.is_tensor(t):
        raise _lib.HipKernelError(f"{name}: expected a CUDA {dtype} tensor, got {type(t).__name__}")
    if not (t.is_cuda and t.dtype == dtype):
        raise _lib.HipKernelError(f"{name}: expected a CUDA {dtype} tensor, got {t.device} {t.dtype}")
    return t


_WORKSPACE = {}
_WS_OVERRIDE = None
WORKSPACE_BYTES = 64 << 20


def new_workspace(device):
    """fp32 split-K scratch.  A captured hipGraph must own the scratch its kernels point at: GraphedUNet allocates one
    per concurrent branch BEFORE capture and installs it with ``workspace(...)`` (a scratch allocated lazily inside a
    capture would live in that graph's private memory pool while being cached here for other graphs)."""
    return torch.empty(WORKSPACE_BYTES // 4, device=device, dtype=torch.float32)


class workspace:
    """Context manager: launches inside use ``ws`` as their split-K scratch (one per concurrently running stream)."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        global _WS_OVERRIDE
        self.prev, _WS_OVERRIDE = _WS_OVERRIDE, self.ws

    def __exit__(self, *exc):
        global _WS_OVERRIDE
        _WS_OVERRIDE = self.prev
        return False


def _workspace(device):
    """The installed scratch, else a persistent one per (device, stream): launches on one stream are serialised, so
    sharing within a stream is safe."""
    if _WS_OVERRIDE is not None:
        return _WS_OVERRIDE
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _WORKSPACE.get(key)
    if ws is None:
        ws = _WORKSPACE[key] = new_workspace(device)
    return ws


class RowStats:
    """LayerNorm statistics of a token matrix as the PRODUCING GEMM left them: ``parts`` [nparts, M, 2] fp32 partial
    (sum, sum of squares) per column tile (gemm(..., emit_stats=True)), finalised by the consumer (gemm(row_stats=RowStats))."""
    __slots__ = ("parts", "nparts", "eps")

    def __init__(self, parts, nparts, eps):
        self.parts, self.nparts, self.eps = parts, nparts, eps


_STATS_PARTS_CACHE = {}


def gemm(a, w, bias=None, *, a2=None, act=ACT_NONE, residual=None, row_bias=None, rows_per_group=0, rb_mod=0,
         row_stats=None, col_sum=None, emit_stats=False, ln_eps=1e-5,
         out=None, out_fp32=False, alpha=1.0, tile=0, split_k=0, batch=1, a_bs=0, w_bs=0, c_bs=0, r_bs=0,
         M=None, N=None, K=None, lda=None, ldw=None, ldc=None):
    """out[M,N'] = epilogue(alpha * [a|a2] @ w^T); see insv2v_gemm in include/insv2v_hip.h."""
    lib = _lib.load()
    _req(a, torch.float16, "gemm.a"), _req(w, torch.float16, "gemm.w")
    M = a.shape[0] if M is None else M
    N = w.shape[0] if N is None else N
    k1 = a.shape[1]
    K = (k1 + (a2.shape[1] if a2 is not None else 0)) if K is None else K
    n_out = N // 2 if act == ACT_GEGLU else N
    if out is None:
        shape = (M, n_out) if batch == 1 else (batch, M, n_out)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if out_fp32 else torch.float16)
    d = GemmDesc()
    d.a, d.w, d.c = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.lda = a.stride(0) if lda is None else lda
    d.ldw = w.stride(0) if ldw is None else ldw
    d.ldc = (out.stride(-2) if ldc is None else ldc)
    if a2 is not None:
        _req(a2, torch.float16, "gemm.a2")
        d.a2, d.lda2, d.k_split = a2.data_ptr(), a2.stride(0), k1
    if bias is not None:
        d.bias = _req(bias, torch.float32, "gemm.bias").data_ptr()
    if row_bias is not None:
        d.row_bias, d.ld_rb, d.rows_per_group = _req(row_bias, torch.float32, "gemm.row_bias").data_ptr(), row_bias.stride(0), rows_per_group
    if residual is not None:
        d.residual, d.ldr = _req(residual, torch.float16, "gemm.residual").data_ptr(), residual.stride(-2)
    scratch = None
    if isinstance(row_stats, RowStats):
        d.row_stats = row_stats.parts.data_ptr()
        d.stats_parts, d.ln_eps = row_stats.nparts, row_stats.eps
        scratch = torch.empty((M, 2), device=a.device, dtype=torch.float32)  # only touched by kernels that park (mean, rstd); held until the launch
        d.stats_scratch = scratch.data_ptr()
        d.col_sum = _req(col_sum, torch.float32, "gemm.col_sum").data_ptr()
    elif row_stats is not None:
        d.row_stats = _req(row_stats, torch.float32, "gemm.row_stats").data_ptr()
        d.col_sum = _req(col_sum, torch.float32, "gemm.col_sum").data_ptr()
    d.rb_mod = rb_mod
    d.M, d.N, d.K, d.act, d.c_fp32, d.alpha, d.tile = M, N, K, act, int(out.dtype == torch.float32), alpha, tile
    d.batch, d.a_bs, d.w_bs, d.c_bs, d.r_bs = batch, a_bs, w_bs, c_bs, r_bs
    if batch == 1:
        ws = _workspace(a.device)
        d.workspace, d.workspace_bytes, d.split_k = ws.data_ptr(), ws.numel() * 4, split_k
    stats = None
    if emit_stats:
        # statistics of the rows just produced, for the LayerNorm that consumes them (one pair per column tile of the kernel the
        # library picks for this problem; problems it cannot instrument fall back to the statistics pass over the output)
        key = (M, N, K, d.lda, d.ldc, d.ldr, residual is not None, act, tile, out.dtype, a.data_ptr() % 16, out.data_ptr() % 16, batch, split_k,
               residual.data_ptr() % 16 if residual is not None else 0,
               # the dispatch (pick_pingpong: 160-column parts on gemm_r8, 128 on the 128x128 tile) also looks at these
               row_bias is not None, rows_per_group, d.ld_rb, rb_mod, a2 is not None, row_stats is not None)
        nparts = _STATS_PARTS_CACHE.get(key)
        if nparts is None:
            nparts = _STATS_PARTS_CACHE[key] = int(lib.insv2v_gemm_stats_parts(_byref(d)))
        if nparts > 0:
            stats = RowStats(torch.empty((nparts, M, 2), device=a.device, dtype=torch.float32), nparts, ln_eps)
            d.stats_out = stats.parts.data_ptr()
    with _timed("gemm_kernel", 2.0 * M * N * K * batch, ("lin", M, N, K, batch, act, residual is not None)):
        rc = lib.insv2v_gemm(_byref(d), _stream())
        if rc == -2 and stats is not None:   # documented behaviour: a problem that cannot emit statistics gets the statistics pass
            stats, d.stats_out = None, None
            rc = lib.insv2v_gemm(_byref(d), _stream())
        check(rc, "insv2v_gemm")
    del scratch
    if emit_stats:
        return out, (stats if stats is not None else layernorm_stats(out, ln_eps))
    return out


def copy_rows(dst, src):
    """dst <- src (a device-to-device copy of token rows on the current stream), visible to the launch recorder like any other launch:
    the duplicated CFG-branch rows of unet.forward_cl(cfg_clips=...)."""
    with _timed("copy", 0.0, ("copy", src.shape[0], src.shape[1])):
        dst.copy_(src)
    return dst


def ffn_fused_supported(C, hidden):
    """True if insv2v_ffn_fused handles this width (the register-resident kernel exists for C = 320, hidden = 1280)."""
    return int(_lib.load().insv2v_ffn_stream_elems(C, hidden, 0)) > 0


def ffn_fused(x, wstream, hidden, eps=1e-5, out=None, post_residual=None):
    """out = x + FeedForward_geglu(LayerNorm(x)) in one launch (insv2v_ffn_fused); wstream from fused.pack_ffn_stream.
    post_residual: the stream also carries the module's trailing Linear (pack_ffn_stream(post=...)):
    out = Wp (x + FF(LN(x))) + bp + post_residual."""
    lib = _lib.load()
    _req(x, torch.float16, "ffn.x"), _req(wstream, torch.float16, "ffn.wstream")
    M, C = x.shape
    if wstream.numel() != int(lib.insv2v_ffn_stream_elems(C, hidden, int(post_residual is not None))):
        raise _lib.HipKernelError(f"ffn_fused: weight stream of {wstream.numel()} halfs does not match C={C}, hidden={hidden}, post={post_residual is not None}")
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=torch.float16)
    d = FfnDesc()
    d.x, d.out, d.wstream, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), wstream.data_ptr(), x.stride(0), out.stride(0)
    d.M, d.C, d.hidden, d.eps = M, C, hidden, eps
    if post_residual is not None:
        d.post, d.post_residual, d.ld_post = 1, _req(post_residual, torch.float16, "ffn.post_residual").data_ptr(), post_residual.stride(0)
    with _timed("gemm_kernel", 2.0 * M * C * 3 * hidden + (2.0 * M * C * C if post_residual is not None else 0.0), ("ffn", M, C, hidden)):
        check(lib.insv2v_ffn_fused(_byref(d), _stream()), "insv2v_ffn_fused")
    return out


def rowlin_supported(N, K):
    """True if insv2v_rowlin handles a [N, K] Linear (K = 320 or 640, N a multiple of 64)."""
    return int(_lib.load().insv2v_rowlin_stream_elems(N, K)) > 0


def rowlin(x, wstream, N, *, layernorm=False, residual=None, frames=0, rows_per_frame=0, eps=1e-5, out=None, emit_stats=False, stats_eps=1e-5,
           gn_ab=None, gn_rows=0):
    """out = [LayerNorm](x) W^T + bias [+ residual] on the register-resident kernel (insv2v_rowlin); wstream from
    fused.pack_linear_stream (frames > 0: it carries a per-frame bias table and row m uses frame (m // rows_per_frame) % frames)."""
    lib = _lib.load()
    _req(x, torch.float16, "rowlin.x"), _req(wstream, torch.float16, "rowlin.wstream")
    M, K = x.shape
    if wstream.numel() != int(lib.insv2v_rowlin_stream_elems(N, K)):
        raise _lib.HipKernelError(f"rowlin: weight stream of {wstream.numel()} halfs does not match N={N}, K={K}")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float16)
    d = RowLinDesc()
    d.x, d.out, d.wstream, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), wstream.data_ptr(), x.stride(0), out.stride(0)
    if residual is not None:
        d.residual, d.ldr = _req(residual, torch.float16, "rowlin.residual").data_ptr(), residual.stride(0)
    d.M, d.N, d.K, d.layernorm, d.eps = M, N, K, int(layernorm), eps
    d.frame_bias, d.rows_per_frame, d.frames = int(frames > 0), rows_per_frame, frames
    if gn_ab is not None:   # GroupNorm of x applied on the fly ((scale, shift) pairs from groupnorm_stats)
        d.gn_ab, d.gn_rows = _req(gn_ab, torch.float32, "rowlin.gn_ab").data_ptr(), gn_rows
    stats = None
    if emit_stats:   # finished (mean, rstd) of the output rows for a following folded-LayerNorm GEMM
        stats = torch.empty((M, 2), device=x.device, dtype=torch.float32)
        d.stats_out, d.stats_eps = stats.data_ptr(), stats_eps
    with _timed("gemm_kernel", 2.0 * M * N * K, ("rowlin", M, N, K, int(layernorm), residual is not None)):
        check(lib.insv2v_rowlin(_byref(d), _stream()), "insv2v_rowlin")
    return (out, stats) if emit_stats else out


def tattn_fused_supported(C, heads, frames):
    """True if insv2v_tattn_fused handles this temporal attention block (C = 320, 8 heads, exactly 16 frames)."""
    return int(_lib.load().insv2v_tattn_stream_elems(C, heads, frames)) > 0


def tattn_fused(x, wstream, samples, HW, heads, frames, eps=1e-5, out=None):
    """out = x + to_out(attention over the frames(LayerNorm(x) + pe -> q, k, v)) in one launch (insv2v_tattn_fused); x rows ordered
    (sample, frame, pixel); wstream from fused.pack_tattn_stream."""
    lib = _lib.load()
    _req(x, torch.float16, "tattn.x"), _req(wstream, torch.float16, "tattn.wstream")
    M, C = x.shape
    if M != samples * frames * HW:
        raise _lib.HipKernelError(f"tattn_fused: {M} rows != {samples} samples x {frames} frames x {HW} pixels")
    if wstream.numel() != int(lib.insv2v_tattn_stream_elems(C, heads, frames)):
        raise _lib.HipKernelError(f"tattn_fused: weight stream of {wstream.numel()} halfs does not match C={C}, heads={heads}, frames={frames}")
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=torch.float16)
    d = TattnDesc()
    d.x, d.out, d.wstream, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), wstream.data_ptr(), x.stride(0), out.stride(0)
    d.samples, d.HW, d.C, d.heads, d.frames, d.eps, d.scale = samples, HW, C, heads, frames, eps, (C // heads) ** -0.5
    with _timed("gemm_kernel", 2.0 * M * C * 4 * C + 4.0 * M * frames * C, ("tattn", M, C, heads, frames)):
        check(lib.insv2v_tattn_fused(_byref(d), _stream()), "insv2v_tattn_fused")
    return out


def tattn_attn_supported(C, heads, frames):
    """True if insv2v_tattn_attn handles this temporal attention block (C = 640, 8 heads, exactly 16 frames)."""
    return int(_lib.load().insv2v_tattn_attn_stream_elems(C, heads, frames)) > 0


def tattn_attn(x, wstream, samples, HW, heads, frames, eps=1e-5, out=None):
    """out = attention over the frames(LayerNorm(x) + pe -> q, k, v), WITHOUT to_out / residual, in one launch (insv2v_tattn_attn, C = 640);
    x rows ordered (sample, frame, pixel); wstream from fused.pack_tattn_qkv_stream."""
    lib = _lib.load()
    _req(x, torch.float16, "tattn_attn.x"), _req(wstream, torch.float16, "tattn_attn.wstream")
    M, C = x.shape
    if M != samples * frames * HW:
        raise _lib.HipKernelError(f"tattn_attn: {M} rows != {samples} samples x {frames} frames x {HW} pixels")
    if wstream.numel() != int(lib.insv2v_tattn_attn_stream_elems(C, heads, frames)):
        raise _lib.HipKernelError(f"tattn_attn: weight stream of {wstream.numel()} halfs does not match C={C}, heads={heads}, frames={frames}")
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=torch.float16)
    d = TattnDesc()
    d.x, d.out, d.wstream, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), wstream.data_ptr(), x.stride(0), out.stride(0)
    d.samples, d.HW, d.C, d.heads, d.frames, d.eps, d.scale = samples, HW, C, heads, frames, eps, (C // heads) ** -0.5
    with _timed("gemm_kernel", 2.0 * M * C * 3 * C + 4.0 * M * frames * C, ("tattn_attn", M, C, heads, frames)):
        check(lib.insv2v_tattn_attn(_byref(d), _stream()), "insv2v_tattn_attn")
    return out


def xattn_fused_supported(C, heads, ctx_len, rows_per_sample):
    """True if insv2v_xattn_fused handles this text cross-attention block (C = 320, 8 heads, 64 < ctx_len <= 96, samples in 128-row tiles)."""
    return int(_lib.load().insv2v_xattn_stream_elems(C, heads, 0)) > 0 and 64 < ctx_len <= 96 and rows_per_sample % 128 == 0


def xattn_fused(x, wstream, kvstream, rows_per_sample, heads, ctx_len, eps=1e-5, out=None, pre_residual=None):
    """out = x + to_out(attention(LayerNorm(x) -> q; text K, V of the row's sample)) in one launch (insv2v_xattn_fused).
    wstream from fused.pack_xattn_stream, kvstream [samples, ...] from fused.pack_xattn_kv.
    pre_residual: x is the SELF-attention output and wstream (pack_xattn_stream(pre=...)) starts with that attention's output projection:
    x1 = to_out1(x) + pre_residual replaces x above and never exists in memory."""
    lib = _lib.load()
    _req(x, torch.float16, "xattn.x"), _req(wstream, torch.float16, "xattn.wstream"), _req(kvstream, torch.float16, "xattn.kvstream")
    M, C = x.shape
    if M % rows_per_sample or kvstream.shape[0] != M // rows_per_sample:
        raise _lib.HipKernelError(f"xattn_fused: {M} rows, {rows_per_sample} per sample, K/V streams of {kvstream.shape[0]} samples")
    if wstream.numel() != int(lib.insv2v_xattn_stream_elems(C, heads, 2 if pre_residual is not None else 0)) or \
            kvstream[0].numel() != int(lib.insv2v_xattn_stream_elems(C, heads, 1)):
        raise _lib.HipKernelError(f"xattn_fused: weight / K-V streams do not match C={C}, heads={heads}")
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=torch.float16)
    d = XattnDesc()
    d.x, d.out, d.wstream, d.kvstream, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), wstream.data_ptr(), kvstream.data_ptr(), x.stride(0), out.stride(0)
    d.M, d.rows_per_sample, d.C, d.heads, d.ctx_len, d.eps, d.scale = M, rows_per_sample, C, heads, ctx_len, eps, (C // heads) ** -0.5
    if pre_residual is not None:
        d.pre_residual, d.ld_pre = _req(pre_residual, torch.float16, "xattn.pre_residual").data_ptr(), pre_residual.stride(0)
    with _timed("gemm_kernel", (6.0 if pre_residual is not None else 4.0) * M * C * C + 4.0 * M * ctx_len * C, ("xattn", M, C, heads, ctx_len, pre_residual is not None)):
        check(lib.insv2v_xattn_fused(_byref(d), _stream()), "insv2v_xattn_fused")
    return out


def xattn_attn_supported(C, heads, ctx_len, rows_per_sample):
    """True if insv2v_xattn_attn handles this text cross-attention (C = 640, 8 heads, 64 < ctx_len <= 96, samples in 128-row tiles)."""
    return int(_lib.load().insv2v_xattn_attn_stream_elems(C, heads, 0)) > 0 and 64 < ctx_len <= 96 and rows_per_sample % 128 == 0


def xattn_attn(x, wstream, kvstream, rows_per_sample, heads, ctx_len, eps=1e-5, out=None):
    """out = attention(LayerNorm(x) -> q; text K, V of the row's sample), WITHOUT to_out / residual, in one launch (insv2v_xattn_attn, C = 640).
    wstream from fused.pack_xattn_q_stream, kvstream [samples, ...] from fused.pack_xattn640_kv."""
    lib = _lib.load()
    _req(x, torch.float16, "xattn_attn.x"), _req(wstream, torch.float16, "xattn_attn.wstream"), _req(kvstream, torch.float16, "xattn_attn.kvstream")
    M, C = x.shape
    if M % rows_per_sample or kvstream.shape[0] != M // rows_per_sample:
        raise _lib.HipKernelError(f"xattn_attn: {M} rows, {rows_per_sample} per sample, K/V streams of {kvstream.shape[0]} samples")
    if wstream.numel() != int(lib.insv2v_xattn_attn_stream_elems(C, heads, 0)) or kvstream[0].numel() != int(lib.insv2v_xattn_attn_stream_elems(C, heads, 1)):
        raise _lib.HipKernelError(f"xattn_attn: weight / K-V streams do not match C={C}, heads={heads}")
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=torch.float16)
    d = XattnDesc()
    d.x, d.out, d.wstream, d.kvstream, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), wstream.data_ptr(), kvstream.data_ptr(), x.stride(0), out.stride(0)
    d.M, d.rows_per_sample, d.C, d.heads, d.ctx_len, d.eps, d.scale = M, rows_per_sample, C, heads, ctx_len, eps, (C // heads) ** -0.5
    with _timed("gemm_kernel", 2.0 * M * C * C + 4.0 * M * ctx_len * C, ("xattn_attn", M, C, heads, ctx_len)):
        check(lib.insv2v_xattn_attn(_byref(d), _stream()), "insv2v_xattn_attn")
    return out


def _conv_geometry(geom, stride, pad, upsample):
    NB, IH, IW = geom
    IHu, IWu = (IH * 2, IW * 2) if upsample else (IH, IW)
    pt, pl = pad
    # PyTorch conv arithmetic; the VAE's asymmetric (0,1,0,1) pad is pad=(0,0) with one extra row/col
    if pt == 0 and stride == 2:
        return (IHu + 1 - 3) // 2 + 1, (IWu + 1 - 3) // 2 + 1
    return (IHu + 2 * pt - 3) // stride + 1, (IWu + 2 * pl - 3) // stride + 1


_FUSE_CACHE = {}


class operand_window:
    """Context manager for tests: insv2v_gemm treats ``nbytes`` as the size of one operand window, so a small problem takes the path of
    an operand beyond 2 GiB (row / image ranges, one launch each).  The product never uses it."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)

    def __enter__(self):
        self.prev = int(_lib.load().insv2v_set_operand_window(self.nbytes))
        _STATS_PARTS_CACHE.clear()   # insv2v_gemm_stats_parts answers for the window in force (a split problem emits none)

    def __exit__(self, *exc):
        _lib.load().insv2v_set_operand_window(self.prev)
        _STATS_PARTS_CACHE.clear()
        return False


def tap_weights(w):
    """[Cout <= 4, Cin, 3, 3] convolution weights -> the [40, Cin] fp16 matrix of conv3x3_narrow: row 4 t + c = w[c, :, ky, kx], t = 3 ky + kx
    (rows of absent output channels and rows 36-39 are zero)."""
    cout, cin = w.shape[:2]
    if cout > 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError(f"tap_weights: a 3x3 convolution with <= 4 output channels, got {tuple(w.shape)}")
    wt = torch.zeros(40, cin, dtype=torch.float32)
    wt[:36].reshape(9, 4, cin)[:, :cout] = w.detach().float().cpu().permute(2, 3, 0, 1).reshape(9, cout, cin)
    return wt.half()


def conv3x3_narrow(x, geom, wtap, bias, cout):
    """Stride-1, pad-1 3x3 convolution with <= 4 output channels as ONE plain GEMM over the input channels (the nine taps' partial outputs
    of every pixel, fp32) + insv2v_tap_gather (the shifted sum): x [NB*H*W, Cin] fp16, wtap = tap_weights(w) on the device -> fp32 [NB*H*W, cout].
    The direct implicit-GEMM form pads the output channels to a 64-wide tile: 16 x the multiply-adds at cout = 4."""
    lib = _lib.load()
    NB, H, W = geom
    y9 = gemm(x, wtap, None, out_fp32=True)
    out = torch.empty((x.shape[0], cout), device=x.device, dtype=torch.float32)
    with _timed("elementwise", 0.0, ("tapgather", x.shape[0], cout)):
        check(lib.insv2v_tap_gather(y9.data_ptr(), y9.stride(0), _ptr(bias), out.data_ptr(), out.stride(0), NB, H, W, cout, _stream()), "insv2v_tap_gather")
    return out


def conv3x3_fuses_groupnorm(geom, cin, cout, k_split=0):
    """True if a stride-1, pad-1 3x3 convolution of this geometry runs on the patch-tiled kernel, which can apply the
    preceding GroupNorm(+SiLU) to its input patch in LDS (conv3x3(gn_ab=...)); asked of the library, cached."""
    key = (tuple(geom), cin, cout, k_split)
    if key not in _FUSE_CACHE:
        lib = _lib.load()
        NB, IH, IW = geom
        d = GemmDesc()
        d.M, d.N, d.K, d.batch, d.mode = NB * IH * IW, cout, 9 * cin, 1, 1
        d.NB, d.IH, d.IW, d.OH, d.OW, d.Cin = NB, IH, IW, IH, IW, cin
        d.stride, d.pad_t, d.pad_l, d.k_split = 1, 1, 1, k_split
        d.workspace, d.workspace_bytes = 1, WORKSPACE_BYTES  # only the decision logic looks at these
        _FUSE_CACHE[key] = bool(lib.insv2v_conv3x3_fuses_groupnorm(_byref(d)))
    return _FUSE_CACHE[key]


def conv3x3(x, geom, w, bias=None, *, x2=None, stride=1, pad=(1, 1), upsample=False, residual=None, row_bias=None,
            rows_per_group=0, out_fp32=False, tile=0, split_k=0, gn_ab=None, gn_images_per_sample=0, gn_silu=False, out=None, act=ACT_NONE):
    """3x3 convolution over channels-last pixels.  x: [NB*IH*IW, C1] (+x2 [.., C2]); w: [N, 9*(C1+C2)];
    geom = (NB, IH, IW).  Returns ([NB*OH*OW, N], (NB, OH, OW)).
    gn_ab ([nsamples, C1+C2, 2] fp32 from groupnorm_stats): x is the RAW tensor and the kernel applies
    act(x*scale + shift) to its input on the fly (only where conv3x3_fuses_groupnorm says so).
    An input beyond the 2 GiB descriptor window of the kernels' LDS-DMA loads (the normalised [1 474 560, 960] concatenation entering
    the first level-0 up block at 20 stacked clips) is convolved by insv2v_gemm in image-aligned parts, one launch each (round 5: the
    split lives behind the C ABI, for the Linear form too)."""
    lib = _lib.load()
    _req(x, torch.float16, "conv.x"), _req(w, torch.float16, "conv.w")
    NB, IH, IW = geom
    pt, pl = pad
    OH, OW = _conv_geometry(geom, stride, pad, upsample)
    N = w.shape[0]
    cin = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    M = NB * OH * OW
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32 if out_fp32 else torch.float16)
    d = GemmDesc()
    d.a, d.w, d.c = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.lda, d.ldw, d.ldc = x.stride(0), w.stride(0), out.stride(0)
    if x2 is not None:
        d.a2, d.lda2, d.k_split = _req(x2, torch.float16, "conv.x2").data_ptr(), x2.stride(0), x.shape[1]
    if bias is not None:
        d.bias = _req(bias, torch.float32, "conv.bias").data_ptr()
    if row_bias is not None:
        d.row_bias, d.ld_rb, d.rows_per_group = _req(row_bias, torch.float32, "conv.row_bias").data_ptr(), row_bias.stride(0), rows_per_group
    if residual is not None:
        d.residual, d.ldr = _req(residual, torch.float16, "conv.residual").data_ptr(), residual.stride(0)
    d.M, d.N, d.K, d.c_fp32, d.alpha, d.tile, d.batch, d.act = M, N, 9 * cin, int(out_fp32), 1.0, tile, 1, act
    d.mode, d.NB, d.IH, d.IW, d.OH, d.OW, d.Cin = 1, NB, IH, IW, OH, OW, cin
    d.stride, d.pad_t, d.pad_l, d.upsample = stride, pt, pl, int(upsample)
    if gn_ab is not None:
        d.gn_ab = _req(gn_ab, torch.float32, "conv.gn_ab").data_ptr()
        d.gn_images_per_sample, d.gn_silu = gn_images_per_sample, int(gn_silu)
    ws = _workspace(x.device)
    d.workspace, d.workspace_bytes, d.split_k = ws.data_ptr(), ws.numel() * 4, split_k
    with _timed("gemm_kernel", 2.0 * M * N * 9 * cin, ("conv", M, N, 9 * cin, stride, int(upsample), residual is not None)):
        check(lib.insv2v_gemm(_byref(d), _stream()), "insv2v_gemm(conv3x3)")
    return out, (NB, OH, OW)


def _gn_chunks(nsamples, rows_per_sample):
    """Chunks per sample of the GroupNorm statistics pass.  (More, smaller chunks - up to 512 per sample - were A/B-ed in round 2
    and lost 0.2 ms per UNet step to the longer finalize; the pass itself now keeps four 16-byte loads per thread in flight.)"""
    return max(1, min(rows_per_sample // 32, max(1, 1024 // nsamples), 128))


# ---- Winograd F(2x2, 3x3) convolution (insv2v_winograd_input -> grouped insv2v_gemm -> insv2v_winograd_output; csrc/winograd.hip)
_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def winograd_weights(w, device, upsample=False):
    """[Cout, Cin, 3, 3] fp32 convolution weights -> U [16, Cout, Cin] fp16, U[i*4 + j] = (G g G^T)[i][j] (computed in fp32, rounded once).
    upsample: the 9 matrices of the nearest-x2-upsample form, patch indices (i, j) in {0, 1, 3}^2 (the others meet an all-zero V)."""
    u = torch.einsum("ai,ocij,bj->aboc", _WINO_G, w.float(), _WINO_G)   # [4, 4, Cout, Cin]
    if upsample:
        u = u[[0, 1, 3]][:, [0, 1, 3]]
    return u.reshape(-1, w.shape[0], w.shape[1]).to(device=device, dtype=torch.float16).contiguous()


def winograd_ok(geom, cin, c1=0, upsample=False):
    """Shapes insv2v_winograd_input accepts (else the caller uses conv3x3): even H, W (any with upsample: one tile per input pixel); image
    rows of at most 128 pixels (an image's 64-channel slice is staged in LDS whole or in bands of tile rows)."""
    _, H, W = geom
    return (upsample or (H % 2 == 0 and W % 2 == 0)) and W <= 128 and cin % 64 == 0 and c1 % 64 == 0


def winograd_conv3x3(x, geom, U, bias=None, *, x2=None, gn_ab=None, gn_images_per_sample=0, gn_silu=False, row_bias=None, rows_per_group=0,
                     residual=None, out=None, tile=0, upsample=False):
    """Stride-1, pad-1 3x3 convolution over channels-last pixels in Winograd form: x [NB*H*W, C1] (+ x2 [.., C2]), U = winograd_weights(w)
    [16, Cout, C1 + C2]; optional GroupNorm (+SiLU) of the input from the (scale, shift) table gn_ab (groupnorm_stats).  Returns [NB*H*W, Cout]."""
    lib = _lib.load()
    _req(x, torch.float16, "winograd.x"), _req(U, torch.float16, "winograd.U")
    NB, H, W = geom
    C1 = x.shape[1]
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    Cout = U.shape[1]
    ng = 9 if upsample else 16                     # transformed taps that are not identically zero
    OH, OW = (2 * H, 2 * W) if upsample else (H, W)
    assert U.shape[0] == ng and U.shape[2] == C and x.shape[0] == NB * H * W
    tiles = NB * H * W if upsample else NB * (H // 2) * (W // 2)
    grows = -(-tiles // 256) * 256
    v = torch.empty((ng * grows, C), device=x.device, dtype=torch.float16)
    di = WinogradInDesc()
    di.x, di.v, di.ldx, di.v_group_rows = x.data_ptr(), v.data_ptr(), x.stride(0), grows
    if x2 is not None:
        di.x2, di.ldx2, di.C1 = _req(x2, torch.float16, "winograd.x2").data_ptr(), x2.stride(0), C1
    di.NB, di.H, di.W, di.C, di.upsample = NB, H, W, C, int(upsample)
    if gn_ab is not None:
        di.gn_ab, di.gn_images_per_sample, di.gn_silu = _req(gn_ab, torch.float32, "winograd.gn_ab").data_ptr(), gn_images_per_sample, int(gn_silu)
    with _timed("groupnorm", 0.0, ("wino_in", NB * H * W, C)):
        check(lib.insv2v_winograd_input(_byref(di), _stream()), "insv2v_winograd_input")
    m = torch.empty((ng * grows, Cout), device=x.device, dtype=torch.float16)
    d = GemmDesc()
    d.a, d.w, d.c = v.data_ptr(), U.data_ptr(), m.data_ptr()
    d.lda, d.ldw, d.ldc = C, C, Cout
    d.M, d.N, d.K, d.batch, d.alpha, d.tile = ng * grows, Cout, C, 1, 1.0, tile
    d.w_group_rows, d.w_group_stride = grows, Cout * C
    # (recorded work = the ALGORITHMIC work of the convolution, 2 * pixels * Cout * 9 * C as SURVEY.md 8d counts it; the launch performs
    #  4 / 9 of those multiply-adds - the tag carries the launched product's shape)
    with _timed("gemm_kernel", 2.0 * NB * OH * OW * Cout * 9 * C, ("wino_gemm", ng * grows, Cout, C)):
        check(lib.insv2v_gemm(_byref(d), _stream()), "insv2v_gemm (grouped)")
    del v
    if out is None:
        out = torch.empty((NB * OH * OW, Cout), device=x.device, dtype=torch.float16)
    do = WinogradOutDesc()
    do.m, do.y, do.m_group_rows, do.ldy = m.data_ptr(), out.data_ptr(), grows, out.stride(0)
    if bias is not None:
        do.bias = _req(bias, torch.float32, "winograd.bias").data_ptr()
    if row_bias is not None:
        do.row_bias, do.ld_rb, do.rows_per_group = _req(row_bias, torch.float32, "winograd.row_bias").data_ptr(), row_bias.stride(0), rows_per_group
    if residual is not None:
        do.residual, do.ldr = _req(residual, torch.float16, "winograd.residual").data_ptr(), residual.stride(0)
    do.NB, do.H, do.W, do.Cout, do.upsample = NB, H, W, Cout, int(upsample)
    with _timed("groupnorm", 0.0, ("wino_out", NB * OH * OW, Cout)):
        check(lib.insv2v_winograd_output(_byref(do), _stream()), "insv2v_winograd_output")
    return out


def groupnorm(x, nsamples, rows_per_sample, gamma, beta, groups, eps, silu=False, x2=None):
    """GroupNorm(+SiLU) of channels-last tokens; nsamples*rows_per_sample == x.shape[0].
    (Normalising in groups of samples so that the apply pass re-reads what the statistics pass just streamed was measured and is
    slower: profiles/r04_groupnorm_sample_groups_experiment.txt.)"""
    lib = _lib.load()
    _req(x, torch.float16, "groupnorm.x")
    C1 = x.shape[1]
    Ct = C1 + (x2.shape[1] if x2 is not None else 0)
    assert nsamples * rows_per_sample == x.shape[0]
    y = torch.empty((x.shape[0], Ct), device=x.device, dtype=torch.float16)
    nchunks = _gn_chunks(nsamples, rows_per_sample)
    # stats [nsamples,G,2] + partials [nsamples,nchunks,G,3]; allocated per call so graph capture owns it
    scratch = torch.empty(nsamples * groups * (2 + 3 * nchunks), device=x.device, dtype=torch.float32)
    d = GroupNormDesc()
    d.x, d.y, d.gamma, d.beta, d.partials = x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), scratch.data_ptr()
    d.ldx, d.ldy = x.stride(0), y.stride(0)
    if x2 is not None:
        d.x2, d.ldx2, d.C1 = _req(x2, torch.float16, "groupnorm.x2").data_ptr(), x2.stride(0), C1
    d.nsamples, d.rows_per_sample, d.C, d.G, d.nchunks, d.silu, d.eps = nsamples, rows_per_sample, Ct, groups, nchunks, int(silu), eps
    with _timed("groupnorm", 0.0, ("gn", nsamples, rows_per_sample, Ct)):
        check(lib.insv2v_groupnorm(_byref(d), _stream()), "insv2v_groupnorm")
    return y


def groupnorm_stats(x, nsamples, rows_per_sample, gamma, beta, groups, eps, x2=None):
    """GroupNorm statistics only -> ab [nsamples, C, 2] fp32 = (rstd*gamma, beta - mean*rstd*gamma) per channel, for a
    consumer that normalises on the fly (conv3x3(gn_ab=...)): one read of the tensor, no normalised copy."""
    lib = _lib.load()
    _req(x, torch.float16, "groupnorm.x")
    C1 = x.shape[1]
    Ct = C1 + (x2.shape[1] if x2 is not None else 0)
    assert nsamples * rows_per_sample == x.shape[0]
    nchunks = _gn_chunks(nsamples, rows_per_sample)
    scratch = torch.empty(nsamples * groups * (2 + 3 * nchunks), device=x.device, dtype=torch.float32)
    ab = torch.empty((nsamples, Ct, 2), device=x.device, dtype=torch.float32)
    d = GroupNormDesc()
    d.x, d.gamma, d.beta, d.partials, d.ab = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), scratch.data_ptr(), ab.data_ptr()
    d.ldx, d.stats_only = x.stride(0), 1
    if x2 is not None:
        d.x2, d.ldx2, d.C1 = _req(x2, torch.float16, "groupnorm.x2").data_ptr(), x2.stride(0), C1
    d.nsamples, d.rows_per_sample, d.C, d.G, d.nchunks, d.eps = nsamples, rows_per_sample, Ct, groups, nchunks, eps
    with _timed("groupnorm", 0.0, ("gnstats", nsamples, rows_per_sample, Ct)):
        check(lib.insv2v_groupnorm(_byref(d), _stream()), "insv2v_groupnorm(stats_only)")
    return ab


def layernorm(x, gamma, beta, eps=1e-5, pe=None, rows_per_frame=0, frames=0, pe_start=0):
    lib = _lib.load()
    _req(x, torch.float16, "layernorm.x")
    y = torch.empty_like(x)
    d = LayerNormDesc()
    d.x, d.y, d.gamma, d.beta = x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    d.ldx, d.ldy, d.rows, d.C, d.eps = x.stride(0), y.stride(0), x.shape[0], x.shape[1], eps
    if pe is not None:
        d.pe, d.rows_per_frame, d.frames, d.pe_start = _req(pe, torch.float32, "layernorm.pe").data_ptr(), rows_per_frame, frames, pe_start
    with _timed("layernorm", 0.0, ("ln", x.shape[0], x.shape[1])):
        check(lib.insv2v_layernorm(_byref(d), _stream()), "insv2v_layernorm")
    return y


def layernorm_stats(x, eps=1e-5):
    """(mean, rstd) per token row -> fp32 [rows, 2]; consumed by gemm(row_stats=...)."""
    lib = _lib.load()
    _req(x, torch.float16, "layernorm_stats.x")
    stats = torch.empty((x.shape[0], 2), device=x.device, dtype=torch.float32)
    with _timed("layernorm", 0.0, ("lnstats", x.shape[0], x.shape[1])):
        check(lib.insv2v_layernorm_stats(x.data_ptr(), stats.data_ptr(), x.stride(0), x.shape[0], x.shape[1], eps, _stream()),
              "insv2v_layernorm_stats")
    return stats


def attention(q_ptr, k_ptr, v_ptr, out, *, batch, heads, head_dim, seq_q, seq_k, scale,
              q_rs, k_rs, v_rs, o_rs, q_addr, kv_addr, o_addr, causal=False, qkv_bias=None):
    """*_addr = (inner, outer_stride, step): base offset of problem z = (z//inner)*outer + (z%inner)*step.
    qkv_bias: fp16 table [seq, 3 * heads * head_dim] added to the q | k | v rows of sequence position i as they are loaded (the temporal
    positional encoding pushed through the projections); only the <= 16-row form (attention_short_supported) takes it."""
    lib = _lib.load()
    d = AttentionDesc()
    if qkv_bias is not None:
        _req(qkv_bias, torch.float16, "attention.qkv_bias")
        cq = heads * head_dim
        if qkv_bias.shape[0] < max(seq_q, seq_k) or qkv_bias.shape[1] != 3 * cq:
            raise _lib.HipKernelError(f"attention: bias table {tuple(qkv_bias.shape)} for {max(seq_q, seq_k)} rows x 3 x {cq} columns")
        bp = qkv_bias.data_ptr()
        d.q_bias, d.k_bias, d.v_bias, d.bias_rs = bp, bp + 2 * cq, bp + 4 * cq, qkv_bias.stride(0)
    d.q, d.k, d.v, d.o = q_ptr, k_ptr, v_ptr, out.data_ptr()
    d.q_rs, d.k_rs, d.v_rs, d.o_rs = q_rs, k_rs, v_rs, o_rs
    d.q_inner, d.q_outer, d.q_step = q_addr
    d.kv_inner, d.kv_outer, d.kv_step = kv_addr
    d.o_inner, d.o_outer, d.o_step = o_addr
    d.batch, d.heads, d.head_dim, d.seq_q, d.seq_k, d.scale, d.causal = batch, heads, head_dim, seq_q, seq_k, scale, int(causal)
    with _timed("attn_kernel", 4.0 * batch * heads * seq_q * seq_k * head_dim, ("attn", batch, heads, head_dim, seq_q, seq_k)):
        check(lib.insv2v_attention(_byref(d), _stream()), "insv2v_attention")
    return out


def attention_short_supported(heads, head_dim, seq):
    """True if insv2v_attention runs the one-wave-per-head kernel for this problem (the only one that takes qkv_bias)."""
    return seq <= 16 and heads <= 16 and head_dim in (16, 32, 40, 64, 80, 128, 160) and heads * ((head_dim + 15) // 16) * 16 * 20 * 2 <= 64 * 1024


def embed_tokens(ids, tok, pos):
    """CLIP text embeddings: ids int64 [n, L] (device) -> fp16 [n*L, C] = tok[ids] + pos[position]."""
    lib = _lib.load()
    _req(tok, torch.float16, "embed.tok")
    _req(pos, torch.float16, "embed.pos")
    if ids.dtype != torch.int64 or not ids.is_cuda or not ids.is_contiguous():
        raise _lib.HipKernelError("embed.ids must be a contiguous int64 device tensor")
    n, L = ids.shape
    if L > pos.shape[0]:
        raise ValueError(f"Sequence length must be less than max_position_embeddings (got {L} > {pos.shape[0]})")
    out = torch.empty((n * L, tok.shape[1]), device=tok.device, dtype=torch.float16)
    check(lib.insv2v_embed_tokens(ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), out.data_ptr(), n * L, L, tok.shape[1],
                                  tok.shape[0], _stream()), "insv2v_embed_tokens")
    return out


def softmax_rows(x, scale=1.0):
    lib = _lib.load()
    _req(x, torch.float16, "softmax.x")
    x2 = x.reshape(-1, x.shape[-1])
    check(lib.insv2v_softmax_rows(x2.data_ptr(), x2.data_ptr(), x2.stride(0), x2.stride(0), x2.shape[0], x2.shape[1],
                                  scale, _stream()), "insv2v_softmax_rows")
    return x


def timestep_embedding(t_dev, dim, shift=0.0):
    lib = _lib.load()
    _req(t_dev, torch.float32, "timestep")
    out = torch.empty((t_dev.shape[0], dim), device=t_dev.device, dtype=torch.float16)
    check(lib.insv2v_timestep_embedding(t_dev.data_ptr(), out.data_ptr(), t_dev.shape[0], dim, shift, _stream()),
          "insv2v_timestep_embedding")
    return out


def build_unet_input(latent, img_cond, out, t_out, timestep, nbranch, branch_rows=0, t_stride=0):
    """branch_rows / t_stride: rows of ``out`` / entries of ``t_out`` between consecutive branches (0: back to back) - the branch-major
    layout of a stack of clips (inference._stacked_gen)."""
    lib = _lib.load()
    F, _, h, w = latent.shape[-4:]
    check(lib.insv2v_build_unet_input(_req(latent, torch.float32, "latent").data_ptr(),
                                      _req(img_cond, torch.float32, "img_cond").data_ptr(), out.data_ptr(),
                                      _ptr(t_out), float(timestep), nbranch, F, h, w, out.shape[-1], branch_rows, t_stride, _stream()),
          "insv2v_build_unet_input")
    return out


def cfg_step(eps_in, latent, *, nbranch, text_cfg=1.0, img_cfg=1.0, sqrt_a=1.0, sqrt_1ma=0.0, coef=(0, 0, 0, 0),
             latent_out=None, pred_x0=None, eps_out=None, latent_ref=None, correct=0, delta_q=None, noise=None,
             rescale_stats=None, guidance_rescale=0.0, branch_stride=0):
    lib = _lib.load()
    F, _, h, w = latent.shape[-4:]
    d = StepDesc()
    d.eps_in, d.latent = _req(eps_in, torch.float32, "eps_in").data_ptr(), _req(latent, torch.float32, "latent").data_ptr()
    d.latent_ref, d.delta_q, d.noise, d.rescale_stats = _ptr(latent_ref), _ptr(delta_q), _ptr(noise), _ptr(rescale_stats)
    d.latent_out, d.pred_x0, d.eps_out = _ptr(latent_out), _ptr(pred_x0), _ptr(eps_out)
    d.nbranch, d.F, d.h, d.w, d.correct = nbranch, F, h, w, correct
    d.R = latent_ref.shape[-4] if latent_ref is not None else 0
    d.text_cfg, d.img_cfg, d.sqrt_a, d.sqrt_1ma = text_cfg, img_cfg, sqrt_a, sqrt_1ma
    d.c_x0, d.c_eps, d.c_xt, d.c_noise = coef
    d.guidance_rescale, d.branch_stride = guidance_rescale, branch_stride
    check(lib.insv2v_cfg_step(_byref(d), _stream()), "insv2v_cfg_step")


def cfg_stats(eps_in, stats, F, h, w, text_cfg, img_cfg, branch_stride=0):
    lib = _lib.load()
    check(lib.insv2v_cfg_stats(eps_in.data_ptr(), stats.data_ptr(), F, h, w, text_cfg, img_cfg, branch_stride, _stream()), "insv2v_cfg_stats")


def warp_image(image, flow):
    lib = _lib.load()
    _req(image, torch.float32, "warp.image"), _req(flow, torch.float32, "warp.flow")
    image, flow = image.contiguous(), flow.contiguous()
    N, Cc, H, W = image.shape
    out = torch.empty_like(image)
    check(lib.insv2v_warp_image(image.data_ptr(), flow.data_ptr(), out.data_ptr(), N, Cc, H, W, _stream()), "insv2v_warp_image")
    return out


def resize_flow(flow, size):
    lib = _lib.load()
    _req(flow, torch.float32, "resize_flow.flow")
    flow = flow.contiguous()
    N, _, h, w = flow.shape
    H, W = size
    out = torch.empty((N, 2, H, W), device=flow.device, dtype=torch.float32)
    check(lib.insv2v_resize_flow(flow.data_ptr(), out.data_ptr(), N, h, w, H, W, _stream()), "insv2v_resize_flow")
    return out


def flow_correction(eps_cfg, latent, latent_ref, flows, sqrt_a, sqrt_1ma):
    lib = _lib.load()
    F, _, h, w = latent.shape[-4:]
    R = latent_ref.shape[-4]
    out = torch.empty((F - R, 4, h, w), device=latent.device, dtype=torch.float32)
    check(lib.insv2v_flow_correction(eps_cfg.data_ptr(), latent.data_ptr(), latent_ref.data_ptr(), flows.data_ptr(),
                                     out.data_ptr(), F, R, h, w, sqrt_a, sqrt_1ma, _stream()), "insv2v_flow_correction")
    return out


def nchw_to_nhwc_f16(x, ldo, scale=1.0):
    lib = _lib.load()
    _req(x, torch.float32, "nchw_to_nhwc.x")
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N * H * W, ldo), device=x.device, dtype=torch.float16)
    check(lib.insv2v_nchw_to_nhwc_f16(x.data_ptr(), out.data_ptr(), N, Cc, H, W, ldo, scale, _stream()), "insv2v_nchw_to_nhwc_f16")
    return out


def nhwc_to_nchw_f32(x, N, Cc, H, W, scale=1.0):
    lib = _lib.load()
    out = torch.empty((N, Cc, H, W), device=x.device, dtype=torch.float32)
    check(lib.insv2v_nhwc_to_nchw_f32(x.data_ptr(), int(x.dtype == torch.float32), out.data_ptr(), N, Cc, H, W,
                                      x.stride(0), scale, _stream()), "insv2v_nhwc_to_nchw_f32")
    return out


def posterior_sample(moments, noise, N, H, W, scale):
    lib = _lib.load()
    _req(moments, torch.float32, "posterior.moments"), _req(noise, torch.float32, "posterior.noise")
    z = torch.empty((N, 4, H, W), device=moments.device, dtype=torch.float32)
    check(lib.insv2v_posterior_sample(moments.data_ptr(), noise.contiguous().data_ptr(), z.data_ptr(), N, H, W,
                                      moments.stride(0), scale, _stream()), "insv2v_posterior_sample")
    return z


# ----------------------------------------------------------------------------- optical-flow estimator (insv2v/raft.py)
def im2col(x, geom, C, kh, kw, stride=1, pad=(0, 0), x2=None):
    """x fp16 [N*IH*IW, C1] (+ x2 [.., C - C1]) -> fp16 [N*OH*OW, kh*kw*C] and the output geometry (N, OH, OW)."""
    lib = _lib.load()
    _req(x, torch.float16, "im2col.x")
    N, IH, IW = geom
    OH, OW = (IH + 2 * pad[0] - kh) // stride + 1, (IW + 2 * pad[1] - kw) // stride + 1
    out = torch.empty((N * OH * OW, kh * kw * C), device=x.device, dtype=torch.float16)
    d = _lib.Im2colDesc()
    d.x, d.out, d.ldx, d.ldo = x.data_ptr(), out.data_ptr(), x.stride(0), out.stride(0)
    if x2 is not None:
        _req(x2, torch.float16, "im2col.x2")
        d.x2, d.ldx2, d.C1 = x2.data_ptr(), x2.stride(0), C - x2.shape[1]
    else:
        d.C1 = C
    d.N, d.IH, d.IW, d.C, d.KH, d.KW = N, IH, IW, C, kh, kw
    d.stride_h = d.stride_w = stride
    d.pad_h, d.pad_w, d.OH, d.OW = pad[0], pad[1], OH, OW
    with _timed("im2col", 0.0, ("im2col", N * OH * OW, kh * kw * C)):
        check(lib.insv2v_im2col(_byref(d), _stream()), "insv2v_im2col")
    return out, (N, OH, OW)


def instance_norm(x, N, HW, relu=False, eps=1e-5):
    """nn.InstanceNorm2d (no affine) (+ ReLU) over channels-last fp16 [N*HW, C]; returns a new tensor."""
    lib = _lib.load()
    _req(x, torch.float16, "instance_norm.x")
    Cc = x.shape[1]
    nchunks = max(1, min(64, HW // 64))
    part = torch.empty((N * nchunks * Cc * 2,), device=x.device, dtype=torch.float32)
    y = torch.empty((x.shape[0], Cc), device=x.device, dtype=torch.float16)
    with _timed("instnorm", 0.0, ("instnorm", N, HW, Cc)):
        check(lib.insv2v_instance_norm(x.data_ptr(), y.data_ptr(), part.data_ptr(), N, HW, Cc, x.stride(0), y.stride(0), nchunks, eps,
                                       int(relu), _stream()), "insv2v_instance_norm")
    return y


def ew(op, a, b=None, c=None, out=None):
    """Element-wise op on fp16 [rows, C] views (column slices of wider buffers are fine): see INSV2V_EW_* in the header."""
    lib = _lib.load()
    _req(a, torch.float16, "ew.a")
    rows, Cc = a.shape
    if out is None:
        out = torch.empty((rows, Cc), device=a.device, dtype=torch.float16)
    check(lib.insv2v_ew(op, a.data_ptr(), _ptr(b), _ptr(c), out.data_ptr(), rows, Cc, a.stride(0), b.stride(0) if b is not None else 0,
                        c.stride(0) if c is not None else 0, out.stride(0), _stream()), "insv2v_ew")
    return out


def avgpool2x2(x, n, h, w):
    lib = _lib.load()
    _req(x, torch.float32, "avgpool.x")
    y = torch.empty((n, h // 2, w // 2), device=x.device, dtype=torch.float32)
    check(lib.insv2v_avgpool2x2(x.data_ptr(), y.data_ptr(), n, h, w, _stream()), "insv2v_avgpool2x2")
    return y


def corr_lookup(pyramid, coords, B, h, w, radius, ldo):
    lib = _lib.load()
    out = torch.empty((B * h * w, ldo), device=coords.device, dtype=torch.float16)
    d = _lib.CorrLookupDesc()
    ptrs = [t.data_ptr() for t in pyramid] + [None] * (4 - len(pyramid))
    d.pyr0, d.pyr1, d.pyr2, d.pyr3 = ptrs
    d.coords, d.out, d.ldo = _req(coords, torch.float32, "corr_lookup.coords").data_ptr(), out.data_ptr(), ldo
    d.B, d.h, d.w, d.levels, d.radius = B, h, w, len(pyramid), radius
    with _timed("corr_lookup", 0.0, ("corr_lookup", B * h * w, ldo)):
        check(lib.insv2v_corr_lookup(_byref(d), _stream()), "insv2v_corr_lookup")
    return out


def raft_flow_rows(coords1, delta, rows, B, h, w):
    """coords1 [B,2,h,w] fp32 += delta rows (fp32 [B*h*w, ld], or None); rows (fp16 [B*h*w, n] view or None) <- coords1 - pixel grid."""
    lib = _lib.load()
    check(lib.insv2v_raft_flow_rows(coords1.data_ptr(), _ptr(delta), delta.stride(0) if delta is not None else 0, _ptr(rows),
                                    rows.stride(0) if rows is not None else 0, rows.shape[1] if rows is not None else 0, B, h, w, _stream()),
          "insv2v_raft_flow_rows")


def convex_upsample(coords1, mask, B, h, w):
    lib = _lib.load()
    _req(mask, torch.float16, "convex_upsample.mask")
    out = torch.empty((B, 2, 8 * h, 8 * w), device=coords1.device, dtype=torch.float32)
    check(lib.insv2v_convex_upsample(coords1.data_ptr(), mask.data_ptr(), mask.stride(0), out.data_ptr(), B, h, w, _stream()),
          "insv2v_convex_upsample")
    return out
