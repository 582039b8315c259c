"""Kernel-level parity: every C-ABI entry point against a plain PyTorch fp32 reference of the same
op, on the GPU, through the same ctypes binding the product uses.

Tolerances (fp16 storage, fp32 accumulate): GEMM/conv outputs are compared against an fp32
computation on the SAME fp16-rounded operands, so the only differences are accumulation order and
the final fp16 rounding: |err| <= 2e-3 * max|ref| + 2e-3.  Norm/attention kernels carry extra
fp16 roundings of intermediates (P in fp16): 4e-3 relative to max|ref|.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(shape, generator=g) * scale).to(dev())


def close(out, ref, rel=2e-3, abs_=2e-3, what=""):
    out, ref = out.float(), ref.float()
    err = (out - ref).abs().max().item()
    tol = rel * ref.abs().max().item() + abs_
    assert math.isfinite(err) and err <= tol, f"{what}: max err {err:.4g} > tol {tol:.4g} (ref max {ref.abs().max().item():.4g})"


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,tile", [(128, 128, 64, 1), (256, 320, 320, 0), (1000, 136, 72, 0), (77, 64, 768, 4),
                                         (384, 640, 1280, 2), (512, 192, 128, 3), (3, 1280, 320, 0), (4608, 1280, 1280, 0)])
def test_gemm_plain(M, N, K, tile):
    from insv2v import ops
    a, w, b = rnd(M, K).half(), rnd(N, K, scale=K ** -0.5).half(), rnd(N)
    out = ops.gemm(a, w, b, tile=tile)
    ref = a.float() @ w.float().t() + b
    close(out, ref, what=f"gemm {M}x{N}x{K} tile{tile}")


def test_gemm_transpose_detecting():
    """A = identity-like, asymmetric W: catches row/col swaps in the MFMA C layout."""
    from insv2v import ops
    M = N = K = 128
    a = torch.eye(M, device=dev()).half()
    w = (torch.arange(N * K, device=dev()).reshape(N, K).float() % 97 / 97).half()
    out = ops.gemm(a, w)
    close(out, w.float().t(), what="gemm identity")


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9, 31, 35, 45, 36, 38])
def test_gemm_epilogues(tile):
    from insv2v import ops
    M, N, K = 320, 256, 192
    a, w, b = rnd(M, K).half(), rnd(N, K, scale=K ** -0.5).half(), rnd(N)
    res = rnd(M, N, seed=3).half()
    rb = rnd(4, N, seed=5)
    out = ops.gemm(a, w, b, residual=res, row_bias=rb, rows_per_group=80, tile=tile, alpha=0.5)
    ref = 0.5 * (a.float() @ w.float().t()) + b + rb.repeat_interleave(80, 0) + res.float()
    close(out, ref, what="gemm bias+rowbias+residual")
    out = ops.gemm(a, w, b, act=ops.ACT_SILU, out_fp32=True, tile=tile)
    close(out, F.silu(a.float() @ w.float().t() + b), what="gemm silu fp32-out")
    assert out.dtype == torch.float32


@pytest.mark.parametrize("tile", [0, 1, 2, 5, 6, 8, 36])
def test_gemm_geglu(tile):
    from insv2v import ops
    from insv2v.unet import interleave32
    M, C = 200, 64
    a = rnd(M, C).half()
    w = rnd(8 * C, C, scale=C ** -0.5)
    b = rnd(8 * C, seed=2)
    wi, bi = interleave32(w).half().contiguous(), interleave32(b).contiguous()
    res = rnd(M, 4 * C, seed=9).half()
    out = ops.gemm(a, wi, bi, act=ops.ACT_GEGLU, residual=res, tile=tile)
    y = a.float() @ w.half().float().t() + b
    h, g = y.chunk(2, dim=-1)
    close(out, h * F.gelu(g) + res.float(), what="gemm geglu")
    assert out.shape == (M, 4 * C)


@pytest.mark.parametrize("M,N,K,res,tile", [(1000, 320, 320, True, 0), (4608, 1280, 640, False, 0), (73728, 320, 320, True, 0), (520, 640, 320, True, 4),
                                            (300, 320, 1280, True, 5), (256, 64, 64, False, 7), (1152, 1280, 1280, True, 0), (384, 320, 128, False, 8),
                                            (8192 + 300, 1280, 1280, True, 240), (9000, 640, 640, False, 240), (46080, 1280, 1280, True, 0)])
def test_gemm_emits_layernorm_statistics(M, N, K, res, tile):
    """Producer-side LayerNorm statistics: an N = C GEMM's epilogue writes per-row partial (sum, sum of squares) of the fp16
    values it stores, one pair per column tile; they must reproduce the statistics pass over the output, and a folded-LayerNorm
    consumer fed with them must equal one fed by insv2v_layernorm_stats - on the tile kernel AND on the persistent kernels
    (which get finished (mean, rstd) pairs from the internal finalize launch)."""
    from insv2v import ops
    a, w, b = (rnd(M, K) * 1.5 + 0.3).half(), rnd(N, K, scale=K ** -0.5).half(), rnd(N) + 0.5
    r = rnd(M, N, seed=3).half() if res else None
    out, st = ops.gemm(a, w, b, residual=r, emit_stats=True, tile=tile)
    assert isinstance(st, ops.RowStats) and st.parts.shape[1:] == (M, 2) and st.nparts == st.parts.shape[0]
    r8 = tile == 240 or (M, N) == (46080, 1280)   # the 256x320 ping-pong kernel (forced / by dispatch): 160-column parts, sums taken BEFORE the fp16 rounding
    if r8:
        assert st.nparts == N // 160, f"expected gemm_r8's {N // 160} parts, got {st.nparts}"
    close(out, a.float() @ w.float().t() + b + (r.float() if res else 0), what="producer output")
    s = st.parts.double().sum(0)
    x = out.double()
    # the sums the kernel documents: of the fp16 values stored (128x128 tile) / of the fp32 values before that rounding (gemm_r8)
    xs = (a.double() @ w.double().t() + b.double() + (r.double() if res else 0)) if r8 else x
    close(s[:, 0], xs.sum(1), rel=1e-5, abs_=1e-3, what="sum")
    close(s[:, 1], (xs * xs).sum(1), rel=1e-5, abs_=1e-3, what="sum of squares")
    # consumers: folded LayerNorm from the partial sums == from the statistics pass
    for N2, act, t2 in ((N, ops.ACT_NONE, 0), (3 * N if N <= 640 else N, ops.ACT_NONE, 0), (N, ops.ACT_NONE, 4)):
        w2 = rnd(N2, N, scale=N ** -0.5, seed=7).half()
        cs, b2 = w2.float().sum(1).contiguous(), rnd(N2, seed=8)
        got = ops.gemm(out, w2, b2, row_stats=st, col_sum=cs, act=act, tile=t2)
        ref = ops.gemm(out, w2, b2, row_stats=ops.layernorm_stats(out, 1e-5), col_sum=cs, act=act, tile=t2)
        close(got, ref, rel=1e-3, abs_=1e-3, what=f"consumer N2={N2} tile={t2}")
        xn = (x - x.mean(1, keepdim=True)) * (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        close(got, xn.float() @ w2.float().t() + b2, rel=4e-3, abs_=4e-3, what=f"consumer vs fp32 LayerNorm N2={N2}")


@pytest.mark.parametrize("tile", [5, 240])
@pytest.mark.parametrize("offset", [0.0, 8.0, 30.0])
def test_producer_layernorm_statistics_with_large_row_offsets(offset, tile):
    """the producer-emitted LayerNorm statistics use the single-pass form var = E[x^2] - mean^2 in fp32 (tile epilogue partial
    sums, ln_finalize, the row kernels' stats_out), where the statistics pass is two-pass and centred.  Residual-stream rows with |mean| >> std
    lose precision to cancellation: this pins the loss.  Rows of standard deviation ~1 around `offset` (fp16 itself resolves |mean| / std up to
    ~2048): finished (mean, rstd) from the partial sums against the two-pass kernel - stated tolerance on rstd 2e-4 relative at offset 0,
    1e-3 at 8, 1e-2 at 30 (E[x^2] ~ 900: fp32 rounding of the sums ~ 1e-4 absolute on a variance of ~1)."""
    from insv2v import ops
    M, N, K = 4096, 320, 320
    a, w = rnd(M, K).half(), rnd(N, K, scale=K ** -0.5).half()
    b = torch.full((N,), offset, device=dev()) + 0.1 * rnd(N, seed=2)
    out, st = ops.gemm(a, w, b, emit_stats=True, tile=tile)   # 5: the 128x128 tile's epilogue; 240: gemm_r8's in-lane sums (160-column parts)
    assert isinstance(st, ops.RowStats) and st.nparts == (3 if tile == 5 else 2)
    ref = ops.layernorm_stats(out, 1e-5)
    if tile == 240:   # gemm_r8 sums the values BEFORE their fp16 rounding (include/insv2v_hip.h): the exact statistics of those are the yardstick
        xs = a.double() @ w.double().t() + b.double()
        mu = xs.mean(1)
        yard = torch.stack([mu, (xs.var(1, unbiased=False) + 1e-5).rsqrt()], 1).float()
    else:
        yard = ref
    s = st.parts.double().sum(0)
    mean = s[:, 0] / N
    rstd = (s[:, 1] / N - mean * mean + 1e-5).clamp_min(1e-12).rsqrt()    # what ln_finalize computes, in fp64 from the fp32 partials
    tol = {0.0: 2e-4, 8.0: 1e-3, 30.0: 1e-2}[offset]
    assert ((mean.float() - yard[:, 0]).abs() / yard[:, 0].abs().clamp_min(1.0)).max().item() <= 1e-5
    err = ((rstd.float() - yard[:, 1]).abs() / yard[:, 1]).max().item()
    print(f"[parity] producer LayerNorm statistics (tile {tile}), rows around {offset}: max relative rstd error {err:.3e}")
    assert err <= tol, f"rstd from single-pass partial sums off by {err:.3e} at row offset {offset} (tolerance {tol})"
    # and the consumer that finalises them itself (fp32 in the kernel): a folded-LayerNorm GEMM fed with the partials vs the statistics pass
    w2 = rnd(N, N, scale=N ** -0.5, seed=7).half()
    cs = w2.float().sum(1).contiguous()
    got = ops.gemm(out, w2, None, row_stats=st, col_sum=cs)
    want = ops.gemm(out, w2, None, row_stats=ref, col_sum=cs)
    close(got, want, rel=10 * tol, abs_=10 * tol, what=f"folded-LayerNorm consumer at row offset {offset}")


@pytest.mark.parametrize("M", [128, 1000, 73728 + 17])
def test_ffn_fused_vs_fp32(M):
    """insv2v_ffn_fused (C = 320: LayerNorm -> Linear(320, 2560) -> h * gelu_erf(g) -> Linear(1280, 320) -> + x in one register-resident
    kernel) against fp32 torch on the same fp16-rounded operands; ragged last row tile; and against the two-GEMM path."""
    from insv2v import ops
    from insv2v.fused import pack_ffn_stream
    from insv2v.unet import fold_layernorm, interleave32
    C, NH = 320, 1280
    x = (rnd(M, C) * 1.3 + 0.2).half()
    w1, b1 = rnd(2 * NH, C, scale=C ** -0.5), rnd(2 * NH, seed=1) * 0.3
    w2, b2 = rnd(C, NH, scale=NH ** -0.5, seed=2).half(), rnd(C, seed=3) * 0.3
    gamma, beta = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    wf, col, bf = fold_layernorm(w1.cpu(), gamma.cpu(), beta.cpu(), b1.cpu())
    stream = pack_ffn_stream(wf.float(), bf, w2.float().cpu(), b2.cpu()).to(dev())
    out = ops.ffn_fused(x, stream, NH)
    xf = x.float()
    xn = (xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    y = xn @ wf.float().to(dev()).t() + bf.to(dev())
    h, g = y.chunk(2, dim=-1)
    ref = (h * F.gelu(g)).half().float() @ w2.float().t() + b2 + xf
    close(out, ref, rel=4e-3, abs_=4e-3, what=f"ffn_fused M={M}")
    # the two-GEMM path the other widths use
    g2 = ops.gemm(x, interleave32(wf).to(dev()), interleave32(bf).to(dev()), act=ops.ACT_GEGLU, row_stats=ops.layernorm_stats(x),
                  col_sum=interleave32(col).to(dev()))
    two = ops.gemm(g2, w2, b2, residual=x)
    close(out, two, rel=4e-3, abs_=4e-3, what=f"ffn_fused vs two GEMMs M={M}")
    # + the trailing projection and its residual in the same launch (insv2v_ffn_fused post=1)
    wp, bp = rnd(C, C, scale=C ** -0.5, seed=6).half(), rnd(C, seed=7) * 0.3
    r2 = rnd(M, C, seed=8).half()
    stream_p = pack_ffn_stream(wf.float(), bf, w2.float().cpu(), b2.cpu(), post=(wp.float().cpu(), bp.cpu())).to(dev())
    outp = ops.ffn_fused(x, stream_p, NH, post_residual=r2)
    refp = ref.half().float() @ wp.float().t() + bp + r2.float()
    close(outp, refp, rel=4e-3, abs_=4e-3, what=f"ffn_fused + proj_out M={M}")
    with pytest.raises(_lib_error()):
        ops.ffn_fused(x[:, :64].contiguous(), stream, 256)


def _lib_error():
    from insv2v import _lib
    return _lib.HipKernelError


@pytest.mark.parametrize("M,N,ln,res,frame,K", [(128, 320, False, False, False, 320), (1000, 320, False, True, False, 320), (777, 960, True, False, False, 320),
                                                  (2 * 16 * 24, 960, True, False, True, 320), (73728 + 5, 320, True, False, False, 320),
                                                  (4096, 64, False, True, False, 320), (3 * 8 * 48, 960, True, False, True, 320),
                                                  (500, 640, False, True, False, 640), (18432 + 3, 1920, True, False, False, 640),
                                                  (2 * 16 * 24, 1920, True, False, True, 640), (2000, 640, False, False, False, 640)])
def test_rowlin_vs_fp32(M, N, ln, res, frame, K):
    """insv2v_rowlin (K = 320 / 640 Linear on the register-resident kernel): plain / residual / in-register LayerNorm / per-frame bias
    table (the temporal positional encoding), ragged last row tile, against fp32 torch on the same fp16-rounded operands and
    against insv2v_gemm with its folded LayerNorm; the optional statistics of the OUTPUT rows against the statistics pass."""
    from insv2v import ops
    from insv2v.fused import pack_linear_stream
    x = (rnd(M, K) * 1.4 + 0.3).half()
    w, b = rnd(N, K, scale=K ** -0.5).half(), rnd(N, seed=1) * 0.5
    r = rnd(M, N, seed=3).half() if res else None
    F_, HW = (16, M // (2 * 16)) if M == 2 * 16 * 24 else (8, 48)
    table = (rnd(F_, N, seed=5) * 0.5) if frame else None
    stream = pack_linear_stream(w.float().cpu(), None if frame else b.cpu(), table.cpu() if frame else None).to(dev())
    out, st = ops.rowlin(x, stream, N, layernorm=ln, residual=r, frames=F_ if frame else 0, rows_per_frame=HW if frame else 0, emit_stats=True)
    close(st, ops.layernorm_stats(out, 1e-5), rel=2e-4, abs_=2e-4, what="rowlin output-row statistics")
    assert torch.equal(out, ops.rowlin(x, stream, N, layernorm=ln, residual=r, frames=F_ if frame else 0, rows_per_frame=HW if frame else 0))
    xf = x.float()
    xn = (xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() if ln else xf
    bias = table[(torch.arange(M, device=dev()) // HW) % F_] if frame else b
    ref = xn @ w.float().t() + bias + (r.float() if res else 0)
    close(out, ref, rel=3e-3, abs_=3e-3, what=f"rowlin {M}x{N} ln={ln} res={res} frame={frame}")
    if ln and not frame:
        two = ops.gemm(x, w, b, row_stats=ops.layernorm_stats(x), col_sum=w.float().sum(1).contiguous())
        close(out, two, rel=3e-3, abs_=3e-3, what="rowlin vs folded-LayerNorm GEMM")
    with pytest.raises(_lib_error()):
        ops.rowlin(x[:, :64].contiguous(), stream, N)


def test_conv3x3_split_into_image_aligned_parts():
    """insv2v_gemm cuts a convolution whose input / output reaches beyond the 2 GiB descriptor window into image-aligned parts (one launch
    each); with the window shrunk (ops.operand_window) so that a small problem takes that path, the result must be bit-identical to the
    single launch - with a per-row-group bias (the time embedding: groups of F images), a residual and a two-source input."""
    from insv2v import ops
    NB, H, W, C1, C2, N, F_ = 12, 16, 16, 64, 64, 64, 2
    x, x2 = rnd(NB * H * W, C1).half(), rnd(NB * H * W, C2, seed=2).half()
    w, b = rnd(N, 9 * (C1 + C2), scale=(9 * (C1 + C2)) ** -0.5).half(), rnd(N, seed=1)
    rb, res = rnd(NB // F_, N, seed=3), rnd(NB * H * W, N, seed=4).half()
    kw = dict(x2=x2, row_bias=rb, rows_per_group=F_ * H * W, residual=res)
    one, _ = ops.conv3x3(x, (NB, H, W), w, b, **kw)
    with ops.operand_window(x.numel() * 2 // 3 + 1):       # -> 3 parts of 4 images
        parts, g = ops.conv3x3(x, (NB, H, W), w, b, **kw)
    assert g == (NB, H, W) and torch.equal(one, parts)
    with ops.operand_window(x.numel() * 2 // 5 + 1):       # 5 does not divide 6 groups -> 6 parts
        parts, _ = ops.conv3x3(x, (NB, H, W), w, b, **kw)
    assert torch.equal(one, parts)
    with ops.operand_window(1024):                          # below one row group: refused, not silently wrong
        with pytest.raises(_lib_error()):
            ops.conv3x3(x, (NB, H, W), w, b, **kw)
    # stride 2 (the output is a quarter of the input: parts are cut by whichever side is larger) and the nearest-x2 up-sampling form
    one, g1 = ops.conv3x3(x, (NB, H, W), w[:, :9 * C1].contiguous(), b, stride=2)
    with ops.operand_window(x.numel() * 2 // 4 + 1):
        parts, g2 = ops.conv3x3(x, (NB, H, W), w[:, :9 * C1].contiguous(), b, stride=2)
    assert g1 == g2 and torch.equal(one, parts)
    one, g1 = ops.conv3x3(x, (NB, H, W), w[:, :9 * C1].contiguous(), b, upsample=True)
    with ops.operand_window(one.numel() * 2 // 3 + 1):
        parts, g2 = ops.conv3x3(x, (NB, H, W), w[:, :9 * C1].contiguous(), b, upsample=True)
    assert g1 == g2 and torch.equal(one, parts)


def test_gemm_split_into_row_ranges():
    """The Linear form of the same split (round 5): rows in ranges aligned to the row-bias groups, every range its own launch, finished
    and partial LayerNorm statistics, a residual and a two-source A operand carried along - bit-identical to the single launch."""
    from insv2v import ops
    M, K, N, G = 6144, 256, 320, 1536
    a, a2 = rnd(M, K).half(), rnd(M, 64, seed=2).half()
    w, b = rnd(N, K + 64, scale=(K + 64) ** -0.5).half(), rnd(N, seed=1)
    rb, res = rnd(M // G, N, seed=3), rnd(M, N, seed=4).half()
    kw = dict(a2=a2, row_bias=rb, rows_per_group=G, residual=res)
    one = ops.gemm(a, w, b, **kw)
    for div in (2, 3, 4):
        with ops.operand_window(M * N * 2 // div + 1):
            assert torch.equal(one, ops.gemm(a, w, b, **kw)), div
    with ops.operand_window(G * N * 2 - 1):                # below one bias group: refused
        with pytest.raises(_lib_error()):
            ops.gemm(a, w, b, **kw)
    # per-frame bias pattern (rb_mod): ranges are whole periods of the pattern
    rbf = rnd(4, N, seed=5)
    one = ops.gemm(a, w[:, :K].contiguous(), b, row_bias=rbf, rows_per_group=128, rb_mod=4)
    with ops.operand_window(M * N * 2 // 3 + 1):
        assert torch.equal(one, ops.gemm(a, w[:, :K].contiguous(), b, row_bias=rbf, rows_per_group=128, rb_mod=4))
    # folded LayerNorm: finished (mean, rstd) rows, and partial sums of a producing GEMM (finalised over the whole problem first)
    wl = w[:, :K].contiguous()
    cs = wl.float().sum(1).contiguous()
    st = ops.layernorm_stats(a)
    # (tile forced: the dispatch picks tiles by problem size, and two tile shapes evaluate the folded-LayerNorm epilogue in different orders)
    one = ops.gemm(a, wl, b, row_stats=st, col_sum=cs, tile=5)
    with ops.operand_window(M * N * 2 // 2 + 1):
        assert torch.equal(one, ops.gemm(a, wl, b, row_stats=st, col_sum=cs, tile=5))
        close(ops.gemm(a, wl, b, row_stats=st, col_sum=cs), one, rel=2e-3, abs_=2e-3, what="split folded-LayerNorm GEMM, dispatched tiles")
    wp = rnd(K, K, scale=K ** -0.5, seed=6).half()
    h, hs = ops.gemm(a, wp, emit_stats=True)
    if isinstance(hs, ops.RowStats):
        one = ops.gemm(h, wl, b, row_stats=hs, col_sum=cs, tile=5)
        with ops.operand_window(M * N * 2 // 2 + 1):
            assert torch.equal(one, ops.gemm(h, wl, b, row_stats=hs, col_sum=cs, tile=5))
    # a split problem cannot emit statistics: the wrapper falls back to the statistics pass, the values stay those of the single launch
    o1, s1 = ops.gemm(a, wp, emit_stats=True)
    with ops.operand_window(M * K * 2 // 2 + 1):
        o2, s2 = ops.gemm(a, wp, emit_stats=True)
    assert torch.equal(o1, o2) and not isinstance(s2, ops.RowStats)
    # GEGLU (output half as wide as N) and an fp32 output
    wg = rnd(2 * N, K, scale=K ** -0.5, seed=7).half()
    one = ops.gemm(a, wg, None, act=ops.ACT_GEGLU)
    with ops.operand_window(M * K * 2 // 3 + 1):
        assert torch.equal(one, ops.gemm(a, wg, None, act=ops.ACT_GEGLU))
    one = ops.gemm(a, wl, b, out_fp32=True)
    with ops.operand_window(M * N * 4 // 3 + 1):
        assert torch.equal(one, ops.gemm(a, wl, b, out_fp32=True))


def test_gemm_operands_beyond_2gib():
    """A real operand beyond the window: the [1 474 597, 960] fp16 output (2.8 GB) of a folded-LayerNorm q/k/v projection through
    insv2v_gemm (the unfused path of clips with more than 16 frames), checked on row blocks at the start, across the 2^31-byte mark and at
    the ragged end against fp32 torch."""
    from insv2v import ops
    M, K, N = 1474560 + 37, 320, 960
    g = torch.Generator(device="cpu").manual_seed(5)
    blk = (torch.randn(4096, K, generator=g) * 1.4 + 0.3).half().to(dev())
    x = blk.repeat(M // 4096 + 1, 1)[:M].contiguous()
    x[1118000:1119000] += 0.25
    w, b = rnd(N, K, scale=K ** -0.5).half(), rnd(N, seed=1) * 0.5
    assert M * N * 2 > 2 ** 31
    out = ops.gemm(x, w, b, row_stats=ops.layernorm_stats(x), col_sum=w.float().sum(1).contiguous())
    for lo, hi in [(0, 512), (1118000, 1119000), (M - 300, M)]:
        xf = x[lo:hi].float()
        xn = (xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        close(out[lo:hi], xn @ w.float().t() + b, rel=3e-3, abs_=3e-3, what=f"gemm LN rows {lo}:{hi} of a 2.8 GB output")
    del out, x
    torch.cuda.empty_cache()


def test_rowlin_operands_beyond_2gib():
    """insv2v_rowlin rebases its buffer descriptors per row tile, so the operands themselves may exceed the 2 GiB descriptor window: the
    fused q/k/v rows of 20 stacked clips are [1 474 560, 960] fp16 = 2.8 GB (insv2v.inference.max_clips_in_flight).  LayerNorm form
    (the spatial self-attention q/k/v) and residual form with a > 2 GiB residual, checked on row blocks at the start, across the
    2^31-byte mark and at the ragged end against fp32 torch, and against the same rows computed as a small problem (bit-identical:
    a row's arithmetic does not depend on its position)."""
    from insv2v import ops
    from insv2v.fused import pack_linear_stream
    M, K, N = 1474560 + 37, 320, 960
    g = torch.Generator(device="cpu").manual_seed(5)
    blk = (torch.randn(4096, K, generator=g) * 1.4 + 0.3).half().to(dev())
    x = blk.repeat(M // 4096 + 1, 1)[:M].contiguous()
    x[1118000:1119000] += 0.25          # rows around byte offset 2^31 of the output (row 1 118 481) differ from their 4096-row period
    w, b = rnd(N, K, scale=K ** -0.5).half(), rnd(N, seed=1) * 0.5
    stream = pack_linear_stream(w.float().cpu(), b.cpu(), None).to(dev())
    assert M * N * 2 > 2 ** 31
    out = ops.rowlin(x, stream, N, layernorm=True)
    res = out                            # a > 2 GiB residual operand for the second form
    out2 = ops.rowlin(x, stream, N, residual=res)
    for lo, hi in [(0, 512), (1118000, 1119000), (M - 300, M)]:
        xf = x[lo:hi].float()
        xn = (xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        ref = xn @ w.float().t() + b
        close(out[lo:hi], ref, rel=3e-3, abs_=3e-3, what=f"rowlin LN rows {lo}:{hi} of a 2.8 GB output")
        close(out2[lo:hi], xf @ w.float().t() + b + out[lo:hi].float(), rel=3e-3, abs_=3e-3, what=f"rowlin +res rows {lo}:{hi}")
    lo = 1118000 - 1118000 % 256         # tile-aligned block as a problem of its own
    small = ops.rowlin(x[lo:lo + 2048].contiguous(), stream, N, layernorm=True)
    assert torch.equal(small, out[lo:lo + 2048])
    del out, out2, res, x
    torch.cuda.empty_cache()


def test_wide_store_kernels_under_co_residency():
    """Regression for the gfx950 16-byte-store hazard (profiles/r02_gemm_debug.md): the persistent GEMMs
    (tiles 230 / 240 / 210 / 211 and the shapes dispatched to them automatically) and the register-resident row kernels store 16 bytes per
    lane with several waves per SIMD; their correctness must not depend on what else shares the SIMD.  Each is run WHILE a second
    stream keeps other kernels (128x128-tile GEMMs, GroupNorm) in flight, several times, and compared element by element with the
    128x128 tile kernel run alone: the two differ only in fp32 summation order, i.e. by at most one fp16 rounding step."""
    from insv2v import ops
    from insv2v.fused import pack_linear_stream
    side = torch.cuda.Stream()
    M = 36864
    bg_a, bg_w = rnd(8192, 640, seed=11).half(), rnd(640, 640, scale=640 ** -0.5, seed=12).half()
    bg_x, bg_g, bg_b = rnd(16 * 1536, 320, seed=13).half(), 1 + 0.1 * rnd(320, seed=14), 0.1 * rnd(320, seed=15)

    def background(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                ops.gemm(bg_a, bg_w, tile=5)
                ops.groupnorm(bg_x, 16, 1536, bg_g, bg_b, 32, 1e-5, silu=True)

    def one_ulp_close(out, ref, what):
        d = (out.float() - ref.float()).abs()
        tol = ref.float().abs() * 2.0 ** -9 + 2.0 ** -12   # two fp16 ulps of slack for values straddling a binade
        bad = int((d > tol).sum())
        assert bad == 0, f"{what}: {bad} elements beyond one fp16 rounding step, worst {d.max().item():.4g}"

    cases = [(960, 320, 230, False), (640, 320, 230, True), (960, 320, 240, False), (640, 320, 240, True), (960, 320, 210, False), (320, 320, 211, True), (2560, 320, 0, False),
             (960, 320, 0, False), (1920, 640, 0, False)]
    for N, K, tile, res in cases:
        a, w, b = rnd(M, K, seed=N).half(), rnd(N, K, scale=K ** -0.5, seed=N + 1).half(), rnd(N, seed=N + 2)
        r = rnd(M, N, seed=N + 3).half() if res else None
        ref = ops.gemm(a, w, b, residual=r, tile=5)
        torch.cuda.synchronize()
        for rep in range(3):
            background(6)
            out = ops.gemm(a, w, b, residual=r, tile=tile)
            torch.cuda.synchronize()
            one_ulp_close(out, ref, f"gemm N={N} K={K} tile={tile} under co-residency, repetition {rep}")
    # the register-resident Linear (two workgroups per CU, 16-byte stores behind v_permlane32_swap)
    for N, res in ((320, True), (960, False)):
        a, w, b = rnd(M, 320, seed=N + 7).half(), rnd(N, 320, scale=320 ** -0.5, seed=N + 8).half(), rnd(N, seed=N + 9)
        r = rnd(M, N, seed=N + 10).half() if res else None
        st = pack_linear_stream(w.float().cpu(), b.cpu()).to(dev())
        ref = ops.gemm(a, w, b, residual=r, tile=5)
        torch.cuda.synchronize()
        for rep in range(3):
            background(6)
            out = ops.rowlin(a, st, N, residual=r)
            torch.cuda.synchronize()
            one_ulp_close(out, ref, f"rowlin N={N} under co-residency, repetition {rep}")


@pytest.mark.parametrize("samples,HW", [(1, 8), (2, 12), (3, 1536)])
def test_tattn_fused_vs_fp32(samples, HW):
    """insv2v_tattn_fused (C = 320, 8 heads x 40, 16 frames: LayerNorm -> +pe -> q/k/v -> attention over the frames of every pixel ->
    to_out -> + residual in one register-resident launch) against fp32 torch on the same fp16-rounded weights, and against the
    unfused path (row-linear q/k/v + insv2v_attention + row-linear out-projection); ragged last pixel tile."""
    from insv2v import ops
    from insv2v.fused import pack_tattn_stream, pack_linear_stream
    C, H, F_, D = 320, 8, 16, 40
    M = samples * F_ * HW
    x = (rnd(M, C) * 1.3 + 0.2).half()
    wqkv = rnd(3 * C, C, scale=C ** -0.5).half()
    table = rnd(F_, 3 * C, seed=1) * 0.4
    wo, bo = rnd(C, C, scale=C ** -0.5, seed=2).half(), rnd(C, seed=3) * 0.3
    stream = pack_tattn_stream(wqkv.float().cpu(), table.cpu(), wo.float().cpu(), bo.cpu()).to(dev())
    out = ops.tattn_fused(x, stream, samples, HW, H, F_)
    xf = x.float()
    xn = (xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    frame = (torch.arange(M, device=dev()) // HW) % F_
    qkv = (xn @ wqkv.float().t() + table[frame]).half().float().reshape(samples, F_, HW, 3, H, D)
    q, k, v = (qkv[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))         # [b, pixel, head, frame, d]
    a = F.scaled_dot_product_attention(q, k, v)                                     # attention over the frames
    a = a.permute(0, 3, 1, 2, 4).reshape(M, C).half().float()
    ref = a @ wo.float().t() + bo + xf
    close(out, ref, rel=4e-3, abs_=4e-3, what=f"tattn_fused samples={samples} HW={HW}")
    # the unfused path of the other widths
    qkv2 = ops.rowlin(x, pack_linear_stream(wqkv.float().cpu(), None, table.cpu()).to(dev()), 3 * C, layernorm=True, frames=F_, rows_per_frame=HW)
    a2 = torch.empty((M, C), device=dev(), dtype=torch.float16)
    pq = qkv2.data_ptr()
    addr = (HW, F_ * HW * 3 * C, 3 * C)
    ops.attention(pq, pq + 2 * C, pq + 4 * C, a2, batch=samples * HW, heads=H, head_dim=D, seq_q=F_, seq_k=F_, scale=D ** -0.5,
                  q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, q_addr=addr, kv_addr=addr, o_addr=(HW, F_ * HW * C, C))
    two = ops.rowlin(a2, pack_linear_stream(wo.float().cpu(), bo.cpu()).to(dev()), C, residual=x)
    close(out, two, rel=4e-3, abs_=4e-3, what=f"tattn_fused vs unfused samples={samples} HW={HW}")
    assert torch.equal(out, ops.tattn_fused(x, stream, samples, HW, H, F_)), "not deterministic"


@pytest.mark.parametrize("samples,HW", [(1, 8), (2, 13), (3, 384)])
def test_tattn_attn_640_vs_fp32(samples, HW):
    """insv2v_tattn_attn (C = 640, 8 heads x 80, 16 frames: LayerNorm -> +pe -> q/k/v -> attention over the frames of every pixel, the attention
    output to memory) against fp32 torch on the same fp16-rounded weights and against the unfused path (row-linear q/k/v + insv2v_attention);
    ragged last pixel tile."""
    from insv2v import ops
    from insv2v.fused import pack_tattn_qkv_stream, pack_linear_stream
    C, H, F_, D = 640, 8, 16, 80
    M = samples * F_ * HW
    x = (rnd(M, C) * 1.3 + 0.2).half()
    wqkv = rnd(3 * C, C, scale=C ** -0.5).half()
    table = rnd(F_, 3 * C, seed=1) * 0.4
    stream = pack_tattn_qkv_stream(wqkv.float().cpu(), table.cpu()).to(dev())
    assert ops.tattn_attn_supported(C, H, F_)
    out = ops.tattn_attn(x, stream, samples, HW, H, F_)
    xf = x.float()
    xn = (xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    frame = (torch.arange(M, device=dev()) // HW) % F_
    qkv = (xn @ wqkv.float().t() + table[frame]).half().float().reshape(samples, F_, HW, 3, H, D)
    q, k, v = (qkv[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))         # [b, pixel, head, frame, d]
    ref = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(M, C)
    close(out, ref, rel=4e-3, abs_=4e-3, what=f"tattn_attn samples={samples} HW={HW}")
    qkv2 = ops.rowlin(x, pack_linear_stream(wqkv.float().cpu(), None, table.cpu()).to(dev()), 3 * C, layernorm=True, frames=F_, rows_per_frame=HW)
    a2 = torch.empty((M, C), device=dev(), dtype=torch.float16)
    pq = qkv2.data_ptr()
    addr = (HW, F_ * HW * 3 * C, 3 * C)
    ops.attention(pq, pq + 2 * C, pq + 4 * C, a2, batch=samples * HW, heads=H, head_dim=D, seq_q=F_, seq_k=F_, scale=D ** -0.5,
                  q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, q_addr=addr, kv_addr=addr, o_addr=(HW, F_ * HW * C, C))
    close(out, a2, rel=4e-3, abs_=4e-3, what=f"tattn_attn vs unfused samples={samples} HW={HW}")
    assert torch.equal(out, ops.tattn_attn(x, stream, samples, HW, H, F_)), "not deterministic"


@pytest.mark.parametrize("samples,rows,L", [(1, 128, 77), (3, 384, 77), (2, 256, 96), (5, 1536, 65), (15, 24576, 77)])
def test_xattn_fused_vs_fp32(samples, rows, L):
    """insv2v_xattn_fused (C = 320, 8 heads x 40: LayerNorm -> q -> attention over the sample's text tokens -> to_out -> + residual in one
    register-resident launch, text K / V as a per-sample fragment stream) against fp32 torch on the same fp16-rounded operands and against
    the unfused path (row-linear q + insv2v_attention + row-linear out-projection); several samples per launch, ctx_len 65 ... 96."""
    from insv2v import ops
    from insv2v.fused import pack_xattn_stream, pack_xattn_kv, pack_linear_stream
    C, H, D = 320, 8, 40
    M = samples * rows
    x = (rnd(M, C) * 1.3 + 0.2).half()
    wq, bq = rnd(C, C, scale=C ** -0.5).half(), rnd(C, seed=5) * 0.3
    wo, bo = rnd(C, C, scale=C ** -0.5, seed=2).half(), rnd(C, seed=3) * 0.3
    kv = (rnd(samples * L, 2 * C, seed=4) * 1.5).half()
    stream = pack_xattn_stream(wq.float().cpu(), bq.cpu(), wo.float().cpu(), bo.cpu()).to(dev())
    kvs = pack_xattn_kv(kv, samples, L, C, H)
    assert ops.xattn_fused_supported(C, H, L, rows)
    out = ops.xattn_fused(x, stream, kvs, rows, H, L)
    xf = x.float()
    xn = ((xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()).half().float()
    q = (xn @ wq.float().t() + bq).half().float().reshape(samples, rows, H, D).permute(0, 2, 1, 3)
    k = kv[:, :C].float().reshape(samples, L, H, D).permute(0, 2, 1, 3)
    v = kv[:, C:].float().reshape(samples, L, H, D).permute(0, 2, 1, 3)
    a = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(M, C).half().float()
    ref = a @ wo.float().t() + bo + xf
    close(out, ref, rel=4e-3, abs_=4e-3, what=f"xattn_fused samples={samples} rows={rows} L={L}")
    # the unfused path of the other widths
    q2 = ops.rowlin(x, pack_linear_stream(wq.float().cpu(), bq.cpu()).to(dev()), C, layernorm=True)
    a2 = torch.empty((M, C), device=dev(), dtype=torch.float16)
    kp = kv.data_ptr()
    ops.attention(q2.data_ptr(), kp, kp + 2 * C, a2, batch=samples, heads=H, head_dim=D, seq_q=rows, seq_k=L, scale=D ** -0.5,
                  q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C, q_addr=(1, rows * C, 0), kv_addr=(1, L * 2 * C, 0), o_addr=(1, rows * C, 0))
    two = ops.rowlin(a2, pack_linear_stream(wo.float().cpu(), bo.cpu()).to(dev()), C, residual=x)
    close(out, two, rel=4e-3, abs_=4e-3, what=f"xattn_fused vs unfused samples={samples} rows={rows} L={L}")
    assert torch.equal(out, ops.xattn_fused(x, stream, kvs, rows, H, L)), "not deterministic"
    # + the preceding self-attention's output projection in the same launch: x1 = to_out1(a) + h never exists in memory
    wo1, bo1 = rnd(C, C, scale=C ** -0.5, seed=11).half(), rnd(C, seed=12) * 0.3
    a1, hres = (rnd(M, C, seed=13) * 0.8).half(), (rnd(M, C, seed=14) * 1.1 + 0.1).half()
    pre_stream = pack_xattn_stream(wq.float().cpu(), bq.cpu(), wo.float().cpu(), bo.cpu(), pre=(wo1.float().cpu(), bo1.cpu())).to(dev())
    out_pre = ops.xattn_fused(a1, pre_stream, kvs, rows, H, L, pre_residual=hres)
    x1 = ops.rowlin(a1, pack_linear_stream(wo1.float().cpu(), bo1.cpu()).to(dev()), C, residual=hres)
    close(out_pre, ops.xattn_fused(x1, stream, kvs, rows, H, L), rel=4e-3, abs_=4e-3, what=f"xattn_fused with leading out-projection samples={samples} rows={rows}")
    assert torch.equal(out_pre, ops.xattn_fused(a1, pre_stream, kvs, rows, H, L, pre_residual=hres)), "not deterministic"


@pytest.mark.parametrize("samples,rows,L", [(1, 128, 77), (3, 384, 77), (2, 256, 96), (5, 6144, 65)])
def test_xattn_attn_640_vs_fp32(samples, rows, L):
    """insv2v_xattn_attn (C = 640, 8 heads x 80: LayerNorm -> q -> attention over the sample's text tokens, the attention output to memory)
    against fp32 torch on the same fp16-rounded operands and against the unfused path (row-linear q + insv2v_attention)."""
    from insv2v import ops
    from insv2v.fused import pack_xattn_q_stream, pack_xattn640_kv, pack_linear_stream
    C, H, D = 640, 8, 80
    M = samples * rows
    x = (rnd(M, C) * 1.3 + 0.2).half()
    wq, bq = rnd(C, C, scale=C ** -0.5).half(), rnd(C, seed=5) * 0.3
    kv = (rnd(samples * L, 2 * C, seed=4) * 1.5).half()
    stream = pack_xattn_q_stream(wq.float().cpu(), bq.cpu()).to(dev())
    kvs = pack_xattn640_kv(kv, samples, L, C, H)
    assert ops.xattn_attn_supported(C, H, L, rows)
    out = ops.xattn_attn(x, stream, kvs, rows, H, L)
    xf = x.float()
    xn = ((xf - xf.mean(1, keepdim=True)) * (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()).half().float()
    q = (xn @ wq.float().t() + bq).half().float().reshape(samples, rows, H, D).permute(0, 2, 1, 3)
    k = kv[:, :C].float().reshape(samples, L, H, D).permute(0, 2, 1, 3)
    v = kv[:, C:].float().reshape(samples, L, H, D).permute(0, 2, 1, 3)
    ref = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(M, C)
    close(out, ref, rel=4e-3, abs_=4e-3, what=f"xattn_attn samples={samples} rows={rows} L={L}")
    q2 = ops.rowlin(x, pack_linear_stream(wq.float().cpu(), bq.cpu()).to(dev()), C, layernorm=True)
    a2 = torch.empty((M, C), device=dev(), dtype=torch.float16)
    kp = kv.data_ptr()
    ops.attention(q2.data_ptr(), kp, kp + 2 * C, a2, batch=samples, heads=H, head_dim=D, seq_q=rows, seq_k=L, scale=D ** -0.5,
                  q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C, q_addr=(1, rows * C, 0), kv_addr=(1, L * 2 * C, 0), o_addr=(1, rows * C, 0))
    close(out, a2, rel=4e-3, abs_=4e-3, what=f"xattn_attn vs unfused samples={samples} rows={rows} L={L}")
    assert torch.equal(out, ops.xattn_attn(x, stream, kvs, rows, H, L)), "not deterministic"


@pytest.mark.parametrize("K,nsamples,rows", [(320, 6, 96), (640, 3, 384), (320, 48, 1536)])
def test_rowlin_fused_groupnorm(K, nsamples, rows):
    """insv2v_rowlin(gn_ab=...): the per-sample GroupNorm in front of the transformer blocks' proj_in (attention.py:101-103,
    motion_module.py:136-139) applied on the fly from the statistics-only GroupNorm output == GroupNorm kernel followed by the
    row kernel, and == fp32 torch."""
    from insv2v import ops
    from insv2v.fused import pack_linear_stream
    M, N, G = nsamples * rows, K, 32
    x = (rnd(M, K) * 1.7 + 0.4).half()
    gamma, beta = 1 + 0.1 * rnd(K, seed=1), 0.1 * rnd(K, seed=2)
    w, b = rnd(N, K, scale=K ** -0.5, seed=3).half(), rnd(N, seed=4) * 0.3
    st = pack_linear_stream(w.float().cpu(), b.cpu()).to(dev())
    ab = ops.groupnorm_stats(x, nsamples, rows, gamma, beta, G, 1e-6)
    out = ops.rowlin(x, st, N, gn_ab=ab, gn_rows=rows)
    two = ops.rowlin(ops.groupnorm(x, nsamples, rows, gamma, beta, G, 1e-6), st, N)
    close(out, two, rel=3e-3, abs_=3e-3, what="fused GroupNorm vs GroupNorm kernel + rowlin")
    xr = x.float().reshape(nsamples, rows, K).permute(0, 2, 1)
    ref = F.group_norm(xr, G, gamma, beta, 1e-6).permute(0, 2, 1).reshape(M, K) @ w.float().t() + b
    close(out, ref, rel=4e-3, abs_=4e-3, what="fused GroupNorm + rowlin vs fp32")


@pytest.mark.parametrize("split", [0, 2, 3, 8])
def test_gemm_split_k(split):
    """Split-K (forced, and the automatic choice for a small-M / long-K problem) == single pass."""
    from insv2v import ops
    M, N, K = 200, 136, 64 * 64
    a, w, b = rnd(M, K).half(), rnd(N, K, scale=K ** -0.5).half(), rnd(N)
    res = rnd(M, N, seed=3).half()
    rb = rnd(2, N, seed=5)
    ref = 0.5 * (a.float() @ w.float().t()) + b + rb.repeat_interleave(100, 0)
    out = ops.gemm(a, w, b, residual=res, row_bias=rb, rows_per_group=100, alpha=0.5, split_k=split)
    close(out, ref + res.float(), what=f"split-k {split}")
    out = ops.gemm(a, w, b, row_bias=rb, rows_per_group=100, alpha=0.5, act=ops.ACT_SILU, out_fp32=True, split_k=split)
    close(out, F.silu(ref), what=f"split-k {split} silu fp32")


@pytest.mark.parametrize("split", [0, 4])
def test_conv3x3_split_k(split):
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, cin, cout, h, w = 2, 512, 72, 6, 4
    x = rnd(nb, cin, h, w).half().float()
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5).half().float()
    b = rnd(cout, seed=4)
    res = rnd(nb * h * w, cout, seed=6).half()
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    out, _ = ops.conv3x3(to_cl(x), (nb, h, w), wk, bk, residual=res, split_k=split)
    close(out, to_cl(conv_ref(x, wt, b, 1, (1, 1), False)).float() + res.float(), what=f"conv split-k {split}")


@pytest.mark.parametrize("act", [0, 2])
def test_gemm_folded_layernorm(act):
    """LayerNorm folded into the GEMM epilogue (+ per-frame positional-encoding bias) == LayerNorm then Linear."""
    from insv2v import ops
    from insv2v.unet import fold_layernorm, interleave32
    B, Fr, HW, C = 2, 4, 24, 320
    N = 8 * C if act == 2 else 3 * C
    x = (rnd(B * Fr * HW, C) * 2 + 1.5 + rnd(1, C, seed=9)).half()  # non-zero token means: the folded term matters
    g, be = 1 + 0.2 * rnd(C, seed=3), 0.2 * rnd(C, seed=4)
    w, bias = rnd(N, C, scale=C ** -0.5), rnd(N, seed=6)
    pe = rnd(32, C, seed=7)
    wf, col, b = fold_layernorm(w.cpu(), g.cpu(), be.cpu(), bias.cpu())
    pe_bias = (pe @ w.t()).contiguous()
    ln = F.layer_norm(x.float(), (C,), g, be, 1e-5)
    stats = ops.layernorm_stats(x, 1e-5)
    close(stats[:, 0], x.float().mean(1), rel=1e-5, abs_=1e-5, what="ln mean")
    close(stats[:, 1], (x.float().var(1, unbiased=False) + 1e-5).rsqrt(), rel=1e-4, what="ln rstd")
    if act == 2:
        out = ops.gemm(x, interleave32(wf).to(dev()).contiguous(), interleave32(b).to(dev()).contiguous(), act=2,
                       row_stats=stats, col_sum=interleave32(col).to(dev()).contiguous())
        y = ln @ w.t() + bias
        h, gg = y.chunk(2, dim=-1)
        close(out, h * F.gelu(gg), rel=4e-3, what="folded LN + GEGLU")
    else:
        out = ops.gemm(x, wf.to(dev()), b.to(dev()), row_stats=stats, col_sum=col.to(dev()),
                       row_bias=pe_bias[3:3 + Fr], rows_per_group=HW, rb_mod=Fr)
        ref = (ln.reshape(B, Fr, HW, C) + pe[3:3 + Fr][None, :, None, :]).reshape(-1, C) @ w.t() + bias
        close(out, ref, rel=4e-3, what="folded LN + PE bias")


@pytest.mark.parametrize("M,N,K,res,ln", [(16384 + 100, 640, 2560, True, False), (16384, 1664, 1280, False, True), (32768 + 8, 640, 1280, False, False)])
def test_gemm_persistent_partial_column_tile(M, N, K, res, ln):
    """N = 2.5 (6.5) column tiles of the 256x256 persistent kernel (FF2 of UNet level 1 at the stacked-clip sizes): half-empty last tile
    column, residual, folded LayerNorm, ragged last row tile - the automatic dispatch and the forced persistent kernel against fp32 torch."""
    from insv2v import ops
    a, w, b = (rnd(M, K) * 1.2 + 0.3).half(), rnd(N, K, scale=K ** -0.5).half(), rnd(N)
    r = rnd(M, N, seed=5).half() if res else None
    kw = dict(row_stats=ops.layernorm_stats(a), col_sum=w.float().sum(1).contiguous()) if ln else {}
    out = ops.gemm(a, w, b, residual=r, **kw)
    one = ops.gemm(a, w, b, residual=r, tile=230, **kw)
    close(out, one, rel=2e-3, abs_=2e-3, what="automatic dispatch vs forced persistent kernel")
    x = F.layer_norm(a.float(), (K,)) if ln else a.float()
    ref = x @ w.float().t() + b + (r.float() if res else 0)
    close(out, ref, rel=4e-3 if ln else 2e-3, what=f"gemm {M}x{N}x{K} partial column tile")


@pytest.mark.parametrize("tile", [230, 231, 240])
@pytest.mark.parametrize("M,N,K,res,ln,rb,ksplit", [(1000, 328, 192, True, False, False, 0), (257, 256, 64, False, False, False, 0),
                                                       (16384 + 100, 640, 2560, True, False, False, 0), (16384, 1664, 1280, False, True, False, 0),
                                                       (70000, 320, 320, False, True, True, 0), (4096, 320, 960, False, False, False, 640),
                                                       (256 * 300 + 8, 512, 128, True, True, False, 0)])
def test_gemm_q8_vs_fp32(tile, M, N, K, res, ln, rb, ksplit):
    """gemm_q8 (round 4: 256x256 8-phase kernel, interleaved half-tile ownership; 230 = LDS-DMA requests inside the MFMA segments, the
    product schedule, 231 = in the load segments) and gemm_r8 (240: the 256x320 tile, five phases per K tile), forced: residual, folded LayerNorm, per-frame row bias, two-source K (concat), one
    K tile, an odd number of K tiles (the ring parity carries over into the next tile), partial last row / column tiles, more tiles
    than CUs (the persistent stream crosses tile boundaries) - against fp32 torch on the same fp16-rounded operands."""
    from insv2v import ops
    a, w, b = (rnd(M, K) * 1.2 + 0.3).half(), rnd(N, K, scale=K ** -0.5).half(), rnd(N)
    r = rnd(M, N, seed=5).half() if res else None
    kw = dict(row_stats=ops.layernorm_stats(a), col_sum=w.float().sum(1).contiguous()) if ln else {}
    x = F.layer_norm(a.float(), (K,)) if ln else a.float()
    ref = x @ w.float().t() + b + (r.float() if res else 0)
    if rb:
        Fr, HW = 16, 256
        table = rnd(Fr, N, seed=9) * 0.5
        kw.update(row_bias=table, rows_per_group=HW, rb_mod=Fr)
        ref = ref + table[(torch.arange(M, device=dev()) // HW) % Fr]
    if ksplit:
        out = ops.gemm(a[:, :ksplit].contiguous(), w, b, a2=a[:, ksplit:].contiguous(), tile=tile)
    else:
        out = ops.gemm(a, w, b, residual=r, tile=tile, **kw)
    close(out, ref, rel=4e-3 if ln else 2e-3, what=f"gemm_q8 tile {tile} {M}x{N}x{K}")
    assert torch.equal(out, ops.gemm(a[:, :ksplit].contiguous(), w, b, a2=a[:, ksplit:].contiguous(), tile=tile) if ksplit
                       else ops.gemm(a, w, b, residual=r, tile=tile, **kw)), "gemm_q8 is not deterministic"


@pytest.mark.parametrize("tile", [230, 231])
def test_gemm_q8_geglu(tile):
    """GEGLU on gemm_q8: the [h | g] interleaved projection rows are permuted by the LDS-DMA source addresses so that W half 0 holds the
    h blocks and W half 1 the gates of the same outputs; folded LayerNorm in front; N = 2.5 column tiles, ragged M."""
    from insv2v import ops
    from insv2v.unet import fold_layernorm, interleave32
    M, C, NH = 2 * 4096 + 78, 320, 640
    x = (rnd(M, C) * 1.3 + 0.2).half()
    w1, b1 = rnd(2 * NH, C, scale=C ** -0.5), rnd(2 * NH, seed=1) * 0.3
    gamma, beta = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    wf, col, bf = fold_layernorm(w1.cpu(), gamma.cpu(), beta.cpu(), b1.cpu())
    out = ops.gemm(x, interleave32(wf).to(dev()), interleave32(bf).to(dev()), act=ops.ACT_GEGLU, row_stats=ops.layernorm_stats(x),
                   col_sum=interleave32(col).to(dev()), tile=tile)
    y = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w1.half().float().t() + b1
    h, g = y.chunk(2, dim=-1)
    close(out, h * F.gelu(g), rel=4e-3, abs_=4e-3, what=f"gemm_q8 GEGLU tile {tile}")
    ref5 = ops.gemm(x, interleave32(wf).to(dev()), interleave32(bf).to(dev()), act=ops.ACT_GEGLU, row_stats=ops.layernorm_stats(x),
                    col_sum=interleave32(col).to(dev()), tile=5)
    close(out, ref5, rel=2e-3, abs_=2e-3, what="gemm_q8 GEGLU vs the 128x128 tile")


@pytest.mark.parametrize("M,C,NH,tile", [(2 * 4096 + 78, 320, 640, 240), (8192 + 256, 640, 2560, 240), (9216, 1280, 5120, 0), (300, 320, 160, 240)])
def test_gemm_r8_geglu(M, C, NH, tile):
    """GEGLU on gemm_r8 (256x320 tiles, five 32-row fragments per wave): the LDS-DMA source-row map puts 16 value rows and their 16 gate
    rows of the [32 value | 32 gate] interleaved projection into every fragment, the product is in-lane.  Folded LayerNorm in front,
    ragged M, several column tiles, a problem smaller than one tile; forced (tile 240) and by dispatch (N = 10240, K = 1280: the
    level-2 FF1), against fp32 torch and the 128x128 tile kernel."""
    from insv2v import ops
    from insv2v.unet import fold_layernorm, interleave32
    x = (rnd(M, C) * 1.3 + 0.2).half()
    w1, b1 = rnd(2 * NH, C, scale=C ** -0.5), rnd(2 * NH, seed=1) * 0.3
    gamma, beta = 1 + 0.1 * rnd(C, seed=4), 0.1 * rnd(C, seed=5)
    wf, col, bf = fold_layernorm(w1.cpu(), gamma.cpu(), beta.cpu(), b1.cpu())
    args = (x, interleave32(wf).to(dev()), interleave32(bf).to(dev()))
    kw = dict(act=ops.ACT_GEGLU, row_stats=ops.layernorm_stats(x), col_sum=interleave32(col).to(dev()))
    out = ops.gemm(*args, tile=tile, **kw)
    assert out.shape == (M, NH)
    y = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w1.half().float().t() + b1
    h, g = y.chunk(2, dim=-1)
    close(out, h * F.gelu(g), rel=4e-3, abs_=4e-3, what=f"gemm_r8 GEGLU tile {tile}")
    close(out, ops.gemm(*args, tile=5, **kw), rel=2e-3, abs_=2e-3, what="gemm_r8 GEGLU vs the 128x128 tile")


@pytest.mark.parametrize("tile", [230, 231, 240])
@pytest.mark.parametrize("h,w,stride,ups,cat", [(16, 16, 1, False, False), (8, 16, 1, True, False), (32, 32, 2, False, True), (16, 32, 1, False, True)])
def test_conv3x3_q8(tile, h, w, stride, ups, cat):
    """The gathered 3x3 convolution on gemm_q8: per-tap source offsets are refreshed only when the tap / source changes; zero padding,
    stride 2, nearest x2 upsample by index, two-source channel concat, per-sample row bias, residual - against torch and the 128x128 tile."""
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, c1, c2, cout = 6, 128, 64 if cat else 0, 320
    x1 = rnd(nb, c1, h, w).half().float()
    x2 = rnd(nb, c2, h, w, seed=2).half().float() if cat else None
    xin = torch.cat([x1, x2], 1) if cat else x1
    wt = rnd(cout, c1 + c2, 3, 3, scale=(9 * (c1 + c2)) ** -0.5).half().float()
    b = rnd(cout, seed=4)
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    ref = conv_ref(xin, wt, b, stride, (1, 1), ups)
    oh, ow = ref.shape[2], ref.shape[3]
    rb = rnd(nb, cout, seed=6) * 0.5
    res = rnd(nb * oh * ow, cout, seed=7).half()
    kw = dict(x2=to_cl(x2) if cat else None, row_bias=rb, rows_per_group=oh * ow, residual=res, stride=stride, upsample=ups)
    out, geom = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, tile=tile, **kw)
    assert geom == (nb, oh, ow)
    full = to_cl(ref).float() + rb.repeat_interleave(oh * ow, 0) + res.float()
    close(out, full, what=f"conv3x3 on gemm_q8 tile {tile}")
    out5, _ = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, tile=5, **kw)
    close(out, out5, what="conv3x3 gemm_q8 vs the 128x128 tile")


def test_gemm_concat_and_strided():
    from insv2v import ops
    M, K1, K2, N = 300, 128, 64, 96
    big = rnd(M, K1 + 40).half()
    a1 = big[:, :K1]  # row stride K1+40
    a2 = rnd(M, K2, seed=1).half()
    w = rnd(N, K1 + K2, scale=0.1).half()
    out = ops.gemm(a1, w, a2=a2)
    close(out, torch.cat([a1, a2], 1).float() @ w.float().t(), what="gemm concat")


def test_gemm_batched():
    from insv2v import ops
    B, M, N, K = 3, 96, 80, 64
    a, w = rnd(B, M, K).half(), rnd(B, N, K, scale=0.2).half()
    out = torch.empty((B, M, N), device=dev(), dtype=torch.float16)
    ops.gemm(a.reshape(B * M, K), w.reshape(B * N, K), out=out, batch=B, M=M, N=N, K=K, lda=K, ldw=K, ldc=N,
             a_bs=M * K, w_bs=N * K, c_bs=M * N, alpha=0.25)
    close(out, 0.25 * torch.bmm(a.float(), w.float().transpose(1, 2)), what="gemm batched")


def test_gemm_rejects_bad_args():
    from insv2v import ops, _lib
    with pytest.raises(_lib.HipKernelError):
        ops.gemm(rnd(8, 12).half(), rnd(8, 12).half())  # K % 8 != 0
    with pytest.raises(_lib.HipKernelError):
        ops.gemm(torch.zeros(8, 16), torch.zeros(8, 16))  # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------- conv
def conv_ref(x_nchw, w, b, stride, pad, upsample):
    if upsample:
        x_nchw = F.interpolate(x_nchw, scale_factor=2.0, mode="nearest")
    if pad == (0, 0) and stride == 2:
        x_nchw = F.pad(x_nchw, (0, 1, 0, 1))
        return F.conv2d(x_nchw, w, b, stride=2, padding=0)
    return F.conv2d(x_nchw, w, b, stride=stride, padding=1)


def to_cl(x):  # NCHW -> [N*H*W, C] fp16
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).half().contiguous()


@pytest.mark.parametrize("cin,cout,h,w,stride,pad,ups", [(64, 64, 8, 12, 1, (1, 1), False), (128, 320, 16, 24, 1, (1, 1), False),
                                                          (64, 128, 16, 24, 2, (1, 1), False), (64, 64, 8, 12, 1, (1, 1), True),
                                                          (128, 64, 16, 16, 2, (0, 0), False), (64, 4, 8, 12, 1, (1, 1), False),
                                                          (192, 3, 10, 6, 1, (1, 1), False)])
def test_conv3x3(cin, cout, h, w, stride, pad, ups):
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb = 3
    x = rnd(nb, cin, h, w).half().float()
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5).half().float()
    b = rnd(cout, seed=4)
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    out, geom = ops.conv3x3(to_cl(x), (nb, h, w), wk, bk, stride=stride, pad=pad, upsample=ups)
    ref = conv_ref(x, wt, b, stride, pad, ups)
    assert geom == (nb, ref.shape[2], ref.shape[3])
    close(out, to_cl(ref).float(), what=f"conv {cin}->{cout} s{stride} pad{pad} up{ups}")


@pytest.mark.parametrize("tile", [3, 5, 6, 8, 9, 35, 36, 45])
def test_conv3x3_tiles(tile):
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, cin, cout, h, w = 3, 128, 192, 12, 20
    x = rnd(nb, cin, h, w).half().float()
    wt = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5).half().float()
    b = rnd(cout, seed=4)
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    out, _ = ops.conv3x3(to_cl(x), (nb, h, w), wk, bk, tile=tile)
    close(out, to_cl(conv_ref(x, wt, b, 1, (1, 1), False)).float(), what=f"conv tile {tile}")


@pytest.mark.parametrize("h,w,ups", [(8, 16, False), (16, 8, False), (32, 48, False), (16, 24, False), (8, 8, True), (16, 24, True)])
@pytest.mark.parametrize("tile", [100, 101])
def test_conv3x3_halo(h, w, ups, tile):
    """Patch-tiled conv (tile=100): input patch resident in LDS, taps read shifted rows; concat, per-sample row bias, SiLU-free
    epilogue, residual and nearest-x2 upsample must match the gathered kernel's semantics."""
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, c1, c2, cout = 3, 128, 64, 192
    x1, x2 = rnd(nb, c1, h, w).half().float(), rnd(nb, c2, h, w, seed=1).half().float()
    wt = rnd(cout, c1 + c2, 3, 3, scale=(9 * (c1 + c2)) ** -0.5).half().float()
    b, rb = rnd(cout), rnd(nb, cout, seed=8)
    oh, ow = (2 * h, 2 * w) if ups else (h, w)
    res = rnd(nb * oh * ow, cout, seed=6).half()
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    out, geom = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, x2=to_cl(x2), row_bias=rb, rows_per_group=oh * ow,
                            residual=res, upsample=ups, tile=tile)
    assert geom == (nb, oh, ow)
    ref = to_cl(conv_ref(torch.cat([x1, x2], 1), wt, b, 1, (1, 1), ups)).float() + rb.repeat_interleave(oh * ow, 0) + res.float()
    close(out, ref, what=f"halo conv {h}x{w} up{ups}")
    # same arithmetic as the gathered kernel up to fp32 summation order (channel-block-major vs tap-major K order)
    out2, _ = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, x2=to_cl(x2), row_bias=rb, rows_per_group=oh * ow,
                          residual=res, upsample=ups, tile=5)
    assert (out.float() - out2.float()).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("h,w,ups", [(16, 16, False), (32, 48, False), (32, 8, False), (8, 8, True), (16, 24, True)])
def test_conv3x3_halo_256_pixel_patch(h, w, ups):
    """The 16-wave form of the patch-tiled conv (tile=103: 16x16 / 32x8 pixel patch, weight slices requested two taps ahead with a counted
    wait): concat, per-sample row bias, residual, upsample, N not a multiple of the tile - against torch and against the shipped 8-wave form."""
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, c1, c2, cout = 3, 128, 64, 320
    x1, x2 = rnd(nb, c1, h, w).half().float(), rnd(nb, c2, h, w, seed=1).half().float()
    wt = rnd(cout, c1 + c2, 3, 3, scale=(9 * (c1 + c2)) ** -0.5).half().float()
    b, rb = rnd(cout), rnd(nb, cout, seed=8)
    oh, ow = (2 * h, 2 * w) if ups else (h, w)
    res = rnd(nb * oh * ow, cout, seed=6).half()
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    kw = dict(x2=to_cl(x2), row_bias=rb, rows_per_group=oh * ow, residual=res, upsample=ups)
    out, geom = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, tile=103, **kw)
    assert geom == (nb, oh, ow)
    ref = to_cl(conv_ref(torch.cat([x1, x2], 1), wt, b, 1, (1, 1), ups)).float() + rb.repeat_interleave(oh * ow, 0) + res.float()
    close(out, ref, what=f"256-pixel halo conv {h}x{w} up{ups}")
    out2, _ = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, tile=100, **kw)
    assert torch.equal(out, out2), (out.float() - out2.float()).abs().max().item()   # same K order, same accumulation per output


@pytest.mark.parametrize("h,w,frames,silu,concat", [(8, 16, 2, True, False), (16, 8, 2, True, True), (32, 48, 4, True, False),
                                                    (16, 24, 4, False, True)])
def test_conv3x3_fused_groupnorm(h, w, frames, silu, concat):
    """ResnetBlock3D's GroupNorm(5-D: statistics over (C/G, f, h, w)) + SiLU applied INSIDE the patch-tiled convolution
    (resnet.py:177-178,188-194): statistics-only GroupNorm -> (scale, shift) table -> conv3x3(gn_ab=...) on the raw tensor
    == GroupNorm+SiLU then conv (torch fp32), and == the unfused kernel pair within fp16 rounding; zero padding stays zero."""
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    B, c1, c2, cout, G = 2, 128, 64 if concat else 0, 192, 32
    nb, C = B * frames, 128 + (64 if concat else 0)
    x1 = (rnd(nb, c1, h, w) * 1.5 + rnd(1, c1, 1, 1, seed=2)).half().float()
    x2 = (rnd(nb, c2, h, w, seed=1) - 0.5).half().float() if concat else None
    xc = torch.cat([x1, x2], 1) if concat else x1
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    wt = rnd(cout, C, 3, 3, scale=(9 * C) ** -0.5).half().float()
    b, rb = rnd(cout), rnd(B, cout, seed=8)
    res = rnd(nb * h * w, cout, seed=6).half()
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    rows = frames * h * w
    x1c, x2c = to_cl(x1), (to_cl(x2) if concat else None)
    ab = ops.groupnorm_stats(x1c, B, rows, gamma, beta, G, 1e-5, x2=x2c)
    out, _ = ops.conv3x3(x1c, (nb, h, w), wk, bk, x2=x2c, row_bias=rb, rows_per_group=rows, residual=res, tile=100,
                         gn_ab=ab, gn_images_per_sample=frames, gn_silu=silu)
    # torch reference: 5-D GroupNorm over (frames, h, w) per sample, then SiLU, then the conv
    x5 = xc.reshape(B, frames, C, h, w).permute(0, 2, 1, 3, 4)
    n5 = F.group_norm(x5, G, gamma, beta, 1e-5)
    if silu:
        n5 = F.silu(n5)
    n4 = n5.permute(0, 2, 1, 3, 4).reshape(nb, C, h, w)
    ref = to_cl(conv_ref(n4, wt, b, 1, (1, 1), False)).float() + rb.repeat_interleave(rows, 0) + res.float()
    close(out, ref, rel=4e-3, what=f"conv with fused GroupNorm {h}x{w} silu={silu} concat={concat}")
    # the unfused pair (normalised copy in HBM, same fp16 rounding of the normalised activations)
    n = ops.groupnorm(x1c, B, rows, gamma, beta, G, 1e-5, silu=silu, x2=x2c)
    out2, _ = ops.conv3x3(n, (nb, h, w), wk, bk, row_bias=rb, rows_per_group=rows, residual=res, tile=100)
    assert (out.float() - out2.float()).abs().max() <= 4e-3 * ref.abs().max()
    # (scale, shift) table itself
    mean = x5.reshape(B, G, -1).mean(2)
    rstd = (x5.reshape(B, G, -1).var(2, unbiased=False) + 1e-5).rsqrt()
    a_ref = rstd.repeat_interleave(C // G, 1) * gamma[None]
    b_ref = beta[None] - mean.repeat_interleave(C // G, 1) * a_ref
    close(ab[..., 0], a_ref, rel=2e-4, what="GroupNorm scale table")
    close(ab[..., 1], b_ref, rel=2e-4, abs_=2e-4, what="GroupNorm shift table")


def test_conv3x3_fused_groupnorm_only_on_the_patch_kernel():
    from insv2v import ops, _lib
    from insv2v.unet import prep_conv3x3
    assert ops.conv3x3_fuses_groupnorm((48, 32, 48), 320, 320) and ops.conv3x3_fuses_groupnorm((48, 16, 24), 1920, 640, 1280)
    assert not ops.conv3x3_fuses_groupnorm((48, 8, 12), 1280, 1280)      # 8x12 does not tile into 8x16 / 16x8 patches
    assert not ops.conv3x3_fuses_groupnorm((2, 8, 16), 64, 64)           # too few tiles: the gathered kernel is chosen
    wk, bk = prep_conv3x3({"c.weight": torch.zeros(64, 64, 3, 3), "c.bias": torch.zeros(64)}, "c", dev())
    ab = torch.zeros(1, 64, 2, device=dev())
    with pytest.raises(_lib.HipKernelError):  # never silently un-normalised
        ops.conv3x3(rnd(2 * 12 * 20, 64).half(), (2, 12, 20), wk, bk, gn_ab=ab, gn_images_per_sample=2)


def test_conv3x3_halo_rejects_unsupported_geometry():
    from insv2v import ops, _lib
    from insv2v.unet import prep_conv3x3
    wk, bk = prep_conv3x3({"c.weight": torch.zeros(64, 64, 3, 3), "c.bias": torch.zeros(64)}, "c", dev())
    with pytest.raises(_lib.HipKernelError):
        ops.conv3x3(rnd(2 * 12 * 20, 64).half(), (2, 12, 20), wk, bk, tile=100)


def test_conv3x3_concat_rowbias_residual_fp32():
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, c1, c2, cout, h, w = 4, 128, 64, 64, 8, 8
    x1, x2 = rnd(nb, c1, h, w).half().float(), rnd(nb, c2, h, w, seed=1).half().float()
    wt = rnd(cout, c1 + c2, 3, 3, scale=0.03).half().float()
    b, rb = rnd(cout), rnd(2, cout, seed=8)
    res = rnd(nb * h * w, cout, seed=6).half()
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    out, _ = ops.conv3x3(to_cl(x1), (nb, h, w), wk, bk, x2=to_cl(x2), row_bias=rb, rows_per_group=2 * h * w,
                         residual=res, out_fp32=True)
    ref = to_cl(F.conv2d(torch.cat([x1, x2], 1), wt, b, padding=1)).float() + rb.repeat_interleave(2 * h * w, 0) + res.float()
    close(out, ref, what="conv concat+rowbias+residual")


def test_conv_in_channel_padding():
    """conv_in: 8 real channels zero-padded to 64 on both the activation and the weight."""
    from insv2v import ops
    from insv2v.unet import prep_conv3x3
    nb, h, w = 2, 8, 8
    x = rnd(nb, 8, h, w).half().float()
    wt = rnd(64, 8, 3, 3, scale=0.1).half().float()
    b = rnd(64)
    wk, bk = prep_conv3x3({"c.weight": wt.cpu(), "c.bias": b.cpu()}, "c", dev())
    xin = ops.nchw_to_nhwc_f16(x, 64)
    out, _ = ops.conv3x3(xin, (nb, h, w), wk, bk)
    close(out, to_cl(F.conv2d(x, wt, b, padding=1)).float(), what="conv_in pad")


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("ns,rows,C,G,silu", [(3, 16 * 8 * 12, 320, 32, True), (48, 96, 64, 32, False), (2, 1000, 640, 32, True),
                                               (6, 24, 1280, 32, False), (1, 7, 2560, 32, True), (3, 384, 1280, 32, True),
                                               (3, 384, 2560, 32, True), (48, 384, 640, 32, False), (2, 409, 640, 32, True),
                                               (2, 410, 640, 32, True), (5, 96, 1280, 8, False)])
def test_groupnorm(ns, rows, C, G, silu):
    from insv2v import ops
    x = (rnd(ns * rows, C) * 2 + 3 * rnd(1, C, seed=2)).half()  # per-channel offsets: exercises the shifted variance
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    y = ops.groupnorm(x, ns, rows, gamma, beta, G, 1e-5, silu=silu)
    xr = x.float().reshape(ns, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xr, G, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    close(y, ref.permute(0, 2, 1).reshape(ns * rows, C), rel=4e-3, what="groupnorm")


@pytest.mark.parametrize("ns,rows,C,G,silu", [(960, 96, 1280, 32, False), (480, 24, 1280, 32, True), (128, 96, 1280, 32, True), (130, 48, 1280, 32, False),
                                               (256, 17, 1280, 32, False), (144, 30, 2560, 32, True), (200, 40, 640, 16, False)])
def test_groupnorm_per_frame_whole_sample_kernel(ns, rows, C, G, silu):
    """gn_frame_kernel: many samples of <= 245 KB each (the per-frame GroupNorm in front of the spatial transformers of the 8x12 / 4x6
    levels: 96 / 24 rows x 1280 channels) - one workgroup holds a whole sample in registers and reduces per group through LDS.  All three
    workgroup sizes (256 / 512 / 1024 threads), ragged chunk counts, 16 and 32 groups, per-channel offsets (centred variance), SiLU;
    a strided output/input view; and the result must not depend on how many samples the launch holds."""
    from insv2v import ops
    x = (rnd(ns * rows, C) * 2 + 3 * rnd(1, C, seed=2)).half()
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    y = ops.groupnorm(x, ns, rows, gamma, beta, G, 1e-5, silu=silu)
    xr = x.float().reshape(ns, rows, C).permute(0, 2, 1)
    ref = F.group_norm(xr, G, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    close(y, ref.permute(0, 2, 1).reshape(ns * rows, C), rel=4e-3, what="per-frame groupnorm")
    k = 128   # a launch of fewer samples (still the whole-sample kernel): bit-identical rows
    y2 = ops.groupnorm(x[:k * rows], k, rows, gamma, beta, G, 1e-5, silu=silu)
    assert torch.equal(y2, y[:k * rows])
    wide = torch.zeros(ns * rows, C + 64, device=dev(), dtype=torch.float16)   # a strided input view
    wide[:, :C] = x
    y3 = ops.groupnorm(wide[:, :C], ns, rows, gamma, beta, G, 1e-5, silu=silu)
    assert torch.equal(y3, y)


@pytest.mark.parametrize("ns,rows,C1,C2", [(3, 384, 1280, 1280), (3, 96, 1280, 640), (2, 50, 640, 640)])
def test_groupnorm_concat_small_slab_path(ns, rows, C1, C2):
    from insv2v import ops
    x1, x2 = (rnd(ns * rows, C1) + 2).half(), (rnd(ns * rows, C2, seed=5) - 1).half()
    gamma, beta = 1 + 0.1 * rnd(C1 + C2, seed=3), 0.1 * rnd(C1 + C2, seed=4)
    y = ops.groupnorm(x1, ns, rows, gamma, beta, 32, 1e-5, silu=True, x2=x2)
    xr = torch.cat([x1, x2], 1).float().reshape(ns, rows, C1 + C2).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xr, 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(ns * rows, C1 + C2)
    close(y, ref, rel=4e-3, what="groupnorm concat (small slab)")


def test_groupnorm_concat():
    from insv2v import ops
    ns, rows, C1, C2 = 2, 192, 128, 64
    x1, x2 = rnd(ns * rows, C1).half(), (rnd(ns * rows, C2, seed=5) + 1).half()
    gamma, beta = 1 + 0.1 * rnd(C1 + C2, seed=3), 0.1 * rnd(C1 + C2, seed=4)
    y = ops.groupnorm(x1, ns, rows, gamma, beta, 32, 1e-6, silu=True, x2=x2)
    xr = torch.cat([x1, x2], 1).float().reshape(ns, rows, C1 + C2).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xr, 32, gamma, beta, 1e-6)).permute(0, 2, 1).reshape(ns * rows, C1 + C2)
    close(y, ref, rel=4e-3, what="groupnorm concat")


@pytest.mark.parametrize("rows,C", [(1000, 320), (37, 1280), (4, 64), (129, 640)])
def test_layernorm(rows, C):
    from insv2v import ops
    x = (rnd(rows, C) + 0.5).half()
    g, b = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    y = ops.layernorm(x, g, b, 1e-5)
    close(y, F.layer_norm(x.float(), (C,), g, b, 1e-5), rel=4e-3, what="layernorm")


def test_layernorm_pe():
    from insv2v import ops
    B, Fr, HW, C = 2, 8, 12, 64
    x = rnd(B * Fr * HW, C).half()
    g, b = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    pe = rnd(32, C, seed=7)
    y = ops.layernorm(x, g, b, 1e-5, pe=pe, rows_per_frame=HW, frames=Fr, pe_start=3)
    ref = F.layer_norm(x.float(), (C,), g, b, 1e-5).reshape(B, Fr, HW, C) + pe[3:3 + Fr][None, :, None, :]
    close(y, ref.reshape(-1, C), rel=4e-3, what="layernorm+pe")


def test_softmax_rows():
    from insv2v import ops
    x = (rnd(3, 40, 96) * 3).half()
    ref = torch.softmax(x.float() * 0.7, -1)
    y = ops.softmax_rows(x.clone(), 0.7)
    close(y, ref, rel=4e-3, what="softmax")


# ------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, scale):  # [B,H,S,D]
    w = torch.softmax(q.float() @ k.float().transpose(-1, -2) * scale, -1)
    return w @ v.float()


@pytest.mark.parametrize("BF,HW,heads,hd", [(4, 96, 8, 40), (2, 384, 8, 80), (3, 24, 8, 160), (2, 1536, 2, 40), (5, 100, 4, 16),
                                             (2, 70, 2, 64), (2, 40, 1, 128), (5, 96, 8, 160), (3, 80, 8, 160), (2, 65, 3, 160)])
def test_attention_spatial_self(BF, HW, heads, hd):
    from insv2v import ops
    C = heads * hd
    qkv = rnd(BF * HW, 3 * C).half()
    out = torch.empty((BF * HW, C), device=dev(), dtype=torch.float16)
    p = qkv.data_ptr()
    ops.attention(p, p + 2 * C, p + 4 * C, out, batch=BF, heads=heads, head_dim=hd, seq_q=HW, seq_k=HW, scale=hd ** -0.5,
                  q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0),
                  o_addr=(1, HW * C, 0))
    t = qkv.reshape(BF, HW, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = attn_ref(t[0], t[1], t[2], hd ** -0.5).permute(0, 2, 1, 3).reshape(BF * HW, C)
    close(out, ref, rel=4e-3, what="self-attention")


def test_attention_peaked_softmax():
    """Large logits in a late KV tile force the online-softmax rescale branch."""
    from insv2v import ops
    BF, HW, heads, hd = 1, 256, 1, 64
    C = heads * hd
    qkv = rnd(BF * HW, 3 * C).half()
    qkv[200, C:2 * C] = qkv[5, :C] * 6  # key 200 strongly matches query 5
    out = torch.empty((BF * HW, C), device=dev(), dtype=torch.float16)
    p = qkv.data_ptr()
    ops.attention(p, p + 2 * C, p + 4 * C, out, batch=1, heads=1, head_dim=hd, seq_q=HW, seq_k=HW, scale=hd ** -0.5,
                  q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, q_addr=(1, 0, 0), kv_addr=(1, 0, 0), o_addr=(1, 0, 0))
    t = qkv.reshape(1, HW, 3, 1, hd).permute(2, 0, 3, 1, 4)
    ref = attn_ref(t[0], t[1], t[2], hd ** -0.5).permute(0, 2, 1, 3).reshape(HW, C)
    close(out, ref, rel=4e-3, what="attention rescale")


@pytest.mark.parametrize("hd,HW", [(40, 1536), (40, 200), (80, 384), (80, 136), (40, 1000)])
@pytest.mark.parametrize("case", ["rising", "late_spike", "small_steps", "very_negative", "wide"])
def test_attention_folded_softmax_extremes(hd, HW, case):
    """The d = 40 / 80 kernels fold the softmax shift into the MFMAs (-m~ in a padding column of Q) and only CHECK the maximum per
    tile (csrc/attention.hip, FOLD): cases built to stress exactly that, against fp32 torch on the same fp16 inputs -
      rising        key norms grow with the key index: the maximum moves in almost every 64-key tile (exact path every time);
      late_spike    one key in the last tile matches every query with a logit far above the rest (one big move at the end);
      small_steps   the maximum creeps up by less than 2^8 per tile: the lazy path keeps probabilities > 1 for many tiles;
      very_negative the first tile holds the largest logits by far, all later ones are tiny (nothing may underflow the row sum);
      wide          logits spread over +-40 (log2 units) at random.
    Ragged key counts (200, 136, 1000) put masked keys in the last tile."""
    from insv2v import ops
    heads, BF = 2, 2
    C = heads * hd
    g = torch.Generator(device="cpu").manual_seed(hd * 7 + HW)
    q = torch.randn(BF, HW, heads, hd, generator=g)
    k = torch.randn(BF, HW, heads, hd, generator=g)
    v = torch.randn(BF, HW, heads, hd, generator=g)
    idx = torch.arange(HW).view(1, HW, 1, 1).float()
    if case == "rising":
        k = k * (0.5 + 3.0 * idx / HW)
        q = q * 2
    elif case == "late_spike":
        k[:, HW - 3] = 6.0 * torch.sign(q[:, HW // 2])
        q = q.abs() * torch.sign(q[:, HW // 2]).unsqueeze(1)      # every query has the spike key's sign pattern
    elif case == "small_steps":
        k = k * 0.2 + q.mean(1, keepdim=True).sign() * (idx / HW) * 1.5
        q = q.abs() * q.mean(1, keepdim=True).sign()
    elif case == "very_negative":
        k = k * 0.05
        k[:, :8] = 4.0 * torch.sign(q[:, :8])
        q = q.abs() * torch.sign(q[:, :1])
    else:
        q, k = q * 3.5, k * 3.5
    qkv = torch.stack([q, k, v], 2).reshape(BF * HW, 3 * C).half().to(dev())
    out = torch.empty((BF * HW, C), device=dev(), dtype=torch.float16)
    p_ = qkv.data_ptr()
    ops.attention(p_, p_ + 2 * C, p_ + 4 * C, out, batch=BF, heads=heads, head_dim=hd, seq_q=HW, seq_k=HW, scale=hd ** -0.5,
                  q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, q_addr=(1, HW * 3 * C, 0), kv_addr=(1, HW * 3 * C, 0), o_addr=(1, HW * C, 0))
    t = qkv.reshape(BF, HW, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = attn_ref(t[0], t[1], t[2], hd ** -0.5).permute(0, 2, 1, 3).reshape(BF * HW, C)
    assert torch.isfinite(out).all()
    close(out, ref, rel=6e-3, abs_=2e-3, what=f"folded softmax d={hd} seq={HW} {case}")


@pytest.mark.parametrize("B,Fr,HW,heads,hd,L", [(3, 4, 96, 8, 40, 77), (2, 2, 24, 4, 16, 77), (1, 3, 50, 8, 160, 20), (2, 3, 96, 8, 160, 77), (1, 2, 72, 8, 160, 96)])
def test_attention_cross(B, Fr, HW, heads, hd, L):
    from insv2v import ops
    C = heads * hd
    q = rnd(B * Fr * HW, C).half()
    kv = rnd(B * L, 2 * C, seed=2).half()
    out = torch.empty_like(q)
    kp = kv.data_ptr()
    ops.attention(q.data_ptr(), kp, kp + 2 * C, out, batch=B * Fr, heads=heads, head_dim=hd, seq_q=HW, seq_k=L,
                  scale=hd ** -0.5, q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C, q_addr=(1, HW * C, 0),
                  kv_addr=(Fr, L * 2 * C, 0), o_addr=(1, HW * C, 0))
    qh = q.reshape(B, Fr, HW, heads, hd).permute(0, 1, 3, 2, 4)
    kh = kv[:, :C].reshape(B, 1, L, heads, hd).permute(0, 1, 3, 2, 4)
    vh = kv[:, C:].reshape(B, 1, L, heads, hd).permute(0, 1, 3, 2, 4)
    ref = attn_ref(qh, kh, vh, hd ** -0.5).permute(0, 1, 3, 2, 4).reshape(B * Fr * HW, C)
    close(out, ref, rel=4e-3, what="cross-attention")


@pytest.mark.parametrize("B,Fr,HW,heads,hd", [(3, 16, 24, 8, 40), (1, 8, 96, 4, 16), (2, 16, 6, 8, 160), (1, 24, 10, 2, 80), (1, 32, 4, 2, 32)])
def test_attention_temporal(B, Fr, HW, heads, hd):
    from insv2v import ops
    C = heads * hd
    qkv = rnd(B * Fr * HW, 3 * C).half()
    out = torch.zeros((B * Fr * HW, C), device=dev(), dtype=torch.float16)
    p = qkv.data_ptr()
    addr = (HW, Fr * HW * 3 * C, 3 * C)
    ops.attention(p, p + 2 * C, p + 4 * C, out, batch=B * HW, heads=heads, head_dim=hd, seq_q=Fr, seq_k=Fr, scale=hd ** -0.5,
                  q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, q_addr=addr, kv_addr=addr,
                  o_addr=(HW, Fr * HW * C, C))
    t = qkv.reshape(B, Fr, HW, 3, heads, hd).permute(3, 0, 2, 4, 1, 5)  # [3,B,HW,heads,F,hd]
    ref = attn_ref(t[0], t[1], t[2], hd ** -0.5).permute(0, 3, 1, 2, 4).reshape(B * Fr * HW, C)
    close(out, ref, rel=4e-3, what="temporal attention")


# (the last two: more problems than the persistent biased kernel has workgroups - 4 per CU -, so workgroups loop over problems)
@pytest.mark.parametrize("B,Fr,HW,heads,hd", [(2, 16, 6, 8, 160), (3, 16, 24, 8, 40), (1, 8, 96, 4, 16), (1, 12, 10, 8, 80), (2, 16, 1300, 8, 160), (5, 13, 500, 8, 40)])
def test_attention_temporal_frame_bias(B, Fr, HW, heads, hd):
    """insv2v_attention(q_bias / k_bias / v_bias): the per-frame positional-encoding bias of q, k and v added as the <= 16-row kernel loads
    the rows == the same attention on rows that carry the bias already (what the q/k/v GEMM's row bias produced before); the generic
    kernel must refuse the tables."""
    from insv2v import ops, _lib
    C = heads * hd
    qkv = rnd(B * Fr * HW, 3 * C).half()
    table = (rnd(Fr, 3 * C, seed=7) * 0.5).half()
    assert ops.attention_short_supported(heads, hd, Fr)
    frame = (torch.arange(B * Fr * HW, device=dev()) // HW) % Fr
    biased = (qkv + table[frame]).contiguous()          # fp16 add, as the kernel does
    addr = (HW, Fr * HW * 3 * C, 3 * C)
    kw = dict(batch=B * HW, heads=heads, head_dim=hd, seq_q=Fr, seq_k=Fr, scale=hd ** -0.5, q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C,
              o_rs=HW * C, q_addr=addr, kv_addr=addr, o_addr=(HW, Fr * HW * C, C))
    out = torch.zeros((B * Fr * HW, C), device=dev(), dtype=torch.float16)
    ref = torch.zeros_like(out)
    p, pb = qkv.data_ptr(), biased.data_ptr()
    ops.attention(p, p + 2 * C, p + 4 * C, out, qkv_bias=table, **kw)
    ops.attention(pb, pb + 2 * C, pb + 4 * C, ref, **kw)
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    t = biased.reshape(B, Fr, HW, 3, heads, hd).permute(3, 0, 2, 4, 1, 5)
    close(out, attn_ref(t[0], t[1], t[2], hd ** -0.5).permute(0, 3, 1, 2, 4).reshape(B * Fr * HW, C), rel=4e-3, what="temporal attention + frame bias")
    with pytest.raises(_lib.HipKernelError):            # 24 rows: the generic kernel has no bias path and must say so
        q24 = rnd(24 * HW, 3 * C).half()
        ops.attention(q24.data_ptr(), q24.data_ptr() + 2 * C, q24.data_ptr() + 4 * C, torch.zeros((24 * HW, C), device=dev(), dtype=torch.float16),
                      qkv_bias=(rnd(24, 3 * C) * 0.5).half(), batch=HW, heads=heads, head_dim=hd, seq_q=24, seq_k=24, scale=1.0, q_rs=HW * 3 * C,
                      k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, q_addr=addr, kv_addr=addr, o_addr=(HW, 24 * HW * C, C))


@pytest.mark.parametrize("NB,H,W,cin,cout", [(5, 32, 48, 320, 4), (3, 7, 5, 64, 3), (2, 16, 24, 128, 1)])
def test_conv3x3_narrow_vs_fp32_conv2d(NB, H, W, cin, cout):
    """ops.conv3x3_narrow (insv2v_gemm over the input channels for the nine taps' partial outputs + insv2v_tap_gather): the UNet's conv_out form
    (unet.py:432-434) against fp32 F.conv2d of the fp16-rounded operands, and against the implicit-GEMM form it replaces at large stacks."""
    from insv2v import ops
    x = rnd(NB * H * W, cin).half()
    w = (rnd(cout, cin, 3, 3, seed=5) * (9 * cin) ** -0.5).half()
    b = rnd(cout, seed=6) * 0.3
    got = ops.conv3x3_narrow(x, (NB, H, W), ops.tap_weights(w).to(dev()), b, cout)
    ref = F.conv2d(x.float().reshape(NB, H, W, cin).permute(0, 3, 1, 2), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(NB * H * W, cout)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    close(got, ref, rel=1e-3, what="narrow conv3x3")
    if cin % 64 == 0:
        direct, _ = ops.conv3x3(x, (NB, H, W), w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous(), b, out_fp32=True)
        close(got, direct, rel=1e-3, what="narrow conv3x3 vs implicit GEMM")


# ------------------------------------------------------------------------------------------- elementwise
def test_timestep_embedding():
    from insv2v import ops
    from oracle.leaves import timestep_sinusoid
    t = torch.tensor([981.0, 1.0, 500.0], device=dev())
    out = ops.timestep_embedding(t, 320, 0.0)
    ref = timestep_sinusoid(t.cpu(), 320, True, 0.0).to(dev())
    close(out, ref, rel=0, abs_=2e-3, what="timestep embedding")


def test_build_unet_input():
    from insv2v import ops
    Fr, h, w = 4, 8, 12
    lat, cond = rnd(Fr, 4, h, w), rnd(Fr, 4, h, w, seed=2)
    out = torch.full((3 * Fr * h * w, 64), 7.0, device=dev(), dtype=torch.float16)
    t = torch.zeros(3, device=dev())
    ops.build_unet_input(lat, cond, out, t, 981, 3)
    assert t.tolist() == [981.0] * 3
    o = out.reshape(3, Fr, h, w, 64).float()
    lat_cl, cond_cl = lat.permute(0, 2, 3, 1), cond.permute(0, 2, 3, 1)
    for br in range(3):
        close(o[br, ..., :4], lat_cl, what="input latent")
        close(o[br, ..., 4:8], cond_cl if br else torch.zeros_like(cond_cl), what="input cond")
    assert o[..., 8:].abs().max() == 0


def test_branch_major_strides():
    """The branch-major stack of clips (inference._stacked_gen, round 5): build_unet_input / cfg_stats / cfg_step address the branches of a
    clip through a stride; the values must be those of the back-to-back layout."""
    from insv2v import ops
    Fr, h, w, n, c = 4, 8, 12, 3, 1
    rows1 = Fr * h * w
    lat, cond = rnd(Fr, 4, h, w), rnd(Fr, 4, h, w, seed=2)
    flat = torch.zeros((3 * rows1, 64), device=dev(), dtype=torch.float16)
    ops.build_unet_input(lat, cond, flat, torch.zeros(3, device=dev()), 981, 3)
    stack = torch.full((3 * n * rows1, 64), 7.0, device=dev(), dtype=torch.float16)
    t = torch.zeros(3 * n, device=dev())
    ops.build_unet_input(lat, cond, stack[c * rows1:], t[c:], 981, 3, branch_rows=n * rows1, t_stride=n)
    sv = stack.reshape(3, n, rows1, 64)
    assert torch.equal(sv[:, c], flat.reshape(3, rows1, 64))
    assert (sv[:, 0] == 7.0).all() and (sv[:, 2] == 7.0).all()
    assert t.reshape(3, n)[:, c].tolist() == [981.0] * 3 and t.reshape(3, n)[:, 0].abs().max() == 0
    e, e_cl = _eps_cl(Fr, h, w)
    big = torch.zeros((3, n, rows1 * 4), device=dev())
    big[:, c] = e_cl.reshape(3, -1)
    s0, s1 = torch.empty(2, device=dev()), torch.empty(2, device=dev())
    ops.cfg_stats(e_cl, s0, Fr, h, w, 7.5, 1.5)
    ops.cfg_stats(big.reshape(-1)[c * rows1 * 4:], s1, Fr, h, w, 7.5, 1.5, branch_stride=n * rows1 * 4)
    assert torch.equal(s0, s1)
    o0, o1 = torch.empty_like(lat), torch.empty_like(lat)
    kw = dict(nbranch=3, text_cfg=7.5, img_cfg=1.5, sqrt_a=0.8, sqrt_1ma=0.6, coef=(0.3, 0.2, 0.1, 0.0))
    ops.cfg_step(e_cl, lat, latent_out=o0, **kw)
    ops.cfg_step(big.reshape(-1)[c * rows1 * 4:], lat, latent_out=o1, branch_stride=n * rows1 * 4, **kw)
    assert torch.equal(o0, o1)


def _eps_cl(Fr, h, w, seed=0):
    e = rnd(3, Fr, 4, h, w, seed=seed)
    return e, e.permute(0, 1, 3, 4, 2).contiguous()  # reference layout, channels-last


@pytest.mark.parametrize("mode", ["ddim", "correct", "ddpm", "rescale"])
def test_cfg_step(mode):
    from insv2v import ops
    from oracle import pipelines as op, schedulers as osch
    Fr, h, w, R = 8, 8, 12, 3
    e, e_cl = _eps_cl(Fr, h, w)
    lat, ref = rnd(Fr, 4, h, w, seed=1), rnd(R, 4, h, w, seed=2)
    tc, ic = 7.5, 1.5
    noise = e[0] + ic * (e[1] - e[0]) + tc * (e[2] - e[1])
    from insv2v.schedulers import DDIMScheduler, DDPMScheduler
    if mode == "ddpm":
        s, o = DDPMScheduler(), osch.DDPMScheduler()
    else:
        s, o = DDIMScheduler(), osch.DDIMScheduler()
    s.set_timesteps(10), o.set_timesteps(10)
    t = int(s.timesteps[2])
    co = s.coefficients(t)
    vn = rnd(Fr, 4, h, w, seed=5) if mode == "ddpm" else None
    stats = None
    if mode == "rescale":
        noise = op.rescale_noise_cfg(noise[None].cpu(), e[0][None].cpu(), 0.6)[0].to(dev())
        stats = torch.empty(2, device=dev())
        ops.cfg_stats(e_cl, stats, Fr, h, w, tc, ic)
        close(stats, torch.stack([e[0].std(), (e[0] + ic * (e[1] - e[0]) + tc * (e[2] - e[1])).std()]), rel=1e-4, abs_=1e-5, what="cfg stats")
    if mode == "correct":
        a = o.alphas_cumprod[t]
        d = (lat[:R] - (a ** 0.5) * ref) / ((1 - a) ** 0.5) - noise[:R]
        noise = torch.cat([noise[:R] + d, noise[R:] + d.mean(0, keepdim=True)], 0)
    so = o.step(noise.cpu(), t, lat.cpu(), **({"variance_noise": vn.cpu()} if mode == "ddpm" else {}))
    new, pred, eo = torch.empty_like(lat), torch.empty_like(lat), torch.empty_like(lat)
    ops.cfg_step(e_cl, lat, nbranch=3, text_cfg=tc, img_cfg=ic, sqrt_a=co["sqrt_a"], sqrt_1ma=co["sqrt_1ma"], coef=co["coef"],
                 latent_out=new, pred_x0=pred, eps_out=eo, latent_ref=ref if mode == "correct" else None,
                 correct=int(mode == "correct"), noise=vn, rescale_stats=stats, guidance_rescale=0.6 if mode == "rescale" else 0.0)
    close(eo, noise, rel=1e-5, abs_=1e-5, what="eps")
    close(pred, so.pred_original_sample.to(dev()), rel=1e-5, abs_=1e-4, what="pred_x0")
    close(new, so.prev_sample.to(dev()), rel=1e-5, abs_=1e-4, what="prev_sample")


def test_warp_resize_vs_golden(golden):
    from insv2v import flow_utils, synth
    g = golden("flow")
    img = synth.synth_input("flow.img", (4, 4, 32, 48))
    flow = synth.synth_input("flow.flow", (4, 2, 32, 48), scale=3.0)
    close(flow_utils.warp_image(img, flow), torch.from_numpy(g["warp"]).to(dev()), rel=1e-5, abs_=2e-5, what="warp")
    big = synth.synth_input("flow.big", (4, 2, 256, 384), scale=8.0)
    close(flow_utils.resize_flow(big, (32, 48)), torch.from_numpy(g["resize"]).to(dev()), rel=1e-5, abs_=2e-5, what="resize")
    odd = synth.synth_input("flow.odd", (2, 2, 50, 70), scale=8.0)
    close(flow_utils.resize_flow(odd, (32, 48)), torch.from_numpy(g["resize_odd"]).to(dev()), rel=1e-5, abs_=2e-5, what="resize odd")
    ident = flow_utils.warp_image(img, torch.zeros_like(flow))
    close(ident, img.to(dev()), rel=0, abs_=1e-4, what="zero-flow identity")


def test_flow_correction():
    from insv2v import ops
    from oracle.flow import warp_image
    Fr, R, h, w = 6, 2, 8, 12
    eps, lat, ref = rnd(Fr, 4, h, w), rnd(Fr, 4, h, w, seed=1), rnd(R, 4, h, w, seed=2)
    flows = rnd(Fr - R, R, 2, h, w, scale=3.0, seed=3)
    sa, sb = 0.8, 0.6
    out = ops.flow_correction(eps, lat, ref, flows, sa, sb)
    delta = ((lat[:R] - sa * ref) / sb - eps[:R]).cpu()
    for q in range(Fr - R):
        wd = warp_image(delta, flows[q].cpu())
        m = warp_image(torch.ones_like(delta)[:, :1], flows[q].cpu())
        ms = m.sum(0, keepdim=True)
        exp = torch.where(ms > 0.5, wd.sum(0, keepdim=True) / ms, torch.zeros(()))[0]
        close(out[q], exp.to(dev()), rel=1e-4, abs_=1e-4, what=f"flow correction q{q}")


def test_layout_and_posterior():
    from insv2v import ops
    x = rnd(2, 3, 8, 12)
    cl = ops.nchw_to_nhwc_f16(x, 8, 0.5)
    close(cl.reshape(2, 8, 12, 8)[..., :3], 0.5 * x.permute(0, 2, 3, 1), what="nchw->nhwc")
    assert cl.reshape(2, 8, 12, 8)[..., 3:].abs().max() == 0
    back = ops.nhwc_to_nchw_f32(cl, 2, 3, 8, 12, 2.0)
    close(back, x, what="nhwc->nchw")
    mom = rnd(2 * 4 * 6, 8)
    mom[:, 4:] *= 20
    noise = rnd(2, 4, 4, 6, seed=3)
    z = ops.posterior_sample(mom, noise, 2, 4, 6, 0.18215)
    m = mom.reshape(2, 4, 6, 8).permute(0, 3, 1, 2)
    ref = (m[:, :4] + torch.exp(0.5 * m[:, 4:].clamp(-30, 20)) * noise) * 0.18215
    close(z, ref, rel=1e-5, abs_=1e-5, what="posterior sample")


# ------------------------------------------------------------------------------------------- CLIP text-encoder kernels
@pytest.mark.parametrize("d,L", [(64, 77), (16, 77), (64, 130), (32, 16)])
def test_attention_causal(d, L):
    from insv2v import ops
    n, H = 3, 4
    C = H * d
    qkv = rnd(n * L, 3 * C).half()
    out = torch.empty((n * L, C), device=dev(), dtype=torch.float16)
    p = qkv.data_ptr()
    addr = (1, L * 3 * C, 0)
    ops.attention(p, p + 2 * C, p + 4 * C, out, batch=n, heads=H, head_dim=d, seq_q=L, seq_k=L, scale=d ** -0.5,
                  q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, q_addr=addr, kv_addr=addr, o_addr=(1, L * C, 0), causal=True)
    q, k, v = (t.float().reshape(n, L, H, d).transpose(1, 2) for t in qkv.split(C, dim=1))
    ref = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(n * L, C)
    close(out, ref, rel=4e-3, what=f"causal attention d={d} L={L}")


def test_gemm_quick_gelu_and_embed_tokens():
    from insv2v import ops
    a, w, b = rnd(154, 64).half(), rnd(128, 64, scale=0.125).half(), rnd(128, seed=3)
    y = a.float() @ w.float().t() + b
    close(ops.gemm(a, w, b, act=ops.ACT_QUICK_GELU), y * torch.sigmoid(1.702 * y), what="quick_gelu epilogue")
    tok, pos = rnd(50, 64).half(), rnd(77, 64, seed=5).half()
    ids = torch.randint(0, 50, (2, 77), device=dev())
    close(ops.embed_tokens(ids, tok, pos), (tok.float()[ids] + pos.float()[None]).reshape(-1, 64), what="embed_tokens")


# ------------------------------------------------------------------------------------------- C-ABI argument validation
def test_c_abi_rejects_bad_arguments_loudly():
    """Every entry point returns an error code (raised as HipKernelError) instead of launching on unsupported input."""
    from insv2v import ops, _lib
    E = _lib.HipKernelError
    q = rnd(64, 3 * 48).half()
    out = torch.empty((64, 48), device=dev(), dtype=torch.float16)
    p = q.data_ptr()
    kw = dict(batch=1, heads=1, seq_q=64, seq_k=64, scale=1.0, q_rs=144, k_rs=144, v_rs=144, o_rs=48,
              q_addr=(1, 0, 0), kv_addr=(1, 0, 0), o_addr=(1, 0, 0))
    with pytest.raises(E):  # head_dim not among the compiled sizes
        ops.attention(p, p + 96, p + 192, out, head_dim=48, **kw)
    with pytest.raises(E):  # head_dim not a multiple of 8
        ops.attention(p, p + 96, p + 192, out, head_dim=12, **kw)
    with pytest.raises(E):  # row stride not 16-byte aligned
        ops.attention(p, p + 96, p + 192, out, head_dim=40, **{**kw, "q_rs": 143})
    with pytest.raises(E):  # empty key sequence
        ops.attention(p, p + 96, p + 192, out, head_dim=40, **{**kw, "seq_k": 0})
    x = rnd(96, 60).half()
    with pytest.raises(E):  # channels not a multiple of 8
        ops.groupnorm(x, 1, 96, rnd(60), rnd(60), 4, 1e-5)
    x = rnd(96, 64).half()
    with pytest.raises(E):  # channels not divisible by groups
        ops.groupnorm(x, 1, 96, rnd(64), rnd(64), 5, 1e-5)
    with pytest.raises(E):  # fp32 activations are not accepted (no silent conversion / fallback)
        ops.layernorm(rnd(8, 64), rnd(64), rnd(64))
    with pytest.raises(E):  # folded LayerNorm needs the column sums
        ops.gemm(rnd(64, 64).half(), rnd(64, 64).half(), row_stats=torch.zeros(64, 2, device=dev()))
    with pytest.raises(E):  # unknown tile code
        ops.gemm(rnd(64, 64).half(), rnd(64, 64).half(), tile=77)


# ------------------------------------------------------------------------------------------- Winograd F(2x2, 3x3) convolution (round 6)
@pytest.mark.parametrize("NB,H,W,C1,C2,N,gn,rb,res,tile", [(6, 8, 12, 1280, 0, 1280, True, True, False, 0), (4, 16, 24, 640, 640, 640, True, False, True, 0),
                                                             (5, 4, 6, 1280, 0, 1280, False, False, True, 0), (3, 8, 12, 1280, 1280, 1280, True, True, True, 0),
                                                             (2, 16, 24, 1280, 640, 640, True, True, False, 240), (7, 8, 12, 1280, 0, 1280, False, False, False, 230),
                                                             (40, 4, 6, 1280, 0, 1280, True, True, True, 0), (2, 24, 32, 1280, 0, 640, True, True, True, 0), (1, 32, 48, 640, 640, 320, True, False, False, 0)])
def test_winograd_conv3x3_vs_fp32(NB, H, W, C1, C2, N, gn, rb, res, tile):
    """ops.winograd_conv3x3 (input transform with GroupNorm scale / shift + SiLU, the 16 transformed-tap GEMMs as one grouped launch of the
    ping-pong engine, output transform with bias + per-sample row bias + residual) against fp32 F.conv2d of the normalised input: two-source
    channel concat, tile counts that are not a multiple of 256 (padded groups), several images per LDS stage (4x6), images staged in bands of
    tile rows (24x32, 32x48), forced gemm_r8 / gemm_q8.
    Stated single-kernel tolerance 2e-3 of max|ref| (measured 6.5e-4: fp16 storage of V, U and M, profiles/r06_winograd_proto.txt)."""
    from insv2v import ops
    C, M = C1 + C2, NB * H * W
    ips = 2 if NB % 2 == 0 else 1          # images per GroupNorm sample
    x = rnd(M, C1, seed=1).half()
    x2 = rnd(M, C2, seed=2).half() if C2 else None
    w = rnd(N, C, 3, 3, scale=(9 * C) ** -0.5, seed=3)
    b = rnd(N, seed=4)
    assert ops.winograd_ok((NB, H, W), C, C1 if C2 else 0)
    U = ops.winograd_weights(w.cpu(), dev())
    ab = None
    xin = torch.cat([x, x2], 1).float() if C2 else x.float()
    if gn:
        ab = torch.stack([1.0 + 0.2 * rnd(NB // ips, C, seed=5), 0.3 * rnd(NB // ips, C, seed=6)], -1).contiguous()   # (scale, shift)
        sc = ab[..., 0].repeat_interleave(ips * H * W, 0), ab[..., 1].repeat_interleave(ips * H * W, 0)
        xin = F.silu(xin * sc[0] + sc[1])
    xin = xin.half().float()                # the kernel stages the normalised pixels in fp16
    tb = rnd(NB // ips, N, seed=7) if rb else None
    r = rnd(M, N, seed=8).half() if res else None
    out = ops.winograd_conv3x3(x, (NB, H, W), U, b, x2=x2, gn_ab=ab, gn_images_per_sample=ips, gn_silu=gn, row_bias=tb, rows_per_group=ips * H * W,
                               residual=r, tile=tile)
    ref = F.conv2d(xin.reshape(NB, H, W, C).permute(0, 3, 1, 2), w.half().float(), b, padding=1).permute(0, 2, 3, 1).reshape(M, N)
    if rb:
        ref = ref + tb.repeat_interleave(ips * H * W, 0)
    if res:
        ref = ref + r.float()
    close(out, ref, rel=2e-3, abs_=1e-3, what=f"winograd conv {NB}x{H}x{W} {C}->{N}")
    # the direct implicit-GEMM convolution of the same (normalised) input agrees within the two forms' fp16 effects
    direct, _ = ops.conv3x3(xin.half(), (NB, H, W), w.permute(0, 2, 3, 1).reshape(N, 9 * C).half().contiguous(), b, row_bias=tb, rows_per_group=ips * H * W, residual=r)
    close(out, direct, rel=3e-3, abs_=2e-3, what="winograd vs direct convolution")


def test_gemm_grouped_weights_vs_separate_launches():
    """insv2v_gemm with w_group_rows: every 256-aligned row group against its own weight matrix = separate launches on the row ranges, bit for
    bit (same kernel, same tiles), on gemm_r8 (N = 640) and gemm_q8 (N = 1280); invalid descriptors are refused."""
    from insv2v import ops, _lib
    lib = _lib.load()
    for N, tile in ((640, 0), (1280, 230), (1280, 240)):
        G, rows, K = 5, 512, 1280
        a = rnd(G * rows, K, seed=N).half()
        w = rnd(G, N, K, scale=K ** -0.5, seed=N + 1).half()
        out = torch.empty(G * rows, N, device=dev(), dtype=torch.float16)
        d = _lib.GemmDesc()
        d.a, d.w, d.c, d.lda, d.ldw, d.ldc = a.data_ptr(), w.data_ptr(), out.data_ptr(), K, K, N
        d.M, d.N, d.K, d.batch, d.alpha, d.tile = G * rows, N, K, 1, 1.0, tile
        d.w_group_rows, d.w_group_stride = rows, N * K
        _lib.check(lib.insv2v_gemm(ops._byref(d), ops._stream()), "grouped gemm")
        for g in range(G):
            one = ops.gemm(a[g * rows:(g + 1) * rows], w[g], tile=tile if tile else (240 if N % 320 == 0 else 230))
            assert torch.equal(out[g * rows:(g + 1) * rows], one), (N, tile, g)
            close(one, a[g * rows:(g + 1) * rows].float() @ w[g].float().t(), what="grouped gemm group")
        d.w_group_rows = 100
        assert lib.insv2v_gemm(ops._byref(d), ops._stream()) == -1          # not a multiple of 256
        d.w_group_rows, d.act = rows, ops.ACT_SILU
        assert lib.insv2v_gemm(ops._byref(d), ops._stream()) == -2          # nothing rides in a grouped product's epilogue


@pytest.mark.parametrize("NB,H,W,C,N,tile", [(6, 4, 6, 1280, 1280, 0), (5, 8, 12, 1280, 1280, 0), (3, 16, 24, 640, 640, 0), (9, 5, 7, 640, 640, 0), (2, 24, 32, 640, 640, 0), (1, 19, 40, 640, 320, 0)])
def test_winograd_upsample_conv3x3_vs_fp32(NB, H, W, C, N, tile):
    """Upsample3D (resnet.py:48-69: nearest x2, then a 3x3 convolution) as ONE Winograd pass over the low-resolution image: on the upsampled
    grid a tile's 4x4 patch has its two centre rows / columns equal, B^T d B has a zero row and column, 9 of the 16 transformed taps remain
    (one tile per input pixel, 4 x fewer MACs than the direct form).  Against fp32 F.conv2d of the upsampled image and against the direct
    implicit-GEMM convolution (index >> 1 gather); odd image sizes are fine here."""
    from insv2v import ops
    M = NB * H * W
    x = rnd(M, C, seed=1).half()
    w = rnd(N, C, 3, 3, scale=(9 * C) ** -0.5, seed=3)
    b = rnd(N, seed=4)
    assert ops.winograd_ok((NB, H, W), C, upsample=True)
    U = ops.winograd_weights(w.cpu(), dev(), upsample=True)
    assert tuple(U.shape) == (9, N, C)
    out = ops.winograd_conv3x3(x, (NB, H, W), U, b, upsample=True, tile=tile)
    up = F.interpolate(x.float().reshape(NB, H, W, C).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(up, w.half().float(), b, padding=1).permute(0, 2, 3, 1).reshape(4 * M, N)
    close(out, ref, rel=2e-3, abs_=1e-3, what=f"winograd upsample conv {NB}x{H}x{W} {C}->{N}")
    if C % 64 == 0:
        direct, geom = ops.conv3x3(x, (NB, H, W), w.permute(0, 2, 3, 1).reshape(N, 9 * C).half().contiguous(), b, upsample=True)
        assert geom == (NB, 2 * H, 2 * W)
        close(out, direct, rel=3e-3, abs_=2e-3, what="winograd vs direct upsample convolution")
