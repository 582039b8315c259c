/*
 * insv2v_hip.h -- C ABI of libinsv2v_hip.so: the MI355X (gfx950) kernels behind the
 * InsV2V denoising hot path.
 *
 * The reference (amazon-science/instruct-video-to-video) is pure Python/PyTorch and has no
 * FFI layer; the seam is the set of library ops its hot path dispatches.
 * Each entry point below replaces those ops for one stage of the path and cites the reference
 * call sites it stands in for (file:line relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated inside;
 *   - activations are fp16, channels-last: a video tensor (b,c,f,h,w) is stored as the token
 *     matrix [b*f*h*w, c]; weights are fp16 [out, in] (conv: [out, kh, kw, in]); biases,
 *     norm affine parameters and statistics are fp32;
 *   - all sizes are element counts, all strides are in elements;
 *   - return value 0 = launched, negative = rejected argument (INSV2V_E*), positive = hipError_t;
 *   - launches are asynchronous on `stream` and re-entrant per stream (graph-capturable);
 *   - one process drives ONE device (the deployment is one process per GPU): the launchers cache the CU count and kernel
 *     attributes of the first device they run on, and a call made with another device current returns INSV2V_EINVAL.
 */
#ifndef INSV2V_HIP_H
#define INSV2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* insv2v_stream_t; /* hipStream_t */

#define INSV2V_OK 0
#define INSV2V_EINVAL (-1)  /* bad shape / alignment / null pointer */
#define INSV2V_EUNSUPPORTED (-2)

#define INSV2V_ACT_NONE 0
#define INSV2V_ACT_SILU 1
#define INSV2V_ACT_GEGLU 2 /* W rows interleaved [h0..31,g0..31,h32..63,...]; out has N/2 columns */
#define INSV2V_ACT_QUICK_GELU 3 /* x * sigmoid(1.702 x): CLIP text MLP (transformers modeling_clip CLIPMLP, hidden_act "quick_gelu") */

/* activations of the optical-flow estimator's convolutions (torchvision raft_large behind flow_utils.py:134-189: ReLU after every
 * Conv2dNormActivation, sigmoid / tanh in the ConvGRU); insv2v_gemm LINEAR mode and, since round 6, CONV3X3 mode on the 128x128 tile kernel (the 3x3 layers of that network without im2col) */
#define INSV2V_ACT_RELU 4
#define INSV2V_ACT_SIGMOID 5
#define INSV2V_ACT_TANH 6

#define INSV2V_MODE_LINEAR 0
#define INSV2V_MODE_CONV3X3 1

/* ABI version, bumped on any struct change. */
int insv2v_abi_version(void);
/* One-time per-process setup (kernel attributes). Safe to call repeatedly. */
int insv2v_init(void);

/*
 * insv2v_gemm: C = epilogue(alpha * A x W^T), MFMA fp16 -> fp32 accumulate.
 *
 * LINEAR mode: A is [M,K] (row stride lda). If k_split > 0, columns [0,k_split) come from `a`
 *   and [k_split,K) from `a2` (row stride lda2): a channel concat that is never materialised.
 *   Replaces torch.nn.Linear / 1x1 Conv2d at attention.py:64,89 (proj_in/out), resnet.py:172,200
 *   (conv_shortcut), resnet.py:183 (time_emb_proj), unet.py:358-364 (TimestepEmbedding),
 *   motion_module.py:139,146, diffusers Attention.to_q/k/v/to_out and FeedForward
 *   (attention.py:160-190, motion_module.py:200,289-331), vqvae/model.py:149-172 (VAE attn 1x1).
 * CONV3X3 mode: implicit GEMM of a 3x3 convolution over an NHWC image [NB,IH,IW,*]:
 *   M = NB*OH*OW, K = 9*Cin, W is [N, 3,3,Cin]. Input pixel for output (oh,ow), tap (kh,kw) is
 *   (oh*stride+kh-pad_t, ow*stride+kw-pad_l), zero outside; with upsample=1 the input is first
 *   nearest-x2 upsampled (index>>1) without materialising it. Channel concat via k_split (=C1).
 *   Cin must be a multiple of 64 (pad channels on the host) and k_split a multiple of 64.
 *   Replaces F.conv2d at resnet.py:10-18 (InflatedConv3d), :59-69 (Upsample3D), :87-105
 *   (Downsample3D), unet.py:92,225 and vqvae/model.py:35-74,77-136 (VAE convs incl. the
 *   asymmetric (0,1,0,1) pad of Downsample: pad_t=pad_l=0, stride=2).
 * Epilogue (in this order): v = alpha*acc; [folded LayerNorm, see below]; += bias[n];
 *   += row_bias[g*ld_rb + n] with g = m / rows_per_group (the per-sample time embedding add of
 *   resnet.py:183-186), or g = (m / rows_per_group) % rb_mod when rb_mod > 0 (per-frame term);
 *   act (SiLU, or GEGLU h*gelu_erf(g)); += residual[m*ldr + n]; store fp16 (or fp32 if c_fp32).
 * Folded LayerNorm (row_stats != NULL): the GEMM consumes the RAW tokens x while computing
 *   LayerNorm(x) W^T:  with W' = W*diag(gamma) passed as `w`, col_sum[n] = sum_k W'[n,k] and
 *   row_stats[m] = (mean_m, rstd_m) from insv2v_layernorm_stats,
 *       v = rstd_m * (alpha*acc - mean_m * col_sum[n])          (= ((x-mean)*rstd*gamma) . W[n,:])
 *   the beta term (sum_k beta_k W[n,k]) is folded into `bias` by the caller, and the temporal positional
 *   encoding added after the norm (motion_module.py:277-278) becomes the per-frame row_bias table
 *   pe @ W^T.  Removes the normalised copy of the activations from HBM (attention.py:236-259,
 *   motion_module.py:206,214).
 * Batched: grid y = batch, operands advanced by *_bs elements per batch.
 * Split-K: when `workspace` is given (fp32 scratch of workspace_bytes) the library may split K over
 *   split_k workgroups per tile (0 = decide automatically: only for problems too small to fill the GPU
 *   with long K, e.g. the 4x6-latent UNet level) and reduce the fp32 partials in a second kernel that
 *   applies the epilogue; results are deterministic (fixed summation order).
 */
typedef struct insv2v_gemm_desc {
    const void* a;
    const void* a2;
    const void* w;
    void* c;
    const float* bias;
    const float* row_bias;
    const void* residual;
    const float* row_stats; /* [M][2] (mean, rstd) or NULL */
    const float* col_sum;   /* [N] or NULL (required with row_stats) */
    int64_t lda, lda2, ldw, ldc, ldr, ld_rb;
    int64_t a_bs, w_bs, c_bs, r_bs;
    void* workspace;
    int64_t workspace_bytes;
    int32_t M, N, K;
    int32_t k_split;
    int32_t rows_per_group;
    int32_t rb_mod;
    int32_t act;
    int32_t c_fp32;
    int32_t mode;
    int32_t NB, IH, IW, OH, OW, Cin;
    int32_t stride, pad_t, pad_l, upsample;
    int32_t batch;
    int32_t tile; /* 0 = auto; low digit = tile shape, tens digit = LDS ring depth (tools/bench_gemm.py) */
    int32_t split_k; /* 0 = auto (needs workspace), 1 = never, S > 1 = force */
    float alpha;
    /* Fused input GroupNorm (+SiLU) of ResnetBlock3D (resnet.py:177-178,188-194), CONV3X3 mode on the patch-tiled kernel
     * only: the convolution reads the RAW tensor and applies y = act(x*scale + shift) to its input patch in LDS, so the
     * normalised copy never exists in HBM.  gn_ab = [nsamples][Cin][2] fp32 (scale, shift) from insv2v_groupnorm
     * (stats_only); image nb belongs to sample nb / gn_images_per_sample.  Zero padding stays zero.  A call that sets gn_ab
     * but cannot run on the patch-tiled kernel is rejected (INSV2V_EUNSUPPORTED), never silently un-normalised: ask
     * insv2v_conv3x3_fuses_groupnorm first. */
    const float* gn_ab;
    int32_t gn_images_per_sample;
    int32_t gn_silu;
    /* LayerNorm statistics from the PRODUCER instead of a re-read (attention.py:236-259, motion_module.py:206,214: every
     * LayerNorm input is the output of an N = C GEMM).  stats_out != NULL (LINEAR mode, fp16 output): the epilogue also writes,
     * per output row and per column tile tn of insv2v_gemm_stats_parts() tiles, the partial sums (sum v, sum v^2) of the
     * values it stores (128x128 tile: after the fp16 rounding; 256x320 ping-pong tile, 160-column parts: before it):
     * stats_out[(tn*M + m)*2 + {0,1}] fp32 (tile-major, deterministic).  A consumer passes that buffer as
     * row_stats with stats_parts = number of tiles (0 = row_stats already holds (mean, rstd)) and ln_eps: it finalises
     * mean = S1/K, rstd = rsqrt(S2/K - mean^2 + eps) itself (K = the LayerNorm width); kernels that need (mean, rstd) pairs get
     * them from a small internal finalize launch into stats_scratch [M][2] (required with stats_parts > 0). */
    float* stats_out;
    float* stats_scratch;
    int32_t stats_parts;
    float ln_eps;
    /* Grouped weights (ABI 11; LINEAR mode on the 256-row ping-pong kernels only): row block g = m / w_group_rows of A / C is multiplied
     * by the weight matrix at w + g * w_group_stride elements (same [N, K] shape and ldw).  w_group_rows must be a multiple of 256 (a tile
     * never straddles two groups); 0 = one weight matrix.  This is the batched product of the Winograd convolution (insv2v_winograd_*:
     * 16 transformed-tap GEMMs as ONE launch); bias / row_bias / residual / statistics / activation are not combined with it. */
    int64_t w_group_stride;
    int32_t w_group_rows;
    int32_t reserved0;
} insv2v_gemm_desc;
int insv2v_gemm(const insv2v_gemm_desc* d, insv2v_stream_t stream);
/* Operand windows (ABI 10).  The kernels address every operand through 32-bit byte offsets inside one 2 GiB buffer descriptor.  A
 * problem whose activations (`a`, `a2`), output or residual reach beyond that window is run by insv2v_gemm as several launches over row
 * ranges (LINEAR) or image ranges (CONV3X3), each inside the window and aligned to the row-bias groups (rows_per_group x rb_mod) and the
 * GroupNorm samples (gn_images_per_sample); partial row statistics (stats_parts > 0) are finalised over the whole problem first.  Such a
 * problem cannot emit stats_out (INSV2V_EUNSUPPORTED, like every problem that cannot), is not batched and takes no forced split-K.
 * insv2v_set_operand_window(bytes) replaces the window size (0 = default: 2 GiB less 1 MiB) and returns the previous value; it exists so
 * that tests can drive small problems through the split path - the product never calls it. */
int64_t insv2v_set_operand_window(int64_t bytes);
/* Number of column tiles (= partial statistics per row) insv2v_gemm will write for this problem when stats_out is set;
 * 0 if the problem cannot emit statistics (the caller then uses insv2v_layernorm_stats on the output). */
int insv2v_gemm_stats_parts(const insv2v_gemm_desc* d);
/* 1 if insv2v_gemm would run this CONV3X3 problem on the patch-tiled kernel, i.e. accepts gn_ab; else 0. */
int insv2v_conv3x3_fuses_groupnorm(const insv2v_gemm_desc* d);

/*
 * Winograd F(2x2, 3x3) form of a stride-1, pad-1 3x3 convolution (ResnetBlock3D conv1 / conv2 with wide inputs, resnet.py:143,159 behind
 * InflatedConv3d resnet.py:10-18; ABI 11).  y = A^T [ sum_c (G g G^T) . (B^T d B) ] A per 2x2 output tile: 2.25 x fewer MACs.  Three calls:
 *   insv2v_winograd_input  : x [NB*H*W, C] (channel concat x | x2 at C1; optional GroupNorm scale / shift table gn_ab [nsamples][C][2] as
 *                            from insv2v_groupnorm(stats_only), then SiLU - resnet.py:177-178,188; zero padding applies AFTER the norm)
 *                            -> v = 16 matrices V_k [tiles, C], matrix k = i*4 + j at row offset k * v_group_rows (tiles = NB*(H/2)*(W/2),
 *                            tile order (image, ty, tx); v_group_rows >= tiles, a multiple of 256 for the grouped GEMM)
 *   insv2v_gemm            : a = v [16 * v_group_rows, C], w = U [16][Cout][C] (U_k = (G g G^T)_k, host), w_group_rows = v_group_rows,
 *                            w_group_stride = Cout * C  ->  m [16 * v_group_rows, Cout] fp16
 *   insv2v_winograd_output : m -> y [NB*H*W, Cout] fp16 = A^T M A + bias[Cout] + row_bias[(pixel / rows_per_group) * ld_rb + n]
 *                            (time embedding, resnet.py:183-186) + residual[pixel * ldr + n]
 * H, W even; C a multiple of 64 (C1 too); W <= 128 (an image's 64-channel slice is staged in LDS whole, or in bands of tile rows with a
 * one-pixel halo); else INSV2V_EUNSUPPORTED and the caller uses insv2v_gemm CONV3X3.  fp16 storage of V, U, M: 6.5e-4 of max|ref| vs fp32 conv2d (profiles/r06_winograd_proto.txt).
 */
typedef struct insv2v_winograd_in_desc {
    const void* x;
    const void* x2;      /* or NULL */
    const float* gn_ab;  /* or NULL: no normalisation */
    void* v;
    int64_t ldx, ldx2;
    int64_t v_group_rows;
    int32_t NB, H, W, C, C1;
    int32_t gn_images_per_sample;
    int32_t gn_silu;
    int32_t upsample;    /* 1: the convolution runs on the nearest-x2 upsampled image (Upsample3D, resnet.py:48-69): one tile per INPUT pixel,
                          *    9 matrices (groups g = ci*3 + cj over patch indices {0, 1, 3}^2; the others are identically zero), H, W need not be even */
} insv2v_winograd_in_desc;
int insv2v_winograd_input(const insv2v_winograd_in_desc* d, insv2v_stream_t stream);
typedef struct insv2v_winograd_out_desc {
    const void* m;
    const float* bias;      /* or NULL */
    const float* row_bias;  /* or NULL */
    const void* residual;   /* or NULL */
    void* y;
    int64_t m_group_rows;
    int64_t ldr, ldy, ld_rb;
    int32_t NB, H, W, Cout;   /* H, W: the INPUT image (upsample: y has NB * 2H * 2W rows) */
    int32_t rows_per_group;
    int32_t upsample;
} insv2v_winograd_out_desc;
int insv2v_winograd_output(const insv2v_winograd_out_desc* d, insv2v_stream_t stream);

/*
 * insv2v_ffn_fused: out = x + FeedForward_geglu(LayerNorm(x)) as ONE kernel, activations resident in registers, the hidden layer
 * never written (diffusers FeedForward behind norm3 / ff_norm: attention.py:259, motion_module.py:214,
 * i.e. LayerNorm -> Linear(C, 8C) -> h * gelu_erf(g) -> Linear(4C, C) -> + residual).  Supported: C = 320, hidden = 1280 (UNet level
 * 0); other widths return INSV2V_EUNSUPPORTED and the caller uses insv2v_gemm twice.
 * wstream: the layer's weights and biases (LayerNorm gamma / beta folded into the first projection) as ONE fp16 buffer of
 *   insv2v_ffn_stream_elems(C, hidden) elements in MFMA-fragment order - the order the kernel consumes it; layout documented in
 *   insv2v/fused.py (pack_ffn_stream), which is the reference packer.
 */
typedef struct insv2v_ffn_desc {
    const void* x;       /* [M, C] fp16, row stride ldx */
    void* out;           /* [M, C] fp16, row stride ldo (may not alias x) */
    const void* wstream; /* fp16 fragment stream */
    int64_t ldx, ldo;
    int32_t M, C, hidden;
    float eps;           /* LayerNorm eps */
    /* post != 0: the transformer module's trailing Linear (proj_out, attention.py:89 / motion_module.py:146) rides behind the feed-forward:
     *   out = Wp (x + FF(LN(x))) + bp + post_residual, Wp [C, C] appended to wstream (pack_ffn_stream(post=...)); the feed-forward result
     *   stays in registers.  post_residual: [M, C] fp16 (the module's input), row stride ld_post. */
    const void* post_residual;
    int64_t ld_post;
    int32_t post;
} insv2v_ffn_desc;
int insv2v_ffn_fused(const insv2v_ffn_desc* d, insv2v_stream_t stream);
int64_t insv2v_ffn_stream_elems(int32_t C, int32_t hidden, int32_t post);

/*
 * insv2v_rowlin: out = [LayerNorm](x) W^T + bias [+ residual] for the K = 320 / 640 Linear / 1x1-conv layers (UNet levels 0-1), activations
 * resident in registers (same machinery as insv2v_ffn_fused; csrc/fused_rows.hip): Transformer3DModel / TemporalTransformer3DModel
 * proj_in / proj_out (attention.py:64,89; motion_module.py:139,146), Attention.to_q / to_k / to_v / to_out
 * (attention.py:160-190, motion_module.py:289-331).  K must be 320 or 640 and N a multiple of 64, else INSV2V_EUNSUPPORTED (use insv2v_gemm).
 *   layernorm != 0: x is normalised per row in registers (no affine: gamma is folded into W and beta into the bias by the caller,
 *     exactly as for insv2v_gemm's folded LayerNorm) - no statistics pass and no row_stats;
 *   frame_bias != 0: the bias of row m is row (m / rows_per_frame) % frames of a per-frame table (the temporal positional encoding
 *     added after the norm, motion_module.py:277-278, pushed through W; frames <= 16), else a plain bias vector; both live in wstream;
 *   residual (optional): [M, N] fp16 added before the fp16 store.
 * wstream: insv2v_rowlin_stream_elems(N, K) fp16 elements in MFMA-fragment order (insv2v/fused.py pack_linear_stream).
 */
typedef struct insv2v_rowlin_desc {
    const void* x;        /* [M, K] fp16 */
    void* out;            /* [M, N] fp16 */
    const void* residual; /* [M, N] fp16 or NULL */
    const void* wstream;
    int64_t ldx, ldo, ldr;
    int32_t M, N, K;
    int32_t layernorm, frame_bias, rows_per_frame, frames;
    float eps;
    float* stats_out;     /* optional [M][2] fp32: (mean, rsqrt(var + stats_eps)) of the OUTPUT rows, i.e. finished LayerNorm statistics
                             for a following insv2v_gemm(row_stats=...) - a wave stores whole rows, so no partial sums are needed */
    float stats_eps;
    /* gn_ab != NULL (plain form only: no layernorm / frame_bias / residual): a preceding GroupNorm is applied to x on the fly,
     * x <- x * scale + shift with [samples][K][2] fp32 (scale, shift) pairs from insv2v_groupnorm(stats_only); row m belongs to sample
     * m / gn_rows (gn_rows % 32 == 0).  The transformer blocks' GroupNorm -> proj_in pair (attention.py:101-103, motion_module.py:136-139)
     * without the normalised copy. */
    const float* gn_ab;
    int32_t gn_rows;
} insv2v_rowlin_desc;
int insv2v_rowlin(const insv2v_rowlin_desc* d, insv2v_stream_t stream);
int64_t insv2v_rowlin_stream_elems(int32_t N, int32_t K);

/*
 * insv2v_tattn_fused: one temporal self-attention sub-block of TemporalTransformerBlock (motion_module.py:206,270-336:
 * LayerNorm -> (+ positional encoding) -> to_q / to_k / to_v -> scaled-dot-product attention over the frames of every pixel ->
 * to_out -> + residual) as ONE register-resident launch.  Supported: C = 320, 8 heads, exactly 16 frames (UNet level 0, every
 * 16-frame window); anything else returns INSV2V_EUNSUPPORTED and the caller uses insv2v_rowlin / insv2v_gemm + insv2v_attention.
 * x / out: token matrices [samples * 16 * HW, C] fp16 with rows ordered (sample, frame, pixel).  wstream: insv2v_tattn_stream_elems()
 * fp16 elements from insv2v/fused.py pack_tattn_stream (q/k/v weights with the LayerNorm gamma folded in, the per-frame bias table
 * = positional-encoding rows pushed through the weights + W beta, the output projection and its bias).
 */
typedef struct insv2v_tattn_desc {
    const void* x;
    void* out;
    const void* wstream;
    int64_t ldx, ldo;
    int32_t samples, HW, C, heads, frames;
    float eps;   /* LayerNorm eps */
    float scale; /* softmax scale, head_dim^-0.5 */
} insv2v_tattn_desc;
int insv2v_tattn_fused(const insv2v_tattn_desc* d, insv2v_stream_t stream);
int64_t insv2v_tattn_stream_elems(int32_t C, int32_t heads, int32_t frames);
/* The same sub-block at C = 640 (8 heads x 80, 16 frames) WITHOUT the output projection (ABI 8): out [rows, C] = the attention output
 * (LayerNorm -> q/k/v with the per-frame bias -> softmax over the frames -> P.V); to_out + residual follow as insv2v_rowlin.  q, k and v
 * never exist in memory.  Same descriptor; wstream: insv2v_tattn_attn_stream_elems(C, heads, frames) halfs (insv2v/fused.py
 * pack_tattn_qkv_stream). */
int insv2v_tattn_attn(const insv2v_tattn_desc* d, insv2v_stream_t stream);
int64_t insv2v_tattn_attn_stream_elems(int32_t C, int32_t heads, int32_t frames);

/*
 * Fused text cross-attention sub-block of BasicTransformerBlock (attention.py:249-257 = norm2 -> attn2 -> + residual, over diffusers
 * Attention: to_q of the tokens, to_k / to_v of the text context, to_out[0]) for C = 320, 8 heads x 40, 64 < ctx_len <= 96:
 *     out = x + Wo . softmax_keys( (LayerNorm(x) Wq^T + Wq beta) K_b^T * scale ) V_b + bo,   b = row / rows_per_sample
 * replaces insv2v_rowlin (q) + insv2v_attention + insv2v_rowlin (out-proj, residual).  x / out: [M, C] fp16 (row strides ldx / ldo;
 * out must not alias x), rows_per_sample a multiple of 128, M a multiple of rows_per_sample.
 * wstream: insv2v_xattn_stream_elems(C, heads, 0) halfs - q projection (LayerNorm gamma folded in, bias = Wq beta) and output
 * projection as MFMA fragments; kvstream: [M / rows_per_sample] x insv2v_xattn_stream_elems(C, heads, 1) halfs - the text K / V of each
 * sample as fragments, masked per head and zero beyond ctx_len (insv2v/fused.py pack_xattn_stream / pack_xattn_kv).  The K / V of the
 * text are loop-invariant over the sampling loop, so the second stream is built once per prompt.
 */
typedef struct insv2v_xattn_desc {
    const void* x;
    void* out;
    const void* wstream;
    const void* kvstream;
    int64_t ldx, ldo;
    int32_t M, rows_per_sample, C, heads, ctx_len;
    float eps;   /* LayerNorm eps */
    float scale; /* softmax scale, head_dim^-0.5 */
    /* optional (insv2v_xattn_fused only): the out-projection of the preceding SELF-attention rides in front (attention.py:244-247):
     * x is then that attention's output, pre_residual [M, C] (row stride ld_pre) the tokens it is added to, and
     *     x1 = Wo1 . x + bo1 + pre_residual;   out = x1 + Wo . Attn(LayerNorm(x1) ...) + bo
     * with x1 never written to memory; wstream = insv2v_xattn_stream_elems(C, heads, 2) halfs (pack_xattn_stream(pre=...)). */
    const void* pre_residual;
    int64_t ld_pre;
} insv2v_xattn_desc;
int insv2v_xattn_fused(const insv2v_xattn_desc* d, insv2v_stream_t stream);
int64_t insv2v_xattn_stream_elems(int32_t C, int32_t heads, int32_t per_sample_kv);
/* The same sub-block at C = 640 (8 heads x 80) WITHOUT the output projection (ABI 8): out [M, C] = the attention output; to_out + residual
 * follow as insv2v_rowlin.  wstream: insv2v_xattn_attn_stream_elems(C, heads, 0) halfs (q projection only), kvstream: per sample
 * insv2v_xattn_attn_stream_elems(C, heads, 1) halfs (insv2v/fused.py pack_xattn_q_stream / pack_xattn640_kv). */
int insv2v_xattn_attn(const insv2v_xattn_desc* d, insv2v_stream_t stream);
int64_t insv2v_xattn_attn_stream_elems(int32_t C, int32_t heads, int32_t per_sample_kv);

/*
 * GroupNorm (+ optional SiLU) over channels-last data, both reduction domains of the path:
 *   5-D GroupNorm of ResnetBlock3D / conv_norm_out (resnet.py:177-178,188,194; unet.py:427-428):
 *     nsamples = b, rows_per_sample = f*h*w (statistics span all frames);
 *   per-frame GroupNorm of Transformer3DModel / TemporalTransformer3DModel and the VAE
 *     (attention.py:101; motion_module.py:136; vqvae/model.py:32-33): nsamples = b*f, rows = h*w.
 * Input may be a channel concat of two tensors (x: C1 channels, x2: C-C1 channels).
 * Pass 1 writes per-(sample,chunk,group) partial (sum, M2) to `partials`
 *   [nsamples, nchunks, G, 3] (count, mean, M2) fp32 placed after a [nsamples, G, 2] (mean, rstd) header in `partials`
 *   (total nsamples*G*(2+3*nchunks) floats); pass 2 merges them (Chan) into the header; pass 3 applies
 *   y = act((x-mean)*rstd*gamma+beta) into y [rows, C] fp16 (row stride ldy).
 */
typedef struct insv2v_groupnorm_desc {
    const void* x;
    const void* x2;
    void* y;
    const float* gamma;
    const float* beta;
    float* partials;
    int64_t ldx, ldx2, ldy;
    int32_t nsamples, rows_per_sample, C, C1, G;
    int32_t nchunks; /* chunks per sample used by pass 1 (>=1) */
    int32_t silu;
    float eps;
    /* stats_only != 0: passes 1-2 only; writes ab[nsamples][C][2] = (rstd*gamma, beta - mean*rstd*gamma) for a consumer
     * that applies y = act(x*scale + shift) itself (insv2v_gemm gn_ab); y may be NULL. */
    float* ab;
    int32_t stats_only;
} insv2v_groupnorm_desc;
int insv2v_groupnorm(const insv2v_groupnorm_desc* d, insv2v_stream_t stream);

/*
 * LayerNorm over the channel axis of a token matrix [rows, C] (fp16 in/out, fp32 statistics,
 * eps inside the sqrt), optionally followed by the temporal positional-encoding add of
 * motion_module.py:236-242,277-278: y += pe[(row / rows_per_frame) % frames + pe_start, :]
 * (pe fp32 [max_len, C]).  Replaces nn.LayerNorm at attention.py:236,249,259 and
 * motion_module.py:206,214.
 */
typedef struct insv2v_layernorm_desc {
    const void* x;
    void* y;
    const float* gamma;
    const float* beta;
    const float* pe;
    int64_t ldx, ldy;
    int32_t rows, C;
    int32_t rows_per_frame, frames, pe_start;
    float eps;
} insv2v_layernorm_desc;
int insv2v_layernorm(const insv2v_layernorm_desc* d, insv2v_stream_t stream);
/* Per-token LayerNorm statistics only: stats[m] = (mean, rsqrt(var + eps)) of row m of x [rows, C] fp16.
 * Feeds the folded-LayerNorm epilogue of insv2v_gemm. */
int insv2v_layernorm_stats(const void* x, float* stats, int64_t ldx, int32_t rows, int32_t C, float eps,
                           insv2v_stream_t stream);

/*
 * Fused multi-head attention O = softmax(Q K^T * scale) V (flash style, MFMA for both
 * contractions, wave shuffles for the softmax reductions, fp32 online softmax).
 * Problem z in [0,batch): operand base = ptr + (z / *_inner) * *_outer + (z % *_inner) * *_step
 *   + head * head_dim; rows are *_rs apart.  This one addressing rule covers
 *   - spatial self-attention (attention.py:244 -> diffusers Attention, xformers in the reference):
 *     z = b*f, rows = h*w tokens;
 *   - text cross-attention (attention.py:251-256): K/V of sample z/f (kv_inner = f, kv_step = 0),
 *     seq_k = 77;
 *   - temporal self-attention over frames (motion_module.py:270-336, F.scaled_dot_product_attention
 *     in the reference): z = (b, pixel), rows = frames, row stride = h*w*ld.
 * head_dim must be a multiple of 8 and <= 160; seq_k >= 1.
 */
typedef struct insv2v_attention_desc {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    int64_t q_rs, k_rs, v_rs, o_rs;
    int64_t q_outer, q_step, kv_outer, kv_step, o_outer, o_step;
    int32_t q_inner, kv_inner, o_inner;
    int32_t batch, heads, head_dim, seq_q, seq_k;
    float scale;
    int32_t causal; /* != 0: key j is visible to query i only if j <= i (CLIP text encoder, modules/openclip/modules.py:118) */
    /* ABI 8, optional (all three or none): fp16 tables [seq, ...] added to the rows of q / k / v as they are loaded - row i of the
     * sequence gets q_bias[i * bias_rs + head * head_dim + c].  This is the temporal positional encoding pushed through to_q / to_k /
     * to_v (motion_module.py:277-278 adds pe AFTER the norm, so it reaches all three): with the add here, the q/k/v projection in
     * front needs no per-frame row bias and may run on the persistent GEMM kernel.  Only the <= 16-row form (seq_q, seq_k <= 16, the
     * temporal attention over the frames) takes it; INSV2V_EUNSUPPORTED otherwise. */
    const void* q_bias;
    const void* k_bias;
    const void* v_bias;
    int64_t bias_rs;
} insv2v_attention_desc;
int insv2v_attention(const insv2v_attention_desc* d, insv2v_stream_t stream);

/*
 * Token + position embedding of the CLIP text encoder (transformers CLIPTextEmbeddings, called from
 * modules/openclip/modules.py:114-118): out[r, :] = tok[ids[r], :] + pos[r % L, :], fp16 tables, fp32 add, fp16 out.
 * ids: int64 device array of `rows` = n*L token ids, each in [0, vocab) (validated by the caller on the host).
 */
int insv2v_embed_tokens(const int64_t* ids, const void* tok, const void* pos, void* out, int32_t rows, int32_t L,
                        int32_t C, int32_t vocab, insv2v_stream_t stream);

/* Row softmax of an fp16 matrix [rows, cols] in place-capable form (VAE AttnBlock,
 * vqvae/model.py:186-188): y = softmax(x * scale) along cols. */
int insv2v_softmax_rows(const void* x, void* y, int64_t ldx, int64_t ldy, int32_t rows, int32_t cols,
                        float scale, insv2v_stream_t stream);

/* Sinusoidal timestep features (diffusers Timesteps(dim, flip_sin_to_cos=True, shift) as used at
 * unet.py:95,358): out[b, :] = [cos(t*f_k), sin(t*f_k)] fp16, t read from device memory t[b] (fp32). */
int insv2v_timestep_embedding(const float* t, void* out, int32_t batch, int32_t dim, float shift,
                              insv2v_stream_t stream);

/*
 * (ABI 12) Second half of a 3x3 convolution (stride 1, zero padding 1) with <= 4 OUTPUT channels - the UNet's conv_out
 * (unet.py:432-434: InflatedConv3d(320, 4, 3, padding=1), resnet.py:14-21) at the stacked clip counts.  The first half is ONE
 * plain insv2v_gemm: y9[p, 4 t + c] = sum_ci x[p, ci] * W[c, ci, ky, kx], t = 3 ky + kx (fp32 out, >= 36 columns); this call adds the
 * nine shifted taps: out[p, c] = bias[c] + sum_t y9[p + (ky - 1) W + (kx - 1), 4 t + c] over the taps inside the image (fp32,
 * c < cout <= 4).  ld9 / ldo in fp32 elements, ld9 a multiple of 4, y9 16-byte aligned.  NB images of H x W pixels, row-major.
 */
int insv2v_tap_gather(const float* y9, int64_t ld9, const float* bias, float* out, int64_t ldo, int32_t NB, int32_t H, int32_t W,
                      int32_t cout, insv2v_stream_t stream);

/*
 * Build the 3-way classifier-free-guidance UNet input of inference.py:183-189 in channels-last
 * fp16: out[3, F, h, w, ldo] with channels [latent(4) | 0 or img_cond(4) | zero pad], from
 * latent/img_cond fp32 in the reference layout [F, 4, h, w].  Also writes `timestep` to t_out[0..2].
 * nbranch = 3 (text/video CFG) or 1 (CFG off: latent | img_cond).
 */
int insv2v_build_unet_input(const float* latent, const float* img_cond, void* out, float* t_out,
                            float timestep, int32_t nbranch, int32_t F, int32_t h, int32_t w, int32_t ldo,
                            int64_t branch_rows, int32_t t_stride, insv2v_stream_t stream);
/* (ABI 10) branch_rows = rows of `out` between consecutive branches (0 = F*h*w: the clip's branches back to back), t_stride = entries
 * of t_out between them (0 = 1): a stack of n clips laid out BRANCH-major ([branch][clip] samples, so that the two branches that share
 * their UNet input - (no text, video) and (text, video) - form one contiguous range) passes n*F*h*w and n.  The same stride, in fp32
 * elements, is insv2v_step_desc.branch_stride / the branch_stride argument of insv2v_cfg_stats for the UNet output. */

/*
 * Fused CFG combine + long-video noise correction + scheduler step (inference.py:198-210,
 * :270-277; diffusers DDIMScheduler.step / DDPMScheduler.step):
 *   eps = n1 + img_cfg*(n2-n1) + text_cfg*(n3-n2)            (nbranch==3; else eps = n1)
 *   [guidance rescale, inference.py:13-24, when rescale > 0: needs eps_stats from insv2v_cfg_stats]
 *   [correction, when R > 0: d_r = (latent_r - sqrt_a*latent_ref_r)/sqrt_1ma - eps_r for r<R;
 *      eps_r += d_r; eps_q += mean_r d_r for q>=R  (flow==NULL)  or the flow-warped masked average
 *      of inference.py:374-386 (delta_q supplied by insv2v_flow_correction)]
 *   x0 = (latent - sqrt_1ma*eps)/sqrt_a;  prev = c_x0*x0 + c_eps*eps + c_xt*latent + c_noise*noise
 * eps_in: fp32 [nbranch, F, h, w, 4] channels-last (UNet conv_out output); latent, latent_out,
 * pred_x0, eps_out: fp32 [F,4,h,w] (reference layout).  eps_out/pred_x0/noise may be NULL.
 */
typedef struct insv2v_step_desc {
    const float* eps_in;
    const float* latent;
    const float* latent_ref; /* [R,4,h,w] or NULL */
    const float* delta_q;    /* [F-R,4,h,w] precomputed flow correction for query frames or NULL */
    const float* noise;      /* DDPM variance noise [F,4,h,w] or NULL */
    const float* rescale_stats; /* device [2]: std_text, std_cfg or NULL */
    float* latent_out;
    float* pred_x0;
    float* eps_out;
    int32_t nbranch, F, h, w, R;
    int32_t correct; /* 0 none, 1 mean over refs, 2 use delta_q */
    float text_cfg, img_cfg;
    float sqrt_a, sqrt_1ma;       /* sqrt(alpha_bar_t), sqrt(1-alpha_bar_t) */
    float c_x0, c_eps, c_xt, c_noise;
    float guidance_rescale;
    int64_t branch_stride; /* fp32 elements of eps_in between consecutive branches; 0 = F*h*w*4 */
} insv2v_step_desc;
int insv2v_cfg_step(const insv2v_step_desc* d, insv2v_stream_t stream);
/* std over all elements of n1 (branch 1) and of the CFG-combined eps -> stats[0..1] (inference.py:18-19). */
int insv2v_cfg_stats(const float* eps_in, float* stats, int32_t F, int32_t h, int32_t w, float text_cfg,
                     float img_cfg, int64_t branch_stride, insv2v_stream_t stream);

/* warp_image of misc_utils/flow_utils.py:25-57: bilinear grid_sample(align_corners=True, zero
 * padding) of image [N,C,H,W] fp32 at (x+flow_x, y+flow_y); flow [N,2,H,W] fp32. */
int insv2v_warp_image(const float* image, const float* flow, float* out, int32_t N, int32_t C, int32_t H,
                      int32_t W, insv2v_stream_t stream);
/* resize_flow of misc_utils/flow_utils.py:59-86: scale (u,v) by (W/w, H/h) then bilinear resize
 * (align_corners=False) [N,2,h,w] -> [N,2,H,W]. */
int insv2v_resize_flow(const float* flow, float* out, int32_t N, int32_t h, int32_t w, int32_t H, int32_t W,
                       insv2v_stream_t stream);
/*
 * Flow-warped noise correction for the query frames (inference.py:296-301, :374-386):
 * eps_cfg fp32 [F,4,h,w] is the CFG-combined noise; delta_r = (latent_r - sqrt_a*ref_r)/sqrt_1ma - eps_r;
 * for each query q: out_q = sum_r warp(delta_r, flow_qr) / sum_r warp(1, flow_qr) where the mask
 * sum > 0.5, else 0.  flows: fp32 [F-R, R, 2, h, w] already at latent resolution.
 */
int insv2v_flow_correction(const float* eps_cfg, const float* latent, const float* latent_ref,
                           const float* flows, float* delta_q, int32_t F, int32_t R, int32_t h, int32_t w,
                           float sqrt_a, float sqrt_1ma, insv2v_stream_t stream);

/* Layout conversions at the VAE boundary (instruct_p2p_video.py:57-79):
 * frames fp32 [N,C,H,W] -> fp16 [N,H,W,ldo] (zero channel pad) and back (with optional scale). */
int insv2v_nchw_to_nhwc_f16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ldo,
                            float scale, insv2v_stream_t stream);
int insv2v_nhwc_to_nchw_f32(const void* x, int32_t x_is_fp32, float* y, int32_t N, int32_t C, int32_t H,
                            int32_t W, int32_t ldx, float scale, insv2v_stream_t stream);
/* Diagonal-Gaussian posterior sample of kl_autoencoder/autoencoder.py:10-23, times `scale`:
 * moments fp32 [N,H,W,8] channels-last (mean|logvar) + noise fp32 [N,4,H,W] -> z fp32 [N,4,H,W]. */
int insv2v_posterior_sample(const float* moments, const float* noise, float* z, int32_t N, int32_t H,
                            int32_t W, int32_t ldm, float scale, insv2v_stream_t stream);

/*
 * ---- optical-flow estimator (ABI 9) --------------------------------------------------------------------------------------------------
 * The kernels behind RAFTFlow (misc_utils/flow_utils.py:134-189; built at pl_trainer/inference/inference.py:294, called per query
 * frame at :303-311).  The network itself is third-party (torchvision raft_large, not under the reference tree): every convolution of it
 * is insv2v_im2col + insv2v_gemm (LINEAR; bias and ReLU / sigmoid / tanh in the epilogue), the rest are the entries below.
 * Activations are channels-last fp16 token matrices [n*h*w, C], correspondences and flows fp32 [n, 2, h, w] as in the reference.
 */
typedef struct insv2v_im2col_desc {
    const void* x;   /* fp16 [N*IH*IW, C1 of ldx]: input channels [0, C1) */
    const void* x2;  /* optional second source: channels [C1, C) (row stride ldx2) - torch.cat along channels, never materialised */
    void* out;       /* fp16 [N*OH*OW, ldo]; column (ky * KW + kx) * C + c; zero outside the image */
    int64_t ldx, ldx2, ldo;
    int32_t N, IH, IW, C, C1, KH, KW, stride_h, stride_w, pad_h, pad_w, OH, OW;
} insv2v_im2col_desc;
/* F.conv2d's input gather for kernels the implicit-GEMM path does not cover (7x7 stride 2, 1x5, 5x1, 1x1 stride 2, channel counts
 * that are no multiple of 64): C, C1 multiples of 8.  OH / OW must equal floor((I + 2 pad - K) / stride) + 1. */
int insv2v_im2col(const insv2v_im2col_desc* d, insv2v_stream_t stream);
/* nn.InstanceNorm2d (affine=False, biased variance) + optional ReLU over [N, HW, C] fp16 (torchvision's feature encoder);
 * y != x; partials = fp32 scratch of N * nchunks * C * 2 floats; deterministic (no atomics), shifted sums. */
int insv2v_instance_norm(const void* x, void* y, float* partials, int32_t N, int32_t HW, int32_t C, int64_t ldx, int64_t ldy,
                         int32_t nchunks, float eps, int32_t relu, insv2v_stream_t stream);
#define INSV2V_EW_RELU 1     /* out = relu(a) */
#define INSV2V_EW_ADD_RELU 2 /* out = relu(a + b): ResidualBlock */
#define INSV2V_EW_TANH 3     /* out = tanh(a): initial hidden state */
#define INSV2V_EW_GRU_RH 4   /* out = a * b: ConvGRU r * h (a already sigmoid-ed) */
#define INSV2V_EW_GRU_OUT 5  /* out = (1 - c) * b + c * a: ConvGRU h' = (1 - z) h + z q */
int insv2v_ew(int32_t op, const void* a, const void* b, const void* c, void* out, int64_t rows, int32_t C, int64_t lda, int64_t ldb,
              int64_t ldc, int64_t ldo, insv2v_stream_t stream);
/* F.avg_pool2d(kernel 2, stride 2) over the last two dims of fp32 [n, h, w]: one level of CorrBlock.build_pyramid */
int insv2v_avgpool2x2(const float* x, float* y, int64_t n, int32_t h, int32_t w, insv2v_stream_t stream);
typedef struct insv2v_corr_lookup_desc {
    const float* pyr0;   /* level l: fp32 [B*h*w, h >> l, w >> l] (the all-pairs correlation and its pooled copies) */
    const float* pyr1;
    const float* pyr2;
    const float* pyr3;
    const float* coords; /* fp32 [B, 2, h, w]: current correspondences (x, y) */
    void* out;           /* fp16 [B*h*w, ldo]: levels * (2 radius + 1)^2 look-ups per pixel, remaining columns zero */
    int64_t ldo;
    int32_t B, h, w, levels, radius;
} insv2v_corr_lookup_desc;
/* CorrBlock.index_pyramid: bilinear (align_corners=True, zero padding) samples at coords / 2^l + (d_i, d_j), d in [-radius, radius];
 * channel l * side^2 + i * side + j, the FIRST offset applied to x (torchvision's delta order). */
int insv2v_corr_lookup(const insv2v_corr_lookup_desc* d, insv2v_stream_t stream);
/* coords1 += delta (fp32 rows [B*h*w, ldd], columns 0 / 1; NULL = no update), then rows[pix][0..1] = coords1 - pixel grid as fp16
 * (columns 2 .. ncols-1 zero; rows may be NULL): the flow fed to the motion encoder and appended to its output */
int insv2v_raft_flow_rows(float* coords1, const float* delta, int64_t ldd, void* rows, int64_t ldf, int32_t ncols, int32_t B, int32_t h,
                          int32_t w, insv2v_stream_t stream);
/* upsample_flow: convex combination of the 3x3 neighbourhood of 8 * (coords1 - grid) with softmax weights from the mask predictor's
 * 576 logits per pixel (fp16 rows, k * 64 + i * 8 + j) -> fp32 [B, 2, 8h, 8w] */
int insv2v_convex_upsample(const float* coords1, const void* mask, int64_t ldm, float* out, int32_t B, int32_t h, int32_t w,
                           insv2v_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* INSV2V_HIP_H */
