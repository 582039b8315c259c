// Optical-flow estimator (RAFT-large) support kernels: everything of torchvision's raft_large (reference: RAFTFlow,
// misc_utils/flow_utils.py:134-189, called at pl_trainer/inference/inference.py:294,303-311) that is not a GEMM.
//
// The network's convolutions (7x7 stride 2, 3x3 stride 1/2, 1x1 stride 2, 1x5, 5x1, on 2 ... 384 channels) run as
// insv2v_im2col + insv2v_gemm (LINEAR, bias + ReLU / sigmoid / tanh in the epilogue); activations are channels-last fp16 token
// matrices [n*h*w, C] like everywhere else in the library, correspondences / flows fp32 [n, 2, h, w] like the reference's.
// All of it runs ONCE per window - HBM-bound gathers, no MFMA: coalesced 16-byte accesses, one launch per op.
#include "common.h"

// ---- im2col: out[(n, oh, ow)][(ky * KW + kx) * C + c] = x[n][oh * sh + ky - ph][ow * sw + kx - pw][c], zeros outside -------------------
// x = channels [0, C1) of row stride ldx, x2 (optional) = channels [C1, C) of row stride ldx2: the ConvGRU's [r * h | x] concat.
__global__ __launch_bounds__(256) void im2col_kernel(const half_t* x, const half_t* x2, half_t* out, int N, int IH, int IW, int C, int C1,
                                                     int64_t ldx, int64_t ldx2, int KH, int KW, int sh, int sw, int ph, int pw, int OH, int OW,
                                                     int64_t ldo) {
    const int cchunks = C >> 3, per_row = KH * KW * cchunks;
    const int64_t total = (int64_t)N * OH * OW * per_row;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / per_row;
        const int r = (int)(idx - row * per_row), tap = r / cchunks, c0 = (r - tap * cchunks) * 8;
        const int ky = tap / KW, kx = tap - ky * KW;
        const int ow = (int)(row % OW), t = (int)(row / OW), oh = t % OH, n = t / OH;
        const int ih = oh * sh + ky - ph, iw = ow * sw + kx - pw;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((unsigned)ih < (unsigned)IH && (unsigned)iw < (unsigned)IW) {
            const int64_t pix = ((int64_t)n * IH + ih) * IW + iw;
            v = c0 < C1 ? *(const uint4*)(x + pix * ldx + c0) : *(const uint4*)(x2 + pix * ldx2 + (c0 - C1));
        }
        *(uint4*)(out + row * ldo + tap * C + c0) = v;
    }
}

extern "C" int insv2v_im2col(const insv2v_im2col_desc* dp, insv2v_stream_t stream) {
    if (!dp || !dp->x || !dp->out) return INSV2V_EINVAL;
    const insv2v_im2col_desc d = *dp;
    const int C1 = d.x2 ? d.C1 : d.C;
    if (d.N <= 0 || d.C <= 0 || (d.C & 7) || (C1 & 7) || C1 <= 0 || C1 > d.C || (d.ldx & 7) || (d.ldo & 7) || d.KH <= 0 || d.KW <= 0 ||
        d.stride_h <= 0 || d.stride_w <= 0 || d.ldo < (int64_t)d.KH * d.KW * d.C || (d.x2 && (d.ldx2 & 7)))
        return INSV2V_EINVAL;
    if ((((uintptr_t)d.x | (uintptr_t)d.out) & 15) || (d.x2 && ((uintptr_t)d.x2 & 15))) return INSV2V_EINVAL;
    const int OH = (d.IH + 2 * d.pad_h - d.KH) / d.stride_h + 1, OW = (d.IW + 2 * d.pad_w - d.KW) / d.stride_w + 1;
    if (OH != d.OH || OW != d.OW) return INSV2V_EINVAL;
    const int64_t total = (int64_t)d.N * OH * OW * d.KH * d.KW * (d.C >> 3);
    const int blocks = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const half_t*)d.x, (const half_t*)d.x2, (half_t*)d.out,
                       d.N, d.IH, d.IW, d.C, C1, d.ldx, d.ldx2, d.KH, d.KW, d.stride_h, d.stride_w, d.pad_h, d.pad_w, OH, OW, d.ldo);
    return launch_status();
}

// ---- InstanceNorm2d (affine = False, eps 1e-5, biased variance) + optional ReLU, in place or out of place -----------------------------
// pass 1: per (image, row chunk): shifted per-channel sums (shift = the image's first row: no E[x^2] - E[x]^2 cancellation), deterministic
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const half_t* x, float* part, int HW, int C, int64_t ld, int nchunks, int rows_per_chunk) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int cc = C >> 3, lanes = 256 / cc > 0 ? 256 / cc : 1;   // threads = cc * lanes (<= 256)
    const int tid = threadIdx.x, c8 = tid % cc, pl = tid / cc;
    __shared__ float sm[2][256][8];
    float s[8] = {}, q[8] = {};
    if (pl < lanes) {
        const half_t* base = x + (int64_t)n * HW * ld + c8 * 8;
        const half8 kv = *(const half8*)base;
        const int r1 = min((chunk + 1) * rows_per_chunk, HW);
        for (int r = chunk * rows_per_chunk + pl; r < r1; r += lanes) {
            const half8 v = *(const half8*)(base + (int64_t)r * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = (float)v[e] - (float)kv[e]; s[e] += dlt; q[e] += dlt * dlt; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[0][tid][e] = s[e]; sm[1][tid][e] = q[e]; }
    __syncthreads();
    if (pl == 0 && tid < cc) {
        for (int j = 1; j < lanes; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += sm[0][j * cc + c8][e]; q[e] += sm[1][j * cc + c8][e]; }
        float* o = part + (((int64_t)n * nchunks + chunk) * C + c8 * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[2 * e] = s[e]; o[2 * e + 1] = q[e]; }
    }
}
// pass 2: every thread folds the chunk sums of its 8 channels (same order everywhere: deterministic), then normalises its rows
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const half_t* x, half_t* y, const float* part, int HW, int C, int64_t ld, int64_t ldy,
                                                            int nchunks, int rows_per_block, float eps, int relu) {
    const int n = blockIdx.y;
    const int cc = C >> 3, lanes = 256 / cc > 0 ? 256 / cc : 1;
    const int tid = threadIdx.x, c8 = tid % cc, pl = tid / cc;
    if (pl >= lanes) return;
    const half_t* base = x + (int64_t)n * HW * ld + c8 * 8;
    const half8 kv = *(const half8*)base;
    float mean[8], rstd[8];
    {
        float s[8] = {}, q[8] = {};
        for (int ch = 0; ch < nchunks; ++ch) {
            const float* o = part + (((int64_t)n * nchunks + ch) * C + c8 * 8) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += o[2 * e]; q[e] += o[2 * e + 1]; }
        }
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float m = s[e] * inv;
            mean[e] = (float)kv[e] + m;
            rstd[e] = rsqrtf(fmaxf(q[e] * inv - m * m, 0.f) + eps);
        }
    }
    const int r1 = min((blockIdx.x + 1) * rows_per_block, HW);
    for (int r = blockIdx.x * rows_per_block + pl; r < r1; r += lanes) {
        const half8 v = *(const half8*)(base + (int64_t)r * ld);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = ((float)v[e] - mean[e]) * rstd[e];
            if (relu) t = fmaxf(t, 0.f);
            o[e] = (half_t)t;
        }
        *(half8*)(y + ((int64_t)n * HW + r) * ldy + c8 * 8) = o;
    }
}

extern "C" int insv2v_instance_norm(const void* x, void* y, float* partials, int32_t N, int32_t HW, int32_t C, int64_t ldx, int64_t ldy,
                                    int32_t nchunks, float eps, int32_t relu, insv2v_stream_t stream) {
    if (!x || !y || !partials || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || C > 2048 || (ldx & 7) || (ldy & 7) || nchunks <= 0 || nchunks > HW)
        return INSV2V_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) || x == y) return INSV2V_EINVAL;   // (in place, a block could overwrite the shift row another still reads)
    const int rpc = (HW + nchunks - 1) / nchunks;
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(nchunks, N), dim3(256), 0, as_stream(stream), (const half_t*)x, partials, HW, C, ldx, nchunks, rpc);
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3(nchunks, N), dim3(256), 0, as_stream(stream), (const half_t*)x, (half_t*)y, (const float*)partials,
                       HW, C, ldx, ldy, nchunks, rpc, eps, relu);
    return launch_status();
}

// ---- element-wise ops on fp16 token matrices ([rows, C], 8 channels per thread) ------------------------------------------------------------
//  RELU      out = relu(a)                      ADD_RELU  out = relu(a + b)                 (ResidualBlock: relu(downsample(x) + y))
//  TANH      out = tanh(a)                      (context encoder: hidden state = tanh, context = relu of the two channel halves)
//  GRU_RH    out = b * a       (a = r, already sigmoid-ed by the GEMM; b = h)               (ConvGRU: r * h)
//  GRU_OUT   out = (1 - c) * b + c * a   (a = q = tanh(convq), b = h, c = z)                (ConvGRU: h' = (1 - z) h + z q)
__global__ __launch_bounds__(256) void ew_kernel(int op, const half_t* a, const half_t* b, const half_t* c, half_t* out, int64_t rows, int C,
                                                 int64_t lda, int64_t ldb, int64_t ldc, int64_t ldo) {
    const int cc = C >> 3;
    const int64_t total = rows * cc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cc;
        const int c0 = (int)(idx - r * cc) * 8;
        const half8 va = *(const half8*)(a + r * lda + c0);
        half8 vb = {}, vc = {}, o;
        if (b) vb = *(const half8*)(b + r * ldb + c0);
        if (c) vc = *(const half8*)(c + r * ldc + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float fa = (float)va[e], fb = (float)vb[e], fc = (float)vc[e];
            float v;
            switch (op) {
                case INSV2V_EW_RELU: v = fmaxf(fa, 0.f); break;
                case INSV2V_EW_ADD_RELU: v = fmaxf(fa + fb, 0.f); break;
                case INSV2V_EW_TANH: v = act_raft_f(fa, INSV2V_ACT_TANH); break;
                case INSV2V_EW_GRU_RH: v = fa * fb; break;
                default: v = (1.0f - fc) * fb + fc * fa; break;
            }
            o[e] = (half_t)v;
        }
        *(half8*)(out + r * ldo + c0) = o;
    }
}

extern "C" int insv2v_ew(int32_t op, const void* a, const void* b, const void* c, void* out, int64_t rows, int32_t C, int64_t lda, int64_t ldb,
                         int64_t ldc, int64_t ldo, insv2v_stream_t stream) {
    if (!a || !out || rows <= 0 || C <= 0 || (C & 7) || (lda & 7) || (ldo & 7) || op < INSV2V_EW_RELU || op > INSV2V_EW_GRU_OUT) return INSV2V_EINVAL;
    if ((op == INSV2V_EW_ADD_RELU || op == INSV2V_EW_GRU_RH || op == INSV2V_EW_GRU_OUT) && (!b || (ldb & 7))) return INSV2V_EINVAL;
    if (op == INSV2V_EW_GRU_OUT && (!c || (ldc & 7))) return INSV2V_EINVAL;
    if (((uintptr_t)a | (uintptr_t)out | (uintptr_t)b | (uintptr_t)c) & 15) return INSV2V_EINVAL;
    const int64_t total = rows * (C >> 3);
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(ew_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), op, (const half_t*)a, (const half_t*)b, (const half_t*)c,
                       (half_t*)out, rows, C, lda, ldb, ldc, ldo);
    return launch_status();
}

// ---- correlation pyramid: 2x2 average pooling over the LAST two dims of [n, h, w] fp32 (F.avg_pool2d(kernel 2, stride 2): floor) --------
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* x, float* y, int64_t n, int h, int w) {
    const int oh = h >> 1, ow = w >> 1;
    const int64_t total = n * oh * ow;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % ow), i = (int)((idx / ow) % oh);
        const int64_t m = idx / ((int64_t)oh * ow);
        const float* p = x + (m * h + 2 * i) * w + 2 * j;
        y[idx] = 0.25f * (p[0] + p[1] + p[w] + p[w + 1]);
    }
}
extern "C" int insv2v_avgpool2x2(const float* x, float* y, int64_t n, int32_t h, int32_t w, insv2v_stream_t stream) {
    if (!x || !y || n <= 0 || h < 2 || w < 2) return INSV2V_EINVAL;
    const int64_t total = n * (h >> 1) * (w >> 1);
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, y, n, h, w);
    return launch_status();
}

// ---- correlation look-up (CorrBlock.index_pyramid): per pixel and pyramid level (2r + 1)^2 bilinear samples around centroid / 2^level ---
// pyr[l] = [B * h * w, h >> l, w >> l] fp32; coords = [B, 2, h, w] fp32 (channel 0 = x, 1 = y).  Output fp16 rows [B*h*w, ldo]: channel
// l * side^2 + i * side + j samples (x + d_i, y + d_j) - torchvision's delta order, the first offset goes to x - with zero padding
// (F.grid_sample, align_corners=True: the normalisation 2 x / (w - 1) - 1 and its inverse cancel for w > 1); columns [L * side^2, ldo) = 0.
__global__ __launch_bounds__(256) void corr_lookup_kernel(insv2v_corr_lookup_desc p) {
    const int side = 2 * p.radius + 1, per = side * side, nch = p.levels * per;
    const int64_t npix = (int64_t)p.B * p.h * p.w;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < npix * p.ldo; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = idx / p.ldo;
        const int ch = (int)(idx - pix * p.ldo);
        float v = 0.f;
        if (ch < nch) {
            const int l = ch / per, k = ch - l * per, i = k / side, j = k - i * side;
            const int b = (int)(pix / ((int64_t)p.h * p.w)), hw = (int)(pix - (int64_t)b * p.h * p.w);
            const float scale = 1.0f / (float)(1 << l);
            const float cx = p.coords[((int64_t)b * 2 + 0) * p.h * p.w + hw] * scale + (float)(i - p.radius);
            const float cy = p.coords[((int64_t)b * 2 + 1) * p.h * p.w + hw] * scale + (float)(j - p.radius);
            const int hl = p.h >> l, wl = p.w >> l;
            const float* img = (l == 0 ? p.pyr0 : l == 1 ? p.pyr1 : l == 2 ? p.pyr2 : p.pyr3) + pix * hl * wl;
            const float fx = floorf(cx), fy = floorf(cy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float ax = cx - fx, ay = cy - fy;
            auto at = [&](int yy, int xx) { return ((unsigned)yy < (unsigned)hl && (unsigned)xx < (unsigned)wl) ? img[yy * wl + xx] : 0.f; };
            v = (1.f - ay) * ((1.f - ax) * at(y0, x0) + ax * at(y0, x0 + 1)) + ay * ((1.f - ax) * at(y0 + 1, x0) + ax * at(y0 + 1, x0 + 1));
        }
        ((half_t*)p.out)[idx] = (half_t)v;
    }
}
extern "C" int insv2v_corr_lookup(const insv2v_corr_lookup_desc* dp, insv2v_stream_t stream) {
    if (!dp || !dp->coords || !dp->out || dp->levels < 1 || dp->levels > 4 || dp->radius < 0 || dp->B <= 0 || dp->h <= 0 || dp->w <= 0) return INSV2V_EINVAL;
    const int side = 2 * dp->radius + 1;
    if (dp->ldo < dp->levels * side * side) return INSV2V_EINVAL;
    const float* pyr[4] = {dp->pyr0, dp->pyr1, dp->pyr2, dp->pyr3};
    for (int l = 0; l < dp->levels; ++l)
        if (!pyr[l] || (dp->h >> l) < 2 || (dp->w >> l) < 2) return INSV2V_EINVAL;   // (w = 1 divides by zero in the reference's normalisation)
    const int64_t total = (int64_t)dp->B * dp->h * dp->w * dp->ldo;
    const int blocks = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
    hipLaunchKernelGGL(corr_lookup_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), *dp);
    return launch_status();
}

// ---- correspondence update + flow rows: coords1 += delta (fp32 rows [B*h*w, ldd], columns 0 / 1 = dx / dy; nullptr = no update), -----
// flow_rows[pix][0 / 1] = coords1 - coords0 (the pixel grid) as fp16, remaining columns of the ldf-wide row zero
__global__ __launch_bounds__(256) void flow_rows_kernel(float* coords1, const float* delta, int64_t ldd, half_t* rows, int64_t ldf, int ncols, int B, int h, int w) {
    const int64_t npix = (int64_t)B * h * w;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(pix / ((int64_t)h * w)), hw = (int)(pix - (int64_t)b * h * w), y = hw / w, x = hw - y * w;
        float* cx = coords1 + ((int64_t)b * 2 + 0) * h * w + hw;
        float* cy = coords1 + ((int64_t)b * 2 + 1) * h * w + hw;
        float vx = *cx, vy = *cy;
        if (delta) { vx += delta[pix * ldd]; vy += delta[pix * ldd + 1]; *cx = vx; *cy = vy; }
        if (rows) {
            half_t* o = rows + pix * ldf;
            o[0] = (half_t)(vx - (float)x); o[1] = (half_t)(vy - (float)y);
            for (int e = 2; e < ncols; ++e) o[e] = (half_t)0.f;
        }
    }
}
extern "C" int insv2v_raft_flow_rows(float* coords1, const float* delta, int64_t ldd, void* rows, int64_t ldf, int32_t ncols, int32_t B, int32_t h,
                                     int32_t w, insv2v_stream_t stream) {
    if (!coords1 || B <= 0 || h <= 0 || w <= 0 || (rows && (ncols < 2 || ldf < ncols)) || (delta && ldd < 2)) return INSV2V_EINVAL;
    const int64_t npix = (int64_t)B * h * w;
    hipLaunchKernelGGL(flow_rows_kernel, dim3((int)((npix + 255) / 256)), dim3(256), 0, as_stream(stream), coords1, delta, ldd, (half_t*)rows, ldf, ncols, B, h, w);
    return launch_status();
}

// ---- convex upsampling (upsample_flow): out[b][c][8 y + i][8 x + j] = sum_k softmax_k(mask[b, y, x][k * 64 + i * 8 + j]) * 8 * flow9_k -----
// flow = coords1 - pixel grid at 1/8 resolution; flow9_k = the 3x3 neighbourhood (F.unfold, padding 1: zeros outside), k = ky * 3 + kx.
// mask rows fp16 [B*h*w, ldm] (576 logits, already multiplied by 0.25 in the producing GEMM).
__global__ __launch_bounds__(64) void convex_upsample_kernel(const float* coords1, const half_t* mask, int64_t ldm, float* out, int B, int h, int w) {
    const int64_t pix = blockIdx.x;
    const int b = (int)(pix / ((int64_t)h * w)), hw = (int)(pix - (int64_t)b * h * w), y = hw / w, x = hw - y * w;
    const int sub = threadIdx.x, i = sub >> 3, j = sub & 7;
    const half_t* m = mask + pix * ldm;
    float lg[9], mx = -1e30f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { lg[k] = (float)m[k * 64 + sub]; mx = fmaxf(mx, lg[k]); }
    float den = 0.f, fx = 0.f, fy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float e = __expf(lg[k] - mx);
        den += e;
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
            fx += e * 8.0f * (coords1[((int64_t)b * 2 + 0) * h * w + yy * w + xx] - (float)xx);
            fy += e * 8.0f * (coords1[((int64_t)b * 2 + 1) * h * w + yy * w + xx] - (float)yy);
        }
    }
    const int64_t H = 8 * h, W = 8 * w;
    out[(((int64_t)b * 2 + 0) * H + 8 * y + i) * W + 8 * x + j] = fx / den;
    out[(((int64_t)b * 2 + 1) * H + 8 * y + i) * W + 8 * x + j] = fy / den;
}
extern "C" int insv2v_convex_upsample(const float* coords1, const void* mask, int64_t ldm, float* out, int32_t B, int32_t h, int32_t w,
                                      insv2v_stream_t stream) {
    if (!coords1 || !mask || !out || B <= 0 || h <= 0 || w <= 0 || ldm < 576) return INSV2V_EINVAL;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)((int64_t)B * h * w)), dim3(64), 0, as_stream(stream), coords1, (const half_t*)mask, ldm, out, B, h, w);
    return launch_status();
}
