// Winograd F(2x2, 3x3) form of the stride-1, pad-1 3x3 convolutions with wide inputs (ResnetBlock3D conv1 / conv2 at UNet levels 1-3,
// resnet.py:143,159 behind InflatedConv3d, resnet.py:10-18).  Round 6.
//
// Why: the GEMM engine runs at the board's power cap (profiles/r06_engine_ceiling.txt) - what is left is fewer FLOPs.  Y = A^T [ (G g G^T) .
// (B^T d B) ] A computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per (channel pair) instead of 36: 2.25 x fewer MACs.
// Per convolution three launches:
//   1. insv2v_winograd_input : GroupNorm scale/shift + SiLU (the apply pass that exists anyway: resnet.py:177-178,188) and V = B^T d B,
//      written as 16 matrices V_k [tiles, Cin] (4 x the tensor; every pixel normalised once, staged in LDS per image and 64-channel slice)
//   2. insv2v_gemm with grouped weights (w_group_rows): M_k = V_k U_k^T, k = 0..15, as ONE launch of the 256-row ping-pong engine
//      (U_k = (G g G^T)_k [Cout, Cin], transformed on the host at load time)
//   3. insv2v_winograd_output: Y = A^T M A + bias + time-embedding row bias + residual -> [pixels, Cout]
// The transform passes are HBM-bound (5 + 6 units of the tensor against the 2.25 x shorter GEMM): it pays where Cin >= 1280 (K >= 11 520):
// 0.60 - 0.78 of the direct implicit-GEMM convolution at the B = 60 stack, not at Cin = 640 (0.96) or level 0 (1.17) -
// profiles/r06_winograd_proto.txt.  fp16 storage of V, U and M costs 6.5e-4 of max|ref| against fp32 F.conv2d (direct fp16 output: 3.4e-4;
// stated single-kernel tolerance 2e-3).
#include "common.h"

namespace {

// LDS slot of staged pixel sp (128 B each): the two pixels of every odd pair are swapped, so that the pixels p and p + 2 two neighbouring tiles
// read lie in different halves of a 256-byte bank row (PMC: 36 % of the input transform's LDS cycles were 2-way conflicts without it)
template <bool UP> __device__ __forceinline__ int lds_px(int sp) { return UP ? sp : sp ^ ((sp >> 1) & 1); }   // (UP reads neighbouring pixels: no conflicts; its W may be odd)

// ---- input transform.  One workgroup = IPB consecutive images x one 64-channel slice; LDS holds those images' (normalised) pixels
// [IPB][H*W][64] fp16.  Thread = (work item tid / 8, 8-channel chunk tid % 8).
// UP (Upsample3D's nearest x2 + 3x3 convolution, resnet.py:48-69, in one Winograd pass): the 4x4 patch of output tile (y, x) on the upsampled grid is
// the low-resolution 3x3 neighbourhood with its centre row / column doubled, so B^T d B has a zero row and column (index 2): 9 of the 16 matrices
// remain - one tile per LOW-resolution pixel, groups g = ci * 3 + cj over (i, j) in {0, 1, 3}^2: 4 x fewer MACs than the direct form.
template <bool NORM, bool UP>
__global__ __launch_bounds__(256) void wino_input_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, const float* __restrict__ ab,
                                                         half_t* __restrict__ V, int64_t ldx, int64_t ldx2, int C1, int C, int NB, int H, int W,
                                                         int ipb, int images_per_sample, int silu, int64_t group_rows, int band_tr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, chunk = tid & 7, item0 = tid >> 3;
    const int nb0 = (int)blockIdx.x * ipb, c0 = (int)blockIdx.y * 64;
    // band blockIdx.z of the image: tile rows [tr0, tr1); staged pixel rows [row0, row0 + nrows) (row0 may be -1: rows outside the image are
    // never staged and read as zero).  One band (band_tr = all tile rows): the whole image, ipb images per workgroup.
    const int trows = UP ? H : H >> 1;
    const int tr0 = (int)blockIdx.z * band_tr, tr1 = min(tr0 + band_tr, trows);
    const bool whole = band_tr >= trows;
    const int row0 = whole ? 0 : (UP ? tr0 - 1 : 2 * tr0 - 1);
    const int nrows = whole ? H : (UP ? (tr1 - tr0) + 2 : 2 * (tr1 - tr0) + 2);
    const int HW = H * W, SW = nrows * W;   // pixels of an image / staged pixels per image
    const bool second = x2 != nullptr && c0 >= C1;
    const half_t* src = second ? x2 + (c0 - C1) + chunk * 8 : x + c0 + chunk * 8;
    const int64_t ld = second ? ldx2 : ldx;
    const int nimg = min(ipb, NB - nb0);
    // phase 1: every pixel of the slice normalised once -> LDS
    for (int it = item0; it < nimg * SW; it += 32) {
        const int img = it / SW, sp = it - img * SW;
        const int iy = row0 + sp / W;
        if ((unsigned)iy >= (unsigned)H) continue;
        half8 v = *(const half8*)(src + ((int64_t)(nb0 + img) * HW + sp + row0 * W) * ld);
        if (NORM) {
            const float* p = ab + ((int64_t)((nb0 + img) / images_per_sample) * C + c0 + chunk * 8) * 2;
            const float4 p0 = *(const float4*)p, p1 = *(const float4*)(p + 4), p2 = *(const float4*)(p + 8), p3 = *(const float4*)(p + 12);
            const float sc[8] = {p0.x, p0.z, p1.x, p1.z, p2.x, p2.z, p3.x, p3.z}, sh[8] = {p0.y, p0.w, p1.y, p1.w, p2.y, p2.w, p3.y, p3.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = fmaf((float)v[e], sc[e], sh[e]);
                if (silu) f = silu_f(f);
                v[e] = (half_t)f;
            }
        }
        *(half8*)(smem + ((int64_t)img * SW + lds_px<UP>(sp)) * 128 + chunk * 16) = v;   // staged pixel (img, sp)
    }
    __syncthreads();
    // phase 2: one 4x4 patch -> 16 (UP: 9) transformed values per channel; zero padding outside the image (applied AFTER the norm, like the conv's)
    if constexpr (UP) {
        const int bt = (tr1 - tr0) * W;   // tiles of this band per image
        for (int it = item0; it < nimg * bt; it += 32) {
            const int img = it / bt, tb_ = it - img * bt;
            const int y = tr0 + tb_ / W, xx = tb_ - (tb_ / W) * W;
            const int t = y * W + xx;
            const char* base = smem + (int64_t)img * SW * 128 + chunk * 16;
            float hz[3][3][8];   // [low row y-1, y, y+1][(x-1) - x, 2 x, x - (x+1)]
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = y - 1 + r;
                float d[3][8];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ix = xx - 1 + c;
                    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ok) v = *(const half8*)(base + lds_px<UP>((iy - row0) * W + ix) * 128);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[c][e] = (float)v[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    hz[r][0][e] = d[0][e] - d[1][e];
                    hz[r][1][e] = 2.f * d[1][e];
                    hz[r][2][e] = d[1][e] - d[2][e];
                }
            }
            const int64_t trow = (int64_t)(nb0 + img) * HW + t;
            half_t* dst = V + trow * C + c0 + chunk * 8;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                half8 o[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[0][e] = (half_t)(hz[0][j][e] - hz[1][j][e]);
                    o[1][e] = (half_t)(2.f * hz[1][j][e]);
                    o[2][e] = (half_t)(hz[1][j][e] - hz[2][j][e]);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) *(half8*)(dst + (int64_t)(i * 3 + j) * group_rows * C) = o[i];
            }
        }
        return;
    }
    const int th = H >> 1, tw = W >> 1, ntile = th * tw;
    const int bt = (tr1 - tr0) * tw;   // tiles of this band per image
    for (int it = item0; it < nimg * bt; it += 32) {
        const int img = it / bt, tb_ = it - img * bt;
        const int ty = tr0 + tb_ / tw, tx = tb_ - (tb_ / tw) * tw;
        const int t = ty * tw + tx;
        const char* base = smem + (int64_t)img * SW * 128 + chunk * 16;
        float hz[4][4][8];   // horizontal transform of the four patch rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r;
            float d[4][8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = 2 * tx - 1 + c;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *(const half8*)(base + lds_px<UP>((iy - row0) * W + ix) * 128);
#pragma unroll
                for (int e = 0; e < 8; ++e) d[c][e] = (float)v[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hz[r][0][e] = d[0][e] - d[2][e];
                hz[r][1][e] = d[1][e] + d[2][e];
                hz[r][2][e] = d[2][e] - d[1][e];
                hz[r][3][e] = d[1][e] - d[3][e];
            }
        }
        const int64_t trow = (int64_t)(nb0 + img) * ntile + t;
        half_t* dst = V + trow * C + c0 + chunk * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half8 o[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[0][e] = (half_t)(hz[0][j][e] - hz[2][j][e]);
                o[1][e] = (half_t)(hz[1][j][e] + hz[2][j][e]);
                o[2][e] = (half_t)(hz[2][j][e] - hz[1][j][e]);
                o[3][e] = (half_t)(hz[1][j][e] - hz[3][j][e]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *(half8*)(dst + (int64_t)(i * 4 + j) * group_rows * C) = o[i];
        }
    }
}

// ---- output transform.  Thread = (tile, 8 output channels): 16 loads, 2x2 pixels out.
template <bool RES, bool RB, bool UP>
__global__ __launch_bounds__(256) void wino_output_kernel(const half_t* __restrict__ Mo, const float* __restrict__ bias, const float* __restrict__ row_bias,
                                                          const half_t* __restrict__ res, half_t* __restrict__ y, int64_t group_rows, int64_t ntiles_total,
                                                          int Cout, int H, int W, int64_t ld_rb, int rows_per_group, int64_t ldr, int64_t ldy) {
    const int c8 = Cout >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tile = idx / c8;
    if (tile >= ntiles_total) return;
    const int ch = (int)(idx - tile * c8) * 8;
    // UP: one tile per low-resolution pixel (ty, tx) = (y, x) of the H x W input, 2x2 output pixels of the 2H x 2W image; 9 matrices, the
    // patch row / column 2 of M is zero
    const int th = UP ? H : H >> 1, tw = UP ? W : W >> 1, ntile = th * tw;
    const int OW = UP ? 2 * W : W, OH = UP ? 2 * H : H;
    const int nb = (int)(tile / ntile), t = (int)(tile - (int64_t)nb * ntile);
    const int ty = t / tw, tx = t - ty * tw;
    const half_t* src = Mo + tile * Cout + ch;
    float s[2][4][8];   // vertical transform: rows (m0 + m1 + m2), (m1 - m2 - m3) of every patch column
    if constexpr (UP) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            half8 m[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) m[i] = *(const half8*)(src + (int64_t)(i * 3 + j) * group_rows * Cout);
            const int jj = j == 2 ? 3 : j;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[0][jj][e] = (float)m[0][e] + (float)m[1][e];
                s[1][jj][e] = (float)m[1][e] - (float)m[2][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s[0][2][e] = s[1][2][e] = 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half8 m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = *(const half8*)(src + (int64_t)(i * 4 + j) * group_rows * Cout);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[0][j][e] = (float)m[0][e] + (float)m[1][e] + (float)m[2][e];
                s[1][j][e] = (float)m[1][e] - (float)m[2][e] - (float)m[3][e];
            }
        }
    }
    float b[8];
    {
        const float4 b0 = bias ? *(const float4*)(bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f), b1 = bias ? *(const float4*)(bias + ch + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
    const int64_t m00 = ((int64_t)nb * OH + 2 * ty) * OW + 2 * tx;
    if (RB) {   // the 2x2 pixels of a tile lie in one image, an image in one row-bias group
        const float* rb = row_bias + (m00 / rows_per_group) * ld_rb + ch;
        const float4 r0 = *(const float4*)rb, r1 = *(const float4*)(rb + 4);
        b[0] += r0.x; b[1] += r0.y; b[2] += r0.z; b[3] += r0.w; b[4] += r1.x; b[5] += r1.y; b[6] += r1.z; b[7] += r1.w;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int64_t m = m00 + (int64_t)i * OW + jj;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = (jj == 0 ? s[i][0][e] + s[i][1][e] + s[i][2][e] : s[i][1][e] - s[i][2][e] - s[i][3][e]) + b[e];
            if (RES) {
                const half8 r = *(const half8*)(res + m * ldr + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
            *(half8*)(y + m * ldy + ch) = o;
        }
}

}  // namespace

extern "C" int insv2v_winograd_input(const insv2v_winograd_in_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    const insv2v_winograd_in_desc& d = *dp;
    if (!d.x || !d.v || d.NB <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0) return INSV2V_EINVAL;
    if ((!d.upsample && ((d.H & 1) || (d.W & 1))) || (d.C % 64) || (d.x2 && (d.C1 <= 0 || d.C1 >= d.C || (d.C1 % 64)))) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.v & 15) || (d.x2 && ((d.ldx2 & 7) || ((uintptr_t)d.x2 & 15)))) return INSV2V_EINVAL;
    if (d.gn_ab && (d.gn_images_per_sample <= 0 || ((uintptr_t)d.gn_ab & 15))) return INSV2V_EINVAL;
    const int HW = d.H * d.W, ntile = d.upsample ? HW : HW / 4;
    const int64_t tiles = (int64_t)d.NB * ntile;
    if (d.v_group_rows < tiles) return INSV2V_EINVAL;
    // One image's 64-channel slice is staged in at most 64 KiB of LDS: whole images (several per workgroup when they are small) or, for
    // larger ones, bands of tile rows with a one-pixel halo above and below (grid z)
    const int trows = d.upsample ? d.H : d.H / 2;
    int band_tr = trows, ipb = 1;
    if (HW * 128 <= 64 * 1024) {
        ipb = 32 / ntile;                                   // at least one work item per 8-thread group
        if (ipb < 1) ipb = 1;
        while (ipb > 1 && ipb * HW * 128 > 64 * 1024) --ipb;
    } else {
        const int max_rows = 64 * 1024 / (d.W * 128);       // staged pixel rows that fit
        band_tr = d.upsample ? max_rows - 2 : (max_rows - 2) / 2;
        if (band_tr < 1) return INSV2V_EUNSUPPORTED;
    }
    const int nbands = (trows + band_tr - 1) / band_tr;
    const int srows = nbands == 1 ? d.H : (d.upsample ? band_tr + 2 : 2 * band_tr + 2);
    const dim3 grid((unsigned)((d.NB + ipb - 1) / ipb), (unsigned)(d.C / 64), (unsigned)nbands);
    const size_t lds = (size_t)ipb * srows * d.W * 128;
#define WINO_IN(NORM, UP)                                                                                                                             \
    hipLaunchKernelGGL((wino_input_kernel<NORM, UP>), grid, dim3(256), lds, as_stream(stream), (const half_t*)d.x, (const half_t*)d.x2, d.gn_ab, (half_t*)d.v, \
                       d.ldx, d.ldx2, d.x2 ? d.C1 : d.C, d.C, d.NB, d.H, d.W, ipb, d.gn_ab ? d.gn_images_per_sample : 1, d.gn_ab ? d.gn_silu : 0, d.v_group_rows, band_tr)
    if (d.gn_ab) { if (d.upsample) WINO_IN(true, true); else WINO_IN(true, false); }
    else { if (d.upsample) WINO_IN(false, true); else WINO_IN(false, false); }
#undef WINO_IN
    return launch_status();
}

extern "C" int insv2v_winograd_output(const insv2v_winograd_out_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    const insv2v_winograd_out_desc& d = *dp;
    if (!d.m || !d.y || d.NB <= 0 || d.H <= 0 || d.W <= 0 || d.Cout <= 0) return INSV2V_EINVAL;
    if ((!d.upsample && ((d.H & 1) || (d.W & 1))) || (d.Cout & 7)) return INSV2V_EUNSUPPORTED;
    if ((d.ldy & 7) || ((uintptr_t)d.m & 15) || ((uintptr_t)d.y & 15) || (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15)))) return INSV2V_EINVAL;
    if ((d.bias && ((uintptr_t)d.bias & 15)) || (d.row_bias && (((uintptr_t)d.row_bias & 15) || (d.ld_rb & 3) || d.rows_per_group <= 0 || d.rows_per_group % (d.H * d.W * (d.upsample ? 4 : 1)))))
        return INSV2V_EINVAL;
    const int64_t tiles = d.upsample ? (int64_t)d.NB * d.H * d.W : (int64_t)d.NB * (d.H / 2) * (d.W / 2);
    if (d.m_group_rows < tiles) return INSV2V_EINVAL;
    const int64_t n = tiles * (d.Cout / 8);
    const dim3 grid((unsigned)((n + 255) / 256));
#define WINO_OUT(RES, RB, UP)                                                                                                                        \
    hipLaunchKernelGGL((wino_output_kernel<RES, RB, UP>), grid, dim3(256), 0, as_stream(stream), (const half_t*)d.m, d.bias, d.row_bias, (const half_t*)d.residual, \
                       (half_t*)d.y, d.m_group_rows, tiles, d.Cout, d.H, d.W, d.ld_rb, d.rows_per_group, d.ldr, d.ldy)
    if (d.upsample) {
        if (d.residual) { if (d.row_bias) WINO_OUT(true, true, true); else WINO_OUT(true, false, true); }
        else { if (d.row_bias) WINO_OUT(false, true, true); else WINO_OUT(false, false, true); }
    } else {
        if (d.residual) { if (d.row_bias) WINO_OUT(true, true, false); else WINO_OUT(true, false, false); }
        else { if (d.row_bias) WINO_OUT(false, true, false); else WINO_OUT(false, false, false); }
    }
#undef WINO_OUT
    return launch_status();
}
