// insv2v_gemm: fp16 MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogue.
//
// Roofline: MFMA-bound (dense contraction).  One workgroup = 4 waves (2x2) computes a
// BM x BN output tile with v_mfma_f32_32x32x16_f16; the MFMA "A" operand is the weight
// fragment (rows = output channels n) and the "B" operand the activation fragment
// (cols = tokens m), so each lane ends up with 4 CONSECUTIVE output channels of one token
// and stores them as one 8-byte write into the channels-last output.
// K is consumed in 64-wide slices staged through LDS (144-byte padded rows: conflict-free
// ds_read_b128 for the 32-row fragment pattern), double buffered, global loads for slice
// t+1 issued before the MFMAs of slice t and written to LDS after them.
// In CONV3X3 mode the activation rows are gathered on the fly (tap-shifted pixels, zero
// fill at the border, optional nearest-x2 upsample and channel concat), so im2col, the
// upsampled tensor and torch.cat are never materialised in HBM.
#include "common.h"

#define BK 64
#define LDS_LD 72  // halfs per LDS row (64 + 8 pad) = 144 B

struct RowInfo {  // per-thread metadata of one staged activation row
    int64_t base;  // linear: m*lda ; conv: nb*IH*IW (pixel index base)
    int oh, ow;    // conv only (already multiplied by stride, minus pad)
    bool valid;
};

template <int MI, int NI, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(insv2v_gemm_desc p) {
    constexpr int BM = MI * 64, BN = NI * 64;
    constexpr int RA = BM / 32, RW = BN / 32;  // 16-byte chunks per thread per slice
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sA = (half_t*)smem;                   // [2][BM][LDS_LD]
    half_t* sW = sA + 2 * BM * LDS_LD;            // [2][BN][LDS_LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int bm0 = tm * BM, bn0 = tn * BN;
    const int z = blockIdx.y;

    const half_t* A = (const half_t*)p.a + z * p.a_bs;
    const half_t* A2 = p.a2 ? (const half_t*)p.a2 + z * p.a_bs : nullptr;
    const half_t* Wp = (const half_t*)p.w + z * p.w_bs;

    const int crow = tid >> 3, cchunk = tid & 7;  // staging: row crow+32*i, 16B chunk cchunk

    RowInfo ri[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = bm0 + crow + 32 * i;
        ri[i].valid = m < p.M;
        if (MODE == INSV2V_MODE_LINEAR) {
            ri[i].base = (int64_t)m;
            ri[i].oh = ri[i].ow = 0;
        } else {
            int mm = ri[i].valid ? m : 0;
            int ow = mm % p.OW, t = mm / p.OW;
            int oh = t % p.OH, nb = t / p.OH;
            ri[i].base = (int64_t)nb * p.IH * p.IW;
            ri[i].oh = oh * p.stride - p.pad_t;
            ri[i].ow = ow * p.stride - p.pad_l;
        }
    }
    bool wvalid[RW];
    int64_t wbase[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        int n = bn0 + crow + 32 * i;
        wvalid[i] = n < p.N;
        wbase[i] = (int64_t)n * p.ldw;
    }

    uint4 ra[RA], rw[RW];
    const int nk = (p.K + BK - 1) / BK;
    const int IHu = p.upsample ? p.IH * 2 : p.IH, IWu = p.upsample ? p.IW * 2 : p.IW;

    auto load_slice = [&](int kt) {
        const int k0 = kt * BK;
        const int kc = k0 + cchunk * 8;
        if (MODE == INSV2V_MODE_LINEAR) {
            const bool kval = kc < p.K;
            const bool second = p.k_split > 0 && k0 >= p.k_split;
            const half_t* src = second ? A2 : A;
            const int64_t ld = second ? p.lda2 : p.lda;
            const int koff = second ? kc - p.k_split : kc;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ri[i].valid && kval) v = *(const uint4*)(src + ri[i].base * ld + koff);
                ra[i] = v;
            }
        } else {
            const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
            const int kh = tap / 3, kw = tap - kh * 3;
            const bool second = p.k_split > 0 && ci0 >= p.k_split;
            const half_t* src = second ? A2 : A;
            const int64_t ld = second ? p.lda2 : p.lda;
            const int coff = (second ? ci0 - p.k_split : ci0) + cchunk * 8;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                int ih = ri[i].oh + kh, iw = ri[i].ow + kw;
                bool ok = ri[i].valid && ih >= 0 && ih < IHu && iw >= 0 && iw < IWu;
                if (p.upsample) { ih >>= 1; iw >>= 1; }
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *(const uint4*)(src + (ri[i].base + (int64_t)ih * p.IW + iw) * ld + coff);
                ra[i] = v;
            }
        }
        const bool kvalw = kc < p.K;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (wvalid[i] && kvalw) v = *(const uint4*)(Wp + wbase[i] + kc);
            rw[i] = v;
        }
    };
    auto store_slice = [&](int buf) {
        half_t* a = sA + buf * BM * LDS_LD;
        half_t* w = sW + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < RA; ++i) *(uint4*)(a + (crow + 32 * i) * LDS_LD + cchunk * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < RW; ++i) *(uint4*)(w + (crow + 32 * i) * LDS_LD + cchunk * 8) = rw[i];
    };

    floatx16 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_slice(0);
    store_slice(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_slice(kt + 1);
        const half_t* a = sA + cur * BM * LDS_LD + (wm * MI * 32 + frow) * LDS_LD + fk;
        const half_t* w = sW + cur * BN * LDS_LD + (wn * NI * 32 + frow) * LDS_LD + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            half8 fa[MI], fw[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) fa[j] = *(const half8*)(a + j * 32 * LDS_LD + kk * 16);
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = *(const half8*)(w + i * 32 * LDS_LD + kk * 16);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_slice(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds, for token m, channels n0+8q+4*(lane>>5)+{0..3}, q=0..3 -------------
    const bool geglu = p.act == INSV2V_ACT_GEGLU;
    char* Cb = (char*)p.c + (int64_t)z * p.c_bs * (p.c_fp32 ? 4 : 2);
    const half_t* Rp = p.residual ? (const half_t*)p.residual + z * p.r_bs : nullptr;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = bm0 + wm * MI * 32 + j * 32 + (lane & 31);
        if (m >= p.M) continue;
        const float* rb = p.row_bias ? p.row_bias + (int64_t)(m / p.rows_per_group) * p.ld_rb : nullptr;
#pragma unroll
        for (int i = 0; i < (NI); ++i) {
            if (geglu && (i & 1)) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * NI * 32 + i * 32 + 8 * q + 4 * (lane >> 5);  // tile-local n of v[0]
                const int n = bn0 + nl;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * q + e] * p.alpha;
                    if (n + e < p.N) {
                        if (p.bias) x += p.bias[n + e];
                        if (rb) x += rb[n + e];
                    }
                    v[e] = x;
                }
                int on = n, oN = p.N;
                if (geglu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float g = acc[(i + 1) % NI][j][4 * q + e] * p.alpha;
                        if (n + 32 + e < p.N && p.bias) g += p.bias[n + 32 + e];
                        v[e] = v[e] * gelu_erf_f(g);
                    }
                    on = (n >> 6) * 32 + (n & 31);
                    oN = p.N >> 1;
                } else if (p.act == INSV2V_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                if (on >= oN) continue;
                if (Rp) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (on + e < oN) v[e] += (float)Rp[(int64_t)m * p.ldr + on + e];
                }
                if (p.c_fp32) {
                    float* dst = (float*)Cb + (int64_t)m * p.ldc + on;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (on + e < oN) dst[e] = v[e];
                } else {
                    half_t* dst = (half_t*)Cb + (int64_t)m * p.ldc + on;
                    if (on + 3 < oN && ((p.ldc & 3) == 0)) {
                        half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *(half4*)dst = h;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (on + e < oN) dst[e] = (half_t)v[e];
                    }
                }
            }
        }
    }
}

template <int MI, int NI, int MODE>
static int launch_cfg(const insv2v_gemm_desc& d, hipStream_t s) {
    constexpr int BM = MI * 64, BN = NI * 64;
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<MI, NI, MODE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    dim3 grid(tiles, d.batch > 0 ? d.batch : 1);
    hipLaunchKernelGGL((gemm_kernel<MI, NI, MODE>), grid, dim3(256), lds, s, d);
    return launch_status();
}

template <int MODE>
static int dispatch_tile(const insv2v_gemm_desc& d, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch_cfg<2, 2, MODE>(d, s);
        case 2: return launch_cfg<1, 2, MODE>(d, s);
        case 3: return launch_cfg<2, 1, MODE>(d, s);
        case 4: return launch_cfg<1, 1, MODE>(d, s);
    }
    return INSV2V_EINVAL;
}

static int pick_tile(const insv2v_gemm_desc& d) {
    // Largest tile that still yields >= ~2 workgroups per CU; GEGLU needs a 128-wide N tile
    // (each wave must own an [h|g] pair of 32-row weight blocks).
    const long batch = d.batch > 0 ? d.batch : 1;
    auto blocks = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * batch; };
    const bool geglu = d.act == INSV2V_ACT_GEGLU;
    if (blocks(128, 128) >= 512) return 1;
    if (geglu) return blocks(128, 128) >= 256 ? 1 : 2;
    if (d.N <= 64) return blocks(128, 64) >= 512 ? 3 : 4;
    if (blocks(64, 128) >= 512) return 2;
    if (blocks(128, 64) >= 512) return 3;
    return 4;
}

extern "C" int insv2v_gemm(const insv2v_gemm_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    insv2v_gemm_desc d = *dp;
    if (!d.a || !d.w || !d.c || d.M <= 0 || d.N <= 0 || d.K <= 0) return INSV2V_EINVAL;
    if ((d.K & 7) || (d.lda & 7) || (d.ldw & 7)) return INSV2V_EINVAL;
    if (((uintptr_t)d.a | (uintptr_t)d.w) & 15) return INSV2V_EINVAL;
    if (d.k_split) {
        if (!d.a2 || (d.k_split % BK) || (d.lda2 & 7) || ((uintptr_t)d.a2 & 15)) return INSV2V_EINVAL;
    }
    if (d.row_bias && d.rows_per_group <= 0) return INSV2V_EINVAL;
    if (d.act == INSV2V_ACT_GEGLU && (d.N % 64)) return INSV2V_EINVAL;
    if (d.batch <= 0) d.batch = 1;
    if (d.alpha == 0.f) d.alpha = 1.f;
    if (d.mode == INSV2V_MODE_CONV3X3) {
        if (d.Cin <= 0 || (d.Cin % BK) || d.K != 9 * d.Cin) return INSV2V_EINVAL;
        if ((long)d.NB * d.OH * d.OW != d.M || d.stride < 1) return INSV2V_EINVAL;
    } else if (d.mode != INSV2V_MODE_LINEAR) {
        return INSV2V_EUNSUPPORTED;
    }
    int tile = d.tile ? d.tile : pick_tile(d);
    if (d.act == INSV2V_ACT_GEGLU && (tile == 3 || tile == 4)) tile = 2;
    hipStream_t s = as_stream(stream);
    return d.mode == INSV2V_MODE_CONV3X3 ? dispatch_tile<INSV2V_MODE_CONV3X3>(d, tile, s)
                                         : dispatch_tile<INSV2V_MODE_LINEAR>(d, tile, s);
}
