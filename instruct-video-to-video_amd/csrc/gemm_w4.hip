// gemm_w4: persistent fp16 MFMA GEMM / implicit-GEMM 3x3 convolution for the UNet's many short-K, memory-heavy
// problems.  4 waves per workgroup, TWO workgroups per CU.
//
// Why: with one 8-wave workgroup per CU (gemm_p8.hip) a tile's epilogue - ~130 KB of stores, the residual reads and the
// fp32 arithmetic - runs while the CU's matrix pipes idle, and at K = 320 ... 1280 the epilogue is as long as the K loop
// (tools/gemm_check ablations, profiles/r02_gemm_p8_ablation.txt).  Here every wave keeps p8's 128 x 64 register tile
// (same LDS read traffic per FLOP) but a workgroup is only 4 waves = one wave per SIMD, and two independent workgroups
// share a CU: while one is in its epilogue / waiting for operands the other one's MFMAs own the matrix pipes.
//
//  * tile = (128 WM) x (64 WN) with WM x WN = 4 waves: 128 x 256 (1 x 4) for wide N, 256 x 128 (2 x 2) for N = 320/640.
//  * K tile = 32 (64-byte LDS rows, 16-byte chunk index XOR-swizzled by (row>>2)&3 on the DMA source side and on the
//    fragment read), 3-slot LDS-DMA ring (72 KiB): per K tile ONE s_barrier: counted s_waitcnt vmcnt (the next K tile's
//    pieces stay in flight) -> barrier -> request K tile s+2 into the slot K tile s-1 just vacated -> 12 ds_read_b128
//    -> 16 MFMA 32x32x16.
//  * persistent: grid = 2 x CUs; the K-tile stream continues across output tiles, so a tile's first loads are in
//    flight while the previous tile's epilogue runs (same stream cursor as gemm_p8.hip).
//  * epilogue: gemm_p8's LDS-free, branch-free form (permlane32_swap pairs -> 16-byte stores, out-of-range buffer
//    offsets instead of branches, park vectors read with inline-asm ds_read), see the comments there.
#include "common.h"
#include "gemm_dma.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int BK4 = 32;                 // halfs per K tile
constexpr int PARK4_B = 4096;
#define SB() __builtin_amdgcn_sched_barrier(0)

typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
// Two v_permlane32_swap (a[32..63] <-> b[0..31]) behind explicit wait states.  hipcc pads the documented 2 wait states
// between a VALU write and the swap that reads it, but with a second wave resident on the SIMD that was not enough on
// gfx950: lanes 12-15 of every 16 still saw the operand's OLD contents (e.g. the unconverted fp32 feeding v_cvt_pk_f16_f32,
// profiles/r02_gemm_debug.md).  The "+v" ties put every producer before the statement, s_nop 7 gives 8 wait states.
__device__ __forceinline__ void swap32x2(unsigned& a0, unsigned& b0, unsigned& a1, unsigned& b1) {
    asm volatile("s_nop 7\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 3"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
}
// fp32 pair -> packed fp16 (round to nearest even) with the classic two-convert + pack sequence: the single
// v_cvt_pk_f16_f32 hipcc picks on gfx950 left lanes 12-15 of every 16 unconverted when a second wave shared the SIMD
// (profiles/r02_gemm_debug.md)
// two fp32 -> one packed fp16 pair, round to nearest even: gfx950's v_cvt_pk_f16_f32 (ONE instruction; rounds 1-5 spent two v_cvt_f16_f32 and a
// v_pack_b32_f16 here, because the only packed conversion of earlier parts, v_cvt_pkrtz, rounds toward zero)
__device__ __forceinline__ unsigned pack_h2(float x, float y) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2v));
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[0]; }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[1]; }

struct W4Args : insv2v_gemm_desc { int tile_delay; };  // -1 = automatic, 0 = none, n > 0 = n delay units

template <int MODE, bool GEGLU, bool HAS_RES, int WM, int WN, int TM, int DBG = 0>
__global__ __launch_bounds__(WM * WN * 64, WM * WN / 2) void gemm_w4_kernel(W4Args p) {
    // WM x WN waves, each owning TM x 2 fragments of 32 x 32: (32 TM) tokens x 64 channels.  4 waves x (128 x 64) keep 256
    // VGPRs per wave (2 waves per SIMD with two workgroups per CU); 8 waves x (64 x 64) fit 128 VGPRs -> 4 waves per
    // SIMD, which is what lets one workgroup's epilogue overlap the other's K loop.
    constexpr int NW = WM * WN;
    static_assert(NW == 4 || NW == 8, "four or eight waves per workgroup");
    constexpr int BM = 32 * TM * WM, BN = 64 * WN;
    constexpr int SLOT_B = (BM + BN) * 64;          // one K tile: A rows then W rows, 64 bytes each
    constexpr int RING_B = 3 * SLOT_B;
    constexpr int NPA = BM / (16 * NW), NPW = BN / (16 * NW);  // LDS-DMA pieces (16 rows x 64 B) per wave per K tile
    static_assert(NPA * 16 * NW == BM && NPW * 16 * NW == BN, "tile rows must divide over the waves");
    constexpr int NP = NPA + NPW;
    static_assert(RING_B + 2 * PARK4_B <= 81920, "two workgroups per CU");
    // park layout: bias | col_sum | row bias (BN floats each) | (mean, rstd) x BM
    constexpr int PK_CS = BN * 4, PK_RB = 2 * BN * 4, PK_ST = 3 * BN * 4;
    static_assert(PK_ST + BM * 8 <= PARK4_B, "park area");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    // ---- tile rasterisation: XCD-aware remap, then groups of GROUP_M tile rows walked column by column ----
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    auto tile_origin = [&](int v, int& bm0, int& bn0) {
        const int bid = xcd_remap(v, ntiles);
        constexpr int GROUP_M = 8;
        const int per_group = GROUP_M * tiles_n;
        const int gidx = bid / per_group, first_m = gidx * GROUP_M;
        const int gsz = min(GROUP_M, tiles_m - first_m), rin = bid - gidx * per_group;
        const int tn = rin / gsz, tm = first_m + rin - tn * gsz;
        bm0 = tm * BM; bn0 = tn * BN;
    };

    const srd_t rA = make_srd(p.a), rA2 = make_srd(p.a2 ? p.a2 : p.a), rW = make_srd(p.w);
    const bool ln = p.row_stats != nullptr;

    // ---- staging side.  A piece is 16 rows x 64 B: row = piece*16 + lane/4, LDS chunk slot lane%4, source chunk =
    // slot ^ ((row>>2)&3).  Wave `wid` fills A pieces wid + NW i (i < NPA) and W pieces wid + NW i (i < NPW).
    const int prow = wid * 16 + (lane >> 2);                        // row of piece `wid`; piece wid + NW i is 16 NW i rows further
    const int chunk8 = ((lane & 3) ^ ((prow >> 2) & 3)) * 8;       // halfs (row + 16 NW i has the same key)
    int arow[NPA], aoh[NPA], aow[NPA];
    unsigned woff[NPW];
    const int nk = p.K / BK4;  // >= 2 (host check): a tile's park vectors are retired by the wait of its second K tile
    const int IHu = p.upsample ? p.IH * 2 : p.IH, IWu = p.upsample ? p.IW * 2 : p.IW;
    const int ups = p.upsample ? 1 : 0;
    struct Cursor { int v, kt, k0, kh, kw, ci0; } cur = {(int)blockIdx.x, 0, 0, 0, 0, 0};  // wave-uniform scalars
    auto set_stage_tile = [&](int v) {
        int bm0, bn0;
        tile_origin(v, bm0, bn0);
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = bm0 + i * 16 * NW + prow;
            if (MODE == INSV2V_MODE_LINEAR) {
                arow[i] = m < p.M ? m : -1;
                aoh[i] = aow[i] = 0;
            } else {
                const int mm = m < p.M ? m : 0;
                const int ow = mm % p.OW, t = mm / p.OW;
                const int oh = t % p.OH, nb = t / p.OH;
                arow[i] = m < p.M ? nb * p.IH * p.IW : -1;
                aoh[i] = oh * p.stride - p.pad_t;
                aow[i] = ow * p.stride - p.pad_l;
            }
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int n = bn0 + i * 16 * NW + prow;
            woff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + chunk8) * 2) : OOB_OFFSET;
        }
    };
    auto advance = [&]() {  // next K tile of the stream
        if (++cur.kt == nk) {
            cur.v += gridDim.x; cur.kt = 0; cur.k0 = 0; cur.kh = cur.kw = cur.ci0 = 0;
            if (cur.v < ntiles) set_stage_tile(cur.v);
            return;
        }
        cur.k0 += BK4;
        if (MODE != INSV2V_MODE_LINEAR) {
            cur.ci0 += BK4;
            if (cur.ci0 >= p.Cin) {
                cur.ci0 = 0;
                if (++cur.kw == 3) { cur.kw = 0; ++cur.kh; }
            }
        }
    };
    auto stage = [&](int slot) {  // the cursor's K tile -> ring slot
        char* dst = smem + slot * SLOT_B + wid * 1024;
        if (MODE == INSV2V_MODE_LINEAR) {
            const bool second = p.k_split > 0 && cur.k0 >= p.k_split;
            const int ld = (int)(second ? p.lda2 : p.lda);
            const int soff = (second ? cur.k0 - p.k_split : cur.k0) * 2;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int m = arow[i];
                dma16(second ? rA2 : rA, m >= 0 ? (unsigned)((m * ld + chunk8) * 2) : OOB_OFFSET, soff, dst + i * NW * 1024);
            }
        } else {
            const bool second = p.k_split > 0 && cur.ci0 >= p.k_split;
            const int ld = (int)(second ? p.lda2 : p.lda);
            const int soff = (second ? cur.ci0 - p.k_split : cur.ci0) * 2;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int ih = aoh[i] + cur.kh, iw = aow[i] + cur.kw;
                const bool ok = arow[i] >= 0 && (unsigned)ih < (unsigned)IHu && (unsigned)iw < (unsigned)IWu;
                const int pix = arow[i] + (ih >> ups) * p.IW + (iw >> ups);
                dma16(second ? rA2 : rA, ok ? (unsigned)((pix * ld + chunk8) * 2) : OOB_OFFSET, soff, dst + i * NW * 1024);
            }
        }
        const int wsoff = cur.k0 * 2;
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma16(rW, woff[i], wsoff, dst + BM * 64 + i * NW * 1024);
    };
    auto row_group = [&](int m) { int g = m / p.rows_per_group; if (p.rb_mod > 0) g %= p.rb_mod; return g; };
    // park vectors of one tile: one (partial) LDS-DMA piece per wave; lanes beyond the vector are masked off
    auto stage_park = [&](int pb, int bm0, int bn0) {
        char* dst = smem + RING_B + pb * PARK4_B;
        if (wid == 0) {
            const int n = bn0 + lane * 4;
            if (lane < BN / 4) dma16(make_srd(p.bias ? (const void*)p.bias : p.w), (p.bias && n < p.N) ? (unsigned)(n * 4) : OOB_OFFSET, 0, dst);
        } else if (wid == 1) {
            const int n = bn0 + lane * 4;
            if (lane < BN / 4) dma16(make_srd(ln ? (const void*)p.col_sum : p.w), (ln && n < p.N) ? (unsigned)(n * 4) : OOB_OFFSET, 0, dst + PK_CS);
        } else if (wid == 2) {
            const int n = bn0 + lane * 4;
            const int g = p.row_bias ? row_group(bm0) : 0;  // every row of the tile is in this group (checked on the host)
            if (lane < BN / 4)
                dma16(make_srd(p.row_bias ? (const void*)p.row_bias : p.w), (p.row_bias && n < p.N) ? (unsigned)((g * (int)p.ld_rb + n) * 4) : OOB_OFFSET, 0, dst + PK_RB);
        } else if (wid == 3) {
#pragma unroll
            for (int i = 0; i < (BM + 127) / 128; ++i) {
                const int m = bm0 + i * 128 + lane * 2;
                if (i * 128 + lane * 2 < BM)
                    dma16(make_srd(ln ? (const void*)p.row_stats : p.w), (ln && m < p.M) ? (unsigned)(m * 8) : OOB_OFFSET, 0, dst + PK_ST + i * 1024);
            }
        }
    };

    // ---- fragment addressing (bytes): row = ... + (lane & 31), 16-byte chunk (kk*2 + lane/32) ^ ((row>>2)&3)
    const int frow = lane & 31, fhi = lane >> 5, fsw = (frow >> 2) & 3;
    int coff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) coff[kk] = ((kk * 2 + fhi) ^ fsw) * 16;
    const char* aBase = smem + (wm * 32 * TM + frow) * 64;
    const char* wBase = smem + BM * 64 + (wn * 64 + frow) * 64;

    floatx16 acc[2][TM];  // [i: 32-channel block][j: 32-token block]
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto compute = [&](int slot) {
        half8 fa[TM][2], fw[2][2];
        const char* a = aBase + slot * SLOT_B;
        const char* w = wBase + slot * SLOT_B;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fw[i][kk] = *(const half8*)(w + i * 32 * 64 + coff[kk]);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[j][kk] = *(const half8*)(a + j * 32 * 64 + coff[kk]);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i][kk], fa[j][kk], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue (see gemm_p8.hip for the why of its form) ----
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const srd_t rC = make_srd(p.c), rR = make_srd(p.residual ? p.residual : p.c);
    auto park6 = [&](unsigned a, floatx4& b0, floatx4& b1, floatx4& r0, floatx4& r1, floatx4& c0, floatx4& c1) {
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:32\n\tds_read_b128 %2, %6 offset:%7\n\t"
                     "ds_read_b128 %3, %6 offset:%8\n\tds_read_b128 %4, %6 offset:%9\n\tds_read_b128 %5, %6 offset:%10\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(b0), "=&v"(b1), "=&v"(r0), "=&v"(r1), "=&v"(c0), "=&v"(c1)
                     : "v"(a), "n"(PK_RB), "n"(PK_RB + 32), "n"(PK_CS), "n"(PK_CS + 32) : "memory");
    };
    auto stat4 = [&](unsigned a, float2& s0, float2& s1, float2& s2, float2& s3) {
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:256\n\tds_read_b64 %2, %4 offset:512\n\t"
                     "ds_read_b64 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a) : "memory");
    };
    auto epilogue = [&](int bm0, int bn0, int pb) {
        if (DBG == 2) {  // timing ablation: no epilogue; one dummy store keeps the accumulators live
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += acc[i][j][r];
            if (s == 12345.678f) ((half_t*)p.c)[tid] = (half_t)s;
            return;
        }
        const unsigned park = lds0 + RING_B + pb * PARK4_B;
        constexpr int NIQ = GEGLU ? 1 : 2;
        const int oN = GEGLU ? (p.N >> 1) : p.N;
        const int nl0 = wn * 64;                                           // tile-local first channel of this wave
        const int on0 = GEGLU ? ((bn0 + nl0) >> 1) : bn0 + nl0;            // first output column of this wave
        const int m0 = bm0 + wm * 32 * TM + frow;
        // v = rstd * (alpha * acc - mean * col_sum) + bias  ==  fma(ra, acc, fma(rm, col_sum, bias))
        float ra[TM], rm[TM];
        unsigned offc[TM], offr[TM];
        {
            float2 st[4];  // TM = 2 reads two rows past its block: inside the park area, unused
            stat4(park + PK_ST + (wm * 32 * TM + frow) * 8, st[0], st[1], st[2], st[3]);
#pragma unroll
            for (int rbk = 0; rbk < TM; ++rbk) {
                const int m = m0 + rbk * 32;
                const float mean = ln ? st[rbk].x : 0.f, rstd = ln ? st[rbk].y : 1.f;
                ra[rbk] = rstd * p.alpha; rm[rbk] = -rstd * mean;
                offc[rbk] = m < p.M ? (unsigned)(m * (int)p.ldc * 2 + fhi * 16) : OOB_OFFSET;
                offr[rbk] = m < p.M ? (unsigned)(m * (int)p.ldr * 2 + fhi * 16) : OOB_OFFSET;
            }
        }
        // Residual: the 16-byte pieces of the next channel group are requested before the current group is computed and
        // stored; hipcc's own counted vmcnt waits retire them (loads and stores of a wave retire in issue order).
        auto load_res = [&](int g, uint4v (&rv)[TM]) {
            const int on = on0 + (g >> 1) * 32 + (g & 1) * 16;
            const bool okc = on + fhi * 8 + 8 <= oN;
#pragma unroll
            for (int rbk = 0; rbk < TM; ++rbk)
                rv[rbk] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[rbk] : OOB_OFFSET, on * 2, 0);
        };
        uint4v rvc[TM], rvn[TM];
        if (HAS_RES) load_res(0, rvc);
#pragma unroll
        for (int g = 0; g < NIQ * 2; ++g) {
            const int iq = g >> 1, qp = g & 1;
            if (HAS_RES && g + 1 < NIQ * 2) load_res(g + 1, rvn);
            // bias (+ the tile's row-bias vector), column sums of the 2 x 4 channels this lane owns in quarters 2qp, 2qp+1
            float bs[2][4], cs[2][4], gbs[2][4], gcs[2][4];
            {
                floatx4 tb[2], tr[2], tc[2], gb[2], gr[2], gc[2];
                const unsigned a = park + (nl0 + iq * 32 + 16 * qp + 4 * fhi) * 4;  // quarter q = 2qp; q + 1 is 32 bytes on
                park6(a, tb[0], tb[1], tr[0], tr[1], tc[0], tc[1]);
                if (GEGLU) park6(a + 128, gb[0], gb[1], gr[0], gr[1], gc[0], gc[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bs[h][e] = tb[h][e] + tr[h][e]; cs[h][e] = tc[h][e];
                        if (GEGLU) { gbs[h][e] = gb[h][e] + gr[h][e]; gcs[h][e] = gc[h][e]; }
                    }
            }
            const int on = on0 + iq * 32 + qp * 16;
            const bool okc = on + fhi * 8 + 8 <= oN;
#pragma unroll
            for (int rbk = 0; rbk < TM; ++rbk) {
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * qp + h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(ra[rbk], acc[iq][rbk][4 * q + e], fmaf(rm[rbk], cs[h][e], bs[h][e]));
                        if (GEGLU) x *= gelu_erf_f(fmaf(ra[rbk], acc[1][rbk][4 * q + e], fmaf(rm[rbk], gcs[h][e], gbs[h][e])));
                        v[h][e] = x;
                    }
                }
                if (HAS_RES) {  // un-swap the residual piece into the fragment layout, add in fp32
                    unsigned r0 = rvc[rbk][0], r1 = rvc[rbk][1], r2 = rvc[rbk][2], r3 = rvc[rbk][3];
                    swap32x2(r0, r2, r1, r3);
                    v[0][0] += h_lo(r0); v[0][1] += h_hi(r0); v[0][2] += h_lo(r1); v[0][3] += h_hi(r1);
                    v[1][0] += h_lo(r2); v[1][1] += h_hi(r2); v[1][2] += h_lo(r3); v[1][3] += h_hi(r3);
                }
                unsigned a0 = pack_h2(v[0][0], v[0][1]), a1 = pack_h2(v[0][2], v[0][3]);
                unsigned b0 = pack_h2(v[1][0], v[1][1]), b1 = pack_h2(v[1][2], v[1][3]);
                swap32x2(a0, b0, a1, b1);
                const uint4v out = {a0, a1, b0, b1};
                __builtin_amdgcn_raw_buffer_store_b128(out, rC, okc ? offc[rbk] : OOB_OFFSET, on * 2, 0);
                // Keep the store's data registers untouched for a few cycles: with a second wave on the SIMD the 16-byte
                // store was still reading lanes 12-15 of every 16 when the next VALU instruction reused the register
                // (those lanes stored the NEXT value - profiles/r02_gemm_debug.md).  The "v" inputs pin the registers.
                asm volatile("s_nop 7" ::"v"(out));
            }
            if (HAS_RES && g + 1 < NIQ * 2) {
#pragma unroll
                for (int rbk = 0; rbk < TM; ++rbk) rvc[rbk] = rvn[rbk];
            }
        }
    };

    // ---- de-phase the two workgroups of a CU: tiles of one launch all take the same time, so two workgroups that
    // start together would run their K loops together and their epilogues together; the one that got the second wave
    // slot of its SIMD starts half a K loop later (delay in units of ~450 cycles, default = number of K tiles)
    if (p.tile_delay != 0) {
        const unsigned hw_id = __builtin_amdgcn_s_getreg((3 << 11) | 4);  // HW_REG_HW_ID[3:0] = wave slot on the SIMD
        if (hw_id & 1) {
            const int n = p.tile_delay > 0 ? p.tile_delay : nk;
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(7);
        }
    }

    // ---- prologue: the first two K tiles of the stream ----
    int cbm0, cbn0;           // tile being computed
    int cv = blockIdx.x, cpb = 0;
    if (cv >= ntiles) return;
    tile_origin(cv, cbm0, cbn0);
    set_stage_tile(cv);
    stage_park(0, cbm0, cbn0);
    stage(0);
    advance();
    if (cur.v < ntiles) { stage(1); advance(); }
    zero_acc();

    int slot = 0, nslot = 2;  // ring slot of the K tile being computed / of the K tile requested next
    for (; cv < ntiles; cv += gridDim.x) {
        tile_origin(cv, cbm0, cbn0);
        for (int t = 0; t < nk; ++t) {
            // K tile s has landed once at most the NP pieces of K tile s+1 (if it was requested) are outstanding
            const bool next_requested = (t + 1 < nk) || (cv + (int)gridDim.x < ntiles);
            if (next_requested) wait_vmcnt<NP>(); else wait_vmcnt<0>();
            SB();
            __builtin_amdgcn_s_barrier();   // everyone's pieces of K tile s are in LDS; everyone is done with K tile s-1
            SB();
            if (t == 0 && cv != (int)blockIdx.x) stage_park(cpb, cbm0, cbn0);
            if (cur.v < ntiles) { stage(nslot); advance(); }
            compute(slot);
            slot = slot == 2 ? 0 : slot + 1;
            nslot = nslot == 2 ? 0 : nslot + 1;
        }
        epilogue(cbm0, cbn0, cpb);
        zero_acc();
        cpb ^= 1;
    }
}

template <int MODE, bool GEGLU, bool HAS_RES, int WM, int WN, int TM, int DBG = 0>
int launch_w4(const insv2v_gemm_desc& d, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 64 * WN;
    constexpr int LDS_B = 3 * (BM + BN) * 64 + 2 * PARK4_B;
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_w4_kernel<MODE, GEGLU, HAS_RES, WM, WN, TM, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    static const int dbg_wgs = getenv("INSV2V_W4_WGS") ? atoi(getenv("INSV2V_W4_WGS")) : 2;  // debugging: workgroups per CU
    const int grid = tiles < dbg_wgs * num_cu ? tiles : dbg_wgs * num_cu;
    static const int dbg_delay = getenv("INSV2V_W4_DELAY") ? atoi(getenv("INSV2V_W4_DELAY")) : -1;
    W4Args a;
    static_cast<insv2v_gemm_desc&>(a) = d;
    a.tile_delay = grid > num_cu ? dbg_delay : 0;
    hipLaunchKernelGGL((gemm_w4_kernel<MODE, GEGLU, HAS_RES, WM, WN, TM, DBG>), dim3(grid), dim3(WM * WN * 64), LDS_B, s, a);
    return launch_status();
}

template <int WM, int WN, int TM>
int dispatch_w4(const insv2v_gemm_desc& d, int dbg, hipStream_t s) {
    const bool conv = d.mode == INSV2V_MODE_CONV3X3, gg = d.act == INSV2V_ACT_GEGLU, res = d.residual != nullptr;
    if (dbg == 1) return conv ? INSV2V_EUNSUPPORTED : launch_w4<INSV2V_MODE_LINEAR, false, false, WM, WN, TM, 2>(d, s);  // no epilogue
    if (conv) return res ? launch_w4<INSV2V_MODE_CONV3X3, false, true, WM, WN, TM>(d, s) : launch_w4<INSV2V_MODE_CONV3X3, false, false, WM, WN, TM>(d, s);
    if (gg) return launch_w4<INSV2V_MODE_LINEAR, true, false, WM, WN, TM>(d, s);
    return res ? launch_w4<INSV2V_MODE_LINEAR, false, true, WM, WN, TM>(d, s) : launch_w4<INSV2V_MODE_LINEAR, false, false, WM, WN, TM>(d, s);
}

}  // namespace

// variant: 0 = 128 x 256 tile (waves 1 x 4), 1 = 256 x 128 tile (waves 2 x 2); +10 = timing ablation without epilogue
int insv2v_gemm_w4(const insv2v_gemm_desc& d, int variant, hipStream_t s) {
    if (d.batch > 1 || d.c_fp32 || d.split_k > 1) return INSV2V_EUNSUPPORTED;
    if ((d.K % BK4) || d.K < 2 * BK4 || (d.N & 7) || (d.ldc & 7) || ((uintptr_t)d.c & 15)) return INSV2V_EUNSUPPORTED;
    if (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15))) return INSV2V_EUNSUPPORTED;
    if (d.k_split && (d.k_split % BK4)) return INSV2V_EUNSUPPORTED;
    const bool gg = d.act == INSV2V_ACT_GEGLU;
    if (gg && ((d.N % 64) || d.residual)) return INSV2V_EUNSUPPORTED;
    if (!gg && d.act != INSV2V_ACT_NONE) return INSV2V_EUNSUPPORTED;  // SiLU / quick-GELU GEMMs are tiny (time embedding, CLIP)
    if (d.row_stats && (d.M & 1)) return INSV2V_EUNSUPPORTED;          // (mean, rstd) pairs are fetched two rows per lane
    const int v = variant % 10;  // 0: 4 waves 128x256, 1: 4 waves 256x128, 2: 8 waves 128x256, 3: 8 waves 256x128
    const int bm = (v == 1 || v == 3) ? 256 : 128;
    if (v >= 2 && d.act == INSV2V_ACT_GEGLU) return INSV2V_EUNSUPPORTED;  // the 8-wave tiles exist for the memory-bound N = C GEMMs
    // the row-bias vector is parked per tile: every tile must lie inside one group
    if (d.row_bias && ((d.ld_rb & 3) || (d.rows_per_group % bm && d.M > d.rows_per_group))) return INSV2V_EUNSUPPORTED;
    if ((int64_t)d.M * d.ldc * 2 >= ((int64_t)1 << 31) || (d.residual && (int64_t)d.M * d.ldr * 2 >= ((int64_t)1 << 31))) return INSV2V_EUNSUPPORTED;
    const bool conv = d.mode == INSV2V_MODE_CONV3X3;
    if (conv && ((d.Cin % BK4) || gg)) return INSV2V_EUNSUPPORTED;
    const int dbg = variant / 10;
    switch (v) {
        case 0: return dispatch_w4<1, 4, 4>(d, dbg, s);
        case 1: return dispatch_w4<2, 2, 4>(d, dbg, s);
        case 2: return dispatch_w4<2, 4, 2>(d, dbg, s);
        case 3: return dispatch_w4<4, 2, 2>(d, dbg, s);
    }
    return INSV2V_EINVAL;
}
