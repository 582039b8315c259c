// gemm_as: "A-stationary" fp16 MFMA GEMM for the UNet's short-K linears (K = 320 at level 0, 640 at level 1).
//
// Why a fourth GEMM form: at K = 320 a classic output-stationary tile spends as long in its prologue + epilogue as in
// its K loop (5 K tiles of 64), and neither the 256x256 8-phase kernel (gemm_p8.hip, one workgroup per CU) nor the
// 2 x 4-wave persistent kernel (gemm_w4.hip) overlaps the two: a lone wave per SIMD drives the matrix pipe at ~50 %
// (profiles/r02_gemm_debug.md).  Here the roles are swapped:
//   * a wave keeps its 64 token rows x K of the ACTIVATION in registers for the whole work item (the B operand of
//     v_mfma_f32_32x32x16_f16: lane (hi, r) holds X[row r][16 kk + 8 hi .. +8], K/16 x 4 VGPRs per 32 rows) - loaded once
//     straight from global memory in fragment layout, no LDS traffic for it at all;
//   * the WEIGHT streams through LDS in tiles of 32 output channels x K (the A operand), shared by the 4 waves of the
//     workgroup: 3-slot LDS-DMA ring, ONE s_barrier per tile = per 2 x K/16 MFMAs per wave; every ds_read_b128 of a weight
//     fragment feeds 2 MFMAs (the wave's two 32-row blocks);
//   * a tile's 32 x 32 results are complete after K/16 MFMAs and are finished (folded LayerNorm, bias, GEGLU, residual,
//     fp16) and stored right away: there is no accumulator that outlives a tile, no separate epilogue phase - the VALU
//     work of tile t overlaps the MFMAs of the other workgroup on the CU (2 x 4 waves per CU, <= 256 VGPRs each).
//   * weight rows are padded by 16 bytes in LDS (row stride K*2+16 B == 9 or 1 sixteen-byte slots mod 16) which makes
//     the 32-row fragment reads conflict free without a swizzle; the LDS-DMA writes lane-linear 1 KiB pieces, so the
//     padding is realised on the SOURCE side (per-lane global offsets computed once).
//   * work item = (256-row block, range of channel tiles); the channel range is split so that the grid is a few
//     rounds of 2 x CUs workgroups (a 73 728-row GEMM is only 288 row blocks).
// Epilogue vectors (bias, LayerNorm column sums) ride in the spare pieces of the tile's LDS slot.
#include "common.h"
#include "gemm_dma.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int AS_TILE_B = 24 * 1024;   // one ring slot: 21-piece weight tile (K = 320) + bias / col_sum pieces
constexpr int AS_NPW = 6;              // LDS-DMA pieces per wave per tile

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swap32x2(unsigned& a0, unsigned& b0, unsigned& a1, unsigned& b1) {
    asm volatile("s_nop 7\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 3"
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
}
__device__ __forceinline__ unsigned pack_h2(float x, float y) {
    unsigned lo, hi, r;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(lo) : "v"(x));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hi) : "v"(y));
    asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[0]; }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[1]; }

struct AsArgs : insv2v_gemm_desc { int n_splits, tiles_per_item, delay; };

// K: contraction length (compile time: the activation fragments are a register array).  QB: 32-row blocks per wave.
// GEGLU: a tile is 16 value rows + the 16 matching gate rows of W (one 32-row A operand), 16 outputs per token.
// Phase profile (PROF builds, tile code 231): cycles (s_memtime) wave 0 of every workgroup spends per tile in
// [0] the counted vmcnt wait, [1] the barrier, [2] issuing the LDS-DMA of tile t+2, [3] the K/16 x QB MFMAs (until the last
// result is readable), [4] the epilogue up to its last store being issued, [5] per item: activation loads + first wait; [6] tiles, [7] items.
__device__ unsigned long long g_as_prof[8];
#define AS_T(x) do { if (PROF) { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)

template <int K, int QB, bool GEGLU, bool HAS_RES, bool PROF = false>
__global__ __launch_bounds__(256, 2) void gemm_as_kernel(AsArgs p) {
    constexpr int KT = K / 16;
    constexpr int RS = K * 2 + 16;              // LDS bytes per weight row
    constexpr int WPIECES = (32 * RS + 1023) / 1024;
    static_assert(WPIECES <= 21 && AS_NPW * 4 == 24, "weight tile + bias, col_sum, spare pieces fill 24 pieces");
    constexpr int PK_BIAS = 21 * 1024, PK_CS = 22 * 1024;
    constexpr int OUT_W = GEGLU ? 16 : 32;      // output channels per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;

    const int ntn = GEGLU ? (p.N >> 1) / 16 : p.N / 32;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mblk = bid / p.n_splits, ns = bid - mblk * p.n_splits;
    const int t0 = ns * p.tiles_per_item, t1 = min(ntn, t0 + p.tiles_per_item);
    if (t0 >= t1) return;
    unsigned long long tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0, tp5 = 0, tstart = 0;
    unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0};
    AS_T(tstart);
    const int m0 = mblk * (128 * QB) + wid * (32 * QB);

    const srd_t rA = make_srd(p.a), rW = make_srd(p.w), rC = make_srd(p.c), rR = make_srd(p.residual ? p.residual : p.c);
    const bool ln = p.row_stats != nullptr;

    // ---- LDS-DMA source offsets: LDS byte (piece*1024 + lane*16) of the tile image <- weight row / column
    unsigned woff[AS_NPW];
#pragma unroll
    for (int i = 0; i < AS_NPW; ++i) {
        const int off = (wid + 4 * i) * 1024 + lane * 16;
        const int row = off / RS, col = off - row * RS;
        const int srow = GEGLU ? (row < 16 ? row : (p.N >> 1) - 16 + row) : row;   // relative to the tile's first value row
        woff[i] = (row < 32 && col < K * 2) ? (unsigned)(srow * (int)p.ldw * 2 + col) : OOB_OFFSET;
    }
    // bias / col_sum pieces (waves 1 / 2, i = 5): 32 floats; GEGLU: 16 value + 16 gate entries
    const int vlane = GEGLU ? ((lane & 4) ? (p.N >> 1) - 16 + (lane & 3) * 4 + 16 : (lane & 3) * 4) : lane * 4;  // float index rel. tile
    const srd_t rV = make_srd(wid == 1 ? (p.bias ? (const void*)p.bias : p.w) : (ln ? (const void*)p.col_sum : p.w));
    const bool vec_ok = wid == 1 ? p.bias != nullptr : ln;
    auto stage = [&](int t, int slot) {
        char* dst = smem + slot * AS_TILE_B + wid * 1024;
        const int n0 = t * OUT_W;
        const int soff = n0 * (int)p.ldw * 2;
#pragma unroll
        for (int i = 0; i < AS_NPW - 1; ++i) dma16(rW, woff[i], soff, dst + i * 4096);
        if (wid == 0) {
            dma16(rW, woff[AS_NPW - 1], soff, dst + (AS_NPW - 1) * 4096);
        } else if (wid == 3) {
            dma16(rW, OOB_OFFSET, 0, dst + (AS_NPW - 1) * 4096);  // spare piece: keeps every wave at AS_NPW requests per tile
        } else {  // lanes >= 8 fetch out of range (zeros into the unused rest of the piece): one request per wave either way
            dma16(rV, (vec_ok && lane < 8) ? (unsigned)(vlane * 4) : OOB_OFFSET, n0 * 4, dst + (AS_NPW - 1) * 4096);
        }
    };

    // ---- the activation rows of this wave, in B-operand fragment layout
    half8 xf[QB][KT];
    float ra[QB], rm[QB];
    unsigned offc[QB], offr[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int m = m0 + b * 32 + r;
        const unsigned ao = m < p.M ? (unsigned)(m * (int)p.lda * 2 + hi * 16) : OOB_OFFSET;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) xf[b][kk] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rA, ao, kk * 32, 0));
        float mean = 0.f, rstd = 1.f;
        if (ln && m < p.M) { mean = p.row_stats[2 * m]; rstd = p.row_stats[2 * m + 1]; }
        ra[b] = rstd * p.alpha; rm[b] = -rstd * mean;
        offc[b] = m < p.M ? (unsigned)(m * (int)p.ldc * 2 + hi * 16) : OOB_OFFSET;
        offr[b] = m < p.M ? (unsigned)(m * (int)p.ldr * 2 + hi * 16) : OOB_OFFSET;
    }

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int fbase = r * RS + hi * 16;  // fragment read: + slot base + kk*32
    const int oN = GEGLU ? (p.N >> 1) : p.N;

    stage(t0, 0);
    if (t0 + 1 < t1) stage(t0 + 1, 1);
    // De-phase the two workgroups that share a CU: equal work items started together stay in lockstep (both in their MFMA
    // segment, then both in their VALU segment, contending for the same unit each time); the one in the odd wave slot of its
    // SIMD starts half a tile later so that one workgroup's MFMAs run under the other's epilogue.
    if (p.delay > 0 && (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1))  // HW_REG_HW_ID[3:0] = wave slot on the SIMD
        for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(4);

    constexpr int NQP = GEGLU ? 1 : 2;   // 16-channel store groups per tile
    int slot = 0;
    for (int t = t0; t < t1; ++t) {
        // Tile t has landed once only YOUNGER requests are outstanding (a wave's vector-memory operations retire in issue
        // order).  DMA(t) was issued at the top of iteration t-2, so younger = stores(t-2), DMA(t+1), stores(t-1): the wait must
        // not cover the stores - their acknowledgement from L2 takes longer than a tile.  (Residual loads are consumed inside
        // their own iteration.)  The first wait also covers the activation loads.
        constexpr int ST = QB * NQP;   // stores per tile per wave
        const bool more = t + 1 < t1;
        AS_T(tp0);
        if (t == t0) wait_vmcnt<0>();
        else if (t == t0 + 1) { if (more) wait_vmcnt<AS_NPW + ST>(); else wait_vmcnt<ST>(); }
        else { if (more) wait_vmcnt<AS_NPW + 2 * ST>(); else wait_vmcnt<2 * ST>(); }
        AS_T(tp1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // every wave's pieces of tile t are in LDS; every wave is done reading tile t-1
        __builtin_amdgcn_sched_barrier(0);
        AS_T(tp2);
        if (t + 2 < t1) stage(t + 2, slot == 0 ? 2 : slot - 1);
        AS_T(tp3);

        // Weight fragments: inline-asm ds_read_b128 with a hand-counted lgkmcnt so that WDEPTH reads stay in flight under
        // the MFMAs (left to itself hipcc keeps ONE fragment register here - 244 VGPRs are live - and waits for every read
        // right before the MFMA pair that uses it).  LDS returns in order: fragment kk is there once <= min(WDEPTH-1, KT-1-kk)
        // younger reads are outstanding.
        constexpr int WDEPTH = 3;
        const unsigned wt = lds0 + slot * AS_TILE_B + fbase;
        half8 wf[WDEPTH];
#define AS_DSR(dst, kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(wt), "n"((kk) * 32))
        AS_DSR(wf[0], 0);
        AS_DSR(wf[1], 1);
        AS_DSR(wf[2], 2);
        floatx16 acc[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            // the "+v" tie makes the MFMAs below depend on the wait (hipcc does not know the asm above is an LDS read)
            if (kk + 3 <= KT) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[kk % WDEPTH]));
            else if (kk + 2 == KT) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[kk % WDEPTH]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[kk % WDEPTH]));
#pragma unroll
            for (int b = 0; b < QB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk % WDEPTH], xf[b][kk], acc[b], 0, 0, 0);
            if (kk + WDEPTH < KT) AS_DSR(wf[kk % WDEPTH], kk + WDEPTH);
        }
#undef AS_DSR
        if (PROF) {  // a VALU read of the last accumulators: the hardware interlock waits for the MFMAs
            float d0, d1;
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(d0), "=v"(d1) : "v"(acc[0][15]), "v"(acc[QB - 1][15]));
            asm volatile("" :: "v"(d0), "v"(d1));
        }
        AS_T(tp4);

        // ---- finish + store the tile.  Lane (hi, r): token row r of block b, channels n0 + 8q + 4hi + e (value index 4q + e).
        const unsigned pk = lds0 + slot * AS_TILE_B + hi * 16;
        floatx4 bq[4], cq[4];
        asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%9+32\n\tds_read_b128 %2, %8 offset:%9+64\n\t"
                     "ds_read_b128 %3, %8 offset:%9+96\n\tds_read_b128 %4, %8 offset:%10\n\tds_read_b128 %5, %8 offset:%10+32\n\t"
                     "ds_read_b128 %6, %8 offset:%10+64\n\tds_read_b128 %7, %8 offset:%10+96\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(cq[0]), "=&v"(cq[1]), "=&v"(cq[2]), "=&v"(cq[3])
                     : "v"(pk), "n"(PK_BIAS), "n"(PK_CS) : "memory");
        const int on0 = t * OUT_W;
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            uint4v rv[NQP];
            if (HAS_RES) {
#pragma unroll
                for (int qp = 0; qp < NQP; ++qp) {
                    const bool okc = on0 + qp * 16 + hi * 8 + 8 <= oN;
                    rv[qp] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[b] : OOB_OFFSET, (on0 + qp * 16) * 2, 0);
                }
            }
#pragma unroll
            for (int qp = 0; qp < NQP; ++qp) {
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * qp + h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(ra[b], acc[b][4 * q + e], fmaf(rm[b], cq[q][e], bq[q][e]));
                        if (GEGLU) x *= gelu_erf_f(fmaf(ra[b], acc[b][4 * (q + 2) + e], fmaf(rm[b], cq[q + 2][e], bq[q + 2][e])));
                        v[h][e] = x;
                    }
                }
                if (HAS_RES) {
                    unsigned r0 = rv[qp][0], r1 = rv[qp][1], r2 = rv[qp][2], r3 = rv[qp][3];
                    swap32x2(r0, r2, r1, r3);
                    v[0][0] += h_lo(r0); v[0][1] += h_hi(r0); v[0][2] += h_lo(r1); v[0][3] += h_hi(r1);
                    v[1][0] += h_lo(r2); v[1][1] += h_hi(r2); v[1][2] += h_lo(r3); v[1][3] += h_hi(r3);
                }
                unsigned a0 = pack_h2(v[0][0], v[0][1]), a1 = pack_h2(v[0][2], v[0][3]);
                unsigned b0 = pack_h2(v[1][0], v[1][1]), b1 = pack_h2(v[1][2], v[1][3]);
                swap32x2(a0, b0, a1, b1);
                const uint4v out = {a0, a1, b0, b1};
                const bool okc = on0 + qp * 16 + hi * 8 + 8 <= oN;
                __builtin_amdgcn_raw_buffer_store_b128(out, rC, okc ? offc[b] : OOB_OFFSET, (on0 + qp * 16) * 2, 0);
                asm volatile("s_nop 7" ::"v"(out));  // 16-byte store data registers are read late (profiles/r02_gemm_debug.md)
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
        if (PROF) {
            AS_T(tp5);
            if (t == t0) acc_t[5] += tp1 - tstart; else acc_t[0] += tp1 - tp0;
            acc_t[1] += tp2 - tp1; acc_t[2] += tp3 - tp2; acc_t[3] += tp4 - tp3; acc_t[4] += tp5 - tp4;
        }
    }
    if (PROF && tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(&g_as_prof[i], acc_t[i]);
        atomicAdd(&g_as_prof[6], (unsigned long long)(t1 - t0));
        atomicAdd(&g_as_prof[7], 1ull);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Persistent form (tile code 232).  What the phase profile of the kernel above asks for (profiles/r02_gemm_as_experiment.txt):
//   * ONE 4-wave workgroup per CU with the whole register file (512 VGPRs per wave): two accumulator sets, so the epilogue of
//     tile t-1 is issued IN THE SAME instruction stream as the MFMAs of tile t, one slice behind every MFMA (a 32x32x16 MFMA
//     occupies the matrix pipe for 32 cycles = 7 VALU issue slots; a tile's epilogue is ~1200 cycles against 40 MFMAs);
//   * every workgroup owns a CONTIGUOUS range of the (row block, channel tile) sequence, total/gridDim tiles each (+-1): perfect
//     balance, and the 160 KB activation block is loaded once per (at most 2-3) row-block segments instead of once per item.
template <int K, bool GEGLU, bool HAS_RES, bool PROF = false>
__global__ __launch_bounds__(256, 1) void gemm_asp_kernel(AsArgs p) {
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tseg = 0;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int QB = 2, KT = K / 16, RS = K * 2 + 16;
    constexpr int PK_BIAS = 21 * 1024, PK_CS = 22 * 1024;
    constexpr int OUT_W = GEGLU ? 16 : 32, NQP = GEGLU ? 1 : 2;
    constexpr int NG = QB * NQP;        // 16-channel store groups per tile: (b, qp)
    constexpr int SUB = KT / NG;        // k steps (MFMA pairs) per store group
    static_assert(SUB * NG == KT && SUB >= (GEGLU ? 10 : 5), "epilogue slices must fit the k steps");
    constexpr int EP = NG + (HAS_RES ? NG : 0);   // vector-memory operations of one tile's epilogue (stores + residual loads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hi = lane >> 5;
    const int ntn = GEGLU ? (p.N >> 1) / 16 : p.N / 32;
    const int mblocks = (p.M + 255) / 256;
    const long total = (long)mblocks * ntn;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int g0 = (int)(total * v / gridDim.x), g1 = (int)(total * (v + 1) / gridDim.x);
    if (g0 >= g1) return;

    const srd_t rA = make_srd(p.a), rW = make_srd(p.w), rC = make_srd(p.c), rR = make_srd(p.residual ? p.residual : p.c);
    const bool ln = p.row_stats != nullptr;
    unsigned woff[AS_NPW];
#pragma unroll
    for (int i = 0; i < AS_NPW; ++i) {
        const int off = (wid + 4 * i) * 1024 + lane * 16;
        const int row = off / RS, col = off - row * RS;
        const int srow = GEGLU ? (row < 16 ? row : (p.N >> 1) - 16 + row) : row;
        woff[i] = (row < 32 && col < K * 2) ? (unsigned)(srow * (int)p.ldw * 2 + col) : OOB_OFFSET;
    }
    const int vlane = GEGLU ? ((lane & 4) ? (p.N >> 1) + (lane & 3) * 4 : (lane & 3) * 4) : lane * 4;
    const srd_t rV = make_srd(wid == 1 ? (p.bias ? (const void*)p.bias : p.w) : (ln ? (const void*)p.col_sum : p.w));
    const bool vec_ok = wid == 1 ? p.bias != nullptr : ln;
    auto stage = [&](int t, int slot) {
        char* dst = smem + slot * AS_TILE_B + wid * 1024;
        const int n0 = t * OUT_W;
        const int soff = n0 * (int)p.ldw * 2;
#pragma unroll
        for (int i = 0; i < AS_NPW - 1; ++i) dma16(rW, woff[i], soff, dst + i * 4096);
        if (wid == 0) dma16(rW, woff[AS_NPW - 1], soff, dst + (AS_NPW - 1) * 4096);
        else if (wid == 3) dma16(rW, OOB_OFFSET, 0, dst + (AS_NPW - 1) * 4096);
        else dma16(rV, (vec_ok && lane < 8) ? (unsigned)(vlane * 4) : OOB_OFFSET, n0 * 4, dst + (AS_NPW - 1) * 4096);
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int fbase = r * RS + hi * 16;
    const int oN = GEGLU ? (p.N >> 1) : p.N;

    for (int g = g0; g < g1;) {
        // ================= one segment: tiles [ta, tb) of row block mblk =================
        const int mblk = g / ntn, ta = g - mblk * ntn, tb = min(ntn, ta + (g1 - g));
        g += tb - ta;
        const int m0 = mblk * 256 + wid * 64;
        AS_T(tseg);
        stage(ta, 0);
        if (ta + 1 < tb) stage(ta + 1, 1);
        half8 xf[QB][KT];
        float ra[QB], rm[QB];
        unsigned offc[QB], offr[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const int m = m0 + b * 32 + r;
            const unsigned ao = m < p.M ? (unsigned)(m * (int)p.lda * 2 + hi * 16) : OOB_OFFSET;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) xf[b][kk] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rA, ao, kk * 32, 0));
            float mean = 0.f, rstd = 1.f;
            if (ln && m < p.M) { mean = p.row_stats[2 * m]; rstd = p.row_stats[2 * m + 1]; }
            ra[b] = rstd * p.alpha; rm[b] = -rstd * mean;
            offc[b] = m < p.M ? (unsigned)(m * (int)p.ldc * 2 + hi * 16) : OOB_OFFSET;
            offr[b] = m < p.M ? (unsigned)(m * (int)p.ldr * 2 + hi * 16) : OOB_OFFSET;
        }

        floatx16 acc[2][QB];     // [tile parity][row block]
        floatx4 bq[4], cq[4];   // bias / column sums of the PENDING tile (read at the end of its own MFMA loop)
        // one slice of the PENDING tile's epilogue (tile t-1: accumulators / vectors of parity PP), placed behind the MFMAs of
        // k step kk; `half` 0 = between the step's two MFMAs, 1 = after the second
        float ev[2][4];
        uint4v rv[NG];
        unsigned pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;
        auto epi_slice = [&](auto PPc, int kk, int half, int on0) {
            constexpr int PP = decltype(PPc)::value;
            const int gi = kk / SUB, s = kk - gi * SUB;
            const int b = gi / NQP, qp = gi - b * NQP;
            if (!GEGLU) {
                if (s < 2) {              // 8 channels of quarter q = 2 qp + s: folded LayerNorm + bias
                    const int q = 2 * qp + s;
#pragma unroll
                    for (int e = 2 * half; e < 2 * half + 2; ++e)
                        ev[s][e] = fmaf(ra[b], acc[PP][b][4 * q + e], fmaf(rm[b], cq[q][e], bq[q][e]));
                } else if (s == 2) {
                    if (HAS_RES) {
                        if (half == 0) {
                            unsigned r0 = rv[gi][0], r1 = rv[gi][1], r2 = rv[gi][2], r3 = rv[gi][3];
                            swap32x2(r0, r2, r1, r3);
                            rv[gi][0] = r0; rv[gi][1] = r1; rv[gi][2] = r2; rv[gi][3] = r3;
                        } else {
                            ev[0][0] += h_lo(rv[gi][0]); ev[0][1] += h_hi(rv[gi][0]); ev[0][2] += h_lo(rv[gi][1]); ev[0][3] += h_hi(rv[gi][1]);
                            ev[1][0] += h_lo(rv[gi][2]); ev[1][1] += h_hi(rv[gi][2]); ev[1][2] += h_lo(rv[gi][3]); ev[1][3] += h_hi(rv[gi][3]);
                        }
                    }
                } else if (s == 3) {
                    if (half == 0) {
                        pa0 = pack_h2(ev[0][0], ev[0][1]); pa1 = pack_h2(ev[0][2], ev[0][3]);
                        pb0 = pack_h2(ev[1][0], ev[1][1]); pb1 = pack_h2(ev[1][2], ev[1][3]);
                    } else {
                        swap32x2(pa0, pb0, pa1, pb1);
                    }
                } else if (s == 4 && half == 0) {
                    const uint4v out = {pa0, pa1, pb0, pb1};
                    const bool okc = on0 + qp * 16 + hi * 8 + 8 <= oN;
                    __builtin_amdgcn_raw_buffer_store_b128(out, rC, okc ? offc[b] : OOB_OFFSET, (on0 + qp * 16) * 2, 0);
                    asm volatile("s_nop 7" ::"v"(out));
                }
            } else {
                if (s < 8) {              // output o = s: quarter h = s / 4 (value) and h + 2 (gate), element e = s % 4
                    const int h = s >> 2, e = s & 3;
                    if (half == 0) ev[h][e] = gelu_erf_f(fmaf(ra[b], acc[PP][b][4 * (h + 2) + e], fmaf(rm[b], cq[h + 2][e], bq[h + 2][e])));
                    else ev[h][e] *= fmaf(ra[b], acc[PP][b][4 * h + e], fmaf(rm[b], cq[h][e], bq[h][e]));
                } else if (s == 8) {
                    if (half == 0) {
                        pa0 = pack_h2(ev[0][0], ev[0][1]); pa1 = pack_h2(ev[0][2], ev[0][3]);
                        pb0 = pack_h2(ev[1][0], ev[1][1]); pb1 = pack_h2(ev[1][2], ev[1][3]);
                    } else {
                        swap32x2(pa0, pb0, pa1, pb1);
                    }
                } else if (s == 9 && half == 0) {
                    const uint4v out = {pa0, pa1, pb0, pb1};
                    const bool okc = on0 + hi * 8 + 8 <= oN;
                    __builtin_amdgcn_raw_buffer_store_b128(out, rC, okc ? offc[b] : OOB_OFFSET, on0 * 2, 0);
                    asm volatile("s_nop 7" ::"v"(out));
                }
            }
        };
        // one tile: wait for its weights, request tile t+2, K/16 x 2 MFMAs into acc[P] with the pending epilogue (parity P^1) woven in
        auto tile = [&](auto Pc, auto EPIc, int t, int slot, int j) {
            constexpr int P = decltype(Pc)::value;
            constexpr bool EPI = decltype(EPIc)::value;
            // Tile t has landed once only YOUNGER requests are outstanding (issue-order retirement).  j = tiles since the segment
            // started: 0 waits for everything (activation loads included); DMA(t) was issued in iteration j-2 (prologue for j < 2),
            // younger are DMA(t+1) and the epilogue operations issued since: none in iteration 0 (no pending tile).
            const bool more = t + 1 < tb;
            AS_T(tq0);
            if (j == 0) wait_vmcnt<0>();
            else if (j == 1) { if (more) wait_vmcnt<AS_NPW>(); else wait_vmcnt<0>(); }
            else if (j == 2) { if (more) wait_vmcnt<AS_NPW + EP>(); else wait_vmcnt<EP>(); }
            else { if (more) wait_vmcnt<AS_NPW + 2 * EP>(); else wait_vmcnt<2 * EP>(); }
            AS_T(tq1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            AS_T(tq2);
            if (t + 2 < tb) stage(t + 2, slot == 0 ? 2 : slot - 1);
            AS_T(tq3);
            const int on0p = (t - 1) * OUT_W;   // the pending tile's first output channel
            if (EPI && HAS_RES) {
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int b = gi / NQP, qp = gi - b * NQP;
                    const bool okc = on0p + qp * 16 + hi * 8 + 8 <= oN;
                    rv[gi] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[b] : OOB_OFFSET, (on0p + qp * 16) * 2, 0);
                }
            }
            constexpr int WDEPTH = 3;
            const unsigned wt = lds0 + slot * AS_TILE_B + fbase;
            half8 wf[WDEPTH];
#define AS_DSR(dst, kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(wt), "n"((kk) * 32))
            AS_DSR(wf[0], 0);
            AS_DSR(wf[1], 1);
            AS_DSR(wf[2], 2);
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[P][b][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                // LDS returns in order: <= 2 younger reads outstanding => fragment kk (and the older bias / column-sum reads) are in.
                // The "+v" ties make the consumers depend on the wait.
                if (kk + 3 <= KT) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[kk % WDEPTH]));
                else if (kk + 2 == KT) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[kk % WDEPTH]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[kk % WDEPTH]));
                acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk % WDEPTH], xf[0][kk], acc[P][0], 0, 0, 0);
                if (EPI) { __builtin_amdgcn_sched_barrier(0); epi_slice(std::integral_constant<int, P ^ 1>{}, kk, 0, on0p); __builtin_amdgcn_sched_barrier(0); }
                acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk % WDEPTH], xf[1][kk], acc[P][1], 0, 0, 0);
                if (EPI) { __builtin_amdgcn_sched_barrier(0); epi_slice(std::integral_constant<int, P ^ 1>{}, kk, 1, on0p); __builtin_amdgcn_sched_barrier(0); }
                if (kk + WDEPTH < KT) AS_DSR(wf[kk % WDEPTH], kk + WDEPTH);
            }
#undef AS_DSR
            // this tile's bias / column sums for ITS epilogue (woven into the next tile, or the flush): the pending tile's slices are
            // done with the old values by now; the wait sits here, after the MFMAs are queued
            const unsigned pk = lds0 + slot * AS_TILE_B + hi * 16;
            floatx4 nb0, nb1, nb2, nb3, nc0, nc1, nc2, nc3;
            asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%9+32\n\tds_read_b128 %2, %8 offset:%9+64\n\t"
                         "ds_read_b128 %3, %8 offset:%9+96\n\tds_read_b128 %4, %8 offset:%10\n\tds_read_b128 %5, %8 offset:%10+32\n\t"
                         "ds_read_b128 %6, %8 offset:%10+64\n\tds_read_b128 %7, %8 offset:%10+96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(nb0), "=&v"(nb1), "=&v"(nb2), "=&v"(nb3), "=&v"(nc0), "=&v"(nc1), "=&v"(nc2), "=&v"(nc3)
                         : "v"(pk), "n"(PK_BIAS), "n"(PK_CS) : "memory");
            bq[0] = nb0; bq[1] = nb1; bq[2] = nb2; bq[3] = nb3; cq[0] = nc0; cq[1] = nc1; cq[2] = nc2; cq[3] = nc3;
            if (PROF) {
                AS_T(tq4);
                if (j == 0) acc_t[5] += tq1 - tseg; else acc_t[0] += tq1 - tq0;
                acc_t[1] += tq2 - tq1; acc_t[2] += tq3 - tq2; acc_t[3] += tq4 - tq3;
            }
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int slot = 0, j = 0, t = ta;
        tile(P0{}, std::false_type{}, t, slot, j);
        ++t; ++j; slot = 1;
        int par = 1;  // parity of the NEXT tile
        while (t < tb) {
            if (par) tile(P1{}, std::true_type{}, t, slot, j); else tile(P0{}, std::true_type{}, t, slot, j);
            par ^= 1; ++t; ++j; slot = slot == 2 ? 0 : slot + 1;
        }
        // flush the last tile's epilogue (parity par^1), then make the ring reusable by the next segment
        {
            const int on0p = (tb - 1) * OUT_W;
            if (HAS_RES) {
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int b = gi / NQP, qp = gi - b * NQP;
                    const bool okc = on0p + qp * 16 + hi * 8 + 8 <= oN;
                    rv[gi] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[b] : OOB_OFFSET, (on0p + qp * 16) * 2, 0);
                }
            }
            if (par) {
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) { epi_slice(P0{}, kk, 0, on0p); epi_slice(P0{}, kk, 1, on0p); }
            } else {
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) { epi_slice(P1{}, kk, 0, on0p); epi_slice(P1{}, kk, 1, on0p); }
            }
        }
        if (PROF) { unsigned long long te; AS_T(te); acc_t[4] += te - tq4; acc_t[6] += tb - ta; acc_t[7] += 1; }
        if (g < g1) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();   // every wave is done with the ring before the next segment restages slots 0 / 1
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PROF && tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&g_as_prof[i], acc_t[i]);
    }
}

template <int K, int QB, bool GEGLU, bool HAS_RES, bool PROF = false>
int launch_as(const insv2v_gemm_desc& d, hipStream_t s) {
    static const int lds_pad = getenv("INSV2V_AS_LDSPAD") ? atoi(getenv("INSV2V_AS_LDSPAD")) : 0;  // debugging: force one workgroup per CU
    const int LDS_B = 3 * AS_TILE_B + lds_pad;
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_as_kernel<K, QB, GEGLU, HAS_RES, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int bm = 128 * QB;
    const int mblocks = (d.M + bm - 1) / bm;
    const int ntn = GEGLU ? (d.N / 2) / 16 : d.N / 32;
    // split the channel tiles of a row block over n_splits work items: fill whole rounds of 2 workgroups per CU, but keep
    // >= 3 tiles per item (an item re-reads its activation block, ~1.5 tiles' worth of time)
    static const int dbg_split = getenv("INSV2V_AS_SPLIT") ? atoi(getenv("INSV2V_AS_SPLIT")) : 0;
    int best = 1;
    double best_score = -1.0;
    const int slots = 2 * num_cu;
    for (int sp = 1; sp <= ntn; ++sp) {
        const int tpi = (ntn + sp - 1) / sp;
        if (tpi < 2 && sp > 1) break;
        const int used = (ntn + tpi - 1) / tpi;  // items that actually have tiles
        if (used != sp) continue;
        const long items = (long)mblocks * sp;
        const long rounds = (items + slots - 1) / slots;
        const double score = (double)items / (double)(rounds * slots) * (double)tpi / (tpi + 1.5);
        if (score > best_score) { best_score = score; best = sp; }
    }
    if (dbg_split > 0) best = dbg_split < ntn ? dbg_split : ntn;
    AsArgs a;
    static_cast<insv2v_gemm_desc&>(a) = d;
    a.n_splits = best;
    a.tiles_per_item = (ntn + best - 1) / best;
    static const int delay = getenv("INSV2V_AS_DELAY") ? atoi(getenv("INSV2V_AS_DELAY")) : 0;
    a.delay = delay;
    hipLaunchKernelGGL((gemm_as_kernel<K, QB, GEGLU, HAS_RES, PROF>), dim3(mblocks * best), dim3(256), LDS_B, s, a);
    return launch_status();
}

template <int K, bool GEGLU, bool HAS_RES, bool PROF = false>
int launch_asp(const insv2v_gemm_desc& d, hipStream_t s) {
    constexpr int LDS_B = 3 * AS_TILE_B;
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_asp_kernel<K, GEGLU, HAS_RES, PROF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const long total = (long)((d.M + 255) / 256) * (GEGLU ? (d.N / 2) / 16 : d.N / 32);
    const int grid = total < num_cu ? (int)total : num_cu;
    AsArgs a;
    static_cast<insv2v_gemm_desc&>(a) = d;
    a.n_splits = a.tiles_per_item = a.delay = 0;
    hipLaunchKernelGGL((gemm_asp_kernel<K, GEGLU, HAS_RES, PROF>), dim3(grid), dim3(256), LDS_B, s, a);
    return launch_status();
}

template <int K, int QB>
int dispatch_as(const insv2v_gemm_desc& d, hipStream_t s) {
    if (d.act == INSV2V_ACT_GEGLU) return launch_as<K, QB, true, false>(d, s);
    return d.residual ? launch_as<K, QB, false, true>(d, s) : launch_as<K, QB, false, false>(d, s);
}

}  // namespace

// Status (round 2): EXPERIMENTAL, reachable only with tile codes 230 (this file's first kernel), 232 (persistent form) and
// 231 / 233 (their phase-profile builds).  Correct for linear / folded-LayerNorm / residual
// problems with K = 320; on the level-0 shapes it is within +-10 % of the dispatched kernels (fused q/k/v 73 728 x 960 x 320:
// 67-75 us vs 77-80 us; out-proj + residual 46 vs 42 us), i.e. not the 2x its structure promised: a lone wave needs
// ~1.9 us per 32-channel tile where its instruction stream adds up to ~1.3 us, and two workgroups per CU overlap by only 1.36x
// (profiles/r02_gemm_as_experiment.txt).  The GEGLU epilogue below is wired but NOT validated - the entry point rejects it.
int insv2v_gemm_as(const insv2v_gemm_desc& d, int variant, hipStream_t s) {
    if (d.act == INSV2V_ACT_GEGLU) return INSV2V_EUNSUPPORTED;  // the GEGLU epilogues are wired but fail the harness check: rejected
    if (d.mode != INSV2V_MODE_LINEAR || d.batch > 1 || d.c_fp32 || d.split_k > 1 || d.k_split || d.row_bias) return INSV2V_EUNSUPPORTED;
    const bool gg = d.act == INSV2V_ACT_GEGLU;
    if (!gg && d.act != INSV2V_ACT_NONE) return INSV2V_EUNSUPPORTED;
    if (gg ? ((d.N % 32) || d.residual) : (d.N % 32)) return INSV2V_EUNSUPPORTED;
    if ((d.lda & 7) || (d.ldw & 7) || (d.ldc & 7) || ((uintptr_t)d.a & 15) || ((uintptr_t)d.w & 15) || ((uintptr_t)d.c & 15)) return INSV2V_EUNSUPPORTED;
    if (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15))) return INSV2V_EUNSUPPORTED;
    if (d.bias && ((uintptr_t)d.bias & 15)) return INSV2V_EUNSUPPORTED;
    if (d.row_stats && (!d.col_sum || ((uintptr_t)d.col_sum & 15))) return INSV2V_EUNSUPPORTED;
    if ((int64_t)d.M * d.ldc * 2 >= ((int64_t)1 << 31) || (int64_t)d.M * d.lda * 2 >= ((int64_t)1 << 31) ||
        (int64_t)d.N * d.ldw * 2 >= ((int64_t)1 << 31) || (d.residual && (int64_t)d.M * d.ldr * 2 >= ((int64_t)1 << 31)))
        return INSV2V_EUNSUPPORTED;
    if (d.K == 320) {
        if (variant == 1) return d.residual ? INSV2V_EUNSUPPORTED : launch_as<320, 2, false, false, true>(d, s);  // phase profile
        if (variant == 3) return (gg || d.residual) ? INSV2V_EUNSUPPORTED : launch_asp<320, false, false, true>(d, s);  // phase profile of the persistent form
        if (variant == 2) {  // persistent form
            if (gg) return launch_asp<320, true, false>(d, s);
            return d.residual ? launch_asp<320, false, true>(d, s) : launch_asp<320, false, false>(d, s);
        }
        return dispatch_as<320, 2>(d, s);
    }
    return INSV2V_EUNSUPPORTED;
}

// debugging aid of tools/gemm_check (not part of the public ABI): read and clear the phase profile of the PROF build
extern "C" int insv2v_debug_gemm_as_profile(unsigned long long* out8) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return INSV2V_EINVAL;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_as_prof), sizeof(z)) != hipSuccess) return INSV2V_EINVAL;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_as_prof), z, sizeof(z)) != hipSuccess) return INSV2V_EINVAL;
    return 0;
}
