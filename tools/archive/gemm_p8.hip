// gemm_p8: 256 x 256 fp16 MFMA GEMM / implicit-GEMM 3x3 convolution with an 8-phase (ping-pong) K loop.
//
// Why a second GEMM kernel: the 128 x 128 tile of gemm.hip streams 32 KB of operands per 2.1 MFLOP and is capped by
// the L2 -> LDS operand stream (~70 GB/s per CU, profiles/r01_gemm_decomposition.txt); its 32 x 64 wave tile also
// keeps the LDS read port ~75 % busy.  Here one workgroup (8 waves, 1 per CU, 128 KiB LDS ring) owns a 256 x 256
// tile: half the operand bytes per FLOP, and each wave owns a 128 x 64 sub-tile (2x less LDS read traffic per FLOP).
//
// Structure (cdna_hip_programming.md section 5, "256^2 8-phase template"; schedule derived here):
//  * A K tile (64 wide) lives in LDS as four 16 KiB half-tiles [A_lo, A_hi, W_lo, W_hi] (128 rows x 128 B, chunk
//    index XOR-swizzled by (row>>1)&7 on the DMA *source* side and again on the fragment read), double buffered.
//  * Waves are 2 (token halves, wm) x 4 (64-channel groups, wn).  A wave's 128 x 64 output is four 64 x 32
//    quadrants; one PHASE = {ds_read the register sub-tile the quadrant needs, issue ONE half-tile of LDS-DMA for a
//    later K tile} -> s_barrier -> 8 MFMA 32x32x16 (quadrant x K=64) -> s_barrier.  4 phases per K tile.
//  * Waves 4-7 (wm = 1) run one barrier interval behind waves 0-3, so on every SIMD one wave is in its MFMA segment
//    while its partner reads LDS / issues DMA (s_setprio(1) around the MFMAs arbitrates in favour of the former).
//  * LDS-DMA completion is awaited with a COUNTED s_waitcnt vmcnt(N) once per K tile (never 0 in steady state), one
//    phase before the first read of the data, so loads stay in flight across barriers.
//  * Hazard rules used (same section): read a staged half one phase after the wait that retires it; restage a half
//    >= 2 phases after its last ds_read (SCHED 0), or 1 phase after when an lgkmcnt(0) ahead of the reading phase's
//    first barrier retired the reads (SCHED 1, one phase more prefetch distance).
//
// Quadrant order (a0,w0) (a0,w1) (a1,w1) (a1,w0): reads per phase 12 / 4 / 8 / 0 ds_read_b128; W halves are last
// read in phase 1, A halves in phase 2, which is what lets the next-but-one K tile start streaming in phase 2/3.
//
// Persistent: the grid is one workgroup per CU; each walks tiles v = blockIdx.x, +gridDim.x, ... (same XCD-aware
// rasterisation as gemm.hip) and the LDS-DMA stream simply continues into the next tile's first K tiles while the
// current tile is finished, so a tile's first-load latency is hidden behind the previous tile's epilogue - this is
// what the many short-K (K = 320 ... 1280: 5-20 K tiles) GEMMs of the UNet need.
//
// Epilogue: same semantics as gemm.hip (alpha, folded LayerNorm, bias, row bias, SiLU / quick-GELU / GEGLU, residual,
// one rounding to fp16) but without LDS (the ring belongs to the next tile by then): a lane holds 4 consecutive
// channels of one token per 32x32 fragment quarter; v_permlane32_swap pairs two quarters so every lane owns 8
// consecutive channels = one 16-byte store (guide T21); residual rows are fetched with the same 16-byte pattern and
// un-swapped.  Bias / column sums / token statistics / the tile's row-bias vector arrive by LDS-DMA into a double-buffered
// park area behind the ring.
#include "common.h"
#include "gemm_dma.h"
#include <type_traits>

namespace {

constexpr int HALF_B = 128 * 128;       // one half-tile: 128 rows x 64 halfs
constexpr int RING_B = 8 * HALF_B;      // [part 0..3][buffer 0..1]: slot (part*2 + b)
constexpr int PARK_B = 5 * 1024;        // bias[256], col_sum[256], (mean, rstd)[256], row_bias[256]
constexpr int LDS_B = RING_B + 2 * PARK_B;

#define SB() __builtin_amdgcn_sched_barrier(0)
#define BARRIER() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

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
__device__ __forceinline__ unsigned pack_h2(float x, float y) {
    unsigned lo, hi, r;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(lo) : "v"(x));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hi) : "v"(y));
    asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[0]; }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[1]; }

template <int MODE, int SCHED, bool GEGLU, bool HAS_RES, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_p8_kernel(insv2v_gemm_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;

    // ---- tile rasterisation: XCD-aware remap, then groups of GROUP_M tile rows walked column by column ----
    const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + 255) >> 8, ntiles = tiles_m * tiles_n;
    auto tile_origin = [&](int v, int& bm0, int& bn0) {
        const int bid = xcd_remap(v, ntiles);
        constexpr int GROUP_M = 8;
        const int per_group = GROUP_M * tiles_n;
        const int gidx = bid / per_group, first_m = gidx * GROUP_M;
        const int gsz = min(GROUP_M, tiles_m - first_m), rin = bid - gidx * per_group;
        const int tn = rin / gsz, tm = first_m + rin - tn * gsz;
        bm0 = tm << 8; bn0 = tn << 8;
    };

    const srd_t rA = make_srd(p.a), rA2 = make_srd(p.a2 ? p.a2 : p.a), rW = make_srd(p.w);
    const bool ln = p.row_stats != nullptr;
    const srd_t rBias = make_srd(p.bias ? (const void*)p.bias : p.w), rCs = make_srd(ln ? (const void*)p.col_sum : p.w),
                rSt = make_srd(ln ? (const void*)p.row_stats : p.w), rRb = make_srd(p.row_bias ? (const void*)p.row_bias : p.w);

    // ---- staging side: the K-tile stream runs ahead of the compute side, across tile boundaries ----
    // a half-tile is 16 pieces of 1 KiB (8 rows x 128 B); wave `wid` fills pieces wid and wid+8:
    // row (within the half) = i*64 + wid*8 + lane/8, LDS chunk slot lane%8, source chunk = slot ^ ((row>>1)&7)
    const int prow = wid * 8 + (lane >> 3);
    const int chunk8 = ((lane & 7) ^ ((prow >> 1) & 7)) * 8;  // halfs
    int arow[4];            // linear: token row m (or -1); conv: first pixel index of the row's image (or -1)
    int aoh[4], aow[4];     // conv: output position * stride - pad
    unsigned woff[4];
    const int nk = DBG == 1 ? 1 : p.K / BK;  // DBG 1: one K tile only (timing ablation, wrong results)
    const int IHu = p.upsample ? p.IH * 2 : p.IH, IWu = p.upsample ? p.IW * 2 : p.IW;
    const int ups = p.upsample ? 1 : 0;
    struct Cursor { int v, kt, k0, kh, kw, ci0; } cur = {(int)blockIdx.x, 0, 0, 0, 0, 0};  // wave-uniform scalars
    auto set_stage_tile = [&](int v) {
        int bm0, bn0;
        tile_origin(v, bm0, bn0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // r = half*2 + i
            const int row = (r >> 1) * 128 + (r & 1) * 64 + prow;
            const int m = bm0 + row;
            if (MODE == INSV2V_MODE_LINEAR) {
                arow[r] = m < p.M ? m : -1;
                aoh[r] = aow[r] = 0;
            } else {
                const int mm = m < p.M ? m : 0;
                const int ow = mm % p.OW, t = mm / p.OW;
                const int oh = t % p.OH, nb = t / p.OH;
                arow[r] = m < p.M ? nb * p.IH * p.IW : -1;
                aoh[r] = oh * p.stride - p.pad_t;
                aow[r] = ow * p.stride - p.pad_l;
            }
            const int n = bn0 + row;
            woff[r] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + chunk8) * 2) : OOB_OFFSET;
        }
    };
    auto advance = [&]() {  // next K tile of the stream
        if (++cur.kt == nk) {
            cur.v += gridDim.x; cur.kt = 0; cur.k0 = 0; cur.kh = cur.kw = cur.ci0 = 0;
            if (cur.v < ntiles) set_stage_tile(cur.v);
            return;
        }
        cur.k0 += BK;
        if (MODE != INSV2V_MODE_LINEAR) {
            cur.ci0 += BK;
            if (cur.ci0 >= p.Cin) {
                cur.ci0 = 0;
                if (++cur.kw == 3) { cur.kw = 0; ++cur.kh; }
            }
        }
    };
    // LDS slot of (part, buffer): part 0 A_lo, 1 A_hi, 2 W_lo, 3 W_hi
    auto slot = [&](int part, int b) { return smem + (part * 2 + b) * HALF_B; };
    auto stage_w = [&](int b, int h) {
        char* dst = slot(2 + h, b) + wid * 1024;
        const int soff = cur.k0 * 2;
        dma16(rW, woff[h * 2], soff, dst);
        dma16(rW, woff[h * 2 + 1], soff, dst + 8192);
    };
    auto stage_a = [&](int b, int h) {
        char* dst = slot(h, b) + wid * 1024;
        if (MODE == INSV2V_MODE_LINEAR) {
            const bool second = p.k_split > 0 && cur.k0 >= p.k_split;
            const int ld = (int)(second ? p.lda2 : p.lda);
            const int soff = (second ? cur.k0 - p.k_split : cur.k0) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = arow[h * 2 + i];
                const unsigned v = m >= 0 ? (unsigned)((m * ld + chunk8) * 2) : OOB_OFFSET;
                dma16(second ? rA2 : rA, v, soff, dst + i * 8192);
            }
        } else {
            const bool second = p.k_split > 0 && cur.ci0 >= p.k_split;
            const int ld = (int)(second ? p.lda2 : p.lda);
            const int soff = (second ? cur.ci0 - p.k_split : cur.ci0) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = h * 2 + i;
                const int ih = aoh[r] + cur.kh, iw = aow[r] + cur.kw;
                const bool ok = arow[r] >= 0 && (unsigned)ih < (unsigned)IHu && (unsigned)iw < (unsigned)IWu;
                const int pix = arow[r] + (ih >> ups) * p.IW + (iw >> ups);
                dma16(second ? rA2 : rA, ok ? (unsigned)((pix * ld + chunk8) * 2) : OOB_OFFSET, soff, dst + i * 8192);
            }
        }
    };
    // Park area of tile parity pb: bias | col_sum | (mean, rstd) | tile-uniform row bias.  One LDS-DMA piece per wave
    // (waves 0-4), out-of-range entries arrive as zeros; a wave's own counted vmcnt retires its piece with the ring.
    auto row_group = [&](int m) { int g = m / p.rows_per_group; if (p.rb_mod > 0) g %= p.rb_mod; return g; };
    auto stage_park = [&](int pb, int bm0, int bn0) {
        char* dst = smem + RING_B + pb * PARK_B + wid * 1024;
        if (wid == 0) {
            const int n = bn0 + lane * 4;
            dma16(rBias, (p.bias && n < p.N) ? (unsigned)(n * 4) : OOB_OFFSET, 0, dst);
        } else if (wid == 1) {
            const int n = bn0 + lane * 4;
            dma16(rCs, (ln && n < p.N) ? (unsigned)(n * 4) : OOB_OFFSET, 0, dst);
        } else if (wid == 2 || wid == 3) {
            const int m = bm0 + (wid - 2) * 128 + lane * 2;
            dma16(rSt, (ln && m < p.M) ? (unsigned)(m * 8) : OOB_OFFSET, 0, dst);
        } else if (wid == 4) {
            const int n = bn0 + lane * 4;
            const int g = p.row_bias ? row_group(bm0) : 0;  // every row of the tile is in this group (checked on the host)
            dma16(rRb, (p.row_bias && n < p.N) ? (unsigned)((g * (int)p.ld_rb + n) * 4) : OOB_OFFSET, 0, dst);
        }
    };

    // ---- fragment addressing (bytes): row = ... + (lane & 31), 16-byte chunk (kk*2 + lane/32) ^ ((row>>1)&7)
    const int frow = lane & 31, fhi = lane >> 5, fsw = (frow >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) coff[kk] = ((kk * 2 + fhi) ^ fsw) * 16;
    const char* aBase = smem + (wm * 2) * HALF_B + frow * 128;                           // A half wm, buffer 0
    const char* wBase = smem + ((2 + (wn >> 1)) * 2) * HALF_B + ((wn & 1) * 64 + frow) * 128;  // W half wn/2, buffer 0

    half8 fa[2][4], fw0[4], fw1[4];
    floatx16 acc[2][2][2];  // [iq (channel quadrant)][jq (token quadrant)][j]
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][k][r] = 0.f;
    };
    auto read_a = [&](int b, int jq) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                fa[j][kk] = *(const half8*)(aBase + b * HALF_B + (jq * 64 + j * 32) * 128 + coff[kk]);
    };
    auto read_w = [&](int b, int iq, half8 (&fw)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fw[kk] = *(const half8*)(wBase + b * HALF_B + iq * 32 * 128 + coff[kk]);
    };
    auto mma = [&](floatx16 (&c)[2], const half8 (&fw)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk], fa[j][kk], c[j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of the tile at (bm0, bn0), park buffer pb; no LDS ring access, no barriers ----
    // Straight-line on purpose: hipcc's waitcnt pass turns every control-flow join and every compiler-visible LDS read
    // that follows an LDS-DMA into s_waitcnt vmcnt(0), which would wait for this tile's own global stores and the next
    // tile's DMA stream.  So: bounds are enforced with out-of-range buffer offsets (loads return 0, stores are dropped),
    // a missing residual reads zeros the same way, and the park vectors are read with inline-asm ds_read + one manual
    // lgkmcnt wait per channel group.  Channel-group major: bias / column sums are read once per 16 channels.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const srd_t rC = make_srd(p.c), rR = make_srd(p.residual ? p.residual : p.c);
    // park reads: loads AND their wait inside one asm statement (early-clobber outputs), so the compiler never touches a
    // destination register before the data has landed (guide 5.7 item 1)
    auto park6 = [&](unsigned a, floatx4& b0, floatx4& b1, floatx4& r0, floatx4& r1, floatx4& c0, floatx4& c1) {
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:32\n\tds_read_b128 %2, %6 offset:4096\n\t"
                     "ds_read_b128 %3, %6 offset:4128\n\tds_read_b128 %4, %6 offset:1024\n\tds_read_b128 %5, %6 offset:1056\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(b0), "=&v"(b1), "=&v"(r0), "=&v"(r1), "=&v"(c0), "=&v"(c1) : "v"(a) : "memory");
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
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s += acc[i][j][k][r];
            if (s == 12345.678f) ((half_t*)p.c)[tid] = (half_t)s;
            return;
        }
        const unsigned park = lds0 + RING_B + pb * PARK_B;  // bias | +1024 col_sum | +2048 (mean, rstd) | +4096 row bias
        constexpr int NIQ = GEGLU ? 1 : 2;
        const int oN = GEGLU ? (p.N >> 1) : p.N;
        const int nl0 = wn * 64;                                           // tile-local first channel of this wave
        const int on0 = GEGLU ? ((bn0 + nl0) >> 1) : bn0 + nl0;            // first output column of this wave
        const int m0 = bm0 + wm * 128 + frow;
        // per row block (32 tokens): statistics and the byte offsets of this lane's row in C / residual
        // v = rstd * (alpha * acc - mean * col_sum) + bias  ==  fma(ra, acc, fma(rm, col_sum, bias))
        float ra[4], rm[4];
        unsigned offc[4], offr[4];
        {
            float2 st[4];
            stat4(park + 2048 + (wm * 128 + frow) * 8, st[0], st[1], st[2], st[3]);
#pragma unroll
            for (int rbk = 0; rbk < 4; ++rbk) {
                const int m = m0 + rbk * 32;
                const float mean = ln ? st[rbk].x : 0.f, rstd = ln ? st[rbk].y : 1.f;
                ra[rbk] = rstd * p.alpha; rm[rbk] = -rstd * mean;
                offc[rbk] = (m < p.M && DBG != 3) ? (unsigned)(m * (int)p.ldc * 2 + fhi * 16) : OOB_OFFSET;
                offr[rbk] = (m < p.M && DBG != 4) ? (unsigned)(m * (int)p.ldr * 2 + fhi * 16) : OOB_OFFSET;
            }
        }
        // Residual: the 16-byte pieces of two channel groups (32 columns) are requested together and awaited with a full
        // vmcnt(0) before they are used.  (A counted wait is not safe here: VGPR-destination loads and the LDS-DMA pieces
        // of the next K tiles do not retire in issue order relative to each other - consuming the data behind hipcc's
        // own counted wait gave rows with stale lanes.)
        uint4v rv[2][4];
        auto load_res2 = [&](int g0) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int on = on0 + ((g0 + gg) >> 1) * 32 + ((g0 + gg) & 1) * 16;
                const bool okc = on + fhi * 8 + 8 <= oN;
#pragma unroll
                for (int rbk = 0; rbk < 4; ++rbk)
                    rv[gg][rbk] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[rbk] : OOB_OFFSET, on * 2, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SB();
        };
#pragma unroll
        for (int g = 0; g < NIQ * 2; ++g) {
            const int iq = g >> 1, qp = g & 1;
            if (HAS_RES && (g & 1) == 0) load_res2(g);
            // bias (+ tile-uniform row bias), column sums of the 2 x 4 channels this lane owns in quarters q = 2qp, 2qp+1
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
            for (int rbk = 0; rbk < 4; ++rbk) {
                const int jq = rbk >> 1, j = rbk & 1;
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * qp + h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(ra[rbk], acc[iq][jq][j][4 * q + e], fmaf(rm[rbk], cs[h][e], bs[h][e]));
                        if (DBG == 5) x = bs[h][e];                       // debugging: park reads only
                        if (DBG == 6) x = acc[iq][jq][j][4 * q + e];      // debugging: raw accumulators
                        if (GEGLU) x *= gelu_erf_f(fmaf(ra[rbk], acc[1][jq][j][4 * q + e], fmaf(rm[rbk], gcs[h][e], gbs[h][e])));
                        v[h][e] = x;
                    }
                }
                if (HAS_RES) {  // un-swap the residual piece into the fragment layout, add in fp32
                    unsigned r0 = rv[g & 1][rbk][0], r1 = rv[g & 1][rbk][1], r2 = rv[g & 1][rbk][2], r3 = rv[g & 1][rbk][3];
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
        }
    };

    // ---- prologue: first K tile of the stream completely, then the head of the second in steady-state order ----
    int cbm0, cbn0;           // tile being computed
    int cv = blockIdx.x, cpb = 0;
    tile_origin(cv, cbm0, cbn0);
    set_stage_tile(cv);
    stage_park(0, cbm0, cbn0);
    stage_w(0, 0); stage_w(0, 1); stage_a(0, 0); stage_a(0, 1);
    advance();
    if (SCHED == 0) {
        if (cur.v < ntiles) { stage_w(1, 0); wait_vmcnt<2>(); } else wait_vmcnt<0>();
    } else {
        if (cur.v < ntiles) { stage_w(1, 0); stage_w(1, 1); wait_vmcnt<4>(); } else wait_vmcnt<0>();
    }
    BARRIER();                 // the first K tile has landed for every wave
    if (wm == 1) BARRIER();    // stagger: waves 4-7 run one barrier interval behind
    zero_acc();

    // One K tile = 4 phases.  B = ring buffer parity of the K tile being computed (wave-uniform); `first` = first K
    // tile of its output tile (park vectors are requested then).
    auto tile_step = [&](const int B, bool first, int nbm0, int nbn0, int npb) {
        // ---- phase 0: quadrant (a0, w0)
        read_w(B, 0, fw0);
        SB();
        read_a(B, 0);
        if (first) stage_park(npb, nbm0, nbn0);
        if (cur.v < ntiles) { if (SCHED == 0) stage_w(B ^ 1, 1); else stage_a(B ^ 1, 0); }
        BARRIER();
        mma(acc[0][0], fw0);
        BARRIER();
        // ---- phase 1: quadrant (a0, w1)
        read_w(B, 1, fw1);
        if (cur.v < ntiles) { if (SCHED == 0) stage_a(B ^ 1, 0); else { stage_a(B ^ 1, 1); advance(); } }
        if (SCHED == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // w1 reads retired before anyone passes the barrier
        BARRIER();
        mma(acc[1][0], fw1);
        BARRIER();
        // ---- phase 2: quadrant (a1, w1)
        read_a(B, 1);
        if (cur.v < ntiles) { if (SCHED == 0) { stage_a(B ^ 1, 1); advance(); } else stage_w(B, 0); }
        BARRIER();
        mma(acc[1][1], fw1);
        BARRIER();
        // ---- phase 3: quadrant (a1, w0); retire the next K tile of the stream (read from the next phase on)
        if (SCHED == 0) {
            if (cur.v < ntiles) { stage_w(B, 0); wait_vmcnt<2>(); } else wait_vmcnt<0>();
        } else {
            if (cur.v < ntiles) { stage_w(B, 1); wait_vmcnt<4>(); } else wait_vmcnt<0>();
        }
        BARRIER();
        mma(acc[0][1], fw0);
        BARRIER();
    };
    int par = 0;
    for (; cv < ntiles; cv += gridDim.x) {
        tile_origin(cv, cbm0, cbn0);
        for (int t = 0; t < nk; ++t) {
            // the first K tile of the very first tile had its park vectors requested by the prologue
            const bool first = t == 0 && cv != (int)blockIdx.x;
            tile_step(par, first, cbm0, cbn0, cpb);
            par ^= 1;
        }
        epilogue(cbm0, cbn0, cpb);
        zero_acc();
        cpb ^= 1;
    }
    if (wm == 0) BARRIER();    // pairs with the last barrier of the staggered half
}

template <int MODE, int SCHED, bool GEGLU, bool HAS_RES, int DBG = 0>
int launch_p8(const insv2v_gemm_desc& d, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_p8_kernel<MODE, SCHED, GEGLU, HAS_RES, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    hipLaunchKernelGGL((gemm_p8_kernel<MODE, SCHED, GEGLU, HAS_RES, DBG>), dim3(tiles < num_cu ? tiles : num_cu), dim3(512), LDS_B, s, d);
    return launch_status();
}

}  // namespace

// variant: 0 = SCHED 0 (restage two phases after the last read), 1 = SCHED 1 (one phase, deeper prefetch)
int insv2v_gemm_p8(const insv2v_gemm_desc& d, int variant, hipStream_t s) {
    if (d.batch > 1 || d.c_fp32 || d.split_k > 1) return INSV2V_EUNSUPPORTED;
    if ((d.K % BK) || (d.N & 7) || (d.ldc & 7) || ((uintptr_t)d.c & 15)) return INSV2V_EUNSUPPORTED;
    if (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15))) return INSV2V_EUNSUPPORTED;
    if (d.k_split && (d.k_split % BK)) return INSV2V_EUNSUPPORTED;
    if (d.act == INSV2V_ACT_GEGLU && (d.N % 64)) return INSV2V_EUNSUPPORTED;
    if (d.row_stats && (d.M & 1)) return INSV2V_EUNSUPPORTED;  // (mean, rstd) pairs are fetched two rows per lane
    // the row-bias vector is parked per tile: every 256-row tile must lie inside one group
    if (d.row_bias && ((d.ld_rb & 3) || (d.rows_per_group % 256 && d.M > d.rows_per_group))) return INSV2V_EUNSUPPORTED;
    if ((int64_t)d.M * d.ldc * 2 >= ((int64_t)1 << 31) || (d.residual && (int64_t)d.M * d.ldr * 2 >= ((int64_t)1 << 31))) return INSV2V_EUNSUPPORTED;
    const bool conv = d.mode == INSV2V_MODE_CONV3X3;
    if (conv && (d.Cin % BK)) return INSV2V_EUNSUPPORTED;
    const bool gg = d.act == INSV2V_ACT_GEGLU;
    if (!gg && d.act != INSV2V_ACT_NONE) return INSV2V_EUNSUPPORTED;  // SiLU / quick-GELU GEMMs are tiny (time embedding, CLIP)
    if (conv && gg) return INSV2V_EUNSUPPORTED;
    const bool res = d.residual != nullptr;
    if (gg && res) return INSV2V_EUNSUPPORTED;
    switch (variant) {
        case 0:
            if (conv) return res ? launch_p8<INSV2V_MODE_CONV3X3, 0, false, true>(d, s) : launch_p8<INSV2V_MODE_CONV3X3, 0, false, false>(d, s);
            if (gg) return launch_p8<INSV2V_MODE_LINEAR, 0, true, false>(d, s);
            return res ? launch_p8<INSV2V_MODE_LINEAR, 0, false, true>(d, s) : launch_p8<INSV2V_MODE_LINEAR, 0, false, false>(d, s);
        case 1:
            if (conv || gg) return INSV2V_EUNSUPPORTED;
            return res ? launch_p8<INSV2V_MODE_LINEAR, 1, false, true>(d, s) : launch_p8<INSV2V_MODE_LINEAR, 1, false, false>(d, s);
        case 5: return launch_p8<INSV2V_MODE_LINEAR, 0, false, false, 5>(d, s);
        case 6: return launch_p8<INSV2V_MODE_LINEAR, 0, false, false, 6>(d, s);
        case 3:  // timing ablation: no epilogue
            if (conv) return INSV2V_EUNSUPPORTED;
            return launch_p8<INSV2V_MODE_LINEAR, 0, false, false, 2>(d, s);
    }
    return INSV2V_EINVAL;
}
