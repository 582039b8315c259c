// gemm_q8: 256 x 256 fp16 MFMA GEMM / implicit-GEMM 3x3 convolution, 8-phase K loop with INTERLEAVED half-tile ownership.
//
// Round 4 successor of gemm_p8.hip (same tile, same LDS image, same epilogue semantics).  What round 3's kernel lacked
// against the guide's 256^2 template (cdna_hip_programming.md section 5: "3 half-tiles prefetched ahead", vmcnt(6)):
//
//  * In gemm_p8 a wave's 128 token rows are ONE A half-tile and its 64 channels lie in ONE W half-tile, so every half is read
//    in two phases (A: phases 0 and 2, W: phases 0 and 1), is dead only late, and the ring can keep just ONE half-tile in
//    flight behind the per-K-tile wait (vmcnt(2)): a half requested in phase 2 is awaited in phase 3 - the L2 / HBM latency
//    of every K tile is exposed (SQ_WAIT_ANY dominates, profiles/r03_final_pmc_pipe_utilisation.txt).
//    Here a wave owns token rows {wm*64 .. +63} of BOTH A halves and channel rows {wn*32 .. +31} of BOTH W halves:
//    quadrant (jq, iq) = A half jq x W half iq, every half-tile is consumed in exactly ONE phase
//        phase 0: W0 (4 ds_read_b128) + A0 (8)    -> MFMA (a0, w0)
//        phase 1: W1 (4)                           -> MFMA (a0, w1)
//        phase 2: A1 (8)                           -> MFMA (a1, w1)
//        phase 3: -                                -> MFMA (a1, w0)
//    and is restaged for the next-but-one K tile right behind its last read:
//        phase 1: W0(t+2)   (W0's reads are retired by an lgkmcnt(8) in front of phase 0's barrier: restage one phase later)
//        phase 2: A0(t+2)   phase 3: W1(t+2)   phase 0 of K tile t+1: A1(t+2)
//    One counted wait per K tile, in phase 3: vmcnt(6) - the three halves just requested stay in flight across the barrier,
//    everything older (= K tile t+1 complete) has landed and is read from the next phase on.  A half-tile now has 4 - 7 phases
//    between its request and its first read instead of 1 - 3.
//  * The two wave groups (wm = 0 / 1, one barrier interval apart) ran their EPILOGUES one after the other (no barrier inside an
//    epilogue, the next barrier pairs them again), so a tile paid two epilogues back to back with the matrix pipes idle - at
//    K = 320 as long as the K loop (profiles/r02_gemm_p8_ablation.txt).  Here the groups re-join before the epilogue (group 0
//    takes one extra barrier) and run it CONCURRENTLY, two waves per SIMD interleaving their VALU / store streams, and
//    the stagger is re-established at the next tile's first barrier.
//  * The read segments of a phase carry no address arithmetic: per-lane source offsets are kept per (tile, source, tap) and only
//    refreshed when one of those changes; ring slots and fragment offsets are compile-time (the K loop is unrolled by the two
//    ring buffers); there is no "stream finished" branch around the LDS-DMA requests (a finished stream requests out-of-range
//    offsets = zero fill, no memory traffic).
//
// Everything else (XCD-aware rasterisation, persistent grid, park area, LDS-free epilogue with v_permlane32_swap + 16-byte
// stores, hazard workarounds) is gemm_p8's; see that file for the reasoning.
#include "common.h"
#include "gemm_dma.h"
#include <type_traits>

namespace {

constexpr int HALF_B = 128 * 128;       // one half-tile: 128 rows x 64 halfs
constexpr int RING_B = 8 * HALF_B;      // slot (part*2 + b); part 0 A0, 1 A1, 2 W0, 3 W1
constexpr int PARK_B = 5 * 1024;        // bias[256], col_sum[256], (mean, rstd)[256], row_bias[256]
constexpr int LDS_B = RING_B + 2 * PARK_B;
constexpr int STAMP_B = 8 * 512;       // DBG 4 only: 32 (begin, end) stamp pairs per wave

#ifndef Q8_MID_SPREAD
#define Q8_MID_SPREAD 1   // compile-time A/B: LDS-DMA requests spread over the MFMA segment by a scheduler pipeline (1) / in two lumps (0)
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
#define BARRIER() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
// two-convert + pack (v_cvt_pkrtz rounds toward zero), pinned store data: tools/archive/gemm_p8.hip has the history
// two fp32 -> one packed fp16 pair, round to nearest even: gfx950's v_cvt_pk_f16_f32 (ONE instruction; rounds 1-5 spent two v_cvt_f16_f32 and a
// v_pack_b32_f16 here, because the only packed conversion of earlier parts, v_cvt_pkrtz, rounds toward zero)
__device__ __forceinline__ unsigned pack_h2(float x, float y) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2v));
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[0]; }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(half2v, u)[1]; }

template <int I> using ic = std::integral_constant<int, I>;

// DBG: 0 product; 2 no epilogue (timing ablation); 3 groups do NOT re-join for the epilogue (A/B of the concurrent epilogue);
//      4 = no epilogue + s_memtime stamps at the start and end of every MFMA segment of the first 32 phases of blocks 0-7, copied to
//      p.workspace as [block][wave][64] u64 (tools/gemm_check --stamps): interval lengths and who waits for whom at the barriers.
// VAR: 0 = LDS-DMA requests in the read segment of a phase (in front of its first barrier); 1 = inside the MFMA segment (behind the
//      2nd and 5th MFMA), the read segments carry ds_reads only
template <int MODE, bool GEGLU, bool HAS_RES, int DBG = 0, int VAR = 0>
__global__ __launch_bounds__(512) void gemm_q8_kernel(insv2v_gemm_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int G = (int)gridDim.x;

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

    // ---- staging side ----
    // a half-tile is 16 pieces of 1 KiB (8 rows x 128 B); wave `wid` fills pieces wid and wid+8:
    // LDS row (within the half) = i*64 + wid*8 + lane/8, chunk slot lane%8, source chunk = slot ^ ((row>>1)&7).
    // A half hh, LDS row r  <->  tile row hh*128 + r.
    // W half hh, LDS row r  <->  tile column hh*128 + r; GEGLU: r = pair*32 + i <-> column pair*64 + hh*32 + i, so that half 0
    // holds the h blocks and half 1 the gate blocks of the [h | g] interleaved projection and a wave finds both in its quadrants.
    const int prow = wid * 8 + (lane >> 3);
    const int chunk8 = ((lane & 7) ^ ((prow >> 1) & 7)) * 8;  // halfs
    int arow[4];            // linear: token row m (or -1); conv: first pixel index of the row's image (or -1)
    int aoh[4], aow[4];     // conv: output position * stride - pad
    unsigned aoff[4];       // byte offset of this lane's 16 bytes for the cursor's (tile, source, tap); OOB_OFFSET = zero fill
    unsigned woff[4];
    const int nk = p.K / BK;
    const int IHu = p.upsample ? p.IH * 2 : p.IH, IWu = p.upsample ? p.IW * 2 : p.IW;
    const int ups = p.upsample ? 1 : 0;
    struct Cursor { int v, kt, k0, kh, kw, ci0, soffA; bool second; unsigned wbase; } cur = {(int)blockIdx.x, 0, 0, 0, 0, 0, 0, false, 0u};  // wave-uniform
    auto refresh_aoff = [&]() {
        const int ld = (int)(cur.second ? p.lda2 : p.lda);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == INSV2V_MODE_LINEAR) {
                aoff[r] = arow[r] >= 0 ? (unsigned)((arow[r] * ld + chunk8) * 2) : OOB_OFFSET;
            } else {
                const int ih = aoh[r] + cur.kh, iw = aow[r] + cur.kw;
                const bool ok = arow[r] >= 0 && (unsigned)ih < (unsigned)IHu && (unsigned)iw < (unsigned)IWu;
                const int pix = arow[r] + (ih >> ups) * p.IW + (iw >> ups);
                aoff[r] = ok ? (unsigned)((pix * ld + chunk8) * 2) : OOB_OFFSET;
            }
        }
    };
    auto set_stage_rows = [&](int v) {
        int bm0 = 0, bn0 = 0;
        const bool live = v < ntiles;
        if (live) tile_origin(v, bm0, bn0);
        // grouped weights (the Winograd product): the tile's row group selects the weight matrix - a scalar byte offset of the W requests
        cur.wbase = (MODE == INSV2V_MODE_LINEAR && live && p.w_group_rows > 0) ? (unsigned)((int64_t)(bm0 / p.w_group_rows) * p.w_group_stride * 2) : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // r = half*2 + i
            const int hh = r >> 1, rr = (r & 1) * 64 + prow;
            const int m = bm0 + hh * 128 + rr;
            const bool okm = live && m < p.M;
            if (MODE == INSV2V_MODE_LINEAR) {
                arow[r] = okm ? m : -1;
                aoh[r] = aow[r] = 0;
            } else {
                const int mm = okm ? m : 0;
                const int ow = mm % p.OW, t = mm / p.OW;
                const int oh = t % p.OH, nb = t / p.OH;
                arow[r] = okm ? nb * p.IH * p.IW : -1;
                aoh[r] = oh * p.stride - p.pad_t;
                aow[r] = ow * p.stride - p.pad_l;
            }
            // round 6 (16x16x32 MFMAs): inside every 32-row block, LDS row cb * 16 + 4 g + r holds W row 8 g + 4 cb + r, so that the two MFMAs
            // of a block's channel halves (cb = 0, 1) leave lane group g = lane / 16 with the eight consecutive channels 8 g .. 8 g + 7
            const int r5 = rr & 31, pr = (rr & ~31) | (((r5 >> 2) & 3) * 8 + (r5 >> 4) * 4 + (r5 & 3));
            const int n = bn0 + (GEGLU ? (pr >> 5) * 64 + hh * 32 + (pr & 31) : hh * 128 + pr);
            woff[r] = (live && n < p.N) ? (unsigned)(((int64_t)n * p.ldw + chunk8) * 2) : OOB_OFFSET;
        }
    };
    // Next K tile of the stream.  Per-lane offsets are recomputed only on a tile / source / tap change, at ONE call site each and
    // without early returns: with the refresh inlined at several exits hipcc's structurizer treated the (wave-uniform) cursor
    // branches as divergent, moved the cursor into VGPRs and wrapped every LDS-DMA request in a waterfall loop.
    auto advance = [&]() {
        bool newtile = false, refresh = false;
        if (++cur.kt == nk) {
            cur.v += G; cur.kt = 0; cur.k0 = 0; cur.kh = cur.kw = cur.ci0 = 0; cur.soffA = 0; cur.second = false;
            newtile = true;
        } else {
            cur.k0 += BK; cur.soffA += BK * 2;
            if (MODE == INSV2V_MODE_LINEAR) {
                if (p.k_split > 0 && cur.k0 == p.k_split) { cur.second = true; cur.soffA = 0; refresh = true; }
            } else {
                cur.ci0 += BK;
                if (cur.ci0 >= p.Cin) {
                    cur.ci0 = 0; cur.soffA = 0;
                    if (++cur.kw == 3) { cur.kw = 0; ++cur.kh; }
                    cur.second = false; refresh = true;
                } else if (p.k_split > 0 && cur.ci0 == p.k_split) {
                    cur.second = true; cur.soffA = 0; refresh = true;
                }
            }
        }
        if (newtile) set_stage_rows(cur.v);
        if (newtile || refresh) refresh_aoff();
    };
    // stage half-tile PART (0 A0, 1 A1, 2 W0, 3 W1) of the cursor's K tile into ring buffer BUF: two 1 KiB pieces per wave
    auto stage_piece = [&](auto part_c, auto buf_c, int i) {
        constexpr int PART = decltype(part_c)::value, BUF = decltype(buf_c)::value;
        char* dst = smem + (PART * 2 + BUF) * HALF_B + wid * 1024 + i * 8192;
        if (PART >= 2) dma16(rW, woff[(PART - 2) * 2 + i], cur.k0 * 2 + (int)cur.wbase, dst);
        else dma16(cur.second ? rA2 : rA, aoff[PART * 2 + i], cur.soffA, dst);
    };
    auto stage = [&](auto part_c, auto buf_c) { stage_piece(part_c, buf_c, 0); stage_piece(part_c, buf_c, 1); };
    // Park area of tile parity pb: bias | col_sum | (mean, rstd) | tile-uniform row bias, in natural tile-local order.
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

    // ---- fragment addressing (bytes).  Round 6: v_mfma_f32_16x16x32_f16 (see gemm_r8.hip: 10 % less energy per FLOP than 32x32x16 in this
    // structure, which runs at the board's power cap).  A 16 x 32 fragment: lane l reads row (l & 15) of a 16-row block, 16-byte chunk
    // (kb * 4 + l / 16) ^ ((row >> 1) & 7); the block bases wm*64, tb*16, wn*32, cb*16 are multiples of 16, so the swizzle term only depends
    // on the lane.  One address register per k half and operand, everything else is an immediate offset.
    const int l15 = lane & 15, lq = lane >> 4, fsw = l15 >> 1;
    const char* aRd[2];
    const char* wRd[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int co = ((kb * 4 + lq) ^ fsw) * 16;
        aRd[kb] = smem + (wm * 64 + l15) * 128 + co;
        wRd[kb] = smem + 4 * HALF_B + (wn * 32 + l15) * 128 + co;
    }

    half8 fa[4][2], fw0[2][2], fw1[2][2];   // [token block of 16][k half], [channel half of the 32-channel block][k half]
    floatx4 acc[2][2][4][2];  // [iq (W half)][jq (A half)][token block][channel half]: 128 registers
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) acc[i][j][tb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    };
    auto read_a = [&](auto buf_c, auto jq_c) {
        constexpr int B = decltype(buf_c)::value, JQ = decltype(jq_c)::value;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) fa[tb][kb] = *(const half8*)(aRd[kb] + (JQ * 2 + B) * HALF_B + tb * 16 * 128);
    };
    auto read_w = [&](auto buf_c, auto iq_c, half8 (&fw)[2][2]) {
        constexpr int B = decltype(buf_c)::value, IQ = decltype(iq_c)::value;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) fw[cb][kb] = *(const half8*)(wRd[kb] + (IQ * 2 + B) * HALF_B + cb * 16 * 128);
    };
    // DBG 4: two s_memtime stamps per MFMA segment, parked in LDS behind the park area (consumed where lgkmcnt is 0 anyway)
    int stamp_i = 0;
    const unsigned stamp_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + LDS_B + wid * 512;
    // (the wait inside the statements keeps hipcc from copying a not-yet-returned SGPR pair; it delays the first MFMA until the
    //  segment's last fragment read has landed instead of its first - a DBG-build-only perturbation)
    auto stamp_begin = [&](unsigned long long& t0) {
        if (DBG == 4) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    };
    auto stamp_end = [&](unsigned long long t0) {
        if (DBG == 4) {
            unsigned long long t1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            if (stamp_i < 32) {
                const unsigned a = stamp_lds + stamp_i * 16;
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" ::"v"(a), "v"(t0), "v"(t1) : "memory");
            }
            ++stamp_i;
        }
    };
    // the 16 MFMAs of one quadrant (2 k halves x 4 token blocks x 2 channel halves); VAR 1: the two LDS-DMA pieces of half-tile PART ->
    // buffer BUF behind the 4th and the 10th MFMA
    auto mma = [&](floatx4 (&c)[4][2], const half8 (&fw)[2][2], auto part_c, auto buf_c) {
        unsigned long long t0 = 0;
        stamp_begin(t0);
        __builtin_amdgcn_s_setprio(1);
#if Q8_MID_SPREAD
        // (see gemm_r8.hip: a 16-cycle MFMA slot holds three other instructions, an LDS-DMA request is eight or nine - the phase's two requests
        // are spread over the segment by a scheduler pipeline instead of stalling the MFMAs behind them)
        if (VAR == 1) { stage_piece(part_c, buf_c, 0); stage_piece(part_c, buf_c, 1); }
#endif
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    c[tb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb][kb], fa[tb][kb], c[tb][cb], 0, 0, 0);
#if !Q8_MID_SPREAD
                    const int idx = kb * 8 + tb * 2 + cb;
                    if (VAR == 1 && (idx == 3 || idx == 9)) { SB(); stage_piece(part_c, buf_c, idx == 3 ? 0 : 1); SB(); }
#endif
                }
#if Q8_MID_SPREAD
        if (VAR == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        stamp_end(t0);
    };

    // ---- epilogue of the tile at (bm0, bn0), park buffer pb; no LDS ring access, no barriers.  Lane l = (token l & 15 of every 16-token
    // block, channel group g = l / 16): rows bm0 + jq*128 + wm*64 + tb*16 + (l & 15); channels iq*128 + wn*32 + 8 g .. + 7 (GEGLU: outputs
    // wn*32 + 8 g .. + 7, values in quadrants iq = 0, gates in iq = 1) - one 16-byte store per (block, token block), residual in the same layout ----
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const srd_t rC = make_srd(p.c), rR = make_srd(p.residual ? p.residual : p.c);
    auto park6 = [&](unsigned a, floatx4& b0, floatx4& b1, floatx4& r0, floatx4& r1, floatx4& c0, floatx4& c1) {
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:4096\n\t"
                     "ds_read_b128 %3, %6 offset:4112\n\tds_read_b128 %4, %6 offset:1024\n\tds_read_b128 %5, %6 offset:1040\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(b0), "=&v"(b1), "=&v"(r0), "=&v"(r1), "=&v"(c0), "=&v"(c1) : "v"(a) : "memory");
    };
    auto stat4 = [&](unsigned a, float2& s0, float2& s1, float2& s2, float2& s3) {  // the lane's token in the four 16-row blocks of one A half
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b64 %2, %4 offset:256\n\t"
                     "ds_read_b64 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a) : "memory");
    };
    auto epilogue = [&](int bm0, int bn0, int pb) {
        if (DBG == 2 || DBG == 4) {  // timing ablation: no epilogue; one dummy store keeps the accumulators live
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s += acc[i][j][tb][cb][r];
            if (s == 12345.678f) ((half_t*)p.c)[tid] = (half_t)s;
            return;
        }
        const unsigned park = lds0 + RING_B + pb * PARK_B;  // bias | +1024 col_sum | +2048 (mean, rstd) | +4096 row bias
        constexpr int NIQ = GEGLU ? 1 : 2;
        const int oN = GEGLU ? (p.N >> 1) : p.N;
        const int e15 = lane & 15, eg = lane >> 4;
        // tile-local first channel (park index) / first output column of the wave's 32-channel block of W half iq
        auto chan0 = [&](int iq) { return GEGLU ? wn * 64 : iq * 128 + wn * 32; };
        auto ocol0 = [&](int iq) { return GEGLU ? (bn0 >> 1) + wn * 32 : bn0 + chan0(iq); };
        constexpr bool LINM = MODE == INSV2V_MODE_LINEAR;   // (a convolution never carries a folded LayerNorm: (ra, rm) = (alpha, 0))
        float ra[LINM ? 8 : 1], rm[LINM ? 8 : 1];
        unsigned offc[8], offr[8];   // row block rbk = jq*4 + tb
        {
            if (!LINM) { ra[0] = p.alpha; rm[0] = 0.f; }
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                float2 st[4];
                if (LINM) stat4(park + 2048 + (jq * 128 + wm * 64 + e15) * 8, st[0], st[1], st[2], st[3]);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    const int rbk = jq * 4 + tb;
                    const int m = bm0 + jq * 128 + wm * 64 + tb * 16 + e15;
                    if (LINM) {
                        const float mean = ln ? st[tb].x : 0.f, rstd = ln ? st[tb].y : 1.f;
                        ra[rbk] = rstd * p.alpha; rm[rbk] = -rstd * mean;
                    }
                    offc[rbk] = m < p.M ? (unsigned)(m * (int)p.ldc * 2 + eg * 16) : OOB_OFFSET;
                    offr[rbk] = m < p.M ? (unsigned)(m * (int)p.ldr * 2 + eg * 16) : OOB_OFFSET;
                }
            }
        }
        uint4v rv[8];
        auto load_res = [&](int iq) {
            const int on = ocol0(iq);
            const bool okc = on + eg * 8 + 8 <= oN;
#pragma unroll
            for (int rbk = 0; rbk < 8; ++rbk) rv[rbk] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[rbk] : OOB_OFFSET, on * 2, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SB();
        };
#pragma unroll
        for (int iq = 0; iq < NIQ; ++iq) {
            if (HAS_RES) load_res(iq);
            float bs[8], cs[8], gbs[8], gcs[8];
            {
                floatx4 tb_[2], tr[2], tc[2], gb[2], gr[2], gc[2];
                const unsigned a = park + (chan0(iq) + 8 * eg) * 4;
                park6(a, tb_[0], tb_[1], tr[0], tr[1], tc[0], tc[1]);
                if (GEGLU) park6(a + 128, gb[0], gb[1], gr[0], gr[1], gc[0], gc[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bs[4 * h + e] = tb_[h][e] + tr[h][e]; cs[4 * h + e] = tc[h][e];
                        if (GEGLU) { gbs[4 * h + e] = gb[h][e] + gr[h][e]; gcs[4 * h + e] = gc[h][e]; }
                    }
            }
            const int on = ocol0(iq);
            const bool okc = on + eg * 8 + 8 <= oN;
#pragma unroll
            for (int rbk = 0; rbk < 8; ++rbk) {
                const int jq = rbk >> 2, tb = rbk & 3, ri = LINM ? rbk : 0;
                float v[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(ra[ri], acc[iq][jq][tb][h][e], fmaf(rm[ri], cs[4 * h + e], bs[4 * h + e]));
                        if (GEGLU) x *= gelu_erf_f(fmaf(ra[ri], acc[1][jq][tb][h][e], fmaf(rm[ri], gcs[4 * h + e], gbs[4 * h + e])));
                        v[4 * h + e] = x;
                    }
                if (HAS_RES) {
                    const uint4v r = rv[rbk];
                    v[0] += h_lo(r[0]); v[1] += h_hi(r[0]); v[2] += h_lo(r[1]); v[3] += h_hi(r[1]);
                    v[4] += h_lo(r[2]); v[5] += h_hi(r[2]); v[6] += h_lo(r[3]); v[7] += h_hi(r[3]);
                }
                const uint4v out = {pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
                __builtin_amdgcn_raw_buffer_store_b128(out, rC, okc ? offc[rbk] : OOB_OFFSET, on * 2, 0);
                asm volatile("s_nop 7" ::"v"(out));  // 16-byte store data pinned (2-waves-per-SIMD store hazard, profiles/r02_gemm_debug.md)
            }
        }
    };

    // ---- prologue: K tile 0 completely, then W0, A0, W1 of K tile 1 (steady-state order; its A1 follows in phase 0) ----
    int cbm0, cbn0;
    int cv = blockIdx.x, cpb = 0;
    tile_origin(cv, cbm0, cbn0);
    set_stage_rows(cv);
    refresh_aoff();
    stage_park(0, cbm0, cbn0);
    stage(ic<2>{}, ic<0>{}); stage(ic<0>{}, ic<0>{}); stage(ic<3>{}, ic<0>{}); stage(ic<1>{}, ic<0>{});
    advance();
    stage(ic<2>{}, ic<1>{}); stage(ic<0>{}, ic<1>{}); stage(ic<3>{}, ic<1>{});
    wait_vmcnt<6>();
    BARRIER();                 // K tile 0 has landed for every wave
    zero_acc();

    // One K tile = 4 phases on ring buffer B (compile-time); `first` = first K tile of its output tile.
    auto tile_step = [&](auto buf_c, bool first, int nbm0, int nbn0, int npb) {
        constexpr int B = decltype(buf_c)::value;
        // ---- phase 0: quadrant (a0, w0); request A1 of the next K tile (last half of it), move the cursor on
        read_w(ic<B>{}, ic<0>{}, fw0);
        SB();
        read_a(ic<B>{}, ic<0>{});
        SB();
        if (first) stage_park(npb, nbm0, nbn0);
        if (VAR == 0) { stage(ic<1>{}, ic<B ^ 1>{}); advance(); }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the W0 reads (issued first) are retired: W0 is restaged next phase
        BARRIER();
        mma(acc[0][0], fw0, ic<1>{}, ic<B ^ 1>{});
        BARRIER();
        // ---- phase 1: quadrant (a0, w1)
        read_w(ic<B>{}, ic<1>{}, fw1);
        if (VAR == 0) stage(ic<2>{}, ic<B>{}); else advance();
        BARRIER();
        mma(acc[1][0], fw1, ic<2>{}, ic<B>{});
        BARRIER();
        // ---- phase 2: quadrant (a1, w1)
        read_a(ic<B>{}, ic<1>{});
        if (VAR == 0) stage(ic<0>{}, ic<B>{});
        BARRIER();
        mma(acc[1][1], fw1, ic<0>{}, ic<B>{});
        BARRIER();
        // ---- phase 3: quadrant (a1, w0); the next K tile is complete behind the halves requested last (VAR 0: three, VAR 1: two)
        if (VAR == 0) { stage(ic<3>{}, ic<B>{}); wait_vmcnt<6>(); } else wait_vmcnt<4>();
        BARRIER();
        mma(acc[0][1], fw0, ic<3>{}, ic<B>{});
        BARRIER();
    };
    int par = 0;
    for (; cv < ntiles; cv += G) {
        tile_origin(cv, cbm0, cbn0);
        if (wm == 1) BARRIER();    // stagger: waves 4-7 run one barrier interval behind
        bool first = cv != (int)blockIdx.x;  // the very first tile's park vectors were requested by the prologue
        int t = 0;
        if (par) { tile_step(ic<1>{}, first, cbm0, cbn0, cpb); first = false; t = 1; par = 0; }
        for (; t + 1 < nk; t += 2) {
            tile_step(ic<0>{}, first, cbm0, cbn0, cpb);
            first = false;
            tile_step(ic<1>{}, false, cbm0, cbn0, cpb);
        }
        if (t < nk) { tile_step(ic<0>{}, first, cbm0, cbn0, cpb); par = 1; }
        if (DBG != 3) { if (wm == 0) BARRIER(); }   // re-join: both groups run the epilogue concurrently
        epilogue(cbm0, cbn0, cpb);
        if (DBG == 3) { if (wm == 0) BARRIER(); }
        zero_acc();
        cpb ^= 1;
    }
    if (DBG == 4 && blockIdx.x < 8 && p.workspace) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long v = *(const unsigned long long*)(smem + LDS_B + wid * 512 + lane * 8);
        ((unsigned long long*)p.workspace)[((int)blockIdx.x * 8 + wid) * 64 + lane] = v;
    }
}

template <int MODE, bool GEGLU, bool HAS_RES, int DBG = 0, int VAR = 0>
int launch_q8(const insv2v_gemm_desc& d, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_q8_kernel<MODE, GEGLU, HAS_RES, DBG, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B + STAMP_B);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    hipLaunchKernelGGL((gemm_q8_kernel<MODE, GEGLU, HAS_RES, DBG, VAR>), dim3(tiles < num_cu ? tiles : num_cu), dim3(512), LDS_B + (DBG == 4 ? STAMP_B : 0), s, d);
    return launch_status();
}

}  // namespace

// variant: 0 = product (LDS-DMA requests inside the MFMA segments); 1 = requests in the load segments (round-4 A/B: 5-18 % slower);
// 2 = no epilogue (timing); 3 = the two wave groups' epilogues one after the other (A/B); 4 / 5 = s_memtime stamps of schedule 1 / 0
int insv2v_gemm_q8(const insv2v_gemm_desc& d, int variant, hipStream_t s) {
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
    if (!gg && d.act != INSV2V_ACT_NONE) return INSV2V_EUNSUPPORTED;
    if (conv && gg) return INSV2V_EUNSUPPORTED;
    const bool res = d.residual != nullptr;
    if (gg && res) return INSV2V_EUNSUPPORTED;
    constexpr int L = INSV2V_MODE_LINEAR, C = INSV2V_MODE_CONV3X3;
    switch (variant) {
        case 0:
            if (conv) return res ? launch_q8<C, false, true, 0, 1>(d, s) : launch_q8<C, false, false, 0, 1>(d, s);
            if (gg) return launch_q8<L, true, false, 0, 1>(d, s);
            return res ? launch_q8<L, false, true, 0, 1>(d, s) : launch_q8<L, false, false, 0, 1>(d, s);
        case 1:
            if (conv) return res ? launch_q8<C, false, true, 0, 0>(d, s) : launch_q8<C, false, false, 0, 0>(d, s);
            if (gg) return launch_q8<L, true, false, 0, 0>(d, s);
            return res ? launch_q8<L, false, true, 0, 0>(d, s) : launch_q8<L, false, false, 0, 0>(d, s);
        case 2:
            if (conv || gg || res) return INSV2V_EUNSUPPORTED;
            return launch_q8<L, false, false, 2, 1>(d, s);
        case 3:
            if (conv) return INSV2V_EUNSUPPORTED;
            if (gg) return launch_q8<L, true, false, 3, 1>(d, s);
            return res ? launch_q8<L, false, true, 3, 1>(d, s) : launch_q8<L, false, false, 3, 1>(d, s);
        case 4:
            if (conv || gg || res) return INSV2V_EUNSUPPORTED;
            return launch_q8<L, false, false, 4, 1>(d, s);
        case 5:
            if (conv || gg || res) return INSV2V_EUNSUPPORTED;
            return launch_q8<L, false, false, 4, 0>(d, s);
    }
    return INSV2V_EINVAL;
}
