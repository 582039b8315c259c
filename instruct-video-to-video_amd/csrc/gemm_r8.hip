// gemm_r8: 256 x 320 fp16 MFMA GEMM / implicit-GEMM 3x3 convolution on the round-4 8-wave ping-pong engine (gemm_q8.hip).
//
// Why a 320-wide tile: every output width of the UNet is a multiple of 320 (320, 640, 960, 1280, 1920, 2560, 3840, 5120,
// 10240), but N = 320 / 640 / 960 / 1920 are 1.25 / 2.5 / 3.75 / 7.5 tiles of 256, which is why round 3 kept the level-0/1
// convolutions (22.7 % of the kernel time) and the N = C linears on 128-wide tiles (2.5 tiles for N = 320: 83 % tile efficiency,
// and the input patch re-read per column tile - VERDICT r3 item 3).  Here one workgroup owns 256 tokens x 320 channels: N = 320 is
// ONE column tile, the activations are read once per workgroup.
//
// Structure (differences to gemm_q8 only):
//  * 8 waves as 4 (token rows, wm) x 2 (channel halves, wn); a wave owns 64 tokens x 160 channels = 2 x 5 fragments of 32 x 32:
//    160 accumulator registers.  Group 0 = waves 0-3 = tokens 0-127, group 1 = tokens 128-255, one barrier interval behind.
//  * A K tile (64 wide) = A0, A1 (128 token rows each, 16 KiB; A_h is read by group h only) + W0 .. W4 (W block p = the p-th
//    32-channel fragment of BOTH channel halves: 64 rows, 8 KiB).  FIVE phases per K tile: phase p reads W_p (4 ds_read_b128;
//    phase 0 also the wave's 8 A fragments, which stay in registers for the five phases) and issues 8 MFMAs (fragment p x 2 token
//    blocks x K = 64).  Every unit is consumed in exactly one phase and restaged for the next-but-one K tile two or more phases later:
//        phase 0: W2, W3 of K tile t+1      phase 1: W4 (t+1), then the stream cursor moves on
//        phase 2: A0 (t+2)                  phase 3: A1 (t+2)                 phase 4: W0, W1 (t+2)
//    all INSIDE the MFMA segments (behind the 2nd and 5th MFMA; tools/phase_rate.hip: an LDS-DMA request next to ds_reads in a load
//    segment costs ~100 cycles per interval, behind an MFMA ~17).  One counted wait per K tile: vmcnt(4) in phase 4's load segment
//    (A0, A1 of t+2 stay in flight, K tile t+1 is complete).  9 requests per wave per K tile; 7+ phases between request and first read.
//  * Ring: [A0 b0 | A0 b1 | A1 b0 | A1 b1 | W0 b0 | W0 b1 | ... | W4 b1] = 144 KiB, so every fragment read is base register +
//    immediate; park area (bias, column sums, token statistics, row bias: 2 x 8 KiB) behind it: all 160 KiB of the CU.
//  * LINEAR mode addresses a half / block with ONE per-lane offset register per operand: a request adds its row-block offset and clamps
//    to the operand's last row (the scalar offset of a buffer load is outside the hardware's range check).  CONV3X3 keeps one offset per
//    piece (per-pixel padding).
//  * No GEGLU (those widths are multiples of 256: gemm_q8).
#include "common.h"
#include "gemm_dma.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int BM = 256, BN = 320;
constexpr int AH_B = 128 * 128;          // A half: 128 rows x 128 B
constexpr int WB_B = 64 * 128;           // W block: 64 rows x 128 B
constexpr int W_BASE = 4 * AH_B;         // W region behind [A0 b0 | A0 b1 | A1 b0 | A1 b1]
constexpr int RING_B = 4 * AH_B + 10 * WB_B;   // 147 456
// every parked vector owns a 2 KiB slot: its second LDS-DMA piece (entries 256 .. 319) zero-fills a whole KiB
constexpr int PK_BIAS = 0, PK_CS = 2048, PK_ST = 4096, PK_RB = 6144, PARK_B = 8192;
constexpr int LDS_B = RING_B + 2 * PARK_B;     // 163 840 = all of the CU's 160 KiB

#ifndef R8_MID_SPREAD
#define R8_MID_SPREAD 1   // compile-time A/B: LDS-DMA requests spread over the MFMA segment by a scheduler pipeline (1) / in two lumps (0)
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
#define BARRIER() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
// two-convert + pack (the compiler's v_cvt_pkrtz rounds toward zero), pinned store data: tools/archive/gemm_p8.hip has the history
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

// (Round 4, negative: a start-up stagger of the workgroups - ((b / 8) % 32) / 32 x 40 000 ... 160 000 cycles - to keep the CUs' HBM-bound
//  epilogues apart changed nothing or lost up to 5 %: profiles/r04_gemm_r8_epilogue_cycles.txt.  Code removed.)
// DBG: 0 product; 2 no epilogue (timing ablation); 4 product + per-wave cycle totals (s_memtime) of the K loops, the re-join barrier and the
//      epilogues, written to p.workspace as [block][wave][4] u64 = (K loops, re-join wait, epilogues, tiles) - tools/gemm_check --stamps
// GEGLU (LINEAR, no residual): out[m][o] = h * gelu(g) with W rows stored in [32 value rows | 32 gate rows] blocks (unet.interleave32, the
// layout the 256x256 kernels use).  A wave owns FIVE 32-row fragments, so value and gate cannot sit in different fragments as in gemm_q8:
// the DMA's source-row map puts 16 value rows and their 16 gate rows into EVERY fragment - fragment t = wn * 5 + pb of the tile holds
// outputs t * 16 .. + 15: LDS row c < 16 <-> W row (t >> 1) * 64 + (t & 1) * 16 + c, row c >= 16 <-> the same + 32.  In the accumulator
// layout a lane's register quarters 0, 1 are then values and quarters 2, 3 the gates of the SAME eight outputs: in-lane product, one
// 16-byte store per fragment and row block (half the stores of the plain form).
// SPLIT: the A operand has two sources (channel concat, k_split > 0); without it the source descriptor and row stride are loop constants
// (four s_cselect per LDS-DMA request less between the MFMAs)
// CIM (CONV3X3, stride 1, pad 1, no upsample): K tiles in CHANNEL-BLOCK-MAJOR / tap-minor order.  Tap-major order re-reads a tile's 256
// activation rows once per tap with FIVE K tiles (Cin = 320; more for wider inputs) of every CU of the XCD in between - 32 CUs x 5 x 72 KiB
// = 11 MB through a 4 MiB L2 - so every tap's rows came from the fabric again (round 4: 2.4x the algorithmic bytes over the GEMM family).
// Here the nine taps of ONE 64-channel block follow each other: a tile's 256 (+ halo) cache lines of that block are fetched once and hit
// in L2 for the other eight taps.  A lane keeps the centre-tap PIXEL index of its four token rows and their 9 validity bits per tile
// (one refresh per tile instead of one per tap); a request forms its offset as (pixel + tap shift) x row bytes.
template <int MODE, bool HAS_RES, int DBG = 0, bool SPLIT = true, bool STATS = false, bool GEGLU = false, bool CIM = false>
__global__ __launch_bounds__(512) void gemm_r8_kernel(insv2v_gemm_desc p, int dephase) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1, grp = wid >> 2;
    const int G = (int)gridDim.x;
    constexpr bool LIN = MODE == INSV2V_MODE_LINEAR;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    // De-phasing by SLACK (round 5).  Equal tiles on a persistent grid finish together, so the epilogues of all CUs hit HBM at once with
    // the matrix pipes idle (profiles/r04_gemm_r8_epilogue_cycles.txt).  Round 4's start-up stagger of EVERY workgroup spread them but
    // pushed the end of the launch out by the stagger.  When the tile count is not a multiple of the grid, the workgroups that run one
    // tile less have a whole tile time to spare: only THEY start late, spread evenly over `dephase` permille of a tile time - their
    // epilogues fall into the others' K loops, the launch ends when it would have ended anyway.
    if (dephase > 0 && ntiles > G) {
        const int n0 = ntiles % G;   // workgroups [0, n0) run one tile more than the rest
        if (n0 != 0 && (int)blockIdx.x >= n0) {
            const float tile_cycles = (float)(p.K / BK) * 3700.f + (HAS_RES ? 30000.f : 10000.f);   // (s_memtime units, r04 stamps)
            const float frac = (float)((int)blockIdx.x - n0 + 1) / (float)(G - n0 + 1) * (float)dephase * 0.001f;
            const unsigned long long wait = (unsigned long long)(frac * tile_cycles);
            unsigned long long t0, t1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            do {
                __builtin_amdgcn_s_sleep(64);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            } while (t1 - t0 < wait);
        }
    }
    auto tile_origin = [&](int v, int& bm0, int& bn0) {
        const int bid = xcd_remap(v, ntiles);
        constexpr int GROUP_M = 8;
        const int per_group = GROUP_M * tiles_n;
        const int gidx = bid / per_group, first_m = gidx * GROUP_M;
        const int gsz = min(GROUP_M, tiles_m - first_m), rin = bid - gidx * per_group;
        const int tn = rin / gsz, tm = first_m + rin - tn * gsz;
        bm0 = tm * BM; bn0 = tn * BN;
    };

    // Rows beyond M / N: the scalar offset of a buffer load is NOT part of the hardware's range check, so a row-block offset carried there
    // cannot be cut off by the descriptor's size.  The per-lane offset of every request is clamped to the operand's LAST row instead (one
    // v_add + one v_min per request): rows beyond the operand read valid memory (a copy of the last row) and their results are masked at
    // the store.  CONV3X3 marks padding / rows beyond M per piece with an out-of-range offset (zero fill).
    const srd_t rA = make_srd(p.a), rA2 = make_srd(p.a2 ? p.a2 : p.a), rW = make_srd(p.w);
    const bool ln = p.row_stats != nullptr;

    // ---- staging side ----
    // a piece = 8 rows x 128 B (1 KiB): lane -> row lane/8, chunk slot lane%8, source chunk = slot ^ ((row>>1)&7).
    // A half h, piece (i, wid): LDS rows i*64 + wid*8 + lane/8  <->  tile row h*128 + i*64 + wid*8 + lane/8.
    // W block pb, piece wid:    LDS rows wid*8 + lane/8         <->  tile column (wid>>2)*160 + pb*32 + (wid&3)*8 + lane/8.
    const int prow = wid * 8 + (lane >> 3);
    const int chunk8 = ((lane & 7) ^ ((prow >> 1) & 7)) * 8;  // halfs
    unsigned aoff[4];       // conv: per-piece byte offsets (CIM: centre-tap pixel indices); linear: only aoff[0] = this lane's offset in the tile's first piece
    unsigned amask[2] = {0, 0};   // CIM: tap validity of the four pieces' rows, 9 bits each (pieces 0, 1 | 2, 3)
    unsigned woff;          // this lane's byte offset in W block 0's piece
    unsigned amax = 0;      // linear: this lane's offset in the LAST row of the current A source (clamp)
    const unsigned wmax = (unsigned)(((int64_t)(p.N - 1) * p.ldw + chunk8) * 2);   // ... in the last row of W
    const int nk = p.K / BK;
    const int IHu = p.upsample ? p.IH * 2 : p.IH, IWu = p.upsample ? p.IW * 2 : p.IW;
    const int ups = p.upsample ? 1 : 0;
    struct Cursor { int v, kt, k0, kh, kw, ci0, soffA, abm0; bool second; int tap, dpix; unsigned wbase; } cur = {(int)blockIdx.x, 0, 0, 0, 0, 0, 0, 0, false, 0, 0, 0u};  // wave-uniform
    if (CIM) cur.dpix = -p.IW - 1;
    auto refresh_aoff = [&]() {
        const int ld = (int)((SPLIT && cur.second) ? p.lda2 : p.lda);
        if (LIN) {
            aoff[0] = ((unsigned)(cur.abm0 + prow) * (unsigned)ld + (unsigned)chunk8) * 2u;
            amax = ((unsigned)(p.M - 1) * (unsigned)ld + (unsigned)chunk8) * 2u;
        } else {
            // (output row, column, image) of the lane's four token rows are recomputed here - once per tap / source / tile change - instead
            // of being kept in eight registers across the K loop and the epilogue; exact: m < 2^24, so m / OW in fp32 is off by at most one
            const float rOW = 1.0f / (float)p.OW, rOH = 1.0f / (float)p.OH;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = cur.abm0 + (r >> 1) * 128 + (r & 1) * 64 + prow;
                const bool okm = m < p.M;
                const int mm = okm ? m : 0;
                int t = (int)((float)mm * rOW);
                int ow = mm - t * p.OW;
                if (ow < 0) { ow += p.OW; --t; } else if (ow >= p.OW) { ow -= p.OW; ++t; }
                int nb = (int)((float)t * rOH);
                int oh = t - nb * p.OH;
                if (oh < 0) { oh += p.OH; --nb; } else if (oh >= p.OH) { oh -= p.OH; ++nb; }
                if (CIM) {   // stride 1, pad 1: output pixel = centre-tap input pixel; tap (kh, kw) is valid iff (oh + kh - 1, ow + kw - 1) is inside
                    aoff[r] = (unsigned)(nb * p.IH * p.IW + oh * p.IW + ow);
                    const unsigned vh = (okm && oh > 0 ? 1u : 0u) | (okm ? 2u : 0u) | (okm && oh + 1 < p.IH ? 4u : 0u);
                    const unsigned vw = (ow > 0 ? 1u : 0u) | 2u | (ow + 1 < p.IW ? 4u : 0u);
                    const unsigned bits = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << 3 : 0u) | ((vh & 4u) ? vw << 6 : 0u);
                    if (r & 1) amask[r >> 1] |= bits << 9; else amask[r >> 1] = bits;
                } else {
                    const int ih = oh * p.stride - p.pad_t + cur.kh, iw = ow * p.stride - p.pad_l + cur.kw;
                    const bool ok = okm && (unsigned)ih < (unsigned)IHu && (unsigned)iw < (unsigned)IWu;
                    const int pix = nb * p.IH * p.IW + (ih >> ups) * p.IW + (iw >> ups);
                    aoff[r] = ok ? (unsigned)((pix * ld + chunk8) * 2) : OOB_OFFSET;
                }
            }
        }
    };
    auto set_stage_rows = [&](int v) {
        int bm0 = 0, bn0 = 0;
        const bool live = v < ntiles;
        if (live) tile_origin(v, bm0, bn0);
        else { bm0 = p.M + BM; bn0 = p.N + BN; }   // a finished stream requests rows beyond the operands: zero fill, no traffic
        cur.abm0 = bm0;
        // grouped weights (the Winograd product): the tile's row group selects the weight matrix - a scalar byte offset of the W requests
        cur.wbase = (LIN && live && p.w_group_rows > 0) ? (unsigned)((int64_t)(bm0 / p.w_group_rows) * p.w_group_stride * 2) : 0u;
        const int c32 = (wid & 3) * 8 + (lane >> 3);   // row of the 32-row W block
        // plain: LDS row cb * 16 + 4 g + r  <->  W row 8 g + 4 cb + r of the 32-channel fragment: the two 16 x 16 MFMAs of a fragment (cb = 0, 1)
        // leave a lane (g = lane / 16) with the EIGHT consecutive channels 8 g .. 8 g + 7 of its token - one 16-byte store, no cross-lane step
        const int pc = ((c32 >> 2) & 3) * 8 + (c32 >> 4) * 4 + (c32 & 3);
        const int n = GEGLU ? bn0 + (c32 >> 4) * 32 + (c32 & 15) : bn0 + (wid >> 2) * 160 + pc;
        woff = ((unsigned)n * (unsigned)p.ldw + (unsigned)chunk8) * 2u;   // (n <= N + 640: no 32-bit wrap, N * ldw * 2 < 2^31)
    };
    auto advance = [&]() {   // one call site per refresh, no early return (see gemm_q8.hip)
        bool newtile = false, refresh = false;
        if (++cur.kt == nk) {
            cur.v += G; cur.kt = 0; cur.k0 = 0; cur.kh = cur.kw = cur.ci0 = 0; cur.soffA = 0; cur.second = false;
            if (CIM) { cur.tap = 0; cur.dpix = -p.IW - 1; }
            newtile = true;
        } else if (CIM) {   // next tap of the channel block, or tap 0 of the next block; nothing per lane changes
            if (++cur.tap == 9) { cur.tap = 0; cur.ci0 += BK; }
            const int kh = (cur.tap * 11) >> 5, kw = cur.tap - 3 * kh;
            cur.dpix = (kh - 1) * p.IW + kw - 1;
            cur.second = SPLIT && p.k_split > 0 && cur.ci0 >= p.k_split;
            cur.soffA = (cur.ci0 - (cur.second ? p.k_split : 0)) * 2;
            cur.k0 = cur.tap * p.Cin + cur.ci0;
        } else {
            cur.k0 += BK; cur.soffA += BK * 2;
            if (LIN) {
                if (SPLIT && p.k_split > 0 && cur.k0 == p.k_split) { cur.second = true; cur.soffA = 0; refresh = true; }
            } else {
                cur.ci0 += BK;
                if (cur.ci0 >= p.Cin) {
                    cur.ci0 = 0; cur.soffA = 0;
                    if (++cur.kw == 3) { cur.kw = 0; ++cur.kh; }
                    cur.second = false; refresh = true;
                } else if (SPLIT && p.k_split > 0 && cur.ci0 == p.k_split) {
                    cur.second = true; cur.soffA = 0; refresh = true;
                }
            }
        }
        if (newtile) set_stage_rows(cur.v);
        if (newtile || refresh) refresh_aoff();
    };
    // piece i (0 / 1) of A half H of the cursor's K tile -> ring buffer BUF
    auto stage_a = [&](auto h_c, auto buf_c, int i) {
        constexpr int H = decltype(h_c)::value, BUF = decltype(buf_c)::value;
        char* dst = smem + (H * 2 + BUF) * AH_B + i * 8192 + wid * 1024;
        if (LIN) {
            const int ld = (int)((SPLIT && cur.second) ? p.lda2 : p.lda);
            dma16((SPLIT && cur.second) ? rA2 : rA, min(aoff[0] + (unsigned)((H * 128 + i * 64) * ld * 2), amax), cur.soffA, dst);
        } else if (CIM) {
            constexpr int R = H * 2;   // pieces (H, 0), (H, 1) = rows r = 2H, 2H + 1: bits 0-8 and 9-17 of amask[H]
            const unsigned ld2 = (unsigned)((SPLIT && cur.second) ? p.lda2 : p.lda) * 2u;
            const bool ok = (amask[H] >> (i * 9 + cur.tap)) & 1u;
            const unsigned off = __umul24(aoff[R + i] + (unsigned)cur.dpix, ld2) + (unsigned)(chunk8 * 2);   // pixel < 2^24, row bytes < 2^24, product < 2^31
            dma16((SPLIT && cur.second) ? rA2 : rA, ok ? off : OOB_OFFSET, cur.soffA, dst);
        } else {
            dma16((SPLIT && cur.second) ? rA2 : rA, aoff[H * 2 + i], cur.soffA, dst);
        }
    };
    // this wave's piece of W block PB of the cursor's K tile -> ring buffer BUF
    auto stage_w = [&](auto pb_c, auto buf_c) {
        constexpr int PB = decltype(pb_c)::value, BUF = decltype(buf_c)::value;
        char* dst = smem + W_BASE + (PB * 2 + BUF) * WB_B + wid * 1024;
        if (GEGLU) {
            const int t = (wid >> 2) * 5 + PB;   // fragment of the tile (wave-uniform)
            dma16(rW, min(woff + (unsigned)(((t >> 1) * 64 + (t & 1) * 16) * (int)p.ldw * 2), wmax), cur.k0 * 2 + (int)cur.wbase, dst);
        } else {
            dma16(rW, min(woff + (unsigned)(PB * 32 * (int)p.ldw * 2), wmax), cur.k0 * 2 + (int)cur.wbase, dst);
        }
    };
    // Park area of tile parity pb: bias[320] | col_sum[320] | (mean, rstd)[256] | tile-uniform row bias[320]; one LDS-DMA piece per wave
    auto row_group = [&](int m) { int g = m / p.rows_per_group; if (p.rb_mod > 0) g %= p.rb_mod; return g; };
    auto stage_park = [&](int pb, int bm0, int bn0) {
        char* park = smem + RING_B + pb * PARK_B;
        const srd_t rBias = make_srd(p.bias ? (const void*)p.bias : p.w), rCs = make_srd(ln ? (const void*)p.col_sum : p.w),
                    rSt = make_srd(ln ? (const void*)p.row_stats : p.w), rRb = make_srd(p.row_bias ? (const void*)p.row_bias : p.w);
        const int hi = wid & 1;                         // second piece of a 320-entry vector: entries 256 .. 319 (16 lanes)
        const int n = bn0 + hi * 256 + lane * 4;
        const bool okn = n < p.N && (hi == 0 || lane < 16);
        if (wid < 2) {
            dma16(rBias, (p.bias && okn) ? (unsigned)(n * 4) : OOB_OFFSET, 0, park + PK_BIAS + hi * 1024);
        } else if (wid < 4) {
            dma16(rCs, (ln && okn) ? (unsigned)(n * 4) : OOB_OFFSET, 0, park + PK_CS + hi * 1024);
        } else if (wid < 6) {
            const int m = bm0 + hi * 128 + lane * 2;
            dma16(rSt, (ln && m < p.M) ? (unsigned)(m * 8) : OOB_OFFSET, 0, park + PK_ST + hi * 1024);
        } else {
            const int g = p.row_bias ? row_group(bm0) : 0;  // every row of the tile is in this group (checked on the host)
            dma16(rRb, (p.row_bias && okn) ? (unsigned)((g * (int)p.ld_rb + n) * 4) : OOB_OFFSET, 0, park + PK_RB + hi * 1024);
        }
    };

    // ---- fragment addressing (bytes).  Round 6: v_mfma_f32_16x16x32_f16 instead of 32x32x16 - inside one and the same 8-phase structure
    // the 32x32x16 shape draws 10 % more energy per FLOP (0.88 vs 0.80 pJ) and the board, which runs this engine at its power cap, clocks
    // 1.55 instead of 1.78 GHz under it: 1 287 vs 1 414 TFLOP/s at 8192^3 on one box (profiles/r06_mfma_shape_ab.txt).
    // A 16 x 32 fragment: lane l reads row (l & 15) of a 16-row block, 16-byte chunk (kb * 4 + l / 16) ^ ((row >> 1) & 7) - conflict-free on
    // the unchanged LDS image (block bases are multiples of 16 rows, so the swizzle term is (l & 15) >> 1).
    const int l15 = lane & 15, lq = lane >> 4, fsw = l15 >> 1;
    const char* aRd[2];     // A half of this wave's group, buffer 0, token block 0, k half kb
    const char* wRd[2];     // W block 0, buffer 0, channel half 0
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int co = ((kb * 4 + lq) ^ fsw) * 16;
        aRd[kb] = smem + grp * 2 * AH_B + ((wm & 1) * 64 + l15) * 128 + co;
        wRd[kb] = smem + W_BASE + (wn * 32 + l15) * 128 + co;
    }

    half8 fa[4][2], fw[2][2];   // [token block of 16][k half], [channel half of the fragment][k half]
    floatx4 acc[5][4][2];       // [p (32-channel fragment)][token block][channel half]: 160 registers
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[i][tb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    };
    auto read_a = [&](auto buf_c) {
        constexpr int B = decltype(buf_c)::value;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) fa[tb][kb] = *(const half8*)(aRd[kb] + B * AH_B + tb * 16 * 128);
    };
    auto read_w = [&](auto buf_c, auto pb_c) {
        constexpr int B = decltype(buf_c)::value, PB = decltype(pb_c)::value;
        if (PB < 4) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) fw[cb][kb] = *(const half8*)(wRd[kb] + (PB * 2 + B) * WB_B + cb * 16 * 128);
        } else {
            // block 4 lies beyond the 64 KiB immediate range of wRd: one v_add per read, with an addend the compiler cannot hoist
            // (hoisted, the block-4 addresses cost registers this kernel does not have)
            int far;
            asm volatile("s_mov_b32 %0, 0x10000" : "=s"(far));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) fw[cb][kb] = *(const half8*)(wRd[kb] + far + B * WB_B + cb * 16 * 128);
        }
    };
    // the 16 MFMAs of fragment PB (2 k halves x 4 token blocks x 2 channel halves; an accumulator is revisited after 8); `mid(i)` runs behind
    // the 4th (i = 0) and the 10th (i = 1) MFMA: the phase's LDS-DMA requests
    // DBG 5: per phase, the summed length of the MFMA segments and of the stretch from one MFMA segment's start to the next one's (= two
    // barrier intervals), low 32 bits of s_memtime, in registers; written to p.workspace as [block][wave][12] u32 at the end
    unsigned dbg_seg[5] = {0, 0, 0, 0, 0}, dbg_gap[5] = {0, 0, 0, 0, 0}, dbg_last = 0, dbg_n = 0;
    auto mma = [&](auto pb_c, auto&& mid) {
        constexpr int PB = decltype(pb_c)::value;
        unsigned long long tb_ = 0;
        if (DBG == 5) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");
            if (dbg_n) dbg_gap[(PB + 4) % 5] += (unsigned)tb_ - dbg_last;
            dbg_last = (unsigned)tb_;
        }
        __builtin_amdgcn_s_setprio(1);
#if R8_MID_SPREAD
        // A 16x16x32 MFMA occupies the pipe for 16 cycles: room for three other instructions of this wave, not for the eight or nine of an
        // LDS-DMA request (address arithmetic, M0, the load) in one lump - behind one MFMA they stall the next ones (the 32x32x16 slots were
        // twice as long).  The two requests of the phase are handed to the scheduler with the MFMAs and a pipeline of
        // {1 MFMA, <= 2 VALU, <= 1 SALU, <= 1 VMEM} groups spreads them over the segment.
        mid(0); mid(1);
#endif
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    acc[PB][tb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb][kb], fa[tb][kb], acc[PB][tb][cb], 0, 0, 0);
#if !R8_MID_SPREAD
                    if (kb * 8 + tb * 2 + cb == 3) { SB(); mid(0); SB(); }
                    if (kb * 8 + tb * 2 + cb == 9) { SB(); mid(1); SB(); }
#endif
                }
#if R8_MID_SPREAD
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        if (DBG == 5) {
            unsigned long long te;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te)::"memory");
            dbg_seg[PB] += (unsigned)te - (unsigned)tb_;
            if (PB == 4) ++dbg_n;
        }
    };

    // (Round 4, negative: touching the residual block's cache lines with four dummy 4-byte loads per lane one K tile ahead of the epilogue
    //  halves the epilogue's wait - 33 500 -> 23 500 cycles per conv tile - but the loads sit in front of the K loop's counted vmcnt and
    //  cost the loop what the epilogue gains: profiles/r04_gemm_r8_epilogue_cycles.txt.  The epilogues of all CUs fall together and run at
    //  the chip's HBM rate; a start-up stagger does not keep them apart.)
    // ---- epilogue of the tile at (bm0, bn0), park buffer pb: no LDS ring access, no barriers, straight-line ----
    // Lane l = (token l & 15 of every 16-token block, channel group g = l / 16): after the two MFMAs of a fragment's channel halves it
    // holds channels wn*160 + pbk*32 + 8 g .. + 7 of its token (the W source-row map above) - bias / column-sum vectors are two adjacent
    // float4, the residual arrives in the same layout, one 16-byte store per (fragment, token block).
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // (bias, row bias, column sums) of 8 consecutive channels (OFF2 = 16) or of 4 values and their 4 gates (GEGLU: OFF2 = 128)
    auto park6 = [&](unsigned a, floatx4& b0, floatx4& b1, floatx4& r0, floatx4& r1, floatx4& c0, floatx4& c1) {
        if constexpr (GEGLU)
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:128\n\tds_read_b128 %2, %6 offset:6144\n\t"
                         "ds_read_b128 %3, %6 offset:6272\n\tds_read_b128 %4, %6 offset:2048\n\tds_read_b128 %5, %6 offset:2176\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(r0), "=&v"(r1), "=&v"(c0), "=&v"(c1) : "v"(a) : "memory");
        else
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:6144\n\t"
                         "ds_read_b128 %3, %6 offset:6160\n\tds_read_b128 %4, %6 offset:2048\n\tds_read_b128 %5, %6 offset:2064\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(r0), "=&v"(r1), "=&v"(c0), "=&v"(c1) : "v"(a) : "memory");
    };
    auto stat4 = [&](unsigned a, float2& s0, float2& s1, float2& s2, float2& s3) {   // (mean, rstd) of the lane's token in the four blocks
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b64 %2, %4 offset:256\n\tds_read_b64 %3, %4 offset:384\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a) : "memory");
    };
    auto epilogue = [&](int bm0, int bn0, int pb) {
        if (DBG == 2 || DBG == 5) {  // timing ablation: no epilogue; one dummy store keeps the accumulators live
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s += acc[i][tb][cb][r];
            if (s == 12345.678f) ((half_t*)p.c)[tid] = (half_t)s;
            return;
        }
        // the lane id is re-derived behind an opaque statement: every per-lane constant of the epilogue (row offsets, park addresses)
        // is then computed HERE instead of being hoisted across the K loop, where there are no registers left for them
        unsigned elane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        const int e15 = (int)(elane & 15), eg = (int)(elane >> 4);
        const srd_t rC = make_srd(p.c), rR = make_srd(p.residual ? p.residual : p.c);
        const unsigned park = lds0 + RING_B + pb * PARK_B;
        // v = rstd * (alpha * acc - mean * col_sum) + bias  ==  fma(ra, acc, fma(rm, col_sum, bias))
        // (a convolution never carries a folded LayerNorm: its (ra, rm) are the scalars (alpha, 0), not eight registers)
        float ra[LIN ? 4 : 1], rm[LIN ? 4 : 1];
        unsigned offc[4], offr[4];
        {
            float2 st[4];
            if (LIN) stat4(park + PK_ST + (wm * 64 + e15) * 8, st[0], st[1], st[2], st[3]);
            else { ra[0] = p.alpha; rm[0] = 0.f; }
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                const int m = bm0 + wm * 64 + tb * 16 + e15;
                if (LIN) {
                    const float mean = ln ? st[tb].x : 0.f, rstd = ln ? st[tb].y : 1.f;
                    ra[tb] = rstd * p.alpha; rm[tb] = -rstd * mean;
                }
                offc[tb] = m < p.M ? (unsigned)(m * (int)p.ldc * 2 + eg * (GEGLU ? 8 : 16)) : OOB_OFFSET;
                offr[tb] = m < p.M ? (unsigned)(m * (int)p.ldr * 2 + eg * 16) : OOB_OFFSET;
            }
        }
        if constexpr (GEGLU) {
#pragma unroll
            for (int pbk = 0; pbk < 5; ++pbk) {
                const int t = wn * 5 + pbk;
                const int nl = (t >> 1) * 64 + (t & 1) * 16 + 4 * eg;     // tile-local W row of the lane's first value; + 32: its gate
                float bh[4], ch[4], bg[4], cg[4];
                {
                    floatx4 tb_[2], tr[2], tc[2];
                    park6(park + nl * 4, tb_[0], tb_[1], tr[0], tr[1], tc[0], tc[1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bh[e] = tb_[0][e] + tr[0][e]; ch[e] = tc[0][e]; bg[e] = tb_[1][e] + tr[1][e]; cg[e] = tc[1][e]; }
                }
                const int on = (bn0 >> 1) + t * 16;                        // first output column of the fragment
                const bool okc = on + eg * 4 + 4 <= (p.N >> 1);
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = fmaf(ra[LIN ? tb : 0], acc[pbk][tb][0][e], fmaf(rm[LIN ? tb : 0], ch[e], bh[e]));
                        const float g = fmaf(ra[LIN ? tb : 0], acc[pbk][tb][1][e], fmaf(rm[LIN ? tb : 0], cg[e], bg[e]));
                        v[e] = x * gelu_erf_f(g);
                    }
                    const uint2v out = {pack_h2(v[0], v[1]), pack_h2(v[2], v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(out, rC, okc ? offc[tb] : OOB_OFFSET, on * 2, 0);
                    asm volatile("s_nop 7" ::"v"(out));
                }
            }
            return;
        }
        // Residual pieces: a vmcnt(0) also waits for the stores issued before it (loads and stores share the counter and do not retire in
        // order with respect to each other, so no counted wait is safe), i.e. one store round trip per wait.  Three waits per tile instead
        // of five, two of them behind a fragment's arithmetic: fragments 0, 1 are requested up front, fragment f + 2 re-uses the registers
        // of fragment f as soon as that one is consumed.
        uint4v rv[2][4];
        auto load_res = [&](auto pbk_c, auto slot_c) {
            constexpr int PBK = decltype(pbk_c)::value, SLOT = decltype(slot_c)::value;
            const int on = bn0 + wn * 160 + PBK * 32;
            const bool okc = on + eg * 8 + 8 <= p.N;
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) rv[SLOT][tb] = __builtin_amdgcn_raw_buffer_load_b128(rR, okc ? offr[tb] : OOB_OFFSET, on * 2, 0);
        };
        if (HAS_RES) {
            load_res(ic<0>{}, ic<0>{}); load_res(ic<1>{}, ic<1>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SB();
        }
        // STATS: (sum v, sum v^2) of the wave's 160 output columns per row - a lane holds 40 of them (its 8 channels of the five fragments),
        // the lanes 16 / 32 / 48 on hold the others: in-lane sums + two exchanges, one float2 per row and column half
        float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pbk = 0; pbk < 5; ++pbk) {
            const int slot = pbk & 1;
            if (HAS_RES && (pbk == 2 || pbk == 4)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SB(); }
            const int cl = wn * 160 + pbk * 32 + 8 * eg;     // tile-local first channel of the lane's eight
            float bs[8], cs[8];
            {
                floatx4 tb_[2], tr[2], tc[2];
                park6(park + cl * 4, tb_[0], tb_[1], tr[0], tr[1], tc[0], tc[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bs[4 * h + e] = tb_[h][e] + tr[h][e]; cs[4 * h + e] = tc[h][e]; }
            }
            const int on = bn0 + wn * 160 + pbk * 32;
            const bool okc = on + eg * 8 + 8 <= p.N;
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                float v[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * h + e] = fmaf(ra[LIN ? tb : 0], acc[pbk][tb][h][e], fmaf(rm[LIN ? tb : 0], cs[4 * h + e], bs[4 * h + e]));
                if (HAS_RES) {
                    const uint4v r = rv[slot][tb];
                    v[0] += h_lo(r[0]); v[1] += h_hi(r[0]); v[2] += h_lo(r[1]); v[3] += h_hi(r[1]);
                    v[4] += h_lo(r[2]); v[5] += h_hi(r[2]); v[6] += h_lo(r[3]); v[7] += h_hi(r[3]);
                }
                if (STATS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { st1[tb] += v[e]; st2[tb] = fmaf(v[e], v[e], st2[tb]); }
                }
                const uint4v out = {pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
                __builtin_amdgcn_raw_buffer_store_b128(out, rC, okc ? offc[tb] : OOB_OFFSET, on * 2, 0);
                asm volatile("s_nop 7" ::"v"(out));  // 16-byte store data pinned (2-waves-per-SIMD store hazard, profiles/r02_gemm_debug.md)
            }
            if (HAS_RES) {
                SB();
                if (pbk == 0) load_res(ic<2>{}, ic<0>{});
                if (pbk == 1) load_res(ic<3>{}, ic<1>{});
                if (pbk == 2) load_res(ic<4>{}, ic<0>{});
                SB();
            }
        }
        if (STATS) {   // part = column half of the whole matrix: bn0 / 160 + wn; tile-major [part][M][2] like the 128x128 kernel's
            const int part = bn0 / 160 + wn;
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                const int m = bm0 + wm * 64 + tb * 16 + e15;
                float s1 = st1[tb] + __shfl_xor(st1[tb], 16, 64), s2 = st2[tb] + __shfl_xor(st2[tb], 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (eg == 0 && m < p.M) *(float2*)(p.stats_out + ((int64_t)part * p.M + m) * 2) = make_float2(s1, s2);
            }
        }
    };

    // ---- prologue: K tile 0 completely, then A0, A1, W0, W1 of K tile 1 (what phases 2-4 of a preceding K tile would have requested) ----
    int cbm0, cbn0;
    int cv = blockIdx.x, cpb = 0;
    tile_origin(cv, cbm0, cbn0);
    set_stage_rows(cv);
    refresh_aoff();
    stage_park(0, cbm0, cbn0);
    stage_a(ic<0>{}, ic<0>{}, 0); stage_a(ic<0>{}, ic<0>{}, 1); stage_a(ic<1>{}, ic<0>{}, 0); stage_a(ic<1>{}, ic<0>{}, 1);
    stage_w(ic<0>{}, ic<0>{}); stage_w(ic<1>{}, ic<0>{}); stage_w(ic<2>{}, ic<0>{}); stage_w(ic<3>{}, ic<0>{}); stage_w(ic<4>{}, ic<0>{});
    advance();
    stage_a(ic<0>{}, ic<1>{}, 0); stage_a(ic<0>{}, ic<1>{}, 1); stage_a(ic<1>{}, ic<1>{}, 0); stage_a(ic<1>{}, ic<1>{}, 1);
    stage_w(ic<0>{}, ic<1>{}); stage_w(ic<1>{}, ic<1>{});
    wait_vmcnt<6>();
    BARRIER();                 // K tile 0 has landed for every wave
    zero_acc();

    // One K tile = 5 phases on ring buffer B (compile-time); `first` = first K tile of its output tile.
    auto tile_step = [&](auto buf_c, bool first, int nbm0, int nbn0, int npb) {
        constexpr int B = decltype(buf_c)::value;
        // ---- phase 0: fragment 0; the wave's A fragments for all five phases
        read_w(ic<B>{}, ic<0>{});
        SB();
        read_a(ic<B>{});
        SB();
        if (first) stage_park(npb, nbm0, nbn0);
        BARRIER();
        mma(ic<0>{}, [&](int i) { if (i == 0) stage_w(ic<2>{}, ic<B ^ 1>{}); else stage_w(ic<3>{}, ic<B ^ 1>{}); });
        BARRIER();
        // ---- phase 1
        read_w(ic<B>{}, ic<1>{});
        BARRIER();
        mma(ic<1>{}, [&](int i) { if (i == 0) stage_w(ic<4>{}, ic<B ^ 1>{}); });
        BARRIER();
        // ---- phase 2: the stream cursor moves to the next-but-one K tile
        read_w(ic<B>{}, ic<2>{});
        advance();
        BARRIER();
        mma(ic<2>{}, [&](int i) { stage_a(ic<0>{}, ic<B>{}, i); });
        BARRIER();
        // ---- phase 3
        read_w(ic<B>{}, ic<3>{});
        BARRIER();
        mma(ic<3>{}, [&](int i) { stage_a(ic<1>{}, ic<B>{}, i); });
        BARRIER();
        // ---- phase 4: the next K tile is complete behind the four A pieces requested last
        read_w(ic<B>{}, ic<4>{});
        wait_vmcnt<4>();
        BARRIER();
        mma(ic<4>{}, [&](int i) { if (i == 0) stage_w(ic<0>{}, ic<B>{}); else stage_w(ic<1>{}, ic<B>{}); });
        BARRIER();
    };
    int par = 0;
    unsigned long long cyc_loop = 0, cyc_join = 0, cyc_epi = 0, ntl = 0, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    auto stamp = [&](unsigned long long& t) { if (DBG == 4) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); };
    for (; cv < ntiles; cv += G) {
        tile_origin(cv, cbm0, cbn0);
        stamp(ts0);
        if (grp == 1) BARRIER();    // stagger: waves 4-7 run one barrier interval behind
        bool first = cv != (int)blockIdx.x;  // the very first tile's park vectors were requested by the prologue
        int t = 0;
        if (par) { tile_step(ic<1>{}, first, cbm0, cbn0, cpb); first = false; t = 1; par = 0; }
        for (; t + 1 < nk; t += 2) {
            tile_step(ic<0>{}, first, cbm0, cbn0, cpb);
            first = false;
            tile_step(ic<1>{}, false, cbm0, cbn0, cpb);
        }
        if (t < nk) { tile_step(ic<0>{}, first, cbm0, cbn0, cpb); par = 1; }
        stamp(ts1);
        if (grp == 0) BARRIER();    // re-join: both groups run the epilogue concurrently
        stamp(ts2);
        epilogue(cbm0, cbn0, cpb);
        stamp(ts3);
        if (DBG == 4) { cyc_loop += ts1 - ts0; cyc_join += ts2 - ts1; cyc_epi += ts3 - ts2; ++ntl; }
        zero_acc();
        cpb ^= 1;
    }
    if (DBG == 5 && p.workspace && blockIdx.x < 64 && lane == 0) {
        unsigned* o = (unsigned*)p.workspace + ((int)blockIdx.x * 8 + wid) * 12;
#pragma unroll
        for (int i = 0; i < 5; ++i) { o[i] = dbg_seg[i]; o[5 + i] = dbg_gap[i]; }
        o[10] = dbg_n; o[11] = 0;
    }
    if (DBG == 4 && p.workspace && blockIdx.x < 64 && lane == 0) {
        unsigned long long* o = (unsigned long long*)p.workspace + ((int)blockIdx.x * 8 + wid) * 4;
        o[0] = cyc_loop; o[1] = cyc_join; o[2] = cyc_epi; o[3] = ntl;
    }
}

template <int MODE, bool HAS_RES, int DBG = 0, bool SPLIT = true, bool STATS = false, bool GEGLU = false, bool CIM = false>
int launch_r8(const insv2v_gemm_desc& d, hipStream_t s) {
    static bool attr_set = false;
    static int num_cu = 0;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_r8_kernel<MODE, HAS_RES, DBG, SPLIT, STATS, GEGLU, CIM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    // INSV2V_R8_DEPHASE: permille of a tile time over which the workgroups with one tile to spare start late (0 = off)
    static const int dephase = getenv("INSV2V_R8_DEPHASE") ? atoi(getenv("INSV2V_R8_DEPHASE")) : 0;
    hipLaunchKernelGGL((gemm_r8_kernel<MODE, HAS_RES, DBG, SPLIT, STATS, GEGLU, CIM>), dim3(tiles < num_cu ? tiles : num_cu), dim3(512), LDS_B, s, d, dephase);
    return launch_status();
}

}  // namespace

// variant: 0 = product; 2 = no epilogue (timing)
int insv2v_gemm_r8(const insv2v_gemm_desc& d, int variant, hipStream_t s) {
    if (d.batch > 1 || d.c_fp32 || d.split_k > 1) return INSV2V_EUNSUPPORTED;
    if ((d.K % BK) || (d.N & 7) || (d.ldc & 7) || ((uintptr_t)d.c & 15)) return INSV2V_EUNSUPPORTED;
    if (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15))) return INSV2V_EUNSUPPORTED;
    if (d.k_split && (d.k_split % BK)) return INSV2V_EUNSUPPORTED;
    const bool geglu = d.act == INSV2V_ACT_GEGLU;
    if (d.act != INSV2V_ACT_NONE && !geglu) return INSV2V_EUNSUPPORTED;
    // GEGLU: W rows in [32 value | 32 gate] blocks, whole 320-row tiles (= 160 outputs), no residual / second source / statistics
    if (geglu && (d.mode != INSV2V_MODE_LINEAR || (d.N % BN) || d.residual || d.k_split > 0 || d.stats_out || variant != 0)) return INSV2V_EUNSUPPORTED;
    if (d.row_stats && (d.M & 1)) return INSV2V_EUNSUPPORTED;  // (mean, rstd) pairs are fetched two rows per lane
    // the row-bias vector is parked per tile: every 256-row tile must lie inside one group
    if (d.row_bias && ((d.ld_rb & 3) || (d.rows_per_group % 256 && d.M > d.rows_per_group))) return INSV2V_EUNSUPPORTED;
    if ((int64_t)d.M * d.ldc * 2 >= ((int64_t)1 << 31) || (d.residual && (int64_t)d.M * d.ldr * 2 >= ((int64_t)1 << 31))) return INSV2V_EUNSUPPORTED;
    if ((d.bias && ((uintptr_t)d.bias & 15)) || (d.col_sum && ((uintptr_t)d.col_sum & 15)) || (d.row_bias && ((uintptr_t)d.row_bias & 15)))
        return INSV2V_EUNSUPPORTED;
    const bool conv = d.mode == INSV2V_MODE_CONV3X3;
    if (conv && ((d.Cin % BK) || d.M >= (1 << 24))) return INSV2V_EUNSUPPORTED;   // (row -> pixel by fp32 division: exact below 2^24 rows)
    const bool res = d.residual != nullptr;
    constexpr int L = INSV2V_MODE_LINEAR, C = INSV2V_MODE_CONV3X3;
    if (geglu) return launch_r8<L, false, 0, false, false, true>(d, s);
    if (d.stats_out) {   // partial row statistics of the output: LINEAR, whole 320-column tiles, one A source
        if (conv || d.k_split > 0 || (d.N % BN) || variant != 0) return INSV2V_EUNSUPPORTED;
        return res ? launch_r8<L, true, 0, false, true>(d, s) : launch_r8<L, false, 0, false, true>(d, s);
    }
    // stride-1 / pad-1 convolutions (every ResnetBlock3D convolution) walk K channel-block-major (template flag CIM); INSV2V_R8_CIM=0
    // keeps the tap-major order for A/B runs
    static const bool cim_on = !(getenv("INSV2V_R8_CIM") && atoi(getenv("INSV2V_R8_CIM")) == 0);
    const bool cim = conv && cim_on && d.stride == 1 && !d.upsample && d.pad_t == 1 && d.pad_l == 1 && d.IH == d.OH && d.IW == d.OW;
    if (cim && (variant == 0 || variant == 4)) {
        if (variant == 4) return res ? launch_r8<C, true, 4, true, false, false, true>(d, s) : launch_r8<C, false, 4, true, false, false, true>(d, s);
        if (d.k_split > 0) return res ? launch_r8<C, true, 0, true, false, false, true>(d, s) : launch_r8<C, false, 0, true, false, false, true>(d, s);
        return res ? launch_r8<C, true, 0, false, false, false, true>(d, s) : launch_r8<C, false, 0, false, false, false, true>(d, s);
    }
    switch (variant) {
        case 0:
            if (d.k_split > 0) {
                if (conv) return res ? launch_r8<C, true>(d, s) : launch_r8<C, false>(d, s);
                return res ? launch_r8<L, true>(d, s) : launch_r8<L, false>(d, s);
            }
            if (conv) return res ? launch_r8<C, true, 0, false>(d, s) : launch_r8<C, false, 0, false>(d, s);
            return res ? launch_r8<L, true, 0, false>(d, s) : launch_r8<L, false, 0, false>(d, s);
        case 2:
            if (conv) return launch_r8<C, false, 2>(d, s);
            return launch_r8<L, false, 2>(d, s);
        case 5:
            if (conv) return launch_r8<C, false, 5>(d, s);
            return launch_r8<L, false, 5>(d, s);
        case 4:
            if (conv) return res ? launch_r8<C, true, 4>(d, s) : launch_r8<C, false, 4>(d, s);
            return res ? launch_r8<L, true, 4>(d, s) : launch_r8<L, false, 4>(d, s);
    }
    return INSV2V_EINVAL;
}
