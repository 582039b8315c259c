// insv2v_attention: flash-style fused attention for spatial self-, text cross- and temporal
// self-attention (one addressing rule, see include/insv2v_hip.h).
//
// Roofline: MFMA-bound for long sequences (spatial, 1536 tokens), HBM/launch-bound for the
// 16-frame temporal case.  One wave owns QB blocks of 16 query rows; all contractions use
// v_mfma_f32_16x16x32_f16:
//   S^T = K . Q^T   (A = K fragment from LDS, B = Q fragment held in registers), so a lane owns
//                   ONE query column (lane&15) and 4 consecutive keys per 16-key block;
//   softmax         row max / row sum over keys = in-lane reduction + 2 wave shuffles (xor 16,32);
//                   the running max/sum and the output rescale factor are lane-local;
//   O^T = V^T . P^T (A = V^T fragment from a transposed LDS tile, B = the lane's own P values,
//                   no data movement: the MFMA k-slot <-> key mapping is chosen to match what
//                   the lane already holds).
// Every K / V^T fragment read from LDS feeds QB MFMAs (QB = 2 for long sequences: 128 query rows
// per workgroup halve both the LDS reads and the L2 traffic per FLOP).  K/V tiles of 64 keys
// (32 for the short temporal sequences) are double-buffered in LDS; global loads for tile t+1 are
// issued before the MFMAs of tile t.  Key masking is only executed on the ragged last tile (every tile when causal).
#include "common.h"
#include "gemm_dma.h"
#include <cstdlib>
#include <algorithm>

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2h(float a, float b) {
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);  // P in [0,1]: round-toward-zero costs <= 1 fp16 ulp
    return __builtin_bit_cast(unsigned, h);
}

typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// Max over the 4 lanes {qc, qc+16, qc+32, qc+48} that share a query column, on the VALU (no LDS round trip as with
// ds_bpermute): v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second; fed two copies of v they leave
// (row 0|0|2|2, row 1|1|3|3) and (lower|lower, upper|upper).  Inline asm: hipcc folds the builtin with identical operands.
__device__ __forceinline__ float col_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = b = fmaxf(a, b);
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float col_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = b = a + b;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// Second launch bound = waves per SIMD the register allocation must allow.  The 8-wave, 128-query workgroups of the long
// d <= 64 sequences need 129 VGPRs unconstrained - one over the 128 that let TWO workgroups share a CU (4 waves per SIMD), which
// is what overlaps one workgroup's softmax (VALU, v_exp_f32 bound) with the other's MFMAs; inside one workgroup the per-tile
// barrier keeps all waves in the same phase.
//
// FOLD (head dims with a spare, zero-padded contraction column: 40, 80): the softmax is VALU-bound - per 16 x 32 score block 112 cycles
// of MFMA stand against ~230 cycles of max / scale-and-shift / v_exp_f32 / convert / row-sum, and only the exponential (128 of them) is
// irreducible.  The rest moves into the matrix pipe:
//   * Q is pre-multiplied by scale * log2(e) when it is loaded, and its first padding column holds -m~ (the running maximum, rounded
//     to fp16) against a 1 in that column of every K fragment: the MFMAs return s' = scale' q.k - m~ and the probabilities are
//     exp2(s') directly.  m~ is common to a whole row, so its rounding cancels in the normalisation;
//   * the maximum is only CHECKED per tile (v_max3 over the lane's values, one ballot): as long as no score exceeds m~ by more than
//     2^8 nothing is rescaled - probabilities up to 256 are as exact in fp16 as those below 1.  The first tile, and any tile where a
//     lane sees s' > 8, takes the exact path: column maximum, m~ moved, accumulators rescaled, the tile's scores shifted;
//   * ONES (the output tiles have a spare row too: 40): the row sum of the ROUNDED probabilities is a row of ones in V^T - it comes
//     out of the P.V MFMAs as output channel D and is rescaled with the accumulators.
// REP (with SINGLE): the workgroup serves the p.kv_inner CONSECUTIVE problems that share one K / V (the text cross-attention of the frames of a
// sample: kv_step == 0).  The tile is loaded and transposed once; the problems' queries stream through it - per problem only the Q
// fragments come in and the output goes out (round 6: one workgroup per (frame, head) spent 20 us on a latency chain of K / V load ->
// LDS -> barrier -> 60 MFMAs -> store, 302 us per B = 60 launch for 472 MB of q + o).
#ifndef ATTN_REP_PREFETCH
#define ATTN_REP_PREFETCH 1
#endif
#ifndef ATTN_SINGLE_ST16
#define ATTN_SINGLE_ST16 1
#endif
template <int D, int NW, int KVT, int QB, bool FOLD = false, bool SINGLE = false, bool REP = false>
__global__ __launch_bounds__(NW * 64, (NW == 8 && D == 40) ? 4 : SINGLE ? 3 : (D == 80 && QB == 2) ? 2 : 1) void attn_kernel(insv2v_attention_desc p) {
    // SINGLE: the whole key sequence is ONE tile (seq_k <= KVT, checked on the host): one K / V^T buffer instead of two, so that two
    // workgroups share a CU's LDS (the 96-token attention of the 8x12 level at d = 160: 64 KB instead of 2 x 43 KB)
    constexpr int NBUF = SINGLE ? 1 : 2;
    static_assert(!(SINGLE && FOLD), "the single-tile form is for the unfolded kernel");
    static_assert(!REP || SINGLE, "only the single-tile form keeps its tile across problems");
    constexpr int DP = (D + 31) / 32 * 32;  // head dim zero-padded to the MFMA K granularity (LDS/registers only)
    constexpr int DTA = (D + 15) / 16;      // output column tiles actually computed
    static_assert(!FOLD || DP > D, "FOLD needs a zero-padded contraction column");
    constexpr bool ONES = FOLD && (D % 16) != 0;              // spare output row = channel D
    constexpr int FKK = D / 32, FG = (D % 32) / 8, FE = D % 8;  // contraction column D: k-step, lane group, element of the fragment
    // VDMA (with FOLD): the V tile goes global -> LDS by LDS-DMA like K - no staging registers, no transposing ds_writes (24 VALU + 4
    // ds_write2 + 2 buffer loads per thread and tile in a kernel that is VALU-issue bound: profiles/r04_attn_pipelined_experiment.txt).
    // The LDS image is per 16-channel block [64 keys][16 channels] (32-byte rows, a half block [64][8] for d = 40), built by the
    // per-lane SOURCE offsets of the DMA; the A fragments of P.V (4 consecutive keys of one channel per lane) come out of
    // ds_read_b64_tr_b16: the 16 lanes of a group address a [4 keys][16 channels] block as 4 x 8 bytes per key, lane q receives
    // column q (tools/tr_probe.hip).  All 64 lanes of a read cover 512 contiguous bytes: no bank conflicts.  The lanes of channels
    // 40-47 (d = 40) address a 16-byte constant instead: (1, 0, 0, 0 | 0, 0, 0, 0) = the row of ones that yields the row sum.
    constexpr bool VDMA = FOLD && KVT == 64 && (D % 8) == 0 && (D % 16 == 0 || ONES);
    // 16-byte output stores (see the P.V loop): 350 -> 322 us for the 96-token self-attention launch of the B = 60 stack; the REP form loses
    // 5 % with them (same-box A/B, profiles/r06_attn_rep.txt) and keeps the 8-byte ones
    constexpr bool ST16 = SINGLE && !REP && ATTN_SINGLE_ST16 && (DTA % 2 == 0) && (D % 32 == 0);
    constexpr int NCBF = D / 16;                  // full 16-channel blocks
    constexpr bool VHALF = (D % 16) != 0;        // + one 8-channel block
    constexpr int VIMG = NCBF * KVT * 32 + (VHALF ? KVT * 16 : 0);   // bytes of one V tile image
    constexpr int NPV = VIMG / 1024, NPVW = (NPV + NW - 1) / NW;    // 1 KiB DMA pieces per tile, per wave
    constexpr int VT_LD = KVT + 4;       // halfs per row of the transposed V tile: (KVT+4)/2 dwords = 2 (mod 32)
                                         // -> the 16 lanes of a ds_read2_b64 group hit 16 distinct bank pairs
    constexpr int NKB = KVT / 32;        // 32-key MFMA blocks per tile
    // K tile rows: for DP == 64 (d = 40/64) unpadded 128-byte rows with the 16-byte chunk index XOR-swizzled
    // by ((key>>1)&7) (conflict-free ds_read_b128 for the 16-key x 4-chunk fragment pattern); otherwise
    // rows padded by 16 bytes.
    constexpr bool KSWZ = DP == 64;
    constexpr int KLD = KSWZ ? DP : DP + 8;  // halfs per K row in LDS
    constexpr int DT = DP / 16;          // max output column tiles
    constexpr int KS = DP / 32;          // k-steps of the QK^T contraction
    constexpr int NT = NW * 64;
    constexpr int KCH = DP / 8;          // 16B chunks per K row (zero padded to DP)
    constexpr int K_ITERS = (KVT * KCH + NT - 1) / NT;
    constexpr int VP = KVT / 2;          // key pairs per tile
    constexpr int V_ITERS = (VP * KCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = (half_t*)smem;                // [2][KVT][KLD]
    half_t* sVt = sK + NBUF * KVT * KLD;       // [NBUF][DP][VT_LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, qc = lane & 15;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs in linear id order (x fastest), which would put the
    // query blocks of one (batch, head) - they stream the SAME K/V - and the neighbouring heads of one token row - they
    // share 128-byte lines of the fused qkv rows - on 8 different L2s.  Remap so consecutive ids share an XCD.
    // grid = (query blocks * heads * batch) folded into x: no 65 535 limit on batch or heads
    const int nqb = (p.seq_q + 16 * NW * QB - 1) / (16 * NW * QB), nhd = p.heads;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lin % nqb, head = (lin / nqb) % nhd;
    const int nrep = REP ? p.kv_inner : 1;                 // problems this workgroup serves: z0 .. z0 + nrep - 1
    const int z0 = (lin / (nqb * nhd)) * nrep;
    constexpr int d = D;
    const half_t* K = (const half_t*)p.k + (int64_t)(z0 / p.kv_inner) * p.kv_outer + (int64_t)(z0 % p.kv_inner) * p.kv_step + head * d;
    const half_t* V = (const half_t*)p.v + (int64_t)(z0 / p.kv_inner) * p.kv_outer + (int64_t)(z0 % p.kv_inner) * p.kv_step + head * d;

    // Q fragments (B operand): lane (g,qc) holds Q[q][kk*32 + g*8 .. +8] for each of its QB query blocks
    int qrow[QB];
    half8 qf[QB][KS];
#pragma unroll
    for (int b = 0; b < QB; ++b) qrow[b] = (qblk * NW + wid) * (16 * QB) + b * 16 + qc;
    auto load_q = [&](int z) {
        const half_t* Q = (const half_t*)p.q + (int64_t)(z / p.q_inner) * p.q_outer + (int64_t)(z % p.q_inner) * p.q_step + head * d;
        // (plain loads under the row guard: hipcc forms ONE guarded region for a block's fragments.  Buffer loads with out-of-range offsets
        //  instead - the K / V tiles' form - were measured slower here: 244 vs 200 us for the REP launch, profiles/r06_attn_rep.txt)
#pragma unroll
        for (int b = 0; b < QB; ++b) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int c = kk * 32 + g * 8;
                half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (qrow[b] < p.seq_q && c < d) v = *(const half8*)(Q + (int64_t)qrow[b] * p.q_rs + c);
                if (FOLD) {
                    const float c2q = p.scale * 1.4426950408889634f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * c2q);
                }
                qf[b][kk] = v;
            }
        }
    };
    load_q(z0);
    // FOLD: the 1 of contraction column D in every K fragment of k-step FKK (the column is zero padding in LDS): one v_or per fragment
    const unsigned kone = (FOLD && g == FG) ? ((FE & 1) ? 0x3C000000u : 0x00003C00u) : 0u;

    uint4 rk[K_ITERS], rv[V_ITERS][2];
    // Tile loads are buffer loads whose offset is out of range (-> zeros) for pieces outside the problem: a plain load inside a
    // divergent `if` costs a control-flow join at which hipcc waits vmcnt(0) - the K, V(even) and V(odd) requests of a tile then went
    // out one L2 round trip after the other at the top of every iteration.  (Offsets are relative to the (z, head) base: < 2^31.)
    const srd_t rK = make_srd(K), rVv = make_srd(V);
    // K tiles go global -> LDS by LDS-DMA whenever the tile image is a whole number of 1 KiB pieces: no staging registers (the
    // 128-VGPR variants spilled them and waited for the load on the spot), no ds_write.  The DMA writes lane-linearly, so padding /
    // swizzle are applied to the per-lane SOURCE offset; pieces outside the problem use an out-of-range offset (zeros).
    constexpr int KROW_B = KLD * 2;
    constexpr bool KDMA = (KVT * KROW_B) % 1024 == 0;
    constexpr int NPK = KVT * KROW_B / 1024, NPKW = (NPK + NW - 1) / NW;
    static_assert(!VDMA || KDMA, "the V DMA path shares the K path's end-of-tile wait");
    int kd_key[KDMA ? NPKW : 1];
    unsigned kd_src[KDMA ? NPKW : 1];
    // The offset register of an LDS-DMA request must not be reused while the request is in flight: hipcc treats it as the load's
    // pending destination and puts vmcnt(0) in front of the next instruction that touches it.  kd_off[] is kept live (pinned by an
    // empty asm) until the end of the iteration, where the tile is awaited anyway.
    unsigned kd_off[KDMA ? NPKW : 1] = {};
    if (KDMA) {
#pragma unroll
        for (int i = 0; i < NPKW; ++i) {
            const int o = (wid + NW * i) * 1024 + lane * 16;
            const int key = o / KROW_B, slot = (o - key * KROW_B) >> 4;
            const int ch = KSWZ ? (slot ^ ((key >> 1) & 7)) : slot;
            kd_key[i] = (wid + NW * i < NPK && ch * 8 < d) ? key : (1 << 30);   // never < seq_k
            kd_src[i] = (unsigned)((key * (int)p.k_rs + ch * 8) * 2);
        }
    }
    // V pieces: piece pc < 2 NCBF = 32 keys x 32 bytes of block pc / 2 (lane = key, half row); the last piece of d = 40 = 64 keys x 16 bytes
    int vd_key[VDMA ? NPVW : 1];
    unsigned vd_src[VDMA ? NPVW : 1];
    unsigned vd_off[VDMA ? NPVW : 1] = {};
    char* const sV = (char*)(sK + NBUF * KVT * KLD);                  // VDMA: [2][VIMG] + 16 constant bytes
    if (VDMA) {
#pragma unroll
        for (int i = 0; i < NPVW; ++i) {
            const int pc = wid + NW * i;
            const bool full = pc < 2 * NCBF;
            const int key = full ? (pc & 1) * 32 + (lane >> 1) : lane;
            const int ch = full ? (pc >> 1) * 16 + (lane & 1) * 8 : NCBF * 16;
            vd_key[i] = pc < NPV ? key : (1 << 30);
            vd_src[i] = (unsigned)((key * (int)p.v_rs + ch) * 2);
        }
        if (tid < 4) ((unsigned*)(sV + 2 * VIMG))[tid] = tid == 0 ? 0x00003C00u : 0u;   // (1, 0, 0, 0, 0, 0, 0, 0)
    }
    auto load_vdma = [&](int t, int buf) {
        const int key0 = t * KVT;
#pragma unroll
        for (int i = 0; i < NPVW; ++i) {
            vd_off[i] = key0 + vd_key[i] < p.seq_k ? vd_src[i] : OOB_OFFSET;
            if (wid + NW * i < NPV)   // wave-uniform
                dma16(rVv, vd_off[i], key0 * (int)p.v_rs * 2, sV + buf * VIMG + (wid + NW * i) * 1024);
        }
    };
    auto load_k = [&](int t, int buf) {
        const int key0 = t * KVT;
        if (KDMA) {
            char* dst = (char*)(sK + buf * KVT * KLD) + wid * 1024;
#pragma unroll
            for (int i = 0; i < NPKW; ++i) {
                kd_off[i] = key0 + kd_key[i] < p.seq_k ? kd_src[i] : OOB_OFFSET;
                if (wid + NW * i < NPK)   // wave-uniform
                    dma16(rK, kd_off[i], key0 * (int)p.k_rs * 2, dst + i * NW * 1024);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int e = tid + NT * i;
            const int key = e / KCH, ch = e - key * KCH;  // chunk fastest: coalesced rows
            const bool ok = e < KVT * KCH && key0 + key < p.seq_k && ch * 8 < d;
            const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(rK, ok ? (unsigned)(((key0 + key) * (int)p.k_rs + ch * 8) * 2) : OOB_OFFSET, 0, 0);
            rk[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto load_v = [&](int t) {
        const int key0 = t * KVT;
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            const int e = tid + NT * i;
            const int ch = e / VP, kp = e - ch * VP;  // key pair fastest: 4-byte transposed LDS writes
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int key = key0 + 2 * kp + h;
                const bool ok = e < VP * KCH && key < p.seq_k && ch * 8 < d;
                const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(rVv, ok ? (unsigned)((key * (int)p.v_rs + ch * 8) * 2) : OOB_OFFSET, 0, 0);
                rv[i][h] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto load_tile = [&](int t, int buf) {
        if constexpr (VDMA) load_vdma(t, buf);
        else load_v(t);
        load_k(t, buf);
    };
    auto store_tile = [&](int buf) {
        half_t* k = sK + buf * KVT * KLD;
        half_t* vt = sVt + buf * DP * VT_LD;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            const int e = tid + NT * i;
            const int key = e / KCH, ch = e - key * KCH;
            const int pc = KSWZ ? (ch ^ ((key >> 1) & 7)) : ch;
            if (!KDMA && e < KVT * KCH) *(uint4*)(k + key * KLD + pc * 8) = rk[i];
        }
        if constexpr (VDMA) return;
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            const int e = tid + NT * i;
            const int ch = e / VP, kp = e - ch * VP;
            if (e < VP * KCH) {
                const unsigned short* h0 = (const unsigned short*)&rv[i][0];
                const unsigned short* h1 = (const unsigned short*)&rv[i][1];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned w = (unsigned)h0[j] | ((unsigned)h1[j] << 16);
                    if (ONES && j == D % 8 && ch == D / 8) w = 0x3C003C00u;   // row D of V^T = ones: output channel D = sum of P
                    *(unsigned*)(vt + (ch * 8 + j) * VT_LD + 2 * kp) = w;
                }
            }
        }
    };

    floatx4 acc[QB][DT];
    float m_run[QB], l_run[QB];
    const float c2 = p.scale * 1.4426950408889634f;

    // VDMA: per-lane LDS byte addresses of the transpose reads, relative to the tile image (full blocks) / absolute (half block)
    const unsigned sv_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sV;
    const unsigned va_full = sv_lds + (g * 4 + (qc >> 2)) * 32 + (qc & 3) * 8;
    const bool va_const = (qc & 3) >= 2;                            // lanes of channels 8-15 of the half block: the constant
    const unsigned va_half0 = va_const ? sv_lds + 2 * VIMG + ((qc & 3) == 3 ? 8 : 0) : sv_lds + NCBF * KVT * 32 + (g * 4 + (qc >> 2)) * 16 + (qc & 3) * 8;
    const unsigned va_hstep_buf = va_const ? 0u : (unsigned)VIMG, va_hstep_kb = va_const ? 0u : 512u, va_hstep_hi = va_const ? 0u : 256u;

    const int ntiles = (p.seq_k + KVT - 1) / KVT;
    load_tile(0, 0);
    store_tile(0);
    if (KDMA) wait_vmcnt<0>();  // other waves read the K pieces this wave DMA'd: do not rely on hipcc waiting before the barrier
    __syncthreads();

#pragma unroll 1
  for (int rep = 0; rep < nrep; ++rep) {
    const int z = z0 + rep;
#if !ATTN_REP_PREFETCH
    if (REP && rep > 0) load_q(z);
#endif
    half_t* O = (half_t*)p.o + (int64_t)(z / p.o_inner) * p.o_outer + (int64_t)(z % p.o_inner) * p.o_step + head * d;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        m_run[b] = FOLD ? 0.f : -1.0e30f;
        l_run[b] = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i) acc[b][i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    }
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        // tile t+1 is requested after the Q.K^T MFMAs (the scores' registers are the pressure peak) and lands under softmax + P.V
        const half_t* k = sK + cur * KVT * KLD;
        const half_t* vt = sVt + cur * DP * VT_LD;
        const int key0 = t * KVT;
        const bool ragged = key0 + KVT > p.seq_k;

        // ---- scores: s[b][kb][sub] = 16 keys x 16 queries, lane: keys key0+kb*32+sub*16+g*4+r
        floatx4 s[QB][NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int b = 0; b < QB; ++b) s[b][kb][sub] = (floatx4){0.f, 0.f, 0.f, 0.f};
                const half_t* kr = k + (kb * 32 + sub * 16 + qc) * KLD;
                const int ksw = KSWZ ? ((qc >> 1) & 7) : 0;  // (row>>1)&7: rows differ from qc by multiples of 16
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    half8 kf = *(const half8*)(kr + (((kk * 4 + g) ^ ksw) * 8));
                    if (FOLD && kk == FKK) {
                        uint4v ku = __builtin_bit_cast(uint4v, kf);
                        ku[FE / 2] |= kone;
                        kf = __builtin_bit_cast(half8, ku);
                    }
#pragma unroll
                    for (int b = 0; b < QB; ++b)
                        s[b][kb][sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[b][kk], s[b][kb][sub], 0, 0, 0);
                }
            }
        if (t + 1 < ntiles) load_tile(t + 1, cur ^ 1);
#if ATTN_REP_PREFETCH
        if (REP && rep + 1 < nrep) load_q(z + 1);   // the next problem's queries land under this problem's softmax, P.V and stores
#endif
        half8 pf[QB][NKB];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            if (ragged || p.causal) {
                const int kmax = p.causal ? min(p.seq_k - 1, qrow[b]) : p.seq_k - 1;  // last visible key of this lane's query
                const int rel = kmax - key0 - g * 4;   // (computed inside the branch: constants against one register, no per-key adds)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kb * 32 + sub * 16 + r > rel) s[b][kb][sub][r] = -1.0e30f;
            }
            if constexpr (FOLD) {
                // the lane's largest s' (scores arrive relative to m~ and in log2 units)
                float mx = s[b][0][0][0];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[b][kb][sub][r]);
                if (t == 0 || __builtin_amdgcn_ballot_w64(mx > 8.f) != 0) {   // wave-uniform: the exact path
                    mx = col_max(mx);
                    const float want = m_run[b] + (t == 0 ? mx : fmaxf(mx, 0.f));
                    const float mnew = (float)(half_t)want;       // what the Q fragment can hold
                    const float shift = mnew - m_run[b];           // exact: both are fp16 values
                    m_run[b] = mnew;
                    if (t > 0) {
                        const float alpha = __builtin_amdgcn_exp2f(-shift);
                        if (!ONES) l_run[b] *= alpha;
#pragma unroll
                        for (int i = 0; i < DTA; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[b][i][r] *= alpha;
                    }
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s[b][kb][sub][r] -= shift;
                    if (g == FG) qf[b][FKK][FE] = (half_t)(-mnew);
                }
                const half2v ones = {(_Float16)1.f, (_Float16)1.f};
                float psum = 0.f;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    uint4v pk;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned u = pack2h(__builtin_amdgcn_exp2f(s[b][kb][sub][2 * h]), __builtin_amdgcn_exp2f(s[b][kb][sub][2 * h + 1]));
                            if (!ONES) psum = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, u), ones, psum, false);
                            pk[sub * 2 + h] = u;
                        }
                    pf[b][kb] = __builtin_bit_cast(half8, pk);
                }
                if (!ONES) l_run[b] += psum;
                continue;
            }
            float mx = m_run[b];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[b][kb][sub][r]);
            mx = col_max(mx);
            // the running maximum rarely moves after the first tiles: rescale only when some query of the wave needs it
            if (__builtin_amdgcn_ballot_w64(mx > m_run[b]) != 0) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[b] - mx) * c2);
                m_run[b] = mx;
                l_run[b] *= alpha;
#pragma unroll
                for (int i = 0; i < DTA; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[b][i][r] *= alpha;
            }
            const float2v c2v = {c2, c2}, mcv = {-mx * c2, -mx * c2};
            const half2v ones = {(_Float16)1.f, (_Float16)1.f};
            float psum = 0.f;  // sum of the ROUNDED probabilities (what the PV MFMAs see): v_dot2_f32_f16 with ones
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                uint4v pk;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float2v x = {s[b][kb][sub][2 * h], s[b][kb][sub][2 * h + 1]};
                        x = __builtin_elementwise_fma(x, c2v, mcv);  // v_pk_fma_f32
                        const unsigned u = pack2h(__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1]));
                        psum = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, u), ones, psum, false);
                        pk[sub * 2 + h] = u;
                    }
                pf[b][kb] = __builtin_bit_cast(half8, pk);
            }
            l_run[b] += psum;
        }

        // ---- O^T += V^T . P^T ; A fragment: Vt[dt*16+qc][kb*32 + g*4 + {0..3}] | [.. + 16 + g*4 + {0..3}]
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int i = 0; i < DTA; ++i) {
                half4 lo, hi;
                if constexpr (VDMA) {
                    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
                    typedef __attribute__((address_space(3))) fp16x4* lds4_t;
                    if (i < NCBF) {
                        const unsigned a = va_full + cur * VIMG + i * (KVT * 32) + kb * 1024;
                        lo = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(uintptr_t)a));
                        hi = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(uintptr_t)(a + 512)));
                    } else {
                        const unsigned a = va_half0 + cur * va_hstep_buf + kb * va_hstep_kb;
                        lo = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(uintptr_t)a));
                        hi = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(uintptr_t)(a + va_hstep_hi)));
                    }
                } else {
                    // ST16: the V^T rows (channels) a lane feeds are permuted so that after the MFMAs of a tile PAIR lane group g holds the 8
                    // consecutive channels 32 (i / 2) + 8 g .. + 7 of its query (16-byte stores; the rows stay conflict-free: 50-dword stride)
                    const int vrow = ST16 ? (i >> 1) * 32 + 8 * (qc >> 2) + 4 * (i & 1) + (qc & 3) : i * 16 + qc;
                    const half_t* vr = vt + vrow * VT_LD + kb * 32 + g * 4;
                    lo = *(const half4*)vr;
                    hi = *(const half4*)(vr + 16);
                }
                const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int b = 0; b < QB; ++b)
                    acc[b][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[b][kb], acc[b][i], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) store_tile(cur ^ 1);
        if (KDMA) {
#pragma unroll
            for (int i = 0; i < NPKW; ++i) asm volatile("" :: "v"(kd_off[i]));
            if (VDMA) {
#pragma unroll
                for (int i = 0; i < NPVW; ++i) asm volatile("" :: "v"(vd_off[i]));
            }
            wait_vmcnt<0>();  // explicit: the next tile's K pieces must have landed before any wave passes the barrier
        }
        if (!REP) __syncthreads();   // (REP: the one tile is read-only after the prologue)
    }

#pragma unroll
    for (int b = 0; b < QB; ++b) {
        // ONES: the row sum is output channel D = tile D / 16, row D % 16 = lane group (D % 16) / 4, register D % 4 of the lane's column
        const float inv = 1.f / (ONES ? __shfl(acc[b][D / 16][D % 4], qc + 16 * ((D % 16) / 4), 64) : col_sum(l_run[b]));
        if (ST16) {
            if (qrow[b] < p.seq_q) {
                half_t* orow = O + (int64_t)qrow[b] * p.o_rs;
#pragma unroll
                for (int pp = 0; pp < DTA / 2; ++pp) {
                    const int c = pp * 32 + g * 8;
                    if (c < d) {
                        const floatx4 a0 = acc[b][2 * pp], a1 = acc[b][2 * pp + 1];
                        const half8 h = {(half_t)(a0[0] * inv), (half_t)(a0[1] * inv), (half_t)(a0[2] * inv), (half_t)(a0[3] * inv),
                                         (half_t)(a1[0] * inv), (half_t)(a1[1] * inv), (half_t)(a1[2] * inv), (half_t)(a1[3] * inv)};
                        *(half8*)(orow + c) = h;
                    }
                }
            }
            continue;
        }
        if (qrow[b] < p.seq_q) {
            half_t* orow = O + (int64_t)qrow[b] * p.o_rs;
#pragma unroll
            for (int i = 0; i < DTA; ++i) {
                const int c = i * 16 + g * 4;
                if (c < d) {
                    half4 h = {(half_t)(acc[b][i][0] * inv), (half_t)(acc[b][i][1] * inv), (half_t)(acc[b][i][2] * inv),
                               (half_t)(acc[b][i][3] * inv)};
                    *(half4*)(orow + c) = h;
                }
            }
        }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// attn_short_kernel: attention over <= 16 keys / queries (the motion modules' temporal self-attention over 16 frames,
// motion_module.py:270-336).  HBM-bound: per (pixel, head) it reads 3 x 16 x d halfs and writes 16 x d.  The generic
// kernel spends 18 KB of LDS per 1-wave workgroup on double-buffered K / V^T tiles, which caps a CU at 8 waves and left
// this case at ~2.3 TB/s.  Here one workgroup = one problem z (pixel), one WAVE per head:
//   * Q and K fragments of v_mfma_f32_16x16x32_f16 are read straight from global memory in fragment layout (lane =
//     (row, 8-half chunk): 16 rows x 64 B per load instruction), no LDS;
//   * S^T = K Q^T puts 4 consecutive keys of one query in each lane (softmax = in-lane + shuffles over xor 16, 32), which
//     is exactly the B-operand layout of v_mfma_f32_16x16x16_f16 for O^T = V^T P^T (16 keys = one k step, no padding);
//   * only V goes through LDS, transposed ([d][16 keys], 40-byte rows), in a per-wave slice: no workgroup barrier;
//   * all 8 heads of a token row are read by the same workgroup at the same time (full 128-byte lines from L1/L2).
// 1.6-6.4 KB of LDS and ~48 VGPRs per wave: 32 waves per CU.
#ifndef ATTN_SHORT_MAXT
#define ATTN_SHORT_MAXT 512
#endif
template <int D, bool BIAS = false>
__global__ __launch_bounds__(BIAS ? ATTN_SHORT_MAXT : 1024) void attn_short_kernel(insv2v_attention_desc p) {
    constexpr int KS = (D + 31) / 32;        // k steps of the QK^T contraction (head dim zero-padded in registers)
    constexpr int DT = (D + 15) / 16;        // 16-wide output column tiles
    constexpr int KCH = D / 8;               // 16-byte chunks per row
    constexpr int VT_LD = 20;                // halfs per transposed V row: 16 keys + 4 pad (40 B: conflict-free 8-byte reads)
    constexpr int V_ITERS = (8 * KCH + 63) / 64;  // (key pair, chunk) items per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: scalar base pointers
    const int g = lane >> 4, qc = lane & 15;
    half_t* vt = (half_t*)smem + head * (DT * 16 * VT_LD);
    // BIAS: the per-frame tables are the same for every problem, and as large as a problem's own q / k / v rows (16 x 3 x heads x D halfs) - too
    // large to stay in the 32 KiB L1.  Read per problem they made the L1 <- L2 traffic twice the algorithmic bytes and cost 31 % of the launch
    // (PMC + with / without A/B: profiles/r06_attn_short_bias.txt).  The workgroup is persistent over problems (grid = two per CU) and
    // every lane keeps ITS bias fragments in registers: 166 VGPRs at d = 160 (one workgroup of 8 waves per CU instead of two), still 25 %
    // faster per launch (295 -> 222 us at 5 760 problems; without the tables: 202 us).
    const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 qb[KS], kb[KS], vb[V_ITERS][2];
    if (BIAS) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk * 32 + g * 8;
            const bool okq = c < D && qc < p.seq_q, okk = c < D && qc < p.seq_k;
            const half8 tq = *(const half8*)((const half_t*)p.q_bias + (okq ? (int64_t)qc * p.bias_rs + head * D + c : 0));
            const half8 tk = *(const half8*)((const half_t*)p.k_bias + (okk ? (int64_t)qc * p.bias_rs + head * D + c : 0));
            qb[kk] = okq ? tq : z8; kb[kk] = okk ? tk : z8;
        }
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            const int e = lane + 64 * i;
            const int ch = e >> 3, kp = e & 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int key = 2 * kp + h;
                const bool okv = ch < KCH && key < p.seq_k;
                const half8 b = *(const half8*)((const half_t*)p.v_bias + (okv ? (int64_t)key * p.bias_rs + head * D + ch * 8 : 0));
                vb[i][h] = okv ? b : z8;
            }
        }
    }
#pragma unroll 1
  for (int z = blockIdx.x; z < p.batch; z += gridDim.x) {
    const half_t* Q = (const half_t*)p.q + (int64_t)(z / p.q_inner) * p.q_outer + (int64_t)(z % p.q_inner) * p.q_step + head * D;
    const half_t* K = (const half_t*)p.k + (int64_t)(z / p.kv_inner) * p.kv_outer + (int64_t)(z % p.kv_inner) * p.kv_step + head * D;
    const half_t* V = (const half_t*)p.v + (int64_t)(z / p.kv_inner) * p.kv_outer + (int64_t)(z % p.kv_inner) * p.kv_step + head * D;
    half_t* O = (half_t*)p.o + (int64_t)(z / p.o_inner) * p.o_outer + (int64_t)(z % p.o_inner) * p.o_step + head * D;

    const srd_t rV = make_srd(V);   // wave-uniform base (problem z, head = wave)
    // ---- all global loads up front: Q / K fragments and this lane's V pieces
    half8 qf[KS], kf[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int c = kk * 32 + g * 8;
        // unconditional loads (element 0 of the problem for lanes outside it) + select: a load inside a divergent `if` costs a
        // control-flow join, and hipcc waits vmcnt(0) at joins - three serial memory round trips instead of one
        const bool okq = c < D && qc < p.seq_q, okk = c < D && qc < p.seq_k;
        half8 tq = *(const half8*)(Q + (okq ? (int64_t)qc * p.q_rs + c : 0));
        half8 tk = *(const half8*)(K + (okk ? (int64_t)qc * p.k_rs + c : 0));
        if (BIAS) {   // per-row (frame) bias tables: the positional encoding pushed through the projections (zeros outside the problem)
            tq += qb[kk];
            tk += kb[kk];
        }
        qf[kk] = okq ? tq : z8; kf[kk] = okk ? tk : z8;
    }
    uint4 rv[V_ITERS][2];
#pragma unroll
    for (int i = 0; i < V_ITERS; ++i) {
        const int e = lane + 64 * i;
        const int ch = e >> 3, kp = e & 7;  // key pair fastest: 4-byte transposed LDS writes
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // buffer load with an out-of-range offset for pieces outside the problem (returns zeros): hipcc turns a select on the
            // address of a plain load back into a branch
            const int key = 2 * kp + h;
            const bool okv = ch < KCH && key < p.seq_k;
            uint4v t = __builtin_amdgcn_raw_buffer_load_b128(rV, okv ? (unsigned)((key * (int)p.v_rs + ch * 8) * 2) : OOB_OFFSET, 0, 0);
            if (BIAS) t = __builtin_bit_cast(uint4v, __builtin_bit_cast(half8, t) + vb[i][h]);   // (out-of-range pieces: zeros + zeros)
            rv[i][h] = make_uint4(t[0], t[1], t[2], t[3]);
        }
    }
    // keep the V requests HERE (hipcc otherwise sinks them behind the Q.K^T MFMAs, next to their LDS writes: a second serial round trip)
    __builtin_amdgcn_sched_barrier(0);
    // ---- S^T = K . Q^T : lane (g, qc) holds keys 4g .. 4g+3 of query qc
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kk], qf[kk], s, 0, 0, 0);
    // ---- V^T into this wave's LDS slice (rows beyond D of the last column tile are never read back into valid outputs)
#pragma unroll
    for (int i = 0; i < V_ITERS; ++i) {
        const int e = lane + 64 * i;
        const int ch = e >> 3, kp = e & 7;
        if (ch < KCH) {
            const unsigned short* h0 = (const unsigned short*)&rv[i][0];
            const unsigned short* h1 = (const unsigned short*)&rv[i][1];
#pragma unroll
            for (int j = 0; j < 8; ++j) *(unsigned*)(vt + (ch * 8 + j) * VT_LD + 2 * kp) = (unsigned)h0[j] | ((unsigned)h1[j] << 16);
        }
    }
    // ---- softmax over the keys of this lane's query
    const float c2 = p.scale * 1.4426950408889634f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (g * 4 + r >= p.seq_k) s[r] = -1.0e30f;
    float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    mx = col_max(mx);
    const float mc = -mx * c2;
    float e[4], l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        e[r] = __builtin_amdgcn_exp2f(fmaf(s[r], c2, mc));
        l += e[r];
    }
    l = col_sum(l);
    const float inv = 1.f / l;
    const half4 pf = {(half_t)e[0], (half_t)e[1], (half_t)e[2], (half_t)e[3]};
    __builtin_amdgcn_wave_barrier();  // the V^T slice is written and read by this wave only (LDS ops of a wave are in order)
    // ---- O^T = V^T . P^T, one 16-column tile at a time; lane holds O[query qc][dt*16 + 4g .. +4].  Every lane feeds the
    // MFMA (its A row is a head-dim index, not a query), only the store is limited to real queries.
    half_t* orow = O + (int64_t)qc * p.o_rs;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const half4 vf = *(const half4*)(vt + (dt * 16 + qc) * VT_LD + g * 4);
        floatx4 o = {0.f, 0.f, 0.f, 0.f};
        o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, o, 0, 0, 0);
        const int c = dt * 16 + g * 4;
        if (c < D && qc < p.seq_q) {
            const half4 h = {(half_t)(o[0] * inv), (half_t)(o[1] * inv), (half_t)(o[2] * inv), (half_t)(o[3] * inv)};
            *(half4*)(orow + c) = h;
        }
    }
    // the next problem's V^T writes follow this problem's reads of the same (wave-private) slice in program order
    __builtin_amdgcn_wave_barrier();
  }
}

template <int D>
static int launch_short(const insv2v_attention_desc& d, hipStream_t s) {
    constexpr int DT = (D + 15) / 16;
    const size_t lds = (size_t)d.heads * DT * 16 * 20 * sizeof(half_t);
    if (lds > 64 * 1024) return INSV2V_EUNSUPPORTED;  // beyond the default dynamic-LDS limit: the caller falls back to attn_kernel
    if (d.q_bias && d.heads * 64 > ATTN_SHORT_MAXT) return INSV2V_EUNSUPPORTED;   // the biased form is compiled for <= 8 heads (its register budget)
    // persistent over problems (see the kernel: the bias fragments are read once per wave): a few workgroups per CU, problems dealt round-robin
    // so that workgroups running side by side read neighbouring token rows.  INSV2V_ATTN_SHORT_WGS = workgroups per CU (0: one per problem).
    static const int wgs_per_cu = getenv("INSV2V_ATTN_SHORT_WGS") ? atoi(getenv("INSV2V_ATTN_SHORT_WGS")) : 2;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        cus = prop.multiProcessorCount;
    }
    const int grid = (wgs_per_cu > 0 && d.q_bias) ? (int)std::min<int64_t>(d.batch, (int64_t)cus * wgs_per_cu) : d.batch;
    if (d.q_bias) hipLaunchKernelGGL((attn_short_kernel<D, true>), dim3(grid), dim3(d.heads * 64), lds, s, d);
    else hipLaunchKernelGGL((attn_short_kernel<D>), dim3(grid), dim3(d.heads * 64), lds, s, d);
    return launch_status();
}

static int dispatch_short(const insv2v_attention_desc& d, hipStream_t s) {
    switch (d.head_dim) {
        case 16: return launch_short<16>(d, s);
        case 32: return launch_short<32>(d, s);
        case 40: return launch_short<40>(d, s);
        case 64: return launch_short<64>(d, s);
        case 80: return launch_short<80>(d, s);
        case 128: return launch_short<128>(d, s);
        case 160: return launch_short<160>(d, s);
    }
    return INSV2V_EUNSUPPORTED;
}

template <int D, int NW, int QB, bool FOLD = false, int KVT_ = 0, bool SINGLE = false, bool REP = false>
static int launch_attn(const insv2v_attention_desc& d, hipStream_t s) {
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KVT = KVT_ ? KVT_ : (NW >= 4 ? 64 : 32);
    constexpr size_t lds = (size_t)(SINGLE ? 1 : 2) * (KVT * (DP + 8) + DP * (KVT + 4)) * sizeof(half_t);
    if constexpr (!FOLD && !SINGLE && (D == 40 || D == 80) && NW >= 4) {   // the folded softmax (INSV2V_ATTN_FOLD=0: the round-3 form, for A/B)
        static const int fold = getenv("INSV2V_ATTN_FOLD") ? atoi(getenv("INSV2V_ATTN_FOLD")) : 1;
        if (fold) return launch_attn<D, NW, QB, true>(d, s);
    }
    if (SINGLE && d.seq_k > KVT) return INSV2V_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_kernel<D, NW, KVT, QB, FOLD, SINGLE, REP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    constexpr int rows = 16 * NW * QB;
    if (REP && (d.kv_inner <= 1 || d.kv_step != 0 || d.batch % d.kv_inner)) return INSV2V_EUNSUPPORTED;
    const int64_t nwg = (int64_t)((d.seq_q + rows - 1) / rows) * d.heads * (REP ? d.batch / d.kv_inner : d.batch);
    if (nwg > 0x7fffffff) return INSV2V_EUNSUPPORTED;
    hipLaunchKernelGGL((attn_kernel<D, NW, KVT, QB, FOLD, SINGLE, REP>), dim3((unsigned)nwg), dim3(NW * 64), lds, s, d);
    return launch_status();
}

// Head dims of the path: 40/80/160 (UNet levels, 8 heads), 16/32/64/128 (reduced-width test configs).
template <int NW, int QB>
static int dispatch_dp(const insv2v_attention_desc& d, hipStream_t s) {
    switch (d.head_dim) {
        case 16: return launch_attn<16, NW, QB>(d, s);
        case 32: return launch_attn<32, NW, QB>(d, s);
        case 40: return launch_attn<40, NW, QB>(d, s);
        case 64: return launch_attn<64, NW, QB>(d, s);
        case 80: return launch_attn<80, NW, QB>(d, s);
        case 128: return launch_attn<128, NW, QB>(d, s);
        case 160: return launch_attn<160, NW, QB>(d, s);
    }
    return INSV2V_EUNSUPPORTED;
}

extern "C" int insv2v_attention(const insv2v_attention_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    insv2v_attention_desc d = *dp;
    if (!d.q || !d.k || !d.v || !d.o) return INSV2V_EINVAL;
    if (d.head_dim <= 0 || (d.head_dim & 7) || d.head_dim > 160) return INSV2V_EINVAL;
    if (d.seq_q <= 0 || d.seq_k <= 0 || d.batch <= 0 || d.heads <= 0) return INSV2V_EINVAL;
    if ((d.q_rs & 7) || (d.k_rs & 7) || (d.v_rs & 7) || (d.o_rs & 3)) return INSV2V_EINVAL;
    if (d.q_inner <= 0) d.q_inner = 1;
    if (d.kv_inner <= 0) d.kv_inner = 1;
    if (d.o_inner <= 0) d.o_inner = 1;
    hipStream_t s = as_stream(stream);
    // <= 16 queries and keys (temporal attention over the frames): one wave per head, no K tile in LDS
    static const int short_on = getenv("INSV2V_ATTN_SHORT") ? atoi(getenv("INSV2V_ATTN_SHORT")) : 1;
    const bool biased = d.q_bias || d.k_bias || d.v_bias;
    if (biased && (!d.q_bias || !d.k_bias || !d.v_bias || (d.bias_rs & 7) || ((uintptr_t)d.q_bias & 15) || ((uintptr_t)d.k_bias & 15) || ((uintptr_t)d.v_bias & 15)))
        return INSV2V_EINVAL;
    if ((short_on || biased) && d.seq_q <= 16 && d.seq_k <= 16 && !d.causal && d.heads <= 16) {
        const int rc = dispatch_short(d, s);
        if (rc != INSV2V_EUNSUPPORTED || biased) return rc;
    }
    if (biased) return INSV2V_EUNSUPPORTED;   // only the <= 16-row kernel adds the tables: never silently drop them
    // 16 query rows per wave and query block: short query sequences (temporal, seq = frames) use
    // 1-wave workgroups; long ones 4 waves x 2 query blocks = 128 rows per workgroup.
    // the 96-token spatial self- / text cross-attention of the 8x12 level (d = 160): one workgroup of six waves per (frame, head), the
    // whole key sequence as one 96-key tile in a single LDS buffer (two workgroups per CU) - the generic form below runs it as two
    // half-filled 64-row workgroups with double-buffered 64-key tiles, one workgroup per CU (profiles/r04_attn_d160_single_tile.txt)
    static const int single_on = getenv("INSV2V_ATTN_SINGLE") ? atoi(getenv("INSV2V_ATTN_SINGLE")) : 1;
    // (three waves x two query blocks instead - every K / V fragment feeding two MFMAs - needs 256 VGPRs + spills and is slower: 401 vs 362 us)
    // (its 16-byte output stores need 16-byte aligned output rows)
    const bool o16 = !(d.o_rs & 7) && !((uintptr_t)d.o & 15) && !(d.o_outer & 7) && !(d.o_step & 7);
    if (single_on && o16 && d.head_dim == 160 && !d.causal && d.seq_q > 64 && d.seq_q <= 96 && d.seq_k <= 96) {
        // consecutive problems sharing one K / V (text cross-attention: the frames of a sample): one workgroup per (sample, head) keeps the tile
        static const int rep_on = getenv("INSV2V_ATTN_REP") ? atoi(getenv("INSV2V_ATTN_REP")) : 1;
        if (rep_on && d.kv_inner > 1 && d.kv_step == 0 && d.batch % d.kv_inner == 0) return launch_attn<160, 6, 1, false, 96, true, true>(d, s);
        return launch_attn<160, 6, 1, false, 96, true>(d, s);
    }
    if (d.seq_q <= 16) return dispatch_dp<1, 1>(d, s);
    if (d.seq_q <= 32) return dispatch_dp<2, 1>(d, s);
    if (d.seq_q >= 512 && d.seq_q % 256 == 0 && d.head_dim <= 96) return dispatch_dp<8, 2>(d, s);
    if (d.seq_q >= 128 && d.head_dim <= 96) return dispatch_dp<4, 2>(d, s);
    return dispatch_dp<4, 1>(d, s);
}
