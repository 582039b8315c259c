// insv2v_gemm: fp16 MFMA GEMM / implicit-GEMM 3x3 convolution with fused epilogue.
//
// Roofline: MFMA-bound by FLOPs, in practice bound by the L2->LDS operand stream (profiles/r01_gemm_decomposition.txt).
// One workgroup = WM x WN (x KG) waves computes a BM x BN output tile with v_mfma_f32_32x32x16_f16 (default: 8 waves as
// 4 x 2, 128 x 128 tile, 2 workgroups per CU); the MFMA "A" operand is the weight fragment (rows = output channels n) and
// the "B" operand the activation fragment (cols = tokens m), so each lane ends up with 4 CONSECUTIVE output channels of
// one token.
//
// K is consumed in 64-wide slices through an LDS ring filled by LDS-DMA
// (buffer_load_dwordx4 ... lds: HBM/L2 -> LDS without a VGPR round trip and without ds_write
// traffic, which measured as the limiter of a register-staged version).  Addresses are
// SRD base + per-lane 32-bit byte offset (loop invariant) + wave-uniform scalar slice offset, so
// the K loop carries almost no address arithmetic; out-of-range rows, conv zero padding and the
// K tail use an offset beyond the descriptor's range, for which the hardware returns zeros.  LDS rows are
// 128 B (64 halfs), unpadded, with the 16-byte chunk index XOR-swizzled by ((row>>1)&7):
// the DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address and
// again on the fragment read, which makes every ds_read_b128 of a 32-row fragment
// conflict-free.
// STAGES=2: slices 0,1 requested up front; then wait(slice t) -> barrier -> issue slice t+1 -> MFMA slice t  (2 workgroups/CU)
// STAGES>2: counted vmcnt keeps S-2 slices in flight across the barrier (kept for A/B measurement:
// LDS capacity, not prefetch depth, limits the bytes in flight, so deeper rings did not pay).
// Tile ids are rasterised XCD-aware in groups of 8 tile rows (8 x 8 patches per XCD) for L2 locality.
//
// In CONV3X3 mode the activation rows are gathered on the fly (tap-shifted pixels, zero
// fill at the border, optional nearest-x2 upsample and channel concat), so im2col, the
// upsampled tensor and torch.cat are never materialised in HBM.  Stride-1 convs whose output divides into
// 8x16 / 16x8 pixel patches go to conv_halo_kernel instead (input patch resident in LDS across the 9 taps).
//
// Epilogue (tile_epilogue): alpha, folded LayerNorm, bias, per-sample / per-frame row bias, SiLU / quick-GELU / GEGLU
// in registers; the fp32 tile is staged through the (now idle) LDS so global stores and residual loads are contiguous
// 16-byte chunks of whole output rows.  Bias / column sums / token statistics are parked in LDS before the K loop.
#include "common.h"
#include <algorithm>
#include "gemm_dma.h"
#include <type_traits>
#include <cstdlib>

typedef unsigned uint4r __attribute__((ext_vector_type(4)));  // 128-bit register operand of inline asm

struct RowInfo {  // per-thread metadata of one staged activation row
    int base;      // linear: m ; conv: nb*IH*IW (pixel index of the image's first pixel)
    int oh, ow;    // conv only (already multiplied by stride, minus pad)
    bool valid;
};

// Debug-only phase timer (tools/gemm_phase_prof.py builds a separate library with -DINSV2V_GEMM_PROF; the shipped
// library never contains it): thread 0 accumulates 100 MHz wall-clock ticks per phase of every interior workgroup.
#ifdef INSV2V_GEMM_PROF
__device__ unsigned long long g_prof[8];
#define PROF_MARK(i) do { if (tid == 0) prof_t[i] = wall_clock64(); } while (0)
#else
#define PROF_MARK(i) do { } while (0)
#endif

// Sum over the aligned group of W consecutive lanes (W = 4, 8, 16) on the VALU: quad_perm, quad_perm, row_half_mirror, row_mirror
template <int W>
__device__ __forceinline__ float lane_group_sum(float v) {
    static_assert(W == 4 || W == 8 || W == 16, "a DPP row is 16 lanes");
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});                 // lanes 1,0,3,2
    v += dpp(v, std::integral_constant<int, 0x4E>{});                 // lanes 2,3,0,1
    if (W >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
    if (W >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

// Whether a workgroup's output takes the vectorised fp16 path of tile_epilogue (whole-launch property)
__device__ __forceinline__ bool fast_output_ok(const insv2v_gemm_desc& p, const char* Cb, const half_t* Rp, int oN) {
    const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)Cb & (p.c_fp32 ? 31 : 15)) == 0) &&
                        (!Rp || (((p.ldr & 7) == 0) && (((uintptr_t)Rp & 15) == 0)));
    const bool res32 = !Rp || (int64_t)p.M * p.ldr * 2 < ((int64_t)1 << 31);
    return vec_ok && !p.c_fp32 && (oN & 7) == 0 && res32;
}

// Shared tile epilogue.  acc holds the wave's MI x NI fragments; grow(r) maps tile-local row r to the global output
// row (token / pixel index), rows_full says every row of the tile exists.  sBias / sCs / sStat are the LDS copies of
// bias, folded-LayerNorm column sums and per-row (mean, rstd) made by the caller before its K loop.
template <int WM, int WN, int MI, int NI, int KG, bool HAS_LN, class RowMap>
__device__ __forceinline__ void tile_epilogue(const insv2v_gemm_desc& p, floatx16 (&acc)[NI][MI], char* smem, const float* sBias,
                                              const float* sCs, const float2* sStat, int tid, int wm, int wn, int kg, int bn0,
                                              int z, int zs, bool rows_full, RowMap grow
#ifdef INSV2V_GEMM_PROF
                                              , unsigned long long* prof_t
#endif
) {
    constexpr int NWV = WM * WN * KG, NT = NWV * 64;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    const int lane = tid & 63;
    const bool geglu = p.act == INSV2V_ACT_GEGLU;
    const int oN = geglu ? (p.N >> 1) : p.N;          // output columns
    const int on0 = geglu ? (bn0 >> 1) : bn0;         // first output column of this tile
    constexpr int CLD = BN + 4;                        // floats per staged row (fp32: one rounding, after the residual add)
    float* sC = (float*)smem;
    char* Cb = (char*)p.c + ((int64_t)z * p.c_bs + (int64_t)zs * p.M * p.ldc) * (p.c_fp32 ? 4 : 2);  // zs > 0 only for split-K partial slabs
    const half_t* Rp = p.residual ? (const half_t*)p.residual + z * p.r_bs : nullptr;
    if (KG > 1) {
        // partial sums of the K groups meet in the staging buffer: every lane of group g > 0 parks its
        // accumulators at the positions the SAME lane of group 0 owns, so no ordering beyond the barrier is needed
        for (int g = 1; g < KG; ++g) {
            if (kg == g) {
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(float4*)(sC + (wm * MI * 32 + j * 32 + (lane & 31)) * CLD + wn * NI * 32 + i * 32 + 8 * q + 4 * (lane >> 5)) =
                                make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 t = *(const float4*)(sC + (wm * MI * 32 + j * 32 + (lane & 31)) * CLD + wn * NI * 32 + i * 32 + 8 * q + 4 * (lane >> 5));
                            acc[i][j][4 * q] += t.x; acc[i][j][4 * q + 1] += t.y; acc[i][j][4 * q + 2] += t.z; acc[i][j][4 * q + 3] += t.w;
                        }
            }
            if (g + 1 < KG) __syncthreads();
        }
    }
    typedef unsigned uint4e __attribute__((ext_vector_type(4)));
    const srd_t rRB = make_srd(p.row_bias ? (const void*)p.row_bias : (const void*)p.w);
    // whole-workgroup choice: 16-byte row-bias loads need N % 4 == 0, an aligned table and offsets below 2^31
    const bool rb_vec = p.row_bias && (p.N & 3) == 0 && (p.ld_rb & 3) == 0 && (((uintptr_t)p.row_bias & 15) == 0);
    if (KG == 1 || kg == 0) {
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int ml = wm * MI * 32 + j * 32 + (lane & 31);
        const int m = grow(ml);
        // Row bias (per-sample time embedding / per-frame positional-encoding table): 16-byte BUFFER loads whose offset is out of
        // range (-> zeros) where no bias applies.  A plain load inside `if (rb)` is a divergent branch per fragment quarter, and hipcc
        // waits vmcnt(0) at every join: 8 serial L2 round trips per thread in this epilogue (profiles/r02_epilogue_loads.txt).
        const float* rb = nullptr;
        unsigned rb_off = OOB_OFFSET;   // byte offset of this row's bias row; vector path only when N % 4 == 0
        if (p.row_bias && m < p.M) {
            int grp = m / p.rows_per_group;
            if (p.rb_mod > 0) grp %= p.rb_mod;
            rb = p.row_bias + (int64_t)grp * p.ld_rb;
            rb_off = (unsigned)(grp * (int)p.ld_rb * 4);
        }
        // all of this row's bias quarters at once (NI x 4 requests in flight; one per fragment quarter in the loop below was one
        // L2 round trip each)
        floatx4 rbv[NI][4];
        if (rb_vec) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = bn0 + wn * NI * 32 + i * 32 + 8 * q + 4 * (lane >> 5);
                    rbv[i][q] = __builtin_bit_cast(floatx4, (uint4e)__builtin_amdgcn_raw_buffer_load_b128(rRB, (rb && n + 3 < p.N) ? rb_off + (unsigned)(n * 4) : OOB_OFFSET, 0, 0));
                }
        }
        // folded LayerNorm: v = rstd*(alpha*acc - mean*col_sum[n]) (+ bias terms)
        float ln_m = 0.f, ln_r = 1.f;
        if (HAS_LN) { const float2 st = sStat[ml]; ln_m = st.x; ln_r = st.y; }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (geglu && (i & 1)) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * NI * 32 + i * 32 + 8 * q + 4 * (lane >> 5);  // tile-local n of v[0]
                const int n = bn0 + nl;
                float v[4];
                const float4 bt = *(const float4*)(sBias + nl), ct = HAS_LN ? *(const float4*)(sCs + nl) : make_float4(0.f, 0.f, 0.f, 0.f);
                float bsum[4] = {bt.x, bt.y, bt.z, bt.w};
                const float cs[4] = {ct.x, ct.y, ct.z, ct.w};
                if (rb_vec) {
                    const floatx4 t = rbv[i][q];
                    bsum[0] += t[0]; bsum[1] += t[1]; bsum[2] += t[2]; bsum[3] += t[3];
                } else if (rb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) bsum[e] += rb[n + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ln_r * (acc[i][j][4 * q + e] * p.alpha - ln_m * cs[e]) + bsum[e];
                int onl = nl;  // tile-local output column
                if (geglu) {
                    const float4 gbt = *(const float4*)(sBias + nl + 32), gct = HAS_LN ? *(const float4*)(sCs + nl + 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float gb[4] = {gbt.x, gbt.y, gbt.z, gbt.w};
                    const float gcs[4] = {gct.x, gct.y, gct.z, gct.w};
                    if (rb_vec) {
                        const floatx4 t = rbv[(i + 1) % NI][q];   // the gate block is the next 32 channels: its own quarter q
                        gb[0] += t[0]; gb[1] += t[1]; gb[2] += t[2]; gb[3] += t[3];
                    } else if (rb && n + 35 < p.N) {
                        const float4 t = *(const float4*)(rb + n + 32);
                        gb[0] += t.x; gb[1] += t.y; gb[2] += t.z; gb[3] += t.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] *= gelu_erf_f(ln_r * (acc[(i + 1) % NI][j][4 * q + e] * p.alpha - ln_m * gcs[e]) + gb[e]);
                    onl = (nl >> 6) * 32 + (nl & 31);
                } else if (p.act == INSV2V_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                } else if (p.act == INSV2V_ACT_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
                } else if (p.act >= INSV2V_ACT_RELU) {   // RAFT's small GEMMs (ReLU / sigmoid / tanh, oracle/raft.py)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_raft_f(v[e], p.act);
                }
                *(float4*)(sC + ml * CLD + onl) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    }
    PROF_MARK(3);
    __syncthreads();
    PROF_MARK(4);
    const int OW8 = (geglu ? BN / 2 : BN) / 8;  // 16-byte chunks per staged output row
    const bool vec_ok = ((p.ldc & 7) == 0) && (((uintptr_t)Cb & (p.c_fp32 ? 31 : 15)) == 0) &&
                        (!Rp || (((p.ldr & 7) == 0) && (((uintptr_t)Rp & 15) == 0)));
    // fast path: fp16 tile whose valid columns are whole 16-byte chunks -> fully unrolled, constant divisors, ALL residual loads
    // issued up front.  Edge tiles (last rows, N = 320 with 128-wide tiles: every third tile) take it too: chunks outside the
    // problem load with an out-of-range buffer offset and skip the store.  (They used to fall to the generic loop below, whose
    // per-chunk residual load + wait made them - and with them the whole launch - 4-8 memory round trips longer.)
    if (fast_output_ok(p, Cb, Rp, oN)) {
        const srd_t rRes = make_srd(Rp ? (const void*)Rp : (const void*)p.w);
        auto copy_rows = [&](auto w8_tag) {
            constexpr int W8 = decltype(w8_tag)::value;
            constexpr int ITERS = (BM * W8) / NT;
            static_assert((BM * W8) % NT == 0, "staged tile must divide evenly over the workgroup");
            half8 rv[ITERS];
            bool ok[ITERS];
            int mrow[ITERS];
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = tid + it * NT, row = idx / W8, ch = idx % W8;
                mrow[it] = grow(row);
                ok[it] = (rows_full || mrow[it] < p.M) && on0 + ch * 8 + 8 <= oN;
                if (Rp) rv[it] = __builtin_bit_cast(half8, (uint4e)__builtin_amdgcn_raw_buffer_load_b128(
                            rRes, ok[it] ? (unsigned)((mrow[it] * (int)p.ldr + on0 + ch * 8) * 2) : OOB_OFFSET, 0, 0));
            }
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = tid + it * NT, row = idx / W8, ch = idx % W8;
                const float4 f0 = *(const float4*)(sC + row * CLD + ch * 8);
                const float4 f1 = *(const float4*)(sC + row * CLD + ch * 8 + 4);
                const float fv[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)(Rp ? fv[e] + (float)rv[it][e] : fv[e]);
                if (ok[it]) *(half8*)((half_t*)Cb + (int64_t)mrow[it] * p.ldc + on0 + ch * 8) = hv;
                if constexpr (W8 == 4 || W8 == 8 || W8 == 16) {
                    // LayerNorm statistics of the NEXT op from the values just stored (fp16-rounded, after the residual add): the W8
                    // lanes holding one row's chunks reduce on the VALU; one (sum, sum of squares) pair per row and column tile
                    if (p.stats_out) {
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float x = ok[it] ? (float)hv[e] : 0.f; s1 += x; s2 = fmaf(x, x, s2); }
                        s1 = lane_group_sum<W8>(s1);
                        s2 = lane_group_sum<W8>(s2);
                        if (ch == 0 && (rows_full || mrow[it] < p.M))
                            *(float2*)(p.stats_out + ((int64_t)(bn0 / BN) * p.M + mrow[it]) * 2) = make_float2(s1, s2);
                    }
                }
            }
        };
        if (geglu) copy_rows(std::integral_constant<int, BN / 16>{});
        else copy_rows(std::integral_constant<int, BN / 8>{});
#ifdef INSV2V_GEMM_PROF
        wait_vmcnt<0>();
        PROF_MARK(5);
        if (tid == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(&g_prof[i], prof_t[i + 1] - prof_t[i]);
            atomicAdd(&g_prof[5], 1ull);
        }
#endif
        return;
    }
    for (int idx = tid; idx < BM * OW8; idx += NT) {
        const int row = idx / OW8, ch = idx - row * OW8;
        const int m = grow(row), on = on0 + ch * 8;
        if (m >= p.M || on >= oN) continue;
        const float4 f0 = *(const float4*)(sC + row * CLD + ch * 8);
        const float4 f1 = *(const float4*)(sC + row * CLD + ch * 8 + 4);
        float fv[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        if (p.c_fp32) {  // conv_out / VAE moments / time embedding / split-K partial slabs
            float* dst32 = (float*)Cb + (int64_t)m * p.ldc + on;
            if (Rp) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (on + e < oN) fv[e] += (float)Rp[(int64_t)m * p.ldr + on + e];
            }
            if (vec_ok && on + 7 < oN) {
                *(float4*)dst32 = make_float4(fv[0], fv[1], fv[2], fv[3]);
                *(float4*)(dst32 + 4) = make_float4(fv[4], fv[5], fv[6], fv[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (on + e < oN) dst32[e] = fv[e];
            }
            continue;
        }
        half_t* dst = (half_t*)Cb + (int64_t)m * p.ldc + on;
        if (vec_ok && on + 7 < oN) {
            half8 hv;
            if (Rp) {
                const half8 rv = *(const half8*)(Rp + (int64_t)m * p.ldr + on);
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)(fv[e] + (float)rv[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)fv[e];
            }
            *(half8*)dst = hv;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (on + e < oN) dst[e] = (half_t)(fv[e] + (Rp ? (float)Rp[(int64_t)m * p.ldr + on + e] : 0.f));
        }
    }
}

// WM x WN waves, each owning MI x NI fragments of 32x32: BM = WM*MI*32 tokens, BN = WN*NI*32 channels.
// KG > 1: KG groups of WM x WN waves share the tile; group g multiplies the g-th 64/KG-wide part of every K slice
// (same DMA ring, 1/KG of the LDS fragment reads per MFMA) and the partial sums meet in LDS before the epilogue.
template <int WM, int WN, int MI, int NI, int MODE, int STAGES, int KG = 1>
__global__ __launch_bounds__(WM * WN * KG * 64) void gemm_kernel(insv2v_gemm_desc p) {
    constexpr int NWV = WM * WN * KG;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int LD = BK;                          // halfs per LDS row (128 B, unpadded, XOR-swizzled chunks)
    constexpr int RPP = 8 * NWV;                    // tile rows filled by one DMA instruction of every wave
    constexpr int RA = BM / RPP, RW = BN / RPP;     // DMA instructions per wave per slice (activations / weights)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sA = (half_t*)smem;                     // [STAGES][BM][LD]
    half_t* sW = sA + STAGES * BM * LD;             // [STAGES][BN][LD]
    constexpr int RING_B = STAGES * (BM + BN) * BK * 2, STAGE_B = BM * (BN + 4) * 4;
    float* sBias = (float*)(smem + (RING_B > STAGE_B ? RING_B : STAGE_B));  // [BN] bias, [BN] col_sum, [BM] (mean, rstd)
    float* sCs = sBias + BN;
    float2* sStat = (float2*)(sCs + BN);

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef INSV2V_GEMM_PROF
    unsigned long long prof_t[6];
#endif
    PROF_MARK(0);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wid / (WM * WN), wl = wid % (WM * WN);
    const int wm = wl / WN, wn = wl % WN;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // Rasterisation: after the XCD remap each XCD runs a contiguous run of ~64 tile ids at a time (32 CUs x 2
    // workgroups).  Ids walk groups of GROUP_M tile rows column by column, so such a run is a ~8 x 8 patch that
    // needs 8 + 8 operand panels from beyond its L2 instead of 1 + 64 (what row-major order costs when N is wide).
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int gidx = bid / per_group, first_m = gidx * GROUP_M;
    const int gsz = min(GROUP_M, tiles_m - first_m), rin = bid - gidx * per_group;
    const int tn = rin / gsz, tm = first_m + rin - tn * gsz;
    const int bm0 = tm * BM, bn0 = tn * BN;
    const int z = blockIdx.y;

    const half_t* A = (const half_t*)p.a + z * p.a_bs;
    const half_t* A2 = p.a2 ? (const half_t*)p.a2 + z * p.a_bs : A;
    const half_t* Wp = (const half_t*)p.w + z * p.w_bs;
    const srd_t rA = make_srd(A), rA2 = make_srd(A2), rW = make_srd(Wp);

    // staging map: DMA instruction i of wave `wid` fills the 8-row group (i*NWV + wid):
    // row = group*8 + lane/8, LDS chunk slot lane%8, i.e. exactly lane-linear 1 KiB per instruction;
    // the lane fetches the LOGICAL chunk slot ^ ((row>>1)&7) (swizzle on the source side).
    const int crow = wid * 8 + (lane >> 3);
    const int cslot = lane & 7;

    RowInfo ri[RA];
    int achunk[RA];
    unsigned aoff1[RA], aoff2[RA];  // per-lane byte offsets into source 1 / 2 (linear: final; conv: un-shifted tap)
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int row = crow + RPP * i;
        achunk[i] = cslot ^ ((row >> 1) & 7);
        const int m = bm0 + row;
        ri[i].valid = m < p.M;
        if (MODE == INSV2V_MODE_LINEAR) {
            ri[i].base = m;
            ri[i].oh = ri[i].ow = 0;
            aoff1[i] = ri[i].valid ? (unsigned)(((int64_t)m * p.lda + achunk[i] * 8) * 2) : OOB_OFFSET;
            aoff2[i] = ri[i].valid ? (unsigned)(((int64_t)m * p.lda2 + achunk[i] * 8) * 2) : OOB_OFFSET;
        } else {
            int mm = ri[i].valid ? m : 0;
            int ow = mm % p.OW, t = mm / p.OW;
            int oh = t % p.OH, nb = t / p.OH;
            ri[i].base = nb * p.IH * p.IW;
            ri[i].oh = oh * p.stride - p.pad_t;
            ri[i].ow = ow * p.stride - p.pad_l;
            const int64_t pix = (int64_t)ri[i].base + (int64_t)ri[i].oh * p.IW + ri[i].ow;  // may be < 0 at the border
            aoff1[i] = (unsigned)((pix * p.lda + achunk[i] * 8) * 2);
            aoff2[i] = (unsigned)((pix * p.lda2 + achunk[i] * 8) * 2);
        }
    }
    int wchunk[RW];
    unsigned woff[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int row = crow + RPP * i;
        wchunk[i] = cslot ^ ((row >> 1) & 7);
        const int n = bn0 + row;
        woff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + wchunk[i] * 8) * 2) : OOB_OFFSET;
    }

    // split-K: workgroup blockIdx.z consumes slices [kt0, kt0 + nk) and writes an fp32 partial tile
    const int nk_total = (p.K + BK - 1) / BK;
    const int zs = blockIdx.z, nsplit = p.split_k > 1 ? p.split_k : 1;

    // (Requesting this thread's residual pieces here, before the K loop, was tried in round 2: correct, +28 VGPRs, and no faster -
    // out-proj 73 728 x 320 x 320 46 vs 43-44 us, 3-stream UNet step unchanged - the residual round trip is not what a short-K tile
    // waits for.  tools/gemm_phase_prof.py shows the phases.)
    const int nk_per = (nk_total + nsplit - 1) / nsplit;
    const int kt0 = zs * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const int IHu = p.upsample ? p.IH * 2 : p.IH, IWu = p.upsample ? p.IW * 2 : p.IW;

    // Slices are requested in increasing order: everything slice-dependent is a wave-uniform scalar
    // kept in a cursor (no division, no 64-bit arithmetic in the loop).
    struct Cursor {
        int k0, kh, kw, ci0;
    } cur_k = {kt0 * BK, 0, 0, 0};
    if (MODE != INSV2V_MODE_LINEAR) {
        const int tap = cur_k.k0 / p.Cin;  // once per workgroup
        cur_k.ci0 = cur_k.k0 - tap * p.Cin;
        cur_k.kh = tap / 3;
        cur_k.kw = tap - cur_k.kh * 3;
    }
    auto advance = [&]() {
        cur_k.k0 += BK;
        if (MODE != INSV2V_MODE_LINEAR) {
            cur_k.ci0 += BK;
            if (cur_k.ci0 >= p.Cin) {
                cur_k.ci0 = 0;
                if (++cur_k.kw == 3) { cur_k.kw = 0; ++cur_k.kh; }
            }
        }
    };
    auto issue_slice = [&](int buf) {
        char* a = (char*)(sA + buf * BM * LD) + wid * 1024;
        char* w = (char*)(sW + buf * BN * LD) + wid * 1024;
        const bool ktail = cur_k.k0 + BK > p.K;  // only the last slice of a K that is not a multiple of 64
        if (MODE == INSV2V_MODE_LINEAR) {
            const bool second = p.k_split > 0 && cur_k.k0 >= p.k_split;
            const int soff = (second ? cur_k.k0 - p.k_split : cur_k.k0) * 2;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                unsigned v = second ? aoff2[i] : aoff1[i];
                if (ktail && cur_k.k0 + achunk[i] * 8 >= p.K) v = OOB_OFFSET;
                dma16(second ? rA2 : rA, v, soff, a + i * (NWV * 1024));
            }
        } else {
            const bool second = p.k_split > 0 && cur_k.ci0 >= p.k_split;
            const int ld = (int)(second ? p.lda2 : p.lda);
            const int cl = second ? cur_k.ci0 - p.k_split : cur_k.ci0;
            const int tapoff = ((cur_k.kh * p.IW + cur_k.kw) * ld + cl) * 2;  // uniform byte shift of this tap / channel block
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ih = ri[i].oh + cur_k.kh, iw = ri[i].ow + cur_k.kw;
                const bool ok = ri[i].valid && (unsigned)ih < (unsigned)IHu && (unsigned)iw < (unsigned)IWu;
                unsigned v;
                if (p.upsample)  // nearest x2: source pixel (ih>>1, iw>>1); not separable per tap
                    v = (unsigned)((((ri[i].base + (ih >> 1) * p.IW + (iw >> 1)) * ld) + cl + achunk[i] * 8) * 2);
                else
                    v = (second ? aoff2[i] : aoff1[i]) + (unsigned)tapoff;
                dma16(second ? rA2 : rA, ok ? v : OOB_OFFSET, 0, a + i * (NWV * 1024));
            }
        }
        const int wsoff = cur_k.k0 * 2;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            unsigned v = woff[i];
            if (ktail && cur_k.k0 + wchunk[i] * 8 >= p.K) v = OOB_OFFSET;
            dma16(rW, v, wsoff, w + i * (NWV * 1024));
        }
        advance();
    };

    floatx16 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    auto compute = [&](int buf) {
        const half_t* a = sA + buf * BM * LD + (wm * MI * 32 + frow) * LD;
        const half_t* w = sW + buf * BN * LD + (wn * NI * 32 + frow) * LD;
        // swizzle term of this lane's fragment rows: rows differ by multiples of 32 across j/i, so
        // ((row>>1)&7) depends on frow only
        const int sw = (frow >> 1) & 7;
#pragma unroll
        for (int kq = 0; kq < BK / 16 / KG; ++kq) {
            const int kk = kg * (BK / 16 / KG) + kq;
            const int c = ((kk * 2 + fhalf) ^ sw) * 8;
            half8 fa[MI], fw[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) fa[j] = *(const half8*)(a + j * 32 * LD + c);
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = *(const half8*)(w + i * 32 * LD + c);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    };

    // S-stage ring: slices kt+1 .. kt+S-2 stay in flight (counted vmcnt) while slice kt is consumed;
    // slice kt+S-1 is issued right after the barrier into the buffer slice kt-1 just vacated.
    // Epilogue vectors (bias, folded-LayerNorm column sums, per-token mean/rstd) are requested BEFORE the
    // first slices and parked in a small LDS area behind the ring/staging buffer, so the epilogue never
    // waits on a global load.  Out-of-range entries are neutral (0 / rstd 1).
    const bool ln = p.row_stats != nullptr;
    float pre_b = 0.f, pre_c = 0.f;
    float2 pre_s = make_float2(0.f, 1.f);
    if (tid < BN && bn0 + tid < p.N) {
        if (p.bias) pre_b = p.bias[bn0 + tid];
        if (ln) pre_c = p.col_sum[bn0 + tid];
    }
    if (tid < BM && ln && bm0 + tid < p.M) {
        if (p.stats_parts > 0) {  // partial sums written by the producing GEMM's epilogue, one pair per column tile
            float s1 = 0.f, s2 = 0.f;
            for (int j = 0; j < p.stats_parts; ++j) {
                const float2 t = ((const float2*)p.row_stats)[(int64_t)j * p.M + bm0 + tid];
                s1 += t.x; s2 += t.y;
            }
            const float mean = s1 / p.K;
            pre_s = make_float2(mean, rsqrtf(fmaxf(s2 / p.K - mean * mean, 0.f) + p.ln_eps));
        } else {
            pre_s = ((const float2*)p.row_stats)[bm0 + tid];
        }
    }

    constexpr int LPT = RA + RW;  // LDS-DMA instructions per wave per slice
    // The whole ring is requested up front (slices 0 .. STAGES-1): the first two slices' memory latencies overlap instead of
    // the second one starting only after the first has landed - one latency saved per tile, which matters for the many
    // short-K (5-20 slice) tiles.  From iteration 1 on, slice kt+STAGES-1 goes into the buffer slice kt-1 just vacated.
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (s < nk) issue_slice(s);
    if (tid < BN) { sBias[tid] = pre_b; sCs[tid] = pre_c; }
    if (tid < BM) sStat[tid] = pre_s;
    int cur = 0, nxt = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // younger slices allowed to stay in flight while slice kt is awaited
        const int behind = kt == 0 ? min(STAGES - 1, nk - 1) : min(STAGES - 2, nk - 1 - kt);
        if (behind <= 0) wait_vmcnt<0>();
        else if (behind == 1) wait_vmcnt<LPT>();
        else if (behind == 2) wait_vmcnt<2 * LPT>();
        else wait_vmcnt<3 * LPT>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef INSV2V_GEMM_PROF
        if (kt == 0) PROF_MARK(1);
#endif
        if (kt > 0 && kt + STAGES - 1 < nk) issue_slice(nxt);
        compute(cur);
        cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();
    PROF_MARK(2);

    // ---- epilogue ---------------------------------------------------------------------------------
    tile_epilogue<WM, WN, MI, NI, KG, true>(p, acc, smem, sBias, sCs, sStat, tid, wm, wn, kg, bn0, z, zs, bm0 + BM <= p.M,
                                            [&](int r) { return bm0 + r; }
#ifdef INSV2V_GEMM_PROF
                                            , prof_t
#endif
    );
}


// ------------------------------------------------------------------------------------------------------------
// conv_halo_kernel: stride-1 3x3 convolution whose output tile is a TH x TW pixel PATCH of one image (8x16 or 16x8 =
// 128 rows).  The 9 taps of a 64-channel block all read the same (TH+2)x(TW+2) input patch, so that patch is brought
// into LDS ONCE per channel block (<= 180 pixel rows x 128 B) and every tap's activation fragments are read from it at
// shifted rows; only the weight slices (BN x 64) stream per tap.  Compared with gathering a fresh 128-row activation
// slice per tap this cuts the L2->LDS operand stream of the K loop from 9 x 32 KB to 9 x 16 KB + 23 KB per channel
// block (0.58x) - the stream is what bounds the implicit-GEMM kernel (profiles/r01_gemm_decomposition.txt).
// K order is channel-block major, tap minor.  Nearest-x2 upsampling reads a (TH/2+2)x(TW/2+2) source patch; channel
// concat picks the source per channel block; zero padding = out-of-range DMA offsets.  2 workgroups / CU.
// GN = true: the input is the RAW tensor and p.gn_ab holds per-(sample, channel) (scale, shift): every thread applies
// y = act(x*scale + shift) to the 16 bytes of the patch it requested itself, in LDS, one tap after the piece was requested
// (the per-tap vmcnt(0) + barrier has retired it by then); pixels outside the image stay zero.  The (scale, shift) pairs
// of a channel block (512 B) arrive by one LDS-DMA piece with the patch.  The LDS reads / writes of this pass are
// inline asm: a compiler-visible LDS access behind an LDS-DMA gets an s_waitcnt vmcnt(0) (profiles/r02_gemm_debug.md).
// WS = weight slices in flight (LDS slots): 2 = the slice of the next tap is requested while this tap computes (vmcnt(0) per tap);
// 3 = two taps ahead with a counted wait (experiment, tile 103: ONE 16-wave workgroup per CU owning a 256-pixel patch).
template <int WM, int WN, int MI, int NI, int KG, bool GN, int WS = 2>
__global__ __launch_bounds__(WM * WN * KG * 64) void conv_halo_kernel(insv2v_gemm_desc p, int tw_shift) {
    constexpr int NWV = WM * WN * KG;
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    static_assert(BM == 128 || BM == 256, "the pixel patch is 128 (8x16 / 16x8) or 256 (16x16 / 32x8) output pixels");
    static_assert(WS == 2 || (WS == 3 && !GN && KG == 1), "the deep weight ring has no fused-GroupNorm / K-group form");
    constexpr int LD = BK;
    constexpr int RPP = 8 * NWV, RW = BN / RPP;
    constexpr int HPIECES = BM == 128 ? 23 : 43;      // 1 KiB DMA pieces (8 pixel rows each) covering <= 180 (<= 340) patch rows
    constexpr int HP = (HPIECES + NWV - 1) / NWV;     // pieces per wave
    constexpr int HALO_B = HPIECES * 1024;
    constexpr int KQ = BK / 16 / KG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sH = smem;                                  // [2][HALO_B] input patches (double buffered over channel blocks)
    half_t* sW = (half_t*)(smem + 2 * HALO_B);        // [WS][BN][LD] weight slices
    constexpr int RING_B = 2 * HALO_B + WS * BN * BK * 2, STAGE_B = BM * (BN + 4) * 4;
    float* sBias = (float*)(smem + (RING_B > STAGE_B ? RING_B : STAGE_B));
    constexpr int AB_OFF = (RING_B > STAGE_B ? RING_B : STAGE_B) + BN * 4;  // [2][64][2] floats (GN only)

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef INSV2V_GEMM_PROF
    unsigned long long prof_t[6];
#endif
    PROF_MARK(0);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wid / (WM * WN), wl = wid % (WM * WN);
    // Round 3: N = 320 leaves the third 128-column tile half empty.  Waves are dealt to the SIMDs round robin (wid & 3), so the
    // column half of a wave is wl / WM, not wl % WN: the waves whose 64 columns lie beyond N then sit one per SIMD, skip their
    // fragment reads + MFMAs (they still issue their share of the DMA and keep the barriers), and every SIMD of the CU gets that
    // matrix-pipe / LDS time back for the other workgroup (bit 8 of tw_shift = the old mapping without the skip, for A/B).
    const bool legacy_map = (tw_shift >> 8) & 1;
    tw_shift &= 255;
    const int wm = legacy_map ? wl / WN : wl % WM, wn = legacy_map ? wl % WN : wl / WM;
    const int TW = 1 << tw_shift, TH = BM >> tw_shift;
    const int tiles_x = p.OW >> tw_shift, tiles_y = p.OH / TH, tiles_img = tiles_x * tiles_y;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = p.NB * tiles_img;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int gidx = bid / per_group, first_m = gidx * GROUP_M;
    const int gsz = min(GROUP_M, tiles_m - first_m), rin = bid - gidx * per_group;
    const int tn = rin / gsz, tm = first_m + rin - tn * gsz;
    const int bn0 = tn * BN;
    const int nb = tm / tiles_img, trem = tm - nb * tiles_img;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int up = p.upsample ? 1 : 0;
    // source patch: rows sy0 .. sy0+PH-1, cols sx0 .. sx0+PW-1 (may stick out of the image: zero padding)
    const int sy0 = (oy0 - 1) >> up, sx0 = (ox0 - 1) >> up;
    const int PH = (TH >> up) + 2, PW = (TW >> up) + 2, HROWS = PH * PW;

    const half_t* A = (const half_t*)p.a;
    const half_t* A2 = p.a2 ? (const half_t*)p.a2 : A;
    const srd_t rA = make_srd(A), rA2 = make_srd(A2), rW = make_srd(p.w);

    // LDS swizzle of the patch: the 16-byte chunk index of pixel (hy, hx) is XORed with a key chosen so that the 16 lanes
    // of every ds_read_b128 group land on 16 distinct bank slots.  A group holds fragment rows {0-3, 12-15, 20-27} (or
    // {4-11, 16-19, 28-31}); in an 8x16 patch these are 16 different hx (mod 16) spread over two patch rows -> key from hx
    // alone; in a 16x8 patch they are 4 patch rows x 4 pixels -> the row parity supplies the fourth bit.  PW is even, so the
    // 128-byte half of the 256-byte bank row is hx & 1.  (Keyed on the linear row index the reads were 24-39 % conflicts.)
    auto patch_key = [&](int hy, int hx) { return ((hx >> 1) + (tw_shift == 3 ? ((hy & 1) << 2) : 0)) & 7; };
    // patch staging map: piece q = i*NWV + wid holds patch rows q*8 .. q*8+7 (row = pixel, 128 B = one channel block)
    unsigned hoff1[HP], hoff2[HP];
    int hchunk[HP];  // GN: logical 8-channel chunk of this lane's 16 bytes of piece i, or -1 for zero padding
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int r = (i * NWV + wid) * 8 + (lane >> 3);
        const int hy = r / PW, hx = r - hy * PW;
        const int chunk = (lane & 7) ^ patch_key(hy, hx);
        const int iy = sy0 + hy, ix = sx0 + hx;
        const bool ok = r < HROWS && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int64_t pix = ((int64_t)nb * p.IH + iy) * p.IW + ix;
        hoff1[i] = ok ? (unsigned)((pix * p.lda + chunk * 8) * 2) : OOB_OFFSET;
        hoff2[i] = ok ? (unsigned)((pix * p.lda2 + chunk * 8) * 2) : OOB_OFFSET;
        hchunk[i] = ok ? chunk : -1;
    }
    const int crow = wid * 8 + (lane >> 3), cslot = lane & 7;
    unsigned woff[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int row = crow + RPP * i;
        const int n = bn0 + row;
        woff[i] = n < p.N ? (unsigned)(((int64_t)n * p.ldw + (cslot ^ ((row >> 1) & 7)) * 8) * 2) : OOB_OFFSET;
    }
    const int ncb = p.Cin / BK, nk = ncb * 9;

    auto issue_halo = [&](int i, int cb) {  // piece i of this wave, channel block cb -> patch buffer cb & 1
        const int q = i * NWV + wid;
        if (q >= HPIECES) return;
        const int ci0 = cb * BK;
        const bool second = p.k_split > 0 && ci0 >= p.k_split;
        dma16(second ? rA2 : rA, second ? hoff2[i] : hoff1[i], (second ? ci0 - p.k_split : ci0) * 2, sH + (cb & 1) * HALO_B + q * 1024);
    };
    // ---- fused GroupNorm (+SiLU) of the input patch
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const srd_t rAB = make_srd(GN ? (const void*)p.gn_ab : p.w);
    const int gn_sample = GN ? nb / p.gn_images_per_sample : 0;
    auto issue_ab = [&](int cb) {  // (scale, shift) of channel block cb: 64 x 2 floats = 512 B = half a DMA piece
        if (GN && wid == 0 && lane < 32)
            dma16(rAB, (unsigned)((((int64_t)gn_sample * p.Cin + cb * BK) * 2 + lane * 4) * 4), 0, smem + AB_OFF + (cb & 1) * 512);
    };
    auto normalize = [&](int i, int cb) {  // this lane's 16 bytes of piece i of patch buffer cb & 1
        const int q = i * NWV + wid;
        if (!GN || q >= HPIECES) return;
        const unsigned addr = lds0 + (cb & 1) * HALO_B + q * 1024 + lane * 16;
        const unsigned abaddr = lds0 + AB_OFF + (cb & 1) * 512 + (hchunk[i] < 0 ? 0 : hchunk[i]) * 64;
        uint4r v;
        floatx4 c0, c1, c2, c3;
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                     "ds_read_b128 %3, %6 offset:32\n\tds_read_b128 %4, %6 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(addr), "v"(abaddr) : "memory");
        const half8 x = __builtin_bit_cast(half8, v);
        const float sc[8] = {c0[0], c0[2], c1[0], c1[2], c2[0], c2[2], c3[0], c3[2]};
        const float sh[8] = {c0[1], c0[3], c1[1], c1[3], c2[1], c2[3], c3[1], c3[3]};
        half8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = fmaf((float)x[e], sc[e], sh[e]);
            if (p.gn_silu) t = silu_f(t);
            y[e] = hchunk[i] < 0 ? (half_t)0.f : (half_t)t;
        }
        const uint4r o = __builtin_bit_cast(uint4r, y);
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(o) : "memory");
    };
    auto issue_w = [&](int buf, int cb, int tap) {
        char* w = (char*)(sW + buf * BN * LD) + wid * 1024;
        const int soff = (tap * p.Cin + cb * BK) * 2;
#pragma unroll
        for (int i = 0; i < RW; ++i) dma16(rW, woff[i], soff, w + i * (NWV * 1024));
    };

    floatx16 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    int fpy[MI], fpx[MI];  // output pixel (within the patch) of this lane's activation fragment rows
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int ml = wm * MI * 32 + j * 32 + frow;
        fpy[j] = ml >> tw_shift;
        fpx[j] = ml & (TW - 1);
    }
    const int wsw = (frow >> 1) & 7;
    const bool wave_live = legacy_map || bn0 + wn * NI * 32 < p.N;   // wave-uniform
    auto compute = [&](int wbuf, int cb, int kh, int kw) {
        if (!wave_live) return;
        const half_t* hb = (const half_t*)(sH + (cb & 1) * HALO_B);
        const half_t* w = sW + wbuf * BN * LD + (wn * NI * 32 + frow) * LD;
        int arow[MI], akey[MI];
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int uy = oy0 + fpy[j] + kh - 1, ux = ox0 + fpx[j] + kw - 1;  // tap position in (upsampled) image coords
            const int hy = (uy >> up) - sy0, hx = (ux >> up) - sx0;
            arow[j] = hy * PW + hx;
            akey[j] = patch_key(hy, hx);
        }
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int kk = kg * KQ + kq;
            const int cl = kk * 2 + fhalf;
            half8 fa[MI], fw[NI];
#pragma unroll
            for (int j = 0; j < MI; ++j) fa[j] = *(const half8*)(hb + arow[j] * LD + ((cl ^ akey[j]) * 8));
#pragma unroll
            for (int i = 0; i < NI; ++i) fw[i] = *(const half8*)(w + i * 32 * LD + ((cl ^ wsw) * 8));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    };

    float pre_b = 0.f;
    if (tid < BN && bn0 + tid < p.N && p.bias) pre_b = p.bias[bn0 + tid];
    issue_ab(0);
#pragma unroll
    for (int i = 0; i < HP; ++i) issue_halo(i, 0);
    issue_w(0, 0, 0);
    int cbn = 0, tapn = 1;   // WS = 3: (channel block, tap) of the next weight slice to request
    if (WS == 3) {
        if (tapn == 9) { tapn = 0; ++cbn; }
        if (nk > 1) issue_w(1, cbn, tapn);
        if (++tapn == 9) { tapn = 0; ++cbn; }
    }
    if (tid < BN) sBias[tid] = pre_b;
    if (GN) {  // the first patch is normalised before its first tap
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // the (scale, shift) piece of wave 0 has landed for everyone
#pragma unroll
        for (int i = 0; i < HP; ++i) normalize(i, 0);
    }
    int cb = 0, kh = 0, kw = 0, tap = 0;
    int wslot = 0, islot = 2;   // WS = 3: ring slot computed from / requested into
    for (int s = 0; s < nk; ++s) {
        // WS = 3: the last request of every step is the weight slice two taps ahead (patch pieces are requested BEFORE it), and a wave's
        // memory operations retire in issue order: all but the newest RW pieces landed = this tap's slice and every patch piece are in LDS
        if (WS == 3 && s + 1 < nk) wait_vmcnt<RW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef INSV2V_GEMM_PROF
        if (s == 0) PROF_MARK(1);
#endif
        // GN: the piece requested one tap ago has landed (wait above) -> normalise it before this tap's DMA is issued
        if (GN && cb + 1 < ncb) {
#pragma unroll
            for (int i = 0; i < HP; ++i)  // static piece index: a runtime index would put hchunk[] in scratch
                if (tap == i + 1) normalize(i, cb + 1);
        }
        if (WS == 2 && s + 1 < nk) {
            const bool wrap = tap == 8;
            issue_w((s + 1) & 1, wrap ? cb + 1 : cb, wrap ? 0 : tap + 1);
        }
        if (tap == 0 && cb + 1 < ncb) issue_ab(cb + 1);
        if (cb + 1 < ncb) {  // the next block's patch trickles in, one piece per tap (static piece index: no scratch)
#pragma unroll
            for (int i = 0; i < HP; ++i)
                if (tap == i) issue_halo(i, cb + 1);
        }
        if (WS == 3) {
            if (s + 2 < nk) {
                issue_w(islot, cbn, tapn);
                if (++tapn == 9) { tapn = 0; ++cbn; }
            }
            islot = islot == 2 ? 0 : islot + 1;
        }
        compute(WS == 3 ? wslot : (s & 1), cb, kh, kw);
        if (WS == 3) wslot = wslot == 2 ? 0 : wslot + 1;
        if (++kw == 3) { kw = 0; ++kh; }
        if (++tap == 9) { tap = 0; kh = 0; ++cb; }
    }
    wait_vmcnt<0>();
    __syncthreads();
    PROF_MARK(2);
    tile_epilogue<WM, WN, MI, NI, KG, false>(p, acc, smem, sBias, nullptr, nullptr, tid, wm, wn, kg, bn0, 0, 0, true,
                                             [&](int r) { return (nb * p.OH + oy0 + (r >> tw_shift)) * p.OW + ox0 + (r & (TW - 1)); }
#ifdef INSV2V_GEMM_PROF
                                             , prof_t
#endif
    );
}

template <int WM, int WN, int MI, int NI, int KG, bool GN = false, int WS = 2>
static int launch_halo(const insv2v_gemm_desc& d, int tw_shift, hipStream_t s) {
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr size_t ring = 2 * (BM == 128 ? 23 : 43) * 1024 + WS * (size_t)BN * BK * sizeof(half_t);
    constexpr size_t stage = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t lds = (ring > stage ? ring : stage) + (size_t)BN * sizeof(float) + (GN ? 1024 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<WM, WN, MI, NI, KG, GN, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = d.NB * (d.OH * d.OW / BM) * ((d.N + BN - 1) / BN);
    static const int legacy_map = getenv("INSV2V_HALO_LEGACY_MAP") ? atoi(getenv("INSV2V_HALO_LEGACY_MAP")) : 0;
    if (legacy_map) tw_shift |= 256;
    hipLaunchKernelGGL((conv_halo_kernel<WM, WN, MI, NI, KG, GN, WS>), dim3(tiles), dim3(WM * WN * KG * 64), lds, s, d, tw_shift);
    return launch_status();
}

// patch geometry of the halo kernel for this problem: log2(TW), or -1 when it does not apply
static int halo_tw_shift(const insv2v_gemm_desc& d) {
    if (d.mode != INSV2V_MODE_CONV3X3 || d.stride != 1 || d.pad_t != 1 || d.pad_l != 1 || (d.Cin % BK) || d.batch > 1) return -1;
    if (d.k_split > 0 && (d.k_split % BK)) return -1;
    if (d.row_stats || d.rb_mod > 0 || d.act == INSV2V_ACT_GEGLU) return -1;
    const int up = d.upsample ? 2 : 1;
    if (d.OH != d.IH * up || d.OW != d.IW * up) return -1;
    if (d.OH % 8 == 0 && d.OW % 16 == 0) return 4;   // 8 x 16 patch
    if (d.OH % 16 == 0 && d.OW % 8 == 0) return 3;   // 16 x 8 patch
    return -1;
}

template <int WM, int WN, int MI, int NI, int MODE, int STAGES, int KG = 1>
static int launch_cfg(const insv2v_gemm_desc& d, hipStream_t s) {
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr size_t ring = (size_t)STAGES * (BM + BN) * BK * sizeof(half_t);
    constexpr size_t stage = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t lds = (ring > stage ? ring : stage) + (size_t)(2 * BN + 2 * BM) * sizeof(float);
    if (lds > 160 * 1024) return INSV2V_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<WM, WN, MI, NI, MODE, STAGES, KG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    dim3 grid(tiles, d.batch > 0 ? d.batch : 1, d.split_k > 1 ? d.split_k : 1);
    hipLaunchKernelGGL((gemm_kernel<WM, WN, MI, NI, MODE, STAGES, KG>), grid, dim3(WM * WN * KG * 64), lds, s, d);
    return launch_status();
}

// tile shapes: 1 = 128x128, 2 = 64x128, 3 = 128x64, 4 = 64x64 (4 waves); 5 = 128x128, 6 = 256x128, 7 = 128x64 (8 waves);
// 8 = 128x128, 9 = 128x64 (8 waves as 2 K groups of 4)
template <int MODE, int STAGES>
static int dispatch_tile(const insv2v_gemm_desc& d, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch_cfg<2, 2, 2, 2, MODE, STAGES>(d, s);
        case 2: return launch_cfg<2, 2, 1, 2, MODE, STAGES>(d, s);
        case 3: return launch_cfg<2, 2, 2, 1, MODE, STAGES>(d, s);
        case 4: return launch_cfg<2, 2, 1, 1, MODE, STAGES>(d, s);
        case 5: return launch_cfg<4, 2, 1, 2, MODE, STAGES>(d, s);
        case 6: return launch_cfg<4, 2, 2, 2, MODE, STAGES>(d, s);
        case 7: return launch_cfg<4, 2, 1, 1, MODE, STAGES>(d, s);
        case 8: return launch_cfg<2, 2, 2, 2, MODE, STAGES, 2>(d, s);  // 128x128, 2 K groups of 4 waves
        case 9: return launch_cfg<2, 2, 2, 1, MODE, STAGES, 2>(d, s);  // 128x64,  2 K groups of 4 waves
    }
    return INSV2V_EINVAL;
}

static int pick_tile(const insv2v_gemm_desc& d) {
    // Measured on MI355X (tools/bench_gemm.py, profiles/): the kernel is bound by memory latency x the LDS
    // capacity available for slices in flight, so wave-level parallelism decides: the 8-wave 128x128 tile
    // (2 workgroups = 16 waves per CU) wins whenever it yields >= ~200 workgroups; tiny problems
    // (M = 1152 at the lowest UNet level) use 64x64 tiles (4-5 workgroups/CU).
    // GEGLU needs each wave to own an [h|g] pair of 32-row weight blocks (NI = 2): tiles 5 / 2.
    const long batch = d.batch > 0 ? d.batch : 1;
    auto blocks = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * batch; };
    const long b11 = blocks(128, 128);
    if (d.act == INSV2V_ACT_GEGLU) return b11 >= 200 ? 5 : 2;
    if (d.mode == INSV2V_MODE_CONV3X3) return (b11 >= 200 && d.N >= 128) ? 5 : 4;
    // Round 2 re-check (profiles/r02_ring_depth_tile_sweep.txt): in ISOLATION the 64x64 tile wins on the single-branch N = C GEMMs
    // (1536 x 1280 x 1280 + residual 15.0 vs 19.3 us, 24 576 x 320 x 320 17.8 vs 20.1 us), but a rule that picked it there made the
    // 3-stream UNet step SLOWER (-2.7 % vs -4.3 % against the same baseline): three branches' kernels overlap, and many small
    // workgroups crowd the other branches' tiles out.  Deeper LDS rings (tile codes 3x / 4x) lose wherever they cost a workgroup per CU.
    return blocks(128, 64) >= 200 ? 5 : 4;
}

// Split-K second pass: out = epilogue(sum_s partial[s]) with the same epilogue semantics as the main
// kernel (alpha, bias, row bias, SiLU, residual; fp16 or fp32 store).  8 outputs per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(insv2v_gemm_desc p, const float* ws, int nsplit) {
    const int64_t nchunk = (int64_t)p.M * (p.N / 8);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nchunk) return;
    const int m = (int)(idx / (p.N / 8)), n = (int)(idx - (int64_t)m * (p.N / 8)) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = ws + (int64_t)m * p.N + n;
    const int64_t slab = (int64_t)p.M * p.N;
    // everything this thread needs is requested up front (16-byte loads): residual, bias, row bias, then the slabs four at a time
    const bool res_vec = p.residual && (p.ldr & 7) == 0 && (((uintptr_t)p.residual & 15) == 0);
    half8 rh = {0, 0, 0, 0, 0, 0, 0, 0};
    if (res_vec) rh = *(const half8*)((const half_t*)p.residual + (int64_t)m * p.ldr + n);
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, r0 = b0, r1 = b0;
    if (p.bias) {
        if (((uintptr_t)p.bias & 15) == 0) { b0 = *(const float4*)(p.bias + n); b1 = *(const float4*)(p.bias + n + 4); }
        else { b0 = make_float4(p.bias[n], p.bias[n + 1], p.bias[n + 2], p.bias[n + 3]); b1 = make_float4(p.bias[n + 4], p.bias[n + 5], p.bias[n + 6], p.bias[n + 7]); }
    }
    if (p.row_bias) {
        const float* rb = p.row_bias + (int64_t)(m / p.rows_per_group) * p.ld_rb + n;
        if ((p.ld_rb & 3) == 0 && (((uintptr_t)p.row_bias & 15) == 0)) { r0 = *(const float4*)rb; r1 = *(const float4*)(rb + 4); }
        else { r0 = make_float4(rb[0], rb[1], rb[2], rb[3]); r1 = make_float4(rb[4], rb[5], rb[6], rb[7]); }
    }
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = *(const float4*)(src + (s + u) * slab); b[u] = *(const float4*)(src + (s + u) * slab + 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // fixed summation order s = 0, 1, 2, ...: deterministic
            v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
        }
    }
    for (; s < nsplit; ++s) {
        const float4 a = *(const float4*)(src + s * slab), b = *(const float4*)(src + s * slab + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = v[e] * p.alpha;
        if (p.bias) x += bb[e];
        if (p.row_bias) x += rr[e];
        if (p.act == INSV2V_ACT_SILU) x = silu_f(x);
        else if (p.act == INSV2V_ACT_QUICK_GELU) x = quick_gelu_f(x);
        else if (p.act >= INSV2V_ACT_RELU) x = act_raft_f(x, p.act);
        if (p.residual) x += res_vec ? (float)rh[e] : (float)((const half_t*)p.residual)[(int64_t)m * p.ldr + n + e];
        v[e] = x;
    }
    if (p.c_fp32) {
        float* dst = (float*)p.c + (int64_t)m * p.ldc + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = v[e];
    } else {
        half_t* dst = (half_t*)p.c + (int64_t)m * p.ldc + n;
        if ((p.ldc & 7) == 0 && (((uintptr_t)p.c & 15) == 0)) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
            *(half8*)dst = h;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = (half_t)v[e];
        }
    }
}

// (mean, rstd) pairs from the producer's partial sums, for the kernels that park finished statistics (gemm_p8 / gemm_w4)
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* parts, float2* stats, int M, int nparts, int C, float eps) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < nparts; ++j) { const float2 t = parts[(int64_t)j * M + m]; s1 += t.x; s2 += t.y; }
    const float mean = s1 / C;
    stats[m] = make_float2(mean, rsqrtf(fmaxf(s2 / C - mean * mean, 0.f) + eps));
}

// Persistent big-tile kernels (gemm_p8.hip: 256x256, one 8-wave workgroup per CU; gemm_w4.hip: 128x256, two 4-wave
// workgroups per CU) for the linear shapes where they measured faster than the 128x128 tile on MI355X
// (tools/gemm_check, profiles/r02_gemm_check_*.txt; both the 3-branch batched and the single-branch token counts):
// the GEGLU FF1 and the fused q/k/v projections, i.e. wide-N GEMMs without a residual.  GEMMs with a residual
// (N = C) and every convolution stay on the 2-workgroup 128x128 / halo kernels, whose four waves per SIMD overlap
// the epilogue with the next tile.  Returns 0 = no, 1 = p8, 2 = w4.
// Round 4: which of the two 8-wave ping-pong kernels (gemm_q8: 256x256, gemm_r8: 256x320) - if any - runs a linear GEMM without
// activation or a convolution.  Every UNet width is a multiple of 320, so gemm_r8 has no partial column tiles where gemm_q8 has
// (N = 320 / 640 / 960 / 1920); what decides is how the tile count quantises over the CUs (persistent grids: rounds of tiles), against
// the finer-grained round-1 kernels (128x128 tiles / patch-tiled conv, two workgroups per CU) at ~0.8 of the new engine's rate
// (tools/gemm_check --set unet30: profiles/r04_gemm_r8_unet30.txt).  Returns 0 = neither, 1 = gemm_q8, 2 = gemm_r8.
static int num_cus_gemm() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}
static int pick_pingpong(const insv2v_gemm_desc& d) {
    static const int enabled = getenv("INSV2V_GEMM_R8") ? atoi(getenv("INSV2V_GEMM_R8")) : 1;
    static int cus = 0;
    if (!enabled || d.act != INSV2V_ACT_NONE || d.batch > 1 || d.c_fp32 || d.M < 8192 || d.N < 256) return 0;
    // statistics of the output rows: only gemm_r8's LINEAR form emits them (two 160-column partial sums per tile row)
    static const int r8_stats = getenv("INSV2V_R8_STATS") ? atoi(getenv("INSV2V_R8_STATS")) : 1;
    if (d.stats_out && (!r8_stats || d.mode != INSV2V_MODE_LINEAR || d.k_split || (d.N % 320))) return 0;
    if (d.mode == INSV2V_MODE_LINEAR && d.K < 256) return 0;
    if (d.row_bias && ((d.ld_rb & 3) || (d.rows_per_group % 256 && d.M > d.rows_per_group))) return 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    const long tm = (d.M + 255) / 256;
    const double old_cost = (double)tm * d.N / cus / 0.80;
    const long q_tiles = tm * ((d.N + 255) / 256), r_tiles = tm * ((d.N + 319) / 320);
    const double q_cost = (double)((q_tiles + cus - 1) / cus) * 256 * 1.03, r_cost = (double)((r_tiles + cus - 1) / cus) * 320;
    // (only the UNet's widths: the VAE's 128 / 256 / 512-channel convolutions stay where round 3 measured them)
    if (d.N % 320 == 0 && (r_cost <= q_cost || d.stats_out) && r_cost < old_cost) return 2;
    // (INSV2V_Q8_MIN_N: the narrowest output gemm_q8 takes by dispatch; round 6: 256, i.e. the VAE's 256 / 512-channel
    //  convolutions on the 16x16x32 engine: profiles/r06_vae_q8_ab.txt)
    static const int q8_min_n = getenv("INSV2V_Q8_MIN_N") ? atoi(getenv("INSV2V_Q8_MIN_N")) : 256;
    if (d.N % 256 == 0 && d.N >= q8_min_n && (d.N % 320 != 0 || q_cost < r_cost) && q_cost < old_cost && d.K >= 1152 && !d.stats_out) return 1;
    return 0;
}
static int pick_persistent(const insv2v_gemm_desc& d) {
    static const int enabled = getenv("INSV2V_GEMM_PERSISTENT") ? atoi(getenv("INSV2V_GEMM_PERSISTENT")) : 1;
    if (!enabled || d.mode != INSV2V_MODE_LINEAR || d.batch > 1 || d.c_fp32 || d.k_split) return 0;
    // the persistent kernels park ONE row-bias vector per tile: every 256-row tile must lie inside one bias group
    if (d.row_bias && ((d.ld_rb & 3) || (d.rows_per_group % 256 && d.M > d.rows_per_group))) return 0;
    // Round 3 (stacked clips, M = 23 040 ... 92 160 at levels 1-2): with K >= 1280 the 256x256 kernel's epilogue is amortised and it
    // wins with or without a residual - FF2 23 040 x 1 280 x 5 120 358 vs 424 us, q/k/v 23 040 x 3 840 x 1 280 259 vs 325 us, FF2
    // 92 160 x 640 x 2 560 462 vs 482 us; at the single-clip sizes (M*N below ~2e7) the 128x128 tile stays ahead
    // (tools/bench_tiles_r03.py, profiles/r03_tile_sweep_stacked.txt)
    if (d.act == INSV2V_ACT_NONE && d.K >= 1280 && d.N >= 640 && (int64_t)d.M * d.N >= (int64_t)20 << 20) return 1;
    if (d.residual) return 0;
    if (d.act == INSV2V_ACT_GEGLU) {
        if (d.K <= 320) return d.M >= 8192 ? 2 : 0;
        return d.M >= 1024 ? 1 : 0;
    }
    if (d.act != INSV2V_ACT_NONE || d.N < 960 || d.K > 640 || d.M < 4096) return 0;
    if (d.M >= 12288) return 2;
    return d.K == 640 ? 1 : 2;
}

// Automatic split-K: only for problems that cannot fill the chip (fewer than ~1 workgroup per CU with
// 128x128 tiles) and whose K is long enough that every split still runs >= 16 slices.
static int pick_split(const insv2v_gemm_desc& d) {
    if (d.split_k == 1 || !d.workspace || d.batch > 1 || d.act == INSV2V_ACT_GEGLU || (d.N & 7) || d.row_stats || d.rb_mod > 0) return 1;
    const long b11 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
    const int nk = (d.K + BK - 1) / BK;
    int s = d.split_k;
    if (s == 0) {
        if (b11 >= 200 || nk < 64) return 1;
        s = (int)(512 / b11);
        if (s > nk / 16) s = nk / 16;
        if (s > 8) s = 8;
    }
    if (s < 2) return 1;
    if ((int64_t)s * d.M * d.N * 4 > d.workspace_bytes) return 1;
    return s;
}

// Column-tile width of the kernel that will run a statistics-emitting GEMM (0 = cannot emit): such a problem always takes the
// round-1 tile kernel's vectorised fp16 epilogue, no split-K, no persistent kernel.
static int stats_tile_width(const insv2v_gemm_desc& d) {
    if (d.mode != INSV2V_MODE_LINEAR || d.c_fp32 || d.act == INSV2V_ACT_GEGLU || d.batch > 1 || (d.N & 7) || (d.ldc & 7) ||
        ((uintptr_t)d.c & 15) || (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15) || (int64_t)d.M * d.ldr * 2 >= ((int64_t)1 << 31))))
        return 0;
    int shape = d.tile % 10;
    if (d.tile >= 240 && d.tile <= 249) return (d.N % 320 || d.k_split) ? 0 : 160;   // gemm_r8 forced
    if (d.tile >= 100) return 0;
    if (d.tile == 0) {   // what the dispatch would do with a statistics-emitting problem
        insv2v_gemm_desc dd = d;
        if (!dd.stats_out) dd.stats_out = (float*)(uintptr_t)16;
        if (dd.batch <= 0) dd.batch = 1;
        if (pick_pingpong(dd) == 2) return 160;
    }
    if (shape == 0) shape = pick_tile(d);
    switch (shape) {
        case 1: case 2: case 5: case 6: case 8: return 128;
        case 3: case 4: case 7: case 9: return 64;
    }
    return 0;
}

// Bytes of one operand window of insv2v_gemm (0 = the hardware's: 2 GiB less a margin).  Tests shrink it so that small problems take the
// split path (insv2v_set_operand_window); the product never touches it.
static int64_t g_operand_window = 0;
extern "C" int64_t insv2v_set_operand_window(int64_t bytes) {
    const int64_t prev = g_operand_window;
    g_operand_window = bytes > 0 ? bytes : 0;
    return prev;
}
static int64_t operand_window() { return g_operand_window > 0 ? g_operand_window : ((int64_t)1 << 31) - ((int64_t)1 << 20); }
// Does an operand of this problem reach beyond one window (insv2v_gemm then runs it as row / image ranges, without output statistics)?
// One helper for insv2v_gemm and insv2v_gemm_stats_parts, so that the two cannot disagree.
static bool beyond_window(const insv2v_gemm_desc& d) {
    const bool conv = d.mode == INSV2V_MODE_CONV3X3;
    const int64_t a_rows = conv ? (int64_t)d.NB * d.IH * d.IW : (int64_t)d.M;
    const int64_t esz = d.c_fp32 ? 4 : 2;
    const int64_t big_in = std::max(a_rows * d.lda * 2, d.k_split ? a_rows * d.lda2 * 2 : (int64_t)0);
    const int64_t big_out = std::max((int64_t)d.M * d.ldc * esz, d.residual ? (int64_t)d.M * d.ldr * 2 : (int64_t)0);
    return big_in >= operand_window() || big_out >= operand_window();
}

extern "C" int insv2v_gemm_stats_parts(const insv2v_gemm_desc* dp) {
    if (!dp) return 0;
    insv2v_gemm_desc d = *dp;
    if (d.batch <= 0) d.batch = 1;
    if (beyond_window(d)) return 0;   // a split problem emits no statistics: say so before the caller allocates for them
    const int w = stats_tile_width(d);
    return w ? (d.N + w - 1) / w : 0;
}

// the persistent kernels park finished (mean, rstd) pairs: partial sums from a producer are finalised by a small launch first
static insv2v_gemm_desc finished_stats_of(insv2v_gemm_desc dd, hipStream_t s) {
    if (dd.stats_parts > 0) {
        hipLaunchKernelGGL(ln_finalize_kernel, dim3((dd.M + 255) / 256), dim3(256), 0, s, (const float2*)dd.row_stats,
                           (float2*)dd.stats_scratch, dd.M, dd.stats_parts, dd.K, dd.ln_eps);
        dd.row_stats = dd.stats_scratch; dd.stats_parts = 0;
    }
    return dd;
}

extern "C" int insv2v_gemm(const insv2v_gemm_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    insv2v_gemm_desc d = *dp;
    if (!d.a || !d.w || !d.c || d.M <= 0 || d.N <= 0 || d.K <= 0) return INSV2V_EINVAL;
    if ((d.K & 7) || (d.lda & 7) || (d.ldw & 7)) return INSV2V_EINVAL;
    if (((uintptr_t)d.a | (uintptr_t)d.w) & 15) return INSV2V_EINVAL;
    if (d.k_split) {
        if (!d.a2 || (d.k_split % BK) || (d.lda2 & 7) || ((uintptr_t)d.a2 & 15)) return INSV2V_EINVAL;
    }
    if (d.row_bias && d.rows_per_group <= 0) return INSV2V_EINVAL;
    if (d.row_stats && (!d.col_sum || d.batch > 1)) return INSV2V_EINVAL;
    if (d.stats_parts < 0 || (d.stats_parts > 0 && (!d.row_stats || !d.stats_scratch))) return INSV2V_EINVAL;
    if (d.stats_out && stats_tile_width(d) == 0) return INSV2V_EUNSUPPORTED;
    if (d.act == INSV2V_ACT_GEGLU && (d.N % 64)) return INSV2V_EINVAL;
    if (d.act < 0 || d.act > INSV2V_ACT_TANH) return INSV2V_EINVAL;                                   // unknown activation code
    // (ReLU / sigmoid / tanh - the optical-flow network's - live in the 128x128 tile kernel's epilogue and the split-K reduction, for both modes;
    //  the patch-tiled and ping-pong kernels never see those codes: the dispatch below keeps such a call away from them)
    if (d.batch <= 0) d.batch = 1;
    if (d.alpha == 0.f) d.alpha = 1.f;
    if (d.w_group_rows < 0 || (d.w_group_rows > 0 && (d.w_group_rows % 256 || d.w_group_stride <= 0))) return INSV2V_EINVAL;
    if (d.w_group_rows > 0) {   // grouped weights: the 256-row ping-pong kernels only, nothing riding in the epilogue
        if (d.mode != INSV2V_MODE_LINEAR || d.batch > 1 || d.c_fp32 || d.act != INSV2V_ACT_NONE || d.residual || d.row_bias || d.row_stats || d.stats_out ||
            d.k_split || d.split_k > 1 || d.K < 256 || (d.N % 256 && d.N % 320))
            return INSV2V_EUNSUPPORTED;
    }
    if (d.gn_ab && (d.mode != INSV2V_MODE_CONV3X3 || d.gn_images_per_sample <= 0 || d.upsample)) return INSV2V_EINVAL;
    if (d.mode == INSV2V_MODE_CONV3X3) {
        if (d.Cin <= 0 || (d.Cin % BK) || d.K != 9 * d.Cin) return INSV2V_EINVAL;
        if ((long)d.NB * d.OH * d.OW != d.M || d.stride < 1) return INSV2V_EINVAL;
    } else if (d.mode != INSV2V_MODE_LINEAR) {
        return INSV2V_EUNSUPPORTED;
    }
    // LDS-DMA addressing uses 32-bit byte offsets against a 2 GiB descriptor window per operand.  A problem whose activations, output or
    // residual reach beyond one window runs as several launches over row ranges (LINEAR) / image ranges (CONV3X3: images are independent),
    // every part inside the window and aligned to the row-bias / GroupNorm groups, so that every kernel family keeps its 32-bit offsets
    // (round 5; the Python wrapper used to cut convolutions only).  Statistics of the output (stats_out: part-major with stride M) are not
    // emitted by a split problem: the caller gets INSV2V_EUNSUPPORTED and takes the statistics pass, as for any problem that cannot emit them.
    {
        const bool conv = d.mode == INSV2V_MODE_CONV3X3;
        const int64_t lim = operand_window();
        if ((int64_t)d.N * d.ldw * 2 >= ((int64_t)1 << 31)) return INSV2V_EUNSUPPORTED;
        const int64_t esz = d.c_fp32 ? 4 : 2;
        if (beyond_window(d)) {
            if (d.batch > 1 || d.stats_out || d.split_k > 1) return INSV2V_EUNSUPPORTED;
            insv2v_gemm_desc f = finished_stats_of(d, as_stream(stream));   // partial row statistics are finalised over the whole problem first
            auto lcm = [](int64_t a, int64_t b) { int64_t x = a, y = b; while (y) { const int64_t t = x % y; x = y; y = t; } return a / x * b; };
            const int64_t opix = conv ? (int64_t)d.OH * d.OW : 1, ipix = conv ? (int64_t)d.IH * d.IW : 1;   // output / input rows per unit
            int64_t unit = conv ? 1 : 2;                                                                 // images (conv) / rows (linear) per part: a multiple of this
            if (d.row_bias) {
                const int64_t grp = (int64_t)d.rows_per_group * (d.rb_mod > 0 ? d.rb_mod : 1);         // rows after which the bias pattern repeats / moves on
                if (conv && grp % opix) return INSV2V_EUNSUPPORTED;
                unit = lcm(unit, conv ? grp / opix : grp);
            }
            if (conv && d.gn_ab) unit = lcm(unit, d.gn_images_per_sample);
            if (!conv && unit < 256) unit = lcm(unit, 256);                                             // whole tiles where nothing else decides
            if (f.w_group_rows > 0) unit = lcm(unit, f.w_group_rows);                                   // grouped weights: whole groups per part
            const int64_t total = conv ? d.NB : d.M;
            const int64_t in_row = std::max((int64_t)d.lda * 2, d.k_split ? (int64_t)d.lda2 * 2 : (int64_t)0) * ipix;
            const int64_t out_row = std::max((int64_t)d.ldc * esz, d.residual ? (int64_t)d.ldr * 2 : (int64_t)0) * opix;
            int64_t per = (lim - 1) / std::max(in_row, out_row) / unit * unit;
            if (per <= 0) return INSV2V_EUNSUPPORTED;
            // parts as even as possible (every distinct part size is its own dispatch decision, equal sizes share it)
            const int64_t nparts = (total + per - 1) / per;
            per = ((total + nparts - 1) / nparts + unit - 1) / unit * unit;
            for (int64_t u0 = 0; u0 < total; u0 += per) {
                const int64_t nu = std::min(per, total - u0);
                insv2v_gemm_desc s = f;
                const int64_t r_in = u0 * ipix, r_out = u0 * opix;
                s.a = (const char*)f.a + r_in * f.lda * 2;
                if (f.k_split) s.a2 = (const char*)f.a2 + r_in * f.lda2 * 2;
                s.c = (char*)f.c + r_out * f.ldc * esz;
                if (f.residual) s.residual = (const char*)f.residual + r_out * f.ldr * 2;
                if (f.row_stats) s.row_stats = f.row_stats + r_out * 2;
                if (f.w_group_rows > 0) s.w = (const char*)f.w + (r_out / f.w_group_rows) * f.w_group_stride * 2;
                if (f.row_bias && f.rb_mod <= 0) s.row_bias = f.row_bias + (r_out / f.rows_per_group) * f.ld_rb;
                if (conv && f.gn_ab) s.gn_ab = f.gn_ab + (u0 / f.gn_images_per_sample) * (int64_t)f.Cin * 2;
                s.M = (int32_t)(nu * opix);
                if (conv) s.NB = (int32_t)nu;
                const int rc = insv2v_gemm(&s, stream);
                if (rc != 0) return rc;
            }
            return 0;
        }
    }
    if (d.w_group_rows > 0) {   // grouped weights (validated above): gemm_r8 / gemm_q8 by the rounds of tiles each needs
        const int64_t groups = ((int64_t)d.M + d.w_group_rows - 1) / d.w_group_rows;
        if (groups * d.w_group_stride * 2 >= ((int64_t)1 << 31)) return INSV2V_EUNSUPPORTED;
        const int cus = num_cus_gemm();
        const long tm = (d.M + 255) / 256, q_tiles = tm * ((d.N + 255) / 256), r_tiles = tm * ((d.N + 319) / 320);
        const bool r_ok = d.N % 320 == 0, q_ok = d.N % 256 == 0;
        const double q_cost = (double)((q_tiles + cus - 1) / cus) * 256 * 1.03, r_cost = (double)((r_tiles + cus - 1) / cus) * 320;
        const bool use_r = d.tile / 10 == 24 ? true : d.tile / 10 == 23 ? false : (r_ok && (!q_ok || r_cost <= q_cost));
        if (use_r ? !r_ok : !q_ok) return INSV2V_EUNSUPPORTED;
        return use_r ? insv2v_gemm_r8(d, 0, as_stream(stream)) : insv2v_gemm_q8(d, 0, as_stream(stream));
    }
    // tile code: low digit = tile shape (0 auto), tens digit = ring depth S (0 default = 2, or 2 / 3).
    // (An L2 prefetch of slices 3 steps ahead was measured and removed: 30-45 % slower, profiles/README.md.)
    auto finished_stats = [&](const insv2v_gemm_desc& dd) { return finished_stats_of(dd, as_stream(stream)); };
    if (d.tile >= 200 && d.tile <= 206) return INSV2V_EUNSUPPORTED;  // round 3's 256x256 kernel (gemm_p8): retired in round 6, source kept under tools/archive/
    if (d.tile >= 230 && d.tile <= 239) {  // 256x256 8-phase kernel, interleaved half-tile ownership (gemm_q8.hip), forced
        if (d.split_k > 1 || d.stats_out) return INSV2V_EUNSUPPORTED;
        d.split_k = 1;
        return insv2v_gemm_q8(finished_stats(d), d.tile - 230, as_stream(stream));
    }
    if (d.tile >= 240 && d.tile <= 249) {  // 256x320 tile on the round-4 engine (gemm_r8.hip), forced
        if (d.split_k > 1) return INSV2V_EUNSUPPORTED;
        d.split_k = 1;
        return insv2v_gemm_r8(finished_stats(d), d.tile - 240, as_stream(stream));
    }
    if (d.tile >= 210 && d.tile <= 221) {  // 4-wave persistent kernel (gemm_w4.hip), forced: 210 = 128x256, 211 = 256x128
        if (d.split_k > 1 || d.stats_out) return INSV2V_EUNSUPPORTED;
        d.split_k = 1;
        return insv2v_gemm_w4(finished_stats(d), d.tile - 210, as_stream(stream));
    }
    int shape = d.tile % 10, pipe = d.tile / 10;
    const int nsplit = d.stats_out ? 1 : pick_split(d);
    insv2v_gemm_desc full = d;
    if (nsplit > 1) {  // main pass: raw fp32 partial slabs [nsplit, M, N] in the workspace, no epilogue
        d.c = d.workspace; d.ldc = d.N; d.c_fp32 = 1; d.c_bs = 0;
        d.bias = nullptr; d.row_bias = nullptr; d.residual = nullptr; d.act = INSV2V_ACT_NONE; d.alpha = 1.f;
        d.row_stats = nullptr; d.col_sum = nullptr;
        d.split_k = nsplit;
        if (shape == 0) shape = 5;
    } else {
        d.split_k = 1;
    }
    if (nsplit <= 1 && d.tile == 0 && !d.gn_ab) {   // (with stats_out: gemm_r8 or nothing, pick_pingpong)
        const int pp = pick_pingpong(d);
        if (pp) {
            const insv2v_gemm_desc dd = finished_stats(d);
            const int rc = pp == 2 ? insv2v_gemm_r8(dd, 0, as_stream(stream)) : insv2v_gemm_q8(dd, 0, as_stream(stream));
            // (a statistics buffer sized for gemm_r8's 160-column parts must not reach a kernel with another part width)
            if (rc != INSV2V_EUNSUPPORTED || d.stats_out) return rc;
        }
    }
    if (nsplit <= 1 && d.tile == 0 && !d.stats_out && d.act == INSV2V_ACT_GEGLU && d.mode == INSV2V_MODE_LINEAR && !d.k_split && !d.residual) {
        // Round 4: the GEGLU FF1 of levels 1-3 (N = 5120 / 10240 = whole 320-row tiles) on the 256x320 kernel where its rounds of tiles
        // cost no more than the 256x256 kernel's: 7 % faster at K = 640, 3.5 % at K = 1280 (profiles/r04_gemm_r8_geglu.txt)
        static const int r8_geglu = getenv("INSV2V_R8_GEGLU") ? atoi(getenv("INSV2V_R8_GEGLU")) : 1;
        const int cus = num_cus_gemm();
        if (r8_geglu && cus > 0 && d.N % 320 == 0 && d.M >= 8192 && d.K >= 640 && !d.c_fp32 && d.batch <= 1) {
            const long tm = (d.M + 255) / 256, q_tiles = tm * ((d.N + 255) / 256), r_tiles = tm * (d.N / 320);
            const double q_cost = (double)((q_tiles + cus - 1) / cus) * 256 * 1.03, r_cost = (double)((r_tiles + cus - 1) / cus) * 320;
            if (r_cost <= q_cost) {
                const int rc = insv2v_gemm_r8(finished_stats(d), 0, as_stream(stream));
                if (rc != INSV2V_EUNSUPPORTED) return rc;
            }
        }
    }
    if (nsplit <= 1 && d.tile == 0 && !d.stats_out) {
        const int pick = pick_persistent(d);
        if (pick) {
            // eligibility is static (shape / alignment): an EUNSUPPORTED answer comes before any launch of the kernel itself
            const insv2v_gemm_desc dd = finished_stats(d);
            // (N = 640 - FF2 of level 1 - is 2.5 column tiles of the 256x256 kernel; running the last 128 columns on the 128x128 tile as a
            //  second launch was built and measured: no gain end to end, profiles/r03_gemm_split_columns_experiment.txt)
            // round 4: the 256x256 choice runs on gemm_q8 (interleaved half-tile ownership, LDS-DMA inside the MFMA segments, concurrent
            // epilogues): 8-25 % faster than gemm_p8 on every UNet shape (profiles/r04_gemm_q8_vs_p8.txt)
            const int rc = pick == 1 ? insv2v_gemm_q8(dd, 0, as_stream(stream)) : insv2v_gemm_w4(dd, 0, as_stream(stream));
            if (rc != INSV2V_EUNSUPPORTED) return rc;
        }
    }
    if (nsplit <= 1 && (d.tile == 100 || d.tile == 0)) {
        const int tws = halo_tw_shift(d);
        if (d.tile == 100 && tws < 0) return INSV2V_EUNSUPPORTED;
        if (d.tile == 100 && d.act >= INSV2V_ACT_RELU) return INSV2V_EUNSUPPORTED;
        if (tws >= 0 && d.act < INSV2V_ACT_RELU && (d.tile == 100 || ((long)d.M / 128) * ((d.N + 127) / 128) >= 200))
            return d.gn_ab ? launch_halo<4, 2, 1, 2, 1, true>(d, tws, as_stream(stream)) : launch_halo<4, 2, 1, 2, 1>(d, tws, as_stream(stream));
    }
    if (d.gn_ab) return INSV2V_EUNSUPPORTED;  // only the patch-tiled kernel normalises its input (never silently skip the norm)
    // Round 3: convolutions the patch kernel cannot tile (8x12 / 4x6 latents, stride 2) at the stacked-clip sizes: the 256x256 persistent
    // kernel beats the gathered 128x128 tile once M fills it (23 040 x 1 280 x 11 520: 712 vs 838 us, x 23 040: 1 357 vs 1 597 us;
    // at M = 4 608 it loses 340 vs 192 us, at M = 5 760 320 vs 206 us, at M = 11 520 (10 stacked clips at the 4x6 level) it wins 330 vs 384 us -
    // tools/bench_conv_tiles_r03.py, profiles/r03_conv_tile_sweep_stacked.txt, r03_conv_tile_sweep_B30.txt)
    if (nsplit <= 1 && d.tile == 0 && d.mode == INSV2V_MODE_CONV3X3 && d.batch == 1 && !d.c_fp32 && d.M >= 10240 && d.N >= 640 && d.K >= 5760) {
        static const int enabled = getenv("INSV2V_GEMM_PERSISTENT") ? atoi(getenv("INSV2V_GEMM_PERSISTENT")) : 1;
        if (enabled) {
            const int rc = insv2v_gemm_q8(d, 0, as_stream(stream));
            if (rc != INSV2V_EUNSUPPORTED) return rc;
        }
    }
    if (d.tile >= 101 && d.tile <= 103 && d.act >= INSV2V_ACT_RELU) return INSV2V_EUNSUPPORTED;
    if (nsplit <= 1 && d.tile == 103) {  // ONE 16-wave workgroup per CU on a 256-pixel patch, weight slices requested two taps ahead, for A/B measurement
        int tws = halo_tw_shift(d);
        if (tws < 0 || d.gn_ab) return INSV2V_EUNSUPPORTED;
        if (d.OH % 16 == 0 && d.OW % 16 == 0) tws = 4;          // 16 x 16 patch
        else if (d.OH % 32 == 0 && d.OW % 8 == 0) tws = 3;      // 32 x 8 patch
        else return INSV2V_EUNSUPPORTED;
        return launch_halo<8, 2, 1, 2, 1, false, 3>(d, tws, as_stream(stream));
    }
    if (nsplit <= 1 && d.tile == 102) {  // 4 waves with 64 x 64 wave tiles (1 KB of LDS reads per MFMA instead of 1.5), for A/B measurement
        const int tws = halo_tw_shift(d);
        if (tws < 0 || d.gn_ab) return INSV2V_EUNSUPPORTED;
        return launch_halo<2, 2, 2, 2, 1>(d, tws, as_stream(stream));
    }
    if (nsplit <= 1 && d.tile == 101) {  // K-group variant, kept for A/B measurement
        const int tws = halo_tw_shift(d);
        if (tws < 0) return INSV2V_EUNSUPPORTED;
        return launch_halo<2, 2, 2, 2, 2>(d, tws, as_stream(stream));
    }
    if (shape == 0) shape = pick_tile(d);
    if (d.act == INSV2V_ACT_GEGLU && (shape == 3 || shape == 4 || shape == 7 || shape == 9)) shape = 2;
    if (pipe == 0) pipe = 2;
    hipStream_t s = as_stream(stream);
    const bool conv = d.mode == INSV2V_MODE_CONV3X3;
    int rc = INSV2V_EINVAL;
    switch (pipe) {
        case 2: rc = conv ? dispatch_tile<INSV2V_MODE_CONV3X3, 2>(d, shape, s) : dispatch_tile<INSV2V_MODE_LINEAR, 2>(d, shape, s); break;
        case 3: rc = conv ? dispatch_tile<INSV2V_MODE_CONV3X3, 3>(d, shape, s) : dispatch_tile<INSV2V_MODE_LINEAR, 3>(d, shape, s); break;
        case 4: rc = conv ? dispatch_tile<INSV2V_MODE_CONV3X3, 4>(d, shape, s) : dispatch_tile<INSV2V_MODE_LINEAR, 4>(d, shape, s); break;
    }
    if (rc != 0 || nsplit <= 1) return rc;
    const int64_t nchunk = (int64_t)full.M * (full.N / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nchunk + 255) / 256)), dim3(256), 0, s, full, (const float*)full.workspace, nsplit);
    return launch_status();
}

extern "C" int insv2v_conv3x3_fuses_groupnorm(const insv2v_gemm_desc* dp) {
    if (!dp || dp->mode != INSV2V_MODE_CONV3X3 || dp->upsample || dp->split_k > 1) return 0;
    insv2v_gemm_desc d = *dp;
    if (d.batch <= 0) d.batch = 1;
    if (d.tile != 0 && d.tile != 100) return 0;
    if (halo_tw_shift(d) < 0) return 0;
    if (d.tile == 0 && (pick_split(d) > 1 || ((long)d.M / 128) * ((d.N + 127) / 128) < 200)) return 0;
    return 1;
}

#ifdef INSV2V_GEMM_PROF
extern "C" int insv2v_prof_read(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
    }
    return (int)e;
}
#endif
