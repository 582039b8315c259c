// Register-resident row kernels for the C = 320 / 640 levels of the UNet (csrc/fused_rows.hip):
//   insv2v_ffn_fused   out = x + W2 . ( h * gelu_erf(g) ) + b2,  [h; g] = W1 . LayerNorm(x) + b1  [-> proj_out + residual]   (C = 320)
//   insv2v_rowlin      out = [LayerNorm | GroupNorm](x) . W^T + bias | per-frame bias [+ residual]        (every K = 320 / 640 Linear)
//   insv2v_tattn_fused one temporal attention sub-block incl. to_out + residual (C = 320);  insv2v_tattn_attn: up to the attention output (C = 640)
//   insv2v_xattn_fused one text cross-attention sub-block incl. to_out + residual (C = 320); insv2v_xattn_attn: up to the attention output (C = 640)
// (diffusers FeedForward(geglu) behind norm3 / ff_norm: attention.py:259, motion_module.py:214; the Linear / 1x1-conv layers of
// attention.py:64,89,160-190 and motion_module.py:139,146,289-331 at the 320-channel level.)
//
// Why: at K = 320 a tile of the ordinary GEMM kernels spends as long in its prologue and global epilogue as in its five K slices
// (330-700 TFLOP/s), and the feed-forward writes and re-reads a 73 728 x 1 280 hidden tensor.  Here the
// activations never leave the register file:
//   * a wave owns 32 tokens; their 320 (normalised) channels sit in 80 VGPRs as MFMA B-operand fragments, loaded once;
//     LayerNorm statistics = in-lane sums + one cross-lane exchange (no statistics pass, no folded-LayerNorm epilogue);
//   * every weight fragment (A operand of one v_mfma_f32_32x32x16_f16: 64 lanes x 16 B = 1 KiB) comes from ONE linear fp16 stream
//     laid out on the host in exactly the order the MFMAs consume it, brought in by LDS-DMA through a ring of slots shared by the 4
//     waves of a workgroup (128 tokens, one workgroup per CU, persistent over row tiles); a fragment read is a conflict-free
//     ds_read_b128 at lane x 16; fragments are read 8 at a time, one group ahead of the MFMAs that use them;
//   * biases ride in one extra k-step against a constant fragment (ones, or the one-hot of the token's frame for the temporal
//     positional-encoding table) - no bias tables, no epilogue arithmetic;
//   * feed-forward: the hidden layer is walked in chunks of 32 units: S = W1_chunk . x (2 x 21 MFMAs), GEGLU in registers, and the
//     fp16 result IS the B operand of the second contraction O += W2_chunk . P (20 MFMAs into 160 accumulator registers): the C
//     layout of one MFMA and the B layout of the next differ only by a permutation of k that is applied to the weights on the host
//     (insv2v/fused.py).  GEGLU of chunk k is issued between the MFMAs of S for chunk k+1 (two S buffers).
// One wave per SIMD (up to 512 registers): nothing overlaps a wave's MFMAs but its own instruction stream, and measured on MI355X
// every non-MFMA instruction costs ~6 cycles that do not hide (profiles/r03_ffn_fused_ablation.txt) - hence one s_waitcnt per
// fragment group, scalar-only ring bookkeeping and 32 KiB slots (one barrier per 32 MFMAs) in the feed-forward.
// Roofline: MFMA; HBM traffic = x once in, out once out + the L2-resident weight stream.
#include "common.h"
#include "gemm_dma.h"
#include <type_traits>
#include <cstdlib>

namespace {
template <int V> using ic = std::integral_constant<int, V>;

constexpr int FC = 320;                 // channels
constexpr int KS1 = FC / 16;            // 20 k-steps over the channels (+1 bias step)
constexpr int NCHUNK = 4 * FC / 32;     // 40 chunks of 32 hidden units
constexpr int CT = FC / 32;             // 10 output channel tiles
constexpr int W1_FR = 2 * (KS1 + 1);    // 42 fragments of a chunk's first contraction (also: of a pair of output tiles of a Linear)
constexpr int W2_FR = 2 * CT;           // 20 fragments of a chunk's second contraction

__device__ __forceinline__ unsigned pk2(float a, float b) {
    const half2v h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(unsigned, h);
}

typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// ---- the weight ring.  Stream slot q (PASS_SLOTS per pass over the weights, wrapping) lives in ring slot q % NS; a slot is SLOT_FR
// fragments; wave w requests pieces w*PPS .. w*PPS+PPS-1 (1 KiB each) of every slot.  Requests run NS-1 slots ahead of the reads.
// Fragments are consumed in groups of 8, one group BEHIND their read, and every consumed group requests 2 pieces: when slot q is
// acquired, everything up to slot q + NS - 2 has been requested except the 2 pieces attached to the group consumed after the
// acquire, so slot q has landed once at most PPS (NS - 2) - 2 pieces are outstanding (loads and stores of a wave retire in issue
// order, so other memory operations in between only make this wait conservative).  All bookkeeping is wave-uniform (SALU).
template <int SLOT_FR_, int NS_, int GS_ = 8>
struct Ring {
    // GS = fragments per read / consume group (8, or 4 where a fragment feeds two MFMAs and 32 registers of read-ahead are enough);
    // a consumed group requests PPG = GS / 4 pieces
    static constexpr int SLOT_FR = SLOT_FR_, NS = NS_, GS = GS_, SLOT_B = SLOT_FR_ * 1024, PPS = SLOT_FR_ / 4, GPS = SLOT_FR_ / GS_, PPG = GS_ / 4;
    static_assert(PPS == PPG * GPS && (GS_ == 8 || GS_ == 4), "whole pieces per fragment group");
    char* smem;
    srd_t rW;
    unsigned lane16;
    int iss_lds, iss_soff, pass_bytes, wave_off, rd_off;   // byte offsets (wave-uniform)
    const char* rd;                                          // this lane's view of the slot being read

    __device__ __forceinline__ void init(char* smem_, const void* stream, int pass_slots, int wid, int lane) {
        smem = smem_;
        rW = make_srd(stream);
        lane16 = (unsigned)(lane * 16);
        wave_off = wid * PPS * 1024;
        iss_lds = 0; iss_soff = 0;
        pass_bytes = pass_slots * SLOT_B;
        rd_off = (NS - 1) * SLOT_B;
        rd = smem_;
#pragma unroll 1
        for (int s = 0; s < NS - 1; ++s) {
            pieces<0, PPS>();
            advance();
        }
    }
    // piece I of the slot being requested: pieces 0 .. 3 / 4 .. 7 share one scalar offset and one LDS base (M0), the KiB inside rides in the
    // instruction's immediate offset (added to both addresses): two scalar operations per request less
    template <int I>
    __device__ __forceinline__ void piece() {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(smem + iss_lds + wave_off + (I >> 2) * 4096), 16, lane16,
                                                 iss_soff + wave_off + (I >> 2) * 4096, (I & 3) * 1024, 0);
    }
    template <int I0, int N>
    __device__ __forceinline__ void pieces() {
        if constexpr (N > 0) { piece<I0>(); pieces<I0 + 1, N - 1>(); }
    }
    __device__ __forceinline__ void advance() {
        iss_lds = iss_lds + SLOT_B == NS * SLOT_B ? 0 : iss_lds + SLOT_B;
        iss_soff = iss_soff + SLOT_B == pass_bytes ? 0 : iss_soff + SLOT_B;
    }
    // piece `which` (0 .. PPG-1) of consumption phase ph (= consumed-group index mod GPS)
    template <int DBG, int PH, int WHICH>
    __device__ __forceinline__ void refill() {
        if (DBG & 1) return;
        piece<PPG * PH + WHICH>();
        if (WHICH == PPG - 1 && PH == GPS - 1) advance();
    }
    template <int DBG>
    __device__ __forceinline__ void acquire() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPS * (NS - 2) - PPG) : "memory");  // own pieces landed; own reads of older slots returned
        if (!(DBG & 4)) __builtin_amdgcn_s_barrier();   // everyone's pieces are in LDS; everyone is done with the previous slot
        asm volatile("" ::: "memory");
        rd_off = rd_off + SLOT_B == NS * SLOT_B ? 0 : rd_off + SLOT_B;
        rd = smem + rd_off + lane16;
    }
    __device__ __forceinline__ half8 frag(int i) const { return *(const half8*)(rd + i * 1024); }
    // group g of a section (sections start on a slot boundary) -> register buffer; acquires the slot at its first group
    template <int DBG, int G>
    __device__ __forceinline__ void read_group(half8 (&fb)[GS_]) {
        if (G % GPS == 0) acquire<DBG>();
#pragma unroll
        for (int i = 0; i < GS; ++i) fb[i] = frag((G % GPS) * GS + i);
        // keep the reads together, ahead of the MFMAs of the previous group
        if (!(DBG & 16)) __builtin_amdgcn_sched_barrier(0);
    }
};

__device__ __forceinline__ void zero16(floatx16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ---- this lane's channels of its token as B-operand fragments, 16 bytes per load: k-step s, slots 0-7 = channels 16 s + 8 half .. +7
// (the "natural" k order: fused.py packs the weights of a layer that reads its input from memory with it; layers that consume an
// MFMA result in registers use the C-layout order instead); optionally LayerNorm-ed without affine (gamma / beta live in the weights)
// gn_off != OOB: a preceding per-sample GroupNorm is applied on the fly, x <- x * scale[c] + shift[c] with (scale, shift) pairs of the row's
// sample at byte offset gn_off of rG (insv2v_groupnorm stats_only output): the normalised copy of the activations never exists.
// GroupNorm on load with the sample's (scale, shift) table staged in LDS (round 5): the wave copies the 16 KS pairs (2.5 KiB at K = 320) of
// its 32 rows' sample once per tile - 3 loads per lane instead of 4 KS per lane - and every lane reads its 8 channels per k-step from
// there (two distinct addresses per instruction: a broadcast).  `tab` = this wave's staging area, tab_off = byte offset of the sample's
// table in rG (wave-uniform), or OOB.
template <int KS>
__device__ __forceinline__ void stage_gn_table(char* tab, srd_t rG, unsigned tab_off, int lane) {
    constexpr int BYTES = 16 * KS * 8;
#pragma unroll
    for (int i = 0; i < (BYTES + 1023) / 1024; ++i) {
        const int o = i * 1024 + lane * 16;
        if (o < BYTES) {
            const uint4v v = (uint4v)__builtin_amdgcn_raw_buffer_load_b128(rG, tab_off == OOB_OFFSET ? OOB_OFFSET : tab_off + o, 0, 0);
            *(uint4v*)(tab + o) = v;
        }
    }
}
template <int KS>
__device__ __forceinline__ void gn_apply_lds(half8 (&xf)[KS], const char* tab, int half) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        floatx4 ab[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ab[j] = *(const floatx4*)(tab + half * 64 + s * 128 + j * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[s][e] = (half_t)fmaf((float)xf[s][e], ab[e >> 1][(e & 1) * 2], ab[e >> 1][(e & 1) * 2 + 1]);
    }
}
// LOAD / XFORM: the two halves separately - the row Linear requests the NEXT tile's rows before its last epilogue (round 5)
template <int KS, bool LN, bool GN = false, bool LOAD = true, bool XFORM = true>
__device__ __forceinline__ void load_rows(half8 (&xf)[KS], srd_t rX, unsigned xoff, float eps, srd_t rG = srd_t(), unsigned gn_off = 0) {
    if (LOAD) {
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = __builtin_bit_cast(half8, (uint4v)__builtin_amdgcn_raw_buffer_load_b128(rX, xoff, s * 32, 0));
    }
    if (!XFORM) return;
    if (GN) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            floatx4 ab[4];   // 8 channels x (scale, shift)
#pragma unroll
            for (int j = 0; j < 4; ++j) ab[j] = __builtin_bit_cast(floatx4, (uint4v)__builtin_amdgcn_raw_buffer_load_b128(rG, gn_off, s * 128 + j * 16, 0));
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[s][e] = (half_t)fmaf((float)xf[s][e], ab[e >> 1][(e & 1) * 2], ab[e >> 1][(e & 1) * 2 + 1]);
            // (rows already in registers - the prefetching row Linear: without a fence every (scale, shift) load is hoisted to the top, 320
            //  registers of them; four k-steps = 64 registers in flight cover the L2 latency)
            if (!LOAD && (s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!LN) return;
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += (float)xf[s][e];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / (16 * KS));
    // (opaque re-definitions between the three passes: otherwise hipcc keeps all 16 KS converted floats alive next to the packed
    //  halfs - 480 registers per 32-token block at K = 640 - instead of converting again)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xf[s]));
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)xf[s][e] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 32, 64);
    const float rstd = rsqrtf(var * (1.f / (16 * KS)) + eps);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xf[s]));
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[s][e] = (half_t)(((float)xf[s][e] - mean) * rstd);
}

// ---- one 32-channel accumulator tile -> memory, 16 bytes per lane and store.  In the MFMA C layout lane (token, half) owns channels
// 8 q + 4 half .. +3 of register quad q; v_permlane32_swap exchanges quad 2j+1 of the lower lane half with quad 2j of the upper one,
// after which the lower lane holds channels 16 j .. 16 j + 7 and the upper lane 16 j + 8 .. + 15 of its token: two 16-byte stores
// per tile instead of four 8-byte ones (row-scattered 8-byte stores are store-issue bound at ~7 B/clk/CU - this kernel's first
// version spent most of its time there).  The optional residual arrives by 16-byte loads in the same layout and is added in fp32.
// off = byte offset of (token row, channel 8 half) or OOB; soff0 = byte offset of the tile's first channel.
template <bool RES>
__device__ __forceinline__ void load_res_tile(uint4v (&rv)[2], srd_t rR, unsigned off, int soff0) {
    if (!RES) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) rv[j] = (uint4v)__builtin_amdgcn_raw_buffer_load_b128(rR, off, soff0 + j * 32, 0);
}
// Two fp32 values (+ the two fp16 residual values of one dword) -> one packed fp16 dword.  With a residual: v_fma_mixlo / mixhi_f16
// (acc * 1.0 + residual half, fp32 arithmetic, one rounding to fp16 - the values of convert + add + convert-pack, in 2 instructions per pair
// instead of 5).  ROW_EPI_LEGACY keeps the round-3 arithmetic for A/B builds.
__device__ __forceinline__ unsigned pack_res2(float a, float b, unsigned res) {
#ifdef ROW_EPI_LEGACY
    const half2v h = __builtin_bit_cast(half2v, res);
    return pk2(a + (float)h[0], b + (float)h[1]);
#else
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(res));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(b), "v"(res));
    return d;
#endif
}
// (sum x, sum x^2) of the two fp16 values of a packed dword, fp32 accumulation: two v_dot2_f32_f16 instead of 2 converts + 2 adds + 2 FMAs
__device__ __forceinline__ void stats2(unsigned o, float& s1, float& s2) {
#ifdef ROW_EPI_LEGACY
    const half2v h = __builtin_bit_cast(half2v, o);
    const float x = (float)h[0], y = (float)h[1];
    s1 += x; s2 = fmaf(x, x, s2); s1 += y; s2 = fmaf(y, y, s2);
#else
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 v = __builtin_bit_cast(f16x2, o), one = {(_Float16)1.f, (_Float16)1.f};
    s1 = __builtin_amdgcn_fdot2(v, one, s1, false);
    s2 = __builtin_amdgcn_fdot2(v, v, s2, false);
#endif
}
template <bool RES>
__device__ __forceinline__ void store_tile(const floatx16& acc, const uint4v (&rv)[2], srd_t rO, unsigned off, int soff0, float* s1 = nullptr, float* s2 = nullptr) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (scalars first: __builtin_bit_cast applied directly to a vector subscript takes element 0 with this hipcc, ROCm 7.2)
            const float alo = acc[8 * j + e], ahi = acc[8 * j + 4 + e];
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, alo), __builtin_bit_cast(unsigned, ahi), false, false);
            const unsigned lo = r[0], hi = r[1];
            v[e] = __builtin_bit_cast(float, lo);
            v[4 + e] = __builtin_bit_cast(float, hi);
        }
        uint4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned rk = rv[j][k];
            o[k] = RES ? pack_res2(v[2 * k], v[2 * k + 1], rk) : pk2(v[2 * k], v[2 * k + 1]);
        }
        if (s1) {   // LayerNorm statistics of the NEXT op, from the fp16 values being stored
#pragma unroll
            for (int k = 0; k < 4; ++k) { const unsigned ok = o[k]; stats2(ok, *s1, *s2); }
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rO, off, soff0 + j * 32, 0);
        // Keep the store's data registers untouched for a few cycles: with a second wave on the SIMD (two row-linear workgroups per
        // CU) a 16-byte store still reads part of its data when the next VALU instruction reuses the registers - the hazard found
        // in round 2 (profiles/r02_gemm_debug.md); here it showed as NaNs in the M = 73 733 test.  The "v" input pins them.
        asm volatile("s_nop 7" ::"v"(o));
    }
}

// LayerNorm (no affine) of a token's channels held as natural-order fragments (in place); see load_rows
template <int KS>
__device__ __forceinline__ void layernorm_frags(half8 (&xf)[KS], float eps) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += (float)xf[s][e];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / (16 * KS));
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xf[s]));
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = (float)xf[s][e] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 32, 64);
    const float rstd = rsqrtf(var * (1.f / (16 * KS)) + eps);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xf[s]));
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[s][e] = (half_t)(((float)xf[s][e] - mean) * rstd);
}

// one accumulator tile (+ residual) -> the two 16-byte chunks a lane would store (store_tile without the store): chunk j holds channels
// 16 j + 8 half .. + 7 of the lane's token, i.e. exactly the natural-order B fragment of k-step 2 t + j of a following contraction
template <bool RES>
__device__ __forceinline__ void finish_tile(const floatx16& acc, const uint4v (&rv)[2], half8 (&out)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float alo = acc[8 * j + e], ahi = acc[8 * j + 4 + e];
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, alo), __builtin_bit_cast(unsigned, ahi), false, false);
            const unsigned lo = r[0], hi = r[1];
            v[e] = __builtin_bit_cast(float, lo);
            v[4 + e] = __builtin_bit_cast(float, hi);
        }
        uint4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned rk = rv[j][k];
            o[k] = RES ? pack_res2(v[2 * k], v[2 * k + 1], rk) : pk2(v[2 * k], v[2 * k + 1]);
        }
        out[j] = __builtin_bit_cast(half8, o);
    }
}

template <int... I, class Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& f) { (f(ic<I>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void pack_tile(const floatx16& a, half8& k0, half8& k1) {   // C layout -> the two operand fragments (k-steps)
    const uint4v u0 = {pk2(a[0], a[1]), pk2(a[2], a[3]), pk2(a[4], a[5]), pk2(a[6], a[7])};
    const uint4v u1 = {pk2(a[8], a[9]), pk2(a[10], a[11]), pk2(a[12], a[13]), pk2(a[14], a[15])};
    k0 = __builtin_bit_cast(half8, u0);
    k1 = __builtin_bit_cast(half8, u1);
}

// ===================================================================================================== feed-forward
struct FfnArgs {
    const half_t* x;
    half_t* out;
    const half_t* wstream;
    const half_t* res2;    // POST: residual of the trailing Linear (the transformer module's input), row stride ldr2
    int64_t ldx, ldo, ldr2;
    int M;
    float eps;
};
// stream per pass, in 64-fragment sections: [b2: 10][W1(0): 42][pad 12] | stage k = 0..38: [W1(k+1): 42][W2(k): 20][pad 2] | [W2(39): 20][pad 12]
constexpr int FFN_SLOT_FR = 32, FFN_NS = 4;
constexpr int FFN_PASS_SLOTS = (64 + 64 * (NCHUNK - 1) + 32) / FFN_SLOT_FR;
// POST: the transformer module's trailing Linear (proj_out, attention.py:89 / motion_module.py:146) + its residual ride behind the feed-
// forward: out = Wp . (x + FF(LN(x))) + bp + res2.  The feed-forward result never leaves the registers: its accumulator tiles (+ x) packed
// to fp16 are the B fragments of the projection.  Stream: + [output tiles in pairs x 21 k-steps: 210][pad 14] = 7 more slots.
constexpr int FFN_POST_FR = 224, FFN_POST_SLOTS = FFN_POST_FR / FFN_SLOT_FR;
typedef unsigned uint2v __attribute__((__vector_size__(8)));   // (the 8-byte buffer-load builtin traffics in GCC-style vectors)

// DBG (timing ablations, results are garbage; selected with INSV2V_FFN_DBG, never in production): 1 = no ring refills,
// 2 = no GEGLU arithmetic, 4 = no slot barriers; valid results: 32 = round-3 schedule (GEGLU lump), 64 / 128 = see the launcher
template <int DBG, bool POST = false>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the weight ring, nothing else
    typedef Ring<FFN_SLOT_FR, (DBG & 256) ? 5 : FFN_NS> R;
    constexpr bool ILV = (DBG & (32 | 2 | 8)) == 0;   // 32: the round-3 form (GEGLU of a stage in one lump)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int ntiles = (p.M + 127) / 128;
    const srd_t rX = make_srd(p.x);
    R ring;
    ring.init(smem, p.wstream, FFN_PASS_SLOTS + (POST ? FFN_POST_SLOTS : 0), wid, lane);

    // constant B fragment of the bias k-step: k-slots 0 and 1 of the lower lane half are 1 (bias hi + lo parts)
    half8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) { ones[0] = (half_t)1.f; ones[1] = (half_t)1.f; }

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + wid * 32 + tok;
        const bool mok = m < p.M;
        const unsigned xoff = mok ? (unsigned)(((int64_t)m * p.ldx + 8 * half) * 2) : OOB_OFFSET;
        half8 xf[KS1];
        load_rows<KS1, true>(xf, rX, xoff, p.eps);

        floatx16 O[CT];
        floatx16 Sh, Sg, Nh, Ng, Nh2, Ng2;   // (Nh2 / Ng2: DBG & 8 - odd k-steps of S accumulate separately: four MFMA chains instead of two)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) zero16(O[ct]);
        zero16(Sh); zero16(Sg);
        half8 pf[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
        auto geglu = [&](const floatx16& sh, const floatx16& sg) {
            uint4v u0, u1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (DBG & 2) {
                    u0[j] = pk2(sh[2 * j] + sg[2 * j], sh[2 * j + 1] + sg[2 * j + 1]);
                    u1[j] = pk2(sh[8 + 2 * j] + sg[8 + 2 * j], sh[8 + 2 * j + 1] + sg[8 + 2 * j + 1]);
                    continue;
                }
                u0[j] = pk2(sh[2 * j] * gelu_erf_f(sg[2 * j]), sh[2 * j + 1] * gelu_erf_f(sg[2 * j + 1]));
                u1[j] = pk2(sh[8 + 2 * j] * gelu_erf_f(sg[8 + 2 * j]), sh[8 + 2 * j + 1] * gelu_erf_f(sg[8 + 2 * j + 1]));
            }
            pf[0] = __builtin_bit_cast(half8, u0);
            pf[1] = __builtin_bit_cast(half8, u1);
        };
        half8 fb[2][8];
        // what a fragment position means: kind 0 = prologue section, 1 = steady stage, 2 = final section.  (nh, ng) = the S being accumulated
        auto consume_group = [&](auto kind_, auto g_, floatx16& nh, floatx16& ng) {
            constexpr int kind = decltype(kind_)::value, g = decltype(g_)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = g * 8 + i;
                const half8 a = fb[g & 1][i];
                if (kind == 0) {                    // [b2: 10] [W1(0): 42] [pad]
                    if (f < CT) O[f < CT ? f : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ones, O[f < CT ? f : 0], 0, 0, 0);
                    else if (f < CT + W1_FR) {
                        const int w = f - CT, s = w >> 1;
                        const half8 b = s < KS1 ? xf[s < KS1 ? s : 0] : ones;
                        if (w & 1) ng = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ng, 0, 0, 0);
                        else nh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, nh, 0, 0, 0);
                    }
                } else if (kind == 1) {             // [W1(k+1): 42] [W2(k): 20] [pad 2]
                    if (f < W1_FR) {
                        const int s = f >> 1;
                        const half8 b = s < KS1 ? xf[s < KS1 ? s : 0] : ones;
                        if ((DBG & 8) && (s & 1)) {
                            if (f & 1) Ng2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Ng2, 0, 0, 0);
                            else Nh2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Nh2, 0, 0, 0);
                        } else {
                            if (f & 1) ng = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ng, 0, 0, 0);
                            else nh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, nh, 0, 0, 0);
                        }
                    } else if (f < W1_FR + W2_FR) {
                        const int j = f - W1_FR, s2 = j / CT, ct = j - s2 * CT;
                        O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[s2], O[ct], 0, 0, 0);
                    }
                } else {                            // [W2(39): 20] [pad 12]
                    if (f < W2_FR) {
                        const int s2 = f / CT, ct = f - s2 * CT;
                        O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[s2], O[ct], 0, 0, 0);
                    }
                }
                if (i == 3) ring.template refill<DBG, g % R::GPS, 0>();
                if (i == 7) ring.template refill<DBG, g % R::GPS, 1>();
            }
        };
#define RD(g) ring.template read_group<DBG, g>(fb[(g) & 1])
#define CG(kind, g, nh, ng) consume_group(ic<kind>{}, ic<g>{}, nh, ng)

        // ---- prologue section: O = b2, S(0) = W1(0) . x + b1
        RD(0);
        RD(1); CG(0, 0, Sh, Sg);
        RD(2); CG(0, 1, Sh, Sg);
        RD(3); CG(0, 2, Sh, Sg);
        RD(4); CG(0, 3, Sh, Sg);
        RD(5); CG(0, 4, Sh, Sg);
        RD(6); CG(0, 5, Sh, Sg);
        RD(7); CG(0, 6, Sh, Sg);
        // (group 7 of the prologue is padding: zeros; the first stage "consumes" it against P = 0)

        // ---- steady state: stage k = S(k+1), then O += W2(k) . P(k); the tail of W2(k) is consumed at the start of stage k+1
        if constexpr (ILV) {
            // GEGLU(k) = 8 pairs of hidden units, ONE pair per fragment group, spread between that group's MFMAs by the scheduler pipeline
            // below (~3 VALU per MFMA; the lump this replaces sat between two MFMAs with the matrix pipe idle: ~190 VALU, a quarter of
            // the kernel).  S(k) is complete behind the 2nd MFMA of stage k-1's group 5, so its pairs 0, 1 ride in groups 5, 6 of stage
            // k-1 and pairs 2 .. 7 in groups 7, 0 .. 4 of stage k; W2(k) starts in group 5.  P(k-1) is still read by stage k's group 7
            // (tail of W2(k-1)): two P buffers, and the two S sets, swap roles from stage to stage (no copies).
            uint4v PA[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, PB[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            auto gpair = [&](auto pr_, const floatx16& sh, const floatx16& sg, uint4v (&P)[2]) {
                constexpr int pr = decltype(pr_)::value, e0 = pr < 4 ? 2 * pr : 8 + 2 * (pr - 4);
                float h0 = sh[e0], h1 = sh[e0 + 1], g0 = sg[e0], g1 = sg[e0 + 1];
                // (DBG & 128: the pair's S values pinned in accumulator registers here, so their moves to the VALU side belong to this pair)
                if constexpr ((DBG & 128) != 0) asm volatile("" : "+a"(h0), "+a"(h1), "+a"(g0), "+a"(g1));
                P[pr >> 2][pr & 3] = pk2(h0 * gelu_erf_relu_f(g0), h1 * gelu_erf_relu_f(g1));
            };
            // group g of a steady stage: MFMAs of W1(k+1) into (nh, ng) / of W2 against Pw; pair pr of the GEGLU of (sh, sg) into Pg
            auto cgi = [&](auto g_, auto pr_, const floatx16& sh, const floatx16& sg, floatx16& nh, floatx16& ng, uint4v (&Pw)[2], uint4v (&Pg)[2]) {
                constexpr int g = decltype(g_)::value;
                // the eight fragments of this group were read one group ago, behind them only the eight reads just issued
                if (!(DBG & 64)) __builtin_amdgcn_s_waitcnt(0xC87F);   // lgkmcnt(8), nothing else
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = g * 8 + i;
                    const half8 a = fb[g & 1][i];
                    if (f < W1_FR) {
                        const int s = f >> 1;
                        const half8 b = s < KS1 ? xf[s < KS1 ? s : 0] : ones;
                        if (f & 1) ng = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, ng, 0, 0, 0);
                        else nh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, nh, 0, 0, 0);
                    } else if (f < W1_FR + W2_FR) {
                        const int j = f - W1_FR, s2 = j / CT, ct = j - s2 * CT;
                        O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, Pw[s2]), O[ct], 0, 0, 0);
                    }
                    if (i == 3) ring.template refill<DBG, g % R::GPS, 0>();
                    if (i == 7) ring.template refill<DBG, g % R::GPS, 1>();
                }
                // (behind the MFMAs in program order: group 5's first two complete the S its pair reads; the pipeline below places it)
                gpair(pr_, sh, sg, Pg);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            };
            // (sh, sg) = S(k), (nh, ng) <- S(k+1), Pc = P(k), Pn = P(k-1) until group 7 is through, then P(k+1)
            auto stage = [&](floatx16& sh, floatx16& sg, floatx16& nh, floatx16& ng, uint4v (&Pc)[2], uint4v (&Pn)[2]) {
                RD(0); cgi(ic<7>{}, ic<2>{}, sh, sg, nh, ng, Pn, Pc);
                zero16(nh); zero16(ng);
                RD(1); cgi(ic<0>{}, ic<3>{}, sh, sg, nh, ng, Pc, Pc);
                RD(2); cgi(ic<1>{}, ic<4>{}, sh, sg, nh, ng, Pc, Pc);
                RD(3); cgi(ic<2>{}, ic<5>{}, sh, sg, nh, ng, Pc, Pc);
                RD(4); cgi(ic<3>{}, ic<6>{}, sh, sg, nh, ng, Pc, Pc);
                RD(5); cgi(ic<4>{}, ic<7>{}, sh, sg, nh, ng, Pc, Pc);
                RD(6); cgi(ic<5>{}, ic<0>{}, nh, ng, nh, ng, Pc, Pn);
                RD(7); cgi(ic<6>{}, ic<1>{}, nh, ng, nh, ng, Pc, Pn);
            };
            gpair(ic<0>{}, Sh, Sg, PA); gpair(ic<1>{}, Sh, Sg, PA);
            static_assert((NCHUNK - 1) % 2 == 1, "steady stages: pairs + one");
#pragma unroll 1
            for (int k = 0; k < (NCHUNK - 1) / 2; ++k) { stage(Sh, Sg, Nh, Ng, PA, PB); stage(Nh, Ng, Sh, Sg, PB, PA); }
            stage(Sh, Sg, Nh, Ng, PA, PB);
            // ---- final section: tail of W2(38) against P(38) = PA, the rest of GEGLU(39), W2(39) against PB
            pf[0] = __builtin_bit_cast(half8, PA[0]); pf[1] = __builtin_bit_cast(half8, PA[1]);
            RD(0); CG(1, 7, Sh, Sg);
            gpair(ic<2>{}, Nh, Ng, PB); gpair(ic<3>{}, Nh, Ng, PB); gpair(ic<4>{}, Nh, Ng, PB);
            gpair(ic<5>{}, Nh, Ng, PB); gpair(ic<6>{}, Nh, Ng, PB); gpair(ic<7>{}, Nh, Ng, PB);
            pf[0] = __builtin_bit_cast(half8, PB[0]); pf[1] = __builtin_bit_cast(half8, PB[1]);
        } else {
#pragma unroll 1
            for (int k = 0; k < NCHUNK - 1; ++k) {
                RD(0); CG(1, 7, Nh, Ng);
                zero16(Nh); zero16(Ng);
                if (DBG & 8) { zero16(Nh2); zero16(Ng2); }
                geglu(Sh, Sg);
                RD(1); CG(1, 0, Nh, Ng);
                RD(2); CG(1, 1, Nh, Ng);
                RD(3); CG(1, 2, Nh, Ng);
                RD(4); CG(1, 3, Nh, Ng);
                RD(5); CG(1, 4, Nh, Ng);
                RD(6); CG(1, 5, Nh, Ng);
                RD(7); CG(1, 6, Nh, Ng);
                if (DBG & 8) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { Sh[r] = Nh[r] + Nh2[r]; Sg[r] = Ng[r] + Ng2[r]; }
                } else {
                    Sh = Nh; Sg = Ng;
                }
            }
            // ---- final section: tail of W2(38), GEGLU(39), W2(39)
            RD(0); CG(1, 7, Nh, Ng);
            geglu(Sh, Sg);
        }
        RD(1); CG(2, 0, Nh, Ng);
        RD(2); CG(2, 1, Nh, Ng);
        CG(2, 2, Nh, Ng);
        ring.template refill<DBG, 3 % R::GPS, 0>(); ring.template refill<DBG, 3 % R::GPS, 1>();
#undef CG
#undef RD

        const srd_t rO = make_srd(p.out);
        const unsigned ooff = mok ? (unsigned)(((int64_t)m * p.ldo + 8 * half) * 2) : OOB_OFFSET;
        if constexpr (!POST) {
            // ---- epilogue: out = O + x (raw, re-read: L2-hot)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint4v rv[2];
                load_res_tile<true>(rv, rX, xoff, ct * 64);
                store_tile<true>(O[ct], rv, rO, ooff, ct * 64);
            }
        } else {
            // ---- h = O + x in the C layout (8-byte residual pieces: channels 32 ct + 8 q + 4 half .. +3), packed: the B fragments of the
            // trailing projection (k-steps 2 ct, 2 ct + 1), into the registers that held the normalised x
            const unsigned xoff4 = mok ? (unsigned)(((int64_t)m * p.ldx + 4 * half) * 2) : OOB_OFFSET;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint2v res[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) res[q] = __builtin_amdgcn_raw_buffer_load_b64(rX, xoff4, (ct * 32 + q * 8) * 2, 0);
                floatx16 hsum;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned rlo = res[q][0], rhi = res[q][1];   // (scalars first: bit_cast on a vector subscript takes element 0)
                    const half2v r0 = __builtin_bit_cast(half2v, rlo), r1 = __builtin_bit_cast(half2v, rhi);
                    hsum[4 * q] = O[ct][4 * q] + (float)r0[0]; hsum[4 * q + 1] = O[ct][4 * q + 1] + (float)r0[1];
                    hsum[4 * q + 2] = O[ct][4 * q + 2] + (float)r1[0]; hsum[4 * q + 3] = O[ct][4 * q + 3] + (float)r1[1];
                }
                pack_tile(hsum, xf[2 * ct], xf[2 * ct + 1]);
            }
            // ---- out = Wp . h + bp + res2: output tiles in pairs (two MFMA chains), fragment f of the section = (k-step (f % 42) >> 1, tile
            // 2 (f / 42) + (f & 1)); the pipeline restarts here (one fragment-read latency per row tile)
            const srd_t rR2 = make_srd(p.res2);
            const unsigned roff2 = mok ? (unsigned)(((int64_t)m * p.ldr2 + 8 * half) * 2) : OOB_OFFSET;
            floatx16 acc0, acc1;
            uint4v resv[2][2];
            auto consume_post = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                static_for<8>([&](auto i_) {
                    constexpr int i = decltype(i_)::value, f = g * 8 + i;
                    if constexpr (f < 5 * W1_FR) {
                        constexpr int pr = f / W1_FR, s = (f % W1_FR) >> 1;
                        const half8 b = s < KS1 ? xf[s < KS1 ? s : 0] : ones;
                        if constexpr ((f & 1) == 0) {
                            if constexpr (s == 0) { zero16(acc0); load_res_tile<true>(resv[0], rR2, roff2, pr * 128); load_res_tile<true>(resv[1], rR2, roff2, pr * 128 + 64); }
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[g & 1][i], b, acc0, 0, 0, 0);
                        } else {
                            if constexpr (s == 0) zero16(acc1);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[g & 1][i], b, acc1, 0, 0, 0);
                            if constexpr (s == KS1) {
                                store_tile<true>(acc0, resv[0], rO, ooff, pr * 128);
                                store_tile<true>(acc1, resv[1], rO, ooff, pr * 128 + 64);
                            }
                        }
                    }
                    if (i == 3) ring.template refill<DBG, g % R::GPS, 0>();
                    if (i == 7) ring.template refill<DBG, g % R::GPS, 1>();
                });
            };
            constexpr int NGP = FFN_POST_FR / 8;
            ring.template read_group<DBG, 0>(fb[0]);
            static_for<NGP - 1>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                ring.template read_group<DBG, g + 1>(fb[(g + 1) & 1]);
                consume_post(ic<g>{});
            });
            consume_post(ic<NGP - 1>{});
        }
    }
    wait_vmcnt<0>();   // no LDS-DMA may land after this workgroup's LDS has been handed to another one
}

// ===================================================================================================== Linear, K = 320
struct RowLinArgs {
    const half_t* x;
    half_t* out;
    const half_t* residual;
    const half_t* wstream;
    int64_t ldx, ldo, ldr;
    int M, N;
    int rows_per_frame, frames;   // FRAME: bias row = (m / rows_per_frame) % frames (<= 16)
    float eps;
    float* stats;                 // optional [M][2] (mean, rstd) of the OUTPUT rows, for the LayerNorm that follows
    float stats_eps;
    const float* gn_ab;           // GN: [samples][K][2] (scale, shift) of a preceding GroupNorm; sample = m / gn_rows
    int gn_rows;
};
// stream per pass: per PAIR of 32-row output tiles (2p, 2p+1) one section of GP groups of 8 fragments (a whole number of 16-fragment slots):
//   [for k-step s = 0..KS: (tile 2p, tile 2p+1)] = 2 (KS + 1) fragments, then padding;  s = KS is the bias step
//   K = 320: 42 fragments in 6 groups (3 slots); K = 640: 82 fragments in 12 groups (6 slots)
constexpr int LIN_SLOT_FR = 16;
// K = 640 forms that may run two token blocks per wave: bit (LN << 2 | FRAME << 1 | RES); measured per form, profiles/r03_rowlin_tb2.txt
// Scheduler pipelines (one MFMA : N VALU) in every fragment group of the attention-block row kernels (round 5,
// profiles/r05_rows_sched_pipelines.txt): the two temporal-attention kernels gain 2.5 / 3.8 % per launch with N = 3, the text
// cross-attention kernels nothing (N = 3) or lose 1 % (N = 5)
#ifndef ROWS_SGB_TATTN
#define ROWS_SGB_TATTN 3
#endif
#ifndef ROWS_SGB_XATTN
#define ROWS_SGB_XATTN 0
#endif
#ifndef ROWS_PIN_AGPR
#define ROWS_PIN_AGPR 0
#endif
#ifndef ROWLIN_PREFETCH
#define ROWLIN_PREFETCH 1
#endif
#ifndef ROWLIN_PREFETCH_GN
#define ROWLIN_PREFETCH_GN 0
#endif
#ifndef ROWLIN_GN_LDS
#define ROWLIN_GN_LDS 1
#endif
#ifndef ROWLIN_TB2_DEFAULT
#define ROWLIN_TB2_DEFAULT 0xff
#endif
template <int KS> struct LinCfg {
    static constexpr int FR = 2 * (KS + 1);                  // fragments of a pair
    static constexpr int GP = (FR + 15) / 16 * 2;            // groups per pair section
    // K = 320: 160-250 registers suffice, so TWO workgroups share a CU (64 KiB ring each): one wave's MFMAs cover the other's fragment
    // reads, waits and ring bookkeeping - the overlap a lone wave per SIMD cannot have.  K = 640 holds 160 registers of activations:
    // one workgroup per CU with the deep ring.
    static constexpr int WGS = KS <= 20 ? 2 : 1;
    static constexpr int NS = KS <= 20 ? 4 : 9;
    // K = 640, TB = 2 (template parameter of the kernel): a wave owns TWO 32-token blocks (256 rows per workgroup), so every weight fragment
    // read from LDS feeds two MFMAs - one MFMA per fragment keeps the LDS port as busy as the matrix pipe (1 KiB per 32 cycles per SIMD)
    // and capped the kernel at ~33 % matrix utilisation; 320 activation + 64 accumulator + 64 fragment registers of the 512 a lone wave
    // per SIMD may use (the forms that also hold residual tiles spill 8-142 registers outside the pair loop and still win at 10 stacked
    // clips: +3-8 % per launch; the GroupNorm form stays at TB = 1).
};


// ring depth of a row Linear: the GroupNorm-on-load form at K = 640 gives one of its nine 16 KiB slots to the four waves' (scale, shift)
// tables (4 x 5 KiB behind the ring: 148 KiB of the CU's 160) - 160 table loads per lane and tile from L2 become 5 per lane + LDS broadcasts
template <int KS, bool GN>
constexpr int lin_ring_slots() { return (GN && ROWLIN_GN_LDS && KS > 20) ? LinCfg<KS>::NS - 1 : LinCfg<KS>::NS; }

template <int KS, bool LN, bool FRAME, bool RES, bool GN = false, int TB = 1>
__global__ __launch_bounds__(256, LinCfg<KS>::WGS) void rowlin_kernel(RowLinArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef LinCfg<KS> Cfg;
    constexpr int GS = TB == 2 ? 4 : 8;                     // fragments per read group: at TB = 2 four fragments are eight MFMAs
    typedef Ring<LIN_SLOT_FR, lin_ring_slots<KS, GN>(), GS> R;
    constexpr int GP = Cfg::GP * (8 / GS), FR = Cfg::FR, TROWS = 128 * TB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int ntiles = (p.M + TROWS - 1) / TROWS;
    const int npairs = p.N >> 6;
    R ring;
    ring.init(smem, p.wstream, npairs * (GP / R::GPS), wid, lane);

    // PF (one token block per wave): the rows of a workgroup's NEXT tile are requested right behind the last MFMA group of the current
    // one, so their HBM latency runs under the last pair's epilogue instead of in front of the next tile's first MFMA (with two token
    // blocks the 320 row registers would stay allocated through the epilogue and spill; so does the GroupNorm-on-load form - 836 bytes
    // of scratch, 2x slower - which therefore keeps its loads in front of the transform).  Measured: -2 ... -3 % per launch on the
    // LayerNorm forms (q/k/v), within noise elsewhere (profiles/r05_rowlin_prefetch.txt): the row Linears are not latency-chain bound.
    constexpr bool PF = TB == 1 && ROWLIN_PREFETCH && (!GN || ROWLIN_PREFETCH_GN);
    half8 xf[TB][KS];
    auto request_rows = [&](int t) {
        const srd_t rXn = make_srd(p.x + (int64_t)t * TROWS * p.ldx);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int ml = (wid * TB + tb) * 32 + tok;
            const unsigned xo = (t * TROWS + ml) < p.M ? (unsigned)(((int64_t)ml * p.ldx + 8 * half) * 2) : OOB_OFFSET;
            load_rows<KS, LN, GN, true, false>(xf[tb], rXn, xo, p.eps);
        }
    };
    if (PF && (int)blockIdx.x < ntiles) request_rows(blockIdx.x);
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int m[TB];
        bool mok[TB];
        unsigned xoff[TB], ooff[TB], roff[TB];
        half8 bstep[TB];
        // descriptors based at the tile's first row (64-bit), lane offsets relative to it: operands beyond 2 GiB (the fused q/k/v rows
        // of 20 stacked clips: [1 474 560, 960] fp16 = 2.8 GB) need no wider offsets
        const int64_t trow0 = (int64_t)tile * TROWS;
        const srd_t rX = make_srd(p.x + trow0 * p.ldx), rO = make_srd(p.out + trow0 * p.ldo),
                    rR = make_srd(RES ? (const void*)(p.residual + trow0 * p.ldr) : (const void*)p.x);
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int ml = (wid * TB + tb) * 32 + tok;
            m[tb] = tile * TROWS + ml;
            mok[tb] = m[tb] < p.M;
            xoff[tb] = mok[tb] ? (unsigned)(((int64_t)ml * p.ldx + 8 * half) * 2) : OOB_OFFSET;
            ooff[tb] = mok[tb] ? (unsigned)(((int64_t)ml * p.ldo + 8 * half) * 2) : OOB_OFFSET;
            roff[tb] = (RES && mok[tb]) ? (unsigned)(((int64_t)ml * p.ldr + 8 * half) * 2) : OOB_OFFSET;
            if constexpr (GN && ROWLIN_GN_LDS) {
                // a wave's 32 rows share one sample (gn_rows % 32 == 0): its table goes through this wave's 16 KS * 8 bytes behind the ring
                char* tab = smem + R::NS * R::SLOT_B + (wid * TB + tb) * (16 * KS * 8);
                const int m0w = tile * TROWS + (wid * TB + tb) * 32;
                const unsigned toff = m0w < p.M ? (unsigned)((int64_t)(m0w / p.gn_rows) * (16 * KS) * 8) : OOB_OFFSET;
                stage_gn_table<KS>(tab, make_srd(p.gn_ab), toff, lane);
                load_rows<KS, LN, false, !PF, false>(xf[tb], rX, xoff[tb], p.eps);
                gn_apply_lds<KS>(xf[tb], tab, half);
                if (LN) load_rows<KS, LN, false, false, true>(xf[tb], rX, xoff[tb], p.eps);
            } else if constexpr (GN) {
                const unsigned goff = mok[tb] ? (unsigned)((((int64_t)(m[tb] / p.gn_rows) * (16 * KS) + 8 * half) * 2) * 4) : OOB_OFFSET;
                load_rows<KS, LN, true, !PF, true>(xf[tb], rX, xoff[tb], p.eps, make_srd(p.gn_ab), goff);
            } else {
                load_rows<KS, LN, false, !PF, true>(xf[tb], rX, xoff[tb], p.eps);
            }
            // The row fragments are B operands of every MFMA of the tile.  Pinned in ACCUMULATOR registers (gfx950 MFMAs read A / B from
            // either file): left to itself the allocator spills part of them to AGPRs and moves a fragment back with four
            // v_accvgpr_read in front of every MFMA that uses it - ten times per fragment at N = 640 (round 5, ROWS_PIN_AGPR).
            // MEASURED SLOWER and off: the pair loop loses its 4 reads per MFMA pair, yet the plain form gains 2 %, the residual form loses
            // 9 %, the LayerNorm form 38 % (profiles/r05_rows_pin_agpr.txt) - an MFMA whose B operand AND accumulator come from the
            // accumulator file is not the cheaper instruction it looks like, and 320 v_accvgpr_write per tile are not free
            if constexpr (ROWS_PIN_AGPR != 0 && (TB == 2 || (ROWS_PIN_AGPR & 2))) {
#pragma unroll
                for (int s = 0; s < KS; ++s) asm volatile("" : "+a"(xf[tb][s]));
            }
            // B fragment of the bias k-step: ones in slots 0, 1 of the lower half (bias hi + lo), or the one-hot of the token's frame
            // (slot = frame & 7 of lane half frame >> 3) against the per-frame table
            bstep[tb] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
            if (FRAME) {
                const int fr = mok[tb] ? (m[tb] / p.rows_per_frame) % p.frames : 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) bstep[tb][e] = (half == (fr >> 3) && e == (fr & 7)) ? (half_t)1.f : (half_t)0.f;
            } else if (half == 0) {
                bstep[tb][0] = (half_t)1.f; bstep[tb][1] = (half_t)1.f;
            }
        }

        floatx16 acc0[TB], acc1[TB];
        uint4v resv[TB][2][2];
        half8 fb[2][GS];
        // group g of a pair section: fragments GS g .. GS g + GS - 1; fragment f = (k-step f >> 1, tile f & 1) for f < FR.  TB = 2: every
        // weight fragment read from LDS feeds TWO MFMAs (the wave's two 32-token blocks)
        auto consume_group = [&](auto g_) {
            constexpr int g = decltype(g_)::value;
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int f = g * GS + i;
                if (f < FR) {
                    const int s = f >> 1;
#pragma unroll
                    for (int tb = 0; tb < TB; ++tb) {
                        const half8 b = s < KS ? xf[tb][s < KS ? s : 0] : bstep[tb];
                        if (f == 0) zero16(acc0[tb]);
                        if (f == 1) zero16(acc1[tb]);
                        if (f & 1) acc1[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[g & 1][i], b, acc1[tb], 0, 0, 0);
                        else acc0[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[g & 1][i], b, acc0[tb], 0, 0, 0);
                    }
                }
                if (i == 3) ring.template refill<0, g % R::GPS, 0>();
                if (GS == 8 && i == 7) ring.template refill<0, g % R::GPS, 1>();
            }
        };
        auto prefetch_res = [&](int pair) {
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                load_res_tile<RES>(resv[tb][0], rR, roff[tb], pair * 128);
                load_res_tile<RES>(resv[tb][1], rR, roff[tb], pair * 128 + 64);
            }
        };
        float st1[TB], st2[TB];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) { st1[tb] = 0.f; st2[tb] = 0.f; }
        const bool want_stats = p.stats != nullptr;   // wave-uniform
        auto epilogue = [&](int pair) {   // tiles 2 pair, 2 pair + 1
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                store_tile<RES>(acc0[tb], resv[tb][0], rO, ooff[tb], pair * 128, want_stats ? &st1[tb] : nullptr, want_stats ? &st2[tb] : nullptr);
                store_tile<RES>(acc1[tb], resv[tb][1], rO, ooff[tb], pair * 128 + 64, want_stats ? &st1[tb] : nullptr, want_stats ? &st2[tb] : nullptr);
            }
        };
#pragma unroll 1
        for (int pr = 0; pr < npairs; ++pr) {
            ring.template read_group<0, 0>(fb[0]);
            if (pr > 0) { consume_group(ic<GP - 1>{}); epilogue(pr - 1); }   // (the pass's first pair has no predecessor; its refill phase
                                                                            //  is the one of the final consume_group below)
            prefetch_res(pr);
            static_for<GP - 1>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                ring.template read_group<0, g + 1>(fb[(g + 1) & 1]);
                consume_group(ic<g>{});
            });
        }
        consume_group(ic<GP - 1>{});
        if (PF && tile + (int)gridDim.x < ntiles) request_rows(tile + gridDim.x);
        epilogue(npairs - 1);
        if (want_stats) {   // every output element of a token was stored by exactly one of its two lanes
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const float s1 = st1[tb] + __shfl_xor(st1[tb], 32, 64), s2 = st2[tb] + __shfl_xor(st2[tb], 32, 64);
                const float mean = s1 / p.N;
                if (half == 0 && mok[tb]) ((float2*)p.stats)[m[tb]] = make_float2(mean, rsqrtf(fmaxf(s2 / p.N - mean * mean, 0.f) + p.stats_eps));
            }
        }
    }
    wait_vmcnt<0>();
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    }
    return n;
}

template <class Args>
int launch_rows(const void* kernel, bool& attr_set, int lds, const Args& args, int M, hipStream_t s, int wgs_per_cu = 1) {
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ncu = num_cus() * wgs_per_cu;
    if (ncu <= 0) return INSV2V_EINVAL;
    const int ntiles = (M + 127) / 128;
    Args a = args;
    void* kargs[] = {&a};
    hipError_t le = hipLaunchKernel(kernel, dim3(ntiles < ncu ? ntiles : ncu), dim3(256), kargs, lds, s);
    if (le != hipSuccess) return (int)le;
    return launch_status();
}

}  // namespace

extern "C" int insv2v_ffn_fused(const insv2v_ffn_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    const insv2v_ffn_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || d.M <= 0) return INSV2V_EINVAL;
    if (d.C != FC || d.hidden != 4 * FC) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 7) || (d.ldo & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.out & 15) || ((uintptr_t)d.wstream & 15)) return INSV2V_EINVAL;
    if ((int64_t)d.M * d.ldx * 2 >= ((int64_t)1 << 31) || (int64_t)d.M * d.ldo * 2 >= ((int64_t)1 << 31)) return INSV2V_EUNSUPPORTED;
    static const int dbg = getenv("INSV2V_FFN_DBG") ? atoi(getenv("INSV2V_FFN_DBG")) : 0;
    // 0 = production; 1 / 2 / 4 / 7 / 8 / 16 / 24 = timing ablations and scheduling variants (tools/bench_ffn.py, profiles/); 32 = GEGLU in one lump per
    // stage (round 3), 64 = no explicit fragment wait, 128 = S values pinned in accumulator registers per pair (profiles/r05_ffn_interleaved_geglu.txt)
    static const void* kernels[13] = {(const void*)ffn_fused_kernel<0>, (const void*)ffn_fused_kernel<1>, (const void*)ffn_fused_kernel<2>, (const void*)ffn_fused_kernel<4>,
                                      (const void*)ffn_fused_kernel<7>, (const void*)ffn_fused_kernel<8>, (const void*)ffn_fused_kernel<16>, (const void*)ffn_fused_kernel<24>,
                                      (const void*)ffn_fused_kernel<32>, (const void*)ffn_fused_kernel<64>, (const void*)ffn_fused_kernel<128>, (const void*)ffn_fused_kernel<5>,
                                      (const void*)ffn_fused_kernel<256>};
    static const int codes[13] = {0, 1, 2, 4, 7, 8, 16, 24, 32, 64, 128, 5, 256};
    int v = 0;
    for (int i = 0; i < 13; ++i) if (codes[i] == dbg) v = i;
    static bool attr_set[13] = {};
    const FfnArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.wstream, (const half_t*)d.post_residual, d.ldx, d.ldo, d.ld_post, d.M, d.eps};
    if (d.post) {
        if (!d.post_residual || (d.ld_post & 7) || ((uintptr_t)d.post_residual & 15) || (int64_t)d.M * d.ld_post * 2 >= ((int64_t)1 << 31)) return INSV2V_EINVAL;
        static bool post_attr = false;
        return launch_rows((const void*)ffn_fused_kernel<0, true>, post_attr, FFN_NS * FFN_SLOT_FR * 1024, a, d.M, as_stream(stream));
    }
    return launch_rows(kernels[v], attr_set[v], (codes[v] == 256 ? 5 : FFN_NS) * FFN_SLOT_FR * 1024, a, d.M, as_stream(stream));
}

// Size in fp16 elements of the weight stream insv2v_ffn_fused expects for (C, hidden); 0 if unsupported.
extern "C" int64_t insv2v_ffn_stream_elems(int32_t C, int32_t hidden, int32_t post) {
    if (C != FC || hidden != 4 * FC) return 0;
    return (int64_t)(FFN_PASS_SLOTS + (post ? FFN_POST_SLOTS : 0)) * FFN_SLOT_FR * 512;
}

template <int KS>
static int launch_rowlin(const insv2v_rowlin_desc& d, const RowLinArgs& a, hipStream_t s) {
    const int v = (d.layernorm ? 4 : 0) | (d.frame_bias ? 2 : 0) | (d.residual ? 1 : 0);
    static const void* kernels[8] = {(const void*)rowlin_kernel<KS, false, false, false>, (const void*)rowlin_kernel<KS, false, false, true>,
                                     (const void*)rowlin_kernel<KS, false, true, false>, (const void*)rowlin_kernel<KS, false, true, true>,
                                     (const void*)rowlin_kernel<KS, true, false, false>, (const void*)rowlin_kernel<KS, true, false, true>,
                                     (const void*)rowlin_kernel<KS, true, true, false>, (const void*)rowlin_kernel<KS, true, true, true>};
    static bool attr_set[8] = {};
    if (d.gn_ab) {   // fused input GroupNorm: only the plain form (proj_in of the transformer blocks) exists
        if (v != 0) return INSV2V_EUNSUPPORTED;
        static bool gn_attr = false;
        return launch_rows((const void*)rowlin_kernel<KS, false, false, false, true>, gn_attr, lin_ring_slots<KS, true>() * LIN_SLOT_FR * 1024 + (ROWLIN_GN_LDS ? 4 * 16 * KS * 8 : 0), a, d.M, s, LinCfg<KS>::WGS);
    }
    if constexpr (KS == 40) {   // two token blocks per wave where the register file holds them: bit v of the mask (INSV2V_ROWLIN_TB2 overrides, for A/B)
        static const int tb2 = getenv("INSV2V_ROWLIN_TB2") ? atoi(getenv("INSV2V_ROWLIN_TB2")) : ROWLIN_TB2_DEFAULT;
        static const void* k2[8] = {(const void*)rowlin_kernel<KS, false, false, false, false, 2>, (const void*)rowlin_kernel<KS, false, false, true, false, 2>,
                                    (const void*)rowlin_kernel<KS, false, true, false, false, 2>, (const void*)rowlin_kernel<KS, false, true, true, false, 2>,
                                    (const void*)rowlin_kernel<KS, true, false, false, false, 2>, (const void*)rowlin_kernel<KS, true, false, true, false, 2>,
                                    (const void*)rowlin_kernel<KS, true, true, false, false, 2>, (const void*)rowlin_kernel<KS, true, true, true, false, 2>};
        static bool attr2[8] = {};
        // only where the launch keeps >= 2 rounds of 256-row tiles (5 stacked clips at level 1 = 1.4 rounds: slower, profiles/r03_rowlin_tb2.txt)
        if (((tb2 >> v) & 1) && (d.M + 255) / 256 >= 2 * num_cus()) return launch_rows(k2[v], attr2[v], LinCfg<KS>::NS * LIN_SLOT_FR * 1024, a, (d.M + 1) / 2, s, LinCfg<KS>::WGS);
    }
    return launch_rows(kernels[v], attr_set[v], LinCfg<KS>::NS * LIN_SLOT_FR * 1024, a, d.M, s, LinCfg<KS>::WGS);
}

static bool rowlin_k_ok(int K) { return K == 320 || K == 640; }

extern "C" int insv2v_rowlin(const insv2v_rowlin_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    const insv2v_rowlin_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || d.M <= 0 || d.N <= 0) return INSV2V_EINVAL;
    if (!rowlin_k_ok(d.K) || (d.N & 63)) return INSV2V_EUNSUPPORTED;
    if (d.frame_bias && (d.rows_per_frame <= 0 || d.frames <= 0 || d.frames > 16)) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 7) || (d.ldo & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.out & 15) || ((uintptr_t)d.wstream & 15)) return INSV2V_EINVAL;
    if (d.residual && ((d.ldr & 7) || ((uintptr_t)d.residual & 15))) return INSV2V_EINVAL;
    // (descriptors are rebased per 128/256-row tile: only a tile's own extent has to fit the 2 GiB window, the operands may be larger)
    const int64_t lim = (int64_t)1 << 31;
    if (256 * (int64_t)d.ldx * 2 >= lim || 256 * (int64_t)d.ldo * 2 >= lim || (d.residual && 256 * (int64_t)d.ldr * 2 >= lim)) return INSV2V_EUNSUPPORTED;
    const RowLinArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.residual, (const half_t*)d.wstream, d.ldx, d.ldo, d.ldr,
                          d.M, d.N, d.rows_per_frame, d.frames, d.eps, d.stats_out, d.stats_eps, d.gn_ab, d.gn_rows};
    if (d.gn_ab && (d.gn_rows <= 0 || (d.gn_rows % 32) || ((uintptr_t)d.gn_ab & 15))) return INSV2V_EUNSUPPORTED;   // a wave's 32 rows share one sample
    return d.K == 320 ? launch_rowlin<20>(d, a, as_stream(stream)) : launch_rowlin<40>(d, a, as_stream(stream));
}

// fp16 elements of the weight stream insv2v_rowlin expects for a [N, K] Linear; 0 if unsupported
extern "C" int64_t insv2v_rowlin_stream_elems(int32_t N, int32_t K) {
    if (!rowlin_k_ok(K) || N <= 0 || (N & 63)) return 0;
    const int gp = K == 320 ? LinCfg<20>::GP : LinCfg<40>::GP;
    return (int64_t)(N >> 6) * gp * 8 * 512;
}

namespace {
// ===================================================================================================== temporal attention block
// insv2v_tattn_fused: one TemporalTransformerBlock attention sub-block (motion_module.py:270-336 behind the LayerNorm of :206) at
// C = 320, 8 heads x 40, 16 frames, as ONE register-resident launch:
//     out = x + Wo . Attn_over_frames( LayerNorm(x) Wqkv^T + (beta, positional-encoding) bias ) + bo
// A wave owns 2 pixels x 16 frames = 32 tokens (rows (b, f, p) of the token matrix: the frame axis is a row stride of HW).  q, k and v
// tiles come out of the MFMAs in the C layout, and every later contraction reads them as an operand IN PLACE:
//   * q^T / k^T tiles ([32 channels] x [32 tokens]) packed to fp16 are legal B / A fragments of S^T = K . Q^T for a permuted channel
//     order (any order works, it is a contraction index); a head is 40 channels = 5 "octets" (8 channels = one register quad of both
//     lane halves): two full k-steps + one half-zero k-step (only the Q side is masked);
//   * S^T is 32 keys x 32 queries: the 16 x 16 diagonal blocks are the two pixels, the rest is discarded by a register select on the
//     query's pixel; softmax over 16 keys = 8 in-lane values + one exchange with the other lane half;
//   * V is computed with the MFMA operands swapped ([tokens] x [channels]), which makes its packed tile the A fragment of
//     O^T = V^T . P^T (k = keys, in exactly the order the probabilities sit in the lane); P of the other pixel's keys is zero;
//   * O^T tiles ([channels] x [queries]) normalised by 1 / l and packed are the B fragments of the output projection.
// Weight stream (insv2v/fused.py pack_tattn_stream), 864 fragments = 54 slots per pass:
//   for head group G = 0, 1 (4 heads = 5 channel tiles): [Q/K of tile tl, k-step s: (q, k)] x 5 x 21 | [V pair (0,1)] [V pair (2,3)] [V 4] | pad 5
//   [output tiles in pairs x 21] | pad 14;   k-step 20 = bias step (per-frame table for q/k/v, plain bias for the output)
struct TattnArgs {
    const half_t* x;
    half_t* out;
    const half_t* wstream;
    int64_t ldx, ldo;
    int HW, npix;          // pixels per sample, total pixels (samples x HW); rows = npix x 16
    float eps, scale;
};
constexpr int TA_H = 8, TA_D = 40, TA_F = 16;
constexpr int TA_SEC_G = 320, TA_QK = 210, TA_VP = 84, TA_SEC_O = 224, TA_TOTAL = 2 * TA_SEC_G + TA_SEC_O;   // fragment positions
struct TaOp { int kind, G, t, s; };   // kind 0 pad, 1 Q, 2 K, 3 V (t = local tile 0..4), 4 OUT (t = output tile 0..9)
constexpr TaOp ta_op(int f) {
    if (f < 2 * TA_SEC_G) {
        const int G = f / TA_SEC_G;
        int r = f % TA_SEC_G;
        if (r < TA_QK) return {1 + (r % 42 & 1), G, r / 42, (r % 42) >> 1};
        r -= TA_QK;
        if (r < TA_VP) return {3, G, 2 * (r / 42) + (r % 42 & 1), (r % 42) >> 1};
        r -= TA_VP;
        if (r < 21) return {3, G, 4, r};
        return {0, 0, 0, 0};
    }
    const int r = f - 2 * TA_SEC_G;
    if (r < 210) return {4, 0, 2 * (r / 42) + (r % 42 & 1), (r % 42) >> 1};
    return {0, 0, 0, 0};
}


__global__ __launch_bounds__(256, 1) void tattn_fused_kernel(TattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Ring<16, 9> R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int pp = tok >> 4, fr = tok & 15;           // pixel of the wave's pair, frame
    const int ntiles = (p.npix + 7) / 8;
    const srd_t rX = make_srd(p.x), rO = make_srd(p.out);
    R ring;
    ring.init(smem, p.wstream, TA_TOTAL / 16, wid, lane);

    half8 ones = {0, 0, 0, 0, 0, 0, 0, 0}, fhot = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) { ones[0] = (half_t)1.f; ones[1] = (half_t)1.f; }
#pragma unroll
    for (int e = 0; e < 8; ++e) fhot[e] = (half == (fr >> 3) && e == (fr & 7)) ? (half_t)1.f : (half_t)0.f;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const float c2 = p.scale * 1.4426950408889634f;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int pix = tile * 8 + wid * 2 + pp;
        const bool mok = pix < p.npix;
        const int b = pix / p.HW, pl = pix - b * p.HW;
        const int64_t m = ((int64_t)b * TA_F + fr) * p.HW + pl;
        const unsigned xoff = mok ? (unsigned)((m * p.ldx + 8 * half) * 2) : OOB_OFFSET;
        const unsigned ooff = mok ? (unsigned)((m * p.ldo + 8 * half) * 2) : OOB_OFFSET;
        half8 xn[KS1];
        load_rows<KS1, true>(xn, rX, xoff, p.eps);

        half8 afr[KS1];                    // attention output, packed: the B fragments of the output projection
        half8 qs[10], ks[10];              // q / k of the current head group, packed per k-step (2 octets each)
        half8 PB[4][2];                    // probabilities of the group's 4 heads: [key k-step 0 | 1]
        float invl[4];
        floatx16 acc0, acc1;               // Q / K, V pair, output pair
        uint4v resv[2][2];
        half8 fb[2][8];

        // ---- per-group attention scores -> PB, invl
        auto scores = [&]() {
            floatx16 S[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) zero16(S[h]);
            // step 0 / 1: the two full k-steps, step 2: the single octet (Q side half-zero); heads interleaved: 4 independent MFMA chains
            static_for<3>([&](auto st_) {
                static_for<4>([&](auto h_) {
                    constexpr int st = decltype(st_)::value, h = decltype(h_)::value;
                    constexpr int lo = 5 * h;                               // first octet of the head within the group
                    if constexpr (st < 2) {
                        constexpr int kst = (lo & 1) ? (lo + 1) / 2 + st : lo / 2 + st;
                        S[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[kst], qs[kst], S[h], 0, 0, 0);
                    } else {
                        constexpr int o = (lo & 1) ? lo : lo + 4;           // the unpaired octet
                        constexpr int kst = o >> 1;
                        uint4v u = __builtin_bit_cast(uint4v, qs[kst]);
                        if (o & 1) { u[0] = 0; u[1] = 0; } else { u[2] = 0; u[3] = 0; }
                        S[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[kst], __builtin_bit_cast(half8, u), S[h], 0, 0, 0);
                    }
                });
            });
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float sel[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float s0 = S[h][j], s1 = S[h][8 + j]; sel[j] = pp ? s1 : s0; }
                float mx = sel[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sel[j]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mc = -mx * c2;
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(fmaf(sel[j], c2, mc));
                const uint4v u = {pk2(e[0], e[1]), pk2(e[2], e[3]), pk2(e[4], e[5]), pk2(e[6], e[7])};
                const half8 pe = __builtin_bit_cast(half8, u);
                float l = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) l += (float)pe[j];             // the ROUNDED probabilities, as the P.V MFMAs see them
                l += __shfl_xor(l, 32, 64);
                invl[h] = 1.f / l;
                PB[h][0] = pp ? zero8 : pe;
                PB[h][1] = pp ? pe : zero8;
            }
        };
        // ---- O^T of local tile tl of group G from its packed V tile -> afr
        auto pv_tile = [&](auto G_, auto tl_, const floatx16& accV) {
            constexpr int G = decltype(G_)::value, tl = decltype(tl_)::value;
            constexpr int ha = (4 * tl) / 5, hb = (4 * tl + 3) / 5;
            half8 v0, v1;
            pack_tile(accV, v0, v1);
            floatx16 Oa, Ob;
            zero16(Oa);
            Oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, PB[ha][0], Oa, 0, 0, 0);
            Oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, PB[ha][1], Oa, 0, 0, 0);
            if (hb != ha) {
                zero16(Ob);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, PB[hb][0], Ob, 0, 0, 0);
                Ob = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, PB[hb][1], Ob, 0, 0, 0);
            }
            floatx16 o;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int h = (4 * tl + qd) / 5;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * qd + e] = (h == ha ? Oa[4 * qd + e] : Ob[4 * qd + e]) * invl[h];
            }
            pack_tile(o, afr[2 * (5 * G + tl)], afr[2 * (5 * G + tl) + 1]);
        };

        auto consume_group = [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            static_for<8>([&](auto i_) {
                constexpr int i = decltype(i_)::value, f = g * 8 + i;
                constexpr TaOp op = ta_op(f);
                const half8 a = fb[g & 1][i];
                if constexpr (op.kind == 1 || op.kind == 2) {          // q / k tile of the group: A = weights, B = tokens
                    const half8 bop = op.s < KS1 ? xn[op.s < KS1 ? op.s : 0] : fhot;
                    if constexpr (op.kind == 1) {
                        if (op.s == 0) zero16(acc0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                    } else {
                        if (op.s == 0) zero16(acc1);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                        if constexpr (op.s == KS1) {                    // tile complete
                            pack_tile(acc0, qs[2 * op.t], qs[2 * op.t + 1]);
                            pack_tile(acc1, ks[2 * op.t], ks[2 * op.t + 1]);
                            if constexpr (op.t == 4) scores();
                        }
                    }
                } else if constexpr (op.kind == 3) {                    // v tile, operands swapped: A = tokens, B = weights -> [token][channel]
                    const half8 aop = op.s < KS1 ? xn[op.s < KS1 ? op.s : 0] : fhot;
                    constexpr bool second = op.t == 1 || op.t == 3;
                    if constexpr (second) {
                        if (op.s == 0) zero16(acc1);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, a, acc1, 0, 0, 0);
                        if constexpr (op.s == KS1) { pv_tile(ic<op.G>{}, ic<op.t - 1>{}, acc0); pv_tile(ic<op.G>{}, ic<op.t>{}, acc1); }
                    } else {
                        if (op.s == 0) zero16(acc0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, a, acc0, 0, 0, 0);
                        if constexpr (op.s == KS1 && op.t == 4) pv_tile(ic<op.G>{}, ic<4>{}, acc0);
                    }
                } else if constexpr (op.kind == 4) {                    // output projection, tiles in pairs
                    const half8 bop = op.s < KS1 ? afr[op.s < KS1 ? op.s : 0] : ones;
                    if constexpr ((op.t & 1) == 0) {
                        if (op.s == 0) { zero16(acc0); load_res_tile<true>(resv[0], rX, xoff, op.t * 64); load_res_tile<true>(resv[1], rX, xoff, op.t * 64 + 64); }
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                    } else {
                        if (op.s == 0) zero16(acc1);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                        if constexpr (op.s == KS1) {
                            store_tile<true>(acc0, resv[0], rO, ooff, (op.t - 1) * 64);
                            store_tile<true>(acc1, resv[1], rO, ooff, op.t * 64);
                        }
                    }
                }
                if (i == 3) ring.template refill<0, g % R::GPS, 0>();
                if (i == 7) ring.template refill<0, g % R::GPS, 1>();
            });
#if ROWS_SGB_TATTN
            // (round 5 experiment) one MFMA, then up to ROWS_SGB_TATTN of whatever VALU work sits in this group's region, eight times
            _Pragma("unroll") for (int i_sgb = 0; i_sgb < 8; ++i_sgb) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, ROWS_SGB_TATTN, 0); }
#endif
        };
        constexpr int NG = TA_TOTAL / 8;   // 108 groups per pass
        ring.template read_group<0, 0>(fb[0]);
        static_for<NG - 1>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            ring.template read_group<0, g + 1>(fb[(g + 1) & 1]);
            consume_group(ic<g>{});
        });
        consume_group(ic<NG - 1>{});
    }
    wait_vmcnt<0>();
}

}  // namespace

extern "C" int insv2v_tattn_fused(const insv2v_tattn_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    const insv2v_tattn_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || d.samples <= 0 || d.HW <= 0) return INSV2V_EINVAL;
    if (d.C != FC || d.heads != TA_H || d.frames != TA_F) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 7) || (d.ldo & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.out & 15) || ((uintptr_t)d.wstream & 15)) return INSV2V_EINVAL;
    const int64_t rows = (int64_t)d.samples * TA_F * d.HW;
    if (rows * d.ldx * 2 >= ((int64_t)1 << 31) || rows * d.ldo * 2 >= ((int64_t)1 << 31)) return INSV2V_EUNSUPPORTED;
    const TattnArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.wstream, d.ldx, d.ldo, d.HW, d.samples * d.HW, d.eps, d.scale};
    static bool attr_set = false;
    // launch_rows sizes the grid from a row count in 128-row tiles: a tile here is 8 pixels x 16 frames = 128 rows
    return launch_rows((const void*)tattn_fused_kernel, attr_set, 9 * 16 * 1024, a, (int)((int64_t)a.npix * 16 > 0x7fffffff ? 0x7fffffff : a.npix * 16), as_stream(stream));
}

extern "C" int64_t insv2v_tattn_stream_elems(int32_t C, int32_t heads, int32_t frames) {
    if (C != FC || heads != TA_H || frames != TA_F) return 0;
    return (int64_t)TA_TOTAL * 512;
}

namespace {
// ===================================================================================================== temporal attention, C = 640
// insv2v_tattn_attn: LayerNorm -> (+pe) -> q/k/v -> attention over the 16 frames of every pixel at C = 640 (8 heads x 80), WITHOUT the output
// projection: 640 channels of activations (160 registers) + the packed attention output for a K = 640 projection (160 more) do not fit next
// to the working set, so the attention output [rows, 640] goes to memory and insv2v_rowlin adds to_out + residual.  q, k and v (a
// [rows, 1920] tensor written and re-read per block before) never exist in memory.  Same scheme as tattn_fused_kernel; a head is 80
// channels = 5 whole k-steps of a 160-channel group (2 heads per group, 4 groups), so nothing is masked.  One group's weights = 624
// fragments = 39 ring slots; the group loop is a run-time loop around one unrolled group body (the instruction stream of four would not
// stay in the instruction cache).
// Stream per group G (channel tiles c = 5G .. 5G+4): [tile c: k-step s = 0..40: (q, k)] x 5 | [v pair (5G, 5G+1)] [v pair (5G+2, 5G+3)] [v 5G+4] | pad 9
constexpr int TB_KS = 40, TB_GROUP_FR = 624, TB_QK = 410, TB_VP = 164;
struct TbOp { int kind, t, s; };   // kind 0 pad, 1 Q, 2 K, 3 V (t = local tile 0..4)
constexpr TbOp tb_op(int r) {
    if (r < TB_QK) return {1 + (r % 82 & 1), r / 82, (r % 82) >> 1};
    r -= TB_QK;
    if (r < TB_VP) return {3, 2 * (r / 82) + (r % 82 & 1), (r % 82) >> 1};
    r -= TB_VP;
    if (r < 41) return {3, 4, r};
    return {0, 0, 0};
}

__global__ __launch_bounds__(256, 1) void tattn640_kernel(TattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Ring<16, 9> R;
    constexpr int KS = TB_KS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int pp = tok >> 4, fr = tok & 15;
    const int ntiles = (p.npix + 7) / 8;
    const srd_t rX = make_srd(p.x), rO = make_srd(p.out);
    R ring;
    ring.init(smem, p.wstream, 4 * TB_GROUP_FR / 16, wid, lane);

    half8 fhot = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) fhot[e] = (half == (fr >> 3) && e == (fr & 7)) ? (half_t)1.f : (half_t)0.f;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const float c2 = p.scale * 1.4426950408889634f;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int pix = tile * 8 + wid * 2 + pp;
        const bool mok = pix < p.npix;
        const int b = pix / p.HW, pl = pix - b * p.HW;
        const int64_t m = ((int64_t)b * TA_F + fr) * p.HW + pl;
        const unsigned xoff = mok ? (unsigned)((m * p.ldx + 8 * half) * 2) : OOB_OFFSET;
        const unsigned ooff = mok ? (unsigned)((m * p.ldo + 8 * half) * 2) : OOB_OFFSET;
        half8 xn[KS];
        load_rows<KS, true>(xn, rX, xoff, p.eps);

#pragma unroll 1
        for (int G = 0; G < 4; ++G) {
            half8 qs[10], ks[10];
            half8 PB[2][2];
            float invl[2];
            floatx16 acc0, acc1;
            const uint4v nores[2] = {};
            half8 fb[2][8];

            auto scores = [&]() {
                floatx16 S[2];
                zero16(S[0]); zero16(S[1]);
                static_for<5>([&](auto st_) {   // heads interleaved: two independent MFMA chains
                    constexpr int st = decltype(st_)::value;
                    S[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[st], qs[st], S[0], 0, 0, 0);
                    S[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ks[5 + st], qs[5 + st], S[1], 0, 0, 0);
                });
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float sel[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float s0 = S[h][j], s1 = S[h][8 + j]; sel[j] = pp ? s1 : s0; }
                    float mx = sel[0];
#pragma unroll
                    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, sel[j]);
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const float mc = -mx * c2;
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(fmaf(sel[j], c2, mc));
                    const uint4v u = {pk2(e[0], e[1]), pk2(e[2], e[3]), pk2(e[4], e[5]), pk2(e[6], e[7])};
                    const half8 pe = __builtin_bit_cast(half8, u);
                    float l = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) l += (float)pe[j];
                    l += __shfl_xor(l, 32, 64);
                    invl[h] = 1.f / l;
                    PB[h][0] = pp ? zero8 : pe;
                    PB[h][1] = pp ? pe : zero8;
                }
            };
            // O^T of local tile tl from its packed V tile -> memory (channels 160 G + 32 tl ..)
            auto pv_tile = [&](auto tl_, const floatx16& accV) {
                constexpr int tl = decltype(tl_)::value;
                constexpr int ha = (32 * tl) / 80, hb = (32 * tl + 31) / 80;
                half8 v0, v1;
                pack_tile(accV, v0, v1);
                floatx16 Oa, Ob;
                zero16(Oa);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, PB[ha][0], Oa, 0, 0, 0);
                Oa = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, PB[ha][1], Oa, 0, 0, 0);
                if (hb != ha) {
                    zero16(Ob);
                    Ob = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, PB[hb][0], Ob, 0, 0, 0);
                    Ob = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, PB[hb][1], Ob, 0, 0, 0);
                }
                floatx16 o;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int h = (32 * tl + 8 * qd) / 80;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[4 * qd + e] = (h == ha ? Oa[4 * qd + e] : Ob[4 * qd + e]) * invl[h];
                }
                store_tile<false>(o, nores, rO, ooff, (160 * G + 32 * tl) * 2);
            };

            auto consume_group = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                static_for<8>([&](auto i_) {
                    constexpr int i = decltype(i_)::value, f = g * 8 + i;
                    constexpr TbOp op = tb_op(f);
                    const half8 a = fb[g & 1][i];
                    if constexpr (op.kind == 1 || op.kind == 2) {          // q / k tile of the group: A = weights, B = tokens
                        const half8 bop = op.s < KS ? xn[op.s < KS ? op.s : 0] : fhot;
                        if constexpr (op.kind == 1) {
                            if (op.s == 0) zero16(acc0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                        } else {
                            if (op.s == 0) zero16(acc1);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                            if constexpr (op.s == KS) {
                                pack_tile(acc0, qs[2 * op.t], qs[2 * op.t + 1]);
                                pack_tile(acc1, ks[2 * op.t], ks[2 * op.t + 1]);
                                if constexpr (op.t == 4) scores();
                            }
                        }
                    } else if constexpr (op.kind == 3) {                    // v tile, operands swapped: [token][channel]
                        const half8 aop = op.s < KS ? xn[op.s < KS ? op.s : 0] : fhot;
                        constexpr bool second = op.t == 1 || op.t == 3;
                        if constexpr (second) {
                            if (op.s == 0) zero16(acc1);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, a, acc1, 0, 0, 0);
                            if constexpr (op.s == KS) { pv_tile(ic<op.t - 1>{}, acc0); pv_tile(ic<op.t>{}, acc1); }
                        } else {
                            if (op.s == 0) zero16(acc0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aop, a, acc0, 0, 0, 0);
                            if constexpr (op.s == KS && op.t == 4) pv_tile(ic<4>{}, acc0);
                        }
                    }
                    if (i == 3) ring.template refill<0, g % R::GPS, 0>();
                    if (i == 7) ring.template refill<0, g % R::GPS, 1>();
                });
#if ROWS_SGB_TATTN
                // (round 5 experiment) one MFMA, then up to ROWS_SGB_TATTN of whatever VALU work sits in this group's region, eight times
                _Pragma("unroll") for (int i_sgb = 0; i_sgb < 8; ++i_sgb) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, ROWS_SGB_TATTN, 0); }
#endif
            };
            constexpr int NG = TB_GROUP_FR / 8;   // 78 groups per head group
            ring.template read_group<0, 0>(fb[0]);
            static_for<NG - 1>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                ring.template read_group<0, g + 1>(fb[(g + 1) & 1]);
                consume_group(ic<g>{});
            });
            consume_group(ic<NG - 1>{});
        }
    }
    wait_vmcnt<0>();
}

}  // namespace

extern "C" int insv2v_tattn_attn(const insv2v_tattn_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    const insv2v_tattn_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || d.samples <= 0 || d.HW <= 0) return INSV2V_EINVAL;
    if (d.C != 640 || d.heads != 8 || d.frames != TA_F) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 7) || (d.ldo & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.out & 15) || ((uintptr_t)d.wstream & 15)) return INSV2V_EINVAL;
    const int64_t rows = (int64_t)d.samples * TA_F * d.HW;
    if (rows * d.ldx * 2 >= ((int64_t)1 << 31) || rows * d.ldo * 2 >= ((int64_t)1 << 31)) return INSV2V_EUNSUPPORTED;
    const TattnArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.wstream, d.ldx, d.ldo, d.HW, d.samples * d.HW, d.eps, d.scale};
    static bool attr_set = false;
    return launch_rows((const void*)tattn640_kernel, attr_set, 9 * 16 * 1024, a, (int)((int64_t)a.npix * 16 > 0x7fffffff ? 0x7fffffff : a.npix * 16), as_stream(stream));
}

extern "C" int64_t insv2v_tattn_attn_stream_elems(int32_t C, int32_t heads, int32_t frames) {
    if (C != 640 || heads != 8 || frames != TA_F) return 0;
    return (int64_t)4 * TB_GROUP_FR * 512;
}

namespace {
// ===================================================================================================== cross-attention block
// insv2v_xattn_fused: the text cross-attention sub-block of BasicTransformerBlock (attention.py:249-257: norm2 -> attn2 + residual) at
// C = 320, 8 heads x 40, up to 96 text tokens, as ONE register-resident launch:
//     out = x + Wo . Attn( LayerNorm(x) Wq^T + Wq beta ;  K_b, V_b ) + bo          (b = the sample of the token row)
// The text K / V of a sample are only 2 x 77 x 320 halfs and loop-invariant over the sampling loop, so they are a second WEIGHT STREAM:
// insv2v/fused.py pack_xattn_kv lays them out per sample as MFMA A fragments, masked per head, in the order consumed here, and the
// ring pulls them through LDS between the shared q-projection and output-projection weights (a 128-row tile lies inside one sample).
//   * q tiles ([32 channels] x [32 tokens], C layout) packed to fp16 are the B fragments of S^T = K_h . Q_h^T: a head is 40 channels = 5
//     octets = two full k-steps + one half k-step whose other octet is ZERO IN THE K FRAGMENT (no masking in the kernel);
//   * S^T is [96 keys] x [32 tokens] per head (3 accumulator tiles): softmax over the keys = in-lane over 48 values + one exchange with the
//     other lane half; keys >= ctx_len get an additive -1e30; the NORMALISED probabilities packed to fp16 are the B fragments (k = keys)
//     of O^T += V_h^T . P^T, with V_h^T fragments zero outside the head's channels so the group's 5 output tiles simply accumulate;
//   * O^T tiles packed are the B fragments of the output projection; residual and store as in the row Linears.
// Stream per tile, 39 slots of 16 fragments: [Q: 5 tile pairs x 21 k-steps, pad 14] [per sample: 8 heads x (9 K + 12 V), pad 8]
// [OUT: 5 tile pairs x 21, pad 14].  The per-sample slots are requested 8 slots ahead like all others, which is still inside the tile.
struct XattnArgs {
    const half_t* x;
    half_t* out;
    const half_t* wstream;
    const half_t* kvstream;
    int64_t ldx, ldo;
    int M, rows_per_sample, ctx_len;
    float eps, scale;
    const half_t* pre_res;   // PRE: residual of the leading Linear (x is then the self-attention output), row stride ld_pre
    int64_t ld_pre;
};
constexpr int XA_Q_FR = 224, XA_KV_FR = 176, XA_O_FR = 224, XA_TOTAL = XA_Q_FR + XA_KV_FR + XA_O_FR, XA_SLOTS = XA_TOTAL / 16;
constexpr int XA_QS = XA_Q_FR / 16, XA_KVS = XA_KV_FR / 16;
// PRE: the out-projection of the preceding self-attention (attention.py:244-247: hidden = attn1(norm1(hidden)) + hidden) rides in front:
// x1 = Wo1 . a + bo1 + h never leaves the registers - its finished tiles (finish_tile) ARE the natural-order fragments the LayerNorm and the
// q projection read, and the raw copy is the residual of the block's output.  Stream: + [output tiles in pairs x 21: 210][pad 14] = 14 slots.
constexpr int XA_PRE_FR = 224;
struct XaOp { int kind, a, b, c; };   // 0 pad | 1 Q (tile a, k-step b) | 2 K (head a of 8, key tile b, step c) | 3 V (head a, tile select b, key k-step c) | 4 OUT (tile a, k-step b) | 5 PRE (tile a, k-step b)
template <bool PRE>
constexpr XaOp xa_op(int f) {
    if (PRE) {
        if (f < 210) return {5, 2 * (f / 42) + (f % 42 & 1), (f % 42) >> 1, 0};
        if (f < XA_PRE_FR) return {0, 0, 0, 0};
        f -= XA_PRE_FR;
    }
    if (f < XA_Q_FR) {
        if (f < 210) return {1, 2 * (f / 42) + (f % 42 & 1), (f % 42) >> 1, 0};
        return {0, 0, 0, 0};
    }
    f -= XA_Q_FR;
    if (f < XA_KV_FR) {
        if (f >= 168) return {0, 0, 0, 0};
        const int gh = f / 21, r = f % 21;
        if (r < 9) return {2, gh, r % 3, r / 3};
        return {3, gh, (r - 9) & 1, (r - 9) >> 1};
    }
    f -= XA_KV_FR;
    if (f < 210) return {4, 2 * (f / 42) + (f % 42 & 1), (f % 42) >> 1, 0};
    return {0, 0, 0, 0};
}
// group-local k-step (16 channels of the 160-channel head group) of step st of head h: see tattn_fused_kernel::scores
constexpr int xa_kstep(int h, int st) {
    const int lo = 5 * h;
    if (st < 2) return (lo & 1) ? (lo + 1) / 2 + st : lo / 2 + st;
    return ((lo & 1) ? lo : lo + 4) >> 1;
}

// Ring<16, 9> whose SOURCE is resolved per compile-time stream slot: the shared weights or the tile's per-sample K / V
template <int NPRE>   // slots of a leading shared-weight section in front of the q section
struct XRing {
    static constexpr int SLOT_FR = 16, NS = 9, SLOT_B = SLOT_FR * 1024, PPS = SLOT_FR / 4, GPS = SLOT_FR / 8;
    char* smem;
    srd_t rW, rKV;
    unsigned lane16;
    int iss_lds, wave_off, rd_off, kv_soff;
    const char* rd;
    template <int SLOT>
    __device__ __forceinline__ void piece(int i) {
        constexpr bool kv = SLOT >= NPRE + XA_QS && SLOT < NPRE + XA_QS + XA_KVS;
        constexpr int base = (kv ? SLOT - NPRE - XA_QS : (SLOT < NPRE + XA_QS ? SLOT : SLOT - XA_KVS)) * SLOT_B;
        if (kv) dma16(rKV, lane16, kv_soff + base + wave_off + i * 1024, smem + iss_lds + wave_off + i * 1024);
        else dma16(rW, lane16, base + wave_off + i * 1024, smem + iss_lds + wave_off + i * 1024);
    }
    __device__ __forceinline__ void advance() { iss_lds = iss_lds + SLOT_B == NS * SLOT_B ? 0 : iss_lds + SLOT_B; }
    __device__ __forceinline__ void init(char* smem_, const void* w, const void* kvs, int wid, int lane) {
        smem = smem_;
        rW = make_srd(w);
        rKV = make_srd(kvs);
        lane16 = (unsigned)(lane * 16);
        wave_off = wid * PPS * 1024;
        iss_lds = 0; kv_soff = 0;
        rd_off = (NS - 1) * SLOT_B;
        rd = smem_;
        static_for<NS - 1>([&](auto s_) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) piece<decltype(s_)::value>(i);
            advance();
        });
    }
    template <int SLOT>
    __device__ __forceinline__ void refill(int ph, int which) {
        piece<SLOT>(2 * ph + which);
        if (which == 1 && ph == GPS - 1) advance();
    }
    __device__ __forceinline__ void acquire() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPS * (NS - 2) - 2) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rd_off = rd_off + SLOT_B == NS * SLOT_B ? 0 : rd_off + SLOT_B;
        rd = smem + rd_off + lane16;
    }
    template <int G>
    __device__ __forceinline__ void read_group(half8 (&fb)[8]) {
        if (G % GPS == 0) acquire();
#pragma unroll
        for (int i = 0; i < 8; ++i) fb[i] = *(const half8*)(rd + ((G % GPS) * 8 + i) * 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
};

template <bool PRE>
__global__ __launch_bounds__(256, 1) void xattn_fused_kernel(XattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPRE = PRE ? XA_PRE_FR / 16 : 0, TOTAL = XA_TOTAL + (PRE ? XA_PRE_FR : 0), SLOTS = TOTAL / 16;
    typedef XRing<NPRE> RingT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int ntiles = (p.M + 127) / 128;
    const srd_t rX = make_srd(p.x), rO = make_srd(p.out), rH = make_srd(PRE ? (const void*)p.pre_res : (const void*)p.x);
    RingT ring;
    ring.init(smem, p.wstream, p.kvstream, wid, lane);

    half8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) { ones[0] = (half_t)1.f; ones[1] = (half_t)1.f; }
    const float c2 = p.scale * 1.4426950408889634f;
    // additive key mask of the third key tile (keys 64 + (r & 3) + 8 (r >> 2) + 4 half): ctx_len is in (64, 96]
    float kmask[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) kmask[r] = (64 + (r & 3) + 8 * (r >> 2) + 4 * half) < p.ctx_len ? 0.f : -1.0e30f;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + wid * 32 + tok;
        const bool mok = m < p.M;
        const unsigned xoff = mok ? (unsigned)(((int64_t)m * p.ldx + 8 * half) * 2) : OOB_OFFSET;
        const unsigned ooff = mok ? (unsigned)(((int64_t)m * p.ldo + 8 * half) * 2) : OOB_OFFSET;
        ring.kv_soff = __builtin_amdgcn_readfirstlane((tile * 128) / p.rows_per_sample) * (XA_KV_FR * 1024);
        const unsigned hoff = (PRE && mok) ? (unsigned)(((int64_t)m * p.ld_pre + 8 * half) * 2) : OOB_OFFSET;
        half8 xn[KS1];                     // PRE: first the self-attention output (operand of the leading Linear), then LayerNorm(x1)
        half8 x1[PRE ? KS1 : 1];           // PRE: x1 = leading Linear + residual, raw: the residual of the block's output
        load_rows<KS1, !PRE>(xn, rX, xoff, p.eps);

        half8 qs[KS1];                     // q of all 10 channel tiles, packed per k-step
        half8 afr[KS1];                    // attention output, packed: the B fragments of the output projection
        half8 P[6];                        // normalised probabilities of the current head: key k-steps 0..5
        floatx16 S[3], O[5];
        floatx16 acc0, acc1;
        uint4v resv[2][2];
        half8 fb[2][8];

        auto softmax = [&]() {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[2][r] += kmask[r];
            float mx = S[0][0];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mc = -mx * c2;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(S[kt][r], c2, mc)); S[kt][r] = e; l += e; }
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.f / l;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[kt][r] *= inv;
                pack_tile(S[kt], P[2 * kt], P[2 * kt + 1]);
            }
        };

        auto consume_group = [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            constexpr int islot = (g / RingT::GPS + RingT::NS - 1) % SLOTS;   // the stream slot whose pieces this group requests
            static_for<8>([&](auto i_) {
                constexpr int i = decltype(i_)::value, f = g * 8 + i;
                constexpr XaOp op = xa_op<PRE>(f);
                const half8 a = fb[g & 1][i];
                if constexpr (op.kind == 5) {                           // PRE: x1 = Wo1 . a + bo1 + h, tiles in pairs, finished into fragments
                    const half8 bop = op.b < KS1 ? xn[op.b < KS1 ? op.b : 0] : ones;
                    if constexpr ((op.a & 1) == 0) {
                        if (op.b == 0) { zero16(acc0); load_res_tile<true>(resv[0], rH, hoff, op.a * 64); load_res_tile<true>(resv[1], rH, hoff, op.a * 64 + 64); }
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                    } else {
                        if (op.b == 0) zero16(acc1);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                        if constexpr (op.b == KS1) {
                            half8 t0[2], t1[2];
                            finish_tile<true>(acc0, resv[0], t0);
                            finish_tile<true>(acc1, resv[1], t1);
                            x1[PRE ? 2 * (op.a - 1) : 0] = t0[0]; x1[PRE ? 2 * (op.a - 1) + 1 : 0] = t0[1];
                            x1[PRE ? 2 * op.a : 0] = t1[0]; x1[PRE ? 2 * op.a + 1 : 0] = t1[1];
                            if constexpr (op.a == 9) {              // all of x1 is there: it replaces the operand, normalised
#pragma unroll
                                for (int k = 0; k < KS1; ++k) xn[k] = x1[PRE ? k : 0];
                                layernorm_frags<KS1>(xn, p.eps);
                            }
                        }
                    }
                } else if constexpr (op.kind == 1) {                           // q projection, tiles in pairs
                    const half8 bop = op.b < KS1 ? xn[op.b < KS1 ? op.b : 0] : ones;
                    if constexpr ((op.a & 1) == 0) {
                        if (op.b == 0) zero16(acc0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                    } else {
                        if (op.b == 0) zero16(acc1);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                        if constexpr (op.b == KS1) {
                            pack_tile(acc0, qs[2 * (op.a - 1)], qs[2 * (op.a - 1) + 1]);
                            pack_tile(acc1, qs[2 * op.a], qs[2 * op.a + 1]);
                        }
                    }
                } else if constexpr (op.kind == 2) {                    // scores of head op.a: S^T[key tile op.b] += K . Q^T
                    constexpr int G = op.a >> 2, h = op.a & 3, kst = 10 * G + xa_kstep(h, op.c);
                    if (op.c == 0) zero16(S[op.b]);
                    S[op.b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qs[kst], S[op.b], 0, 0, 0);
                    if constexpr (op.c == 2 && op.b == 2) softmax();
                } else if constexpr (op.kind == 3) {                    // O^T[tile] += V_h^T . P^T
                    constexpr int G = op.a >> 2, h = op.a & 3, t = (40 * h) / 32 + op.b;
                    if constexpr (h == 0 && op.b == 0 && op.c == 0) {
#pragma unroll
                        for (int q = 0; q < 5; ++q) zero16(O[q]);
                    }
                    O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, P[op.c], O[t], 0, 0, 0);
                    if constexpr (h == 3 && op.b == 1 && op.c == 5) {
#pragma unroll
                        for (int q = 0; q < 5; ++q) pack_tile(O[q], afr[2 * (5 * G + q)], afr[2 * (5 * G + q) + 1]);
                    }
                } else if constexpr (op.kind == 4) {                    // output projection + residual, tiles in pairs
                    const half8 bop = op.b < KS1 ? afr[op.b < KS1 ? op.b : 0] : ones;
                    if constexpr ((op.a & 1) == 0) {
                        if (op.b == 0) {
                            zero16(acc0);
                            if (PRE) {   // the residual x1 never left the registers: its fragments are the 16-byte chunks store_tile adds
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    resv[0][j] = __builtin_bit_cast(uint4v, x1[PRE ? 2 * op.a + j : 0]);
                                    resv[1][j] = __builtin_bit_cast(uint4v, x1[PRE ? 2 * op.a + 2 + j : 0]);
                                }
                            } else {
                                load_res_tile<true>(resv[0], rX, xoff, op.a * 64); load_res_tile<true>(resv[1], rX, xoff, op.a * 64 + 64);
                            }
                        }
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                    } else {
                        if (op.b == 0) zero16(acc1);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                        if constexpr (op.b == KS1) {
                            store_tile<true>(acc0, resv[0], rO, ooff, (op.a - 1) * 64);
                            store_tile<true>(acc1, resv[1], rO, ooff, op.a * 64);
                        }
                    }
                }
                if (i == 3) ring.template refill<islot>(g % RingT::GPS, 0);
                if (i == 7) ring.template refill<islot>(g % RingT::GPS, 1);
            });
#if ROWS_SGB_XATTN
            // (round 5 experiment) one MFMA, then up to ROWS_SGB_XATTN of whatever VALU work sits in this group's region, eight times
            _Pragma("unroll") for (int i_sgb = 0; i_sgb < 8; ++i_sgb) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, ROWS_SGB_XATTN, 0); }
#endif
        };
        constexpr int NG = TOTAL / 8;   // 78 (PRE: 106) groups per tile
        ring.template read_group<0>(fb[0]);
        static_for<NG - 1>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            ring.template read_group<g + 1>(fb[(g + 1) & 1]);
            consume_group(ic<g>{});
        });
        consume_group(ic<NG - 1>{});
    }
    wait_vmcnt<0>();
}

}  // namespace

extern "C" int insv2v_xattn_fused(const insv2v_xattn_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    const insv2v_xattn_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || !d.kvstream || d.M <= 0 || d.rows_per_sample <= 0) return INSV2V_EINVAL;
    if (d.C != FC || d.heads != 8 || d.ctx_len <= 64 || d.ctx_len > 96) return INSV2V_EUNSUPPORTED;
    if ((d.rows_per_sample % 128) || (d.M % d.rows_per_sample)) return INSV2V_EUNSUPPORTED;   // a workgroup's 128 rows share one sample's K / V
    if ((d.ldx & 7) || (d.ldo & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.out & 15) || ((uintptr_t)d.wstream & 15) || ((uintptr_t)d.kvstream & 15)) return INSV2V_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)d.M * d.ldx * 2 >= lim || (int64_t)d.M * d.ldo * 2 >= lim || (int64_t)(d.M / d.rows_per_sample) * XA_KV_FR * 1024 >= lim) return INSV2V_EUNSUPPORTED;
    const XattnArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.wstream, (const half_t*)d.kvstream, d.ldx, d.ldo, d.M, d.rows_per_sample,
                         d.ctx_len, d.eps, d.scale, (const half_t*)d.pre_residual, d.ld_pre};
    static bool attr_set = false;
    if (d.pre_residual) {
        if ((d.ld_pre & 7) || ((uintptr_t)d.pre_residual & 15) || (int64_t)d.M * d.ld_pre * 2 >= lim) return INSV2V_EINVAL;
        static bool pre_attr = false;
        return launch_rows((const void*)xattn_fused_kernel<true>, pre_attr, XRing<0>::NS * XRing<0>::SLOT_B, a, d.M, as_stream(stream));
    }
    return launch_rows((const void*)xattn_fused_kernel<false>, attr_set, XRing<0>::NS * XRing<0>::SLOT_B, a, d.M, as_stream(stream));
}

// fp16 elements of the shared weight stream (q + output projections) and of ONE sample's K / V stream; 0 if unsupported
extern "C" int64_t insv2v_xattn_stream_elems(int32_t C, int32_t heads, int32_t per_sample_kv) {   // per_sample_kv: 0 = shared weights, 1 = one sample's K / V, 2 = shared weights with the leading Linear
    if (C != FC || heads != 8) return 0;
    return (int64_t)(per_sample_kv == 1 ? XA_KV_FR : XA_Q_FR + XA_O_FR + (per_sample_kv == 2 ? XA_PRE_FR : 0)) * 512;
}

namespace {
// ===================================================================================================== cross-attention, C = 640
// insv2v_xattn_attn: LayerNorm -> q -> attention over the sample's text tokens at C = 640 (8 heads x 80), WITHOUT the output projection
// (same register argument as tattn640_kernel): the attention output [rows, 640] goes to memory, to_out + residual follow as insv2v_rowlin.
// A head = 5 whole k-steps of a 160-channel group; per group the ring pulls 13 slots of q weights (5 tiles x 41 k-steps) and 5 slots of the
// sample's K / V fragments (2 heads x (15 K + 18 V)); the group loop is a run-time loop around one unrolled group body.
constexpr int XB_Q_FR = 208, XB_KV_FR = 80, XB_GROUP_FR = XB_Q_FR + XB_KV_FR, XB_QS = XB_Q_FR / 16, XB_KVS = XB_KV_FR / 16, XB_SLOTS = XB_GROUP_FR / 16;
struct XbOp { int kind, a, b, c; };   // 0 pad | 1 Q (tile a, k-step b) | 2 K (head a, key tile b, step c) | 3 V (head a, tile select b, key k-step c)
constexpr XbOp xb_op(int f) {
    if (f < XB_Q_FR) {
        if (f < 164) return {1, 2 * (f / 82) + (f % 82 & 1), (f % 82) >> 1, 0};
        if (f < 205) return {1, 4, f - 164, 0};
        return {0, 0, 0, 0};
    }
    const int r = f - XB_Q_FR;
    if (r >= 66) return {0, 0, 0, 0};
    const int h = r / 33, q = r % 33;
    if (q < 15) return {2, h, q % 3, q / 3};
    return {3, h, (q - 15) % 3, (q - 15) / 3};
}

struct XbRing {   // Ring<16, 9>; source per stream slot: q weights of group Gi or the tile's sample K / V of group Gi (resolved from a static slot + run-time group)
    static constexpr int SLOT_FR = 16, NS = 9, SLOT_B = SLOT_FR * 1024, PPS = SLOT_FR / 4, GPS = SLOT_FR / 8;
    char* smem;
    srd_t rW, rKV;
    unsigned lane16;
    int iss_lds, wave_off, rd_off, kv_soff;
    const char* rd;
    template <int SLOT>   // SLOT in [0, XB_SLOTS): slot of group Gi
    __device__ __forceinline__ void piece(int i, int Gi) {
        constexpr bool kv = SLOT >= XB_QS;
        if (kv) dma16(rKV, lane16, kv_soff + (Gi * XB_KVS + SLOT - XB_QS) * SLOT_B + wave_off + i * 1024, smem + iss_lds + wave_off + i * 1024);
        else dma16(rW, lane16, (Gi * XB_QS + SLOT) * SLOT_B + wave_off + i * 1024, smem + iss_lds + wave_off + i * 1024);
    }
    __device__ __forceinline__ void advance() { iss_lds = iss_lds + SLOT_B == NS * SLOT_B ? 0 : iss_lds + SLOT_B; }
    __device__ __forceinline__ void init(char* smem_, const void* w, const void* kvs, int wid, int lane) {
        smem = smem_;
        rW = make_srd(w);
        rKV = make_srd(kvs);
        lane16 = (unsigned)(lane * 16);
        wave_off = wid * PPS * 1024;
        iss_lds = 0; kv_soff = 0;
        rd_off = (NS - 1) * SLOT_B;
        rd = smem_;
        static_for<NS - 1>([&](auto s_) {   // slots 0 .. 7 of group 0: q weights
#pragma unroll
            for (int i = 0; i < PPS; ++i) piece<decltype(s_)::value>(i, 0);
            advance();
        });
    }
    __device__ __forceinline__ void acquire() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPS * (NS - 2) - 2) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rd_off = rd_off + SLOT_B == NS * SLOT_B ? 0 : rd_off + SLOT_B;
        rd = smem + rd_off + lane16;
    }
    template <int G>
    __device__ __forceinline__ void read_group(half8 (&fb)[8]) {
        if (G % GPS == 0) acquire();
#pragma unroll
        for (int i = 0; i < 8; ++i) fb[i] = *(const half8*)(rd + ((G % GPS) * 8 + i) * 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
};

__global__ __launch_bounds__(256, 1) void xattn640_kernel(XattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 40;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int ntiles = (p.M + 127) / 128;
    const srd_t rX = make_srd(p.x), rO = make_srd(p.out);
    XbRing ring;
    ring.init(smem, p.wstream, p.kvstream, wid, lane);

    half8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) { ones[0] = (half_t)1.f; ones[1] = (half_t)1.f; }
    const float c2 = p.scale * 1.4426950408889634f;
    float kmask[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) kmask[r] = (64 + (r & 3) + 8 * (r >> 2) + 4 * half) < p.ctx_len ? 0.f : -1.0e30f;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + wid * 32 + tok;
        const bool mok = m < p.M;
        const unsigned xoff = mok ? (unsigned)(((int64_t)m * p.ldx + 8 * half) * 2) : OOB_OFFSET;
        const unsigned ooff = mok ? (unsigned)(((int64_t)m * p.ldo + 8 * half) * 2) : OOB_OFFSET;
        ring.kv_soff = __builtin_amdgcn_readfirstlane((tile * 128) / p.rows_per_sample) * (4 * XB_KV_FR * 1024);
        half8 xn[KS];
        load_rows<KS, true>(xn, rX, xoff, p.eps);

#pragma unroll 1
        for (int G = 0; G < 4; ++G) {
            half8 qs[10];
            half8 P[6];
            floatx16 S[3], O[5];
            floatx16 acc0, acc1;
            const uint4v nores[2] = {};
            half8 fb[2][8];

            auto softmax = [&]() {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[2][r] += kmask[r];
                float mx = S[0][0];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[kt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mc = -mx * c2;
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(fmaf(S[kt][r], c2, mc)); S[kt][r] = e; l += e; }
                l += __shfl_xor(l, 32, 64);
                const float inv = 1.f / l;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[kt][r] *= inv;
                    pack_tile(S[kt], P[2 * kt], P[2 * kt + 1]);
                }
            };

            auto consume_group = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                constexpr int ahead = g / XbRing::GPS + XbRing::NS - 1, islot = ahead % XB_SLOTS;   // slot requested by this group: of group G or G + 1
                const int Gi = (G + (ahead >= XB_SLOTS ? 1 : 0)) & 3;                               // (wraps into the next tile's group 0: q weights only)
                static_for<8>([&](auto i_) {
                    constexpr int i = decltype(i_)::value, f = g * 8 + i;
                    constexpr XbOp op = xb_op(f);
                    const half8 a = fb[g & 1][i];
                    if constexpr (op.kind == 1) {                           // q projection of the group's 5 tiles: pairs (0,1), (2,3), then 4
                        const half8 bop = op.b < KS ? xn[op.b < KS ? op.b : 0] : ones;
                        if constexpr ((op.a & 1) == 0) {
                            if (op.b == 0) zero16(acc0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc0, 0, 0, 0);
                            if constexpr (op.a == 4 && op.b == KS) pack_tile(acc0, qs[8], qs[9]);
                        } else {
                            if (op.b == 0) zero16(acc1);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bop, acc1, 0, 0, 0);
                            if constexpr (op.b == KS) {
                                pack_tile(acc0, qs[2 * (op.a - 1)], qs[2 * (op.a - 1) + 1]);
                                pack_tile(acc1, qs[2 * op.a], qs[2 * op.a + 1]);
                            }
                        }
                    } else if constexpr (op.kind == 2) {                    // scores of head op.a: S^T[key tile op.b] += K . Q^T over its 5 k-steps
                        if (op.c == 0) zero16(S[op.b]);
                        S[op.b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qs[5 * op.a + op.c], S[op.b], 0, 0, 0);
                        if constexpr (op.c == 4 && op.b == 2) softmax();
                    } else if constexpr (op.kind == 3) {                    // O^T[tile] += V_h^T . P^T; head 0: tiles 0-2, head 1: tiles 2-4
                        constexpr int t = 2 * op.a + op.b;
                        if constexpr (op.a == 0 && op.b == 0 && op.c == 0) {
#pragma unroll
                            for (int q = 0; q < 5; ++q) zero16(O[q]);
                        }
                        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, P[op.c], O[t], 0, 0, 0);
                        if constexpr (op.a == 1 && op.b == 2 && op.c == 5) {
#pragma unroll
                            for (int q = 0; q < 5; ++q) store_tile<false>(O[q], nores, rO, ooff, (160 * G + 32 * q) * 2);
                        }
                    }
                    if (i == 3) { ring.template piece<islot>(2 * (g % XbRing::GPS), Gi); }
                    if (i == 7) { ring.template piece<islot>(2 * (g % XbRing::GPS) + 1, Gi); if (g % XbRing::GPS == XbRing::GPS - 1) ring.advance(); }
                });
#if ROWS_SGB_XATTN
                // (round 5 experiment) one MFMA, then up to ROWS_SGB_XATTN of whatever VALU work sits in this group's region, eight times
                _Pragma("unroll") for (int i_sgb = 0; i_sgb < 8; ++i_sgb) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, ROWS_SGB_XATTN, 0); }
#endif
            };
            constexpr int NG = XB_GROUP_FR / 8;   // 36 groups per head group
            ring.template read_group<0>(fb[0]);
            static_for<NG - 1>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                ring.template read_group<g + 1>(fb[(g + 1) & 1]);
                consume_group(ic<g>{});
            });
            consume_group(ic<NG - 1>{});
        }
    }
    wait_vmcnt<0>();
}

}  // namespace

extern "C" int insv2v_xattn_attn(const insv2v_xattn_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    const insv2v_xattn_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || !d.kvstream || d.M <= 0 || d.rows_per_sample <= 0) return INSV2V_EINVAL;
    if (d.C != 640 || d.heads != 8 || d.ctx_len <= 64 || d.ctx_len > 96) return INSV2V_EUNSUPPORTED;
    if ((d.rows_per_sample % 128) || (d.M % d.rows_per_sample)) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 7) || (d.ldo & 7) || ((uintptr_t)d.x & 15) || ((uintptr_t)d.out & 15) || ((uintptr_t)d.wstream & 15) || ((uintptr_t)d.kvstream & 15)) return INSV2V_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)d.M * d.ldx * 2 >= lim || (int64_t)d.M * d.ldo * 2 >= lim || (int64_t)(d.M / d.rows_per_sample) * 4 * XB_KV_FR * 1024 >= lim) return INSV2V_EUNSUPPORTED;
    const XattnArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.wstream, (const half_t*)d.kvstream, d.ldx, d.ldo, d.M, d.rows_per_sample,
                         d.ctx_len, d.eps, d.scale};
    static bool attr_set = false;
    return launch_rows((const void*)xattn640_kernel, attr_set, XbRing::NS * XbRing::SLOT_B, a, d.M, as_stream(stream));
}

// fp16 elements of the q weight stream / of ONE sample's K / V stream of insv2v_xattn_attn; 0 if unsupported
extern "C" int64_t insv2v_xattn_attn_stream_elems(int32_t C, int32_t heads, int32_t per_sample_kv) {
    if (C != 640 || heads != 8) return 0;
    return (int64_t)4 * (per_sample_kv ? XB_KV_FR : XB_Q_FR) * 512;
}
