// insv2v_ffn_fused: LayerNorm -> GEGLU feed-forward -> + residual as ONE kernel for C = 320 (UNet level 0):
//     out = x + W2 . ( h * gelu_erf(g) ) + b2,   [h; g] = W1 . LayerNorm(x) + b1
// (diffusers FeedForward(geglu) behind norm3 / ff_norm: attention.py:259, motion_module.py:214).
//
// Why: as three launches (row statistics, FF1 + GEGLU, FF2 + residual) the level-0 feed-forward writes and re-reads its
// 73 728 x 1 280 hidden tensor (189 MB per branch triple) and runs at 530-710 TFLOP/s because each short-K tile pays a
// prologue and a global epilogue.  Here the activations never leave the register file:
//   * a wave owns 32 tokens; their 320 normalised channels sit in 80 VGPRs as MFMA B-operand fragments (loaded once,
//     LayerNorm statistics = in-lane sums + one cross-lane exchange);
//   * the hidden layer is walked in chunks of 32 units: S = W1_chunk . x (2 x 21 v_mfma_f32_32x32x16_f16, biases ride in a
//     21st k-step against a constant "ones" fragment), GEGLU in registers, and the fp16 result IS the B operand of the
//     second contraction O += W2_chunk . P (20 MFMAs into 160 accumulator registers) - the C layout of one MFMA and
//     the B layout of the next differ only by a permutation of k that is applied to the weights on the host;
//   * GEGLU of chunk k is issued between the MFMAs of S for chunk k+1 (software pipeline, two S buffers);
//   * LDS holds nothing but the weight stream: one linear fp16 buffer in exactly the order the MFMAs consume it
//     (1 KiB fragment = 64 lanes x 16 B), brought in by LDS-DMA through a ring of 16 KiB slots shared by the 4 waves
//     (128 tokens per workgroup, one workgroup per CU, persistent over row tiles); a fragment read is a conflict-free
//     ds_read_b128 at lane x 16.
// Roofline: MFMA (181 GFLOP per 73 728 tokens); HBM traffic = x once in, out once out (94 MB) + the L2-resident 2.6 MB stream.
#include "common.h"
#include "gemm_dma.h"
#include <type_traits>
#include <cstdlib>

namespace {
template <int V> using ic = std::integral_constant<int, V>;

constexpr int FC = 320;                 // channels
constexpr int KS1 = FC / 16;            // 20 k-steps of the first contraction (+1 bias step)
constexpr int NCHUNK = 4 * FC / 32;     // 40 chunks of 32 hidden units
constexpr int CT = FC / 32;             // 10 output channel tiles
constexpr int SLOT_FR = 16;             // fragments per ring slot
constexpr int SLOT_B = SLOT_FR * 1024;
constexpr int NS = 9;                   // ring slots (144 KiB)
constexpr int W1_FR = 2 * (KS1 + 1);    // 42 fragments of a chunk's first contraction
constexpr int W2_FR = 2 * CT;           // 20 fragments of a chunk's second contraction
// stream layout per pass (slots): [b2: 1] [W1(0): 3] [stage k = 0..38: W1(k+1) + W2(k): 4 each] [W2(39): 2]
constexpr int PASS_SLOTS = 1 + 3 + 4 * (NCHUNK - 1) + 2;

struct FfnArgs {
    const half_t* x;
    half_t* out;
    const half_t* wstream;
    int64_t ldx, ldo;
    int M;
    float eps;
};

__device__ __forceinline__ unsigned pk2(float a, float b) {
    const half2v h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(unsigned, h);
}

// (the 8-byte buffer load / store builtins traffic in GCC-style vectors)
typedef unsigned uint2v __attribute__((__vector_size__(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// DBG (timing ablations, results are garbage; selected with INSV2V_FFN_DBG, never in production): 1 = no ring refills,
// 2 = no GEGLU arithmetic, 4 = no slot barriers
template <int DBG>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the weight ring, nothing else
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, half = lane >> 5;
    const int ntiles = (p.M + 127) / 128;

    const srd_t rW = make_srd(p.wstream), rX = make_srd(p.x);
    // ---- weight ring: stream slot q (of PASS_SLOTS per pass, wrapping) lives in ring slot q % NS; wave w requests
    // fragments 4w .. 4w+3 of every slot.  issue_q / its ring position and stream offset are wave-uniform scalars.
    int iss_ring = 0, iss_pass = 0;         // ring slot / slot-within-pass of the NEXT slot to request
    auto issue_piece = [&](int i) {         // piece i (0..3) of this wave for the slot being requested
        dma16(rW, (unsigned)(lane * 16), iss_pass * SLOT_B + (wid * 4 + i) * 1024, smem + iss_ring * SLOT_B + (wid * 4 + i) * 1024);
    };
    auto issue_advance = [&]() {
        iss_ring = iss_ring + 1 == NS ? 0 : iss_ring + 1;
        iss_pass = iss_pass + 1 == PASS_SLOTS ? 0 : iss_pass + 1;
    };
    int rd_ring = NS - 1;                   // ring slot being read (advanced by acquire)
    const char* rd = smem;
    // acquire the next slot.  Requests run NS-1 slots ahead of the reads MINUS the two pieces whose refill is attached to the group
    // consumed after this acquire (groups are consumed one step behind their read): when slot q is acquired, the slots up to
    // q + NS - 2 have been requested except the last 2 pieces of the newest, so slot q has landed once at most
    // 4 (NS - 2) - 2 pieces are outstanding.
    auto acquire = [&]() {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (NS - 2) - 2) : "memory");  // own pieces landed; own reads of older slots returned
        if (!(DBG & 4)) __builtin_amdgcn_s_barrier();       // everyone's pieces are in LDS; everyone is done with the previous slot
        asm volatile("" ::: "memory");
        rd_ring = rd_ring + 1 == NS ? 0 : rd_ring + 1;
        rd = smem + rd_ring * SLOT_B + lane * 16;
    };
    auto frag = [&](int i) { return *(const half8*)(rd + i * 1024); };
    // the slot vacated by the previous acquire is refilled piecewise, between the MFMAs of the current slot
    auto refill = [&](int i) { if (DBG & 1) return; issue_piece(i); if (i == 3) issue_advance(); };

    // prologue: NS-1 slots in flight
#pragma unroll 1
    for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(i);
        issue_advance();
    }

    // constant B fragment of the bias k-step: k-slots 0 and 1 of the lower lane half are 1 (bias hi + lo parts)
    half8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
    if (half == 0) { ones[0] = (half_t)1.f; ones[1] = (half_t)1.f; }

#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + wid * 32 + tok;
        const bool mok = m < p.M;
        // ---- this lane's 160 channels of its token: k-step s, slots 0-3 = channels 16s + 4 half .. +3, slots 4-7 = the same + 8
        const unsigned xoff = mok ? (unsigned)(((int64_t)m * p.ldx + 4 * half) * 2) : OOB_OFFSET;
        half8 xf[KS1];
        {
            uint2v raw[KS1][2];
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                raw[s][0] = __builtin_amdgcn_raw_buffer_load_b64(rX, xoff, s * 32, 0);
                raw[s][1] = __builtin_amdgcn_raw_buffer_load_b64(rX, xoff, s * 32 + 16, 0);
            }
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const uint4v u = {raw[s][0][0], raw[s][0][1], raw[s][1][0], raw[s][1][1]};
                xf[s] = __builtin_bit_cast(half8, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)xf[s][e];
            }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.f / FC);
            float var = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xf[s][e] - mean; var = fmaf(d, d, var); }
            var += __shfl_xor(var, 32, 64);
            const float rstd = rsqrtf(var * (1.f / FC) + p.eps);
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[s][e] = (half_t)(((float)xf[s][e] - mean) * rstd);
        }

        floatx16 O[CT];
        floatx16 Sh, Sg, Nh, Ng;
        auto zero16 = [](floatx16& a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
        };
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
        // Fragments are read 8 at a time (half a slot) into one of two register buffers, one group AHEAD of the MFMAs that
        // consume them: the reads of group g+1 are issued before the MFMAs of group g, so a fragment's LDS latency hides behind
        // eight MFMAs.  A slot is acquired before its first group is read; its two groups carry the refill pieces 0,1 / 2,3.
        half8 fb[2][8];
        auto read_group = [&](auto g_) {           // group g of the current section -> buffer g & 1
            constexpr int g = decltype(g_)::value;
            if ((g & 1) == 0) acquire();
#pragma unroll
            for (int i = 0; i < 8; ++i) fb[g & 1][i] = frag((g & 1) * 8 + i);
        };
        // what a fragment position means: kind 0 = prologue section, 1 = steady stage, 2 = final section
        auto consume_group = [&](auto kind_, auto g_) {
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
                        if (w & 1) Sg = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Sg, 0, 0, 0);
                        else Sh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Sh, 0, 0, 0);
                    }
                } else if (kind == 1) {             // [W1(k+1): 42] [W2(k): 20] [pad 2]
                    if (f < W1_FR) {
                        const int s = f >> 1;
                        const half8 b = s < KS1 ? xf[s < KS1 ? s : 0] : ones;
                        if (f & 1) Ng = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Ng, 0, 0, 0);
                        else Nh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, Nh, 0, 0, 0);
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
                if (i == 3) refill((g & 1) * 2);
                if (i == 7) refill((g & 1) * 2 + 1);
            }
        };
        // refill pieces of a group whose consumption step does not exist (pipeline fill / drain)
        auto refill_only = [&](auto g_) { constexpr int g = decltype(g_)::value; refill((g & 1) * 2); refill((g & 1) * 2 + 1); };

        // ---- prologue section: O = b2, S(0) = W1(0) . x + b1   (4 slots)
        read_group(ic<0>{});
        read_group(ic<1>{}); consume_group(ic<0>{}, ic<0>{});
        read_group(ic<2>{}); consume_group(ic<0>{}, ic<1>{});
        read_group(ic<3>{}); consume_group(ic<0>{}, ic<2>{});
        read_group(ic<4>{}); consume_group(ic<0>{}, ic<3>{});
        read_group(ic<5>{}); consume_group(ic<0>{}, ic<4>{});
        read_group(ic<6>{}); consume_group(ic<0>{}, ic<5>{});
        read_group(ic<7>{}); consume_group(ic<0>{}, ic<6>{});
        // (group 7 of the prologue is padding: zeros; the first stage "consumes" it against pf = 0)

        // ---- steady state: stage k = S(k+1) with GEGLU(k) woven in, then O += W2(k) . P(k); the tail of W2(k) is consumed at the
        // start of stage k+1, before GEGLU(k+1) replaces P
#pragma unroll 1
        for (int k = 0; k < NCHUNK - 1; ++k) {
            read_group(ic<0>{}); consume_group(ic<1>{}, ic<7>{});
            zero16(Nh); zero16(Ng);
            geglu(Sh, Sg);
            read_group(ic<1>{}); consume_group(ic<1>{}, ic<0>{});
            read_group(ic<2>{}); consume_group(ic<1>{}, ic<1>{});
            read_group(ic<3>{}); consume_group(ic<1>{}, ic<2>{});
            read_group(ic<4>{}); consume_group(ic<1>{}, ic<3>{});
            read_group(ic<5>{}); consume_group(ic<1>{}, ic<4>{});
            read_group(ic<6>{}); consume_group(ic<1>{}, ic<5>{});
            read_group(ic<7>{}); consume_group(ic<1>{}, ic<6>{});
            Sh = Nh; Sg = Ng;
        }
        // ---- final section: tail of W2(38), GEGLU(39), W2(39)   (2 slots)
        read_group(ic<0>{}); consume_group(ic<1>{}, ic<7>{});
        geglu(Sh, Sg);
        read_group(ic<1>{}); consume_group(ic<2>{}, ic<0>{});
        read_group(ic<2>{}); consume_group(ic<2>{}, ic<1>{});
        consume_group(ic<2>{}, ic<2>{});
        refill_only(ic<3>{});

        // ---- epilogue: out = O + x (raw, re-read: L2-hot), 4 consecutive channels per lane and register quad
        {
            const srd_t rO = make_srd(p.out);
            const unsigned ooff = mok ? (unsigned)(((int64_t)m * p.ldo + 4 * half) * 2) : OOB_OFFSET;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint2v res[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) res[q] = __builtin_amdgcn_raw_buffer_load_b64(rX, xoff, (ct * 32 + q * 8) * 2, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // copy the elements to scalars first: __builtin_bit_cast applied directly to a vector subscript takes element 0
                    // for both with this hipcc (ROCm 7.2; the load is then narrowed to one dword - 4 of 8 residual channels wrong)
                    const unsigned rlo = res[q][0], rhi = res[q][1];
                    const half2v r0 = __builtin_bit_cast(half2v, rlo), r1 = __builtin_bit_cast(half2v, rhi);
                    const uint2v o = {pk2(O[ct][4 * q] + (float)r0[0], O[ct][4 * q + 1] + (float)r0[1]),
                                      pk2(O[ct][4 * q + 2] + (float)r1[0], O[ct][4 * q + 3] + (float)r1[1])};
                    __builtin_amdgcn_raw_buffer_store_b64(o, rO, ooff, (ct * 32 + q * 8) * 2, 0);
                }
            }
        }
        // stores may retire out of order with the ring's loads: drain before the counted waits are trusted again
        wait_vmcnt<0>();
    }
    wait_vmcnt<0>();   // no LDS-DMA may land after this workgroup's LDS has been handed to another one
}

}  // namespace

extern "C" int insv2v_ffn_fused(const insv2v_ffn_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    const insv2v_ffn_desc& d = *dp;
    if (!d.x || !d.out || !d.wstream || d.M <= 0) return INSV2V_EINVAL;
    if (d.C != FC || d.hidden != 4 * FC) return INSV2V_EUNSUPPORTED;
    if ((d.ldx & 3) || (d.ldo & 3) || ((uintptr_t)d.x & 7) || ((uintptr_t)d.out & 7) || ((uintptr_t)d.wstream & 15)) return INSV2V_EINVAL;
    if ((int64_t)d.M * d.ldx * 2 >= ((int64_t)1 << 31) || (int64_t)d.M * d.ldo * 2 >= ((int64_t)1 << 31)) return INSV2V_EUNSUPPORTED;
    static bool attr_set = false;
    static int num_cu = 0;
    constexpr int LDS_B = NS * SLOT_B;
    static const int dbg = getenv("INSV2V_FFN_DBG") ? atoi(getenv("INSV2V_FFN_DBG")) : 0;
    const void* kernels[8] = {(const void*)ffn_fused_kernel<0>, (const void*)ffn_fused_kernel<1>, (const void*)ffn_fused_kernel<2>, (const void*)ffn_fused_kernel<3>,
                              (const void*)ffn_fused_kernel<4>, (const void*)ffn_fused_kernel<5>, (const void*)ffn_fused_kernel<6>, (const void*)ffn_fused_kernel<7>};
    if (!attr_set) {
        for (int i = 0; i < 8; ++i) {
            hipError_t e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
            if (e != hipSuccess) return (int)e;
        }
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return INSV2V_EINVAL;
        num_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    FfnArgs a = {(const half_t*)d.x, (half_t*)d.out, (const half_t*)d.wstream, d.ldx, d.ldo, d.M, d.eps};
    const int ntiles = (d.M + 127) / 128;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    void* args[] = {&a};
    hipError_t le = hipLaunchKernel(kernels[dbg & 7], dim3(grid), dim3(256), args, LDS_B, as_stream(stream));
    if (le != hipSuccess) return (int)le;
    return launch_status();
}

// Size in fp16 elements of the weight stream insv2v_ffn_fused expects for (C, hidden); 0 if unsupported.
extern "C" int64_t insv2v_ffn_stream_elems(int32_t C, int32_t hidden) {
    if (C != FC || hidden != 4 * FC) return 0;
    return (int64_t)PASS_SLOTS * SLOT_FR * 512;
}
