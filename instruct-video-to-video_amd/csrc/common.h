// Shared device helpers for the gfx950 kernels (wave64, MFMA fragment types).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/insv2v_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }  // x * sigmoid(1.702 x)
// the optical-flow network's activations (INSV2V_ACT_RELU / SIGMOID / TANH; only the round-1 tile kernel's epilogue takes them)
__device__ __forceinline__ float act_raft_f(float x, int act) {
    if (act == INSV2V_ACT_RELU) return fmaxf(x, 0.f);
    if (act == INSV2V_ACT_SIGMOID) return 1.0f / (1.0f + __expf(-x));
    if (act != INSV2V_ACT_TANH) return x;       // (insv2v_gemm rejects unknown codes; nothing silently becomes tanh)
    const float e = __expf(-2.0f * fabsf(x));   // tanh(|x|) = (1 - e) / (1 + e): no overflow for large |x|
    return copysignf((1.0f - e) / (1.0f + e), x);
}
// erf-GELU, x * Phi(x), with Phi(-|x|) = 2^Q(|x|), Q a degree-5 polynomial fitted (minimax on the GELU value itself,
// tools/fit_gelu.py) to log2 of the normal CDF: |abs err| <= 8e-7 over all finite inputs, relative error <= 3e-5 around 0
// (fp16 resolution is 4.9e-4), Q -> -inf for large |x| so no clamp is needed.  gelu(x) = 0.5 x + |x| (0.5 - T):
// 5 FMA + one v_exp_f32 + 3 VALU ops (libm erff: ~60; Abramowitz-Stegun 7.1.26 with its v_rcp_f32: ~22).  The GEGLU
// epilogue applies it to every FF1 output and is VALU-bound at K = 320.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float a = fabsf(x);
    float q = fmaf(-0.0004733019319801221f, a, 0.007084501019364234f);
    q = fmaf(q, a, -0.05182722931942957f);
    q = fmaf(q, a, -0.4599926224444887f);
    q = fmaf(q, a, -1.1507877598128362f);
    q = fmaf(q, a, -1.0000376369909822f);
    const float t = __builtin_amdgcn_exp2f(q);
    return fmaf(a, 0.5f - t, 0.5f * x);
}
// the same value as max(x, 0) - |x| T (0.5 x + 0.5 |x| = relu(x)): one VALU less, |x| as a source modifier
__device__ __forceinline__ float gelu_erf_relu_f(float x) {
    const float a = __builtin_fabsf(x);
    float q = fmaf(-0.0004733019319801221f, a, 0.007084501019364234f);
    q = fmaf(q, a, -0.05182722931942957f);
    q = fmaf(q, a, -0.4599926224444887f);
    q = fmaf(q, a, -1.1507877598128362f);
    q = fmaf(q, a, -1.0000376369909822f);
    const float t = __builtin_amdgcn_exp2f(q);
    return fmaf(-a, t, fmaxf(x, 0.f));
}

// GEGLU of TWO outputs in packed fp16 arithmetic: x * gelu_erf(g) for the pair, returned as one packed dword (round 5, the GEGLU
// epilogue of the 256 x 320 GEMM: the fp32 form above costs ~15 VALU per output with the matrix pipes idle - a fifth of a K = 640
// launch).  The same polynomial evaluated with v_pk_fma_f16 on |g| (two outputs per instruction), v_exp_f16 per half, relu and the
// final products packed: ~10 per output.  Against the exact erf-GELU of the fp16-rounded g: rms absolute error 2.09e-4 vs 2.05e-4 for
// the fp32 form rounded to fp16 (tools/fit_gelu.py --pk), i.e. the output's own fp16 rounding dominates.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned geglu_pk_f16(float x0, float x1, float g0, float g1) {
    const h2_t x = {(_Float16)x0, (_Float16)x1}, g = {(_Float16)g0, (_Float16)g1};
    const h2_t a = __builtin_elementwise_abs(g);
    const h2_t c5 = (_Float16)-0.0004733019319801221f, c4 = (_Float16)0.007084501019364234f, c3 = (_Float16)-0.05182722931942957f,
               c2 = (_Float16)-0.4599926224444887f, c1 = (_Float16)-1.1507877598128362f, c0 = (_Float16)-1.0000376369909822f, zero = (_Float16)0.f;
    h2_t q = __builtin_elementwise_fma(c5, a, c4);
    q = __builtin_elementwise_fma(q, a, c3);
    q = __builtin_elementwise_fma(q, a, c2);
    q = __builtin_elementwise_fma(q, a, c1);
    q = __builtin_elementwise_fma(q, a, c0);
    const h2_t t = __builtin_elementwise_exp2(q);
    const h2_t r = __builtin_elementwise_fma(-a, t, __builtin_elementwise_max(g, zero));
    return __builtin_bit_cast(unsigned, x * r);
}

// XCD-aware, bijective remap of a linear workgroup id (guide T1): blocks that are
// consecutive after the remap run on the same XCD and share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx * 2) return bid;
    int q = nwg / nx, r = nwg % nx, xcd = bid % nx, idx = bid / nx;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline hipStream_t as_stream(insv2v_stream_t s) { return (hipStream_t)s; }
// One process drives ONE device (one process per GPU): the launchers cache the CU count and the per-function
// dynamic-LDS attribute of the first device they run on.  A call with another device current is refused instead of mis-sizing a
// persistent grid.
// (round 6, ADVICE r5: ONE latch for the whole library - an atomic in elementwise.hip, set by insv2v_init or the first launcher
// call with compare-exchange - instead of one unsynchronised static per translation unit.)
bool insv2v_one_device_check();
static inline bool one_device() { return insv2v_one_device_check(); }
static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
