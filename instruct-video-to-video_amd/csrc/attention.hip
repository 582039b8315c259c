// insv2v_attention: flash-style fused attention for spatial self-, text cross- and temporal
// self-attention (one addressing rule, see include/insv2v_hip.h).
//
// Roofline: MFMA-bound for long sequences (spatial, 1536 tokens), HBM/launch-bound for the
// 16-frame temporal case.  Layout of one wave (16 query rows), all contractions on
// v_mfma_f32_16x16x32_f16:
//   S^T = K . Q^T   (A = K fragment from LDS, B = Q fragment held in registers), so a lane owns
//                   ONE query column (lane&15) and 4 consecutive keys per 16-key block;
//   softmax         row max / row sum over keys = in-lane reduction + 2 wave shuffles (xor 16,32);
//                   the running max/sum and the output rescale factor are lane-local;
//   O^T = V^T . P^T (A = V^T fragment from a transposed LDS tile, B = the lane's own P values,
//                   no data movement: the MFMA k-slot <-> key mapping is chosen to match what
//                   the lane already holds).
// K/V tiles of 64 keys (32 for the short temporal sequences) are staged through double-buffered LDS by all waves of the workgroup;
// global loads for tile t+1 are issued before the MFMAs of tile t.
#include "common.h"


template <int DP, int NW, int KVT>
__global__ __launch_bounds__(NW * 64) void attn_kernel(insv2v_attention_desc p) {
    constexpr int KV_TILE = KVT;         // keys per staged tile (32 or 64)
    constexpr int VT_LD = KVT + 8;       // halfs per row of the transposed V tile
    constexpr int NKB = KVT / 32;        // 32-key MFMA blocks per tile
    constexpr int KLD = DP + 8;          // halfs per K row in LDS
    constexpr int DT = DP / 16;          // max output column tiles
    constexpr int KS = DP / 32;          // k-steps of the QK^T contraction
    constexpr int NT = NW * 64;
    constexpr int KCH = DP / 8;          // 16B chunks per K row (zero padded to DP)
    constexpr int K_ITERS = (KV_TILE * KCH + NT - 1) / NT;
    constexpr int V_ITERS = (KV_TILE * KCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = (half_t*)smem;                   // [2][64][KLD]
    half_t* sVt = sK + 2 * KV_TILE * KLD;         // [2][DP][VT_LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, qc = lane & 15;
    const int head = blockIdx.y, z = blockIdx.z;
    const int d = p.head_dim;
    const half_t* Q = (const half_t*)p.q + (int64_t)(z / p.q_inner) * p.q_outer + (int64_t)(z % p.q_inner) * p.q_step + head * d;
    const half_t* K = (const half_t*)p.k + (int64_t)(z / p.kv_inner) * p.kv_outer + (int64_t)(z % p.kv_inner) * p.kv_step + head * d;
    const half_t* V = (const half_t*)p.v + (int64_t)(z / p.kv_inner) * p.kv_outer + (int64_t)(z % p.kv_inner) * p.kv_step + head * d;
    half_t* O = (half_t*)p.o + (int64_t)(z / p.o_inner) * p.o_outer + (int64_t)(z % p.o_inner) * p.o_step + head * d;

    const int q = blockIdx.x * (16 * NW) + wid * 16 + qc;
    const bool qvalid = q < p.seq_q;

    // Q fragment (B operand): lane (g,qc) holds Q[q][kk*32 + g*8 .. +8]
    half8 qf[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        int c = kk * 32 + g * 8;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qvalid && c < d) v = *(const half8*)(Q + (int64_t)q * p.q_rs + c);
        qf[kk] = v;
    }

    uint4 rk[K_ITERS], rv[V_ITERS];
    auto load_tile = [&](int t) {
        const int key0 = t * KV_TILE;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            int e = tid + NT * i;
            int key = e / KCH, ch = e - key * KCH;  // chunk fastest: coalesced rows
            uint4 v = make_uint4(0, 0, 0, 0);
            if (e < KV_TILE * KCH && key0 + key < p.seq_k && ch * 8 < d)
                v = *(const uint4*)(K + (int64_t)(key0 + key) * p.k_rs + ch * 8);
            rk[i] = v;
        }
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            int e = tid + NT * i;
            int ch = e / KV_TILE, key = e - ch * KV_TILE;  // key fastest: conflict-free transposed LDS write
            uint4 v = make_uint4(0, 0, 0, 0);
            if (e < KV_TILE * KCH && key0 + key < p.seq_k && ch * 8 < d)
                v = *(const uint4*)(V + (int64_t)(key0 + key) * p.v_rs + ch * 8);
            rv[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        half_t* k = sK + buf * KV_TILE * KLD;
        half_t* vt = sVt + buf * DP * VT_LD;
#pragma unroll
        for (int i = 0; i < K_ITERS; ++i) {
            int e = tid + NT * i;
            int key = e / KCH, ch = e - key * KCH;
            if (e < KV_TILE * KCH) *(uint4*)(k + key * KLD + ch * 8) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < V_ITERS; ++i) {
            int e = tid + NT * i;
            int ch = e / KV_TILE, key = e - ch * KV_TILE;
            if (e < KV_TILE * KCH) {
                const half_t* h = (const half_t*)&rv[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) vt[(ch * 8 + j) * VT_LD + key] = h[j];
            }
        }
    };

    floatx4 acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;
    const float c2 = p.scale * 1.4426950408889634f;

    const int ntiles = (p.seq_k + KV_TILE - 1) / KV_TILE;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) load_tile(t + 1);
        const half_t* k = sK + cur * KV_TILE * KLD;
        const half_t* vt = sVt + cur * DP * VT_LD;
        const int key0 = t * KV_TILE;

        // ---- scores: s[kb][sub] = 16 keys x 16 queries, lane: keys key0+kb*32+sub*16+g*4+r
        floatx4 s[NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                floatx4 a = {0.f, 0.f, 0.f, 0.f};
                if (key0 + kb * 32 < p.seq_k) {
                    const half_t* kr = k + (kb * 32 + sub * 16 + qc) * KLD + g * 8;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        half8 kf = *(const half8*)(kr + kk * 32);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[kk], a, 0, 0, 0);
                    }
                }
                s[kb][sub] = a;
            }
        float mx = m_run;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int key = key0 + kb * 32 + sub * 16 + g * 4 + r;
                    float v = key < p.seq_k ? s[kb][sub][r] : -1.0e30f;
                    s[kb][sub][r] = v;
                    mx = fmaxf(mx, v);
                }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float alpha = exp2f((m_run - mx) * c2);
        m_run = mx;
        float psum = 0.f;
        half8 pf[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = exp2f((s[kb][sub][r] - mx) * c2);
                    psum += pv;
                    pf[kb][sub * 4 + r] = (half_t)pv;
                }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] *= alpha;

        // ---- O^T += V^T . P^T ; A fragment: Vt[dt*16+qc][kb*32 + g*4 + {0..3}] | [.. + 16 + g*4 + {0..3}]
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (key0 + kb * 32 < p.seq_k) {
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    if (i * 16 < d) {
                        const half_t* vr = vt + (i * 16 + qc) * VT_LD + kb * 32 + g * 4;
                        half4 lo = *(const half4*)vr;
                        half4 hi = *(const half4*)(vr + 16);
                        half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[kb], acc[i], 0, 0, 0);
                    }
                }
            }
        }
        if (t + 1 < ntiles) store_tile(cur ^ 1);
        __syncthreads();
    }

    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (qvalid) {
        half_t* orow = O + (int64_t)q * p.o_rs;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            int c = i * 16 + g * 4;
            if (c < d) {
                half4 h = {(half_t)(acc[i][0] * inv), (half_t)(acc[i][1] * inv), (half_t)(acc[i][2] * inv),
                           (half_t)(acc[i][3] * inv)};
                *(half4*)(orow + c) = h;
            }
        }
    }
}

template <int DP, int NW>
static int launch_attn(const insv2v_attention_desc& d, hipStream_t s) {
    constexpr int KVT = NW >= 4 ? 64 : 32;
    constexpr size_t lds = (size_t)2 * (KVT * (DP + 8) + DP * (KVT + 8)) * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_kernel<DP, NW, KVT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((d.seq_q + 16 * NW - 1) / (16 * NW), d.heads, d.batch);
    hipLaunchKernelGGL((attn_kernel<DP, NW, KVT>), grid, dim3(NW * 64), lds, s, d);
    return launch_status();
}

template <int NW>
static int dispatch_dp(const insv2v_attention_desc& d, hipStream_t s) {
    const int dp = (d.head_dim + 31) / 32 * 32;
    switch (dp) {
        case 32: return launch_attn<32, NW>(d, s);
        case 64: return launch_attn<64, NW>(d, s);
        case 96: return launch_attn<96, NW>(d, s);
        case 128: return launch_attn<128, NW>(d, s);
        case 160: return launch_attn<160, NW>(d, s);
    }
    return INSV2V_EUNSUPPORTED;
}

extern "C" int insv2v_attention(const insv2v_attention_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    insv2v_attention_desc d = *dp;
    if (!d.q || !d.k || !d.v || !d.o) return INSV2V_EINVAL;
    if (d.head_dim <= 0 || (d.head_dim & 7) || d.head_dim > 160) return INSV2V_EINVAL;
    if (d.seq_q <= 0 || d.seq_k <= 0 || d.batch <= 0 || d.heads <= 0) return INSV2V_EINVAL;
    if ((d.q_rs & 7) || (d.k_rs & 7) || (d.v_rs & 7) || (d.o_rs & 3)) return INSV2V_EINVAL;
    if (d.q_inner <= 0) d.q_inner = 1;
    if (d.kv_inner <= 0) d.kv_inner = 1;
    if (d.o_inner <= 0) d.o_inner = 1;
    if (d.batch > 65535 || d.heads > 65535) return INSV2V_EUNSUPPORTED;
    hipStream_t s = as_stream(stream);
    // 16 query rows per wave: short query sequences (temporal, seq = frames) use 1-wave workgroups.
    if (d.seq_q <= 16) return dispatch_dp<1>(d, s);
    if (d.seq_q <= 32) return dispatch_dp<2>(d, s);
    return dispatch_dp<4>(d, s);
}
