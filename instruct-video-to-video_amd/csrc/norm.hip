// GroupNorm(+SiLU), LayerNorm(+temporal PE) and row softmax for channels-last fp16 tokens.
//
// Roofline: HBM-bound (one read + one write of the activation; GroupNorm reads twice).
// Every global access is a 16-byte (8 x fp16) vector per lane, lanes of a wave cover
// consecutive channel chunks of a token, so a wave reads whole contiguous token rows.
// Statistics are fp32.  GroupNorm variance uses per-channel SHIFTED sums (shift = the
// sample's first token) merged with Chan's parallel formula, which is deterministic (no
// atomics) and free of the E[x^2]-E[x]^2 cancellation.
#include "common.h"
#include <cstdlib>

struct Moments {  // count, mean, sum of squared deviations
    float n, mean, m2;
};
__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    if (b.n == 0.f) return a;
    if (a.n == 0.f) return b;
    Moments r;
    r.n = a.n + b.n;
    float d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
    return r;
}

__device__ __forceinline__ half8 load8(const half_t* x, const half_t* x2, int64_t ldx, int64_t ldx2, int C1,
                                       int64_t row, int c0) {
    const half_t* p = (c0 < C1) ? x + row * ldx + c0 : x2 + row * ldx2 + (c0 - C1);
    return *(const half8*)p;
}

// pass 1: per (sample, chunk): partial (mean, M2) of every group.  block = CC*P threads,
// thread (pl, cc) owns channel chunk cc and tokens pl, pl+P, ... of the chunk.
__global__ void gn_partial_kernel(insv2v_groupnorm_desc p, int CC, int P, int rows_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Moments* sm = (Moments*)smem;  // [P][C] then reused as [C]
    const int tid = threadIdx.x, cc = tid % CC, pl = tid / CC;
    const int sample = blockIdx.y, chunk = blockIdx.x;
    const int C1 = p.x2 ? p.C1 : p.C;
    const half_t* x = (const half_t*)p.x;
    const half_t* x2 = (const half_t*)p.x2;
    const int64_t row0 = (int64_t)sample * p.rows_per_sample;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(r0 + rows_per_chunk, p.rows_per_sample);
    const int c0 = cc * 8;

    half8 kv = load8(x, x2, p.ldx, p.ldx2, C1, row0, c0);  // per-channel shift
    float k[8], s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { k[e] = (float)kv[e]; s[e] = 0.f; q[e] = 0.f; }
    int cnt = 0;
    int r = r0 + pl;
    // four rows per step, all loads issued before the first use: with one load in flight per thread this pass ran at HBM
    // LATENCY (20 us for 15.7 MB, profiles/r02_final_rocprofv3_kernel_stats_bench.csv)
    for (; r + 3 * P < r1; r += 4 * P) {
        half8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load8(x, x2, p.ldx, p.ldx2, C1, row0 + r + u * P, c0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float d = (float)v[u][e] - k[e];
                s[e] += d;
                q[e] += d * d;
            }
        cnt += 4;
    }
    for (; r < r1; r += P) {
        half8 v = load8(x, x2, p.ldx, p.ldx2, C1, row0 + r, c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float d = (float)v[e] - k[e];
            s[e] += d;
            q[e] += d * d;
        }
        ++cnt;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        Moments m;
        m.n = (float)cnt;
        float inv = cnt ? 1.f / cnt : 0.f;
        m.mean = k[e] + s[e] * inv;
        m.m2 = fmaxf(q[e] - s[e] * s[e] * inv, 0.f);
        sm[pl * p.C + c0 + e] = m;
    }
    __syncthreads();
    // merge token lanes per channel
    for (int c = tid; c < p.C; c += blockDim.x) {
        Moments m = sm[c];
        for (int j = 1; j < P; ++j) m = merge(m, sm[j * p.C + c]);
        sm[c] = m;  // lane j=0 slot; only this thread touches column c
    }
    __syncthreads();
    const int cpg = p.C / p.G;
    for (int g = tid; g < p.G; g += blockDim.x) {
        Moments m = sm[g * cpg];
        for (int j = 1; j < cpg; ++j) m = merge(m, sm[g * cpg + j]);
        float* out = p.partials + (int64_t)p.nsamples * p.G * 2 +
                     (((int64_t)sample * p.nchunks + chunk) * p.G + g) * 3;
        out[0] = m.n;
        out[1] = m.mean;
        out[2] = m.m2;
    }
}

// pass 2: merge the chunk partials of one (sample, group) with one wave -> (mean, rstd).
__global__ __launch_bounds__(256) void gn_finalize_kernel(insv2v_groupnorm_desc p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sample = blockIdx.y, g = blockIdx.x * 4 + wid;
    if (g >= p.G) return;
    const float* in = p.partials + (int64_t)p.nsamples * p.G * 2 + ((int64_t)sample * p.nchunks * p.G + g) * 3;
    Moments m = {0.f, 0.f, 0.f};
    for (int c = lane; c < p.nchunks; c += 64) {
        const float* q = in + (int64_t)c * p.G * 3;
        Moments b = {q[0], q[1], q[2]};
        m = merge(m, b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Moments b = {__shfl_xor(m.n, o, 64), __shfl_xor(m.mean, o, 64), __shfl_xor(m.m2, o, 64)};
        m = merge(m, b);
    }
    if (lane == 0) {
        float var = m.m2 / m.n;
        p.partials[((int64_t)sample * p.G + g) * 2 + 0] = m.mean;
        p.partials[((int64_t)sample * p.G + g) * 2 + 1] = rsqrtf(var + p.eps);
    }
    if (p.ab) {  // per-channel (scale, shift) for a consumer that normalises on the fly (insv2v_gemm gn_ab)
        const float mean = m.mean, rstd = rsqrtf(m.m2 / m.n + p.eps);
        const int cpg = p.C / p.G;
        for (int c = lane; c < cpg; c += 64) {
            const int ch = g * cpg + c;
            const float a = rstd * p.gamma[ch];
            p.ab[((int64_t)sample * p.C + ch) * 2 + 0] = a;
            p.ab[((int64_t)sample * p.C + ch) * 2 + 1] = p.beta[ch] - mean * a;
        }
    }
}

// pass 3: y = act((x - mean) * rstd * gamma + beta)
__global__ void gn_apply_kernel(insv2v_groupnorm_desc p, int CC, int P) {
    const int tid = threadIdx.x, cc = tid % CC, pl = tid / CC;
    const int sample = blockIdx.y;
    const int C1 = p.x2 ? p.C1 : p.C;
    const half_t* x = (const half_t*)p.x;
    const half_t* x2 = (const half_t*)p.x2;
    half_t* y = (half_t*)p.y;
    const int64_t row0 = (int64_t)sample * p.rows_per_sample;
    const int c0 = cc * 8, cpg = p.C / p.G;
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int c = c0 + e, g = c / cpg;
        float mean = p.partials[((int64_t)sample * p.G + g) * 2 + 0];
        float rstd = p.partials[((int64_t)sample * p.G + g) * 2 + 1];
        a[e] = rstd * p.gamma[c];
        b[e] = p.beta[c] - mean * a[e];
    }
    const int step = gridDim.x * P;
    int r = blockIdx.x * P + pl;
    auto finish = [&](half8 v, int row) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (float)v[e] * a[e] + b[e];
            if (p.silu) t = silu_f(t);
            o[e] = (half_t)t;
        }
        *(half8*)(y + (row0 + row) * p.ldy + c0) = o;
    };
    for (; r + 3 * step < p.rows_per_sample; r += 4 * step) {  // four rows per thread in flight (the grid leaves ~4 rows per thread)
        half8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load8(x, x2, p.ldx, p.ldx2, C1, row0 + r + u * step, c0);
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(v[u], r + u * step);
    }
    for (; r < p.rows_per_sample; r += step) finish(load8(x, x2, p.ldx, p.ldx2, C1, row0 + r, c0), r);
}

// Single-launch GroupNorm for small (sample, group) slabs: one workgroup owns one (sample, group), keeps its
// rows x cpg elements in registers (<= 256 threads x GNS_MAXCH chunks), two-pass statistics in fp32, one read
// and one write of the tensor instead of 2 reads + 1 write over three launches.  VW = halfs per chunk.
#define GNS_MAXCH 16
template <int VW>
__global__ __launch_bounds__(256) void gn_small_kernel(insv2v_groupnorm_desc p) {
    typedef half_t vec_t __attribute__((ext_vector_type(VW)));
    __shared__ float red[8];
    __shared__ float sgam[256], sbet[256];   // this group's affine parameters (cpg <= 256, checked on the host)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.x, sample = blockIdx.y;
    const int cpg = p.C / p.G, cpr = cpg / VW;             // chunks per row of this group
    const int nchunk = p.rows_per_sample * cpr;
    const int C1 = p.x2 ? p.C1 : p.C;
    const half_t* x = (const half_t*)p.x;
    const half_t* x2 = (const half_t*)p.x2;
    const int64_t row0 = (int64_t)sample * p.rows_per_sample;
    if (tid < cpg) { sgam[tid] = p.gamma[g * cpg + tid]; sbet[tid] = p.beta[g * cpg + tid]; }   // visible after the first barrier below
    vec_t v[GNS_MAXCH];
    float s = 0.f;
    // all of the thread's loads first, branch-free (out-of-range chunks re-read chunk `tid` and are zeroed): inside per-chunk
    // `if` regions hipcc waits for each load before the next one is issued and the pass runs at memory latency
    const int rs = tid < nchunk ? tid / cpr : 0, cs = tid < nchunk ? tid - rs * cpr : 0;   // chunk `tid`: (row, chunk in row)
    const int dr = 256 / cpr, dc = 256 - dr * cpr;                                           // +256 chunks = +dr rows +dc chunks
    int rr = rs, cc = cs;
#pragma unroll
    for (int i = 0; i < GNS_MAXCH; ++i) {
        const bool ok = tid + 256 * i < nchunk;
        const int r = ok ? rr : rs, c0 = g * cpg + (ok ? cc : cs) * VW;
        const half_t* src = c0 < C1 ? x + (row0 + r) * p.ldx + c0 : x2 + (row0 + r) * p.ldx2 + (c0 - C1);
        v[i] = *(const vec_t*)src;
        rr += dr; cc += dc;
        if (cc >= cpr) { cc -= cpr; ++rr; }
    }
#pragma unroll
    for (int i = 0; i < GNS_MAXCH; ++i) {
        if (tid + 256 * i >= nchunk) {
#pragma unroll
            for (int e = 0; e < VW; ++e) v[i][e] = (half_t)0.f;
        }
#pragma unroll
        for (int e = 0; e < VW; ++e) s += (float)v[i][e];
    }
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    const float n = (float)p.rows_per_sample * cpg;
    const float mean = (red[0] + red[1] + red[2] + red[3]) / n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < GNS_MAXCH; ++i) {
        if (tid + 256 * i < nchunk) {
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float d = (float)v[i][e] - mean;
                q += d * d;
            }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wid] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / n + p.eps);
    half_t* y = (half_t*)p.y;
    rr = rs; cc = cs;
#pragma unroll
    for (int i = 0; i < GNS_MAXCH; ++i) {
        const int idx = tid + 256 * i;
        const int r = rr, c0 = g * cpg + cc * VW;
        rr += dr; cc += dc;
        if (cc >= cpr) { cc -= cpr; ++rr; }
        if (idx < nchunk) {
            vec_t o;
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                float t = ((float)v[i][e] - mean) * rstd * sgam[c0 - g * cpg + e] + sbet[c0 - g * cpg + e];
                if (p.silu) t = silu_f(t);
                o[e] = (half_t)t;
            }
            *(vec_t*)(y + (row0 + r) * p.ldy + c0) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// gn_frame_kernel: the per-frame GroupNorm in front of the spatial transformers of the 8x12 / 4x6 levels (attention.py:236-244 on
// `(b f) c h w`: sample = ONE frame = 96 / 24 token rows x 1280 channels = 245 / 61 KB).  gn_small gives every (sample, group) its
// own workgroup, whose rows are 80-byte pieces at a 2560-byte stride and whose 30 720 workgroups hold 7.7 KB each: 0.5-1.6 TB/s.
// Here ONE workgroup owns the whole sample in registers (NT threads x <= 15 chunks of 16 bytes), reads and writes it as one contiguous
// stream (thread t, chunk i = 16-byte chunk t + NT i of the row-major sample), and reduces per group through LDS: every chunk lies in
// exactly one group (cpg % 8 == 0); its sum goes to part[chunk]; the NT / G threads of a group add that group's R x cpg / 8 entries
// and finish with a shuffle.  Two passes over the registers (mean, then centred squares): cancellation-free like the other kernels.
#define GNF_MAXCH 15
template <int NT>
__global__ __launch_bounds__(NT) void gn_frame_kernel(insv2v_groupnorm_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int sample = blockIdx.x;
    // channel split (grid y): groups are independent, so a sample's channels may be shared out over CS workgroups (whole groups each): smaller
    // workgroups, several resident per CU - one's loads run under another's reductions and stores (round 6; one 1024-thread workgroup per CU
    // and sample ran load / reduce / store phases back to back: 2.1 TB/s)
    const int CS = gridDim.y, Cw = p.C / CS, Gw = p.G / CS, coff = blockIdx.y * Cw;
    const int crow = Cw >> 3;                          // 16-byte chunks per row (of this workgroup's channels)
    const int cpr = (p.C / p.G) >> 3;                  // chunks per row of one group
    const int nchunk = p.rows_per_sample * crow;
    float* part = (float*)smem;                        // [nchunk]
    float* gstat = part + nchunk;                      // [G] mean, then [G] rstd
    float* sgam = gstat + 2 * Gw;                      // [C] gamma, [C] beta
    float* sbet = sgam + Cw;
    for (int c = tid; c < Cw; c += NT) { sgam[c] = p.gamma[coff + c]; sbet[c] = p.beta[coff + c]; }
    const half_t* x = (const half_t*)p.x + (int64_t)sample * p.rows_per_sample * p.ldx + coff;
    half_t* y = (half_t*)p.y + (int64_t)sample * p.rows_per_sample * p.ldy + coff;
    // chunk k = tid + NT i  ->  (row, chunk in row); stepping by NT = dr rows + dc chunks
    const int r0 = tid / crow, c0 = tid - r0 * crow;
    const int dr = NT / crow, dc = NT - dr * crow;
    half8 v[GNF_MAXCH];
    {
        int rr = r0, cc = c0;
#pragma unroll
        for (int i = 0; i < GNF_MAXCH; ++i) {
            const bool ok = tid + NT * i < nchunk;
            v[i] = *(const half8*)(x + (int64_t)(ok ? rr : r0) * p.ldx + (ok ? cc : c0) * 8);   // branch-free: idle slots re-read chunk `tid`
            rr += dr; cc += dc;
            if (cc >= crow) { cc -= crow; ++rr; }
        }
    }
    const int TPG = NT / Gw;                           // threads per group in the reductions (host: G divides NT, TPG <= 64 a power of two)
    const int rg = tid / TPG, rp = tid - rg * TPG;     // this thread reduces group rg, entries rp, rp + TPG, ...
    const int ng = p.rows_per_sample * cpr;            // entries of a group
    const float inv_n = 1.f / ((float)p.rows_per_sample * (float)(p.C / p.G));   // (a group's channel count does not depend on the split)
    auto group_total = [&]() {                         // sum of part[] over group rg (valid in every lane of the group's TPG lanes)
        float s = 0.f;
        for (int e = rp; e < ng; e += TPG) {
            const int r = e / cpr, j = e - r * cpr;
            s += part[r * crow + rg * cpr + j];
        }
        for (int o = TPG >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        return s;
    };
    // pass 1: mean
#pragma unroll
    for (int i = 0; i < GNF_MAXCH; ++i) {
        if (tid + NT * i < nchunk) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[i][e];
            part[tid + NT * i] = s;
        }
    }
    __syncthreads();
    {
        const float m = group_total() * inv_n;
        if (rp == 0) gstat[rg] = m;
    }
    __syncthreads();
    // pass 2: centred squares
    {
        int cc = c0;
#pragma unroll
        for (int i = 0; i < GNF_MAXCH; ++i) {
            if (tid + NT * i < nchunk) {
                const float m = gstat[cc / cpr];
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - m; q = fmaf(d, d, q); }
                part[tid + NT * i] = q;
            }
            cc += dc;
            if (cc >= crow) cc -= crow;
        }
    }
    __syncthreads();
    {
        const float var = group_total() * inv_n;
        if (rp == 0) gstat[Gw + rg] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    // apply, store
    {
        int rr = r0, cc = c0;
#pragma unroll
        for (int i = 0; i < GNF_MAXCH; ++i) {
            if (tid + NT * i < nchunk) {
                const int g = cc / cpr;
                const float m = gstat[g], rs = gstat[Gw + g];
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = ((float)v[i][e] - m) * rs * sgam[cc * 8 + e] + sbet[cc * 8 + e];
                    if (p.silu) t = silu_f(t);
                    o[e] = (half_t)t;
                }
                *(half8*)(y + (int64_t)rr * p.ldy + cc * 8) = o;
            }
            rr += dr; cc += dc;
            if (cc >= crow) { cc -= crow; ++rr; }
        }
    }
}

template <int NT>
static int launch_gn_frame(const insv2v_groupnorm_desc& d, hipStream_t s, int CS = 1) {
    const size_t lds = ((size_t)d.rows_per_sample * (d.C / CS / 8) + 2 * (d.G / CS) + 2 * (d.C / CS)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gn_frame_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (lds > 96 * 1024) return INSV2V_EUNSUPPORTED;
    hipLaunchKernelGGL(gn_frame_kernel<NT>, dim3(d.nsamples, CS), dim3(NT), lds, s, d);
    return launch_status();
}

extern "C" int insv2v_groupnorm(const insv2v_groupnorm_desc* dp, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!dp) return INSV2V_EINVAL;
    insv2v_groupnorm_desc d = *dp;
    if (!d.x || !d.gamma || !d.beta || !d.partials) return INSV2V_EINVAL;
    if (d.stats_only ? !d.ab : !d.y) return INSV2V_EINVAL;
    if (d.C <= 0 || d.G <= 0 || (d.C % d.G) || (d.C & 7) || d.C > 8192) return INSV2V_EINVAL;
    if ((d.ldx & 7) || (d.ldy & 7) || d.nsamples <= 0 || d.rows_per_sample <= 0) return INSV2V_EINVAL;
    if (d.x2 && ((d.C1 & 7) || d.C1 <= 0 || d.C1 >= d.C || (d.ldx2 & 7))) return INSV2V_EINVAL;
    if (d.nchunks <= 0) return INSV2V_EINVAL;
    hipStream_t s = as_stream(stream);
    {   // whole samples of <= 245 KB (the per-frame GroupNorm of the 8x12 / 4x6 levels), many of them: one workgroup per sample
        static const int frame_on = getenv("INSV2V_GN_FRAME") ? atoi(getenv("INSV2V_GN_FRAME")) : 1;
        const int cpg = d.C / d.G;
        const int64_t nchunk = (int64_t)d.rows_per_sample * (d.C / 8);
        if (frame_on && !d.stats_only && !d.ab && !d.x2 && (cpg % 8) == 0 && d.nsamples >= 128 && nchunk >= 2048 && nchunk <= 1024 * GNF_MAXCH &&
            (d.G == 32 || d.G == 16) && d.C <= 2560) {
            // channel split CS (see the kernel): as many parts as keep whole groups, a power-of-two thread count per group and >= 2048 chunks
            // per workgroup.  INSV2V_GN_FRAME_SPLIT overrides (1 = the round-5 form: one workgroup per sample)
            static const int split_env = getenv("INSV2V_GN_FRAME_SPLIT") ? atoi(getenv("INSV2V_GN_FRAME_SPLIT")) : 0;
            int CS = split_env > 0 ? split_env : (nchunk > 512 * GNF_MAXCH ? 4 : nchunk > 256 * GNF_MAXCH ? 2 : 1);
            while (CS > 1 && (d.G % CS || nchunk / CS < 1024)) CS >>= 1;
            const int64_t nc = nchunk / CS;
            const int rc = nc <= 256 * GNF_MAXCH ? launch_gn_frame<256>(d, s, CS) : nc <= 512 * GNF_MAXCH ? launch_gn_frame<512>(d, s, CS) : launch_gn_frame<1024>(d, s, CS);
            if (rc != INSV2V_EUNSUPPORTED) return rc;
        }
    }
    {   // small slabs: one launch, one read + one write (cpg*2 B >= 32 B keeps the strided row pieces sector-sized)
        const int cpg = d.C / d.G;
        const int vw = (cpg % 8 == 0) ? 8 : ((cpg % 4 == 0) ? 4 : 0);
        const bool src_ok = !d.x2 || (d.C1 % (vw ? vw : 1) == 0);
        if (!d.stats_only && !d.ab && vw && cpg >= 16 && cpg <= 256 && src_ok && (int64_t)d.rows_per_sample * (cpg / vw) <= 256 * GNS_MAXCH) {
            if (vw == 8) hipLaunchKernelGGL(gn_small_kernel<8>, dim3(d.G, d.nsamples), dim3(256), 0, s, d);
            else hipLaunchKernelGGL(gn_small_kernel<4>, dim3(d.G, d.nsamples), dim3(256), 0, s, d);
            return launch_status();
        }
    }
    const int CC = d.C / 8;
    if (CC > 1024) return INSV2V_EUNSUPPORTED;
    int P = 256 / CC;
    if (P < 1) P = 1;
    if (P > 16) P = 16;
    const int threads = CC * P;
    int nchunks = d.nchunks;
    if (nchunks > d.rows_per_sample) nchunks = d.rows_per_sample;
    d.nchunks = nchunks;
    const int rows_per_chunk = (d.rows_per_sample + nchunks - 1) / nchunks;
    // a trailing chunk may be empty when rows do not divide: it contributes n = 0.
    size_t lds = (size_t)P * d.C * sizeof(Moments);
    if (lds > 64 * 1024) return INSV2V_EUNSUPPORTED;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, d.nsamples), dim3(threads), lds, s, d, CC, P, rows_per_chunk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((d.G + 3) / 4, d.nsamples), dim3(256), 0, s, d);
    if (d.stats_only) return launch_status();
    long want = ((long)d.rows_per_sample + P * 4 - 1) / (P * 4);
    long cap = 4096 / d.nsamples;
    if (cap < 1) cap = 1;
    int gx = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, d.nsamples), dim3(threads), 0, s, d, CC, P);
    return launch_status();
}

// ------------------------------------------------------------------------------ LayerNorm
// one wave per LN_ROWS token rows; lane owns 16-byte chunks lane, lane+64, ... (C <= 2048).  All
// loads of the wave's rows are issued before the first reduction so several KB per wave are in flight.
#define LN_MAXCH 4
#define LN_ROWS 4
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(insv2v_layernorm_desc p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wid) * LN_ROWS;
    if (row0 >= p.rows) return;
    const int CC = p.C >> 3;
    half8 v[LN_ROWS][NCH];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const int row = min(row0 + r, p.rows - 1);
        const half_t* x = (const half_t*)p.x + (int64_t)row * p.ldx;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            const half8 t = *(const half8*)(x + (ch < CC ? ch : 0) * 8);  // unconditional load (chunk 0 for idle lanes): no branch, no wait between loads
            v[r][i] = ch < CC ? t : z;
        }
    }
    float mean[LN_ROWS], rstd[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[r][i][e];
        mean[r] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        mean[r] /= p.C;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + 64 * i < CC) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = (float)v[r][i][e] - mean[r];
                    s += d * d;
                }
            }
        }
        rstd[r] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) rstd[r] = rsqrtf(rstd[r] / p.C + p.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch >= CC) continue;
        float g[8], b[8];
        const float4 g0 = *(const float4*)(p.gamma + ch * 8), g1 = *(const float4*)(p.gamma + ch * 8 + 4);
        const float4 b0 = *(const float4*)(p.beta + ch * 8), b1 = *(const float4*)(p.beta + ch * 8 + 4);
        g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) {
            const int row = row0 + r;
            if (row >= p.rows) break;
            const float* pe = p.pe ? p.pe + (int64_t)((row / p.rows_per_frame) % p.frames + p.pe_start) * p.C + ch * 8 : nullptr;
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = ((float)v[r][i][e] - mean[r]) * rstd[r] * g[e] + b[e];
                if (pe) t += pe[e];
                o[e] = (half_t)t;
            }
            *(half8*)((half_t*)p.y + (int64_t)row * p.ldy + ch * 8) = o;
        }
    }
}

// LayerNorm statistics only (mean, rstd per token): the read-only half of LayerNorm, for GEMMs that fold
// the normalisation into their epilogue.  Same row/lane mapping as layernorm_kernel.
template <int NCH>
__global__ __launch_bounds__(256) void ln_stats_kernel(const half_t* xp, float* stats, int64_t ldx, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wid) * LN_ROWS;
    if (row0 >= rows) return;
    const int CC = C >> 3;
    half8 v[LN_ROWS][NCH];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const half_t* x = xp + (int64_t)min(row0 + r, rows - 1) * ldx;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + 64 * i;
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            const half8 t = *(const half8*)(x + (ch < CC ? ch : 0) * 8);  // unconditional load: all LN_ROWS x NCH loads in flight together
            v[r][i] = ch < CC ? t : z;
        }
    }
    float mean[LN_ROWS], var[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[r][i][e];
        mean[r] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        mean[r] /= C;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + 64 * i < CC) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)v[r][i][e] - mean[r];
                    s += d * d;
                }
            }
        }
        var[r] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) var[r] += __shfl_xor(var[r], o, 64);
    if (lane < LN_ROWS && row0 + lane < rows) {
        float m = mean[0], q = var[0];
#pragma unroll
        for (int r = 1; r < LN_ROWS; ++r)
            if (lane == r) { m = mean[r]; q = var[r]; }
        *(float2*)(stats + 2 * (int64_t)(row0 + lane)) = make_float2(m, rsqrtf(q / C + eps));
    }
}

extern "C" int insv2v_layernorm_stats(const void* x, float* stats, int64_t ldx, int32_t rows, int32_t C, float eps,
                                      insv2v_stream_t stream) {
    if (!x || !stats || rows <= 0 || (C & 7) || C <= 0 || C > 64 * 8 * LN_MAXCH || (ldx & 7)) return INSV2V_EINVAL;
    const int nch = (C / 8 + 63) / 64;
    dim3 grid((rows + 4 * LN_ROWS - 1) / (4 * LN_ROWS));
    hipStream_t s = as_stream(stream);
    const half_t* xp = (const half_t*)x;
    switch (nch) {
        case 1: hipLaunchKernelGGL(ln_stats_kernel<1>, grid, dim3(256), 0, s, xp, stats, ldx, rows, C, eps); break;
        case 2: hipLaunchKernelGGL(ln_stats_kernel<2>, grid, dim3(256), 0, s, xp, stats, ldx, rows, C, eps); break;
        case 3: hipLaunchKernelGGL(ln_stats_kernel<3>, grid, dim3(256), 0, s, xp, stats, ldx, rows, C, eps); break;
        default: hipLaunchKernelGGL(ln_stats_kernel<4>, grid, dim3(256), 0, s, xp, stats, ldx, rows, C, eps); break;
    }
    return launch_status();
}

extern "C" int insv2v_layernorm(const insv2v_layernorm_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    insv2v_layernorm_desc d = *dp;
    if (!d.x || !d.y || !d.gamma || !d.beta || d.rows <= 0) return INSV2V_EINVAL;
    if ((d.C & 7) || d.C <= 0 || d.C > 64 * 8 * LN_MAXCH || (d.ldx & 7) || (d.ldy & 7)) return INSV2V_EINVAL;
    if (d.pe && (d.rows_per_frame <= 0 || d.frames <= 0 || d.pe_start < 0)) return INSV2V_EINVAL;
    const int nch = (d.C / 8 + 63) / 64;
    dim3 grid((d.rows + 4 * LN_ROWS - 1) / (4 * LN_ROWS));
    hipStream_t s = as_stream(stream);
    switch (nch) {
        case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, d); break;
        case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, d); break;
        case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, s, d); break;
        default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, d); break;
    }
    return launch_status();
}

// ------------------------------------------------------------------------------ row softmax
__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* x, half_t* y, int64_t ldx, int64_t ldy,
                                                           int cols, float scale) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const half_t* xr = x + (int64_t)row * ldx;
    half_t* yr = y + (int64_t)row * ldy;
    const int CC = cols >> 3;
    float mx = -3.0e38f;
    for (int ch = tid; ch < CC; ch += 256) {
        half8 v = *(const half8*)(xr + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float c = scale * 1.4426950408889634f;
    float sum = 0.f;
    for (int ch = tid; ch < CC; ch += 256) {
        half8 v = *(const half8*)(xr + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += exp2f(((float)v[e] - mx) * c);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int ch = tid; ch < CC; ch += 256) {
        half8 v = *(const half8*)(xr + ch * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(exp2f(((float)v[e] - mx) * c) * inv);
        *(half8*)(yr + ch * 8) = o;
    }
}

extern "C" int insv2v_softmax_rows(const void* x, void* y, int64_t ldx, int64_t ldy, int32_t rows, int32_t cols,
                                   float scale, insv2v_stream_t stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || (cols & 7) || (ldx & 7) || (ldy & 7)) return INSV2V_EINVAL;
    if (scale < 0.f) return INSV2V_EINVAL;  // max is taken before scaling
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, as_stream(stream), (const half_t*)x,
                       (half_t*)y, ldx, ldy, cols, scale);
    return launch_status();
}
