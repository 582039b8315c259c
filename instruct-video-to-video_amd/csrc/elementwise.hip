// Small HBM-bound kernels of the sampling loop: UNet input assembly, timestep features,
// fused CFG-combine / noise-correction / scheduler step, optical-flow warp, VAE boundary
// layout conversions.  Latents follow the reference layout [F,4,h,w] fp32 (they are
// API-visible); UNet-side tensors are channels-last.
#include "common.h"
#include <atomic>
#include <cstdio>

__global__ void timestep_embedding_kernel(const float* t, half_t* out, int batch, int dim, float shift) {
    const int half_dim = dim / 2;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * half_dim) return;
    int b = i / half_dim, k = i - b * half_dim;
    float e = -9.210340371976184f * (float)k;  // -ln(10000) * k
    e = e / ((float)half_dim - shift);
    float arg = t[b] * expf(e);
    out[(int64_t)b * dim + k] = (half_t)cosf(arg);             // flip_sin_to_cos=True: cos first
    out[(int64_t)b * dim + half_dim + k] = (half_t)sinf(arg);
}
extern "C" int insv2v_timestep_embedding(const float* t, void* out, int32_t batch, int32_t dim, float shift,
                                         insv2v_stream_t stream) {
    if (!t || !out || batch <= 0 || dim <= 0 || (dim & 1)) return INSV2V_EINVAL;
    int n = batch * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), t,
                       (half_t*)out, batch, dim, shift);
    return launch_status();
}

__global__ void build_unet_input_kernel(const float* latent, const float* cond, half_t* out, float* t_out,
                                        float timestep, int nbranch, int F, int h, int w, int ldo, int64_t branch_rows, int t_stride) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [nbranch,F,h,w]
    int64_t total = (int64_t)nbranch * F * h * w;
    if (i < nbranch && t_out) t_out[i * t_stride] = timestep;
    if (i >= total) return;
    int x = i % w;
    int64_t r = i / w;
    int y = r % h; r /= h;
    int f = r % F;
    int br = r / F;
    half_t* o = out + ((int64_t)br * branch_rows + (i - (int64_t)br * F * h * w)) * ldo;
    const int64_t plane = (int64_t)h * w;
    const int64_t base = (int64_t)f * 4 * plane + (int64_t)y * w + x;
    const bool use_cond = (nbranch == 1) || (br > 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        o[c] = (half_t)latent[base + c * plane];
        o[4 + c] = use_cond ? (half_t)cond[base + c * plane] : (half_t)0.f;
    }
    for (int c = 8; c < ldo; ++c) o[c] = (half_t)0.f;
}
extern "C" int insv2v_build_unet_input(const float* latent, const float* img_cond, void* out, float* t_out,
                                       float timestep, int32_t nbranch, int32_t F, int32_t h, int32_t w,
                                       int32_t ldo, int64_t branch_rows, int32_t t_stride, insv2v_stream_t stream) {
    if (!latent || !img_cond || !out || (nbranch != 1 && nbranch != 3) || ldo < 8) return INSV2V_EINVAL;
    int64_t total = (int64_t)nbranch * F * h * w;
    if (branch_rows == 0) branch_rows = (int64_t)F * h * w;
    if (t_stride == 0) t_stride = 1;
    if (branch_rows < (int64_t)F * h * w || t_stride < 0) return INSV2V_EINVAL;
    hipLaunchKernelGGL(build_unet_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), latent, img_cond, (half_t*)out, t_out, timestep, nbranch, F, h, w, ldo, branch_rows, t_stride);
    return launch_status();
}

// ------------------------------------------------------------------ CFG + correction + step
__device__ __forceinline__ float cfg_eps(const insv2v_step_desc& p, int f, int c, int y, int x) {
    const int64_t hw = (int64_t)p.h * p.w;
    if (p.nbranch == 0) return p.eps_in[((int64_t)f * 4 + c) * hw + (int64_t)y * p.w + x];
    const int64_t pix = ((int64_t)f * p.h + y) * p.w + x;
    const int64_t bstride = p.branch_stride > 0 ? p.branch_stride : (int64_t)p.F * hw * 4;
    float n1 = p.eps_in[pix * 4 + c];
    if (p.nbranch == 1) return n1;
    float n2 = p.eps_in[bstride + pix * 4 + c];
    float n3 = p.eps_in[2 * bstride + pix * 4 + c];
    float e = n1 + p.img_cfg * (n2 - n1) + p.text_cfg * (n3 - n2);
    if (p.guidance_rescale > 0.f && p.rescale_stats) {
        float resc = e * (p.rescale_stats[0] / p.rescale_stats[1]);
        e = p.guidance_rescale * resc + (1.f - p.guidance_rescale) * e;
    }
    return e;
}

// one thread per (c,y,x); loops over frames so the mean over reference frames is thread-local.
__global__ void cfg_step_kernel(insv2v_step_desc p, int do_step) {
    const int64_t hw = (int64_t)p.h * p.w;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * hw) return;
    int c = i / hw;
    int64_t r = i - c * hw;
    int y = r / p.w, x = r - (int64_t)y * p.w;
    float dmean = 0.f;
    if (p.correct == 1) {
        float acc = 0.f;
        for (int f = 0; f < p.R; ++f) {
            int64_t li = ((int64_t)f * 4 + c) * hw + r;
            float e = cfg_eps(p, f, c, y, x);
            acc += (p.latent[li] - p.sqrt_a * p.latent_ref[li]) / p.sqrt_1ma - e;
        }
        dmean = acc / (float)p.R;
    }
    for (int f = 0; f < p.F; ++f) {
        int64_t li = ((int64_t)f * 4 + c) * hw + r;
        float e = cfg_eps(p, f, c, y, x);
        float xt = p.latent[li];
        if (p.correct) {
            if (f < p.R) {
                float d = (xt - p.sqrt_a * p.latent_ref[li]) / p.sqrt_1ma - e;
                e = e + d;
            } else if (p.correct == 1) {
                e = e + dmean;
            } else {
                e = e + p.delta_q[((int64_t)(f - p.R) * 4 + c) * hw + r];
            }
        }
        if (p.eps_out) p.eps_out[li] = e;
        if (do_step) {
            float x0 = (xt - p.sqrt_1ma * e) / p.sqrt_a;
            float prev = p.c_x0 * x0 + p.c_eps * e + p.c_xt * xt;
            if (p.noise) prev += p.c_noise * p.noise[li];
            if (p.pred_x0) p.pred_x0[li] = x0;
            p.latent_out[li] = prev;
        }
    }
}
extern "C" int insv2v_cfg_step(const insv2v_step_desc* dp, insv2v_stream_t stream) {
    if (!dp) return INSV2V_EINVAL;
    insv2v_step_desc d = *dp;
    if (!d.eps_in || !d.latent) return INSV2V_EINVAL;
    if (d.nbranch != 0 && d.nbranch != 1 && d.nbranch != 3) return INSV2V_EINVAL;
    if (d.correct < 0 || d.correct > 2 || d.branch_stride < 0) return INSV2V_EINVAL;
    if (d.correct && (!d.latent_ref || d.R <= 0 || d.R > d.F)) return INSV2V_EINVAL;
    if (d.correct == 2 && !d.delta_q && d.R < d.F) return INSV2V_EINVAL;
    const int do_step = d.latent_out != nullptr;
    if (!do_step && !d.eps_out) return INSV2V_EINVAL;
    int64_t n = 4ll * d.h * d.w;
    hipLaunchKernelGGL(cfg_step_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, as_stream(stream), d, do_step);
    return launch_status();
}

// unbiased std of n1 and of the CFG-combined eps over all elements (single workgroup, two-pass).
__global__ __launch_bounds__(1024) void cfg_stats_kernel(const float* eps_in, float* stats, int F, int h, int w,
                                                         float text_cfg, float img_cfg, int64_t bs) {
    __shared__ float red[2][16];
    __shared__ float mean[2];
    const int64_t n = (int64_t)F * h * w * 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto vals = [&](int64_t i, float& a, float& b) {
        float n1 = eps_in[i], n2 = eps_in[bs + i], n3 = eps_in[2 * bs + i];
        a = n1;
        b = n1 + img_cfg * (n2 - n1) + text_cfg * (n3 - n2);
    };
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = tid; i < n; i += 1024) {
        float a, b;
        vals(i, a, b);
        s0 += a;
        s1 += b;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) { red[0][wid] = s0; red[1][wid] = s1; }
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int j = 0; j < 16; ++j) t += red[tid][j];
        mean[tid] = t / (float)n;
    }
    __syncthreads();
    const float m0 = mean[0], m1 = mean[1];
    s0 = s1 = 0.f;
    for (int64_t i = tid; i < n; i += 1024) {
        float a, b;
        vals(i, a, b);
        s0 += (a - m0) * (a - m0);
        s1 += (b - m1) * (b - m1);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    __syncthreads();
    if (lane == 0) { red[0][wid] = s0; red[1][wid] = s1; }
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int j = 0; j < 16; ++j) t += red[tid][j];
        stats[tid] = sqrtf(t / (float)(n - 1));
    }
}
extern "C" int insv2v_cfg_stats(const float* eps_in, float* stats, int32_t F, int32_t h, int32_t w, float text_cfg,
                                float img_cfg, int64_t branch_stride, insv2v_stream_t stream) {
    if (!eps_in || !stats || F <= 0 || h <= 0 || w <= 0 || branch_stride < 0) return INSV2V_EINVAL;
    const int64_t bs = branch_stride > 0 ? branch_stride : (int64_t)F * h * w * 4;
    hipLaunchKernelGGL(cfg_stats_kernel, dim3(1), dim3(1024), 0, as_stream(stream), eps_in, stats, F, h, w, text_cfg, img_cfg, bs);
    return launch_status();
}

// ------------------------------------------------------------------ optical-flow warp
// grid_sample(bilinear, align_corners=True, padding zeros) at pixel (x+u, y+v); the normalise /
// un-normalise round trip of flow_utils.py:52-55 + grid_sample is replayed in fp32.
__device__ __forceinline__ void warp_coords(float gx, float gy, int W, int H, float& ix, float& iy) {
    float xn = 2.f * (gx / (float)(W - 1) - 0.5f);
    float yn = 2.f * (gy / (float)(H - 1) - 0.5f);
    ix = ((xn + 1.f) / 2.f) * (float)(W - 1);
    iy = ((yn + 1.f) / 2.f) * (float)(H - 1);
}
struct Bilin {
    int x0, y0;
    float w00, w01, w10, w11;  // (y0,x0) (y0,x1) (y1,x0) (y1,x1), zero when out of bounds
};
__device__ __forceinline__ Bilin bilin_setup(float ix, float iy, int W, int H) {
    Bilin b;
    float fx = floorf(ix), fy = floorf(iy);
    b.x0 = (int)fx;
    b.y0 = (int)fy;
    float ax = ix - fx, ay = iy - fy;
    bool x0ok = b.x0 >= 0 && b.x0 < W, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < W;
    bool y0ok = b.y0 >= 0 && b.y0 < H, y1ok = b.y0 + 1 >= 0 && b.y0 + 1 < H;
    b.w00 = (x0ok && y0ok) ? (1.f - ax) * (1.f - ay) : 0.f;
    b.w01 = (x1ok && y0ok) ? ax * (1.f - ay) : 0.f;
    b.w10 = (x0ok && y1ok) ? (1.f - ax) * ay : 0.f;
    b.w11 = (x1ok && y1ok) ? ax * ay : 0.f;
    return b;
}
template <typename Fn>
__device__ __forceinline__ float bilin_sample(const Bilin& b, int W, Fn at) {
    float v = 0.f;
    if (b.w00 != 0.f) v += b.w00 * at(b.y0, b.x0);
    if (b.w01 != 0.f) v += b.w01 * at(b.y0, b.x0 + 1);
    if (b.w10 != 0.f) v += b.w10 * at(b.y0 + 1, b.x0);
    if (b.w11 != 0.f) v += b.w11 * at(b.y0 + 1, b.x0 + 1);
    return v;
}

__global__ void warp_image_kernel(const float* img, const float* flow, float* out, int N, int C, int H, int W) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t hw = (int64_t)H * W;
    if (i >= N * hw) return;
    int n = i / hw;
    int64_t r = i - n * hw;
    int y = r / W, x = r - (int64_t)y * W;
    float ix, iy;
    warp_coords((float)x + flow[((int64_t)n * 2) * hw + r], (float)y + flow[((int64_t)n * 2 + 1) * hw + r], W, H, ix, iy);
    Bilin b = bilin_setup(ix, iy, W, H);
    for (int c = 0; c < C; ++c) {
        const float* pl = img + ((int64_t)n * C + c) * hw;
        out[((int64_t)n * C + c) * hw + r] = bilin_sample(b, W, [&](int yy, int xx) { return pl[(int64_t)yy * W + xx]; });
    }
}
extern "C" int insv2v_warp_image(const float* image, const float* flow, float* out, int32_t N, int32_t C, int32_t H,
                                 int32_t W, insv2v_stream_t stream) {
    if (!image || !flow || !out || N <= 0 || C <= 0 || H <= 1 || W <= 1) return INSV2V_EINVAL;
    int64_t n = (int64_t)N * H * W;
    hipLaunchKernelGGL(warp_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), image, flow, out, N, C, H, W);
    return launch_status();
}

// F.interpolate(bilinear, align_corners=False) of the pre-scaled flow.
__global__ void resize_flow_kernel(const float* flow, float* out, int N, int h, int w, int H, int W) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t HW = (int64_t)H * W;
    if (i >= (int64_t)N * 2 * HW) return;
    int nc = i / HW;
    int64_t r = i - nc * HW;
    int Y = r / W, X = r - (int64_t)Y * W;
    const float sx = (float)w / (float)W, sy = (float)h / (float)H;
    float fx = fmaxf(sx * ((float)X + 0.5f) - 0.5f, 0.f), fy = fmaxf(sy * ((float)Y + 0.5f) - 0.5f, 0.f);
    int x0 = (int)fx, y0 = (int)fy;
    int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    float lx = fx - (float)x0, ly = fy - (float)y0;
    const float* pl = flow + (int64_t)nc * h * w;
    const float mul = (nc & 1) ? (float)((double)H / (double)h) : (float)((double)W / (double)w);
    float v00 = pl[(int64_t)y0 * w + x0] * mul, v01 = pl[(int64_t)y0 * w + x1] * mul;
    float v10 = pl[(int64_t)y1 * w + x0] * mul, v11 = pl[(int64_t)y1 * w + x1] * mul;
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}
extern "C" int insv2v_resize_flow(const float* flow, float* out, int32_t N, int32_t h, int32_t w, int32_t H, int32_t W,
                                  insv2v_stream_t stream) {
    if (!flow || !out || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return INSV2V_EINVAL;
    int64_t n = (int64_t)N * 2 * H * W;
    hipLaunchKernelGGL(resize_flow_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), flow, out, N, h, w, H, W);
    return launch_status();
}

__global__ void flow_correction_kernel(const float* eps, const float* latent, const float* ref, const float* flows,
                                       float* delta_q, int F, int R, int h, int w, float sqrt_a, float sqrt_1ma) {
    const int64_t hw = (int64_t)h * w;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Q = F - R;
    if (i >= Q * hw) return;
    int qi = i / hw;
    int64_t r = i - qi * hw;
    int y = r / w, x = r - (int64_t)y * w;
    float wsum[4] = {0.f, 0.f, 0.f, 0.f}, msum = 0.f;
    for (int rf = 0; rf < R; ++rf) {
        const float* fl = flows + (((int64_t)qi * R + rf) * 2) * hw;
        float ix, iy;
        warp_coords((float)x + fl[r], (float)y + fl[hw + r], w, h, ix, iy);
        Bilin b = bilin_setup(ix, iy, w, h);
        msum += bilin_sample(b, w, [&](int, int) { return 1.f; });
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t pl = ((int64_t)rf * 4 + c) * hw;
            wsum[c] += bilin_sample(b, w, [&](int yy, int xx) {
                int64_t li = pl + (int64_t)yy * w + xx;
                return (latent[li] - sqrt_a * ref[li]) / sqrt_1ma - eps[li];
            });
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        delta_q[((int64_t)qi * 4 + c) * hw + r] = msum > 0.5f ? wsum[c] / msum : 0.f;
}
extern "C" int insv2v_flow_correction(const float* eps_cfg, const float* latent, const float* latent_ref,
                                      const float* flows, float* delta_q, int32_t F, int32_t R, int32_t h, int32_t w,
                                      float sqrt_a, float sqrt_1ma, insv2v_stream_t stream) {
    if (!eps_cfg || !latent || !latent_ref || !flows || !delta_q || R <= 0 || R >= F || h <= 1 || w <= 1) return INSV2V_EINVAL;
    int64_t n = (int64_t)(F - R) * h * w;
    hipLaunchKernelGGL(flow_correction_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, as_stream(stream), eps_cfg,
                       latent, latent_ref, flows, delta_q, F, R, h, w, sqrt_a, sqrt_1ma);
    return launch_status();
}

// ------------------------------------------------------------------ VAE boundary
__global__ void nchw_to_nhwc_kernel(const float* x, half_t* y, int N, int C, int H, int W, int ldo, float scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over N*H*W
    int64_t hw = (int64_t)H * W;
    if (i >= N * hw) return;
    int n = i / hw;
    int64_t r = i - n * hw;
    half_t* o = y + i * ldo;
    for (int c = 0; c < ldo; ++c) o[c] = c < C ? (half_t)(x[((int64_t)n * C + c) * hw + r] * scale) : (half_t)0.f;
}
extern "C" int insv2v_nchw_to_nhwc_f16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ldo,
                                       float scale, insv2v_stream_t stream) {
    if (!x || !y || N <= 0 || C <= 0 || ldo < C) return INSV2V_EINVAL;
    int64_t n = (int64_t)N * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, (half_t*)y, N, C, H, W, ldo, scale);
    return launch_status();
}
__global__ void nhwc_to_nchw_kernel(const void* x, int is32, float* y, int N, int C, int H, int W, int ldx, float scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over N*C*H*W (output order)
    int64_t hw = (int64_t)H * W;
    if (i >= (int64_t)N * C * hw) return;
    int64_t r = i % hw;
    int64_t nc = i / hw;
    int c = nc % C, n = nc / C;
    int64_t src = ((int64_t)n * hw + r) * ldx + c;
    float v = is32 ? ((const float*)x)[src] : (float)((const half_t*)x)[src];
    y[i] = v * scale;
}
extern "C" int insv2v_nhwc_to_nchw_f32(const void* x, int32_t x_is_fp32, float* y, int32_t N, int32_t C, int32_t H,
                                       int32_t W, int32_t ldx, float scale, insv2v_stream_t stream) {
    if (!x || !y || N <= 0 || C <= 0 || ldx < C) return INSV2V_EINVAL;
    int64_t n = (int64_t)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, x_is_fp32, y, N, C, H, W, ldx, scale);
    return launch_status();
}
__global__ void posterior_sample_kernel(const float* mom, const float* noise, float* z, int N, int H, int W, int ldm, float scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over N*4*H*W
    int64_t hw = (int64_t)H * W;
    if (i >= (int64_t)N * 4 * hw) return;
    int64_t r = i % hw;
    int64_t nc = i / hw;
    int c = nc % 4, n = nc / 4;
    const float* m = mom + ((int64_t)n * hw + r) * ldm;
    float logvar = fminf(fmaxf(m[4 + c], -30.f), 20.f);
    z[i] = (m[c] + expf(0.5f * logvar) * noise[i]) * scale;
}
extern "C" int insv2v_posterior_sample(const float* moments, const float* noise, float* z, int32_t N, int32_t H,
                                       int32_t W, int32_t ldm, float scale, insv2v_stream_t stream) {
    if (!moments || !noise || !z || N <= 0 || ldm < 8) return INSV2V_EINVAL;
    int64_t n = (int64_t)N * 4 * H * W;
    hipLaunchKernelGGL(posterior_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), moments, noise, z, N, H, W, ldm, scale);
    return launch_status();
}

// ---- narrow-output 3x3 convolution, second half (ABI 12): out[p, c] = bias[c] + sum over the 9 taps t of y9[p + offset(t), 4 t + c], zero padding.
// y9 = x . Wtap^T is one plain GEMM over the Cin channels (row 4 t + c of Wtap = W[c, :, ky, kx], t = 3 ky + kx; insv2v_gemm, fp32 out): the
// convolution's 9 Cin multiply-adds per output are done ONCE per pixel and tap instead of on a tile padded from <= 4 to 64 output channels
// (conv_out of the UNet, 320 -> 4 at 1 474 560 pixels: 819 us of MFMA work on padding; unet.py:432-434, resnet.py InflatedConv3d).
__global__ __launch_bounds__(256) void tap_gather_kernel(const float* __restrict__ y9, int64_t ld9, const float* __restrict__ bias, float* __restrict__ out,
                                                         int64_t ldo, int64_t M, int H, int W, int cout) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int x = (int)(m % W), y = (int)((m / W) % H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) { acc.x = bias[0]; if (cout > 1) acc.y = bias[1]; if (cout > 2) acc.z = bias[2]; if (cout > 3) acc.w = bias[3]; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float4 v = *(const float4*)(y9 + (m + (int64_t)(ky - 1) * W + (kx - 1)) * ld9 + (ky * 3 + kx) * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    float* o = out + m * ldo;
    o[0] = acc.x; if (cout > 1) o[1] = acc.y; if (cout > 2) o[2] = acc.z; if (cout > 3) o[3] = acc.w;
}
extern "C" int insv2v_tap_gather(const float* y9, int64_t ld9, const float* bias, float* out, int64_t ldo, int32_t NB, int32_t H, int32_t W,
                                 int32_t cout, insv2v_stream_t stream) {
    if (!one_device()) return INSV2V_EINVAL;
    if (!y9 || !out || NB <= 0 || H <= 0 || W <= 0 || cout <= 0 || cout > 4 || ld9 < 36 || (ld9 & 3) || ((uintptr_t)y9 & 15) || ldo < cout) return INSV2V_EINVAL;
    const int64_t M = (int64_t)NB * H * W;
    hipLaunchKernelGGL(tap_gather_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), y9, ld9, bias, out, ldo, M, H, W, cout);
    return launch_status();
}

extern "C" int insv2v_abi_version(void) { return 12; }
// The process's device (one process per GPU): latched once, by insv2v_init or by the first launcher that asks.
static std::atomic<int> g_first_device{-1};
bool insv2v_one_device_check() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    int expected = -1;
    if (g_first_device.compare_exchange_strong(expected, dev)) return true;
    if (expected != dev) {
        static std::atomic<bool> told{false};
        if (!told.exchange(true))
            fprintf(stderr, "insv2v: launch refused - device %d is current, but this process latched device %d on its first call "
                            "(one process drives one GPU; call torch.cuda.set_device before the first insv2v call)\n", dev, expected);
        return false;
    }
    return true;
}
extern "C" int insv2v_init(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    return insv2v_one_device_check() ? 0 : INSV2V_EINVAL;
}

// ---- CLIP text embeddings: one thread per 8 channels of one token row
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* ids, const half_t* tok, const half_t* pos, half_t* out,
                                                           int rows, int L, int C, int vocab) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = C / 8;
    if (idx >= (int64_t)rows * cpr) return;
    const int r = (int)(idx / cpr), c = (int)(idx - (int64_t)r * cpr) * 8;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // the host wrapper rejects out-of-range ids; never read outside the table
    const half8 t = *(const half8*)(tok + id * C + c);
    const half8 q = *(const half8*)(pos + (int64_t)(r % L) * C + c);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)t[e] + (float)q[e]);
    *(half8*)(out + (int64_t)r * C + c) = o;
}

extern "C" int insv2v_embed_tokens(const int64_t* ids, const void* tok, const void* pos, void* out, int32_t rows, int32_t L,
                                   int32_t C, int32_t vocab, insv2v_stream_t stream) {
    if (!ids || !tok || !pos || !out || rows <= 0 || L <= 0 || C <= 0 || (C & 7) || vocab <= 0) return INSV2V_EINVAL;
    const int64_t n = (int64_t)rows * (C / 8);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), ids,
                       (const half_t*)tok, (const half_t*)pos, (half_t*)out, rows, L, C, vocab);
    return launch_status();
}
