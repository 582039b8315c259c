// Micro-benchmark: issue rate of the fp16 MFMA shapes on gfx950 (independent accumulator chains, operands in registers).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o instruct-video-to-video_amd/build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float s = 0.f;
    if (SHAPE == 0) {  // 32x32x16
        floatx16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < CHAINS; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    } else if (SHAPE == 1) {  // 16x16x32
        floatx4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) for (int v = 0; v < 4; ++v) acc[c][v] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < CHAINS; ++c) for (int v = 0; v < 4; ++v) s += acc[c][v];
    } else {  // 16x16x16 (legacy)
        floatx4 acc[CHAINS];
        half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        for (int c = 0; c < CHAINS; ++c) for (int v = 0; v < 4; ++v) acc[c][v] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[c], 0, 0, 0);
        for (int c = 0; c < CHAINS; ++c) for (int v = 0; v < 4; ++v) s += acc[c][v];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 32x32x16 MFMAs (2 chains) with NV independent v_fma_f32 (VGPR operands, not the accumulators) placed behind every MFMA of the
// SAME wave: does the VALU work hide in the 32-cycle shadow of the MFMA?
template <int NV>
__global__ __launch_bounds__(256) void kv(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    floatx16 acc[2];
    for (int c = 0; c < 2; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
    const float m = 1.0001f, d = 0.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i % 8]) : "v"(m), "v"(d));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 2; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void runv(int waves_per_simd) {
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, iters = 20000;
    const int blocks = cus * waves_per_simd;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kv<NV>), dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kv<NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("32x32x16 + %2d v_fma_f32 behind every MFMA, waves/SIMD %d: %6.2f ns per MFMA per wave (MFMA-only: see above)\n", NV, waves_per_simd,
           ms * 1e6 / ((double)iters * 2));
    hipFree(out);
}

template <int SHAPE, int CHAINS>
void run(const char* name, double flops_per_mfma, int waves_per_simd) {
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, iters = 20000;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * CHAINS;
    printf("%-10s chains %d waves/SIMD %d: %7.1f TF/s  (%.2f ns per MFMA per SIMD)\n", name, CHAINS, waves_per_simd,
           n * flops_per_mfma / ms / 1e9, ms * 1e6 / ((double)iters * CHAINS * waves_per_simd));
    hipFree(out);
}

// Sustained run for clock logging (round 3): 32x32x16, 4 chains, one wave per SIMD, operands all zero ("zero") or the non-zero
// per-thread values above scaled by `scale` ("rand"), for ~`seconds`; prints TF/s once per launch (~0.25 s) with a wall-clock stamp so
// a parallel `rocm-smi --showclocks` log can be lined up.  With s_memtime ticks per launch it also reports the shader clock the
// kernel itself saw: cycles per MFMA x MFMAs / time.
__global__ __launch_bounds__(256) void ksus(float* out, unsigned long long* cyc, int iters, float scale) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)((threadIdx.x * 0.001f + i) * scale); b[i] = (_Float16)((i * 0.5f - threadIdx.x * 0.002f) * scale); }
    floatx16 acc[4];
    for (int c = 0; c < 4; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
// Round 6: the two fp16 shapes side by side at the power cap - same uniform [-1,1) operands (a different fragment per chain, as a GEMM's
// K loop has), 16 accumulator registers' worth of FLOPs per chain group: SHAPE 0 = 4 chains of 32x32x16 (64 accumulator registers),
// SHAPE 1 = 16 chains of 16x16x32 (64 accumulator registers); one wave per SIMD (waves = 1) or two.
template <int SHAPE>
__global__ __launch_bounds__(512) void kshape(float* out, int iters) {
    unsigned x = threadIdx.x * 2654435761u + 12345u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return (_Float16)((float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f); };
    float s = 0.f;
    if (SHAPE == 0) {
        half8 a[4], b[4];
        for (int c = 0; c < 4; ++c) for (int i = 0; i < 8; ++i) { a[c][i] = rnd(); b[c][i] = rnd(); }
        floatx16 acc[4];
        for (int c = 0; c < 4; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[(c + (it & 1)) & 3], acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    } else {
        half8 a[4], b[4];
        for (int c = 0; c < 4; ++c) for (int i = 0; i < 8; ++i) { a[c][i] = rnd(); b[c][i] = rnd(); }
        floatx4 acc[16];
        for (int c = 0; c < 16; ++c) for (int v = 0; v < 4; ++v) acc[c][v] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c & 3], b[((c >> 2) + (it & 1)) & 3], acc[c], 0, 0, 0);
        for (int c = 0; c < 16; ++c) for (int v = 0; v < 4; ++v) s += acc[c][v];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#include <chrono>
#include <cstring>
template <int SHAPE>
static int sustained_shape(double seconds, int waves) {
    int dev = 0; hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, iters = 1000000;
    float* out; (void)hipMalloc(&out, (size_t)cus * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const auto w0 = std::chrono::steady_clock::now();
    double best = 0, last = 0;
    for (;;) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kshape<SHAPE>, dim3(cus), dim3(256 * waves), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        const double flops = (double)cus * 4 * waves * iters * (SHAPE == 0 ? 4 * 32768.0 : 16 * 16384.0);
        last = flops / ms / 1e9; if (last > best) best = last;
        if (el > seconds) break;
    }
    printf("bare %s chains, uniform [-1,1) operands, %d wave(s) per SIMD: %.1f TFLOP/s sustained (last launch; best %.1f)\n", SHAPE == 0 ? "32x32x16" : "16x16x32", waves, last, best);
    return 0;
}
static int sustained(const char* mode, double seconds) {
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, iters = 4000000;
    const float scale = strcmp(mode, "zero") == 0 ? 0.f : 1.f;
    float* out; hipMalloc(&out, (size_t)cus * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const auto w0 = std::chrono::steady_clock::now();
    for (;;) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(ksus, dim3(cus), dim3(256), 0, 0, out, cyc, iters, scale);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        const double nmf = (double)iters * 4;   // MFMAs per wave
        printf("t=%6.2fs %s operands: %7.1f TF/s, %.2f ns per MFMA per SIMD, %.2f s_memtime ticks per MFMA (100 MHz counter => %.0f MHz if 32 clk/MFMA)\n", el, mode,
               (double)cus * 4 * nmf * 32768 / ms / 1e9, ms * 1e6 / nmf, (double)c / nmf, 32.0 / (ms * 1e6 / nmf) * 1e3);
        fflush(stdout);
        if (el > seconds) break;
    }
    return 0;
}

// Round 5: what LDS operand reads cost at the power cap.  512 threads (two waves per SIMD, like the 8-wave GEMM engine), ten independent
// 32x32x16 chains per wave (160 accumulator registers); R of every ten MFMAs take a FRESH A fragment read from LDS (ds_read_b128 of
// random fp16 data, conflict-free, immediate offsets), the others reuse the previous one.  R = 7 is gemm_r8's 64 x 160 wave tile
// (0.7 reads per MFMA), R = 10 the row kernels (one per MFMA), R = 4 / 5 what a 128 x 160 wave tile would need (0.45).  Sustained, so the
// rate settles where the board's power management leaves it; `rocm-smi` is sampled beside it (tools/r05/run12_lds_energy.sh).
template <int R>
__global__ __launch_bounds__(512) void klds(float* out, const half8* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* lds = (half8*)smem;
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) lds[i] = src[i];
    __syncthreads();
    half8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.37f * i - 0.002f * (threadIdx.x & 63) - 1.1f);
    // two fragment sets swap roles: the reads of one iteration feed the MFMAs of the next (a whole iteration of latency cover)
    constexpr int NF = R > 0 ? R : 1;
    half8 X[NF], Y[NF];
    for (int c = 0; c < NF; ++c) { X[c] = lds[(threadIdx.x & 63) + 64 * c]; Y[c] = lds[(threadIdx.x & 63) + 64 * (c + NF)]; }
    floatx16 acc[10];
    for (int c = 0; c < 10; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it += 2) {
        const half8* base = lds + ((it & 2) * 1024 + lane);   // 16 KiB windows: one address register, immediate offsets
#pragma unroll
        for (int c = 0; c < R; ++c) Y[c] = base[c * 64];
#pragma unroll
        for (int c = 0; c < 10; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[c % NF], b, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < R; ++c) X[c] = base[(c + 16) * 64];
#pragma unroll
        for (int c = 0; c < 10; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Y[c % NF], b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 10; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int R>
static int sustained_lds(double seconds) {
    int dev = 0; hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, iters = 400000;
    float* out; (void)hipMalloc(&out, (size_t)cus * 512 * 4);
    half8* src; (void)hipMalloc(&src, 65536);
    {
        _Float16* h = (_Float16*)malloc(65536);
        unsigned x = 12345u;
        for (int i = 0; i < 32768; ++i) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((x >> 8) & 0xffff) / 32768.0f - 1.0f); }
        (void)hipMemcpy(src, h, 65536, hipMemcpyHostToDevice);
        free(h);
    }
    (void)hipFuncSetAttribute((const void*)klds<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const auto w0 = std::chrono::steady_clock::now();
    for (;;) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(klds<R>, dim3(cus), dim3(512), 65536, 0, out, src, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        const double nmf = (double)iters * 10;   // MFMAs per wave
        printf("t=%6.2fs lds R=%d of 10: %7.1f TF/s, %.2f ns per MFMA per SIMD (2 waves per SIMD)\n", el, R, (double)cus * 8 * nmf * 32768 / ms / 1e9, ms * 1e6 / (2 * nmf));
        fflush(stdout);
        if (el > seconds) break;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && strcmp(argv[1], "lds") == 0) {
        const int r = atoi(argv[2]);
        const double sec = argc >= 4 ? atof(argv[3]) : 6.0;
        switch (r) {
            case 0: return sustained_lds<0>(sec);
            case 2: return sustained_lds<2>(sec);
            case 4: return sustained_lds<4>(sec);
            case 5: return sustained_lds<5>(sec);
            case 7: return sustained_lds<7>(sec);
            case 10: return sustained_lds<10>(sec);
        }
        return 1;
    }
    if (argc >= 2 && strcmp(argv[1], "shape32") == 0) return sustained_shape<0>(argc >= 3 ? atof(argv[2]) : 5.0, argc >= 4 ? atoi(argv[3]) : 1);
    if (argc >= 2 && strcmp(argv[1], "shape16") == 0) return sustained_shape<1>(argc >= 3 ? atof(argv[2]) : 5.0, argc >= 4 ? atoi(argv[3]) : 1);
    if (argc >= 2 && (strcmp(argv[1], "zero") == 0 || strcmp(argv[1], "rand") == 0)) return sustained(argv[1], argc >= 3 ? atof(argv[2]) : 5.0);
    run<0, 1>("32x32x16", 32768, 1); run<0, 2>("32x32x16", 32768, 1); run<0, 4>("32x32x16", 32768, 1); run<0, 4>("32x32x16", 32768, 2);
    run<1, 1>("16x16x32", 16384, 1); run<1, 2>("16x16x32", 16384, 1); run<1, 4>("16x16x32", 16384, 1); run<1, 8>("16x16x32", 16384, 1); run<1, 8>("16x16x32", 16384, 2);
    run<2, 4>("16x16x16", 8192, 1); run<2, 8>("16x16x16", 8192, 2);
    runv<0>(1); runv<2>(1); runv<4>(1); runv<6>(1); runv<8>(1); runv<12>(1); runv<16>(1);
    runv<0>(2); runv<4>(2); runv<8>(2); runv<16>(2);
    return 0;
}
