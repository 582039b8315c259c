// phase_rate: what does one barrier interval of the 8-wave ping-pong ("8-phase") GEMM loop cost on gfx950, piece by piece?
//
//   build : hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/phase_rate.hip -o instruct-video-to-video_amd/build/phase_rate
//   usage : phase_rate [iters]
//
// One 512-thread workgroup per CU (256 blocks), 128 KiB of LDS like gemm_p8 / gemm_q8.  Waves 0-3 (group 0) and 4-7 (group 1) sit
// two per SIMD.  Every variant is a loop of `iters` PHASES; a phase is two barrier intervals:
//      group 0:  [load segment] s_barrier [8 x v_mfma_f32_32x32x16_f16] s_barrier
//      group 1:  the same, one interval later (it takes one extra barrier in front)
// so in every interval one wave of a SIMD issues MFMAs while its partner is in its load segment.  The ideal interval is the
// MFMA issue time alone: 8 x 32 = 256 cycles.  Variants add the load segment's contents one at a time:
//      0  barriers only (no MFMA, no loads)                 -> s_barrier round trip
//      1  MFMAs + barriers, empty load segment              -> the structure's floor
//      2  + 12 ds_read_b128 in the load segment (lgkmcnt(0) waited in the MFMA segment, as the GEMMs do)
//      3  + 2 LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB each) per load segment, vmcnt(6) once per 4 phases... per phase here
//      4  = 3 without the ds_reads (DMA only)
//      5  MFMAs only, NO barriers, both groups (what two free-running waves per SIMD reach: 16 MFMAs per phase per SIMD)
//      6  = 3 but ONE barrier per interval pair dropped: group-local... (not built)
// Output: shader cycles per interval (s_memtime, wave 0 of block 0 and the mean over blocks), and the implied matrix-pipe
// utilisation 256 / cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t srd_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
#define BARRIER() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

template <int VAR>
__global__ __launch_bounds__(512) void phase_kernel(const _Float16* src, float* sink, unsigned long long* cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;
    const srd_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(src), 0, 0x7FFFFFFF, 0x00020000);
    // fragment read addresses: conflict-free pattern of the GEMMs (row = lane & 31, chunk swizzled by (row>>1)&7)
    const int frow = lane & 31, fhi = lane >> 5, fsw = (frow >> 1) & 7;
    const char* rd[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) rd[kk] = smem + ((wid & 3) * 32 + frow) * 128 + (((kk * 2 + fhi) ^ fsw) * 16);
    for (int i = tid; i < 32768; i += 512) ((float*)smem)[i] = 0.001f * (float)((i * 7 + 3) & 255);
    __syncthreads();
    half8 fa[2][4], fw[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        fw[kk] = *(const half8*)(rd[kk]);
        fa[0][kk] = *(const half8*)(rd[kk] + 16384);
        fa[1][kk] = *(const half8*)(rd[kk] + 32768);
    }
    floatx16 c[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { c[0][r] = 0.f; c[1][r] = 0.f; }
    const int rstride = VAR == 11 ? 5760 : VAR == 12 ? 16384 : 0;
    const unsigned voff = rstride ? (unsigned)(((blockIdx.x * 64 + wid * 8 + (lane >> 3)) * rstride + (lane & 7) * 16) & 0x7fffffff)
                                  : (unsigned)((blockIdx.x * 512 + tid) * 16) & 0x001fffffu;   // a 64 MiB window, L2-resident after the first pass
    int soff = 0;
    const int WIN = VAR == 10 ? 0x7fffffff & ~16383 : (VAR == 11 || VAR == 12) ? 0x0fff : 0x00ffffff;   // 11 / 12: k advances by 128 B per piece pair inside 4 KiB rows   // VAR 10: stream 2 GiB (HBM), else a 16 MiB window (L2 / MALL)
    __syncthreads();
    if (VAR != 5 && grp == 1) BARRIER();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        // ---- load segment
        auto dma2 = [&]() {
            char* dst = smem + 49152 + ((it & 3) * 16384) + wid * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 8192), 16, voff, soff + 8192, 0, 0);
            soff = (soff + 16384) & WIN;
        };
        auto dma1 = [&](int i) {
            char* dst = smem + 49152 + ((it & 3) * 16384) + wid * 1024 + i * 8192;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, (VAR == 11 || VAR == 12) ? soff + i * rstride * 2048 : soff + i * 8192, 0, 0);
            if (i) soff = (VAR == 11 || VAR == 12) ? ((soff + 128) & WIN) : ((soff + 16384) & WIN);
        };
        if (VAR == 6) { dma2(); SB(); }
        if (VAR == 2 || VAR == 3 || VAR == 6 || VAR == 7 || VAR == 8 || VAR == 9 || VAR == 10 || VAR == 11 || VAR == 12) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fw[kk] = *(const half8*)(rd[kk] + ((it & 1) << 16));
            SB();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fa[0][kk] = *(const half8*)(rd[kk] + 16384 + ((it & 1) << 16));
                fa[1][kk] = *(const half8*)(rd[kk] + 32768 + ((it & 1) << 16));
            }
            SB();
        }
        if (VAR == 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); }
        if (VAR == 3 || VAR == 4 || VAR == 8 || VAR == 10) dma2();
        if (VAR == 3 || VAR == 4 || VAR == 6 || VAR == 7 || VAR == 8 || VAR == 9 || VAR == 10 || VAR == 11 || VAR == 12) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (VAR != 5) BARRIER();
        if (VAR == 9) { dma2(); SB(); }
        // ---- MFMA segment
        if (VAR != 0) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk], fa[j][kk], c[j], 0, 0, 0);
                    if ((VAR == 7 || VAR == 11 || VAR == 12) && (kk * 2 + j == 1 || kk * 2 + j == 4)) { SB(); dma1(kk * 2 + j == 4); SB(); }
                }
            __builtin_amdgcn_s_setprio(0);
        }
        if (VAR != 5) BARRIER();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (VAR != 5 && grp == 0) BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c[0][r] + c[1][r];
    if (s == 12345.678f) sink[tid] = s;
    if (lane == 0) cycles[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int VAR>
static void run(const char* name, const _Float16* src, float* sink, unsigned long long* dcyc, int iters, int nblk) {
    CK(hipFuncSetAttribute((const void*)phase_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(phase_kernel<VAR>, dim3(nblk), dim3(512), 131072, 0, src, sink, dcyc, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nblk * 8);
    CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double per_phase = mean / iters, per_interval = per_phase / 2;
    const double mfma_cyc = VAR == 0 ? 0 : 256.0;
    printf("%-58s %8.1f cycles / interval (block 0 wave 0: %8.1f)  pipe use %5.1f %%   wall %7.1f us, %.0f MHz\n", name, per_interval,
           (double)h[0] / iters / 2, 100.0 * mfma_cyc / per_interval, ms * 1e3, mean / (ms * 1e3));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nblk = prop.multiProcessorCount;
    _Float16* src; CK(hipMalloc(&src, (2l << 30) + (64l << 20)));
    CK(hipMemset(src, 0x3c, (2l << 30) + (64l << 20)));
    float* sink; CK(hipMalloc(&sink, 4096));
    unsigned long long* dcyc; CK(hipMalloc(&dcyc, nblk * 8 * 8));
    printf("phase_rate: %d blocks x 512 threads, %d phases (2 intervals each); ideal interval = 256 cycles (8 MFMA 32x32x16)\n", nblk, iters);
    run<0>("0 barriers only", src, sink, dcyc, iters, nblk);
    run<1>("1 MFMA + barriers, empty load segment", src, sink, dcyc, iters, nblk);
    run<2>("2 + 12 ds_read_b128 per load segment", src, sink, dcyc, iters, nblk);
    run<3>("3 + 2 LDS-DMA pieces per load segment (vmcnt(6))", src, sink, dcyc, iters, nblk);
    run<4>("4 MFMA + barriers + 2 LDS-DMA pieces, no ds_read", src, sink, dcyc, iters, nblk);
    run<5>("5 MFMA only, no barriers (2 free-running waves per SIMD)", src, sink, dcyc, iters, nblk);
    run<6>("6 = 3 with the LDS-DMA pieces IN FRONT of the ds_reads", src, sink, dcyc, iters, nblk);
    run<7>("7 ds_reads in the load segment, LDS-DMA behind MFMA 2 and 5", src, sink, dcyc, iters, nblk);
    run<8>("8 = 3 with lgkmcnt(0) between the ds_reads and the DMA", src, sink, dcyc, iters, nblk);
    run<9>("9 ds_reads in the load segment, LDS-DMA at the head of the MFMA segment", src, sink, dcyc, iters, nblk);
    run<10>("10 = 3 streaming 2 GiB instead of a 16 MiB window", src, sink, dcyc, iters, nblk);
    run<11>("11 = 7 with GEMM-shaped pieces: 8 rows x 128 B, row stride 5760 B", src, sink, dcyc, iters, nblk);
    run<12>("12 = 7 with GEMM-shaped pieces: 8 rows x 128 B, row stride 16 KiB", src, sink, dcyc, iters, nblk);
    return 0;
}
