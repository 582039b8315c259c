// store_hazard: minimal stand-alone probe of the gfx950 hazard behind the `s_nop 7` / pinned-register workarounds in the persistent GEMM
// epilogues (gemm_p8 / gemm_q8 / gemm_r8 / gemm_w4 / fused_rows: profiles/r02_gemm_debug.md, VERDICT r3 "weak" item 14).
//
//   build : hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_hazard.hip -o instruct-video-to-video_amd/build/store_hazard
//           (the kernels are hand-written inline asm, so what is measured is exactly the instruction sequence printed below)
//
// Claim under test: `buffer_store_dwordx4 v[a:a+3]` reads its data registers LATE; a VALU instruction that overwrites v[a] right behind
// the store can reach lanes 12-15 of every 16 first when a second wave shares the SIMD.  The probe issues, per iteration,
//     v_mov  v[d..d+3] <- pattern A (per lane, per iteration)
//     buffer_store_dwordx4 v[d..d+3]
//     [PAD x s_nop 0]
//     v_mov  v[d..d+3] <- pattern B              (the "next value")
// with 1, 2 or 4 waves per SIMD resident (workgroup size), optionally while the partner waves run back-to-back VALU or MFMA work, and
// counts stored dwords that show pattern B instead of A, histogrammed by lane % 16.  PAD = 0 .. 8 wait states.
// A second kernel does the same for `v_permlane32_swap` reading a register written by the VALU instruction directly in front of it
// (the other suspicion of round 2): swap(a, b) right after v_mov a, with PAD wait states in between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __amdgpu_buffer_rsrc_t srd_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n\t"
#define NOPS_2 "s_nop 1\n\t"
#define NOPS_4 "s_nop 3\n\t"
#define NOPS_8 "s_nop 7\n\t"

// BUSY: 0 = partner waves idle at a barrier-free spin of s_nop, 1 = partners issue VALU back to back, 2 = partners issue MFMAs
template <int PAD, int BUSY>
__global__ void store_kernel(unsigned* out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const srd_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);
    if (wid != 0) {   // partner waves: keep the SIMDs' issue ports busy for the whole life of wave 0
        float a = (float)lane, b = 1.0001f;
        typedef _Float16 half8 __attribute__((ext_vector_type(8)));
        typedef float floatx16 __attribute__((ext_vector_type(16)));
        half8 x = {1, 2, 3, 4, 5, 6, 7, 8};
        floatx16 c = {};
        for (int i = 0; i < iters * 6; ++i) {
            if (BUSY == 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) a = __builtin_fmaf(a, b, 0.5f);
            } else if (BUSY == 2) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, c, 0, 0, 0);
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (a + c[0] == 12345.678f) out[0] = 1;
        return;
    }
    unsigned voff = (unsigned)(((blockIdx.x * 64 + lane) * 16));
    for (int it = 0; it < iters; ++it) {
        const unsigned pa = 0xA0000000u | (unsigned)(it << 8) | (unsigned)lane, pb = 0xB0000000u | (unsigned)(it << 8) | (unsigned)lane;
        const unsigned soff = (unsigned)it * (gridDim.x * 1024);
#define BODY(NOPS)                                                                                                                  \
        asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\ts_nop 7\n\t"              \
                     "buffer_store_dwordx4 v[40:43], %2, %3, %4 offen\n\t" NOPS                                                    \
                     "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %1\n\t"                       \
                     :: "v"(pa), "v"(pb), "v"(voff), "s"(rs), "s"(soff) : "v40", "v41", "v42", "v43", "memory")
        if (PAD == 0) BODY(NOPS_0); else if (PAD == 1) BODY(NOPS_1); else if (PAD == 2) BODY(NOPS_2); else if (PAD == 4) BODY(NOPS_4); else BODY(NOPS_8);
#undef BODY
    }
}

// the permlane suspicion: VALU write of `a`, PAD wait states, v_permlane32_swap a, b; result stored and checked
template <int PAD, int BUSY>
__global__ void swap_kernel(unsigned* out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (wid != 0) {
        float a = (float)lane, b = 1.0001f;
        for (int i = 0; i < iters * 4; ++i) {
            if (BUSY == 1) {
#pragma unroll
                for (int k = 0; k < 16; ++k) a = __builtin_fmaf(a, b, 0.5f);
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (a == 12345.678f) out[0] = 1;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        unsigned a = 0, b = 0;
        const unsigned pa = 0xA0000000u | (unsigned)(it << 8) | (unsigned)lane, pb = 0xB0000000u | (unsigned)(it << 8) | (unsigned)lane;
#define BODY(NOPS)                                                                                                  \
        asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\t" NOPS \
                     "v_permlane32_swap_b32 %0, %1\n\ts_nop 7" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb))
        if (PAD == 0) BODY(NOPS_0); else if (PAD == 1) BODY(NOPS_1); else if (PAD == 2) BODY(NOPS_2); else if (PAD == 4) BODY(NOPS_4); else BODY(NOPS_8);
#undef BODY
        // after the swap: a = [pa of lanes 0-31 | pb of lanes 0-31], b = [pa of lanes 32-63 | pb of lanes 32-63]
        out[((size_t)it * gridDim.x + blockIdx.x) * 128 + lane] = a;
        out[((size_t)it * gridDim.x + blockIdx.x) * 128 + 64 + lane] = b;
    }
}

template <int PAD, int BUSY>
static void run_store(int waves_per_simd, unsigned* dout, int iters, int nblk) {
    const int threads = waves_per_simd * 4 * 64;
    CK(hipMemset(dout, 0, (size_t)iters * nblk * 1024));
    hipLaunchKernelGGL((store_kernel<PAD, BUSY>), dim3(nblk), dim3(threads), 0, 0, dout, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)iters * nblk * 256);
    CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0, hist[16] = {0};
    for (int it = 0; it < iters; ++it)
        for (int b = 0; b < nblk; ++b)
            for (int l = 0; l < 64; ++l)
                for (int k = 0; k < 4; ++k) {
                    const unsigned v = h[((size_t)it * nblk + b) * 256 + l * 4 + k], want = 0xA0000000u | (unsigned)(it << 8) | (unsigned)l;
                    if (v != want) { ++bad; ++hist[l & 15]; }
                }
    printf("store   pad %d  %d wave(s)/SIMD  partners %-5s : %8ld wrong dwords of %ld; by lane%%16:", PAD, waves_per_simd,
           BUSY == 0 ? "idle" : BUSY == 1 ? "VALU" : "MFMA", bad, (long)h.size());
    for (int i = 0; i < 16; ++i) printf(" %ld", hist[i]);
    printf("\n");
}

template <int PAD, int BUSY>
static void run_swap(int waves_per_simd, unsigned* dout, int iters, int nblk) {
    const int threads = waves_per_simd * 4 * 64;
    CK(hipMemset(dout, 0, (size_t)iters * nblk * 512));
    hipLaunchKernelGGL((swap_kernel<PAD, BUSY>), dim3(nblk), dim3(threads), 0, 0, dout, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)iters * nblk * 128);
    CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0, hist[16] = {0};
    for (int it = 0; it < iters; ++it)
        for (int b = 0; b < nblk; ++b)
            for (int l = 0; l < 64; ++l) {
                const unsigned tag = (unsigned)(it << 8);
                const unsigned wa = l < 32 ? (0xA0000000u | tag | l) : (0xB0000000u | tag | (l - 32));
                const unsigned wb = l < 32 ? (0xA0000000u | tag | (l + 32)) : (0xB0000000u | tag | l);
                const unsigned va = h[((size_t)it * nblk + b) * 128 + l], vb = h[((size_t)it * nblk + b) * 128 + 64 + l];
                if (va != wa) { ++bad; ++hist[l & 15]; }
                if (vb != wb) { ++bad; ++hist[l & 15]; }
            }
    printf("swap32  pad %d  %d wave(s)/SIMD  partners %-5s : %8ld wrong dwords of %ld; by lane%%16:", PAD, waves_per_simd, BUSY ? "VALU" : "idle", bad, (long)h.size() );
    for (int i = 0; i < 16; ++i) printf(" %ld", hist[i]);
    printf("\n");
}

int main() {
    CK(hipSetDevice(0));
    const int iters = 256, nblk = 256;
    unsigned* dout; CK(hipMalloc(&dout, (size_t)iters * nblk * 1024 + 4096));
    printf("store_hazard: %d workgroups x %d iterations; wave 0 of every workgroup runs the probed sequence, the other waves are 'partners'\n", nblk, iters);
    for (int wps : {1, 2, 4}) {
        run_store<0, 0>(wps, dout, iters, nblk); run_store<0, 1>(wps, dout, iters, nblk); run_store<0, 2>(wps, dout, iters, nblk);
        run_store<1, 1>(wps, dout, iters, nblk); run_store<2, 1>(wps, dout, iters, nblk); run_store<4, 1>(wps, dout, iters, nblk); run_store<8, 1>(wps, dout, iters, nblk);
    }
    for (int wps : {1, 2, 4}) {
        run_swap<0, 0>(wps, dout, iters, nblk); run_swap<0, 1>(wps, dout, iters, nblk); run_swap<1, 1>(wps, dout, iters, nblk);
        run_swap<2, 1>(wps, dout, iters, nblk); run_swap<4, 1>(wps, dout, iters, nblk);
    }
    return 0;
}
