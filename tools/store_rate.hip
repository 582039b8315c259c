// store_rate: what does the 256 x 256 tile epilogue's store pattern cost per CU on gfx950?
//
//   build : hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_rate.hip -o instruct-video-to-video_amd/build/store_rate
//
// One 512-thread workgroup per CU writes `tiles` fp16 tiles of 256 rows x 256 columns (128 KiB each) of a row-major [M, N] matrix
// with 16-byte stores, 16 store instructions per wave per tile, in different lane -> (row, 16-byte chunk) assignments:
//   0  32 rows x 32 B per instruction   (gemm_p8 / gemm_q8 today: lane = (row, half), the MFMA C layout after one permlane32_swap)
//   1  16 rows x 64 B
//   2   8 rows x 128 B                  (a full cache line per 8 lanes)
//   3   4 rows x 256 B
//   4   2 rows x 512 B                  (a whole tile row per 32 lanes)
// Reported: shader cycles per tile per CU (s_memtime around the loop incl. the final vmcnt(0)), bytes per cycle per CU, and the
// wall-clock bandwidth of the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t srd_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int VAR>
__global__ __launch_bounds__(512) void store_kernel(void* out, int ld, int tiles_n, int tiles, unsigned long long* cycles) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const srd_t rc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);
    // lanes of one instruction cover RPI rows x CPI 16-byte chunks; the wave covers its share of the tile with 16 instructions
    constexpr int CPI = VAR == 0 ? 2 : VAR == 1 ? 4 : VAR == 2 ? 8 : VAR == 3 ? 16 : 32;   // chunks per row per instruction
    constexpr int RPI = 64 / CPI;
    const int lrow = lane / CPI, lchunk = lane % CPI;
    uint4v v = {(unsigned)tid, 1u, 2u, 3u};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < tiles; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        // the tile = 256 rows x 32 chunks; instruction i of wave w: a block of RPI rows x CPI chunks; blocks are numbered row-block major
        // so that one wave's 16 instructions stay inside a band of rows (as the GEMM waves own row bands)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int blk = wid * 16 + i;                 // 0..127
            const int cblocks = 32 / CPI;                 // chunk blocks per row
            const int rb = blk / cblocks, cb = blk % cblocks;
            const int row = tm * 256 + rb * RPI + lrow, chunk = tn * 32 + cb * CPI + lchunk;
            __builtin_amdgcn_raw_buffer_store_b128(v, rc, (unsigned)(row * ld * 2 + chunk * 16), 0, 0);
            v[1] += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cycles[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int VAR>
static void run(const char* name, void* out, int M, int N, unsigned long long* dcyc, int nblk) {
    const int tiles_n = N / 256, tiles_m = M / 256, per = tiles_m * tiles_n / nblk;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(store_kernel<VAR>, dim3(nblk), dim3(512), 0, 0, out, N, tiles_n, per, dcyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(nblk * 8);
    CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    printf("%-34s M=%6d N=%5d  %8.0f cycles / tile / CU = %5.1f B/clk/CU   chip %6.2f TB/s (%7.1f us)\n", name, M, N, mean / per, 131072.0 * per / mean,
           (double)per * nblk * 131072 / (ms * 1e-3) * 1e-12, ms * 1e3);
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nblk = prop.multiProcessorCount;
    void* out; CK(hipMalloc(&out, 1l << 30));
    unsigned long long* dcyc; CK(hipMalloc(&dcyc, nblk * 8 * 8));
    // FEW workgroups (32 of 256 CUs busy, 8 tiles each): the per-CU store path without chip-wide bandwidth contention - what a CU's
    // epilogue sees when the other CUs are in their K loops
    run<0>("0: 32 rows x 32 B, 32 CUs only", out, 8192, 2048, dcyc, 32);
    run<1>("1: 16 rows x 64 B, 32 CUs only", out, 8192, 2048, dcyc, 32);
    run<2>("2:  8 rows x 128 B, 32 CUs only", out, 8192, 2048, dcyc, 32);
    run<3>("3:  4 rows x 256 B, 32 CUs only", out, 8192, 2048, dcyc, 32);
    run<4>("4:  2 rows x 512 B, 32 CUs only", out, 8192, 2048, dcyc, 32);
    // one tile per CU (a 32 MiB burst = the whole L2): does the write-back L2 absorb it?
    run<0>("0: 32 rows x 32 B, ONE tile per CU", out, 4096, 4096, dcyc, nblk);
    run<2>("2:  8 rows x 128 B, ONE tile per CU", out, 4096, 4096, dcyc, nblk);
    run<0>("0: 32 rows x 32 B, TWO tiles per CU", out, 8192, 4096, dcyc, nblk);
    run<2>("2:  8 rows x 128 B, TWO tiles per CU", out, 8192, 4096, dcyc, nblk);
    for (int N : {2560, 8192}) {
        const int M = N == 2560 ? 65536 : 8192;
        run<0>("0: 32 rows x 32 B per instruction", out, M, N, dcyc, nblk);
        run<1>("1: 16 rows x 64 B", out, M, N, dcyc, nblk);
        run<2>("2:  8 rows x 128 B", out, M, N, dcyc, nblk);
        run<3>("3:  4 rows x 256 B", out, M, N, dcyc, nblk);
        run<4>("4:  2 rows x 512 B", out, M, N, dcyc, nblk);
    }
    return 0;
}
