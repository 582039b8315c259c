// tr_probe: what does ds_read_b64_tr_b16 return?  LDS holds halfs whose value is their own index; every lane passes a byte address from
// a simple pattern and the four halfs each lane receives are printed (as LDS half indices).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tr_probe.hip -o instruct-video-to-video_amd/build/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void probe(unsigned short* out, int pattern) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* lds = (unsigned short*)smem;
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned addr;
    if (pattern == 0) addr = lane * 8;                                            // 64 consecutive 8-byte granules
    else if (pattern == 1) addr = (lane & 15) * 8 + (lane >> 4) * 2048;           // each 16-lane group its own region
    else if (pattern == 2) addr = ((lane & 15) >> 2) * 128 + (lane & 3) * 8 + (lane >> 4) * 512;  // 4 rows x 128 B stride, 4 granules per row
    else addr = ((lane & 15) & 3) * 128 + ((lane & 15) >> 2) * 8 + (lane >> 4) * 512;             // lane -> (row = l & 3, granule = l >> 2)
    addr += (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned long long r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(r >> (16 * j));
}

int main() {
    CK(hipSetDevice(0));
    unsigned short* d; CK(hipMalloc(&d, 64 * 4 * 2));
    std::vector<unsigned short> h(256);
    for (int pat = 0; pat < 4; ++pat) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, d, pat);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
        printf("pattern %d (values = half index in LDS; lane address in halfs shown first)\n", pat);
        for (int l = 0; l < 64; ++l) {
            unsigned a;
            if (pat == 0) a = l * 8; else if (pat == 1) a = (l & 15) * 8 + (l >> 4) * 2048;
            else if (pat == 2) a = ((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 512; else a = ((l & 15) & 3) * 128 + ((l & 15) >> 2) * 8 + (l >> 4) * 512;
            printf("  lane %2d addr %5u : %5u %5u %5u %5u\n", l, a / 2, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        }
    }
    return 0;
}
