// Stand-alone probe of the ROCm 7.2 crash the product works around by parking retired hipGraphs (insv2v/inference.py, _GRAVEYARD):
// capture a chain of kernel launches -> instantiate -> launch -> DESTROY graph + exec -> capture the SAME chain again -> launch.
// No torch, no product library.  Modes (argv[1]): 0 = one stream; 1 = fork / join over three streams inside the capture (the 3 CFG-branch
// streams of the single-clip graph); 2 = mode 1 with the buffers freed and re-allocated between the two captures (what a caching
// allocator's private graph pool does when the graph that owned it dies).  build: hipcc --offload-arch=gfx950 -O2 graph_recapture.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void axpy(float* y, const float* x, float a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = a * x[i] + y[i]; }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, rounds = argc > 2 ? atoi(argv[2]) : 4, n = 1 << 22;
    float *x, *y[3];
    hipStream_t s[3]; hipEvent_t fork, join[3];
    for (int i = 0; i < 3; ++i) { CK(hipStreamCreate(&s[i])); CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming)); }
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipMalloc(&x, n * 4)); CK(hipMemset(x, 0, n * 4));
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&y[i], n * 4)); CK(hipMemset(y[i], 0, n * 4)); }
    for (int r = 0; r < rounds; ++r) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        if (mode >= 1) { CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0)); CK(hipStreamWaitEvent(s[2], fork, 0)); }
        for (int k = 0; k < 200; ++k)
            for (int b = 0; b < (mode >= 1 ? 3 : 1); ++b) hipLaunchKernelGGL(axpy, dim3(n / 256), dim3(256), 0, s[b], y[b], x, 1.0f + k, n);
        if (mode >= 1) for (int b = 1; b < 3; ++b) { CK(hipEventRecord(join[b], s[b])); CK(hipStreamWaitEvent(s[0], join[b], 0)); }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int it = 0; it < 3; ++it) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));      // the step the product avoids
        if (mode == 2) for (int i = 0; i < 3; ++i) { CK(hipFree(y[i])); CK(hipMalloc(&y[i], n * 4)); CK(hipMemset(y[i], 0, n * 4)); }
        printf("round %d: capture -> launch x3 -> destroy ok\n", r); fflush(stdout);
    }
    printf("mode %d: %d destroy / recapture rounds survived\n", mode, rounds);
    return 0;
}
