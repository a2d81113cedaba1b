// Scratch: what a fresh process pays before its first kernel has run (the floor under the CLI's wall time).
// build: hipcc -O2 --offload-arch=gfx950 exp_hipstart.hip -o exp_hipstart
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int* p) { if (p) p[0] = 1; }
int main() {
    double t0 = now(), t;
    hipInit(0);                                     t = now(); printf("hipInit            %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipSetDevice(0); hipFree(nullptr);              t = now(); printf("device + context   %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); t = now(); printf("stream             %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    int* d; hipMalloc(&d, 1 << 20);                 t = now(); printf("first hipMalloc    %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s); t = now(); printf("first kernel       %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    void* h; hipHostMalloc(&h, 32 << 20, hipHostMallocDefault); t = now(); printf("pinned 32 MB       %7.1f ms\n", (t - t0) * 1e3); t0 = t;
    hipHostFree(h); hipFree(d);
    return 0;
}
