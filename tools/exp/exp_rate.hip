// VALU issue-rate probe for a few integer instructions on gfx950 (one wave per SIMD, long unrolled chains).
// build: hipcc -O3 --offload-arch=gfx950 -o exp_rate exp_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * (i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] * a[i] + seed;                              // v_mul_lo_u32 (+add)
                if (OP == 1) a[i] = __builtin_amdgcn_udot4(a[i], 0x01041040u, a[i], false);
                if (OP == 2) a[i] = (a[i] << 10) | (a[i] ^ seed);                    // shift-or class
                if (OP == 3) a[i] = __builtin_amdgcn_perm(a[i], seed, 0x07030602u) + 1u;
                if (OP == 4) a[i] = __builtin_amdgcn_ubfe(a[i], 3, 17) + seed;
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name) {
    uint32_t* d;
    hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 12345u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(2048), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 2048.0 * 4 * iters * 16 * 8;   // wave-instructions of the probed kind
    printf("%-12s %8.3f ms  %6.2f wave-instr / ns chip-wide  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms,
           ops / (ms * 1e6), (ms * 1e6 * 2.4) / (ops / 1024.0));
    hipFree(d);
}

int main() {
    run<0>("mul_lo+add"); run<1>("dot4"); run<2>("shl/xor/or"); run<3>("perm+add"); run<4>("bfe+add");
    return 0;
}
