// Scratch: read-pattern ablation for the dense pass (load + pack only).  Not part of the product.
// build: hipcc -O3 --offload-arch=gfx950 exp_load.hip -o exp_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 LDNT(const uint8_t* p) { const u32x4 v = __builtin_nontemporal_load((const u32x4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint32_t pack4x2(uint32_t d) { return __builtin_amdgcn_udot4(d & 0x06060606u, 0x01041040u, 0u, false); }
__device__ __forceinline__ uint32_t pack16(uint4 v) {
    const uint32_t a = (pack4x2(v.x) << 8) | pack4x2(v.y), b = (pack4x2(v.z) << 8) | pack4x2(v.w);
    return (a << 15) | (b >> 1);
}
// RB row bytes (1008 / 1024); MODE 0 contiguous range per wave, 1 rows strided over all waves, 2 contiguous range per
// WORKGROUP with its waves interleaved row by row; DEPTH loads in flight per wave
template <int RB, int MODE, int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS) void k_load(const uint8_t* __restrict__ bases, uint64_t n_rows, uint32_t* __restrict__ out) {
    constexpr int kWaves = THREADS / 64;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave, n_waves = (uint64_t)gridDim.x * kWaves;
    uint64_t row, step, end;
    if (MODE == 0) { const uint64_t per = (n_rows + n_waves - 1) / n_waves; row = gw * per; step = 1; end = row + per < n_rows ? row + per : n_rows; }
    else if (MODE == 1) { row = gw; step = n_waves; end = n_rows; }
    else if (MODE == 2) { const uint64_t per = (n_rows + gridDim.x - 1) / gridDim.x; row = blockIdx.x * per + wave; step = kWaves; end = blockIdx.x * per + per < n_rows ? blockIdx.x * per + per : n_rows; }
    else {   // MODE 3: PAIRS of adjacent rows strided over the grid (depth 2 = the two rows of a pair)
        uint32_t sink3 = 0;
        const uint64_t n_pairs = n_rows / 2;
        const uint8_t* p = bases + gw * 2 * RB + (uint64_t)lane * 16;
        const uint64_t st = n_waves * 2 * RB;
        uint4 r0 = make_uint4(0,0,0,0), r1 = r0;
        if (gw < n_pairs) { r0 = LDNT(p); r1 = LDNT(p + RB); }
        for (uint64_t j = gw; j < n_pairs; j += n_waves, p += st) {
            const uint8_t* nx = j + n_waves < n_pairs ? p + st : p;
            sink3 ^= pack16(r0); r0 = LDNT(nx);
            sink3 ^= pack16(r1); r1 = LDNT(nx + RB);
        }
        if (sink3 == 0x12345678) out[0] = sink3;
        return;
    }
    const uint64_t stride = step * RB;
    const uint8_t* ptr = bases + row * RB + (uint64_t)lane * 16;
    uint4 raw[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) raw[d] = (row + d * step < end) ? LDNT(ptr + d * stride) : make_uint4(0, 0, 0, 0);
    uint32_t sink = 0;
    for (; row < end; row += DEPTH * step, ptr += DEPTH * stride) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            sink ^= pack16(raw[d]);
            raw[d] = LDNT(row + (d + DEPTH) * step < end ? ptr + (d + DEPTH) * stride : ptr);
        }
    }
    if (sink == 0x12345678) out[0] = sink;
}
template <int RB, int MODE, int DEPTH, int THREADS> void run(const uint8_t* bases, uint64_t n, uint32_t* out, int blocks) {
    const uint64_t n_rows = (n - 1024) / RB;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, tot = 0;
    for (int it = 0; it < 8; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_load<RB, MODE, DEPTH, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, bases, n_rows, out);
        CK(hipGetLastError()); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) { tot += ms; if (ms < best) best = ms; }
    }
    printf("row %4d B  %-22s depth %d  threads %4d blocks %5d  avg %.4f ms  best %.4f ms  -> %.0f GB/s\n", RB,
           MODE == 0 ? "contiguous per wave" : MODE == 1 ? "strided over grid" : MODE == 2 ? "contiguous per group" : "pairs strided over grid", DEPTH, THREADS, blocks, tot / 7, best, n / best / 1e6);
}
int main() {
    const uint64_t n = 500000000ull;
    uint8_t* bases; uint32_t* out;
    CK(hipMalloc(&bases, n + 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(bases, 0x41, n + 4096)); CK(hipMemset(out, 0, 64));
    for (int blocks : {256, 384, 512}) {
        run<1008, 3, 2, 1024>(bases, n, out, blocks);
        run<1008, 1, 2, 1024>(bases, n, out, blocks);
        run<1008, 0, 2, 1024>(bases, n, out, blocks);
        run<1008, 3, 2, 1024>(bases, n, out, blocks);
        run<1008, 1, 2, 1024>(bases, n, out, blocks);
        run<1008, 0, 2, 1024>(bases, n, out, blocks);
    }
    for (int blocks : {256}) {
        run<1008, 0, 2, 1024>(bases, n, out, blocks);
        run<1024, 0, 2, 1024>(bases, n, out, blocks);
        run<1008, 1, 2, 1024>(bases, n, out, blocks);
        run<1024, 1, 2, 1024>(bases, n, out, blocks);
        run<1024, 2, 2, 1024>(bases, n, out, blocks);
        run<1024, 0, 4, 1024>(bases, n, out, blocks);
        run<1024, 1, 4, 1024>(bases, n, out, blocks);
        run<1024, 2, 4, 1024>(bases, n, out, blocks);
        run<1024, 1, 8, 1024>(bases, n, out, blocks);
    }
    for (int blocks : {1024, 2048, 4096}) {
        run<1024, 1, 2, 256>(bases, n, out, blocks);
        run<1024, 1, 4, 256>(bases, n, out, blocks);
        run<1024, 2, 4, 256>(bases, n, out, blocks);
    }
    return 0;
}
