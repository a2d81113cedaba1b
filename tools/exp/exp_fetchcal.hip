// Scratch: what FETCH_SIZE / WRITE_SIZE report per byte for the access shapes of the comparison's kernels.  Not part
// of the product.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streams (16 B per lane: the
// counter shows half the bytes) and says "other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern".  Every kernel below moves a KNOWN number of bytes over a buffer far larger than the 256 MiB
// Infinity Cache; run it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and divide.
//   k_stream<W>     coalesced stream, W = 4 / 8 / 16 bytes per lane (k_parts_scatter reads 8 + 4, k_accumulate's `where` 4)
//   k_gather<W>     one W-byte read per lane at a random W-aligned address (list references: 4, inline lists / lists: 16)
//   k_scatter16     one 16-byte store per lane at a random 16-aligned address (k_parts_scatter's records)
// build: hipcc -O3 --offload-arch=gfx950 exp_fetchcal.hip -o exp_fetchcal ; usage: exp_fetchcal [n_access = 2^26]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
    return x;
}
template <class T> __device__ __forceinline__ uint32_t fold(T v);
template <> __device__ __forceinline__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint64_t>(uint64_t v) { return (uint32_t)v ^ (uint32_t)(v >> 32); }
template <> __device__ __forceinline__ uint32_t fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <class T>
__global__ __launch_bounds__(256) void k_stream(const T* __restrict__ src, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t x = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) x ^= fold<T>(src[i]);
    if (x == 0x12345678u) out[0] = x;
}
template <class T>
__global__ __launch_bounds__(256) void k_gather(const T* __restrict__ src, uint64_t n_elems, uint64_t n_access, uint32_t* __restrict__ out) {
    uint32_t x = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_access; i += (uint64_t)gridDim.x * 256) x ^= fold<T>(src[mix(i) % n_elems]);
    if (x == 0x12345678u) out[0] = x;
}
__global__ __launch_bounds__(256) void k_scatter16(uint4* __restrict__ dst, uint64_t n_elems, uint64_t n_access) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_access; i += (uint64_t)gridDim.x * 256)
        dst[mix(i) % n_elems] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void k_stream_store16(uint4* __restrict__ dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main(int argc, char** argv) {
    const uint64_t n_access = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 26);
    const uint64_t bytes = 4ull << 30;                    // 4 GiB: 16 x the Infinity Cache
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    const dim3 grid(256 * 8), block(256);
    printf("n_access %llu; expected bytes: stream4 %llu stream8 %llu stream16 %llu gather4 %llu gather8 %llu gather16 %llu scatter16 %llu store16 %llu\n",
           (unsigned long long)n_access, (unsigned long long)(n_access * 4), (unsigned long long)(n_access * 8), (unsigned long long)(n_access * 16),
           (unsigned long long)(n_access * 4), (unsigned long long)(n_access * 8), (unsigned long long)(n_access * 16), (unsigned long long)(n_access * 16),
           (unsigned long long)(n_access * 16));
    // wall time per kernel (HIP events): a gather cannot have moved more bytes than its time allows at the measured
    // ~6.5 TB/s read rate -- which settles whether a random access costs a 64-byte sector or a 128-byte line
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timed = [&](const char* name, auto launch, double bytes_useful) {
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("time %-12s %8.3f ms  useful %7.1f GB/s  if 64 B/access %7.1f GB/s  if 128 B/access %7.1f GB/s\n", name, ms, bytes_useful / ms / 1e6,
                   64.0 * n_access / ms / 1e6, 128.0 * n_access / ms / 1e6);
        };
        timed("stream16", [&] { hipLaunchKernelGGL(k_stream<uint4>, grid, block, 0, 0, (const uint4*)buf, n_access, out); }, 16.0 * n_access);
        timed("gather4", [&] { hipLaunchKernelGGL(k_gather<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, n_access, out); }, 4.0 * n_access);
        timed("gather16", [&] { hipLaunchKernelGGL(k_gather<uint4>, grid, block, 0, 0, (const uint4*)buf, bytes / 16, n_access, out); }, 16.0 * n_access);
        timed("scatter16", [&] { hipLaunchKernelGGL(k_scatter16, grid, block, 0, 0, (uint4*)buf, bytes / 16, n_access); }, 16.0 * n_access);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, n_access, out);
        hipLaunchKernelGGL(k_stream<uint64_t>, grid, block, 0, 0, (const uint64_t*)buf, n_access, out);
        hipLaunchKernelGGL(k_stream<uint4>, grid, block, 0, 0, (const uint4*)buf, n_access, out);
        hipLaunchKernelGGL(k_gather<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, n_access, out);
        hipLaunchKernelGGL(k_gather<uint64_t>, grid, block, 0, 0, (const uint64_t*)buf, bytes / 8, n_access, out);
        hipLaunchKernelGGL(k_gather<uint4>, grid, block, 0, 0, (const uint4*)buf, bytes / 16, n_access, out);
        hipLaunchKernelGGL(k_scatter16, grid, block, 0, 0, (uint4*)buf, bytes / 16, n_access);
        hipLaunchKernelGGL(k_stream_store16, grid, block, 0, 0, (uint4*)buf, n_access);
        CK(hipDeviceSynchronize());
    }
    printf("done\n");
    return 0;
}
