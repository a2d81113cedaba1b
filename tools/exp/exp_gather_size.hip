// Scratch: random 4-byte gathers (2^26 of them) over tables of different sizes: does a table that fits the 256 MiB Infinity Cache
// (or an XCD's 4 MiB L2) serve random lines faster than HBM does?  (The row sums' list references: a 196-293 MB table read 4.9 x 10^7
// times at random)  Not part of the product.
// build: hipcc -O3 --offload-arch=gfx950 exp_gather_size.hip -o exp_gather_size
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31; return x; }
template <class T>
__global__ __launch_bounds__(256) void k_gather(const T* __restrict__ src, uint64_t n_elems, uint64_t n_access, uint64_t salt, uint32_t* __restrict__ out) {
    uint32_t x = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_access; i += (uint64_t)gridDim.x * 256) x ^= (uint32_t)src[mix(i + salt) % n_elems];
    if (x == 0x12345678u) out[0] = x;
}
int main() {
    const uint64_t n_access = 1ull << 26;
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, 4ull << 30)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 1, 4ull << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t sizes_mb[] = {2, 16, 64, 128, 192, 256, 384, 512, 1024, 4096};
    for (uint64_t mb : sizes_mb) {
        for (int width = 4; width <= 2; width /= 2) {}
        const uint64_t bytes = mb << 20;
        for (int w = 0; w < 2; ++w) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (w == 0) hipLaunchKernelGGL(k_gather<uint32_t>, dim3(2048), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, n_access, (uint64_t)rep << 40, out);
                else hipLaunchKernelGGL(k_gather<uint16_t>, dim3(2048), dim3(256), 0, 0, (const uint16_t*)buf, bytes / 2, n_access, (uint64_t)rep << 40, out);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("table %5llu MB, %d-byte reads: %7.3f ms for 2^26 random reads = %6.1f G reads/s\n", (unsigned long long)mb, w == 0 ? 4 : 2, ms, n_access / ms / 1e6);
        }
    }
    return 0;
}
