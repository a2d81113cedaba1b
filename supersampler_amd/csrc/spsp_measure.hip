// spsp_measure.hip -- HBM calibration for the roofline ("calibrate with a device copy kernel on the
// box and use the measured figure as denominator").  Two plain streaming kernels over buffers far larger than the
// 256 MiB Infinity Cache: a copy (read + write, the usual yardstick: MI355X_MICROARCH.md quotes 6.29 TB/s for a
// float4 copy) and a read-only pass (the dense scan pass is read-only: a read stream does not pay for the bus
// turning around, so its ceiling is the fairer denominator for that kernel).  Same load shape as the product's dense
// pass: one non-temporal 16-byte load per lane, four in flight, rows strided over the grid.
#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

typedef uint32_t u32x4_m __attribute__((ext_vector_type(4)));
constexpr int kMeasThreads = 1024, kMeasUnroll = 4;

__global__ __launch_bounds__(kMeasThreads) void k_measure_copy(const u32x4_m* __restrict__ src, u32x4_m* __restrict__ dst, uint64_t n_vec) {
    const uint64_t stride = (uint64_t)gridDim.x * kMeasThreads;
    uint64_t i = (uint64_t)blockIdx.x * kMeasThreads + threadIdx.x;
    for (; i + (kMeasUnroll - 1) * stride < n_vec; i += kMeasUnroll * stride) {
        u32x4_m v[kMeasUnroll];
#pragma unroll
        for (int u = 0; u < kMeasUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kMeasUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n_vec; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// every word is folded into a checksum the kernel has to deliver (one word per workgroup), so no load can be dropped
__global__ __launch_bounds__(kMeasThreads) void k_measure_read(const u32x4_m* __restrict__ src, uint64_t n_vec, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_x[kMeasThreads / 64];
    const uint64_t stride = (uint64_t)gridDim.x * kMeasThreads;
    uint64_t i = (uint64_t)blockIdx.x * kMeasThreads + threadIdx.x;
    uint32_t x = 0;
    for (; i + (kMeasUnroll - 1) * stride < n_vec; i += kMeasUnroll * stride) {
        u32x4_m v[kMeasUnroll];
#pragma unroll
        for (int u = 0; u < kMeasUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < kMeasUnroll; ++u) x ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n_vec; i += stride) { const u32x4_m v = __builtin_nontemporal_load(src + i); x ^= v.x ^ v.y ^ v.z ^ v.w; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x ^= __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0) s_x[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t y = 0;
        for (int w = 0; w < kMeasThreads / 64; ++w) y ^= s_x[w];
        sums[blockIdx.x] = y;
    }
}

}  // namespace spsp

using namespace spsp;

extern "C" int spsp_measure_hbm_device(spsp_ctx* ctx, uint64_t bytes, uint32_t reps, spsp_hbm_rates* out) {
    if (!ctx || !out || reps == 0) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    bytes &= ~((uint64_t)4095);
    if (bytes < (1ull << 20)) { set_error("calibration buffer too small"); return SPSP_ERR_ARG; }
    void *a = nullptr, *b = nullptr;
    uint32_t* sums = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const uint32_t grid = (ctx->n_cu ? ctx->n_cu : 256) * 2;       // two 1024-lane workgroups per CU, like the dense pass on its own CUs
    int rc = SPSP_OK;
    auto done = [&](int r) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (sums) (void)hipFree(sums);
        return r;
    };
#define MEAS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = hip_fail(e_, #call, __FILE__, __LINE__); return done(rc); } } while (0)
    MEAS_HIP(hipMalloc(&a, bytes));
    MEAS_HIP(hipMalloc(&b, bytes));
    MEAS_HIP(hipMalloc((void**)&sums, (size_t)grid * 4));
    MEAS_HIP(hipMemsetAsync(a, 0x5a, bytes, ctx->stream));
    MEAS_HIP(hipMemsetAsync(b, 0, bytes, ctx->stream));
    MEAS_HIP(hipEventCreate(&e0));
    MEAS_HIP(hipEventCreate(&e1));
    const uint64_t n_vec = bytes / 16;
    float ms = 0;
    // copy
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_measure_copy, dim3(grid), dim3(kMeasThreads), 0, ctx->stream, (const u32x4_m*)a, (u32x4_m*)b, n_vec);
    MEAS_HIP(hipEventRecord(e0, ctx->stream));
    for (uint32_t r = 0; r < reps; ++r) hipLaunchKernelGGL(k_measure_copy, dim3(grid), dim3(kMeasThreads), 0, ctx->stream, (const u32x4_m*)a, (u32x4_m*)b, n_vec);
    MEAS_HIP(hipEventRecord(e1, ctx->stream));
    MEAS_HIP(hipEventSynchronize(e1));
    MEAS_HIP(hipEventElapsedTime(&ms, e0, e1));
    out->copy_ms = ms / reps;
    out->copy_GBps = 2.0 * (double)bytes / 1e9 / (out->copy_ms / 1e3);
    // read only: alternate between the two buffers so that no launch re-reads what the one before it left in a cache
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_measure_read, dim3(grid), dim3(kMeasThreads), 0, ctx->stream, (const u32x4_m*)(w ? b : a), n_vec, sums);
    MEAS_HIP(hipEventRecord(e0, ctx->stream));
    for (uint32_t r = 0; r < reps; ++r) hipLaunchKernelGGL(k_measure_read, dim3(grid), dim3(kMeasThreads), 0, ctx->stream, (const u32x4_m*)((r & 1) ? b : a), n_vec, sums);
    MEAS_HIP(hipEventRecord(e1, ctx->stream));
    MEAS_HIP(hipEventSynchronize(e1));
    MEAS_HIP(hipEventElapsedTime(&ms, e0, e1));
    MEAS_HIP(hipGetLastError());
    out->read_ms = ms / reps;
    out->read_GBps = (double)bytes / 1e9 / (out->read_ms / 1e3);
    out->bytes = bytes;
    out->reps = reps;
    out->n_cu = ctx->n_cu ? ctx->n_cu : 256;
#undef MEAS_HIP
    return done(SPSP_OK);
}
