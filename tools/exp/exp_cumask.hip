// Scratch: which CUs does a hipExtStreamCreateWithCUMask stream get?  (bit -> XCC / SE / CU mapping)  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_where(uint32_t* out) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the CU busy for a while so that the grid spreads over everything the stream may use
    uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
static void show(const char* name, hipStream_t s, uint32_t* d_out, int blocks) {
    std::vector<uint32_t> h(2 * blocks);
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(1024), 0, s, d_out);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<uint32_t, std::map<uint32_t, int>> per;   // xcc -> (se,sh,cu) -> count
    for (int b = 0; b < blocks; ++b) { const uint32_t hw = h[2 * b + 1]; per[h[2 * b] & 15][(hw >> 8) & 0xff]++; }
    printf("%s:", name);
    int total = 0;
    for (auto& x : per) { printf("  xcc%u:%zu CUs", x.first, x.second.size()); total += (int)x.second.size(); }
    printf("  = %d CUs\n", total);
}
int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 8 * 4096));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("multiProcessorCount %d\n", p.multiProcessorCount);
    hipStream_t s0; CK(hipStreamCreate(&s0));
    show("unmasked", s0, d_out, 2048);
    for (int variant = 0; variant < 4; ++variant) {
        std::vector<uint32_t> mask(8, 0);
        const char* name;
        if (variant == 0) { name = "bits 0-15"; mask[0] = 0xffff; }
        else if (variant == 1) { name = "bits 16-255"; for (int i = 16; i < 256; ++i) mask[i / 32] |= 1u << (i % 32); }
        else if (variant == 2) { name = "bits 0-31"; mask[0] = 0xffffffffu; }
        else { name = "bits {32j, 32j+1}"; for (int j = 0; j < 8; ++j) mask[j] = 3u; }
        hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, mask.data()));
        show(name, s, d_out, 2048);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
