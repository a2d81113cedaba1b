// Scratch ablation harness, round 2: where does k_dense_pair's time go after the masked-merge lookup?
// build: hipcc -O3 --offload-arch=gfx950 exp_dense2.hip -o exp_dense2      (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_fill(uint8_t* b, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        b[i] = "ACGT"[x & 3];
    }
}
__device__ __forceinline__ uint32_t pack4x2(uint32_t d) { return __builtin_amdgcn_udot4(d & 0x06060606u, 0x01041040u, 0u, false); }
__device__ __forceinline__ uint32_t pack16(uint4 v) {
    const uint32_t a = (pack4x2(v.x) << 8) | pack4x2(v.y), b = (pack4x2(v.z) << 8) | pack4x2(v.w);
    return (a << 15) | (b >> 1);
}
__device__ __forceinline__ uint32_t next_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 LD(const uint8_t* p) {   // non-temporal, as the product
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
static hipStream_t g_stream = 0;
constexpr int kWaves = 16, kTab = 65536;
// VAR 0 load+pack; 1 +halo; 2 +8 byte lookups, masked merge; 3 same, conflict-free addresses; 4 four byte lookups;
// 5 four ds_read_b32 lookups (word table, 14-bit address) + 4 masks; 6 as 5 with conflict-free addresses; 7 as 2 with 2 lookups
// packed-input variant of the same bodies: one dword (16 bases) per lane and row, four rows in flight
template <int VAR>
__global__ __launch_bounds__(1024) void k_varp(const uint32_t* __restrict__ b32, uint64_t n_rows, const uint8_t* __restrict__ gtab, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t tab[kTab];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    { const uint4* src = (const uint4*)gtab; uint4* dst = (uint4*)tab;
      for (uint32_t i = threadIdx.x; i < kTab / 16; i += 1024) dst[i] = src[i]; }
    __syncthreads();
    const uint32_t* wtab = (const uint32_t*)tab;
    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave, n_waves = (uint64_t)gridDim.x * kWaves;
    const uint64_t per = (n_rows + n_waves - 1) / n_waves;
    uint64_t row = gw * per;
    const uint64_t end = row + per < n_rows ? row + per : n_rows;
    const uint32_t* p = b32 + row * 63 + lane;
    uint32_t r[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) r[d] = row + d < end ? __builtin_nontemporal_load(p + d * 63) : 0u;
    uint32_t sink = 0;
    auto body = [&](uint32_t hi) {
        const uint32_t nxt = next_lane(hi);
        uint32_t ce = 0, co = 0;
        if (VAR == 2 || VAR == 3) {
            const uint32_t mid = __builtin_amdgcn_alignbit(hi, nxt, 16);
            uint32_t t[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t a0 = (hi >> (16 - 4 * q)) & 0xffffu, a1 = (mid >> (16 - 4 * q)) & 0xffffu;
                if (VAR == 3) { a0 = lane * 4 + q * 512; a1 = lane * 4 + 256 + q * 512; }
                t[q] = tab[a0]; t[q + 4] = tab[a1];
            }
            const uint32_t pe = ((t[0] << 8 | t[2]) << 16) | (t[4] << 8 | t[6]);
            const uint32_t po = ((t[1] << 8 | t[3]) << 16) | (t[5] << 8 | t[7]);
            const uint32_t se = __builtin_amdgcn_alignbit(hi, nxt, 22) & 0x03030303u;
            const uint32_t so = __builtin_amdgcn_alignbit(hi, nxt, 18) & 0x03030303u;
            ce = pe & __builtin_amdgcn_perm(0u, 0x88442211u, se);
            co = po & __builtin_amdgcn_perm(0u, 0x88442211u, so);
        } else if (VAR == 5 || VAR == 6) {
            const uint64_t W = ((uint64_t)hi << 32) | nxt;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t a = (uint32_t)(W >> (46 - 8 * q)) & 0x3fffu;
                if (VAR == 6) a = lane + q * 64;
                const uint32_t w = wtab[a];
                const uint32_t p2 = (uint32_t)(W >> (60 - 8 * q)) & 15u, s1 = (uint32_t)(W >> (44 - 8 * q)) & 3u;
                const uint32_t mk = (1u << p2) | (0x10000u << (p2 & 3u)) | (0x100000u << s1) | (0x1000000u << s1);
                ce |= (w & mk) ? (1u << q) : 0u;
            }
        } else { ce = hi ^ nxt; ce = ce == 0x12345u; }
        const unsigned long long bl = __ballot((ce | co) != 0) & 0x7fffffffffffffffull;
        if (bl) { if (lane == 0) sink += __popcll(bl); }
    };
    for (; row + 3 < end; row += 4, p += 4 * 63) {
#pragma unroll
        for (int d = 0; d < 4; ++d) { const uint32_t hi = r[d]; r[d] = __builtin_nontemporal_load(row + 4 + d < end ? p + (4 + d) * 63 : p); body(hi); }
    }
    if (sink == 0x12345678) out[0] = sink;
}
template <int VAR> void runp(const char* name, const uint32_t* b32, uint64_t n, const uint8_t* tab, uint32_t* out, int blocks) {
    const uint64_t n_rows = (n - 1024) / 1008;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, tot = 0;
    for (int it = 0; it < 8; ++it) {
        CK(hipEventRecord(a, g_stream));
        hipLaunchKernelGGL((k_varp<VAR>), dim3(blocks), dim3(1024), 0, g_stream, b32, n_rows, tab, out);
        CK(hipGetLastError()); CK(hipEventRecord(b, g_stream)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) { tot += ms; if (ms < best) best = ms; }
    }
    printf("PACKED %-34s blocks=%4d  avg %.4f ms  best %.4f ms  -> %.2e positions/s\n", name, blocks, tot / 7, best, n / best * 1e3);
}

template <int VAR, bool STRIDED>
__global__ __launch_bounds__(1024) void k_var(const uint8_t* __restrict__ bases, uint64_t n, const uint8_t* __restrict__ gtab,
                                             uint64_t n_rows, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t tab[kTab];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    { const uint4* src = (const uint4*)gtab; uint4* dst = (uint4*)tab;
      for (uint32_t i = threadIdx.x; i < kTab / 16; i += 1024) dst[i] = src[i]; }
    __syncthreads();
    const uint32_t* wtab = (const uint32_t*)tab;
    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave, n_waves = (uint64_t)gridDim.x * kWaves;
    const uint64_t per = (n_rows + n_waves - 1) / n_waves;
    const uint64_t stride = STRIDED ? n_waves * 1008 : 1008;
    uint64_t row = STRIDED ? gw : gw * per;
    const uint64_t step = STRIDED ? n_waves : 1;
    const uint64_t end = STRIDED ? n_rows : (row + per < n_rows ? row + per : n_rows);
    const uint8_t* ptr = bases + row * 1008 + (uint64_t)lane * 16;
    uint4 raw0 = make_uint4(0,0,0,0), raw1 = raw0;
    if (row < end) raw0 = LD(ptr);
    if (row + step < end) raw1 = LD(ptr + stride);
    uint32_t sink = 0;
    auto body = [&](uint4& raw, uint64_t r, const uint8_t* at) {
        const uint32_t hi = pack16(raw);
        raw = LD(r + 2 * step < end ? at + 2 * stride : at);
        if (VAR == 0) { sink ^= hi; return; }
        const uint32_t nxt = next_lane(hi);
        if (VAR == 1) { sink ^= hi ^ nxt; return; }
        const uint32_t mid = __builtin_amdgcn_alignbit(hi, nxt, 16);
        uint32_t ce = 0, co = 0;
        if (VAR == 2 || VAR == 3 || VAR == 4 || VAR == 7) {
            uint32_t t[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t a0 = (hi >> (16 - 4 * q)) & 0xffffu, a1 = (mid >> (16 - 4 * q)) & 0xffffu;
                if (VAR == 3) { a0 = lane * 4 + q * 512; a1 = lane * 4 + 256 + q * 512; }
                const bool skip = (VAR == 4 && (q & 1)) || (VAR == 7 && q != 0);
                t[q] = skip ? a0 : tab[a0]; t[q + 4] = skip ? a1 : tab[a1];
            }
            const uint32_t pe = ((t[0] << 8 | t[2]) << 16) | (t[4] << 8 | t[6]);
            const uint32_t po = ((t[1] << 8 | t[3]) << 16) | (t[5] << 8 | t[7]);
            const uint32_t se = __builtin_amdgcn_alignbit(hi, nxt, 22) & 0x03030303u;
            const uint32_t so = __builtin_amdgcn_alignbit(hi, nxt, 18) & 0x03030303u;
            ce = pe & __builtin_amdgcn_perm(0u, 0x88442211u, se);
            co = po & __builtin_amdgcn_perm(0u, 0x88442211u, so);
        } else {
            // word table: address = 7 bases (14 bits), one ds_read_b32 answers 4 positions; mask of 4 one-hot fields
            const uint64_t W = ((uint64_t)hi << 32) | nxt;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t a = (uint32_t)(W >> (46 - 8 * q)) & 0x3fffu;     // bases 4q+2 .. 4q+8
                if (VAR == 6) a = lane + q * 64;
                const uint32_t w = wtab[a];
                const uint32_t p2 = (uint32_t)(W >> (60 - 8 * q)) & 15u;  // two bases in front
                const uint32_t s1 = (uint32_t)(W >> (44 - 8 * q)) & 3u;   // base behind
                const uint32_t mk = (1u << p2) | (0x10000u << (p2 & 3u)) | (0x100000u << s1) | (0x1000000u << s1);
                ce |= (w & mk) ? (1u << q) : 0u;
            }
        }
        const unsigned long long bl = __ballot((ce | co) != 0) & 0x7fffffffffffffffull;
        if (bl) { if (lane == 0) sink += __popcll(bl); }
    };
    for (; row + step < end; row += 2 * step, ptr += 2 * stride) { body(raw0, row, ptr); body(raw1, row + step, ptr + stride); }
    if (row < end) body(raw0, row, ptr);
    if (sink == 0x12345678) out[0] = sink;
}

template <int VAR, bool STRIDED> void run(const char* name, const uint8_t* bases, uint64_t n, const uint8_t* tab, uint32_t* out, int blocks) {
    const uint64_t n_rows = (n - 1024) / 1008;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, tot = 0;
    for (int it = 0; it < 8; ++it) {
        CK(hipEventRecord(a, g_stream));
        hipLaunchKernelGGL((k_var<VAR, STRIDED>), dim3(blocks), dim3(1024), 0, g_stream, bases, n, tab, n_rows, out);
        CK(hipGetLastError()); CK(hipEventRecord(b, g_stream)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) { tot += ms; if (ms < best) best = ms; }
    }
    printf("%-34s %s blocks=%4d  avg %.4f ms  best %.4f ms  -> %.0f GB/s\n", name, STRIDED ? "strided" : "contig ", blocks, tot / 7, best, n / best / 1e6);
}

int main(int argc, char** argv) {
    const uint64_t n = 500000000ull;
    uint8_t *bases, *tab; uint32_t* out;
    CK(hipMalloc(&bases, n + 64)); CK(hipMalloc(&tab, kTab)); CK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, bases, n, 7ull);
    std::vector<uint8_t> h(kTab, 0);
    for (int i = 0; i < kTab; ++i) if ((i * 2654435761u >> 20) % 331 == 0) h[i] = 1u << (i & 7);   // ~0.3% of entries
    CK(hipMemcpy(tab, h.data(), kTab, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0, 64));
    if (argc > 1 && atoi(argv[1]) == 0) {   // "0": packed-input bodies on the whole chip
        const uint32_t* b32 = (const uint32_t*)bases;      // (the ASCII bytes read as packed words: random data either way)
        for (int blocks : {256, 512}) {
            runp<0>("load only", b32, n, tab, out, blocks);
            runp<2>("8 byte lookups, masked", b32, n, tab, out, blocks);
            runp<3>("same, conflict-free", b32, n, tab, out, blocks);
            runp<5>("four word lookups (4 pos each)", b32, n, tab, out, blocks);
            runp<6>("same, conflict-free", b32, n, tab, out, blocks);
        }
        return 0;
    }
    if (argc > 1) {   // argv[1] = CUs of a masked stream (first 256 - argv[1] CUs are left out)
        const int cus = atoi(argv[1]);
        uint32_t mask[8] = {0};
        for (int c = 256 - cus; c < 256; ++c) mask[c / 32] |= 1u << (c % 32);
        CK(hipExtStreamCreateWithCUMask(&g_stream, 8, mask));
        for (int blocks : {cus, 2 * cus}) {
            run<0, false>("0 load+pack", bases, n, tab, out, blocks);
            run<0, true>("0 load+pack", bases, n, tab, out, blocks);
            run<2, false>("2 +8 byte lookups, masked", bases, n, tab, out, blocks);
            run<2, true>("2 +8 byte lookups, masked", bases, n, tab, out, blocks);
            run<3, false>("3 same, conflict-free", bases, n, tab, out, blocks);
            run<4, false>("4 four byte lookups", bases, n, tab, out, blocks);
        }
        return 0;
    }
    for (int blocks : {256, 512}) {
        run<0, false>("0 load+pack", bases, n, tab, out, blocks);
        run<0, true>("0 load+pack", bases, n, tab, out, blocks);
        run<1, false>("1 +halo dpp", bases, n, tab, out, blocks);
        run<2, false>("2 +8 byte lookups, masked", bases, n, tab, out, blocks);
        run<2, true>("2 +8 byte lookups, masked", bases, n, tab, out, blocks);
        run<3, false>("3 same, conflict-free", bases, n, tab, out, blocks);
        run<4, false>("4 four byte lookups", bases, n, tab, out, blocks);
        run<7, false>("7 two byte lookups", bases, n, tab, out, blocks);
        run<5, false>("5 four word lookups (4 pos each)", bases, n, tab, out, blocks);
        run<6, false>("6 same, conflict-free", bases, n, tab, out, blocks);
    }
    return 0;
}
