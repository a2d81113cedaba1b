// Device-side primitives shared by the scan and compare kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace spsp {

// XXH64 of one little-endian 64-bit word, seed 1312 -- the reference's
// Subsampler::unrevhash (SubSampler.cpp:64-67 -> include/xxhash64.h:100-150).
// Closed form for an 8-byte input: no stripe loop, one "remaining 8 bytes"
// round and the avalanche.  gfx950 has no 64x64 multiply; each `*` below is a
// few v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32.
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

__device__ __forceinline__ uint64_t xxh64_u64(uint64_t x) {
    constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL,
                       P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL,
                       P5 = 2870177450012600261ULL;
    uint64_t h = 1312ULL + P5 + 8ULL;
    h ^= rotl64(x * P2, 31) * P1;
    h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

// Reverse complement of a full 64-bit window of 32 bases (first base in the top
// two bits; A=0 C=1 T=2 G=3 so complement = code ^ 2): reverse the order of the
// 2-bit groups, then flip the high bit of each group.
__device__ __forceinline__ uint64_t rc_window64(uint64_t w) {
    uint64_t r = __brevll(w);
    r = ((r & 0x5555555555555555ULL) << 1) | ((r >> 1) & 0x5555555555555555ULL);
    return r ^ 0xAAAAAAAAAAAAAAAAULL;
}
__device__ __forceinline__ uint32_t rc_window32(uint32_t w) {
    uint32_t r = __brev(w);
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    return r ^ 0xAAAAAAAAu;
}
// reverse complement of an m-mer held in the low 2m bits (reference rcbc, utils.cpp:449-462)
__device__ __forceinline__ uint32_t rc_mmer32(uint32_t v, uint32_t m) { return rc_window32(v) >> (32 - 2 * m); }

}  // namespace spsp
