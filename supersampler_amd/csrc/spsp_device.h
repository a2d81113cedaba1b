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

// 16 ASCII bases -> 32 bits, first base in the two most significant bits.
// code = (c >> 1) & 3 (reference utils.cpp:13-16: A=0 C=1 T=2 G=3).
// Four bases per dword with ONE dot instruction: bytes & 0x06 hold 2*code, and v_dot4_u32_u8 with byte
// weights 64,16,4,1 (first base = lowest byte = most significant field) sums them into 2 * (b0<<6|b1<<4|b2<<2|b3).
// The factor 2 is carried through the shift-or merges and dropped by the last shift, so 16 bases cost
// 4 and + 4 dot4 + 4 merges (a 32-bit integer multiply, the previous form, is a quarter-rate instruction).
__device__ __forceinline__ uint32_t pack4x2(uint32_t d) {
    return __builtin_amdgcn_udot4(d & 0x06060606u, 0x01041040u, 0u, false);
}
__device__ __forceinline__ uint32_t pack16(uint4 v) {
    const uint32_t a = (pack4x2(v.x) << 8) | pack4x2(v.y);   // 2 * (first 8 bases)
    const uint32_t b = (pack4x2(v.z) << 8) | pack4x2(v.w);   // 2 * (last 8 bases)
    return (a << 15) | (b >> 1);
}

}  // namespace spsp
