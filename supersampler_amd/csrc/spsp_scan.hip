// spsp_scan.hip -- path A on gfx950: the minimizer / FracMinHash scan.
//
// What the reference does serially per base (SubSampler.cpp:357-455: rolling
// m-mers, XXH64 of the canonical m-mer, running minimum with a full rescan of
// the k-mer whenever the minimizer leaves the window, super-k-mer emission) is
// re-formulated for the GPU as a dense pass plus a sparse pass:
//
//   dense  (k_dense_*)   every m-mer position: 2-bit pack -> canonical m-mer
//                        -> XXH64 -> `hash <= T` -> one bit in a hit bitmap.
//                        1 byte read and 1/8 byte written per position.
//   sparse (k_expand,    a k-mer is selected iff ANY m-mer of its window has
//           k_resolve)   hash <= T (min <= T  <=>  exists <= T), so only the
//                        hits matter.  Hits closer than w = k-m+1 form a
//                        cluster; a cluster always starts with the reference's
//                        "new m-mer beats the minimum" event (SubSampler.cpp:374)
//                        which resets its whole state, so clusters are
//                        independent and one lane replays the reference's state
//                        machine literally over the cluster's hits -- including
//                        its position/strand quirks (SubSampler.cpp:89-93,
//                        132-166) that split super-k-mers.
//
// Output = exactly the argument stream of Subsampler::handle_superkmer.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

// ---------------------------------------------------------------- geometry --
constexpr int kThreads = 256;                // 4 waves
constexpr int kChunk = 16;                   // bases per lane per row (one dwordx4 load)
constexpr int kRows = 4;                     // rows per tile
constexpr int kRowPos = kThreads * kChunk;   // 4096 positions per row
constexpr int kTilePos = kRows * kRowPos;    // 16384 positions per workgroup
constexpr int kTileWords = kTilePos / 32;    // 512 bitmap words per tile

__device__ __forceinline__ uint32_t load_pack(const uint8_t* __restrict__ bases, uint64_t n, uint64_t pos) {
    if (pos + kChunk <= n) {
        return pack16(*reinterpret_cast<const uint4*>(bases + pos));
    }
    uint32_t w = 0;
    for (int j = 0; j < kChunk; ++j) {
        uint32_t c = (pos + j < n) ? bases[pos + j] : 0u;
        w = (w << 2) | ((c >> 1) & 3u);
    }
    return w;
}

// One 16-byte chunk of a row, as a NON-TEMPORAL load (global_load_dwordx4 ... nt): the dense pass reads every byte
// exactly once, and without the hint its stream churns through L2 -- measured on the pair-table pass, 5 x 10^8
// positions on the whole chip: 0.096-0.099 ms with plain loads, 0.085-0.089 ms with nt (5.8 TB/s).
typedef uint32_t spsp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_row16(const uint8_t* __restrict__ p) {
    const spsp_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const spsp_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ASCII <-> packed input (SPSP_SCAN_PACKED_INPUT): thread t makes / expands dword t = bases 16t .. 16t+15
__global__ __launch_bounds__(256) void k_pack_bases(const uint8_t* __restrict__ bases, uint64_t n, uint32_t* __restrict__ packed, uint64_t n_dw_padded) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_dw_padded; t += (uint64_t)gridDim.x * blockDim.x)
        packed[t] = t * kChunk < n ? load_pack(bases, n, t * kChunk) : 0u;
}
__global__ __launch_bounds__(256) void k_unpack_bases(const uint32_t* __restrict__ packed, uint64_t n, uint8_t* __restrict__ bases) {
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t * kChunk < n; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = packed[t];
        for (uint32_t j = 0; j < (uint32_t)kChunk && t * kChunk + j < n; ++j) bases[t * kChunk + j] = "ACTG"[(w >> (30 - 2 * j)) & 3u];
    }
}

// ------------------------------------------------- dense pass, direct form ---
// XXH64 at every position: the variant for dense selections (small -s), where
// nearly every lane has to hash anyway.  A workgroup stages one tile of kTilePos
// positions as 2-bit words in LDS (16-base halo from the next tile) and the grid
// strides over the tiles.
__global__ __launch_bounds__(kThreads) void k_dense_direct(const uint8_t* __restrict__ bases, uint64_t n, uint32_t m,
                                                          uint64_t thr, uint64_t n_tiles,
                                                          uint32_t* __restrict__ bitmap,
                                                          uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t packed[kRows * kThreads + 1];
    __shared__ uint32_t s_cnt;
    const uint32_t t = threadIdx.x;
    const uint64_t n_mmers = n >= m ? n - m + 1 : 0;
    const uint32_t mm = (1u << (2 * m)) - 1u;  // m <= 15
    const uint32_t sh0 = 64 - 2 * m;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t tile_base = tile * kTilePos;
        __syncthreads();  // previous iteration's readers of packed[] / s_cnt are done
        if (t == 0) s_cnt = 0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const uint32_t c = r * kThreads + t;
            packed[c] = load_pack(bases, n, tile_base + (uint64_t)c * kChunk);
        }
        if (t == 0) packed[kRows * kThreads] = load_pack(bases, n, tile_base + (uint64_t)kTilePos);
        __syncthreads();
        uint32_t local = 0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const uint32_t c = r * kThreads + t;
            const uint64_t p0 = tile_base + (uint64_t)c * kChunk;
            const uint64_t W = ((uint64_t)packed[c] << 32) | packed[c + 1];
            const uint64_t R = rc_window64(W);
            uint32_t mask16 = 0;
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                const uint32_t f = (uint32_t)(W >> (sh0 - 2 * j)) & mm;
                const uint32_t rc = (uint32_t)(R >> (2 * j)) & mm;
                const uint32_t x = f < rc ? f : rc;
                mask16 |= (xxh64_u64(x) <= thr ? 1u : 0u) << j;
            }
            // positions past the last m-mer of the buffer never count
            if (p0 + kChunk > n_mmers) {
                const uint32_t keep = p0 >= n_mmers ? 0u : (uint32_t)(n_mmers - p0);
                mask16 &= (keep >= 16) ? 0xffffu : ((1u << keep) - 1u);
            }
            const uint32_t other = __shfl_down(mask16, 1);
            if ((t & 1u) == 0) bitmap[p0 >> 5] = mask16 | (other << 16);
            local += __popc(mask16);
        }
        if (local) atomicAdd(&s_cnt, local);
        __syncthreads();
        if (t == 0) tile_count[tile] = s_cnt;
    }
}

// --------------------------------------------- dense pass, pair-table form ---
// The selected set is tiny next to the m-mer universe (default k31/m11/s1000:
// ~100 canonical 11-mers out of 2^21), so the question "can the m-mer starting
// here be selected at all?" is memoised in LDS and XXH64 only runs on the rare
// survivors.  One lookup answers it for TWO adjacent positions:
//
//   key tables    K9[q]   q = first 9 bases (18 bits) of an m-mer, either strand: set iff some m-mer x with that
//                         prefix has XXH64(canon(x)) <= T;  M9[q] the same for bases 1..9 of x (m >= 10),
//                         K8[q] for its first 8 bases (m = 9)
//   pair table    P[e]    e = 9 consecutive bases (18 bits) starting at an even offset of the lane's chunk -> 2 bits
//                         bit0 = K9[e]: the position AT e;  bit1 = M9[e]: the position IN FRONT of e, whose bases
//                         1..9 are e (m = 9: K8[last 8 bases of e], the position behind)
//                         (64 KiB in LDS, 4 entries per byte)
//
// A lane owns 16 consecutive positions; its 32-base register window gives the
// 9-base pair keys with one bit-field extract each (eight, nine from m = 10 on: the
// last position's entry starts in the next lane's chunk), so the hot loop is
// ~3 VALU + 1 ds_read_u8 per TWO positions, with no branch and no hash (pair_lookup16).
// Survivors (<1 % of positions when this variant is chosen) go to a per-wave
// queue in LDS and are hashed 64 at a time with every lane busy.  Waves never
// synchronise with each other: the 16-base halo comes from the neighbouring
// lane by one DPP move (lane 63 is halo only), verified hits go to the wave's own list.
constexpr int kPairWaves = 16;              // waves per workgroup (1024 lanes)
constexpr int kPairTabBytes = 65536;        // 2^18 entries x 2 bits
constexpr int kQueueCap = 64;               // survivor slots per wave (16 bytes each): 64 KiB table + 16 KiB queues = 80 KiB, two workgroups per CU

__global__ void k_build_key8(uint32_t m, uint64_t thr, uint32_t* __restrict__ key8, uint32_t* __restrict__ key9,
                             uint32_t* __restrict__ mid9) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1ull << (2 * m))) return;
    const uint32_t v = (uint32_t)x;
    const uint32_t rc = rc_mmer32(v, m);
    if (rc < v) return;                       // visit each canonical m-mer once
    if (xxh64_u64(v) > thr) return;
    const uint32_t sh = 2 * m - 16;           // m >= 8: first 8 bases
    const uint32_t a = v >> sh, b = rc >> sh; // either strand may appear in the genome
    atomicOr(&key8[a >> 5], 1u << (a & 31));
    atomicOr(&key8[b >> 5], 1u << (b & 31));
    if (key9) {                               // m >= 9: first 9 bases, for the first position of a pair
        const uint32_t a9 = v >> (sh - 2), b9 = rc >> (sh - 2);
        atomicOr(&key9[a9 >> 5], 1u << (a9 & 31));
        atomicOr(&key9[b9 >> 5], 1u << (b9 & 31));
    }
    if (mid9) {                               // m >= 10: bases 1..9 of the m-mer, for the position IN FRONT of a pair
        const uint32_t am = (v >> (2 * m - 20)) & 0x3ffffu, bm = (rc >> (2 * m - 20)) & 0x3ffffu;
        atomicOr(&mid9[am >> 5], 1u << (am & 31));
        atomicOr(&mid9[bm >> 5], 1u << (bm & 31));
    }
}

__global__ void k_build_pairtab(const uint32_t* __restrict__ key8, const uint32_t* __restrict__ key9,
                                const uint32_t* __restrict__ mid9, uint8_t* __restrict__ tab) {
    const uint32_t byte = blockIdx.x * blockDim.x + threadIdx.x;
    if (byte >= (uint32_t)kPairTabBytes) return;
    uint32_t v = 0;
    for (uint32_t s2 = 0; s2 < 4; ++s2) {
        const uint32_t e = byte * 4 + s2;     // 9-base window, first base most significant
        const uint32_t q0 = e >> 2, q1 = e & 0xffffu;
        // the entry's 9 bases ARE the first position's 9-base prefix: test it with K9 (4x fewer false survivors
        // than K8, same byte); the second position only has 8 of its bases inside the entry
        (void)q0;
        // second bit: m = 9 -- the position behind (8 of its bases are in the entry); m >= 10 -- the position IN FRONT,
        // whose bases 1..9 are this entry: a nine-base test, four times sharper (k_dense_pair<true>)
        const uint32_t r0 = (key9[e >> 5] >> (e & 31)) & 1u;
        const uint32_t r1 = mid9 ? (mid9[e >> 5] >> (e & 31)) & 1u : (key8[q1 >> 5] >> (q1 & 31)) & 1u;
        v |= (r0 << s2) | (r1 << (s2 + 4));   // sub-entry s2: bit s2 = first position, bit s2+4 = second
    }
    tab[byte] = (uint8_t)v;
}

// ------------------------------------------------- per-wave hit lists --------
// The table variants of the dense pass give every wave a CONTIGUOUS range of rows and its own slice of a
// raw hit array: survivors are verified in position order (the queue is first in, first out and a drain round
// keeps lane order), so a wave's slice is sorted and wave w's hits all precede wave w+1's.  No bitmap, no
// atomics, no memset: k_compact only has to prefix the per-wave counts.  A wave that finds more hits than
// its slice holds keeps counting (stores are suppressed); the host grows the slices and runs the pass again.
struct WaveLists {
    Hit* raw;            // [n_waves][cap]
    uint32_t* cnt;       // [n_waves] hits found (may exceed cap)
    uint32_t cap;
    uint64_t rows_per_wave;
};

// one drain round: lane i holds candidate i (or none); real hits are appended in lane order
__device__ __forceinline__ uint32_t append_hits(bool is_hit, uint64_t pos, uint32_t canon, uint32_t f, uint64_t hash,
                                                Hit* __restrict__ out, uint32_t out_cap, uint32_t out_n) {
    const unsigned long long hitmask = __ballot(is_hit);
    if (!hitmask) return out_n;
    const uint32_t at = out_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(hitmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hitmask, 0u));
    if (is_hit && at < out_cap) {
        Hit h;
        h.pos = pos; h.hash = hash; h.canon = canon; h.rec = 0;
        h.flags = canon != f ? 1u : 0u;      // rec / usable are filled in by k_compact
        h.pad = 0;
        out[at] = h;
    }
    return out_n + (uint32_t)__popcll(hitmask);
}

// A wave-row is 63 chunks of 16 positions: lane 63 only supplies the halo of lane
// 62 (its chunk is lane 0 of the next row), so every lane runs the same code and
// no lane needs a second load.
constexpr int kRowChunks = 63;
constexpr int kRowPosPair63 = kRowChunks * kChunk;   // 1008 positions per wave-row

// The next lane's packed chunk (lane 63: 0 -- it is the halo lane): one DPP move on the VALU instead of a
// ds_bpermute, which would take a slot of the LDS pipeline the table reads saturate.
__device__ __forceinline__ uint32_t next_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}

// Survivors of one lane's 16 positions from its packed window (hi = own 16 bases, nxt = next 16).
// Pair q (positions 2q, 2q+1) reads the table byte of its first 8 bases; the sub-entry of that byte is chosen by
// the pair's 9th base s_q: bit s_q = first position, bit s_q + 4 = second.  The eight bytes are NOT shifted one by
// one: the bytes of the even pairs (q = 0,2,4,6) are merged into one word, those of the odd pairs into another,
// and each word is ANDed with a mask whose byte for pair q is 0x11 << s_q.  The s_q of the even pairs are 8 bits
// apart in the window, so one v_alignbit brings them to bits 1:0 of the four bytes and ONE v_perm_b32 with the
// constant 0x88442211 as byte table turns them into the mask.  ~20 VALU per 16 positions beside the address
// extraction (the shifted form took 36).  Which position a surviving bit stands for is worked out when the
// survivor is verified (pair_code_to_offset), not here.
// MID (m >= 10): the second bit of an entry stands for the position in front of the pair (see k_build_pairtab), so
// position 2q - 1 is answered by pair q: pair 0's second bit belongs to the lane in front and is masked out here, and
// position 15 needs a ninth lookup, "pair 8" = bases 16..24 of the window (PairSurv::x).
struct PairSurv { uint32_t e, o, x; };
template <bool MID>
__device__ __forceinline__ PairSurv pair_lookup16(const uint8_t* __restrict__ tab, uint32_t hi, uint32_t nxt) {
    const uint32_t mid = __builtin_amdgcn_alignbit(hi, nxt, 16);   // bases 8..23
    uint32_t t[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                    // pairs 0..3 from hi, 4..7 from mid
        t[q] = tab[(hi >> (16 - 4 * q)) & 0xffffu];
        t[q + 4] = tab[(mid >> (16 - 4 * q)) & 0xffffu];
    }
    const uint32_t t8 = MID ? (uint32_t)tab[nxt >> 16] : 0u;                 // bases 16..23
    const uint32_t pe = ((t[0] << 8 | t[2]) << 16) | (t[4] << 8 | t[6]);    // byte 3 - q/2 = pair q (q even)
    const uint32_t po = ((t[1] << 8 | t[3]) << 16) | (t[5] << 8 | t[7]);    // byte 3 - q/2 = pair q (q odd)
    // base 2q+8 of the 64-bit window hi:nxt sits at bits 47-4q : 46-4q
    uint32_t se = __builtin_amdgcn_alignbit(hi, nxt, 22) & 0x03030303u;
    const uint32_t so = __builtin_amdgcn_alignbit(hi, nxt, 18) & 0x03030303u;
    if (MID) se |= 0x04000000u;                      // pair 0 (byte 3): selectors 4..7 = the first-bit-only masks
    PairSurv r;
    r.e = pe & __builtin_amdgcn_perm(0x08040201u, 0x88442211u, se);
    r.o = po & __builtin_amdgcn_perm(0u, 0x88442211u, so);
    r.x = MID ? t8 & (0x10u << ((nxt >> 14) & 3u)) : 0u;                    // base 24 picks the sub-entry; second bit only
    return r;
}
// survivor code = bit index in PairSurv::e (0..31), 32 + bit index in PairSurv::o, 64 + bit index in PairSurv::x
// -> position offset (0..15)
template <bool MID>
__device__ __forceinline__ uint32_t pair_code_to_offset(uint32_t code) {
    const uint32_t second = (code >> 2) & 1u;
    if (code >= 64u) return 15u;
    const uint32_t q = 2 * (3 - ((code >> 3) & 3u)) + (code >> 5);
    return MID ? 2 * q - second : 2 * q + second;
}
__device__ __forceinline__ uint32_t pair_count(PairSurv c) { return __popc(c.e) + __popc(c.o) + __popc(c.x); }
// code of the only survivor of a lane that holds exactly one
__device__ __forceinline__ uint32_t pair_single_code(PairSurv c) {
    return (uint32_t)(__ffs(c.e | c.o | c.x) - 1) | (c.o ? 32u : 0u) | (c.x ? 64u : 0u);
}

// Verify the queued survivors of one wave (all of them: the queue never holds more than 64), lane i taking
// entry i.  Entries are {position of the lane's chunk relative to the wave's first position, survivor bit
// (or 0x100 | offset), window hi, window nxt}: offset and m-mer are decoded here, with all lanes busy, not in
// the push.  Deliberately NOT inlined: the hash is ~150 instructions and the scan kernel reaches this
// from several places -- inlined copies blow the instruction cache of the hot loop.
template <bool MID>
__device__ __attribute__((noinline)) uint32_t drain_pair(const uint4* __restrict__ queue, uint32_t qn, uint64_t base_pos,
                                                         uint64_t n_mmers, uint32_t m, uint64_t thr,
                                                         Hit* __restrict__ out, uint32_t out_cap, uint32_t out_n) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t mm = (1u << (2 * m)) - 1u;
    bool is_hit = false;
    uint64_t pos = 0, hash = 0;
    uint32_t f = 0, x = 0;
    if (lane < qn) {
        const uint4 e = queue[lane];
        const uint32_t off = (e.y & 0x100u) ? (e.y & 15u) : pair_code_to_offset<MID>(e.y);
        const uint64_t W = ((uint64_t)e.z << 32) | e.w;
        pos = base_pos + e.x + off;
        f = (uint32_t)(W >> (64 - 2 * m - 2 * off)) & mm;
        const uint32_t rc = rc_mmer32(f, m);
        x = f < rc ? f : rc;
        hash = xxh64_u64(x);
        is_hit = pos < n_mmers && hash <= thr;
    }
    return append_hits(is_hit, pos, x, f, hash, out, out_cap, out_n);
}

// PACKED: `bases` holds 2-bit codes, 16 bases per little-endian dword with the first base in bits 31:30 -- exactly the
// word pack16 makes of 16 ASCII bytes (k_pack_bases; SPSP_SCAN_PACKED_INPUT).  A lane's chunk is ONE dword, a wave-row
// 252 bytes: a quarter of the traffic, no packing arithmetic, four rows in flight per wave.  On this byte model
// (0.25 B per position) the pass is bound by its LDS lookups, not by HBM.
template <bool MID, bool PACKED>
__global__ __launch_bounds__(64 * kPairWaves) void k_dense_pair(const uint8_t* __restrict__ bases, uint64_t n, uint32_t m,
                                                               uint64_t thr, const uint8_t* __restrict__ pairtab,
                                                               uint64_t n_rows, WaveLists L) {
    __shared__ __attribute__((aligned(16))) uint8_t tab[kPairTabBytes];
    extern __shared__ __attribute__((aligned(16))) uint32_t qbase[];   // per wave: kQueueCap x {rel pos, bit, hi, nxt}
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row bookkeeping stays on the SALU
    uint4* queue = reinterpret_cast<uint4*>(qbase) + wave * kQueueCap;
    {   // table -> LDS, 64 bytes per lane
        const uint4* src = reinterpret_cast<const uint4*>(pairtab);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t i = threadIdx.x; i < kPairTabBytes / 16; i += 64 * kPairWaves) dst[i] = src[i];
    }
    __syncthreads();
    const uint64_t n_mmers = n >= m ? n - m + 1 : 0;
    const uint64_t gw = (uint64_t)blockIdx.x * kPairWaves + wave;
    // rows of this wave: [first, first + n_my), a contiguous range (measured: strided rows, the streaming order,
    // are 3 % faster in the dense kernel alone -- and would need a merge of the lists afterwards)
    const uint64_t first = gw * L.rows_per_wave;
    const uint64_t lim = first + L.rows_per_wave < n_rows ? first + L.rows_per_wave : n_rows;
    const uint64_t n_my = first < lim ? lim - first : 0;
    const uint64_t base_pos = first * kRowPosPair63;
    constexpr uint64_t row_bytes = kRowPosPair63;
    Hit* out = L.raw + gw * L.cap;
    uint32_t out_n = 0;       // hits of this wave so far (wave-uniform)
    uint32_t qn = 0;          // survivors waiting in this wave's queue (wave-uniform, lives in an SGPR)

    auto drain = [&]() {
        out_n = drain_pair<MID>(queue, qn, base_pos, n_mmers, m, thr, out, L.cap, out_n);
        qn = 0;
    };
    // General form for ONE row: any number of survivors per lane, queued in position order.  Usual case (a lane or
    // two holding two survivors: one row pair in five at the default sampling) -- they are appended behind what is
    // queued and hashed with the next full round; only more than a queue-full is hashed 64 at a time on the spot.
    auto push_row = [&](PairSurv c, uint32_t rel, uint32_t hi, uint32_t nxt) {
        if (!__ballot((c.e | c.o | c.x) != 0)) return;
        uint32_t pm = c.x ? 0x8000u : 0u;                 // bit j = position offset j survives
        for (uint32_t t = c.e; t; t &= t - 1) pm |= 1u << pair_code_to_offset<MID>(__ffs(t) - 1);
        for (uint32_t t = c.o; t; t &= t - 1) pm |= 1u << pair_code_to_offset<MID>(32 + __ffs(t) - 1);
        const uint32_t cnt = __popc(pm);
        uint32_t idx = 0, total = 0;                      // exclusive prefix / total of cnt over the lanes
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const unsigned long long plane = __ballot((cnt >> b) & 1u);
            idx += __builtin_amdgcn_mbcnt_hi((uint32_t)(plane >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)plane, 0u)) << b;
            total += (uint32_t)__popcll(plane) << b;
        }
        if (qn + total > (uint32_t)kQueueCap) drain();
        if (total <= (uint32_t)kQueueCap) {
            uint32_t at = qn + idx;
            while (pm) {
                const uint32_t off = __ffs(pm) - 1;
                pm &= pm - 1;
                queue[at++] = make_uint4(rel, 0x100u | off, hi, nxt);
            }
            qn += total;
            if (qn >= (uint32_t)kQueueCap) drain();
            return;
        }
        for (uint32_t base = 0; base < total; base += 64) {   // adversarial input: repeats
            while (pm && idx < base + 64) {
                const uint32_t off = __ffs(pm) - 1;
                pm &= pm - 1;
                queue[idx - base] = make_uint4(rel, 0x100u | off, hi, nxt);
                ++idx;
            }
            qn = total - base < 64 ? total - base : 64;
            drain();
        }
    };
    // Queue the survivors of TWO consecutive rows of this wave: ca belongs to the row whose lane chunk starts at
    // rel_a (relative to base_pos) with window (hia,nxa), cb to the next row.  Usual case: no lane holds two
    // survivors of one row -- one ballot per row gives the queue places, row a first (position order).
    // (lane 63 only feeds lane 62: its own lookups are dropped from the ballots, on the scalar unit)
    constexpr unsigned long long kRowLanes = (1ull << kRowChunks) - 1;
    auto handle = [&](PairSurv ca, PairSurv cb, uint32_t rel_a, uint32_t hia, uint32_t nxa, uint32_t rel_b, uint32_t hib, uint32_t nxb) {
        const unsigned long long ha = __ballot((ca.e | ca.o | ca.x) != 0) & kRowLanes, hb = __ballot((cb.e | cb.o | cb.x) != 0) & kRowLanes;
        if (!(ha | hb)) return;
        const bool mine_a = (ha >> lane) & 1u, mine_b = (hb >> lane) & 1u;
        const uint32_t na = (uint32_t)__popcll(ha), total = na + (uint32_t)__popcll(hb);
        if (!(__ballot(pair_count(ca) > 1 || pair_count(cb) > 1) & kRowLanes) && total <= (uint32_t)kQueueCap) {
            if (qn + total > (uint32_t)kQueueCap) drain();
            if (mine_a) queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(ha >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ha, 0u))] =
                            make_uint4(rel_a, pair_single_code(ca), hia, nxa);
            if (mine_b) queue[qn + na + __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u))] =
                            make_uint4(rel_b, pair_single_code(cb), hib, nxb);
            qn += total;
            if (qn >= (uint32_t)kQueueCap) drain();
            return;
        }
        const PairSurv none = {0u, 0u, 0u};
        push_row(lane < (uint32_t)kRowChunks ? ca : none, rel_a, hia, nxa);
        push_row(lane < (uint32_t)kRowChunks ? cb : none, rel_b, hib, nxb);
    };

    if (PACKED) {
        const uint32_t* b32 = reinterpret_cast<const uint32_t*>(bases);
        const uint64_t n_dw = (n + kChunk - 1) / kChunk;                        // dwords that hold bases (the tail of the last one is 0)
        const uint64_t full_p = n_dw >= 64 ? (n_dw - 64) / kRowChunks + 1 : 0;  // rows whose 64 dwords (halo included) exist
        const uint64_t nf_all = first < full_p ? full_p - first : 0;
        const uint64_t nf = nf_all < n_my ? nf_all : n_my;
        const uint32_t* p = b32 + first * kRowChunks + lane;
        constexpr uint32_t rs = kRowChunks;                                     // dwords per row
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        if (0 < nf) r0 = __builtin_nontemporal_load(p);
        if (1 < nf) r1 = __builtin_nontemporal_load(p + rs);
        if (2 < nf) r2 = __builtin_nontemporal_load(p + 2 * rs);
        if (3 < nf) r3 = __builtin_nontemporal_load(p + 3 * rs);
        uint64_t i = 0;
        uint32_t rel = lane * kChunk;
        const uint32_t rel_step = (uint32_t)kRowPosPair63;
        for (; i + 3 < nf; i += 4, p += 4 * rs, rel += 4 * rel_step) {
            // unconditional refills (see the ASCII form): past its last row a wave re-reads its current one
            const uint32_t hia = r0; r0 = __builtin_nontemporal_load(i + 4 < nf ? p + 4 * rs : p);
            const uint32_t nxa = next_lane(hia);
            const PairSurv ca = pair_lookup16<MID>(tab, hia, nxa);
            const uint32_t hib = r1; r1 = __builtin_nontemporal_load(i + 5 < nf ? p + 5 * rs : p);
            const uint32_t nxb = next_lane(hib);
            const PairSurv cb = pair_lookup16<MID>(tab, hib, nxb);
            handle(ca, cb, rel, hia, nxa, rel + rel_step, hib, nxb);
            const uint32_t hic = r2; r2 = __builtin_nontemporal_load(i + 6 < nf ? p + 6 * rs : p);
            const uint32_t nxc = next_lane(hic);
            const PairSurv cc = pair_lookup16<MID>(tab, hic, nxc);
            const uint32_t hid = r3; r3 = __builtin_nontemporal_load(i + 7 < nf ? p + 7 * rs : p);
            const uint32_t nxd = next_lane(hid);
            const PairSurv cd = pair_lookup16<MID>(tab, hid, nxd);
            handle(cc, cd, rel + 2 * rel_step, hic, nxc, rel + 3 * rel_step, hid, nxd);
        }
        const PairSurv none = {0u, 0u, 0u};
        for (; i < n_my; ++i, rel += rel_step) {        // up to three rows left, and the rows at the end of the buffer
            const uint64_t at = (first + i) * kRowChunks + lane;
            const uint32_t hi = at < n_dw ? b32[at] : 0u;
            const uint32_t nxt = next_lane(hi);
            handle(pair_lookup16<MID>(tab, hi, nxt), none, rel, hi, nxt, rel, hi, nxt);
        }
        drain();
        if (lane == 0) L.cnt[gw] = out_n;
        return;
    }
    // rows whose 64 chunks lie completely inside the buffer take the vector path
    const uint64_t full_rows = n >= 64 * kChunk ? (n - 64 * kChunk) / kRowPosPair63 + 1 : 0;
    const uint64_t n_fast_all = first < full_rows ? full_rows - first : 0;
    const uint64_t n_fast = n_fast_all < n_my ? n_fast_all : n_my;
    uint64_t i = 0;                                            // index among this wave's rows
    const uint8_t* ptr = bases + first * kRowPosPair63 + (uint64_t)lane * kChunk;
    // Two rows in flight per wave, in two named register sets: each set is refilled
    // right after it has been packed, so its wait sits a whole loop trip later.
    uint4 raw0 = make_uint4(0, 0, 0, 0), raw1 = raw0;
    if (i < n_fast) raw0 = load_row16(ptr);
    if (i + 1 < n_fast) raw1 = load_row16(ptr + row_bytes);
    auto body = [&](uint4& raw, uint64_t r, const uint8_t* at, uint32_t& hi, uint32_t& nxt) -> PairSurv {
        hi = pack16(raw);
        // unconditional refill (a branch around the load would force a full vmcnt(0) wait right here):
        // past the last row the wave re-reads its current row, whose value is never used
        raw = load_row16(r + 2 < n_fast ? at + 2 * row_bytes : at);
        nxt = next_lane(hi);
        return pair_lookup16<MID>(tab, hi, nxt);
    };
    uint32_t rel = lane * kChunk;                              // this lane's chunk of the current row, relative to base_pos
    const uint32_t rel_step = (uint32_t)row_bytes;
    for (; i + 1 < n_fast; i += 2, ptr += 2 * row_bytes, rel += 2 * rel_step) {
        uint32_t hia, nxa, hib, nxb;
        const PairSurv ca = body(raw0, i, ptr, hia, nxa);
        const PairSurv cb = body(raw1, i + 1, ptr + row_bytes, hib, nxb);
        handle(ca, cb, rel, hia, nxa, rel + rel_step, hib, nxb);
    }
    const PairSurv no_row = {0u, 0u, 0u};
    if (i < n_fast) {
        uint32_t hia, nxa;
        const PairSurv ca = body(raw0, i, ptr, hia, nxa);
        handle(ca, no_row, rel, hia, nxa, rel, hia, nxa);
        ++i; rel += rel_step;
    }
    // the last (at most two) rows of the buffer touch its end: byte-wise loads
    for (; i < n_my; ++i, rel += rel_step) {
        const uint32_t hi = load_pack(bases, n, (first + i) * kRowPosPair63 + (uint64_t)lane * kChunk);
        const uint32_t nxt = next_lane(hi);
        handle(pair_lookup16<MID>(tab, hi, nxt), no_row, rel, hi, nxt, rel, hi, nxt);
    }
    drain();
    if (lane == 0) L.cnt[gw] = out_n;
}

// ------------------------------------- dense pass, single-position table form ---
// Same structure as k_dense_pair with a sharper, one-position table for the
// configurations whose 8-base pair keys are too crowded (m = 13, 15, or dense
// sampling): K10[q] = some m-mer (either strand) whose first 10 bases are q has
// XXH64(canonical) <= T; 2^20 bits = 128 KiB of LDS, one ds_read_u8 per position.
// Survivor rates up to ~30 % go through a first-in-first-out ring per wave, so every
// hash round runs with 64 busy lanes and the verified hits stay in position order.
constexpr int kKey10Bytes = 131072;
constexpr int kQueueCap1 = 192;              // 16 waves x 192 x 8 B + table = 152 KiB
constexpr int kPushWindow = 128;             // survivors queued between two drains: < 64 left + 128 <= kQueueCap1

__global__ void k_build_key10(uint32_t m, uint64_t thr, uint32_t* __restrict__ key10) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1ull << (2 * m))) return;
    const uint32_t v = (uint32_t)x;
    const uint32_t rc = rc_mmer32(v, m);
    if (rc < v) return;
    if (xxh64_u64(v) > thr) return;
    const uint32_t sh = 2 * m - 20;           // m >= 10: first 10 bases
    const uint32_t a = v >> sh, b = rc >> sh;
    atomicOr(&key10[a >> 5], 1u << (a & 31));
    atomicOr(&key10[b >> 5], 1u << (b & 31));
}

__device__ __forceinline__ uint32_t single_lookup16(const uint8_t* __restrict__ tab, uint32_t hi, uint32_t nxt) {
    const uint32_t w1 = (hi << 12) | (nxt >> 20);   // bases 6..21
    const uint32_t w2 = (hi << 24) | (nxt >> 8);    // bases 12..27
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t src = j < 6 ? hi : (j < 12 ? w1 : w2);
        const int jj = j < 6 ? j : (j < 12 ? j - 6 : j - 12);
        const uint32_t addr = (src >> (15 - 2 * jj)) & 0x1ffffu;   // 10-base key >> 3
        const uint32_t bit = (src >> (12 - 2 * jj)) & 7u;
        acc = __builtin_amdgcn_alignbit((uint32_t)tab[addr] >> bit, acc, 1);
    }
    return acc >> 16;   // bit j = position j survives
}

__global__ __launch_bounds__(64 * kPairWaves) void k_dense_single(const uint8_t* __restrict__ bases, uint64_t n, uint32_t m,
                                                                 uint64_t thr, const uint8_t* __restrict__ key10,
                                                                 uint64_t n_rows, WaveLists L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds1[];
    uint8_t* tab = lds1;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint2* queue = reinterpret_cast<uint2*>(lds1 + kKey10Bytes) + wave * kQueueCap1;   // ring of {rel pos, m-mer}
    {
        const uint4* src = reinterpret_cast<const uint4*>(key10);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t i = threadIdx.x; i < kKey10Bytes / 16; i += 64 * kPairWaves) dst[i] = src[i];
    }
    __syncthreads();
    const uint64_t n_mmers = n >= m ? n - m + 1 : 0;
    const uint32_t mm = (1u << (2 * m)) - 1u;
    const uint64_t gw = (uint64_t)blockIdx.x * kPairWaves + wave;
    const uint64_t first = gw * L.rows_per_wave;              // see k_dense_pair
    const uint64_t lim = first + L.rows_per_wave < n_rows ? first + L.rows_per_wave : n_rows;
    const uint64_t n_my = first < lim ? lim - first : 0;
    const uint64_t base_pos = first * kRowPosPair63;
    constexpr uint64_t row_bytes = kRowPosPair63;
    Hit* out = L.raw + gw * L.cap;
    uint32_t out_n = 0, qn = 0, head = 0;
    // inlined on purpose (unlike k_dense_pair's): at survivor rates of 5-20 % a row drains several times, and a
    // real call would spill the caller's live registers every time
    auto drain = [&](uint32_t keep_below) {
        while (qn >= keep_below && qn > 0) {
            const uint32_t take = qn < 64 ? qn : 64;
            bool is_hit = false;
            uint64_t pos = 0, hash = 0;
            uint32_t f = 0, x = 0;
            if (lane < take) {
                uint32_t at = head + lane;
                if (at >= (uint32_t)kQueueCap1) at -= kQueueCap1;
                const uint2 e = queue[at];
                pos = base_pos + e.x; f = e.y;
                const uint32_t rc = rc_mmer32(f, m);
                x = f < rc ? f : rc;
                hash = xxh64_u64(x);
                is_hit = pos < n_mmers && hash <= thr;
            }
            out_n = append_hits(is_hit, pos, x, f, hash, out, L.cap, out_n);
            head += take; if (head >= (uint32_t)kQueueCap1) head -= kQueueCap1;
            qn -= take;
        }
    };
    // queue one row's survivors in position order (lane-major), kPushWindow at a time
    auto handle = [&](uint32_t cand, uint32_t rel, uint32_t hi, uint32_t nxt) {
        if (!__ballot(cand != 0)) return;
        const uint32_t cnt = __popc(cand);
        uint32_t idx = 0, total = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const unsigned long long plane = __ballot((cnt >> b) & 1u);
            idx += __builtin_amdgcn_mbcnt_hi((uint32_t)(plane >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)plane, 0u)) << b;
            total += (uint32_t)__popcll(plane) << b;
        }
        const uint64_t W = ((uint64_t)hi << 32) | nxt;
        uint32_t c = cand;
        for (uint32_t base = 0; base < total; base += kPushWindow) {
            const uint32_t tail = head + qn;              // ring slot of survivor `base` (before wrapping)
            while (c && idx < base + kPushWindow) {
                const uint32_t j = __ffs(c) - 1;
                c &= c - 1;
                uint32_t at = tail + (idx - base);
                if (at >= (uint32_t)kQueueCap1) at -= kQueueCap1;
                if (at >= (uint32_t)kQueueCap1) at -= kQueueCap1;
                queue[at] = make_uint2(rel + j, (uint32_t)(W >> (64 - 2 * m - 2 * j)) & mm);
                ++idx;
            }
            qn += total - base < (uint32_t)kPushWindow ? total - base : (uint32_t)kPushWindow;
            drain(64);
        }
    };
    const bool halo_lane = lane >= kRowChunks;
    uint32_t rel = lane * kChunk;
    const uint32_t rel_step = (uint32_t)row_bytes;
    const uint64_t full_rows = n >= 64 * kChunk ? (n - 64 * kChunk) / kRowPosPair63 + 1 : 0;
    const uint64_t n_fast_all = first < full_rows ? full_rows - first : 0;
    const uint64_t n_fast = n_fast_all < n_my ? n_fast_all : n_my;
    uint64_t i = 0;
    const uint8_t* ptr = bases + first * kRowPosPair63 + (uint64_t)lane * kChunk;
    uint4 raw0 = make_uint4(0, 0, 0, 0), raw1 = raw0;
    if (i < n_fast) raw0 = load_row16(ptr);
    if (i + 1 < n_fast) raw1 = load_row16(ptr + row_bytes);
    auto body = [&](uint4& raw, uint64_t r, const uint8_t* at, uint32_t rl) {
        const uint32_t hi = pack16(raw);
        raw = load_row16(r + 2 < n_fast ? at + 2 * row_bytes : at);
        const uint32_t nxt = next_lane(hi);
        const uint32_t c = single_lookup16(tab, hi, nxt);
        handle(halo_lane ? 0u : c, rl, hi, nxt);
    };
    for (; i + 1 < n_fast; i += 2, ptr += 2 * row_bytes, rel += 2 * rel_step) {
        body(raw0, i, ptr, rel);
        body(raw1, i + 1, ptr + row_bytes, rel + rel_step);
    }
    if (i < n_fast) { body(raw0, i, ptr, rel); ++i; rel += rel_step; }
    for (; i < n_my; ++i, rel += rel_step) {
        const uint32_t hi = load_pack(bases, n, (first + i) * kRowPosPair63 + (uint64_t)lane * kChunk);
        const uint32_t nxt = next_lane(hi);
        const uint32_t c = single_lookup16(tab, hi, nxt);
        handle(halo_lane ? 0u : c, rel, hi, nxt);
    }
    drain(1);
    if (lane == 0) L.cnt[gw] = out_n;
}

// ------------------------------------------------- exclusive scan (1 block) --
// out[i] = sum(in[0..i)), out[n] = total.  u32, single workgroup of 1024 lanes;
// the arrays it runs over (tiles, hits) are tiny next to the dense pass.
__global__ __launch_bounds__(1024) void k_exclusive_scan(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                        uint64_t n_max, const uint32_t* __restrict__ n_dev,
                                                        uint64_t* __restrict__ total_host, uint32_t* __restrict__ total_dev) {
    constexpr int E = 8;                       // consecutive elements per lane per round
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry_s;
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    // n may only be known on the device (a count produced earlier on this stream)
    uint64_t n = n_max;
    if (n_dev) { const uint64_t v = *n_dev; n = v < n_max ? v : n_max; }
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += 1024 * E) {
        const uint64_t i0 = base + (uint64_t)t * E;
        uint32_t v[E];
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < E; ++u) { v[u] = i0 + u < n ? in[i0 + u] : 0u; sum += v[u]; }
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) wave_sum[wid] = x;
        __syncthreads();
        uint32_t pre = 0, all = 0;
        for (uint32_t w = 0; w < 16; ++w) { if (w < wid) pre += wave_sum[w]; all += wave_sum[w]; }
        const uint32_t carry = carry_s;
        uint32_t run = carry + pre + x - sum;
#pragma unroll
        for (int u = 0; u < E; ++u) { if (i0 + u < n) out[i0 + u] = run; run += v[u]; }
        __syncthreads();
        if (t == 0) carry_s = carry + all;
        __syncthreads();
    }
    if (t == 0) {
        const uint32_t total = carry_s;
        out[n] = total;
        if (total_host) *total_host = total;
        if (total_dev) *total_dev = total;
    }
}

// Two-level form for the tile-count scan on the scan pipeline's critical path: every workgroup scans its own
// segment (a power of two >= 1 Ki elements, chosen so that there are at most 64 segments) and leaves the
// segment sum; consumers add the few preceding segment sums themselves.
constexpr int kSegThreads = 256, kSegChunk = 4 * kSegThreads;
static inline uint32_t seg_shift_for(uint64_t n) {
    uint32_t sh = 10;
    while (((n + (1ull << sh) - 1) >> sh) > 64) ++sh;
    return sh;
}
__global__ __launch_bounds__(kSegThreads) void k_scan_segments(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                              uint64_t n, uint32_t seg_shift,
                                                              uint32_t* __restrict__ seg_sum) {
    __shared__ uint32_t wave_sum[kSegThreads / 64];
    __shared__ uint32_t s_carry;
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint64_t seg0 = (uint64_t)blockIdx.x << seg_shift;
    const uint64_t seg1 = seg0 + (1ull << seg_shift) < n ? seg0 + (1ull << seg_shift) : n;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t c0 = seg0; c0 < seg1; c0 += kSegChunk) {
        const uint64_t i0 = c0 + (uint64_t)t * 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i0 + 4 <= seg1) v = *reinterpret_cast<const uint4*>(in + i0);   // segments start on 4 KiB boundaries
        else {
            if (i0 < seg1) v.x = in[i0];
            if (i0 + 1 < seg1) v.y = in[i0 + 1];
            if (i0 + 2 < seg1) v.z = in[i0 + 2];
        }
        const uint32_t sum = v.x + v.y + v.z + v.w;
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) wave_sum[wid] = x;
        __syncthreads();
        uint32_t pre = s_carry, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kSegThreads / 64; ++w) { if (w < wid) pre += wave_sum[w]; all += wave_sum[w]; }
        const uint32_t e0 = pre + x - sum;
        const uint4 o = make_uint4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
        if (i0 + 4 <= seg1) *reinterpret_cast<uint4*>(out + i0) = o;
        else {
            if (i0 < seg1) out[i0] = o.x;
            if (i0 + 1 < seg1) out[i0 + 1] = o.y;
            if (i0 + 2 < seg1) out[i0 + 2] = o.z;
        }
        __syncthreads();
        if (t == 0) s_carry += all;
        __syncthreads();
    }
    if (t == 0) seg_sum[blockIdx.x] = s_carry;
}
// prefix of the segment sums, computed by whoever needs it (a handful of segments: cheaper than a launch)
__device__ __forceinline__ uint32_t seg_prefix(const uint32_t* __restrict__ seg_sum, uint32_t seg) {
    uint32_t p = 0;
    for (uint32_t i = 0; i < seg; ++i) p += seg_sum[i];
    return p;
}
__global__ __launch_bounds__(64) void k_scan_top(const uint32_t* __restrict__ seg_sum, uint32_t* __restrict__ seg_off,
                                                uint32_t n_seg, uint64_t* __restrict__ total_host,
                                                uint32_t* __restrict__ total_dev) {
    const uint32_t lane = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_seg; base += 64) {
        const uint32_t i = base + lane;
        const uint32_t v = i < n_seg ? seg_sum[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (i < n_seg) seg_off[i] = carry + x - v;
        carry += __shfl(x, 63);
    }
    if (lane == 0) {
        if (total_host) *total_host = carry;
        if (total_dev) *total_dev = carry;
    }
}

// ------------------------------------------------------------- expand pass --
// One WAVE per dense tile (four tiles per workgroup, no workgroup barrier): bitmap
// bits -> Hit records in position order, and the tile's words/count are left zero.
constexpr int kExpandTilesPerWg = kThreads / 64;

// m bases starting at pos as a 2-bit value: five aligned dwords cover any 15-byte span
__device__ __forceinline__ uint32_t mmer_at(const uint8_t* __restrict__ bases, uint64_t n, uint64_t pos, uint32_t m) {
    const uint64_t a0 = pos & ~3ull;
    const uint32_t sh = (uint32_t)(pos & 3) * 8;
    uint32_t d[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint64_t a = a0 + 4 * i;
        d[i] = a + 4 <= n ? *reinterpret_cast<const uint32_t*>(bases + a) : 0u;
        if (a + 4 > n && a < n) {            // ragged end of the buffer
            for (uint32_t b = 0; a + b < n; ++b) d[i] |= (uint32_t)bases[a + b] << (8 * b);
        }
    }
    uint32_t f = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = sh ? (d[i] >> sh) | (d[i + 1] << (32 - sh)) : d[i];   // bytes pos+4i .. pos+4i+3
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if ((uint32_t)(4 * i + b) < m) f = (f << 2) | ((w >> (8 * b + 1)) & 3u);
    }
    return f;
}

__global__ __launch_bounds__(kThreads) void k_expand(const uint8_t* __restrict__ bases, uint64_t n, uint32_t k,
                                                    uint32_t m, uint32_t* __restrict__ bitmap,
                                                    uint32_t* __restrict__ tile_count,
                                                    const uint32_t* __restrict__ tile_off,
                                                    const uint32_t* __restrict__ seg_sum, uint32_t n_seg,
                                                    uint32_t seg_shift, uint64_t n_tiles, const uint64_t* __restrict__ rec_off, uint32_t n_rec,
                                                    Hit* __restrict__ hits, uint32_t hits_cap,
                                                    uint64_t* __restrict__ total_host, uint32_t* __restrict__ total_dev,
                                                    uint32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    const uint32_t lane = threadIdx.x & 63;
    if (blockIdx.x == 0)                            // the count pass of k_resolve adds into these
        for (uint32_t i = threadIdx.x; i < n_chunks; i += kThreads) chunk_sum[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // the hit total, for the host and for k_resolve
        const uint32_t total = seg_prefix(seg_sum, n_seg);
        *total_host = total; *total_dev = total;
    }
    // the tile index is wave-uniform: held in an SGPR so that every per-tile lookup below is a scalar load
    const uint64_t b = (uint64_t)blockIdx.x * kExpandTilesPerWg + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (b >= n_tiles) return;
    if (tile_count[b] == 0) return;
    // the tile's 512 words as two coalesced 16-byte loads per lane: lane holds words
    // [4*lane, 4*lane+4) of each 256-word half, so position order = (half, lane, word)
    uint4* tile = reinterpret_cast<uint4*>(bitmap + b * kTileWords);
    uint4 q[2];
    q[0] = tile[lane]; q[1] = tile[64 + lane];
    // record holding the tile's first position: a binary search on the scalar unit (wave-uniform operands; the
    // record table sits in the scalar cache) -- a vector load with 64 scattered probe addresses costs the
    // texture path ~64 cycles per instruction, which was this kernel's largest item.  The hits then only step forward.
    const uint64_t tile_pos = b * kTilePos;
    uint32_t rlo = 0, rhi = n_rec;              // invariant: rec_off[rlo] <= tile_pos < rec_off[rhi]
    while (rhi - rlo > 1) {
        const uint32_t mid = (rlo + rhi) >> 1;
        if (rec_off[mid] <= tile_pos) rlo = mid; else rhi = mid;
    }
    // bounds of the three records a tile usually touches
    const uint64_t ro0 = rec_off[rlo];
    const uint64_t ro1 = rec_off[rlo + 1 <= n_rec ? rlo + 1 : n_rec];
    const uint64_t ro2 = rec_off[rlo + 2 <= n_rec ? rlo + 2 : n_rec];
    const uint32_t base_off = tile_off[b] + seg_prefix(seg_sum, (uint32_t)(b >> seg_shift));
    uint32_t cnt[2];
    uint32_t lrank[2];                          // tile-local rank of the lane's first hit in each half
    uint32_t total = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        cnt[h] = __popc(q[h].x) + __popc(q[h].y) + __popc(q[h].z) + __popc(q[h].w);
        uint32_t x = cnt[h];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        lrank[h] = total + x - cnt[h];
        total += __shfl(x, 63);
        // leave the bitmap all-zero behind us: the pair-table dense pass publishes hits with
        // atomicOr into a zeroed bitmap, and this saves it a 1/8 B-per-position memset
        if (cnt[h]) tile[64 * h + lane] = make_uint4(0, 0, 0, 0);
    }
    if (lane == 0) tile_count[b] = 0;          // every lane has read it (same wave, program order)
    // Hits are few and scattered over the lanes (one or two per tile at the default rate), so the lanes first
    // hand their hit positions over through LDS in rank order and then lane l builds the record of hit l: one
    // pass of m-mer loads + hash + store with up to 64 hits in flight, whatever lane found them.
    __shared__ uint32_t s_pos[kExpandTilesPerWg][64];
    uint32_t* my_pos = s_pos[threadIdx.x >> 6];
    uint32_t w[2][4] = {{q[0].x, q[0].y, q[0].z, q[0].w}, {q[1].x, q[1].y, q[1].z, q[1].w}};
    uint32_t next[2] = {lrank[0], lrank[1]};    // rank of the lane's next unpublished hit in each half
    for (uint32_t cb = 0; cb < total; cb += 64) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            while (next[h] < cb + 64 && next[h] < lrank[h] + cnt[h]) {
                uint32_t wi, bit;
                if (w[h][0]) { wi = 0; bit = __ffs(w[h][0]) - 1; w[h][0] &= w[h][0] - 1; }
                else if (w[h][1]) { wi = 1; bit = __ffs(w[h][1]) - 1; w[h][1] &= w[h][1] - 1; }
                else if (w[h][2]) { wi = 2; bit = __ffs(w[h][2]) - 1; w[h][2] &= w[h][2] - 1; }
                else { wi = 3; bit = __ffs(w[h][3]) - 1; w[h][3] &= w[h][3] - 1; }
                my_pos[next[h] - cb] = (256u * h + 4u * lane + wi) * 32u + bit;
                ++next[h];
            }
        }
        // same wave, LDS operations complete in issue order: the reads below see the writes above
        if (cb + lane < total) {
            const uint64_t pos = tile_pos + my_pos[lane];
            const uint32_t f = mmer_at(bases, n, pos, m);
            const uint32_t rc = rc_mmer32(f, m);
            Hit hrec;
            hrec.pos = pos;
            hrec.canon = f < rc ? f : rc;
            hrec.hash = xxh64_u64(hrec.canon);
            uint32_t r = rlo;
            uint64_t r0 = ro0, r1 = ro1;
            if (pos >= ro1 && rlo + 1 < n_rec) {
                r = rlo + 1; r0 = ro1; r1 = ro2;
                if (pos >= ro2 && rlo + 2 < n_rec) {              // several records inside one tile: rare
                    while (r + 1 < n_rec && rec_off[r + 1] <= pos) ++r;
                    r0 = rec_off[r]; r1 = rec_off[r + 1];
                }
            }
            hrec.rec = r;
            const bool usable = (pos + m <= r1) && (r1 - r0 >= k);
            hrec.flags = (hrec.canon != f ? 1u : 0u) | (usable ? 2u : 0u);
            hrec.pad = 0;
            const uint32_t rk = base_off + cb + lane;
            if (rk < hits_cap) hits[rk] = hrec;   // the host sees n_hits > hits_cap and retries with room
        }
    }
}

// ------------------------------------------ dense pass, blocked-Bloom form ---
// Fine sampling at long minimizers (k63 m15 s100: 1.1 x 10^5 selectable canonical 15-mers) fills a fifth of the
// 10-base prefix table, and hashing a fifth of all positions costs more than the table test of all of them.  The
// prefix table cannot be sharpened: 2^20 bits hold 2.2 x 10^5 keys (both strands) at ~5 bits each, whatever the
// index.  So this form halves the keys -- it tests the CANONICAL m-mer, two extra instructions per position --
// and spends the bits as a blocked Bloom filter: the top 15 bits of the canonical value pick one 32-bit word of a
// 128 KiB table, three 5-bit fields of the rest pick three bits in it; ONE ds_read_b32 per position, ~2 % of the
// positions (instead of ~20 %) go on to XXH64.  Everything else is k_dense_single.
template <int M>
__global__ void k_build_bloom(uint64_t thr, uint32_t* __restrict__ tab) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (1ull << (2 * M))) return;
    const uint32_t v = (uint32_t)x;
    if (rc_mmer32(v, M) < v) return;          // canonical values only
    if (xxh64_u64(v) > thr) return;
    atomicOr(&tab[v >> (2 * M - 15)], (1u << (v & 31u)) | (1u << ((v >> 5) & 31u)) | (1u << ((v >> 10) & 31u)));
}

// survivors of one lane's 16 positions: bit j = the canonical m-mer at offset j passes the filter
template <int M>
__device__ __forceinline__ uint32_t bloom_lookup16(const uint32_t* __restrict__ tab, uint32_t hi, uint32_t nxt) {
    const uint64_t R = rc_window64(((uint64_t)hi << 32) | nxt);
    const uint32_t rhi = (uint32_t)(R >> 32), rlo = (uint32_t)R;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // m-mer in the top 2M bits of a word; the bits below are other bases: they only ever break ties between a
        // palindromic m-mer and itself
        const uint32_t f = j == 0 ? hi : __builtin_amdgcn_alignbit(hi, nxt, 32 - 2 * j);
        const int rs = 2 * j + 2 * M - 32;                 // the reverse complement of offset j: bits [2j, 2j + 2M) of R
        const uint32_t r = rs < 0 ? (rlo << -rs) : (rs == 0 ? rlo : __builtin_amdgcn_alignbit(rhi, rlo, rs & 31));
        const uint32_t c = f < r ? f : r;
        const uint32_t w = tab[c >> 17];
        const uint32_t v = c >> (32 - 2 * M);
        const uint32_t t = (w >> (v & 31u)) & (w >> ((v >> 5) & 31u)) & (w >> ((v >> 10) & 31u));
        acc = __builtin_amdgcn_alignbit(t, acc, 1);
    }
    return acc >> 16;
}

constexpr int kBloomBytes = 131072;

// PACKED: 2-bit input as k_dense_pair<., true> reads it (one dword per lane and row, the ingest's k_clean_write<PACK> makes it):
// a quarter of the stream and no pack16 in a kernel that is bound by its instruction issue
template <int M, bool PACKED>
__global__ __launch_bounds__(64 * kPairWaves) void k_dense_bloom(const uint8_t* __restrict__ bases, uint64_t n, uint64_t thr,
                                                                const uint32_t* __restrict__ bloom, uint64_t n_rows, WaveLists L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds1[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(lds1);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint2* queue = reinterpret_cast<uint2*>(lds1 + kBloomBytes) + wave * kQueueCap1;   // ring of {rel pos, m-mer}
    {
        const uint4* src = reinterpret_cast<const uint4*>(bloom);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t i = threadIdx.x; i < kBloomBytes / 16; i += 64 * kPairWaves) dst[i] = src[i];
    }
    __syncthreads();
    constexpr uint32_t m = M;
    const uint64_t n_mmers = n >= m ? n - m + 1 : 0;
    const uint32_t mm = (1u << (2 * m)) - 1u;
    const uint64_t gw = (uint64_t)blockIdx.x * kPairWaves + wave;
    const uint64_t first = gw * L.rows_per_wave;              // see k_dense_pair
    const uint64_t lim = first + L.rows_per_wave < n_rows ? first + L.rows_per_wave : n_rows;
    const uint64_t n_my = first < lim ? lim - first : 0;
    const uint64_t base_pos = first * kRowPosPair63;
    constexpr uint64_t row_bytes = kRowPosPair63;
    Hit* out = L.raw + gw * L.cap;
    uint32_t out_n = 0, qn = 0, head = 0;
    auto drain = [&](uint32_t keep_below) {
        while (qn >= keep_below && qn > 0) {
            const uint32_t take = qn < 64 ? qn : 64;
            bool is_hit = false;
            uint64_t pos = 0, hash = 0;
            uint32_t f = 0, x = 0;
            if (lane < take) {
                uint32_t at = head + lane;
                if (at >= (uint32_t)kQueueCap1) at -= kQueueCap1;
                const uint2 e = queue[at];
                pos = base_pos + e.x; f = e.y;
                const uint32_t rc = rc_mmer32(f, m);
                x = f < rc ? f : rc;
                hash = xxh64_u64(x);
                is_hit = pos < n_mmers && hash <= thr;
            }
            out_n = append_hits(is_hit, pos, x, f, hash, out, L.cap, out_n);
            head += take; if (head >= (uint32_t)kQueueCap1) head -= kQueueCap1;
            qn -= take;
        }
    };
    auto handle = [&](uint32_t cand, uint32_t rel, uint32_t hi, uint32_t nxt) {
        if (!__ballot(cand != 0)) return;
        const uint32_t cnt = __popc(cand);
        uint32_t idx = 0, total = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const unsigned long long plane = __ballot((cnt >> b) & 1u);
            idx += __builtin_amdgcn_mbcnt_hi((uint32_t)(plane >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)plane, 0u)) << b;
            total += (uint32_t)__popcll(plane) << b;
        }
        const uint64_t W = ((uint64_t)hi << 32) | nxt;
        uint32_t c = cand;
        for (uint32_t base = 0; base < total; base += kPushWindow) {
            const uint32_t tail = head + qn;
            while (c && idx < base + kPushWindow) {
                const uint32_t j = __ffs(c) - 1;
                c &= c - 1;
                uint32_t at = tail + (idx - base);
                if (at >= (uint32_t)kQueueCap1) at -= kQueueCap1;
                if (at >= (uint32_t)kQueueCap1) at -= kQueueCap1;
                queue[at] = make_uint2(rel + j, (uint32_t)(W >> (64 - 2 * m - 2 * j)) & mm);
                ++idx;
            }
            qn += total - base < (uint32_t)kPushWindow ? total - base : (uint32_t)kPushWindow;
            drain(64);
        }
    };
    const bool halo_lane = lane >= kRowChunks;
    uint32_t rel = lane * kChunk;
    const uint32_t rel_step = (uint32_t)row_bytes;
    if (PACKED) {
        const uint32_t* b32 = reinterpret_cast<const uint32_t*>(bases);
        const uint64_t n_dw = (n + kChunk - 1) / kChunk;                        // dwords that hold bases (the tail of the last one is 0)
        const uint64_t full_p = n_dw >= 64 ? (n_dw - 64) / kRowChunks + 1 : 0;  // rows whose 64 dwords (halo included) exist
        const uint64_t nf_all = first < full_p ? full_p - first : 0;
        const uint64_t nf = nf_all < n_my ? nf_all : n_my;
        const uint32_t* p = b32 + first * kRowChunks + lane;
        constexpr uint32_t rs = kRowChunks;                                     // dwords per row
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        if (0 < nf) r0 = __builtin_nontemporal_load(p);
        if (1 < nf) r1 = __builtin_nontemporal_load(p + rs);
        if (2 < nf) r2 = __builtin_nontemporal_load(p + 2 * rs);
        if (3 < nf) r3 = __builtin_nontemporal_load(p + 3 * rs);
        uint64_t i = 0;
        auto row = [&](uint32_t& r, uint64_t ahead, uint32_t rl) {             // (unconditional refill: past its last row a wave re-reads its current one)
            const uint32_t hi = r; r = __builtin_nontemporal_load(i + ahead < nf ? p + ahead * rs : p);
            const uint32_t nxt = next_lane(hi);
            const uint32_t c = bloom_lookup16<M>(tab, hi, nxt);
            handle(halo_lane ? 0u : c, rl, hi, nxt);
        };
        for (; i + 3 < nf; i += 4, p += 4 * rs, rel += 4 * rel_step) {
            row(r0, 4, rel);
            row(r1, 5, rel + rel_step);
            row(r2, 6, rel + 2 * rel_step);
            row(r3, 7, rel + 3 * rel_step);
        }
        for (; i < n_my; ++i, rel += rel_step) {        // up to three rows left, and the rows at the end of the buffer
            const uint64_t at = (first + i) * kRowChunks + lane;
            const uint32_t hi = at < n_dw ? b32[at] : 0u;
            const uint32_t nxt = next_lane(hi);
            const uint32_t c = bloom_lookup16<M>(tab, hi, nxt);
            handle(halo_lane ? 0u : c, rel, hi, nxt);
        }
        drain(1);
        if (lane == 0) L.cnt[gw] = out_n;
        return;
    }
    const uint64_t full_rows = n >= 64 * kChunk ? (n - 64 * kChunk) / kRowPosPair63 + 1 : 0;
    const uint64_t n_fast_all = first < full_rows ? full_rows - first : 0;
    const uint64_t n_fast = n_fast_all < n_my ? n_fast_all : n_my;
    uint64_t i = 0;
    const uint8_t* ptr = bases + first * kRowPosPair63 + (uint64_t)lane * kChunk;
    uint4 raw0 = make_uint4(0, 0, 0, 0), raw1 = raw0;
    if (i < n_fast) raw0 = load_row16(ptr);
    if (i + 1 < n_fast) raw1 = load_row16(ptr + row_bytes);
    auto body = [&](uint4& raw, uint64_t r, const uint8_t* at, uint32_t rl) {
        const uint32_t hi = pack16(raw);
        raw = load_row16(r + 2 < n_fast ? at + 2 * row_bytes : at);
        const uint32_t nxt = next_lane(hi);
        const uint32_t c = bloom_lookup16<M>(tab, hi, nxt);
        handle(halo_lane ? 0u : c, rl, hi, nxt);
    };
    for (; i + 1 < n_fast; i += 2, ptr += 2 * row_bytes, rel += 2 * rel_step) {
        body(raw0, i, ptr, rel);
        body(raw1, i + 1, ptr + row_bytes, rel + rel_step);
    }
    if (i < n_fast) { body(raw0, i, ptr, rel); ++i; rel += rel_step; }
    for (; i < n_my; ++i, rel += rel_step) {
        const uint32_t hi = load_pack(bases, n, (first + i) * kRowPosPair63 + (uint64_t)lane * kChunk);
        const uint32_t nxt = next_lane(hi);
        const uint32_t c = bloom_lookup16<M>(tab, hi, nxt);
        handle(halo_lane ? 0u : c, rel, hi, nxt);
    }
    drain(1);
    if (lane == 0) L.cnt[gw] = out_n;
}

// ------------------------------------------------------------ compact pass --
// Per-wave hit lists of the table variants -> one array in position order, with the record of every hit
// filled in.  Wave w's hits precede wave w+1's and each list is sorted, so the place of a hit is the sum of
// the counts before its wave plus its index: every workgroup sums the counts in front of its own
// kCompactWaves lists itself (a few KiB from L2 -- cheaper than a scan launch); the LAST workgroup thereby
// holds the grand total and the largest count and publishes them (overflow of a list = largest > cap).
constexpr int kCompactThreads = 256, kCompactWaves = 64;
__global__ __launch_bounds__(kCompactThreads) void k_compact(WaveLists L, uint32_t n_lists, uint64_t n, uint32_t k, uint32_t m,
                                                            const uint64_t* __restrict__ rec_off, uint32_t n_rec,
                                                            Hit* __restrict__ hits, uint32_t hits_cap,
                                                            uint64_t* __restrict__ total_host, uint32_t* __restrict__ total_dev,
                                                            uint32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    __shared__ uint32_t s_red[kCompactThreads / 64][2];
    __shared__ uint32_t s_pre[kCompactWaves + 1];
    __shared__ uint32_t s_rec[2];
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    if (blockIdx.x == 0)                            // the count pass of k_resolve adds into these
        for (uint32_t i = t; i < n_chunks; i += kCompactThreads) chunk_sum[i] = 0;
    const uint32_t w0 = blockIdx.x * kCompactWaves;
    // sum and maximum of the counts in front of this workgroup's lists
    uint32_t sum = 0, mx = 0;
    for (uint32_t i = t; i < w0; i += kCompactThreads) {
        const uint32_t c = L.cnt[i];
        sum += c < L.cap ? c : L.cap;
        mx = c > mx ? c : mx;
    }
    // own lists: clamped counts -> exclusive prefix (first wave)
    uint32_t own = 0, own_raw = 0;
    if (t < (uint32_t)kCompactWaves && w0 + t < n_lists) { own_raw = L.cnt[w0 + t]; own = own_raw < L.cap ? own_raw : L.cap; }
    if (wid == 0) {
        uint32_t x = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        s_pre[lane] = x - own;
        if (lane == 63) s_pre[64] = x;
        mx = own_raw > mx ? own_raw : mx;
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) { sum += __shfl_xor(sum, d); const uint32_t o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
    if (lane == 0) { s_red[wid][0] = sum; s_red[wid][1] = mx; }
    // records this workgroup's positions can fall into: [rlo, rhi]
    const uint64_t span = L.rows_per_wave * kRowPosPair63;
    if (t == 64 || t == 128) {
        const uint64_t want = t == 64 ? (uint64_t)w0 * span : ((uint64_t)w0 + kCompactWaves) * span;
        uint32_t rlo = 0, rhi = n_rec;              // invariant: rec_off[rlo] <= want (rec_off[0] = 0) < rec_off[rhi] or rhi = n_rec
        while (rhi - rlo > 1) {
            const uint32_t mid = (rlo + rhi) >> 1;
            if (rec_off[mid] <= want) rlo = mid; else rhi = mid;
        }
        s_rec[t == 64 ? 0 : 1] = rlo;
    }
    __syncthreads();
    uint32_t base = 0, gmax = 0;
#pragma unroll
    for (int w = 0; w < kCompactThreads / 64; ++w) { base += s_red[w][0]; gmax = s_red[w][1] > gmax ? s_red[w][1] : gmax; }
    const uint32_t mine = s_pre[kCompactWaves];
    if (blockIdx.x == gridDim.x - 1 && t == 0) {    // the hit total and the fullest list, for the host and for k_resolve
        total_host[0] = (uint64_t)base + mine;
        total_host[2] = gmax;
        *total_dev = base + mine;
    }
    const uint32_t rec_lo = s_rec[0], rec_hi = s_rec[1];
    for (uint32_t i = t; i < mine; i += kCompactThreads) {
        uint32_t lo = 0, hi = kCompactWaves;        // list holding item i: last l with s_pre[l] <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pre[mid] <= i) lo = mid; else hi = mid;
        }
        Hit h = L.raw[(uint64_t)(w0 + lo) * L.cap + (i - s_pre[lo])];
        uint32_t a = rec_lo, b = rec_hi + 1 < n_rec ? rec_hi + 1 : n_rec;   // rec_off[a] <= pos < rec_off[b]
        while (b - a > 1) {
            const uint32_t mid = (a + b) >> 1;
            if (rec_off[mid] <= h.pos) a = mid; else b = mid;
        }
        // equal offsets (empty records): the record holding pos is the LAST one starting at or before it
        const uint64_t r0 = rec_off[a], r1 = rec_off[a + 1];
        h.rec = a;
        const bool usable = (h.pos + m <= r1) && (r1 - r0 >= k);
        h.flags = (h.flags & 1u) | (usable ? 2u : 0u);
        const uint32_t rk = base + i;
        if (rk < hits_cap) hits[rk] = h;            // the host sees n_hits > hits_cap and retries with room
    }
}

// dense-only callers (spsp_scan_hits_device): the number of hits = sum of the per-wave counts
__global__ __launch_bounds__(1024) void k_sum_counts(const uint32_t* __restrict__ cnt, uint32_t n_lists,
                                                    uint64_t* __restrict__ total_host) {
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < n_lists; i += 1024) s += cnt[i];
#pragma unroll
    for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum, s);
    __syncthreads();
    if (threadIdx.x == 0) *total_host = s_sum;
}

// ------------------------------------------------------------ resolve pass --
// Literal replay of the reference state machine over one cluster of hits.
struct Rescan { uint32_t mini; uint32_t rev; uint64_t position; uint64_t hash; };

// The hits a workgroup of k_resolve works on, staged in LDS: a cluster is replayed with a chain of dependent reads
// (next hit, window bounds, rescans), and from LDS a link of that chain costs ~100 cycles instead of an L2 round trip.
// Indices outside the staged window (a cluster that runs far past the workgroup's hits) fall back to global memory.
struct HitView {
    const Hit* g;        // all hits, position order
    const Hit* s;        // LDS copy of g[lo, hi)
    uint32_t lo, hi;
    __device__ __forceinline__ Hit at(uint32_t i) const { return (i >= lo && i < hi) ? s[i - lo] : g[i]; }
};

// regular_minimizer_pos (SubSampler.cpp:81-169) restricted to the hits
// H[lo..hi] of the window starting at m-mer position ws.  Non-hit m-mers can
// never win (their hash is > T >= any hit's) and can never tie (XXH64 on 8
// bytes is a bijection), so skipping them leaves every assignment identical.
__device__ __forceinline__ Rescan rescan_hits(const HitView& V, uint32_t hb, uint64_t r0, uint32_t lo, uint32_t hi,
                                              uint64_t ws, uint64_t km) {
    Rescan r;
    {
        const Hit h = V.at(hb + hi);
        const uint64_t off = (h.pos - r0) - ws;
        r.mini = h.canon; r.hash = h.hash; r.rev = h.flags & 1u;
        if (off == km) r.position = r.rev ? 0 : km;  // rightmost m-mer: :88-93 (reverse => position 0, sic)
        else r.position = off;                       // first hit met strictly beats the non-hits right of it
    }
    for (uint32_t idx = hi; idx-- > lo;) {
        const Hit h = V.at(hb + idx);
        const uint64_t off = (h.pos - r0) - ws;
        const uint64_t ii = km - off;  // the reference's loop index i
        const uint32_t lrev = h.flags & 1u;
        if (r.hash > h.hash) {                                       // :117-129
            r.position = off; r.mini = h.canon; r.rev = lrev; r.hash = h.hash;
        } else if (h.canon == r.mini) {                              // :132-166
            if (lrev == r.rev) {
                if (r.rev && r.position > ii) r.position = ii;       // :151-157 (sic)
                if (!r.rev && r.position > off) r.position = off;    // :158-164
            }
        }
    }
    return r;
}

// cluster = hits hb .. hb + cnt - 1 of V
// `cut`: the last of the cnt hits is the FIRST hit of the next piece of a cluster cut in two (head_kind): this piece ends
// with the iteration in which that hit enters and takes over -- the super-k-mer it closes is this piece's last.
template <bool WRITE>
__device__ uint32_t run_cluster(const HitView& V, uint32_t hb, uint32_t cnt, uint64_t r0, uint64_t n, uint32_t k,
                                uint32_t m, uint32_t rec, spsp_superkmer* __restrict__ out, uint32_t room, bool cut) {
    const uint64_t km = k - m, w = km + 1;
    auto q = [&](uint32_t i) -> uint64_t { return V.at(hb + i).pos - r0; };
    uint32_t nem = 0;
    auto emit = [&](uint64_t start, uint64_t len, uint32_t mini, uint32_t rev) {
        if (WRITE && nem < room) {
            spsp_superkmer e;
            e.rec = rec; e.minimizer = mini; e.start = start; e.len = (uint32_t)len; e.rev = rev;
            out[nem] = e;
        }
        ++nem;
    };
    const uint64_t q0 = q(0);
    uint32_t minimizer, rev;
    uint64_t hash_min, position_min, last_position, i;
    uint32_t lo = 0;  // first hit with position >= current window start
    if (q0 > km) {
        // iteration i = q0-w: the hit enters as the rightmost m-mer and beats an
        // unselected minimum (SubSampler.cpp:374-388); the old super-k-mer is not selected.
        const Hit h = V.at(hb);
        minimizer = h.canon; hash_min = h.hash; position_min = q0; rev = h.flags & 1u;
        last_position = q0 - km; i = q0 - km;
    } else {
        // record start (SubSampler.cpp:359-365): rescan of k-mer 0
        uint32_t hi = 0;
        while (hi + 1 < cnt && q(hi + 1) <= km) ++hi;
        const Rescan r = rescan_hits(V, hb, r0, 0, hi, 0, km);
        minimizer = r.mini; hash_min = r.hash; position_min = r.position; rev = r.rev;
        last_position = 0; i = 0;
    }
    uint32_t old_min = minimizer, old_rev = rev;
    uint32_t nh = 0;  // first hit with position >= i + w
    bool closed = false;
    while (i + k < n) {                                           // :367
        const uint64_t pn = i + w;                                // position of the entering m-mer
        while (nh < cnt && q(nh) < pn) ++nh;
        const Hit hn = nh < cnt ? V.at(hb + nh) : Hit{};
        const bool enters = nh < cnt && hn.pos - r0 == pn;
        bool dump = false;
        if (enters && hn.hash < hash_min) {                       // :374-388
            minimizer = hn.canon; hash_min = hn.hash; position_min = pn; rev = hn.flags & 1u;
        } else if (i >= position_min) {                           // :391-398
            while (lo < cnt && q(lo) < i + 1) ++lo;
            if (lo >= cnt || q(lo) > pn) {
                // no hit left in the window: the new minimizer is unselected, the
                // cluster's last super-k-mer is emitted and the cluster is over.
                emit(last_position, i + k - last_position, old_min, old_rev);
                closed = true;
                break;
            }
            const uint32_t hi = enters ? nh : nh - 1;
            const Rescan r = rescan_hits(V, hb, r0, lo, hi, i + 1, km);
            minimizer = r.mini; rev = r.rev; hash_min = r.hash;
            position_min = r.position + i + 1;
            dump = true;
        }
        if (old_min != minimizer || dump) {                       // :401-435 (old one is selected)
            emit(last_position, i + k - last_position, old_min, old_rev);
            last_position = i + 1; old_min = minimizer; old_rev = rev;
        }
        if (cut && enters && nh + 1 == cnt) { closed = true; break; }   // the next piece starts from exactly this state
        // skip iterations in which nothing can happen: the next event is the
        // next hit entering or the tracked minimizer leaving the window.
        uint64_t ni = position_min;
        const uint32_t nx = enters ? nh + 1 : nh;
        if (nx < cnt) { const uint64_t e = q(nx) - w; if (e < ni) ni = e; }
        i = ni > i + 1 ? ni : i + 1;
    }
    if (!closed) emit(last_position, n - last_position, old_min, old_rev);  // :441-454 tail
    return nem;
}

// Where a replay may start.  1: the first hit of a cluster (no usable hit of its record within w positions in front of
// it).  2: a hit whose hash is strictly below every hit of the w positions in front of it, beyond the record's first
// window -- when it enters, the reference's state machine (SubSampler.cpp:374-388) takes it as the new minimizer whatever
// it was tracking (the tracked m-mer is one of those w, its hash is larger, so is its canonical value: XXH64 of 8 bytes
// is a bijection), closes the running super-k-mer and goes on from a state that depends on this hit alone: the same state
// a cluster starts from (run_cluster, q0 > km).  So a cluster can be cut in front of every such hit, the piece in front of
// it replayed up to and including that iteration (`cut`), the piece behind it like a cluster of its own: identical output.
// A strict window minimum comes along every ~w positions at the latest when every m-mer is selected (s = 1: one
// cluster per RECORD, replayed by one lane -- 7 s per 10^8 bases before this).
__device__ __forceinline__ uint32_t head_kind(const HitView& V, uint32_t x, uint64_t w, uint64_t km, const uint64_t* __restrict__ rec_off) {
    const Hit me = V.at(x);
    if (!(me.flags & 2u)) return 0u;
    bool any = false;
    for (uint32_t g = x; g-- > 0;) {
        const Hit o = V.at(g);
        if (me.pos - o.pos > w) break;
        if ((o.flags & 2u) && o.rec == me.rec) {
            if (o.hash <= me.hash) return 0u;
            any = true;
        }
    }
    if (!any) return 1u;
    return me.pos - rec_off[me.rec] > km ? 2u : 0u;
}

// One lane per hit; cluster heads replay their cluster.  The count pass (WRITE = false) leaves, besides the
// per-hit counts, one sum per wave and -- with one atomic per wave -- one sum per chunk of 64 waves; the write
// pass turns those into its output offset itself (a masked wave load of each level + one wave reduction), so
// no scan kernel sits between the two passes.
constexpr int kResolveThreads = 128;
constexpr int kResolveBefore = 8, kResolveAfter = 120;        // hits staged in LDS in front of / behind the workgroup's own
template <bool WRITE>
__global__ __launch_bounds__(kResolveThreads) void k_resolve(const Hit* __restrict__ hits, const uint32_t* __restrict__ n_hits_dev,
                                                             uint32_t hits_cap, const uint64_t* __restrict__ rec_off, uint32_t k,
                                                             uint32_t m, uint32_t* __restrict__ emit_count,
                                                             uint32_t* __restrict__ wave_sum, uint32_t* __restrict__ chunk_sum,
                                                             uint32_t n_chunks, uint64_t* __restrict__ total_host,
                                                             spsp_superkmer* __restrict__ out, uint32_t out_cap, uint32_t n_super) {
    __shared__ __attribute__((aligned(16))) Hit s_hit[kResolveThreads + kResolveBefore + kResolveAfter];
    const uint32_t h = blockIdx.x * kResolveThreads + threadIdx.x, lane = threadIdx.x & 63, gw = h >> 6;
    uint32_t n_hits = *n_hits_dev;
    if (n_hits > hits_cap) n_hits = hits_cap;   // overflowed: this pass is discarded by the host
    HitView V;
    V.g = hits; V.s = s_hit;
    {
        const uint32_t b0 = blockIdx.x * kResolveThreads;
        V.lo = b0 > (uint32_t)kResolveBefore ? b0 - kResolveBefore : 0u;
        const uint64_t end = (uint64_t)b0 + kResolveThreads + kResolveAfter;
        V.hi = end < n_hits ? (uint32_t)end : n_hits;
        if (V.hi < V.lo) V.hi = V.lo;
        const uint4* src = reinterpret_cast<const uint4*>(hits + V.lo);
        uint4* dst = reinterpret_cast<uint4*>(s_hit);
        for (uint32_t i = threadIdx.x; i < 2 * (V.hi - V.lo); i += kResolveThreads) dst[i] = src[i];
    }
    __syncthreads();
    const uint64_t w = k - m + 1, km = k - m;
    // where replays start among the staged hits (a piece that runs past them asks hit by hit)
    __shared__ uint8_t s_kind[kResolveThreads + kResolveBefore + kResolveAfter];
    for (uint32_t x = V.lo + threadIdx.x; x < V.hi; x += kResolveThreads) s_kind[x - V.lo] = (uint8_t)head_kind(V, x, w, km, rec_off);
    __syncthreads();
    Hit me{};
    bool head = false, cut = false;
    uint32_t cnt = 0;
    if (h < n_hits) {
        me = V.at(h);
        head = s_kind[h - V.lo] != 0;
        if (head) {
            cnt = 1;
            Hit a = me;
            while (h + cnt < n_hits) {
                const uint32_t x = h + cnt;
                const Hit b = V.at(x);
                if (!(b.flags & 2u) || b.rec != me.rec || b.pos - a.pos > w) break;
                if (x < V.hi ? s_kind[x - V.lo] != 0 : head_kind(V, x, w, km, rec_off) != 0) { cut = true; ++cnt; break; }   // (the piece sees the hit that ends it)
                a = b;
                ++cnt;
            }
        }
    }
    uint64_t r0 = 0, r1 = 0;
    if (head) { r0 = rec_off[me.rec]; r1 = rec_off[me.rec + 1]; }
    if (!WRITE) {
        const uint32_t c = head ? run_cluster<false>(V, h, cnt, r0, r1 - r0, k, m, me.rec, nullptr, 0u, cut) : 0u;
        if (h < hits_cap) emit_count[h] = c;
        uint32_t t = c;
#pragma unroll
        for (int d = 32; d; d >>= 1) t += __shfl_xor(t, d);
        if (lane == 0) {
            wave_sum[gw] = t;
            if (t) atomicAdd(&chunk_sum[gw >> 6], t);
        }
        return;
    }
    // write pass: offset = chunks before mine + waves of my chunk before mine + lanes of my wave before me
    const uint32_t c = h < hits_cap ? emit_count[h] : 0u;
    uint32_t x = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    // (n_super != 0: a third level, sums of 64 chunks made by k_super_sums between the passes -- with two levels a wave of an
    // input that is all hits summed 10^5 chunk sums, quadratic in the hits; few hits: two levels and no launch in between)
    const uint32_t wi = gw & 63u, ci = gw >> 6, si = ci >> 6;
    const uint32_t* super_sum = chunk_sum + n_chunks;
    uint32_t v = lane < wi ? wave_sum[(gw & ~63u) + lane] : 0u;
    if (n_super) {
        if ((ci & ~63u) + lane < ci) v += chunk_sum[(ci & ~63u) + lane];
        for (uint32_t j = lane; j < si; j += 64) v += super_sum[j];
    } else {
        for (uint32_t j = lane; j < ci; j += 64) v += chunk_sum[j];
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    if (gw == 0) {   // number of super-k-mers, for the host
        uint32_t tv = 0;
        if (n_super) { for (uint32_t j = lane; j < n_super; j += 64) tv += super_sum[j]; }
        else for (uint32_t j = lane; j < n_chunks; j += 64) tv += chunk_sum[j];
#pragma unroll
        for (int d = 32; d; d >>= 1) tv += __shfl_xor(tv, d);
        if (lane == 0) *total_host = tv;
    }
    if (head) {
        const uint32_t at = v + x - c;
        run_cluster<true>(V, h, cnt, r0, r1 - r0, k, m, me.rec, out + at, at < out_cap ? out_cap - at : 0u, cut);
    }
}

// super_sum[j] = chunk sums 64 j .. 64 j + 63 (one wave each), written behind the chunk sums
__global__ __launch_bounds__(64) void k_super_sums(uint32_t* __restrict__ chunk_sum, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    uint32_t v = c < n_chunks ? chunk_sum[c] : 0u;
#pragma unroll
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    if (threadIdx.x == 0) chunk_sum[n_chunks + blockIdx.x] = v;
}

// Long inputs (the super-k-mers of a metagenome segment: 8 x 10^5 counts) in blocks: one workgroup's rounds over all of them
// were 0.75 ms of a 7 ms key extraction.  Block sums -> their scan (the single-workgroup kernel, which also publishes the
// total) -> every block scanned from its offset.
constexpr uint32_t kScanBlock = 8192;
__global__ __launch_bounds__(1024) void k_scan_block_sums(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t wave_sum[16];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * 8;
    uint32_t sum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += base + u < n ? in[base + u] : 0u;
#pragma unroll
    for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63u) == 0) wave_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t all = 0; for (uint32_t w = 0; w < 16; ++w) all += wave_sum[w]; sums[blockIdx.x] = all; }
}
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                     const uint32_t* __restrict__ block_off) {
    __shared__ uint32_t wave_sum[16];
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)t * 8;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = i0 + u < n ? in[i0 + u] : 0u; sum += v[u]; }
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wid; ++w) pre += wave_sum[w];
    uint32_t run = block_off[blockIdx.x] + pre + x - sum;
#pragma unroll
    for (int u = 0; u < 8; ++u) { if (i0 + u < n) out[i0 + u] = run; run += v[u]; }
}
int launch_scan_u32(spsp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* total_host) {
    if (n > 4 * (uint64_t)kScanBlock) {
        const uint32_t blocks = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
        int rc = ctx->scan_blocks.reserve((size_t)(2 * blocks + 2) * 4);
        if (rc) return rc;
        uint32_t* sums = ctx->scan_blocks.as<uint32_t>();
        uint32_t* offs = sums + blocks;                                  // [blocks + 1]: the scan's own total lands in offs[blocks]
        hipLaunchKernelGGL(k_scan_block_sums, dim3(blocks), dim3(1024), 0, ctx->stream, d_in, n, sums);
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)sums, offs, (uint64_t)blocks, (const uint32_t*)nullptr,
                           total_host, d_out + n);                       // out[n] = total, as the one-kernel form leaves it
        hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(1024), 0, ctx->stream, d_in, d_out, n, (const uint32_t*)offs);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    }
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, ctx->stream, d_in, d_out, n, (const uint32_t*)nullptr,
                       total_host, (uint32_t*)nullptr);
    SPSP_HIP(hipGetLastError());
    return SPSP_OK;
}

// ----------------------------------------------------------------- helpers --
int check_params(const spsp_params* p) {
    if (!p) { set_error("params is NULL"); return SPSP_ERR_ARG; }
    if (p->m < 1 || p->m > 15) { set_error("m=%u out of range 1..15", p->m); return SPSP_ERR_ARG; }
    if (p->k < p->m || p->k > 63) { set_error("k=%u out of range m..63", p->k); return SPSP_ERR_ARG; }
    return SPSP_OK;
}

// Picks the dense-pass variant from the expected survivor rate of each memoised table:
// P(hash <= T) times the m-mers (both strands) that share one table bit.
enum { kDenseDirect = 0, kDenseSingle = 1, kDensePair = 2, kDenseBloom = 3 };
static int pick_dense(const spsp_params* p);
bool scan_reads_packed(const spsp_params* p) { const int v = pick_dense(p); return v == kDensePair || v == kDenseBloom; }
static int pick_dense(const spsp_params* p) {
    const bool bloom_ok = p->m == 13 || p->m == 15;
    if (p->flags & SPSP_SCAN_DIRECT_HASH) return kDenseDirect;
    if ((p->flags & SPSP_SCAN_BLOOM_FILTER) && bloom_ok) return kDenseBloom;
    if ((p->flags & SPSP_SCAN_LDS_FILTER) && p->m >= 10) return kDenseSingle;
    if ((p->flags & SPSP_SCAN_PAIR_FILTER) && p->m >= 9) return kDensePair;
    const double frac = (double)p->threshold / 18446744073709551616.0;  // P(hash <= T)
    const uint32_t bits = 2 * p->m;
    // pair table: the weaker of an entry's two tests sees 8 bases at m = 9, nine from m = 10 on (k_build_pairtab)
    if (p->m >= 9 && frac * (double)(1u << (bits - (p->m >= 10 ? 18 : 16))) < 0.01) return kDensePair;
    // expected share of positions that go on to XXH64: prefix table (both strands' m-mers per 10-base prefix) against
    // the blocked Bloom filter (three bits per canonical m-mer in one of 2^15 words, ~fill^3)
    const double single = p->m >= 10 ? frac * (double)(1u << (bits - 20)) : 1.0;
    double bloom = 1.0;
    if (bloom_ok) {
        const double keys_per_word = frac * (double)(1ull << bits) / 2.0 / 32768.0;
        const double fill = 1.0 - exp(-3.0 * keys_per_word / 32.0);
        bloom = fill * fill * fill;
    }
    // measured on 5 x 10^8 positions: prefix-table test 0.235 ms + ~1.0 ms x (share hashed); Bloom test 0.26 ms + the same
    if (single < 0.10 || (single < 0.30 && !(bloom < 0.5 * single))) return kDenseSingle;
    if (bloom < 0.30) return kDenseBloom;
    return kDenseDirect;
}

static int ensure_bloom(spsp_ctx* ctx, const spsp_params* p) {
    if (ctx->bloom_valid && ctx->bloom_m == p->m && ctx->bloom_thr == p->threshold) return SPSP_OK;
    int rc = ctx->bloom.reserve((size_t)kBloomBytes);
    if (rc) return rc;
    SPSP_HIP(hipMemsetAsync(ctx->bloom.p, 0, (size_t)kBloomBytes, ctx->stream));
    const uint64_t total = 1ull << (2 * p->m);
    if (p->m == 15) hipLaunchKernelGGL(k_build_bloom<15>, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, p->threshold, ctx->bloom.as<uint32_t>());
    else hipLaunchKernelGGL(k_build_bloom<13>, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, p->threshold, ctx->bloom.as<uint32_t>());
    SPSP_HIP(hipGetLastError());
    ctx->bloom_m = p->m; ctx->bloom_thr = p->threshold; ctx->bloom_valid = true;
    return SPSP_OK;
}

static int ensure_key10(spsp_ctx* ctx, const spsp_params* p) {
    if (ctx->filter_valid && ctx->filter_m == p->m && ctx->filter_thr == p->threshold) return SPSP_OK;
    int rc = ctx->filter.reserve((size_t)kKey10Bytes);
    if (rc) return rc;
    SPSP_HIP(hipMemsetAsync(ctx->filter.p, 0, (size_t)kKey10Bytes, ctx->stream));
    const uint64_t total = 1ull << (2 * p->m);
    hipLaunchKernelGGL(k_build_key10, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, p->m,
                       p->threshold, ctx->filter.as<uint32_t>());
    SPSP_HIP(hipGetLastError());
    ctx->filter_m = p->m; ctx->filter_thr = p->threshold; ctx->filter_valid = true;
    return SPSP_OK;
}

static int ensure_pairtab(spsp_ctx* ctx, const spsp_params* p) {
    if (ctx->pair_valid && ctx->pair_m == p->m && ctx->pair_thr == p->threshold) return SPSP_OK;
    int rc = ctx->pairtab.reserve((size_t)kPairTabBytes + 8192 + 32768 + 32768);
    if (rc) return rc;
    uint32_t* key8 = reinterpret_cast<uint32_t*>(ctx->pairtab.as<uint8_t>() + kPairTabBytes);  // 2^16 bits
    uint32_t* key9 = key8 + 8192 / 4;                                                           // 2^18 bits
    uint32_t* mid9 = p->m >= 10 ? key9 + 32768 / 4 : nullptr;                                   // 2^18 bits
    SPSP_HIP(hipMemsetAsync(key8, 0, 8192 + 32768 + 32768, ctx->stream));
    const uint64_t total = 1ull << (2 * p->m);
    hipLaunchKernelGGL(k_build_key8, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, p->m,
                       p->threshold, key8, key9, mid9);
    SPSP_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_build_pairtab, dim3(kPairTabBytes / 256), dim3(256), 0, ctx->stream, key8, key9, mid9,
                       ctx->pairtab.as<uint8_t>());
    SPSP_HIP(hipGetLastError());
    ctx->pair_m = p->m; ctx->pair_thr = p->threshold; ctx->pair_valid = true;
    return SPSP_OK;
}

// geometry of the per-wave hit lists for one dense launch of a table variant
struct ListPlan { uint32_t grid, n_lists, cap; uint64_t n_rows, rows_per_wave; };
static int plan_lists(spsp_ctx* ctx, const spsp_params* p, int variant, uint64_t n_bases, ListPlan* P) {
    P->n_rows = (n_bases + kRowPosPair63 - 1) / kRowPosPair63;
    const uint64_t want = (P->n_rows + kPairWaves - 1) / kPairWaves;
    static const int per_cu_env = getenv("SPSP_PAIR_BLOCKS_PER_CU") ? atoi(getenv("SPSP_PAIR_BLOCKS_PER_CU")) : 0;  // tuning knob
    // One 1024-lane workgroup per CU unless the caller says its stream owns its CUs (spsp_set_cu_count).  k_dense_pair
    // (80 KiB of LDS) fits twice, and alone it runs 3-4 % faster that way (more loads in flight); with one, the other
    // half of the CU's wave slots and LDS stays free for the kernels of other streams that share the CU (a pipelined step
    // on unpartitioned streams: 0.158 vs 0.185 ms)
    const int per_cu = per_cu_env > 0 ? per_cu_env : ctx->dense_blocks_per_cu;
    const uint64_t cap_blocks = (uint64_t)ctx->n_cu * (per_cu > 0 ? per_cu : 1);
    P->grid = (uint32_t)(want < cap_blocks ? want : cap_blocks);
    P->n_lists = P->grid * kPairWaves;
    P->rows_per_wave = (P->n_rows + P->n_lists - 1) / P->n_lists;
    if (P->rows_per_wave * kRowPosPair63 >= 0xffff0000ull) { set_error("input too large for one call"); return SPSP_ERR_OVERFLOW; }
    // slice per wave: 1.5 x the expected hits + slack for the Poisson tail; grown by the caller after an overflow
    const double frac = (double)p->threshold / 18446744073709551616.0;
    const double expect = (double)P->rows_per_wave * kRowPosPair63 * frac;
    uint64_t cap = (uint64_t)(expect * 1.5) + 48;
    static const char* dbg_hits = getenv("SPSP_DEBUG_HITS_CAP");      // test hook: tiny lists, so the overflow paths run
    if (dbg_hits && ctx->list_cap == 0) cap = (uint64_t)atoll(dbg_hits) / 8 + 1;
    if (ctx->list_cap > cap && ctx->list_cap_threshold == p->threshold) cap = ctx->list_cap;   // grown by an earlier overflow at this threshold: keep
    const uint64_t most = P->rows_per_wave * kRowPosPair63;           // a wave cannot find more hits than it has positions
    if (cap > most) cap = most;
    if (cap > 0xfffffff0ull) { set_error("input too large for one call"); return SPSP_ERR_OVERFLOW; }
    P->cap = (uint32_t)cap;
    return SPSP_OK;
}

// SPSP_SCAN_PACKED_INPUT: the pair-table pass and the blocked-Bloom pass read 2-bit input; the other variants (and the bitmap
// fallback) get an ASCII copy made on the device.  Returns the buffer and flags the pass should use.
static int unpack_if_needed(spsp_ctx* ctx, spsp_params* p, const uint8_t** d_bases, uint64_t n_bases, bool use_bitmap) {
    if (!(p->flags & SPSP_SCAN_PACKED_INPUT)) return SPSP_OK;
    if (!use_bitmap && (pick_dense(p) == kDensePair || pick_dense(p) == kDenseBloom)) return SPSP_OK;
    int rc = ctx->unpacked.reserve((size_t)n_bases + 64);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_bases, dim3((uint32_t)std::min<uint64_t>((n_bases / kChunk + 256) / 256, (uint64_t)ctx->n_cu * 32)), dim3(256), 0,
                       ctx->stream, reinterpret_cast<const uint32_t*>(*d_bases), n_bases, ctx->unpacked.as<uint8_t>());
    SPSP_HIP(hipGetLastError());
    *d_bases = ctx->unpacked.as<uint8_t>();
    p->flags &= ~SPSP_SCAN_PACKED_INPUT;
    return SPSP_OK;
}

int pack_bases_impl(spsp_ctx* ctx, const uint8_t* d_bases, uint64_t n_bases, uint32_t** d_packed) {
    const uint64_t n_dw = (n_bases + kChunk - 1) / kChunk + 64;           // + halo dwords, zero
    int rc = ctx->packed.reserve((size_t)n_dw * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_bases, dim3((uint32_t)std::min<uint64_t>((n_dw + 255) / 256, (uint64_t)ctx->n_cu * 32)), dim3(256), 0, ctx->stream,
                       d_bases, n_bases, ctx->packed.as<uint32_t>(), n_dw);
    SPSP_HIP(hipGetLastError());
    *d_packed = ctx->packed.as<uint32_t>();
    return SPSP_OK;
}

// dense pass of one attempt; *lists = per-wave hit lists were produced (table variants), else bitmap + tile counts
static int launch_dense(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                        uint64_t n_tiles, bool use_bitmap, ListPlan* LP, bool* lists) {
    int rc;
    const int variant = use_bitmap ? kDenseDirect : pick_dense(p);
    *lists = variant != kDenseDirect;
    const bool packed_in = (p->flags & SPSP_SCAN_PACKED_INPUT) != 0;      // (only the pair-table and Bloom passes get here with it: unpack_if_needed)
    if (packed_in && variant != kDensePair && variant != kDenseBloom) { set_error("internal: packed input reached the %d dense variant", variant); return SPSP_ERR_ARG; }
    if ((rc = ctx->d_scalar.reserve(64))) return rc;
    if (*lists) {
        if ((rc = plan_lists(ctx, p, variant, n_bases, LP))) return rc;
        if ((rc = ctx->wave_hits.reserve((size_t)LP->n_lists * LP->cap * sizeof(Hit)))) return rc;
        if ((rc = ctx->wave_cnt.reserve((size_t)LP->n_lists * 4))) return rc;
        if (variant == kDenseSingle && (rc = ensure_key10(ctx, p))) return rc;
        if (variant == kDensePair && (rc = ensure_pairtab(ctx, p))) return rc;
        if (variant == kDenseBloom && (rc = ensure_bloom(ctx, p))) return rc;
    } else {
        if ((rc = ctx->bitmap.reserve((size_t)n_tiles * kTileWords * 4))) return rc;
        if ((rc = ctx->tile_count.reserve((size_t)n_tiles * 4))) return rc;
        if ((rc = ctx->tile_off.reserve((size_t)(n_tiles + 1) * 4))) return rc;
    }
    // The kernel's own completion signal serves as the event other streams wait on (spsp_wait_dense, the sparse stages'
    // stream) and, when this launch is bracketed, its start/stop as the timing pair: hipExtLaunchKernelGGL attaches
    // both to the dispatch packet, so no event packet sits between two dense passes that run back to back
    // (an event record of its own cost the pipelined step 3-7 us per dense pass).
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    const bool timed = ctx->ev_pair(kEvDense, &ev_start, &ev_stop);
    if (!timed) {
        if (!ctx->dense_done) SPSP_HIP(hipEventCreateWithFlags(&ctx->dense_done, hipEventDisableTiming));
        ev_stop = ctx->dense_done;
    }
    if (*lists) {
        const WaveLists L{ctx->wave_hits.as<Hit>(), ctx->wave_cnt.as<uint32_t>(), LP->cap, LP->rows_per_wave};
        if (variant == kDensePair) {
            const size_t lds = (size_t)kPairWaves * kQueueCap * 16;                // + 64 KiB static table
            if (!ctx->attr_pair_set) {
                const void* ks[4] = {reinterpret_cast<const void*>(&k_dense_pair<false, false>), reinterpret_cast<const void*>(&k_dense_pair<true, false>),
                                     reinterpret_cast<const void*>(&k_dense_pair<false, true>), reinterpret_cast<const void*>(&k_dense_pair<true, true>)};
                for (const void* kf : ks) SPSP_HIP(hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->attr_pair_set = true;
            }
#define SPSP_PAIR(MIDV, PK) hipExtLaunchKernelGGL((k_dense_pair<MIDV, PK>), dim3(LP->grid), dim3(64 * kPairWaves), lds, ctx->stream, ev_start, ev_stop, 0, \
                                                  d_bases, n_bases, p->m, p->threshold, ctx->pairtab.as<uint8_t>(), LP->n_rows, L)
            if (packed_in) { if (p->m >= 10) SPSP_PAIR(true, true); else SPSP_PAIR(false, true); }
            else { if (p->m >= 10) SPSP_PAIR(true, false); else SPSP_PAIR(false, false); }
#undef SPSP_PAIR
        } else if (variant == kDenseBloom) {
            const size_t lds = (size_t)kBloomBytes + (size_t)kPairWaves * kQueueCap1 * 8;
            if (!ctx->attr_bloom_set) {
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_bloom<15, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_bloom<13, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_bloom<15, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_bloom<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->attr_bloom_set = true;
            }
#define SPSP_BLOOM(MV, PK) hipExtLaunchKernelGGL((k_dense_bloom<MV, PK>), dim3(LP->grid), dim3(64 * kPairWaves), lds, ctx->stream, ev_start, ev_stop, 0, d_bases, n_bases, \
                                                 p->threshold, ctx->bloom.as<uint32_t>(), LP->n_rows, L)
            if (packed_in) { if (p->m == 15) SPSP_BLOOM(15, true); else SPSP_BLOOM(13, true); }
            else { if (p->m == 15) SPSP_BLOOM(15, false); else SPSP_BLOOM(13, false); }
#undef SPSP_BLOOM
        } else {
            const size_t lds = (size_t)kKey10Bytes + (size_t)kPairWaves * kQueueCap1 * 8;
            if (!ctx->attr_single_set) {
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_single),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->attr_single_set = true;
            }
            hipExtLaunchKernelGGL(k_dense_single, dim3(LP->grid), dim3(64 * kPairWaves), lds, ctx->stream, ev_start, ev_stop, 0, d_bases, n_bases, p->m,
                               p->threshold, ctx->filter.as<uint8_t>(), LP->n_rows, L);
        }
    } else {
        // k_dense_direct stores every bitmap word and tile count of the tiles it covers: nothing to clear
        const uint64_t cap = (uint64_t)ctx->n_cu * 16;
        const uint32_t grid = (uint32_t)(n_tiles < cap ? n_tiles : cap);
        hipExtLaunchKernelGGL(k_dense_direct, dim3(grid), dim3(kThreads), 0, ctx->stream, ev_start, ev_stop, 0, d_bases, n_bases, p->m,
                           p->threshold, n_tiles, ctx->bitmap.as<uint32_t>(), ctx->tile_count.as<uint32_t>());
    }
    SPSP_HIP(hipGetLastError());
    ctx->dense_marker = ev_stop;
    return SPSP_OK;
}
// bitmap form: tile counts -> tile offsets (first of the sparse stages)
static int launch_tile_scan(spsp_ctx* ctx, uint64_t n_tiles, hipStream_t stream) {
    int rc;
    const uint32_t seg_shift = seg_shift_for(n_tiles);
    const uint32_t n_seg = (uint32_t)((n_tiles + (1ull << seg_shift) - 1) >> seg_shift);
    if ((rc = ctx->seg_a.reserve((size_t)n_seg * 8))) return rc;
    hipLaunchKernelGGL(k_scan_segments, dim3(n_seg), dim3(kSegThreads), 0, stream, ctx->tile_count.as<uint32_t>(),
                       ctx->tile_off.as<uint32_t>(), n_tiles, seg_shift, ctx->seg_a.as<uint32_t>());
    SPSP_HIP(hipGetLastError());
    return SPSP_OK;
}

int scan_hits_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                   uint64_t* n_hits) {
    int rc = check_params(p);
    if (rc) return rc;
    if (((uintptr_t)d_bases & 15u) != 0) { set_error("d_bases must be 16-byte aligned"); return SPSP_ERR_ARG; }
    *n_hits = 0;
    if (n_bases < p->m) return SPSP_OK;
    const uint64_t n_tiles = (n_bases + kTilePos - 1) / kTilePos;
    if (n_tiles > 0x7fffffffull) { set_error("input too large for one call"); return SPSP_ERR_OVERFLOW; }
    ListPlan LP{};
    bool lists = false;
    spsp_params pp = *p;
    if ((rc = unpack_if_needed(ctx, &pp, &d_bases, n_bases, false))) return rc;
    p = &pp;
    if ((rc = launch_dense(ctx, p, d_bases, n_bases, n_tiles, false, &LP, &lists))) return rc;
    if (lists) {
        hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(1024), 0, ctx->stream, ctx->wave_cnt.as<uint32_t>(), LP.n_lists,
                           ctx->h_scalar + 0);
    } else {
        if ((rc = launch_tile_scan(ctx, n_tiles, ctx->stream))) return rc;
        const uint32_t seg_shift = seg_shift_for(n_tiles);
        const uint32_t n_seg = (uint32_t)((n_tiles + (1ull << seg_shift) - 1) >> seg_shift);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(64), 0, ctx->stream, ctx->seg_a.as<uint32_t>(),
                           ctx->seg_a.as<uint32_t>() + n_seg, n_seg, ctx->h_scalar + 0, ctx->d_scalar.as<uint32_t>() + 0);
    }
    SPSP_HIP(hipGetLastError());
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    *n_hits = ctx->h_scalar[0];
    return SPSP_OK;
}

// Whole scan on the stream with ONE host synchronisation at the end: the sparse
// stages are launched with grids sized by buffer capacity and read the live counts
// from device memory.  Capacities start from the expected hit density and grow (and
// the affected stages re-run) in the rare call that overflows them.
//
// The job is split at that synchronisation: scan_begin queues one attempt and returns,
// scan_end waits, inspects the counts and -- only on overflow -- queues again.  Between
// the two the host is free to queue other work (another stream's comparison, the next
// batch's copy); spsp_scan_device is begin + end.
static int scan_enqueue(spsp_ctx* ctx) {
    ScanJob& J = ctx->scan_job;
    const spsp_params* p = &J.p;
    int rc;
    if (ctx->hits_cap > 0xfffffff0ull || ctx->out_cap > 0xfffffff0ull) {
        set_error("too many selected m-mers / super-k-mers for one call; split the input");
        return SPSP_ERR_OVERFLOW;
    }
    const uint32_t hits_cap = (uint32_t)ctx->hits_cap, out_cap = (uint32_t)ctx->out_cap;
    J.hits_cap = hits_cap; J.out_cap = out_cap;
    if ((rc = ctx->hits.reserve((size_t)hits_cap * sizeof(Hit)))) return rc;
    if ((rc = ctx->emit_count.reserve((size_t)hits_cap * 4))) return rc;
    if ((rc = ctx->scan_tmp.reserve((size_t)out_cap * sizeof(spsp_superkmer)))) return rc;
    const uint32_t rblocks = (hits_cap + kResolveThreads - 1) / kResolveThreads;
    const uint32_t n_waves = rblocks * (kResolveThreads / 64), n_chunks = (n_waves + 63) / 64;
    const uint32_t n_super = n_chunks > 2048 ? (n_chunks + 63) / 64 : 0u;   // sums of 64 chunks behind the chunk sums: from 8 x 10^6 hits on
    if ((rc = ctx->seg_b.reserve((size_t)(n_chunks + n_super + n_waves) * 4))) return rc;
    uint32_t* chunk_sum = ctx->seg_b.as<uint32_t>();
    uint32_t* wave_sum = chunk_sum + n_chunks + n_super;
    if ((rc = ctx->d_scalar.reserve(64))) return rc;
    uint32_t* d_sc = ctx->d_scalar.as<uint32_t>();
    if (J.redo_from == 0) {
        ListPlan LP{};
        if ((rc = unpack_if_needed(ctx, &J.p, &J.d_bases, J.n_bases, J.use_bitmap))) return rc;
        if ((rc = launch_dense(ctx, p, J.d_bases, J.n_bases, J.n_tiles, J.use_bitmap, &LP, &J.lists))) return rc;
        if (J.lists) { J.n_lists = LP.n_lists; J.list_cap = LP.cap; J.rows_per_wave = LP.rows_per_wave; }
    }
    // the sparse stages may have a stream of their own (spsp_scan_tail_stream): behind this dense pass, beside the next
    const hipStream_t sparse = ctx->sparse_stream();
    if (sparse != ctx->stream && J.redo_from == 0) SPSP_HIP(hipStreamWaitEvent(sparse, ctx->dense_marker, 0));
    if (J.redo_from == 0 && !J.lists && (rc = launch_tile_scan(ctx, J.n_tiles, sparse))) return rc;
    if (J.redo_from <= 1) {
        if (J.lists) {
            const WaveLists L{ctx->wave_hits.as<Hit>(), ctx->wave_cnt.as<uint32_t>(), J.list_cap, J.rows_per_wave};
            hipLaunchKernelGGL(k_compact, dim3((J.n_lists + kCompactWaves - 1) / kCompactWaves), dim3(kCompactThreads), 0, sparse,
                               L, J.n_lists, J.n_bases, p->k, p->m, J.d_rec_off, J.n_rec, ctx->hits.as<Hit>(), hits_cap,
                               ctx->h_scalar + 0, d_sc + 0, chunk_sum, n_chunks);
        } else {
            ctx->h_scalar[2] = 0;   // no lists: nothing can overflow them (no kernel of this attempt writes the slot)
            const uint32_t seg_shift = seg_shift_for(J.n_tiles);
            const uint32_t n_seg_t = (uint32_t)((J.n_tiles + (1ull << seg_shift) - 1) >> seg_shift);
            hipLaunchKernelGGL(k_expand, dim3((uint32_t)((J.n_tiles + kExpandTilesPerWg - 1) / kExpandTilesPerWg)),
                               dim3(kThreads), 0, sparse, J.d_bases, J.n_bases, p->k, p->m, ctx->bitmap.as<uint32_t>(),
                               ctx->tile_count.as<uint32_t>(), ctx->tile_off.as<uint32_t>(), ctx->seg_a.as<uint32_t>(),
                               n_seg_t, seg_shift, J.n_tiles, J.d_rec_off, J.n_rec, ctx->hits.as<Hit>(), hits_cap, ctx->h_scalar + 0, d_sc + 0,
                               chunk_sum, n_chunks);
        }
        SPSP_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_resolve<false>, dim3(rblocks), dim3(kResolveThreads), 0, sparse, ctx->hits.as<Hit>(), d_sc + 0,
                           hits_cap, J.d_rec_off, p->k, p->m, ctx->emit_count.as<uint32_t>(), wave_sum, chunk_sum, n_chunks,
                           (uint64_t*)nullptr, (spsp_superkmer*)nullptr, 0u, n_super);
        SPSP_HIP(hipGetLastError());
        if (n_super) { hipLaunchKernelGGL(k_super_sums, dim3(n_super), dim3(64), 0, sparse, chunk_sum, n_chunks); SPSP_HIP(hipGetLastError()); }
    }
    hipLaunchKernelGGL(k_resolve<true>, dim3(rblocks), dim3(kResolveThreads), 0, sparse, ctx->hits.as<Hit>(), d_sc + 0,
                       hits_cap, J.d_rec_off, p->k, p->m, ctx->emit_count.as<uint32_t>(), wave_sum, chunk_sum, n_chunks,
                       ctx->h_scalar + 1, ctx->scan_tmp.as<spsp_superkmer>(), out_cap, n_super);
    SPSP_HIP(hipGetLastError());
    // what scan_end waits on: this job's last kernel, not the whole stream (a caller may already have queued
    // the next batch's work behind it)
    if (!ctx->scan_done) SPSP_HIP(hipEventCreateWithFlags(&ctx->scan_done, hipEventDisableTiming));
    SPSP_HIP(hipEventRecord(ctx->scan_done, sparse));
    return SPSP_OK;
}

int scan_begin_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                    const uint64_t* d_rec_off, uint32_t n_rec) {
    ScanJob& J = ctx->scan_job;
    if (J.pending) { set_error("a scan is already pending on this context: call spsp_scan_device_end first"); return SPSP_ERR_ARG; }
    int rc = check_params(p);
    if (rc) return rc;
    if (((uintptr_t)d_bases & 15u) != 0) { set_error("d_bases must be 16-byte aligned"); return SPSP_ERR_ARG; }
    J = ScanJob{};
    J.p = *p; J.d_bases = d_bases; J.n_bases = n_bases; J.d_rec_off = d_rec_off; J.n_rec = n_rec;
    J.pending = true;
    J.empty = (n_rec == 0 || n_bases < p->k);
    if (J.empty) return SPSP_OK;
    J.n_tiles = (n_bases + kTilePos - 1) / kTilePos;
    if (J.n_tiles > 0x7fffffffull) { J.pending = false; set_error("input too large for one call"); return SPSP_ERR_OVERFLOW; }
    const double frac = (double)p->threshold / 18446744073709551616.0;
    uint64_t want_hits = (uint64_t)((double)n_bases * frac * 1.25) + 4096;
    // Capacities are per call (the resolve grids are as large as the capacity): what an overflow of an earlier call taught
    // -- hits and super-k-mers per base at this threshold -- is kept as a RATE, not as a size.  (Kept as sizes, one call that
    // selects every m-mer of 500 Mbp left every later call of the context with grids of 4 x 10^6 workgroups: 65 ms each.)
    uint64_t want_out = 0;
    if (ctx->learn_threshold == p->threshold && ctx->learn_valid) {
        want_hits = std::max(want_hits, (uint64_t)((double)n_bases * ctx->learn_hits_per_base * 1.125) + 4096);
        want_out = (uint64_t)((double)n_bases * ctx->learn_out_per_base * 1.125) + 4096;
    }
    if (want_hits > n_bases) want_hits = n_bases;
    ctx->hits_cap = want_hits;
    ctx->out_cap = std::max(want_hits, want_out);
    // test hooks: start from deliberately small buffers so the overflow/retry paths run
    static const char* dbg_hits = getenv("SPSP_DEBUG_HITS_CAP");
    static const char* dbg_out = getenv("SPSP_DEBUG_OUT_CAP");
    if (dbg_hits) ctx->hits_cap = (uint64_t)atoll(dbg_hits);
    if (dbg_out) ctx->out_cap = (uint64_t)atoll(dbg_out);
    if (dbg_hits) ctx->list_cap = 0;
    if (ctx->list_cap_threshold != p->threshold) ctx->list_cap = 0;
    J.redo_from = 0;
    J.use_bitmap = false;
    // a threshold that selects (nearly) every m-mer: the scan by segments (spsp_stats.hip::k_seg_scan) -- every position a
    // 32-byte hit record through compact and resolve was 245 ms per 500 Mbp at -s 1.  From -s 1.25 on (-s 2: 3.4 ms) the
    // dense + sparse passes are faster again.
    static const char* dbg_seg = getenv("SPSP_DEBUG_SEG_SCAN");       // "0": never, "1": whatever the threshold (A/B, tests)
    J.segments = (dbg_seg ? dbg_seg[0] == '1' : frac >= 0.8) && n_bases < (1ull << 32);   // (32-bit places for the emitted super-k-mers)
    if (J.segments) {
        if ((rc = ctx->ev_begin(kEvScan))) { J.pending = false; return rc; }
        rc = seg_scan_count(ctx, p, d_bases, (p->flags & SPSP_SCAN_PACKED_INPUT) != 0, n_bases, d_rec_off, n_rec);
        if (!rc) {
            if (!ctx->scan_done && hipEventCreateWithFlags(&ctx->scan_done, hipEventDisableTiming) != hipSuccess) rc = SPSP_ERR_HIP;
            if (!rc && hipEventRecord(ctx->scan_done, ctx->stream) != hipSuccess) rc = SPSP_ERR_HIP;
        }
        const int rc2 = ctx->ev_end(kEvScan);
        if (rc || rc2) J.pending = false;
        return rc ? rc : rc2;
    }
    if ((rc = ctx->ev_begin(kEvScan))) { J.pending = false; return rc; }
    rc = scan_enqueue(ctx);
    const int rc2 = ctx->ev_end(kEvScan);   // brackets the first attempt (a retry after an overflow is not timed)
    if (rc || rc2) J.pending = false;
    return rc ? rc : rc2;
}

int scan_end_impl(spsp_ctx* ctx, spsp_superkmer** d_out, uint64_t* n_out) {
    ScanJob& J = ctx->scan_job;
    *d_out = nullptr; *n_out = 0;
    if (!J.pending) { set_error("no scan is pending on this context"); return SPSP_ERR_ARG; }
    J.pending = false;
    if (J.empty) return SPSP_OK;
    static const char* dbg_out = getenv("SPSP_DEBUG_OUT_CAP");
    static const char* dbg_budget = getenv("SPSP_DEBUG_LIST_BUDGET");   // test hook: bytes the grown hit lists may take
    int rc;
    if (J.segments) {
        SPSP_HIP(hipEventSynchronize(ctx->scan_done));
        const uint64_t n_em = (uint32_t)ctx->h_scalar[0];
        const uint32_t left_halo = (uint32_t)ctx->h_scalar[1];
        if (left_halo == 0) {
            if (n_em == 0) return SPSP_OK;
            if ((rc = ctx->scan_tmp.reserve((size_t)n_em * sizeof(spsp_superkmer) + 64))) return rc;
            if ((rc = seg_scan_emit(ctx, &J.p, J.d_bases, (J.p.flags & SPSP_SCAN_PACKED_INPUT) != 0, J.n_bases, J.d_rec_off, J.n_rec,
                                    ctx->scan_tmp.as<spsp_superkmer>(), n_em))) return rc;
            SPSP_HIP(hipStreamSynchronize(ctx->stream));
            *d_out = ctx->scan_tmp.as<spsp_superkmer>();
            *n_out = n_em;
            return SPSP_OK;
        }
        // a chain left its tile's halo (a long run without a reset: a homopolymer, a short-period repeat): the product scan
        J.segments = false;
        if ((rc = scan_enqueue(ctx))) return rc;
    }
    for (int attempt = 0; attempt < 5; ++attempt) {
        SPSP_HIP(hipEventSynchronize(ctx->scan_done));
        const uint64_t n_hits = ctx->h_scalar[0], n_em = ctx->h_scalar[1], fullest = J.lists ? ctx->h_scalar[2] : 0;
        if (fullest > J.list_cap) {
            // a wave found more hits than its list holds (it kept counting): run the dense pass again with lists
            // that fit -- unless they would be out of proportion (an input that is nearly all hits in places),
            // where the bitmap form of the dense pass takes over
            const uint64_t want = fullest + fullest / 8 + 16;
            const uint64_t budget = dbg_budget ? (uint64_t)atoll(dbg_budget) : std::max<uint64_t>(256ull << 20, 8 * J.n_bases);
            if (want * J.n_lists * sizeof(Hit) > budget) J.use_bitmap = true;
            else { ctx->list_cap = want; ctx->list_cap_threshold = J.p.threshold; }
            if (n_hits > J.hits_cap) ctx->hits_cap = n_hits + n_hits / 8 + 1024;   // (clamped counts: a lower bound)
            if (ctx->out_cap < ctx->hits_cap && !dbg_out) ctx->out_cap = ctx->hits_cap;
            J.redo_from = 0;
        } else if (n_hits > J.hits_cap) {  // room for every hit; the lists are intact, a consumed bitmap is not
            ctx->hits_cap = n_hits + n_hits / 8 + 1024;
            if (ctx->out_cap < ctx->hits_cap && !dbg_out) ctx->out_cap = ctx->hits_cap;
            J.redo_from = J.lists ? 1 : 0;
        } else if (n_em > J.out_cap) {     // hits and offsets are intact: only the write pass repeats
            ctx->out_cap = n_em + n_em / 8 + 1024;
            J.redo_from = 2;
        } else {
            *d_out = n_em ? ctx->scan_tmp.as<spsp_superkmer>() : nullptr;
            *n_out = n_em;
            if (attempt > 0) {                     // an overflow happened: the next call at this threshold starts with room
                ctx->learn_threshold = J.p.threshold; ctx->learn_valid = true;
                ctx->learn_hits_per_base = (double)n_hits / (double)J.n_bases;
                ctx->learn_out_per_base = (double)n_em / (double)J.n_bases;
            }
            return SPSP_OK;
        }
        if (attempt == 4) break;
        if ((rc = scan_enqueue(ctx))) return rc;
    }
    set_error("scan buffers kept overflowing");
    return SPSP_ERR_OVERFLOW;
}

int scan_device_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                     const uint64_t* d_rec_off, uint32_t n_rec, spsp_superkmer** d_out, uint64_t* n_out) {
    *d_out = nullptr; *n_out = 0;
    int rc = scan_begin_impl(ctx, p, d_bases, n_bases, d_rec_off, n_rec);
    if (rc) return rc;
    return scan_end_impl(ctx, d_out, n_out);
}

}  // namespace spsp
