// spsp_bigkeys.hip -- the comparator's keys of a genome (or sketch) of ANY size, on the device.
//
// The reference's k-mer index is unbounded: minimizer_map is a std::map of hash maps that grow with the genome
// (SubSampler.h:62, SubSampler.cpp:274-300, `count++` on a uint8), and the comparator's color_map takes whatever a
// bucket holds (Comparator.cpp:186-260).  The per-genome LDS forms of spsp_keys.hip / spsp_decode.hip hold 4096-8192
// k-mers; what does not fit goes through the two stages of this file, with no host code in between:
//
//   dedupe   k_big_insert   every raw record (minimizer | orientation << 31, canonical k-mer) of a flagged segment claims
//                           or finds the slot of its (segment, key) in ONE open-addressing table in HBM: a CAS per record,
//                           FULL keys compared against the claiming record, the occurrences counted in the slot word
//            k_big_emit     the claimer of every group whose count passes handle_superkmer's uint8 rule emits the key,
//                           unless the other orientation's group does (same rule as k_keys_fused, spsp_keys.hip);
//                           places inside the segment's output slice come from ONE atomic per workgroup (its lanes' counts
//                           are summed in LDS first)
//   sort     k_bigsort_chunks / k_bigsort_merge   (callers that promise sorted sketches) bitonic sort of 2048-key chunks
//                           in LDS, then merge passes over global memory, merge-path partitioned: one 2048-key output
//                           tile per workgroup, both inputs staged in LDS, 8 keys per lane
//
// The table is never cleared between calls: a slot word carries the epoch (an 8-bit call counter) of the call that
// claimed it, and a slot of another epoch is free.  The dedupe kernels are launched over ALL raw places of a call and
// find their segment by a search on the segment starts; a gate word (set by the LDS form when it meets a segment it
// cannot hold) lets every workgroup leave at once in the common case that nothing is flagged.
#include <algorithm>
#include <cstring>
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

constexpr uint32_t kBigThreads = 256, kBigPer = 8;                 // places per workgroup = kBigThreads * kBigPer
constexpr uint32_t kBigTile = 2048, kBigSortThreads = 1024, kBigMergeThreads = 256, kBigVT = kBigTile / kBigMergeThreads;

__device__ __forceinline__ uint64_t big_mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
// 64 hash bits of (segment, key): the home slot comes from bits 16.., the 8-bit fingerprint kept in the slot word from the top
__device__ __forceinline__ uint64_t big_hash(uint32_t seg, uint32_t mo, uint64_t lo, uint64_t hi) {
    uint64_t h = big_mix(lo ^ 0x9E3779B97F4A7C15ULL);
    h = big_mix(h + (uint64_t)mo * 0xD6E8FEB86659FD93ULL + (uint64_t)seg * 0xA0761D6478BD642FULL);
    return big_mix(h ^ hi);
}
__device__ __forceinline__ uint32_t big_home(uint64_t h, uint32_t mask) { return (uint32_t)(h >> 16) & mask; }
__device__ __forceinline__ uint32_t big_fp(uint64_t h) { return (uint32_t)(h >> 56); }

// segment of raw place p: the last s with seg_first[s] <= p (segments with no place share their start with the next one)
__device__ __forceinline__ uint32_t big_segment(const uint32_t* __restrict__ seg_first, uint32_t n_seg, uint32_t p) {
    uint32_t lo = 0, hi = n_seg;                                   // answer in [lo, hi)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_first[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

// slot word: [63:48] occurrences (wraps at 2^16; the uint8 rule reads it mod 256), [47:40] epoch, [39:32] fingerprint of the
// claimer's key, [31:0] claiming record + 1.  The fingerprint only SKIPS comparisons (a probe that meets another key's slot --
// three in ten at the table's load -- read that key's record, three random lines, to learn that it differs: a third of both
// kernels' time at 4 x 10^7 keys); equal fingerprints are followed by the full comparison as before, so nothing is decided by it.
__device__ __forceinline__ uint32_t slot_epoch(unsigned long long wd) { return (uint32_t)(wd >> 40) & 0xffu; }
__device__ __forceinline__ uint32_t slot_fp(unsigned long long wd) { return (uint32_t)(wd >> 32) & 0xffu; }

template <bool HAS_HI>
__global__ __launch_bounds__(kBigThreads) void k_big_insert(const uint32_t* __restrict__ raw_mn, const uint64_t* __restrict__ raw_lo,
                                                           const uint64_t* __restrict__ raw_hi, const uint32_t* __restrict__ seg_first,
                                                           const uint32_t* __restrict__ seg_cnt, const uint32_t* __restrict__ seg_big, uint32_t n_seg,
                                                           uint32_t n_places, const uint32_t* __restrict__ gate, unsigned long long* __restrict__ slot,
                                                           uint32_t mask, uint32_t epoch) {
    if (gate && *gate == 0) return;
    uint32_t s = 0xffffffffu, first = 0, cnt = 0, big = 0;
#pragma unroll 1
    for (uint32_t u = 0; u < kBigPer; ++u) {
        const uint32_t p = (blockIdx.x * kBigPer + u) * kBigThreads + threadIdx.x;
        if (p >= n_places) break;
        if (s == 0xffffffffu || p - first >= cnt) {
            s = big_segment(seg_first, n_seg, p);
            first = seg_first[s]; cnt = seg_cnt[s]; big = seg_big[s];
        }
        if (!big || p - first >= cnt) continue;
        const uint32_t mo = raw_mn[p];
        if (mo == 0xffffffffu) continue;                           // a place without a k-mer
        const uint64_t lo = raw_lo[p], hi = HAS_HI ? raw_hi[p] : 0ull;
        // (a claim carries its own occurrence: one atomic per distinct key instead of a CAS and an add on the same line --
        // a metagenome's keys are nearly all distinct, and the table's atomics are what this kernel's time is)
        const uint64_t hh = big_hash(s, mo, lo, hi);
        const uint32_t fp = big_fp(hh);
        const unsigned long long mine = (1ull << 48) | ((unsigned long long)epoch << 40) | ((unsigned long long)fp << 32) | (unsigned long long)(p + 1);
        uint32_t h = big_home(hh, mask);
        unsigned long long cur = slot[h];
        bool claimed = false;
        for (;;) {                                                 // ends: at least twice as many slots as records
            if (slot_epoch(cur) != epoch) {                        // free (an older call's word, or never used)
                const unsigned long long prev = atomicCAS(&slot[h], cur, mine);
                if (prev == cur) { claimed = true; break; }
                cur = prev;                                        // somebody was faster (or the plain load was stale): look at what is there
                continue;
            }
            const uint32_t c = (uint32_t)cur - 1;                  // the claimer's key was written by the kernel before this one
            if (slot_fp(cur) == fp && c - first < cnt && raw_lo[c] == lo && raw_mn[c] == mo && (!HAS_HI || raw_hi[c] == hi)) break;
            h = (h + 1) & mask;
            cur = slot[h];
        }
        if (!claimed) atomicAdd(&slot[h], 1ull << 48);
    }
}

template <bool HAS_HI>
__global__ __launch_bounds__(kBigThreads) void k_big_emit(const uint32_t* __restrict__ raw_mn, const uint64_t* __restrict__ raw_lo,
                                                         const uint64_t* __restrict__ raw_hi, const uint32_t* __restrict__ seg_first,
                                                         const uint32_t* __restrict__ seg_cnt, const uint32_t* __restrict__ seg_big, uint32_t n_seg,
                                                         uint32_t n_places, const uint32_t* __restrict__ gate, const unsigned long long* __restrict__ slot,
                                                         uint32_t mask, uint32_t epoch, uint32_t abundance, uint32_t* __restrict__ o_mn,
                                                         uint64_t* __restrict__ o_lo, uint64_t* __restrict__ o_hi, uint32_t* __restrict__ distinct) {
    if (gate && *gate == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = 0xffffffffu, first = 0, cnt = 0, big = 0;
    uint32_t emit_mask = 0;                                        // bit u: this lane's u-th place emits its key
    uint32_t s_lo_seen = 0xffffffffu;                              // segment of this lane's first emitting place (segments only grow with u)
    auto usable = [&](unsigned long long wd) { return ((uint32_t)(wd >> 48) & 255u) >= abundance; };   // uint8 count (SubSampler.h:24)
#pragma unroll 1
    for (uint32_t u = 0; u < kBigPer; ++u) {
        const uint32_t p = (blockIdx.x * kBigPer + u) * kBigThreads + threadIdx.x;
        bool emit = false;
        uint32_t mo = 0; uint64_t lo = 0, hi = 0;
        if (p < n_places) {
            if (s == 0xffffffffu || p - first >= cnt) {
                s = big_segment(seg_first, n_seg, p);
                first = seg_first[s]; cnt = seg_cnt[s]; big = seg_big[s];
            }
            if (big && p - first < cnt && (mo = raw_mn[p]) != 0xffffffffu) {
                lo = raw_lo[p]; hi = HAS_HI ? raw_hi[p] : 0ull;
                const uint64_t hh = big_hash(s, mo, lo, hi);
                const uint32_t fp = big_fp(hh);
                uint32_t h = big_home(hh, mask);
                unsigned long long cur;
                for (;;) {                                         // this record's group: k_big_insert left it on this chain
                    cur = slot[h];
                    const uint32_t c = (uint32_t)cur - 1;
                    if (c == p) break;                             // its own claim
                    if (slot_epoch(cur) == epoch && slot_fp(cur) == fp && c - first < cnt && raw_lo[c] == lo && raw_mn[c] == mo && (!HAS_HI || raw_hi[c] == hi)) break;
                    h = (h + 1) & mask;
                }
                if ((uint32_t)cur == p + 1 && usable(cur)) {       // one lane per (key, orientation) group: its claimer
                    emit = true;
                    if (s_lo_seen == 0xffffffffu) s_lo_seen = s;
                    if (mo >> 31) {                                // the forward-oriented group of the same canonical key emits if it is usable
                        const uint32_t sib = mo & 0x7fffffffu;
                        const uint64_t hs = big_hash(s, sib, lo, hi);
                        const uint32_t fps = big_fp(hs);
                        uint32_t h2 = big_home(hs, mask);
                        for (;;) {
                            const unsigned long long c2 = slot[h2];
                            if (slot_epoch(c2) != epoch) break;    // no such group
                            const uint32_t c = (uint32_t)c2 - 1;
                            if (slot_fp(c2) == fps && c - first < cnt && raw_lo[c] == lo && raw_mn[c] == sib && (!HAS_HI || raw_hi[c] == hi)) { if (usable(c2)) emit = false; break; }
                            h2 = (h2 + 1) & mask;
                        }
                    }
                }
            }
        }
        if (emit) emit_mask |= 1u << u;
    }
    // Places in the segment's output slice.  A workgroup's places nearly always lie in ONE segment: then its lanes' counts
    // are summed in LDS and the workgroup draws its room with one atomic (one per wave made k_big_emit wait on 600 000
    // same-address atomics at 4 x 10^7 keys); a workgroup that straddles segments draws per wave and segment.
    __shared__ uint32_t s_base, s_ref;
    const uint32_t p_first = blockIdx.x * kBigPer * kBigThreads;
    if (threadIdx.x == 0) s_ref = big_segment(seg_first, n_seg, p_first < n_places ? p_first : n_places - 1);
    __syncthreads();
    const uint32_t ref = s_ref;
    const bool one_segment = __syncthreads_and(emit_mask == 0 || (s_lo_seen == ref && s == ref)) != 0;
    if (one_segment) {
        // output places in PLACE order (round u, then lane): the stores of one round are consecutive across a wave's lanes.
        // (Lane-major places -- every lane's up to eight keys side by side -- made each store instruction touch 64 different
        // lines: 2.5 of this kernel's 2.8 ms at 4 x 10^7 keys.)
        __shared__ uint32_t s_cnt[kBigPer][kBigThreads / 64];
        const uint32_t wv = threadIdx.x >> 6;
        unsigned long long votes[kBigPer];
#pragma unroll
        for (uint32_t u = 0; u < kBigPer; ++u) {
            votes[u] = __ballot((emit_mask >> u) & 1u);
            if (lane == 0) s_cnt[u][wv] = (uint32_t)__popcll(votes[u]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (uint32_t u = 0; u < kBigPer; ++u)
                for (uint32_t w2 = 0; w2 < kBigThreads / 64; ++w2) { const uint32_t c = s_cnt[u][w2]; s_cnt[u][w2] = tot; tot += c; }
            s_base = tot ? atomicAdd(&distinct[ref], tot) : 0u;
        }
        __syncthreads();
        const uint32_t base = seg_first[ref] + s_base;
#pragma unroll
        for (uint32_t u = 0; u < kBigPer; ++u) {
            if (!((emit_mask >> u) & 1u)) continue;
            const uint32_t p = (blockIdx.x * kBigPer + u) * kBigThreads + threadIdx.x;
            const uint32_t at = base + s_cnt[u][wv] + (uint32_t)__popcll(votes[u] & ((1ull << lane) - 1ull));
            o_mn[at] = raw_mn[p] & 0x7fffffffu; o_lo[at] = raw_lo[p];
            if (HAS_HI) o_hi[at] = raw_hi[p];
        }
        return;
    }
#pragma unroll 1
    for (uint32_t u = 0; u < kBigPer; ++u) {
        const uint32_t p = (blockIdx.x * kBigPer + u) * kBigThreads + threadIdx.x;
        const bool emit = (emit_mask >> u) & 1u;
        uint32_t sl_mine = 0, first_mine = 0;
        if (emit) { sl_mine = big_segment(seg_first, n_seg, p); first_mine = seg_first[sl_mine]; }
        unsigned long long todo = __ballot(emit);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t sl = __shfl(sl_mine, leader);
            const unsigned long long same = __ballot(emit && sl_mine == sl);
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(&distinct[sl], (uint32_t)__popcll(same));
            base = __shfl(base, leader);
            if (emit && sl_mine == sl) {
                const uint32_t at = first_mine + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                o_mn[at] = raw_mn[p] & 0x7fffffffu; o_lo[at] = raw_lo[p];
                if (HAS_HI) o_hi[at] = raw_hi[p];
            }
            todo &= ~same;
        }
    }
}

// ------------------------------------------------------------------ sort --
struct BigTileDesc { uint32_t off, n, tile, pad; };                // segment start, segment length, tile number inside the segment

template <bool HAS_HI>
__global__ __launch_bounds__(kBigSortThreads) void k_bigsort_chunks(const BigTileDesc* __restrict__ tiles, const uint32_t* __restrict__ i_mn,
                                                                   const uint64_t* __restrict__ i_lo, const uint64_t* __restrict__ i_hi,
                                                                   uint32_t* __restrict__ o_mn, uint64_t* __restrict__ o_lo, uint64_t* __restrict__ o_hi) {
    __shared__ uint64_t s_lo[kBigTile];
    __shared__ uint64_t s_hi[HAS_HI ? kBigTile : 1];
    __shared__ uint32_t s_mn[kBigTile];
    const BigTileDesc T = tiles[blockIdx.x];
    const uint32_t t = threadIdx.x, t0 = T.tile * kBigTile, n = T.n - t0 < kBigTile ? T.n - t0 : kBigTile;
    const uint32_t g0 = T.off + t0;
    for (uint32_t i = t; i < kBigTile; i += kBigSortThreads) {
        if (i < n) { s_mn[i] = i_mn[g0 + i]; s_lo[i] = i_lo[g0 + i]; if (HAS_HI) s_hi[i] = i_hi[g0 + i]; }
        else { s_mn[i] = 0xffffffffu; s_lo[i] = ~0ull; if (HAS_HI) s_hi[i] = ~0ull; }
    }
    __syncthreads();
    auto greater = [&](uint32_t a, uint32_t b) {
        if (s_mn[a] != s_mn[b]) return s_mn[a] > s_mn[b];
        if (HAS_HI && s_hi[a] != s_hi[b]) return s_hi[a] > s_hi[b];
        return s_lo[a] > s_lo[b];
    };
    for (uint32_t size = 2; size <= kBigTile; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const uint32_t idx = t;                                // kBigTile / 2 comparators, one per thread
            const uint32_t i = ((idx / stride) * (stride << 1)) + (idx % stride), j = i + stride;
            const bool asc = (i & size) == 0;
            if (greater(i, j) == asc) {
                const uint32_t tm = s_mn[i]; s_mn[i] = s_mn[j]; s_mn[j] = tm;
                const uint64_t tl = s_lo[i]; s_lo[i] = s_lo[j]; s_lo[j] = tl;
                if (HAS_HI) { const uint64_t th = s_hi[i]; s_hi[i] = s_hi[j]; s_hi[j] = th; }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = t; i < n; i += kBigSortThreads) { o_mn[g0 + i] = s_mn[i]; o_lo[g0 + i] = s_lo[i]; if (HAS_HI) o_hi[g0 + i] = s_hi[i]; }
}

// one merge pass: runs of `run` keys (a multiple of the tile) are merged in pairs; a workgroup makes one output tile
template <bool HAS_HI>
__global__ __launch_bounds__(kBigMergeThreads) void k_bigsort_merge(const BigTileDesc* __restrict__ tiles, uint32_t run, const uint32_t* __restrict__ i_mn,
                                                                   const uint64_t* __restrict__ i_lo, const uint64_t* __restrict__ i_hi,
                                                                   uint32_t* __restrict__ o_mn, uint64_t* __restrict__ o_lo, uint64_t* __restrict__ o_hi) {
    __shared__ uint64_t s_lo[kBigTile];
    __shared__ uint64_t s_hi[HAS_HI ? kBigTile : 1];
    __shared__ uint32_t s_mn[kBigTile];
    __shared__ uint32_t s_split[2];
    const BigTileDesc T = tiles[blockIdx.x];
    const uint32_t t = threadIdx.x, t0 = T.tile * kBigTile;
    const uint32_t pair0 = (t0 / (2 * run)) * (2 * run);           // (run is a multiple of the tile: a tile lies inside one pair)
    const uint32_t rest = T.n - pair0;
    const uint32_t lenA = rest < run ? rest : run, lenB = rest - lenA < run ? rest - lenA : run;
    const uint32_t A = T.off + pair0, B = A + lenA;                // both runs in the input arrays
    const uint32_t d0 = t0 - pair0, d1 = d0 + kBigTile < lenA + lenB ? d0 + kBigTile : lenA + lenB;
    auto g_le = [&](uint32_t a, uint32_t b) {                      // input[a] <= input[b]
        const uint32_t ma = i_mn[a], mb = i_mn[b];
        if (ma != mb) return ma < mb;
        if (HAS_HI) { const uint64_t ha = i_hi[a], hb = i_hi[b]; if (ha != hb) return ha < hb; }
        return i_lo[a] <= i_lo[b];
    };
    if (t < 2) {                                                   // how many keys of run A are among the first d outputs of the pair
        const uint32_t d = t == 0 ? d0 : d1;
        uint32_t lo = d > lenB ? d - lenB : 0u, hi = d < lenA ? d : lenA;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (g_le(A + mid, B + (d - 1 - mid))) lo = mid + 1; else hi = mid; }
        s_split[t] = lo;
    }
    __syncthreads();
    const uint32_t a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
    const uint32_t na = a1 - a0, nb = b1 - b0, total = na + nb;
    for (uint32_t i = t; i < total; i += kBigMergeThreads) {
        const uint32_t src = i < na ? A + a0 + i : B + b0 + (i - na);
        s_mn[i] = i_mn[src]; s_lo[i] = i_lo[src]; if (HAS_HI) s_hi[i] = i_hi[src];
    }
    __syncthreads();
    auto s_le = [&](uint32_t a, uint32_t b) {
        if (s_mn[a] != s_mn[b]) return s_mn[a] < s_mn[b];
        if (HAS_HI && s_hi[a] != s_hi[b]) return s_hi[a] < s_hi[b];
        return s_lo[a] <= s_lo[b];
    };
    const uint32_t d = t * kBigVT < total ? t * kBigVT : total;
    uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_le(mid, na + (d - 1 - mid))) lo = mid + 1; else hi = mid; }
    uint32_t ai = lo, bi = d - lo;
    const uint32_t out0 = T.off + t0 + d;
#pragma unroll
    for (uint32_t v = 0; v < kBigVT; ++v) {
        if (d + v >= total) break;
        const bool take_a = bi >= nb || (ai < na && s_le(ai, na + bi));
        const uint32_t src = take_a ? ai++ : na + bi++;
        o_mn[out0 + v] = s_mn[src]; o_lo[out0 + v] = s_lo[src]; if (HAS_HI) o_hi[out0 + v] = s_hi[src];
    }
}

// ------------------------------------------------------------------ host side --
int big_dedupe_launch(spsp_ctx* ctx, bool has_hi, const uint32_t* raw_mn, const uint64_t* raw_lo, const uint64_t* raw_hi,
                      const uint32_t* d_seg_first, const uint32_t* d_seg_cnt, const uint32_t* d_seg_big, uint32_t n_seg, uint64_t n_places,
                      const uint32_t* d_gate, uint32_t abundance, uint32_t* out_mn, uint64_t* out_lo, uint64_t* out_hi, uint32_t* d_distinct) {
    if (n_places == 0 || n_seg == 0) return SPSP_OK;
    if (n_places > 0x7ffffff0ull) { set_error("too many k-mer places for one call"); return SPSP_ERR_OVERFLOW; }
    uint64_t slots = 1024;
    while (slots < 2 * n_places) slots <<= 1;
    int rc;
    if (ctx->b_table.cap < slots * 8 || !ctx->b_table.p) {
        if ((rc = ctx->b_table.reserve((size_t)slots * 8))) return rc;
        ctx->big_epoch = 0;                                        // fresh memory: no word of it means anything
    }
    if (ctx->big_epoch == 0 || ctx->big_epoch == 0xffu) {          // first use, or the 8-bit epoch wraps: every word becomes "free"
        SPSP_HIP(hipMemsetAsync(ctx->b_table.p, 0, ctx->b_table.cap, ctx->stream));
        ctx->big_epoch = 0;
    }
    const uint32_t epoch = ++ctx->big_epoch;
    const uint32_t blocks = (uint32_t)((n_places + (uint64_t)kBigThreads * kBigPer - 1) / ((uint64_t)kBigThreads * kBigPer));
    unsigned long long* slot = ctx->b_table.as<unsigned long long>();
    const uint32_t mask = (uint32_t)(slots - 1);
    if (has_hi) {
        hipLaunchKernelGGL(k_big_insert<true>, dim3(blocks), dim3(kBigThreads), 0, ctx->stream, raw_mn, raw_lo, raw_hi, d_seg_first, d_seg_cnt, d_seg_big, n_seg,
                           (uint32_t)n_places, d_gate, slot, mask, epoch);
        hipLaunchKernelGGL(k_big_emit<true>, dim3(blocks), dim3(kBigThreads), 0, ctx->stream, raw_mn, raw_lo, raw_hi, d_seg_first, d_seg_cnt, d_seg_big, n_seg,
                           (uint32_t)n_places, d_gate, (const unsigned long long*)slot, mask, epoch, abundance, out_mn, out_lo, out_hi, d_distinct);
    } else {
        hipLaunchKernelGGL(k_big_insert<false>, dim3(blocks), dim3(kBigThreads), 0, ctx->stream, raw_mn, raw_lo, (const uint64_t*)nullptr, d_seg_first, d_seg_cnt,
                           d_seg_big, n_seg, (uint32_t)n_places, d_gate, slot, mask, epoch);
        hipLaunchKernelGGL(k_big_emit<false>, dim3(blocks), dim3(kBigThreads), 0, ctx->stream, raw_mn, raw_lo, (const uint64_t*)nullptr, d_seg_first, d_seg_cnt,
                           d_seg_big, n_seg, (uint32_t)n_places, d_gate, (const unsigned long long*)slot, mask, epoch, abundance, out_mn, out_lo,
                           (uint64_t*)nullptr, d_distinct);
    }
    SPSP_HIP(hipGetLastError());
    return SPSP_OK;
}

// sorts every segment [off, off + n) of (mn, lo, hi) by (minimizer, kmer_hi, kmer_lo) in place; (t_mn, t_lo, t_hi) is scratch
// of the same extent.  Queued on the context's stream; the tile list is uploaded from a vector that must outlive the
// copy, so the call synchronises the stream once before it returns.
int big_sort_segments(spsp_ctx* ctx, bool has_hi, uint32_t* mn, uint64_t* lo, uint64_t* hi, uint32_t* t_mn, uint64_t* t_lo, uint64_t* t_hi,
                      const std::vector<std::pair<uint32_t, uint32_t>>& segs) {
    std::vector<BigTileDesc> tiles;
    uint32_t longest = 0;
    for (auto& sg : segs) {
        if (sg.second == 0) continue;
        longest = std::max(longest, sg.second);
        for (uint32_t t = 0; t * (uint64_t)kBigTile < sg.second; ++t) tiles.push_back(BigTileDesc{sg.first, sg.second, t, 0});
    }
    if (tiles.empty()) return SPSP_OK;
    int rc;
    if ((rc = ctx->b_tiles.reserve(tiles.size() * sizeof(BigTileDesc)))) return rc;
    SPSP_HIP(hipMemcpyAsync(ctx->b_tiles.p, tiles.data(), tiles.size() * sizeof(BigTileDesc), hipMemcpyHostToDevice, ctx->stream));
    uint32_t passes = 0;
    for (uint64_t run = kBigTile; run < longest; run <<= 1) ++passes;
    const BigTileDesc* d_tiles = ctx->b_tiles.as<BigTileDesc>();
    const dim3 grid((uint32_t)tiles.size());
    // the chunk sort writes where an even number of merge passes brings the keys home
    uint32_t *s_mn = mn, *d_mn = (passes & 1) ? t_mn : mn;
    uint64_t *s_lo = lo, *d_lo = (passes & 1) ? t_lo : lo, *s_hi = hi, *d_hi = (passes & 1) ? t_hi : hi;
    if (has_hi) hipLaunchKernelGGL(k_bigsort_chunks<true>, grid, dim3(kBigSortThreads), 0, ctx->stream, d_tiles, s_mn, s_lo, s_hi, d_mn, d_lo, d_hi);
    else hipLaunchKernelGGL(k_bigsort_chunks<false>, grid, dim3(kBigSortThreads), 0, ctx->stream, d_tiles, s_mn, s_lo, (const uint64_t*)nullptr, d_mn, d_lo, (uint64_t*)nullptr);
    uint64_t run = kBigTile;
    for (uint32_t ps = 0; ps < passes; ++ps, run <<= 1) {
        s_mn = d_mn; s_lo = d_lo; s_hi = d_hi;
        d_mn = s_mn == mn ? t_mn : mn; d_lo = s_lo == lo ? t_lo : lo; d_hi = s_hi == hi ? t_hi : hi;
        if (has_hi) hipLaunchKernelGGL(k_bigsort_merge<true>, grid, dim3(kBigMergeThreads), 0, ctx->stream, d_tiles, (uint32_t)run, s_mn, s_lo, s_hi, d_mn, d_lo, d_hi);
        else hipLaunchKernelGGL(k_bigsort_merge<false>, grid, dim3(kBigMergeThreads), 0, ctx->stream, d_tiles, (uint32_t)run, s_mn, s_lo, (const uint64_t*)nullptr, d_mn, d_lo,
                                (uint64_t*)nullptr);
    }
    SPSP_HIP(hipGetLastError());
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    return SPSP_OK;
}

}  // namespace spsp
