// spsp_build.hip -- the sketch builder on the device (round 5; SURVEY.md 8a A7 / A8, VERDICT r4 "missing" 3).
//
// What the reference does per file behind its scan -- handle_superkmer (SubSampler.cpp:243-302: every k-mer of every selected
// super-k-mer, oriented so that the minimizer reads canonically, indexed per minimizer with a uint8 count and the place of
// the minimizer inside it), then the emission (:458-504) that walks every bucket with find_first_kmer / find_next /
// reconstruct_superkmer (:512-620: greedy, in INSERTION order, neighbours tried A, T, C, G, `seen` marks) and writes maximal
// super-k-mers 2-bit packed (strCompressor, utils.cpp:48-68) and the others as "prefix\nsuffix\n" lines -- was the host's
// share of the sketching path (spsp_host.cpp: sketch_build_core): 0.45 ms per 5 Mbp genome and thread, 0.30 s of the 0.57 s a
// 4 Gbp metagenome file takes.  Here, for ALL files of a batch at once:
//
//   k_bld_keys        (file, minimizer) key of every super-k-mer                                  -> stable radix sort
//   k_rs_hist / k_rs_scatter   LSD radix sort, 8 bits a pass, stable: super-k-mers grouped by bucket = (file, minimizer),
//                     stream order kept inside a bucket (the order handle_superkmer inserts in)
//   k_bld_counts      k-mer places of every sorted super-k-mer, bucket starts                     -> scans
//   k_bld_places      one lane per k-mer PLACE: the oriented k-mer (a shift out of the bases), where the minimizer sits in it
//   k_bld_insert      one open-addressing table in HBM over (bucket, k-mer): occurrences counted, the FIRST place kept
//                     (insertion order: places are numbered bucket by bucket in stream order)
//   k_bld_walk        one lane per bucket: the literal emission loop over the bucket's places -- cursor in insertion order,
//                     left then right extension, usable = unseen and (count mod 256) >= abundance -- codes of maximal
//                     super-k-mers and text of the others into the bucket's scratch
//   k_bld_assemble    [m ASCII][u32 n][blob][lines]["\n\n"] per bucket, buckets back to back per file
//
// The host adds the header line (it has the stream's counts) and gzips.  Buckets never meet (the index is keyed by minimizer,
// the walk stays in its bucket), so a lane per bucket is the parallelism the algorithm has; inside a bucket everything is
// the reference's order.  SPSP_BUILD=host keeps the host builder (the A/B partner and the fallback for what does not fit
// 31-bit place numbers).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

typedef unsigned __int128 u128b;

// ---------------------------------------------------------------------------------------------- radix sort (keys u64, values u32)
constexpr uint32_t kRsThreads = 256, kRsPer = 8, kRsTile = kRsThreads * kRsPer;
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t n_blocks,
                                                       uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRsTile;
    for (uint32_t u = 0; u < kRsPer; ++u) {
        const uint32_t i = base + u * kRsThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];            // digit-major: the scan gives every (digit, block) its place
}
// stable: the elements of a tile leave in their order, digit by digit.  Round u takes elements u * 256 .. u * 256 + 255 of the
// tile: a lane's rank among the lanes of its wave with the same digit comes from eight ballots, the waves add their counts to
// the running per-digit counter one after the other.
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t shift,
                                                          uint32_t n_blocks, const uint32_t* __restrict__ offs, uint64_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t run[256];                                          // where the next element of digit d of this tile goes
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    run[t] = offs[t * n_blocks + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * kRsTile;
    for (uint32_t u = 0; u < kRsPer; ++u) {
        const uint32_t i = base + u * kRsThreads + t;
        const bool live = i < n;
        const uint64_t key = live ? keys[i] : 0ull;
        const uint32_t val = live ? vals[i] : 0u;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const unsigned long long vote = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? vote : ~vote; }
        const uint32_t below = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)), mine = (uint32_t)__popcll(same);
        uint32_t at = 0;
        for (uint32_t w = 0; w < kRsThreads / 64; ++w) {                   // (wave-uniform branch; the barrier is reached by all)
            if (w == wave && live) {
                if (below == 0) { at = run[d]; run[d] = at + mine; }      // the first lane of every digit group of this wave
            }
            __syncthreads();
        }
        // the group's first lane holds the base: the others read it from that lane
        const int first_lane = __ffsll((long long)same) - 1;
        at = __shfl(at, first_lane < 0 ? 0 : first_lane);
        if (live) { keys_out[at + below] = key; vals_out[at + below] = val; }
    }
}

// sorts (keys, vals) by the low `bits` bits of the keys, stable; the result is in (*keys_io, *vals_io) (buffers may swap)
static int radix_sort_pairs(spsp_ctx* ctx, uint64_t** keys_io, uint32_t** vals_io, uint64_t* keys_tmp, uint32_t* vals_tmp, uint32_t n, uint32_t bits) {
    if (n == 0) return SPSP_OK;
    const uint32_t n_blocks = (n + kRsTile - 1) / kRsTile;
    int rc = ctx->bl_hist.reserve((size_t)(256 * n_blocks + 1) * 4 * 2 + 64);
    if (rc) return rc;
    uint32_t* hist = ctx->bl_hist.as<uint32_t>();
    uint32_t* offs = hist + 256 * n_blocks + 1;
    uint64_t* ka = *keys_io; uint64_t* kb = keys_tmp;
    uint32_t* va = *vals_io; uint32_t* vb = vals_tmp;
    for (uint32_t shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3(n_blocks), dim3(kRsThreads), 0, ctx->stream, (const uint64_t*)ka, n, shift, n_blocks, hist);
        if ((rc = launch_scan_u32(ctx, hist, offs, (uint64_t)256 * n_blocks, nullptr))) return rc;
        hipLaunchKernelGGL(k_rs_scatter, dim3(n_blocks), dim3(kRsThreads), 0, ctx->stream, (const uint64_t*)ka, (const uint32_t*)va, n, shift, n_blocks,
                           (const uint32_t*)offs, kb, vb);
        SPSP_HIP(hipGetLastError());
        std::swap(ka, kb); std::swap(va, vb);
    }
    *keys_io = ka; *vals_io = va;
    return SPSP_OK;
}

// ---------------------------------------------------------------------------------------------- keys, counts, places
__global__ __launch_bounds__(256) void k_bld_keys(const spsp_superkmer* __restrict__ sk, uint32_t n_sk, const uint32_t* __restrict__ file_sk, uint32_t n_files,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sk) return;
    uint32_t a = 0, z = n_files;                                           // the last file whose first super-k-mer is <= i
    while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if (file_sk[mid] <= i) a = mid; else z = mid; }
    keys[i] = ((uint64_t)a << 30) | (uint64_t)(sk[i].minimizer & 0x3fffffffu);
    vals[i] = i;
}
// per sorted super-k-mer j: its k-mer places, and whether it opens a bucket
__global__ __launch_bounds__(256) void k_bld_counts(const spsp_superkmer* __restrict__ sk, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                   uint32_t n_sk, uint32_t k, uint32_t* __restrict__ cnt, uint32_t* __restrict__ opens) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sk) return;
    const uint32_t len = sk[perm[j]].len;
    cnt[j] = len >= k ? len - k + 1 : 0u;
    opens[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}
// bucket b: its first sorted super-k-mer (bucket_sk[b]); bucket of sorted super-k-mer j = bucket_of[j] - 1 after the INCLUSIVE use below
__global__ __launch_bounds__(256) void k_bld_buckets(const uint32_t* __restrict__ opens, const uint32_t* __restrict__ open_off, uint32_t n_sk,
                                                    uint32_t* __restrict__ bucket_sk, uint32_t* __restrict__ bucket_of) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sk) return;
    const uint32_t b = open_off[j] + opens[j] - 1;                         // (open_off is exclusive: the buckets opened in front of j)
    bucket_of[j] = b;
    if (opens[j]) bucket_sk[b] = j;
}

__device__ __forceinline__ u128b bld_rc(u128b v, uint32_t k) {             // reverse complement of the k-mer in the low 2k bits
    const u128b top = v << (128u - 2u * k);
    const u128b rcw = ((u128b)rc_window64((uint64_t)top) << 64) | (u128b)rc_window64((uint64_t)(top >> 64));
    return k == 64 ? rcw : rcw & ((((u128b)1) << (2 * k)) - 1);
}

constexpr uint32_t kBldThreads = 256, kBldPer = 4, kBldTile = kBldThreads * kBldPer;
// one lane per place o (places numbered sorted super-k-mer by sorted super-k-mer, window by window in ORIENTED order): the
// oriented k-mer and the first place of the minimizer inside it (kmerstr.find(minimizer), SubSampler.cpp:262-263; 0xff = npos)
__global__ __launch_bounds__(kBldThreads) void k_bld_places(const uint8_t* __restrict__ bases, bool packed, const uint64_t* __restrict__ rec_off,
                                                           const spsp_superkmer* __restrict__ sk, const uint32_t* __restrict__ perm,
                                                           const uint32_t* __restrict__ place_first, const uint32_t* __restrict__ bucket_of, uint32_t n_sk,
                                                           uint32_t k, uint32_t m, uint64_t* __restrict__ kv_lo, uint64_t* __restrict__ kv_hi,
                                                           uint8_t* __restrict__ pmin, uint32_t* __restrict__ place_bucket) {
    __shared__ uint32_t s_first[kBldTile + 1];
    __shared__ uint32_t s_j0;
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t total = place_first[n_sk];
    const uint32_t P0 = blockIdx.x * kBldTile;
    if (P0 >= total) return;
    if (t < 64) {                                                          // the last sorted super-k-mer whose first place is <= P0
        uint32_t lo = 0, hi = n_sk;
        while (hi - lo > 1) {
            const uint32_t span = hi - lo - 1, step = (span + 63) / 64;
            const uint32_t at = lo + (lane + 1) * step;
            const bool le = at < hi && place_first[at] <= P0;
            const uint32_t nle = (uint32_t)__popcll(__ballot(le));
            const uint32_t new_lo = lo + nle * step, next = new_lo + step;
            hi = next < hi ? next : hi;
            lo = new_lo;
        }
        if (lane == 0) s_j0 = lo;
    }
    __syncthreads();
    const uint32_t j0 = s_j0;
    for (uint32_t x = t; x <= kBldTile; x += kBldThreads) s_first[x] = j0 + x <= n_sk ? place_first[j0 + x] : 0xffffffffu;
    __syncthreads();
    const uint32_t* words = reinterpret_cast<const uint32_t*>(bases);
    const uint32_t mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
#pragma unroll 1
    for (uint32_t u = 0; u < kBldPer; ++u) {
        const uint32_t o = P0 + u * kBldThreads + t;
        if (o >= total) break;
        uint32_t x = 0;                                                    // the last staged super-k-mer whose first place is <= o
#pragma unroll
        for (uint32_t step = kBldTile / 2; step; step >>= 1) if (s_first[x + step] <= o) x += step;
        uint32_t j = j0 + x, first = s_first[x];
        spsp_superkmer e = sk[perm[j]];
        uint32_t cnt = e.len >= k ? e.len - k + 1 : 0u;
        if (o - first >= cnt) {                                            // (more than a tile of super-k-mers without a place: searched, not assumed)
            uint32_t lo = j, hi = n_sk;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (place_first[mid] <= o) lo = mid; else hi = mid; }
            j = lo; first = place_first[j]; e = sk[perm[j]]; cnt = e.len >= k ? e.len - k + 1 : 0u;
        }
        const uint32_t tw = o - first;                                     // window of the ORIENTED super-k-mer
        const uint32_t fw = e.rev ? cnt - 1 - tw : tw;                     // the same k-mer's window in the record's direction
        const uint64_t q = rec_off[e.rec] + e.start + fw;
        u128b fwd;
        if (packed) {
            const uint32_t* W = words + (q >> 4);
            const uint32_t sh = 2u * (uint32_t)(q & 15u);
            const u128b top = ((u128b)W[0] << 96) | ((u128b)W[1] << 64) | ((u128b)W[2] << 32) | (u128b)W[3];
            const u128b win = sh ? (top << sh) | ((u128b)W[4] >> (32u - sh)) : top;
            fwd = win >> (128u - 2u * k);
        } else {
            fwd = 0;
            for (uint32_t b = 0; b < k; ++b) fwd = (fwd << 2) | (((uint32_t)bases[q + b] >> 1) & 3u);
        }
        const u128b kv = e.rev ? bld_rc(fwd, k) : fwd;
        uint32_t pm = 0xffu;
        for (uint32_t s2 = 0; s2 + m <= k; ++s2) {                         // first m-mer of the k-mer that reads as the minimizer
            if (((uint32_t)(kv >> (2 * (k - m - s2))) & mmask) == e.minimizer) { pm = s2; break; }
        }
        kv_lo[o] = (uint64_t)kv;
        if (kv_hi) kv_hi[o] = (uint64_t)(kv >> 64);
        pmin[o] = (uint8_t)pm;
        place_bucket[o] = bucket_of[j];
    }
}

// ---------------------------------------------------------------------------------------------- the index
__device__ __forceinline__ uint64_t bld_mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t bld_hash(uint32_t bucket, uint64_t lo, uint64_t hi) {
    uint64_t h = bld_mix(lo ^ 0x9E3779B97F4A7C15ULL);
    h = bld_mix(h + (uint64_t)bucket * 0xD6E8FEB86659FD93ULL);
    return bld_mix(h ^ hi);
}
// slot word: [63:48] occurrences (the uint8 rule reads them mod 256), [47:40] fingerprint, [31:0] a place holding the key + 1;
// first[slot] = the key's FIRST place (insertion order), bit 31 set once the emission has used the k-mer (`seen`)
template <bool HAS_HI>
__global__ __launch_bounds__(256) void k_bld_insert(const uint64_t* __restrict__ kv_lo, const uint64_t* __restrict__ kv_hi, const uint32_t* __restrict__ place_bucket,
                                                   uint32_t n_places, unsigned long long* __restrict__ slot, uint32_t* __restrict__ first, uint32_t mask) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_places) return;
    const uint32_t b = place_bucket[o];
    const uint64_t lo = kv_lo[o], hi = HAS_HI ? kv_hi[o] : 0ull;
    const uint64_t hh = bld_hash(b, lo, hi);
    const uint32_t fp = (uint32_t)(hh >> 56);
    const unsigned long long mine = (1ull << 48) | ((unsigned long long)fp << 40) | (unsigned long long)(o + 1);
    uint32_t h = (uint32_t)(hh >> 16) & mask;
    unsigned long long cur = slot[h];
    bool claimed = false;
    for (;;) {                                                             // ends: twice as many slots as places
        if (cur == 0) {
            const unsigned long long prev = atomicCAS(&slot[h], 0ull, mine);
            if (prev == 0) { claimed = true; break; }
            cur = prev;
            continue;
        }
        const uint32_t c = (uint32_t)cur - 1;
        if (((uint32_t)(cur >> 40) & 255u) == fp && place_bucket[c] == b && kv_lo[c] == lo && (!HAS_HI || kv_hi[c] == hi)) break;
        h = (h + 1) & mask;
        cur = slot[h];
    }
    if (!claimed) atomicAdd(&slot[h], 1ull << 48);
    atomicMin(&first[h], o);
}

// ---------------------------------------------------------------------------------------------- the emission walk
struct BldOut { uint32_t n_codes, n_text, n_entries, n_skm, n_max, pad; };
template <bool HAS_HI>
__global__ __launch_bounds__(64) void k_bld_walk(const uint64_t* __restrict__ kv_lo, const uint64_t* __restrict__ kv_hi, const uint8_t* __restrict__ pmin,
                                                const uint32_t* __restrict__ place_bucket, const uint32_t* __restrict__ place_first,
                                                const uint32_t* __restrict__ bucket_sk, uint32_t n_buckets, uint32_t n_sk,
                                                const unsigned long long* __restrict__ slot, uint32_t* __restrict__ first, uint32_t mask, uint32_t k,
                                                uint32_t m, uint32_t abundance, const uint64_t* __restrict__ bucket_key, uint8_t* __restrict__ codes,
                                                uint8_t* __restrict__ text, uint32_t text_per_place, BldOut* __restrict__ outs) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t j0 = bucket_sk[b], j1 = b + 1 < n_buckets ? bucket_sk[b + 1] : n_sk;
    const uint32_t p0 = place_first[j0], p1 = place_first[j1];
    const uint32_t mn = (uint32_t)bucket_key[j0] & 0x3fffffffu;
    const uint32_t half = k - m, full = 2 * k - m;
    const u128b kmask = k == 64 ? ~(u128b)0 : ((((u128b)1) << (2 * k)) - 1);
    const uint32_t mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    uint8_t* cz = codes + 2ull * p0;                                       // this bucket's scratch: codes of maximal super-k-mers (one a byte) ...
    uint8_t* tx = text + (uint64_t)text_per_place * p0;                    // ... and the lines of the others
    uint32_t n_codes = 0, n_text = 0, n_entries = 0, n_skm = 0, n_max = 0;
    // the slot of k-mer `v` of this bucket, or 0xffffffff; *word = its slot word
    auto find = [&](u128b v, unsigned long long* word) -> uint32_t {
        const uint64_t lo = (uint64_t)v, hi = HAS_HI ? (uint64_t)(v >> 64) : 0ull;
        const uint64_t hh = bld_hash(b, lo, hi);
        const uint32_t fp = (uint32_t)(hh >> 56);
        uint32_t h = (uint32_t)(hh >> 16) & mask;
        for (;;) {
            const unsigned long long cur = slot[h];
            if (cur == 0) return 0xffffffffu;
            const uint32_t c = (uint32_t)cur - 1;
            if (((uint32_t)(cur >> 40) & 255u) == fp && place_bucket[c] == b && kv_lo[c] == lo && (!HAS_HI || kv_hi[c] == hi)) { *word = cur; return h; }
            h = (h + 1) & mask;
        }
    };
    auto usable = [&](uint32_t h, unsigned long long word) { return !(first[h] >> 31) && ((uint32_t)(word >> 48) & 255u) >= abundance; };
    // Subsampler::find_next (SubSampler.cpp:566-602): neighbours tried in A, T, C, G order; the one taken is marked seen
    auto step = [&](u128b from, bool left, u128b* nx_out, uint32_t* base_out) -> bool {
        const uint32_t order[4] = {0u, 2u, 1u, 3u};
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const uint32_t c = order[q];
            const u128b nx = left ? ((from >> 2) | ((u128b)c << (2 * k - 2))) : (((from << 2) | c) & kmask);
            unsigned long long word = 0;
            const uint32_t h = find(nx, &word);
            if (h != 0xffffffffu && usable(h, word)) { first[h] |= 0x80000000u; *nx_out = nx; *base_out = c; return true; }
        }
        return false;
    };
    for (uint32_t p = p0; p < p1; ++p) {                                   // find_first_kmer: the bucket's k-mers in insertion order
        const u128b start = (u128b)kv_lo[p] | (HAS_HI ? (u128b)kv_hi[p] << 64 : (u128b)0);
        unsigned long long word = 0;
        const uint32_t hs = find(start, &word);                            // (always found: this place inserted it)
        if ((first[hs] & 0x7fffffffu) != p) continue;                      // a later occurrence of a k-mer already in the index
        ++n_entries;
        if (!usable(hs, word)) continue;
        first[hs] |= 0x80000000u;
        // reconstruct_superkmer (SubSampler.cpp:512-564): `left` / `right` hold the bases added on either side of the start k-mer
        const uint32_t pm = pmin[p];
        uint64_t n_left = (uint64_t)half - pm, n_right = pm;
        u128b left = 0, right = 0, cur = start;
        uint32_t nl = 0, nr = 0;
        while (nl + k + nr != full) {
            if (n_left != 0) {
                u128b nx; uint32_t c;
                const bool found = step(cur, true, &nx, &c);
                n_left -= 1;
                if (found) { left |= (u128b)c << (2 * nl); ++nl; } else n_left = 0;      // (left: base i of the prefix counted from the start k-mer outwards)
                if (n_left == 0) cur = start; else if (found) cur = nx;
            } else if (n_right != 0) {
                u128b nx; uint32_t c;
                const bool found = step(cur, false, &nx, &c);
                n_right -= 1;
                if (!found) break;
                right = (right << 2) | c; ++nr;
                cur = nx;
            } else break;
        }
        const uint32_t slen = nl + k + nr;
        auto base_at = [&](uint32_t x) -> uint32_t {                       // base x of the reconstructed super-k-mer
            if (x < nl) return (uint32_t)(left >> (2 * (nl - 1 - x))) & 3u;
            if (x < nl + k) return (uint32_t)(start >> (2 * (k - 1 - (x - nl)))) & 3u;
            return (uint32_t)(right >> (2 * (nr - 1 - (x - nl - k)))) & 3u;
        };
        if (slen == full) {
            ++n_max;
            for (uint32_t x = 0; x < half; ++x) cz[n_codes++] = (uint8_t)base_at(x);
            for (uint32_t x = 0; x < half; ++x) cz[n_codes++] = (uint8_t)base_at(k + x);
        } else {
            uint32_t mv = 0, at = slen;                                    // skmer_str.find(minstr): the first m-mer equal to the minimizer
            for (uint32_t x = 0; x < slen; ++x) {
                mv = ((mv << 2) | base_at(x)) & mmask;
                if (x + 1 >= m && mv == mn) { at = x + 1 - m; break; }
            }
            const char nuc[4] = {'A', 'C', 'T', 'G'};
            for (uint32_t x = 0; x < at && x < slen; ++x) tx[n_text++] = (uint8_t)nuc[base_at(x)];
            tx[n_text++] = '\n';
            for (uint32_t x = at + m; x < slen; ++x) tx[n_text++] = (uint8_t)nuc[base_at(x)];
            tx[n_text++] = '\n';
        }
        ++n_skm;
    }
    outs[b] = BldOut{n_codes, n_text, n_entries, n_skm, n_max, 0};
}

__global__ __launch_bounds__(256) void k_bld_sizes(const BldOut* __restrict__ outs, uint32_t n_buckets, uint32_t m, uint32_t* __restrict__ len) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t nc = outs[b].n_codes;
    const uint32_t blob = nc ? 1u + (nc + 3u) / 4u : 0u;                   // strCompressor (utils.cpp:48-68): the count mod 4, then four codes a byte
    len[b] = outs[b].n_entries ? m + 4u + blob + outs[b].n_text + 2u : 0u; // (a bucket without an index entry does not exist)
}
// one wave per bucket: [m ASCII][u32 n][blob][lines]["\n\n"]
__global__ __launch_bounds__(64) void k_bld_assemble(const BldOut* __restrict__ outs, const uint32_t* __restrict__ out_off, const uint32_t* __restrict__ bucket_sk,
                                                    const uint32_t* __restrict__ place_first, const uint64_t* __restrict__ bucket_key, uint32_t n_buckets,
                                                    uint32_t m, const uint8_t* __restrict__ codes, const uint8_t* __restrict__ text, uint32_t text_per_place,
                                                    uint8_t* __restrict__ out) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const BldOut O = outs[b];
    if (!O.n_entries) return;
    const uint32_t j0 = bucket_sk[b], p0 = place_first[j0];
    const uint32_t mn = (uint32_t)bucket_key[j0] & 0x3fffffffu;
    uint8_t* w = out + out_off[b];
    const char nuc[4] = {'A', 'C', 'T', 'G'};
    if (lane < m) w[lane] = (uint8_t)nuc[(mn >> (2 * (m - 1 - lane))) & 3u];
    const uint32_t nc = O.n_codes, blob = nc ? 1u + (nc + 3u) / 4u : 0u;
    if (lane < 4) w[m + lane] = (uint8_t)(blob >> (8 * lane));
    uint8_t* bw = w + m + 4;
    const uint8_t* cz = codes + 2ull * p0;
    if (nc) {
        if (lane == 0) bw[0] = (uint8_t)(nc & 3u);
        for (uint32_t q = lane; q < (nc + 3) / 4; q += 64) {               // the accumulator of strCompressor, byte by byte (H1: it starts at 0)
            uint32_t c = 0;
            const uint32_t have = nc - 4 * q < 4 ? nc - 4 * q : 4;
            for (uint32_t x = 0; x < have; ++x) { c = (c + cz[4 * q + x]) & 255u; if (x + 1 < 4) c = (c << 2) & 255u; }
            bw[1 + q] = (uint8_t)c;
        }
    }
    uint8_t* tw = bw + blob;
    const uint8_t* tx = text + (uint64_t)text_per_place * p0;
    for (uint32_t x = lane; x < O.n_text; x += 64) tw[x] = tx[x];
    if (lane < 2) tw[O.n_text + lane] = '\n';
}
__global__ __launch_bounds__(256) void k_bld_file_stats(const BldOut* __restrict__ outs, const uint64_t* __restrict__ bucket_key, const uint32_t* __restrict__ bucket_sk,
                                                       uint32_t n_buckets, unsigned long long* __restrict__ fstats) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const BldOut O = outs[b];
    const uint32_t f = (uint32_t)(bucket_key[bucket_sk[b]] >> 30);
    if (O.n_entries) atomicAdd(&fstats[4 * f + 0], 1ull);                  // actual_minimizer_number
    atomicAdd(&fstats[4 * f + 1], (unsigned long long)O.n_entries);        // seen_kmers_at_reconstruction
    atomicAdd(&fstats[4 * f + 2], (unsigned long long)O.n_skm);            // seen_superkmers_at_reconstruction
    atomicAdd(&fstats[4 * f + 3], (unsigned long long)O.n_max);            // seen_max_superkmers_at_reconstruction
}

// ---------------------------------------------------------------------------------------------- host side
// The bodies (everything behind the header line) of the sketches of n_files files whose super-k-mers are the ranges
// h_file_sk[f] .. h_file_sk[f + 1] of the scan's stream d_sk; stats: the four counters of the emission per file.
// SPSP_ERR_OVERFLOW: more than 2^31 k-mer places -- the caller takes the host builder.
int sketch_build_device_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, const uint64_t* d_rec_off, const spsp_superkmer* d_sk,
                             uint64_t n_sk64, const uint32_t* h_file_sk, uint32_t n_files, std::vector<std::string>* bodies, std::vector<uint64_t>* file_stats) {
    bodies->assign(n_files, std::string());
    file_stats->assign((size_t)n_files * 4, 0);
    if (n_sk64 == 0 || n_files == 0) return SPSP_OK;
    if (n_sk64 > 0x7ffffff0ull / 64 || n_files > (1u << 30)) { set_error("too many super-k-mers for the device builder"); return SPSP_ERR_OVERFLOW; }
    const uint32_t n_sk = (uint32_t)n_sk64, k = p->k, m = p->m;
    const bool has_hi = k > 32;
    int rc;
    if ((rc = ctx->bl_keys.reserve((size_t)n_sk * 8 * 2 + 64)) || (rc = ctx->bl_vals.reserve((size_t)n_sk * 4 * 2 + 64)) ||
        (rc = ctx->bl_meta.reserve((size_t)(n_sk + 2) * 4 * 6 + (size_t)(n_files + 1) * 4 + (size_t)n_files * 32 + 256))) return rc;
    uint64_t* keys = ctx->bl_keys.as<uint64_t>();
    uint32_t* vals = ctx->bl_vals.as<uint32_t>();
    uint32_t* cnt = ctx->bl_meta.as<uint32_t>();
    uint32_t* place_first = cnt + (n_sk + 2);
    uint32_t* opens = place_first + (n_sk + 2);
    uint32_t* open_off = opens + (n_sk + 2);
    uint32_t* bucket_sk = open_off + (n_sk + 2);
    uint32_t* bucket_of = bucket_sk + (n_sk + 2);
    uint32_t* d_file_sk = bucket_of + (n_sk + 2);
    unsigned long long* d_fstats = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(d_file_sk + n_files + 1) + 7) & ~(uintptr_t)7);
    SPSP_HIP(hipMemcpyAsync(d_file_sk, h_file_sk, (size_t)(n_files + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SPSP_HIP(hipMemsetAsync(d_fstats, 0, (size_t)n_files * 32, ctx->stream));
    const uint32_t gsk = (n_sk + 255) / 256;
    hipLaunchKernelGGL(k_bld_keys, dim3(gsk), dim3(256), 0, ctx->stream, d_sk, n_sk, (const uint32_t*)d_file_sk, n_files, keys, vals);
    uint32_t file_bits = 0;
    while ((1ull << file_bits) < n_files) ++file_bits;
    uint64_t* ks = keys; uint32_t* vs = vals;
    if ((rc = radix_sort_pairs(ctx, &ks, &vs, keys + n_sk, vals + n_sk, n_sk, 30 + file_bits))) return rc;
    if (ks != keys) { keys = ks; vals = vs; }                              // (the sorted arrays: whichever half the last pass wrote)
    hipLaunchKernelGGL(k_bld_counts, dim3(gsk), dim3(256), 0, ctx->stream, d_sk, (const uint64_t*)keys, (const uint32_t*)vals, n_sk, k, cnt, opens);
    if ((rc = launch_scan_u32(ctx, cnt, place_first, n_sk, ctx->h_scalar + 6))) return rc;
    if ((rc = launch_scan_u32(ctx, opens, open_off, n_sk, ctx->h_scalar + 7))) return rc;
    hipLaunchKernelGGL(k_bld_buckets, dim3(gsk), dim3(256), 0, ctx->stream, (const uint32_t*)opens, (const uint32_t*)open_off, n_sk, bucket_sk, bucket_of);
    SPSP_HIP(hipGetLastError());
    SPSP_HIP(hipStreamSynchronize(ctx->stream));                           // the totals size everything behind this point
    const uint64_t n_places = ctx->h_scalar[6], n_buckets = ctx->h_scalar[7];
    if (n_places == 0) return SPSP_OK;
    if (n_places > 0x7ffffff0ull) { set_error("too many k-mer places for the device builder"); return SPSP_ERR_OVERFLOW; }
    uint64_t slots = 1024;
    while (slots < 2 * n_places) slots <<= 1;
    // scratch per place: two codes (a maximal super-k-mer of k - m + 1 k-mers writes 2 (k - m)), k + 2 bytes of lines (a
    // super-k-mer of q k-mers writes at most k + q + 1)
    const uint32_t text_per_place = k + 2;
    if ((rc = ctx->bl_lo.reserve((size_t)n_places * 8 + 64)) || (has_hi && (rc = ctx->bl_hi.reserve((size_t)n_places * 8 + 64))) ||
        (rc = ctx->bl_pmin.reserve((size_t)n_places + 64)) || (rc = ctx->bl_pb.reserve((size_t)n_places * 4 + 64)) ||
        (rc = ctx->bl_slot.reserve((size_t)slots * 8)) || (rc = ctx->bl_first.reserve((size_t)slots * 4)) ||
        (rc = ctx->bl_codes.reserve((size_t)n_places * 2 + 64)) || (rc = ctx->bl_text.reserve((size_t)n_places * text_per_place + 64)) ||
        (rc = ctx->bl_outs.reserve((size_t)n_buckets * sizeof(BldOut) + (size_t)(2 * n_buckets + 2) * 4 + 64))) return rc;
    uint64_t* kv_lo = ctx->bl_lo.as<uint64_t>();
    uint64_t* kv_hi = has_hi ? ctx->bl_hi.as<uint64_t>() : nullptr;
    SPSP_HIP(hipMemsetAsync(ctx->bl_slot.p, 0, (size_t)slots * 8, ctx->stream));
    SPSP_HIP(hipMemsetAsync(ctx->bl_first.p, 0xff, (size_t)slots * 4, ctx->stream));
    hipLaunchKernelGGL(k_bld_places, dim3((uint32_t)((n_places + kBldTile - 1) / kBldTile)), dim3(kBldThreads), 0, ctx->stream, d_bases, packed, d_rec_off, d_sk,
                       (const uint32_t*)vals, (const uint32_t*)place_first, (const uint32_t*)bucket_of, n_sk, k, m, kv_lo, kv_hi, ctx->bl_pmin.as<uint8_t>(),
                       ctx->bl_pb.as<uint32_t>());
    const uint32_t mask = (uint32_t)(slots - 1);
    unsigned long long* slot = ctx->bl_slot.as<unsigned long long>();
    uint32_t* first = ctx->bl_first.as<uint32_t>();
    const uint32_t gpl = (uint32_t)((n_places + 255) / 256);
    BldOut* outs = ctx->bl_outs.as<BldOut>();
    uint32_t* out_len = reinterpret_cast<uint32_t*>(outs + n_buckets);
    uint32_t* out_off = out_len + n_buckets;                               // (n_buckets + 1 entries: the scan's total at the end)
    const uint32_t ab = p->abundance;
    const uint32_t gb = (uint32_t)((n_buckets + 63) / 64);
#define SPSP_BLD(HI) \
    hipLaunchKernelGGL(k_bld_insert<HI>, dim3(gpl), dim3(256), 0, ctx->stream, (const uint64_t*)kv_lo, (const uint64_t*)kv_hi, (const uint32_t*)ctx->bl_pb.as<uint32_t>(), \
                       (uint32_t)n_places, slot, first, mask); \
    hipLaunchKernelGGL(k_bld_walk<HI>, dim3(gb), dim3(64), 0, ctx->stream, (const uint64_t*)kv_lo, (const uint64_t*)kv_hi, (const uint8_t*)ctx->bl_pmin.as<uint8_t>(), \
                       (const uint32_t*)ctx->bl_pb.as<uint32_t>(), (const uint32_t*)place_first, (const uint32_t*)bucket_sk, (uint32_t)n_buckets, n_sk, \
                       (const unsigned long long*)slot, first, mask, k, m, ab, (const uint64_t*)keys, ctx->bl_codes.as<uint8_t>(), ctx->bl_text.as<uint8_t>(), \
                       text_per_place, outs)
    if (has_hi) { SPSP_BLD(true); } else { SPSP_BLD(false); }
#undef SPSP_BLD
    hipLaunchKernelGGL(k_bld_sizes, dim3((uint32_t)((n_buckets + 255) / 256)), dim3(256), 0, ctx->stream, (const BldOut*)outs, (uint32_t)n_buckets, m, out_len);
    if ((rc = launch_scan_u32(ctx, out_len, out_off, n_buckets, ctx->h_scalar + 6))) return rc;
    hipLaunchKernelGGL(k_bld_file_stats, dim3((uint32_t)((n_buckets + 255) / 256)), dim3(256), 0, ctx->stream, (const BldOut*)outs, (const uint64_t*)keys,
                       (const uint32_t*)bucket_sk, (uint32_t)n_buckets, d_fstats);
    SPSP_HIP(hipGetLastError());
    // first bucket of every file, for the split of the output: bucket_of at the file's first sorted super-k-mer -- the files'
    // super-k-mers are contiguous in the sorted order too (the file is the key's top), in file order
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t total = ctx->h_scalar[6];
    if (total > 0xfffffff0ull) { set_error("sketch payloads of one batch exceed 4 GiB"); return SPSP_ERR_OVERFLOW; }
    if ((rc = ctx->bl_out.reserve((size_t)total + 64))) return rc;
    hipLaunchKernelGGL(k_bld_assemble, dim3((uint32_t)n_buckets), dim3(64), 0, ctx->stream, (const BldOut*)outs, (const uint32_t*)out_off, (const uint32_t*)bucket_sk,
                       (const uint32_t*)place_first, (const uint64_t*)keys, (uint32_t)n_buckets, m, (const uint8_t*)ctx->bl_codes.as<uint8_t>(),
                       (const uint8_t*)ctx->bl_text.as<uint8_t>(), text_per_place, ctx->bl_out.as<uint8_t>());
    SPSP_HIP(hipGetLastError());
    // per file: where its first bucket's bytes start = out_off[bucket_of[first sorted super-k-mer of the file]]; the sorted order
    // keeps the files' super-k-mer COUNTS, so the file's first sorted super-k-mer is h_file_sk[f] as well
    std::vector<uint32_t> h_bucket_of_first(n_files + 1, 0), h_off((size_t)n_buckets + 1);
    std::vector<uint8_t> h_out((size_t)total);
    SPSP_HIP(hipMemcpyAsync(h_off.data(), out_off, (size_t)(n_buckets + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (total) SPSP_HIP(hipMemcpyAsync(h_out.data(), ctx->bl_out.p, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint32_t> h_bof(n_sk ? n_files : 0);
    for (uint32_t f = 0; f < n_files; ++f)
        if (h_file_sk[f] < h_file_sk[f + 1]) SPSP_HIP(hipMemcpyAsync(&h_bof[f], bucket_of + h_file_sk[f], 4, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipMemcpyAsync(file_stats->data(), d_fstats, (size_t)n_files * 32, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t next_b = (uint32_t)n_buckets;
    for (uint32_t f = n_files; f-- > 0;) {                                 // a file's buckets end where the next file with buckets starts
        if (h_file_sk[f] >= h_file_sk[f + 1]) continue;
        const uint32_t b0 = h_bof[f];
        (*bodies)[f].assign(reinterpret_cast<const char*>(h_out.data()) + h_off[b0], (size_t)(h_off[next_b] - h_off[b0]));
        next_b = b0;
    }
    return SPSP_OK;
}

}  // namespace spsp
