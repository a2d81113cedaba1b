// spsp_keys.hip -- from the scan's super-k-mer stream straight to the comparator's keys, on the device.
//
// In the reference a genome reaches the comparison through a file: handle_superkmer explodes every selected
// super-k-mer into its k-mers, oriented so that the minimizer reads canonically, and counts each one in a per-minimizer
// map with a uint8 count (SubSampler.cpp:243-302, SubSampler.h:23-27); the emission writes every k-mer whose count is
// >= abundance into some super-k-mer (:458-620: find_first_kmer / find_next only ever follow such k-mers, so a written
// super-k-mer holds those and nothing else); the comparator reads the super-k-mers back, walks their k-mers and keeps
// the DISTINCT canonical ones per bucket (Comparator.cpp:186-260).  Composed, a genome's keys are
//
//     { (minimizer, canon(x)) :  x an oriented k-mer of a selected super-k-mer,  (occurrences of x) mod 256 >= abundance }
//
// (for k = m without the count condition: the reader takes the minimizer of every bucket that exists, see
// sketch_keys_begin_impl) -- which needs neither the string reconstruction nor the file.  This is that composition for MANY genomes at once --
// genome g = records [first_rec[g], first_rec[g+1]) of one scan -- producing the arrays spsp_compare_device takes:
//
//   k_keys_sizes    k-mers per selected super-k-mer -> (scan) first raw key of each
//   k_keys_emit     one lane per super-k-mer: rolls the k-mers and their reverse complements from the bases (ASCII or
//                   2-bit words) -> raw keys (minimizer, canonical k-mer, "the oriented form is the reverse complement")
//   k_keys_ranges   one lane per genome: its super-k-mers (binary search on the record numbers) -> its raw key range
//   k_keys_sort     one workgroup per genome: bitonic sort in LDS by (minimizer, k-mer, orientation); occurrences per
//                   oriented k-mer -> the uint8 rule; distinct surviving canonical keys, compacted in place
//   k_keys_compact  genomes back to back
//
// A genome with more raw keys than the per-genome LDS forms hold (8192 / 6144 k-mers; 4096 with k > 32) is flagged by the
// workgroup that meets it and goes through the global-memory stages of spsp_bigkeys.hip, queued behind the LDS kernel in
// the same _begin call: one open-addressing table in HBM with the same (key, orientation) groups, counts and uint8 rule,
// and -- for the sorted form -- a merge sort of its distinct keys where they finally lie.  The reference's index is
// unbounded (SubSampler.h:62, SubSampler.cpp:274-300); so is this one, and nothing of it runs on the host.
#include <algorithm>
#include <cstring>
#include <string>

#include <vector>
#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

typedef unsigned __int128 u128d;
constexpr int kKeySortThreads = 1024;
constexpr uint32_t kKeyCapLo = 8192, kKeyCapHi = 4096;

__global__ void k_keys_sizes(const spsp_superkmer* __restrict__ sk, uint32_t n_sk, uint32_t k, uint32_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sk) cnt[i] = sk[i].len >= k ? sk[i].len - k + 1 : 0u;
}

__global__ __launch_bounds__(256) void k_keys_emit(const uint8_t* __restrict__ bases, bool packed, const uint64_t* __restrict__ rec_off,
                                                  const spsp_superkmer* __restrict__ sk, const uint32_t* __restrict__ raw_first,
                                                  uint32_t n_sk, uint32_t k, uint32_t* __restrict__ r_mn, uint64_t* __restrict__ r_lo,
                                                  uint64_t* __restrict__ r_hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sk) return;
    const spsp_superkmer e = sk[i];
    if (e.len < k) return;
    const uint64_t src = rec_off[e.rec] + e.start;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(bases);
    const u128d mask = k == 64 ? ~(u128d)0 : (((u128d)1 << (2 * k)) - 1);
    u128d fwd = 0, rc = 0;
    uint32_t o = raw_first[i];
    for (uint32_t t = 0; t < e.len; ++t) {
        const uint64_t q = src + t;
        const uint32_t c = packed ? (words[q >> 4] >> (30u - 2u * (uint32_t)(q & 15u))) & 3u : ((uint32_t)bases[q] >> 1) & 3u;
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((u128d)(c ^ 2u) << (2 * (k - 1)));
        if (t + 1 < k) continue;
        // handle_superkmer stores the k-mer as it reads in the super-k-mer's orientation (reverse complemented when the
        // minimizer reads reversed, SubSampler.cpp:246-249); the comparator canonises (utils.cpp:470-472)
        const u128d canon = fwd < rc ? fwd : rc;
        const u128d oriented = e.rev ? rc : fwd;
        r_mn[o] = e.minimizer | (oriented != canon ? 0x80000000u : 0u);   // bit 31: "the oriented form is the reverse complement"
        r_lo[o] = (uint64_t)canon;
        if (r_hi) r_hi[o] = (uint64_t)(canon >> 64);
        ++o;
    }
}

// genome g = records [first_rec[g], first_rec[g + 1]); the stream is in record order
__global__ void k_keys_ranges(const spsp_superkmer* __restrict__ sk, uint32_t n_sk, const uint32_t* __restrict__ raw_first,
                              const uint32_t* __restrict__ first_rec, uint32_t n_genomes, uint32_t* __restrict__ raw_off) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_genomes) return;
    const uint32_t want = first_rec[g];
    uint32_t lo = 0, hi = n_sk;                              // first super-k-mer with rec >= want
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sk[mid].rec < want) lo = mid + 1; else hi = mid; }
    raw_off[g] = raw_first[lo];                              // (raw_first has n_sk + 1 entries)
}

template <bool HAS_HI>
__global__ __launch_bounds__(kKeySortThreads) void k_keys_sort(uint32_t* __restrict__ r_mn, uint64_t* __restrict__ r_lo, uint64_t* __restrict__ r_hi,
                                                              const uint32_t* __restrict__ raw_off, uint32_t abundance, uint32_t* __restrict__ distinct,
                                                              uint32_t* __restrict__ raw_cnt, uint32_t* __restrict__ big, uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_k[];
    constexpr uint32_t CAP = HAS_HI ? kKeyCapHi : kKeyCapLo;
    uint64_t* s_lo = reinterpret_cast<uint64_t*>(lds_k);
    uint64_t* s_hi = s_lo + CAP;                                   // (HAS_HI only)
    uint32_t* s_mn = reinterpret_cast<uint32_t*>(s_hi + (HAS_HI ? CAP : 0));
    uint8_t* s_or = reinterpret_cast<uint8_t*>(s_mn + CAP);
    __shared__ uint32_t wave_sum[kKeySortThreads / 64];
    const uint32_t g = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint32_t r0 = raw_off[g], n = raw_off[g + 1] - r0;
    if (t == 0) { raw_cnt[g] = n; big[g] = n > CAP ? 1u : 0u; }
    if (n == 0) { if (t == 0) distinct[g] = 0; return; }
    // more than the LDS holds: the global-memory stages queued behind this kernel take the genome (spsp_bigkeys.hip)
    if (n > CAP) { if (t == 0) { distinct[g] = 0; atomicOr(&flags[0], 1u); atomicAdd(&flags[1], 1u); } return; }
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (uint32_t i = t; i < n2; i += kKeySortThreads) {
        if (i < n) { const uint32_t mo = r_mn[r0 + i]; s_mn[i] = mo & 0x7fffffffu; s_or[i] = (uint8_t)(mo >> 31); s_lo[i] = r_lo[r0 + i]; if (HAS_HI) s_hi[i] = r_hi[r0 + i]; }
        else { s_mn[i] = 0xffffffffu; s_lo[i] = ~0ull; if (HAS_HI) s_hi[i] = ~0ull; s_or[i] = 1; }
    }
    __syncthreads();
    auto greater = [&](uint32_t a, uint32_t b) {                   // (minimizer, k-mer, orientation)
        if (s_mn[a] != s_mn[b]) return s_mn[a] > s_mn[b];
        if (HAS_HI && s_hi[a] != s_hi[b]) return s_hi[a] > s_hi[b];
        if (s_lo[a] != s_lo[b]) return s_lo[a] > s_lo[b];
        return s_or[a] > s_or[b];
    };
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t idx = t; idx < (n2 >> 1); idx += kKeySortThreads) {
                const uint32_t i = ((idx / stride) * (stride << 1)) + (idx % stride), j = i + stride;
                const bool asc = (i & size) == 0;
                if (greater(i, j) == asc) {
                    const uint32_t tm = s_mn[i]; s_mn[i] = s_mn[j]; s_mn[j] = tm;
                    const uint64_t tl = s_lo[i]; s_lo[i] = s_lo[j]; s_lo[j] = tl;
                    if (HAS_HI) { const uint64_t th = s_hi[i]; s_hi[i] = s_hi[j]; s_hi[j] = th; }
                    const uint8_t to = s_or[i]; s_or[i] = s_or[j]; s_or[j] = to;
                }
            }
            __syncthreads();
        }
    }
    auto same_key = [&](uint32_t a, uint32_t b) { return s_mn[a] == s_mn[b] && s_lo[a] == s_lo[b] && (!HAS_HI || s_hi[a] == s_hi[b]); };
    // occurrences of the oriented k-mer that starts at i (a run of equal key AND orientation), read through the
    // reference's uint8 counter (SubSampler.h:24): 256 occurrences count as 0
    auto usable_from = [&](uint32_t i, uint32_t* next) {
        uint32_t j = i + 1;
        while (j < n && same_key(i, j) && s_or[j] == s_or[i]) ++j;
        *next = j;
        return ((j - i) & 255u) >= abundance;
    };
    constexpr uint32_t PER = CAP / kKeySortThreads;
    uint32_t keep[PER];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = t * PER + u;
        keep[u] = 0;
        if (i < n && (i == 0 || !same_key(i, i - 1))) {            // first record of a canonical key: its (at most two) orientations
            uint32_t j;
            bool ok = usable_from(i, &j);
            if (j < n && same_key(i, j)) { uint32_t j2; ok = usable_from(j, &j2) || ok; }
            keep[u] = ok ? 1u : 0u;
        }
        cnt += keep[u];
    }
    uint32_t x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    for (uint32_t w = 0; w < kKeySortThreads / 64; ++w) { if (w < wid) pre += wave_sum[w]; all += wave_sum[w]; }
    uint32_t rank = pre + x - cnt;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = t * PER + u;
        if (!keep[u]) continue;
        r_mn[r0 + rank] = s_mn[i]; r_lo[r0 + rank] = s_lo[i];
        if (HAS_HI) r_hi[r0 + rank] = s_hi[i];
        ++rank;
    }
    if (t == 0) distinct[g] = all;
}

// The same rule without the sort and without the raw arrays, for a caller that does not need the keys in order (the
// comparison's partition form only needs them DISTINCT within a genome: spsp_compare_keys_unordered).  ONE kernel per
// call, one workgroup per genome: it finds its super-k-mers (two binary searches on the record numbers), gives super-k-mer
// q the table records [q w, (q + 1) w), w = k - m + 1 -- every lane rolls ONE k-mer and its reverse complement from the
// bases -- groups them in an LDS table per (canonical k-mer, orientation) -- a slot word is (occurrences << 13 | claiming
// record + 1), full keys are compared against the claiming record -- and the claimer of a usable group emits the key
// unless the other orientation's group is usable too and comes first.  A tenth of the sorted form's time (its bitonic
// sort is 91 barrier-separated passes over 13-byte records), which is what lets the key extraction run inside a 0.1 ms
// step.
constexpr uint32_t kDedupCapLo = 6144, kDedupCapHi = 4096, kDedupSkmWords = 4096;   // k-mer places per genome; staged super-k-mer words (16 KiB)
__device__ __forceinline__ uint64_t keys_mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
// One super-k-mer's k-mer places as raw records (minimizer | orientation << 31, canonical k-mer) for the table in HBM of
// spsp_bigkeys.hip: place0 + j = k-mer j, 0xffffffff where the super-k-mer has fewer than w.  Rolled base by base from
// global memory: only the genomes that do not fit a workgroup's LDS come this way.
// The same records, one lane per k-mer PLACE (round 5): a workgroup takes 1024 consecutive places, finds the super-k-mer of
// the first one (64 probes per search step), stages the place offsets of the <= 1024 super-k-mers that can own them in LDS
// and every lane cuts its k-mer out of the bases by a shift -- coalesced stores, no serial roll.  (One lane per super-k-mer
// wrote its ~50 places one after the other, every store of a wave to 64 different lines: 2.7 ms for 4 x 10^7 places of a
// metagenome segment, BASELINE configs[4].)
constexpr uint32_t kPlaceThreads = 256, kPlacePer = 4, kPlaceTile = kPlaceThreads * kPlacePer;
__global__ __launch_bounds__(kPlaceThreads) void k_keys_emit_places(const uint8_t* __restrict__ bases, bool packed, const uint64_t* __restrict__ rec_off,
                                                                   const spsp_superkmer* __restrict__ sk, const uint32_t* __restrict__ raw_first,
                                                                   const uint32_t* __restrict__ cnt, uint32_t n_sk, uint32_t k,
                                                                   uint32_t* __restrict__ r_mn, uint64_t* __restrict__ r_lo, uint64_t* __restrict__ r_hi) {
    __shared__ uint32_t s_first[kPlaceTile + 1];
    __shared__ uint32_t s_i0;
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t total = raw_first[n_sk - 1] + cnt[n_sk - 1];
    const uint32_t P0 = blockIdx.x * kPlaceTile;
    if (P0 >= total) return;
    if (t < 64) {                                                  // the last super-k-mer whose first place is <= P0
        uint32_t lo = 0, hi = n_sk;                                // answer in [lo, hi): raw_first[lo] <= P0 (raw_first[0] = 0)
        while (hi - lo > 1) {
            const uint32_t span = hi - lo - 1, step = (span + 63) / 64;       // probes lo + step, lo + 2 step, ...
            const uint32_t at = lo + (lane + 1) * step;
            const bool le = at < hi && raw_first[at] <= P0;
            const uint32_t nle = (uint32_t)__popcll(__ballot(le));           // probes 1 .. nle are <= P0 (monotone)
            const uint32_t new_lo = lo + nle * step, next = new_lo + step;
            hi = next < hi ? next : hi;
            lo = new_lo;
        }
        if (lane == 0) s_i0 = lo;
    }
    __syncthreads();
    const uint32_t i0 = s_i0;
    for (uint32_t x = t; x <= kPlaceTile; x += kPlaceThreads) s_first[x] = i0 + x < n_sk ? raw_first[i0 + x] : 0xffffffffu;
    __syncthreads();
    const uint32_t* words = reinterpret_cast<const uint32_t*>(bases);
    const u128d mask = k == 64 ? ~(u128d)0 : (((u128d)1 << (2 * k)) - 1);
#pragma unroll 1
    for (uint32_t u = 0; u < kPlacePer; ++u) {
        const uint32_t o = P0 + u * kPlaceThreads + t;
        if (o >= total) break;
        uint32_t x = 0;                                            // the last staged super-k-mer whose first place is <= o (one without places
#pragma unroll                                                     // shares its offset with the next one: the last of equals owns the place)
        for (uint32_t step = kPlaceTile / 2; step; step >>= 1) if (s_first[x + step] <= o) x += step;
        uint32_t i = i0 + x, first = s_first[x];
        spsp_superkmer e = sk[i];
        if (o - first >= (e.len >= k ? e.len - k + 1 : 0u)) {      // (more than 1024 super-k-mers without a place in front of this one: never from
            uint32_t lo = i, hi = n_sk;                            // the scan, which emits no super-k-mer below k bases -- searched, not assumed)
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (raw_first[mid] <= o) lo = mid; else hi = mid; }
            i = lo; first = raw_first[i]; e = sk[i];
        }
        const uint64_t q = rec_off[e.rec] + e.start + (o - first);          // first base of this place's k-mer
        u128d fwd;
        if (packed) {
            const uint32_t* W = words + (q >> 4);
            const uint32_t sh = 2u * (uint32_t)(q & 15u);
            const u128d top = ((u128d)W[0] << 96) | ((u128d)W[1] << 64) | ((u128d)W[2] << 32) | (u128d)W[3];
            const u128d win = sh ? (top << sh) | ((u128d)W[4] >> (32u - sh)) : top;      // 64 bases from q on (256 readable bytes follow the last word)
            fwd = win >> (128u - 2u * k);
        } else {
            fwd = 0;
            for (uint32_t b = 0; b < k; ++b) fwd = (fwd << 2) | (((uint32_t)bases[q + b] >> 1) & 3u);
        }
        const u128d top_aligned = fwd << (128u - 2u * k);
        const u128d rcw = ((u128d)rc_window64((uint64_t)top_aligned) << 64) | (u128d)rc_window64((uint64_t)(top_aligned >> 64));
        const u128d rc = rcw & mask;
        const u128d canon = fwd < rc ? fwd : rc;
        const u128d oriented = e.rev ? rc : fwd;
        r_mn[o] = e.minimizer | (oriented != canon ? 0x80000000u : 0u);
        r_lo[o] = (uint64_t)canon;
        if (r_hi) r_hi[o] = (uint64_t)(canon >> 64);
    }
}

__device__ __noinline__ void roll_places(const uint8_t* __restrict__ bases, bool packed, const uint64_t* __restrict__ rec_off, const spsp_superkmer e,
                                         uint32_t k, uint32_t w, uint32_t place0, uint32_t* __restrict__ r_mn, uint64_t* __restrict__ r_lo,
                                         uint64_t* __restrict__ r_hi) {
    const uint64_t src = rec_off[e.rec] + e.start;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(bases);
    const u128d mask = k == 64 ? ~(u128d)0 : (((u128d)1 << (2 * k)) - 1);
    u128d fwd = 0, rc = 0;
    uint32_t made = 0;
    for (uint32_t t = 0; t < e.len && made < w; ++t) {
        const uint64_t b = src + t;
        const uint32_t c = packed ? (words[b >> 4] >> (30u - 2u * (uint32_t)(b & 15u))) & 3u : ((uint32_t)bases[b] >> 1) & 3u;
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((u128d)(c ^ 2u) << (2 * (k - 1)));
        if (t + 1 < k) continue;
        const u128d canon = fwd < rc ? fwd : rc;                   // (orientation as in k_keys_emit)
        const u128d oriented = e.rev ? rc : fwd;
        r_mn[place0 + made] = e.minimizer | (oriented != canon ? 0x80000000u : 0u);
        r_lo[place0 + made] = (uint64_t)canon;
        if (r_hi) r_hi[place0 + made] = (uint64_t)(canon >> 64);
        ++made;
    }
    for (; made < w; ++made) r_mn[place0 + made] = 0xffffffffu;
}

template <bool HAS_HI>
__global__ __launch_bounds__(kKeySortThreads) void k_keys_fused(const uint8_t* __restrict__ bases, bool packed, uint64_t n_bases_readable,
                                                               const uint64_t* __restrict__ rec_off,
                                                               const spsp_superkmer* __restrict__ sk, uint32_t n_sk, const uint32_t* __restrict__ first_rec,
                                                               uint32_t k, uint32_t w, uint32_t abundance, uint32_t* __restrict__ r_mn,
                                                               uint64_t* __restrict__ r_lo, uint64_t* __restrict__ r_hi, uint32_t* __restrict__ raw_off,
                                                               uint32_t* __restrict__ distinct, uint32_t* __restrict__ raw_cnt, uint32_t* __restrict__ big,
                                                               uint32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_d[];
    constexpr uint32_t FULL = HAS_HI ? kDedupCapHi : kDedupCapLo, CAP = FULL, SLOTS = 2 * CAP, PER = FULL / kKeySortThreads;
    constexpr uint32_t kNone = 0xffffffffu;                        // record that is not this workgroup's (or holds no k-mer)
    uint64_t* k_lo = reinterpret_cast<uint64_t*>(lds_d);
    uint64_t* k_hi = k_lo + CAP;                                   // (HAS_HI only)
    uint32_t* k_mn = reinterpret_cast<uint32_t*>(k_hi + (HAS_HI ? CAP : 0));   // minimizer | orientation << 31
    uint32_t* slot = k_mn + CAP;
    // the genome's super-k-mers as 2-bit words (16 bases per word, first base in bits 31:30), CH words each: staged once,
    // so that a record's k-mer is two or three LDS words and a shift instead of k loads from global memory
    constexpr uint32_t CH = HAS_HI ? 8 : 4, SKM_MAX = kDedupSkmWords / CH;
    uint32_t* skw = slot + SLOTS;
    __shared__ uint32_t wave_sum[kKeySortThreads / 64];
    __shared__ uint32_t s_count;
    const uint32_t g = blockIdx.x, seg = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    // this genome's super-k-mers: [q0, q1) (the stream is in record order)
    // (wave 0 searches 64 ways at a time -- three dependent loads per bound instead of fifteen -- and tells the others)
    __shared__ uint32_t s_q[2];
    if (wid == 0) {
        for (int which = 0; which < 2; ++which) {
            const uint32_t want = first_rec[g + which];
            uint32_t lo = 0, hi = n_sk;                      // answer = first index in [lo, hi] whose rec >= want (hi: none)
            while (hi - lo > 0) {
                const uint32_t span = hi - lo, step = (span + 63) / 64;
                const uint32_t at = lo + lane * step;        // probes lo, lo + step, ...
                const bool below = at < hi && sk[at].rec < want;
                const uint32_t nb = (uint32_t)__popcll(__ballot(below));   // probes 0 .. nb - 1 are below (monotone)
                if (nb == 0) { hi = lo; break; }
                const uint32_t last_below = lo + (nb - 1) * step;
                lo = last_below + 1;
                const uint32_t next_probe = lo + step - 1;   // the first probe not below (or past the end)
                hi = next_probe < hi ? next_probe : hi;
            }
            if (lane == 0) s_q[which] = lo;
        }
    }
    __syncthreads();
    const uint32_t q0 = s_q[0], q1 = s_q[1];
    const uint32_t n = (q1 - q0) * w;                              // the genome's k-mer places (some empty)
    const uint32_t r0 = q0 * w;                                    // this genome's room in the staging arrays: its k-mer places
    const bool too_big = n > FULL || q1 - q0 > SKM_MAX;
    if (t == 0) { raw_off[seg] = r0; raw_cnt[seg] = n; big[seg] = too_big ? 1u : 0u; s_count = 0; }
    if (n == 0) { if (t == 0) distinct[seg] = 0; return; }
    // more k-mer places (or super-k-mers) than this workgroup's LDS holds: the table in HBM (spsp_bigkeys.hip) takes the
    // genome.  Its raw records are written here, into the genome's slice of the context's staging arrays, so that those
    // stages read nothing of the caller's -- they may be queued right behind this kernel or, the first time a context
    // meets such a genome, from _end.  flags[0] is their gate, flags[1] counts such genomes for the report.
    if (too_big) {
        if (t == 0) { distinct[seg] = 0; atomicOr(&flags[0], 1u); atomicAdd(&flags[1], 1u); }
        for (uint32_t q = q0 + t; q < q1; q += kKeySortThreads) roll_places(bases, packed, rec_off, sk[q], k, w, q * w, r_mn, r_lo, HAS_HI ? r_hi : nullptr);
        return;
    }
    for (uint32_t x = t; x < SLOTS; x += kKeySortThreads) slot[x] = 0;
    {
        const uint32_t* gw = reinterpret_cast<const uint32_t*>(bases);
        for (uint32_t c = t; c < (q1 - q0) * CH; c += kKeySortThreads) {
            const spsp_superkmer e = sk[q0 + c / CH];
            const uint32_t b0 = (c % CH) * 16;                 // first base of this word inside the super-k-mer
            uint32_t word = 0;
            if (b0 < e.len) {
                const uint64_t q = rec_off[e.rec] + e.start + b0;
                const uint32_t have = e.len - b0 < 16 ? e.len - b0 : 16;
                if (packed) {
                    const uint32_t sh = 2u * (uint32_t)(q & 15u);
                    const uint32_t w0 = gw[q >> 4], w1 = sh ? gw[(q >> 4) + 1] : 0u;   // (256 readable bytes follow the last word)
                    word = sh ? (w0 << sh) | (w1 >> (32u - sh)) : w0;
                } else if (((q & ~15ull) + 32) <= n_bases_readable) {
                    // two aligned 16-byte loads cover bases q .. q + 15: packed separately, joined, shifted into place
                    const uint4* v = reinterpret_cast<const uint4*>(bases + (q & ~15ull));
                    const uint64_t both = ((uint64_t)pack16(v[0]) << 32) | pack16(v[1]);
                    word = (uint32_t)((both << (2u * (uint32_t)(q & 15u))) >> 32);
                } else {
                    for (uint32_t b = 0; b < have; ++b) word |= (((uint32_t)bases[q + b] >> 1) & 3u) << (30u - 2u * b);
                }
                if (have < 16) word &= ~0u << (32u - 2u * have);
            }
            skw[c] = word;
        }
    }
    __syncthreads();
    auto home = [&](uint32_t mo, uint64_t lo, uint64_t hi) {
        uint64_t h = keys_mix(lo ^ 0x9E3779B97F4A7C15ULL);
        h = keys_mix(h + (uint64_t)mo * 0xD6E8FEB86659FD93ULL);
        if (HAS_HI) h = keys_mix(h ^ hi);
        return (uint32_t)(((h & 0xffffffffull) * SLOTS) >> 32);
    };
    const u128d mask = ((u128d)1 << (2 * k)) - 1;                  // k <= 63
    uint32_t hs[PER], id[PER];                                     // home slot and table record of this lane's u-th k-mer (kNone: not ours)
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {                           // place r = u * threads + t = k-mer r % w of super-k-mer q0 + r / w
        const uint32_t r = u * kKeySortThreads + t;
        hs[u] = 0; id[u] = kNone;
        if (r >= n) continue;
        const spsp_superkmer e = sk[q0 + r / w];
        const uint32_t j = r % w;
        if (e.len < k || j > e.len - k) continue;
        const uint32_t* sw = skw + (r / w) * CH;
        u128d fwd = 0, rc = 0;
        if (!HAS_HI) {
            // k <= 32: bases j .. j + k - 1 lie inside the three words from j / 16 on (48 bases, j % 16 + k <= 47)
            const uint32_t p = j >> 4, o = j & 15u;
            const uint32_t w0 = sw[p], w1 = p + 1 < CH ? sw[p + 1] : 0u, w2 = p + 2 < CH ? sw[p + 2] : 0u;
            const u128d X = ((u128d)w0 << 64) | ((u128d)w1 << 32) | w2;
            const uint64_t f = (uint64_t)(X >> (2u * (48u - o - k))) & (uint64_t)mask;
            fwd = f;
            rc = rc_window64(f << (64u - 2u * k)) & (uint64_t)mask;    // reversed and complemented; the padding lands above bit 2k
        } else {
            for (uint32_t b = 0; b < k; ++b) {
                const uint32_t q = j + b;
                const uint32_t c = (sw[q >> 4] >> (30u - 2u * (q & 15u))) & 3u;
                fwd = (fwd << 2) | c;
                rc = (rc >> 2) | ((u128d)(c ^ 2u) << (2 * (k - 1)));
            }
            fwd &= mask;
        }
        // handle_superkmer stores the k-mer as it reads in the super-k-mer's orientation (reverse complemented when the
        // minimizer reads reversed, SubSampler.cpp:246-249); the comparator canonises (utils.cpp:470-472)
        const u128d canon = fwd < rc ? fwd : rc;
        const uint32_t mo = e.minimizer | (((e.rev ? rc : fwd) != canon) ? 0x80000000u : 0u);
        const uint64_t lo = (uint64_t)canon, hi = (uint64_t)(canon >> 64);
        const uint32_t i = atomicAdd(&s_count, 1u);                // (at most n <= CAP records)
        id[u] = i;
        k_mn[i] = mo; k_lo[i] = lo;
        if (HAS_HI) k_hi[i] = hi;
        hs[u] = home(mo, lo, hi);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t r = id[u];
        if (r == kNone) continue;
        const uint32_t mo = k_mn[r];
        const uint64_t lo = k_lo[r], hi = HAS_HI ? k_hi[r] : 0ull;
        uint32_t h = hs[u];
        for (;;) {                                                 // ends: twice as many slots as records
            uint32_t cur = slot[h];
            if (cur == 0) cur = atomicCAS(&slot[h], 0u, r + 1);
            if (cur == 0) break;                                   // claimed
            const uint32_t c = (cur & 0x1fffu) - 1;
            if (k_lo[c] == lo && k_mn[c] == mo && (!HAS_HI || k_hi[c] == hi)) break;
            h = h + 1 == SLOTS ? 0u : h + 1;
        }
        hs[u] = h;
        atomicAdd(&slot[h], 1u << 13);
    }
    __syncthreads();
    // the reference's uint8 counter per ORIENTED k-mer (SubSampler.h:24): 256 occurrences read as 0
    auto usable = [&](uint32_t wd) { return ((wd >> 13) & 255u) >= abundance; };
    uint32_t keep[PER];
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t r = id[u];
        keep[u] = 0;
        if (r == kNone) continue;
        const uint32_t wd = slot[hs[u]];
        if ((wd & 0x1fffu) != r + 1 || !usable(wd)) continue;      // one lane per (key, orientation) group: its claimer
        const uint32_t mo = k_mn[r];
        uint32_t emit = 1;
        if (mo >> 31) {                                            // the forward-oriented group of the same canonical key emits if it is usable
            const uint32_t sib = mo & 0x7fffffffu;
            const uint64_t lo = k_lo[r], hi = HAS_HI ? k_hi[r] : 0ull;
            uint32_t h = home(sib, lo, hi);
            for (;;) {
                const uint32_t cur = slot[h];
                if (cur == 0) break;                               // no such group
                const uint32_t c = (cur & 0x1fffu) - 1;
                if (k_lo[c] == lo && k_mn[c] == sib && (!HAS_HI || k_hi[c] == hi)) { if (usable(cur)) emit = 0; break; }
                h = h + 1 == SLOTS ? 0u : h + 1;
            }
        }
        keep[u] = emit;
    }
    // places in the output: one workgroup prefix over the lanes' counts, a lane's keys one after the other (the order of the
    // keys inside a genome means nothing in this form)
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) cnt += keep[u];
    uint32_t x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0, base = 0;
    for (uint32_t w2 = 0; w2 < kKeySortThreads / 64; ++w2) { if (w2 < wid) pre += wave_sum[w2]; base += wave_sum[w2]; }
    uint32_t at = r0 + pre + x - cnt;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t r = id[u];
        if (!keep[u]) continue;
        r_mn[at] = k_mn[r] & 0x7fffffffu; r_lo[at] = k_lo[r];
        if (HAS_HI) r_hi[at] = k_hi[r];
        ++at;
    }
    if (t == 0) distinct[seg] = base;
}

// genomes back to back: a genome's distinct keys lie at the start of its slice of the staging arrays -- (r_*) for the
// LDS forms, (b_*) for a genome the global-memory stages took
__global__ __launch_bounds__(256) void k_keys_compact(const uint32_t* __restrict__ r_mn, const uint64_t* __restrict__ r_lo,
                                                     const uint64_t* __restrict__ r_hi, const uint32_t* __restrict__ b_mn,
                                                     const uint64_t* __restrict__ b_lo, const uint64_t* __restrict__ b_hi,
                                                     const uint32_t* __restrict__ raw_off, const uint32_t* __restrict__ distinct,
                                                     const uint32_t* __restrict__ big, const uint32_t* __restrict__ out_off,
                                                     uint32_t* __restrict__ o_mn, uint64_t* __restrict__ o_lo, uint64_t* __restrict__ o_hi,
                                                     uint32_t n_genomes, uint32_t* __restrict__ flags, uint32_t* __restrict__ host_out) {
    const uint32_t g = blockIdx.y;
    __shared__ uint32_t s_o0;
    if (!out_off) {                                          // unordered form: no scan launch in front -- the counts of the genomes before this one
        uint32_t part = 0;
        for (uint32_t j = threadIdx.x; j < g; j += 256) part += distinct[j];
#pragma unroll
        for (int d = 32; d; d >>= 1) part += __shfl_xor(part, d);
        __shared__ uint32_t s_w[4];
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) s_o0 = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    const uint32_t n = distinct[g], r0 = raw_off[g], o0 = out_off ? out_off[g] : s_o0;
    const bool from_b = big[g] != 0;
    const uint32_t* s_mn = from_b ? b_mn : r_mn;
    const uint64_t* s_lo = from_b ? b_lo : r_lo;
    const uint64_t* s_hi = from_b ? b_hi : r_hi;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        o_mn[o0 + e] = s_mn[r0 + e]; o_lo[o0 + e] = s_lo[r0 + e];
        if (o_hi) o_hi[o0 + e] = s_hi[r0 + e];
    }
    // the offsets and the report travel to pinned host memory from here: the job ends with this kernel.
    // host_out: [0, n_genomes] key offsets; [n_genomes + 1] number of genomes the global-memory stages took;
    //           [n_genomes + 2 + g] "genome g was one of them" (the sorted form sorts those where they lie, in _end)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        host_out[g] = o0;
        host_out[n_genomes + 2 + g] = from_b ? 1u : 0u;
        if (g == n_genomes - 1) {
            host_out[n_genomes] = o0 + n;
            // (zeroed by the host at _begin; a second run queued from _end finds the word reported and cleared: it stays)
            if (flags[1]) host_out[n_genomes + 1] = flags[1];
            flags[0] = 0; flags[1] = 0;                         // for the next extraction on this context (every reader of this one has finished)
        }
    }
}

// [the table in HBM for the flagged genomes] -> [offsets, sorted form] -> compaction, on the context's stream.  with_big:
// queue k_big_insert / k_big_emit (gate: a device word that is zero when no genome was flagged, or nullptr = run).
static int keys_finish_queue(spsp_ctx* ctx, bool with_big, const uint32_t* d_gate) {
    const KeysJob& J = ctx->keys_job;
    const uint32_t ng = J.n_genomes;
    uint32_t* d_first_rec = ctx->dc_meta.as<uint32_t>();
    uint32_t* d_raw_off = d_first_rec + (ng + 2);
    uint32_t* d_distinct = d_raw_off + (ng + 2);
    uint32_t* d_out_off = d_distinct + (ng + 2);
    uint32_t* d_raw_cnt = d_out_off + (ng + 2);
    uint32_t* d_big = d_raw_cnt + (ng + 2);
    uint32_t* d_flags = ctx->c_flags.as<uint32_t>() + 16;
    uint32_t* h_out = ctx->h_keys + (ng + 1);
    uint32_t* a_mn = ctx->a_mn.as<uint32_t>();
    uint64_t *a_lo = ctx->a_lo.as<uint64_t>(), *a_hi = J.has_hi ? ctx->a_hi.as<uint64_t>() : (uint64_t*)nullptr;
    uint32_t* b_mn = ctx->b_mn.as<uint32_t>();
    uint64_t *b_lo = ctx->b_lo.as<uint64_t>(), *b_hi = J.has_hi ? ctx->b_hi.as<uint64_t>() : (uint64_t*)nullptr;
    int rc;
    if (with_big && J.bound &&
        (rc = big_dedupe_launch(ctx, J.has_hi, a_mn, a_lo, a_hi, d_raw_off, d_raw_cnt, d_big, ng, J.bound, d_gate, J.abundance, b_mn, b_lo, b_hi, d_distinct))) return rc;
    if (J.flat && (rc = launch_scan_u32(ctx, d_distinct, d_out_off, ng, ctx->h_scalar + 7))) return rc;
    // (8 workgroups of 256 copy a genome's few thousand keys; a call with few, huge genomes gets as many as its keys need)
    const uint32_t gx = (uint32_t)std::min<uint64_t>(2048, std::max<uint64_t>(8, J.bound / ng / 2048));
    hipLaunchKernelGGL(k_keys_compact, dim3(gx, ng), dim3(256), 0, ctx->stream, a_mn, a_lo, a_hi, b_mn, b_lo, b_hi, d_raw_off, d_distinct, d_big,
                       J.flat ? (const uint32_t*)d_out_off : (const uint32_t*)nullptr, ctx->c_min.as<uint32_t>(), ctx->c_lo.as<uint64_t>(),
                       J.has_hi ? ctx->c_hi.as<uint64_t>() : (uint64_t*)nullptr, ng, d_flags, h_out);
    SPSP_HIP(hipGetLastError());
    return SPSP_OK;
}

int sketch_keys_begin_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, const uint64_t* d_rec_off,
                           const spsp_superkmer* d_sk, uint64_t n_sk, const uint32_t* h_first_rec, uint32_t n_genomes, bool unordered,
                           uint64_t n_bases_readable) {
    int rc = check_params(p);
    if (rc) return rc;
    if (ctx->keys_pending) { set_error("a key extraction is already pending on this context"); return SPSP_ERR_ARG; }
    if (n_genomes == 0) { set_error("no genomes"); return SPSP_ERR_ARG; }
    if (n_genomes > 65535) { set_error("at most 65535 genomes per call"); return SPSP_ERR_ARG; }
    for (uint32_t g = 0; g < n_genomes; ++g)
        if (h_first_rec[g + 1] < h_first_rec[g]) { set_error("genome record ranges must be non-decreasing (genome %u)", g); return SPSP_ERR_ARG; }
    if (n_sk > 0x7ffffff0ull / 64) { set_error("too many super-k-mers for one call"); return SPSP_ERR_OVERFLOW; }
    const uint32_t n = (uint32_t)n_sk;
    const bool has_hi = p->k > 32;
    const uint32_t w = p->k - p->m + 1;
    const uint64_t bound = (uint64_t)n * w;                        // raw keys: a super-k-mer holds at most k - m + 1 k-mers
    // pinned staging: first_rec in; offsets, report and per-genome flags out
    const size_t need = (size_t)(n_genomes + 1) + (size_t)(2 * n_genomes + 2);
    if (ctx->h_keys_cap < need) {
        if (ctx->h_keys) { SPSP_HIP(hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->h_keys); ctx->h_keys = nullptr; ctx->h_keys_cap = 0; }
        size_t cap = 1024;
        while (cap < need) cap *= 2;
        SPSP_HIP(hipHostMalloc((void**)&ctx->h_keys, cap * 4, hipHostMallocDefault));
        ctx->h_keys_cap = cap;
    }
    memcpy(ctx->h_keys, h_first_rec, (size_t)(n_genomes + 1) * 4);
    ctx->h_keys[(n_genomes + 1) + (n_genomes + 1)] = 0;            // "genomes beyond the LDS forms": written by the compaction only when there are any
    // a_*: the LDS forms' staging (raw records in, a genome's distinct keys out, slice by slice); b_*: the output slices of
    // the genomes the global-memory stages take; c_*: the keys of all genomes back to back (what the comparison reads)
    if ((rc = ctx->a_cnt.reserve((size_t)(n + 1) * 4)) || (rc = ctx->a_off.reserve((size_t)(n + 2) * 4)) ||
        (rc = ctx->a_mn.reserve((size_t)bound * 4 + 64)) || (rc = ctx->a_lo.reserve((size_t)bound * 8 + 64)) ||
        (has_hi && (rc = ctx->a_hi.reserve((size_t)bound * 8 + 64))) ||
        (rc = ctx->b_mn.reserve((size_t)bound * 4 + 64)) || (rc = ctx->b_lo.reserve((size_t)bound * 8 + 64)) ||
        (has_hi && (rc = ctx->b_hi.reserve((size_t)bound * 8 + 64))) ||
        (rc = ctx->dc_meta.reserve((size_t)(n_genomes + 2) * 4 * 6 + 64)) ||
        (rc = ctx->c_min.reserve((size_t)bound * 4 + 64)) || (rc = ctx->c_lo.reserve((size_t)bound * 8 + 64)) ||
        (has_hi && (rc = ctx->c_hi.reserve((size_t)bound * 8 + 64)))) return rc;
    uint32_t* d_first_rec = ctx->dc_meta.as<uint32_t>();
    uint32_t* d_raw_off = d_first_rec + (n_genomes + 2);
    uint32_t* d_distinct = d_raw_off + (n_genomes + 2);
    uint32_t* d_out_off = d_distinct + (n_genomes + 2);
    uint32_t* d_raw_cnt = d_out_off + (n_genomes + 2);
    uint32_t* d_big = d_raw_cnt + (n_genomes + 2);
    if ((rc = ctx->c_flags.reserve(256))) return rc;
    uint32_t* d_flags = ctx->c_flags.as<uint32_t>() + 16;          // two words of its own behind the comparison's sixteen (which k_parts_prepare clears and k_parts_group counts in)
    // The unordered form runs on a stream that carries a key extraction per step: its workgroups read their two record
    // bounds straight from the pinned staging block (no copy packet in front), and the gate words are cleared by the
    // compaction kernel that reports them (no fill packet either; cleared here after a call that did not get that far).
    // The sorted form keeps the copy: several of its kernels read the bounds.
    const bool flat_front = !unordered || bound > (uint64_t)n_genomes * 32768ull;   // (see below)
    if (flat_front) SPSP_HIP(hipMemcpyAsync(d_first_rec, ctx->h_keys, (size_t)(n_genomes + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    else d_first_rec = ctx->h_keys;
    if (!ctx->keys_flags_clear) SPSP_HIP(hipMemsetAsync(d_flags, 0, 8, ctx->stream));
    ctx->keys_flags_clear = false;
    // k == m: a bucket's only k-mer is its minimizer, and the reader takes the minimizer of EVERY bucket that exists for a
    // k-mer (an empty blob reads as the bare minimizer, Comparator.cpp:88-90,193-198) -- also of one whose k-mer stayed below
    // -a or wrapped to 0: its map entry made the bucket exist (SubSampler.cpp:283-300).  The count rule does not apply.
    const uint32_t ab = p->k == p->m ? 0u : (p->abundance ? p->abundance : 1u);
    uint32_t* a_mn = ctx->a_mn.as<uint32_t>();
    uint64_t *a_lo = ctx->a_lo.as<uint64_t>(), *a_hi = has_hi ? ctx->a_hi.as<uint64_t>() : (uint64_t*)nullptr;
    // Unordered form with few, very large genomes (a metagenome record set as ONE sketch, BASELINE configs[4]): one workgroup
    // per genome would roll millions of k-mers alone.  Such a call takes the sorted form's front end -- one lane per
    // super-k-mer over the whole stream -- and leaves out the final sort only.
    const bool flat = flat_front;
    ctx->keys_job = KeysJob{has_hi, flat, n_genomes, bound, ab, false};
    if (!flat) {
        const size_t lds_d = (has_hi ? (size_t)kDedupCapHi * 28 : (size_t)kDedupCapLo * 20) + kDedupSkmWords * 4;
        if (!ctx->attr_dedupe_set) {
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_keys_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kDedupCapHi * 28 + kDedupSkmWords * 4)));
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_keys_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kDedupCapLo * 20 + kDedupSkmWords * 4)));
            ctx->attr_dedupe_set = true;
        }
        if (has_hi) hipLaunchKernelGGL((k_keys_fused<true>), dim3(n_genomes), dim3(kKeySortThreads), lds_d, ctx->stream, d_bases, packed, n_bases_readable, d_rec_off,
                                       d_sk, n, d_first_rec, p->k, w, ab, a_mn, a_lo, a_hi, d_raw_off, d_distinct, d_raw_cnt, d_big, d_flags);
        else hipLaunchKernelGGL((k_keys_fused<false>), dim3(n_genomes), dim3(kKeySortThreads), lds_d, ctx->stream, d_bases, packed, n_bases_readable, d_rec_off,
                                d_sk, n, d_first_rec, p->k, w, ab, a_mn, a_lo, a_hi, d_raw_off, d_distinct, d_raw_cnt, d_big, d_flags);
        SPSP_HIP(hipGetLastError());
    } else {
        if (n) {
            hipLaunchKernelGGL(k_keys_sizes, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_sk, n, p->k, ctx->a_cnt.as<uint32_t>());
            SPSP_HIP(hipGetLastError());
        }
        if ((rc = launch_scan_u32(ctx, ctx->a_cnt.as<uint32_t>(), ctx->a_off.as<uint32_t>(), n, ctx->h_scalar + 7))) return rc;
        if (n) {
            // one lane per place from a few places per super-k-mer on (SPSP_DEBUG_KEYS_EMIT=sk: one lane per super-k-mer always)
            static const char* dbg_emit = getenv("SPSP_DEBUG_KEYS_EMIT");
            if (dbg_emit && dbg_emit[0] == 's')
                hipLaunchKernelGGL(k_keys_emit, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_bases, packed, d_rec_off, d_sk, ctx->a_off.as<uint32_t>(), n, p->k,
                                   a_mn, a_lo, a_hi);
            else
                hipLaunchKernelGGL(k_keys_emit_places, dim3((uint32_t)((bound + kPlaceTile - 1) / kPlaceTile)), dim3(kPlaceThreads), 0, ctx->stream, d_bases, packed,
                                   d_rec_off, d_sk, ctx->a_off.as<uint32_t>(), ctx->a_cnt.as<uint32_t>(), n, p->k, a_mn, a_lo, a_hi);
            SPSP_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(k_keys_ranges, dim3((n_genomes + 1 + 255) / 256), dim3(256), 0, ctx->stream, d_sk, n, ctx->a_off.as<uint32_t>(), d_first_rec,
                           n_genomes, d_raw_off);
        SPSP_HIP(hipGetLastError());
        const size_t lds = has_hi ? (size_t)kKeyCapHi * 21 : (size_t)kKeyCapLo * 13;
        if (!ctx->attr_keys_set) {
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_keys_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kKeyCapHi * 21)));
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_keys_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kKeyCapLo * 13)));
            ctx->attr_keys_set = true;
        }
        if (has_hi) hipLaunchKernelGGL(k_keys_sort<true>, dim3(n_genomes), dim3(kKeySortThreads), lds, ctx->stream, a_mn, a_lo, a_hi, d_raw_off, ab, d_distinct, d_raw_cnt, d_big, d_flags);
        else hipLaunchKernelGGL(k_keys_sort<false>, dim3(n_genomes), dim3(kKeySortThreads), lds, ctx->stream, a_mn, a_lo, (uint64_t*)nullptr, d_raw_off, ab, d_distinct, d_raw_cnt, d_big, d_flags);
        SPSP_HIP(hipGetLastError());
    }
    // Behind the LDS kernel: the table in HBM for the genomes it flagged (their raw records lie in the staging arrays, in
    // both forms), then the compaction.  A context whose last extraction met no such genome leaves the two table kernels
    // out -- a key extraction per 0.1 ms step pays for every launch on its stream -- and runs them from _end in the call
    // that does meet one; from then on they are queued here, behind a gate word that lets them leave at once.
    static const char* force = getenv("SPSP_DEBUG_KEYS_BIG");      // test hook: "early" / "late" pins the choice
    const bool early = force ? force[0] == 'e' : ctx->keys_expect_big;
    if ((rc = keys_finish_queue(ctx, early, early ? d_flags : nullptr))) return rc;
    ctx->keys_job.big_queued = early;
    if (!ctx->keys_done) SPSP_HIP(hipEventCreateWithFlags(&ctx->keys_done, hipEventDisableTiming));
    SPSP_HIP(hipEventRecord(ctx->keys_done, ctx->stream));
    ctx->keys_pending = true;
    ctx->keys_genomes = n_genomes;
    ctx->keys_has_hi = has_hi;
    ctx->keys_sorted = !unordered;
    ctx->keys_flags_clear = true;                                  // (k_keys_compact leaves the words at zero)
    return SPSP_OK;
}

int sketch_keys_end_impl(spsp_ctx* ctx, void** d_mn, void** d_lo, void** d_hi, uint64_t* sk_off) {
    if (!ctx->keys_pending) { set_error("no key extraction is pending on this context"); return SPSP_ERR_ARG; }
    ctx->keys_pending = false;
    SPSP_HIP(hipEventSynchronize(ctx->keys_done));
    const uint32_t n_genomes = ctx->keys_genomes;
    const uint32_t* h_out = ctx->h_keys + (n_genomes + 1);
    ctx->keys_big_genomes = h_out[n_genomes + 1];
    if (ctx->keys_big_genomes && !ctx->keys_job.big_queued) {
        // the first extraction of this context that meets a genome beyond the LDS forms: the table kernels were not queued.
        // Everything they read is the context's own (raw records, segment tables): the caller's inputs are not touched.
        int rc = keys_finish_queue(ctx, true, nullptr);
        if (rc) return rc;
        SPSP_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->keys_expect_big = ctx->keys_big_genomes != 0;
    for (uint32_t g = 0; g <= n_genomes; ++g) sk_off[g] = h_out[g];
    *d_mn = ctx->c_min.p; *d_lo = ctx->c_lo.p; *d_hi = ctx->keys_has_hi ? ctx->c_hi.p : nullptr;
    if (ctx->keys_sorted && ctx->keys_big_genomes) {
        // the sorted form promises sorted sketches: the keys of the genomes the table took are distinct and in place, in no
        // order.  Sizes are known here, so the merge sort is queued from _end -- over context-owned buffers only.
        std::vector<std::pair<uint32_t, uint32_t>> segs;
        for (uint32_t g = 0; g < n_genomes; ++g)
            if (h_out[n_genomes + 2 + g]) segs.emplace_back(h_out[g], h_out[g + 1] - h_out[g]);
        const bool hh = ctx->keys_has_hi;
        const int rc = big_sort_segments(ctx, hh, ctx->c_min.as<uint32_t>(), ctx->c_lo.as<uint64_t>(), hh ? ctx->c_hi.as<uint64_t>() : nullptr,
                                         ctx->a_mn.as<uint32_t>(), ctx->a_lo.as<uint64_t>(), hh ? ctx->a_hi.as<uint64_t>() : nullptr, segs);
        if (rc) return rc;
    }
    return SPSP_OK;
}

}  // namespace spsp

using namespace spsp;

extern "C" {

int spsp_sketch_keys_device_begin(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases, const void* d_rec_off,
                                  const void* d_superkmers, uint64_t n_superkmers, const uint32_t* h_first_rec, uint32_t n_genomes, uint32_t flags) {
    if (!ctx || !p || !h_first_rec || (n_superkmers && (!d_bases || !d_rec_off || !d_superkmers))) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return sketch_keys_begin_impl(ctx, p, (const uint8_t*)d_bases, (p->flags & SPSP_SCAN_PACKED_INPUT) != 0, (const uint64_t*)d_rec_off,
                                  (const spsp_superkmer*)d_superkmers, n_superkmers, h_first_rec, n_genomes, (flags & SPSP_KEYS_UNORDERED) != 0, n_bases);
}

int spsp_sketch_keys_device_end(spsp_ctx* ctx, void** d_minimizer, void** d_kmer_lo, void** d_kmer_hi, uint64_t* sk_off) {
    if (!ctx || !d_minimizer || !d_kmer_lo || !d_kmer_hi || !sk_off) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return sketch_keys_end_impl(ctx, d_minimizer, d_kmer_lo, d_kmer_hi, sk_off);
}

uint32_t spsp_sketch_keys_big_genomes(spsp_ctx* ctx) { return ctx ? ctx->keys_big_genomes : 0u; }

int spsp_sketch_keys_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases, const void* d_rec_off,
                            const void* d_superkmers, uint64_t n_superkmers, const uint32_t* h_first_rec, uint32_t n_genomes, uint32_t flags,
                            void** d_minimizer, void** d_kmer_lo, void** d_kmer_hi, uint64_t* sk_off) {
    const int rc = spsp_sketch_keys_device_begin(ctx, p, d_bases, n_bases, d_rec_off, d_superkmers, n_superkmers, h_first_rec, n_genomes, flags);
    if (rc) return rc;
    return spsp_sketch_keys_device_end(ctx, d_minimizer, d_kmer_lo, d_kmer_hi, sk_off);
}

}  // extern "C"
