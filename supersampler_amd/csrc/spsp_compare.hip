// spsp_compare.hip -- path B on gfx950: all-vs-all bucketed k-mer intersection.
//
// The reference walks the minimizer buckets in an N-way merge and, per bucket,
// colours every canonical k-mer with the set of files holding it
// (Comparator.cpp:177-264), then adds 1 to score_A[i,j] for every pair of
// files sharing a k-mer (compute_scores :269-287).  Summed over buckets that is
//     inter[i][j] = | { (minimizer, canonical k-mer) of i } ∩ { ... of j } |
// because a k-mer only ever meets k-mers of its own bucket.
//
// GPU formulation (integer only, no MFMA):
//   1. dictionary: every distinct (minimizer, k-mer) key of the rows this rank
//      owns gets a row id (open-addressing table of 64-bit fingerprints,
//      claimed with one CAS (the winner records itself as the slot's owner);
//      every later lookup compares the FULL key with the owner's, so a
//      fingerprint collision is detected and the build retried with a new
//      seed -- results never depend on the fingerprint).
//   2. colour matrix A[row][N bits]: bit j set iff sketch j holds the key --
//      the reference's vector<bool>(N+1) colour sets, stored densely.
//   3. accumulate: for an owned sketch i, inter[i][j] = sum over i's keys of
//      bit j of the key's row: a sparse-row sum over the colour matrix.  A lane
//      owns one 64-bit word and counts its 64 columns bit-sliced (eight 64-bit
//      adds per word), so the work is ~sum_i n_i * N/64 word-adds instead of
//      the N^2 * n comparisons of pairwise merging.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

#ifndef SPSP_EXP
#define SPSP_EXP 0          // timing experiments (tools/exp/build_variant.sh): kernels with parts of their work left out; 0 in the product
#endif

namespace spsp {

struct Keys {
    const uint32_t* mn;
    const uint64_t* lo;
    const uint64_t* hi;  // may be null (k <= 32)
    uint64_t fp_mask;    // all ones; narrowed only by the collision-path test hook
};

// full key of every occupied table slot (written once by the slot's owner in the insert pass): the fill pass
// checks a candidate slot with loads that depend on the slot index only, not on a second hop through the owner
struct SlotKeys {
    uint64_t* lo;
    uint64_t* hi;
    uint32_t* mn;
};

__device__ __forceinline__ bool key_less(const Keys& K, uint64_t a, uint64_t b) {
    if (K.mn[a] != K.mn[b]) return K.mn[a] < K.mn[b];
    if (K.hi && K.hi[a] != K.hi[b]) return K.hi[a] < K.hi[b];
    return K.lo[a] < K.lo[b];
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
// Which of `passes` key classes a key belongs to, for builds whose colour matrix is capped (see run loop in
// compare_job_end).  Independent of the table fingerprint and of the exchange's rank partition (other hash bits).
__device__ __forceinline__ uint32_t pass_of(uint64_t lo, uint32_t mn, uint64_t hi, bool has_hi, uint32_t passes) {
    uint64_t h = mix64(lo ^ 0xD6E8FEB86659FD93ULL);
    h = mix64(h + (uint64_t)mn * 0xC2B2AE3D27D4EB4FULL);
    if (has_hi) h = mix64(h ^ hi);
    return (uint32_t)(((h & 0xffffffffull) * passes) >> 32);
}
__device__ __forceinline__ uint64_t fingerprint(const Keys& K, uint64_t e, uint64_t seed) {
    uint64_t f = mix64(K.lo[e] + seed);
    f = mix64(f ^ ((uint64_t)K.mn[e] * 0x9E3779B97F4A7C15ULL));
    if (K.hi) f = mix64(f + K.hi[e]);
    f &= K.fp_mask;
    return f ? f : 1;
}
__device__ __forceinline__ uint64_t home_slot(uint64_t fp, uint32_t log2cap) {
    return (fp * 0x9E3779B97F4A7C15ULL) >> (64 - log2cap);
}
// sketch that owns entry e: last j with sk_off[j] <= e
__device__ __forceinline__ uint32_t sketch_of(const uint64_t* __restrict__ sk_off, uint32_t n, uint64_t e) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sk_off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// One launch clears everything a dictionary build starts from: the table, the colour matrix (when its size is
// already known) and the per-attempt flags -- three memsets' worth of launches and gaps otherwise.
__global__ __launch_bounds__(256) void k_prepare(uint4* __restrict__ table, uint64_t table_vec, uint4* __restrict__ matrix,
                                                uint64_t matrix_vec, uint32_t* __restrict__ flags, uint32_t n_flags,
                                                const uint64_t* __restrict__ host_skoff, uint64_t* __restrict__ dev_skoff,
                                                uint32_t n_skoff) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint64_t i = t; i < table_vec; i += stride) table[i] = z;
    for (uint64_t i = t; i < matrix_vec; i += stride) matrix[i] = z;
    if (t < n_flags) flags[t] = 0;
    // the sketch offsets come straight out of the caller's (pinned, device-visible) staging copy: no separate
    // host-to-device copy, and its completion wait, in front of the pipeline
    for (uint64_t i = t; i < n_skoff; i += stride) dev_skoff[i] = host_skoff[i];
}

// flags[0]: input not strictly sorted inside a sketch; flags[1]: fingerprint collision
// rows a call owns: row_first, row_first + row_stride, ... below row_limit (strided over ranks: first < stride; a block
// of rows: stride 1)
__host__ __device__ __forceinline__ bool owned_row(uint32_t j, uint32_t row_first, uint32_t row_stride, uint32_t row_limit) {
    return j >= row_first && j < row_limit && (j - row_first) % row_stride == 0;
}

__global__ void k_insert(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n, uint64_t S, uint32_t row_first,
                         uint32_t row_stride, uint32_t row_limit, uint64_t seed, uint64_t* __restrict__ table, uint32_t log2cap,
                         uint32_t* __restrict__ owner, SlotKeys SK, uint32_t* __restrict__ flags, uint32_t passes,
                         uint32_t pass) {
    // grid.y = sketch, grid.x = 256-key chunk of it (no per-entry search for the owning sketch)
    const uint32_t j = blockIdx.y;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    if (e > sk_off[j] && !key_less(K, e - 1, e)) atomicOr(&flags[0], 1u);
    if (!owned_row(j, row_first, row_stride, row_limit)) return;   // not an owned (and printed) row
    if (passes > 1 && pass_of(K.lo[e], K.mn[e], K.hi ? K.hi[e] : 0, K.hi != nullptr, passes) != pass) return;
    const uint64_t fp = fingerprint(K, e, seed);
    const uint64_t mask = (1ull << log2cap) - 1;
    uint64_t pos = home_slot(fp, log2cap);
    for (uint64_t probes = 0;; ++probes) {
        const unsigned long long old = atomicCAS((unsigned long long*)&table[pos], 0ull, (unsigned long long)fp);
        if (old == 0ull) {   // the claiming entry is the slot's owner: its index and its full key go next to the slot
            owner[pos] = (uint32_t)e;
            SK.lo[pos] = K.lo[e]; SK.mn[pos] = K.mn[e];
            if (K.hi) SK.hi[pos] = K.hi[e];
            break;
        }
        if (old == fp) break;
        if (probes > mask) { atomicOr(&flags[5], 1u); break; }   // table full (cannot happen at load <= 1/2): never spin
        pos = (pos + 1) & mask;
    }
}

// Occupied slot -> dense row id.  Same-address atomics retire at ~90 per microsecond on this
// part, so a workgroup first counts its 16 Ki slots, reserves one run of ids with a single
// atomic, then hands them out from a workgroup-wide prefix sum.
constexpr int kRowThreads = 1024, kRowSlots = 16;
__global__ __launch_bounds__(kRowThreads) void k_assign_rows(const uint64_t* __restrict__ table, uint64_t cap,
                                                            uint32_t* __restrict__ rowid, uint32_t* __restrict__ n_rows) {
    __shared__ uint32_t wave_sum[kRowThreads / 64];
    __shared__ uint32_t s_base;
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint64_t base_slot = (uint64_t)blockIdx.x * kRowThreads * kRowSlots;
    uint32_t occ = 0;   // bit u: slot base_slot + u * kRowThreads + t is occupied
#pragma unroll
    for (int u = 0; u < kRowSlots; ++u) {
        const uint64_t sl = base_slot + (uint64_t)u * kRowThreads + t;
        if (sl < cap && table[sl] != 0) occ |= 1u << u;
    }
    const uint32_t cnt = __popc(occ);
    uint32_t x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0, total = 0;
    for (uint32_t w = 0; w < kRowThreads / 64; ++w) { if (w < wid) pre += wave_sum[w]; total += wave_sum[w]; }
    if (t == 0) s_base = total ? atomicAdd(n_rows, total) : 0u;
    __syncthreads();
    uint32_t id = s_base + pre + x - cnt;
#pragma unroll
    for (int u = 0; u < kRowSlots; ++u)
        if (occ & (1u << u)) rowid[base_slot + (uint64_t)u * kRowThreads + t] = id++;
}

// Colours: every entry of every sketch looks its key up; found => set bit j of
// the key's row, and (for owned sketches) remember the row for the accumulation.
__global__ void k_fill(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n, uint64_t S, uint32_t row_first,
                       uint32_t row_stride, uint32_t row_limit, uint64_t seed, const uint64_t* __restrict__ table, uint32_t log2cap,
                       const uint32_t* __restrict__ owner, const uint32_t* __restrict__ rowid, SlotKeys SK, uint32_t W,
                       unsigned long long* __restrict__ A, uint32_t* __restrict__ row_of_entry,
                       uint32_t* __restrict__ flags, uint32_t passes, uint32_t pass) {
    const uint32_t j = blockIdx.y;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    if (passes > 1 && pass_of(K.lo[e], K.mn[e], K.hi ? K.hi[e] : 0, K.hi != nullptr, passes) != pass) {
        if (owned_row(j, row_first, row_stride, row_limit)) row_of_entry[e] = 0xffffffffu;   // not in this pass: k_accumulate skips it
        return;
    }
    // every exit that does not find the key's row (collision, full table: attempts the host discards) must still leave
    // a valid marker behind, or the row sums queued behind this pass would follow a stale index
    if (owned_row(j, row_first, row_stride, row_limit)) row_of_entry[e] = 0xffffffffu;
    const uint64_t fp = fingerprint(K, e, seed);
    const uint64_t mask = (1ull << log2cap) - 1;
    uint64_t pos = home_slot(fp, log2cap);
    for (uint64_t probes = 0;; ++probes) {
        if (probes > mask) { atomicOr(&flags[5], 1u); return; }
        const uint64_t v = table[pos];
        if (v == 0) return;  // key not held by any owned sketch: contributes to no owned row
        if (v == fp) {
            const bool own = owned_row(j, row_first, row_stride, row_limit);
            const bool same = SK.lo[pos] == K.lo[e] && SK.mn[pos] == K.mn[e] && (!K.hi || SK.hi[pos] == K.hi[e]);
            if (same) {
                const uint32_t r = rowid ? rowid[pos] : owner[pos];   // direct mode: the owner's entry index is the row
                atomicOr(&A[(uint64_t)r * W + (j >> 6)], 1ull << (j & 63));
                if (own) row_of_entry[e] = r;
            } else if (own) {
                // an owned key was inserted under this fingerprint, so this IS its slot: a different full key
                // here means two distinct keys share a fingerprint -> the host rebuilds with another seed
                atomicOr(&flags[1], 1u);
            }
            return;  // equal keys share the first slot with this fingerprint; nothing further down matches
        }
        pos = (pos + 1) & mask;
    }
}

// ---------------------------------------------------------------------------
// Sparse form, for many sketches.  A key is held by a handful of sketches however many there are, so a dense
// colour row (N bits) is almost all zeros once N reaches the thousands: at 10^4 sketches the matrix is 21 GB
// and the row sums stream 65 GB.  Here every distinct key gets the LIST of the sketches holding it (u16 ids,
// length first) and the row sums walk the lists:
//   k_insert_sparse  CAS insert + one count per key on its slot; the slot of every entry is remembered
//   k_assign_ranges  slot -> list position (workgroup prefix + one atomic per 16 Ki slots), writes the lengths
//   k_fill_sparse    full-key check (collision flag), then each entry takes a place in its slot's list (count-down)
//   k_accumulate_sparse  one workgroup per sketch and block of 16 Ki columns: LDS counters, one LDS add per
//                    (key of i, other holder of the key)
// Traffic per key is its list (a few bytes) instead of N/8 bytes.  Used when all rows are owned (single GPU).
// A sketch list = u16 length, then the u16 ids of the sketches holding the key, starting on a 16-byte boundary
// of `ids`.  Its reference (one u32 per sketch entry) carries the place in 16-byte units (26 bits: 2^29 u16 of
// lists) and, for lists shorter than 63, the length -- so the row sums issue the list's loads without first
// waiting for its length word.
constexpr uint32_t kRefOffBits = 26, kRefOffMask = (1u << kRefOffBits) - 1, kRefLenMax = 63;
__host__ __device__ __forceinline__ uint32_t list_u16(uint32_t len) { return (len + 1 + 7) & ~7u; }
__device__ __forceinline__ uint32_t list_ref(uint32_t o, uint32_t len) { return (o >> 3) | ((len < kRefLenMax ? len : kRefLenMax) << kRefOffBits); }
constexpr uint32_t kNoList = 0xffffffffu;       // the key takes part in no pair of this job
constexpr uint32_t kNoWhere = 0xffffffffu;      // partition form: the key's record did not fit its part (the call is redone)

__global__ void k_insert_sparse(Keys K, const uint64_t* __restrict__ sk_off, uint64_t seed, uint64_t* __restrict__ table,
                                uint32_t log2cap, uint32_t* __restrict__ cnt, SlotKeys SK, uint32_t* __restrict__ slot_of_entry,
                                uint32_t* __restrict__ flags) {
    const uint32_t j = blockIdx.y;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    if (e > sk_off[j] && !key_less(K, e - 1, e)) atomicOr(&flags[0], 1u);
    const uint64_t fp = fingerprint(K, e, seed);
    const uint64_t mask = (1ull << log2cap) - 1;
    uint64_t pos = home_slot(fp, log2cap);
    for (uint64_t probes = 0;; ++probes) {
        const unsigned long long old = atomicCAS((unsigned long long*)&table[pos], 0ull, (unsigned long long)fp);
        if (old == 0ull) {
            SK.lo[pos] = K.lo[e]; SK.mn[pos] = K.mn[e];
            if (K.hi) SK.hi[pos] = K.hi[e];
            break;
        }
        if (old == fp) break;
        if (probes > mask) { atomicOr(&flags[5], 1u); slot_of_entry[e] = 0xffffffffu; return; }   // later passes skip the entry
        pos = (pos + 1) & mask;
    }
    atomicAdd(&cnt[pos], 1u);                   // result unused: a returning atomic here costs 60 % more (measured)
    slot_of_entry[e] = (uint32_t)pos;
}

__global__ __launch_bounds__(kRowThreads) void k_assign_ranges(const uint32_t* __restrict__ cnt, uint64_t cap,
                                                              uint32_t* __restrict__ off, uint16_t* __restrict__ ids,
                                                              uint32_t* __restrict__ total) {
    __shared__ uint32_t wave_sum[kRowThreads / 64];
    __shared__ uint32_t s_base;
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint64_t base_slot = ((uint64_t)blockIdx.x * kRowThreads + t) * kRowSlots;   // 16 consecutive slots per lane
    uint32_t c[kRowSlots];
    uint32_t sum = 0;
#pragma unroll
    for (int u = 0; u < kRowSlots; ++u) {
        const uint64_t sl = base_slot + u;
        c[u] = sl < cap ? cnt[sl] : 0u;
        sum += c[u] ? list_u16(c[u]) : 0u;      // list = length word + ids, padded to whole 16-byte words
    }
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    for (uint32_t w = 0; w < kRowThreads / 64; ++w) { if (w < wid) pre += wave_sum[w]; all += wave_sum[w]; }
    if (t == 0) s_base = all ? atomicAdd(total, all) : 0u;
    __syncthreads();
    uint32_t at = s_base + pre + x - sum;
#pragma unroll
    for (int u = 0; u < kRowSlots; ++u) {
        if (!c[u]) continue;
        off[base_slot + u] = at;
        ids[at] = (uint16_t)c[u];               // a key is held at most once per sketch: length <= 65535
        at += list_u16(c[u]);
    }
}

__global__ void k_fill_sparse(Keys K, const uint64_t* __restrict__ sk_off, uint32_t* __restrict__ slot_of_entry, SlotKeys SK,
                              uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, uint16_t* __restrict__ ids,
                              uint32_t* __restrict__ flags) {
    const uint32_t j = blockIdx.y;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    const uint32_t pos = slot_of_entry[e];
    if (pos == 0xffffffffu) return;             // table overflow was flagged by the insert pass
    const bool same = SK.lo[pos] == K.lo[e] && SK.mn[pos] == K.mn[e] && (!K.hi || SK.hi[pos] == K.hi[e]);
    if (!same) atomicOr(&flags[1], 1u);         // two distinct keys, one fingerprint: the host rebuilds with another seed
    const uint32_t o = off[pos];
    const uint32_t idx = atomicSub(&cnt[pos], 1u);   // counts down c .. 1: the places behind the length word
    ids[o + idx] = (uint16_t)j;
    slot_of_entry[e] = list_ref(o, ids[o]);     // the row sums go straight to the list
}

constexpr int kSparseCols = 16384;
constexpr int kFlags = 8;   // [0] unsorted input, [1] fingerprint collision, [2] n_rows (dictionary forms) / records of overflowed parts (partition
                            // form: sizes the spill), [3] malformed slot, [4] slot overflow, [5] table full, [6] a key part overflowed its
                            // capacity (partition form), [7] records dealt into parts -- these eight travel to the host with every job.
                            // Device-side words behind them (16 in all, cleared by k_parts_prepare): [8] u16 of spilled lists handed out,
                            // [9] bit columns handed out, [10] / [11] k_row_order's verdict, [12] / [13] records with a list / records in a
                            // sample of the parts (k_parts_group), [14..15] the cell count of a comparison returned as cells
// grid.y = owned row (sketch row_first + y * row_stride), grid.x = block of `cols` columns, grid.z = slice of the
// row's keys (split > 1: the slices add into cells zeroed by k_zero_rows).  Counters live in dynamic LDS,
// `copies` of each, interleaved (counter c of copy k at c * copies + k): the lists of one row's keys name the
// same few sketches over and over, and 64 lanes adding to ONE LDS word serialise -- with a copy per lane
// group the same column is spread over `copies` banks.  A thread takes kAccU keys at a time: the list
// references are one coalesced load and the lists' first words are issued together.
constexpr int kSparseThreads = 1024, kAccU = 2, kAccW = 4, kAccR = 2;
// HALF: 16-bit counters, two to a word (no row holds 65 536 keys): half the LDS per workgroup
// TOUCH (cells out, HALF, one column block, no split): the workgroup STAYS and takes row after row (rows_y of them in all:
// what grid.y would have been), and a row costs what it touches -- the counters are cleared once, whoever makes a counter
// non-zero notes it in a list, the row's cells are the list's columns and only they are cleared again.  Without it a row
// costs a clear and a scan of all N counters whatever it holds: a key-partitioned rank's 10 000 rows of ~600 keys were 0.20 ms
// of that and of workgroup starts.  A row that touches more than kTouchCap counters (a family of thousands) is scanned as before.
constexpr uint32_t kTouchCap = 1024;
template <bool HALF, bool TOUCH, int T>
__global__ __launch_bounds__(T) void k_accumulate_sparse(const uint32_t* __restrict__ list_of_entry,
                                                                     const uint32_t* __restrict__ where,
                                                                     const uint16_t* __restrict__ ids,
                                                                     const uint64_t* __restrict__ sk_begin,
                                                                     const uint64_t* __restrict__ sk_end, uint32_t n,
                                                                     uint32_t row_first, uint32_t row_stride, uint32_t row_limit,
                                                                     uint32_t cols, uint32_t copies_log2, uint32_t split,
                                                                     uint32_t* __restrict__ inter,
                                                                     const uint32_t* __restrict__ flags,
                                                                     uint32_t* __restrict__ host_flags,
                                                                     unsigned long long* __restrict__ cells, unsigned long long cells_cap,
                                                                     unsigned long long* __restrict__ cells_count, uint32_t xcd_rows, bool add,
                                                                     const uint32_t* __restrict__ row_order, uint32_t long_limit,
                                                                     const uint32_t* __restrict__ multi, uint32_t multi_slots, uint32_t rows_y) {
    static_assert(!TOUCH || HALF, "the touched-counter form packs 16-bit counters");
    extern __shared__ uint32_t s_cnt[];
    __shared__ uint32_t s_list[TOUCH ? kTouchCap : 1];
    __shared__ uint32_t s_nt;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < kFlags) host_flags[threadIdx.x] = flags[threadIdx.x];
    // a part overflowed: its workgroup of k_parts_group left without writing list references, so the ones in place are
    // whatever the buffer held before -- nothing may be followed from them (the host repeats the call with more parts)
    if (flags[6]) return;
    // Which row: workgroups go to the 8 XCDs round-robin by their flat number, so consecutive rows -- the members of a family,
    // which hold the same keys and therefore fetch the SAME holder lists -- would run on eight different L2s.  With xcd_rows
    // (= rows per XCD; grids of one column block and one slice only) XCD x takes the rows [x xcd_rows, (x + 1) xcd_rows) in
    // order: a family's members run side by side behind one L2 and a list line comes from HBM once per family, not once per
    // member (MI355X_MICROARCH.md: 4 MiB of L2 per XCD, not coherent across XCDs; a family's lists are ~0.25 MB).
    const uint32_t copies = 1u << copies_log2, mine = threadIdx.x & (copies - 1);
    if (TOUCH) {
        for (uint32_t x = threadIdx.x; x < ((cols << copies_log2) >> 1); x += T) s_cnt[x] = 0;
        if (threadIdx.x == 0) s_nt = 0;
        __syncthreads();
    }
    if (TOUCH) { split = 1; add = false; long_limit = 0; }                     // (what the host guarantees for this form: the code for the rest folds away)
    uint32_t by = blockIdx.y;
    // (rows in fixed strides: drawing them from a per-XCD ticket counter was measured for the long rows of an all-vs-all call --
    // where this form loses anyway, see the host side -- and for a rank's short ones, 0.161 -> 0.171 ms: not kept)
    do {                                                                       // (TOUCH: row after row; else once -- not a loop to the compiler)
    uint32_t r = by;
    if (xcd_rows) { r = (by & 7u) * xcd_rows + (by >> 3); if ((by >> 3) >= xcd_rows) continue; }
    if (row_order && r >= n) continue;
    // (row_order: every row owned; flags[10] / flags[11] = sketches with a sketch of the same signature close in front of them in the
    // new order / in the input's: the new order is taken when the input keeps fewer than half as many together)
    const bool reorder = row_order && 2u * flags[11] < flags[10];
    const uint32_t i = reorder ? row_order[r] : row_first + r * row_stride, col0 = TOUCH ? 0u : blockIdx.x * cols;
    if (i >= n || i >= row_limit) continue;
    if (col0 + cols <= i + 1 || i + 1 >= n) continue;       // no column > i in this block / at all
    uint64_t e0 = sk_begin[i], e1 = sk_end[i];
    if (long_limit && e1 - e0 > long_limit) continue;       // a row far longer than the others: summed in slices by a launch of its own
    if (split > 1) {
        const uint64_t per = (e1 - e0 + split - 1) / split;
        e0 += per * blockIdx.z;
        if (e0 + per < e1) e1 = e0 + per;
        if (e0 >= e1) continue;
    }
    if (!TOUCH) {
        for (uint32_t x = threadIdx.x; x < ((cols << copies_log2) >> (HALF ? 1 : 0)); x += T) s_cnt[x] = 0;
        __syncthreads();
    }
#if SPSP_EXP & 8
    uint32_t exp_chk = 0;
#endif
    const uint32_t lane = threadIdx.x & 63u;
    auto count = [&](uint32_t jj) {
#if SPSP_EXP & 8
        exp_chk += jj;                                        // (timing experiment: the lists are read, nothing is added in LDS)
        return;
#endif
#if SPSP_EXP & 64
        if (jj > i && jj - col0 < cols) { atomicAdd(&s_cnt[(threadIdx.x * 33u + jj) & 4095u], 1u); }   // (timing experiment: adds without same-address conflicts)
        return;
#endif
        if (jj > i && jj - col0 < cols) {
            const uint32_t idx = ((jj - col0) << copies_log2) + mine;
            if (TOUCH) {
                const uint32_t sh = (idx & 1u) << 4;
                const uint32_t old = atomicAdd(&s_cnt[idx >> 1], 1u << sh);
                if (((old >> sh) & 0xffffu) == 0) {            // this counter's first: the lanes that have one now note them together
                    const unsigned long long firsts = __ballot(1);
                    const int leader = __ffsll((long long)firsts) - 1;
                    uint32_t base = 0;
                    if ((int)lane == leader) base = atomicAdd(&s_nt, (uint32_t)__popcll(firsts));
                    base = __shfl(base, leader);
                    const uint32_t at_l = base + (uint32_t)__popcll(firsts & ((1ull << lane) - 1ull));
                    if (at_l < kTouchCap) s_list[at_l] = idx;
                }
            } else if (HALF) atomicAdd(&s_cnt[idx >> 1], 1u << ((idx & 1u) << 4));    // (a half never carries: a counter is at most the row's key count)
            else atomicAdd(&s_cnt[idx], 1u);
        }
    };
    auto counter = [&](uint32_t idx) { return HALF ? (s_cnt[idx >> 1] >> ((idx & 1u) << 4)) & 0xffffu : s_cnt[idx]; };
    // kAccR list references are fetched together (the partition form reaches them through `where`: two dependent
    // loads, the second scattered -- eight of each in flight per thread hide the extra hop), then the lists kAccU at a time
    const bool use_multi = multi && (multi_slots >> 31 ? true : 5u * flags[12] < 2u * flags[13]);   // (top bit of multi_slots: the host insists, SPSP_DEBUG_MULTI=1)     // (records with a list / records, in a sample of the parts: k_parts_group)
    for (uint64_t eb = e0; eb < e1; eb += (uint64_t)kAccR * T) {     // (every lane makes every round: the waves pool their long lists, below)
        const uint64_t e = eb + threadIdx.x;
        uint32_t refs[kAccR];
        if (where) {                                         // the reference sits where the key's record went
            uint32_t at[kAccR];
#pragma unroll
            for (int r = 0; r < kAccR; ++r) { const uint64_t eu = e + (uint64_t)r * T; at[r] = eu < e1 ? where[eu] : kNoWhere; }
#pragma unroll
            for (int r = 0; r < kAccR; ++r) {
                // (multi: a bit per record slot of the parts -- has the key a list? -- in front of the list reference's miss)
                bool fetch = at[r] != kNoWhere;
                if (fetch && use_multi && at[r] < (multi_slots & 0x7fffffffu)) fetch = (multi[at[r] >> 5] >> (at[r] & 31u)) & 1u;
#if SPSP_EXP & 32
                refs[r] = fetch && at[r] == 0xfffffff1u ? list_of_entry[at[r]] : kNoList;   // (timing experiment: no reference read)
#else
                refs[r] = fetch ? list_of_entry[at[r]] : kNoList;
#endif
#if SPSP_EXP & 16
                if (refs[r] != 0xfffffff2u) refs[r] = kNoList;                              // (timing experiment: no list read)
#endif
            }
        } else {
#pragma unroll
            for (int r = 0; r < kAccR; ++r) { const uint64_t eu = e + (uint64_t)r * T; refs[r] = eu < e1 ? list_of_entry[eu] : kNoList; }
        }
#pragma unroll
        for (int g = 0; g < kAccR; g += kAccU) {
        uint32_t ref[kAccU];
#pragma unroll
        for (int u = 0; u < kAccU; ++u) ref[u] = refs[g + u];
        // the first kAccW 16-byte words of each SHORT list (length + 31 ids) are requested together; of a longer list only the first
        uint4 w[kAccU][kAccW];
#pragma unroll
        for (int u = 0; u < kAccU; ++u) {
            const uint4* L = reinterpret_cast<const uint4*>(ids) + (ref[u] & kRefOffMask);
            const uint32_t lenf = ref[u] >> kRefOffBits;     // (63 = "63 or more")
#pragma unroll
            for (int q = 0; q < kAccW; ++q)
                w[u][q] = (ref[u] != kNoList && (q == 0 || (lenf >= 8u * q && lenf < 8u * kAccW))) ? L[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kAccU; ++u) {
            const bool has = ref[u] != kNoList;
            uint32_t len = 0;
            if (has) { len = ref[u] >> kRefOffBits; if (len == kRefLenMax) len = w[u][0].x & 0xffffu; }
            // list element t (element 0 is the length, ids are 1..len) = half-word t & 7 of word t >> 3
            auto word = [&](const uint4& v, uint32_t first, uint32_t ln) {
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t h = 0; h < 8; ++h) {
                    const uint32_t t = first + h;
                    if (t >= 1 && t <= ln) count((d[h >> 1] >> (16 * (h & 1))) & 0xffffu);
                }
            };
            const bool is_long = has && len >= 8u * kAccW;
            if (has) {
                word(w[u][0], 0u, len);
                if (!is_long) {
#pragma unroll
                    for (int q = 1; q < kAccW; ++q) if (len >= 8u * q) word(w[u][q], 8u * q, len);
                }
            }
            // LONG lists (keys of a large family) are read by the wave TOGETHER.  Lane by lane -- every lane walking its own list,
            // 16 bytes at a time -- a wave's load touches 64 different lines for 1 KiB of use, the lines of 2 048 lanes do not fit the
            // CU's vector cache, and every 16 bytes cost a 128-byte line from the L2 again: 19 GB of lists were 150 GB of L2 traffic at
            // 317 holders per key.  The words 1.. of the wave's long lists form ONE sequence (a prefix sum over the lanes); lane l of
            // round r takes word 64 r + l of it -- whole lines, every lane busy, two rounds in flight -- and finds the list it belongs
            // to by a binary search over the lanes' prefix (shuffles).
            if (__any(is_long)) {
                const uint32_t nwp = is_long ? (len >> 3) : 0u;
                uint32_t incl = nwp;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += y; }
                const uint32_t excl = incl - nwp, total = __shfl(incl, 63);
                auto locate = [&](uint32_t gw, uint32_t& r_, uint32_t& ln_, uint32_t& wi_) {
                    uint32_t lo = 0;
#pragma unroll
                    for (uint32_t step = 32; step; step >>= 1) {
                        const uint32_t cand = lo + step;
                        const uint32_t pc = __shfl(excl, cand & 63u);
                        if (cand < 64u && pc <= gw) lo = cand;
                    }
                    r_ = __shfl(ref[u], lo); ln_ = __shfl(len, lo);
                    wi_ = gw - __shfl(excl, lo) + 1u;
                };
                for (uint32_t g0 = 0; g0 < total; g0 += 128u) {
                    uint32_t rA, lnA, wA, rB, lnB, wB;
                    const uint32_t gA = g0 + lane, gB = g0 + 64u + lane;
                    locate(gA < total ? gA : 0u, rA, lnA, wA);
                    locate(gB < total ? gB : 0u, rB, lnB, wB);
                    const uint4 vA = gA < total ? (reinterpret_cast<const uint4*>(ids) + (rA & kRefOffMask))[wA] : make_uint4(0, 0, 0, 0);
                    const uint4 vB = gB < total ? (reinterpret_cast<const uint4*>(ids) + (rB & kRefOffMask))[wB] : make_uint4(0, 0, 0, 0);
                    if (gA < total) word(vA, 8u * wA, lnA);
                    if (gB < total) word(vB, 8u * wB, lnB);
                }
            }
        }
        }
    }
#if SPSP_EXP & 8
    if (exp_chk == 0x12345u) s_cnt[1] = exp_chk;
#endif
    __syncthreads();
    if (TOUCH) {
        const uint32_t nt = s_nt;
        if (nt <= kTouchCap) {
            // the row's cells: the columns in the list (a column with several copies of its counter is in the list once per copy
            // that counted: the lowest such copy speaks for it)
            const uint32_t rounds = (nt + 63u) >> 6;
            for (uint32_t q = threadIdx.x >> 6; q < rounds; q += T / 64) {
                const uint32_t t = (q << 6) + lane;
                uint32_t v = 0, x = 0;
                if (t < nt) {
                    const uint32_t idx = s_list[t];
                    x = idx >> copies_log2;
                    bool speaks = true;
                    for (uint32_t k = 0; k < copies; ++k) {
                        const uint32_t c = counter((x << copies_log2) + k);
                        if (c && k < (idx & (copies - 1u))) speaks = false;
                        v += c;
                    }
                    if (!speaks) v = 0;
                }
                const unsigned long long hit = __ballot(v != 0);
                if (hit) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(cells_count, (unsigned long long)__popcll(hit));
                    base = __shfl(base, 0);
                    const unsigned long long mine_at = base + (unsigned long long)__popcll(hit & ((1ull << lane) - 1ull));
                    if (v && mine_at < cells_cap) cells[mine_at] = ((unsigned long long)i << 48) | ((unsigned long long)(col0 + x) << 32) | v;
                }
            }
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < nt; t += T) s_cnt[s_list[t] >> 1] = 0;     // (both halves: the other is zero or in the list too)
            if (threadIdx.x == 0) s_nt = 0;
            __syncthreads();
            continue;
        }
    }
    if (TOUCH) {
        // a row that touched more counters than the list holds (a family of thousands): every counter looked at, every counter cleared
#pragma unroll 1
        for (uint32_t x0 = 0; x0 < cols; x0 += T) {
            const uint32_t x = x0 + threadIdx.x;
            uint32_t v = 0;
            if (x < cols && x > i && x < n)
                for (uint32_t k = 0; k < copies; ++k) v += counter((x << copies_log2) + k);
            const unsigned long long hit = __ballot(v != 0);
            if (hit) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(cells_count, (unsigned long long)__popcll(hit));
                base = __shfl(base, 0);
                const unsigned long long mine_at = base + (unsigned long long)__popcll(hit & ((1ull << lane) - 1ull));
                if (v && mine_at < cells_cap) cells[mine_at] = ((unsigned long long)i << 48) | ((unsigned long long)x << 32) | v;
            }
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < ((cols << copies_log2) >> 1); x += T) s_cnt[x] = 0;
        if (threadIdx.x == 0) s_nt = 0;
        __syncthreads();
        continue;
    }
    if (cells) {
        // sparse result (the key-partitioned split: a rank's partial row is nearly all zeros): the row's non-zero cells
        // leave as packed words i << 48 | j << 32 | count -- the dense row is not written.  One atomic per WAVE that has any
        // (not per wave and column step: all of them hit the same counter, 11 ns each when millions queue up -- 75 ms at
        // 65 535 sketches of unrelated genomes against 5 ms for the dense form), and with several column blocks per row ONE
        // per workgroup: the waves count first, the counters stay in registers.
        constexpr int kSteps = kSparseCols / T;
        __shared__ uint32_t s_wave[T / 64];
        __shared__ unsigned long long s_base;
        const uint32_t wave = threadIdx.x >> 6;
        auto value = [&](int st) -> uint32_t {
            const uint32_t x = threadIdx.x + (uint32_t)st * T, col = col0 + x;
            uint32_t v = 0;
            if (x < cols && col > i && col < n)
                for (uint32_t k = 0; k < copies; ++k) v += counter((x << copies_log2) + k);
            return v;
        };
        // the wave counts first (which of its column steps have any cell: most have none), then writes those steps only
        static_assert(kSteps <= 64, "one bit per column step");
        uint32_t wave_total = 0;
        unsigned long long steps = 0;
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            const unsigned long long hit = __ballot(value(st) != 0);
            if (hit) { wave_total += (uint32_t)__popcll(hit); steps |= 1ull << st; }
        }
        unsigned long long at;
        if (gridDim.x == 1) {
            // one column block per row (up to 16 Ki sketches): at most 16 waves x rows atomics and no barrier -- the two barriers of
            // the workgroup form were 0.03 ms of the 0.18 ms a key-partitioned rank's 10 000 short rows take
            unsigned long long base = 0;
            if (lane == 0 && wave_total) base = atomicAdd(cells_count, (unsigned long long)wave_total);
            at = __shfl(base, 0);
        } else {
            if (lane == 0) s_wave[wave] = wave_total;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t total = 0;
                for (uint32_t w = 0; w < T / 64; ++w) { const uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
                s_base = total ? atomicAdd(cells_count, (unsigned long long)total) : 0ull;
            }
            __syncthreads();
            at = s_base + s_wave[wave];
        }
        while (steps) {                                  // (wave-uniform: the mask came from ballots)
            const int st = __ffsll((long long)steps) - 1;
            steps &= steps - 1;
            const uint32_t v = value(st);
            const unsigned long long hit = __ballot(v != 0);
            const unsigned long long mine_at = at + (unsigned long long)__popcll(hit & ((1ull << lane) - 1ull));
            if (v && mine_at < cells_cap)
                cells[mine_at] = ((unsigned long long)i << 48) | ((unsigned long long)(col0 + threadIdx.x + (uint32_t)st * T) << 32) | v;
            at += (unsigned long long)__popcll(hit);
        }
        continue;
    }
    for (uint32_t x = threadIdx.x; x < cols; x += T) {
        const uint32_t col = col0 + x;
        if (col > i && col < n) {
            uint32_t v = 0;
            for (uint32_t k = 0; k < copies; ++k) v += counter((x << copies_log2) + k);
            if (split > 1) { if (v) atomicAdd(&inter[(uint64_t)i * n + col], v); }
            else if (add) { if (v) inter[(uint64_t)i * n + col] += v; }     // (a later key class of a very large input: the cell is this workgroup's alone)
            else inter[(uint64_t)i * n + col] = v;
        }
    }
    } while (TOUCH && (by += gridDim.y) < rows_y);
}
// cells (i, j > i) of the owned rows = 0 (only needed when the row sums are split over several workgroups)
__global__ __launch_bounds__(256) void k_zero_rows(uint32_t n, uint32_t row_first, uint32_t row_stride, uint32_t row_limit,
                                                  uint32_t* __restrict__ inter) {
    const uint32_t i = row_first + blockIdx.y * row_stride;
    if (i >= n || i >= row_limit) return;
    for (uint32_t col = i + 1 + blockIdx.x * 256 + threadIdx.x; col < n; col += gridDim.x * 256) inter[(uint64_t)i * n + col] = 0;
}

// ---------------------------------------------------------------------------
// Partition form: the dictionary without global atomics.  Equal keys have equal hashes, so the keys are first
// dealt into P hash classes ("parts", a few thousand records each) and every part is then grouped by ONE
// workgroup entirely in LDS -- full keys are compared, so there is no fingerprint and no collision retry.
//   k_parts_prepare   part counters and flags cleared, sketch offsets brought over from the pinned staging copy
//   k_parts_scatter   record {kmer_lo, minimizer | sketch << 32, entry, [kmer_hi]} -> its part; a workgroup counts
//                     its chunk per part in LDS and reserves room with ONE global atomic per part it touches
//   k_parts_group     per part: LDS hash of record indices (CAS claims a slot, later holders of the key chain
//                     themselves in with an exchange); every chain head then writes the key's sketch list
//                     (length + u16 ids) into the part's slice of `ids` and the list's place for each of its entries
//   k_accumulate_sparse  as in the sketch-list form above: row sums by walking the lists
// Parts have a fixed capacity; one that overflows (heavily duplicated keys) raises flags[6] and the host
// falls back to the global-dictionary forms.
constexpr uint32_t kColFlag = 0x80000000u;      // in place of a list offset: the key has a column of the bit matrix, not a list (k_spill_pairs)
constexpr int kPartCap = 4096, kPartSlots = 7936, kGroupThreads = 1024;   // (record index + 1 fits the 13 low bits of a slot word)
constexpr int kScatThreads = 1024, kScatPer = 4, kScatSub = kScatThreads * kScatPer;   // entries per sub-chunk
constexpr int kMaxKeyParts = 32000;   // x ~2 900 records: 9 x 10^7 keys per comparison (BASELINE configs[3] has 5.2 x 10^7); x 16 Ki u16 of lists
                                      // each stays inside the 2^29 u16 a list reference can address

// key class of a pass (bits the part and the slot of a key depend on least)
__device__ __forceinline__ uint32_t key_class(uint64_t h, uint32_t classes) { return (uint32_t)((((h >> 8) & 0xffffffull) * classes) >> 24); }
__device__ __forceinline__ uint64_t key_hash(uint64_t lo, uint32_t mn, uint64_t hi, bool has_hi) {
    uint64_t h = mix64(lo ^ 0xA0761D6478BD642FULL);
    h = mix64(h + (uint64_t)mn * 0xE7037ED1A0B428DBULL);
    if (has_hi) h = mix64(h ^ hi);
    return h;
}

// Rows of similar sketches side by side, whatever order the sketches came in.  The row sums deal the rows to the XCDs in
// runs so that the rows of a family -- which fetch the same holder lists -- sit behind one L2; that presumes families come
// in runs.  sig[j] = the smallest key hash of sketch j (a min-hash: two sketches share it with the probability of their
// Jaccard index).  A key's part is the top of its hash, so a sketch's smallest hash lies in the first part it has a key in:
// the records of the first few parts hold it for every sketch (64 parts: ~19 keys of each).
constexpr uint32_t kSigParts = 64;
template <bool HAS_HI>
__global__ __launch_bounds__(1024) void k_row_signature(const uint64_t* __restrict__ recs, const uint32_t* __restrict__ part_cnt, uint32_t cap,
                                                       unsigned long long* __restrict__ sig) {
    constexpr uint32_t W = HAS_HI ? 3 : 2;
    const uint32_t p = blockIdx.x, n = min(part_cnt[p], cap);
    const uint64_t* base = recs + (uint64_t)p * cap * W;
    for (uint32_t r0 = 0; r0 < n; r0 += 1024) {          // (every lane makes every round: the waves vote)
        const uint32_t r = r0 + threadIdx.x;
        const bool valid = r < n;
        uint32_t sk = 0;
        unsigned long long h = ~0ull;
        if (valid) {
            const uint64_t lo = base[(uint64_t)r * W], w1 = base[(uint64_t)r * W + 1], hi = HAS_HI ? base[(uint64_t)r * W + 2] : 0ull;
            sk = (uint32_t)(w1 >> 32);
            h = (unsigned long long)key_hash(lo, (uint32_t)w1, hi, HAS_HI);
        }
        // records reach a part in runs of one sketch: a wave of one sketch sends ONE atomic (a sketch of 3 x 10^6 keys among
        // 10 000 small ones had 11 000 records here, all on one address: 0.12 ms)
        const unsigned long long live = __ballot(valid);
        if (!live) continue;
        const uint32_t sk0 = __shfl(sk, __ffsll((long long)live) - 1);
        if (__all(!valid || sk == sk0)) {
#pragma unroll
            for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(h, d); h = o < h ? o : h; }
            if ((threadIdx.x & 63u) == 0) atomicMin(&sig[sk0], h);
        } else if (valid) atomicMin(&sig[sk], h);
    }
}
// The order: sketches dealt into 8 192 buckets by the top of their signature (a counting sort in ONE workgroup: equal
// signatures land in the same bucket, a bucket holds one or two sketches on average, n <= 16 Ki).  stats[0] += sketches that
// have a sketch of the same signature among the 32 in front of them in this order, stats[1] += the same count in INPUT order:
// the row sums take the new order only when the input keeps fewer than half as many together -- sketches that come family by
// family are already in a better order than one min-hash can make.
constexpr uint32_t kOrderBuckets = 8192, kOrderWindow = 16, kOrderMost = 16384;
__global__ __launch_bounds__(1024) void k_row_order(const unsigned long long* __restrict__ sig, uint32_t n, uint32_t* __restrict__ order, uint32_t* __restrict__ stats) {
    extern __shared__ uint32_t lds_o[];
    uint32_t* hist = lds_o;                       // [kOrderBuckets]
    uint32_t* tag = lds_o + kOrderBuckets;        // [n]: the signatures, mixed once more (a smallest hash has no high bits), 32 bits of them
    __shared__ uint32_t wave_sum[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wid = t >> 6;
    for (uint32_t b = t; b < kOrderBuckets; b += 1024) hist[b] = 0;
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024) {
        const uint32_t g = (uint32_t)(mix64(sig[i]) >> 32);
        tag[i] = g;
        atomicAdd(&hist[g >> 19], 1u);
    }
    __syncthreads();
    uint32_t near_new = 0, near_in = 0;
    for (uint32_t p = t; p < n; p += 1024) {      // input order: a sketch of the same signature among the kOrderWindow in front?
        const uint32_t a = tag[p];
        bool f = false;
        for (uint32_t w = 1; w <= kOrderWindow && w <= p; ++w) f |= tag[p - w] == a;
        near_in += f ? 1u : 0u;
    }
    // exclusive prefix over the buckets, 8 per thread
    uint32_t c[8], sum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { c[u] = hist[t * 8 + u]; sum += c[u]; }
    uint32_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < wid; ++w) pre += wave_sum[w];
    uint32_t at = pre + x - sum;
#pragma unroll
    for (int u = 0; u < 8; ++u) { hist[t * 8 + u] = at; at += c[u]; }
    __syncthreads();
    uint32_t mine[kOrderMost / 1024], where_[kOrderMost / 1024];
#pragma unroll
    for (uint32_t u = 0; u < kOrderMost / 1024; ++u) {
        const uint32_t i = t + u * 1024;
        mine[u] = 0; where_[u] = 0;
        if (i < n) { mine[u] = tag[i]; where_[u] = atomicAdd(&hist[mine[u] >> 19], 1u); order[where_[u]] = i; }
    }
    __syncthreads();                              // every tag has been read: the array now holds them in the new order
#pragma unroll
    for (uint32_t u = 0; u < kOrderMost / 1024; ++u) if (t + u * 1024 < n) tag[where_[u]] = mine[u];
    __syncthreads();
    for (uint32_t p = t; p < n; p += 1024) {
        const uint32_t a = tag[p];
        bool f = false;
        for (uint32_t w = 1; w <= kOrderWindow && w <= p; ++w) f |= tag[p - w] == a;
        near_new += f ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) { near_new += __shfl_xor(near_new, d); near_in += __shfl_xor(near_in, d); }
    if (lane == 0) { if (near_new) atomicAdd(&stats[0], near_new); if (near_in) atomicAdd(&stats[1], near_in); }
}

// sub_sk[c] = sketch holding entry c * kScatSub (worked out by the host, which has the offsets anyway)
__global__ __launch_bounds__(256) void k_parts_prepare(uint32_t* __restrict__ part_cnt, uint32_t n_parts, uint32_t* __restrict__ flags,
                                                      const uint64_t* __restrict__ host_skoff, uint64_t* __restrict__ dev_skoff,
                                                      uint32_t n_skoff, const uint32_t* __restrict__ host_sub, uint32_t* __restrict__ dev_sub,
                                                      uint32_t n_sub, uint32_t* __restrict__ zero_inter, uint32_t zero_n,
                                                      uint4* __restrict__ filter, uint32_t filter_vec) {
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = t; i < n_parts; i += stride) part_cnt[i] = 0;
    for (uint32_t i = t; i < filter_vec; i += stride) filter[i] = make_uint4(0, 0, 0, 0);
    // small-problem form: the parts ADD into the pair matrix, so the cells they add into -- (i, j > i), the cells every
    // form of the comparison owns -- start from zero; the diagonal and the lower triangle stay the caller's (spsp.h)
    for (uint32_t i = t; i < zero_n * zero_n; i += stride)
        if (i % zero_n > i / zero_n) zero_inter[i] = 0;
    if (t < 16) flags[t] = 0;
    for (uint32_t i = t; i < n_skoff; i += stride) dev_skoff[i] = host_skoff[i];
    for (uint32_t i = t; i < n_sub; i += stride) dev_sub[i] = host_sub[i];
}

// Row-partitioned calls (a rank of a multi-GPU comparison owns some rows): only keys held by an OWNED
// sketch can add to an owned row, so the dictionary is built from the owned sketches' keys and the keys of the other
// sketches that one of them also holds.  A blocked Bloom filter over the owned keys (16 bits per key, two bits in one
// 32-bit word: ~1.4 % false positives, no false negatives) stands in front of the scatter: a foreign key that misses
// it is read once (12 or 20 bytes, coalesced) and dropped -- no record, no `where` word, no place in a part.  Foreign
// keys that pass by accident only occupy a record: full keys are compared when the part is grouped.
__device__ __forceinline__ uint32_t filter_bits(uint64_t h) { return (1u << ((h >> 32) & 31u)) | (1u << ((h >> 37) & 31u)); }
template <bool HAS_HI>
__global__ __launch_bounds__(256) void k_parts_filter(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n, uint32_t row_first,
                                                     uint32_t row_stride, uint32_t row_limit, uint32_t* __restrict__ filter,
                                                     uint32_t fmask) {
    const uint32_t j = row_first + blockIdx.y * row_stride;        // grid.y = owned sketch, grid.x = 256-key chunk of it
    if (j >= n || j >= row_limit) return;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    const uint64_t h = key_hash(K.lo[e], K.mn[e], HAS_HI ? K.hi[e] : 0ull, HAS_HI);
    atomicOr(&filter[(uint32_t)h & fmask], filter_bits(h));
}

// A workgroup owns E x kScatThreads consecutive entries of the concatenated key arrays, E per thread, all held in
// registers: one round of (independent) loads, LDS counts per part with the entry's rank, ONE global atomic per
// part the chunk touches (lane p takes part p: a wave's atomics are one contiguous 256-byte request), stores.
// (E = 4: with 8 or 16 the unrolled hashes spill registers -- measured 20x slower.)
template <bool HAS_HI, int E>
__global__ __launch_bounds__(kScatThreads, 4) void k_parts_scatter(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n,
                                                               const uint32_t* __restrict__ sub_sk, uint64_t S, uint32_t n_parts,
                                                               uint32_t cap, uint32_t* __restrict__ part_cnt, uint64_t* __restrict__ recs,
                                                               uint32_t* __restrict__ where, uint32_t* __restrict__ flags, bool check_order,
                                                               const uint32_t* __restrict__ filter, uint32_t fmask, uint64_t e_first,
                                                               uint32_t row_first, uint32_t row_stride, uint32_t row_limit,
                                                               uint32_t classes, uint32_t cls) {
    constexpr uint32_t W = HAS_HI ? 3 : 2;
    extern __shared__ uint32_t hist[];                   // [n_parts]
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint64_t base = e_first + (uint64_t)blockIdx.x * E * kScatThreads;   // (the grid starts at the first owned sketch's chunk)
    for (uint32_t p = t; p < n_parts; p += kScatThreads) hist[p] = 0;
    uint64_t lo[E], hi[E];
    uint32_t mn[E], pr[E];                                // pr = part | rank << 15
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const uint64_t e = base + (uint64_t)u * kScatThreads + t;
        lo[u] = 0; mn[u] = 0; hi[u] = 0;
        if (e < S) { lo[u] = K.lo[e]; mn[u] = K.mn[e]; if (HAS_HI) hi[u] = K.hi[e]; }
    }
    // sketch of every entry and the sortedness check, while the keys are the only thing in flight
    uint32_t sk_of[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const uint64_t e = base + (uint64_t)u * kScatThreads + t;
        // the key in front of e: the neighbouring lane's, except for the first lane of a wave
        uint64_t plo = __shfl_up(lo[u], 1), phi = HAS_HI ? __shfl_up(hi[u], 1) : 0ull;
        uint32_t pmn = __shfl_up(mn[u], 1);
        sk_of[u] = 0;
        if (e >= S) continue;
        if (lane == 0 && e > 0) { plo = K.lo[e - 1]; pmn = K.mn[e - 1]; if (HAS_HI) phi = K.hi[e - 1]; }
        uint32_t j = sub_sk[e / kScatSub];
        while (j + 1 < n && sk_off[j + 1] <= e) ++j;      // sketch of entry e (empty sketches are stepped over)
        sk_of[u] = j;
        if (check_order && e > sk_off[j]) {               // strictly increasing inside a sketch (spsp_compare_keys_unordered: the caller vouches
            const bool less = pmn != mn[u] ? pmn < mn[u] : (HAS_HI && phi != hi[u]) ? phi < hi[u] : plo < lo[u];   // for distinct keys instead)
            if (!less) atomicOr(&flags[0], 1u);
        }
    }
    // an owned row i counts holders j > i only: nothing of the sketches in front of the first owned row is dealt.
    // Filtered form: a key of a sketch this call does not own is kept only if an owned sketch may hold it too
    uint64_t hsh[E];
    bool keep[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const uint64_t e = base + (uint64_t)u * kScatThreads + t;
        hsh[u] = key_hash(lo[u], mn[u], hi[u], HAS_HI);
        keep[u] = e < S && sk_of[u] >= row_first;        // (sketches in front of the first owned row are never counted by an owned row)
        if (classes > 1 && keep[u] && key_class(hsh[u], classes) != cls) {   // very large inputs: one class of keys per pass (compare_job_begin)
            keep[u] = false;
            where[e] = kNoWhere;                         // (the row sums of this pass read every entry of their row)
        }
        if (filter && keep[u] && !owned_row(sk_of[u], row_first, row_stride, row_limit)) {
            const uint32_t bits = filter_bits(hsh[u]);
            keep[u] = (filter[(uint32_t)hsh[u] & fmask] & bits) == bits;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u) {
        pr[u] = 0xffffffffu;
        if (keep[u]) {
            const uint32_t part = (uint32_t)(((hsh[u] >> 32) * n_parts) >> 32);
            pr[u] = part | (atomicAdd(&hist[part], 1u) << 15);
        }
    }
    __syncthreads();
    for (uint32_t p = t; p < n_parts; p += kScatThreads) {
        const uint32_t c = hist[p];
        if (c) hist[p] = atomicAdd(&part_cnt[p], c);     // the chunk's records of part p start here
    }                                                    // (eight reservations in flight per thread instead of one: no change at 17 000 parts, measured)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const uint64_t e = base + (uint64_t)u * kScatThreads + t;
        if (!keep[u]) continue;                          // (beyond the end, or dropped by the filter: no row sum reads its `where`)
        const uint32_t j = sk_of[u];
        const uint32_t part = pr[u] & 0x7fffu, at = hist[part] + (pr[u] >> 15);
        // where the record goes, in entry order (coalesced): the row sums find the key's list through it, so the
        // record need not carry its entry number and k_parts_group need not scatter one word per key back
        if (where) where[e] = at < cap ? part * cap + at : kNoWhere;     // (the small-problem form keeps no index: k_parts_group_small)
        if (at >= cap) continue;                         // overflow: the grouping kernel sees the count and raises the flag
        uint64_t* r = recs + ((uint64_t)part * cap + at) * W;
        if (HAS_HI) { r[0] = lo[u]; r[1] = (uint64_t)mn[u] | ((uint64_t)j << 32); r[2] = hi[u]; }
        else *reinterpret_cast<ulonglong2*>(r) = make_ulonglong2(lo[u], (uint64_t)mn[u] | ((uint64_t)j << 32));   // one 16-byte store
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tiles: the scatter over (block of sketches) x (range of every sketch) instead of over 4096 consecutive entries (round 5).
// With consecutive entries a workgroup deals ~one sketch, so the holders of one key -- the members of a family -- are dealt
// by as many workgroups, at different times: every record is a 16-byte store of its own into the tail of its part and every
// record's list reference sits in a line of its own (one miss per key in the row sums).  A TILE is range s of T -- the
// entries [s len / T, (s + 1) len / T) -- of each of tile_sk consecutive sketches, ~4 000 entries in all.  Sketches of a family
// hold mostly the same keys in the same (sorted) order, so a key sits at nearly the same relative place in each of them: its
// holders inside the block are dealt in the same round by ONE workgroup, take consecutive ranks in the key's part (nothing
// else of the round goes to that part, as a rule: 4096 records over thousands of parts) and leave as one run; their list
// references share lines that the family's rows -- side by side behind one L2 (xcd_rows) -- fetch once.  Ranges by PLACE,
// not by key: splitter keys (every sketch's lower bound of the pivot sketch's quantiles, found by binary or galloping
// search) were built first and cost 0.11 ms at configs[3] -- 14 random lines per search whatever the search -- for the same
// row-sum time: a holder that lands in the neighbouring range only splits its key's run in two.  Scheduling only: same
// records, same parts, same `where` semantics, every entry in exactly one tile whatever the keys are.
constexpr uint32_t kTileSkMax = 64;                    // sketches per block: a power of two up to this (stage_sk_off)
// tile_info[tile] = block | s << 32 | T << 48: the tile is range s of the T ranges of block `block` (made by the host, which
// has the offsets: stage_sk_off)
template <bool HAS_HI>
__global__ __launch_bounds__(kScatThreads, 4) void k_parts_scatter_tiles(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n,
                                                                     const uint64_t* __restrict__ tile_info, uint32_t tile_sk,
                                                                     uint32_t n_parts, uint32_t cap,
                                                                     uint32_t* __restrict__ part_cnt, uint64_t* __restrict__ recs,
                                                                     uint32_t* __restrict__ where, uint32_t* __restrict__ flags, bool check_order,
                                                                     uint32_t row_first, uint32_t classes, uint32_t cls) {
    constexpr int E = 4;
    constexpr uint32_t W = HAS_HI ? 3 : 2;
    extern __shared__ uint32_t hist[];                   // [n_parts]
    __shared__ uint32_t s_a[kTileSkMax], s_pre[kTileSkMax + 1];
    const uint32_t t = threadIdx.x, lane = t & 63u, tile = blockIdx.x;
    const uint64_t info = tile_info[tile];
    const uint32_t b = (uint32_t)info;
    if (t < 64) {
        const uint32_t j = b * tile_sk + t;
        uint32_t a = 0, z = 0;
        if (t < tile_sk && j < n) {                       // range s of T: the same share of every sketch
            const uint64_t j0 = sk_off[j], len = sk_off[j + 1] - j0, s_ = (info >> 32) & 0xffffu, T_ = info >> 48;
            a = (uint32_t)(j0 + s_ * len / T_); z = (uint32_t)(j0 + (s_ + 1) * len / T_);
        }
        uint32_t incl = z - a;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += y; }
        s_a[t] = a; s_pre[t] = incl - (z - a);
        if (t == 63) s_pre[64] = incl;
    }
    __syncthreads();
    const uint32_t total = s_pre[64];
    for (uint32_t x0 = 0; x0 < total; x0 += (uint32_t)E * kScatThreads) {     // (one round, as a rule: tiles are made for 0.9 of a round)
        for (uint32_t p = t; p < n_parts; p += kScatThreads) hist[p] = 0;
        uint64_t lo[E], hi[E], ent[E];
        uint32_t mn[E], sk_of[E], pr[E];
        bool first_of_run[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const uint32_t x = x0 + (uint32_t)u * kScatThreads + t;
            lo[u] = 0; mn[u] = 0; hi[u] = 0; ent[u] = ~0ull; sk_of[u] = 0; first_of_run[u] = false;
            if (x < total) {
                uint32_t q = 0;                           // the sketch of the block whose run holds place x: largest q with s_pre[q] <= x
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) if (s_pre[q + step] <= x) q += step;
                const uint64_t e = (uint64_t)s_a[q] + (x - s_pre[q]);
                ent[u] = e; sk_of[u] = b * tile_sk + q; first_of_run[u] = x == s_pre[q];
                lo[u] = K.lo[e]; mn[u] = K.mn[e]; if (HAS_HI) hi[u] = K.hi[e];
            }
        }
        uint64_t hsh[E];
        bool keep[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            // the key in front of e: the neighbouring lane's, except for the first lane of a wave and the first entry of a run
            uint64_t plo = __shfl_up(lo[u], 1), phi = HAS_HI ? __shfl_up(hi[u], 1) : 0ull;
            uint32_t pmn = __shfl_up(mn[u], 1);
            const uint64_t e = ent[u];
            const bool valid = e != ~0ull;
            if (check_order && valid && e > sk_off[sk_of[u]]) {   // strictly increasing inside a sketch (spsp_compare_keys_unordered: the caller vouches for distinct keys instead)
                if (lane == 0 || first_of_run[u]) { plo = K.lo[e - 1]; pmn = K.mn[e - 1]; if (HAS_HI) phi = K.hi[e - 1]; }
                const bool less = pmn != mn[u] ? pmn < mn[u] : (HAS_HI && phi != hi[u]) ? phi < hi[u] : plo < lo[u];
                if (!less) atomicOr(&flags[0], 1u);
            }
            hsh[u] = key_hash(lo[u], mn[u], hi[u], HAS_HI);
            keep[u] = valid && sk_of[u] >= row_first;     // (sketches in front of the first owned row are never counted by an owned row)
            if (classes > 1 && keep[u] && key_class(hsh[u], classes) != cls) { keep[u] = false; where[e] = kNoWhere; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E; ++u) {
            pr[u] = 0xffffffffu;
            if (keep[u]) {
                const uint32_t part = (uint32_t)(((hsh[u] >> 32) * n_parts) >> 32);
                pr[u] = part | (atomicAdd(&hist[part], 1u) << 15);
            }
        }
        __syncthreads();
        for (uint32_t p = t; p < n_parts; p += kScatThreads) {
            const uint32_t c = hist[p];
#if SPSP_EXP & 1
            if (c) hist[p] = (blockIdx.x * 29u) % (cap / 2);     // (timing experiment: no reservation -- results are wrong)
#else
            if (c) hist[p] = atomicAdd(&part_cnt[p], c);
#endif
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E; ++u) {
            if (!keep[u]) continue;
            const uint32_t part = pr[u] & 0x7fffu, at = hist[part] + (pr[u] >> 15);
#if !(SPSP_EXP & 4)
            where[ent[u]] = at < cap ? part * cap + at : kNoWhere;
#endif
            if (at >= cap) continue;
#if SPSP_EXP & 2
            continue;
#endif
            uint64_t* r = recs + ((uint64_t)part * cap + at) * W;
            if (HAS_HI) { r[0] = lo[u]; r[1] = (uint64_t)mn[u] | ((uint64_t)sk_of[u] << 32); r[2] = hi[u]; }
            else *reinterpret_cast<ulonglong2*>(r) = make_ulonglong2(lo[u], (uint64_t)mn[u] | ((uint64_t)sk_of[u] << 32));
        }
        __syncthreads();                                  // (the next round clears the counters these stores read)
    }
}

// One workgroup per part.  A slot word is (records of the slot's key so far) << 13 | (claiming record + 1): a
// record CASes its index into the first free slot of its probe sequence or finds the slot of its key (full-key
// compare against the claiming record, kept in LDS), then draws its rank in the key's list by adding 1 << 13.
// The claiming record of every key held by >= 2 sketches reserves the list in the part's slice of `ids`, and
// every record writes its sketch id and its list reference.  80 KiB of LDS: two workgroups per CU.
template <bool HAS_HI>
__global__ __launch_bounds__(kGroupThreads) void k_parts_group(const uint64_t* __restrict__ recs, const uint32_t* __restrict__ part_cnt,
                                                              uint16_t* __restrict__ ids, uint32_t* __restrict__ list_of_slot,
                                                              uint32_t* __restrict__ flags, uint32_t t_bits, uint32_t max_cols,
                                                              unsigned long long* __restrict__ bits, uint32_t n_sk,
                                                              unsigned long long* __restrict__ multi) {
    constexpr uint32_t W = HAS_HI ? 3 : 2;
    constexpr uint32_t R = kPartCap / kGroupThreads;      // records per thread
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_g[];
    uint64_t* k_lo = reinterpret_cast<uint64_t*>(lds_g);                                  // [kPartCap]
    uint64_t* k_hi = k_lo + kPartCap;                                                     // [kPartCap] (HAS_HI only)
    uint32_t* k_mn = reinterpret_cast<uint32_t*>(k_hi + (HAS_HI ? kPartCap : 0));         // [kPartCap]
    uint32_t* slot = k_mn + kPartCap;                                                     // [kPartSlots]
    uint32_t* cursor = slot + kPartSlots;
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t n = part_cnt[p];
    if (t == 0) atomicAdd(&flags[7], n);                  // records in all parts: sizes the parts of a filtered call's next attempt
    if (n > (uint32_t)kPartCap) { if (t == 0) { atomicOr(&flags[6], 1u); atomicAdd(&flags[2], n); } return; }   // ([2]: records of all overflowed parts -- sizes the spill)
    for (uint32_t x = t; x < (uint32_t)kPartSlots; x += kGroupThreads) slot[x] = 0;
    if (t == 0) { cursor[0] = 0; cursor[1] = 0; }
    const uint64_t* base = recs + (uint64_t)p * kPartCap * W;
    uint64_t lo[R], hi[R];
    uint32_t mn[R], sk[R], hs[R], rank[R];
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        hs[u] = 0; lo[u] = 0; hi[u] = 0; mn[u] = 0; sk[u] = 0; rank[u] = 0;
        {   // loaded whether or not the record exists (the slice is allocated in full): no wait for the count first
            uint64_t w1;
            if (HAS_HI) { lo[u] = base[(uint64_t)r * W]; w1 = base[(uint64_t)r * W + 1]; hi[u] = base[(uint64_t)r * W + 2]; }
            else { const ulonglong2 v = reinterpret_cast<const ulonglong2*>(base)[r]; lo[u] = v.x; w1 = v.y; hi[u] = 0ull; }
            mn[u] = (uint32_t)w1; sk[u] = (uint32_t)(w1 >> 32);
            k_lo[r] = lo[u]; k_mn[r] = mn[u];
            if (HAS_HI) k_hi[r] = hi[u];
            hs[u] = (uint32_t)(((key_hash(lo[u], mn[u], hi[u], HAS_HI) & 0xffffffffull) * kPartSlots) >> 32);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        if (r >= n) continue;
        uint32_t h = hs[u];
        for (;;) {                                        // ends: the table has about twice as many slots as a part has records
            uint32_t cur = slot[h];
            if (cur == 0) cur = atomicCAS(&slot[h], 0u, r + 1);
            if (cur == 0) break;                          // claimed
            const uint32_t c = (cur & 0x1fffu) - 1;
            if (k_lo[c] == lo[u] && k_mn[c] == mn[u] && (!HAS_HI || k_hi[c] == hi[u])) break;
            h = h + 1 == (uint32_t)kPartSlots ? 0u : h + 1;
        }
        hs[u] = h;
        rank[u] = atomicAdd(&slot[h], 1u << 13) >> 13;
    }
    __syncthreads();
    uint32_t cnt[R];
    bool claimer[R];
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        const uint32_t w = r < n ? slot[hs[u]] : 0u;
        cnt[u] = w >> 13;
        claimer[u] = r < n && (w & 0x1fffu) == r + 1;
    }
    // multi: one bit per record slot -- does the record's key have a list?  The row sums look here (8.7 MB at configs[3]: it
    // stays in the L2s) before they fetch a list reference (70 MB, a line from HBM each): a key held by one sketch costs them
    // no miss.  A wave's 64 consecutive slots are one 8-byte store.
    // (flags[12] / flags[13] = records that have a list / records, of every 64th part: the row sums use the bits only when fewer than 2 in 5 do -- with more, the extra
    // dependent read costs more than the misses it saves: 0.84 -> 0.99 ms at configs[3], 0.84 -> 0.32 for unrelated sketches)
    if (multi) {                                          // (not made when the context's last comparison had no use for it: job_parts)
        uint32_t listed = 0;
#pragma unroll
        for (uint32_t u = 0; u < R; ++u) {
            const uint32_t r = u * kGroupThreads + t;
            const unsigned long long m = __ballot(r < n && cnt[u] >= 2 && cnt[u] < t_bits);
            if ((t & 63u) == 0) { multi[((uint64_t)p * kPartCap + r) >> 6] = m; listed += (uint32_t)__popcll(m); }
        }
        if ((p & 63u) == 0) {                             // (a sample: one part in 64 -- an atomic per workgroup on one word was 0.15 ms)
            if ((t & 63u) == 0 && listed) atomicAdd(cursor + 1, listed);      // (the word behind the list cursor: summed per workgroup)
            __syncthreads();
            if (t == 0) { atomicAdd(&flags[12], cursor[1]); atomicAdd(&flags[13], n); }
        }
    }
    __syncthreads();                                      // (every count has been read: the slots are reused for the list places)
    const uint32_t ids_base = p * (4u * kPartCap);
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        if (!claimer[u] || cnt[u] < 2) continue;          // one thread per key held by >= 2 sketches
        if (cnt[u] >= t_bits) {                           // spill attempts only (else t_bits = 0xffffffff): a column instead of a list, see k_spill_pairs
            const uint32_t col = atomicAdd(&flags[9], 1u);
            if (col >= max_cols) atomicOr(&flags[5], 1u);
            slot[hs[u]] = kColFlag | (col < max_cols ? col : 0u);
            continue;
        }
        const uint32_t o = ids_base + atomicAdd(cursor, list_u16(cnt[u]));   // 8 u16 per 2..7 records: fits 4 x kPartCap
        ids[o] = (uint16_t)cnt[u];                        // a key is held at most once per sketch: <= 65535
        slot[hs[u]] = o;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        if (r >= n) continue;
        uint32_t* out = list_of_slot + (size_t)p * kPartCap + r;         // record order: coalesced
        if (cnt[u] < 2) { *out = kNoList; continue; }                    // held by one sketch: no pair to count
        const uint32_t o = slot[hs[u]];
        if (o & kColFlag) {
            const uint32_t col = o & ~kColFlag;
            atomicOr(&bits[(uint64_t)(col >> 6) * n_sk + sk[u]], 1ull << (col & 63u));
            *out = kNoList;
            continue;
        }
        ids[o + 1 + rank[u]] = (uint16_t)sk[u];
        *out = list_ref(o, cnt[u]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Spill: parts that overflow.  A part holds kPartCap records and is planned for 2 900; keys arrive with ALL their holders,
// so a collection whose genomes come in families of a hundred or more (one species sequenced many times) makes parts
// that do not fit -- whatever their number, since a key held by 5 000 sketches is 5 000 records of one part.  The records
// of exactly those parts (every record of them: a key's holders must stay together) are grouped in a table in HBM instead:
//   k_spill_insert   entry of an overflowed part -> slot of its key (claimed by entry number, full keys compared through
//                    the claimer's entry), its rank among the key's holders
//   k_spill_ranges   slot -> place of the key's holder list behind the parts' lists (workgroup prefix, one atomic per
//                    workgroup), or -- a key held by t_bits sketches or more -- a COLUMN of the bit matrix below
//   k_spill_fill     holder ids into the lists / bits into the columns; `where` of the entry -> its key's list reference
// and the row sums run as for any other key.  Keys held by a large share of the sketches leave the lists altogether: one
// such key costs a row-sum workgroup h LDS adds per holder (h^2 / 2 in all), while as one bit per sketch it costs every PAIR
// of sketches one AND + popcount per 64 keys (k_spill_pairs: wavefront-wide popcounts, no MFMA -- BASELINE north_star).
constexpr int kSpillThreads = 256;
__device__ __forceinline__ uint32_t spill_home(uint64_t h, uint32_t log2cap) { return (uint32_t)(mix64(h ^ 0x8EBC6AF09C88C6E3ULL) >> (64 - log2cap)); }

// sketch of entry e: from the sketch of its 4096-entry sub-chunk (the table the scatter uses), a few steps on (a binary search
// over the offsets -- 14 dependent reads per entry -- was most of these kernels' time)
__device__ __forceinline__ uint32_t spill_sketch(const uint64_t* __restrict__ sk_off, uint32_t n, const uint32_t* __restrict__ sub_sk, uint64_t e) {
    uint32_t j = sub_sk[e / kScatSub];
    while (j + 1 < n && sk_off[j + 1] <= e) ++j;
    return j;
}

template <bool HAS_HI>
__global__ __launch_bounds__(kSpillThreads) void k_spill_insert(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n, uint64_t S, uint64_t e_first,
                                                               uint32_t row_first, uint32_t n_parts, const uint32_t* __restrict__ part_cnt,
                                                               uint32_t* __restrict__ tbl, uint32_t log2cap, uint32_t* __restrict__ cnt,
                                                               uint32_t* __restrict__ where, uint32_t* __restrict__ rank_of, uint32_t room,
                                                               uint32_t* __restrict__ flags, uint32_t classes, uint32_t cls,
                                                               const uint32_t* __restrict__ sub_sk) {
    if (flags[2] > room) return;                          // more records than this attempt has room for: the host repeats it with the count
    const uint64_t e = e_first + (uint64_t)blockIdx.x * kSpillThreads + threadIdx.x;
    if (e >= S) return;
    const uint64_t lo = K.lo[e], hi = HAS_HI ? K.hi[e] : 0ull;
    const uint32_t mn = K.mn[e];
    const uint64_t h = key_hash(lo, mn, hi, HAS_HI);
    if (classes > 1 && key_class(h, classes) != cls) return;
    if (part_cnt[(uint32_t)(((h >> 32) * n_parts) >> 32)] <= (uint32_t)kPartCap) return;
    if (row_first && spill_sketch(sk_off, n, sub_sk, e) < row_first) return;   // (the scatter deals nothing of the sketches in front of the first owned row)
    const uint32_t mask = (1u << log2cap) - 1u;
    uint32_t pos = spill_home(h, log2cap);
    for (uint32_t probes = 0;; ++probes) {
        uint32_t cur = tbl[pos];
        if (cur == 0) cur = atomicCAS(&tbl[pos], 0u, (uint32_t)e + 1u);
        if (cur == 0) break;                              // claimed
        const uint64_t c = cur - 1u;
        if (K.lo[c] == lo && K.mn[c] == mn && (!HAS_HI || K.hi[c] == hi)) break;
        if (probes > mask) { atomicOr(&flags[5], 1u); where[e] = kNoWhere; return; }
        pos = (pos + 1u) & mask;
    }
    atomicAdd(&cnt[pos], 1u);                             // (result unused: the keys that nearly every sketch holds put thousands of these on one word, and
                                                          // a returning atomic waits for its turn; a list key draws its place in k_spill_fill instead)
    where[e] = pos;                                       // (until k_spill_fill has run)
}

// flags[8] = u16 of lists handed out, flags[9] = columns handed out
__global__ __launch_bounds__(kRowThreads) void k_spill_ranges(const uint32_t* __restrict__ cnt, uint64_t cap, uint32_t* __restrict__ off,
                                                             uint16_t* __restrict__ ids, uint32_t ids_base, uint32_t ids_room,
                                                             uint32_t* __restrict__ lref, uint32_t t_bits, uint32_t max_cols, uint32_t room,
                                                             uint32_t* __restrict__ flags) {
    if (flags[2] > room) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[6] = 0;           // the overflowed parts are taken care of: the row sums may run
    __shared__ uint32_t wave_sum[kRowThreads / 64], wave_cols[kRowThreads / 64];
    __shared__ uint32_t s_base, s_cols;
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint64_t base_slot = ((uint64_t)blockIdx.x * kRowThreads + t) * kRowSlots;   // 16 consecutive slots per lane
    uint32_t c[kRowSlots];
    uint32_t sum = 0, cols = 0;
#pragma unroll
    for (int u = 0; u < kRowSlots; ++u) {
        const uint64_t sl = base_slot + u;
        c[u] = sl < cap ? cnt[sl] : 0u;
        if (c[u] >= t_bits) ++cols;
        else if (c[u] >= 2) sum += list_u16(c[u]);       // (a key of one holder takes part in no pair: no list)
    }
    uint32_t x = sum, y = cols;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x2 = __shfl_up(x, d), y2 = __shfl_up(y, d);
        if (lane >= (uint32_t)d) { x += x2; y += y2; }
    }
    if (lane == 63) { wave_sum[wid] = x; wave_cols[wid] = y; }
    __syncthreads();
    uint32_t pre = 0, all = 0, pre_c = 0, all_c = 0;
    for (uint32_t w = 0; w < kRowThreads / 64; ++w) {
        if (w < wid) { pre += wave_sum[w]; pre_c += wave_cols[w]; }
        all += wave_sum[w]; all_c += wave_cols[w];
    }
    if (t == 0) { s_base = all ? atomicAdd(&flags[8], all) : 0u; s_cols = all_c ? atomicAdd(&flags[9], all_c) : 0u; }
    __syncthreads();
    uint32_t at = s_base + pre + x - sum, col = s_cols + pre_c + y - cols;
    if ((all && s_base + all > ids_room) || (all_c && s_cols + all_c > max_cols)) { if (t == 0) atomicOr(&flags[5], 1u); return; }   // (cannot happen: sized from the count)
#pragma unroll
    for (int u = 0; u < kRowSlots; ++u) {
        const uint64_t sl = base_slot + u;
        if (c[u] < 2) { if (c[u] == 1) off[sl] = kNoWhere; continue; }          // (held by one sketch: k_spill_fill reads this, the counts count down under it)
        if (c[u] >= t_bits) { off[sl] = kColFlag | col; ++col; continue; }
        off[sl] = at;
        ids[ids_base + at] = (uint16_t)c[u];
        lref[sl] = list_ref(ids_base + at, c[u]);
        at += list_u16(c[u]);
    }
}

template <bool HAS_HI>
__global__ __launch_bounds__(kSpillThreads) void k_spill_fill(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n, uint64_t S, uint64_t e_first,
                                                             uint32_t row_first, uint32_t n_parts, const uint32_t* __restrict__ part_cnt,
                                                             uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                                                             uint16_t* __restrict__ ids, uint32_t ids_base, uint32_t lref_base,
                                                             unsigned long long* __restrict__ bits, uint32_t* __restrict__ where,
                                                             const uint32_t* __restrict__ rank_of, uint32_t room, const uint32_t* __restrict__ flags,
                                                             uint32_t classes, uint32_t cls, const uint32_t* __restrict__ sub_sk) {
    if (flags[2] > room || flags[5]) return;
    const uint64_t e = e_first + (uint64_t)blockIdx.x * kSpillThreads + threadIdx.x;
    if (e >= S) return;
    const uint64_t h = key_hash(K.lo[e], K.mn[e], HAS_HI ? K.hi[e] : 0ull, HAS_HI);
    if (classes > 1 && key_class(h, classes) != cls) return;
    if (part_cnt[(uint32_t)(((h >> 32) * n_parts) >> 32)] <= (uint32_t)kPartCap) return;
    const uint32_t j = spill_sketch(sk_off, n, sub_sk, e);
    if (j < row_first) return;
    const uint32_t pos = where[e];
    if (pos == kNoWhere) return;
    const uint32_t o = off[pos];
    if (o == kNoWhere) { where[e] = kNoWhere; return; }   // held by this sketch only
    if (o & kColFlag) {                                   // one bit per holder: word (column / 64) of sketch j
        const uint32_t col = o & ~kColFlag;
        atomicOr(&bits[(uint64_t)(col >> 6) * n + j], 1ull << (col & 63u));
        where[e] = kNoWhere;                              // (no list: the row sums pass it by)
        return;
    }
    ids[ids_base + o + atomicSub(&cnt[pos], 1u)] = (uint16_t)j;     // the count runs down c .. 1: the places behind the length word
    where[e] = lref_base + pos;
}

// Pair counts of the keys that have a column: cell (i, j > i) += popcount(bits[.][i] & bits[.][j]) over the words in use.
// A workgroup takes a 64 x 64 tile of pairs, a thread 4 x 4 of them (rows p * 16 + t / 16, columns q * 16 + t % 16: the
// sixteen lanes of a row read consecutive LDS words and store consecutive cells); the two 64-sketch panels pass through
// LDS kPairWords words at a time (a word of 64 consecutive sketches is one 512-byte line run).  v_and + v_bcnt
// (popcount-accumulate): four VALU instructions per pair and 64 keys -- the kernel runs at the VALU issue rate (4.6 x 10^13
// lane operations/s measured), so larger tiles buy nothing (128 x 128, 8 x 8 per thread: 15 % slower, fewer waves per SIMD).
// Tile pairs are dealt to the XCDs in runs, so that the tiles of one row of tiles (same first panel) sit behind one L2.
constexpr int kPairTile = 64, kPairWords = 16, kPairThreads = 256, kPairR = kPairTile / 16;
__global__ __launch_bounds__(kPairThreads) void k_spill_pairs(const unsigned long long* __restrict__ bits, uint32_t n, uint32_t tiles,
                                                             uint32_t row_first, uint32_t row_stride, uint32_t row_limit,
                                                             uint32_t* __restrict__ inter, const uint32_t* __restrict__ flags) {
    const uint32_t words = (flags[9] + 63u) >> 6;
    if (words == 0 || flags[6] || flags[5]) return;
    const uint32_t total = tiles * (tiles + 1) / 2, per_xcd = gridDim.x >> 3;          // (the grid is a multiple of 8)
    uint32_t left = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (left >= total) return;
    // tile pair (bi <= bj) from its flat number: row bi of the upper triangle holds tiles - bi pairs
    uint32_t bi = 0;
    while (left >= tiles - bi) { left -= tiles - bi; ++bi; }
    const uint32_t bj = bi + left;
    __shared__ unsigned long long a[kPairWords][kPairTile], b[kPairWords][kPairTile];
    const uint32_t t = threadIdx.x, ti = t >> 4, tj = t & 15u;
    uint32_t acc[kPairR][kPairR] = {};
    for (uint32_t w0 = 0; w0 < words; w0 += kPairWords) {
        for (uint32_t x = t; x < kPairWords * kPairTile; x += kPairThreads) {
            const uint32_t w = w0 + x / kPairTile, c = x % kPairTile;
            const uint32_t si = bi * kPairTile + c, sj = bj * kPairTile + c;
            a[x / kPairTile][c] = (w < words && si < n) ? bits[(uint64_t)w * n + si] : 0ull;
            b[x / kPairTile][c] = (w < words && sj < n) ? bits[(uint64_t)w * n + sj] : 0ull;
        }
        __syncthreads();
#pragma unroll 2
        for (uint32_t w = 0; w < kPairWords; ++w) {
            unsigned long long va[kPairR], vb[kPairR];
#pragma unroll
            for (int q = 0; q < kPairR; ++q) { va[q] = a[w][q * 16 + ti]; vb[q] = b[w][q * 16 + tj]; }
#pragma unroll
            for (int p = 0; p < kPairR; ++p)
#pragma unroll
                for (int q = 0; q < kPairR; ++q) acc[p][q] += (uint32_t)__popcll(va[p] & vb[q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < kPairR; ++p) {
        const uint32_t i = bi * kPairTile + p * 16 + ti;
        if (i >= n || !owned_row(i, row_first, row_stride, row_limit)) continue;
#pragma unroll
        for (int q = 0; q < kPairR; ++q) {
            const uint32_t j = bj * kPairTile + q * 16 + tj;
            if (j < n && j > i && acc[p][q]) inter[(uint64_t)i * n + j] += acc[p][q];   // (the row sums have stored the cell: same stream, earlier kernel)
        }
    }
}

// Small problems (at most kSmallN sketches, k <= 32, every row owned): grouping AND counting in one kernel.  Parts of
// half the size (2048 records: 40 KiB of keys and slots), the holders of every key are listed in LDS, every record adds
// 1 to the cell (its sketch, other holder) for the holders above it -- an N x N matrix of 16-bit counters in LDS,
// two per word -- and the part's non-zero cells go to the pair matrix with global atomics (a few hundred per part when
// the sketches fall into families).  No list references, no `where` index, no row-sum kernel: the chain is three
// launches instead of four and moves a third of the bytes; it is what lets the comparison of bench.py's 100
// sketches run beside a dense pass on 32 CUs instead of 64.
constexpr int kSmallCap = 2048, kSmallSlots = 3968, kSmallN = 128, kSmallMean = 1450;
constexpr int kSmallHl = kSmallCap + 3 * (kSmallCap / 2);   // lists of >= 2 holders padded to 4: at most cap / 2 such keys
__global__ __launch_bounds__(kGroupThreads) void k_parts_group_small(const uint64_t* __restrict__ recs, const uint32_t* __restrict__ part_cnt,
                                                                    uint32_t n_sk, uint32_t* __restrict__ inter, uint32_t* __restrict__ flags) {
    constexpr uint32_t R = kSmallCap / kGroupThreads;     // records per thread
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_s[];
    uint64_t* k_lo = reinterpret_cast<uint64_t*>(lds_s);                    // [kSmallCap]
    uint32_t* k_mn = reinterpret_cast<uint32_t*>(k_lo + kSmallCap);         // [kSmallCap]
    uint32_t* slot = k_mn + kSmallCap;                                      // [kSmallSlots]
    uint32_t* mat = slot + kSmallSlots;                                     // [kSmallN * kSmallN / 2]: cell c in half c & 1 of word c >> 1
    uint8_t* hl = reinterpret_cast<uint8_t*>(mat + kSmallN * kSmallN / 2);  // [kSmallHl]: holder lists of the keys held more than once, one byte per
    uint32_t* cursor = reinterpret_cast<uint32_t*>(hl + kSmallHl);          // holder, every list on a 4-byte boundary (read a word at a time)
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t n = part_cnt[p];
    if (n > (uint32_t)kSmallCap) { if (t == 0) atomicOr(&flags[6], 1u); return; }
    for (uint32_t x = t; x < (uint32_t)kSmallSlots; x += kGroupThreads) slot[x] = 0;
    for (uint32_t x = t; x < (uint32_t)(kSmallN * kSmallN / 2); x += kGroupThreads) mat[x] = 0;
    if (t == 0) *cursor = 0;
    const ulonglong2* base = reinterpret_cast<const ulonglong2*>(recs) + (uint64_t)p * kSmallCap;
    uint64_t lo[R];
    uint32_t mn[R], sk[R], hs[R], rank[R];
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        const ulonglong2 v = base[r];                     // (loaded whether or not the record exists: the slice is allocated in full)
        lo[u] = v.x; mn[u] = (uint32_t)v.y; sk[u] = (uint32_t)(v.y >> 32); rank[u] = 0;
        k_lo[r] = lo[u]; k_mn[r] = mn[u];
        hs[u] = (uint32_t)(((key_hash(lo[u], mn[u], 0ull, false) & 0xffffffffull) * kSmallSlots) >> 32);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        if (r >= n) continue;
        uint32_t h = hs[u];
        for (;;) {                                        // ends: about twice as many slots as a part has records
            uint32_t cur = slot[h];
            if (cur == 0) cur = atomicCAS(&slot[h], 0u, r + 1);
            if (cur == 0) break;                          // claimed
            const uint32_t c = (cur & 0x1fffu) - 1;
            if (k_lo[c] == lo[u] && k_mn[c] == mn[u]) break;
            h = h + 1 == (uint32_t)kSmallSlots ? 0u : h + 1;
        }
        hs[u] = h;
        rank[u] = atomicAdd(&slot[h], 1u << 13) >> 13;
    }
    __syncthreads();
    uint32_t cnt[R];
    bool claimer[R];
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        const uint32_t w = r < n ? slot[hs[u]] : 0u;
        cnt[u] = w >> 13;
        claimer[u] = r < n && (w & 0x1fffu) == r + 1;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < R; ++u)
        if (claimer[u] && cnt[u] >= 2) slot[hs[u]] = atomicAdd(cursor, (cnt[u] + 3u) & ~3u);   // where the key's holders are listed in hl (4-aligned)
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        if (r < n && cnt[u] >= 2) hl[slot[hs[u]] + rank[u]] = (uint8_t)sk[u];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < R; ++u) {
        const uint32_t r = u * kGroupThreads + t;
        if (r >= n || cnt[u] < 2) continue;               // held by one sketch: no pair to count
        const uint32_t* hw = reinterpret_cast<const uint32_t*>(hl + slot[hs[u]]);
        for (uint32_t j = 0; j < cnt[u]; j += 4) {        // four holders per LDS word (the byte-wise walk was 17 of the kernel's ~30 LDS instructions per record)
            const uint32_t four = hw[j >> 2];
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t other = (four >> (8 * b)) & 0xffu;
                if (j + b < cnt[u] && other > sk[u]) {    // (a sketch holds a key at most once: the scatter checked the order)
                    const uint32_t cell = sk[u] * kSmallN + other;
                    atomicAdd(&mat[cell >> 1], (cell & 1u) ? 0x10000u : 1u);    // a part holds 2048 records: no half overflows
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t x = t; x < (uint32_t)(kSmallN * kSmallN / 2); x += kGroupThreads) {
        const uint32_t w = mat[x];
        if (!w) continue;
        const uint32_t c0 = 2 * x, a = c0 / kSmallN, b = c0 % kSmallN;      // cells c0 and c0 + 1 share row a (kSmallN is even)
        if (w & 0xffffu) atomicAdd(&inter[(uint64_t)a * n_sk + b], w & 0xffffu);
        if (w >> 16) atomicAdd(&inter[(uint64_t)a * n_sk + b + 1], w >> 16);
    }
}

// inter[i][*] for one owned sketch i and one block of 64 colour words (4096 columns).
// A lane owns ONE 64-bit word of the colour rows (64 columns): a key's row is read
// with one coalesced wave load and every lane adds its word's 64 bits into 64 private
// counters.  The adds are bit-sliced: (w >> b) & 0x0101..01 extracts bits b, b+8, ...
// as eight byte lanes, so eight 64-bit adds cover the whole word; the byte counters
// are spilled into 32-bit counters before they can wrap.  When a row is narrower than
// 64 words several keys share one wave load (lane group g handles key g).
// The four waves of the workgroup split the sketch's keys and meet in LDS; results
// are written with plain coalesced stores (every column > i, zeros included).
constexpr int kAccThreads = 256;
__global__ __launch_bounds__(kAccThreads) void k_accumulate(const uint32_t* __restrict__ row_of_entry,
                                                           const uint64_t* __restrict__ A, uint32_t W,
                                                           uint32_t lanes_per_key,
                                                           const uint64_t* __restrict__ sk_begin,
                                                           const uint64_t* __restrict__ sk_end, uint32_t n,
                                                           uint32_t row_first, uint32_t row_stride, uint32_t row_limit,
                                                           uint32_t* __restrict__ inter, const uint32_t* __restrict__ flags,
                                                           uint32_t* __restrict__ host_flags, bool add) {
    // every pass that can raise a flag has finished: hand them to the host (pinned memory) from here, so the
    // pipeline ends with this kernel and not with a device-to-host copy behind it
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kFlags) host_flags[threadIdx.x] = flags[threadIdx.x];
    const uint32_t i = row_first + blockIdx.y * row_stride;
    if (i >= n || i >= row_limit) return;
    const uint32_t wb = blockIdx.x;                         // block of 64 words
    const uint32_t first_wd = (i + 1) >> 6;                 // first word holding a column > i
    if (wb * 64 + 63 < first_wd) return;
    __shared__ uint32_t s_cnt[64 * 64];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (uint32_t x = t; x < 64 * 64; x += kAccThreads) s_cnt[x] = 0;
    const uint32_t groups = 64 / lanes_per_key;             // keys per wave load
    const uint32_t grp = lane / lanes_per_key, wl = lane % lanes_per_key;
    const uint32_t word = wb * 64 + wl;
    const bool active = word < W && word >= first_wd;
    const uint64_t e0 = sk_begin[i], e1 = sk_end[i];
    const uint64_t M8 = 0x0101010101010101ULL;
    uint32_t cnt[64];
#pragma unroll
    for (int b = 0; b < 64; ++b) cnt[b] = 0;
    uint64_t acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = 0;
    uint32_t pending = 0;                                    // keys folded into acc[] since the last spill
    auto spill = [&]() {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt[8 * j + b] += (uint32_t)(acc[b] >> (8 * j)) & 0xffu;
            acc[b] = 0;
        }
        pending = 0;
    };
    const uint64_t step = (uint64_t)(kAccThreads / 64) * groups;
    constexpr int U = 8;   // independent row gathers in flight per lane
    for (uint64_t e = e0 + (uint64_t)wave * groups + grp; e < e1 + (U - 1) * step; e += U * step) {
        uint32_t rows[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t eu = e + u * step;
            rows[u] = (active && eu < e1) ? row_of_entry[eu] : 0xffffffffu;
        }
        uint64_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = rows[u] != 0xffffffffu ? A[(uint64_t)rows[u] * W + word] : 0ull;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[b] += (v[u] >> b) & M8;
        }
        pending += U;
        if (pending >= 248) spill();
    }
    spill();
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 64; ++b)
        if (cnt[b]) atomicAdd(&s_cnt[wl * 64 + b], cnt[b]);
    __syncthreads();
    const uint32_t cols = lanes_per_key * 64;
    for (uint32_t x = t; x < cols; x += kAccThreads) {
        const uint32_t col = wb * 4096 + x;
        if (col > i && col < n) {   // every cell has one writer; later key-class passes add to the first pass's counts
            uint32_t* cell = &inter[(uint64_t)i * n + col];
            *cell = add ? *cell + s_cnt[x] : s_cnt[x];
        }
    }
}

// ---------------------------------------------------------------------------
// Host driver shared by the two input forms (flat key arrays / exchange slots):
// dictionary build -> colour matrix -> row sums, with the collision retry.
// Like the scan it is split at its host synchronisation: compare_job_begin queues the
// (speculative) pipeline and returns, compare_job_end waits, checks the flags and --
// only after a fingerprint collision or for inputs too large to speculate -- queues more.
// room = records the table and the lists are sized for; the key table has 2^log2cap slots; keys held by t_bits sketches or
// more get a column of the bit matrix (0xffffffff: none do); lists / list references of spilled keys start at ids_base / lref_base
struct SpillPlan { uint64_t room = 0; uint32_t log2cap = 0, t_bits = 0xffffffffu, max_cols = 0, ids_base = 0, ids_room = 0, lref_base = 0; };
struct ComparePlan {
    uint64_t S_own;        // upper bound on the keys inserted (sizes the table and the speculative matrix)
    uint64_t S_entries;    // entry index space (sizes row_of_entry)
    uint32_t n, n_own, row_first, row_stride, row_limit;
    const uint64_t *sk_begin, *sk_end;   // device: entry range of sketch i
    uint32_t* d_inter;
    const uint32_t* list_ref = nullptr;  // list reference per entry -- or per record slot, with `where` = slot of every entry
    const uint32_t* where = nullptr;
    uint64_t max_row = 0;                // keys of the longest sketch (0 = unknown): bounds every pair count
    const uint32_t* row_order = nullptr; // rows in the order of their sketches' min-hash (k_row_order), or null
    const uint32_t* multi = nullptr;     // partition form: one bit per record slot -- has the record's key a list? (k_parts_group)
    uint32_t multi_slots = 0;
};
struct CompareJob {
    ComparePlan P;
    std::function<int(uint64_t seed, uint64_t fp_mask, uint32_t log2cap, uint32_t passes, uint32_t pass)> insert;
    std::function<int(uint64_t seed, uint64_t fp_mask, uint32_t log2cap, uint32_t W, bool direct_rows, uint32_t passes, uint32_t pass)> fill;
    uint32_t log2cap = 0, W = 0, lanes_per_key = 64, sblocks = 0;
    uint64_t cap = 0, seed = 0x5350535053505350ULL;
    bool speculative = false;
    uint32_t n_skoff = 0;           // flat form: sketch offsets to bring over from the pinned staging copy
    bool clear_all_flags = false;   // no pass before the first attempt has written flags
    bool direct_rows = false;   // speculative and every row owned from entry 0: row id = the owner's entry index
    int attempt = 0;
    // sparse form (many sketches, all rows owned): set by the flat entry point, see k_insert_sparse
    std::function<int(uint64_t seed, uint64_t fp_mask, uint32_t log2cap)> insert_sparse;
    std::function<int()> fill_sparse;
    bool sparse = false;
    uint32_t passes = 1, pass = 0;  // large builds: the keys are split into classes and the dictionary + colour
                                    // matrix are built class by class, so the matrix never exceeds its budget
    // partition form (flat entry point): see k_parts_scatter
    std::function<int(uint32_t n_parts, bool small, bool filtered, uint32_t fmask, uint32_t classes, uint32_t cls)> scatter_parts;
    std::function<int(uint32_t n_parts, const SpillPlan&, bool want_multi)> group_parts;
    std::function<int(uint32_t n_parts)> group_small;     // small problems: grouping + counting in one kernel (k_parts_group_small)
    bool parts = false, small = false;
    uint32_t n_parts = 0, parts_attempt = 0, n_sub = 0;
    // filtered form (row-partitioned calls): Bloom filter over the owned sketches' keys in front of the scatter
    std::function<int(uint32_t fmask)> build_filter;
    bool filtered = false;
    uint32_t filter_words = 0;
    uint64_t parts_entries = 0;     // records the parts are sized for
    bool bracket_closed = false;    // the kEvCompare bracket of the begin call has been closed already
    // very large inputs (more keys than kMaxKeyParts parts hold): the keys go through the partition form one hash class at a
    // time, the first class's row sums store the cells, the later ones add
    bool has_hi = false;            // k > 32: records carry a second key word
    bool ordered = false;           // this attempt makes a row order (k_row_order)
    bool want_multi = false;        // this attempt makes the has-a-list bits (k_parts_group) and reads back what share of the records have one
    uint32_t classes = 1, cls = 0;
    uint64_t S_behind = 0;          // keys of the first owned row and later sketches: what the scatter deals
    // spill (partition form, unfiltered): the records of parts that overflow are grouped in a table in HBM (k_spill_insert)
    SpillPlan spill;                // room = 0: not part of this attempt
    std::function<int(uint32_t n_parts, const SpillPlan&, int phase, uint32_t classes, uint32_t cls)> spill_parts;   // phase 0: buffers cleared (in front of the scatter), 1: the kernels (behind the grouping)
};
// flags: [0] unsorted input, [1] fingerprint collision, [2] n_rows, [3] malformed slot, [4] slot overflow, [5] table full
static uint64_t job_fp_mask(const CompareJob& J) {
    // test hook: fingerprints of the first attempt cut to a few bits, so distinct keys collide and the retry runs
    static const char* dbg_fp = getenv("SPSP_DEBUG_FP_BITS");
    return (dbg_fp && J.attempt == 0) ? ((1ull << atoi(dbg_fp)) - 1) : ~0ull;
}
// what compare_end waits on: the job's last queued kernel / copy, not the whole stream
static int job_mark_done(spsp_ctx* ctx) {
    if (!ctx->compare_done) SPSP_HIP(hipEventCreateWithFlags(&ctx->compare_done, hipEventDisableTiming));
    SPSP_HIP(hipEventRecord(ctx->compare_done, ctx->stream));
    return SPSP_OK;
}
static int launch_accumulate_sparse(spsp_ctx* ctx, const ComparePlan& P, uint32_t* flags, bool may_emit_cells = false, bool add = false) {
    uint32_t cols = 64;
    while (cols < P.n && cols < (uint32_t)kSparseCols) cols <<= 1;
    // a row's keys are walked by ONE workgroup unless the sketches are huge (few sketches of millions of keys):
    // then slices of the row add into cells cleared first
    const uint64_t per_row = P.n_own ? P.S_own / P.n_own : 0;
    uint32_t split = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, per_row / 65536), 64);
    // ... or a FEW rows are (one eukaryote among ten thousand bacteria): those rows get launches of their own, in slices, and
    // the launch of all rows passes them by -- one workgroup walking 3 x 10^6 keys was 5 ms of tail behind a 0.9 ms kernel, and
    // its length had switched the 16-bit counters off for every row (a counter is bounded by the SHORTER sketch of its pair)
    constexpr uint64_t kLongRow = 65535;
    std::vector<uint32_t> long_rows;
    if (split == 1 && P.max_row > kLongRow && ctx->h_skoff) {
        for (uint32_t i = P.row_first; i < std::min(P.row_limit, P.n) && long_rows.size() <= 64; i += P.row_stride)
            if (ctx->h_skoff[i + 1] - ctx->h_skoff[i] > kLongRow) long_rows.push_back(i);
        if (long_rows.size() > 64) { long_rows.clear(); split = 2; }        // many: every row in slices (the form for huge sketches)
    }
    const uint32_t long_limit = long_rows.empty() ? 0u : (uint32_t)kLongRow;
    if (split > 1 && !add) {
        hipLaunchKernelGGL(k_zero_rows, dim3(std::max(1u, std::min((P.n + 255) / 256, 64u)), P.n_own), dim3(256), 0, ctx->stream,
                           P.n, P.row_first, P.row_stride, P.row_limit, P.d_inter);
        SPSP_HIP(hipGetLastError());
    }
    uint32_t copies_log2 = 0;                                // as many copies of the counters as 64 KiB of LDS hold, up to 16
    while (copies_log2 < 4 && ((size_t)cols << (copies_log2 + 1)) * 4 <= (size_t)kSparseCols * 4) ++copies_log2;
    // rows of one family behind one L2 (see the kernel): when a row is ONE workgroup and there are rows enough to fill the chip
    static const char* dbg_xcd = getenv("SPSP_DEBUG_ACC_XCD");     // "0": rows in launch order (A/B, profiles/r04_acc_xcd.md)
    const uint32_t col_blocks = (P.n + cols - 1) / cols;
    const bool by_xcd = col_blocks == 1 && split == 1 && P.n_own >= 512 && !(dbg_xcd && dbg_xcd[0] == '0');
    const uint32_t xcd_rows = by_xcd ? (P.n_own + 7) / 8 : 0u;
    // 16-bit counters when no row can make one overflow: half the LDS per workgroup -> twice the rows in flight per CU (the
    // kernel waits on two dependent random reads per key: more rows in flight is more of them in flight)
    static const char* dbg_half = getenv("SPSP_DEBUG_ACC_HALF");   // "0": 32-bit counters (A/B)
    const bool half = split == 1 && P.max_row > 0 && (P.max_row <= kLongRow || long_limit) && !(dbg_half && dbg_half[0] == '0');
    // (256-lane workgroups for short rows were measured for the key-partitioned ranks' ~600-key rows: 0.178 -> 0.192 ms -- a row's fixed
    // cost is clearing and scanning its N counters, which takes four times as many rounds with a quarter of the lanes; not kept)
    auto kern = half ? &k_accumulate_sparse<true, false, kSparseThreads> : &k_accumulate_sparse<false, false, kSparseThreads>;
    // a caller that wants the result as sparse cells (compare_cells_run) gets them straight from the row sums
    // when ONE workgroup makes a row (no split, no long rows) -- else the dense matrix is written and sparsified afterwards
    const bool direct = may_emit_cells && ctx->cells_req.armed && split == 1 && long_rows.empty();
    // ... and, when the rows are SHORT (a key-partitioned rank's: ~600 keys of each of 10 000 sketches), by workgroups of 256 lanes
    // that stay and take row after row, a row costing what it touches (TOUCH, see the kernel): 0.204 -> 0.161 ms for such a rank.
    // Long rows keep a workgroup each: noting first touches takes LDS adds that RETURN, ~20 per key, and a 4 800-key row's clear
    // and scan are little beside its keys (configs[3] all-vs-all as cells: 0.775 ms against 0.95 with 1 024 lanes staying, 1.14 with 256)
    static const char* dbg_touch = getenv("SPSP_DEBUG_ACC_TOUCH");   // "0": never; "s" / "l": always, 256 / 1 024 lanes (A/B, tests)
    const bool touch_forced = dbg_touch && (dbg_touch[0] == 's' || dbg_touch[0] == 'l');
    const bool touch = direct && half && col_blocks == 1 && !(dbg_touch && dbg_touch[0] == '0') && (per_row <= 2048 || touch_forced);
    const uint32_t rows_y = by_xcd ? xcd_rows * 8 : P.n_own;
    uint32_t grid_y = rows_y, threads = kSparseThreads;
    size_t lds = ((size_t)cols << copies_log2) * (half ? 2 : 4);
    if (touch) {
        // counters for the N columns there are, not for the next power of two (one column block): more workgroups to a CU
        cols = (P.n + 63u) & ~63u;
        copies_log2 = 0;
        while (copies_log2 < 4 && ((size_t)cols << (copies_log2 + 1)) * 2 <= 32u * 1024u) ++copies_log2;
        lds = ((size_t)cols << copies_log2) * 2;
        const bool small_wg = !(dbg_touch && dbg_touch[0] == 'l');
        threads = small_wg ? 256 : kSparseThreads;
        const uint32_t per_cu = small_wg ? (uint32_t)std::min<size_t>(8, (160u * 1024u) / (lds + kTouchCap * 4 + 512)) : 2u;
        grid_y = std::min(rows_y, std::max(8u, (uint32_t)ctx->n_cu * std::max(1u, per_cu)) & ~7u);
        if (grid_y < 8 || (grid_y & 7u)) grid_y = rows_y;                                        // (fewer than eight rows: one workgroup each)
        kern = small_wg ? &k_accumulate_sparse<true, true, 256> : &k_accumulate_sparse<true, true, kSparseThreads>;
    }
    hipLaunchKernelGGL(kern, dim3(col_blocks, grid_y, split), dim3(threads), lds, ctx->stream,
                       P.list_ref ? P.list_ref : ctx->c_row.as<uint32_t>(), P.where, ctx->c_matrix.as<uint16_t>(), P.sk_begin, P.sk_end,
                       P.n, P.row_first, P.row_stride, P.row_limit, cols, copies_log2, split, P.d_inter, flags,
                       reinterpret_cast<uint32_t*>(ctx->h_scalar + 8),
                       direct ? ctx->cells_req.cells : (unsigned long long*)nullptr,
                       (unsigned long long)ctx->cells_req.cap, ctx->cells_req.count, xcd_rows, add, by_xcd ? P.row_order : (const uint32_t*)nullptr, long_limit, P.multi, P.multi_slots, rows_y);
    SPSP_HIP(hipGetLastError());
    for (uint32_t i : long_rows) {
        const uint64_t keys = ctx->h_skoff[i + 1] - ctx->h_skoff[i];
        const uint32_t slices = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2, keys / 32768), 256);
        if (!add) {
            hipLaunchKernelGGL(k_zero_rows, dim3(std::max(1u, std::min((P.n + 255) / 256, 64u)), 1), dim3(256), 0, ctx->stream, P.n, i, 1u, i + 1, P.d_inter);
            SPSP_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL((k_accumulate_sparse<false, false, kSparseThreads>), dim3(col_blocks, 1, slices), dim3(kSparseThreads), ((size_t)cols << copies_log2) * 4, ctx->stream,
                           P.list_ref ? P.list_ref : ctx->c_row.as<uint32_t>(), P.where, ctx->c_matrix.as<uint16_t>(), P.sk_begin, P.sk_end,
                           P.n, i, 1u, i + 1, cols, copies_log2, slices, P.d_inter, flags, reinterpret_cast<uint32_t*>(ctx->h_scalar + 8),
                           (unsigned long long*)nullptr, 0ull, (unsigned long long*)nullptr, 0u, add, (const uint32_t*)nullptr, 0u, P.multi, P.multi_slots, 1u);
        SPSP_HIP(hipGetLastError());
    }
    ctx->cells_req.direct = direct;
    // the number of cells travels to pinned memory behind the kernel: whoever waits for the job (compare_end) has it, no round trip of its own
    if (direct) SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 12, ctx->cells_req.count, 8, hipMemcpyDeviceToHost, ctx->stream));
    return SPSP_OK;
}
static int job_queue_flags(spsp_ctx* ctx);
// partition form: prepare -> scatter -> group -> row sums, queued in one go (small problems: prepare -> scatter -> group + count)
static int job_parts(spsp_ctx* ctx, CompareJob& J) {
    uint32_t* flags = ctx->c_flags.as<uint32_t>();
    ctx->cells_req.direct = false;                            // (set by the row sums of the general form when they emit cells)
    int rc;
    if ((rc = ctx->c_part_cnt.reserve((size_t)J.n_parts * 4))) return rc;
    if (!J.small && (rc = ctx->c_matrix.reserve(((size_t)J.n_parts * 4 * kPartCap + J.spill.ids_room) * sizeof(uint16_t)))) return rc;     // sketch lists
    if (J.spill.room && (rc = ctx->c_lref.reserve(((size_t)J.spill.lref_base + ((size_t)1 << J.spill.log2cap)) * 4))) return rc;             // (before the scatter asks for less)
    if (J.filtered && (rc = ctx->c_filter.reserve((size_t)J.filter_words * 4))) return rc;
    const uint32_t filter_vec = J.filtered ? J.filter_words / 4 : 0u;
    const uint32_t most = std::max(std::max(std::max(J.n_parts, J.n_skoff), std::max(J.n_sub, J.small ? J.P.n * J.P.n : 0u)), filter_vec);
    hipLaunchKernelGGL(k_parts_prepare, dim3(std::min<uint32_t>((most + 255) / 256, J.filtered ? 1024u : 64u)), dim3(256), 0, ctx->stream,
                       ctx->c_part_cnt.as<uint32_t>(), J.n_parts, flags, (const uint64_t*)ctx->h_skoff,
                       ctx->c_skoff.as<uint64_t>(), J.n_skoff, reinterpret_cast<const uint32_t*>(ctx->h_skoff + J.n_skoff),
                       reinterpret_cast<uint32_t*>(ctx->c_skoff.as<uint64_t>() + J.n_skoff), J.n_sub,
                       J.small ? J.P.d_inter : (uint32_t*)nullptr, J.small ? J.P.n : 0u,
                       J.filtered ? ctx->c_filter.as<uint4>() : (uint4*)nullptr, filter_vec);
    SPSP_HIP(hipGetLastError());
    // analysis hook (results are wrong with it): leave stages out to see what each costs a kernel of another stream
    static const int skip = getenv("SPSP_DEBUG_SKIP_STAGES") ? atoi(getenv("SPSP_DEBUG_SKIP_STAGES")) : 0;
    if ((rc = ctx->ev_begin(kEvScatter))) return rc;
    if (J.filtered && (rc = J.build_filter(J.filter_words - 1))) return rc;
    if (J.spill.room && !J.small && (rc = J.spill_parts(J.n_parts, J.spill, 0, J.classes, J.cls))) return rc;
    // rows of similar sketches side by side for the row sums (their holder lists meet in one L2), whatever order the sketches
    // came in: every sketch's smallest key hash from the records of the first parts (k_row_signature), the order they give
    // (k_row_order).  SPSP_DEBUG_ROW_ORDER=0: launch order, 2: an order for every comparison
    static const char* dbg_order = getenv("SPSP_DEBUG_ROW_ORDER");
    const bool all_rows = J.P.row_first == 0 && J.P.row_stride == 1 && J.P.row_limit >= J.P.n && J.P.n_own == J.P.n;
    // (what a context learnt holds for the collection it learnt it on: another one -- other offsets -- is looked at afresh)
    if (ctx->h_skoff) {
        const uint64_t fp = (uint64_t)J.P.n * 0x9E3779B97F4A7C15ULL ^ J.P.S_entries * 0xC2B2AE3D27D4EB4FULL ^ ctx->h_skoff[1] * 0x165667B19E3779F9ULL ^
                            ctx->h_skoff[J.P.n / 2] * 0x27D4EB2F165667C5ULL ^ ctx->h_skoff[J.P.n - J.P.n / 3] * 0x85EBCA77C2B2AE63ULL;
        if (fp != ctx->learnt_on) { ctx->learnt_on = fp; ctx->order_quiet = 0; ctx->multi_quiet = 0; }   // (spsp_compare_forget resets all of it)
    }
    bool ordered = !J.small && all_rows && J.P.n >= (dbg_order && dbg_order[0] == '2' ? 512u : 2048u) && J.P.n <= (uint32_t)kSparseCols && !(dbg_order && dbg_order[0] == '0');
    // (a context whose last comparison came in a good order of its own -- k_row_order's verdict, read back with the job -- skips
    // the making of the order for the next 63: collections are compared batch after batch of the same kind)
    if (ordered && ctx->order_quiet > 0 && !(dbg_order && dbg_order[0] == '2')) { --ctx->order_quiet; ordered = false; }
    J.ordered = ordered;
    if (ordered && ((rc = ctx->c_sig.reserve((size_t)J.P.n * 8)) || (rc = ctx->c_order.reserve((size_t)J.P.n * 4)))) return rc;
    if (!(skip & 1) && (rc = J.scatter_parts(J.n_parts, J.small, J.filtered, J.filter_words - 1, J.classes, J.cls))) return rc;
    if (ordered) {
        const hipStream_t side = ctx->stream;          // (on a stream of their own, beside the grouping kernel, these three short launches cost the same 0.06 ms: measured)
        SPSP_HIP(hipMemsetAsync(ctx->c_sig.p, 0xff, (size_t)J.P.n * 8, side));
        unsigned long long* sig = ctx->c_sig.as<unsigned long long>();
        const uint32_t sp = std::min<uint32_t>(J.n_parts, kSigParts);
        if (J.has_hi) hipLaunchKernelGGL(k_row_signature<true>, dim3(sp), dim3(1024), 0, side, ctx->c_recs.as<uint64_t>(), ctx->c_part_cnt.as<uint32_t>(), (uint32_t)kPartCap, sig);
        else hipLaunchKernelGGL(k_row_signature<false>, dim3(sp), dim3(1024), 0, side, ctx->c_recs.as<uint64_t>(), ctx->c_part_cnt.as<uint32_t>(), (uint32_t)kPartCap, sig);
        static_assert(kOrderMost == kSparseCols, "the order is made for comparisons of one column block");
        if (!ctx->attr_order_set) {
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_row_order), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((kOrderBuckets + kOrderMost) * 4)));
            ctx->attr_order_set = true;
        }
        hipLaunchKernelGGL(k_row_order, dim3(1), dim3(1024), (size_t)(kOrderBuckets + J.P.n) * 4, side, (const unsigned long long*)sig, J.P.n, ctx->c_order.as<uint32_t>(), flags + 10);
        SPSP_HIP(hipGetLastError());
    }
    if ((rc = ctx->ev_end(kEvScatter))) return rc;
    if ((rc = ctx->ev_begin(kEvGroup))) return rc;
    // the has-a-list bits pay when most keys are held by one sketch only (unrelated genomes: row sums 0.84 -> 0.32 ms); a context
    // whose last comparison had lists for two records in five or more leaves them out for its next 63 (they cost the grouping
    // kernel 0.03 ms and, used, the row sums 0.15 ms at configs[3])
    J.want_multi = !J.small;
    static const char* dbg_multi = getenv("SPSP_DEBUG_MULTI");       // "0": never made, "1": always made and always used (A/B)
    if (dbg_multi && dbg_multi[0] == '0') J.want_multi = false;
    else if (dbg_multi && dbg_multi[0] == '1') {}
    else if (J.want_multi && ctx->multi_quiet > 0) { --ctx->multi_quiet; J.want_multi = false; }
    if (!(skip & 2) && (rc = J.small ? J.group_small(J.n_parts) : J.group_parts(J.n_parts, J.spill, J.want_multi))) return rc;
    if (J.spill.room && !J.small && (rc = J.spill_parts(J.n_parts, J.spill, 1, J.classes, J.cls))) return rc;
    if ((rc = ctx->ev_end(kEvGroup))) return rc;
    if (J.small) return job_queue_flags(ctx);             // (no later kernel forwards the flags)
    if ((rc = ctx->ev_begin(kEvAccumulate))) return rc;
    ComparePlan PP = J.P;
    PP.list_ref = ctx->c_lref.as<uint32_t>(); PP.where = ctx->c_where.as<uint32_t>();
    if (J.want_multi) { PP.multi = ctx->c_multi.as<uint32_t>(); PP.multi_slots = J.n_parts * (uint32_t)kPartCap | ((dbg_multi && dbg_multi[0] == '1') ? 0x80000000u : 0u); }
    if (ordered) PP.row_order = ctx->c_order.as<uint32_t>();
    if (ordered || J.want_multi) SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 14, flags + 10, 16, hipMemcpyDeviceToHost, ctx->stream));   // the verdicts, for compare_job_end
    // (an attempt whose parts overflow leaves this kernel at its first line, before any cell is emitted: the retry emits them once)
    // (with a spill the keys that have columns add into the dense matrix behind the row sums: no cells straight from them)
    //  -- nor with key classes: a pair's count comes in several parts)
    if (!(skip & 4) && (rc = launch_accumulate_sparse(ctx, PP, flags, J.spill.room == 0 && J.classes == 1, J.cls > 0))) return rc;
    if (J.spill.room && J.spill.t_bits != 0xffffffffu) {
        const uint32_t tiles = (J.P.n + kPairTile - 1) / kPairTile;
        hipLaunchKernelGGL(k_spill_pairs, dim3((tiles * (tiles + 1) / 2 + 7) / 8 * 8), dim3(kPairThreads), 0, ctx->stream, ctx->c_bits.as<unsigned long long>(),
                           J.P.n, tiles, J.P.row_first, J.P.row_stride, J.P.row_limit, J.P.d_inter, (const uint32_t*)flags);
        SPSP_HIP(hipGetLastError());
    }
    if ((rc = ctx->ev_end(kEvAccumulate))) return rc;
    return job_mark_done(ctx);
}

// sparse form: the whole pipeline in one go (no size depends on a count the host has to read)
static int job_sparse(spsp_ctx* ctx, CompareJob& J) {
    uint32_t* flags = ctx->c_flags.as<uint32_t>();
    const ComparePlan& P = J.P;
    int rc;
    // table and per-slot counts are cleared together (the counts live where the dense form keeps slot owners)
    const uint64_t t_vec = J.cap * 8 / 16, c_vec = (J.cap * 4 + 15) / 16;
    const uint64_t want = (t_vec + c_vec + 255) / 256, cap_blocks = (uint64_t)ctx->n_cu * 8;
    hipLaunchKernelGGL(k_prepare, dim3((uint32_t)std::max<uint64_t>(1, std::min(want, cap_blocks))), dim3(256), 0, ctx->stream,
                       ctx->c_table.as<uint4>(), t_vec, ctx->c_owner.as<uint4>(), c_vec, flags,
                       (J.attempt == 0 && J.clear_all_flags) ? 16u : 3u,
                       (const uint64_t*)ctx->h_skoff, ctx->c_skoff.as<uint64_t>(), J.n_skoff);
    SPSP_HIP(hipGetLastError());
    if ((rc = J.insert_sparse(J.seed, job_fp_mask(J), J.log2cap))) return rc;
    hipLaunchKernelGGL(k_assign_ranges, dim3(J.sblocks), dim3(kRowThreads), 0, ctx->stream, ctx->c_owner.as<uint32_t>(), J.cap,
                       ctx->c_rowid.as<uint32_t>(), ctx->c_matrix.as<uint16_t>(), flags + 2);
    SPSP_HIP(hipGetLastError());
    if ((rc = J.fill_sparse())) return rc;
    if ((rc = ctx->ev_begin(kEvAccumulate))) return rc;
    if ((rc = launch_accumulate_sparse(ctx, P, flags))) return rc;
    if ((rc = ctx->ev_end(kEvAccumulate))) return rc;
    return job_mark_done(ctx);
}

// dictionary build: table (and row ids unless the owner's entry index serves as the row)
static int job_front(spsp_ctx* ctx, CompareJob& J) {
    ctx->cells_req.direct = false;                            // a colour-matrix form: the dense matrix is written
    uint32_t* flags = ctx->c_flags.as<uint32_t>();
    int rc;
    // speculative: the matrix size is known now (one row per inserted key), so it is cleared by the same launch
    const uint64_t m_bytes = J.speculative ? (uint64_t)J.P.S_own * J.W * 8 : 0;
    if (m_bytes && (rc = ctx->c_matrix.reserve((size_t)((m_bytes + 15) & ~15ull)))) return rc;
    const uint64_t t_vec = J.cap * 8 / 16, m_vec = (m_bytes + 15) / 16;
    const uint64_t want = (t_vec + m_vec + 255) / 256, cap_blocks = (uint64_t)ctx->n_cu * 8;
    hipLaunchKernelGGL(k_prepare, dim3((uint32_t)std::max<uint64_t>(1, std::min(want, cap_blocks))), dim3(256), 0, ctx->stream,
                       ctx->c_table.as<uint4>(), t_vec, ctx->c_matrix.as<uint4>(), m_vec, flags,
                       (J.attempt == 0 && J.clear_all_flags) ? 16u : 3u,    // [3], [4] belong to the slot index pass
                       (const uint64_t*)ctx->h_skoff, ctx->c_skoff.as<uint64_t>(), J.n_skoff);
    SPSP_HIP(hipGetLastError());
    if ((rc = J.insert(J.seed, job_fp_mask(J), J.log2cap, J.passes, J.pass))) return rc;
    if (!J.direct_rows) {
        hipLaunchKernelGGL(k_assign_rows, dim3(J.sblocks), dim3(kRowThreads), 0, ctx->stream, ctx->c_table.as<uint64_t>(), J.cap,
                           ctx->c_rowid.as<uint32_t>(), flags + 2);
        SPSP_HIP(hipGetLastError());
    }
    return SPSP_OK;
}
// colour matrix (room for `rows` rows), the row sums, and the flags on their way to the host
static int job_back(spsp_ctx* ctx, CompareJob& J, uint64_t rows) {
    int rc;
    const ComparePlan& P = J.P;
    if (!J.speculative) {   // sized by the row count just read back
        if ((rc = ctx->c_matrix.reserve((size_t)rows * J.W * 8))) return rc;
        SPSP_HIP(hipMemsetAsync(ctx->c_matrix.p, 0, (size_t)rows * J.W * 8, ctx->stream));
    }
    if ((rc = J.fill(J.seed, job_fp_mask(J), J.log2cap, J.W, J.direct_rows, J.passes, J.pass))) return rc;
    if ((rc = ctx->ev_begin(kEvAccumulate))) return rc;
    hipLaunchKernelGGL(k_accumulate, dim3((J.W + 63) / 64, P.n_own), dim3(kAccThreads), 0, ctx->stream,
                       ctx->c_row.as<uint32_t>(), ctx->c_matrix.as<uint64_t>(), J.W, J.lanes_per_key, P.sk_begin, P.sk_end,
                       P.n, P.row_first, P.row_stride, P.row_limit, P.d_inter, ctx->c_flags.as<uint32_t>(),
                       reinterpret_cast<uint32_t*>(ctx->h_scalar + 8), J.pass > 0);
    SPSP_HIP(hipGetLastError());
    if ((rc = ctx->ev_end(kEvAccumulate))) return rc;
    return job_mark_done(ctx);
}
// the flags travel to pinned host memory as the last item of whatever has been queued ...
static int job_queue_flags(spsp_ctx* ctx) {
    uint32_t* pinned = reinterpret_cast<uint32_t*>(ctx->h_scalar + 8);
    SPSP_HIP(hipMemcpyAsync(pinned, ctx->c_flags.p, kFlags * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    return job_mark_done(ctx);
}
// ... and are read after the one synchronisation
static int job_wait_flags(spsp_ctx* ctx, uint32_t* h_flags) {
    SPSP_HIP(hipEventSynchronize(ctx->compare_done));
    memcpy(h_flags, ctx->h_scalar + 8, kFlags * sizeof(uint32_t));
    if (h_flags[5]) { set_error("dictionary table overflow (internal sizing error)"); return SPSP_ERR_HIP; }
    if (h_flags[3]) { set_error("malformed exchange slot (header, sketch count or key count does not match)"); return SPSP_ERR_FORMAT; }
    if (h_flags[4]) { set_error("an exchange slot overflowed its capacity: partition again with a larger slot_cap"); return SPSP_ERR_OVERFLOW; }
    if (h_flags[0] && !ctx->keys_unordered) { set_error("sketch keys must be strictly increasing by (minimizer, kmer_hi, kmer_lo)"); return SPSP_ERR_ARG; }
    return SPSP_OK;
}

// global-dictionary forms (dense colour matrix / sketch lists): plan and queue the first attempt
static int job_begin_dictionary(spsp_ctx* ctx, CompareJob* J) {
    const ComparePlan& P = J->P;
    int rc;
    J->W = (P.n + 63) / 64;
    J->lanes_per_key = 64;
    if (J->W < 64) { J->lanes_per_key = 1; while (J->lanes_per_key < J->W) J->lanes_per_key <<= 1; }
    // A matrix with one row per inserted KEY (an upper bound on the distinct keys) is cheap for small inputs:
    // then the whole pipeline is queued without waiting for the row count and checked once at the end.
    static const char* dbg_budget = getenv("SPSP_DEBUG_MATRIX_BUDGET");    // test hook: bytes the matrix may take
    const uint64_t worst_matrix = (uint64_t)P.S_own * J->W * 8;
    J->speculative = !dbg_budget && worst_matrix <= (256ull << 20);
    // many sketches, every row owned: sketch lists instead of colour rows (SPSP_DEBUG_SPARSE=1/0 forces the choice)
    static const char* dbg_sparse = getenv("SPSP_DEBUG_SPARSE");
    const bool all_owned = P.row_stride == 1 && P.row_first == 0 && P.row_limit >= P.n && P.n_own == P.n;
    J->sparse = J->insert_sparse && all_owned && 8 * P.S_entries < (1ull << 29) &&
                (dbg_sparse ? atoi(dbg_sparse) != 0 : (J->W >= 64 && !dbg_budget));
    if (J->sparse) J->speculative = true;       // queued in one go, checked once
    // Large builds: the colour matrix is rows x N bits and grows with N * (distinct keys) -- at tens of thousands of
    // sketches it would outgrow the card.  The reference bounds the same structure by working bucket by bucket
    // (Comparator.cpp:58-68); here the keys are split into hash classes and one class is built and summed at a time.
    J->passes = 1;
    if (!J->speculative && !J->sparse) {
        uint64_t budget;
        if (dbg_budget) budget = (uint64_t)atoll(dbg_budget);
        else {
            size_t free_b = 0, total_b = 0;
            SPSP_HIP(hipMemGetInfo(&free_b, &total_b));
            budget = (uint64_t)free_b / 2 + ctx->c_matrix.cap / 2;          // what is already reserved counts as available
        }
        if (budget < (1ull << 16)) budget = 1ull << 16;
        const uint64_t want = (worst_matrix + budget - 1) / budget;
        J->passes = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want, 1), 4096);
    }
    // table for one class: twice its keys (+25 % for the spread of the class sizes)
    const uint64_t class_keys = J->passes == 1 ? P.S_own : P.S_own / J->passes + P.S_own / J->passes / 4 + 1024;
    J->log2cap = 10;
    while ((1ull << J->log2cap) < 2 * class_keys) ++J->log2cap;
    J->cap = 1ull << J->log2cap;
    if ((rc = ctx->c_table.reserve((size_t)J->cap * 8))) return rc;
    if ((rc = ctx->c_owner.reserve((size_t)J->cap * 4))) return rc;
    if ((rc = ctx->c_rowid.reserve((size_t)J->cap * 4))) return rc;
    if ((rc = ctx->c_row.reserve((size_t)P.S_entries * 4))) return rc;
    J->sblocks = (uint32_t)((J->cap + (uint64_t)kRowThreads * kRowSlots - 1) / ((uint64_t)kRowThreads * kRowSlots));
    // owned entries form the prefix [0, S_own) of the entry space when rows are not strided over ranks
    J->direct_rows = J->speculative && !J->sparse && P.row_stride == 1 && P.row_first == 0;
    if ((rc = ctx->c_slot_lo.reserve((size_t)J->cap * 8))) return rc;
    if ((rc = ctx->c_slot_hi.reserve((size_t)J->cap * 8))) return rc;
    if ((rc = ctx->c_slot_mn.reserve((size_t)J->cap * 4))) return rc;
    if (J->sparse) {
        if ((rc = ctx->c_matrix.reserve((size_t)8 * P.S_entries * sizeof(uint16_t) + 16))) return rc;   // lists: length + ids, in whole 16-byte words
        if ((rc = job_sparse(ctx, *J))) return rc;
    } else {
        if ((rc = job_front(ctx, *J))) return rc;
        if (J->speculative) { if ((rc = job_back(ctx, *J, P.S_own))) return rc; }   // k_accumulate forwards the flags
        else if ((rc = job_queue_flags(ctx))) return rc;                             // the row count is needed first
    }
    return SPSP_OK;
}

// records a part is planned to hold on average: kPartCap less the spread of the part sizes (keys shared by c
// sketches arrive c at a time); halved for the second attempt
static bool dbg_mean_set() { static const bool v = getenv("SPSP_DEBUG_PART_MEAN") != nullptr; return v; }
static uint32_t parts_small(uint64_t entries) {
    static const char* dbg_small_mean = getenv("SPSP_DEBUG_SMALL_MEAN");   // test hook: tiny parts, so that a small input reaches the part limit
    const uint64_t mean = dbg_small_mean ? (uint64_t)std::max(1, atoi(dbg_small_mean)) : (uint64_t)kSmallMean;
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (entries + mean - 1) / mean), 0xffffffffull);
}
static uint32_t parts_for(uint64_t entries, uint32_t attempt) {
    static const char* dbg_mean = getenv("SPSP_DEBUG_PART_MEAN");   // test hook: tiny parts, so that small inputs reach tens of thousands of parts
    const uint64_t mean = dbg_mean ? (uint64_t)std::max(1, atoi(dbg_mean)) : (attempt == 0 ? 2900 : 1400);
    return (uint32_t)std::max<uint64_t>(1, (entries + mean - 1) / mean);
}

// Can the records of overflowed parts (`records` of them) be grouped beside n_parts parts?  The lists must stay inside
// the 2^29 u16 a list reference can address and the list references inside a `where` word.
static bool spill_enabled() {
    static const char* dbg_spill = getenv("SPSP_DEBUG_SPILL");          // "0": the forms of before (half-size parts, then the global dictionary)
    return !(dbg_spill && dbg_spill[0] == '0');
}
static bool spill_plan(const CompareJob& J, uint64_t records, SpillPlan* sp) {
    if (!spill_enabled()) return false;
    if (!J.spill_parts || J.small || J.filtered || records == 0) return false;
    SpillPlan P;
    P.room = std::min<uint64_t>(records, J.P.S_entries);
    P.log2cap = 10;
    while ((1ull << P.log2cap) < 2 * P.room) ++P.log2cap;
    const uint64_t ids_base = (uint64_t)J.n_parts * 4 * kPartCap, ids_room = 4 * P.room + 8, lref_base = (uint64_t)J.n_parts * kPartCap;
    if (P.log2cap > 31 || ids_base + ids_room > (1ull << 29) || lref_base + (1ull << P.log2cap) >= 0xffffffffull) return false;
    P.ids_base = (uint32_t)ids_base; P.ids_room = (uint32_t)ids_room; P.lref_base = (uint32_t)lref_base;
    // a key held by h sketches costs the row sums h^2 / 2 LDS adds and the pair kernel N^2 / 128 word operations: columns from
    // N / 24 holders on (measured rates); SPSP_DEBUG_SPILL_BITS=<holders> sets the threshold, 0 = lists only
    static const char* dbg_bits = getenv("SPSP_DEBUG_SPILL_BITS");
    P.t_bits = dbg_bits ? (atoi(dbg_bits) > 0 ? (uint32_t)std::max(2, atoi(dbg_bits)) : 0xffffffffu) : std::max(64u, J.P.n / 24u);
    P.max_cols = P.t_bits == 0xffffffffu ? 0u : (uint32_t)(J.P.S_entries / P.t_bits + 1);   // (any part may hold such keys, overflowed or not)
    *sp = P;
    return true;
}

// takes ownership of `job`; on success it is pending on the context until compare_job_end
static int compare_job_begin(spsp_ctx* ctx, CompareJob* job) {
    std::unique_ptr<CompareJob> J(job);
    const ComparePlan& P = J->P;
    int rc;
    // Partition form unless a test hook asks for one of the global-dictionary forms (SPSP_DEBUG_PARTS=0 by name)
    static const bool hooks = getenv("SPSP_DEBUG_MATRIX_BUDGET") || getenv("SPSP_DEBUG_SPARSE") || getenv("SPSP_DEBUG_FP_BITS");
    static const char* dbg_parts = getenv("SPSP_DEBUG_PARTS");
    // More keys than kMaxKeyParts parts hold (9 x 10^7: tens of thousands of sketches): one hash class of the keys per pass
    // through the same parts (SPSP_DEBUG_KEY_CLASSES=<n> forces n classes on small inputs)
    static const char* dbg_classes = getenv("SPSP_DEBUG_KEY_CLASSES");
    // (with several classes, half the parts a single one may take: the lists of a spill fit behind them, spill_plan)
    J->classes = dbg_classes ? (uint32_t)std::max(1, atoi(dbg_classes))
                             : parts_for(P.S_entries, 0) <= (uint32_t)kMaxKeyParts ? 1u : (uint32_t)((parts_for(P.S_entries, 0) + kMaxKeyParts / 2 - 1) / (kMaxKeyParts / 2));
    J->parts = J->scatter_parts && (dbg_parts ? atoi(dbg_parts) != 0 : !hooks) && J->classes <= 256;
    if (J->parts) {
        J->speculative = true;                      // queued in one go, checked once
        // small problems (bench.py's 100 sketches): one kernel groups and counts.  Every row must be owned (the parts
        // add into the whole matrix, which is cleared first) and k <= 32; SPSP_DEBUG_SMALL=0 keeps the general form
        static const bool small_off = getenv("SPSP_DEBUG_SMALL") && atoi(getenv("SPSP_DEBUG_SMALL")) == 0;
        // ... and its parts are half the size, so the part count is checked on its own: the scatter's LDS counters and
        // the 15 part bits of its (part, rank) word hold kMaxKeyParts parts, not more (two sketches of 3 x 10^7 keys each
        // would otherwise ask for 41 000)
        J->small = J->group_small && !small_off && !dbg_mean_set() && P.n <= (uint32_t)kSmallN && P.n_own == P.n && P.row_first == 0 && P.row_stride == 1 &&
                   P.row_limit >= P.n && parts_small(P.S_entries) <= (uint32_t)kMaxKeyParts;
        if (ctx->spill_expect) J->small = false;     // (the context's last comparison overflowed its parts: straight to the form that spills)
        // row-partitioned call that owns at most 3/4 of the keys: the other sketches' keys go through a filter first
        // (k_parts_filter) and the parts are sized for what got through last time (the count comes back with the flags;
        // an attempt that overflows is repeated with the exact count).  SPSP_DEBUG_FILTER=0/1 forces the choice.
        // A filter that lets most keys through costs more than it saves (rows dealt i % G over sketches that come in
        // families: 65 % pass): once a call has measured that, later ones skip it and look again now and then.
        static const char* dbg_filter = getenv("SPSP_DEBUG_FILTER");
        const uint64_t S_behind = P.S_entries - ctx->h_skoff[std::min(P.row_first, P.n)];   // keys of the first owned row and later sketches
        const bool pays = ctx->filter_ratio * (double)P.S_own <= 0.5 * (double)S_behind || (++ctx->filter_skipped & 255u) == 0;
        J->filtered = !J->small && J->build_filter && P.n_own < P.n &&
                      (dbg_filter ? atoi(dbg_filter) != 0 : (4 * P.S_own <= 3 * S_behind && pays));
        J->parts_entries = S_behind;
        if (J->filtered) {
            uint32_t words = 1024;                  // 16 bits per owned key, a power of two of 32-bit words
            while ((uint64_t)words * 2 < P.S_own && words < (1u << 28)) words <<= 1;
            J->filter_words = words;
            const double ratio = std::min(std::max(ctx->filter_ratio, 1.0), (double)S_behind / (double)P.S_own);
            J->parts_entries = std::min<uint64_t>(S_behind, (uint64_t)((double)P.S_own * ratio * 1.08) + 2900);
        }
        J->S_behind = S_behind;
        if (J->classes > 1) {
            J->small = false; J->filtered = false;
            J->parts_entries = S_behind / J->classes + S_behind / J->classes / 32 + 65536;
            if (parts_for(J->parts_entries, 0) > (uint32_t)kMaxKeyParts) { set_error("internal: a key class needs %u parts", parts_for(J->parts_entries, 0)); return SPSP_ERR_ARG; }
        }
        J->n_parts = J->small ? parts_small(P.S_entries) : parts_for(J->parts_entries, 0);
        if ((rc = ctx->c_row.reserve((size_t)P.S_entries * 4))) return rc;
        // the context's last comparison spilled: this one is queued with room for as much again (a collection is compared
        // batch after batch of the same kind); a guess that turns out too small costs one more attempt, nothing else
        if (ctx->spill_expect && !spill_plan(*J, ctx->spill_expect + ctx->spill_expect / 4 + 4096, &J->spill)) J->spill = SpillPlan{};
        if ((rc = job_parts(ctx, *J))) return rc;
    } else if ((rc = job_begin_dictionary(ctx, J.get()))) return rc;
    ctx->compare_job = J.release();
    return SPSP_OK;
}

int compare_job_end(spsp_ctx* ctx) {
    if (!ctx->compare_job) { set_error("no comparison is pending on this context"); return SPSP_ERR_ARG; }
    std::unique_ptr<CompareJob> J(ctx->compare_job);
    ctx->compare_job = nullptr;
    int rc;
    while (J->parts) {
        uint32_t h_flags[kFlags];
        if ((rc = job_wait_flags(ctx, h_flags))) return rc;
        if (J->filtered && h_flags[7]) ctx->filter_ratio = (double)h_flags[7] / (double)J->P.S_own;
        if (!h_flags[6]) {
            if (!J->filtered && !J->small) ctx->spill_expect = h_flags[2];
            if (J->ordered) {
                const uint32_t near_new = (uint32_t)ctx->h_scalar[14], near_in = (uint32_t)(ctx->h_scalar[14] >> 32);
                ctx->order_quiet = 2 * near_in >= near_new ? 63 : 0;
            }
            if (J->want_multi) {
                const uint64_t listed = (uint32_t)ctx->h_scalar[15], sampled = (uint32_t)(ctx->h_scalar[15] >> 32);
                ctx->multi_quiet = (sampled && 5 * listed >= 2 * sampled) ? 63 : 0;
            }
            static const bool trace = getenv("SPSP_DEBUG_SPILL_TRACE") != nullptr;     // test hook: which way the comparison went
            if (trace) fprintf(stderr, "spsp compare: %u sketches, %s form, %u parts, classes %u, row order %s, has-a-list bits %s, spill %s\n", J->P.n,
                               J->small ? "small" : (J->filtered ? "filtered" : "partition"), J->n_parts, J->classes, J->ordered ? "made" : "as given",
                               J->want_multi ? "made" : "left out", J->spill.room ? "yes" : "no");
            if (trace && J->spill.room) fprintf(stderr, "spsp spill: %u records of overflowed parts grouped in HBM (room %llu, %u parts, columns from %u holders)\n",
                                                h_flags[2], (unsigned long long)J->spill.room, J->n_parts, J->spill.t_bits);
            if (J->cls + 1 < J->classes) {              // the next class of keys through the same parts; its row sums add
                ++J->cls;
                J->bracket_closed = true;
                if ((rc = job_parts(ctx, *J))) return rc;
                continue;
            }
            return SPSP_OK;
        }
        // a part overflowed (many sketches share their keys): the same parts once more with the records of the
        // overflowed ones grouped in HBM (spill); where that does not apply, parts half the size, then the
        // global-dictionary forms, which have no such limit
        J->bracket_closed = true;
        if (J->filtered && h_flags[7] > J->parts_entries) {   // more keys passed the filter than the parts were sized for: now the count is known
            J->parts_entries = (uint64_t)h_flags[7] + (uint64_t)h_flags[7] / 64 + 2900;
            if (parts_for(J->parts_entries, J->parts_attempt) <= (uint32_t)kMaxKeyParts) {
                J->n_parts = parts_for(J->parts_entries, J->parts_attempt);
                if ((rc = job_parts(ctx, *J))) return rc;
                continue;
            }
        }
        if (J->small) {                             // the small-problem form's parts are half the size: the general form next
            J->small = false;
            J->n_parts = parts_for(J->P.S_entries, 0);
            if ((rc = ctx->c_row.reserve((size_t)J->P.S_entries * 4))) return rc;
            if ((rc = job_parts(ctx, *J))) return rc;
            continue;
        }
        if (!J->small && h_flags[2] > J->spill.room && spill_plan(*J, h_flags[2], &J->spill)) {   // (the count is exact: the same parts overflow again)
            if ((rc = job_parts(ctx, *J))) return rc;
            continue;
        }
        J->spill = SpillPlan{};
        // no room for the spilled keys' lists behind this many parts (list references address 2^29 u16): twice the classes,
        // so half the parts -- everything starts over, a key's class changes
        if (!J->filtered && spill_enabled() && J->spill_parts && J->n_parts > 4096 && J->classes <= 128 && h_flags[2] > 0) {
            J->classes *= 2; J->cls = 0;
            J->parts_entries = J->S_behind / J->classes + J->S_behind / J->classes / 32 + 65536;
            J->n_parts = parts_for(J->parts_entries, 0);
            if ((rc = job_parts(ctx, *J))) return rc;
            continue;
        }
        if (J->parts_attempt == 0 && parts_for(J->parts_entries, 1) <= (uint32_t)kMaxKeyParts) {
            J->parts_attempt = 1;
            J->n_parts = parts_for(J->parts_entries, 1);
            if ((rc = job_parts(ctx, *J))) return rc;
            continue;
        }
        J->parts = false;
        J->clear_all_flags = true;
        if ((rc = job_begin_dictionary(ctx, J.get()))) return rc;
    }
    for (;;) {
        uint32_t h_flags[kFlags];
        bool collided = false;
        if (J->speculative) {
            if ((rc = job_wait_flags(ctx, h_flags))) return rc;              // collisions surface in the fill pass
            collided = h_flags[1] != 0;
        } else {
            for (;;) {   // key class by key class; the first class (front part) was queued by the begin call / the retry below
                if ((rc = job_wait_flags(ctx, h_flags))) return rc;          // the row count sizes the colour matrix
                if ((rc = job_back(ctx, *J, h_flags[2]))) return rc;
                if (J->attempt == 0 && J->pass + 1 == J->passes && !J->bracket_closed && (rc = ctx->ev_end(kEvCompare))) return rc;   // bracket of the begin call
                if ((rc = job_wait_flags(ctx, h_flags))) return rc;          // collisions of this class
                if (h_flags[1]) { collided = true; break; }
                if (++J->pass == J->passes) break;
                if ((rc = job_front(ctx, *J))) return rc;
                if ((rc = job_queue_flags(ctx))) return rc;
            }
        }
        if (!collided) return SPSP_OK;
        if (J->attempt >= 4) { set_error("fingerprint collisions persisted over 5 seeds"); return SPSP_ERR_HIP; }
        ++J->attempt;
        J->pass = 0;   // a class that collided has already been added into the counts: start over (the first class overwrites)
        J->seed = J->seed * 6364136223846793005ULL + 1442695040888963407ULL;  // new fingerprints, try again
        if (J->sparse) { if ((rc = job_sparse(ctx, *J))) return rc; continue; }
        if ((rc = job_front(ctx, *J))) return rc;
        if (J->speculative) { if ((rc = job_back(ctx, *J, J->P.S_own))) return rc; }
        else if ((rc = job_queue_flags(ctx))) return rc;
    }
}

void compare_job_drop(spsp_ctx* ctx) {
    delete ctx->compare_job;
    ctx->compare_job = nullptr;
}

// pinned copy of the caller's offsets: the queued H2D copy must not read memory the caller may free
// tiles (k_parts_scatter_tiles): behind the sub-chunk table (8-byte aligned), tile_info[tile] = block | s << 32 | T << 48 -- the
// tile is range s of the T ranges of its block of tile_sk sketches; a block gets one range per kTileTarget entries (none when
// it is empty or lies in front of the first owned row).  SPSP_DEBUG_TILES=0: no tiles (the scatter over consecutive entries);
// =<n>: n entries per tile; SPSP_DEBUG_TILE_SK=<8|16|32|64>: sketches per block
constexpr uint32_t kTileTarget = 4032, kTileSk = 32, kTileMinSketches = 2 * kTileSk;
static int stage_sk_off(spsp_ctx* ctx, const uint64_t* h_sk_off, uint32_t n, uint32_t row_first, uint32_t* n_tiles_out, uint32_t* tile_sk_out) {
    // n + 1 offsets, then (partition form) one u32 per sub-chunk of kScatSub entries: the sketch holding its first entry
    const uint64_t S = h_sk_off[n];
    const size_t n_sub = (size_t)((S + kScatSub - 1) / kScatSub);
    static const char* dbg_tiles = getenv("SPSP_DEBUG_TILES");
    static const char* dbg_sk = getenv("SPSP_DEBUG_TILE_SK");
    const uint32_t target = dbg_tiles && atoi(dbg_tiles) >= 32 ? (uint32_t)atoi(dbg_tiles) : kTileTarget;
    uint32_t tile_sk = kTileSk;
    if (dbg_sk && (atoi(dbg_sk) == 8 || atoi(dbg_sk) == 16 || atoi(dbg_sk) == 32 || atoi(dbg_sk) == 64)) tile_sk = (uint32_t)atoi(dbg_sk);
    const bool tiles = !(dbg_tiles && atoi(dbg_tiles) == 0) && n >= kTileMinSketches;
    const uint32_t n_blocks = tiles ? (n + tile_sk - 1) / tile_sk : 0u;
    uint64_t n_tiles = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t j0 = b * tile_sk, j1 = std::min(n, j0 + tile_sk);
        const uint64_t E = h_sk_off[j1] - h_sk_off[j0];
        if (j1 > row_first && E) n_tiles += std::min<uint64_t>((E + target - 1) / target, 0xffffu);
    }
    const size_t sub_words = (n_sub + 1) / 2;              // u64 words of the sub-chunk table
    const size_t need = (size_t)(n + 1) + sub_words + (size_t)n_tiles;
    if (ctx->h_skoff_cap < need) {
        if (ctx->h_skoff) { SPSP_HIP(hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->h_skoff); ctx->h_skoff = nullptr; ctx->h_skoff_cap = 0; }
        size_t cap = 1024;
        while (cap < need) cap *= 2;
        SPSP_HIP(hipHostMalloc((void**)&ctx->h_skoff, cap * 8, hipHostMallocDefault));
        ctx->h_skoff_cap = cap;
    }
    memcpy(ctx->h_skoff, h_sk_off, (size_t)(n + 1) * 8);   // the prepare kernel copies it to c_skoff; no job is pending, so the stream has drained
    uint32_t* sub = reinterpret_cast<uint32_t*>(ctx->h_skoff + n + 1);
    uint32_t j = 0;
    for (size_t c = 0; c < n_sub; ++c) {
        const uint64_t e = (uint64_t)c * kScatSub;
        while (j + 1 < n && h_sk_off[j + 1] <= e) ++j;
        sub[c] = j;
    }
    if (n_sub & 1) sub[n_sub] = 0;
    uint64_t* info = ctx->h_skoff + n + 1 + sub_words;
    uint64_t t = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t j0 = b * tile_sk, j1 = std::min(n, j0 + tile_sk);
        const uint64_t E = h_sk_off[j1] - h_sk_off[j0];
        if (!(j1 > row_first && E)) continue;
        const uint64_t T = std::min<uint64_t>((E + target - 1) / target, 0xffffu);
        for (uint64_t s = 0; s < T; ++s) info[t++] = (uint64_t)b | (s << 32) | (T << 48);
    }
    *n_tiles_out = (uint32_t)n_tiles; *tile_sk_out = tile_sk;
    return ctx->c_skoff.reserve(need * 8);
}

// returns 1 when there is nothing to do (no job queued), 0 when a job is pending, < 0 on error
static int compare_device_begin_inner(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                                      const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit,
                                      uint32_t row_first, uint32_t row_stride, uint32_t* d_inter) {
    if (n == 0) return 1;
    if (n > 65535) { set_error("at most 65535 sketches (the reference's uint32 pair key, Comparator.h:26)"); return SPSP_ERR_ARG; }
    if (row_stride == 0) { set_error("bad row partition %u/%u", row_first, row_stride); return SPSP_ERR_ARG; }
    for (uint32_t i = 0; i < n; ++i)   // grids and ranges are derived from these: a decreasing offset must not reach a kernel
        if (h_sk_off[i + 1] < h_sk_off[i]) { set_error("sketch offsets must be non-decreasing (sketch %u)", i); return SPSP_ERR_ARG; }
    const uint64_t S = h_sk_off[n];
    if (S && k > 32 && !d_hi) { set_error("k=%u needs kmer_hi", k); return SPSP_ERR_ARG; }
    if (S > 0xfffffff0ull) { set_error("too many sketch k-mers for one call"); return SPSP_ERR_OVERFLOW; }
    uint64_t S_own = 0;
    uint32_t n_own = 0;
    if (row_limit > n) row_limit = n;
    for (uint32_t i = row_first; i < row_limit; i += row_stride) { S_own += h_sk_off[i + 1] - h_sk_off[i]; ++n_own; }
    if (n_own == 0) return 1;
    if (S_own == 0) {   // the owned sketches are all empty: their rows are zero (the cells a call owns are always written, spsp.h)
        hipLaunchKernelGGL(k_zero_rows, dim3(std::max(1u, std::min((n + 255) / 256, 64u)), n_own), dim3(256), 0, ctx->stream,
                           n, row_first, row_stride, row_limit, d_inter);
        SPSP_HIP(hipGetLastError());
        return 1;
    }
    int rc;
    // staging copy of the offsets (one job may be pending per context, and the previous one has been collected)
    uint32_t n_tiles = 0, tile_sk = 0;
    if ((rc = stage_sk_off(ctx, h_sk_off, n, row_first, &n_tiles, &tile_sk))) return rc;
    if ((rc = ctx->c_flags.reserve(256))) return rc;
    Keys K{d_min, d_lo, (k > 32) ? d_hi : nullptr, ~0ull};
    const uint64_t* sk = ctx->c_skoff.as<uint64_t>();
    uint32_t* flags = ctx->c_flags.as<uint32_t>();
    uint64_t max_all = 0;
    for (uint32_t i = 0; i < n; ++i) max_all = std::max(max_all, h_sk_off[i + 1] - h_sk_off[i]);
    const dim3 grid_all((uint32_t)((max_all + 255) / 256), n);
    CompareJob* J = new CompareJob;
    J->P = ComparePlan{S_own, S, n, n_own, row_first, row_stride, row_limit, sk, sk + 1, d_inter};
    J->P.max_row = max_all;
    J->clear_all_flags = true;
    J->n_skoff = n + 1;
    J->insert = [=](uint64_t seed, uint64_t fp_mask, uint32_t log2cap, uint32_t passes, uint32_t pass) -> int {
        Keys Km = K; Km.fp_mask = fp_mask;
        hipLaunchKernelGGL(k_insert, grid_all, dim3(256), 0, ctx->stream, Km, sk, n, S, row_first, row_stride, row_limit,
                           seed, ctx->c_table.as<uint64_t>(), log2cap, ctx->c_owner.as<uint32_t>(),
                           SlotKeys{ctx->c_slot_lo.as<uint64_t>(), ctx->c_slot_hi.as<uint64_t>(), ctx->c_slot_mn.as<uint32_t>()}, flags,
                           passes, pass);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    J->fill = [=](uint64_t seed, uint64_t fp_mask, uint32_t log2cap, uint32_t W, bool direct_rows, uint32_t passes, uint32_t pass) -> int {
        Keys Km = K; Km.fp_mask = fp_mask;
        hipLaunchKernelGGL(k_fill, grid_all, dim3(256), 0, ctx->stream, Km, sk, n, S, row_first, row_stride, row_limit, seed,
                           ctx->c_table.as<uint64_t>(), log2cap, ctx->c_owner.as<uint32_t>(),
                           direct_rows ? (const uint32_t*)nullptr : ctx->c_rowid.as<uint32_t>(),
                           SlotKeys{ctx->c_slot_lo.as<uint64_t>(), ctx->c_slot_hi.as<uint64_t>(), ctx->c_slot_mn.as<uint32_t>()},
                           W, ctx->c_matrix.as<unsigned long long>(), ctx->c_row.as<uint32_t>(), flags, passes, pass);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    J->insert_sparse = [=](uint64_t seed, uint64_t fp_mask, uint32_t log2cap) -> int {
        Keys Km = K; Km.fp_mask = fp_mask;
        hipLaunchKernelGGL(k_insert_sparse, grid_all, dim3(256), 0, ctx->stream, Km, sk, seed, ctx->c_table.as<uint64_t>(), log2cap,
                           ctx->c_owner.as<uint32_t>(),
                           SlotKeys{ctx->c_slot_lo.as<uint64_t>(), ctx->c_slot_hi.as<uint64_t>(), ctx->c_slot_mn.as<uint32_t>()},
                           ctx->c_row.as<uint32_t>(), flags);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    J->fill_sparse = [=]() -> int {
        hipLaunchKernelGGL(k_fill_sparse, grid_all, dim3(256), 0, ctx->stream, K, sk, ctx->c_row.as<uint32_t>(),
                           SlotKeys{ctx->c_slot_lo.as<uint64_t>(), ctx->c_slot_hi.as<uint64_t>(), ctx->c_slot_mn.as<uint32_t>()},
                           ctx->c_owner.as<uint32_t>(), ctx->c_rowid.as<uint32_t>(), ctx->c_matrix.as<uint16_t>(), flags);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    const bool has_hi = K.hi != nullptr;
    J->has_hi = has_hi;
    const uint32_t* sub_sk = reinterpret_cast<const uint32_t*>(sk + n + 1);
    const uint32_t n_sub_real = (uint32_t)((S + kScatSub - 1) / kScatSub);
    const uint32_t sub_words = (n_sub_real + 1) / 2;
    J->n_sub = 2 * (sub_words + n_tiles);                        // u32 words behind the offsets that k_parts_prepare brings over: sub-chunk sketches, then the tile table
    const uint64_t* tile_info = sk + n + 1 + sub_words;
    J->build_filter = [=](uint32_t fmask) -> int {
        uint64_t max_own = 0;
        for (uint32_t i = row_first; i < row_limit; i += row_stride) max_own = std::max(max_own, ctx->h_skoff[i + 1] - ctx->h_skoff[i]);
        const dim3 grid((uint32_t)std::max<uint64_t>(1, (max_own + 255) / 256), n_own);
        if (has_hi) hipLaunchKernelGGL(k_parts_filter<true>, grid, dim3(256), 0, ctx->stream, K, sk, n, row_first, row_stride, row_limit, ctx->c_filter.as<uint32_t>(), fmask);
        else hipLaunchKernelGGL(k_parts_filter<false>, grid, dim3(256), 0, ctx->stream, K, sk, n, row_first, row_stride, row_limit, ctx->c_filter.as<uint32_t>(), fmask);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    // (nothing of the sketches in front of the first owned row is dealt: the grid starts at that row's chunk)
    const uint64_t e_own = h_sk_off[row_first] / (4u * kScatThreads) * (4u * kScatThreads);
    J->scatter_parts = [=](uint32_t n_parts, bool small, bool filtered, uint32_t fmask, uint32_t classes, uint32_t cls) -> int {
        const uint32_t cap = small ? (uint32_t)kSmallCap : (uint32_t)kPartCap;
        int r2 = ctx->c_recs.reserve((size_t)n_parts * cap * (has_hi ? 24 : 16));
        if (r2) return r2;
        if (!small && (r2 = ctx->c_where.reserve((size_t)S * 4 + 16))) return r2;
        if (!small && (r2 = ctx->c_lref.reserve((size_t)n_parts * kPartCap * 4))) return r2;
        uint32_t* where = small ? nullptr : ctx->c_where.as<uint32_t>();
        static_assert(kMaxKeyParts < (1 << 15), "k_parts_scatter keeps the part in 15 bits of its (part, rank) word");
        if (n_parts > (uint32_t)kMaxKeyParts) { set_error("internal: %u key parts exceed the scatter's limit of %d", n_parts, kMaxKeyParts); return SPSP_ERR_ARG; }
        // tiles: the general partition form (a filtered call's scatter starts at its first owned row's chunk and drops most of
        // what it reads; the small form keeps no `where` and its 100 sketches are one block anyway)
        if (!small && !filtered && n_tiles) {
            const size_t lds_t = (size_t)n_parts * 4;
            if (lds_t > 48 * 1024 && !ctx->attr_scatter_tiles_set) {
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_scatter_tiles<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKeyParts * 4));
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_scatter_tiles<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKeyParts * 4));
                ctx->attr_scatter_tiles_set = true;
            }
            if (has_hi) hipLaunchKernelGGL(k_parts_scatter_tiles<true>, dim3(n_tiles), dim3(kScatThreads), lds_t, ctx->stream, K, sk, n, tile_info, tile_sk, n_parts,
                                           cap, ctx->c_part_cnt.as<uint32_t>(), ctx->c_recs.as<uint64_t>(), where, flags, !ctx->keys_unordered, row_first, classes, cls);
            else hipLaunchKernelGGL(k_parts_scatter_tiles<false>, dim3(n_tiles), dim3(kScatThreads), lds_t, ctx->stream, K, sk, n, tile_info, tile_sk, n_parts,
                                    cap, ctx->c_part_cnt.as<uint32_t>(), ctx->c_recs.as<uint64_t>(), where, flags, !ctx->keys_unordered, row_first, classes, cls);
            SPSP_HIP(hipGetLastError());
            return SPSP_OK;
        }
        const uint32_t per_wg = 4u * kScatThreads;
        const uint64_t e_first = e_own;
        const dim3 grid((uint32_t)((S - e_first + per_wg - 1) / per_wg));
        const uint32_t* filter = filtered ? ctx->c_filter.as<uint32_t>() : nullptr;
        const size_t lds = (size_t)n_parts * 4;
        if (lds > 48 * 1024 && !ctx->attr_scatter_set) {
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_scatter<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKeyParts * 4));
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_scatter<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKeyParts * 4));
            ctx->attr_scatter_set = true;
        }
#define SPSP_SCATTER(HI, E) hipLaunchKernelGGL((k_parts_scatter<HI, E>), grid, dim3(kScatThreads), lds, ctx->stream, K, sk, n, sub_sk, S, \
                                               n_parts, cap, ctx->c_part_cnt.as<uint32_t>(), ctx->c_recs.as<uint64_t>(), where, flags, !ctx->keys_unordered, \
                                               filter, fmask, e_first, row_first, row_stride, row_limit, classes, cls)
        if (has_hi) SPSP_SCATTER(true, 4);
        else SPSP_SCATTER(false, 4);
#undef SPSP_SCATTER
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    J->group_parts = [=](uint32_t n_parts, const SpillPlan& sp, bool want_multi) -> int {
        const size_t lds = (size_t)kPartCap * (8 + (has_hi ? 8 : 0) + 4) + (size_t)kPartSlots * 4 + 16;
        // (a spill attempt: keys of many holders get columns here too -- a part they do not overflow would otherwise list them)
        const uint32_t t_bits = sp.room ? sp.t_bits : 0xffffffffu, max_cols = sp.room ? sp.max_cols : 0u;
        unsigned long long* bits = (sp.room && sp.max_cols) ? ctx->c_bits.as<unsigned long long>() : (unsigned long long*)nullptr;
        if (want_multi) { const int rm = ctx->c_multi.reserve((size_t)n_parts * kPartCap / 8 + 64); if (rm) return rm; }
        unsigned long long* multi = want_multi ? ctx->c_multi.as<unsigned long long>() : (unsigned long long*)nullptr;
        if (has_hi) {
            if (!ctx->attr_group_hi_set) {
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_group<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->attr_group_hi_set = true;
            }
            hipLaunchKernelGGL(k_parts_group<true>, dim3(n_parts), dim3(kGroupThreads), lds, ctx->stream, ctx->c_recs.as<uint64_t>(),
                               ctx->c_part_cnt.as<uint32_t>(), ctx->c_matrix.as<uint16_t>(), ctx->c_lref.as<uint32_t>(), flags, t_bits, max_cols, bits, n, multi);
        } else {
            if (!ctx->attr_group_set) {
                SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_group<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->attr_group_set = true;
            }
            hipLaunchKernelGGL(k_parts_group<false>, dim3(n_parts), dim3(kGroupThreads), lds, ctx->stream, ctx->c_recs.as<uint64_t>(),
                               ctx->c_part_cnt.as<uint32_t>(), ctx->c_matrix.as<uint16_t>(), ctx->c_lref.as<uint32_t>(), flags, t_bits, max_cols, bits, n, multi);
        }
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    J->spill_parts = [=](uint32_t n_parts, const SpillPlan& sp, int phase, uint32_t classes, uint32_t cls) -> int {
        const uint64_t slots = 1ull << sp.log2cap;
        int r2;
        if (phase == 0) {
            if ((r2 = ctx->c_table.reserve((size_t)slots * 4)) || (r2 = ctx->c_owner.reserve((size_t)slots * 4)) || (r2 = ctx->c_rowid.reserve((size_t)slots * 4))) return r2;
            SPSP_HIP(hipMemsetAsync(ctx->c_table.p, 0, (size_t)slots * 4, ctx->stream));
            SPSP_HIP(hipMemsetAsync(ctx->c_owner.p, 0, (size_t)slots * 4, ctx->stream));
            if (sp.max_cols) {
                const size_t bytes = (size_t)((sp.max_cols + 63) / 64) * n * 8;
                if ((r2 = ctx->c_bits.reserve(bytes))) return r2;
                SPSP_HIP(hipMemsetAsync(ctx->c_bits.p, 0, bytes, ctx->stream));
            }
            return SPSP_OK;
        }
        const dim3 grid((uint32_t)((S - e_own + kSpillThreads - 1) / kSpillThreads));
        uint32_t *tbl = ctx->c_table.as<uint32_t>(), *cnt = ctx->c_owner.as<uint32_t>(), *off = ctx->c_rowid.as<uint32_t>();
        uint32_t *where = ctx->c_where.as<uint32_t>(), *rank_of = ctx->c_row.as<uint32_t>();
        const uint32_t* part_cnt = ctx->c_part_cnt.as<uint32_t>();
        const uint32_t room = (uint32_t)sp.room;
        if (has_hi) hipLaunchKernelGGL(k_spill_insert<true>, grid, dim3(kSpillThreads), 0, ctx->stream, K, sk, n, S, e_own, row_first, n_parts, part_cnt, tbl, sp.log2cap, cnt, where, rank_of, room, flags, classes, cls, sub_sk);
        else hipLaunchKernelGGL(k_spill_insert<false>, grid, dim3(kSpillThreads), 0, ctx->stream, K, sk, n, S, e_own, row_first, n_parts, part_cnt, tbl, sp.log2cap, cnt, where, rank_of, room, flags, classes, cls, sub_sk);
        SPSP_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_spill_ranges, dim3((uint32_t)((slots + (uint64_t)kRowThreads * kRowSlots - 1) / ((uint64_t)kRowThreads * kRowSlots))), dim3(kRowThreads), 0, ctx->stream,
                           (const uint32_t*)cnt, slots, off, ctx->c_matrix.as<uint16_t>(), sp.ids_base, sp.ids_room, ctx->c_lref.as<uint32_t>() + sp.lref_base, sp.t_bits, sp.max_cols, room, flags);
        SPSP_HIP(hipGetLastError());
        unsigned long long* bits = sp.max_cols ? ctx->c_bits.as<unsigned long long>() : (unsigned long long*)nullptr;
        if (has_hi) hipLaunchKernelGGL(k_spill_fill<true>, grid, dim3(kSpillThreads), 0, ctx->stream, K, sk, n, S, e_own, row_first, n_parts, part_cnt, cnt, (const uint32_t*)off,
                                       ctx->c_matrix.as<uint16_t>(), sp.ids_base, sp.lref_base, bits, where, (const uint32_t*)rank_of, room, (const uint32_t*)flags, classes, cls, sub_sk);
        else hipLaunchKernelGGL(k_spill_fill<false>, grid, dim3(kSpillThreads), 0, ctx->stream, K, sk, n, S, e_own, row_first, n_parts, part_cnt, cnt, (const uint32_t*)off,
                                ctx->c_matrix.as<uint16_t>(), sp.ids_base, sp.lref_base, bits, where, (const uint32_t*)rank_of, room, (const uint32_t*)flags, classes, cls, sub_sk);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    if (!has_hi) J->group_small = [=](uint32_t n_parts) -> int {
        const size_t lds = (size_t)kSmallCap * (8 + 4) + (size_t)kSmallHl + (size_t)kSmallSlots * 4 + (size_t)kSmallN * kSmallN * 2 + 16;
        if (!ctx->attr_small_set) {
            SPSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_parts_group_small), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ctx->attr_small_set = true;
        }
        hipLaunchKernelGGL(k_parts_group_small, dim3(n_parts), dim3(kGroupThreads), lds, ctx->stream, ctx->c_recs.as<uint64_t>(),
                           ctx->c_part_cnt.as<uint32_t>(), n, d_inter, flags);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    return compare_job_begin(ctx, J);
}

// ===========================================================================
// Multi-GPU exchange, key-partitioned form (DESIGN.md "multi-GPU").
//
// Equal keys hash to the same rank, so a rank that holds every sketch's keys of ONE hash class can count
// that class's contribution to every pair; the classes are disjoint, hence inter = sum over ranks of the
// partial matrices.  Each rank therefore sends every key exactly once (all-to-all, O(own keys)) instead
// of receiving every rank's keys (all-gather, O(all keys)), and the table, the colour matrix and the row
// sums all shrink by the number of ranks.
//
// Wire format = one fixed-size SLOT per destination rank (fixed size so that the all-to-all needs no
// size negotiation):
//     u32 magic, u32 n_sketches, u32 n_keys (> slot_cap signals overflow; only slot_cap are stored), u32 words
//     u32 cnt[n_sketches]          keys per local sketch in this slot (padded to an even count)
//     record[slot_cap]             words x u64: kmer_lo, (kmer_hi if k > 32), minimizer | local sketch << 32
// Records are grouped by sketch, sketches in order, keys of a sketch in their original (sorted) order.
// (kSlotMagic, kMaxParts, slot_rec_off / slot_words / slot_bytes: spsp_internal.h -- the receiver is spsp_multi.hip)
constexpr int kPartThreads = 1024;

__device__ __forceinline__ uint32_t part_of(uint64_t lo, uint32_t mn, uint64_t hi, bool has_hi, uint32_t parts) {
    uint64_t h = mix64(lo ^ 0xD6E8FEB86659FD93ULL);
    h = mix64(h + (uint64_t)mn * 0xC2B2AE3D27D4EB4FULL);
    if (has_hi) h = mix64(h ^ hi);
    return (uint32_t)(((h >> 32) * parts) >> 32);
}

// sender 1/3: keys of sketch j per destination
__global__ __launch_bounds__(kPartThreads) void k_part_count(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n,
                                                            uint32_t parts, uint32_t* __restrict__ cnt /* [parts][n] */) {
    __shared__ uint32_t hist[kMaxKeyParts];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    if (t < kMaxParts) hist[t] = 0;
    __syncthreads();
    for (uint64_t e = sk_off[j] + t; e < sk_off[j + 1]; e += kPartThreads)
        atomicAdd(&hist[part_of(K.lo[e], K.mn[e], K.hi ? K.hi[e] : 0, K.hi != nullptr, parts)], 1u);
    __syncthreads();
    if (t < parts) cnt[(uint64_t)t * n + j] = hist[t];
}

// sender 2/3: per destination, where each sketch's run starts; slot header + counts
__global__ __launch_bounds__(kPartThreads) void k_part_offsets(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t words,
                                                              uint8_t* __restrict__ slots, uint64_t slot_sz,
                                                              uint32_t* __restrict__ off /* [parts][n] */) {
    __shared__ uint32_t wave_sum[kPartThreads / 64];
    __shared__ uint32_t s_carry;
    const uint32_t p = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    uint32_t* hdr = reinterpret_cast<uint32_t*>(slots + (uint64_t)p * slot_sz);
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kPartThreads) {
        const uint32_t j = base + t;
        const uint32_t v = j < n ? cnt[(uint64_t)p * n + j] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) wave_sum[wid] = x;
        __syncthreads();
        uint32_t pre = s_carry, all = 0;
        for (uint32_t w = 0; w < kPartThreads / 64; ++w) { if (w < wid) pre += wave_sum[w]; all += wave_sum[w]; }
        if (j < n) { off[(uint64_t)p * n + j] = pre + x - v; hdr[4 + j] = v; }
        __syncthreads();
        if (t == 0) s_carry += all;
        __syncthreads();
    }
    if (t == 0) {
        hdr[0] = kSlotMagic; hdr[1] = n; hdr[2] = s_carry; hdr[3] = words;
        if (n & 1u) hdr[4 + n] = 0;
    }
}

// sender 3/3: stable scatter of sketch j's keys into the slots
__global__ __launch_bounds__(kPartThreads) void k_part_scatter(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n,
                                                              uint32_t parts, const uint32_t* __restrict__ off,
                                                              uint8_t* __restrict__ slots, uint64_t slot_sz, uint32_t cap,
                                                              uint32_t words) {
    __shared__ uint32_t cursor[kMaxParts];
    __shared__ uint32_t wave_cnt[kPartThreads / 64][kMaxParts];
    const uint32_t j = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    if (t < kMaxParts) cursor[t] = t < parts ? off[(uint64_t)t * n + j] : 0u;
    const uint64_t e0 = sk_off[j], e1 = sk_off[j + 1];
    const uint64_t rec0 = slot_rec_off(n);
    for (uint64_t base = e0; base < e1; base += kPartThreads) {
        for (uint32_t x = t; x < (kPartThreads / 64) * kMaxParts; x += kPartThreads) (&wave_cnt[0][0])[x] = 0;
        __syncthreads();
        const uint64_t e = base + t;
        const bool live = e < e1;
        uint64_t lo = 0, hi = 0;
        uint32_t mn = 0, p = 0xffffffffu;
        if (live) {
            lo = K.lo[e]; mn = K.mn[e]; hi = K.hi ? K.hi[e] : 0;
            p = part_of(lo, mn, hi, K.hi != nullptr, parts);
        }
        // rank among the wave's earlier lanes with the same destination (keeps the original order)
        uint32_t rank = 0;
        uint64_t todo = __ballot(live);
        while (todo) {
            const uint32_t lead = (uint32_t)__ffsll((unsigned long long)todo) - 1;
            const uint32_t pl = __shfl(p, lead);
            const uint64_t same = __ballot(live && p == pl);
            if (p == pl) rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            if (lane == lead) wave_cnt[wid][pl] = (uint32_t)__popcll(same);
            todo &= ~same;
        }
        __syncthreads();
        if (live) {
            uint32_t at = cursor[p] + rank;
            for (uint32_t w = 0; w < wid; ++w) at += wave_cnt[w][p];
            if (at < cap) {
                uint64_t* rec = reinterpret_cast<uint64_t*>(slots + (uint64_t)p * slot_sz + rec0) + (uint64_t)at * words;
                rec[0] = lo;
                if (words == 3) rec[1] = hi;
                rec[words - 1] = (uint64_t)mn | ((uint64_t)j << 32);
            }
        }
        __syncthreads();
        if (t < parts) {
            uint32_t add = 0;
            for (uint32_t w = 0; w < kPartThreads / 64; ++w) add += wave_cnt[w][t];
            cursor[t] += add;
        }
        __syncthreads();
    }
}

int partition_keys_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo, const uint64_t* d_hi,
                        const uint64_t* h_sk_off, uint32_t n, uint32_t parts, uint32_t cap, uint8_t* d_slots) {
    if (n == 0 || n > 65535) { set_error("1..65535 sketches per rank"); return SPSP_ERR_ARG; }
    if (parts == 0 || parts > kMaxParts) { set_error("1..%u destinations", kMaxParts); return SPSP_ERR_ARG; }
    if (cap == 0) { set_error("slot_cap must be positive"); return SPSP_ERR_ARG; }
    for (uint32_t i = 0; i < n; ++i)
        if (h_sk_off[i + 1] < h_sk_off[i]) { set_error("sketch offsets must be non-decreasing (sketch %u)", i); return SPSP_ERR_ARG; }
    if (k > 32 && !d_hi && h_sk_off[n] > 0) { set_error("k=%u needs kmer_hi", k); return SPSP_ERR_ARG; }   // (no keys: nothing to read)
    if (((uintptr_t)d_slots & 7u) != 0) { set_error("d_slots must be 8-byte aligned"); return SPSP_ERR_ARG; }
    if (h_sk_off[n] > 0xfffffff0ull) { set_error("too many sketch k-mers for one call"); return SPSP_ERR_OVERFLOW; }
    int rc;
    if ((rc = ctx->c_skoff.reserve((size_t)(n + 1) * 8))) return rc;
    if ((rc = ctx->x_cnt.reserve((size_t)parts * n * 4))) return rc;
    if ((rc = ctx->x_off.reserve((size_t)parts * n * 4))) return rc;
    SPSP_HIP(hipMemcpyAsync(ctx->c_skoff.p, h_sk_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    Keys K{d_min, d_lo, (k > 32) ? d_hi : nullptr, ~0ull};
    const uint64_t* sk = ctx->c_skoff.as<uint64_t>();
    const uint32_t words = slot_words(k);
    const uint64_t sz = slot_bytes(n, cap, k);
    hipLaunchKernelGGL(k_part_count, dim3(n), dim3(kPartThreads), 0, ctx->stream, K, sk, n, parts, ctx->x_cnt.as<uint32_t>());
    SPSP_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_part_offsets, dim3(parts), dim3(kPartThreads), 0, ctx->stream, ctx->x_cnt.as<uint32_t>(), n, words,
                       d_slots, sz, ctx->x_off.as<uint32_t>());
    SPSP_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_part_scatter, dim3(n), dim3(kPartThreads), 0, ctx->stream, K, sk, n, parts, ctx->x_off.as<uint32_t>(),
                       d_slots, sz, cap, words);
    SPSP_HIP(hipGetLastError());
    return SPSP_OK;
}

int compare_device_begin_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                              const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit,
                              uint32_t row_first, uint32_t row_stride, uint32_t* d_inter) {
    if (ctx->compare_job) { set_error("a comparison is already pending on this context: call spsp_compare_end first"); return SPSP_ERR_ARG; }
    int rc = ctx->ev_begin(kEvCompare);
    if (rc) return rc;
    rc = compare_device_begin_inner(ctx, k, d_min, d_lo, d_hi, h_sk_off, n, row_limit, row_first, row_stride, d_inter);
    const bool deferred = rc == 0 && ctx->compare_job && !ctx->compare_job->speculative;   // closed in compare_job_end
    const int rc2 = deferred ? SPSP_OK : ctx->ev_end(kEvCompare);
    if (rc < 0) return rc;
    if (rc == 1) {                       // nothing to compare: leave an empty job so that begin/end stay paired
        CompareJob* J = new CompareJob;
        J->speculative = true;
        J->P = ComparePlan{};
        ctx->compare_job = J;
        J->attempt = -1;
    }
    return rc2;
}

int compare_end_impl(spsp_ctx* ctx) {
    if (ctx->compare_job && ctx->compare_job->attempt == -1) { compare_job_drop(ctx); return SPSP_OK; }
    return compare_job_end(ctx);
}

int compare_slots_impl(spsp_ctx* ctx, uint32_t k, const uint8_t* d_slots, uint32_t parts, uint32_t n, uint32_t cap,
                       uint32_t* d_inter) {
    int rc = compare_slots_begin_impl(ctx, k, d_slots, parts, n, cap, d_inter);
    if (rc) return rc;
    rc = compare_end_impl(ctx);
    return rc ? rc : slots_bad_record(ctx);
}

int compare_device_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                        const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit, uint32_t row_first,
                        uint32_t row_stride, uint32_t* d_inter) {
    int rc = compare_device_begin_impl(ctx, k, d_min, d_lo, d_hi, h_sk_off, n, row_limit, row_first, row_stride, d_inter);
    if (rc) return rc;
    return compare_end_impl(ctx);
}

}  // namespace spsp

using namespace spsp;

extern "C" {

int spsp_compare_device(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo,
                        const void* d_kmer_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t n_query, uint32_t row_first,
                        uint32_t row_stride, void* d_inter) {
    if (!ctx || !h_sk_off || !d_inter) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return compare_device_impl(ctx, k, (const uint32_t*)d_minimizer, (const uint64_t*)d_kmer_lo,
                               (const uint64_t*)d_kmer_hi, h_sk_off, n, n_query, row_first, row_stride, (uint32_t*)d_inter);
}

uint64_t spsp_slot_bytes(uint32_t n, uint32_t slot_cap, uint32_t k) { return slot_bytes(n, slot_cap, k); }

int spsp_partition_keys_device(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo,
                               const void* d_kmer_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t parts,
                               uint32_t slot_cap, void* d_slots) {
    if (!ctx || !h_sk_off || !d_slots) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return partition_keys_impl(ctx, k, (const uint32_t*)d_minimizer, (const uint64_t*)d_kmer_lo, (const uint64_t*)d_kmer_hi,
                               h_sk_off, n, parts, slot_cap, (uint8_t*)d_slots);
}

int spsp_compare_slots_device(spsp_ctx* ctx, uint32_t k, const void* d_slots, uint32_t parts, uint32_t n,
                              uint32_t slot_cap, void* d_inter) {
    if (!ctx || !d_slots || !d_inter) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return compare_slots_impl(ctx, k, (const uint8_t*)d_slots, parts, n, slot_cap, (uint32_t*)d_inter);
}

int spsp_compare_device_begin(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo,
                              const void* d_kmer_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t n_query,
                              uint32_t row_first, uint32_t row_stride, void* d_inter) {
    if (!ctx || !h_sk_off || !d_inter) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return compare_device_begin_impl(ctx, k, (const uint32_t*)d_minimizer, (const uint64_t*)d_kmer_lo,
                                     (const uint64_t*)d_kmer_hi, h_sk_off, n, n_query, row_first, row_stride, (uint32_t*)d_inter);
}

int spsp_compare_slots_device_begin(spsp_ctx* ctx, uint32_t k, const void* d_slots, uint32_t parts, uint32_t n,
                                    uint32_t slot_cap, void* d_inter) {
    if (!ctx || !d_slots || !d_inter) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return compare_slots_begin_impl(ctx, k, (const uint8_t*)d_slots, parts, n, slot_cap, (uint32_t*)d_inter);
}

int spsp_compare_end(spsp_ctx* ctx) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    const int rc = compare_end_impl(ctx);
    return rc ? rc : slots_bad_record(ctx);                   // (a comparison of exchange slots: the unpack's record check)
}

int spsp_compare(spsp_ctx* ctx, const spsp_sketch_view* sk, uint32_t n, uint32_t n_query, uint32_t* inter,
                 uint64_t* card) {
    if (!ctx || (n && (!sk || !inter || !card))) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (n == 0) return SPSP_OK;
    SPSP_HIP(hipSetDevice(ctx->device));
    std::vector<uint64_t> off(n + 1, 0);
    bool need_hi = false;
    for (uint32_t i = 0; i < n; ++i) {
        off[i + 1] = off[i] + sk[i].n;
        card[i] = sk[i].n;
        if (sk[i].n && (!sk[i].minimizer || !sk[i].kmer_lo)) { set_error("sketch %u has NULL key arrays", i); return SPSP_ERR_ARG; }
        if (sk[i].kmer_hi) need_hi = true;
    }
    for (uint32_t i = 0; i < n; ++i)
        if (need_hi && sk[i].n && !sk[i].kmer_hi) { set_error("kmer_hi must be given for all sketches or none"); return SPSP_ERR_ARG; }
    memset(inter, 0, sizeof(uint32_t) * (size_t)n * n);
    const uint64_t S = off[n];
    if (S == 0) return SPSP_OK;
    int rc;
    if ((rc = ctx->c_min.reserve((size_t)S * 4))) return rc;
    if ((rc = ctx->c_lo.reserve((size_t)S * 8))) return rc;
    if (need_hi && (rc = ctx->c_hi.reserve((size_t)S * 8))) return rc;
    if ((rc = ctx->c_inter.reserve((size_t)n * n * 4))) return rc;
    for (uint32_t i = 0; i < n; ++i) {
        if (!sk[i].n) continue;
        SPSP_HIP(hipMemcpyAsync(ctx->c_min.as<uint32_t>() + off[i], sk[i].minimizer, sk[i].n * 4, hipMemcpyHostToDevice, ctx->stream));
        SPSP_HIP(hipMemcpyAsync(ctx->c_lo.as<uint64_t>() + off[i], sk[i].kmer_lo, sk[i].n * 8, hipMemcpyHostToDevice, ctx->stream));
        if (need_hi) SPSP_HIP(hipMemcpyAsync(ctx->c_hi.as<uint64_t>() + off[i], sk[i].kmer_hi, sk[i].n * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    SPSP_HIP(hipMemsetAsync(ctx->c_inter.p, 0, (size_t)n * n * 4, ctx->stream));
    rc = compare_device_impl(ctx, need_hi ? 63 : 31, ctx->c_min.as<uint32_t>(), ctx->c_lo.as<uint64_t>(),
                             need_hi ? ctx->c_hi.as<uint64_t>() : nullptr, off.data(), n, n_query, 0, 1, ctx->c_inter.as<uint32_t>());
    if (rc) return rc;
    SPSP_HIP(hipMemcpyAsync(inter, ctx->c_inter.p, (size_t)n * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    return SPSP_OK;
}

}  // extern "C"
