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
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

struct Keys {
    const uint32_t* mn;
    const uint64_t* lo;
    const uint64_t* hi;  // may be null (k <= 32)
    uint64_t fp_mask;    // all ones; narrowed only by the collision-path test hook
};

__device__ __forceinline__ bool key_eq(const Keys& K, uint64_t a, uint64_t b) {
    if (K.lo[a] != K.lo[b] || K.mn[a] != K.mn[b]) return false;
    return K.hi ? K.hi[a] == K.hi[b] : true;
}
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

// flags[0]: input not strictly sorted inside a sketch; flags[1]: fingerprint collision
__global__ void k_insert(Keys K, const uint64_t* __restrict__ sk_off, uint32_t n, uint64_t S, uint32_t row_first,
                         uint32_t row_stride, uint32_t row_limit, uint64_t seed, uint64_t* __restrict__ table, uint32_t log2cap,
                         uint32_t* __restrict__ owner, uint32_t* __restrict__ slot, uint32_t* __restrict__ flags) {
    // grid.y = sketch, grid.x = 256-key chunk of it (no per-entry search for the owning sketch)
    const uint32_t j = blockIdx.y;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    if (e > sk_off[j] && !key_less(K, e - 1, e)) atomicOr(&flags[0], 1u);
    if (j % row_stride != row_first || j >= row_limit) return;   // not an owned (and printed) row
    const uint64_t fp = fingerprint(K, e, seed);
    const uint64_t mask = (1ull << log2cap) - 1;
    uint64_t pos = home_slot(fp, log2cap);
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long*)&table[pos], 0ull, (unsigned long long)fp);
        if (old == 0ull) { owner[pos] = (uint32_t)e; break; }   // the claiming entry is the slot's owner: one plain store
        if (old == fp) break;
        pos = (pos + 1) & mask;
    }
    slot[e] = (uint32_t)pos;
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
                       const uint32_t* __restrict__ owner, const uint32_t* __restrict__ rowid, uint32_t W,
                       unsigned long long* __restrict__ A, uint32_t* __restrict__ row_of_entry,
                       uint32_t* __restrict__ flags) {
    const uint32_t j = blockIdx.y;
    const uint64_t e = sk_off[j] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sk_off[j + 1]) return;
    const uint64_t fp = fingerprint(K, e, seed);
    const uint64_t mask = (1ull << log2cap) - 1;
    uint64_t pos = home_slot(fp, log2cap);
    for (;;) {
        const uint64_t v = table[pos];
        if (v == 0) return;  // key not held by any owned sketch: contributes to no owned row
        if (v == fp) {
            const bool own = j % row_stride == row_first && j < row_limit;
            if (key_eq(K, owner[pos], e)) {
                const uint32_t r = rowid[pos];
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
                                                           const uint64_t* __restrict__ sk_off, uint32_t n,
                                                           uint32_t row_first, uint32_t row_stride, uint32_t row_limit,
                                                           uint32_t* __restrict__ inter) {
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
    const uint64_t e0 = sk_off[i], e1 = sk_off[i + 1];
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
        if (col > i && col < n) inter[(uint64_t)i * n + col] = s_cnt[x];
    }
}

static int compare_device_inner(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                                const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit,
                                uint32_t row_first, uint32_t row_stride, uint32_t* d_inter) {
    if (n == 0) return SPSP_OK;
    if (n > 65535) { set_error("at most 65535 sketches (the reference's uint32 pair key, Comparator.h:26)"); return SPSP_ERR_ARG; }
    if (row_stride == 0 || row_first >= row_stride) { set_error("bad row partition %u/%u", row_first, row_stride); return SPSP_ERR_ARG; }
    if (k > 32 && !d_hi) { set_error("k=%u needs kmer_hi", k); return SPSP_ERR_ARG; }
    const uint64_t S = h_sk_off[n];
    if (S == 0) return SPSP_OK;
    if (S > 0xfffffff0ull) { set_error("too many sketch k-mers for one call"); return SPSP_ERR_OVERFLOW; }
    uint64_t S_own = 0;
    uint32_t n_own = 0;
    if (row_limit > n) row_limit = n;
    for (uint32_t i = row_first; i < row_limit; i += row_stride) { S_own += h_sk_off[i + 1] - h_sk_off[i]; ++n_own; }
    if (S_own == 0 || n_own == 0) return SPSP_OK;
    int rc;
    uint32_t log2cap = 10;
    while ((1ull << log2cap) < 2 * S_own) ++log2cap;
    const uint64_t cap = 1ull << log2cap;
    if ((rc = ctx->c_skoff.reserve((size_t)(n + 1) * 8))) return rc;
    if ((rc = ctx->c_table.reserve((size_t)cap * 8))) return rc;
    if ((rc = ctx->c_owner.reserve((size_t)cap * 4))) return rc;
    if ((rc = ctx->c_rowid.reserve((size_t)cap * 4))) return rc;
    if ((rc = ctx->c_slot.reserve((size_t)S * 4))) return rc;
    if ((rc = ctx->c_row.reserve((size_t)S * 4))) return rc;
    if ((rc = ctx->c_flags.reserve(64))) return rc;
    SPSP_HIP(hipMemcpyAsync(ctx->c_skoff.p, h_sk_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    Keys K{d_min, d_lo, (k > 32) ? d_hi : nullptr, ~0ull};
    // test hook: fingerprints of the first attempt cut to a few bits, so distinct keys collide and the retry runs
    static const char* dbg_fp = getenv("SPSP_DEBUG_FP_BITS");
    const uint64_t* sk = ctx->c_skoff.as<uint64_t>();
    uint32_t* flags = ctx->c_flags.as<uint32_t>();  // [0] unsorted, [1] collision, [2] n_rows
    uint64_t max_all = 0, max_own = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t c = h_sk_off[i + 1] - h_sk_off[i];
        max_all = std::max(max_all, c);
        if (i % row_stride == row_first && i < row_limit) max_own = std::max(max_own, c);
    }
    const dim3 grid_all((uint32_t)((max_all + 255) / 256), n);
    (void)max_own;
    const uint32_t sblocks = (uint32_t)((cap + (uint64_t)kRowThreads * kRowSlots - 1) / ((uint64_t)kRowThreads * kRowSlots));
    const uint32_t W = (n + 63) / 64;
    uint32_t lanes_per_key = 64;
    if (W < 64) { lanes_per_key = 1; while (lanes_per_key < W) lanes_per_key <<= 1; }
    // dictionary build: table and row ids
    auto front = [&](uint64_t seed) -> int {
        SPSP_HIP(hipMemsetAsync(ctx->c_table.p, 0, (size_t)cap * 8, ctx->stream));
        SPSP_HIP(hipMemsetAsync(flags, 0, 64, ctx->stream));
        hipLaunchKernelGGL(k_insert, grid_all, dim3(256), 0, ctx->stream, K, sk, n, S, row_first, row_stride, row_limit, seed,
                           ctx->c_table.as<uint64_t>(), log2cap, ctx->c_owner.as<uint32_t>(),
                           ctx->c_slot.as<uint32_t>(), flags);
        SPSP_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_assign_rows, dim3(sblocks), dim3(kRowThreads), 0, ctx->stream, ctx->c_table.as<uint64_t>(), cap,
                           ctx->c_rowid.as<uint32_t>(), flags + 2);
        SPSP_HIP(hipGetLastError());
        return SPSP_OK;
    };
    // colour matrix (room for `rows` rows) and the row sums
    auto back = [&](uint64_t seed, uint64_t rows) -> int {
        int r2;
        if ((r2 = ctx->c_matrix.reserve((size_t)rows * W * 8))) return r2;
        SPSP_HIP(hipMemsetAsync(ctx->c_matrix.p, 0, (size_t)rows * W * 8, ctx->stream));
        hipLaunchKernelGGL(k_fill, grid_all, dim3(256), 0, ctx->stream, K, sk, n, S, row_first, row_stride, row_limit, seed,
                           ctx->c_table.as<uint64_t>(), log2cap, ctx->c_owner.as<uint32_t>(), ctx->c_rowid.as<uint32_t>(),
                           W, ctx->c_matrix.as<unsigned long long>(), ctx->c_row.as<uint32_t>(), flags);
        SPSP_HIP(hipGetLastError());
        if ((r2 = ctx->ev_begin(kEvAccumulate))) return r2;
        hipLaunchKernelGGL(k_accumulate, dim3((W + 63) / 64, n_own), dim3(kAccThreads), 0, ctx->stream,
                           ctx->c_row.as<uint32_t>(), ctx->c_matrix.as<uint64_t>(), W, lanes_per_key, sk, n, row_first,
                           row_stride, row_limit, d_inter);
        SPSP_HIP(hipGetLastError());
        return ctx->ev_end(kEvAccumulate);
    };
    auto read_flags = [&](uint32_t* h_flags) -> int {
        SPSP_HIP(hipMemcpyAsync(h_flags, flags, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        SPSP_HIP(hipStreamSynchronize(ctx->stream));
        if (h_flags[0]) { set_error("sketch keys must be strictly increasing by (minimizer, kmer_hi, kmer_lo)"); return SPSP_ERR_ARG; }
        return SPSP_OK;
    };
    // A matrix with one row per OWNED KEY (an upper bound on the distinct keys) is cheap for small inputs:
    // then the whole pipeline is queued without waiting for the row count and checked once at the end.
    const bool speculative = (uint64_t)S_own * W * 8 <= (256ull << 20);
    uint64_t seed = 0x5350535053505350ULL;
    for (int attempt = 0;; ++attempt) {
        uint32_t h_flags[3];
        K.fp_mask = (dbg_fp && attempt == 0) ? ((1ull << atoi(dbg_fp)) - 1) : ~0ull;
        if ((rc = front(seed))) return rc;
        if (speculative) {
            if ((rc = back(seed, S_own))) return rc;
        } else {
            if ((rc = read_flags(h_flags))) return rc;          // the row count sizes the colour matrix
            if ((rc = back(seed, h_flags[2]))) return rc;
        }
        if ((rc = read_flags(h_flags))) return rc;              // collisions surface in k_fill
        if (!h_flags[1]) return SPSP_OK;
        if (attempt >= 4) { set_error("fingerprint collisions persisted over 5 seeds"); return SPSP_ERR_HIP; }
        seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;  // new fingerprints, try again
    }
}

int compare_device_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                        const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit, uint32_t row_first,
                        uint32_t row_stride, uint32_t* d_inter) {
    int rc = ctx->ev_begin(kEvCompare);
    if (rc) return rc;
    rc = compare_device_inner(ctx, k, d_min, d_lo, d_hi, h_sk_off, n, row_limit, row_first, row_stride, d_inter);
    const int rc2 = ctx->ev_end(kEvCompare);
    return rc ? rc : rc2;
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
