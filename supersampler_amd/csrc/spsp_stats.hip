// spsp_stats.hip -- the counters of Subsampler::print_stat that need EVERY super-k-mer of the input, selected
// or not: total_superkmer_number (SubSampler.cpp:430,452).  The product scan only ever looks at the selected
// minimizers, so this is a separate, optional pass (sub_sampler runs it for -v 1, the reference's default).
//
// The count is history dependent: the reference tracks a BELIEVED position of the current minimizer
// (regular_minimizer_pos, SubSampler.cpp:81-169, with its tie rules) and every time that position leaves the
// window it rescans and cuts a super-k-mer (`dump`, :391-398) -- also when the rescan finds the same m-mer
// again.  What is NOT history dependent: the tracked minimizer VALUE is always the true minimum of the window,
// so the event "the entering m-mer beats the minimum" (:374-388) happens at the same iterations whatever the
// believed position was, and it resets the whole state.  One lane therefore replays the literal state machine
// over a chunk of iterations from a fresh rescan a little in front of the chunk: once it has met such an event
// before its chunk starts, everything it counts is exact.  Chunks that meet none (period-w repeats,
// homopolymers) are counted again in a second pass, serially from the last exact point.
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

constexpr int kStatChunk = 512;     // iterations (k-mers) per lane
constexpr int kStatThreads = 128;

struct StatMachine {
    const uint8_t* rec;   // bases of the record (ASCII) ...
    const uint32_t* pk;   // ... or the whole 2-bit buffer (16 bases per dword, first base in bits 31:30) and
    uint64_t base;        //     the record's first base in it
    uint64_t len;
    uint32_t k, m, mask;
    // state
    uint64_t hash_min, position_min;
    uint32_t minimizer, old_minimizer, min_seq, min_rc;

    __device__ __forceinline__ uint32_t code(uint64_t at) const {
        if (pk) { const uint64_t q = base + at; return (pk[q >> 4] >> (30u - 2u * (uint32_t)(q & 15u))) & 3u; }
        return ((uint32_t)rec[at] >> 1) & 3u;
    }
    __device__ __forceinline__ void bind(const uint8_t* bases, bool packed, uint64_t first_base) {
        if (packed) { pk = reinterpret_cast<const uint32_t*>(bases); base = first_base; rec = nullptr; }
        else { pk = nullptr; base = 0; rec = bases + first_base; }
    }

    // regular_minimizer_pos (SubSampler.cpp:81-169) of the k-mer starting at `ks`, literally, right to left
    __device__ void rescan(uint64_t ks, uint64_t* position) {
        const uint32_t km = k - m;
        uint32_t f = 0;
        for (uint32_t j = 0; j < m; ++j) f = (f << 2) | code(ks + km + j);
        uint32_t r = rc_mmer32(f, m);
        uint32_t mini = f < r ? f : r;
        bool is_rev = mini != f;
        uint64_t pos = is_rev ? 0 : km;                                 // :88-93 (reverse => position 0, sic)
        uint64_t hash_mini = xxh64_u64(mini);
        for (uint32_t i = 1; i <= km; ++i) {
            const uint32_t off = km - i;
            const uint32_t b = code(ks + off);
            f = (f >> 2) | (b << (2 * m - 2));
            r = ((r << 2) | (b ^ 2u)) & mask;
            const uint32_t canon = f < r ? f : r;
            const bool local_rev = canon != f;
            const uint64_t h = xxh64_u64(canon);
            if (hash_mini > h) {                                         // :117-129
                pos = off; mini = canon; is_rev = local_rev; hash_mini = h;
            } else if (canon == mini && local_rev == is_rev) {           // :132-166
                if (is_rev && pos > i) pos = i;                          // :151-157 (sic)
                if (!is_rev && pos > off) pos = off;                     // :158-164
            }
        }
        minimizer = mini; hash_min = hash_mini; *position = pos;
    }
    // state in front of iteration i0 as a rescan of k-mer i0 leaves it (exact for i0 = 0: SubSampler.cpp:359-365)
    __device__ void start(uint64_t i0) {
        uint64_t pos;
        rescan(i0, &pos);
        position_min = pos + i0;
        old_minimizer = minimizer;
        const uint32_t km = k - m;
        uint32_t f = 0;
        for (uint32_t j = 0; j < m; ++j) f = (f << 2) | code(i0 + km + j);
        min_seq = f; min_rc = rc_mmer32(f, m);
    }
    // one iteration of the loop SubSampler.cpp:367-440; returns 1 when it cuts a super-k-mer; *reset = the
    // entering m-mer became the minimizer (the state no longer depends on what it was before)
    __device__ __forceinline__ uint32_t step(uint64_t i, bool* reset) {
        const uint32_t b = code(i + k);
        min_seq = ((min_seq << 2) | b) & mask;
        min_rc = (min_rc >> 2) | ((b ^ 2u) << (2 * m - 2));
        const uint32_t canon = min_seq < min_rc ? min_seq : min_rc;
        const uint64_t h = xxh64_u64(canon);
        bool dump = false;
        *reset = false;
        if (h < hash_min) {                                              // :374-388
            minimizer = canon; hash_min = h; position_min = i + k - m + 1;
            *reset = true;
        } else if (i >= position_min) {                                  // :391-398
            uint64_t pos;
            rescan(i + 1, &pos);
            position_min = pos + i + 1;
            dump = true;
        }
        if (old_minimizer != minimizer || dump) { old_minimizer = minimizer; return 1u; }   // :401-435
        return 0u;
    }
};

// the file of record r: the last f with file_rec[f] <= r
__device__ __forceinline__ uint32_t stat_file_of(const uint32_t* __restrict__ file_rec, uint32_t n_files, uint32_t r) {
    uint32_t a = 0, z = n_files;
    while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if (file_rec[mid] <= r) a = mid; else z = mid; }
    return a;
}

// chunk c covers the global k-mer iterations [c * kStatChunk, (c + 1) * kStatChunk) of the concatenated records:
// iteration g of record r = its k-mer g - rec_off[r] (only iterations i with i + k < len run the loop; the tail
// super-k-mer of every record with >= 1 k-mer is one more, added by whoever handles its k-mer 0)
__global__ __launch_bounds__(kStatThreads) void k_stat_count(const uint8_t* __restrict__ bases, bool packed, uint64_t base0, const uint64_t* __restrict__ rec_off,
                                                            uint32_t n_rec, uint32_t k, uint32_t m, uint64_t n_chunks,
                                                            uint32_t lookback, uint32_t* __restrict__ chunk_count,
                                                            uint8_t* __restrict__ chunk_open, unsigned long long* __restrict__ total,
                                                            const uint32_t* __restrict__ file_rec, uint32_t n_files) {
    // file_rec (optional): first record of each of n_files files -- the records of SEVERAL files in one launch, total[f] per file
    // (the file pipeline's batch: a launch per 5 Mbp file filled a third of the chip and was 7 ms of a 9 ms batch)
    const uint64_t c = (uint64_t)blockIdx.x * kStatThreads + threadIdx.x;
    unsigned long long mine = 0;
    uint32_t my_file = 0;
    if (c < n_chunks) {
        const uint64_t g0 = c * kStatChunk, g1 = g0 + kStatChunk;
        uint32_t r = 0, hi = n_rec;                          // record holding position g0
        while (hi - r > 1) { const uint32_t mid = (r + hi) >> 1; if (rec_off[mid] <= g0) r = mid; else hi = mid; }
        uint32_t first_count = 0;
        bool first_open = false, first = true;
        for (; r < n_rec && rec_off[r] < g1; ++r) {
            const uint64_t r0 = rec_off[r], len = rec_off[r + 1] - r0;
            if (len < k) { first = false; continue; }
            const uint64_t n_iter = len - k;                  // loop iterations 0 .. n_iter - 1
            const uint64_t a = g0 > r0 ? g0 - r0 : 0, b = g1 - r0 < n_iter ? g1 - r0 : n_iter;
            uint32_t cnt = 0;
            if (a == 0) ++cnt;                                // the record's tail super-k-mer (:441-454)
            bool open = false;
            if (a < b) {
                StatMachine M;
                M.bind(bases, packed, base0 + r0); M.len = len; M.k = k; M.m = m; M.mask = (1u << (2 * m)) - 1u;
                const uint64_t i0 = a > lookback ? a - lookback : 0;
                M.start(i0);
                bool exact = i0 == 0, reset;
                for (uint64_t i = i0; i < a; ++i) { M.step(i, &reset); exact |= reset; }
                open = !exact;
                for (uint64_t i = a; i < b; ++i) cnt += M.step(i, &reset);
            }
            if (first && a > 0) { first_count = cnt; first_open = open; }   // only a chunk's first piece can start inexact
            if (file_rec) {
                const uint32_t f = stat_file_of(file_rec, n_files, r);
                if (f != my_file) { if (mine) atomicAdd(&total[my_file], mine); mine = 0; my_file = f; }
            }
            mine += cnt;
            first = false;
        }
        chunk_count[c] = first_count;
        chunk_open[c] = first_open ? 1 : 0;
    }
    // one atomic per wave (several files: per wave when its lanes ended in the same file -- nearly always)
    const uint32_t f0 = __shfl(my_file, 0);
    if (file_rec && __any(my_file != f0)) { if (mine) atomicAdd(&total[my_file], mine); return; }
#pragma unroll
    for (int d = 32; d; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&total[file_rec ? f0 : 0u], mine);
}

// second pass: every RUN of consecutive open chunks (inside one record) is replayed by one lane from the exact
// state in front of the chunk before the run; the provisional counts are replaced
__global__ __launch_bounds__(kStatThreads) void k_stat_fix(const uint8_t* __restrict__ bases, bool packed, uint64_t base0, const uint64_t* __restrict__ rec_off,
                                                          uint32_t n_rec, uint32_t k, uint32_t m, uint64_t n_chunks,
                                                          uint32_t lookback, const uint32_t* __restrict__ chunk_count,
                                                          const uint8_t* __restrict__ chunk_open, unsigned long long* __restrict__ total,
                                                          const uint32_t* __restrict__ file_rec, uint32_t n_files) {
    const uint64_t c = (uint64_t)blockIdx.x * kStatThreads + threadIdx.x;
    if (c >= n_chunks || !chunk_open[c]) return;
    const uint64_t g0 = c * kStatChunk;
    uint32_t r = 0, hi = n_rec;
    while (hi - r > 1) { const uint32_t mid = (r + hi) >> 1; if (rec_off[mid] <= g0) r = mid; else hi = mid; }
    const uint64_t r0 = rec_off[r], len = rec_off[r + 1] - r0, n_iter = len - k;
    // head of a run? (the chunk in front is closed, or belongs to an earlier record)
    if (c > 0 && chunk_open[c - 1] && (c - 1) * kStatChunk > r0) return;
    // exact start: the chunk in front of the run was closed, so its lane's start point is good for us too
    const uint64_t a = g0 - r0;                              // > 0 for an open chunk
    const uint64_t prev_a = a > (uint64_t)kStatChunk ? a - kStatChunk : 0;
    const uint64_t i0 = prev_a > lookback ? prev_a - lookback : 0;
    StatMachine M;
    M.bind(bases, packed, base0 + r0); M.len = len; M.k = k; M.m = m; M.mask = (1u << (2 * m)) - 1u;
    M.start(i0);
    bool reset;
    for (uint64_t i = i0; i < a; ++i) M.step(i, &reset);
    long long delta = 0;
    for (uint64_t cc = c; cc < n_chunks && chunk_open[cc] && cc * kStatChunk < r0 + n_iter; ++cc) {
        const uint64_t ca = cc * kStatChunk - r0, cb = ca + kStatChunk < n_iter ? ca + kStatChunk : n_iter;
        uint32_t cnt = 0;
        for (uint64_t i = ca; i < cb; ++i) cnt += M.step(i, &reset);
        delta += (long long)cnt - (long long)chunk_count[cc];
    }
    if (delta) atomicAdd(&total[file_rec ? stat_file_of(file_rec, n_files, r) : 0u], (unsigned long long)delta);
}

// h_file_rec / n_files (optional): the records are those of n_files files, file f starting with record h_file_rec[f]: total[f]
// per file from ONE pair of launches (else n_files = 1 and `total` is one number)
int count_superkmers_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases, const uint64_t* d_rec_off,
                          uint32_t n_rec, uint64_t* total, bool packed, uint64_t base0, const uint32_t* h_file_rec, uint32_t n_files) {
    if (!h_file_rec) n_files = 1;
    for (uint32_t f = 0; f < n_files; ++f) total[f] = 0;
    int rc = check_params(p);
    if (rc) return rc;
    if (n_rec == 0 || n_bases < p->k || n_files == 0) return SPSP_OK;
    if (n_files > 1024) { set_error("too many files for one statistics pass"); return SPSP_ERR_ARG; }
    const uint64_t n_chunks = (n_bases + kStatChunk - 1) / kStatChunk;
    if (n_chunks > 0x7fffffffull * kStatThreads) { set_error("input too large for one call"); return SPSP_ERR_OVERFLOW; }
    if ((rc = ctx->st_count.reserve((size_t)n_chunks * 4))) return rc;
    if ((rc = ctx->st_open.reserve((size_t)n_chunks + 8))) return rc;
    if ((rc = ctx->st_total.reserve((size_t)n_files * 12 + 64))) return rc;
    unsigned long long* d_total = ctx->st_total.as<unsigned long long>();
    uint32_t* d_file_rec = reinterpret_cast<uint32_t*>(d_total + n_files);
    SPSP_HIP(hipMemsetAsync(d_total, 0, (size_t)n_files * 8, ctx->stream));
    if (h_file_rec) SPSP_HIP(hipMemcpyAsync(d_file_rec, h_file_rec, (size_t)n_files * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t* file_rec = h_file_rec ? d_file_rec : nullptr;
    const uint32_t w = p->k - p->m + 1;
    const uint32_t lookback = 8 * w < 64 ? 64 : 8 * w;
    const uint32_t blocks = (uint32_t)((n_chunks + kStatThreads - 1) / kStatThreads);
    hipLaunchKernelGGL(k_stat_count, dim3(blocks), dim3(kStatThreads), 0, ctx->stream, d_bases, packed, base0, d_rec_off, n_rec, p->k, p->m, n_chunks,
                       lookback, ctx->st_count.as<uint32_t>(), ctx->st_open.as<uint8_t>(), d_total, file_rec, n_files);
    SPSP_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_stat_fix, dim3(blocks), dim3(kStatThreads), 0, ctx->stream, d_bases, packed, base0, d_rec_off, n_rec, p->k, p->m, n_chunks,
                       lookback, ctx->st_count.as<uint32_t>(), ctx->st_open.as<uint8_t>(), d_total, file_rec, n_files);
    SPSP_HIP(hipGetLastError());
    std::vector<unsigned long long> back(n_files);
    SPSP_HIP(hipMemcpyAsync(back.data(), d_total, (size_t)n_files * 8, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    for (uint32_t f = 0; f < n_files; ++f) total[f] = back[f];
    return SPSP_OK;
}

}  // namespace spsp

using namespace spsp;

extern "C" int spsp_count_superkmers_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                                            const void* d_rec_off, uint32_t n_rec, uint64_t* total_superkmers) {
    if (!ctx || !p || !total_superkmers) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return count_superkmers_impl(ctx, p, (const uint8_t*)d_bases, n_bases, (const uint64_t*)d_rec_off, n_rec, total_superkmers);
}
