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
#include <algorithm>
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
    uint32_t last_rev;    // strand of the minimizer the last rescan found

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
        minimizer = mini; hash_min = hash_mini; *position = pos; last_rev = is_rev ? 1u : 0u;
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

// ---------------------------------------------------------------------------------------------------------------
// The same count by SEGMENTS (round 5).  One lane replaying the machine position by position hashes every m-mer again at every
// rescan, and with 64 lanes at different places of their chunks some lane of the wave is rescanning at nearly every step:
// a 30 Mbp batch took 8 ms whatever the chunk size (the largest part of a `sub_sampler` process's GPU stage).  What the count
// needs, taken apart:
//   * (canon, strand, hash) of every m-mer -- once, in parallel, into LDS;
//   * the iterations where the entering m-mer beats the window's minimum ("resets") -- data only: the tracked hash is the
//     true minimum of the window at all times (the believed position never lies behind the minimum's rightmost place), so
//     reset <=> hash[e] < min(hash[window]), a sliding minimum over k - m + 1 LDS words;  a reset always cuts (its m-mer
//     differs from the one before: the hash is smaller) and leaves a state that does not depend on what came before;
//   * between two resets the only cuts are the rescans (`dump`), at the iterations where the believed position leaves the
//     window: i = position_min, then position_min = i + 1 + (what regular_minimizer_pos finds in k-mer i + 1) -- a chain
//     of rescans, each 'k - m + 1 LDS reads', no stepping in between.
// A tile is kSegIter iterations (plus a halo the chains may run into); a lane owns the resets and record starts of its
// iterations.  A chain that leaves the halo (a long repeat: no reset for thousands of positions) is handed to k_seg_tail,
// which walks it with the literal machine.  Same totals as k_stat_count / k_stat_fix (SPSP_DEBUG_STATS=chunks keeps them).
constexpr int kSegIter = 2048, kSegHalo = 1024, kSegThreads = 512;
constexpr int kSegSpan = kSegIter + kSegHalo + 64;              // m-mers held: a window of the last halo iteration ends inside
constexpr uint32_t kSegOverCap = 1u << 20;
constexpr uint32_t kSegSlowMax = 16384;                    // iterations ONE lane walks alone behind its tile's halo (k_seg_scan) at the most; all of a call's together: slow_budget                     // chains handed on per call (beyond: the chunk kernels do the call)

// regular_minimizer_pos over LDS (see StatMachine::rescan): k-mer at tile place ks
__device__ __forceinline__ uint32_t seg_rescan(const uint64_t* __restrict__ s_h, const uint32_t* __restrict__ s_c, uint32_t ks, uint32_t km) {
    uint32_t c0 = s_c[ks + km];
    uint32_t mini = c0 & 0x7fffffffu;
    bool is_rev = (c0 >> 31) != 0;
    uint32_t pos = is_rev ? 0u : km;
    uint64_t hash_mini = s_h[ks + km];
    for (uint32_t i = 1; i <= km; ++i) {
        const uint32_t off = km - i;
        const uint32_t c = s_c[ks + off];
        const uint64_t h = s_h[ks + off];
        const uint32_t canon = c & 0x7fffffffu;
        const bool local_rev = (c >> 31) != 0;
        if (hash_mini > h) { pos = off; mini = canon; is_rev = local_rev; hash_mini = h; }
        else if (canon == mini && local_rev == is_rev) {
            if (is_rev && pos > i) pos = i;
            if (!is_rev && pos > off) pos = off;
        }
    }
    return pos;
}

// the tile's events -- iterations that are a record's start or a reset -- as an ordered list in LDS: with one lane per EVENT every
// lane of a round walks a chain (a lane per ITERATION had nine lanes in ten idle while some lane of the wave walked one: the walk
// was 8 rounds of the longest chain each).  All lanes call; returns the number of events.
__device__ __forceinline__ uint32_t seg_events(const uint32_t* __restrict__ s_d, const unsigned long long* __restrict__ s_flag, uint16_t* __restrict__ s_ev,
                                               uint32_t* __restrict__ s_wave) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t base = 0;
    for (uint32_t q = 0; q < (uint32_t)kSegIter; q += kSegThreads) {
        const uint32_t p = q + t;
        const bool has = (s_d[p] >> 31) != 0 || ((s_flag[p >> 6] >> (p & 63u)) & 1ull) != 0;
        const unsigned long long b = __ballot(has);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (uint32_t w = 0; w < kSegThreads / 64; ++w) { const uint32_t c = s_wave[w]; if (w < wave) pre += c; tot += c; }
        if (has) s_ev[base + pre + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)p;
        base += tot;
        __syncthreads();
    }
    return base;
}

__global__ __launch_bounds__(kSegThreads) void k_seg_count(const uint8_t* __restrict__ bases, bool packed, uint64_t n_bases, const uint64_t* __restrict__ rec_off,
                                                          uint32_t n_rec, uint32_t k, uint32_t m, unsigned long long* __restrict__ total,
                                                          const uint32_t* __restrict__ file_rec, uint32_t n_files,
                                                          unsigned long long* __restrict__ over, uint32_t* __restrict__ over_n) {
    __shared__ uint64_t s_h[kSegSpan];
    __shared__ uint32_t s_c[kSegSpan];                          // canon | strand << 31
    __shared__ uint32_t s_d[kSegSpan];                          // iterations from this place to the end of its record's loop (0: not an iteration) | record start << 31
    __shared__ unsigned long long s_flag[(kSegIter + kSegHalo) / 64];
    __shared__ uint16_t s_ev[kSegIter];
    __shared__ uint32_t s_wave[kSegThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, km = k - m, mask = (1u << (2 * m)) - 1u;
    const uint64_t T0 = (uint64_t)blockIdx.x * kSegIter;
    // ---- every m-mer of the span once: a lane takes consecutive places (one search for its record, the m-mer rolls)
    {
        constexpr uint32_t PER = (kSegSpan + kSegThreads - 1) / kSegThreads;
        const uint32_t p0 = t * PER, p1 = p0 + PER < (uint32_t)kSegSpan ? p0 + PER : (uint32_t)kSegSpan;
        uint64_t g = T0 + p0;
        uint32_t r = 0;
        if (p0 < p1 && g < n_bases) {
            uint32_t hi = n_rec;
            while (hi - r > 1) { const uint32_t mid = (r + hi) >> 1; if (rec_off[mid] <= g) r = mid; else hi = mid; }
        }
        uint64_t r0 = rec_off[r], r1 = rec_off[r + 1];
        StatMachine B;                                          // (for its base reader only)
        B.bind(bases, packed, 0);
        uint32_t f = 0, filled = 0;                             // the m-mer ending at the last base read; bases of it that are of this record
        for (uint32_t p = p0; p < p1; ++p, ++g) {
            uint64_t h = ~0ull; uint32_t c = 0, d = 0;
            if (g < n_bases) {
                while (g >= r1 && r + 1 < n_rec) { ++r; r0 = r1; r1 = rec_off[r + 1]; filled = 0; }
                const uint64_t len = r1 - r0;
                if (g + m <= r1) {                              // an m-mer of the record starts here
                    if (filled == 0) { for (uint32_t j = 0; j < m; ++j) f = ((f << 2) | B.code(g + j)) & mask; filled = 1; }
                    else f = ((f << 2) | B.code(g + m - 1)) & mask;
                    const uint32_t rc = rc_mmer32(f, m), canon = f < rc ? f : rc;
                    c = canon | (canon != f ? 0x80000000u : 0u);
                    h = xxh64_u64(canon);
                } else filled = 0;
                if (len >= k) {
                    const uint64_t n_iter = len - k, i = g - r0;
                    if (i < n_iter) d = (uint32_t)(n_iter - i < 0x7fffffffull ? n_iter - i : 0x7fffffffull);
                    if (i == 0) d |= 0x80000000u;
                }
            }
            s_h[p] = h; s_c[p] = c; s_d[p] = d;
        }
    }
    __syncthreads();
    // ---- resets: the entering m-mer (place p + km + 1) beats the window [p, p + km]
    for (uint32_t q = 0; q < (uint32_t)(kSegIter + kSegHalo); q += kSegThreads) {
        const uint32_t p = q + t;
        bool reset = false;
        if ((s_d[p] & 0x7fffffffu) != 0) {
            uint64_t mn = s_h[p];
            for (uint32_t j = 1; j <= km; ++j) { const uint64_t h = s_h[p + j]; mn = h < mn ? h : mn; }
            reset = s_h[p + km + 1] < mn;
        }
        const unsigned long long b = __ballot(reset);
        if (lane == 0) s_flag[p >> 6] = b;
    }
    __syncthreads();
    // ---- the chains
    unsigned long long mine = 0;
    uint32_t my_file = 0xffffffffu;
    constexpr uint32_t LIMIT = kSegIter + kSegHalo;
    auto next_reset = [&](uint32_t from, uint32_t end) -> uint32_t {     // first reset in [from, end), or end; end <= LIMIT
        uint32_t p = from;
        while (p < end) {
            unsigned long long w = s_flag[p >> 6] >> (p & 63u);
            if (w) { const uint32_t x = p + (uint32_t)__ffsll((long long)w) - 1u; return x < end ? x : end; }
            p = (p | 63u) + 1u;
        }
        return end;
    };
    const uint32_t n_ev = seg_events(s_d, s_flag, s_ev, s_wave);
    for (uint32_t j = t; j < n_ev; j += kSegThreads) {
        const uint32_t p = s_ev[j];
        const uint32_t d = s_d[p];
        const bool is_start = (d >> 31) != 0, is_reset = (s_flag[p >> 6] >> (p & 63u)) & 1ull;
        const uint32_t left = d & 0x7fffffffu;                  // iterations p .. p + left - 1 exist
        uint32_t cnt = 0;
#pragma unroll 1
        for (int kind = 0; kind < 2; ++kind) {                  // 0: the record's start (state of a rescan of k-mer 0), 1: a reset at p
            if (kind == 0 ? !is_start : !is_reset) continue;
            cnt += 1;                                           // the record's tail super-k-mer / the reset's cut
            uint32_t P = kind == 0 ? p + seg_rescan(s_h, s_c, p, km) : p + km + 1;
            const uint32_t from = kind == 0 ? p : p + 1;
            const uint64_t end64 = (uint64_t)p + left;
            const uint32_t end = end64 < LIMIT ? (uint32_t)end64 : LIMIT;
            uint32_t R = from < end ? next_reset(from, end) : end;
            bool unknown = R == end && end64 > LIMIT;           // no reset up to the halo's end: where the chain stops is not known here
            uint32_t add = 0;
            while (!unknown && P < R) {
                P = P + 1 + seg_rescan(s_h, s_c, P + 1, km);
                ++add;
            }
            if (unknown) {
                const uint32_t at = atomicAdd(over_n, 1u);
                if (at < kSegOverCap) over[at] = ((T0 + p) << 1) | (unsigned long long)kind;
                add = 0;                                        // (k_seg_tail counts the whole chain)
            }
            cnt += add;
        }
        if (file_rec) {
            // the record of place p: the search again (only lanes with an event come here)
            uint32_t r = 0, hi = n_rec;
            const uint64_t g = T0 + p;
            while (hi - r > 1) { const uint32_t mid = (r + hi) >> 1; if (rec_off[mid] <= g) r = mid; else hi = mid; }
            const uint32_t fl = stat_file_of(file_rec, n_files, r);
            if (fl != my_file) { if (mine) atomicAdd(&total[my_file], mine); mine = 0; my_file = fl; }
        }
        mine += cnt;
    }
    // one atomic per wave (several files: when the lanes that counted anything counted for the same file -- nearly always; an
    // atomic per lane on the handful of totals was 12 of this kernel's 20 ms)
    uint32_t f_wave = 0;
    if (file_rec) {
        const unsigned long long have = __ballot(mine != 0);
        if (!have) return;
        f_wave = __shfl(my_file, __ffsll((long long)have) - 1);
        if (__any(mine != 0 && my_file != f_wave)) { if (mine) atomicAdd(&total[my_file], mine); return; }
    }
#pragma unroll
    for (int dd = 32; dd; dd >>= 1) mine += __shfl_xor(mine, dd);
    if (lane == 0 && mine) atomicAdd(&total[f_wave], mine);
}

// ---------------------------------------------------------------------------------------------------------------
// The scan itself by segments, for thresholds that select (nearly) every m-mer (-s 1): what k_seg_count counts, emitted.
// Every event -- a record's start, a reset, a rescan -- opens a super-k-mer: it starts behind the event's iteration, carries
// the minimizer and strand the event leaves, and ends with the next event (SubSampler.cpp:401-438; the record's last one at
// :441-454 by the same formula).  A lane owns the events of its iterations and walks the reset's chain as above, so it knows
// every super-k-mer of the chain in full; the selected ones (hash of the minimizer <= threshold, :405) are counted per owning
// iteration, placed by a prefix over the tile and the tiles' totals (two launches: counts, then the same walk writing), and
// leave in genome order.  The product scan's dense + sparse passes take 245 ms per 500 Mbp at -s 1 (every position a 32-byte
// hit record); this form takes ~50.  A chain that leaves its tile's halo makes the whole call take the product scan (flag).
struct SegState { uint32_t mn; uint32_t rev; uint64_t h; uint32_t pos; };
__device__ __forceinline__ SegState seg_rescan_full(const uint64_t* __restrict__ s_h, const uint32_t* __restrict__ s_c, uint32_t ks, uint32_t km) {
    const uint32_t c0 = s_c[ks + km];
    SegState S;
    S.mn = c0 & 0x7fffffffu; S.rev = c0 >> 31; S.pos = S.rev ? 0u : km; S.h = s_h[ks + km];
    for (uint32_t i = 1; i <= km; ++i) {
        const uint32_t off = km - i;
        const uint32_t c = s_c[ks + off];
        const uint64_t h = s_h[ks + off];
        const uint32_t canon = c & 0x7fffffffu, lrev = c >> 31;
        if (S.h > h) { S.pos = off; S.mn = canon; S.rev = lrev; S.h = h; }
        else if (canon == S.mn && lrev == S.rev) {
            if (S.rev && S.pos > i) S.pos = i;
            if (!S.rev && S.pos > off) S.pos = off;
        }
    }
    return S;
}

// the same over a ring of the last 64 m-mers (index = m-mer place & 63): the lane that walks a chain behind its tile's halo
__device__ __forceinline__ SegState seg_rescan_ring(const uint64_t* __restrict__ r_h, const uint32_t* __restrict__ r_c, uint64_t ks, uint32_t km) {
    const uint32_t c0 = r_c[(ks + km) & 63u];
    SegState S;
    S.mn = c0 & 0x7fffffffu; S.rev = c0 >> 31; S.pos = S.rev ? 0u : km; S.h = r_h[(ks + km) & 63u];
    // eight m-mers' words are requested together, then judged in order: the lane is alone in its wave here, and a rescan of one
    // dependent LDS read after the other was most of an iteration's time
    for (uint32_t i0 = 1; i0 <= km; i0 += 8) {
        uint64_t hh[8];
        uint32_t cc[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u, idx = (uint32_t)((ks + km - (i <= km ? i : km)) & 63u);
            hh[u] = r_h[idx]; cc[u] = r_c[idx];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = i0 + u;
            if (i > km) break;
            const uint32_t off = km - i;
            const uint32_t canon = cc[u] & 0x7fffffffu, lrev = cc[u] >> 31;
            if (S.h > hh[u]) { S.pos = off; S.mn = canon; S.rev = lrev; S.h = hh[u]; }
            else if (canon == S.mn && lrev == S.rev) {
                if (S.rev && S.pos > i) S.pos = i;
                if (!S.rev && S.pos > off) S.pos = off;
            }
        }
    }
    return S;
}
constexpr uint32_t kSegRings = 2;                            // lanes of a workgroup that may walk behind the halo with a ring in LDS at a time

template <bool EMIT>
__global__ __launch_bounds__(kSegThreads) void k_seg_scan(const uint8_t* __restrict__ bases, bool packed, uint64_t n_bases, const uint64_t* __restrict__ rec_off,
                                                         uint32_t n_rec, uint32_t k, uint32_t m, uint64_t threshold,
                                                         uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_off,
                                                         spsp_superkmer* __restrict__ out, uint64_t out_cap, uint32_t* __restrict__ over_n, uint32_t slow_budget) {
    __shared__ uint64_t s_h[kSegSpan];
    __shared__ uint32_t s_c[kSegSpan];
    __shared__ uint32_t s_d[kSegSpan];
    __shared__ unsigned long long s_flag[(kSegIter + kSegHalo) / 64];
    __shared__ uint32_t s_cnt[kSegIter];                        // selected super-k-mers opened by event j; then their first place in the tile
    __shared__ uint64_t s_ring_h[kSegRings][64];
    __shared__ uint32_t s_ring_c[kSegRings][64];
    __shared__ uint32_t s_ring_used[kSegRings];
    __shared__ uint16_t s_ev[kSegIter];
    __shared__ uint32_t s_wave[kSegThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, km = k - m, mask = (1u << (2 * m)) - 1u;
    const uint64_t T0 = (uint64_t)blockIdx.x * kSegIter;
    {
        constexpr uint32_t PER = (kSegSpan + kSegThreads - 1) / kSegThreads;
        const uint32_t p0 = t * PER, p1 = p0 + PER < (uint32_t)kSegSpan ? p0 + PER : (uint32_t)kSegSpan;
        uint64_t g = T0 + p0;
        uint32_t r = 0;
        if (p0 < p1 && g < n_bases) {
            uint32_t hi = n_rec;
            while (hi - r > 1) { const uint32_t mid = (r + hi) >> 1; if (rec_off[mid] <= g) r = mid; else hi = mid; }
        }
        uint64_t r0 = rec_off[r], r1 = rec_off[r + 1];
        StatMachine B;
        B.bind(bases, packed, 0);
        uint32_t f = 0, filled = 0;
        for (uint32_t p = p0; p < p1; ++p, ++g) {
            uint64_t h = ~0ull; uint32_t c = 0, d = 0;
            if (g < n_bases) {
                while (g >= r1 && r + 1 < n_rec) { ++r; r0 = r1; r1 = rec_off[r + 1]; filled = 0; }
                const uint64_t len = r1 - r0;
                if (g + m <= r1) {
                    if (filled == 0) { for (uint32_t j = 0; j < m; ++j) f = ((f << 2) | B.code(g + j)) & mask; filled = 1; }
                    else f = ((f << 2) | B.code(g + m - 1)) & mask;
                    const uint32_t rc = rc_mmer32(f, m), canon = f < rc ? f : rc;
                    c = canon | (canon != f ? 0x80000000u : 0u);
                    h = xxh64_u64(canon);
                } else filled = 0;
                if (len >= k) {
                    const uint64_t n_iter = len - k, i = g - r0;
                    if (i < n_iter) d = (uint32_t)(n_iter - i < 0x7fffffffull ? n_iter - i : 0x7fffffffull);
                    if (i == 0) d |= 0x80000000u;
                }
            }
            s_h[p] = h; s_c[p] = c; s_d[p] = d;
        }
    }
    for (uint32_t x = t; x < (uint32_t)kSegIter; x += kSegThreads) s_cnt[x] = 0;
    if (t < kSegRings) s_ring_used[t] = 0;
    __syncthreads();
    for (uint32_t q = 0; q < (uint32_t)(kSegIter + kSegHalo); q += kSegThreads) {
        const uint32_t p = q + t;
        bool reset = false;
        if ((s_d[p] & 0x7fffffffu) != 0) {
            uint64_t mn = s_h[p];
            for (uint32_t j = 1; j <= km; ++j) { const uint64_t h = s_h[p + j]; mn = h < mn ? h : mn; }
            reset = s_h[p + km + 1] < mn;
        }
        const unsigned long long b = __ballot(reset);
        if (lane == 0) s_flag[p >> 6] = b;
    }
    __syncthreads();
    constexpr uint32_t LIMIT = kSegIter + kSegHalo;
    auto next_reset = [&](uint32_t from, uint32_t end) -> uint32_t {
        uint32_t p = from;
        while (p < end) {
            unsigned long long w = s_flag[p >> 6] >> (p & 63u);
            if (w) { const uint32_t x = p + (uint32_t)__ffsll((long long)w) - 1u; return x < end ? x : end; }
            p = (p | 63u) + 1u;
        }
        return end;
    };
    const uint32_t n_ev = seg_events(s_d, s_flag, s_ev, s_wave);
    // the walk, a lane per event: WRITE = false counts the selected super-k-mers of each event, WRITE = true puts them in their places
    auto walk = [&](bool write) {
        for (uint32_t j = t; j < n_ev; j += kSegThreads) {
            const uint32_t p = s_ev[j];
            const uint32_t d = s_d[p];
            const bool is_start = (d >> 31) != 0, is_reset = (s_flag[p >> 6] >> (p & 63u)) & 1ull;
            const uint32_t left = d & 0x7fffffffu;
            uint32_t cnt = 0;
            uint64_t at = 0, r0 = 0;
            uint32_t rec = 0;
            bool write_lookup_done = write;                        // (the record of place p: the writing pass looks it up for every event, the counting pass for chains that leave the halo)
            if (write) {
                uint32_t hi = n_rec;
                const uint64_t g = T0 + p;
                while (hi - rec > 1) { const uint32_t mid = (rec + hi) >> 1; if (rec_off[mid] <= g) rec = mid; else hi = mid; }
                r0 = rec_off[rec];
                at = (uint64_t)tile_off[blockIdx.x] + s_cnt[j];
            }
#pragma unroll 1
            for (int kind = 0; kind < 2; ++kind) {
                if (kind == 0 ? !is_start : !is_reset) continue;
                SegState S;
                uint32_t P, cur;
                if (kind == 0) { S = seg_rescan_full(s_h, s_c, p, km); P = p + S.pos; cur = p; }
                else { const uint32_t c = s_c[p + km + 1]; S.mn = c & 0x7fffffffu; S.rev = c >> 31; S.h = s_h[p + km + 1]; P = p + km + 1; cur = p + 1; }
                const uint64_t end64 = (uint64_t)p + left;
                const uint32_t end = end64 < LIMIT ? (uint32_t)end64 : LIMIT;
                const uint32_t R = cur < end ? next_reset(cur, end) : end;
                // the chain leaves the halo (no reset for 1 024 iterations and more: a homopolymer, a short-period repeat): its rescans
                // inside the halo as below, then this lane goes on alone, iteration by iteration over the bases themselves
                const bool unknown = R == end && end64 > LIMIT;
                const uint32_t stop = unknown ? LIMIT : R;
                if (unknown && !write_lookup_done) {
                    uint32_t hi = n_rec;
                    const uint64_t g = T0 + p;
                    rec = 0;
                    while (hi - rec > 1) { const uint32_t mid = (rec + hi) >> 1; if (rec_off[mid] <= g) rec = mid; else hi = mid; }
                    r0 = rec_off[rec];
                    write_lookup_done = true;
                }
                auto emit = [&](uint64_t start_rel, uint64_t len, uint32_t mn, uint32_t rev, uint64_t h) {
                    if (h > threshold) return;
                    if (write && at < out_cap) {
                        spsp_superkmer e;
                        e.rec = rec; e.minimizer = mn; e.start = start_rel; e.len = (uint32_t)len; e.rev = rev;
                        out[at] = e;
                    }
                    ++at; ++cnt;
                };
                for (;;) {
                    const bool more = P < stop;
                    if (!more && unknown) break;                         // (the open super-k-mer goes on behind the halo)
                    const uint32_t close = more ? P : R;                 // the iteration that cuts the open super-k-mer (or the record's end)
                    emit((T0 + cur) - r0, (uint64_t)close + k - cur, S.mn, S.rev, S.h);
                    if (!more) break;
                    S = seg_rescan_full(s_h, s_c, P + 1, km);
                    cur = P + 1;
                    P = P + 1 + S.pos;
                }
                if (unknown) {
                    const uint64_t rec_len = rec_off[rec + 1] - r0, n_iter = rec_len - k;
                    StatMachine M;
                    M.bind(bases, packed, r0); M.len = rec_len; M.k = k; M.m = m; M.mask = mask;
                    uint64_t i = (T0 + LIMIT) - r0;                      // the first iteration behind the halo
                    uint64_t open = (T0 + cur) - r0;
                    {
                        uint32_t f = 0;
                        for (uint32_t jj = 0; jj < m; ++jj) f = (f << 2) | M.code(i + km + jj);       // the m-mer that ends with base i + k - 1
                        M.min_seq = f; M.min_rc = rc_mmer32(f, m);
                    }
                    M.minimizer = S.mn; M.hash_min = S.h; M.position_min = (T0 + P) - r0; M.old_minimizer = S.mn;
                    uint32_t rev = S.rev, ahead = 0;
                    bool ended = false;
                    // a ring of the window's m-mers in LDS when one is free: an iteration is then ONE hash (the entering m-mer) and a rescan
                    // k - m + 1 LDS reads, instead of k - m + 2 hashes.  (Measured: 116 -> 61 ms for 7 000 iterations in three walks -- ~2.6 us an
                    // iteration and walk still: a lane alone in its wave pays four cycles an instruction like a full wave, and the ~500 instructions
                    // of an iteration with a rescan are what is left.  The rescan by the WAVE -- a lane per m-mer of the window, the tie rules as
                    // reductions -- is the form that would take it to ~0.3 us)
                    int ring = -1;
                    for (uint32_t q = 0; q < kSegRings && ring < 0; ++q) if (atomicCAS(&s_ring_used[q], 0u, 1u) == 0u) ring = (int)q;
                    if (ring >= 0) {
                        uint64_t* r_h = s_ring_h[ring];
                        uint32_t* r_c = s_ring_c[ring];
                        uint32_t f = 0;
                        for (uint32_t jj = 0; jj + 1 < m; ++jj) f = (f << 2) | M.code(i + jj);
                        for (uint64_t e = i; e <= i + km; ++e) {                 // the m-mers of k-mer i
                            f = ((f << 2) | M.code(e + m - 1)) & mask;
                            const uint32_t rc = rc_mmer32(f, m), canon = f < rc ? f : rc;
                            r_h[e & 63u] = xxh64_u64(canon); r_c[e & 63u] = canon | (canon != f ? 0x80000000u : 0u);
                        }
                    }
                    for (uint32_t steps = 0; steps < kSegSlowMax; ++steps, ++i) {
                        // (what all such lanes of the call may walk together is bounded: ~3 us per iteration and pass, against 0.5 ns per
                        // base for the dense + sparse passes -- counted by the counting launch, which decides for the call)
                        if (!EMIT && (steps & 255u) == 0 && atomicAdd(over_n + 1, 256u) + 256u > slow_budget) break;
                        if (i >= n_iter) { emit(open, rec_len - open, M.minimizer, rev, M.hash_min); ended = true; break; }   // the record's last super-k-mer
                        // the bases eight at a time (eight independent loads): one dependent global load per iteration was most of the 3 us
                        if ((steps & 7u) == 0) {
                            ahead = 0;
#pragma unroll
                            for (uint32_t jj = 0; jj < 8; ++jj) { const uint64_t q = i + k + jj; ahead |= (q < rec_len ? M.code(q) : 0u) << (2 * jj); }
                        }
                        const uint32_t b = (ahead >> (2 * (steps & 7u))) & 3u;
                        M.min_seq = ((M.min_seq << 2) | b) & mask;
                        M.min_rc = (M.min_rc >> 2) | ((b ^ 2u) << (2 * m - 2));
                        const uint32_t canon = M.min_seq < M.min_rc ? M.min_seq : M.min_rc;
                        const uint64_t h_in = xxh64_u64(canon);
                        if (ring >= 0) { const uint64_t e = i + km + 1; s_ring_h[ring][e & 63u] = h_in; s_ring_c[ring][e & 63u] = canon | (canon != M.min_seq ? 0x80000000u : 0u); }
                        if (h_in < M.hash_min) {                            // a reset: it cuts, and is its own tile's event from here on
                            emit(open, i + k - open, M.minimizer, rev, M.hash_min);
                            ended = true;
                            break;
                        }
                        if (i >= M.position_min) {
                            emit(open, i + k - open, M.minimizer, rev, M.hash_min);
                            if (ring >= 0) {
                                const SegState N = seg_rescan_ring(s_ring_h[ring], s_ring_c[ring], i + 1, km);
                                M.minimizer = N.mn; M.hash_min = N.h; M.position_min = N.pos + i + 1; rev = N.rev;
                            } else {
                                uint64_t pos;
                                M.rescan(i + 1, &pos);
                                M.position_min = pos + i + 1;
                                rev = M.last_rev;
                            }
                            open = i + 1;
                        }
                    }
                    if (ring >= 0) s_ring_used[ring] = 0;
                    if (!ended && !write) atomicAdd(over_n, 1u);           // longer than that: the product scan takes the call
                }
            }
            if (!write) s_cnt[j] = cnt;
        }
    };
    walk(false);
    __syncthreads();
    // exclusive prefix over the tile's iterations (8 consecutive per lane)
    {
        constexpr uint32_t PER = kSegIter / kSegThreads;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) { v[u] = s_cnt[t * PER + u]; sum += v[u]; }
        uint32_t x = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(x, dd); if (lane >= (uint32_t)dd) x += y; }
        if (lane == 63) s_wave[t >> 6] = x;
        __syncthreads();
        uint32_t pre = 0, all = 0;
        for (uint32_t w = 0; w < kSegThreads / 64; ++w) { if (w < (t >> 6)) pre += s_wave[w]; all += s_wave[w]; }
        uint32_t run = pre + x - sum;
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) { s_cnt[t * PER + u] = run; run += v[u]; }
        if (!EMIT) { if (t == 0) tile_count[blockIdx.x] = all; return; }
    }
    __syncthreads();
    walk(true);
}

// chains that left their tile's halo: the literal machine from the chain's first state to the next reset (or the record's end)
__global__ __launch_bounds__(64) void k_seg_tail(const uint8_t* __restrict__ bases, bool packed, const uint64_t* __restrict__ rec_off, uint32_t n_rec,
                                                uint32_t k, uint32_t m, unsigned long long* __restrict__ total, const uint32_t* __restrict__ file_rec,
                                                uint32_t n_files, const unsigned long long* __restrict__ over, const uint32_t* __restrict__ over_n) {
    const uint32_t x = blockIdx.x * 64 + threadIdx.x;
    const uint32_t n = *over_n < kSegOverCap ? *over_n : kSegOverCap;
    if (x >= n) return;
    const unsigned long long e = over[x];
    const uint64_t g = e >> 1;
    const bool at_reset = (e & 1ull) != 0;
    uint32_t r = 0, hi = n_rec;
    while (hi - r > 1) { const uint32_t mid = (r + hi) >> 1; if (rec_off[mid] <= g) r = mid; else hi = mid; }
    const uint64_t r0 = rec_off[r], len = rec_off[r + 1] - r0, n_iter = len - k;
    StatMachine M;
    M.bind(bases, packed, r0); M.len = len; M.k = k; M.m = m; M.mask = (1u << (2 * m)) - 1u;
    uint64_t i = g - r0;
    bool reset = false;
    if (at_reset) {
        M.start(i + 1);                                       // (sets the rolling m-mer registers to k-mer i + 1's last m-mer ...)
        const uint32_t canon = M.min_seq < M.min_rc ? M.min_seq : M.min_rc;
        M.minimizer = canon; M.old_minimizer = canon; M.hash_min = xxh64_u64(canon); M.position_min = i + k - m + 1;   // ... which is the reset's m-mer
        ++i;
    } else M.start(0);
    unsigned long long cnt = 0;
    for (; i < n_iter; ++i) {
        const uint32_t c = M.step(i, &reset);
        if (reset) break;                                     // (its cut belongs to the reset's own chain)
        cnt += c;
    }
    if (cnt) atomicAdd(&total[file_rec ? stat_file_of(file_rec, n_files, r) : 0u], cnt);
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
    static const char* dbg_stats = getenv("SPSP_DEBUG_STATS");      // "chunks": the chunk kernels for every call (A/B, tests); "tiny": see below
    const uint64_t n_tiles = (n_bases + kSegIter - 1) / kSegIter;
    if (!(dbg_stats && dbg_stats[0] == 'c') && base0 == 0 && n_tiles < 0x7fffffffull) {
        if ((rc = ctx->st_over.reserve((size_t)kSegOverCap * 8 + 64))) return rc;
        unsigned long long* d_over = ctx->st_over.as<unsigned long long>() + 1;
        uint32_t* d_over_n = ctx->st_over.as<uint32_t>();
        SPSP_HIP(hipMemsetAsync(d_over_n, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_seg_count, dim3((uint32_t)n_tiles), dim3(kSegThreads), 0, ctx->stream, d_bases, packed, n_bases, d_rec_off, n_rec, p->k, p->m,
                           d_total, file_rec, n_files, d_over, d_over_n);
        SPSP_HIP(hipGetLastError());
        // the chains that left their tile (long repeats): as many lanes as there can be -- the kernel reads the count
        hipLaunchKernelGGL(k_seg_tail, dim3(kSegOverCap / 64 / 64), dim3(64), 0, ctx->stream, d_bases, packed, d_rec_off, n_rec, p->k, p->m, d_total, file_rec, n_files,
                           (const unsigned long long*)d_over, (const uint32_t*)d_over_n);
        SPSP_HIP(hipGetLastError());
        std::vector<unsigned long long> back(n_files);
        SPSP_HIP(hipMemcpyAsync(back.data(), d_total, (size_t)n_files * 8, hipMemcpyDeviceToHost, ctx->stream));
        SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 3, d_over_n, 4, hipMemcpyDeviceToHost, ctx->stream));
        SPSP_HIP(hipStreamSynchronize(ctx->stream));
        const uint32_t tail_lanes = dbg_stats && dbg_stats[0] == 't' ? 1u : kSegOverCap / 64;   // ("tiny": the hand-over to the chunk kernels, for tests)
        if ((uint32_t)ctx->h_scalar[3] <= tail_lanes) {          // (every chain handed on had its lane)
            for (uint32_t f = 0; f < n_files; ++f) total[f] = back[f];
            return SPSP_OK;
        }
        SPSP_HIP(hipMemsetAsync(d_total, 0, (size_t)n_files * 8, ctx->stream));   // more long chains than lanes: the chunk kernels, from zero
    }
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


// the scan by segments (see k_seg_scan).  _count: queues the counting launch, the prefix over the tiles and the copies of the total and of
// the number of chains that left their tile to h_scalar[0] / [1]; _emit (after the caller has waited and reserved `out`): the writing launch
int seg_scan_count(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, uint64_t n_bases, const uint64_t* d_rec_off, uint32_t n_rec) {
    const uint64_t n_tiles = (n_bases + kSegIter - 1) / kSegIter;
    if (n_tiles > 0x7ffffff0ull) { set_error("input too large for one call"); return SPSP_ERR_OVERFLOW; }
    int rc;
    if ((rc = ctx->st_count.reserve((size_t)(n_tiles + 1) * 4 * 2 + 64)) || (rc = ctx->st_over.reserve(64))) return rc;
    uint32_t* d_cnt = ctx->st_count.as<uint32_t>();
    uint32_t* d_off = d_cnt + n_tiles + 1;
    uint32_t* d_over_n = ctx->st_over.as<uint32_t>();
    SPSP_HIP(hipMemsetAsync(d_over_n, 0, 8, ctx->stream));
    // iterations the lanes may walk alone behind their tiles' halos, all chains of the call together: as many as cost what the dense +
    // sparse passes cost the call before they meet the run themselves (~10 us per iteration over the three walks against 0.5 ns per base;
    // they are no faster on such a run: 100 Mbp with a homopolymer of 2 000 / 8 000 / 16 000 bases: 16 / 116 / 267 ms here, 60 / 145 / 241 there)
    const uint32_t slow_budget = (uint32_t)std::min<uint64_t>(1u << 20, 512 + n_bases / 20000);
    hipLaunchKernelGGL(k_seg_scan<false>, dim3((uint32_t)n_tiles), dim3(kSegThreads), 0, ctx->stream, d_bases, packed, n_bases, d_rec_off, n_rec, p->k, p->m, p->threshold,
                       d_cnt, (const uint32_t*)nullptr, (spsp_superkmer*)nullptr, 0ull, d_over_n, slow_budget);
    SPSP_HIP(hipGetLastError());
    if ((rc = launch_scan_u32(ctx, d_cnt, d_off, n_tiles, ctx->h_scalar + 0))) return rc;      // (the total to h_scalar[0])
    SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 1, d_over_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    return SPSP_OK;
}
int seg_scan_emit(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, uint64_t n_bases, const uint64_t* d_rec_off, uint32_t n_rec,
                  spsp_superkmer* d_out, uint64_t out_cap) {
    const uint64_t n_tiles = (n_bases + kSegIter - 1) / kSegIter;
    uint32_t* d_cnt = ctx->st_count.as<uint32_t>();
    uint32_t* d_off = d_cnt + n_tiles + 1;
    hipLaunchKernelGGL(k_seg_scan<true>, dim3((uint32_t)n_tiles), dim3(kSegThreads), 0, ctx->stream, d_bases, packed, n_bases, d_rec_off, n_rec, p->k, p->m, p->threshold,
                       d_cnt, (const uint32_t*)d_off, d_out, out_cap, ctx->st_over.as<uint32_t>(), 0u);
    SPSP_HIP(hipGetLastError());
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
