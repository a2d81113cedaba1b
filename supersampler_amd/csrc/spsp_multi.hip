// spsp_multi.hip -- the all-vs-all comparison split over several GPUs, behind the C-ABI.
//
// The reference's comparator is one thread over one merge (Comparator::compare_sketches, Comparator.cpp:39-74): it has no
// sharded form to mirror.  The split used here is by KEY: equal (minimizer, canonical k-mer) keys hash to the same
// device, every device holds every sketch's keys of ONE hash class and counts that class's contribution to EVERY pair,
// and inter = the sum of the partial matrices (the classes are disjoint).  Per device that is 1/G of the dictionary, of
// the lists and of the row sums -- the same share for every device, whatever order the sketches come in -- and each key
// crosses the fabric once (an all-to-all of 1/G-sized slots; an all-gather moves G times as much and leaves the device
// that owns the first rows with every other device's keys to look at).  What comes back is sparse: a
// partial matrix is non-zero only where two sketches share k-mers (95 000 of 5 x 10^7 cells at BASELINE configs[3]), so
// its cells travel as packed words, not as N x N.
//
//   sender    spsp_partition_keys_device (spsp_compare.hip): one fixed-size slot per destination
//   receiver  compare_slots_begin_impl: slot headers to the host (counts), k_slot_unpack -> flat key arrays of all
//             G x n sketches, then the partition-form comparison of spsp_compare.hip with every row owned
//   result    k_matrix_cells: non-zero cells (i < j) of the partial matrix as i << 48 | j << 32 | count;
//             k_matrix_add_cells: inter[i][j] += count on whoever collects them
//   driver    compare_payloads_multi: one context per device (or several on one device), one host thread each:
//             decode own block of sketch files -> partition -> peer copies -> compare -> cells -> host matrix
#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

// slot records -> flat key arrays.  Slot s holds, sketch by sketch, the keys of the sender's n sketches that belong to
// this receiver's hash class; its records land at base[s] + position: global sketch s * n + j keeps its order.
__global__ __launch_bounds__(256) void k_slot_unpack(const uint8_t* __restrict__ slots, uint64_t slot_sz, uint64_t rec_off, uint32_t words,
                                                    const uint32_t* __restrict__ tot, const uint32_t* __restrict__ base, uint32_t n,
                                                    const uint64_t* __restrict__ sk_off, uint32_t* __restrict__ bad,
                                                    uint32_t* __restrict__ o_mn, uint64_t* __restrict__ o_lo, uint64_t* __restrict__ o_hi) {
    const uint32_t s = blockIdx.y;
    const uint32_t n_rec = tot[s], b0 = base[s];
    const uint64_t* rec = reinterpret_cast<const uint64_t*>(slots + (uint64_t)s * slot_sz + rec_off);
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n_rec; e += gridDim.x * 256) {
        const uint64_t* r = rec + (uint64_t)e * words;
        o_lo[b0 + e] = r[0];
        if (o_hi) o_hi[b0 + e] = words == 3 ? r[1] : 0ull;
        o_mn[b0 + e] = (uint32_t)r[words - 1];
        // the record names its sketch (the sender's local number): it must be the one whose run of the counts this position lies in
        const uint32_t j = (uint32_t)(r[words - 1] >> 32);
        if (j >= n || sk_off[(uint64_t)s * n + j] > b0 + e || sk_off[(uint64_t)s * n + j + 1] <= b0 + e) atomicOr(bad, 1u);
    }
}

// non-zero cells (i < j) of rows [row_first, row_limit) as packed words; *count keeps counting past cap (the caller
// then calls again with room for all of them)
__global__ __launch_bounds__(256) void k_matrix_cells(const uint32_t* __restrict__ inter, uint32_t n, uint32_t row_first, uint32_t row_limit,
                                                     unsigned long long* __restrict__ cells, unsigned long long cap,
                                                     unsigned long long* __restrict__ count) {
    const uint32_t i = row_first + blockIdx.y;
    if (i >= row_limit || i >= n) return;
    const uint32_t col0 = blockIdx.x * 1024;
    if (col0 + 1024 <= i + 1) return;                              // no column > i in this block
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t j = col0 + u * 256 + threadIdx.x;
        const uint32_t v = (j > i && j < n) ? inter[(uint64_t)i * n + j] : 0u;
        const unsigned long long hit = __ballot(v != 0);
        if (!hit) continue;
        unsigned long long at = 0;
        const int leader = __ffsll((long long)hit) - 1;
        if ((int)lane == leader) at = atomicAdd(count, (unsigned long long)__popcll(hit));
        at = __shfl(at, leader) + (unsigned long long)__popcll(hit & ((1ull << lane) - 1ull));
        if (v && at < cap) cells[at] = ((unsigned long long)i << 48) | ((unsigned long long)j << 32) | v;
    }
}

__global__ __launch_bounds__(256) void k_matrix_add_cells(uint32_t* __restrict__ inter, uint32_t n, const unsigned long long* __restrict__ cells,
                                                         unsigned long long n_cells, uint32_t* __restrict__ bad) {
    for (unsigned long long e = blockIdx.x * 256ull + threadIdx.x; e < n_cells; e += gridDim.x * 256ull) {
        const unsigned long long c = cells[e];
        const uint32_t i = (uint32_t)(c >> 48), j = (uint32_t)(c >> 32) & 0xffffu;
        if (i >= n || j >= n || i >= j) { atomicOr(bad, 1u); continue; }
        atomicAdd(&inter[(uint64_t)i * n + j], (uint32_t)c);
    }
}

int matrix_cells_impl(spsp_ctx* ctx, const uint32_t* d_inter, uint32_t n, uint32_t row_first, uint32_t row_limit, uint64_t* d_cells,
                      uint64_t cap, uint64_t* n_cells) {
    if (n > 65535) { set_error("at most 65535 sketches (the packed cell holds two 16-bit sketch numbers; Comparator.h:26 has the same bound)"); return SPSP_ERR_ARG; }
    *n_cells = 0;
    if (n < 2 || row_first >= row_limit) return SPSP_OK;
    if (row_limit > n) row_limit = n;
    int rc;
    if ((rc = ctx->c_flags.reserve(256))) return rc;
    unsigned long long* d_count = reinterpret_cast<unsigned long long*>(ctx->c_flags.as<uint32_t>() + 14);   // two words of the flag block nobody else uses
    SPSP_HIP(hipMemsetAsync(d_count, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_matrix_cells, dim3((n + 1023) / 1024, row_limit - row_first), dim3(256), 0, ctx->stream, d_inter, n, row_first, row_limit,
                       reinterpret_cast<unsigned long long*>(d_cells), (unsigned long long)cap, d_count);
    SPSP_HIP(hipGetLastError());
    SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 12, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    *n_cells = ctx->h_scalar[12];
    if (*n_cells > cap) { set_error("the matrix has %llu non-zero cells, room was given for %llu", (unsigned long long)*n_cells, (unsigned long long)cap); return SPSP_ERR_OVERFLOW; }
    return SPSP_OK;
}

// after compare_end of a comparison queued by compare_slots_begin_impl: did k_slot_unpack meet a record that names a sketch its
// position does not belong to?  (the records were compared all the same -- as keys of the sketch their POSITION belongs to;
// the result of such a call is refused)
int slots_bad_record(spsp_ctx* ctx) {
    if (!ctx->m_slots_job) return SPSP_OK;
    ctx->m_slots_job = false;
    if ((uint32_t)ctx->h_scalar[13]) { set_error("an exchange slot is malformed (a record names a sketch its position does not belong to)"); return SPSP_ERR_FORMAT; }
    return SPSP_OK;
}

// grow (may be NULL): the context buffer d_cells lives in -- when the result went through the dense matrix and has more
// cells than `cap`, the buffer is grown and the matrix sparsified again instead of the caller repeating the comparison
int compare_cells_run(spsp_ctx* ctx, const std::function<int()>& begin, uint32_t n, uint32_t row_limit, uint32_t* d_scratch, uint64_t* d_cells,
                      uint64_t cap, uint64_t* n_cells, DevBuf* grow) {
    if (n > 65535) { set_error("at most 65535 sketches (the packed cell holds two 16-bit sketch numbers; Comparator.h:26 has the same bound)"); return SPSP_ERR_ARG; }
    *n_cells = 0;
    int rc;
    if ((rc = ctx->c_flags.reserve(256))) return rc;
    unsigned long long* d_count = reinterpret_cast<unsigned long long*>(ctx->c_flags.as<uint32_t>() + 14);
    SPSP_HIP(hipMemsetAsync(d_count, 0, 8, ctx->stream));
    ctx->cells_req.cells = reinterpret_cast<unsigned long long*>(d_cells); ctx->cells_req.cap = cap; ctx->cells_req.count = d_count;
    ctx->cells_req.armed = row_limit >= n;                         // (query mode: rows are limited by the sparsifier, through the dense matrix)
    ctx->cells_req.direct = false;
    rc = begin();
    if (!rc) rc = compare_end_impl(ctx);
    if (!rc) rc = slots_bad_record(ctx);
    const bool direct = ctx->cells_req.direct;
    ctx->cells_req = spsp_ctx::CellsReq{};
    if (rc) return rc;
    if (!direct) {
        rc = matrix_cells_impl(ctx, d_scratch, n, 0, row_limit, d_cells, cap, n_cells);
        if (rc == SPSP_ERR_OVERFLOW && grow && grow->p == (void*)d_cells) {
            const uint64_t need = *n_cells;
            if ((rc = grow->reserve((size_t)need * 8))) return rc;
            rc = matrix_cells_impl(ctx, d_scratch, n, 0, row_limit, grow->as<uint64_t>(), need, n_cells);
        }
        return rc;
    }
    *n_cells = ctx->h_scalar[12];                                  // (copied to pinned memory behind the row sums: compare_end has waited for it)
    if (*n_cells > cap) { set_error("the matrix has %llu non-zero cells, room was given for %llu", (unsigned long long)*n_cells, (unsigned long long)cap); return SPSP_ERR_OVERFLOW; }
    return SPSP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Receiver of the key-partitioned split.  The slot headers (magic, geometry, keys per sketch) go to the host first: the
// flat form needs the sketch offsets there, and a malformed or overflowed slot is refused before any kernel reads it.
// h_headers (optional): the `parts` slot headers (slot_rec_off(n) bytes each, back to back) in HOST memory already -- the
// senders of a one-process split read them back behind their partition, in the wait they make anyway (compare_payloads_multi),
// so the receiver queues its unpack without a read-back and a wait of its own
int compare_slots_begin_impl(spsp_ctx* ctx, uint32_t k, const uint8_t* d_slots, uint32_t parts, uint32_t n, uint32_t cap,
                             uint32_t* d_inter, const uint8_t* h_headers) {
    if (ctx->compare_job) { set_error("a comparison is already pending on this context: call spsp_compare_end first"); return SPSP_ERR_ARG; }
    if (n == 0 || parts == 0 || parts > kMaxParts) { set_error("bad slot geometry"); return SPSP_ERR_ARG; }
    const uint64_t N = (uint64_t)parts * n;
    if (N > 65535) { set_error("at most 65535 sketches (the reference's uint32 pair key, Comparator.h:26)"); return SPSP_ERR_ARG; }
    const uint64_t E = (uint64_t)parts * cap;
    if (cap == 0 || E > 0xfffffff0ull) { set_error("too many sketch k-mers for one call"); return SPSP_ERR_OVERFLOW; }
    if (((uintptr_t)d_slots & 7u) != 0) { set_error("d_slots must be 8-byte aligned"); return SPSP_ERR_ARG; }
    const uint64_t slot_sz = slot_bytes(n, cap, k), hdr = slot_rec_off(n);
    const uint32_t words = slot_words(k);
    std::vector<uint8_t> h;
    if (h_headers) h.assign(h_headers, h_headers + (size_t)parts * hdr);
    else {
        h.resize((size_t)parts * hdr);
        SPSP_HIP(hipMemcpy2DAsync(h.data(), hdr, d_slots, slot_sz, hdr, parts, hipMemcpyDeviceToHost, ctx->stream));
        SPSP_HIP(hipStreamSynchronize(ctx->stream));
    }
    // (host arrays the queued copies read: kept in the context until its next call, so that nothing here has to wait for them)
    std::vector<uint64_t>& sk_off = ctx->m_h_skoff;
    std::vector<uint32_t>&tot = ctx->m_h_tot, &base = ctx->m_h_base;
    sk_off.assign((size_t)N + 1, 0); tot.assign(parts, 0); base.assign(parts, 0);
    uint64_t at = 0;
    for (uint32_t s = 0; s < parts; ++s) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(h.data() + (size_t)s * hdr);
        if (w[0] != kSlotMagic || w[1] != n || w[3] != words) { set_error("exchange slot %u is malformed (magic / geometry)", s); return SPSP_ERR_FORMAT; }
        if (w[2] > cap) { set_error("an exchange slot overflowed its capacity (%u keys, room for %u): partition again with a larger slot_cap", w[2], cap); return SPSP_ERR_OVERFLOW; }
        uint64_t sum = 0;
        for (uint32_t j = 0; j < n; ++j) { sum += w[4 + j]; sk_off[(size_t)s * n + j + 1] = at + sum; }
        if (sum != w[2]) { set_error("exchange slot %u is malformed (its counts do not add up)", s); return SPSP_ERR_FORMAT; }
        tot[s] = w[2]; base[s] = (uint32_t)at;
        at += sum;
    }
    int rc;
    const bool has_hi = k > 32;
    if ((rc = ctx->m_mn.reserve((size_t)at * 4 + 64)) || (rc = ctx->m_lo.reserve((size_t)at * 8 + 64)) || (has_hi && (rc = ctx->m_hi.reserve((size_t)at * 8 + 64))) ||
        (rc = ctx->x_tot.reserve((size_t)parts * 8 + 8)) || (rc = ctx->x_begin.reserve((size_t)(N + 1) * 8))) return rc;
    if (at) {
        uint32_t* d_tot = ctx->x_tot.as<uint32_t>();
        uint32_t* d_bad = d_tot + 2 * parts;
        SPSP_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
        SPSP_HIP(hipMemcpyAsync(ctx->x_begin.p, sk_off.data(), (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        SPSP_HIP(hipMemcpyAsync(d_tot, tot.data(), (size_t)parts * 4, hipMemcpyHostToDevice, ctx->stream));
        SPSP_HIP(hipMemcpyAsync(d_tot + parts, base.data(), (size_t)parts * 4, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t gx = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, (uint64_t)cap / 1024));
        hipLaunchKernelGGL(k_slot_unpack, dim3(gx, parts), dim3(256), 0, ctx->stream, d_slots, slot_sz, hdr, words, (const uint32_t*)d_tot, (const uint32_t*)(d_tot + parts), n,
                           (const uint64_t*)ctx->x_begin.p, d_bad, ctx->m_mn.as<uint32_t>(), ctx->m_lo.as<uint64_t>(), has_hi ? ctx->m_hi.as<uint64_t>() : (uint64_t*)nullptr);
        SPSP_HIP(hipGetLastError());
        SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 13, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));   // read by slots_bad_record() once the job has been waited for
    }
    const int rc2 = compare_device_begin_impl(ctx, k, ctx->m_mn.as<uint32_t>(), ctx->m_lo.as<uint64_t>(), has_hi ? ctx->m_hi.as<uint64_t>() : nullptr, sk_off.data(),
                                              (uint32_t)N, (uint32_t)N, 0, 1, d_inter);
    ctx->m_slots_job = rc2 == SPSP_OK && at > 0;                   // (its record check is read behind compare_end: slots_bad_record)
    return rc2;
}

// ---------------------------------------------------------------------------------------------------------------
// Several contexts, one host thread each.  Stage by stage (the threads meet between stages):
//   A  decode this context's block of sketch payloads on its device (spsp_decode.hip)            -> keys of its sketches
//   B  partition them by hash class into one slot per context (spsp_partition_keys_device)
//   C  fetch the slots of this context's class from every context (peer copies over xGMI; same-device copies when
//      several contexts share a device)
//   D  partition-form comparison of ALL sketches' keys of that class (every row owned)            -> dense partial matrix
//   E  its non-zero cells to the host, added into the caller's matrix
namespace {
struct MultiShared {
    uint32_t n = 0, n_ctx = 0, per = 0, k = 0, m = 0, n_query = 0;
    std::vector<std::vector<uint64_t>> sk_off;                     // per context: offsets of its `per` sketches (absent ones are empty)
    std::vector<int> rc;
    std::vector<std::string> err;
    uint32_t cap = 0;
};
template <class F>
int run_stage(uint32_t n_ctx, MultiShared& S, F&& body) {
    std::vector<std::thread> pool;
    for (uint32_t d = 0; d < n_ctx; ++d)
        pool.emplace_back([&, d]() { S.rc[d] = body(d); if (S.rc[d]) S.err[d] = spsp_last_error(); });
    for (auto& t : pool) t.join();
    for (uint32_t d = 0; d < n_ctx; ++d)
        if (S.rc[d]) { set_error("%s", S.err[d].c_str()); return S.rc[d]; }
    return SPSP_OK;
}
}  // namespace

int compare_payloads_multi(spsp_ctx* const* ctxs, uint32_t n_ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n,
                           const int* extra_has, const uint32_t* extra_mn, uint32_t n_query, uint32_t* k_out, uint32_t* m_out,
                           uint32_t* inter, uint64_t* card, bool* mirrored, std::vector<uint64_t>* cells_out) {
    if (mirrored) *mirrored = false;
    if (cells_out) cells_out->clear();
    if (n_ctx == 0 || n_ctx > kMaxParts) { set_error("1..%u contexts", kMaxParts); return SPSP_ERR_ARG; }
    if (n == 0) { *k_out = *m_out = 0; return SPSP_OK; }
    // the exchange numbers sketches per * n_ctx (the last block padded): near the 65 535-sketch limit the padding alone can
    // cross it (n = 65 535 over two contexts: 65 536) although the comparison itself is allowed -- then fewer contexts take part
    // (ADVICE r4); one context always fits what the single-device path accepts
    while (n_ctx > 1 && n <= 65535 && (uint64_t)((n + n_ctx - 1) / n_ctx) * n_ctx > 65535) --n_ctx;
    MultiShared S;
    S.n = n; S.n_ctx = n_ctx; S.n_query = n_query;
    S.per = (n + n_ctx - 1) / n_ctx;
    const uint64_t NP = (uint64_t)S.per * n_ctx;                   // sketch numbers of the exchange: context d owns [d per, (d + 1) per)
    if (NP > 65535) { set_error("at most 65535 sketches (the reference's uint32 pair key, Comparator.h:26)"); return SPSP_ERR_ARG; }
    S.sk_off.assign(n_ctx, std::vector<uint64_t>((size_t)S.per + 1, 0));
    S.rc.assign(n_ctx, SPSP_OK); S.err.assign(n_ctx, std::string());
    std::vector<uint32_t> ks(n_ctx, 0), ms(n_ctx, 0);
    // (inter is zero on entry)
    // peers: every context's device reads the others' slots
    for (uint32_t a = 0; a < n_ctx; ++a)
        for (uint32_t b = 0; b < n_ctx; ++b) {
            if (ctxs[a]->device == ctxs[b]->device) continue;
            (void)hipSetDevice(ctxs[a]->device);
            const hipError_t e = hipDeviceEnablePeerAccess(ctxs[b]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); }   // (copies then stage through the host: slower, same result)
            else if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    int rc;
    // A: decode
    rc = run_stage(n_ctx, S, [&](uint32_t d) -> int {
        spsp_ctx* c = ctxs[d];
        SPSP_HIP(hipSetDevice(c->device));
        const uint32_t b0 = std::min<uint64_t>((uint64_t)d * S.per, n), b1 = std::min<uint64_t>((uint64_t)(d + 1) * S.per, n);
        std::vector<uint64_t> off((size_t)(b1 - b0) + 1, 0);
        if (b1 > b0) {
            const int r = sketch_decode_device_impl(c, payloads + b0, lens + b0, b1 - b0, extra_has ? extra_has + b0 : nullptr, extra_mn ? extra_mn + b0 : nullptr,
                                                    &ks[d], &ms[d], off.data());
            if (r) return r;
        }
        for (uint32_t j = 0; j <= S.per; ++j) S.sk_off[d][j] = off[std::min<size_t>(j, off.size() - 1)];
        for (uint32_t i = b0; i < b1; ++i) card[i] = off[i - b0 + 1] - off[i - b0];
        return SPSP_OK;
    });
    if (rc) return rc;
    for (uint32_t d = 0; d < n_ctx; ++d) {
        const uint32_t b0 = std::min<uint64_t>((uint64_t)d * S.per, n), b1 = std::min<uint64_t>((uint64_t)(d + 1) * S.per, n);
        if (b1 == b0) continue;
        if (!S.k) { S.k = ks[d]; S.m = ms[d]; }
        else if (ks[d] != S.k || ms[d] != S.m) { set_error("sketches were made with different k / m (k=%u m=%u and k=%u m=%u)", S.k, S.m, ks[d], ms[d]); return SPSP_ERR_FORMAT; }
    }
    *k_out = S.k; *m_out = S.m;
    uint64_t most = 0, all = 0;
    for (uint32_t d = 0; d < n_ctx; ++d) { most = std::max(most, S.sk_off[d][S.per]); all += S.sk_off[d][S.per]; }
    if (all == 0) return SPSP_OK;
    const bool has_hi = S.k > 32;
    // B + C + D, repeated with exact room when a slot turns out fuller than the hash's mean share suggested
    S.cap = (uint32_t)std::min<uint64_t>(0x7ffffff0ull, most / n_ctx + most / (4 * n_ctx) + 4096);
    std::vector<std::vector<uint8_t>> hdrs(n_ctx);                 // [sender]: its n_ctx slot headers
    for (int attempt = 0;; ++attempt) {
        const uint64_t slot_sz = slot_bytes(S.per, S.cap, S.k);
        std::vector<uint32_t> fullest(n_ctx, 0);
        rc = run_stage(n_ctx, S, [&](uint32_t d) -> int {
            spsp_ctx* c = ctxs[d];
            SPSP_HIP(hipSetDevice(c->device));
            int r;
            if ((r = c->m_send.reserve((size_t)slot_sz * n_ctx + 64)) || (r = c->m_recv.reserve((size_t)slot_sz * n_ctx + 64))) return r;
            if ((r = partition_keys_impl(c, S.k, c->c_min.as<uint32_t>(), c->c_lo.as<uint64_t>(), has_hi ? c->c_hi.as<uint64_t>() : nullptr, S.sk_off[d].data(), S.per,
                                         n_ctx, S.cap, c->m_send.as<uint8_t>()))) return r;
            // how full did the slots get? (header word 2 = keys that wanted in) -- the WHOLE headers come back in this wait: the
            // receivers take them from here (hdrs[sender][destination]) instead of reading them back once more
            const uint64_t hdr = slot_rec_off(S.per);
            hdrs[d].resize((size_t)n_ctx * hdr);
            SPSP_HIP(hipMemcpy2DAsync(hdrs[d].data(), hdr, c->m_send.p, slot_sz, hdr, n_ctx, hipMemcpyDeviceToHost, c->stream));
            SPSP_HIP(hipStreamSynchronize(c->stream));
            for (uint32_t p = 0; p < n_ctx; ++p) fullest[d] = std::max(fullest[d], reinterpret_cast<const uint32_t*>(hdrs[d].data() + (size_t)p * hdr)[2]);
            return SPSP_OK;
        });
        if (rc) return rc;
        const uint32_t need = *std::max_element(fullest.begin(), fullest.end());
        if (need <= S.cap) break;
        if (attempt) { set_error("exchange slots overflowed twice"); return SPSP_ERR_OVERFLOW; }
        S.cap = need;
    }
    const uint64_t slot_sz = slot_bytes(S.per, S.cap, S.k);
    std::vector<std::vector<uint64_t>> host_cells(n_ctx);
    std::vector<uint64_t> n_cells(n_ctx, 0);
    rc = run_stage(n_ctx, S, [&](uint32_t d) -> int {
        spsp_ctx* c = ctxs[d];
        SPSP_HIP(hipSetDevice(c->device));
        // C: slot d of every context (every sender has synchronised its stream at the end of stage B)
        for (uint32_t s = 0; s < n_ctx; ++s) {
            const uint8_t* src = ctxs[s]->m_send.as<uint8_t>() + (size_t)d * slot_sz;
            uint8_t* dst = c->m_recv.as<uint8_t>() + (size_t)s * slot_sz;
            if (ctxs[s]->device == c->device) SPSP_HIP(hipMemcpyAsync(dst, src, slot_sz, hipMemcpyDeviceToDevice, c->stream));
            else SPSP_HIP(hipMemcpyPeerAsync(dst, c->device, src, ctxs[s]->device, slot_sz, c->stream));
        }
        // D + E: every sketch's keys of this class, every row owned; the partial matrix is sparse: its non-zero cells leave
        // the row sums as packed words (query mode: through the dense matrix, rows of the query sketches only)
        int r;
        const size_t cells_n = (size_t)NP * NP;
        if ((r = c->c_inter.reserve(cells_n * 4))) return r;
        const uint32_t row_limit = (uint32_t)std::min<uint64_t>(NP, n_query);
        uint64_t cap = std::max<uint64_t>(1u << 16, (uint64_t)NP * 32);
        for (int attempt = 0; attempt < 2; ++attempt) {
            if ((r = c->m_cells.reserve((size_t)cap * 8))) return r;
            // the headers of slot d of every sender, in sender order: what m_recv holds
            const uint64_t hdr = slot_rec_off(S.per);
            std::vector<uint8_t> mine((size_t)n_ctx * hdr);
            for (uint32_t s = 0; s < n_ctx; ++s) memcpy(mine.data() + (size_t)s * hdr, hdrs[s].data() + (size_t)d * hdr, (size_t)hdr);
            r = compare_cells_run(c, [&]() { return compare_slots_begin_impl(c, S.k, c->m_recv.as<uint8_t>(), n_ctx, S.per, S.cap, c->c_inter.as<uint32_t>(), mine.data()); },
                                  (uint32_t)NP, row_limit, c->c_inter.as<uint32_t>(), c->m_cells.as<uint64_t>(), cap, &n_cells[d], &c->m_cells);
            if (r != SPSP_ERR_OVERFLOW) break;
            cap = n_cells[d];
        }
        if (r) return r;
        host_cells[d].resize((size_t)n_cells[d]);
        if (n_cells[d]) {
            SPSP_HIP(hipMemcpyAsync(host_cells[d].data(), c->m_cells.p, (size_t)n_cells[d] * 8, hipMemcpyDeviceToHost, c->stream));
            SPSP_HIP(hipStreamSynchronize(c->stream));
        }
        return SPSP_OK;
    });
    if (rc) return rc;
    if (cells_out) {
        // the caller prints from the cells: the contexts' partial cells of one pair are added up (sort by pair, sum the runs)
        // Grouped by row first (a counting sort over the row number), then every row's few cells sorted and summed on host
        // threads: one std::sort over everything took 0.7 s of a 1.2 s call when the matrix is dense (4 000 files of one species:
        // three contexts x 8 x 10^6 cells)
        std::vector<uint64_t>& all = *cells_out;
        std::vector<uint64_t> row_at((size_t)NP + 2, 0);
        for (uint32_t d = 0; d < n_ctx; ++d) for (uint64_t cw : host_cells[d]) ++row_at[(size_t)(cw >> 48) + 2];
        for (size_t i = 2; i < row_at.size(); ++i) row_at[i] += row_at[i - 1];
        std::vector<uint64_t> grouped((size_t)row_at.back());
        for (uint32_t d = 0; d < n_ctx; ++d) {
            for (uint64_t cw : host_cells[d]) grouped[(size_t)row_at[(size_t)(cw >> 48) + 1]++] = cw;     // row_at[i + 1] walks from row i's first cell to its end
            std::vector<uint64_t>().swap(host_cells[d]);
        }
        std::vector<uint64_t> kept((size_t)NP + 1, 0);         // cells row i keeps (merged, inside the matrix)
        unsigned workers = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (grouped.size() < (1u << 16)) workers = 1;
        std::atomic<uint64_t> next_row{0};
        auto merge_rows = [&]() {
            for (;;) {
                const uint64_t r0 = next_row.fetch_add(64);
                if (r0 >= NP) return;
                for (uint64_t i = r0; i < std::min<uint64_t>(NP, r0 + 64); ++i) {
                    uint64_t* a = grouped.data() + row_at[(size_t)i], *z = grouped.data() + row_at[(size_t)i + 1];
                    std::sort(a, z);
                    uint64_t* w = a;
                    for (uint64_t* r = a; r < z;) {
                        const uint64_t pair = *r >> 32;
                        uint64_t sum = 0;
                        for (; r < z && (*r >> 32) == pair; ++r) sum += (uint32_t)*r;
                        if ((uint32_t)(pair >> 16) < n && (uint32_t)(pair & 0xffffu) < n) *w++ = (pair << 32) | (uint32_t)sum;
                    }
                    kept[(size_t)i] = (uint64_t)(w - a);
                }
            }
        };
        {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < workers; ++t) pool.emplace_back(merge_rows);
            merge_rows();
            for (auto& th : pool) th.join();
        }
        uint64_t total = 0;
        for (uint64_t i = 0; i < NP; ++i) total += kept[(size_t)i];
        all.resize((size_t)total);
        size_t w = 0;
        for (uint64_t i = 0; i < NP; ++i) { memcpy(all.data() + w, grouped.data() + row_at[(size_t)i], (size_t)kept[(size_t)i] * 8); w += (size_t)kept[(size_t)i]; }
        return SPSP_OK;
    }
    for (uint32_t d = 0; d < n_ctx; ++d)
        for (uint64_t cw : host_cells[d]) {
            const uint32_t i = (uint32_t)(cw >> 48), j = (uint32_t)(cw >> 32) & 0xffffu;
            if (i < n && j < n) {                                              // (exchange numbers >= n are the padding of the last block: no keys, no cells)
                inter[(size_t)i * n + j] += (uint32_t)cw;
                if (mirrored) inter[(size_t)j * n + i] += (uint32_t)cw;        // (the printers then read rows only)
            }
        }
    if (mirrored) *mirrored = true;
    return SPSP_OK;
}

}  // namespace spsp

using namespace spsp;

extern "C" {

int spsp_matrix_cells_device(spsp_ctx* ctx, const void* d_inter, uint32_t n, uint32_t row_first, uint32_t row_limit, void* d_cells, uint64_t cap,
                             uint64_t* n_cells) {
    if (!ctx || !d_inter || !n_cells || (cap && !d_cells)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return matrix_cells_impl(ctx, (const uint32_t*)d_inter, n, row_first, row_limit, (uint64_t*)d_cells, cap, n_cells);
}

int spsp_compare_slots_cells_device(spsp_ctx* ctx, uint32_t k, const void* d_slots, uint32_t parts, uint32_t n, uint32_t slot_cap, void* d_scratch,
                                    void* d_cells, uint64_t cap, uint64_t* n_cells) {
    if (!ctx || !d_slots || !d_scratch || !n_cells || (cap && !d_cells)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    const uint64_t N = (uint64_t)parts * n;
    if (N > 65535) { set_error("at most 65535 sketches"); return SPSP_ERR_ARG; }
    return compare_cells_run(ctx, [&]() { return compare_slots_begin_impl(ctx, k, (const uint8_t*)d_slots, parts, n, slot_cap, (uint32_t*)d_scratch); }, (uint32_t)N, (uint32_t)N,
                             (uint32_t*)d_scratch, (uint64_t*)d_cells, cap, n_cells);
}

int spsp_compare_cells_device(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo, const void* d_kmer_hi, const uint64_t* h_sk_off,
                              uint32_t n, uint32_t n_query, void* d_scratch, void* d_cells, uint64_t cap, uint64_t* n_cells) {
    if (!ctx || !h_sk_off || !d_scratch || !n_cells || (cap && !d_cells)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return compare_cells_run(ctx, [&]() { return compare_device_begin_impl(ctx, k, (const uint32_t*)d_minimizer, (const uint64_t*)d_kmer_lo, (const uint64_t*)d_kmer_hi,
                                                                            h_sk_off, n, n_query, 0, 1, (uint32_t*)d_scratch); },
                             n, n_query < n ? n_query : n, (uint32_t*)d_scratch, (uint64_t*)d_cells, cap, n_cells);
}

int spsp_matrix_add_cells_device(spsp_ctx* ctx, void* d_inter, uint32_t n, const void* d_cells, uint64_t n_cells) {
    if (!ctx || !d_inter || (n_cells && !d_cells)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (n_cells == 0) return SPSP_OK;
    SPSP_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->c_flags.reserve(256))) return rc;
    uint32_t* d_bad = ctx->c_flags.as<uint32_t>() + 14;
    SPSP_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_matrix_add_cells, dim3((uint32_t)std::min<uint64_t>(4096, (n_cells + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)d_inter, n,
                       (const unsigned long long*)d_cells, (unsigned long long)n_cells, d_bad);
    SPSP_HIP(hipGetLastError());
    SPSP_HIP(hipMemcpyAsync(ctx->h_scalar + 12, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    if ((uint32_t)ctx->h_scalar[12]) { set_error("a cell names a sketch outside the matrix (or a pair that is not i < j)"); return SPSP_ERR_FORMAT; }
    return SPSP_OK;
}

}  // extern "C"
