// spsp_abund.hip -- "next" row N4: the -a abundance filter on the GPU.
//
// The reference counts every k-mer of every selected super-k-mer in a per-minimizer map (handle_superkmer,
// SubSampler.cpp:243-302: uint8 count, wraps at 256) and lets find_first_kmer / find_next use a k-mer only if
// `count >= abundance` (SubSampler.cpp:587,608).  A k-mer below the threshold is never written, never followed
// and never marks anything: it only shows up in "After removing duplicate kmers" (its map entry) and in its
// bucket existing.  So with -a > 1 the counting happens here, over the gathered super-k-mers that are on the
// device anyway, and the host sketch builder indexes usable k-mers only:
//
//   k_abund_sizes    k-mers per selected super-k-mer -> (scan) first occurrence number of each
//   k_abund_emit     one lane per super-k-mer: oriented as handle_superkmer stores it (reverse complement when the
//                    minimizer reads reversed), rolls the k-mers (2-bit, 128 bits) -> key arrays by occurrence
//   k_abund_insert   open-addressing table of occurrence numbers: a key claims a slot with one CAS or finds the
//                    slot whose claimer holds the same FULL key (no fingerprints: nothing to retry); one count per slot
//   k_abund_flags    per occurrence: bit 0 = usable ((count mod 256) >= abundance), bit 1 = this occurrence is the
//                    claimer of a key that is not usable (one per distinct dropped k-mer: the builder keeps an empty
//                    map entry for it, which is all the reference keeps of such a k-mer)
#include <cstring>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

__global__ void k_abund_sizes(const spsp_superkmer* __restrict__ sk, uint32_t n_sk, uint32_t k, uint32_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sk) cnt[i] = sk[i].len >= k ? sk[i].len - k + 1 : 0u;
}

__global__ __launch_bounds__(256) void k_abund_emit(const uint8_t* __restrict__ compact, const uint32_t* __restrict__ src_off,
                                                   const spsp_superkmer* __restrict__ sk, const uint32_t* __restrict__ occ_off,
                                                   uint32_t n_sk, uint32_t k, uint32_t* __restrict__ k_mn,
                                                   uint64_t* __restrict__ k_lo, uint64_t* __restrict__ k_hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sk) return;
    const spsp_superkmer e = sk[i];
    if (e.len < k) return;
    const uint8_t* s = compact + src_off[i];
    const uint64_t mask_hi = k > 32 ? ((1ull << (2 * k - 64)) - 1) : 0ull;   // k <= 63
    const uint64_t mask_lo = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t hi = 0, lo = 0;
    uint32_t o = occ_off[i];
    for (uint32_t t = 0; t < e.len; ++t) {
        const uint32_t c = e.rev ? ((((uint32_t)s[e.len - 1 - t] >> 1) & 3u) ^ 2u) : (((uint32_t)s[t] >> 1) & 3u);
        hi = ((hi << 2) | (lo >> 62)) & mask_hi;
        lo = ((lo << 2) | c) & mask_lo;
        if (t + 1 < k) continue;
        k_mn[o] = e.minimizer; k_lo[o] = lo; k_hi[o] = hi;
        ++o;
    }
}

__device__ __forceinline__ uint64_t abund_mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t abund_hash(uint32_t mn, uint64_t lo, uint64_t hi) {
    uint64_t h = abund_mix(lo ^ 0x9E3779B97F4A7C15ULL);
    h = abund_mix(h + (uint64_t)mn * 0xD6E8FEB86659FD93ULL);
    return abund_mix(h ^ hi);
}

// seg_sk (n_seg + 1 super-k-mer numbers; nullptr = one segment): the super-k-mers of a BATCH of files, counted file by file in ONE
// pass -- the reference's map is per file (one Subsampler per file, SubSampler.cpp:787), so a k-mer of another file is
// another key: the segment is hashed with the key and a claimer outside the occurrence's own segment never matches
// (round 5: -a > 1 through the batched file pipeline, one GPU job per batch instead of one per file)
__global__ void k_abund_segments(const uint32_t* __restrict__ seg_sk, uint32_t n_seg, const uint32_t* __restrict__ occ_off, uint32_t* __restrict__ seg_occ) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= n_seg) seg_occ[s] = occ_off[seg_sk[s]];
}
__global__ __launch_bounds__(256) void k_abund_insert(const uint32_t* __restrict__ k_mn, const uint64_t* __restrict__ k_lo,
                                                     const uint64_t* __restrict__ k_hi, uint32_t n_occ, uint32_t* __restrict__ slot,
                                                     uint32_t* __restrict__ count, uint32_t cap_mask, uint32_t* __restrict__ slot_of,
                                                     const uint32_t* __restrict__ seg_occ, uint32_t n_seg) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_occ) return;
    const uint32_t mn = k_mn[o];
    const uint64_t lo = k_lo[o], hi = k_hi[o];
    uint32_t seg = 0, first = 0, last = n_occ;
    if (seg_occ) {                                        // the last segment whose first occurrence is <= o
        uint32_t a = 0, z = n_seg;
        while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if (seg_occ[mid] <= o) a = mid; else z = mid; }
        seg = a; first = seg_occ[a]; last = seg_occ[a + 1];
    }
    uint32_t h = (uint32_t)abund_hash(mn + seg * 0x9E3779B1u, lo, hi) & cap_mask;
    for (;;) {                                            // ends: the table has at least twice as many slots as keys
        uint32_t cur = slot[h];
        if (cur == 0) cur = atomicCAS(&slot[h], 0u, o + 1);
        if (cur == 0) break;                              // claimed
        const uint32_t c = cur - 1;                       // the claimer's key was written by the kernel before this one
        if (c >= first && c < last && k_lo[c] == lo && k_mn[c] == mn && k_hi[c] == hi) break;
        h = (h + 1) & cap_mask;
    }
    atomicAdd(&count[h], 1u);
    slot_of[o] = h;
}

__global__ __launch_bounds__(256) void k_abund_flags(uint32_t n_occ, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ slot,
                                                    const uint32_t* __restrict__ count, uint32_t abundance, uint8_t* __restrict__ flags) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_occ) return;
    const uint32_t h = slot_of[o];
    const bool usable = (count[h] & 255u) >= abundance;   // uint8_t count (SubSampler.h:24): 256 occurrences read as 0
    flags[o] = (usable ? 1u : 0u) | ((!usable && slot[h] == o + 1) ? 2u : 0u);
}

// Flags for every k-mer occurrence of the gathered super-k-mers (the buffers gather_superkmers_impl left on the
// device: ctx->i_compact / ctx->i_dst): malloc'd, one byte per occurrence, numbered super-k-mer by super-k-mer.
// SPSP_ERR_OVERFLOW when the occurrences do not fit 31 bits: the caller then lets the host count.
int abundance_flags_impl(spsp_ctx* ctx, const spsp_params* p, const spsp_superkmer* d_sk, uint64_t n_sk, uint8_t** h_flags, uint64_t* n_occ_out,
                         const uint32_t* h_seg_sk, uint32_t n_seg) {
    *h_flags = nullptr; *n_occ_out = 0;
    if (n_sk == 0) { *h_flags = (uint8_t*)malloc(1); return *h_flags ? SPSP_OK : SPSP_ERR_NOMEM; }
    // a super-k-mer holds at most k - m + 1 <= 63 k-mers: below this many of them the 32-bit prefix sums cannot wrap
    if (n_sk > 0x7ffffff0ull / 64) { set_error("too many super-k-mers for the device abundance pass"); return SPSP_ERR_OVERFLOW; }
    int rc;
    const uint32_t n = (uint32_t)n_sk;
    if ((rc = ctx->a_cnt.reserve((size_t)n * 4)) || (rc = ctx->a_off.reserve((size_t)(n + 1) * 4))) return rc;
    hipLaunchKernelGGL(k_abund_sizes, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_sk, n, p->k, ctx->a_cnt.as<uint32_t>());
    SPSP_HIP(hipGetLastError());
    if ((rc = launch_scan_u32(ctx, ctx->a_cnt.as<uint32_t>(), ctx->a_off.as<uint32_t>(), n, ctx->h_scalar + 7))) return rc;
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t n_occ = ctx->h_scalar[7];
    *n_occ_out = n_occ;
    uint8_t* out = (uint8_t*)malloc((size_t)n_occ + 1);
    if (!out) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    if (n_occ == 0) { *h_flags = out; return SPSP_OK; }
    uint64_t cap = 1024;
    while (cap < 2 * n_occ) cap <<= 1;
    if ((rc = ctx->a_mn.reserve((size_t)n_occ * 4)) || (rc = ctx->a_lo.reserve((size_t)n_occ * 8)) || (rc = ctx->a_hi.reserve((size_t)n_occ * 8)) ||
        (rc = ctx->a_slot.reserve((size_t)cap * 8)) || (rc = ctx->a_slot_of.reserve((size_t)n_occ * 4)) || (rc = ctx->a_flags.reserve((size_t)n_occ))) {
        free(out);
        return rc;
    }
    uint32_t* slot = ctx->a_slot.as<uint32_t>();
    uint32_t* count = slot + cap;
    hipError_t e = hipMemsetAsync(slot, 0, (size_t)cap * 8, ctx->stream);
    if (e != hipSuccess) { free(out); return hip_fail(e, "abundance table", __FILE__, __LINE__); }
    hipLaunchKernelGGL(k_abund_emit, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->i_compact.as<uint8_t>(), ctx->i_dst.as<uint32_t>(),
                       d_sk, ctx->a_off.as<uint32_t>(), n, p->k, ctx->a_mn.as<uint32_t>(), ctx->a_lo.as<uint64_t>(), ctx->a_hi.as<uint64_t>());
    const uint32_t blocks = (uint32_t)((n_occ + 255) / 256);
    uint32_t* d_seg_occ = nullptr;
    if (h_seg_sk && n_seg > 1) {
        if ((rc = ctx->a_seg.reserve((size_t)(n_seg + 1) * 8 + 64))) { free(out); return rc; }
        uint32_t* d_seg_sk = ctx->a_seg.as<uint32_t>();
        d_seg_occ = d_seg_sk + (n_seg + 1);
        e = hipMemcpyAsync(d_seg_sk, h_seg_sk, (size_t)(n_seg + 1) * 4, hipMemcpyHostToDevice, ctx->stream);     // (the caller's array outlives the wait below)
        if (e != hipSuccess) { free(out); return hip_fail(e, "abundance segments", __FILE__, __LINE__); }
        hipLaunchKernelGGL(k_abund_segments, dim3((n_seg + 256) / 256), dim3(256), 0, ctx->stream, (const uint32_t*)d_seg_sk, n_seg, (const uint32_t*)ctx->a_off.as<uint32_t>(), d_seg_occ);
    }
    hipLaunchKernelGGL(k_abund_insert, dim3(blocks), dim3(256), 0, ctx->stream, ctx->a_mn.as<uint32_t>(), ctx->a_lo.as<uint64_t>(),
                       ctx->a_hi.as<uint64_t>(), (uint32_t)n_occ, slot, count, (uint32_t)(cap - 1), ctx->a_slot_of.as<uint32_t>(),
                       (const uint32_t*)d_seg_occ, n_seg);
    hipLaunchKernelGGL(k_abund_flags, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t)n_occ, ctx->a_slot_of.as<uint32_t>(), slot, count,
                       p->abundance, ctx->a_flags.as<uint8_t>());
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, ctx->a_flags.p, (size_t)n_occ, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(out); return hip_fail(e, "abundance pass", __FILE__, __LINE__); }
    *h_flags = out;
    return SPSP_OK;
}

}  // namespace spsp
