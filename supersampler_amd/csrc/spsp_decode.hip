// spsp_decode.hip -- "next" row N2: bulk decode of sketch payloads on the GPU.
//
// What the reference's comparator does per bucket and file while it merges (strDecompressor utils.cpp:71-111,
// inject_minimizer Comparator.cpp:78-92, the k-mer walks :196-260, canonize(128 bit) utils.cpp:470-472) happens
// here for ALL sketches at once: the host only finds the structure of every payload (header, bucket boundaries,
// line ends: a memchr walk), the GPU rebuilds every super-k-mer (prefix + minimizer + suffix), rolls the k-mers
// and their reverse complements, sorts each sketch's canonical (minimizer, k-mer) keys and drops duplicates -- the
// arrays spsp_compare_device takes, already in HBM.
//
//   k_decode_emit     one lane per stored super-k-mer: 2-bit blob bases / ASCII lines -> canonical keys
//   k_decode_sort     one workgroup per sketch: bitonic sort in LDS by (minimizer, kmer_hi, kmer_lo), unique,
//                     distinct count
//   k_exclusive_scan  distinct counts -> offsets (spsp_scan.hip)
//   k_decode_compact  sketches back to back
//
// A sketch that does not fit the sort's LDS (more than 8192 keys, 4096 with k > 32) stays on the device: its raw keys go
// through the table in HBM and the merge sort of spsp_bigkeys.hip (the reference's color_map takes whatever a bucket
// holds, Comparator.cpp:186-260).  Only a sketch that is not laid out as the sketcher writes it (partial blob bytes,
// over-long lines, truncated) is decoded by spsp_sketch_parse_host and uploaded: same keys, already sorted.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

constexpr int kSortThreads = 1024;
constexpr uint32_t kSortCapLo = 8192, kSortCapHi = 4096;

__device__ __forceinline__ void rc128(uint64_t hi, uint64_t lo, uint32_t k, uint64_t* rhi, uint64_t* rlo) {
    // reverse complement of the k-mer in the low 2k bits of (hi:lo): reverse all 64 groups, then shift down
    const uint64_t a = rc_window64(lo), b = rc_window64(hi);       // (a:b) = reverse complement of the 64-base window
    const uint32_t sh = 128 - 2 * k;                               // >= 2
    if (sh >= 64) { *rhi = 0; *rlo = sh == 64 ? a : a >> (sh - 64); }
    else { *rhi = a >> sh; *rlo = (b >> sh) | (a << (64 - sh)); }
}

__global__ __launch_bounds__(256) void k_decode_emit(const uint8_t* __restrict__ text, const DecDesc* __restrict__ desc, uint32_t n_desc,
                                                    uint32_t k, uint32_t m, uint32_t* __restrict__ r_mn, uint64_t* __restrict__ r_lo,
                                                    uint64_t* __restrict__ r_hi) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_desc) return;
    const DecDesc D = desc[d];
    const uint32_t kind = D.info & 3u;
    const uint32_t side = k - m;                                   // maximal: k - m stored bases on either side of the minimizer
    uint32_t n_pre, n_suf;
    if (kind == 0) { n_pre = side; n_suf = side; }
    else if (kind == 1) { n_pre = (D.info >> 2) & 0xffu; n_suf = (D.info >> 10) & 0xffu; }
    else { n_pre = 0; n_suf = 0; }
    const uint32_t total = n_pre + m + n_suf;
    const uint8_t* src = text + D.off;
    uint64_t hi = 0, lo = 0;
    const uint64_t mask_hi = k > 32 ? (k == 64 ? ~0ull : ((1ull << (2 * k - 64)) - 1)) : 0ull;
    const uint64_t mask_lo = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    uint32_t made = 0;
    for (uint32_t t = 0; t < total; ++t) {
        uint32_t c;
        if (t < n_pre) c = kind == 0 ? (src[t >> 2] >> (6 - 2 * (t & 3))) & 3u : ((uint32_t)src[t] >> 1) & 3u;
        else if (t < n_pre + m) c = (D.mn >> (2 * (m - 1 - (t - n_pre)))) & 3u;
        else {
            const uint32_t u = t - n_pre - m;
            c = kind == 0 ? (src[(n_pre + u) >> 2] >> (6 - 2 * ((n_pre + u) & 3))) & 3u : ((uint32_t)src[n_pre + 1 + u] >> 1) & 3u;
        }
        hi = ((hi << 2) | (lo >> 62)) & mask_hi;
        lo = ((lo << 2) | c) & mask_lo;
        if (t + 1 < k) continue;
        uint64_t rh, rl;
        rc128(hi, lo, k, &rh, &rl);
        const bool fwd = hi != rh ? hi < rh : lo <= rl;
        const uint32_t o = D.out + made;
        r_mn[o] = D.mn; r_lo[o] = fwd ? lo : rl;
        if (r_hi) r_hi[o] = fwd ? hi : rh;
        ++made;
    }
}

// The structure of every payload -- header, bucket boundaries, line ends: sketch_parse_structure_host's walk -- on the device,
// one lane per sketch (round 5).  At 10 000 sketch files the host walk (2.4 x 10^6 descriptors made, copied and uploaded:
// 58 MB) was 60 of the 66 ms in front of a 6 ms decode; the text crosses PCIe anyway, a lane reads its 3 KB payload from its
// CU's vector cache and blobs are stepped over unread.  COUNT pass: descriptors and raw keys per sketch (the host needs
// the totals to size the arrays) and a flag for anything the sketcher does not write -- such a sketch is decoded on the host
// as before, and it is the host path that words the error for a malformed one.  WRITE pass: the same walk writes DecDesc.
// Every read is bounded by the sketch's length and every descriptor names bytes inside it (the conditions are the host
// parser's, line for line).
struct DecCount { uint32_t n_desc, n_keys, flags, pad; };
template <bool WRITE>
__global__ __launch_bounds__(64) void k_decode_parse(const uint8_t* __restrict__ text, const uint64_t* __restrict__ text_off, const uint64_t* __restrict__ lens,
                                                    uint32_t n, uint32_t k, uint32_t m, const uint32_t* __restrict__ extra, DecCount* __restrict__ counts,
                                                    const uint64_t* __restrict__ desc_off, const uint64_t* __restrict__ raw_off,
                                                    const uint8_t* __restrict__ skip, DecDesc* __restrict__ desc) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (WRITE && skip[i]) return;                                  // decoded on the host: no descriptors
    const uint8_t* p = text + text_off[i];
    const uint64_t len = lens[i], t0 = text_off[i];
    uint32_t n_desc = 0, flags = 0;
    uint64_t keys = 0;
    const uint64_t d0 = WRITE ? desc_off[i] : 0ull, r0 = WRITE ? raw_off[i] : 0ull;
    auto push = [&](uint64_t off, uint32_t mn, uint32_t info, uint64_t count) {
        if (count == 0) return;
        if (keys + count > 0xfffffff0ull) { flags |= 1u; return; }
        if (WRITE) desc[d0 + n_desc] = DecDesc{t0 + off, mn, info, (uint32_t)(r0 + keys), 0};
        ++n_desc; keys += count;
    };
    // header: "<2k - m> <m> ..." -- anything but two plain decimal numbers that give this call's k and m goes to the host path
    uint64_t pos = 0;
    while (pos < len && p[pos] != '\n') ++pos;
    bool ok = pos < len;
    if (ok) {
        uint64_t q = 0, skm = 0, mm = 0;
        uint32_t digits = 0;
        while (q < pos && p[q] >= '0' && p[q] <= '9' && digits < 6) { skm = skm * 10 + (p[q] - '0'); ++q; ++digits; }
        ok = digits > 0 && digits < 6 && q < pos && p[q] == ' ';
        while (q < pos && p[q] == ' ') ++q;
        digits = 0;
        while (q < pos && p[q] >= '0' && p[q] <= '9' && digits < 6) { mm = mm * 10 + (p[q] - '0'); ++q; ++digits; }
        ok = ok && digits > 0 && digits < 6 && (q == pos || p[q] == ' ');
        ok = ok && mm == m && skm == 2ull * k - m;                 // (the sketcher's header: k - 1 + k - m + 1, SubSampler.cpp:459)
    }
    if (!ok) flags |= 1u;
    else {
        const uint32_t half = k - m;
        if (half > 0 && (2 * half) % 4 != 0) flags |= 1u;          // k - m odd: a blob byte straddles two super-k-mers -- host decoder
        pos += 1;
        while (pos + m <= len) {
            uint32_t mn = 0;
            for (uint32_t j = 0; j < m; ++j) mn = (mn << 2) | (((uint32_t)p[pos + j] >> 1) & 3u);
            pos += m;
            if (pos + 4 > len) break;
            const uint32_t nbytes = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
            pos += 4;
            if (pos + nbytes > len) { flags |= 1u; break; }        // (the host path reports it)
            uint64_t seq_len = 0;
            if (nbytes) { if (p[pos] != 0) flags |= 1u; seq_len = (uint64_t)(nbytes - 1) * 4; }
            if (half > 0) { for (uint64_t x = 0; (x + 1) * 2 * half <= seq_len; ++x) push(pos + 1 + x * (half / 2), mn, 0u, k - m + 1); }
            else if (seq_len == 0) push(pos, mn, 2u, 1);
            pos += nbytes;
            for (;;) {                                             // "prefix\nsuffix\n" until an empty pair
                if (pos >= len) break;
                uint64_t e1 = pos;
                while (e1 < len && p[e1] != '\n') ++e1;
                const bool has1 = e1 < len;
                const uint64_t s1 = pos, l1 = e1 - pos;
                pos = has1 ? e1 + 1 : len;
                uint64_t e2 = pos;
                while (e2 < len && p[e2] != '\n') ++e2;
                const bool has2 = pos < len && e2 < len;
                const uint64_t s2 = pos, l2 = pos < len ? e2 - pos : 0;
                pos = pos < len ? (has2 ? e2 + 1 : len) : len;
                if (l1 == 0 && l2 == 0) break;
                if (!has1 || l1 > 255 || l2 > 255 || s2 != s1 + l1 + 1) { flags |= 1u; continue; }
                const uint64_t total = l1 + m + l2;
                push(s1, mn, 1u | ((uint32_t)l1 << 2) | ((uint32_t)l2 << 10), total >= k ? total - k + 1 : 0);
            }
        }
    }
    if (extra && extra[i] != 0xffffffffu && !(flags & 1u)) {       // the phantom key of the comparator's first-read rule, behind the sketch's own
        if (WRITE) desc[d0 + n_desc] = DecDesc{t0, extra[i], 2u, (uint32_t)(r0 + keys), 0};
        ++n_desc; keys += 1;
    }
    if (!WRITE) counts[i] = DecCount{n_desc, (uint32_t)keys, flags, 0};
}

// sort + unique of one sketch's raw keys [raw_off[s], raw_off[s] + raw_cnt[s]) in LDS; sketches with presorted[s]
// (decoded by the host) are only counted
template <bool HAS_HI>
__global__ __launch_bounds__(kSortThreads) void k_decode_sort(uint32_t* __restrict__ r_mn, uint64_t* __restrict__ r_lo,
                                                             uint64_t* __restrict__ r_hi, const uint64_t* __restrict__ raw_off,
                                                             const uint32_t* __restrict__ raw_cnt, const uint8_t* __restrict__ presorted,
                                                             const uint32_t* __restrict__ big, uint32_t* __restrict__ distinct, uint32_t force_network,
                                                             uint32_t cap_rt, uint32_t second) {
    // cap_rt: keys the LDS arrays of THIS launch hold.  The first launch is sized for the largest sketch as it is (counting inside
    // buckets needs no padding: 4 800 keys are 58 KB, two workgroups a CU, where the network's power of two was 96 KB and one);
    // a sketch that needs the network and whose power of two does not fit marks itself (kNeedsNetwork) and is sorted by the
    // second launch, which has the full arrays and passes every other sketch by.
    constexpr uint32_t kNeedsNetwork = 0xffffffffu;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_s[];
    constexpr uint32_t CAP = HAS_HI ? kSortCapHi : kSortCapLo;
    uint64_t* s_lo = reinterpret_cast<uint64_t*>(lds_s);
    uint64_t* s_hi = s_lo + cap_rt;                                // (HAS_HI only)
    uint32_t* s_mn = reinterpret_cast<uint32_t*>(s_hi + (HAS_HI ? cap_rt : 0));
    __shared__ uint32_t wave_sum[kSortThreads / 64];
    const uint32_t s = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const uint32_t n = raw_cnt[s];
    const uint64_t r0 = raw_off[s];
    if (second) { if (distinct[s] != kNeedsNetwork) return; force_network = 1; }
    else {
        if (presorted[s]) { if (t == 0) distinct[s] = n; return; }
        if (big[s]) { if (t == 0) distinct[s] = 0; return; }       // beyond this workgroup's LDS: k_big_insert / k_big_emit count it
        if (n == 0) { if (t == 0) distinct[s] = 0; return; }
    }
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    if (force_network && n2 > cap_rt) { if (t == 0) distinct[s] = kNeedsNetwork; return; }    // (the first launch with the network pinned: test hook)
    const uint32_t fill = n2 <= cap_rt ? n2 : n;                   // (the padding only where the network could run)
    for (uint32_t i = t; i < fill; i += kSortThreads) {
        if (i < n) { s_mn[i] = r_mn[r0 + i]; s_lo[i] = r_lo[r0 + i]; if (HAS_HI) s_hi[i] = r_hi[r0 + i]; }
        else { s_mn[i] = 0xffffffffu; s_lo[i] = ~0ull; if (HAS_HI) s_hi[i] = ~0ull; }
    }
    __syncthreads();
    // A sketch as the sketcher writes it has its buckets in ascending minimizer order (the reference walks a std::map,
    // SubSampler.cpp:458-504): the raw keys are sorted by minimizer already and only a bucket's k-mers -- tens of them -- are
    // out of order.  Then every key finds its place by COUNTING the smaller keys of its bucket (no exchange network, two
    // barriers): 4.4 -> ~0.5 ms for the 10 000 sketches of configs[3].  A payload whose minimizers do not ascend, or with a
    // bucket of more than kBucketMax keys (a low-complexity genome), takes the bitonic network below.
    constexpr uint32_t kBucketMax = 256, PERK = CAP / kSortThreads;
    __shared__ uint32_t s_slow;
    if (t == 0) s_slow = force_network;
    __syncthreads();
    if (!force_network) {
        // lane t takes keys t, t + 1024, ...: the lanes of a wave hold 64 NEIGHBOURING keys -- mostly one or two buckets, so they loop
        // over the same bounds and read the same LDS words (a lane with eight consecutive keys made every wave wait for its longest
        // bucket eight times over: the pass was bound by its compare loop, 53 us a workgroup whatever ran beside it).  A key's bucket
        // bounds are two binary searches over the (ascending) minimizers.
        uint32_t my_rank[PERK];
        uint64_t my_lo[PERK], my_hi[PERK];
        bool slow = false;
#pragma unroll
        for (uint32_t u = 0; u < PERK; ++u) {
            const uint32_t i = t + u * kSortThreads;
            my_rank[u] = 0xffffffffu;
            if (i >= n) continue;
            const uint32_t mn = s_mn[i];
            if (i && s_mn[i - 1] > mn) slow = true;
            uint32_t b0, b1;
            {
                uint32_t lo_ = 0, hi_ = n;
                while (lo_ < hi_) { const uint32_t mid = (lo_ + hi_) >> 1; if (s_mn[mid] < mn) lo_ = mid + 1; else hi_ = mid; }
                b0 = lo_; hi_ = n;
                while (lo_ < hi_) { const uint32_t mid = (lo_ + hi_) >> 1; if (s_mn[mid] <= mn) lo_ = mid + 1; else hi_ = mid; }
                b1 = lo_;
            }
            // (minimizers that do not ascend make the searches meaningless: the bounds are clamped around i so that the loop below
            // stays short and in range -- that sketch takes the network anyway)
            if (b0 > i || b1 <= i || b1 - b0 > kBucketMax) { slow = true; continue; }
            const uint64_t lo = s_lo[i], hi = HAS_HI ? s_hi[i] : 0ull;
            uint32_t smaller = 0;
            for (uint32_t j = b0; j < b1; ++j) {
                const uint64_t lj = s_lo[j], hj = HAS_HI ? s_hi[j] : 0ull;
                const bool less = HAS_HI ? (hj < hi || (hj == hi && (lj < lo || (lj == lo && j < i)))) : (lj < lo || (lj == lo && j < i));
                smaller += less ? 1u : 0u;
            }
            my_rank[u] = b0 + smaller; my_lo[u] = lo; my_hi[u] = hi;
        }
        if (slow) s_slow = 1;
        __syncthreads();
        const bool any_slow = s_slow != 0;
        if (!any_slow) {
#pragma unroll
            for (uint32_t u = 0; u < PERK; ++u)
                if (my_rank[u] != 0xffffffffu) { s_lo[my_rank[u]] = my_lo[u]; if (HAS_HI) s_hi[my_rank[u]] = my_hi[u]; }
        }
        __syncthreads();
    }
    if (s_slow && n2 > cap_rt) { if (t == 0) distinct[s] = kNeedsNetwork; return; }          // (the second launch's)
    if (s_slow) {
    auto greater = [&](uint32_t a, uint32_t b) {
        if (s_mn[a] != s_mn[b]) return s_mn[a] > s_mn[b];
        if (HAS_HI && s_hi[a] != s_hi[b]) return s_hi[a] > s_hi[b];
        return s_lo[a] > s_lo[b];
    };
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t idx = t; idx < (n2 >> 1); idx += kSortThreads) {
                const uint32_t i = ((idx & ~(stride - 1u)) << 1) | (idx & (stride - 1u)), j = i + stride;   // (stride is a power of two)
                const bool asc = (i & size) == 0;
                if (greater(i, j) == asc) {
                    const uint32_t tm = s_mn[i]; s_mn[i] = s_mn[j]; s_mn[j] = tm;
                    const uint64_t tl = s_lo[i]; s_lo[i] = s_lo[j]; s_lo[j] = tl;
                    if (HAS_HI) { const uint64_t th = s_hi[i]; s_hi[i] = s_hi[j]; s_hi[j] = th; }
                }
            }
            __syncthreads();
        }
    }
    }
    // unique: first of every run of equal keys, ranks by a workgroup prefix sum (CAP / kSortThreads consecutive elements per lane)
    constexpr uint32_t PER = CAP / kSortThreads;
    uint32_t first[PER];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = t * PER + u;
        first[u] = 0;
        if (i < n) first[u] = (i == 0 || s_mn[i] != s_mn[i - 1] || s_lo[i] != s_lo[i - 1] || (HAS_HI && s_hi[i] != s_hi[i - 1])) ? 1u : 0u;
        cnt += first[u];
    }
    uint32_t x = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= (uint32_t)d) x += y; }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    for (uint32_t w = 0; w < kSortThreads / 64; ++w) { if (w < wid) pre += wave_sum[w]; all += wave_sum[w]; }
    uint32_t rank = pre + x - cnt;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t i = t * PER + u;
        if (!first[u]) continue;
        r_mn[r0 + rank] = s_mn[i]; r_lo[r0 + rank] = s_lo[i];
        if (HAS_HI) r_hi[r0 + rank] = s_hi[i];
        ++rank;
    }
    if (t == 0) distinct[s] = all;
}

__global__ __launch_bounds__(256) void k_decode_compact(const uint32_t* __restrict__ r_mn, const uint64_t* __restrict__ r_lo,
                                                       const uint64_t* __restrict__ r_hi, const uint32_t* __restrict__ b_mn,
                                                       const uint64_t* __restrict__ b_lo, const uint64_t* __restrict__ b_hi,
                                                       const uint64_t* __restrict__ raw_off, const uint32_t* __restrict__ big,
                                                       const uint32_t* __restrict__ distinct, const uint32_t* __restrict__ out_off,
                                                       uint32_t* __restrict__ o_mn, uint64_t* __restrict__ o_lo, uint64_t* __restrict__ o_hi) {
    const uint32_t s = blockIdx.y;
    const uint32_t n = distinct[s];
    const uint64_t r0 = raw_off[s];
    const uint32_t o0 = out_off[s];
    const bool from_b = big[s] != 0;                               // (a sketch the global-memory stages took: its keys are in their output slices)
    const uint32_t* s_mn = from_b ? b_mn : r_mn;
    const uint64_t* s_lo = from_b ? b_lo : r_lo;
    const uint64_t* s_hi = from_b ? b_hi : r_hi;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        o_mn[o0 + e] = s_mn[r0 + e]; o_lo[o0 + e] = s_lo[r0 + e];
        if (o_hi) o_hi[o0 + e] = s_hi[r0 + e];
    }
}

// ------------------------------------------------------------------ host side --
// payloads (gunzipped sketch files, host) -> context-owned device key arrays (c_min / c_lo / c_hi: the buffers
// spsp_compare's host form uploads into) + host offsets.  extra[i] (optional): one more key for sketch i, given as a
// bare minimizer -- the phantom key of the comparator's first-read rule (spsp_sketch_chain_host)
int sketch_decode_device_impl(spsp_ctx* ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n,
                              const int* extra_has, const uint32_t* extra_mn, uint32_t* k_out, uint32_t* m_out, uint64_t* sk_off) {
    int rc = SPSP_OK;
    static const bool dbg_times = getenv("SPSP_DEBUG_DECODE_TIMES") != nullptr;
    double tm[6] = {now_s(), 0, 0, 0, 0, 0};
    std::vector<ParsedSketch> P(n);
    std::vector<int> rcs(n, SPSP_OK);
    std::vector<std::string> errs(n);
    // sketches that are not laid out as the sketcher writes them are decoded on the host (sorted, distinct) and uploaded
    // in place: by the SAME worker pool that walks the payload structures, not one after the other on the calling thread.
    // A sketch of any SIZE stays on the device (big[]: more raw keys than the LDS sort holds)
    struct HostKeys { uint32_t* mn = nullptr; uint64_t *lo = nullptr, *hi = nullptr; uint64_t n = 0; };
    std::vector<HostKeys> hk(n);
    std::vector<uint8_t> presorted(n, 0);
    std::vector<uint32_t> big(n, 0);
    auto free_hk = [&]() { for (auto& h : hk) { free(h.mn); free(h.lo); free(h.hi); } };
    // Many sketches: the structure walk runs on the DEVICE (k_decode_parse) -- the payloads go up first, the walk's counts come
    // back (one host wait), flagged sketches take the host decoder, and the descriptors are written where they are used.
    // SPSP_DEBUG_DECODE_WALK=host / device pins the choice (tests run both on the same files).
    static const char* dbg_walk = getenv("SPSP_DEBUG_DECODE_WALK");
    const bool dev_walk = dbg_walk ? dbg_walk[0] == 'd' : n >= 256;
    std::vector<DecCount> counts;
    std::vector<uint64_t> text_off_dev;
    uint64_t n_desc_dev = 0;
    if (dev_walk && n) {
        // k and m: the first sketch's header, by the host parser's rules (every other header is held against them by the walk)
        {
            const uint8_t* nl = (payloads[0] && lens[0]) ? (const uint8_t*)memchr(payloads[0], '\n', lens[0]) : nullptr;
            ParsedSketch H;
            if (!nl) { set_error("sketch has no header line"); return SPSP_ERR_FORMAT; }
            if ((rc = sketch_parse_structure_host(payloads[0], (uint64_t)(nl - payloads[0]) + 1, &H))) return rc;   // (the header line alone)
            for (uint32_t i = 0; i < n; ++i) { P[i].k = H.k; P[i].m = H.m; }
        }
        const uint32_t k = P[0].k, m = P[0].m;
        text_off_dev.assign((size_t)n + 1, 0);
        for (uint32_t i = 0; i < n; ++i) text_off_dev[i + 1] = text_off_dev[i] + ((lens[i] + 15) & ~15ull);
        const uint64_t T = text_off_dev[n];
        if ((rc = ctx->dc_text.reserve((size_t)T + 64)) || (rc = ctx->dc_walk.reserve((size_t)n * (8 + 8 + 4 + 16 + 8 + 8 + 1) + 256))) return rc;
        // one pinned-size staging buffer for everything that goes up: texts back to back, then lens
        // (malloc, not a vector: 30 MB of zeroed pages first touched by ONE thread were 10 ms; the gaps between sketches are never read)
        // ... unless the caller's payloads already lie so, in a few pieces (spsp_compare_files reads its files into a region per reader
        // thread, each payload at the 16-byte-rounded end of the one before): then every piece goes up from where it is
        struct Run { const uint8_t* host; uint64_t dev, bytes; };
        std::vector<Run> runs;
        {
            static const bool allow = getenv("SPSP_DEBUG_DECODE_GATHER") == nullptr;
            bool ok = allow;
            for (uint32_t i = 0; i < n && ok; ++i) {
                if (!lens[i]) continue;
                if (!runs.empty() && payloads[i] == runs.back().host + (text_off_dev[i] - runs.back().dev)) runs.back().bytes = text_off_dev[i] + lens[i] - runs.back().dev;
                else { runs.push_back(Run{payloads[i], text_off_dev[i], lens[i]}); ok = runs.size() <= 64; }
            }
            if (!ok) runs.clear();
        }
        const bool laid_out = !runs.empty();
        struct Free { void operator()(uint8_t* q) const { free(q); } };
        std::unique_ptr<uint8_t, Free> text_buf(laid_out ? nullptr : (uint8_t*)malloc((size_t)T + 64));
        if (!laid_out && !text_buf) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
        if (!laid_out) {
            unsigned workers = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
            if (T < (1u << 20)) workers = 1;
            std::atomic<uint32_t> next(0);
            auto work = [&]() {
                for (;;) {
                    const uint32_t i0 = next.fetch_add(64);
                    if (i0 >= n) break;
                    for (uint32_t i = i0; i < std::min(n, i0 + 64); ++i) if (lens[i]) memcpy(text_buf.get() + text_off_dev[i], payloads[i], (size_t)lens[i]);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            runs.push_back(Run{text_buf.get(), 0, T});
        }
        const double tw0 = now_s();
        uint64_t* d_toff = ctx->dc_walk.as<uint64_t>();
        uint64_t* d_lens = d_toff + n;
        uint64_t* d_doff = d_lens + n;
        uint64_t* d_roff = d_doff + n;
        DecCount* d_counts = reinterpret_cast<DecCount*>(d_roff + n);
        uint32_t* d_extra = reinterpret_cast<uint32_t*>(d_counts + n);
        std::vector<uint32_t> extra((size_t)n, 0xffffffffu);
        bool any_extra = false;
        if (extra_has) for (uint32_t i = 0; i < n; ++i) if (extra_has[i]) { extra[i] = extra_mn[i]; any_extra = true; }
        hipError_t e = hipSuccess;
        for (const Run& r : runs)
            if (e == hipSuccess && r.bytes) e = hipMemcpyAsync(ctx->dc_text.as<uint8_t>() + r.dev, r.host, (size_t)r.bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_toff, text_off_dev.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_lens, lens, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && any_extra) e = hipMemcpyAsync(d_extra, extra.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return hip_fail(e, "sketch upload", __FILE__, __LINE__);
        hipLaunchKernelGGL(k_decode_parse<false>, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->dc_text.as<uint8_t>(), (const uint64_t*)d_toff, (const uint64_t*)d_lens, n, k, m,
                           any_extra ? (const uint32_t*)d_extra : (const uint32_t*)nullptr, d_counts, (const uint64_t*)nullptr, (const uint64_t*)nullptr,
                           (const uint8_t*)nullptr, (DecDesc*)nullptr);
        counts.resize(n);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, (size_t)n * sizeof(DecCount), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return hip_fail(e, "sketch structure walk", __FILE__, __LINE__);
        tm[1] = now_s();
        if (dbg_times) fprintf(stderr, "[spsp decode] device walk: texts assembled %.1f ms, upload + count pass + wait %.1f ms\n", (tw0 - tm[0]) * 1e3, (tm[1] - tw0) * 1e3);
        // flagged sketches: the host decoder (and the host's wording of what is wrong with a malformed one)
        std::vector<uint32_t> odd;
        for (uint32_t i = 0; i < n; ++i) if (counts[i].flags & 1u) odd.push_back(i);
        {
            unsigned workers = std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
            if (workers > odd.size()) workers = odd.empty() ? 1 : (unsigned)odd.size();
            std::atomic<uint32_t> next(0);
            auto work = [&]() {
                for (;;) {
                    const uint32_t x = next.fetch_add(1);
                    if (x >= odd.size()) break;
                    const uint32_t i = odd[x];
                    ParsedSketch Q;
                    rcs[i] = sketch_parse_structure_host(payloads[i], lens[i], &Q);          // (its k and m, for the mismatch message below)
                    if (rcs[i]) { errs[i] = spsp_last_error(); continue; }
                    P[i].k = Q.k; P[i].m = Q.m;
                    uint32_t kk, mm2;
                    rcs[i] = spsp_sketch_parse_host(payloads[i], lens[i], &kk, &mm2, &hk[i].mn, &hk[i].lo, &hk[i].hi, &hk[i].n);
                    if (rcs[i]) { errs[i] = spsp_last_error(); continue; }
                    presorted[i] = 1;
                }
            };
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (presorted[i] || rcs[i]) continue;
            P[i].n_keys = counts[i].n_keys - ((extra_has && extra_has[i]) ? 1u : 0u);
            const uint32_t cap_i = k > 32 ? kSortCapHi : kSortCapLo;
            big[i] = counts[i].n_keys > cap_i ? 1u : 0u;
        }
    } else {
        unsigned workers = std::thread::hardware_concurrency();
        if (workers == 0) workers = 1;
        if (workers > 16) workers = 16;
        if (workers > n) workers = n ? n : 1;
        std::atomic<uint32_t> next(0);
        auto work = [&]() {
            for (;;) {
                const uint32_t i = next.fetch_add(1);
                if (i >= n) break;
                rcs[i] = sketch_parse_structure_host(payloads[i], lens[i], &P[i]);
                if (rcs[i]) { errs[i] = spsp_last_error(); continue; }
                const uint32_t cap_i = P[i].k > 32 ? kSortCapHi : kSortCapLo;
                const uint64_t keys = P[i].n_keys + ((extra_has && extra_has[i]) ? 1 : 0);
                if (P[i].standard) { big[i] = keys > cap_i ? 1u : 0u; continue; }
                uint32_t kk, mm2;
                rcs[i] = spsp_sketch_parse_host(payloads[i], lens[i], &kk, &mm2, &hk[i].mn, &hk[i].lo, &hk[i].hi, &hk[i].n);
                if (rcs[i]) { errs[i] = spsp_last_error(); continue; }
                presorted[i] = 1;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    if (!(dev_walk && n)) tm[1] = now_s();
    for (uint32_t i = 0; i < n; ++i) if (rcs[i]) { free_hk(); set_error("%s", errs[i].c_str()); return rcs[i]; }
    for (uint32_t i = 1; i < n; ++i)
        if (P[i].k != P[0].k || P[i].m != P[0].m) { free_hk(); set_error("sketch %u was made with k=%u m=%u, expected k=%u m=%u", i, P[i].k, P[i].m, P[0].k, P[0].m); return SPSP_ERR_FORMAT; }
    const uint32_t k = n ? P[0].k : 0, m = n ? P[0].m : 0;
    *k_out = k; *m_out = m;
    sk_off[0] = 0;
    if (n == 0) return SPSP_OK;
    const bool has_hi = k > 32;
    std::vector<uint64_t> raw_off(n + 1, 0), text_off(n + 1, 0);
    std::vector<uint32_t> raw_cnt(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t keys = P[i].n_keys + ((extra_has && extra_has[i]) ? 1 : 0);
        if (presorted[i]) {
            if (extra_has && extra_has[i] && hk[i].n == 0) {       // (the phantom only ever joins an empty sketch)
                hk[i].mn[0] = extra_mn[i];
                // k == m: the k-mer is the minimizer; canonical = min(value, reverse complement)
                uint64_t v = extra_mn[i], r = 0;
                for (uint32_t j = 0; j < m; ++j) r |= (uint64_t)(((v >> (2 * j)) & 3u) ^ 2u) << (2 * (m - 1 - j));
                hk[i].lo[0] = v < r ? v : r; hk[i].hi[0] = 0; hk[i].n = 1;
            }
            keys = hk[i].n;
            P[i].desc.clear();
        }
        if (keys > 0xfffffff0ull) { free_hk(); set_error("too many sketch k-mers for one call"); return SPSP_ERR_OVERFLOW; }
        raw_cnt[i] = (uint32_t)keys;
        raw_off[i + 1] = raw_off[i] + keys;
        text_off[i + 1] = text_off[i] + (presorted[i] ? 0 : ((lens[i] + 15) & ~15ull));
    }
    const uint64_t R = raw_off[n];
    if (R > 0xfffffff0ull) { free_hk(); set_error("too many sketch k-mers for one call"); return SPSP_ERR_OVERFLOW; }
    // descriptors with absolute offsets
    // (every sketch's descriptors have their place: written by a few threads -- 2.4 x 10^6 of them at 10 000 sketches --
    // together with the sketch's text into the one buffer that crosses PCIe)
    std::vector<DecDesc> desc;
    std::vector<uint8_t> text_all;
    if (dev_walk) {
        // the descriptors are written on the device: where each sketch's start, and where its raw keys go
        const double td0 = now_s();
        std::vector<uint64_t> doff((size_t)n, 0);
        for (uint32_t i = 0; i < n; ++i) { doff[i] = n_desc_dev; if (!presorted[i]) n_desc_dev += counts[i].n_desc; }
        if ((rc = ctx->dc_desc.reserve((size_t)n_desc_dev * sizeof(DecDesc) + 64))) { free_hk(); return rc; }
        const double td1 = now_s();
        uint64_t* d_toff = ctx->dc_walk.as<uint64_t>();
        uint64_t* d_lens = d_toff + n;
        uint64_t* d_doff = d_lens + n;
        uint64_t* d_roff = d_doff + n;
        DecCount* d_counts = reinterpret_cast<DecCount*>(d_roff + n);
        uint32_t* d_extra = reinterpret_cast<uint32_t*>(d_counts + n);
        uint8_t* d_skip = reinterpret_cast<uint8_t*>(d_extra + n);
        bool any_extra = false;
        if (extra_has) for (uint32_t i = 0; i < n; ++i) any_extra |= extra_has[i] != 0;
        hipError_t e = hipMemcpyAsync(d_doff, doff.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_roff, raw_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_skip, presorted.data(), (size_t)n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // (doff is a local: the copies have read it)
        if (e != hipSuccess) { free_hk(); return hip_fail(e, "sketch upload", __FILE__, __LINE__); }
        if (n_desc_dev) {
            hipLaunchKernelGGL(k_decode_parse<true>, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->dc_text.as<uint8_t>(), (const uint64_t*)d_toff, (const uint64_t*)d_lens, n, k, m,
                               any_extra ? (const uint32_t*)d_extra : (const uint32_t*)nullptr, (DecCount*)nullptr, (const uint64_t*)d_doff, (const uint64_t*)d_roff,
                               (const uint8_t*)d_skip, ctx->dc_desc.as<DecDesc>());
            SPSP_HIP(hipGetLastError());
        }
        if (dbg_times) fprintf(stderr, "[spsp decode] device walk, second pass: since the count pass %.2f ms, offsets + reserve %.2f ms, copies + wait + launch %.2f ms\n",
                               (td0 - tm[1]) * 1e3, (td1 - td0) * 1e3, (now_s() - td1) * 1e3);
    } else {
        std::vector<size_t> d_at((size_t)n + 1, 0);
        for (uint32_t i = 0; i < n; ++i) d_at[i + 1] = d_at[i] + (presorted[i] ? 0 : P[i].desc.size() + ((extra_has && extra_has[i]) ? 1 : 0));
        desc.resize(d_at[n]);
        const bool one_copy = n > 8 && text_off[n] > 0;
        if (one_copy) text_all.resize((size_t)text_off[n]);
        unsigned workers = std::thread::hardware_concurrency();
        if (workers == 0) workers = 1;
        if (workers > 16) workers = 16;
        if (d_at[n] < (1u << 16)) workers = 1;
        std::atomic<uint32_t> next(0);
        auto work = [&]() {
            for (;;) {
                const uint32_t i0 = next.fetch_add(64);
                if (i0 >= n) break;
                for (uint32_t i = i0; i < std::min(n, i0 + 64); ++i) {
                    if (presorted[i]) continue;
                    size_t at = d_at[i];
                    for (DecDesc d : P[i].desc) { d.off += text_off[i]; d.out += (uint32_t)raw_off[i]; desc[at++] = d; }
                    if (extra_has && extra_has[i]) desc[at++] = DecDesc{text_off[i], extra_mn[i], 2u, (uint32_t)(raw_off[i] + P[i].n_keys), 0};
                    if (one_copy && lens[i]) memcpy(text_all.data() + text_off[i], payloads[i], (size_t)lens[i]);
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    tm[2] = now_s();
    auto fail = [&](int r) { free_hk(); return r; };
    if (!dev_walk && (rc = ctx->dc_text.reserve((size_t)text_off[n] + 64))) return fail(rc);
    if (!dev_walk && (rc = ctx->dc_desc.reserve(desc.size() * sizeof(DecDesc) + 64))) return fail(rc);
    if ((rc = ctx->dc_mn.reserve((size_t)R * 4 + 64))) return fail(rc);
    if ((rc = ctx->dc_lo.reserve((size_t)R * 8 + 64))) return fail(rc);
    if (has_hi && (rc = ctx->dc_hi.reserve((size_t)R * 8 + 64))) return fail(rc);
    if ((rc = ctx->dc_meta.reserve((size_t)(n + 1) * 8 + (size_t)n * 4 * 3 + (size_t)n + 64))) return fail(rc);
    if ((rc = ctx->b_seg.reserve((size_t)n * 4 * 2 + 64))) return fail(rc);
    uint32_t any_big = 0;
    for (uint32_t i = 0; i < n; ++i) any_big |= big[i];
    if (any_big && ((rc = ctx->b_mn.reserve((size_t)R * 4 + 64)) || (rc = ctx->b_lo.reserve((size_t)R * 8 + 64)) ||
                    (has_hi && (rc = ctx->b_hi.reserve((size_t)R * 8 + 64))))) return fail(rc);
    uint32_t* d_first32 = ctx->b_seg.as<uint32_t>();
    uint32_t* d_big = d_first32 + n;
    std::vector<uint32_t> first32(n);
    for (uint32_t i = 0; i < n; ++i) first32[i] = (uint32_t)raw_off[i];
    if ((rc = ctx->c_min.reserve((size_t)R * 4 + 64))) return fail(rc);
    if ((rc = ctx->c_lo.reserve((size_t)R * 8 + 64))) return fail(rc);
    if (has_hi && (rc = ctx->c_hi.reserve((size_t)R * 8 + 64))) return fail(rc);
    uint8_t* d_text = ctx->dc_text.as<uint8_t>();
    uint64_t* d_raw_off = ctx->dc_meta.as<uint64_t>();
    uint32_t* d_raw_cnt = reinterpret_cast<uint32_t*>(d_raw_off + n + 1);
    uint32_t* d_distinct = d_raw_cnt + n;
    uint32_t* d_out_off = d_distinct + n;            // n + 1 entries follow... (scan writes n + 1)
    uint8_t* d_presorted = reinterpret_cast<uint8_t*>(d_out_off + n + 1);
    hipError_t e = hipSuccess;
    // the payloads cross in ONE copy (10 000 sketch files: ten thousand pageable copies of 3 KB each were 0.1 s of a 0.16 s stage)
    if (!text_all.empty()) e = hipMemcpyAsync(d_text, text_all.data(), text_all.size(), hipMemcpyHostToDevice, ctx->stream);
    for (uint32_t i = 0; i < n && e == hipSuccess; ++i) {
        if (!presorted[i]) { if (!dev_walk && lens[i] && text_all.empty()) e = hipMemcpyAsync(d_text + text_off[i], payloads[i], (size_t)lens[i], hipMemcpyHostToDevice, ctx->stream); }
        else if (hk[i].n) {
            e = hipMemcpyAsync(ctx->dc_mn.as<uint32_t>() + raw_off[i], hk[i].mn, hk[i].n * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ctx->dc_lo.as<uint64_t>() + raw_off[i], hk[i].lo, hk[i].n * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && has_hi) e = hipMemcpyAsync(ctx->dc_hi.as<uint64_t>() + raw_off[i], hk[i].hi, hk[i].n * 8, hipMemcpyHostToDevice, ctx->stream);
        }
    }
    if (e == hipSuccess && !desc.empty()) e = hipMemcpyAsync(ctx->dc_desc.p, desc.data(), desc.size() * sizeof(DecDesc), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_raw_off, raw_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_raw_cnt, raw_cnt.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_presorted, presorted.data(), (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_first32, first32.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_big, big.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { free_hk(); return hip_fail(e, "sketch upload", __FILE__, __LINE__); }
    tm[3] = now_s();
    const uint64_t n_desc_all = dev_walk ? n_desc_dev : (uint64_t)desc.size();
    if (n_desc_all > 0xfffffff0ull) return fail((set_error("too many stored super-k-mers for one call"), SPSP_ERR_OVERFLOW));
    if (n_desc_all) {
        hipLaunchKernelGGL(k_decode_emit, dim3((uint32_t)((n_desc_all + 255) / 256)), dim3(256), 0, ctx->stream, d_text,
                           ctx->dc_desc.as<DecDesc>(), (uint32_t)n_desc_all, k, m, ctx->dc_mn.as<uint32_t>(), ctx->dc_lo.as<uint64_t>(),
                           has_hi ? ctx->dc_hi.as<uint64_t>() : (uint64_t*)nullptr);
    }
    if (!ctx->attr_sort_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kSortCapHi * 20));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kSortCapLo * 12));
        ctx->attr_sort_set = true;
    }
    static const char* dbg_sort = getenv("SPSP_DEBUG_DECODE_SORT");     // "network": every sketch through the bitonic network (A/B, tests)
    const uint32_t force_network = dbg_sort && dbg_sort[0] == 'n' ? 1u : 0u;
    // the first launch's arrays hold the largest sketch it sorts, as it is (see the kernel); the second has them in full
    const uint32_t cap_full = has_hi ? kSortCapHi : kSortCapLo;
    uint32_t cap1 = 64;
    for (uint32_t i = 0; i < n; ++i) if (!presorted[i] && !big[i] && raw_cnt[i] > cap1) cap1 = raw_cnt[i];
    cap1 = std::min(cap_full, (cap1 + 63u) & ~63u);
    if (force_network) cap1 = cap_full;
    const size_t per_key = has_hi ? 20 : 12;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && cap1 == cap_full) break;                  // (nothing could have been left for it)
        const uint32_t cap_rt = pass == 0 ? cap1 : cap_full;
        if (has_hi) hipLaunchKernelGGL(k_decode_sort<true>, dim3(n), dim3(kSortThreads), (size_t)cap_rt * per_key, ctx->stream, ctx->dc_mn.as<uint32_t>(), ctx->dc_lo.as<uint64_t>(),
                                       ctx->dc_hi.as<uint64_t>(), d_raw_off, d_raw_cnt, d_presorted, d_big, d_distinct, force_network, cap_rt, (uint32_t)pass);
        else hipLaunchKernelGGL(k_decode_sort<false>, dim3(n), dim3(kSortThreads), (size_t)cap_rt * per_key, ctx->stream, ctx->dc_mn.as<uint32_t>(), ctx->dc_lo.as<uint64_t>(),
                                (uint64_t*)nullptr, d_raw_off, d_raw_cnt, d_presorted, d_big, d_distinct, force_network, cap_rt, (uint32_t)pass);
    }
    // the sketches beyond the LDS sort: distinct keys through the table in HBM (no count rule here: the reader takes every
    // k-mer a sketch holds; no orientation bit: k_decode_emit writes canonical keys), sorted once they lie in place
    if (any_big && (rc = big_dedupe_launch(ctx, has_hi, ctx->dc_mn.as<uint32_t>(), ctx->dc_lo.as<uint64_t>(), has_hi ? ctx->dc_hi.as<uint64_t>() : nullptr,
                                           d_first32, d_raw_cnt, d_big, n, R, nullptr, 0u, ctx->b_mn.as<uint32_t>(), ctx->b_lo.as<uint64_t>(),
                                           has_hi ? ctx->b_hi.as<uint64_t>() : nullptr, d_distinct))) return fail(rc);
    if ((rc = launch_scan_u32(ctx, d_distinct, d_out_off, n, ctx->h_scalar + 7))) return fail(rc);
    const uint32_t gx = (uint32_t)std::min<uint64_t>(2048, std::max<uint64_t>(8, R / n / 2048));   // (more workgroups per sketch when sketches are huge)
    hipLaunchKernelGGL(k_decode_compact, dim3(gx, n), dim3(256), 0, ctx->stream, ctx->dc_mn.as<uint32_t>(), ctx->dc_lo.as<uint64_t>(),
                       has_hi ? ctx->dc_hi.as<uint64_t>() : (const uint64_t*)nullptr, ctx->b_mn.as<uint32_t>(), ctx->b_lo.as<uint64_t>(),
                       has_hi ? ctx->b_hi.as<uint64_t>() : (const uint64_t*)nullptr, d_raw_off, d_big, d_distinct, d_out_off, ctx->c_min.as<uint32_t>(),
                       ctx->c_lo.as<uint64_t>(), has_hi ? ctx->c_hi.as<uint64_t>() : (uint64_t*)nullptr);
    std::vector<uint32_t> off32(n + 1);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(off32.data(), d_out_off, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_hk();
    if (e != hipSuccess) return hip_fail(e, "sketch decode", __FILE__, __LINE__);
    tm[4] = now_s();
    if (dbg_times) fprintf(stderr, "[spsp decode] %u sketches: structure walk %.1f ms, descriptors %.1f ms, uploads queued %.1f ms, kernels + wait %.1f ms\n", n,
                           (tm[1] - tm[0]) * 1e3, (tm[2] - tm[1]) * 1e3, (tm[3] - tm[2]) * 1e3, (tm[4] - tm[3]) * 1e3);
    for (uint32_t i = 0; i <= n; ++i) sk_off[i] = off32[i];
    if (any_big) {
        std::vector<std::pair<uint32_t, uint32_t>> segs;
        for (uint32_t i = 0; i < n; ++i) if (big[i]) segs.emplace_back(off32[i], off32[i + 1] - off32[i]);
        if ((rc = big_sort_segments(ctx, has_hi, ctx->c_min.as<uint32_t>(), ctx->c_lo.as<uint64_t>(), has_hi ? ctx->c_hi.as<uint64_t>() : nullptr,
                                    ctx->dc_mn.as<uint32_t>(), ctx->dc_lo.as<uint64_t>(), has_hi ? ctx->dc_hi.as<uint64_t>() : nullptr, segs))) return rc;
    }
    return SPSP_OK;
}

int compare_payloads_impl(spsp_ctx* ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n, const int* extra_has,
                          const uint32_t* extra_mn, uint32_t n_query, uint32_t* k_out, uint32_t* m_out, uint32_t* inter, uint64_t* card, bool* mirrored,
                          std::vector<uint64_t>* cells_out) {
    if (mirrored) *mirrored = false;
    if (cells_out) cells_out->clear();
    std::vector<uint64_t> sk_off((size_t)n + 1, 0);
    int rc = sketch_decode_device_impl(ctx, payloads, lens, n, extra_has, extra_mn, k_out, m_out, sk_off.data());
    if (rc || n == 0) return rc;
    for (uint32_t i = 0; i < n; ++i) card[i] = sk_off[i + 1] - sk_off[i];
    if (sk_off[n] == 0) return SPSP_OK;                            // (inter is zero on entry)
    if ((rc = ctx->c_inter.reserve((size_t)n * n * 4))) return rc;
    const uint32_t k = *k_out;
    if (n >= 1024 && n <= 65535) {
        // a large matrix is mostly zeros (sketches of different species share no k-mer) and 4 n^2 bytes would cross PCIe:
        // the non-zero cells come back instead, straight from the row sums (spsp_multi.hip: compare_cells_run)
        uint64_t cap = std::max<uint64_t>(1u << 16, (uint64_t)n * 32), n_cells = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if ((rc = ctx->m_cells.reserve((size_t)cap * 8))) return rc;
            rc = compare_cells_run(ctx, [&]() { return compare_device_begin_impl(ctx, k, ctx->c_min.as<uint32_t>(), ctx->c_lo.as<uint64_t>(), k > 32 ? ctx->c_hi.as<uint64_t>() : nullptr,
                                                                                  sk_off.data(), n, n_query, 0, 1, ctx->c_inter.as<uint32_t>()); },
                                   n, n_query < n ? n_query : n, ctx->c_inter.as<uint32_t>(), ctx->m_cells.as<uint64_t>(), cap, &n_cells, &ctx->m_cells);
            if (rc != SPSP_ERR_OVERFLOW) break;
            cap = n_cells;
        }
        if (rc) return rc;
        std::vector<uint64_t> cells((size_t)n_cells);
        if (n_cells) {
            SPSP_HIP(hipMemcpyAsync(cells.data(), ctx->m_cells.p, (size_t)n_cells * 8, hipMemcpyDeviceToHost, ctx->stream));
            SPSP_HIP(hipStreamSynchronize(ctx->stream));
        }
        if (cells_out) { cells_out->swap(cells); return SPSP_OK; }   // the caller prints from the cells: no matrix
        for (uint64_t cw : cells) {
            const size_t i = (size_t)(cw >> 48), j = (size_t)((cw >> 32) & 0xffffu);
            inter[i * n + j] = (uint32_t)cw;
            if (mirrored) inter[j * n + i] = (uint32_t)cw;         // (the printers then read rows only)
        }
        if (mirrored) *mirrored = true;
        return SPSP_OK;
    }
    SPSP_HIP(hipMemsetAsync(ctx->c_inter.p, 0, (size_t)n * n * 4, ctx->stream));
    if ((rc = compare_device_impl(ctx, k, ctx->c_min.as<uint32_t>(), ctx->c_lo.as<uint64_t>(), k > 32 ? ctx->c_hi.as<uint64_t>() : nullptr,
                                  sk_off.data(), n, n_query, 0, 1, ctx->c_inter.as<uint32_t>()))) return rc;
    SPSP_HIP(hipMemcpyAsync(inter, ctx->c_inter.p, (size_t)n * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    return SPSP_OK;
}

}  // namespace spsp

using namespace spsp;

extern "C" int spsp_sketch_decode_device(spsp_ctx* ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n,
                                         uint32_t* k, uint32_t* m, void** d_minimizer, void** d_kmer_lo, void** d_kmer_hi,
                                         uint64_t* sk_off) {
    if (!ctx || !k || !m || !d_minimizer || !d_kmer_lo || !d_kmer_hi || !sk_off || (n && (!payloads || !lens))) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    const int rc = sketch_decode_device_impl(ctx, payloads, lens, n, nullptr, nullptr, k, m, sk_off);
    if (rc) return rc;
    *d_minimizer = ctx->c_min.p; *d_kmer_lo = ctx->c_lo.p; *d_kmer_hi = *k > 32 ? ctx->c_hi.p : nullptr;
    return SPSP_OK;
}
