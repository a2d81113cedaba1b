// spsp_host.cpp -- the host side of the two CLIs behind the C-ABI: FASTA
// ingest, sketch (de)serialisation, CSV printing and zstr-compatible file I/O.
// These are the stages SURVEY.md 8(a) keeps on the host (A4, A7-A11, A16);
// they work on 2-bit integers throughout instead of the reference's
// std::string substr/find chains, but produce the same bytes.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "spsp_internal.h"

typedef unsigned __int128 u128;

namespace {

using spsp::set_error;
using spsp::now_s;

inline uint32_t code_of(uint8_t c) { return (c >> 1) & 3u; }  // A=0 C=1 T=2 G=3 (utils.cpp:13-16)
const char kNuc[4] = {'A', 'C', 'T', 'G'};                     // int2nuc (utils.cpp:26-45)

template <class T> T* dup_vec(const std::vector<T>& v) {
    T* p = (T*)malloc(std::max<size_t>(1, v.size()) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

inline uint64_t rev_groups64(uint64_t x) {  // reverse the 32 two-bit groups
    x = __builtin_bswap64(x);
    x = ((x & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL);
    x = ((x & 0x3333333333333333ULL) << 2) | ((x >> 2) & 0x3333333333333333ULL);
    return x;
}
inline u128 revcomp_kmer(u128 v, uint32_t k) {  // rcb, utils.cpp:397-438
    const uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
    const u128 r = ((u128)(rev_groups64(lo) ^ 0xaaaaaaaaaaaaaaaaULL) << 64) | (rev_groups64(hi) ^ 0xaaaaaaaaaaaaaaaaULL);
    return r >> (128 - 2 * k);
}

// ---------------------------------------------------------------- sketching --
struct Entry {
    u128 kmer;
    uint32_t minimizer;
    uint8_t count;    // wraps at 256 like the reference's uint8_t (SubSampler.h:24)
    uint8_t pos_min;
    uint8_t seen;
};

// (minimizer, k-mer) -> entry index; linear probing, grows by doubling.
struct KmerIndex {
    std::vector<uint32_t> slot;  // entry index + 1, 0 = empty
    uint64_t mask = 0;
    size_t used = 0;
    static uint64_t hash(uint32_t mn, u128 km) {
        uint64_t x = (uint64_t)km * 0x9E3779B97F4A7C15ULL ^ ((uint64_t)(km >> 64) + mn) * 0xC2B2AE3D27D4EB4FULL;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        return x;
    }
    void init(size_t cap_pow2) { slot.assign(cap_pow2, 0); mask = cap_pow2 - 1; used = 0; }
    void grow(const std::vector<Entry>& es) {
        std::vector<uint32_t> old;
        old.swap(slot);
        init(old.size() * 2);
        for (uint32_t v : old)
            if (v) place(es, v - 1);
    }
    void place(const std::vector<Entry>& es, uint32_t idx) {
        uint64_t p = hash(es[idx].minimizer, es[idx].kmer) & mask;
        while (slot[p]) p = (p + 1) & mask;
        slot[p] = idx + 1; ++used;
    }
    int64_t find(const std::vector<Entry>& es, uint32_t mn, u128 km) const {
        uint64_t p = hash(mn, km) & mask;
        while (slot[p]) {
            const Entry& e = es[slot[p] - 1];
            if (e.kmer == km && e.minimizer == mn) return slot[p] - 1;
            p = (p + 1) & mask;
        }
        return -1;
    }
};

constexpr uint32_t kPlaceholder = 0xffffffffu;   // `order` entry of a k-mer the device abundance pass dropped

struct Builder {
    uint32_t k, m, abundance;
    u128 kmask;
    uint32_t mmask;
    std::vector<Entry> entries;
    KmerIndex index;
    std::vector<std::pair<uint32_t, uint32_t>> order;  // (minimizer, entry index) in insertion order
    spsp_sketch_stats st;

    // Subsampler::handle_superkmer (SubSampler.cpp:243-302) on 2-bit codes.  fl (optional): the device abundance
    // pass's verdict per k-mer of this super-k-mer (spsp_abund.hip) -- a k-mer that will stay below -a is not indexed;
    // the first occurrence of each such k-mer leaves a placeholder, which is what the reference keeps of it (a map
    // entry that is counted, makes its bucket exist and is never usable)
    void add_superkmer(const uint8_t* s, uint32_t len, uint32_t minimizer, bool rev, const uint8_t* fl = nullptr) {
        st.selected_superkmer_number++;
        st.selected_kmer_number += len - k + 1;
        if (len == 2 * k - m) st.count_maximal_skmer++;
        static thread_local std::vector<uint8_t> o;
        o.resize(len);
        if (rev) for (uint32_t t = 0; t < len; ++t) o[t] = (uint8_t)(code_of(s[len - 1 - t]) ^ 2u);
        else for (uint32_t t = 0; t < len; ++t) o[t] = (uint8_t)code_of(s[t]);
        // start positions (oriented) where the m-mer equals the minimizer, ascending
        static thread_local std::vector<uint32_t> occ;
        occ.clear();
        uint32_t mv = 0;
        for (uint32_t t = 0; t < len; ++t) {
            mv = ((mv << 2) | o[t]) & mmask;
            if (t + 1 >= m && mv == minimizer) occ.push_back(t + 1 - m);
        }
        size_t oc = 0;
        u128 kv = 0;
        for (uint32_t t = 0; t < len; ++t) {
            kv = ((kv << 2) | o[t]) & kmask;
            if (t + 1 < k) continue;
            const uint32_t i = t + 1 - k;  // k-mer start
            if (fl && !(fl[i] & 1u)) {
                if (fl[i] & 2u) order.emplace_back(minimizer, kPlaceholder);
                continue;
            }
            while (oc < occ.size() && occ[oc] < i) ++oc;
            // kmerstr.find(minimizer) : first occurrence inside the k-mer, else npos -> (uint8_t)
            const uint8_t pos_min = (oc < occ.size() && occ[oc] + m <= i + k) ? (uint8_t)(occ[oc] - i) : (uint8_t)0xff;
            const int64_t at = index.find(entries, minimizer, kv);
            if (at >= 0) { entries[at].count++; continue; }
            Entry e; e.kmer = kv; e.minimizer = minimizer; e.count = 1; e.pos_min = pos_min; e.seen = 0;
            entries.push_back(e);
            order.emplace_back(minimizer, (uint32_t)entries.size() - 1);
            if ((index.used + 1) * 2 > index.slot.size()) index.grow(entries);
            index.place(entries, (uint32_t)entries.size() - 1);
        }
    }

    bool usable(const Entry& e) const { return !e.seen && e.count >= abundance; }

    // Subsampler::find_next (SubSampler.cpp:566-602): neighbours tried in A,T,C,G order.
    int64_t step(uint32_t mn, u128 from, bool left) {
        static const uint32_t kTry[4] = {0, 2, 1, 3};
        for (uint32_t c : kTry) {
            const u128 nx = left ? ((from >> 2) | ((u128)c << (2 * k - 2))) : (((from << 2) | c) & kmask);
            const int64_t at = index.find(entries, mn, nx);
            if (at >= 0 && usable(entries[at])) { entries[at].seen = 1; return at; }
        }
        return -1;
    }

    // emission half of parse_fasta_test (SubSampler.cpp:458-504): the header line ...
    static void emit_header(uint32_t k, uint32_t m, uint64_t selected_kmers, double rate, std::string& out) {
        char hdr[128];
        const int hl = snprintf(hdr, sizeof hdr, "%llu %llu %llu %f\n", (unsigned long long)(k - 1 + (k - m + 1)),
                                (unsigned long long)m, (unsigned long long)selected_kmers, rate);
        out.append(hdr, hl);
    }
    // ... and the buckets this builder holds (one, when sketch_build_core builds bucket by bucket)
    void emit(std::string& out) {
        // std::map<uint32_t,...> order over buckets, insertion order inside one (ankerl dense map)
        std::stable_sort(order.begin(), order.end(),
                         [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
        const uint32_t full = 2 * k - m, half = k - m;
        std::vector<uint8_t> buf(3 * k + 8), maxcodes;
        std::string text;
        size_t b0 = 0;
        while (b0 < order.size()) {
            size_t b1 = b0;
            const uint32_t mn = order[b0].first;
            while (b1 < order.size() && order[b1].first == mn) ++b1;
            st.actual_minimizer_number++;
            st.seen_kmers_at_reconstruction += b1 - b0;
            for (uint32_t j = 0; j < m; ++j) out.push_back(kNuc[(mn >> (2 * (m - 1 - j))) & 3]);
            maxcodes.clear(); text.clear();
            size_t cursor = b0;  // find_first_kmer: `seen` only ever turns on, so a cursor visits the same k-mers
            for (;;) {
                while (cursor < b1 && (order[cursor].second == kPlaceholder || !usable(entries[order[cursor].second]))) ++cursor;
                if (cursor >= b1) break;
                Entry& start = entries[order[cursor].second];
                start.seen = 1;
                // reconstruct_superkmer (SubSampler.cpp:512-564)
                uint64_t n_left = (uint64_t)half - start.pos_min, n_right = start.pos_min;
                uint32_t L = k + 4, R = L + k;  // [L,R) holds the growing super-k-mer, room for half on each side
                for (uint32_t j = 0; j < k; ++j) buf[L + j] = (uint8_t)((start.kmer >> (2 * (k - 1 - j))) & 3);
                u128 cur = start.kmer;
                while (R - L != full) {
                    if (n_left != 0) {
                        const int64_t at = step(mn, cur, true);
                        n_left -= 1;
                        if (at >= 0) { buf[--L] = (uint8_t)((entries[at].kmer >> (2 * (k - 1))) & 3); }
                        else n_left = 0;
                        if (n_left == 0) cur = start.kmer; else if (at >= 0) cur = entries[at].kmer;
                    } else if (n_right != 0) {
                        const int64_t at = step(mn, cur, false);
                        n_right -= 1;
                        if (at < 0) break;
                        buf[R++] = (uint8_t)(entries[at].kmer & 3);
                        cur = entries[at].kmer;
                    } else break;
                }
                const uint32_t slen = R - L;
                if (slen == full) {
                    st.seen_max_superkmers_at_reconstruction++;
                    maxcodes.insert(maxcodes.end(), buf.begin() + L, buf.begin() + L + half);
                    maxcodes.insert(maxcodes.end(), buf.begin() + L + k, buf.begin() + L + k + half);
                } else {
                    // skmer_str.find(minstr): first m-mer equal to the minimizer
                    uint32_t mv = 0, p = slen;
                    for (uint32_t t = 0; t < slen; ++t) {
                        mv = ((mv << 2) | buf[L + t]) & mmask;
                        if (t + 1 >= m && mv == mn) { p = t + 1 - m; break; }
                    }
                    for (uint32_t t = 0; t < p && t < slen; ++t) text.push_back(kNuc[buf[L + t]]);
                    text.push_back('\n');
                    for (uint32_t t = p + m; t < slen; ++t) text.push_back(kNuc[buf[L + t]]);
                    text.push_back('\n');
                }
                st.seen_superkmers_at_reconstruction++;
            }
            // strCompressor (utils.cpp:48-68), accumulator starting at 0
            std::string blob;
            if (!maxcodes.empty()) {
                const uint8_t mod = (uint8_t)(maxcodes.size() % 4);
                blob.push_back((char)mod);
                uint8_t c = 0;
                for (size_t i = 0; i < maxcodes.size(); ++i) {
                    c = (uint8_t)(c + maxcodes[i]);
                    if ((i + 1) % 4 == 0) { blob.push_back((char)c); c = 0; }
                    c = (uint8_t)(c << 2);
                }
                if (mod != 0) blob.push_back((char)c);
            }
            const uint32_t sz = (uint32_t)blob.size();
            out.append((const char*)&sz, 4);
            out += blob;
            out += text;
            out += "\n\n";
            b0 = b1;
        }
    }
};

// ------------------------------------------------------------- gzip reading --
int inflate_all(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
    // zstr autodetect (include/zstr.hpp:154-167): gzip or zlib header, else plain
    const bool packed = n >= 2 && ((in[0] == 0x1F && in[1] == 0x8B) ||
                                   (in[0] == 0x78 && (in[1] == 0x01 || in[1] == 0x9C || in[1] == 0xDA)));
    if (!packed) { out.assign(in, in + n); return SPSP_OK; }
    // One inflator per THREAD, reset per member (10 000 sketch files of 3 KB: inflateInit2's 40 KB of state and a zeroed 1 MiB
    // bounce buffer per file were most of the 48 us a file cost its thread), and the output grows in place: the member's ISIZE
    // trailer says how much is coming (a hint only: sizes are taken from what inflate delivers)
    struct Inflator { z_stream zs; bool live = false; ~Inflator() { if (live) inflateEnd(&zs); } };
    static thread_local Inflator I;
    out.clear();
    size_t hint = n * 4 + 64;
    if (n >= 18 && in[0] == 0x1F) { uint32_t isize; memcpy(&isize, in + n - 4, 4); if (isize >= n && isize < (1u << 30)) hint = (size_t)isize + 64; }
    out.resize(hint);
    size_t have = 0, at = 0;
    while (at < n) {  // concatenated members: the inflator starts afresh per member, as zstr does (:198-203)
        if (!I.live) {
            memset(&I.zs, 0, sizeof I.zs);
            if (inflateInit2(&I.zs, 15 + 32) != Z_OK) { set_error("inflateInit2 failed"); return SPSP_ERR_IO; }
            I.live = true;
        } else if (inflateReset2(&I.zs, 15 + 32) != Z_OK) { set_error("inflateReset2 failed"); return SPSP_ERR_IO; }
        z_stream& zs = I.zs;
        zs.avail_in = 0;
        int ret = Z_OK;
        bool truncated = false;
        while (ret != Z_STREAM_END) {
            if (zs.avail_in == 0) {
                if (at >= n) { truncated = true; break; }
                const size_t take = std::min<size_t>(n - at, (size_t)1 << 30);
                zs.next_in = const_cast<Bytef*>(in + at);
                zs.avail_in = (uInt)take;
                at += take;
            }
            if (have == out.size()) out.resize(out.size() * 2 + 4096);
            const size_t room = std::min<size_t>(out.size() - have, (size_t)1 << 30);
            zs.next_out = out.data() + have;
            zs.avail_out = (uInt)room;
            ret = inflate(&zs, Z_NO_FLUSH);
            if (ret != Z_OK && ret != Z_STREAM_END && ret != Z_BUF_ERROR) {
                set_error("inflate failed (%d)", ret);
                return SPSP_ERR_IO;
            }
            have += room - zs.avail_out;
        }
        at -= zs.avail_in;  // bytes not consumed belong to the next member
        if (truncated) break;
    }
    out.resize(have);
    return SPSP_OK;
}

}  // namespace

namespace spsp {

int inflate_all_host(const uint8_t* in, size_t n, std::vector<uint8_t>& out) { return inflate_all(in, n, out); }

// Super-k-mer i's bases come either from the full cleaned sequence (bases + rec_off[rec] + start) or, when the
// ingest ran on the GPU, from a compact buffer holding only the selected super-k-mers (compact + compact_off[i]).
int sketch_build_core(const spsp_params* p, double rate, const uint64_t* rec_off, uint32_t n_rec, const spsp_superkmer* sk,
                      uint64_t n_sk, const uint8_t* bases, const uint8_t* compact, const uint32_t* compact_off,
                      uint8_t** payload, uint64_t* payload_len, spsp_sketch_stats* stats, const uint8_t* kmer_flags) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!payload || !payload_len || (n_rec && !rec_off) || (n_sk && (!sk || (!bases && !(compact && compact_off))))) {
        set_error("NULL argument");
        return SPSP_ERR_ARG;
    }
    // The k-mers of a bucket (one minimizer) meet no k-mer of another: the index is keyed by (minimizer, k-mer), the walk
    // of reconstruct_superkmer stays inside its bucket, buckets are written in minimizer order.  So the super-k-mers are
    // grouped by minimizer (stream order kept inside a group) and every bucket is built and written on its own -- an index
    // that fits a core's cache instead of one over all k-mers of a genome (a 1 Gbp record at -s 100: 10^7 k-mers, 10 us
    // per super-k-mer in one table, 5.0 s), and on several threads when there is enough to do.
    spsp_sketch_stats st;
    memset(&st, 0, sizeof st);
    static const bool dbg_times = getenv("SPSP_DEBUG_BUILD_TIMES") != nullptr;   // analysis: where the builder's time goes (stderr)
    const double bt0 = dbg_times ? now_s() : 0.0;
    uint64_t nb = 0, pos_end = 0, occ = 0;                // occ: k-mer occurrences so far (numbering of kmer_flags)
    uint32_t cur_rec = 0xffffffffu;
    for (uint32_t r = 0; r < n_rec; ++r) {
        const uint64_t len = rec_off[r + 1] - rec_off[r];
        if (len >= p->k) st.read_kmer += len - p->k + 1;
    }
    std::vector<std::pair<uint32_t, uint32_t>> by_mn;     // (minimizer, super-k-mer)
    std::vector<uint64_t> occ_of;                         // first k-mer occurrence of every super-k-mer (kmer_flags)
    if (n_sk > 0xfffffff0ull) { set_error("too many super-k-mers for one sketch"); return SPSP_ERR_OVERFLOW; }
    by_mn.reserve((size_t)n_sk);
    if (kmer_flags) occ_of.reserve((size_t)n_sk);
    for (uint64_t i = 0; i < n_sk; ++i) {
        const spsp_superkmer& e = sk[i];
        if (e.rec >= n_rec || e.len < p->k || e.start + e.len > rec_off[e.rec + 1] - rec_off[e.rec]) {
            set_error("super-k-mer %llu is outside its record", (unsigned long long)i);
            return SPSP_ERR_ARG;
        }
        // nb_mmer_selected bookkeeping (SubSampler.cpp:410-424, 445): a pure function of the stream
        if (e.rec != cur_rec) { cur_rec = e.rec; pos_end = 0; }
        const uint64_t rlen = rec_off[e.rec + 1] - rec_off[e.rec];
        if (e.start + e.len == rlen) nb -= p->m - 1;  // the tail call :441-450
        else {
            if (e.start + p->m - 2 > pos_end) {
                if (pos_end > 0) nb -= p->m - 1;
                nb += e.len;
                nb -= p->k - p->m;
            } else nb += e.start + e.len - (pos_end + 1);
            pos_end = e.start + e.len - 1;
        }
        by_mn.emplace_back(e.minimizer, (uint32_t)i);
        if (kmer_flags) occ_of.push_back(occ);
        st.selected_kmer_number += e.len - p->k + 1;
        occ += e.len - p->k + 1;
    }
    nb -= p->m - 1;  // SubSampler.cpp:458
    st.nb_mmer_selected = nb;
    // grouped by minimizer, stream order kept inside a group: two stable counting passes over 16 bits each (a comparison sort of
    // the 8 x 10^5 super-k-mers of a 4 Gbp metagenome segment was 0.1 s on one thread in front of the threaded part)
    if (by_mn.size() > 4096) {
        std::vector<std::pair<uint32_t, uint32_t>> tmp(by_mn.size());
        std::vector<uint32_t> cnt(65537);
        for (int pass = 0; pass < 2; ++pass) {
            const int sh = 16 * pass;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (const auto& e : by_mn) ++cnt[((e.first >> sh) & 0xffffu) + 1];
            for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
            for (const auto& e : by_mn) tmp[cnt[(e.first >> sh) & 0xffffu]++] = e;
            by_mn.swap(tmp);
        }
    } else
        std::stable_sort(by_mn.begin(), by_mn.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
    struct Bucket { size_t first, last; std::string text; spsp_sketch_stats st; };
    std::vector<Bucket> buckets;
    for (size_t a = 0; a < by_mn.size();) {
        size_t z = a;
        while (z < by_mn.size() && by_mn[z].first == by_mn[a].first) ++z;
        buckets.push_back(Bucket{a, z, std::string(), spsp_sketch_stats{}});
        a = z;
    }
    const double bt1 = dbg_times ? now_s() : 0.0;
    std::vector<uint32_t> todo(buckets.size());           // largest first: the threads finish together
    for (uint32_t i = 0; i < todo.size(); ++i) todo[i] = i;
    std::sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) { return buckets[a].last - buckets[a].first > buckets[b].last - buckets[b].first; });
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        for (;;) {
            const uint32_t t = next.fetch_add(1);
            if (t >= todo.size()) return;
            Bucket& B = buckets[todo[t]];
            Builder b;
            b.k = p->k; b.m = p->m; b.abundance = p->abundance;
            b.kmask = (((u128)1) << (2 * p->k)) - 1;
            b.mmask = (1u << (2 * p->m)) - 1;
            memset(&b.st, 0, sizeof b.st);
            b.index.init(1024);
            for (size_t x = B.first; x < B.last; ++x) {
                const uint32_t i = by_mn[x].second;
                const spsp_superkmer& e = sk[i];
                const uint8_t* src = compact ? compact + compact_off[i] : bases + rec_off[e.rec] + e.start;
                b.add_superkmer(src, e.len, e.minimizer, e.rev != 0, kmer_flags ? kmer_flags + occ_of[i] : nullptr);
            }
            b.emit(B.text);                               // (nothing when every k-mer of the bucket was dropped without a trace: the bucket does not exist)
            B.st = b.st;
        }
    };
    // (the file pipeline builds the sketches of different files on its own worker threads: threads of its own only for a
    // sketch that is large by itself)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_thr = n_sk >= 20000 ? std::min<unsigned>(std::min<unsigned>(n_sk >= 400000 ? 16u : 8u, hw), (unsigned)std::max<size_t>(1, buckets.size())) : 1u;
    if (n_thr <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    const double bt2 = dbg_times ? now_s() : 0.0;
    std::string out;
    Builder::emit_header(p->k, p->m, st.selected_kmer_number, rate, out);
    for (const Bucket& B : buckets) {
        out += B.text;
        st.selected_superkmer_number += B.st.selected_superkmer_number;
        st.count_maximal_skmer += B.st.count_maximal_skmer;
        st.actual_minimizer_number += B.st.actual_minimizer_number;
        st.seen_kmers_at_reconstruction += B.st.seen_kmers_at_reconstruction;
        st.seen_superkmers_at_reconstruction += B.st.seen_superkmers_at_reconstruction;
        st.seen_max_superkmers_at_reconstruction += B.st.seen_max_superkmers_at_reconstruction;
    }
    *payload = (uint8_t*)malloc(out.size() + 1);
    if (!*payload) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    memcpy(*payload, out.data(), out.size());
    *payload_len = out.size();
    if (stats) *stats = st;
    if (dbg_times) fprintf(stderr, "[spsp build] %llu super-k-mers, %zu buckets, %u threads: group by minimizer %.1f ms, buckets %.1f ms, join %.1f ms\n",
                           (unsigned long long)n_sk, buckets.size(), n_thr, (bt1 - bt0) * 1e3, (bt2 - bt1) * 1e3, (now_s() - bt2) * 1e3);
    return SPSP_OK;
}

// What the device builder (spsp_build.hip) leaves to the host: the counters that are a pure function of the scan's stream
// (the same lines as in sketch_build_core above) and the header line.
int sketch_stream_stats(const spsp_params* p, const uint64_t* rec_off, uint32_t n_rec, const spsp_superkmer* sk, uint64_t n_sk, spsp_sketch_stats* st_out) {
    spsp_sketch_stats st;
    memset(&st, 0, sizeof st);
    uint64_t nb = 0, pos_end = 0;
    uint32_t cur_rec = 0xffffffffu;
    for (uint32_t r = 0; r < n_rec; ++r) {
        const uint64_t len = rec_off[r + 1] - rec_off[r];
        if (len >= p->k) st.read_kmer += len - p->k + 1;
    }
    for (uint64_t i = 0; i < n_sk; ++i) {
        const spsp_superkmer& e = sk[i];
        if (e.rec >= n_rec || e.len < p->k || e.start + e.len > rec_off[e.rec + 1] - rec_off[e.rec]) {
            set_error("super-k-mer %llu is outside its record", (unsigned long long)i);
            return SPSP_ERR_ARG;
        }
        if (e.rec != cur_rec) { cur_rec = e.rec; pos_end = 0; }
        const uint64_t rlen = rec_off[e.rec + 1] - rec_off[e.rec];
        if (e.start + e.len == rlen) nb -= p->m - 1;
        else {
            if (e.start + p->m - 2 > pos_end) {
                if (pos_end > 0) nb -= p->m - 1;
                nb += e.len;
                nb -= p->k - p->m;
            } else nb += e.start + e.len - (pos_end + 1);
            pos_end = e.start + e.len - 1;
        }
        st.selected_kmer_number += e.len - p->k + 1;
        st.selected_superkmer_number++;
        if (e.len == 2 * p->k - p->m) st.count_maximal_skmer++;
    }
    nb -= p->m - 1;
    st.nb_mmer_selected = nb;
    *st_out = st;
    return SPSP_OK;
}
void sketch_header_line(uint32_t k, uint32_t m, uint64_t selected_kmers, double rate, std::string& out) { Builder::emit_header(k, m, selected_kmers, rate, out); }

// structure of one payload for the GPU decoder (spsp_decode.hip): header (Comparator.cpp:23-37) and, per bucket,
// [m ASCII][u32 n][blob][lines]["\n\n"] (:186-260).  Every offset it hands to the device is checked against `len` here.
int sketch_parse_structure_host(const uint8_t* payload, uint64_t len, ParsedSketch* P) {
    const uint8_t* nl = (payload && len) ? (const uint8_t*)memchr(payload, '\n', len) : nullptr;
    if (!nl) { set_error("sketch has no header line"); return SPSP_ERR_FORMAT; }
    char* endp = nullptr;
    const std::string header((const char*)payload, nl - payload);
    const long skm = strtol(header.c_str(), &endp, 10);
    const long mm = strtol(endp, &endp, 10);
    if (skm <= 0 || skm > 126 || mm <= 0 || mm > 15 || (skm + mm) / 2 > 63 || (skm + mm) / 2 < mm) { set_error("bad sketch header '%.60s'", header.c_str()); return SPSP_ERR_FORMAT; }
    const uint32_t m = (uint32_t)mm, k = (uint32_t)((skm + mm) / 2), half = (uint32_t)((skm - mm) / 2);
    P->k = k; P->m = m;
    uint64_t pos = (uint64_t)(nl - payload) + 1;
    uint64_t out = 0;
    auto push = [&](uint64_t off, uint32_t mn, uint32_t info, uint64_t count) {
        if (count == 0) return;
        if (out + count > 0xfffffff0ull) { P->standard = false; return; }
        P->desc.push_back(DecDesc{off, mn, info, (uint32_t)out, 0});
        out += count;
    };
    while (pos + m <= len) {
        uint32_t mn = 0;
        for (uint32_t j = 0; j < m; ++j) mn = (mn << 2) | code_of(payload[pos + j]);
        pos += m;
        uint32_t nbytes = 0;
        if (pos + 4 > len) break;
        memcpy(&nbytes, payload + pos, 4);
        pos += 4;
        if (pos + nbytes > len) { set_error("bucket blob runs past the end of the sketch"); return SPSP_ERR_FORMAT; }
        uint64_t seq_len = 0;
        if (nbytes) {
            if (payload[pos] != 0) P->standard = false;           // a partial last byte: never written by the sketcher (k, m odd)
            seq_len = (uint64_t)(nbytes - 1) * 4;
        }
        if (half > 0) {
            if ((2 * half) % 4 != 0) P->standard = false;         // k - m odd (never from bin/sub_sampler, which forces k and m odd, SubSampler.cpp:732-741; possible through spsp_sketch_build_host): a blob byte then straddles two super-k-mers -- host decoder
            for (uint64_t i = 0; (i + 1) * 2 * half <= seq_len; ++i) push(pos + 1 + i * (half / 2), mn, 0u, k - m + 1);
        } else if (seq_len == 0) {
            push(pos, mn, 2u, 1);                                  // k == m: the bare minimizer is one k-mer (Comparator.cpp:88-90,193-198)
        }
        pos += nbytes;
        for (;;) {                                                 // "prefix\nsuffix\n" until an empty pair (:226-260)
            if (pos >= len) break;
            const uint8_t* e1 = (const uint8_t*)memchr(payload + pos, '\n', len - pos);
            const uint64_t s1 = pos, l1 = e1 ? (uint64_t)(e1 - payload) - pos : len - pos;
            pos = e1 ? s1 + l1 + 1 : len;
            const uint8_t* e2 = pos < len ? (const uint8_t*)memchr(payload + pos, '\n', len - pos) : nullptr;
            const uint64_t s2 = pos, l2 = pos < len ? (e2 ? (uint64_t)(e2 - payload) - pos : len - pos) : 0;
            pos = e2 ? s2 + l2 + 1 : len;
            if (l1 == 0 && l2 == 0) break;
            if (!e1 || l1 > 255 || l2 > 255 || s2 != s1 + l1 + 1) { P->standard = false; continue; }
            const uint64_t total = l1 + m + l2;
            push(s1, mn, 1u | ((uint32_t)l1 << 2) | ((uint32_t)l2 << 10), total >= k ? total - k + 1 : 0);
        }
    }
    P->n_keys = out;
    return SPSP_OK;
}

}  // namespace spsp

extern "C" {

// Subsampler::compute_threshold (SubSampler.cpp:622-631) + SubSampler.h:79-83
uint64_t spsp_threshold_host(uint32_t k, uint32_t m, double sampling_rate) {
    if (!(sampling_rate > 1)) return ~0ULL;
    const uint64_t w = (uint64_t)k - m + 1;
    const long double frac = (long double)1 / sampling_rate;
    const long double root = powl((long double)1 - frac, (long double)1 / w);
    const long double scaled = ((long double)1 - root) * ((uint64_t)1 << 63);
    return (uint64_t)scaled * 2;
}

int spsp_fasta_clean_host(const char* text, uint64_t n, uint8_t** bases, uint64_t** rec_off, uint32_t* n_rec) {
    if (!bases || !rec_off || !n_rec || (n && !text)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    // keep[c] = upper-cased base for acgtACGT, 0 for everything else (clean_dna, utils.cpp:675-702)
    struct Keep {
        uint8_t t[256];
        Keep() {
            memset(t, 0, sizeof t);
            for (const char* q = "ACGT"; *q; ++q) { t[(uint8_t)*q] = (uint8_t)*q; t[(uint8_t)(*q + 32)] = (uint8_t)*q; }
        }
    };
    static const Keep keep_tab;
    const uint8_t* keep = keep_tab.t;
    std::vector<uint64_t> off;
    uint8_t* out = (uint8_t*)malloc((size_t)n + 64);
    if (!out) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    uint64_t w = 0, pos = 0;
    bool at_end = false;
    // SubSampler.cpp:334-348 + getLineFasta utils.cpp:706-718: every turn drops
    // one line, then swallows lines up to the next one starting with '>'.
    while (!at_end) {
        off.push_back(w);
        const char* nl = pos < n ? (const char*)memchr(text + pos, '\n', n - pos) : nullptr;
        if (!nl) { pos = n; at_end = true; } else pos = (uint64_t)(nl - text) + 1;
        while (!at_end) {
            if (pos >= n) { at_end = true; break; }
            const uint8_t c0 = (uint8_t)text[pos];
            if (c0 == '>' || c0 == 0xFF) break;  // (char)peek() == EOF aliasing for 0xFF
            const char* e = (const char*)memchr(text + pos, '\n', n - pos);
            const uint64_t stop = e ? (uint64_t)(e - text) : n;
            for (uint64_t i = pos; i < stop; ++i) {
                const uint8_t b = keep[(uint8_t)text[i]];
                out[w] = b;
                w += (b != 0);
            }
            if (!e) { pos = n; at_end = true; } else pos = stop + 1;
        }
    }
    off.push_back(w);
    *bases = out;
    *rec_off = dup_vec(off);
    *n_rec = (uint32_t)(off.size() - 1);
    if (!*rec_off) { free(out); set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    return SPSP_OK;
}

int spsp_sketch_build_host(const spsp_params* p, double rate, const uint8_t* bases, const uint64_t* rec_off,
                           uint32_t n_rec, const spsp_superkmer* sk, uint64_t n_sk, uint8_t** payload,
                           uint64_t* payload_len, spsp_sketch_stats* stats) {
    if (n_sk && !bases) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    return spsp::sketch_build_core(p, rate, rec_off, n_rec, sk, n_sk, bases, nullptr, nullptr, payload, payload_len, stats, nullptr);
}

int spsp_sketch_parse_host(const uint8_t* payload, uint64_t len, uint32_t* k_out, uint32_t* m_out,
                           uint32_t** minimizer, uint64_t** kmer_lo, uint64_t** kmer_hi, uint64_t* n_out) {
    if (!payload || !k_out || !m_out || !minimizer || !kmer_lo || !kmer_hi || !n_out) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    // header "<2k-m> <m> <n> <rate>\n" (Comparator.cpp:23-37)
    const uint8_t* nl = (const uint8_t*)memchr(payload, '\n', len);
    if (!nl) { set_error("sketch has no header line"); return SPSP_ERR_FORMAT; }
    char* endp = nullptr;
    const std::string header((const char*)payload, nl - payload);
    const long skm = strtol(header.c_str(), &endp, 10);
    const long mm = strtol(endp, &endp, 10);
    // (range checks first: strtol saturates at LONG_MAX on garbage and the sums below must not overflow)
    if (skm <= 0 || skm > 126 || mm <= 0 || mm > 15 || (skm + mm) / 2 > 63 || (skm + mm) / 2 < mm) { set_error("bad sketch header '%.60s'", header.c_str()); return SPSP_ERR_FORMAT; }
    const uint32_t m = (uint32_t)mm, k = (uint32_t)((skm + mm) / 2), half = (uint32_t)((skm - mm) / 2);
    const u128 kmask = (((u128)1) << (2 * k)) - 1;
    struct Key { uint32_t mn; uint64_t hi, lo; };
    std::vector<Key> keys;
    std::vector<uint8_t> seq;
    auto push_kmers = [&](const std::vector<uint8_t>& s, size_t from, size_t count_limit, uint32_t mn) {
        // canonical k-mers of s[from..], at most count_limit of them
        u128 kv = 0;
        size_t made = 0;
        for (size_t t = from; t < s.size() && made < count_limit; ++t) {
            kv = ((kv << 2) | s[t]) & kmask;
            if (t + 1 - from < k) continue;
            const u128 rc = revcomp_kmer(kv, k);
            const u128 c = kv < rc ? kv : rc;
            keys.push_back(Key{mn, (uint64_t)(c >> 64), (uint64_t)c});
            ++made;
        }
    };
    uint64_t pos = (uint64_t)(nl - payload) + 1;
    while (pos + m <= len) {
        uint32_t mn = 0;
        for (uint32_t j = 0; j < m; ++j) mn = (mn << 2) | code_of(payload[pos + j]);
        pos += m;
        uint32_t nbytes = 0;
        if (pos + 4 > len) break;
        memcpy(&nbytes, payload + pos, 4);
        pos += 4;
        if (pos + nbytes > len) { set_error("bucket blob runs past the end of the sketch"); return SPSP_ERR_FORMAT; }
        // strDecompressor (utils.cpp:71-111)
        seq.clear();
        if (nbytes) {
            const uint8_t mod = payload[pos];
            const uint64_t last = (mod == 0) ? nbytes : nbytes - 1;
            for (uint64_t i = 1; i < last; ++i) {
                const uint8_t b = payload[pos + i];
                seq.push_back((b >> 6) & 3); seq.push_back((b >> 4) & 3); seq.push_back((b >> 2) & 3); seq.push_back(b & 3);
            }
            if (mod != 0 && last >= 1) {
                uint8_t b = payload[pos + last];
                uint8_t f[4] = {0, 0, 0, 0};
                for (int i = 0; i < (int)(mod & 3) + 1; ++i) { f[(mod & 3) - i] = b & 3; b >>= 2; }
                for (int i = 0; i < (int)(mod & 3); ++i) seq.push_back(f[i]);
            }
        }
        pos += nbytes;
        // inject_minimizer (Comparator.cpp:78-92) + maximal walk (:196-225): every
        // 2(k-m) stored bases are one maximal super-k-mer = k-m+1 k-mers
        if (half > 0) {
            std::vector<uint8_t> sk(2 * half + m);
            for (size_t i = 0; i + 2 * half <= seq.size(); i += 2 * half) {
                for (uint32_t j = 0; j < half; ++j) sk[j] = seq[i + j];
                for (uint32_t j = 0; j < m; ++j) sk[half + j] = (mn >> (2 * (m - 1 - j))) & 3;
                for (uint32_t j = 0; j < half; ++j) sk[half + m + j] = seq[i + half + j];
                push_kmers(sk, 0, k - m + 1, mn);
            }
        } else if (seq.empty()) {
            // k == m: inject_minimizer returns the bare minimizer for an empty blob and, being k long,
            // it is walked as one k-mer (Comparator.cpp:88-90,193-198)
            std::vector<uint8_t> sk(m);
            for (uint32_t j = 0; j < m; ++j) sk[j] = (mn >> (2 * (m - 1 - j))) & 3;
            push_kmers(sk, 0, 1, mn);
        }
        // non-maximal super-k-mers: "prefix\nsuffix\n" until an empty pair (:226-260)
        for (;;) {
            if (pos >= len) break;
            const uint8_t* e1 = (const uint8_t*)memchr(payload + pos, '\n', len - pos);
            const uint64_t s1 = pos, l1 = e1 ? (uint64_t)(e1 - payload) - pos : len - pos;
            pos = e1 ? s1 + l1 + 1 : len;
            const uint8_t* e2 = pos < len ? (const uint8_t*)memchr(payload + pos, '\n', len - pos) : nullptr;
            const uint64_t s2 = pos, l2 = pos < len ? (e2 ? (uint64_t)(e2 - payload) - pos : len - pos) : 0;
            pos = e2 ? s2 + l2 + 1 : len;
            if (l1 == 0 && l2 == 0) break;
            std::vector<uint8_t> sk;
            sk.reserve(l1 + m + l2);
            for (uint64_t j = 0; j < l1; ++j) sk.push_back((uint8_t)code_of(payload[s1 + j]));
            for (uint32_t j = 0; j < m; ++j) sk.push_back((mn >> (2 * (m - 1 - j))) & 3);
            for (uint64_t j = 0; j < l2; ++j) sk.push_back((uint8_t)code_of(payload[s2 + j]));
            push_kmers(sk, 0, (size_t)-1, mn);
        }
    }
    // distinct canonical k-mers per bucket == distinct (minimizer, k-mer) keys
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
        if (a.mn != b.mn) return a.mn < b.mn;
        if (a.hi != b.hi) return a.hi < b.hi;
        return a.lo < b.lo;
    });
    keys.erase(std::unique(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.mn == b.mn && a.hi == b.hi && a.lo == b.lo; }), keys.end());
    const size_t n = keys.size();
    uint32_t* mn = (uint32_t*)malloc(std::max<size_t>(1, n) * 4);
    uint64_t* lo = (uint64_t*)malloc(std::max<size_t>(1, n) * 8);
    uint64_t* hi = (uint64_t*)malloc(std::max<size_t>(1, n) * 8);
    if (!mn || !lo || !hi) { free(mn); free(lo); free(hi); set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    for (size_t i = 0; i < n; ++i) { mn[i] = keys[i].mn; lo[i] = keys[i].lo; hi[i] = keys[i].hi; }
    *minimizer = mn; *kmer_lo = lo; *kmer_hi = hi; *n_out = n; *k_out = k; *m_out = m;
    return SPSP_OK;
}

// The N-way merge starts by reading every file's first minimizer into ONE shared buffer, without an end-of-file
// check (increment_files INITIALIZATION, Comparator.cpp:316-319; buffer = m times 'A', :294).  A sketch with no
// bucket therefore leaves the buffer as the previous file filled it and enters the merge with that minimizer and an
// empty blob; inject_minimizer turns an empty blob into the bare minimizer (:88-90), which is dropped when m < k
// (:193-195) but IS a k-mer when k == m.  So with k == m an empty sketch holds one phantom k-mer that depends on
// its predecessor in the file list.  This emulates that read for one sketch, in file order.
int spsp_sketch_chain_host(const uint8_t* payload, uint64_t len, uint32_t k, uint32_t m, char* read_buffer, int* has_key,
                           uint32_t* minimizer, uint64_t* kmer_lo, uint64_t* kmer_hi) {
    if (!payload || !read_buffer || !has_key || !minimizer || !kmer_lo || !kmer_hi) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    *has_key = 0;
    const uint8_t* nl = (const uint8_t*)memchr(payload, '\n', len);
    if (!nl) { set_error("sketch has no header line"); return SPSP_ERR_FORMAT; }
    const uint64_t body = len - (uint64_t)(nl + 1 - payload);
    const uint64_t got = body < m ? body : m;
    memcpy(read_buffer, nl + 1, (size_t)got);          // istream::read stores what it could read
    if (got == m || k != m) return SPSP_OK;            // a real first bucket, or a phantom that is dropped (m < k)
    uint32_t mn = 0;
    u128 kv = 0;
    for (uint32_t j = 0; j < m; ++j) { const uint32_t c = code_of((uint8_t)read_buffer[j]); mn = (mn << 2) | c; kv = (kv << 2) | c; }
    const u128 rc = revcomp_kmer(kv, k);
    const u128 c = kv < rc ? kv : rc;
    *minimizer = mn; *kmer_lo = (uint64_t)c; *kmer_hi = (uint64_t)(c >> 64);
    *has_key = 1;
    return SPSP_OK;
}

// print_containment / print_jaccard (Comparator.cpp:362-460); operator<<(double)
// with setprecision(p) in the default float format is printf's %.*g.
// sortCSV (sort_csv.cpp:26-111): put the rows and columns of a (symmetric, all-vs-all) Jaccard CSV into the order of
// the original file-of-files -- sub_sampler's output list is in OpenMP completion order (SubSampler.cpp:782-786).
// Values pass through a double and are printed with six significant digits, as the reference's `out << double`.
int spsp_sort_csv_host(const char* csv, uint64_t csv_len, const char* fof, uint64_t fof_len, char** text, uint64_t* len) {
    if (!text || !len || (csv_len && !csv) || (fof_len && !fof)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    *text = nullptr; *len = 0;
    // fof lines -> rank (first occurrence wins, as std::find does; a final newline yields one empty last name)
    std::unordered_map<std::string, uint32_t> rank_of;
    {
        uint32_t line_no = 0;
        uint64_t a = 0;
        for (uint64_t i = 0; i <= fof_len; ++i) {
            if (i == fof_len || fof[i] == '\n') { rank_of.emplace(std::string(fof + a, i - a), line_no++); a = i + 1; }
        }
    }
    // CSV lines
    std::vector<std::pair<const char*, size_t>> lines;
    for (uint64_t a = 0, i = 0; i <= csv_len; ++i)
        if (i == csv_len || csv[i] == '\n') { lines.emplace_back(csv + a, (size_t)(i - a)); a = i + 1; }
    auto tokens = [](const char* p, size_t n) {   // split() of utils.cpp:609-629: the last field ends at the first non-printable byte
        std::vector<std::string> t;
        size_t a = 0;
        for (size_t i = 0; i < n; ++i) if (p[i] == ',') { t.emplace_back(p + a, i - a); a = i + 1; }
        size_t e = a;
        while (e < n && isprint((unsigned char)p[e])) ++e;
        t.emplace_back(p + a, e - a);
        return t;
    };
    if (lines.empty()) { set_error("empty CSV"); return SPSP_ERR_FORMAT; }
    const std::vector<std::string> head = tokens(lines[0].first, lines[0].second);
    const size_t N = head.size();
    std::vector<std::pair<uint32_t, uint32_t>> order;   // (rank in the fof, column in the input)
    for (size_t c = 0; c < N; ++c) {
        auto it = rank_of.find(head[c]);
        if (it == rank_of.end()) { set_error("column '%s' is not in the file of files", head[c].c_str()); return SPSP_ERR_FORMAT; }
        order.emplace_back(it->second, (uint32_t)c);
    }
    std::sort(order.begin(), order.end());
    for (size_t c = 1; c < N; ++c)
        if (order[c].first == order[c - 1].first) { set_error("column '%s' appears twice", head[order[c].second].c_str()); return SPSP_ERR_FORMAT; }
    std::vector<uint32_t> new_of(N);
    for (size_t r = 0; r < N; ++r) new_of[order[r].second] = (uint32_t)r;
    // Rows.  A matrix of thousands of sketches is nearly all "0" cells (10^8 cells at BASELINE configs[3], 95 000 pairs that
    // share anything): the cells that are not literally "0" are kept as (output row, output column, value) -- the reference
    // fills the TRANSPOSED cell (sort_csv.cpp:83) -- and a row is written as runs of "0," between them.  Lines are independent:
    // parsed and printed by a few threads (the round-3 form built 10^8 std::strings and an N x N array of doubles: 8 s at
    // N = 6 000 on one thread).
    size_t n_rows = 0;                                   // consecutive lines that can hold a row (sort_csv.cpp:80 stops at the first that cannot)
    while (1 + n_rows < lines.size() && lines[1 + n_rows].second >= N) ++n_rows;
    const size_t parse_rows = n_rows < N ? n_rows : N;
    struct Cell { uint32_t orow, ocol; double x; };
    unsigned workers = std::thread::hardware_concurrency();
    if (workers == 0) workers = 1;
    if (workers > 16) workers = 16;
    if (N * N < (1u << 20)) workers = 1;
    if (workers > parse_rows) workers = parse_rows ? (unsigned)parse_rows : 1;
    std::vector<std::vector<Cell>> found(workers);
    std::vector<size_t> bad_row(workers, (size_t)-1);       // first row this worker could not parse
    std::vector<std::string> bad_msg(workers);
    auto parse = [&](unsigned w) {
        char buf[64];
        for (size_t r = w; r < parse_rows; r += workers) {
            const char* p = lines[1 + r].first;
            const size_t n = lines[1 + r].second;
            size_t a = 0, c = 0;
            bool failed = false;
            auto cell = [&](size_t b) {                  // cell c = [a, b)
                const size_t l = b - a;
                if (l == 1 && p[a] == '0') return;       // the common case: 0 -> "0"
                if (l == 0 || l >= sizeof buf) { failed = true; return; }
                memcpy(buf, p + a, l); buf[l] = 0;
                char* endp = nullptr;
                const double x = strtod(buf, &endp);
                if (endp == buf) { failed = true; return; }
                found[w].push_back(Cell{new_of[c], new_of[r], x});
            };
            for (size_t i = 0; i < n && c < N && !failed; ++i)
                if (p[i] == ',') { cell(i); if (failed) { char t[96]; snprintf(t, sizeof t, "row %zu, column %zu: not a number", r, c); bad_msg[w] = t; } a = i + 1; ++c; }
            if (!failed && c < N) {                       // the last field ends at the first non-printable byte (split(), utils.cpp:609-629)
                size_t e = a;
                while (e < n && isprint((unsigned char)p[e])) ++e;
                if (c + 1 < N) { char t[96]; snprintf(t, sizeof t, "row %zu has %zu values, expected %zu", r, c + 1, N); bad_msg[w] = t; failed = true; }
                else { cell(e); if (failed) { char t[96]; snprintf(t, sizeof t, "row %zu, column %zu: not a number", r, c); bad_msg[w] = t; } ++c; }
            }
            if (failed) { bad_row[w] = r; return; }
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(parse, w);
        parse(0);
        for (auto& th : pool) th.join();
    }
    {
        unsigned first = workers;
        for (unsigned w = 0; w < workers; ++w) if (bad_row[w] != (size_t)-1 && (first == workers || bad_row[w] < bad_row[first])) first = w;
        if (first != workers) { set_error("%s", bad_msg[first].c_str()); return SPSP_ERR_FORMAT; }
    }
    if (n_rows > N) { set_error("more than %zu rows", N); return SPSP_ERR_FORMAT; }
    if (n_rows != N) { set_error("%zu rows for %zu columns (a containment file, or a query-mode matrix?)", n_rows, N); return SPSP_ERR_FORMAT; }
    // cells by output row, then by column
    std::vector<size_t> at(N + 1, 0);
    for (auto& v : found) for (const Cell& c : v) ++at[c.orow + 1];
    for (size_t i = 0; i < N; ++i) at[i + 1] += at[i];
    std::vector<std::pair<uint32_t, double>> adj(at[N]);
    {
        std::vector<size_t> cur(at.begin(), at.end() - 1);
        for (auto& v : found) for (const Cell& c : v) adj[cur[c.orow]++] = std::make_pair(c.ocol, c.x);
    }
    std::vector<std::vector<Cell>>().swap(found);
    for (size_t i = 0; i < N; ++i) {                      // the diagonal must be 1 (the reference stops at the first that is not, sort_csv.cpp:100-104)
        double d = 0;
        for (size_t e = at[i]; e < at[i + 1]; ++e) if (adj[e].first == i) d = adj[e].second;
        if (d != 1) { set_error("diagonal entry %zu is not 1 (the reference stops here)", i); return SPSP_ERR_FORMAT; }
    }
    std::string head_out;
    for (size_t r = 0; r < N; ++r) { head_out += head[order[r].second]; head_out += (r + 1 != N) ? ',' : '\n'; }
    static const std::string zero_run = []() { std::string z; z.reserve(8192); for (int i = 0; i < 4096; ++i) z += "0,"; return z; }();
    std::vector<std::string> parts(workers);
    auto print = [&](unsigned w) {
        std::string& out = parts[w];
        const size_t r0 = N * w / workers, r1 = N * (w + 1) / workers;
        out.reserve((r1 - r0) * N * 2 + 4096);
        char num[40];
        for (size_t i = r0; i < r1; ++i) {
            std::sort(adj.begin() + at[i], adj.begin() + at[i + 1], [](const std::pair<uint32_t, double>& x, const std::pair<uint32_t, double>& y) { return x.first < y.first; });
            size_t col = 0;
            auto zeros_to = [&](size_t upto) {
                size_t z = upto - col;
                while (z) { const size_t take = z < 4096 ? z : 4096; out.append(zero_run.data(), take * 2); z -= take; }
                col = upto;
            };
            for (size_t e = at[i]; e < at[i + 1]; ++e) {
                if (e + 1 < at[i + 1] && adj[e + 1].first == adj[e].first) continue;      // (a column given twice cannot happen: columns are a permutation)
                zeros_to(adj[e].first);
                out.append(num, (size_t)snprintf(num, sizeof num, "%g", adj[e].second));
                out += ',';
                col = adj[e].first + 1;
            }
            zeros_to(N);
            out.back() = '\n';
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(print, w);
        print(0);
        for (auto& th : pool) th.join();
    }
    size_t total = head_out.size();
    for (auto& p2 : parts) total += p2.size();
    char* buf = (char*)malloc(total + 1);
    if (!buf) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    size_t pos = 0;
    memcpy(buf, head_out.data(), head_out.size()); pos += head_out.size();
    for (auto& p2 : parts) { memcpy(buf + pos, p2.data(), p2.size()); pos += p2.size(); }
    buf[total] = 0;
    *text = buf; *len = total;
    return SPSP_OK;
}

// mirrored: cell (i, j) is also stored at (j, i) -- a row is then read left to right instead of down a column for j < i
// (10^4 sketches: 5 x 10^7 reads 40 KB apart per matrix otherwise)
// printf's %.*g (what `out << setprecision(p) << double` prints, Comparator.cpp:362-460) by std::to_chars: the same characters by
// the standard's word (general format with a precision = "as if by printf %.*g in the C locale"; 15 x 10^6 random scores at five
// precisions compared equal) at a third of the time -- a 10^8-cell matrix of 10^4 sketches is 2 x 10^5 numbers per matrix
static inline int format_g(char* buf, size_t cap, int precision, double v) {
    if (precision >= 0) {
        const std::to_chars_result r = std::to_chars(buf, buf + cap, v, std::chars_format::general, precision);
        if (r.ec == std::errc()) return (int)(r.ptr - buf);
    }
    return snprintf(buf, cap, "%.*g", precision, v);
}
static int csv_impl(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const uint32_t* inter,
                    const uint64_t* card, int precision, double min_threshold, char** text, uint64_t* len, bool mirrored) {
    if (!text || !len || (n && (!names || !inter || !card))) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    std::string head;
    for (uint32_t i = 0; i < n; ++i) { head += names[i]; head += (i + 1 != n) ? ',' : '\n'; }
    if (!jaccard) head += '\n';
    // rows are independent: formatted by a few host threads (row blocks), concatenated in order ("next" row N3)
    const uint32_t rows = n < n_query ? n : n_query;
    // a matrix of thousands of sketches is nearly all zeros (species that share no k-mer): runs of "0," are copied from a
    // constant instead of being appended cell by cell
    static const std::string zero_run = []() { std::string z; z.reserve(8192); for (int i = 0; i < 4096; ++i) z += "0,"; return z; }();
    auto format_rows = [&](uint32_t r0, uint32_t r1, std::string& out) {
        char num[64];
        out.reserve((size_t)(r1 - r0) * n * 2 + 4096);
        for (uint32_t i = r0; i < r1; ++i) {
            const uint32_t* row = inter + (uint64_t)i * n;
            uint32_t zeros = 0;                                   // cells "0," not written yet
            auto flush = [&]() { while (zeros) { const uint32_t take = zeros < 4096 ? zeros : 4096; out.append(zero_run.data(), (size_t)take * 2); zeros -= take; } };
            for (uint32_t j = 0; j < n; ++j) {
                uint32_t sc = 0;
                if (i != j) sc = (mirrored || i < j) ? row[j] : inter[(uint64_t)j * n + i];
                if (i != j && sc == 0) { ++zeros; continue; }
                flush();
                if (i == j) out += '1';
                else {
                    const double score = jaccard ? (double)sc / (double)(card[i] + card[j] - sc) : (double)sc / (double)card[i];
                    if (score < min_threshold) out += '0';
                    else { const int l = format_g(num, sizeof num, precision, score); out.append(num, l); }
                }
                out += ',';
            }
            flush();
            out.back() = '\n';                                    // (the row's last separator)
        }
    };
    unsigned workers = std::thread::hardware_concurrency();
    if (workers == 0) workers = 1;
    if (workers > 16) workers = 16;
    if ((uint64_t)rows * n < (1u << 18)) workers = 1;          // small matrices: not worth a thread
    if (workers > rows) workers = rows ? rows : 1;
    std::vector<std::string> parts(workers);
    {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < workers; ++w) {
            const uint32_t r0 = (uint32_t)((uint64_t)rows * w / workers), r1 = (uint32_t)((uint64_t)rows * (w + 1) / workers);
            if (w + 1 == workers) format_rows(r0, r1, parts[w]);
            else pool.emplace_back(format_rows, r0, r1, std::ref(parts[w]));
        }
        for (auto& th : pool) th.join();
    }
    size_t total = head.size();
    for (auto& p2 : parts) total += p2.size();
    *text = (char*)malloc(total + 1);
    if (!*text) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    size_t at = 0;
    memcpy(*text, head.data(), head.size()); at += head.size();
    for (auto& p2 : parts) { memcpy(*text + at, p2.data(), p2.size()); at += p2.size(); }
    (*text)[total] = 0;
    *len = total;
    return SPSP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Rows of a sparse pair matrix straight into a gzip member (round 5).  A row of a 10^4-sketch matrix is ~12 runs of "0,"
// between ~10 numbers; formatted as text and handed to zlib it is 20 KB through deflate's match finder and crc32 -- 42 ms of
// thread time per 10^8-cell matrix.  Here the row never exists as text: a run of z cells is the two literals "0," and
// ceil((2z - 2) / 258) matches of distance 2 in ONE fixed-Huffman deflate block (RFC 1951 3.2.6), and the member's CRC-32
// takes the run as 16 table steps -- crc(A || B) = crc(A) * x^(8 |B|) + crc(B) over GF(2), with the operators x^(16 * 2^t)
// (2^t cells) as byte tables and crc("0," x 2^t) precomputed.  gunzip gives the bytes the text path writes (the tests read
// every member back through zlib, which checks CRC-32 and ISIZE).
namespace {
struct ZeroRunCrc {
    uint32_t tab[17][4][256];                 // tab[t]: multiplication of a (reflected) CRC value by x^(16 * 2^t), byte-wise
    uint32_t crc[17];                         // crc32 of "0," x 2^t
    static uint32_t times(const uint32_t* mat, uint32_t vec) { uint32_t s = 0; for (int i = 0; vec; vec >>= 1, ++i) if (vec & 1u) s ^= mat[i]; return s; }
    ZeroRunCrc() {
        uint32_t a[32], b[32];
        a[0] = 0xedb88320u;                    // the operator of ONE zero bit (zlib crc32_combine): the polynomial, then the shifts
        for (int n = 1; n < 32; ++n) a[n] = 1u << (n - 1);
        auto square = [](uint32_t* dst, const uint32_t* src) { for (int n = 0; n < 32; ++n) dst[n] = times(src, src[n]); };
        square(b, a); square(a, b); square(b, a); square(a, b);   // 2, 4, 8, 16 bits: a = two zero bytes = one cell
        const char cell[2] = {'0', ','};
        for (int t = 0; t <= 16; ++t) {
            for (int by = 0; by < 4; ++by)
                for (uint32_t v = 0; v < 256; ++v) tab[t][by][v] = times(a, v << (8 * by));
            if (t == 0) crc[0] = (uint32_t)crc32(0L, (const Bytef*)cell, 2);
            else crc[t] = mul(t - 1, crc[t - 1]) ^ crc[t - 1];   // 2^t cells = 2^(t-1) cells followed by 2^(t-1) cells
            square(b, a); memcpy(a, b, sizeof a);
        }
    }
    uint32_t mul(int t, uint32_t c) const { return tab[t][0][c & 255u] ^ tab[t][1][(c >> 8) & 255u] ^ tab[t][2][(c >> 16) & 255u] ^ tab[t][3][c >> 24]; }
    uint32_t append(uint32_t c, uint32_t cells) const {            // crc of (what c covers) followed by `cells` x "0,"
        for (int t = 0; cells; cells >>= 1, ++t) if (cells & 1u) c = mul(t, c) ^ crc[t];
        return c;
    }
};
static const ZeroRunCrc& zero_run_crc() { static const ZeroRunCrc z; return z; }

struct StringSink {
    std::string* out;
    void bytes(const char* p, size_t n) { out->append(p, n); }
    void zeros(uint32_t cells) {
        static const std::string run = []() { std::string z; z.reserve(8192); for (int i = 0; i < 4096; ++i) z += "0,"; return z; }();
        while (cells) { const uint32_t take = cells < 4096 ? cells : 4096; out->append(run.data(), (size_t)take * 2); cells -= take; }
    }
};
struct GzSink {                                // one gzip member: header, ONE fixed-Huffman block, CRC-32, ISIZE
    std::vector<uint8_t>* out;
    uint64_t acc = 0; int nbits = 0;
    uint32_t crc = 0; uint64_t isize = 0;
    static uint32_t rev(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i); return r; }
    // the bit patterns of the fixed code, made once: a literal's (8 or 9 bits), and the match of 258 bytes at distance 2 (13 bits)
    struct Codes {
        uint16_t lit[256]; uint8_t lit_len[256]; uint32_t m258; int m258_len;
        Codes() {
            for (uint32_t b = 0; b < 256; ++b) { if (b < 144) { lit[b] = (uint16_t)rev(0x30u + b, 8); lit_len[b] = 8; } else { lit[b] = (uint16_t)rev(0x190u + (b - 144u), 9); lit_len[b] = 9; } }
            m258 = rev(0xc0u + 5u, 8) | (rev(1u, 5) << 8); m258_len = 13;      // symbol 285 (no extra bits), distance code 1
        }
    };
    static const Codes& codes() { static const Codes c; return c; }
    void put(uint32_t v, int n) {               // n bits, least significant first (n <= 32, at most 7 bits pending: 39 fit)
        acc |= (uint64_t)v << nbits; nbits += n;
        if (nbits >= 32) {                      // four bytes at a time
            const size_t at = out->size();
            out->resize(at + 4);
            const uint32_t w = (uint32_t)acc;
            memcpy(out->data() + at, &w, 4);
            acc >>= 32; nbits -= 32;
        }
    }
    void flush_bytes() { while (nbits >= 8) { out->push_back((uint8_t)acc); acc >>= 8; nbits -= 8; } }
    void begin() {
        static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
        out->insert(out->end(), hdr, hdr + 10);
        put(1, 1); put(1, 2);                   // BFINAL = 1, BTYPE = 01
    }
    void literal(uint8_t b) { const Codes& C = codes(); put(C.lit[b], C.lit_len[b]); }
    void match2(uint32_t len) {                 // `len` bytes (3 .. 258) copied from distance 2
        if (len == 258) { const Codes& C = codes(); put(C.m258, C.m258_len); return; }
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        int c = 28;
        while (base[c] > len) --c;              // (258 has a code of its own; 227 .. 257 are code 284 + extra bits)
        const uint32_t sym = 257u + (uint32_t)c;
        if (sym < 280) put(rev(sym - 256u, 7), 7); else put(rev(0xc0u + (sym - 280u), 8), 8);
        if (extra[c]) put(len - base[c], extra[c]);
        put(rev(1u, 5), 5);                     // distance code 1 = distance 2, no extra bits
    }
    void bytes(const char* p, size_t n) {
        for (size_t i = 0; i < n; ++i) literal((uint8_t)p[i]);
        crc = (uint32_t)crc32(crc, (const Bytef*)p, (uInt)n); isize += n;
    }
    void zeros(uint32_t cells) {
        if (!cells) return;
        literal('0'); literal(',');
        uint64_t rem = 2ull * cells - 2;
        while (rem >= 258) { match2(258); rem -= 258; }
        if (rem >= 4) match2((uint32_t)rem); else if (rem == 2) { literal('0'); literal(','); }
        crc = zero_run_crc().append(crc, cells); isize += 2ull * cells;
    }
    void end() {
        put(0, 7);                              // end of block (symbol 256: seven zero bits)
        flush_bytes();
        if (nbits) { put(0, 8 - nbits); flush_bytes(); }
        for (int i = 0; i < 4; ++i) out->push_back((uint8_t)(crc >> (8 * i)));
        for (int i = 0; i < 4; ++i) out->push_back((uint8_t)((uint32_t)isize >> (8 * i)));
    }
};
}  // namespace

// The same bytes from the SPARSE form of the pair matrix (packed cells i << 48 | j << 32 | count, i < j, every pair at most
// once): a comparison of thousands of sketches returns ~10 non-zero partners per row, and a row is then "0," runs between
// them -- no n x n matrix is built or scanned (10^4 sketches: 2 x 400 MB of reads per matrix before).
static int deflate_member(const uint8_t* data, uint64_t len, int level, std::vector<uint8_t>& out);
// gz_path != nullptr: nothing is returned as text; row blocks of ~8 MB are formatted AND deflated by the workers, one gzip
// member each, and written in order (a valid gzip file that zstr / zlib / gunzip read back as one stream, like the members
// of spsp_write_gz_host): the 200 MB of a 10^8-cell matrix never exist in one piece.  times[0] / times[1]: the workers'
// formatting / deflate + write seconds, summed and divided by the worker count.
static int csv_cells_impl(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const std::vector<uint64_t>& cells,
                          const uint64_t* card, int precision, double min_threshold, char** text, uint64_t* len,
                          const char* gz_path = nullptr, int gz_level = 1, double* times = nullptr) {
    std::string head;
    for (uint32_t i = 0; i < n; ++i) { head += names[i]; head += (i + 1 != n) ? ',' : '\n'; }
    if (!jaccard) head += '\n';
    const uint32_t rows = n < n_query ? n : n_query;
    // partners of every printed row, by column
    std::vector<uint32_t> deg((size_t)rows + 1, 0);
    for (uint64_t c : cells) {
        const uint32_t i = (uint32_t)(c >> 48), j = (uint32_t)(c >> 32) & 0xffffu;
        if (i < rows) ++deg[i + 1];
        if (j < rows) ++deg[j + 1];
    }
    for (uint32_t i = 0; i < rows; ++i) deg[i + 1] += deg[i];
    std::vector<uint64_t> adj((size_t)deg[rows]);                  // partner << 32 | count
    {
        std::vector<uint32_t> at(deg.begin(), deg.end() - 1);
        for (uint64_t c : cells) {
            const uint32_t i = (uint32_t)(c >> 48), j = (uint32_t)(c >> 32) & 0xffffu;
            if (i < rows) adj[at[i]++] = ((uint64_t)j << 32) | (uint32_t)c;
            if (j < rows) adj[at[j]++] = ((uint64_t)i << 32) | (uint32_t)c;
        }
    }
    // rows [r0, r1) into a sink: text (StringSink) or a gzip member made directly (GzSink).  A cell is its text and ',' -- '\n'
    // for the row's last cell; a run of zeros that reaches the end of the row is one cell shorter and ends in "0\n"
    auto format_rows = [&](uint32_t r0, uint32_t r1, auto& sink) {
        char num[64];
        for (uint32_t i = r0; i < r1; ++i) {
            std::sort(adj.begin() + deg[i], adj.begin() + deg[i + 1]);
            uint32_t col = 0;
            auto zeros_to = [&](uint32_t upto) {                 // cells [col, upto) are "0,"
                if (upto > col) {
                    if (upto == n) { sink.zeros(upto - col - 1); sink.bytes("0\n", 2); }
                    else sink.zeros(upto - col);
                }
                col = upto;
            };
            auto cell = [&](const char* p, size_t l, uint32_t at) {   // the cell of column `at`: its text and the separator in one piece
                char piece[72];
                memcpy(piece, p, l);
                piece[l] = at + 1 == n ? '\n' : ',';
                sink.bytes(piece, l + 1);
                col = at + 1;
            };
            bool diag_done = false;
            auto diagonal = [&]() { zeros_to(i); cell("1", 1, i); diag_done = true; };
            for (uint32_t e = deg[i]; e < deg[i + 1]; ++e) {
                const uint32_t j = (uint32_t)(adj[e] >> 32), sc = (uint32_t)adj[e];
                if (!diag_done && j > i) diagonal();
                zeros_to(j);
                const double score = jaccard ? (double)sc / (double)(card[i] + card[j] - sc) : (double)sc / (double)card[i];
                if (score < min_threshold) cell("0", 1, j);
                else { const int l = format_g(num, sizeof num, precision, score); cell(num, (size_t)l, j); }
            }
            if (!diag_done) diagonal();
            zeros_to(n);
        }
    };
    unsigned workers = std::thread::hardware_concurrency();
    if (workers == 0) workers = 1;
    if (workers > 16) workers = 16;
    if (workers > rows) workers = rows ? rows : 1;
    if (gz_path) {
        // rows per block: ~8 MB of text when the rows are mostly "0," -- and never fewer than four blocks per worker: a DENSE matrix
        // (one species: every cell a 9-byte number) of 1 200 rows was ONE block, formatted and deflated by one thread, 0.46 of a 0.48 s call
        uint32_t rpb = std::max<uint32_t>(1, (uint32_t)((8ull << 20) / (2ull * (n ? n : 1))));
        rpb = std::max<uint32_t>(1, std::min<uint32_t>(rpb, rows / (4 * workers)));
        const uint32_t n_blocks = rows ? (rows + rpb - 1) / rpb : 0;
        std::vector<std::vector<uint8_t>> gz((size_t)n_blocks + 1);
        std::vector<int> rcs((size_t)n_blocks + 1, SPSP_OK);
        std::atomic<uint32_t> next(0);
        std::vector<double> t_fmt(workers, 0.0), t_gz(workers, 0.0);
        rcs[0] = deflate_member((const uint8_t*)head.data(), head.size(), gz_level, gz[0]);        // the header line(s): a member of their own
        auto work = [&](unsigned w) {
            for (;;) {
                const uint32_t b = next.fetch_add(1);
                if (b >= n_blocks) break;
                const double t0 = now_s();
                GzSink sink{&gz[b + 1]};                          // formatted and deflated in one go: the rows never exist as text
                sink.begin();
                format_rows(b * rpb, std::min(rows, (b + 1) * rpb), sink);
                sink.end();
                t_fmt[w] += now_s() - t0;
            }
        };
        if (workers > n_blocks) workers = n_blocks ? n_blocks : 1;
        {
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work, w);
            work(0);
            for (auto& th : pool) th.join();
        }
        const double t2 = now_s();
        for (int r : rcs) if (r) { set_error("deflate failed"); return r; }
        FILE* f = fopen(gz_path, "wb");
        if (!f) { set_error("cannot create '%s'", gz_path); return SPSP_ERR_IO; }
        int rc = SPSP_OK;
        for (auto& m : gz) if (!rc && fwrite(m.data(), 1, m.size(), f) != m.size()) { set_error("short write to '%s'", gz_path); rc = SPSP_ERR_IO; }
        if (fclose(f) != 0 && !rc) { set_error("close failed for '%s'", gz_path); rc = SPSP_ERR_IO; }
        if (times) {
            double a = 0, b2 = 0;
            for (unsigned w = 0; w < workers; ++w) { a += t_fmt[w]; b2 += t_gz[w]; }
            times[0] = a / workers; times[1] = b2 / workers + (now_s() - t2);
        }
        return rc;
    }
    std::vector<std::string> parts(workers);
    {
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < workers; ++w) {
            const uint32_t r0 = (uint32_t)((uint64_t)rows * w / workers), r1 = (uint32_t)((uint64_t)rows * (w + 1) / workers);
            auto job = [&format_rows, &parts, n](uint32_t a, uint32_t z, unsigned which) {
                parts[which].reserve((size_t)(z - a) * n * 2 + 4096);
                StringSink sink{&parts[which]};
                format_rows(a, z, sink);
            };
            if (w + 1 == workers) job(r0, r1, w);
            else pool.emplace_back(job, r0, r1, w);
        }
        for (auto& th : pool) th.join();
    }
    size_t total = head.size();
    for (auto& p2 : parts) total += p2.size();
    *text = (char*)malloc(total + 1);
    if (!*text) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    size_t at = 0;
    memcpy(*text, head.data(), head.size()); at += head.size();
    for (auto& p2 : parts) { memcpy(*text + at, p2.data(), p2.size()); at += p2.size(); }
    (*text)[total] = 0;
    *len = total;
    return SPSP_OK;
}

int spsp_csv_cells_host(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const uint64_t* cells, uint64_t n_cells,
                        const uint64_t* card, int precision, double min_threshold, char** text, uint64_t* len) {
    if (!text || !len || (n && (!names || !card)) || (n_cells && !cells)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (n > 65535) { set_error("at most 65535 sketches (a packed cell holds two 16-bit sketch numbers)"); return SPSP_ERR_ARG; }
    std::vector<uint64_t> v(cells, cells + n_cells);
    std::sort(v.begin(), v.end());
    for (size_t e = 0; e < v.size(); ++e) {
        const uint32_t i = (uint32_t)(v[e] >> 48), j = (uint32_t)(v[e] >> 32) & 0xffffu;
        if (i >= j || j >= n) { set_error("cell %zu names the pair (%u, %u): not i < j < n", e, i, j); return SPSP_ERR_FORMAT; }
        if (e && (v[e] >> 32) == (v[e - 1] >> 32)) { set_error("the pair (%u, %u) occurs twice: add partial cells up first (spsp_matrix_add_cells_device)", i, j); return SPSP_ERR_FORMAT; }
    }
    return csv_cells_impl(jaccard, names, n, n_query, v, card, precision, min_threshold, text, len);
}

int spsp_csv_cells_gz_host(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const uint64_t* cells, uint64_t n_cells,
                           const uint64_t* card, int precision, double min_threshold, const char* gz_path) {
    if (!gz_path || (n && (!names || !card)) || (n_cells && !cells)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (n > 65535) { set_error("at most 65535 sketches (a packed cell holds two 16-bit sketch numbers)"); return SPSP_ERR_ARG; }
    std::vector<uint64_t> v(cells, cells + n_cells);
    std::sort(v.begin(), v.end());
    for (size_t e = 0; e < v.size(); ++e) {
        const uint32_t i = (uint32_t)(v[e] >> 48), j = (uint32_t)(v[e] >> 32) & 0xffffu;
        if (i >= j || j >= n) { set_error("cell %zu names the pair (%u, %u): not i < j < n", e, i, j); return SPSP_ERR_FORMAT; }
        if (e && (v[e] >> 32) == (v[e - 1] >> 32)) { set_error("the pair (%u, %u) occurs twice: add partial cells up first (spsp_matrix_add_cells_device)", i, j); return SPSP_ERR_FORMAT; }
    }
    return csv_cells_impl(jaccard, names, n, n_query, v, card, precision, min_threshold, nullptr, nullptr, gz_path, 1, nullptr);
}

int spsp_csv_host(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const uint32_t* inter,
                  const uint64_t* card, int precision, double min_threshold, char** text, uint64_t* len) {
    return csv_impl(jaccard, names, n, n_query, inter, card, precision, min_threshold, text, len, false);
}

// Three system calls for a small file (open, one read that comes back short, close): where system calls are the cost -- 10^4
// sketch files of 3 KB, on hosts that serialise them -- fopen + fstat + two freads + fclose were twice that.
static int slurp(const char* path, uint8_t** raw, uint64_t* n) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { set_error("cannot open '%s'", path); return SPSP_ERR_IO; }
    // the first 64 KiB land in a buffer the thread keeps (a fresh 64 KiB block per file had malloc trimming the heap per file)
    struct Scratch { uint8_t* p = (uint8_t*)malloc((1u << 16) + 64); ~Scratch() { free(p); } };
    static thread_local Scratch scratch;
    size_t cap = 1u << 16;
    uint8_t* buf = scratch.p;
    bool own = false;
    size_t got = 0;
    bool regular_known = false, regular = false;
    while (buf) {
        const ssize_t r = read(fd, buf + got, cap - got);
        if (r < 0) { if (errno == EINTR) continue; if (own) free(buf); close(fd); set_error("cannot read '%s'", path); return SPSP_ERR_IO; }
        if (r == 0) break;
        got += (size_t)r;
        if (got < cap) {
            // a short read ends a REGULAR file (pipes and ttys may come back short at any time)
            if (!regular_known) { struct stat st; regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size == got; regular_known = true; if (regular) break; }
            continue;
        }
        if (!regular_known) {   // a big file: size the buffer once instead of doubling through it
            struct stat st;
            regular_known = true;
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size > cap) cap = (size_t)st.st_size + 1; else cap *= 2;
        } else cap *= 2;
        uint8_t* nb = own ? (uint8_t*)realloc(buf, cap + 64) : (uint8_t*)malloc(cap + 64);
        if (nb && !own) memcpy(nb, buf, got);
        if (!nb) { if (own) free(buf); buf = nullptr; } else buf = nb;
        own = true;
    }
    close(fd);
    if (buf && !own) {   // small file: an exact-size copy out of the scratch
        uint8_t* nb = (uint8_t*)malloc(got + 64);
        if (nb) memcpy(nb, buf, got);
        buf = nb;
    }
    if (!buf) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    *raw = buf; *n = got;
    return SPSP_OK;
}

int spsp_read_file_host(const char* path, uint8_t** data, uint64_t* len) {
    if (!path || !data || !len) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    uint8_t* raw = nullptr; uint64_t n = 0;
    int rc = slurp(path, &raw, &n);
    if (rc) return rc;
    const bool packed = n >= 2 && ((raw[0] == 0x1F && raw[1] == 0x8B) ||
                                   (raw[0] == 0x78 && (raw[1] == 0x01 || raw[1] == 0x9C || raw[1] == 0xDA)));
    if (!packed) { *data = raw; *len = n; return SPSP_OK; }   // plain text passes through (zstr autodetect)
    std::vector<uint8_t> plain;
    rc = inflate_all(raw, n, plain);
    free(raw);
    if (rc) return rc;
    *data = (uint8_t*)malloc(plain.size() + 64);
    if (!*data) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    if (!plain.empty()) memcpy(*data, plain.data(), plain.size());
    *len = plain.size();
    return SPSP_OK;
}

// zstr::ofstream(filename, expbuffer, level) -> gzip container (zstr.hpp:78-82)
static int deflate_member(const uint8_t* data, uint64_t len, int level, std::vector<uint8_t>& out) {
    // one deflator per thread and level, reset per member (deflateInit2 is 268 KB of allocation per call: per sketch, in the
    // file pipeline)
    struct Deflator { z_stream zs; int level = -100; ~Deflator() { if (level != -100) deflateEnd(&zs); } };
    static thread_local Deflator D;
    if (D.level != level) {
        if (D.level != -100) { deflateEnd(&D.zs); D.level = -100; }
        memset(&D.zs, 0, sizeof D.zs);
        if (deflateInit2(&D.zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return SPSP_ERR_IO;
        D.level = level;
    } else if (deflateReset(&D.zs) != Z_OK) return SPSP_ERR_IO;
    z_stream& zs = D.zs;
    out.resize(deflateBound(&zs, (uLong)len) + 64);
    zs.next_in = const_cast<Bytef*>(data);
    zs.avail_in = (uInt)len;
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int ret = deflate(&zs, Z_FINISH);
    const size_t have = out.size() - zs.avail_out;
    if (ret != Z_STREAM_END) return SPSP_ERR_IO;
    out.resize(have);
    return SPSP_OK;
}

int spsp_write_gz_host(const char* path, const uint8_t* data, uint64_t len, int level) {
    if (!path || (len && !data)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    // plain descriptors, not stdio: fopen/fclose take a process-wide lock (the list of open FILEs), and the file pipeline
    // writes one sketch per task on every host thread
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) { set_error("cannot create '%s'", path); return SPSP_ERR_IO; }
    auto write_all = [&](const uint8_t* p, size_t n) {
        while (n) {
            const ssize_t w = write(fd, p, n);
            if (w < 0) { if (errno == EINTR) continue; return false; }
            p += w; n -= (size_t)w;
        }
        return true;
    };
    // One gzip member per 16 MiB of payload.  Small outputs (every sketch) are a single member, exactly what
    // zstr::ofstream writes; large CSVs are compressed member by member on a few threads -- a valid gzip file
    // that zstr / zlib / gunzip read back as one stream (zstr.hpp:198-203 restarts the inflator per member).
    // (beyond 16 MiB the members shrink to 1 MiB at the least, so that the 28 MB sketch of a 4 Gbp record set at -s 100 keeps
    // every worker busy: 0.43 s with 16 MiB members, 0.11 s with 4 MiB -- seven members on sixteen threads)
    unsigned workers = std::thread::hardware_concurrency();
    if (workers == 0) workers = 1;
    if (workers > 16) workers = 16;
    uint64_t chunk = 16ull << 20;
    if (len > chunk) chunk = std::max<uint64_t>(1ull << 20, std::min<uint64_t>(chunk, ((len + workers - 1) / workers + 0xfffffull) & ~0xfffffull));
    const uint64_t n_chunks = len ? (len + chunk - 1) / chunk : 1;
    if (workers > n_chunks) workers = (unsigned)n_chunks;
    int rc = SPSP_OK;
    for (uint64_t c0 = 0; c0 < n_chunks && !rc; c0 += workers) {
        const unsigned batch = (unsigned)std::min<uint64_t>(workers, n_chunks - c0);
        std::vector<std::vector<uint8_t>> outs(batch);
        std::vector<int> rcs(batch, SPSP_OK);
        std::vector<std::thread> pool;
        auto job = [&](unsigned b) {
            const uint64_t off = (c0 + b) * chunk;
            const uint64_t take = len > off ? std::min<uint64_t>(chunk, len - off) : 0;
            rcs[b] = deflate_member(data + off, take, level, outs[b]);
        };
        for (unsigned b = 1; b < batch; ++b) pool.emplace_back(job, b);
        job(0);
        for (auto& th : pool) th.join();
        for (unsigned b = 0; b < batch && !rc; ++b) {
            if (rcs[b]) { set_error("deflate failed"); rc = rcs[b]; break; }
            if (!write_all(outs[b].data(), outs[b].size())) { set_error("short write to '%s'", path); rc = SPSP_ERR_IO; }
        }
    }
    if (close(fd) != 0 && !rc) { set_error("close failed for '%s'", path); rc = SPSP_ERR_IO; }
    return rc;
}

int spsp_stage_times_read(spsp_ctx* ctx, spsp_stage_times* out, int reset) {
    if (!ctx || !out) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    *out = ctx->stages;
    if (reset) ctx->stages = spsp_stage_times{};
    return SPSP_OK;
}

// chatter: 0 = silent; 1 = the stdout lines of the reference's all-versus-all run (Comparator.cpp:56,69,364,414,
// 503,509); 2 = those of its query run (:56,69,364,414)
static int compare_files_impl(spsp_ctx* ctx, const char* const* paths, uint32_t n, uint32_t n_query, int precision,
                              double min_threshold, const char* out_prefix, int chatter, spsp_ctx* const* more = nullptr, uint32_t n_more = 0) {
    // more / n_more: all contexts of a multi-device call (more[0] == ctx): the comparison is then split by key over them
    if (!ctx || !paths || !out_prefix) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    double t0 = now_s(), t1;
    const double t_start = t0;
    ctx->stages.compare_calls += 1;
    static const bool host_decode = getenv("SPSP_HOST_DECODE") != nullptr;   // A/B switch: decode + sort on host threads
    std::vector<uint8_t*> datas(n, nullptr);
    std::vector<uint64_t> lens(n, 0);
    std::vector<int> rcs(n, SPSP_OK);
    std::vector<std::string> errs(n);
    // Many sketch files: every reader thread takes a RANGE of the files and lays their payloads down back to back (each at the
    // 16-byte-rounded end of the one before) in a region of its own that the context keeps; the decoder finds payloads that lie
    // as it would lay them out and uploads them from where they are, a copy per region (spsp_decode.hip).  One pass: open / read /
    // close do not scale with threads on the hosts measured (10 000 files: 13 ms from one thread and from sixteen,
    // tools/exp/open_scaling.sh), the inflating does -- so a file is inflated by the thread that read it while the others wait for
    // the kernel (two passes, all reads then all inflates into one block by the trailers' lengths, were 13 + 12 ms).
    // datas[i] then points INTO a region (own[i] == 0).
    std::vector<uint8_t> own(n, 1);
    auto free_datas = [&]() { for (uint32_t i = 0; i < n; ++i) { if (own[i]) free(datas[i]); datas[i] = nullptr; } };   // (the regions stay with the context)
    unsigned workers = std::thread::hardware_concurrency();
    if (workers == 0) workers = 1;
    if (workers > 16) workers = 16;
    if (workers > n) workers = n ? n : 1;
    auto on_threads = [&](const std::function<void(uint32_t)>& one) {
        std::atomic<uint32_t> next(0);
        auto work = [&]() { for (;;) { const uint32_t i = next.fetch_add(1); if (i >= n) break; one(i); } };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    };
    // One file at a time per worker: no N open streams (Comparator.cpp:45-50).  Decoding (strDecompressor, inject_minimizer, the
    // k-mer walks, sort, unique) happens for all sketches at once on the GPU (spsp_decode.hip).
    static const bool regions_on = getenv("SPSP_DEBUG_READ_BLOCK") == nullptr || getenv("SPSP_DEBUG_READ_BLOCK")[0] != '0';
    if (n >= 256 && regions_on) {
        if (ctx->h_read_regions.size() < workers) ctx->h_read_regions.resize(workers);
        std::vector<uint64_t> at(n, 0);                          // the payload's place in its thread's region
        auto range = [&](unsigned w) {
            spsp_ctx::ReadRegion& G = ctx->h_read_regions[w];
            size_t used = 0;
            auto room = [&](size_t more) -> bool {              // (grows by half: the places are offsets until the range is done)
                if (used + more + 64 <= G.cap) return true;
                size_t cap = std::max<size_t>(G.cap + G.cap / 2, used + more + 64);
                cap = (cap + 0xfffffu) & ~(size_t)0xfffffu;
                uint8_t* q = (uint8_t*)realloc(G.p, cap);
                if (!q) return false;
                G.p = q; G.cap = cap;
                return true;
            };
            struct Inflator { z_stream zs; bool live = false; ~Inflator() { if (live) inflateEnd(&zs); } };
            static thread_local Inflator I;
            std::vector<uint8_t> plain;
            const uint32_t i0 = (uint32_t)((uint64_t)n * w / workers), i1 = (uint32_t)((uint64_t)n * (w + 1) / workers);
            for (uint32_t i = i0; i < i1; ++i) {
                uint8_t* raw = nullptr; uint64_t raw_len = 0;
                rcs[i] = slurp(paths[i], &raw, &raw_len);
                if (rcs[i]) { errs[i] = spsp_last_error(); continue; }
                const bool gz = raw_len >= 18 && raw[0] == 0x1F && raw[1] == 0x8B;
                const bool zl = !gz && raw_len >= 2 && raw[0] == 0x78 && (raw[1] == 0x01 || raw[1] == 0x9C || raw[1] == 0xDA);
                bool placed = false;
                if (gz) {
                    // one member whose trailer tells the length (a forged one must not size a buffer: deflate never expands 1032-fold):
                    // inflated straight to its place
                    uint32_t isize; memcpy(&isize, raw + raw_len - 4, 4);
                    bool ok = (uint64_t)isize <= 1032ull * raw_len + 64 && raw_len < (1ull << 31) && room(isize);
                    if (ok) {
                        if (!I.live) { memset(&I.zs, 0, sizeof I.zs); ok = inflateInit2(&I.zs, 15 + 16) == Z_OK; I.live = ok; }
                        else ok = inflateReset2(&I.zs, 15 + 16) == Z_OK;
                    }
                    if (ok) {
                        uint8_t spill[8];
                        I.zs.next_in = raw; I.zs.avail_in = (uInt)raw_len;
                        I.zs.next_out = isize ? G.p + used : spill; I.zs.avail_out = isize;
                        int ret = inflate(&I.zs, Z_FINISH);
                        if (ret == Z_BUF_ERROR && I.zs.avail_out == 0) {   // the output is full: the trailer is still to be read
                            I.zs.next_out = spill; I.zs.avail_out = 0;
                            ret = inflate(&I.zs, Z_FINISH);
                        }
                        if (ret == Z_STREAM_END && I.zs.avail_in == 0 && I.zs.total_out == isize) { at[i] = used; lens[i] = isize; placed = true; }
                    }
                }
                if (!placed) {
                    // several members, a trailer that does not tell the truth, a zlib wrapper, a damaged file (the general reader and
                    // its error text), or plain text: appended all the same
                    const uint8_t* src = raw; uint64_t len = raw_len;
                    if (gz || zl) {
                        rcs[i] = inflate_all(raw, raw_len, plain);
                        if (rcs[i]) { errs[i] = spsp_last_error(); free(raw); continue; }
                        src = plain.data(); len = plain.size();
                    }
                    if (!room((size_t)len)) { set_error("out of host memory"); rcs[i] = SPSP_ERR_NOMEM; errs[i] = spsp_last_error(); free(raw); continue; }
                    if (len) memcpy(G.p + used, src, (size_t)len);
                    at[i] = used; lens[i] = len;
                }
                free(raw);
                used = (size_t)((used + lens[i] + 15) & ~(uint64_t)15);
            }
            for (uint32_t i = i0; i < i1; ++i) if (!rcs[i]) { datas[i] = G.p + at[i]; own[i] = 0; }
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < workers; ++w) pool.emplace_back(range, w);
        range(0);
        for (auto& th : pool) th.join();
    } else {
        on_threads([&](uint32_t i) {
            rcs[i] = spsp_read_file_host(paths[i], &datas[i], &lens[i]);
            if (rcs[i]) errs[i] = spsp_last_error();
        });
    }
    int rc = SPSP_OK;
    static const bool load_times = getenv("SPSP_DEBUG_DECODE_TIMES") != nullptr;
    if (load_times) fprintf(stderr, "[load] read + gunzip of %u files on %u threads %.4f s\n", n, workers, now_s() - t0);
    for (uint32_t i = 0; i < n && !rc; ++i)
        if (rcs[i]) { set_error("%s", errs[i].c_str()); rc = rcs[i]; }
    // k and m of the first header (every sketch is checked against them by the decoder)
    uint32_t k0 = 0, m0 = 0;
    if (!rc && n) {
        const uint8_t* nl = (const uint8_t*)memchr(datas[0], '\n', lens[0]);
        long skm = 0, mm = 0;
        if (nl) { char* e = nullptr; const std::string h((const char*)datas[0], nl - datas[0]); skm = strtol(h.c_str(), &e, 10); mm = strtol(e, &e, 10); }
        if (!nl || skm <= 0 || skm > 126 || mm <= 0 || mm > 15 || (skm + mm) / 2 > 63 || (skm + mm) / 2 < mm) { set_error("bad sketch header in '%s'", paths[0]); rc = SPSP_ERR_FORMAT; }
        else { m0 = (uint32_t)mm; k0 = (uint32_t)((skm + mm) / 2); }
    }
    // the merge's shared first-read buffer, in file order (see spsp_sketch_chain_host): phantom keys of empty sketches
    std::vector<int> extra_has(n, 0);
    std::vector<uint32_t> extra_mn(n, 0);
    if (!rc && n) {
        char buffer[16];
        memset(buffer, 'A', sizeof buffer);
        for (uint32_t i = 0; i < n && !rc; ++i) {
            int has = 0; uint32_t mn = 0; uint64_t lo = 0, hi = 0;
            rc = spsp_sketch_chain_host(datas[i], lens[i], k0, m0, buffer, &has, &mn, &lo, &hi);
            if (!rc && has) { extra_has[i] = 1; extra_mn[i] = mn; }
        }
    }
    if (load_times) fprintf(stderr, "[load] ... with the first-read chain %.4f s\n", now_s() - t0);
    // the pair matrix: zero pages from calloc (400 MB at 10^4 sketches: touched only where a row is written or read)
    struct Matrix { uint32_t* p = nullptr; ~Matrix() { free(p); } uint32_t* data() { return p; }
                    int zero(size_t cells) { free(p); p = (uint32_t*)calloc(cells ? cells : 1, 4); return p ? SPSP_OK : SPSP_ERR_NOMEM; } } inter;
    bool mirrored = false;                                    // both triangles filled
    std::vector<uint64_t> cells;                              // ... or no matrix at all: a large comparison comes back as its non-zero cells
    bool as_cells = false;
    std::vector<uint64_t> card(n, 0);
    if (!rc && host_decode) {
        // round-1 path: every sketch decoded and sorted by spsp_sketch_parse_host on the host threads, keys uploaded by spsp_compare
        std::vector<spsp_sketch_view> views(n);
        std::vector<void*> owned((size_t)n * 3, nullptr);
        // decode + sort on the host threads (one sketch per task), then the checks in file order
        std::vector<uint32_t> kks(n, 0), mms(n, 0);
        std::vector<uint64_t> cnts(n, 0);
        {
            std::fill(rcs.begin(), rcs.end(), SPSP_OK);
            std::atomic<uint32_t> next(0);
            auto work = [&]() {
                for (;;) {
                    const uint32_t i = next.fetch_add(1);
                    if (i >= n) break;
                    uint32_t* mn = nullptr; uint64_t *lo = nullptr, *hi = nullptr;
                    rcs[i] = spsp_sketch_parse_host(datas[i], lens[i], &kks[i], &mms[i], &mn, &lo, &hi, &cnts[i]);
                    if (rcs[i]) { errs[i] = spsp_last_error(); continue; }
                    owned[3 * (size_t)i] = mn; owned[3 * (size_t)i + 1] = lo; owned[3 * (size_t)i + 2] = hi;
                }
            };
            std::vector<std::thread> pool;
            for (unsigned w = 1; w < workers; ++w) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            for (uint32_t i = 0; i < n && !rc; ++i)
                if (rcs[i]) { set_error("%s", errs[i].c_str()); rc = rcs[i]; }
        }
        for (uint32_t i = 0; i < n && !rc; ++i) {
            const uint32_t kk = kks[i], mm2 = mms[i];
            uint32_t* mn = (uint32_t*)owned[3 * (size_t)i]; uint64_t *lo = (uint64_t*)owned[3 * (size_t)i + 1], *hi = (uint64_t*)owned[3 * (size_t)i + 2];
            uint64_t cnt = cnts[i];
            if (kk != k0 || mm2 != m0) { set_error("'%s' was sketched with k=%u m=%u, expected k=%u m=%u", paths[i], kk, mm2, k0, m0); rc = SPSP_ERR_FORMAT; break; }
            if (extra_has[i] && cnt == 0) {
                int has = 0; char tmp[16]; memset(tmp, 'A', sizeof tmp);
                // (recomputed from the stored minimizer: k == m, the k-mer is the minimizer's canonical form)
                uint64_t v = extra_mn[i], r = 0;
                for (uint32_t j = 0; j < m0; ++j) r |= (uint64_t)(((v >> (2 * j)) & 3u) ^ 2u) << (2 * (m0 - 1 - j));
                mn[0] = extra_mn[i]; lo[0] = v < r ? v : r; hi[0] = 0; cnt = 1; (void)has; (void)tmp;
            }
            views[i].minimizer = mn; views[i].kmer_lo = lo; views[i].kmer_hi = k0 > 32 ? hi : nullptr; views[i].n = cnt;
        }
        free_datas();
        if (!rc && chatter && n) { printf("kmers evaluated are of length: %u minimizer size is %u\n", k0, m0); fflush(stdout); }   // :56
        t1 = now_s(); ctx->stages.load_s += t1 - t0; t0 = t1;
        if (!rc && (rc = inter.zero((size_t)n * n))) set_error("out of host memory");
        if (!rc) rc = spsp_compare(ctx, views.data(), n, n_query, inter.data(), card.data());
        for (void* p : owned) free(p);
    } else if (!rc) {
        if (chatter && n) { printf("kmers evaluated are of length: %u minimizer size is %u\n", k0, m0); fflush(stdout); }   // :56
        t1 = now_s(); ctx->stages.load_s += t1 - t0; t0 = t1;
        uint32_t kk = 0, mm2 = 0;
        as_cells = n >= 1024 && n <= 65535;                    // (the printers then work from the cells: no n x n matrix on the host)
        if (!as_cells && (rc = inter.zero((size_t)n * n))) set_error("out of host memory");
        else if (n_more > 1) rc = spsp::compare_payloads_multi(more, n_more, datas.data(), lens.data(), n, extra_has.data(), extra_mn.data(), n_query, &kk, &mm2,
                                                               inter.data(), card.data(), &mirrored, as_cells ? &cells : nullptr);
        else rc = spsp::compare_payloads_impl(ctx, datas.data(), lens.data(), n, extra_has.data(), extra_mn.data(), n_query, &kk, &mm2,
                                              inter.data(), card.data(), &mirrored, as_cells ? &cells : nullptr);
    }
    free_datas();
    t1 = now_s(); ctx->stages.compare_s += t1 - t0;
    if (rc) return rc;
    if (chatter) {
        printf("Comparisons done\n");                                                         // :69
        if (chatter == 1) std::cout << "Comparisons lasted " << (t1 - t_start) << " sec" << std::endl;   // :503 (cout's default float format)
        fflush(stdout);
    }
    const double t_middle = t1;
    for (int jac = 0; jac < 2 && !rc; ++jac) {
        char* text = nullptr; uint64_t len = 0;
        t0 = now_s();
        if (chatter) { printf(jac ? "Jackard index dump\n" : "Containement index dump \n"); fflush(stdout); }   // :364, :414
        const std::string out_gz = std::string(out_prefix) + (jac ? "_jaccard.csv.gz" : "_containment.csv.gz");
        if (as_cells) {
            // rows formatted from the cells and deflated block by block on the workers, level 1 (Comparator.cpp:363,413)
            double tt[2] = {0, 0};
            rc = csv_cells_impl(jac, paths, n, n_query, cells, card.data(), precision, min_threshold, nullptr, nullptr, out_gz.c_str(), 1, tt);
            ctx->stages.csv_s += tt[0]; ctx->stages.csv_gzip_s += tt[1];
            continue;
        }
        rc = csv_impl(jac, paths, n, n_query, inter.data(), card.data(), precision, min_threshold, &text, &len, mirrored);
        t1 = now_s(); ctx->stages.csv_s += t1 - t0;
        if (rc) break;
        const std::string out = std::string(out_prefix) + (jac ? "_jaccard.csv.gz" : "_containment.csv.gz");
        rc = spsp_write_gz_host(out.c_str(), (const uint8_t*)text, len, 1);  // level 1: Comparator.cpp:363,413
        ctx->stages.csv_gzip_s += now_s() - t1;
        free(text);
    }
    if (!rc && chatter == 1) std::cout << "Jaccard output lasted " << (now_s() - t_middle) << " sec" << std::endl;   // :509
    return rc;
}

int spsp_compare_files(spsp_ctx* ctx, const char* const* paths, uint32_t n, uint32_t n_query, int precision,
                       double min_threshold, const char* out_prefix) {
    return compare_files_impl(ctx, paths, n, n_query, precision, min_threshold, out_prefix, 0);
}
int spsp_compare_files_chatty(spsp_ctx* ctx, const char* const* paths, uint32_t n, uint32_t n_query, int precision,
                              double min_threshold, const char* out_prefix, int all_versus_all) {
    return compare_files_impl(ctx, paths, n, n_query, precision, min_threshold, out_prefix, all_versus_all ? 1 : 2);
}

int spsp_compare_files_multi(const int* devices, uint32_t n_dev, const char* const* paths, uint32_t n, uint32_t n_query, int precision,
                             double min_threshold, const char* out_prefix, int chatter, spsp_stage_times* times) {
    if (!devices || n_dev == 0 || n_dev > 64 || !paths || !out_prefix) { set_error("1..64 devices, file list and output prefix"); return SPSP_ERR_ARG; }
    std::vector<spsp_ctx*> ctxs;
    int rc = SPSP_OK;
    if (getenv("SPSP_DEBUG_MULTI_TRACE")) {                   // which devices a CLI run chose (tests/test_multi_device.py)
        std::string devs;
        for (uint32_t d = 0; d < n_dev; ++d) devs += (d ? "," : "") + std::to_string(devices[d]);
        fprintf(stderr, "spsp multi: %u contexts on devices %s, %u sketches\n", n_dev, devs.c_str(), n);
    }
    for (uint32_t d = 0; d < n_dev && !rc; ++d) {
        spsp_ctx* c = nullptr;
        rc = spsp_create(devices[d], nullptr, &c);
        if (!rc) ctxs.push_back(c);
    }
    if (!rc) rc = compare_files_impl(ctxs[0], paths, n, n_query, precision, min_threshold, out_prefix, chatter < 0 ? 0 : (chatter > 2 ? 2 : chatter), ctxs.data(), (uint32_t)ctxs.size());
    if (!rc && times) *times = ctxs[0]->stages;
    const std::string err = rc ? spsp_last_error() : "";
    for (spsp_ctx* c : ctxs) spsp_destroy(c);
    if (rc) set_error("%s", err.c_str());
    return rc;
}

}  // extern "C"
