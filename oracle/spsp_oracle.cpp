// =============================================================================
//  oracle/spsp_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE
// =============================================================================
//  A single-threaded CPU restatement of the reference algorithm for the hot
//  path named in BASELINE.json.  It is
//  the checker that tests/, __graft_entry__.smoke() and bench.py's
//  `cpu_baseline` leg compare the HIP product against.  Nothing under
//  supersampler_amd/ includes, links or calls this file.
//
//  PARITY STATUS: "parity unpinned" by the reference.  The reference ships no
//  tests, golden vectors or fixtures for this path and
//  executing the compiled reference was denied by the environment in the
//  survey session, so it is never run.  This
//  restatement is instead pinned by first-principle known answers
//  (third-party python-xxhash vectors, hand-derived 2-bit packings, brute
//  force window minima and Python set intersections) in tests/test_oracle.py.
//
//  Every function cites the reference file:line it follows (paths relative to
//  /root/reference).  Undefined behaviour in the reference is resolved as
//  SURVEY.md hazards H1..H4 say (noted inline).
// =============================================================================
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

typedef unsigned __int128 u128;

namespace {

// ---------------------------------------------------------------- A1 codec --
// utils.cpp:13-16  nuc2int: (c/2)%4  => A=0 C=1 T=2 G=3
inline uint64_t nuc2int(char c) { return (uint64_t)(((unsigned char)c / 2) % 4); }
// utils.cpp:20-22  nuc2intrc: complement = code ^ 2
inline uint64_t nuc2intrc(char c) { return nuc2int(c) ^ 2; }
// utils.cpp:26-45
inline char int2nuc(unsigned n) {
    switch (n & 3) { case 0: return 'A'; case 1: return 'C'; case 2: return 'T'; default: return 'G'; }
}
// utils.cpp:158-165
u128 str2num(const char* s, uint64_t len) {
    u128 r = 0;
    for (uint64_t i = 0; i < len; ++i) { r <<= 2; r += nuc2int(s[i]); }
    return r;
}
// utils.cpp:168-183
std::string num2str(u128 num, uint64_t len) {
    std::string s(len, 'A');
    for (uint64_t i = 0; i < len; ++i) { s[len - 1 - i] = int2nuc((unsigned)(num & 3)); num >>= 2; }
    return s;
}
// utils.cpp:131-148
inline char revCompChar(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; }
    return 'A';
}
std::string revComp(const std::string& s) {
    std::string rc(s.size(), 0);
    for (size_t i = 0; i < s.size(); ++i) rc[s.size() - 1 - i] = revCompChar(s[i]);
    return rc;
}

// ------------------------------------------------------------ A2 canonical --
inline uint64_t flip_pairs64(uint64_t x) {
    // utils.cpp:452-457 -- byte swap, nibble swap, 2-bit pair swap
    uint64_t r = __builtin_bswap64(x);
    const uint64_t c1 = 0x0f0f0f0f0f0f0f0fULL, c2 = 0x3333333333333333ULL;
    r = ((r & c1) << 4) | ((r & (c1 << 4)) >> 4);
    r = ((r & c2) << 2) | ((r & (c2 << 2)) >> 2);
    return r;
}
// utils.cpp:449-462  rcbc
inline uint64_t rc64(uint64_t in, uint64_t n) {
    return flip_pairs64(in ^ 0xaaaaaaaaaaaaaaaaULL) >> (64 - 2 * n);
}
// utils.cpp:465-467
inline uint64_t canon64(uint64_t x, uint64_t n) { return std::min(x, rc64(x, n)); }
// utils.cpp:397-438  rcb (128-bit; SSSE3 shuffle there, plain integer here)
inline u128 rc128(u128 in, uint64_t n) {
    uint64_t lo = (uint64_t)in, hi = (uint64_t)(in >> 64);
    uint64_t nlo = flip_pairs64(hi) ^ 0xaaaaaaaaaaaaaaaaULL;
    uint64_t nhi = flip_pairs64(lo) ^ 0xaaaaaaaaaaaaaaaaULL;
    u128 r = ((u128)nhi << 64) | nlo;
    return r >> (128 - 2 * n);
}
// utils.cpp:470-472
inline u128 canon128(u128 x, uint64_t n) { return std::min(x, rc128(x, n)); }

// ----------------------------------------------------------------- A3 hash --
// include/xxhash64.h:100-150,158-163,167-171,182-191 specialised to 8 input
// bytes (SubSampler.cpp:64-67: XXHash64::hash(&x, 8, 1312); the code after the
// first `return` there is dead).
const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL,
               P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL,
               P5 = 2870177450012600261ULL;
inline uint64_t rotl(uint64_t x, unsigned b) { return (x << b) | (x >> (64 - b)); }
inline uint64_t xxh64_u64(uint64_t x, uint64_t seed) {
    uint64_t r = seed + P5;                // xxhash64.h:118 (state[2] holds the seed)
    r += 8;                                // :121 totalLength
    uint64_t single = rotl(0 + x * P2, 31) * P1;  // :188-191 processSingle(0, x)
    r = rotl(r ^ single, 27) * P1 + P4;    // :130
    r ^= r >> 33; r *= P2; r ^= r >> 29; r *= P3; r ^= r >> 32;  // :144-148
    return r;
}
inline uint64_t unrevhash(uint64_t x) { return xxh64_u64(x, 1312); }

// ------------------------------------------------------------ A4 threshold --
// SubSampler.cpp:622-631 and the constructor's selection SubSampler.h:79-83
uint64_t compute_threshold(uint64_t k, uint64_t m, double sampling_rate) {
    if (!(sampling_rate > 1)) return (uint64_t)-1;  // SubSampler.h:79-83
    uint64_t mmerinkmer = k - m + 1;
    long double fraction_sampling = (long double)1 / sampling_rate;
    long double root = powl((long double)1 - fraction_sampling, (long double)1 / mmerinkmer);
    long double result = ((long double)1 - root) * ((uint64_t)1 << 63);
    return (uint64_t)result * 2;
}

// ------------------------------------------------------------------ params --
struct Params {
    uint64_t k, m;
    uint64_t threshold;
    uint64_t abundance;
    double rate;
    uint64_t mmask;  // offsetUpdateMinimizer SubSampler.h:74
    u128 kmask;      // offsetUpdateAnchor    SubSampler.h:73
};
Params make_params(uint64_t k, uint64_t m, double s, uint64_t abundance) {
    Params p;
    p.k = k; p.m = m; p.rate = s; p.abundance = abundance;
    p.threshold = compute_threshold(k, m, s);
    p.mmask = ((uint64_t)1 << (2 * m)) - 1;
    p.kmask = (((u128)1) << (2 * k)) - 1;
    return p;
}

// ------------------------------------------------------------- A5 rescan ----
// SubSampler.cpp:81-169  regular_minimizer_pos: right-to-left over the k-mer.
uint64_t regular_minimizer_pos(const Params& P, u128 seq, uint64_t& position, bool& is_rev) {
    const uint64_t k = P.k, m = P.m;
    is_rev = false;
    uint64_t mmer = (uint64_t)(seq & P.mmask);
    uint64_t mini = canon64(mmer, m);
    position = k - m;
    if (mini != mmer) { is_rev = true; position = 0; }  // :89-93 (sic: 0, not k-m)
    mmer = mini;
    uint64_t hash_mini = unrevhash(mmer);
    for (uint64_t i = 1; i <= k - m; i++) {
        seq >>= 2;
        mmer = (uint64_t)(seq & P.mmask);
        uint64_t canon_mmer = canon64(mmer, m);
        bool local_rev = !(canon_mmer == mmer);
        mmer = canon_mmer;
        uint64_t h = unrevhash(mmer);
        if (hash_mini > h) {                             // :117-129
            position = k - m - i; mini = mmer; is_rev = local_rev; hash_mini = h;
        } else if (mmer == mini) {                       // :132-166
            if (local_rev != is_rev) {
                // different reading orders: keep the first one found (:136-148)
            } else {
                if (is_rev && position > i) {            // :151-157 (sic: position = i)
                    position = i; mini = mmer; is_rev = local_rev; hash_mini = h;
                }
                if (!is_rev && position > k - m - i) {   // :158-164 leftmost forward copy
                    position = k - m - i; mini = mmer; is_rev = local_rev; hash_mini = h;
                }
            }
        }
    }
    return mini;
}

// -------------------------------------------------------------- A6 scan -----
struct Emit { uint32_t rec; uint32_t minimizer; uint64_t start; uint32_t len; uint32_t rev; };
struct ScanStats {
    uint64_t read_kmer, total_kmer_number, total_superkmer_number;
    uint64_t nb_mmer_selected;
};

// SubSampler.cpp:350-455 for one cleaned record `ref` of length n.
void scan_record(const Params& P, const char* ref, uint64_t n, uint32_t rec,
                 std::vector<Emit>& out, ScanStats& st) {
    const uint64_t k = P.k, m = P.m;
    if (n < k) return;                                   // :340-343
    st.read_kmer += n - k + 1;                           // :346
    bool is_rev = false, old_rev = false, dump = false;  // :353, H3: dump=false
    uint32_t old_minimizer, minimizer;
    uint64_t last_position = 0, position_min = 0, i = 0, pos_end = 0;
    u128 seq = str2num(ref, k);                          // :359
    uint64_t min_seq = (uint64_t)str2num(ref + (k - m), m);      // :360
    uint64_t min_rcseq = rc64(min_seq, m);                       // :361
    uint64_t min_canon = std::min(min_seq, min_rcseq);           // :362
    minimizer = (uint32_t)regular_minimizer_pos(P, seq, position_min, old_rev);  // :363
    old_minimizer = minimizer;
    uint64_t hash_min = unrevhash(minimizer);            // :365
    for (; i + k < n; ++i) {                             // :367
        char c = ref[i + k];
        seq = ((seq << 2) + nuc2int(c)) & P.kmask;                       // :29-34
        min_seq = ((min_seq << 2) + nuc2int(c)) & P.mmask;               // :36-41
        min_rcseq = (min_rcseq >> 2) + (nuc2intrc(c) << (2 * m - 2));    // :49-53
        min_canon = std::min(min_seq, min_rcseq);        // :372
        uint64_t new_h = unrevhash(min_canon);           // :373
        if (new_h < hash_min) {                          // :374-388
            minimizer = (uint32_t)min_canon;
            hash_min = new_h;
            position_min = i + k - m + 1;
            is_rev = !(min_canon == min_seq);
        } else if (i >= position_min) {                  // :391-398
            minimizer = (uint32_t)regular_minimizer_pos(P, seq, position_min, is_rev);
            dump = true;
            hash_min = unrevhash(minimizer);
            position_min += (i + 1);
        }
        if (old_minimizer != minimizer || dump) {        // :401
            dump = false;
            if (unrevhash(old_minimizer) <= P.threshold) {   // :405
                if (last_position + m - 2 > pos_end) {       // :410-419
                    if (pos_end > 0) st.nb_mmer_selected -= m - 1;
                    st.nb_mmer_selected += i + k - last_position;
                    st.nb_mmer_selected -= k - m;
                } else {
                    st.nb_mmer_selected += i + k - (pos_end + 1);    // :423
                }
                Emit e; e.rec = rec; e.start = last_position; e.len = (uint32_t)(i + k - last_position);
                e.minimizer = old_minimizer; e.rev = old_rev ? 1 : 0;
                out.push_back(e);                        // :426 handle_superkmer(...)
                pos_end = i + k - 1;
            }
            st.total_kmer_number += (i - last_position + 1);  // :429
            st.total_superkmer_number++;
            last_position = i + 1;
            old_minimizer = minimizer;
            old_rev = is_rev;
        }
    }
    if (n - last_position > k - 1) {                     // :441
        if (unrevhash(old_minimizer) <= P.threshold) {   // :443
            st.nb_mmer_selected -= m - 1;
            Emit e; e.rec = rec; e.start = last_position; e.len = (uint32_t)(i + k - last_position);
            e.minimizer = old_minimizer; e.rev = old_rev ? 1 : 0;
            out.push_back(e);                            // :448
        }
        st.total_kmer_number += (i - last_position + 1); // :451
        st.total_superkmer_number++;
    }
}

// ------------------------------------------------------- A7 k-mer buckets ---
struct KInfo { uint8_t count; uint8_t pos_min; bool seen; };   // SubSampler.h:23-27
struct U128Hash {
    size_t operator()(const u128& x) const {
        uint64_t a = (uint64_t)x, b = (uint64_t)(x >> 64);
        a ^= b + 0x9e3779b97f4a7c15ULL + (a << 6) + (a >> 2);
        a *= 0xff51afd7ed558ccdULL; a ^= a >> 33;
        return (size_t)a;
    }
};
// Insertion-ordered map: stands in for ankerl::unordered_dense::map whose
// iteration order is insertion order (include/unordered_dense.h:429,884-897;
// nothing is ever erased from it on this path).
struct Bucket {
    std::vector<u128> keys;
    std::vector<KInfo> info;
    std::unordered_map<u128, uint32_t, U128Hash> index;
    int find(u128 key) const {
        auto it = index.find(key);
        return it == index.end() ? -1 : (int)it->second;
    }
};
struct SketchStats {
    uint64_t selected_kmer_number, selected_superkmer_number, count_maximal_skmer;
    uint64_t seen_kmers_at_reconstruction, seen_superkmers_at_reconstruction;
    uint64_t seen_max_superkmers_at_reconstruction, seen_unique_kmers_at_reconstruction;
    uint64_t total_kmer_number_at_reconstruction, actual_minimizer_number;
};
typedef std::map<uint32_t, Bucket> MinimizerMap;         // SubSampler.h:62

// SubSampler.cpp:243-302
void handle_superkmer(const Params& P, MinimizerMap& mm, SketchStats& st,
                      std::string superkmer, uint32_t input_minimizer, bool inputrev) {
    const uint64_t k = P.k, m = P.m;
    st.selected_superkmer_number++;
    if (inputrev) superkmer = revComp(superkmer);        // :246-249
    st.selected_kmer_number += superkmer.size() - k + 1; // :250
    if (superkmer.size() == 2 * k - m) st.count_maximal_skmer++;
    const std::string minstr = num2str(input_minimizer, m);
    for (uint64_t i = 0; i + k <= superkmer.size(); ++i) {
        std::string kmerstr = superkmer.substr(i, k);
        uint64_t position_min = kmerstr.find(minstr);    // :264 (npos -> "PB"; never blocks here)
        u128 seq = str2num(kmerstr.data(), k);
        Bucket& b = mm[input_minimizer];                 // :274-300
        int at = b.find(seq);
        if (at >= 0) {
            b.info[at].count++;                          // uint8_t wraps (H4)
        } else {
            KInfo ki; ki.count = 1; ki.pos_min = (uint8_t)position_min; ki.seen = false;  // H2
            b.index[seq] = (uint32_t)b.keys.size();
            b.keys.push_back(seq); b.info.push_back(ki);
        }
    }
}

// SubSampler.cpp:604-620
bool find_first_kmer(const Params& P, Bucket& b, SketchStats& st, u128& out) {
    for (size_t j = 0; j < b.keys.size(); ++j) {
        if (!b.info[j].seen && b.info[j].count >= P.abundance) {
            st.total_kmer_number_at_reconstruction += b.info[j].count;
            st.seen_unique_kmers_at_reconstruction++;
            b.info[j].seen = true;
            out = b.keys[j];
            return true;
        }
    }
    return false;  // reference returns kmer(-1)
}
// SubSampler.cpp:566-602
u128 find_next(const Params& P, Bucket& b, SketchStats& st, u128 start, bool left) {
    const char nucs[] = {'A', 'T', 'C', 'G'};            // :568
    const uint64_t k = P.k;
    for (char nuc : nucs) {
        u128 next = start;
        if (left) { next >>= 2; next += (u128)nuc2int(nuc) << ((2 * k) - 2); }
        else { next <<= 2; next += nuc2int(nuc); next &= P.kmask; }
        int at = b.find(next);
        if (at >= 0 && !b.info[at].seen && b.info[at].count >= P.abundance) {
            b.info[at].seen = true;
            st.seen_unique_kmers_at_reconstruction++;
            st.total_kmer_number_at_reconstruction += b.info[at].count;
            return next;
        }
    }
    return start;
}
// SubSampler.cpp:512-564
std::string reconstruct_superkmer(const Params& P, Bucket& b, SketchStats& st, u128 start) {
    const uint64_t k = P.k, m = P.m;
    std::string superkmer = num2str(start, k);
    uint64_t pm = b.info[b.find(start)].pos_min;
    uint64_t n_left = (k - m) - pm, n_right = pm;
    u128 next, n_start = start;
    while (superkmer.size() != (k * 2 - m)) {
        if (n_left != 0) {
            next = find_next(P, b, st, n_start, true);
            n_left -= 1;
            if (next != n_start) superkmer.insert(superkmer.begin(), num2str(next, k)[0]);
            else n_left = 0;
            if (n_left == 0) n_start = start; else n_start = next;
        } else if (n_right != 0) {
            next = find_next(P, b, st, n_start, false);
            n_right -= 1;
            if (next != n_start) superkmer.push_back(int2nuc((unsigned)(next & 3)));
            else break;
            n_start = next;
        } else break;
    }
    return superkmer;
}

// utils.cpp:48-68  strCompressor (H1: the accumulator starts at 0)
std::string strCompressor(const std::string& str) {
    std::string result;
    if (str.empty()) return result;
    char mod = (char)(str.size() % 4);
    result += mod;
    unsigned char c = 0;
    for (uint64_t i = 0; i < str.size(); ++i) {
        c += (unsigned char)nuc2int(str[i]);
        if ((i + 1) % 4 == 0) { result += (char)c; c = 0; }
        c <<= 2;
    }
    if (mod != 0) result += (char)c;
    return result;
}
// utils.cpp:71-111  strDecompressor
std::string strDecompressor(const std::string& str) {
    std::string result;
    if (str.empty()) return result;
    char mod = str[0];
    uint64_t last = (mod == 0) ? str.size() : str.size() - 1;
    char f[4];
    for (uint64_t i = 1; i < last; ++i) {
        unsigned char p = (unsigned char)str[i];
        f[3] = int2nuc(p % 4); p >>= 2; f[2] = int2nuc(p % 4); p >>= 2;
        f[1] = int2nuc(p % 4); p >>= 2; f[0] = int2nuc(p % 4);
        result.append(f, 4);
    }
    if (mod != 0) {
        unsigned char p = (unsigned char)str[last];
        for (int i = 0; i < mod + 1; ++i) { f[mod - i] = int2nuc(p % 4); p >>= 2; }
        for (int i = 0; i < mod; ++i) result += f[i];
    }
    return result;
}

// SubSampler.cpp:458-504: header + buckets -> uncompressed sketch payload
std::string emit_sketch(const Params& P, MinimizerMap& mm, SketchStats& st) {
    const uint64_t k = P.k, m = P.m;
    std::string out;
    out += std::to_string(k - 1 + (k - m + 1)) + " " + std::to_string(m) + " " +
           std::to_string(st.selected_kmer_number) + " " + std::to_string(P.rate) + "\n";  // :459
    for (auto& kv : mm) {                                // ascending minimizer (std::map)
        Bucket& b = kv.second;
        std::string minstr = num2str(kv.first, m);
        out += minstr;                                   // :466
        uint64_t i = 0;
        std::string max_skmers, skmers;
        st.seen_kmers_at_reconstruction += b.keys.size();
        while (i <= b.keys.size()) {                     // :470
            u128 start;
            if (!find_first_kmer(P, b, st, start)) break;
            std::string s = reconstruct_superkmer(P, b, st, start);
            if (s.size() == (k * 2 - m)) {               // :479-485
                i += (k - m + 1);
                st.seen_max_superkmers_at_reconstruction++;
                max_skmers += s.substr(0, k - m);
                max_skmers += s.substr(k, k - m);
            } else {                                     // :486-494
                i += (s.size() - k + 1);
                uint64_t p = s.find(minstr);
                skmers += s.substr(0, p); skmers += "\n";
                skmers += s.substr(p + m); skmers += "\n";
            }
            st.seen_superkmers_at_reconstruction++;
        }
        std::string compressed = strCompressor(max_skmers);
        uint32_t size_compressed = (uint32_t)compressed.size();
        out.append((const char*)&size_compressed, 4);    // :499-500 little-endian host
        out += compressed;
        out += skmers;
        out += "\n\n";
    }
    st.actual_minimizer_number = mm.size();
    return out;
}

// ------------------------------------------------------------ A9 ingest -----
// utils.cpp:675-702 clean_dna
void clean_dna(std::string& s) {
    std::string r;
    r.reserve(s.size());
    for (char c : s) {
        switch (c) {
            case 'a': case 'A': case 'c': case 'C': case 'g': case 'G': case 't': case 'T':
                r.push_back((char)toupper((unsigned char)c)); break;
            default: break;
        }
    }
    s.swap(r);
}
// utils.cpp:706-718 getLineFasta + the caller loop SubSampler.cpp:334-348,
// replayed over an in-memory (already gunzipped) buffer with iostream
// eof/peek semantics.
void split_fasta(const char* buf, uint64_t n, std::vector<std::string>& records) {
    uint64_t pos = 0;
    bool eof = false;
    auto getline_ = [&](std::string* dst) {
        if (pos >= n) { eof = true; return; }
        uint64_t e = pos;
        while (e < n && buf[e] != '\n') ++e;
        if (dst) dst->append(buf + pos, e - pos);
        if (e >= n) { eof = true; pos = n; } else pos = e + 1;
    };
    while (!eof) {
        std::string result;
        getline_(nullptr);                               // header line dropped (:708)
        for (;;) {
            if (eof || pos >= n) { eof = true; break; }  // peek() == EOF
            char c = buf[pos];
            if (c == '>' || c == (char)0xFF) break;      // :710 (char)EOF aliasing
            getline_(&result);
        }
        clean_dna(result);
        records.push_back(result);
    }
}

// -------------------------------------------------------- A10-A15 compare ---
struct Cursor {       // minimal istream stand-in over a gunzipped sketch payload
    const char* p; uint64_t n, pos; bool eof;
    void read(char* dst, uint64_t len) {
        uint64_t take = std::min(len, n - pos);
        memcpy(dst, p + pos, take); pos += take;
        if (take < len) eof = true;
    }
    std::string getline() {
        std::string s;
        if (pos >= n) { eof = true; return s; }
        uint64_t e = pos;
        while (e < n && p[e] != '\n') ++e;
        s.assign(p + pos, e - pos);
        if (e >= n) { eof = true; pos = n; } else pos = e + 1;
        return s;
    }
};

struct Comparator {
    uint64_t skmer_size = 0, k = 0, m = 0, nb_files = 0, nb_files_eof = 0, query_size = 0;
    std::vector<uint64_t> minimizers, nb_kmer_seen_infile;
    bool run = true;
    std::unordered_map<uint32_t, uint32_t> score_A;      // Comparator.h:26
    std::vector<Cursor> files;

    // Comparator.cpp:23-37
    void get_header_info() {
        for (auto& f : files) {
            std::string header = f.getline();
            long v[4] = {0, 0, 0, 0};
            size_t at = 0;
            for (int j = 0; j < 4; ++j) {
                v[j] = strtol(header.c_str() + at, nullptr, 10);
                size_t sp = header.find(' ', at);
                if (sp == std::string::npos) break;
                at = sp + 1;
            }
            skmer_size = (uint64_t)v[0]; m = (uint64_t)v[1];
            k = (skmer_size + m) / 2;                    // :34
            skmer_size -= m;                             // :35
        }
    }
    // Comparator.cpp:78-92
    std::string inject_minimizer(const std::string& str, const std::string& minstr) {
        std::string result;
        if (!str.empty()) {
            uint64_t half = skmer_size / 2;
            for (uint64_t i = 0; i < str.size(); i += half) {
                result += str.substr(i, half);
                i += half;
                result += minstr;
                if (i <= str.size()) result += str.substr(i, half);
            }
        } else result = minstr;
        return result;
    }
    // Comparator.cpp:291-323
    void increment_files(const std::vector<uint64_t>& indices) {
        std::string buffer(m, 'A');
        if (!indices.empty()) {
            for (uint64_t idx : indices) {
                if (!files[idx].eof) {
                    files[idx].read(&buffer[0], m);
                    if (!files[idx].eof) minimizers[idx] = (uint64_t)str2num(buffer.data(), m);
                    else { minimizers[idx] = (uint64_t)-1; nb_files_eof++; }
                } else { minimizers[idx] = (uint64_t)-1; nb_files_eof++; }
            }
            if (nb_files_eof == nb_files) run = false;
        } else {
            for (uint64_t i = 0; i < files.size(); ++i) {
                files[i].read(&buffer[0], m);
                minimizers[i] = (uint64_t)str2num(buffer.data(), m);
            }
        }
    }
    // Comparator.cpp:328-359
    bool findMin(std::vector<uint64_t>& min_vector) {
        uint64_t mn = (uint64_t)-1;
        bool result = false;
        min_vector.clear();
        for (uint64_t i = 0; i < minimizers.size(); ++i) {
            if (minimizers[i] < mn) {
                mn = minimizers[i]; min_vector.clear(); min_vector.push_back(i);
                result = (i < query_size);
            } else if (minimizers[i] == mn) {
                min_vector.push_back(i);
                if (i < query_size) result = true;
            }
        }
        if (mn == (uint64_t)-1) { run = false; min_vector.clear(); }
        return result;
    }
    // shared enumeration of one (file, bucket): Comparator.cpp:104-149 / :186-260.
    // fn(canon) is called once per k-mer occurrence, maximal part first.
    template <class F> void walk_bucket(uint64_t ind, const std::string& minstr, F fn) {
        Cursor& f = files[ind];
        uint32_t size_buffer = 0;
        f.read((char*)&size_buffer, 4);
        std::string ref;
        uint64_t avail = std::min<uint64_t>(size_buffer, f.n - f.pos);
        ref.resize(avail);                               // a short read sets eof, as istream::read would
        f.read(&ref[0], avail);
        if (avail < size_buffer) f.eof = true;
        ref = strDecompressor(ref);
        ref = inject_minimizer(ref, minstr);
        if (ref.size() < k) ref.clear();
        if (!ref.empty()) {
            uint64_t i = 0;
            while (i + k <= ref.size()) {
                u128 cur = str2num(ref.data() + i, k - 1);
                for (uint64_t j = 0; j < k - m + 1; ++j) {
                    cur = ((cur << 2) + nuc2int(ref.at(i + k - 1))) % ((u128)1 << (2 * k));  // utils.cpp:752-756
                    fn(canon128(cur, k));
                    i++;
                }
                i += k - 1;
            }
        }
        for (;;) {
            std::string s1 = f.getline(), s2 = f.getline();
            if (s1.empty() && s2.empty()) break;
            s1 += minstr + s2;
            u128 cur = str2num(s1.data(), std::min<uint64_t>(k - 1, s1.size()));
            uint64_t i = 0;
            while (i + k <= s1.size()) {
                cur = ((cur << 2) + nuc2int(s1[i + k - 1])) % ((u128)1 << (2 * k));
                fn(canon128(cur, k));
                ++i;
            }
        }
    }
    // Comparator.cpp:97-154
    void skip_bucket(const std::vector<uint64_t>& indices, const std::string& minstr) {
        for (uint64_t ind : indices) {
            std::unordered_map<u128, char, U128Hash> skip_map;
            walk_bucket(ind, minstr, [&](u128 c) { skip_map[c] = 1; });
            nb_kmer_seen_infile[ind] += skip_map.size();
        }
    }
    // Comparator.cpp:177-264 + compute_scores :269-287
    void count_intersection(const std::vector<uint64_t>& indices, const std::string& minstr) {
        std::unordered_map<u128, std::vector<bool>, U128Hash> color_map;
        std::vector<u128> interesting_hits;
        const uint64_t nf = files.size();
        for (uint64_t ind : indices) {
            walk_bucket(ind, minstr, [&](u128 canon) {
                auto it = color_map.find(canon);
                if (it == color_map.end()) {
                    nb_kmer_seen_infile[ind]++;
                    std::vector<bool>& v = color_map[canon];
                    v.resize(nf + 1, false);
                    v[ind] = true;
                } else if (!it->second[ind]) {
                    nb_kmer_seen_infile[ind]++;
                    it->second[ind] = true;
                    if (!it->second[nf]) { interesting_hits.push_back(canon); it->second[nf] = true; }
                }
            });
        }
        std::vector<uint32_t> ones;
        for (const u128& h : interesting_hits) {
            const std::vector<bool>& v = color_map.at(h);
            ones.clear();
            for (uint32_t i = 0; i < nb_files; ++i) if (v[i]) ones.push_back(i);
            for (size_t i = 0; i < ones.size(); ++i)
                for (size_t j = i + 1; j < ones.size(); ++j)
                    score_A[(uint32_t)(ones[i] * nb_files + ones[j])]++;
        }
    }
    // Comparator.cpp:39-74
    void compare_sketches(uint64_t size_query) {
        query_size = size_query;
        nb_files = files.size();
        nb_kmer_seen_infile.assign(nb_files, 0);
        minimizers.assign(nb_files, 0);
        std::vector<uint64_t> indices;
        get_header_info();
        increment_files(indices);
        while (run) {
            bool queryfound = findMin(indices);
            if (indices.empty()) break;                  // reference would index [0]; run is false here
            std::string minstr = num2str(minimizers[indices[0]], m);
            if (indices.size() == 1 || !queryfound) skip_bucket(indices, minstr);
            else count_intersection(indices, minstr);
            increment_files(indices);
        }
    }
};

// Comparator.cpp:362-408 (containment) / :412-460 (jaccard). Exact IEEE
// division (H7: the reference's -Ofast may use a reciprocal).
std::string print_matrix(bool jaccard, const std::vector<std::string>& names, uint64_t nb_files,
                         uint64_t query_size, const uint32_t* inter, const uint64_t* card,
                         int precision, double min_threshold) {
    std::ostringstream out;
    for (uint32_t i = 0; i < nb_files; ++i) {
        out << names[i];
        if (i != nb_files - 1) out << ','; else out << '\n';
    }
    if (!jaccard) out << "\n";                           // :373 extra blank line
    for (uint32_t i = 0; i < nb_files && i < query_size; ++i) {
        for (uint32_t j = 0; j < nb_files; ++j) {
            if (i == j) out << "1";
            else {
                uint32_t a = std::min(i, j), b = std::max(i, j);
                uint32_t sc = inter[(uint64_t)a * nb_files + b];
                if (sc == 0) out << "0";
                else {
                    double score = jaccard
                        ? (double)sc / (double)(card[i] + card[j] - sc)
                        : (double)sc / (double)card[i];
                    if (score < min_threshold) out << '0';
                    else out << std::setprecision(precision) << score;
                }
            }
            if (j != nb_files - 1) out << ','; else out << '\n';
        }
    }
    return out.str();
}

char* dup_bytes(const std::string& s, uint64_t* len) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size()); p[s.size()] = 0;
    if (len) *len = s.size();
    return p;
}

}  // namespace

// ============================================================================
//  extern "C" surface used by tests/ (ctypes) and bench.py's cpu_baseline leg
// ============================================================================
extern "C" {

uint64_t orc_xxh64_u64(uint64_t x, uint64_t seed) { return xxh64_u64(x, seed); }
uint64_t orc_threshold(uint32_t k, uint32_t m, double s) { return compute_threshold(k, m, s); }
uint64_t orc_rc64(uint64_t x, uint32_t n) { return rc64(x, n); }
uint64_t orc_canon64(uint64_t x, uint32_t n) { return canon64(x, n); }
void orc_canon128(uint64_t lo, uint64_t hi, uint32_t n, uint64_t* olo, uint64_t* ohi) {
    u128 c = canon128(((u128)hi << 64) | lo, n);
    *olo = (uint64_t)c; *ohi = (uint64_t)(c >> 64);
}
uint64_t orc_str2num64(const char* s, uint32_t n) { return (uint64_t)str2num(s, n); }
void orc_free(void* p) { free(p); }

char* orc_compress(const char* s, uint64_t n, uint64_t* out_len) {
    return dup_bytes(strCompressor(std::string(s, n)), out_len);
}
char* orc_decompress(const char* s, uint64_t n, uint64_t* out_len) {
    return dup_bytes(strDecompressor(std::string(s, n)), out_len);
}

// rescan of one k-mer given as ASCII (A5); returns minimizer, writes position / rev
uint64_t orc_rescan(uint32_t k, uint32_t m, const char* kmer_ascii, uint64_t* position, uint32_t* rev) {
    Params P = make_params(k, m, 1000, 1);
    bool r; uint64_t pos;
    uint64_t mini = regular_minimizer_pos(P, str2num(kmer_ascii, k), pos, r);
    *position = pos; *rev = r ? 1 : 0;
    return mini;
}

// FASTA text (gunzipped) -> concatenated cleaned records + offsets (A9).
// Returns number of records; *bases / *offsets are malloc'd (n_rec+1 offsets).
uint64_t orc_clean_fasta(const char* text, uint64_t n, char** bases, uint64_t** offsets) {
    std::vector<std::string> recs;
    split_fasta(text, n, recs);
    uint64_t total = 0;
    for (auto& r : recs) total += r.size();
    *bases = (char*)malloc(total + 1);
    *offsets = (uint64_t*)malloc(sizeof(uint64_t) * (recs.size() + 1));
    uint64_t at = 0;
    for (size_t i = 0; i < recs.size(); ++i) {
        (*offsets)[i] = at;
        memcpy(*bases + at, recs[i].data(), recs[i].size());
        at += recs[i].size();
    }
    (*offsets)[recs.size()] = at;
    return recs.size();
}

struct orc_superkmer { uint32_t rec; uint32_t minimizer; uint64_t start; uint32_t len; uint32_t rev; };
struct orc_scan_stats { uint64_t read_kmer, total_kmer_number, total_superkmer_number, nb_mmer_selected; };

// Literal scan (A5-A6) over cleaned records.  threshold given explicitly so
// tests can force dense selection.  Returns count; *out malloc'd.
uint64_t orc_scan(uint32_t k, uint32_t m, uint64_t threshold, const char* bases,
                  const uint64_t* rec_off, uint32_t n_rec, orc_superkmer** out, orc_scan_stats* stats) {
    Params P = make_params(k, m, 1000, 1);
    P.threshold = threshold;
    std::vector<Emit> em;
    ScanStats st; memset(&st, 0, sizeof(st));
    for (uint32_t r = 0; r < n_rec; ++r)
        scan_record(P, bases + rec_off[r], rec_off[r + 1] - rec_off[r], r, em, st);
    st.nb_mmer_selected -= m - 1;                        // SubSampler.cpp:458
    if (stats) {
        stats->read_kmer = st.read_kmer; stats->total_kmer_number = st.total_kmer_number;
        stats->total_superkmer_number = st.total_superkmer_number; stats->nb_mmer_selected = st.nb_mmer_selected;
    }
    orc_superkmer* o = (orc_superkmer*)malloc(sizeof(orc_superkmer) * (em.size() + 1));
    for (size_t i = 0; i < em.size(); ++i) {
        o[i].rec = em[i].rec; o[i].minimizer = em[i].minimizer; o[i].start = em[i].start;
        o[i].len = em[i].len; o[i].rev = em[i].rev;
    }
    *out = o;
    return em.size();
}

// Timed scan for bench.py's cpu_baseline: returns seconds spent in the scan
// loop only (A1-A6), k-mers seen in *kmers.
double orc_scan_timed(uint32_t k, uint32_t m, uint64_t threshold, const char* bases,
                      const uint64_t* rec_off, uint32_t n_rec, uint64_t* kmers, uint64_t* n_emit) {
    Params P = make_params(k, m, 1000, 1);
    P.threshold = threshold;
    std::vector<Emit> em;
    ScanStats st; memset(&st, 0, sizeof(st));
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 0; r < n_rec; ++r)
        scan_record(P, bases + rec_off[r], rec_off[r + 1] - rec_off[r], r, em, st);
    auto t1 = std::chrono::steady_clock::now();
    *kmers = st.read_kmer; *n_emit = em.size();
    return std::chrono::duration<double>(t1 - t0).count();
}

struct orc_sketch_stats {
    uint64_t selected_kmer_number, selected_superkmer_number, count_maximal_skmer;
    uint64_t seen_kmers_at_reconstruction, seen_superkmers_at_reconstruction;
    uint64_t seen_max_superkmers_at_reconstruction, actual_minimizer_number;
    uint64_t read_kmer, total_kmer_number, total_superkmer_number, nb_mmer_selected;
};

// FASTA text -> uncompressed sketch payload (A9 + A1-A8), i.e. what
// parse_fasta_test writes before gzip.  s is the value AFTER the CLI's stof
// (SubSampler.cpp:699).  Returns malloc'd payload.
char* orc_sketch_fasta(const char* text, uint64_t n, uint32_t k, uint32_t m, double s,
                       uint32_t abundance, uint64_t* out_len, orc_sketch_stats* stats) {
    Params P = make_params(k, m, s, abundance);
    std::vector<std::string> recs;
    split_fasta(text, n, recs);
    MinimizerMap mm;
    SketchStats ss; memset(&ss, 0, sizeof(ss));
    ScanStats st; memset(&st, 0, sizeof(st));
    for (size_t r = 0; r < recs.size(); ++r) {
        std::vector<Emit> em;
        scan_record(P, recs[r].data(), recs[r].size(), (uint32_t)r, em, st);
        for (auto& e : em)
            handle_superkmer(P, mm, ss, recs[r].substr(e.start, e.len), e.minimizer, e.rev != 0);
    }
    st.nb_mmer_selected -= m - 1;
    std::string payload = emit_sketch(P, mm, ss);
    if (stats) {
        stats->selected_kmer_number = ss.selected_kmer_number;
        stats->selected_superkmer_number = ss.selected_superkmer_number;
        stats->count_maximal_skmer = ss.count_maximal_skmer;
        stats->seen_kmers_at_reconstruction = ss.seen_kmers_at_reconstruction;
        stats->seen_superkmers_at_reconstruction = ss.seen_superkmers_at_reconstruction;
        stats->seen_max_superkmers_at_reconstruction = ss.seen_max_superkmers_at_reconstruction;
        stats->actual_minimizer_number = ss.actual_minimizer_number;
        stats->read_kmer = st.read_kmer; stats->total_kmer_number = st.total_kmer_number;
        stats->total_superkmer_number = st.total_superkmer_number;
        stats->nb_mmer_selected = st.nb_mmer_selected;
    }
    return dup_bytes(payload, out_len);
}

// N gunzipped sketch payloads -> inter (n*n, entries a<b used) and card (n).
// Follows Comparator::compare_sketches (A10-A15).  Returns 0, or -1 if n>65535.
int orc_compare(const char* const* payloads, const uint64_t* sizes, uint32_t n, uint32_t n_query,
                uint32_t* inter, uint64_t* card, uint32_t* k_out, uint32_t* m_out) {
    if (n > 65535) return -1;
    Comparator c;
    c.files.resize(n);
    for (uint32_t i = 0; i < n; ++i) { c.files[i].p = payloads[i]; c.files[i].n = sizes[i]; c.files[i].pos = 0; c.files[i].eof = false; }
    c.compare_sketches(n_query);
    memset(inter, 0, sizeof(uint32_t) * (uint64_t)n * n);
    for (auto& kv : c.score_A) inter[kv.first] = kv.second;
    for (uint32_t i = 0; i < n; ++i) card[i] = c.nb_kmer_seen_infile[i];
    if (k_out) *k_out = (uint32_t)c.k;
    if (m_out) *m_out = (uint32_t)c.m;
    return 0;
}
// The distinct (minimizer, canonical k-mer) keys ONE sketch contributes to the comparison: the comparator's own merge
// (compare_sketches :39-74) run over this file alone, with the per-bucket enumeration of skip_bucket / count_intersection
// (walk_bucket above: strDecompressor, inject_minimizer, the k-mer walks, canonize) collecting keys instead of counting
// them.  Sorted by (minimizer, k-mer).  What tests hold spsp_sketch_decode_device / spsp_sketch_parse_host against.
// Returns the key count; arrays are malloc'd (orc_free).
uint64_t orc_sketch_keys(const char* payload, uint64_t size, uint32_t* k_out, uint32_t* m_out, uint32_t** mn_out, uint64_t** lo_out,
                         uint64_t** hi_out) {
    Comparator c;
    c.files.resize(1);
    c.files[0].p = payload; c.files[0].n = size; c.files[0].pos = 0; c.files[0].eof = false;
    c.query_size = 1; c.nb_files = 1;
    c.nb_kmer_seen_infile.assign(1, 0);
    c.minimizers.assign(1, 0);
    std::vector<uint64_t> indices;
    c.get_header_info();
    c.increment_files(indices);
    std::vector<std::pair<uint32_t, u128>> keys;
    while (c.run) {
        c.findMin(indices);
        if (indices.empty()) break;
        const uint64_t mn = c.minimizers[indices[0]];
        const std::string minstr = num2str(mn, c.m);
        c.walk_bucket(0, minstr, [&](u128 canon) { keys.emplace_back((uint32_t)mn, canon); });
        c.increment_files(indices);
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const size_t n = keys.size();
    *mn_out = (uint32_t*)malloc(std::max<size_t>(1, n) * 4);
    *lo_out = (uint64_t*)malloc(std::max<size_t>(1, n) * 8);
    *hi_out = (uint64_t*)malloc(std::max<size_t>(1, n) * 8);
    for (size_t i = 0; i < n; ++i) { (*mn_out)[i] = keys[i].first; (*lo_out)[i] = (uint64_t)keys[i].second; (*hi_out)[i] = (uint64_t)(keys[i].second >> 64); }
    if (k_out) *k_out = (uint32_t)c.k;
    if (m_out) *m_out = (uint32_t)c.m;
    return n;
}

double orc_compare_timed(const char* const* payloads, const uint64_t* sizes, uint32_t n, uint32_t n_query,
                         uint32_t* inter, uint64_t* card) {
    auto t0 = std::chrono::steady_clock::now();
    orc_compare(payloads, sizes, n, n_query, inter, card, nullptr, nullptr);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// CSV text (A16).  names joined by '\n' in `names_nl`.
char* orc_csv(int jaccard, const char* names_nl, uint32_t n, uint32_t n_query, const uint32_t* inter,
              const uint64_t* card, int precision, double min_threshold, uint64_t* out_len) {
    std::vector<std::string> names;
    std::string cur;
    for (const char* p = names_nl; ; ++p) {
        if (*p == '\n' || *p == 0) { names.push_back(cur); cur.clear(); if (*p == 0) break; }
        else cur += *p;
    }
    names.resize(n);
    return dup_bytes(print_matrix(jaccard != 0, names, n, n_query, inter, card, precision, min_threshold), out_len);
}

// sortCSV (sort_csv.cpp:26-111 with split(), utils.cpp:609-629): reorder a Jaccard CSV (already gunzipped) to the
// order of the original file-of-files.  Returns NULL where the reference reads memory it never wrote, throws or
// blocks on cin.get(): a header name missing from the fof or listed twice, a row with fewer than N values or an
// unparsable one, fewer than N rows, a diagonal entry != 1.  The reference's stdout chatter is not reproduced.
char* orc_sort_csv(const char* csv, uint64_t csv_len, const char* fof, uint64_t fof_len, uint64_t* out_len) {
    auto lines_of = [](const char* p, uint64_t n) {          // getline until eof: a trailing newline yields a last ""
        std::vector<std::string> v;
        std::string cur;
        for (uint64_t i = 0; i < n; ++i) { if (p[i] == '\n') { v.push_back(cur); cur.clear(); } else cur += p[i]; }
        v.push_back(cur);
        return v;
    };
    auto split = [](const std::string& str) {                // utils.cpp:609-629: the last token stops at the first non-printable
        std::vector<std::string> res;
        size_t pred = 0;
        for (size_t i = 0; i < str.size(); ++i) if (str[i] == ',') { res.push_back(str.substr(pred, i - pred)); pred = i + 1; }
        std::string last;
        for (char c : str.substr(pred)) { if (isprint((unsigned char)c)) last += c; else break; }
        res.push_back(last);
        return res;
    };
    const std::vector<std::string> names_ordered = lines_of(fof, fof_len);          // :35-38
    const std::vector<std::string> in = lines_of(csv, csv_len);
    if (in.empty()) return nullptr;
    const std::vector<std::string> files_names = split(in[0]);                      // :39-40
    const size_t N = files_names.size();
    std::map<uint32_t, uint32_t> sorted_names, old2new;
    std::map<uint32_t, std::string> names;
    for (size_t i = 0; i < N; ++i) {                                                // :49-56
        const size_t pos = std::find(names_ordered.begin(), names_ordered.end(), files_names[i]) - names_ordered.begin();
        if (pos == names_ordered.size() || sorted_names.count((uint32_t)pos)) return nullptr;
        sorted_names[(uint32_t)pos] = (uint32_t)i;
        names[(uint32_t)pos] = files_names[i];
    }
    uint32_t id = 0;
    for (auto const& kv : sorted_names) old2new[kv.second] = id++;                  // :60-65
    std::ostringstream out;
    id = 0;
    for (auto const& kv : names) { out << kv.second; if (++id != N) out << ','; }   // :66-71
    out << std::endl;
    std::vector<double> matrix(N * N, 0.0);
    size_t line_id = 0;
    for (size_t l = 1; l < in.size(); ++l) {                                        // :76-85
        const std::string& line = in[l];
        if (line.size() < N) break;
        if (line_id >= N) return nullptr;
        const std::vector<std::string> values = split(line);
        if (values.size() < N) return nullptr;
        for (size_t i = 0; i < N; ++i) {
            char* endp = nullptr;
            const double v = strtod(values[i].c_str(), &endp);                      // stod: leading number, rest ignored
            if (endp == values[i].c_str()) return nullptr;
            matrix[old2new[(uint32_t)i] * N + old2new[(uint32_t)line_id]] = v;
        }
        ++line_id;
    }
    if (line_id != N) return nullptr;
    for (size_t i = 0; i < N; ++i) {                                                // :89-109
        if (matrix[i * N + i] != 1) return nullptr;                                 // "bug2" + cin.get()
        for (size_t j = 0; j < N; ++j) { out << matrix[i * N + j]; if (j != N - 1) out << ','; }
        out << std::endl;
    }
    return dup_bytes(out.str(), out_len);
}

}  // extern "C"
