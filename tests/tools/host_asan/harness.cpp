// AddressSanitizer / UBSan harness for the HOST side of libspsp (spsp_host.cpp is compiled in directly, the GPU entry
// points it calls are stubbed out): feeds valid sketch payloads, CSVs and FASTA text plus thousands of randomly
// corrupted variants to the parsers.  Every call must return (OK or an error code) -- never crash, never read out of
// bounds.  Sanitizers cannot run on the GPU side of this pool, so this is where they run.
//   build + run: tests/tools/host_asan/run.sh
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

#include "../../../supersampler_amd/csrc/spsp_internal.h"

namespace spsp {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
int gather_superkmers_impl(spsp_ctx*, const uint8_t*, const uint64_t*, const spsp_superkmer*, uint64_t, uint8_t**, uint32_t**) { return SPSP_ERR_NO_DEVICE; }
int clean_device_impl(spsp_ctx*, const uint8_t*, uint64_t, uint8_t**, uint64_t*, uint64_t**, uint32_t*) { return SPSP_ERR_NO_DEVICE; }
int scan_device_impl(spsp_ctx*, const spsp_params*, const uint8_t*, uint64_t, const uint64_t*, uint32_t, spsp_superkmer**, uint64_t*) { return SPSP_ERR_NO_DEVICE; }
int compare_payloads_impl(spsp_ctx*, const uint8_t* const*, const uint64_t*, uint32_t, const int*, const uint32_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint64_t*, bool*, std::vector<uint64_t>*) { return SPSP_ERR_NO_DEVICE; }
int compare_payloads_multi(spsp_ctx* const*, uint32_t, const uint8_t* const*, const uint64_t*, uint32_t, const int*, const uint32_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint64_t*, bool*, std::vector<uint64_t>*) { return SPSP_ERR_NO_DEVICE; }
int check_params(const spsp_params* p) { return (p && p->m >= 1 && p->m <= 15 && p->k >= p->m && p->k <= 63) ? SPSP_OK : SPSP_ERR_ARG; }
}  // namespace spsp
extern "C" {
const char* spsp_last_error(void) { return spsp::g_err.c_str(); }
void spsp_free(void* p) { free(p); }
int spsp_scan(spsp_ctx*, const spsp_params*, const uint8_t*, const uint64_t*, uint32_t, spsp_superkmer**, uint64_t*) { return SPSP_ERR_NO_DEVICE; }
int spsp_compare(spsp_ctx*, const spsp_sketch_view*, uint32_t, uint32_t, uint32_t*, uint64_t*) { return SPSP_ERR_NO_DEVICE; }
int spsp_sketch_text(spsp_ctx*, const spsp_params*, double, const char*, uint64_t, uint8_t**, uint64_t*, spsp_sketch_stats*) { return SPSP_ERR_NO_DEVICE; }
int spsp_copy_to_host(spsp_ctx*, void*, const void*, uint64_t) { return SPSP_ERR_NO_DEVICE; }
int spsp_create(int, void*, spsp_ctx**) { return SPSP_ERR_NO_DEVICE; }
void spsp_destroy(spsp_ctx*) {}
}

static std::string slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

// ThreadSanitizer mode: the host functions that spread work over threads (CSV rows, gzip members, file reads)
// the sketch builder on a stream large enough for its threads (buckets built side by side): the oracle's bytes
static int large_build(const std::string& dir) {
    const std::string bb = slurp(dir + "/build_bases.bin"), bo = slurp(dir + "/build_off.bin"), bs = slurp(dir + "/build_sk.bin");
    if (bs.empty()) return 0;
    std::vector<uint64_t> off(bo.size() / 8);
    memcpy(off.data(), bo.data(), off.size() * 8);
    std::vector<spsp_superkmer> sk(bs.size() / sizeof(spsp_superkmer));
    memcpy(sk.data(), bs.data(), sk.size() * sizeof(spsp_superkmer));
    for (uint32_t ab = 1; ab <= 2; ++ab) {
        const std::string want = slurp(dir + "/build_payload_a" + std::to_string(ab) + ".bin");
        spsp_params P{21, 9, spsp_threshold_host(21, 9, 4.0), ab, 0};
        uint8_t* payload = nullptr; uint64_t plen = 0; spsp_sketch_stats st;
        if (spsp_sketch_build_host(&P, 4.0, (const uint8_t*)bb.data(), off.data(), (uint32_t)off.size() - 1, sk.data(), sk.size(), &payload, &plen, &st) != SPSP_OK) {
            fprintf(stderr, "large sketch build failed: %s\n", spsp_last_error()); return 9;
        }
        const bool same = plen == want.size() && memcmp(payload, want.data(), plen) == 0;
        free(payload);
        if (!same) { fprintf(stderr, "large sketch build: payload differs from the oracle's (abundance %u)\n", ab); return 9; }
    }
    printf("large sketch build (%zu super-k-mers, threads): the oracle's bytes\n", sk.size());
    return 0;
}

static int threads_mode(const std::string& dir) {
    const uint32_t n = 900;
    std::mt19937_64 rng(7);
    std::vector<uint32_t> inter((size_t)n * n, 0);
    std::vector<uint64_t> card(n);
    std::vector<std::string> names(n);
    std::vector<const char*> np(n);
    for (uint32_t i = 0; i < n; ++i) { card[i] = 1000 + rng() % 5000; names[i] = "s" + std::to_string(i); np[i] = names[i].c_str(); }
    for (uint32_t i = 0; i < n; ++i) for (uint32_t j = i + 1; j < n; ++j) if (rng() % 3 == 0) inter[(size_t)i * n + j] = (uint32_t)(rng() % 1000);
    for (int jac = 0; jac < 2; ++jac) {
        char* text = nullptr; uint64_t len = 0;
        if (spsp_csv_host(jac, np.data(), n, n, inter.data(), card.data(), 6, 0.0, &text, &len) != SPSP_OK) return 3;
        const std::string path = dir + "/m" + std::to_string(jac) + ".csv.gz";
        if (spsp_write_gz_host(path.c_str(), (const uint8_t*)text, len, 1) != SPSP_OK) return 4;
        uint8_t* back = nullptr; uint64_t blen = 0;
        if (spsp_read_file_host(path.c_str(), &back, &blen) != SPSP_OK || blen != len || memcmp(back, text, len) != 0) return 5;
        free(back); free(text);
    }
    std::string big(40u << 20, 'A');                      // three 16 MiB gzip members compressed on threads
    for (auto& c : big) c = "ACGT"[rng() & 3];
    const std::string path = dir + "/big.gz";
    if (spsp_write_gz_host(path.c_str(), (const uint8_t*)big.data(), big.size(), 1) != SPSP_OK) return 6;
    uint8_t* back = nullptr; uint64_t blen = 0;
    if (spsp_read_file_host(path.c_str(), &back, &blen) != SPSP_OK || blen != big.size() || memcmp(back, big.data(), blen) != 0) return 7;
    free(back);
    if (int r = large_build(dir)) return r;
    printf("host thread-sanitizer harness: threaded CSV, parallel gzip members, read-back, threaded sketch build OK\n");
    return 0;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    if (argc > 2 && !strcmp(argv[2], "threads")) return threads_mode(dir);
    const int rounds = argc > 2 ? atoi(argv[2]) : 3000;
    std::mt19937_64 rng(12345);
    std::vector<std::string> payloads, csvs, fastas;
    for (int i = 0;; ++i) { std::string s = slurp(dir + "/payload" + std::to_string(i) + ".bin"); if (s.empty()) break; payloads.push_back(s); }
    for (int i = 0;; ++i) { std::string s = slurp(dir + "/csv" + std::to_string(i) + ".txt"); if (s.empty()) break; csvs.push_back(s); }
    for (int i = 0;; ++i) { std::string s = slurp(dir + "/fasta" + std::to_string(i) + ".fa"); if (s.empty()) break; fastas.push_back(s); }
    const std::string fof = slurp(dir + "/fof.txt");
    if (payloads.empty() || csvs.empty() || fastas.empty()) { fprintf(stderr, "no inputs in %s\n", dir.c_str()); return 2; }
    auto mutate = [&](std::string s) {
        const int kind = (int)(rng() % 5);
        if (s.empty()) return s;
        if (kind == 0) s.resize(rng() % (s.size() + 1));                                   // truncate
        else if (kind == 1) for (int j = 0, n = 1 + (int)(rng() % 8); j < n; ++j) s[rng() % s.size()] = (char)(rng() & 0xff);   // flip bytes
        else if (kind == 2) s.insert(rng() % (s.size() + 1), std::string(1 + rng() % 40, (char)(rng() & 0xff)));
        else if (kind == 3) { const size_t a = rng() % s.size(); s.erase(a, 1 + rng() % 30); }
        else { const size_t a = rng() % s.size(); uint32_t big = 0xfffffff0u - (uint32_t)(rng() % 64); if (a + 4 <= s.size()) memcpy(&s[a], &big, 4); }   // absurd length word
        return s;
    };
    uint64_t ok = 0, rejected = 0;
    for (int it = 0; it < rounds; ++it) {
        {   // sketch reader + the merge's first-read rule
            const std::string p = it < (int)payloads.size() ? payloads[it] : mutate(payloads[rng() % payloads.size()]);
            uint32_t k = 0, m = 0, *mn = nullptr; uint64_t *lo = nullptr, *hi = nullptr, n = 0;
            const int rc = spsp_sketch_parse_host((const uint8_t*)p.data(), p.size(), &k, &m, &mn, &lo, &hi, &n);
            if (rc == SPSP_OK) {
                ++ok;
                uint64_t acc = 0;
                for (uint64_t i = 0; i < n; ++i) acc += mn[i] + lo[i] + hi[i];
                if (acc == 0x1234567) printf("!");
                char buf[16]; memset(buf, 'A', sizeof buf);
                int has = 0; uint32_t pm; uint64_t pl, ph;
                spsp_sketch_chain_host((const uint8_t*)p.data(), p.size(), k, m, buf, &has, &pm, &pl, &ph);
                free(mn); free(lo); free(hi);
            } else ++rejected;
        }
        {   // structure walk of the GPU decoder: every byte range a descriptor makes k_decode_emit read must lie inside the payload
            const std::string p = it < (int)payloads.size() ? payloads[it] : mutate(payloads[rng() % payloads.size()]);
            std::vector<uint8_t> exact(p.begin(), p.end());   // heap copy without slack: ASan sees one byte too far
            spsp::ParsedSketch P;
            if (spsp::sketch_parse_structure_host(exact.data(), exact.size(), &P) == SPSP_OK) {
                ++ok;
                uint64_t acc = 0, out = 0;
                for (const spsp::DecDesc& D : P.desc) {
                    const uint32_t kind = D.info & 3u, l1 = (D.info >> 2) & 0xffu, l2 = (D.info >> 10) & 0xffu;
                    const uint64_t nbytes = kind == 0 ? (2ull * (P.k - P.m) + 3) / 4 : kind == 1 ? (uint64_t)l1 + 1 + l2 : 0;
                    if (D.off + nbytes > exact.size() || D.out != out) { fprintf(stderr, "descriptor outside its payload\n"); return 9; }
                    for (uint64_t b = 0; b < nbytes; ++b) acc += exact[D.off + b];
                    const uint32_t total = kind == 0 ? 2 * P.k - P.m : kind == 1 ? l1 + P.m + l2 : P.m;
                    out += total - P.k + 1;
                }
                if (out != P.n_keys) { fprintf(stderr, "key count mismatch\n"); return 9; }
                if (acc == 0x1234567) printf("!");
            } else ++rejected;
        }
        {   // sortCSV
            const std::string c = it < (int)csvs.size() ? csvs[it] : mutate(csvs[rng() % csvs.size()]);
            const std::string f = (rng() % 4) ? fof : mutate(fof);
            char* text = nullptr; uint64_t len = 0;
            if (spsp_sort_csv_host(c.data(), c.size(), f.data(), f.size(), &text, &len) == SPSP_OK) { ++ok; free(text); } else ++rejected;
        }
        {   // FASTA cleaner + sketch builder on an empty stream
            const std::string t = it < (int)fastas.size() ? fastas[it] : mutate(fastas[rng() % fastas.size()]);
            uint8_t* bases = nullptr; uint64_t* off = nullptr; uint32_t n_rec = 0;
            if (spsp_fasta_clean_host(t.data(), t.size(), &bases, &off, &n_rec) == SPSP_OK) {
                ++ok;
                spsp_params P{31, 11, spsp_threshold_host(31, 11, 1000.0), 1, 0};
                uint8_t* payload = nullptr; uint64_t plen = 0; spsp_sketch_stats st;
                if (spsp_sketch_build_host(&P, 1000.0, bases, off, n_rec, nullptr, 0, &payload, &plen, &st) == SPSP_OK) free(payload);
                free(bases); free(off);
            } else ++rejected;
        }
    }
    if (int r = large_build(dir)) return r;
    printf("host sanitizer harness: %llu calls accepted, %llu rejected, no crash\n", (unsigned long long)ok, (unsigned long long)rejected);
    return 0;
}
