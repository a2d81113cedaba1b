#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
// tests/tools/host_asan/fuzz_inflate.cpp -- the one-go gunzip (csrc/spsp_inflate.cpp) against zlib under ASan + UBSan: streams of every
// block kind and strategy, then bit flips, truncations, overwritten and appended bytes.  It must accept exactly what zlib accepts as ONE
// member of the promised length (same bytes), refuse the rest, and never touch memory outside its buffers (exact-size heap blocks).
namespace spsp { int fast_gunzip_member(const uint8_t* in, size_t n, uint8_t* out, size_t out_len); }
static std::vector<uint8_t> gz(const std::vector<uint8_t>& d, int level, int strategy) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, 31, 8, strategy);
    std::vector<uint8_t> out(2 * d.size() + 4096);
    zs.next_in = (Bytef*)d.data(); zs.avail_in = d.size(); zs.next_out = out.data(); zs.avail_out = out.size();
    deflate(&zs, Z_FINISH); out.resize(out.size() - zs.avail_out); deflateEnd(&zs); return out;
}
static int zl(const uint8_t* in, size_t n, std::vector<uint8_t>& out, size_t want) {
    z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, 31);
    out.assign(want + 1, 0);
    zs.next_in = (Bytef*)in; zs.avail_in = n; zs.next_out = out.data(); zs.avail_out = out.size();
    int r = inflate(&zs, Z_FINISH); size_t got = zs.total_out; size_t left = zs.avail_in; inflateEnd(&zs);
    if (r != Z_STREAM_END || got != want || left != 0) return -1;
    out.resize(got); return 0;
}
int main() {
    std::mt19937_64 rng(7);
    long agree_ok = 0, agree_bad = 0, fast_refused = 0;
    for (int it = 0; it < 4000; ++it) {
        size_t n = rng() % 6000;
        std::vector<uint8_t> d(n);
        int alpha = 2 + rng() % 250;
        for (auto& b : d) b = rng() % alpha;
        if (it % 3 == 0) for (size_t i = 100; i < n; ++i) d[i] = d[i - 1 - rng() % 100];
        static const int strat[5] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
        std::vector<uint8_t> z = gz(d, (int)(rng() % 10), strat[rng() % 5]);
        for (int mut = 0; mut < 12; ++mut) {
            std::vector<uint8_t> m = z;
            int kind = mut == 0 ? 0 : 1 + rng() % 4;
            if (kind == 1 && !m.empty()) m[rng() % m.size()] ^= 1u << (rng() % 8);
            if (kind == 2 && m.size() > 1) m.resize(1 + rng() % (m.size() - 1));
            if (kind == 3) for (int x = 0; x < 3 && !m.empty(); ++x) m[rng() % m.size()] = rng();
            if (kind == 4) m.push_back(rng());
            // exact-size heap buffers (+8 readable behind the input, as the contract says) so that ASan sees any stray access
            uint8_t* in = (uint8_t*)malloc(m.size() + 8); memcpy(in, m.data(), m.size()); memset(in + m.size(), 0xAB, 8);
            uint32_t isize = 0; if (m.size() >= 4) memcpy(&isize, m.data() + m.size() - 4, 4);
            size_t want = kind == 0 ? d.size() : (isize < (1u << 20) ? isize : d.size());
            uint8_t* out = (uint8_t*)malloc(want ? want : 1);
            int f = spsp::fast_gunzip_member(in, m.size(), out, want);
            std::vector<uint8_t> ref;
            int z0 = zl(in, m.size(), ref, want);
            if (f == 0) {
                if (z0 != 0 || memcmp(out, ref.data(), want) != 0) { printf("MISMATCH it=%d mut=%d kind=%d fast ok, zlib %d\n", it, mut, kind, z0); return 1; }
                ++agree_ok;
            } else { if (z0 == 0) ++fast_refused; else ++agree_bad; }
            if (f != 0 && z0 == 0 && kind != 4) { printf("REFUSED what zlib takes it=%d mut=%d kind=%d\n", it, mut, kind); return 1; }
            if (kind == 0 && f != 0 && z0 == 0) { printf("REFUSED a valid stream it=%d n=%zu\n", it, n); return 1; }
            free(in); free(out);
        }
    }
    printf("ok: %ld accepted (equal to zlib), %ld refused by both, %ld refused by the fast path only\n", agree_ok, agree_bad, fast_refused);
}
