// spsp_inflate.cpp -- one gzip member of known length, inflated in one go (host).
//
// spsp_compare_files reads collections of small sketch files (10 000 files of 3 KB at BASELINE configs[3]): zlib's inflate
// spends ~20 us on each (a byte-at-a-time state machine made to be resumable at any input or output byte; 4-5 ns per literal),
// which was a quarter of the whole call.  Here the member is in memory in full and the output length is known from the
// trailer, so the decoder is a plain loop: a 64-bit bit buffer refilled eight bytes at a time, one table lookup per symbol
// (10 bits for literals / lengths, 8 for distances; the rare longer codes walk the canonical code bit by bit), no state to save.
// RFC 1951 (stored, fixed and dynamic blocks) and RFC 1952 (header flags, CRC-32 and ISIZE checked: zlib's crc32).
//
// It is an ACCELERATOR, not a second reader: anything it does not like -- a header it cannot parse, an invalid code, output that
// does not come to exactly the promised length, a CRC that does not match, bytes behind the member -- makes it return -1 and the
// caller runs zlib's inflate over the same bytes, which then decides what the file is and words the error (zstr's behaviour:
// zstr.hpp:154-203).
#include <zlib.h>

#include <cstdint>
#include <cstring>

#include "spsp_internal.h"

namespace spsp {
namespace {

constexpr int kLitBits = 10, kDistBits = 8;

struct Huff {
    // primary table: entry = symbol << 4 | code length (0: no code of <= bits bits ends here -> the slow walk)
    uint16_t tab[1 << kLitBits];
    int bits;
    // canonical description for codes longer than `bits` (and for validation)
    uint16_t count[16], first_sym[16];
    uint32_t first_code[16];
    uint16_t sorted[288];
};

inline uint32_t rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// builds the decoder for code lengths len[0 .. n); false if the lengths do not describe a usable prefix code
bool build(Huff& H, const uint8_t* len, int n, int bits) {
    H.bits = bits;
    memset(H.count, 0, sizeof H.count);
    for (int i = 0; i < n; ++i) { if (len[i] > 15) return false; H.count[len[i]]++; }
    H.count[0] = 0;
    // over-subscribed codes are invalid; incomplete ones are allowed only as zlib allows them (a single code of length 1)
    int left = 1;
    int used = 0;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= H.count[l]; if (left < 0) return false; used += H.count[l]; }
    if (left > 0 && !(used == 1 && H.count[1] == 1)) return false;
    uint32_t code = 0;
    uint16_t sym = 0;
    for (int l = 1; l <= 15; ++l) {
        code <<= 1;
        H.first_code[l] = code; H.first_sym[l] = sym;
        code += H.count[l]; sym = (uint16_t)(sym + H.count[l]);
    }
    uint16_t next[16];
    for (int l = 1; l <= 15; ++l) next[l] = H.first_sym[l];
    for (int i = 0; i < n; ++i) if (len[i]) H.sorted[next[len[i]]++] = (uint16_t)i;
    memset(H.tab, 0, sizeof(uint16_t) << bits);
    // codes of up to `bits` bits: every table index whose low bits are the (bit-reversed) code
    uint32_t c[16];
    for (int l = 1; l <= 15; ++l) c[l] = H.first_code[l];
    for (int i = 0; i < n; ++i) {
        const int l = len[i];
        if (!l) continue;
        const uint32_t cd = c[l]++;
        if (l > bits) continue;
        const uint32_t r = rev_bits(cd, l);
        const uint16_t e = (uint16_t)((i << 4) | l);
        for (uint32_t x = r; x < (1u << bits); x += 1u << l) H.tab[x] = e;
    }
    return true;
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;      // one behind the member's last byte (the caller's buffer is readable for 8 bytes behind it)
    uint64_t buf = 0;
    int n = 0;               // valid bits in buf
    inline void refill() {
        if (n > 56) return;
        if (p + 8 <= end) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << n;
            const int take = (63 - n) >> 3;
            p += take; n += take * 8;
        } else {
            while (n <= 56) {
                if (p < end) buf |= (uint64_t)*p << n;
                ++p;                                   // (behind the end: zeros -- whoever CONSUMES them ends up behind the trailer's place and is refused there)
                n += 8;
                if (p > end + 8) break;
            }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
    inline void drop(int k) { buf >>= k; n -= k; }
    inline uint32_t take(int k) { const uint32_t v = peek(k); drop(k); return v; }
};

// one symbol; -1: invalid code.  The bit buffer holds at least 15 bits (the caller refills).
inline int decode(const Huff& H, Bits& B) {
    const uint16_t e = H.tab[B.peek(H.bits)];
    if (e & 15) { B.drop(e & 15); return e >> 4; }
    // longer than the table: walk the canonical code, one bit at a time from the top
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code << 1) | B.take(1);
        if (l > H.bits || true) {
            const uint32_t off = code - H.first_code[l];
            if (code >= H.first_code[l] && off < H.count[l]) return H.sorted[H.first_sym[l] + off];
        }
    }
    return -1;
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Fixed { Huff lit, dist; bool ok; Fixed() { uint8_t l[288]; for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; uint8_t d[30]; memset(d, 5, sizeof d); ok = build(lit, l, 288, kLitBits); Huff& D = dist; ok = ok && build_fixed_dist(D, d); }
               static bool build_fixed_dist(Huff& D, const uint8_t* d) {
                   // 30 codes of 5 bits: incomplete (32 would be complete) -- allowed for the fixed code (RFC 1951 3.2.6: codes 30-31 never occur)
                   uint8_t dd[32]; memcpy(dd, d, 30); dd[30] = dd[31] = 5;
                   return build(D, dd, 32, kDistBits);
               } };

}  // namespace

// in[0 .. n) = one gzip member (and nothing else); its payload must come to exactly out_len bytes.  Returns 0 when `out` holds the
// payload, checked against the member's CRC-32; -1 when the caller should run zlib instead.  `in` must be readable up to in + n + 8.
int fast_gunzip_member(const uint8_t* in, size_t n, uint8_t* out, size_t out_len) {
    if (n < 18 || in[0] != 0x1F || in[1] != 0x8B || in[2] != 8) return -1;
    const uint8_t flg = in[3];
    if (flg & 0xE0) return -1;
    size_t at = 10;
    if (flg & 4) { if (at + 2 > n) return -1; const size_t xlen = in[at] | ((size_t)in[at + 1] << 8); at += 2 + xlen; }
    if (flg & 8) { while (at < n && in[at]) ++at; ++at; }
    if (flg & 16) { while (at < n && in[at]) ++at; ++at; }
    if (flg & 2) at += 2;
    if (at + 8 > n) return -1;
    static const Fixed fixed;
    if (!fixed.ok) return -1;
    Bits B;
    B.p = in + at; B.end = in + n - 8;                       // the trailer is not deflate data
    uint8_t* o = out;
    uint8_t* const oe = out + out_len;
    Huff lit_h, dist_h;
    for (;;) {
        B.refill();
        const uint32_t last = B.take(1), type = B.take(2);
        if (type == 0) {
            B.drop(B.n & 7);                                 // to the byte boundary
            B.refill();
            const uint32_t len = B.take(16), nlen = B.take(16);
            if ((len ^ nlen) != 0xffffu) return -1;
            // the bytes still in the bit buffer belong to the stored data
            const uint8_t* src = B.p - (B.n >> 3);
            if (src + len > B.end || (size_t)(oe - o) < len) return -1;
            memcpy(o, src, len);
            o += len;
            B.p = src + len; B.buf = 0; B.n = 0;
        } else if (type == 1 || type == 2) {
            const Huff* L = &fixed.lit;
            const Huff* D = &fixed.dist;
            if (type == 2) {
                B.refill();
                const uint32_t hlit = B.take(5) + 257, hdist = B.take(5) + 1, hclen = B.take(4) + 4;
                if (hlit > 286 || hdist > 30) return -1;
                uint8_t cl[19];
                memset(cl, 0, sizeof cl);
                for (uint32_t i = 0; i < hclen; ++i) { B.refill(); cl[kClOrder[i]] = (uint8_t)B.take(3); }
                Huff clh;
                if (!build(clh, cl, 19, 7)) return -1;
                uint8_t lens[320];
                uint32_t i = 0;
                while (i < hlit + hdist) {
                    B.refill();
                    const int s = decode(clh, B);
                    if (s < 0) return -1;
                    if (s < 16) lens[i++] = (uint8_t)s;
                    else {
                        uint32_t rep; uint8_t v = 0;
                        if (s == 16) { if (i == 0) return -1; v = lens[i - 1]; rep = 3 + B.take(2); }
                        else if (s == 17) rep = 3 + B.take(3);
                        else rep = 11 + B.take(7);
                        if (i + rep > hlit + hdist) return -1;
                        memset(lens + i, v, rep);
                        i += rep;
                    }
                    if (B.p > B.end + 16) return -1;
                }
                if (lens[256] == 0) return -1;               // no end-of-block code
                if (!build(lit_h, lens, (int)hlit, kLitBits)) return -1;
                // a distance code of ONE code (all matches at one distance, or none) is incomplete and legal
                if (!build(dist_h, lens + hlit, (int)hdist, kDistBits)) {
                    int nz = 0;
                    for (uint32_t d = 0; d < hdist; ++d) nz += lens[hlit + d] != 0;
                    if (nz != 0) return -1;                  // (zero distance codes: a block of literals only -- any match is an error below)
                    memset(dist_h.tab, 0, sizeof(uint16_t) << kDistBits);
                    memset(dist_h.count, 0, sizeof dist_h.count);
                    dist_h.bits = kDistBits;
                }
                L = &lit_h; D = &dist_h;
            }
            for (;;) {
                B.refill();
                int s = decode(*L, B);
                if (s < 0) return -1;
                if (s < 256) {
                    if (o >= oe) return -1;
                    *o++ = (uint8_t)s;
                    // a second literal from the same refill (48 bits are left at the least: two codes of 15 fit)
                    s = decode(*L, B);
                    if (s < 0) return -1;
                    if (s < 256) { if (o >= oe) return -1; *o++ = (uint8_t)s; continue; }
                }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return -1;
                B.refill();
                const uint32_t len = kLenBase[s] + B.take(kLenExtra[s]);
                const int ds = decode(*D, B);
                if (ds < 0 || ds >= 30) return -1;
                B.refill();
                const uint32_t dist = kDistBase[ds] + B.take(kDistExtra[ds]);
                if (dist > (size_t)(o - out) || (size_t)(oe - o) < len) return -1;
                const uint8_t* from = o - dist;
                if (dist >= len) { memcpy(o, from, len); o += len; }
                else for (uint32_t x = 0; x < len; ++x) *o++ = *from++;      // overlapping: a run
            }
        } else return -1;
        if (B.p > B.end + 16) return -1;                     // (far behind the data: a member that never ends)
        if (last) break;
    }
    if (o != oe) return -1;
    // the trailer sits at the next byte boundary, and the member ends with it
    const uint8_t* tail = B.p - (B.n >> 3);
    if (tail != in + n - 8) return -1;
    uint32_t crc, isize;
    memcpy(&crc, tail, 4); memcpy(&isize, tail + 4, 4);
    if (isize != (uint32_t)out_len) return -1;
    uint32_t mine = (uint32_t)crc32(0L, Z_NULL, 0);
    size_t done = 0;
    while (done < out_len) { const size_t step = out_len - done < (1u << 30) ? out_len - done : (1u << 30); mine = (uint32_t)crc32(mine, out + done, (uInt)step); done += step; }
    return mine == crc ? 0 : -1;
}

}  // namespace spsp
