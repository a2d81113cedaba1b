// spsp_ingest.hip -- "next" row N1 of SURVEY.md 8(f): FASTA ingest on the GPU.
//
// The reference reads a record with getLineFasta (utils.cpp:706-718): drop one
// line, concatenate the following lines up to the next one that starts with
// '>', then clean_dna (utils.cpp:675-702) deletes every byte that is not one of
// ACGTacgt and upper-cases the rest.  Over a whole (gunzipped) file that is:
//
//   * line 0 and every line starting with '>' (or 0xFF, the reference's
//     (char)peek()==EOF aliasing) is a DROP line and starts a new record;
//   * every other byte survives iff it is in ACGTacgt.
//
// So ingest is a stream compaction with a one-bit line state.  Three launches:
//   k_clean_tiles   per 4 KiB tile: state transform of the tile (does it end inside
//                   a drop line?), survivor counts with the entry state left open,
//                   record starts
//   k_clean_scan    one workgroup: composes the tile transforms, turns the counts
//                   into output / record offsets
//   k_clean_write   per tile: compacts through LDS, coalesced stores, rec_off[]
// 1 B read + 1 B written per input byte, twice over the input (L2-friendly).
#include <cstdlib>
#include <vector>

#include <cstring>
#include <string>

#include "spsp_internal.h"
#include "spsp_device.h"

namespace spsp {

constexpr int kCleanThreads = 256;
constexpr int kCleanChunk = 16;
constexpr int kCleanTile = kCleanThreads * kCleanChunk;   // 4096 bytes

__device__ __forceinline__ bool is_drop_start(uint32_t c) { return c == '>' || c == 0xFFu; }
// upper-cased base for ACGTacgt, 0 otherwise
__device__ __forceinline__ uint32_t keep_base(uint32_t c) {
    const uint32_t u = c & 0xDFu;   // fold case
    return (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : 0u;
}

// Everything one lane needs to know about its 16 bytes.
struct ChunkInfo {
    uint32_t bytes[4];       // raw little-endian dwords (zero past the end of the text)
    uint32_t valid;          // number of bytes inside the text
    uint32_t keep_if_data;   // bit j: byte j is a base (survives if its line is a data line)
    uint32_t nl_mask;        // bit j: byte j is '\n'
    uint32_t drop_after;     // bit j: byte j is '\n' and the line starting at j+1 exists and is a drop line
};

// Byte-parallel classification (four bytes per register, no per-byte loop):
//   zero_bytes(v)  0x80 in every byte of v that is zero (exact, no borrow between bytes)
//   pack_flags(z)  the four 0x80 flags of a word -> 4 bits (byte 0 -> bit 0)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}
__device__ __forceinline__ uint32_t pack_flags(uint32_t z) { return (((z >> 7) * 0x01020408u) >> 24) & 0xFu; }   // bits 0,8,16,24 -> 24..27, no carries
// 0x80 where the byte is one of ACGTacgt: fold the case, look the base of its 2-bit code (c >> 1) & 3 up with one
// byte permute (A, C, T, G) and compare
__device__ __forceinline__ uint32_t base_bytes(uint32_t w) {
    const uint32_t code = (w >> 1) & 0x03030303u;
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, code);   // selector bytes 0..3 pick 'A','C','T','G'
    return zero_bytes((w & 0xDFDFDFDFu) ^ expect);
}

__device__ __forceinline__ ChunkInfo load_chunk(const uint8_t* __restrict__ text, uint64_t n, uint64_t p0) {
    ChunkInfo c;
    c.valid = p0 >= n ? 0u : (uint32_t)((n - p0) < (uint64_t)kCleanChunk ? (n - p0) : (uint64_t)kCleanChunk);
    if (c.valid == kCleanChunk) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + p0);
        c.bytes[0] = v.x; c.bytes[1] = v.y; c.bytes[2] = v.z; c.bytes[3] = v.w;
    } else {
        c.bytes[0] = c.bytes[1] = c.bytes[2] = c.bytes[3] = 0;
        for (uint32_t j = 0; j < c.valid; ++j) c.bytes[j >> 2] |= (uint32_t)text[p0 + j] << (8 * (j & 3));
    }
    const uint32_t in_mask = c.valid >= 16 ? 0xFFFFu : ((1u << c.valid) - 1u);
    uint32_t keep = 0, nl = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        keep |= pack_flags(base_bytes(c.bytes[d])) << (4 * d);
        nl |= pack_flags(zero_bytes(c.bytes[d] ^ 0x0A0A0A0Au)) << (4 * d);
    }
    c.keep_if_data = keep & in_mask;
    c.nl_mask = nl & in_mask;
    // what follows each newline decides the next line's kind: newlines are rare (one per line), so a loop over them
    c.drop_after = 0;
    if (c.nl_mask) {
        const uint32_t next = (p0 + kCleanChunk < n) ? text[p0 + kCleanChunk] : 0x100u;   // 0x100: nothing follows
        const uint64_t lo = ((uint64_t)c.bytes[1] << 32) | c.bytes[0], hi = ((uint64_t)c.bytes[3] << 32) | c.bytes[2];
        uint32_t rem = c.nl_mask;
        while (rem) {
            const uint32_t j = __ffs(rem) - 1;
            rem &= rem - 1;
            uint32_t follow;
            if (j + 1 < (uint32_t)kCleanChunk) {
                const uint32_t f = j + 1;
                follow = f < c.valid ? (uint32_t)((f < 8 ? lo >> (8 * f) : hi >> (8 * (f - 8))) & 0xFFu) : 0x100u;
            } else follow = next;
            if (follow != 0x100u && is_drop_start(follow)) c.drop_after |= 1u << j;
        }
    }
    return c;
}

// Line-state transform of a span: 0 = identity (no newline inside), 2 = ends in a data
// line, 3 = ends in a drop line.  compose(a, b): a first, then b.
__device__ __forceinline__ uint32_t compose(uint32_t a, uint32_t b) { return b ? b : a; }
__device__ __forceinline__ uint32_t chunk_transform(const ChunkInfo& c) {
    if (!c.nl_mask) return 0u;
    const int last = 31 - __clz(c.nl_mask);
    return 2u | ((c.drop_after >> last) & 1u);
}
// bit j set iff byte j survives, given the state (1 = drop line) of the line entering the chunk: the bytes between
// two newlines share their line's state, so the walk is over the (rare) newlines, not over the bytes
__device__ __forceinline__ uint32_t chunk_keep(const ChunkInfo& c, uint32_t entry_drop) {
    uint32_t keep = 0, drop = entry_drop, from = 0, rem = c.nl_mask;
    while (rem) {
        const uint32_t j = __ffs(rem) - 1;
        rem &= rem - 1;
        if (!drop) keep |= c.keep_if_data & ((2u << j) - 1u) & ~((1u << from) - 1u);    // bytes from..j
        drop = (c.drop_after >> j) & 1u;
        from = j + 1;
    }
    if (!drop) keep |= c.keep_if_data & ~((1u << from) - 1u) & 0xFFFFu;
    return keep;
}

// exclusive scan of the line-state transforms over the workgroup (lane order = byte order)
__device__ __forceinline__ uint32_t block_scan_transform(uint32_t t, uint32_t* s_wave, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t x = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x = compose(y, x);
    }
    if (lane == 63) s_wave[wid] = x;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    for (uint32_t w = 0; w < kCleanThreads / 64; ++w) { if (w < wid) pre = compose(pre, s_wave[w]); all = compose(all, s_wave[w]); }
    uint32_t excl = __shfl_up(x, 1);
    if (lane == 0) excl = 0;
    *total = all;
    __syncthreads();
    return compose(pre, excl);
}
__device__ __forceinline__ uint32_t block_scan_add(uint32_t v, uint32_t* s_wave, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) s_wave[wid] = x;
    __syncthreads();
    uint32_t pre = 0, all = 0;
    for (uint32_t w = 0; w < kCleanThreads / 64; ++w) { if (w < wid) pre += s_wave[w]; all += s_wave[w]; }
    *total = all;
    __syncthreads();
    return pre + x - v;
}

struct TileSummary { uint32_t transform, keep_fixed, keep_if_data, headers; };

__global__ __launch_bounds__(kCleanThreads) void k_clean_tiles(const uint8_t* __restrict__ text, uint64_t n,
                                                              TileSummary* __restrict__ tiles) {
    __shared__ uint32_t s_wave[kCleanThreads / 64];
    const uint64_t p0 = (uint64_t)blockIdx.x * kCleanTile + (uint64_t)threadIdx.x * kCleanChunk;
    const ChunkInfo c = load_chunk(text, n, p0);
    uint32_t tile_t;
    const uint32_t pre = block_scan_transform(chunk_transform(c), s_wave, &tile_t);
    // lanes before the tile's first newline inherit the (still unknown) entry state of the tile
    const uint32_t fixed = pre ? __popc(chunk_keep(c, pre & 1u)) : __popc(chunk_keep(c, 1u));
    const uint32_t extra = pre ? 0u : __popc(chunk_keep(c, 0u)) - fixed;
    // only the three totals are needed: one packed 64-bit reduction (each count fits in 20 bits)
    unsigned long long packed = (unsigned long long)fixed | ((unsigned long long)extra << 20) |
                                ((unsigned long long)__popc(c.drop_after) << 40);
#pragma unroll
    for (int d = 32; d; d >>= 1) packed += __shfl_xor(packed, d);
    __shared__ unsigned long long s_tot[kCleanThreads / 64];
    if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = packed;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (int w = 0; w < kCleanThreads / 64; ++w) all += s_tot[w];
        tiles[blockIdx.x] = TileSummary{tile_t, (uint32_t)(all & 0xFFFFFu), (uint32_t)((all >> 20) & 0xFFFFFu), (uint32_t)(all >> 40)};
    }
}

// one workgroup: entry state, output offset and record offset of every tile; totals.  A lane owns kScanTiles
// consecutive tiles per round (composed locally, then one workgroup scan over the lanes), so a round covers
// 1024 * kScanTiles tiles and a 200 MB text needs a dozen rounds.
constexpr int kScanTiles = 4;
__global__ __launch_bounds__(1024) void k_clean_scan(const TileSummary* __restrict__ tiles, uint64_t n_tiles,
                                                    uint32_t* __restrict__ entry_drop, uint64_t* __restrict__ out_off,
                                                    uint32_t* __restrict__ rec_base, uint64_t* __restrict__ totals) {
    __shared__ uint32_t s_t[16];
    __shared__ unsigned long long s_k[16];
    __shared__ uint32_t s_h[16];
    __shared__ uint32_t c_t;
    __shared__ unsigned long long c_k;
    __shared__ uint32_t c_h;
    const uint32_t t = threadIdx.x, lane = t & 63, wid = t >> 6;
    if (t == 0) { c_t = 3u; c_k = 0; c_h = 1; }   // the file starts inside a drop line (line 0), which is record 0
    __syncthreads();
    for (uint64_t base = 0; base < n_tiles; base += 1024ull * kScanTiles) {
        const uint64_t i0 = base + (uint64_t)t * kScanTiles;
        TileSummary me[kScanTiles];
        uint32_t tl = 0;                                    // transform of this lane's tiles
#pragma unroll
        for (int u = 0; u < kScanTiles; ++u) {
            me[u] = (i0 + u < n_tiles) ? tiles[i0 + u] : TileSummary{0, 0, 0, 0};
            tl = compose(tl, me[u].transform);
        }
        // inclusive scan of the lanes' transforms
        uint32_t x = tl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x = compose(y, x);
        }
        if (lane == 63) s_t[wid] = x;
        __syncthreads();
        uint32_t pre = c_t;
        for (uint32_t w = 0; w < wid; ++w) pre = compose(pre, s_t[w]);
        uint32_t excl = __shfl_up(x, 1);
        if (lane == 0) excl = 0;
        uint32_t state = compose(pre, excl);                // always a constant: the carry starts as one
        // entry state of each of the lane's tiles, their survivor counts, the lane's sums
        uint32_t entry[kScanTiles];
        unsigned long long keep[kScanTiles], ksum = 0;
        uint32_t hsum = 0;
#pragma unroll
        for (int u = 0; u < kScanTiles; ++u) {
            entry[u] = state & 1u;
            keep[u] = (unsigned long long)me[u].keep_fixed + (entry[u] ? 0u : me[u].keep_if_data);
            ksum += keep[u]; hsum += me[u].headers;
            state = compose(state, me[u].transform);
        }
        unsigned long long kx = ksum;
        uint32_t hx = hsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long ky = __shfl_up(kx, d);
            const uint32_t hy = __shfl_up(hx, d);
            if (lane >= (uint32_t)d) { kx += ky; hx += hy; }
        }
        if (lane == 63) { s_k[wid] = kx; s_h[wid] = hx; }
        __syncthreads();
        unsigned long long kpre = c_k, kall = 0;
        uint32_t hpre = c_h, hall = 0, tall = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wid) { kpre += s_k[w]; hpre += s_h[w]; }
            kall += s_k[w]; hall += s_h[w]; tall = compose(tall, s_t[w]);
        }
        unsigned long long ko = kpre + kx - ksum;
        uint32_t ho = hpre + hx - hsum;
#pragma unroll
        for (int u = 0; u < kScanTiles; ++u) {
            if (i0 + u < n_tiles) { entry_drop[i0 + u] = entry[u]; out_off[i0 + u] = ko; rec_base[i0 + u] = ho; }
            ko += keep[u]; ho += me[u].headers;
        }
        __syncthreads();
        if (t == 0) { c_t = compose(c_t, tall); c_k += kall; c_h += hall; }
        __syncthreads();
    }
    if (t == 0) { totals[0] = c_k; totals[1] = c_h; }
}

// PACK: the survivors leave as 2-bit words -- 16 bases per dword, first base in bits 31:30, the layout the dense pass
// reads with SPSP_SCAN_PACKED_INPUT (N1 of SURVEY.md 8f: "2-bit packing + non-ACGT compaction").  The tile's survivors
// sit in LDS at the output's alignment modulo 16, so an aligned 16-byte group of the staging buffer IS one output word:
// whole groups are packed and stored, the first and last group of a tile (shared with the neighbouring tiles) are ORed
// into the zero-initialised buffer.
template <bool PACK>
__global__ __launch_bounds__(kCleanThreads) void k_clean_write(const uint8_t* __restrict__ text, uint64_t n,
                                                              const uint32_t* __restrict__ entry_drop,
                                                              const uint64_t* __restrict__ out_off,
                                                              const uint32_t* __restrict__ rec_base,
                                                              uint8_t* __restrict__ bases, uint64_t* __restrict__ rec_off) {
    __shared__ uint32_t s_wave[kCleanThreads / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[kCleanTile + 16];
    const uint64_t p0 = (uint64_t)blockIdx.x * kCleanTile + (uint64_t)threadIdx.x * kCleanChunk;
    const ChunkInfo c = load_chunk(text, n, p0);
    uint32_t tile_t;
    const uint32_t pre = block_scan_transform(chunk_transform(c), s_wave, &tile_t);
    const uint32_t entry = pre ? (pre & 1u) : entry_drop[blockIdx.x];
    const uint32_t keep = chunk_keep(c, entry);
    // survivors and record starts before this lane: one scan over the packed pair of counts
    uint32_t tile_both;
    const uint32_t both = block_scan_add(__popc(keep) | (__popc(c.drop_after) << 16), s_wave, &tile_both);
    const uint32_t koff = both & 0xFFFFu, hoff = both >> 16, tile_keep = tile_both & 0xFFFFu;
    const uint64_t obase = out_off[blockIdx.x];
    // survivors -> LDS in order, shifted so that LDS offset and global address agree modulo 16: the tile then
    // leaves as aligned 16-byte stores; record starts -> rec_off
    const uint32_t sh = (uint32_t)(obase & 15u);
    uint32_t at = koff, hr = rec_base[blockIdx.x] + hoff;
    uint8_t* dst = s_out + sh + at;
    const uint64_t lo = (((uint64_t)c.bytes[1] << 32) | c.bytes[0]) & 0xDFDFDFDFDFDFDFDFull;   // survivors are ACGTacgt:
    const uint64_t hi = (((uint64_t)c.bytes[3] << 32) | c.bytes[2]) & 0xDFDFDFDFDFDFDFDFull;   // clearing bit 5 upper-cases them
    if (keep == 0xFFFFu && !c.drop_after) {
        // four lanes out of five: the whole chunk survives -- one (unaligned) 16-byte LDS store
        const uint64_t both[2] = {lo, hi};
        __builtin_memcpy(dst, both, 16);
    } else if (c.valid == kCleanChunk && __popc(keep) == 15 && !c.drop_after) {
        // most of the rest: one byte (the newline) goes -- close the gap in registers, store 8 + 4 + 2 + 1 bytes
        const uint32_t j = __ffs(~keep & 0xFFFFu) - 1;
        uint64_t l2 = lo, h2 = hi >> 8;
        if (j < 8) {
            const uint64_t m = (1ull << (8 * j)) - 1ull;
            l2 = (lo & m) | (((lo >> 8) | (hi << 56)) & ~m);
        } else {
            const uint64_t m = (1ull << (8 * (j - 8))) - 1ull;
            h2 = (hi & m) | ((hi >> 8) & ~m);
        }
        const uint32_t h4 = (uint32_t)h2;
        const uint16_t h2b = (uint16_t)(h2 >> 32);
        __builtin_memcpy(dst, &l2, 8);
        __builtin_memcpy(dst + 8, &h4, 4);
        __builtin_memcpy(dst + 12, &h2b, 2);
        dst[14] = (uint8_t)(h2 >> 48);
    } else {
        uint32_t todo = keep | c.drop_after;
        while (todo) {                                  // header lines, N runs, the end of the text: byte by byte
            const uint32_t j = __ffs(todo) - 1;
            todo &= todo - 1;
            if (keep & (1u << j)) s_out[sh + at++] = (uint8_t)((j < 8 ? lo >> (8 * j) : hi >> (8 * (j - 8))) & 0xFFu);
            if (c.drop_after & (1u << j)) rec_off[hr++] = obase + at;   // the record begins where the next survivor will land
        }
    }
    __syncthreads();
    const uint32_t end = sh + tile_keep;                // LDS range [sh, end) holds this tile's output
    if (PACK) {
        uint32_t* words = reinterpret_cast<uint32_t*>(bases) + ((obase - sh) >> 4);
        for (uint32_t g = threadIdx.x; g * 16 < end; g += kCleanThreads) {
            const uint32_t lo = g * 16, hi = lo + 16;
            if (lo >= sh && hi <= end) words[g] = pack16(*reinterpret_cast<const uint4*>(s_out + lo));
            else {
                uint32_t w = 0;
                for (uint32_t x = lo < sh ? sh : lo; x < (hi < end ? hi : end); ++x) w |= (((uint32_t)s_out[x] >> 1) & 3u) << (30u - 2u * (x - lo));
                if (w) atomicOr(&words[g], w);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) rec_off[0] = 0;
        return;
    }
    uint8_t* gbase = bases + (obase - sh);              // 16-byte aligned
    for (uint32_t g = threadIdx.x; g * 16 < end; g += kCleanThreads) {
        const uint32_t lo = g * 16, hi = lo + 16;
        if (lo >= sh && hi <= end) {
            *reinterpret_cast<uint4*>(gbase + lo) = *reinterpret_cast<const uint4*>(s_out + lo);
        } else {                                        // first / last group of the tile: shared with the neighbours
            for (uint32_t x = lo < sh ? sh : lo; x < (hi < end ? hi : end); ++x) gbase[x] = s_out[x];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) rec_off[0] = 0;
}

// copy the bases of every selected super-k-mer into one compact buffer (dst offsets = prefix sums of len)
__global__ void k_gather_superkmers(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ rec_off,
                                    const spsp_superkmer* __restrict__ sk, const uint32_t* __restrict__ dst_off,
                                    uint32_t n_sk, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);   // one wave per super-k-mer
    if (i >= n_sk) return;
    const spsp_superkmer e = sk[i];
    const uint8_t* src = bases + rec_off[e.rec] + e.start;
    uint8_t* dst = out + dst_off[i];
    for (uint32_t j = threadIdx.x & 63; j < e.len; j += 64) dst[j] = src[j];
}
__global__ void k_gather_superkmers_packed(const uint32_t* __restrict__ words, const uint64_t* __restrict__ rec_off,
                                           const spsp_superkmer* __restrict__ sk, const uint32_t* __restrict__ dst_off,
                                           uint32_t n_sk, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);   // one wave per super-k-mer
    if (i >= n_sk) return;
    const spsp_superkmer e = sk[i];
    const uint64_t src = rec_off[e.rec] + e.start;
    uint8_t* dst = out + dst_off[i];
    for (uint32_t j = threadIdx.x & 63; j < e.len; j += 64) {
        const uint64_t q = src + j;
        dst[j] = "ACTG"[(words[q >> 4] >> (30u - 2u * (uint32_t)(q & 15u))) & 3u];      // int2nuc (utils.cpp:26-45)
    }
}
__global__ void k_sk_lens(const spsp_superkmer* __restrict__ sk, uint32_t n_sk, uint32_t* __restrict__ lens) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sk) lens[i] = sk[i].len;
}

int clean_device_impl(spsp_ctx* ctx, const uint8_t* d_text, uint64_t n_text, uint8_t** d_bases, uint64_t* n_bases,
                      uint64_t** d_rec_off, uint32_t* n_rec, bool pack) {
    if (((uintptr_t)d_text & 15u) != 0) { set_error("d_text must be 16-byte aligned"); return SPSP_ERR_ARG; }
    const uint64_t n_tiles = (n_text + kCleanTile - 1) / kCleanTile;
    if (n_tiles > 0x7fffffffull) { set_error("text too large for one call"); return SPSP_ERR_OVERFLOW; }
    int rc;
    if ((rc = ctx->i_tiles.reserve((size_t)(n_tiles + 1) * sizeof(TileSummary)))) return rc;
    if ((rc = ctx->i_entry.reserve((size_t)(n_tiles + 1) * 4))) return rc;
    if ((rc = ctx->i_outoff.reserve((size_t)(n_tiles + 1) * 8))) return rc;
    if ((rc = ctx->i_recbase.reserve((size_t)(n_tiles + 1) * 4))) return rc;
    const size_t packed_bytes = (size_t)((n_text + 15) / 16 + 64) * 4; // words of the survivors (never more than the input) + zero halo
    if (pack) {
        if ((rc = ctx->packed.reserve(packed_bytes))) return rc;
        SPSP_HIP(hipMemsetAsync(ctx->packed.p, 0, packed_bytes, ctx->stream));   // tile seams are ORed in; tail and halo stay zero
    } else if ((rc = ctx->bases.reserve((size_t)n_text + 64))) return rc;  // survivors never outnumber the input
    if ((rc = ctx->d_scalar.reserve(64))) return rc;
    uint64_t* totals = ctx->h_scalar + 4;
    if (n_tiles) {
        hipLaunchKernelGGL(k_clean_tiles, dim3((uint32_t)n_tiles), dim3(kCleanThreads), 0, ctx->stream, d_text, n_text,
                           ctx->i_tiles.as<TileSummary>());
        SPSP_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_clean_scan, dim3(1), dim3(1024), 0, ctx->stream, ctx->i_tiles.as<TileSummary>(), n_tiles,
                       ctx->i_entry.as<uint32_t>(), ctx->i_outoff.as<uint64_t>(), ctx->i_recbase.as<uint32_t>(), totals);
    SPSP_HIP(hipGetLastError());
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t kept = totals[0], recs = totals[1];
    if (recs > 0xfffffff0ull) { set_error("too many FASTA records for one call"); return SPSP_ERR_OVERFLOW; }
    if ((rc = ctx->rec_off.reserve((size_t)(recs + 1) * 8))) return rc;
    if (n_tiles) {
        if (pack) hipLaunchKernelGGL(k_clean_write<true>, dim3((uint32_t)n_tiles), dim3(kCleanThreads), 0, ctx->stream, d_text, n_text,
                                     ctx->i_entry.as<uint32_t>(), ctx->i_outoff.as<uint64_t>(), ctx->i_recbase.as<uint32_t>(),
                                     ctx->packed.as<uint8_t>(), ctx->rec_off.as<uint64_t>());
        else hipLaunchKernelGGL(k_clean_write<false>, dim3((uint32_t)n_tiles), dim3(kCleanThreads), 0, ctx->stream, d_text, n_text,
                                ctx->i_entry.as<uint32_t>(), ctx->i_outoff.as<uint64_t>(), ctx->i_recbase.as<uint32_t>(),
                                ctx->bases.as<uint8_t>(), ctx->rec_off.as<uint64_t>());
        SPSP_HIP(hipGetLastError());
    }
    // rec_off[0] (also for an empty text) and the closing offset
    const uint64_t ends[1] = {kept};
    if (!n_tiles) { const uint64_t zero = 0; SPSP_HIP(hipMemcpyAsync(ctx->rec_off.p, &zero, 8, hipMemcpyHostToDevice, ctx->stream)); }
    SPSP_HIP(hipMemcpyAsync(ctx->rec_off.as<uint64_t>() + recs, ends, 8, hipMemcpyHostToDevice, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    *d_bases = pack ? ctx->packed.as<uint8_t>() : ctx->bases.as<uint8_t>(); *n_bases = kept;
    *d_rec_off = ctx->rec_off.as<uint64_t>(); *n_rec = (uint32_t)recs;
    return SPSP_OK;
}

// bases of the selected super-k-mers -> one compact HOST buffer + offsets (malloc'd; n_sk+1 offsets)
int gather_superkmers_impl(spsp_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_rec_off, const spsp_superkmer* d_sk,
                           uint64_t n_sk, uint8_t** h_compact, uint32_t** h_off, bool packed) {
    *h_compact = nullptr; *h_off = nullptr;
    // destination offsets are 32-bit prefix sums: a super-k-mer is at most 2k - m <= 125 bases long, so below this many
    // of them the total cannot wrap (a whole-genome select-all of > 4 Gbp has to be split by the caller)
    if (n_sk > 0xffffffffull / 126) { set_error("too many selected super-k-mers for one call (%llu): split the input", (unsigned long long)n_sk); return SPSP_ERR_OVERFLOW; }
    uint32_t* off = (uint32_t*)malloc((size_t)(n_sk + 1) * 4);
    if (!off) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    off[0] = 0;
    if (n_sk == 0) { *h_off = off; *h_compact = (uint8_t*)malloc(1); return SPSP_OK; }
    int rc;
    if ((rc = ctx->i_lens.reserve((size_t)n_sk * 4)) || (rc = ctx->i_dst.reserve((size_t)(n_sk + 1) * 4))) { free(off); return rc; }
    hipLaunchKernelGGL(k_sk_lens, dim3((uint32_t)((n_sk + 255) / 256)), dim3(256), 0, ctx->stream, d_sk, (uint32_t)n_sk,
                       ctx->i_lens.as<uint32_t>());
    if ((rc = launch_scan_u32(ctx, ctx->i_lens.as<uint32_t>(), ctx->i_dst.as<uint32_t>(), n_sk, ctx->h_scalar + 6))) { free(off); return rc; }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(off); return hip_fail(e, "gather sizes", __FILE__, __LINE__); }
    const uint64_t total = ctx->h_scalar[6];
    if ((rc = ctx->i_compact.reserve((size_t)total + 64))) { free(off); return rc; }
    if (packed) hipLaunchKernelGGL(k_gather_superkmers_packed, dim3((uint32_t)((n_sk + 3) / 4)), dim3(256), 0, ctx->stream,
                                   reinterpret_cast<const uint32_t*>(d_bases), d_rec_off, d_sk, ctx->i_dst.as<uint32_t>(), (uint32_t)n_sk,
                                   ctx->i_compact.as<uint8_t>());
    else hipLaunchKernelGGL(k_gather_superkmers, dim3((uint32_t)((n_sk + 3) / 4)), dim3(256), 0, ctx->stream, d_bases, d_rec_off,
                            d_sk, ctx->i_dst.as<uint32_t>(), (uint32_t)n_sk, ctx->i_compact.as<uint8_t>());
    uint8_t* buf = (uint8_t*)malloc((size_t)total + 1);
    if (!buf) { free(off); set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    e = hipMemcpyAsync(buf, ctx->i_compact.p, (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(off, ctx->i_dst.p, (size_t)(n_sk + 1) * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(off); free(buf); return hip_fail(e, "gather copy", __FILE__, __LINE__); }
    *h_compact = buf; *h_off = off;
    return SPSP_OK;
}

// Where the sketch is built.  The device builder (spsp_build.hip) is a chain of ~25 launches and four host waits, ~2.5 ms
// whatever the size, then ~1.2 ns per k-mer place; the host builder costs ~110 ns per place and thread and runs file by file
// on the pipeline's workers.  A batch of bacterial genomes at -s 1000 (4 x 10^4 places) is faster on the host threads, a
// metagenome file (4 x 10^7 places: 0.23 s of host builder against 0.05 s) on the device: from 5 x 10^5 places on.
// SPSP_BUILD=device / host pins the choice (the tests run both).
bool build_on_device(uint64_t places) {
    static const char* e = getenv("SPSP_BUILD");
    if (e && e[0] == 'h') return false;
    if (e && e[0] == 'd') return true;
    return places >= 500000;
}
bool ingest_packs(const spsp_params* p) {
    static const bool ascii = getenv("SPSP_INGEST_ASCII") != nullptr;
    return !ascii && !(p->flags & SPSP_SCAN_PACKED_INPUT) && scan_reads_packed(p);
}

}  // namespace spsp

using namespace spsp;

extern "C" {

// the same with the cleaned bases as 2-bit words (the input form of SPSP_SCAN_PACKED_INPUT): what spsp_sketch_text /
// spsp_sketch_files put in front of the pair-table dense pass
int spsp_fasta_clean_packed_device(spsp_ctx* ctx, const void* d_text, uint64_t n_text, void** d_packed, uint64_t* n_bases,
                                   void** d_rec_off, uint32_t* n_rec) {
    if (!ctx || !d_packed || !n_bases || !d_rec_off || !n_rec || (n_text && !d_text)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    uint8_t* b = nullptr; uint64_t* o = nullptr;
    const int rc = clean_device_impl(ctx, (const uint8_t*)d_text, n_text, &b, n_bases, &o, n_rec, true);
    *d_packed = b; *d_rec_off = o;
    return rc;
}

int spsp_fasta_clean_device(spsp_ctx* ctx, const void* d_text, uint64_t n_text, void** d_bases, uint64_t* n_bases,
                            void** d_rec_off, uint32_t* n_rec) {
    if (!ctx || !d_bases || !n_bases || !d_rec_off || !n_rec || (n_text && !d_text)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    uint8_t* b = nullptr; uint64_t* o = nullptr;
    const int rc = clean_device_impl(ctx, (const uint8_t*)d_text, n_text, &b, n_bases, &o, n_rec);
    *d_bases = b; *d_rec_off = o;
    return rc;
}

// FASTA text (host, gunzipped) -> sketch payload with ingest, scan and super-k-mer gather on the GPU:
// only the selected super-k-mers' bases (~0.2 % of the genome at -s 1000) ever come back to the host.
int spsp_sketch_text(spsp_ctx* ctx, const spsp_params* p, double rate, const char* text, uint64_t n_text,
                     uint8_t** payload, uint64_t* payload_len, spsp_sketch_stats* stats) {
    if (!ctx || !payload || !payload_len || (n_text && !text)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    int rc = check_params(p);
    if (rc) return rc;
    SPSP_HIP(hipSetDevice(ctx->device));
    double t0 = now_s(), t1;
    if ((rc = ctx->i_text.reserve((size_t)n_text + 64))) return rc;
    if (n_text) SPSP_HIP(hipMemcpyAsync(ctx->i_text.p, text, (size_t)n_text, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d_bases = nullptr; uint64_t* d_off = nullptr; uint64_t n_bases = 0; uint32_t n_rec = 0;
    // the ingest writes 2-bit words when the dense pass of these parameters reads them (SPSP_INGEST_ASCII=1: A/B switch)
    const bool packed = ingest_packs(p);
    if ((rc = clean_device_impl(ctx, ctx->i_text.as<uint8_t>(), n_text, &d_bases, &n_bases, &d_off, &n_rec, packed))) return rc;
    t1 = now_s(); ctx->stages.ingest_s += t1 - t0; t0 = t1;
    spsp_superkmer* d_sk = nullptr; uint64_t n_sk = 0;
    spsp_params ps = *p;
    if (packed) ps.flags |= SPSP_SCAN_PACKED_INPUT;
    if ((rc = scan_device_impl(ctx, &ps, d_bases, n_bases, d_off, n_rec, &d_sk, &n_sk))) return rc;
    t1 = now_s(); ctx->stages.scan_s += t1 - t0; t0 = t1;
    std::vector<uint64_t> rec_off((size_t)n_rec + 1);
    std::vector<spsp_superkmer> sk((size_t)n_sk);
    SPSP_HIP(hipMemcpyAsync(rec_off.data(), d_off, rec_off.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (n_sk) SPSP_HIP(hipMemcpyAsync(sk.data(), d_sk, (size_t)n_sk * sizeof(spsp_superkmer), hipMemcpyDeviceToHost, ctx->stream));
    // the sketch builder on the device (spsp_build.hip; SPSP_BUILD=host: the host builder over the gathered super-k-mers, as until
    // round 5): only the finished payload comes back
    uint64_t places = 0;
    if (n_sk) {
        SPSP_HIP(hipStreamSynchronize(ctx->stream));                   // (the stream's copy queued above)
        for (const spsp_superkmer& e : sk) places += e.len >= p->k ? e.len - p->k + 1 : 0;
    }
    if (build_on_device(places)) {
        const uint32_t fsk[2] = {0u, (uint32_t)n_sk};
        std::vector<std::string> bodies;
        std::vector<uint64_t> fst;
        rc = n_sk <= 0xfffffff0ull ? sketch_build_device_impl(ctx, p, d_bases, packed, d_off, d_sk, n_sk, fsk, 1, &bodies, &fst) : SPSP_ERR_OVERFLOW;
        if (rc != SPSP_ERR_OVERFLOW) {
            if (rc) return rc;
            SPSP_HIP(hipStreamSynchronize(ctx->stream));               // (the copies of the stream and the offsets queued above)
            spsp_sketch_stats st;
            if ((rc = sketch_stream_stats(p, rec_off.data(), n_rec, sk.data(), n_sk, &st))) return rc;
            st.actual_minimizer_number = fst[0]; st.seen_kmers_at_reconstruction = fst[1];
            st.seen_superkmers_at_reconstruction = fst[2]; st.seen_max_superkmers_at_reconstruction = fst[3];
            std::string head;
            sketch_header_line(p->k, p->m, st.selected_kmer_number, rate, head);
            *payload = (uint8_t*)malloc(head.size() + bodies[0].size() + 1);
            if (!*payload) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
            memcpy(*payload, head.data(), head.size());
            memcpy(*payload + head.size(), bodies[0].data(), bodies[0].size());
            *payload_len = head.size() + bodies[0].size();
            if (stats) *stats = st;
            ctx->stages.build_s += now_s() - t0;
            if (stats && (p->flags & SPSP_SCAN_STATS)) {
                t0 = now_s();
                rc = count_superkmers_impl(ctx, p, d_bases, n_bases, d_off, n_rec, &stats->total_superkmer_number, packed, 0);
                stats->total_kmer_number = stats->read_kmer;
                ctx->stages.scan_s += now_s() - t0;
            }
            return rc;
        }
        rc = SPSP_OK;                                                  // beyond the device builder's numbering: the host builder below
    }
    uint8_t* compact = nullptr; uint32_t* coff = nullptr;
    if ((rc = gather_superkmers_impl(ctx, d_bases, d_off, d_sk, n_sk, &compact, &coff, packed))) return rc;   // synchronises the stream
    // -a > 1: the k-mers are counted here, over the gathered super-k-mers still on the device, and the host builder
    // indexes the usable ones only (SPSP_HOST_ABUNDANCE=1 leaves the counting to the builder, as round 1 did)
    uint8_t* kflags = nullptr; uint64_t n_occ = 0;
    static const bool host_abundance = getenv("SPSP_HOST_ABUNDANCE") != nullptr;
    if (p->abundance > 1 && !host_abundance) {
        rc = abundance_flags_impl(ctx, p, d_sk, n_sk, &kflags, &n_occ);
        if (rc == SPSP_ERR_OVERFLOW) { rc = SPSP_OK; kflags = nullptr; }   // too many occurrences for 32-bit numbering: the host counts
        if (rc) { free(compact); free(coff); return rc; }
    }
    t1 = now_s(); ctx->stages.gather_s += t1 - t0; t0 = t1;
    rc = sketch_build_core(p, rate, rec_off.data(), n_rec, sk.data(), n_sk, nullptr, compact, coff, payload, payload_len, stats, kflags);
    free(kflags);
    ctx->stages.build_s += now_s() - t0;
    if (!rc && stats && (p->flags & SPSP_SCAN_STATS)) {   // print_stat's counters over ALL super-k-mers (SubSampler.cpp:429-430,451-452)
        t0 = now_s();
        rc = count_superkmers_impl(ctx, p, d_bases, n_bases, d_off, n_rec, &stats->total_superkmer_number, packed, 0);
        stats->total_kmer_number = stats->read_kmer;      // every k-mer of a record lies in exactly one super-k-mer
        ctx->stages.scan_s += now_s() - t0;
    }
    free(compact); free(coff);
    return rc;
}

}  // extern "C"
