// Internal declarations shared by the libspsp translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/spsp.h"

namespace spsp {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SPSP_HIP(call)                                                      \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return spsp::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

// Grow-only device buffer owned by a context (re-used across calls so the
// steady state of a batch loop performs no hipMalloc).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// One candidate m-mer: hash <= threshold.  32 bytes.
struct Hit {
    uint64_t pos;    // absolute position of the m-mer in the concatenated bases
    uint64_t hash;   // XXH64(canonical m-mer, seed 1312)
    uint32_t canon;  // canonical 2-bit value
    uint32_t rec;    // record index
    uint32_t flags;  // bit0: occurrence is reverse strand; bit1: usable (inside a record of length >= k)
    uint32_t pad;
};

}  // namespace spsp

namespace spsp {
// HIP-event pairs recorded around one kind of launch
struct EventLog {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> used, spare;
};
enum { kEvDense = 0, kEvScan = 1, kEvAccumulate = 2, kEvCompare = 3, kEvScatter = 4, kEvGroup = 5, kEvKinds = 6 };
}  // namespace spsp

namespace spsp {
// a scan queued by scan_begin_impl and not yet collected by scan_end_impl
struct ScanJob {
    bool pending = false, empty = false;
    int redo_from = 0;           // first stage the next attempt runs: 0 dense, 1 compact (lists intact), 2 write pass
    bool segments = false;       // the attempt in flight is the scan by segments (spsp_stats.hip: thresholds that select nearly everything)
    bool use_bitmap = false;     // hit bitmap + k_expand (dense selections, list-overflow fallback) instead of per-wave lists
    bool lists = false;          // the attempt in flight used per-wave hit lists
    spsp_params p{};
    const uint8_t* d_bases = nullptr;
    uint64_t n_bases = 0, n_tiles = 0;
    const uint64_t* d_rec_off = nullptr;
    uint32_t n_rec = 0, hits_cap = 0, out_cap = 0;
    uint32_t n_lists = 0, list_cap = 0;   // geometry of the per-wave hit lists of the attempt in flight
    uint64_t rows_per_wave = 0;
};
struct CompareJob;   // spsp_compare.hip
// the key extraction queued by sketch_keys_begin_impl: what its last stages need when _end has to queue them (spsp_keys.hip)
struct KeysJob {
    bool has_hi = false, flat = false;   // flat: raw records by one lane per super-k-mer + per-genome LDS sort (the sorted form's kernels)
    uint32_t n_genomes = 0;
    uint64_t bound = 0;          // k-mer places of all genomes together (the extent of the staging arrays)
    uint32_t abundance = 1;
    bool big_queued = false;     // the table kernels for genomes beyond the LDS forms were queued by _begin
};
}  // namespace spsp

struct spsp_ctx {
    bool timing = false;          // any kind enabled
    uint32_t timing_mask = 0;     // bit k: regions of kind k (kEvDense ...) are bracketed by events
    spsp::EventLog evlog[spsp::kEvKinds];
    // begin/end bracket for one timed region; no-ops unless timing is on
    int ev_begin(int kind);
    int ev_end(int kind);
    bool ev_pair(int kind, hipEvent_t* start, hipEvent_t* stop);   // events for hipExtLaunchKernelGGL (no stream packets)
    bool ev_open[spsp::kEvKinds] = {};   // a begin without its end is outstanding
    uint32_t timing_every = 1;           // spsp_timing_sample: every n-th region of a kind is bracketed
    uint32_t ev_seq[spsp::kEvKinds] = {};
    spsp_stage_times stages{};    // whole-file drivers: wall seconds per stage (spsp_stage_times_read)
    int device = 0;
    int n_cu = 256;          // compute units this context's stream may use (spsp_set_cu_count)
    int n_cu_device = 256;   // ... of the device
    int dense_blocks_per_cu = 1;   // table variants of the dense pass: 1024-lane workgroups per CU (spsp_set_cu_count)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t tail_stream = nullptr;   // sparse stages of the scan (spsp_scan_tail_stream); nullptr = the main stream
    bool own_tail_stream = false;
    hipStream_t sparse_stream() const { return tail_stream ? tail_stream : stream; }
    uint64_t* h_scalar = nullptr;  // pinned, 16 slots: [0] hits, [1] super-k-mers (scan); [4..6] ingest totals; [8..10] compare flags
    hipEvent_t dense_done = nullptr;   // recorded behind every dense pass (unless a timing event already is)
    hipEvent_t dense_marker = nullptr; // what spsp_wait_dense waits on
    hipEvent_t tail_event = nullptr;   // spsp_wait_stream: marks the current end of this context's stream
    hipEvent_t scan_done = nullptr;    // behind the last kernel of the queued scan: what spsp_scan_device_end waits on
    hipEvent_t compare_done = nullptr; // likewise for the queued comparison
    uint64_t learnt_on = 0;            // offsets' fingerprint of the collection order_quiet / multi_quiet were learnt on
    uint32_t multi_quiet = 0;          // comparisons that leave out the has-a-list bits (the last one had lists for most records)
    uint32_t order_quiet = 0;          // comparisons that skip the making of a row order (the last one came in a good order of its own)
    bool attr_pair_set = false, attr_single_set = false, attr_bloom_set = false, attr_small_set = false, attr_group_set = false, attr_group_hi_set = false, attr_scatter_set = false, attr_scatter_tiles_set = false, attr_sort_set = false, attr_order_set = false;   // dynamic-LDS attributes set on this context's device
    spsp::ScanJob scan_job;
    spsp::CompareJob* compare_job = nullptr;
    // spsp_sketch_keys_device_begin / _end (spsp_keys.hip)
    bool keys_pending = false, keys_has_hi = false, attr_keys_set = false, keys_flags_clear = false, attr_dedupe_set = false;
    bool keys_unordered = false;       // spsp_compare_keys_unordered: the comparisons of this context do not insist on sorted sketches
    uint32_t keys_genomes = 0;
    bool keys_sorted = false;          // the pending extraction promised sorted sketches (its big genomes are sorted in _end)
    // a comparison whose caller wants the pair matrix as sparse cells (spsp_multi.hip: compare_cells_run)
    struct CellsReq { unsigned long long* cells = nullptr; unsigned long long cap = 0; unsigned long long* count = nullptr; bool armed = false, direct = false; } cells_req;
    spsp::KeysJob keys_job;
    bool keys_expect_big = false;      // the last extraction collected on this context met a genome beyond the LDS forms
    uint32_t keys_big_genomes = 0;     // genomes of the last collected extraction that went through the global-memory stages (spsp_bigkeys.hip)
    hipEvent_t keys_done = nullptr;
    uint32_t* h_keys = nullptr;        // pinned: genome record ranges in, key offsets + overflow report out
    size_t h_keys_cap = 0;
    uint8_t* h_text = nullptr;         // pinned staging for a whole FASTA file (spsp_sketch_file reads plain files straight into it)
    size_t h_text_cap = 0;
    uint64_t* h_skoff = nullptr;       // pinned staging for the sketch offsets of a queued comparison
    size_t h_skoff_cap = 0;
    struct ReadRegion { uint8_t* p = nullptr; size_t cap = 0; };
    std::vector<ReadRegion> h_read_regions;   // spsp_compare_files: one per reader thread, the payloads of its files back to back (kept: page faults and munmaps per call otherwise)
    // scan workspace
    spsp::DevBuf bases, rec_off, bitmap, tile_count, tile_off, hits, emit_count, scan_tmp, d_scalar, seg_a, seg_b;
    spsp::DevBuf wave_hits, wave_cnt;    // per-wave hit lists of the table variants of the dense pass
    spsp::DevBuf packed, unpacked;       // SPSP_SCAN_PACKED_INPUT: spsp_pack_bases_device's output; ASCII copy for the variants that need one
    spsp::DevBuf st_count, st_open, st_total, st_over;      // print_stat counting pass (spsp_stats.hip)
    uint64_t hits_cap = 0, out_cap = 0;  // entries the sparse-stage buffers of the call in flight are sized for (grow on overflow)
    // what the last overflow taught: hits / super-k-mers per base at that threshold (scan_begin_impl sizes the next call by it)
    bool learn_valid = false;
    uint64_t learn_threshold = 0;
    double learn_hits_per_base = 0, learn_out_per_base = 0;
    uint64_t list_cap = 0;               // hits one wave's list holds (grows on overflow)
    uint64_t list_cap_threshold = 0;     // ... learnt at this threshold (another threshold starts from its own expectation)
    // LDS pre-filter table cache (keyed by m, threshold)
    spsp::DevBuf filter;
    uint32_t filter_m = 0;
    uint64_t filter_thr = 0;
    uint32_t filter_shift = 0;
    bool filter_valid = false;
    // blocked Bloom filter over canonical m-mers (k_dense_bloom)
    spsp::DevBuf bloom;
    uint32_t bloom_m = 0;
    uint64_t bloom_thr = 0;
    bool bloom_valid = false;
    // pair-lookup table cache (64 KiB table + 8 KiB key bitmap)
    spsp::DevBuf pairtab;
    uint32_t pair_m = 0;
    uint64_t pair_thr = 0;
    bool pair_valid = false;
    // ingest workspace (GPU-side getLineFasta + clean_dna)
    spsp::DevBuf i_text, i_tiles, i_entry, i_outoff, i_recbase, i_lens, i_dst, i_compact;
    // compare workspace
    spsp::DevBuf c_min, c_lo, c_hi, c_table, c_owner, c_rowid, c_row, c_matrix, c_inter, c_flags, c_skoff, c_slot_lo, c_slot_hi, c_slot_mn, c_part_cnt, c_recs, c_where, c_lref, c_filter, c_bits, c_sig, c_order, c_multi, scan_blocks;
    uint64_t spill_expect = 0;     // records the last unfiltered partition-form comparison had in overflowed parts (0: none) -- see spill_plan
    uint32_t filter_skipped = 0;
    double filter_ratio = 1.0;   // records dealt into parts per owned key in the last filtered comparison (sizes the next one's parts)
    spsp::DevBuf x_cnt, x_off, x_begin, x_end, x_tot;   // key-partitioned exchange (spsp_compare.hip)
    spsp::DevBuf bl_hist, bl_keys, bl_vals, bl_meta, bl_lo, bl_hi, bl_pmin, bl_pb, bl_slot, bl_first, bl_codes, bl_text, bl_outs, bl_out;   // sketch builder on the device (spsp_build.hip)
    spsp::DevBuf dc_text, dc_desc, dc_mn, dc_lo, dc_hi, dc_meta, dc_walk;   // bulk sketch decode (spsp_decode.hip)
    spsp::DevBuf a_cnt, a_off, a_mn, a_lo, a_hi, a_slot, a_slot_of, a_flags, a_seg;   // -a abundance pass (spsp_abund.hip)
    // genomes / sketches beyond the per-segment LDS forms (spsp_bigkeys.hip): output slices, the open-addressing table in HBM
    // (slot words carry the epoch of the call that claimed them: never cleared between calls), the sort's tile list
    spsp::DevBuf b_mn, b_lo, b_hi, b_table, b_tiles, b_seg;
    std::vector<uint64_t> m_h_skoff;                          // host arrays a queued slot unpack reads (spsp_multi.hip)
    std::vector<uint32_t> m_h_tot, m_h_base;
    bool m_slots_job = false;                                 // the pending comparison came from exchange slots: its record check is read behind compare_end
    spsp::DevBuf m_send, m_recv, m_cells, m_mn, m_lo, m_hi;   // key-partitioned split (spsp_multi.hip): slots out / in, sparse cells, unpacked keys
    uint32_t big_epoch = 0;
};

namespace spsp {
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// scan pipeline (spsp_scan.hip)
int scan_device_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                     const uint64_t* d_rec_off, uint32_t n_rec, spsp_superkmer** d_out, uint64_t* n_out);
int scan_begin_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                    const uint64_t* d_rec_off, uint32_t n_rec);
int scan_end_impl(spsp_ctx* ctx, spsp_superkmer** d_out, uint64_t* n_out);
int seg_scan_count(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, uint64_t n_bases, const uint64_t* d_rec_off, uint32_t n_rec);
int seg_scan_emit(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, uint64_t n_bases, const uint64_t* d_rec_off, uint32_t n_rec,
                  spsp_superkmer* d_out, uint64_t out_cap);
int scan_hits_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases,
                   uint64_t* n_hits);
// compare pipeline (spsp_compare.hip)
int compare_device_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                        const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit, uint32_t row_first,
                        uint32_t row_stride, uint32_t* d_inter);
int compare_device_begin_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo,
                              const uint64_t* d_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t row_limit,
                              uint32_t row_first, uint32_t row_stride, uint32_t* d_inter);
int compare_end_impl(spsp_ctx* ctx);
void compare_job_drop(spsp_ctx* ctx);
int check_params(const spsp_params* p);
int pack_bases_impl(spsp_ctx* ctx, const uint8_t* d_bases, uint64_t n_bases, uint32_t** d_packed);
// every super-k-mer of the input, selected or not (spsp_stats.hip)
// packed: d_bases holds 2-bit words (16 bases per dword); base0: first base of rec_off[0]'s record in d_bases, added to every offset
int count_superkmers_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, uint64_t n_bases, const uint64_t* d_rec_off,
                          uint32_t n_rec, uint64_t* total, bool packed = false, uint64_t base0 = 0, const uint32_t* h_file_rec = nullptr, uint32_t n_files = 1);
// bulk sketch decode (spsp_decode.hip): one stored super-k-mer of a sketch payload, as the host's structure walk finds it
struct DecDesc {
    uint64_t off;    // byte offset in the payload (later: in the concatenated payload buffer): blob bytes (kind 0) / prefix line (kind 1)
    uint32_t mn;     // minimizer of the bucket (2-bit value)
    uint32_t info;   // bits 0-1 kind: 0 maximal super-k-mer in the blob, 1 "prefix\nsuffix\n" pair, 2 the bare minimizer (k == m);
                     // kind 1: prefix length bits 2-9, suffix length bits 10-17
    uint32_t out;    // first raw key of this super-k-mer
    uint32_t pad;
};
struct ParsedSketch {
    uint32_t k = 0, m = 0;
    bool standard = true;        // laid out as the sketcher writes it: the GPU path applies
    uint64_t n_keys = 0;         // raw keys (duplicates included)
    std::vector<DecDesc> desc;   // offsets relative to the payload; `out` relative to the sketch
};
// structure of one payload: header + bucket boundaries + line ends (spsp_host.cpp: pure host code, fuzzed under ASan)
int sketch_parse_structure_host(const uint8_t* payload, uint64_t len, ParsedSketch* P);
// decode on the GPU + all-vs-all + copy back: the device half of spsp_compare_files (spsp_decode.hip)
// (inter: n x n, zero on entry; *mirrored = every written cell (i, j > i) was also stored at (j, i))
int compare_payloads_impl(spsp_ctx* ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n, const int* extra_has,
                          const uint32_t* extra_mn, uint32_t n_query, uint32_t* k_out, uint32_t* m_out, uint32_t* inter, uint64_t* card,
                          bool* mirrored = nullptr, std::vector<uint64_t>* cells_out = nullptr);
// (cells_out, for 1024 <= n <= 65535: the non-zero cells i << 48 | j << 32 | count, every pair once, INSTEAD of the matrix: inter may be null)
int sketch_decode_device_impl(spsp_ctx* ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n,
                              const int* extra_has, const uint32_t* extra_mn, uint32_t* k_out, uint32_t* m_out, uint64_t* sk_off);
// ingest (spsp_ingest.hip)
// pack: the cleaned bases leave as 2-bit words (ctx->packed: 16 bases per dword, first base in bits 31:30, zero-filled tail
// + 256 readable bytes) instead of ASCII (ctx->bases); *d_bases then points at the words
int clean_device_impl(spsp_ctx* ctx, const uint8_t* d_text, uint64_t n_text, uint8_t** d_bases, uint64_t* n_bases,
                      uint64_t** d_rec_off, uint32_t* n_rec, bool pack = false);
// does the dense pass chosen for these parameters read 2-bit input directly? (spsp_scan.hip)
bool scan_reads_packed(const spsp_params* p);
bool build_on_device(uint64_t places);    // the sketch builder on the device from 5 x 10^5 k-mer places on (SPSP_BUILD=device / host pins it)
// should the whole-file drivers let the ingest write 2-bit words for these parameters? (spsp_ingest.hip)
bool ingest_packs(const spsp_params* p);
int gather_superkmers_impl(spsp_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_rec_off, const spsp_superkmer* d_sk,
                           uint64_t n_sk, uint8_t** h_compact, uint32_t** h_off, bool packed = false);
// out[i] = sum(in[0..i)), out[n] = total (also stored to *total_host, pinned)
int launch_scan_u32(spsp_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint64_t* total_host);
int sketch_stream_stats(const spsp_params* p, const uint64_t* rec_off, uint32_t n_rec, const spsp_superkmer* sk, uint64_t n_sk, spsp_sketch_stats* st);
void sketch_header_line(uint32_t k, uint32_t m, uint64_t selected_kmers, double rate, std::string& out);
// the sketch builder on the device (spsp_build.hip): the bodies of n_files sketches from the scan's stream; file_stats: per file
// actual_minimizer_number, seen_kmers_at_reconstruction, seen_superkmers_at_reconstruction, seen_max_superkmers_at_reconstruction
int sketch_build_device_impl(spsp_ctx* ctx, const spsp_params* p, const uint8_t* d_bases, bool packed, const uint64_t* d_rec_off, const spsp_superkmer* d_sk,
                             uint64_t n_sk, const uint32_t* h_file_sk, uint32_t n_files, std::vector<std::string>* bodies, std::vector<uint64_t>* file_stats);
// zstr-style inflate of a whole buffer: gzip / zlib members, or the bytes as they are (spsp_host.cpp)
int inflate_all_host(const uint8_t* in, size_t n, std::vector<uint8_t>& out);
// host sketch builder over per-super-k-mer base pointers (spsp_host.cpp)
// -a on the device (spsp_abund.hip): per k-mer occurrence of the gathered super-k-mers, bit 0 usable, bit 1 first of a dropped k-mer
int abundance_flags_impl(spsp_ctx* ctx, const spsp_params* p, const spsp_superkmer* d_sk, uint64_t n_sk, uint8_t** h_flags, uint64_t* n_occ_out,
                         const uint32_t* h_seg_sk = nullptr, uint32_t n_seg = 0);
// exchange slots of the key-partitioned split (wire format: spsp_compare.hip, sender; spsp_multi.hip, receiver)
constexpr uint32_t kSlotMagic = 0x4c535053u;   // "SPSL"
constexpr uint32_t kMaxParts = 64;
__host__ __device__ inline uint64_t slot_rec_off(uint32_t n) { return 16 + (uint64_t)((n + 1) & ~1u) * 4; }
__host__ __device__ inline uint32_t slot_words(uint32_t k) { return k > 32 ? 3u : 2u; }
__host__ __device__ inline uint64_t slot_bytes(uint32_t n, uint32_t cap, uint32_t k) {
    return slot_rec_off(n) + (uint64_t)cap * slot_words(k) * 8;
}
int partition_keys_impl(spsp_ctx* ctx, uint32_t k, const uint32_t* d_min, const uint64_t* d_lo, const uint64_t* d_hi,
                        const uint64_t* h_sk_off, uint32_t n, uint32_t parts, uint32_t cap, uint8_t* d_slots);
int compare_slots_begin_impl(spsp_ctx* ctx, uint32_t k, const uint8_t* d_slots, uint32_t parts, uint32_t n, uint32_t cap,
                             uint32_t* d_inter, const uint8_t* h_headers = nullptr);
int slots_bad_record(spsp_ctx* ctx);
// sparse form of a pair matrix (spsp_multi.hip): non-zero cells (i < j) as i << 48 | j << 32 | count
int matrix_cells_impl(spsp_ctx* ctx, const uint32_t* d_inter, uint32_t n, uint32_t row_first, uint32_t row_limit, uint64_t* d_cells,
                      uint64_t cap, uint64_t* n_cells);
// queue a comparison with begin() and return its pair matrix as sparse cells: straight from the row sums where the form allows
// it (d_scratch then stays unwritten), else through the dense matrix in d_scratch (n x n uint32) and k_matrix_cells
int compare_cells_run(spsp_ctx* ctx, const std::function<int()>& begin, uint32_t n, uint32_t row_limit, uint32_t* d_scratch, uint64_t* d_cells,
                      uint64_t cap, uint64_t* n_cells, DevBuf* grow = nullptr);
// decode + all-vs-all over several contexts (one per device, or several on one): the device half of spsp_compare_files_multi
int compare_payloads_multi(spsp_ctx* const* ctxs, uint32_t n_ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n,
                           const int* extra_has, const uint32_t* extra_mn, uint32_t n_query, uint32_t* k_out, uint32_t* m_out,
                           uint32_t* inter, uint64_t* card, bool* mirrored = nullptr, std::vector<uint64_t>* cells_out = nullptr);
// spsp_bigkeys.hip: distinct keys of flagged segments through one table in HBM (queued, no host wait); segments sorted in place
int big_dedupe_launch(spsp_ctx* ctx, bool has_hi, const uint32_t* raw_mn, const uint64_t* raw_lo, const uint64_t* raw_hi,
                      const uint32_t* d_seg_first, const uint32_t* d_seg_cnt, const uint32_t* d_seg_big, uint32_t n_seg, uint64_t n_places,
                      const uint32_t* d_gate, uint32_t abundance, uint32_t* out_mn, uint64_t* out_lo, uint64_t* out_hi, uint32_t* d_distinct);
int big_sort_segments(spsp_ctx* ctx, bool has_hi, uint32_t* mn, uint64_t* lo, uint64_t* hi, uint32_t* t_mn, uint64_t* t_lo, uint64_t* t_hi,
                      const std::vector<std::pair<uint32_t, uint32_t>>& segs);
int sketch_build_core(const spsp_params* p, double rate, const uint64_t* rec_off, uint32_t n_rec, const spsp_superkmer* sk,
                      uint64_t n_sk, const uint8_t* bases, const uint8_t* compact, const uint32_t* compact_off,
                      uint8_t** payload, uint64_t* payload_len, spsp_sketch_stats* stats, const uint8_t* kmer_flags);
}  // namespace spsp
