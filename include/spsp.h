/* spsp.h -- C-ABI of libspsp: the MI355X-native drop-in for SuperSampler's
 * data-parallel hot path (sketch scan + all-vs-all sketch comparison).
 *
 * The reference (TimRouze/supersampler) has no FFI seam of its own: its public
 * surface is two CLIs and three file formats.  This header is
 * the seam a maintainer would bind instead of calling the member functions
 * cited next to each entry point (paths relative to the reference tree).
 *
 * Conventions: plain pointers and sizes, no exceptions cross the boundary,
 * 0 = ok / negative = error (text from spsp_last_error(), thread-local).
 * Buffers returned through `**out` are owned by the library and released with
 * spsp_free().  A context is bound to
 * one HIP device + one stream and is not shared between threads: the CLIs use
 * one context (= one stream) per in-flight genome.
 *
 * Every entry point whose name does not end in _host runs on the GPU and
 * FAILS (SPSP_ERR_NO_DEVICE) when no gfx950 device is usable -- there is no
 * CPU fallback in this library.
 */
#ifndef SPSP_H
#define SPSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPSP_OK 0
#define SPSP_ERR_ARG (-1)
#define SPSP_ERR_NO_DEVICE (-2)
#define SPSP_ERR_HIP (-3)
#define SPSP_ERR_NOMEM (-4)
#define SPSP_ERR_IO (-5)
#define SPSP_ERR_FORMAT (-6)
#define SPSP_ERR_OVERFLOW (-7)

typedef struct spsp_ctx spsp_ctx;

/* Scan parameters: the state `Subsampler::Subsampler` derives from the command
 * line (SubSampler.h:63-88). k, m odd, m <= 15, m <= k <= 63. */
typedef struct spsp_params {
    uint32_t k;
    uint32_t m;
    uint64_t threshold; /* selection_threshold, from spsp_threshold_host() */
    uint32_t abundance; /* -a, used by the sketch builder only */
    uint32_t flags;     /* SPSP_SCAN_* */
} spsp_params;

#define SPSP_SCAN_DEFAULT 0u
#define SPSP_SCAN_DIRECT_HASH 1u /* force XXH64 at every position (no LDS pre-filter) */
#define SPSP_SCAN_LDS_FILTER 2u  /* force the 2^20-bit memoised LDS pre-filter (one lookup per position) */
#define SPSP_SCAN_PAIR_FILTER 4u /* force the 64 KiB pair table (one lookup per two positions, m >= 9) */
#define SPSP_SCAN_PACKED_INPUT 32u /* device forms only: d_bases holds 2-bit codes (spsp_pack_bases_device), not ASCII */
#define SPSP_SCAN_BLOOM_FILTER 16u /* force the blocked Bloom filter over canonical m-mers (m = 13 or 15) */
#define SPSP_SCAN_STATS 8u       /* spsp_sketch_text / spsp_sketch_file: also count EVERY super-k-mer of the input
                                    (total_superkmer_number of print_stat, SubSampler.cpp:430,452) -- an extra pass */

/* One selected super-k-mer == one call of Subsampler::handle_superkmer
 * (SubSampler.cpp:426,448): ref.substr(start,len) of record `rec`, its
 * minimizer (canonical 2-bit value) and whether that minimizer reads
 * reverse-complemented in the genome. Emitted in genome order. */
typedef struct spsp_superkmer {
    uint32_t rec;
    uint32_t minimizer;
    uint64_t start; /* offset inside the record */
    uint32_t len;
    uint32_t rev;
} spsp_superkmer;

/* One sketch as the comparator sees it after Comparator.cpp:186-260: the
 * DISTINCT canonical k-mers of every bucket, sorted by (minimizer, kmer_hi,
 * kmer_lo). kmer_hi may be NULL when k <= 32. */
typedef struct spsp_sketch_view {
    const uint32_t* minimizer;
    const uint64_t* kmer_lo;
    const uint64_t* kmer_hi;
    uint64_t n;
} spsp_sketch_view;

/* ------------------------------------------------------------ lifecycle -- */
/* how many gfx950 devices this process sees (devices 0 .. count - 1 are usable with spsp_create); 0 = none, see spsp_last_error */
int spsp_device_count(void);
int spsp_create(int device, void* hip_stream /* hipStream_t or NULL = own stream */, spsp_ctx** out);
void spsp_destroy(spsp_ctx* ctx);
const char* spsp_last_error(void);
const char* spsp_version(void);
void spsp_free(void* host_ptr);
/* CU partitioning (MI355X: 256 CUs in 8 XCDs).  spsp_stream_create_cus makes a HIP stream whose kernels may only
 * run on the `n_cu` logical compute units [first_cu, first_cu + n_cu) (hipExtStreamCreateWithCUMask; the mask's bits
 * are dealt round-robin over the XCDs -- measured, tools/exp/exp_cumask.hip -- so a range of 8j bits is j CUs of every
 * XCD).  A pipeline gives the bandwidth-bound dense pass most of the chip and the latency-bound kernels (sparse
 * stages, comparison) a few CUs of their own: they then overlap without slowing each other down, which they do when
 * they share CUs (a dense workgroup next to a comparison workgroup runs at half speed and the dense grid, a static
 * partition, waits for it).  Take n_cu in multiples of 32: a CU count that is not the same in every shader engine
 * (4 per XCD) makes the dispatcher double workgroups up on some CUs (dense pass 0.10 -> 0.15 ms with 240 CUs).
 * spsp_set_cu_count tells a context how many CUs its stream owns (grid sizing; 0 = all) and how many dense-pass
 * workgroups to put on each: 1 (the default) leaves half of every CU's wave slots and LDS to other streams' kernels,
 * 2 is for a stream that has its CUs to itself. */
int spsp_stream_create_cus(int device, uint32_t first_cu, uint32_t n_cu, void** hip_stream);
int spsp_stream_destroy(int device, void* hip_stream);
int spsp_set_cu_count(spsp_ctx* ctx, uint32_t n_cu, uint32_t dense_blocks_per_cu);
/* copy `bytes` from a device buffer returned by this library to host memory, on the context's stream, and wait */
int spsp_copy_to_host(spsp_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);

/* ------------------------------------------------------------ measurement -- */
/* HIP-event timing of the two dominant kernels and of the whole pipelines, on
 * the context's stream (bench.py's `roofline` numbers come from here). */
typedef struct spsp_timing {
    double dense_ms;       /* k_dense: hash + threshold over every m-mer position */
    uint64_t dense_launches;
    double scan_ms;        /* whole spsp_scan_device pipeline */
    uint64_t scan_calls;
    double accumulate_ms;  /* k_accumulate: colour-matrix row sums */
    uint64_t accumulate_launches;
    double compare_ms;     /* whole spsp_compare_device pipeline */
    uint64_t compare_calls;
    double scatter_ms;     /* k_parts_scatter: keys dealt into hash classes (partition form of the comparison) */
    uint64_t scatter_launches;
    double group_ms;       /* k_parts_group: per-class LDS dictionary -> sketch lists */
    uint64_t group_launches;
} spsp_timing;
/* `kinds` = OR of SPSP_TIME_* (0 = off).  Every bracketed region costs two event records on the stream, i.e. two
 * packets the following kernels queue behind (~5 us each on an otherwise idle queue): enable what you read.
 * The scan / compare pipeline brackets close behind the FIRST attempt a begin call queues: the re-run of a call
 * whose buffers overflowed (or whose key classes did) is not inside the bracket. */
#define SPSP_TIME_DENSE 1
#define SPSP_TIME_SCAN 2
#define SPSP_TIME_ACCUMULATE 4
#define SPSP_TIME_COMPARE 8
#define SPSP_TIME_PARTS 16 /* scatter + group kernels */
#define SPSP_TIME_ALL 31
int spsp_timing_enable(spsp_ctx* ctx, int kinds);
/* bracket only every `every`-th region of each kind (1 = all): the two event packets of a bracket cost a pipelined
 * stream ~4 us each (bench.py: 8.6 us of a 0.123 ms step with every dense pass bracketed) */
int spsp_timing_sample(spsp_ctx* ctx, uint32_t every);
/* synchronises the stream, returns the totals since the previous read and resets them */
int spsp_timing_read(spsp_ctx* ctx, spsp_timing* out);

/* HBM calibration for the roofline's denominator ("calibrate with a device copy kernel on the box and use
 * the measured figure"; the reference has no counterpart -- its only timer is Comparator.cpp:499-509): a streaming copy
 * and a streaming read of two freshly allocated buffers of `bytes` each (take >= 1 GiB: the 256 MiB Infinity Cache must
 * not serve them), `reps` launches each behind two warm-up launches, timed with HIP events on the context's stream and
 * on the CUs that stream owns.  copy_GBps counts bytes read + bytes written. */
typedef struct spsp_hbm_rates {
    double copy_GBps, copy_ms;   /* per launch */
    double read_GBps, read_ms;
    uint64_t bytes;
    uint32_t reps, n_cu;
} spsp_hbm_rates;
int spsp_measure_hbm_device(spsp_ctx* ctx, uint64_t bytes, uint32_t reps, spsp_hbm_rates* out);

/* ------------------------------------------------------------- path A ---- */
/* Subsampler::compute_threshold + ctor selection (SubSampler.cpp:622-631,
 * SubSampler.h:79-83). Host long double, as the reference. */
uint64_t spsp_threshold_host(uint32_t k, uint32_t m, double sampling_rate);

/* Replaces the scan loop SubSampler.cpp:357-455 with regular_minimizer_pos
 * :81-169 and unrevhash :64-67. `bases` = cleaned upper-case ASCII records
 * back to back (what getLineFasta returns, utils.cpp:706-718); rec_off has
 * n_rec+1 entries. Output: the handle_superkmer argument stream. */
int spsp_scan(spsp_ctx* ctx, const spsp_params* p, const uint8_t* bases, const uint64_t* rec_off,
              uint32_t n_rec, spsp_superkmer** out, uint64_t* n_out);

/* Same with everything resident in HBM (16-byte aligned d_bases). The result
 * stays on the device in a buffer OWNED BY THE CONTEXT (*d_out: n_out records,
 * valid until the next scan call on this context; do not free). The whole
 * pipeline is queued on the context's stream and the call returns after the one
 * host synchronisation that reads back *n_out. */
int spsp_scan_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                     const void* d_rec_off, uint32_t n_rec, void** d_out, uint64_t* n_out);

/* 2-bit input (0.25 B per k-mer hashed).  spsp_pack_bases_device turns `n_bases` cleaned ASCII bases on the
 * device into the packed form -- 16 bases per little-endian 32-bit word, first base in bits 31:30, A=0 C=1 T=2 G=3, the
 * last word zero-filled and 256 readable bytes behind it -- in a buffer the context owns (valid until its next pack call).
 * A scan whose spsp_params.flags carry SPSP_SCAN_PACKED_INPUT takes that buffer as d_bases (n_bases stays the number of
 * BASES; record offsets are base offsets as ever).  The pair-table and the blocked-Bloom dense passes (m >= 9 at coarse
 * sampling; m = 13 / 15 at fine sampling: BASELINE configs[1] and configs[4]) read it directly: a quarter of the traffic, no
 * packing arithmetic; the other variants unpack it into an ASCII copy first.  Same stream out. */
int spsp_pack_bases_device(spsp_ctx* ctx, const void* d_bases, uint64_t n_bases, void** d_packed);

/* The same call split at its host synchronisation, for callers that pipeline several streams (one context
 * per stream): _begin queues the whole scan on the context's stream and returns at once; _end waits for it,
 * and re-runs the affected stages in the rare call whose hit / super-k-mer buffers overflowed.  One scan
 * may be pending per context.  d_bases / d_rec_off must stay valid until _end returns. */
int spsp_scan_device_begin(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                           const void* d_rec_off, uint32_t n_rec);
int spsp_scan_device_end(spsp_ctx* ctx, void** d_out, uint64_t* n_out);
/* Optional second stream for the scan's sparse stages: everything behind the dense pass (hit compaction, cluster
 * replay) is queued on `hip_stream` (NULL = a stream the context creates itself; pass tail = 0 to switch back),
 * ordered behind the dense pass by an event.  Several contexts created on ONE stream then run their dense passes
 * back to back in that stream's order while each context's sparse stages overlap the next dense pass. */
int spsp_scan_tail_stream(spsp_ctx* ctx, int tail, void* hip_stream);
/* Stream ordering between two contexts of one device: work queued on `waiter` after this call starts only
 * once the dense pass of `scanner`'s most recently queued scan has finished (the dense pass fills every CU;
 * latency-bound work of another stream overlaps best with the sparse stages behind it). */
int spsp_wait_dense(spsp_ctx* waiter, spsp_ctx* scanner);
/* ... only once everything queued on `other` so far has finished. */
int spsp_wait_stream(spsp_ctx* waiter, spsp_ctx* other);

/* total_superkmer_number of Subsampler::print_stat (SubSampler.cpp:430,452,641): how many super-k-mers the scan
 * loop cuts over ALL k-mers of the input, selected or not -- including the cuts its position tracking makes
 * when one m-mer occurs twice in a window (`dump`, :391-398).  Not needed for the sketch: a separate pass. */
int spsp_count_superkmers_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                                 const void* d_rec_off, uint32_t n_rec, uint64_t* total_superkmers);

/* Dense stage only (hash + threshold + hit bitmap), for the roofline
 * measurement: returns the number of m-mers with hash <= threshold. */
int spsp_scan_hits_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                          uint64_t* n_hits);

/* ------------------------------------------------------------- path B ---- */
/* Replaces Comparator::count_intersection / skip_bucket / compute_scores
 * (Comparator.cpp:97-287): inter is n*n, entry [a*n+b] for a<b =
 * sum over buckets |A_a,b ∩ A_b,b| (zero elsewhere); card[i] = nb_kmer_seen_infile[i].
 * Query mode (n_query < n, the sketches of the -q file first): only rows a < n_query are computed --
 * the rows print_jaccard / print_containment emit (Comparator.cpp:374,423); other rows stay zero. */
int spsp_compare(spsp_ctx* ctx, const spsp_sketch_view* sk, uint32_t n, uint32_t n_query,
                 uint32_t* inter, uint64_t* card);

/* Device-resident form over concatenated key arrays (sketch i owns entries
 * [d_sk_off[i], d_sk_off[i+1]) ). Only the rows row_first, row_first + row_stride, ... below n_query
 * are computed (the multi-GPU split of SURVEY.md 8e: every rank holds all sketches after the
 * all-gather and owns a share of the rows -- a block: row_first = its first row, row_stride = 1,
 * n_query = the end of the block; or strided: row_first = rank, row_stride = ranks, n_query = n).
 * A call that owns a part of the rows builds its dictionary from the owned sketches' keys; the other
 * sketches' keys are read once and kept only where an owned sketch may hold them too, and sketches
 * in front of row_first are not read at all (a row counts the sketches behind it). d_inter is a
 * dense n*n uint32 matrix; cells (i, j > i) of owned rows are overwritten, everything else is left
 * untouched. The work is queued on the context's stream: results are complete once that stream has
 * drained. */
int spsp_compare_device(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo,
                        const void* d_kmer_hi /* NULL if k<=32 */, const uint64_t* h_sk_off,
                        uint32_t n, uint32_t n_query /* = n for all-vs-all */, uint32_t row_first,
                        uint32_t row_stride, void* d_inter);

/* ----------------------------------- multi-GPU exchange ---- */
/* The reference is single-process; its all-vs-all merge (Comparator.cpp:97-287) has no sharded form to mirror.
 * Key-partitioned split: equal (minimizer, k-mer) keys hash to the same rank, every rank counts its own hash
 * class for ALL pairs, and inter = the sum of the partial matrices (sparse: spsp_matrix_cells_device).  Each rank sends each of
 * its keys exactly once (all-to-all) -- O(own keys) per rank, against O(all keys) for an all-gather.
 *
 * Sender: scatter this rank's n sketches (concatenated key arrays as in spsp_compare_device) into `parts`
 * fixed-size slots, slot p for rank p, at d_slots + p * spsp_slot_bytes(n, slot_cap, k) (8-byte aligned).
 * Slot layout: u32 magic, u32 n, u32 n_keys, u32 words; u32 cnt[n] (padded to even); slot_cap records of
 * `words` u64 (kmer_lo, [kmer_hi if k > 32], minimizer | local sketch << 32), grouped by sketch, original
 * order kept.  A slot with more than slot_cap keys keeps the first slot_cap and records the true n_keys: the
 * receiver reports SPSP_ERR_OVERFLOW and the caller partitions again with a larger slot_cap. Asynchronous
 * on the context's stream. */
uint64_t spsp_slot_bytes(uint32_t n, uint32_t slot_cap, uint32_t k);
int spsp_partition_keys_device(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo,
                               const void* d_kmer_hi /* NULL if k<=32 */, const uint64_t* h_sk_off, uint32_t n,
                               uint32_t parts, uint32_t slot_cap, void* d_slots);
/* Receiver: d_slots holds `parts` slots, slot s as sent by rank s (same n, slot_cap, k everywhere). Global
 * sketch id = s * n + local id; d_inter is the dense (parts*n)^2 uint32 partial matrix: cells (i, j > i)
 * are overwritten with this rank's share of |K_i ∩ K_j|, everything else is left untouched.  The slot headers
 * (geometry, keys per sketch) are read on the host first -- a malformed slot is SPSP_ERR_FORMAT, one that
 * overflowed at the sender SPSP_ERR_OVERFLOW, before any kernel reads it -- then the records are unpacked into
 * flat key arrays and go through the same partition-form comparison as spsp_compare_device, every row owned. */
int spsp_compare_slots_device(spsp_ctx* ctx, uint32_t k, const void* d_slots, uint32_t parts, uint32_t n,
                              uint32_t slot_cap, void* d_inter);

/* spsp_compare_device / spsp_compare_slots_device split at their host synchronisation (see
 * spsp_scan_device_begin): _begin queues the dictionary build, colour matrix and row sums and returns;
 * spsp_compare_end waits, checks the input / collision flags and, after a fingerprint collision, rebuilds
 * with a new seed.  One comparison may be pending per context; h_sk_off is copied before _begin returns,
 * the device arrays must stay valid until spsp_compare_end returns. */
int spsp_compare_device_begin(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo,
                              const void* d_kmer_hi, const uint64_t* h_sk_off, uint32_t n, uint32_t n_query,
                              uint32_t row_first, uint32_t row_stride, void* d_inter);
int spsp_compare_slots_device_begin(spsp_ctx* ctx, uint32_t k, const void* d_slots, uint32_t parts, uint32_t n,
                                    uint32_t slot_cap, void* d_inter);
int spsp_compare_end(spsp_ctx* ctx);

/* --------------------------------------------- host side of the two CLIs -- */
/* getLineFasta + clean_dna (utils.cpp:675-718) over an already gunzipped
 * buffer: records back to back + n_rec+1 offsets. */
int spsp_fasta_clean_host(const char* text, uint64_t n, uint8_t** bases, uint64_t** rec_off,
                          uint32_t* n_rec);

/* The same two functions on the GPU ("next" row N1): raw gunzipped FASTA text resident in HBM (16-byte
 * aligned) -> cleaned records + offsets in context-owned device buffers, ready for spsp_scan_device. */
int spsp_fasta_clean_device(spsp_ctx* ctx, const void* d_text, uint64_t n_text, void** d_bases, uint64_t* n_bases,
                            void** d_rec_off, uint32_t* n_rec);

/* ... with the cleaned bases written as 2-bit words straight away (utils.cpp:675-718 compacted AND packed in one pass:
 * N1 of SURVEY.md 8f): *d_packed is the layout spsp_pack_bases_device makes -- 16 bases per little-endian dword, first
 * base in bits 31:30, zero tail, 256 readable bytes behind -- in a context-owned buffer, ready for a scan with
 * SPSP_SCAN_PACKED_INPUT; n_bases and the record offsets count BASES as ever.  spsp_sketch_text / spsp_sketch_file(s) use
 * it whenever the dense pass of their parameters reads packed input (the pair-table pass: the default configuration). */
int spsp_fasta_clean_packed_device(spsp_ctx* ctx, const void* d_text, uint64_t n_text, void** d_packed, uint64_t* n_bases,
                                   void** d_rec_off, uint32_t* n_rec);

typedef struct spsp_sketch_stats {
    uint64_t read_kmer, selected_kmer_number, selected_superkmer_number, count_maximal_skmer;
    uint64_t seen_kmers_at_reconstruction, seen_superkmers_at_reconstruction;
    uint64_t seen_max_superkmers_at_reconstruction, actual_minimizer_number, nb_mmer_selected;
    uint64_t total_kmer_number, total_superkmer_number;   /* filled when SPSP_SCAN_STATS is set, else 0 */
} spsp_sketch_stats;

/* handle_superkmer + the emission half of parse_fasta_test
 * (SubSampler.cpp:243-302, 458-504, 512-620; strCompressor utils.cpp:48-68):
 * super-k-mer stream -> uncompressed sketch payload. `rate` is the -s value
 * after stof (SubSampler.cpp:699), printed into the header.  Buckets (minimizers) are
 * built one by one -- they never meet -- and, from 20 000 super-k-mers on, on up to
 * 8 (16 from 400 000 on) host threads of the call's own: same bytes out. */
int spsp_sketch_build_host(const spsp_params* p, double rate, const uint8_t* bases,
                           const uint64_t* rec_off, uint32_t n_rec, const spsp_superkmer* sk,
                           uint64_t n_sk, uint8_t** payload, uint64_t* payload_len,
                           spsp_sketch_stats* stats);

/* FASTA text (host) -> sketch payload with ingest, scan and super-k-mer gather on the GPU: what
 * parse_fasta_test does between openFile and the gzip writer (SubSampler.cpp:306-504). */
int spsp_sketch_text(spsp_ctx* ctx, const spsp_params* p, double rate, const char* text, uint64_t n_text,
                     uint8_t** payload, uint64_t* payload_len, spsp_sketch_stats* stats);

/* Header + bucket reader of the comparator (Comparator.cpp:23-37, 78-92,
 * 186-260; strDecompressor utils.cpp:71-111): payload -> sorted distinct
 * (minimizer, canonical k-mer) keys. Arrays are spsp_free()d one by one. */
int spsp_sketch_parse_host(const uint8_t* payload, uint64_t len, uint32_t* k, uint32_t* m,
                           uint32_t** minimizer, uint64_t** kmer_lo, uint64_t** kmer_hi, uint64_t* n);

/* From a scan straight to the comparator's keys, without the sketch file in between.  Genome g = records
 * [h_first_rec[g], h_first_rec[g + 1]) of ONE scan (d_bases / n_bases / d_rec_off / d_superkmers as given to and returned by
 * spsp_scan_device; SPSP_SCAN_PACKED_INPUT in p->flags when d_bases holds 2-bit words).  The result is what
 * spsp_sketch_parse_host would return for the sketch parse_fasta_test writes for that genome: handle_superkmer's
 * per-k-mer counts with their uint8 wrap and the -a rule (SubSampler.cpp:243-302, 587, 608), the emission / reader round
 * trip (:458-620, Comparator.cpp:186-260) and canonize composed -- the DISTINCT (minimizer, canonical k-mer) keys of every
 * genome, sorted, back to back in device arrays OWNED BY THE CONTEXT (the ones spsp_sketch_decode_device fills: valid
 * until the next decode / keys / spsp_compare call on it; *d_kmer_hi = NULL when k <= 32), sk_off with n_genomes + 1
 * offsets: ready for spsp_compare_device.  A genome of ANY size stays on the device, like the reference's unbounded
 * minimizer_map (SubSampler.h:62): up to 8192 selected k-mer occurrences (4096 with k > 32) a workgroup handles a genome in
 * its LDS; a larger one is flagged by that workgroup and taken by the global-memory stages queued behind it in the same
 * call (one open-addressing table in HBM with the same per-(k-mer, orientation) counts and uint8 rule; then, for the
 * sorted form, a merge sort of the genome's distinct keys where they finally lie, queued from _end).  There is no host
 * path.  _begin queues the work on the context's stream and returns; _end waits for it (an event behind its last
 * kernel).  The caller's device inputs (bases, record offsets, super-k-mers) are read by the work _begin queues and by
 * nothing else: they may be rewritten once that work has run -- spsp_scan_output_wait orders a scan's next write
 * behind it -- without waiting for _end.  One job may be pending per context.
 * flags: SPSP_KEYS_UNORDERED -- every genome's keys DISTINCT but in no particular order: an LDS table per genome instead of
 * the per-genome sort (a tenth of its time; up to 6144 k-mer places and 1024 super-k-mers per genome, 4096 / 512 with
 * k > 32, beyond that the same global-memory table, without the sort).  Such keys are for
 * comparisons on a context that has been told so (spsp_compare_keys_unordered): the comparison itself only needs a
 * sketch to hold a key once; the order is what lets it CHECK that on input it did not make. */
#define SPSP_KEYS_UNORDERED 1u
int spsp_sketch_keys_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases, const void* d_rec_off,
                            const void* d_superkmers, uint64_t n_superkmers, const uint32_t* h_first_rec, uint32_t n_genomes, uint32_t flags,
                            void** d_minimizer, void** d_kmer_lo, void** d_kmer_hi, uint64_t* sk_off);
int spsp_sketch_keys_device_begin(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases, const void* d_rec_off,
                                  const void* d_superkmers, uint64_t n_superkmers, const uint32_t* h_first_rec, uint32_t n_genomes, uint32_t flags);
/* on != 0: the device-form comparisons queued on this context from now on accept sketches whose keys are distinct but
 * unsorted (the caller vouches for "distinct": duplicates inside a sketch would be counted twice) */
int spsp_compare_keys_unordered(spsp_ctx* ctx, int on);
/* A context remembers what its last comparisons looked like (the input came in a good row order, most records had lists,
 * parts spilled, the filter's pass rate) and queues the next one accordingly -- scheduling only, results never depend on it.
 * This forgets all of it: for timing or profiling comparisons of different collections on one context.
 * SPSP_DEBUG_SPILL_TRACE=1 prints the form every comparison took on stderr. */
int spsp_compare_forget(spsp_ctx* ctx);
int spsp_sketch_keys_device_end(spsp_ctx* ctx, void** d_minimizer, void** d_kmer_lo, void** d_kmer_hi, uint64_t* sk_off);
/* how many genomes of the extraction last collected on this context were beyond the per-genome LDS forms and went
 * through the table in HBM (a diagnostic for tests and benchmarks) */
uint32_t spsp_sketch_keys_big_genomes(spsp_ctx* ctx);

/* A scan's output buffer belongs to its context and is rewritten by that context's next scan.  A caller that pipelines --
 * queues scan t + 1 on `scanner` while `reader`'s key extraction of scan t's output (spsp_sketch_keys_device_begin) may
 * still be running on another stream -- calls this in between: the stage of the next scan that writes the output (its
 * last) then starts only behind the reader's latest key extraction; the dense pass is not held up. */
int spsp_scan_output_wait(spsp_ctx* scanner, spsp_ctx* reader);

/* The same decode for MANY sketches at once on the GPU ("next" row N2): payloads[i] = gunzipped sketch i.  The
 * sorted distinct keys of all sketches end up back to back in device arrays OWNED BY THE CONTEXT (valid until the
 * next decode / spsp_compare call on it; *d_kmer_hi = NULL when k <= 32), ready for spsp_compare_device; sk_off gets
 * n + 1 offsets.  A sketch of any size is decoded on the device (beyond 8192 raw keys -- 4096 with k > 32 -- through a
 * table in HBM and a merge sort instead of the per-sketch LDS sort); only a sketch that is not laid out as the sketcher
 * writes it goes through spsp_sketch_parse_host internally: same keys. */
int spsp_sketch_decode_device(spsp_ctx* ctx, const uint8_t* const* payloads, const uint64_t* lens, uint32_t n,
                              uint32_t* k, uint32_t* m, void** d_minimizer, void** d_kmer_lo, void** d_kmer_hi,
                              uint64_t* sk_off);

/* The comparator's N-way merge reads every file's first minimizer into one shared buffer without an end-of-file
 * check (Comparator.cpp:294,316-319).  Call this for the sketches IN FILE ORDER with the same `read_buffer` (m bytes,
 * initialised to 'A'): it performs that read and, for a sketch without any bucket when k == m, returns the one
 * phantom key the reference then counts for it (has_key = 1).  spsp_compare_files does this itself. */
int spsp_sketch_chain_host(const uint8_t* payload, uint64_t len, uint32_t k, uint32_t m, char* read_buffer, int* has_key,
                           uint32_t* minimizer, uint64_t* kmer_lo, uint64_t* kmer_hi);

/* print_jaccard / print_containment (Comparator.cpp:362-460), IEEE division.
 * names: n NUL-terminated strings. */
int spsp_csv_host(int jaccard, const char* const* names, uint32_t n, uint32_t n_query,
                  const uint32_t* inter, const uint64_t* card, int precision, double min_threshold,
                  char** text, uint64_t* len);

/* The same two printers from the SPARSE form of the pair matrix: `cells` = its non-zero entries as packed words
 * i << 48 | j << 32 | count (i < j < n <= 65535, every pair at most once, any order: what spsp_compare_cells_device
 * returns).  A comparison of thousands of sketches has ~10 non-zero partners per row; a row is then written as runs of
 * "0," between them and no n x n matrix is built or scanned.  Same bytes as spsp_csv_host on the dense matrix. */
int spsp_csv_cells_host(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const uint64_t* cells, uint64_t n_cells,
                        const uint64_t* card, int precision, double min_threshold, char** text, uint64_t* len);
/* The same matrix written straight to `gz_path` as the reference writes it (gzip, Comparator.cpp:363,413), without the text
 * ever existing: row blocks become gzip members on the host threads, a run of "0," cells is two literals and a few deflate
 * matches of distance 2, and the member's CRC-32 takes the run in 16 table steps (spsp_compare_files writes its two
 * 10^8-cell matrices this way).  gunzip gives exactly the bytes spsp_csv_cells_host returns. */
int spsp_csv_cells_gz_host(int jaccard, const char* const* names, uint32_t n, uint32_t n_query, const uint64_t* cells, uint64_t n_cells,
                           const uint64_t* card, int precision, double min_threshold, const char* gz_path);

/* sortCSV (sort_csv.cpp:26-111): rows and columns of a symmetric all-vs-all Jaccard CSV (gunzipped text) put into
 * the order of the original file of files.  Inputs the reference mishandles (name missing from the fof or listed
 * twice, short or unparsable rows, diagonal != 1) are rejected with SPSP_ERR_FORMAT. */
int spsp_sort_csv_host(const char* csv, uint64_t csv_len, const char* fof, uint64_t fof_len, char** text, uint64_t* len);

/* zstr-compatible I/O (include/zstr.hpp:136-209 autodetect, :392-407 gzip
 * writer): whole-file read with gzip/zlib/plain autodetection; gzip write. */
int spsp_read_file_host(const char* path, uint8_t** data, uint64_t* len);
int spsp_write_gz_host(const char* path, const uint8_t* data, uint64_t len, int level);

/* Whole-file drivers used by the CLIs (Subsampler::parse_fasta_test
 * SubSampler.cpp:306-510; Comparator::compare_sketches + printers
 * Comparator.cpp:39-74, 362-460). */
int spsp_sketch_file(spsp_ctx* ctx, const spsp_params* p, double rate, const char* fasta_path,
                     const char* out_path, spsp_sketch_stats* stats);
/* The file-of-files loop of the reference's main (`#pragma omp parallel num_threads(c)`, SubSampler.cpp:771-793) inside
 * the library: the n files are taken in list order, a few at a time; `threads` workers read (and gunzip) a batch's files
 * into one pinned buffer, the batch crosses PCIe in one copy and goes through ONE ingest, ONE scan and ONE gather on the
 * GPU (a GPU job per file is a chain of launches and host waits that costs ~0.35 ms however small the file), and the
 * workers then run the sketch builder, gzip and the write per file.  Up to four batches (at most one per worker) are in flight, each on a
 * context (HIP stream) of its own, so reading, the GPU and the builders overlap inside ONE process.  With -a > 1 the
 * k-mer occurrences of the whole batch are counted in one device pass, file by file (the file is part of the key).  A batch
 * with 5 x 10^5 selected k-mer occurrences or more (one metagenome file) has its sketches BUILT on the device as well
 * (handle_superkmer + the emission walk, spsp_build.hip; SPSP_BUILD=device / host pins the choice).  `cb` (may be
 * NULL) is called with phase 0 when file `index` is taken off the queue (inside the queue's lock: the calls come in
 * list order, like the reference's critical(fof) section that prints the name and appends to the output list; with ONE
 * worker right before the file's own phase-1 call, the way the reference's single thread alternates names and reports) and
 * with phase 1 when it is done (rc, its statistics, the error text when rc != 0; one call at a time, like critical(cout)).
 * A file that fails does not stop the others; the call then returns SPSP_ERR_IO.  `times` (may be NULL) receives the
 * stage seconds summed over the workers. */
typedef void (*spsp_file_callback)(void* user, uint32_t index, int phase, int rc, const spsp_sketch_stats* stats, const char* error);
int spsp_sketch_files(int device, const spsp_params* p, double rate, const char* const* fasta_paths, const char* const* out_paths,
                      uint32_t n, uint32_t threads, spsp_file_callback cb, void* user, struct spsp_stage_times* times);
/* The same over several GPUs of one node: the batches are dealt over the devices (slot j of the pipeline lives on
 * devices[j mod n_dev]; up to four batches in flight per device), everything else as above -- sketching shards by genome,
 * there is nothing to exchange.  A device may be named more than once. */
int spsp_sketch_files_multi(const int* devices, uint32_t n_dev, const spsp_params* p, double rate, const char* const* fasta_paths,
                            const char* const* out_paths, uint32_t n, uint32_t threads, spsp_file_callback cb, void* user,
                            struct spsp_stage_times* times);
/* spsp_sketch_files keeps its contexts, device buffers and pinned staging buffers for the next call on the same device
 * (setting them up costs more than sketching a hundred genomes).  This releases the idle ones (device < 0: of every
 * device); optional -- a process that simply exits never needs it. */
void spsp_sketch_files_release(int device);
/* Wall-clock seconds the two whole-file drivers have spent per stage on this context (end-to-end measurement:
 * bench.py's `end_to_end` object).  GPU stages include the host synchronisation that ends them. */
typedef struct spsp_stage_times {
    double read_s;     /* sketch: file read + gunzip (zstr autodetect)                      utils.cpp:357-364 */
    double ingest_s;   /* sketch: H2D copy + getLineFasta/clean_dna on the GPU               utils.cpp:675-718 */
    double scan_s;     /* sketch: the minimizer scan                                         SubSampler.cpp:357-455 */
    double gather_s;   /* sketch: selected super-k-mers' bases back to the host */
    double build_s;    /* sketch: handle_superkmer + emission on the host                    SubSampler.cpp:243-302,458-504 */
    double gzip_s;     /* sketch: gzip level 9 + write                                       SubSampler.cpp:326 */
    double load_s;     /* compare: read + gunzip + decode + sort of all sketches (host threads)  Comparator.cpp:186-260 */
    double compare_s;  /* compare: H2D + all-vs-all on the GPU + D2H                         Comparator.cpp:97-287 */
    double csv_s;      /* compare: both matrices formatted                                   Comparator.cpp:362-460 */
    double csv_gzip_s; /* compare: gzip level 1 + write */
    uint64_t sketch_files, compare_calls;
} spsp_stage_times;
int spsp_stage_times_read(spsp_ctx* ctx, spsp_stage_times* out, int reset);
int spsp_compare_files(spsp_ctx* ctx, const char* const* paths, uint32_t n, uint32_t n_query,
                       int precision, double min_threshold, const char* out_prefix);
/* the same, printing the progress lines of the reference's comparator on stdout where it prints them
 * (Comparator.cpp:56,69,364,414; all-versus-all runs also :503,509) -- for bin/comparator */
int spsp_compare_files_chatty(spsp_ctx* ctx, const char* const* paths, uint32_t n, uint32_t n_query,
                              int precision, double min_threshold, const char* out_prefix, int all_versus_all);

/* The comparator over several GPUs of one node (the reference's compare_sketches, Comparator.cpp:39-74, is
 * one thread over one merge).  One context per entry of `devices` (the same device may be named more than once: contexts
 * then share it), one host thread each.  The split is by KEY: every context decodes a block of the sketch files, deals
 * its keys into one exchange slot per context (spsp_partition_keys_device: equal keys hash to the same slot), fetches
 * the slots of its own hash class from all the others (peer copies over xGMI), compares ALL sketches' keys of that class
 * (spsp_compare_slots_device: 1/n_dev of the dictionary and of the row sums, the same share for every context) and
 * hands the non-zero cells of its partial matrix to the host (spsp_matrix_cells_device), where they add up to the pair
 * matrix the two CSVs are printed from.  Same files out as spsp_compare_files, byte for byte.
 * chatter: 0 silent, 1 / 2 the reference's stdout lines of an all-versus-all / a query run (spsp_compare_files_chatty);
 * times (may be NULL): wall seconds per stage. */
int spsp_compare_files_multi(const int* devices, uint32_t n_dev, const char* const* paths, uint32_t n, uint32_t n_query, int precision,
                             double min_threshold, const char* out_prefix, int chatter, struct spsp_stage_times* times);

/* A pair matrix in sparse form: the non-zero cells (i, j > i) of rows row_first <= i < row_limit of the dense n x n
 * matrix d_inter as packed 64-bit words  i << 48 | j << 32 | count  (n <= 65535, the reference's bound: Comparator.h:26),
 * in no particular order, in d_cells (room for `cap` words).  *n_cells = how many there are; SPSP_ERR_OVERFLOW when
 * that is more than cap (nothing is lost: call again with that much room).  A partial matrix of the key-partitioned
 * split is mostly zeros -- this is what crosses the fabric instead of n x n cells.  Waits for the context's stream. */
int spsp_matrix_cells_device(spsp_ctx* ctx, const void* d_inter, uint32_t n, uint32_t row_first, uint32_t row_limit, void* d_cells,
                             uint64_t cap, uint64_t* n_cells);
/* spsp_compare_device (every row i < n_query, all-vs-all: n_query = n) and spsp_compare_slots_device with the result
 * returned in that sparse form.  Where the comparison's form allows it -- the partition form, all rows, one workgroup
 * per row: every large problem -- the cells leave the row sums directly and the dense matrix is never written (d_scratch,
 * n x n uint32, stays as it was); otherwise d_scratch receives the dense matrix and is sparsified.  SPSP_ERR_OVERFLOW
 * with *n_cells = the room needed when there are more than cap.  Synchronous. */
int spsp_compare_cells_device(spsp_ctx* ctx, uint32_t k, const void* d_minimizer, const void* d_kmer_lo, const void* d_kmer_hi,
                              const uint64_t* h_sk_off, uint32_t n, uint32_t n_query, void* d_scratch, void* d_cells, uint64_t cap,
                              uint64_t* n_cells);
int spsp_compare_slots_cells_device(spsp_ctx* ctx, uint32_t k, const void* d_slots, uint32_t parts, uint32_t n, uint32_t slot_cap,
                                    void* d_scratch, void* d_cells, uint64_t cap, uint64_t* n_cells);
/* d_inter[i][j] += count for every packed cell (the collecting side of the above) */
int spsp_matrix_add_cells_device(spsp_ctx* ctx, void* d_inter, uint32_t n, const void* d_cells, uint64_t n_cells);

#ifdef __cplusplus
}
#endif
#endif /* SPSP_H */
