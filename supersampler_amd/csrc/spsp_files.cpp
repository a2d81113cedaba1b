// spsp_files.cpp -- the whole-file drivers of the sketcher: spsp_sketch_file (one FASTA file -> one sketch file, what
// Subsampler::parse_fasta_test does, SubSampler.cpp:306-510) and spsp_sketch_files (the file-of-files loop of main,
// SubSampler.cpp:771-793, as a pipeline of batches).  Host code that DRIVES the device -- pinned staging buffers, contexts,
// the ingest / scan / gather entry points of the other translation units -- kept apart from spsp_host.cpp, whose pure
// host logic (parsers, builder, printers, zlib I/O) is what tests/tools/host_asan links on its own under the sanitizers.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "spsp_internal.h"

using spsp::set_error;
using spsp::now_s;
static int inflate_all(const uint8_t* in, size_t n, std::vector<uint8_t>& out) { return spsp::inflate_all_host(in, n, out); }

extern "C" {

// A whole file into the context's PINNED staging buffer (grown on demand, reused file after file): a plain FASTA then
// goes to the GPU with one asynchronous copy at the link's rate, instead of malloc + fread + a runtime-staged copy out
// of pageable memory.  Returns 1 when the file is gzip / zlib packed (zstr autodetect, zstr.hpp:154-167): the caller
// inflates it.
static int slurp_pinned(spsp_ctx* ctx, const char* path, uint64_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open '%s'", path); return SPSP_ERR_IO; }
    struct stat st;
    size_t want = (fstat(fileno(f), &st) == 0 && st.st_size > 0) ? (size_t)st.st_size + 1 : (1u << 20);
    size_t got = 0;
    for (;;) {
        if (ctx->h_text_cap < want + 64) {
            size_t cap = std::max<size_t>(want + 64 + want / 4, (size_t)4 << 20);
            uint8_t* nb = nullptr;
            if (hipHostMalloc((void**)&nb, cap, hipHostMallocDefault) != hipSuccess) { fclose(f); set_error("out of pinned host memory"); return SPSP_ERR_NOMEM; }
            if (ctx->h_text) {
                (void)hipStreamSynchronize(ctx->stream);             // a copy out of the old buffer may still be queued
                if (got) memcpy(nb, ctx->h_text, got);
                (void)hipHostFree(ctx->h_text);
            }
            ctx->h_text = nb; ctx->h_text_cap = cap;
        }
        const size_t r = fread(ctx->h_text + got, 1, want - got, f);
        got += r;
        if (r == 0) break;
        if (got == want) want *= 2;                                  // (a file that grew, or a size fstat could not tell)
    }
    fclose(f);
    *n = got;
    const uint8_t* raw = ctx->h_text;
    return (got >= 2 && ((raw[0] == 0x1F && raw[1] == 0x8B) || (raw[0] == 0x78 && (raw[1] == 0x01 || raw[1] == 0x9C || raw[1] == 0xDA)))) ? 1 : 0;
}

int spsp_sketch_file(spsp_ctx* ctx, const spsp_params* p, double rate, const char* fasta_path, const char* out_path,
                     spsp_sketch_stats* stats) {
    if (!ctx || !fasta_path || !out_path) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    uint8_t* text = nullptr; uint64_t tlen = 0;
    bool text_owned = false;
    double t0 = now_s();
    static const bool host_ingest = getenv("SPSP_HOST_INGEST") != nullptr;   // A/B switch: clean on the host, scan on the GPU
    int rc;
    if (host_ingest) { rc = spsp_read_file_host(fasta_path, &text, &tlen); text_owned = true; }
    else {
        SPSP_HIP(hipSetDevice(ctx->device));
        rc = slurp_pinned(ctx, fasta_path, &tlen);
        if (rc == 1) {                                               // packed: inflate out of the pinned copy
            std::vector<uint8_t> plain;
            rc = inflate_all(ctx->h_text, tlen, plain);
            if (!rc) {
                text = (uint8_t*)malloc(plain.size() + 64);
                if (!text) { set_error("out of host memory"); rc = SPSP_ERR_NOMEM; }
                else { if (!plain.empty()) memcpy(text, plain.data(), plain.size()); tlen = plain.size(); text_owned = true; }
            }
        } else if (rc == 0) text = ctx->h_text;
    }
    if (rc) return rc;
    ctx->stages.read_s += now_s() - t0;
    ctx->stages.sketch_files += 1;
    uint8_t* payload = nullptr; uint64_t plen = 0;
    if (host_ingest) {
        uint8_t* bases = nullptr; uint64_t* off = nullptr; uint32_t n_rec = 0;
        rc = spsp_fasta_clean_host((const char*)text, tlen, &bases, &off, &n_rec);
        spsp_superkmer* sk = nullptr; uint64_t n_sk = 0;
        if (!rc) rc = spsp_scan(ctx, p, bases, off, n_rec, &sk, &n_sk);
        if (!rc) rc = spsp_sketch_build_host(p, rate, bases, off, n_rec, sk, n_sk, &payload, &plen, stats);
        free(bases); free(off); free(sk);
    } else {
        // ingest (getLineFasta + clean_dna), scan and super-k-mer gather all run on the GPU
        rc = spsp_sketch_text(ctx, p, rate, (const char*)text, tlen, &payload, &plen, stats);
    }
    if (text_owned) free(text);
    if (rc) { free(payload); return rc; }
    t0 = now_s();
    rc = spsp_write_gz_host(out_path, payload, plen, 9);  // level 9: SubSampler.cpp:326
    ctx->stages.gzip_s += now_s() - t0;
    free(payload);
    return rc;
}

// File-of-files loop, one context per worker and one GPU job per file (spsp_sketch_file): the form used with -a > 1 (the
// abundance pass counts k-mers per file) and as the A/B partner of the batched pipeline below (SPSP_FILES_PER_WORKER=1).
static int sketch_files_per_worker(const std::vector<int>& devices, const spsp_params* p, double rate, const char* const* fasta_paths, const char* const* out_paths,
                      uint32_t n, uint32_t threads, spsp_file_callback cb, void* user, spsp_stage_times* times) {
    if (!p || (n && (!fasta_paths || !out_paths))) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    int rc0 = spsp::check_params(p);
    if (rc0) return rc0;
    if (threads == 0) threads = 1;
    if (threads > n) threads = n ? n : 1;
    std::atomic<uint32_t> next(0);
    std::mutex queue_m, done_m;
    std::vector<spsp_stage_times> per(threads);
    std::vector<int> worker_rc(threads, SPSP_OK);
    std::vector<std::string> worker_err(threads);
    std::atomic<int> failed_files(0);
    auto work = [&](uint32_t w) {
        spsp_ctx* ctx = nullptr;
        memset(&per[w], 0, sizeof per[w]);
        if ((worker_rc[w] = spsp_create(devices[w % devices.size()], nullptr, &ctx))) { worker_err[w] = spsp_last_error(); return; }   // workers dealt over the devices
        for (;;) {
            uint32_t i;
            {   // named critical section `fof` of the reference (:776-786): dequeue + the caller's "started" line, in list order
                std::lock_guard<std::mutex> g(queue_m);
                i = next.load();
                if (i >= n) break;
                next.store(i + 1);
                if (cb) cb(user, i, 0, SPSP_OK, nullptr, nullptr);
            }
            spsp_sketch_stats st;
            memset(&st, 0, sizeof st);
            const int rc = spsp_sketch_file(ctx, p, rate, fasta_paths[i], out_paths[i], &st);
            if (rc) failed_files.fetch_add(1);
            if (cb) {   // critical section `cout` (:791): one file's report at a time
                std::lock_guard<std::mutex> g(done_m);
                cb(user, i, 1, rc, &st, rc ? spsp_last_error() : nullptr);
            }
        }
        per[w] = ctx->stages;
        spsp_destroy(ctx);
    };
    std::vector<std::thread> pool;
    for (uint32_t w = 1; w < threads; ++w) pool.emplace_back(work, w);
    work(0);
    for (auto& th : pool) th.join();
    if (times) {
        memset(times, 0, sizeof *times);
        for (const auto& s : per) {
            times->read_s += s.read_s; times->ingest_s += s.ingest_s; times->scan_s += s.scan_s; times->gather_s += s.gather_s;
            times->build_s += s.build_s; times->gzip_s += s.gzip_s; times->sketch_files += s.sketch_files;
        }
    }
    for (uint32_t w = 0; w < threads; ++w)
        if (worker_rc[w]) { set_error("%s", worker_err[w].c_str()); return worker_rc[w]; }
    if (failed_files.load()) { set_error("%d of %u files could not be sketched (see the callback's reports)", failed_files.load(), n); return SPSP_ERR_IO; }
    return SPSP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's `#pragma omp parallel num_threads(c)` block over the file of files (SubSampler.cpp:771-793) as a
// pipeline of BATCHES.  One GPU job per file is a chain of a dozen small launches, five copies and four host waits:
// ~0.35 ms per 5 Mbp genome whatever the number of streams (measured: 100 files, 1 / 8 / 16 workers with a context each:
// the GPU-side stages summed to 35 ms every time) -- the chain, not the kernels, is what a file costs.  So several files
// travel together: their texts are laid out in ONE pinned slab (every file on a 4 KiB boundary = a tile of the ingest
// kernels, first line forced to be a header line, the gap filled with newlines), cross PCIe in ONE copy and go through
// ONE ingest, ONE scan and ONE gather; the super-k-mer stream is cut at the files' record ranges and the sketch builder,
// gzip and the write run per file on the worker threads.  Batches overlap each other: while one is on the GPU, the
// workers read the next and build the sketches of the one before.
namespace {

struct PipeFile {
    uint32_t index = 0;
    int rc = SPSP_OK;
    std::string err;
    bool packed = false, force_header = false;
    std::vector<uint8_t> inflated;       // gunzipped text of a packed file
    uint64_t text_len = 0, off = 0;      // place in the slab (off is a multiple of kSlabAlign)
    uint32_t first_rec = 0, n_rec = 0;
    uint64_t sk0 = 0, sk1 = 0;           // its super-k-mers in the batch's stream
    uint64_t occ0 = 0;                   // -a > 1: its first k-mer occurrence in the batch's numbering
    std::string body;                    // device builder: everything behind the header line
    bool built = false;
    uint64_t dev_stats[4] = {0, 0, 0, 0};
    uint64_t total_superkmers = 0;       // SPSP_SCAN_STATS
    bool done = false;                   // sketched and written already (one-job-per-file fallback of an over-large batch)
    spsp_sketch_stats st{};
};

constexpr uint64_t kSlabAlign = 4096;    // = the ingest kernels' tile (spsp_ingest.hip kCleanTile): a file starts a tile

struct PipeSlot {
    spsp_ctx* ctx = nullptr;
    uint8_t* slab = nullptr;             // pinned
    size_t slab_cap = 0;
    std::vector<PipeFile> files;
    std::atomic<int> left{0};            // tasks of the current stage still running
    uint64_t total = 0;                  // bytes of the slab in use
    std::vector<uint64_t> rec_off;
    std::vector<spsp_superkmer> sk;
    uint8_t* compact = nullptr;
    uint32_t* coff = nullptr;
    uint8_t* kflags = nullptr;           // -a > 1: the device abundance pass's verdict per k-mer occurrence of the batch (file by file)
    uint8_t* d_text = nullptr;           // the slab's place on the device: every fill task copies what it has read (round 5)
    std::atomic<bool> copy_failed{false};
    bool busy = false;
    int device = 0;              // the device its context lives on
};

class FilePipeline {
public:
    FilePipeline(const std::vector<int>& devices, const spsp_params* p, double rate, const char* const* in, const char* const* out, uint32_t n, uint32_t threads,
                 spsp_file_callback cb, void* user)
        : devices_(devices), p_(*p), rate_(rate), in_(in), out_(out), n_(n), threads_(threads), cb_(cb), user_(user) {}

    int run(spsp_stage_times* times) {
        // sizes decide how many files travel together: batches of ~1/12 of the job, between 8 and 32 MB of text (pinning
        // the slabs costs ~0.2 ms per MB, once per process: the slots are kept for the next call)
        sizes_.assign(n_, 0);
        uint64_t total = 0;
        for (uint32_t i = 0; i < n_; ++i) {
            struct stat st;
            if (stat(in_[i], &st) == 0 && st.st_size > 0) sizes_[i] = (uint64_t)st.st_size;
            total += sizes_[i];
        }
        budget_ = std::max<uint64_t>(8ull << 20, std::min<uint64_t>(32ull << 20, total / 12));
        static const char* dbg_budget = getenv("SPSP_DEBUG_PIPE_BUDGET_MB");   // tuning knob
        if (dbg_budget) budget_ = (uint64_t)std::max(1, atoi(dbg_budget)) << 20;
        const double t_setup0 = now_s();
        t_begin_ = t_setup0;
        // batches in flight: with one worker nothing overlaps the host's work anyway (and every slot costs a context, a
        // pinned slab and device buffers, which a short-lived process pays for in full)
        // (several devices: the batches are dealt over them, slot j on device j mod n -- up to six in flight per device)
        static const char* dbg_slots = getenv("SPSP_DEBUG_PIPE_SLOTS");         // tuning knob: batches in flight per device
        // (six: 100 x 5 Mbp on 16 threads, median of 7 calls 17.4 ms with four, 13.2 with six, 15.4 with eight -- a slot is a batch being read,
        // one on the GPU or one being built; with four the workers ran out of reads while batches sat in their GPU stage)
        // ... for a long job.  A slot is also a context, a pinned slab and a set of device buffers that a short-lived process pays
        // for in full: `sub_sampler` on 100 files, a fresh process each, 0.28-0.29 s with four against 0.30-0.34 with six -- so six from
        // 32 batches on)
        const uint32_t per_dev = dbg_slots ? (uint32_t)std::max(1, atoi(dbg_slots)) : (total / budget_ >= 32 ? 6u : 4u);
        const uint32_t n_slots = std::min<uint32_t>(std::min<uint32_t>(per_dev * (uint32_t)devices_.size(), threads_), n_);
        slots_.resize(n_slots);
        for (size_t j = 0; j < slots_.size(); ++j) {
            auto& s = slots_[j];
            const int dev = devices_[j % devices_.size()];
            s = take_slot(dev);
            s->device = dev;
            if (!s->ctx) {
                const int rc = spsp_create(dev, nullptr, &s->ctx);
                if (rc) { fatal_rc_ = rc; fatal_err_ = spsp_last_error(); break; }
            }
            s->ctx->stages = spsp_stage_times{};
        }
        setup_s_ += now_s() - t_setup0;
        if (!fatal_rc_) {
            {
                std::lock_guard<std::mutex> g(m_);
                for (auto& s : slots_) form_batch(*s);
            }
            std::vector<std::thread> pool;
            for (uint32_t w = 1; w < threads_; ++w) pool.emplace_back([this]() { work(); });
            work();
            for (auto& th : pool) th.join();
        }
        if (times) memset(times, 0, sizeof *times);
        for (auto& s : slots_) {
            if (!s) continue;                                // (context creation failed before this slot was reached)
            if (s->ctx && times) {
                const spsp_stage_times& t = s->ctx->stages;
                times->ingest_s += t.ingest_s; times->scan_s += t.scan_s; times->gather_s += t.gather_s;
            }
            free(s->compact); free(s->coff); free(s->kflags); s->compact = nullptr; s->coff = nullptr; s->kflags = nullptr;
            std::vector<PipeFile>().swap(s->files);
            const int dev = s->device;
            give_slot(dev, std::move(s));
        }
        if (times) { times->read_s = read_s_; times->build_s = build_s_; times->gzip_s = gzip_s_; times->sketch_files = done_files_; }
        if (getenv("SPSP_DEBUG_PIPE_TIMES"))
            fprintf(stderr, "[spsp pipeline] contexts %.1f ms, pinned slabs %.1f ms, teardown follows; budget %llu MB, %u slots\n", setup_s_ * 1e3, slab_s_ * 1e3,
                    (unsigned long long)(budget_ >> 20), (unsigned)slots_.size());
        if (trace_on_) {
            std::sort(trace_.begin(), trace_.end(), [](const TraceEv& a, const TraceEv& b) { return a.t0 < b.t0; });
            for (const TraceEv& e : trace_) fprintf(stderr, "[pipe] slot %d %c %8.3f .. %8.3f ms (%.3f)\n", e.slot, e.stage, e.t0, e.t1, e.t1 - e.t0);
            fprintf(stderr, "[pipe] call %.3f ms\n", (now_s() - t_begin_) * 1e3);
        }
        if (fatal_rc_) { set_error("%s", fatal_err_.c_str()); return fatal_rc_; }
        if (failed_) { set_error("%u of %u files could not be sketched (see the callback's reports)", failed_, n_); return SPSP_ERR_IO; }
        return SPSP_OK;
    }

private:
    // Slots (a context, its device buffers and tables, a pinned slab) outlive the call: the next call on this device takes
    // them over instead of paying for contexts, device allocations and ~0.2 ms per MB of page pinning again.  They are
    // never destroyed (no HIP call may run from a static destructor after the runtime has shut down); the process's end
    // releases them.
    struct SlotPool { std::mutex m; std::vector<std::pair<int, std::unique_ptr<PipeSlot>>> idle; };
    static SlotPool& pool() { static SlotPool* p = new SlotPool(); return *p; }
    static std::unique_ptr<PipeSlot> take_slot(int device) {
        SlotPool& P = pool();
        std::lock_guard<std::mutex> g(P.m);
        for (size_t i = 0; i < P.idle.size(); ++i)
            if (P.idle[i].first == device) { std::unique_ptr<PipeSlot> s = std::move(P.idle[i].second); P.idle.erase(P.idle.begin() + (ptrdiff_t)i); return s; }
        return std::unique_ptr<PipeSlot>(new PipeSlot());
    }
    static void give_slot(int device, std::unique_ptr<PipeSlot> s) {
        if (!s || !s->ctx) return;                           // (a slot whose context could not be created holds nothing)
        SlotPool& P = pool();
        std::lock_guard<std::mutex> g(P.m);
        P.idle.emplace_back(device, std::move(s));
    }

public:
    static void release_idle(int device) {
        SlotPool& P = pool();
        std::lock_guard<std::mutex> g(P.m);
        for (size_t i = 0; i < P.idle.size();) {
            if (device >= 0 && P.idle[i].first != device) { ++i; continue; }
            PipeSlot& s = *P.idle[i].second;
            (void)hipSetDevice(P.idle[i].first);
            if (s.slab) { (void)hipStreamSynchronize(s.ctx->stream); (void)hipHostFree(s.slab); }
            spsp_destroy(s.ctx);
            P.idle.erase(P.idle.begin() + (ptrdiff_t)i);
        }
    }

private:
    // ---- scheduling: one queue of tasks, `threads_` workers; a stage's last task queues the next stage
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(f)); } cv_.notify_one(); }
    void work() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this]() { return !q_.empty() || finished_(); });
                if (q_.empty()) { cv_.notify_all(); return; }
                f = std::move(q_.front());
                q_.pop_front();
                ++running_;
            }
            f();
            {
                std::lock_guard<std::mutex> g(m_);
                --running_;
            }
            cv_.notify_all();
        }
    }
    bool finished_() const { return q_.empty() && running_ == 0 && batches_in_flight_ == 0; }

    // m_ held.  The next files in list order (the "started" reports come in that order, like the reference's critical(fof))
    void form_batch(PipeSlot& s) {
        s.files.clear();
        s.busy = false;
        if (next_ >= n_) return;
        uint64_t bytes = 0;
        while (next_ < n_ && s.files.size() < 64) {
            const uint64_t sz = sizes_[next_] + 2 * kSlabAlign;
            if (!s.files.empty() && bytes + sz > budget_) break;
            PipeFile f;
            f.index = next_;
            s.files.push_back(std::move(f));
            bytes += sz;
            if (cb_ && threads_ > 1) cb_(user_, next_, 0, SPSP_OK, nullptr, nullptr);
            ++next_;
        }
        s.busy = true;
        ++batches_in_flight_;
        s.left.store((int)s.files.size());
        for (size_t j = 0; j < s.files.size(); ++j) q_.push_back([this, &s, j]() { prepare(s, j); });
        cv_.notify_all();
    }

    // stage 1, per file: open, tell packed from plain (zstr autodetect, zstr.hpp:154-167), learn the text's length and
    // first byte; a packed file is inflated here
    void prepare(PipeSlot& s, size_t j) {
        PipeFile& f = s.files[j];
        const double t0 = now_s();
        const int fd = open(in_[f.index], O_RDONLY);
        if (fd < 0) { f.rc = SPSP_ERR_IO; f.err = std::string("cannot open '") + in_[f.index] + "'"; }
        else {
            struct stat st;
            uint8_t head[2] = {0, 0};
            const ssize_t got = pread(fd, head, 2, 0);
            const bool sized = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
            f.packed = got == 2 && ((head[0] == 0x1F && head[1] == 0x8B) || (head[0] == 0x78 && (head[1] == 0x01 || head[1] == 0x9C || head[1] == 0xDA)));
            if (f.packed || !sized) {                       // inflate (or a pipe / device: slurp) into memory now
                std::vector<uint8_t> raw;
                uint8_t buf[1 << 16];
                ssize_t r;
                while ((r = read(fd, buf, sizeof buf)) > 0) raw.insert(raw.end(), buf, buf + r);
                if (f.packed) { f.rc = inflate_all(raw.data(), raw.size(), f.inflated); if (f.rc) f.err = spsp_last_error(); }
                else f.inflated.swap(raw);
                f.packed = true;                            // "text is in f.inflated"
                f.text_len = f.inflated.size();
                f.force_header = f.text_len == 0 || !(f.inflated[0] == '>' || f.inflated[0] == 0xFF);
            } else {
                f.text_len = (uint64_t)st.st_size;
                f.force_header = got < 1 || !(head[0] == '>' || head[0] == 0xFF);
            }
            close(fd);
        }
        add_time(read_s_, now_s() - t0);
        trace(s, 'p', t0);
        if (s.left.fetch_sub(1) == 1) layout(s);
    }

    // between stages 1 and 2 (one thread): every good file gets a tile-aligned place in the slab
    void layout(PipeSlot& s) {
        const double t_lay = now_s();
        uint64_t at = 0;
        for (auto& f : s.files) {
            if (f.rc) continue;
            f.off = at;
            at = (at + (f.force_header ? 1 : 0) + f.text_len + 1 + kSlabAlign - 1) / kSlabAlign * kSlabAlign;   // >= 1 newline behind every file
        }
        s.total = at;
        if (s.slab_cap < at + 64) {
            (void)hipSetDevice(s.device);
            if (s.slab) { (void)hipStreamSynchronize(s.ctx->stream); (void)hipHostFree(s.slab); s.slab = nullptr; s.slab_cap = 0; }
            const size_t cap = (size_t)std::max<uint64_t>(at + at / 8 + 64, budget_ + budget_ / 4);
            const double t_slab = now_s();
            const hipError_t he = hipHostMalloc((void**)&s.slab, cap, hipHostMallocDefault);
            add_time(slab_s_, now_s() - t_slab);
            if (he != hipSuccess) {
                for (auto& f : s.files) if (!f.rc) { f.rc = SPSP_ERR_NOMEM; f.err = "out of pinned host memory"; }
                s.total = 0;
            } else s.slab_cap = cap;
        }
        // the text's place on the device: a fill task copies its file (a large file: its slice) as soon as it has read it, so the
        // PCIe copy runs beside the reads of the other tasks instead of behind all of them (one 4 GB FASTA file: 0.078 s of reading
        // and 0.076 s of copying, one after the other)
        s.d_text = nullptr;
        static const bool copy_late = getenv("SPSP_DEBUG_COPY_LATE") != nullptr;     // A/B: one copy per batch, in the GPU stage
        if (s.total && !copy_late) {
            (void)hipSetDevice(s.device);
            if (s.ctx->i_text.reserve((size_t)s.total + 64) == SPSP_OK) s.d_text = s.ctx->i_text.as<uint8_t>();
        }
        s.left.store((int)s.files.size());
        trace(s, 'l', t_lay);
        for (size_t j = 0; j < s.files.size(); ++j) push([this, &s, j]() { fill(s, j); });
    }
    // bytes [from, to) of the slab to their place on the device (queued on the slot's stream; the GPU stage's kernels follow them)
    static void copy_up(PipeSlot& s, uint64_t from, uint64_t to) {
        if (!s.d_text || to <= from) return;
        (void)hipSetDevice(s.device);
        if (hipMemcpyAsync(s.d_text + from, s.slab + from, (size_t)(to - from), hipMemcpyHostToDevice, s.ctx->stream) != hipSuccess) { (void)hipGetLastError(); s.copy_failed.store(true); }
    }

    // stage 2, per file: the text into its place (a plain file is read straight into the pinned slab)
    void fill(PipeSlot& s, size_t j) {
        PipeFile& f = s.files[j];
        const double t0 = now_s();
        if (!f.rc) {
            uint8_t* dst = s.slab + f.off;
            std::vector<uint8_t> sliced_up;                  // a large file's slices that are on their way to the device already
            if (f.force_header) *dst++ = '>';
            if (f.packed) { if (f.text_len) memcpy(dst, f.inflated.data(), f.text_len); std::vector<uint8_t>().swap(f.inflated); }
            else {
                const int fd = open(in_[f.index], O_RDONLY);
                uint64_t got = 0;
                if (fd >= 0) {
                    // a large file (a whole eukaryote genome) is read in slices on a few threads of its own: one thread copies
                    // the page cache at ~20 GB/s, which was a fifth of the whole call for a 4 GB file
                    const uint64_t slice = 256ull << 20;
                    const unsigned parts = (unsigned)std::min<uint64_t>(8, f.text_len / slice);
                    auto take = [&](uint64_t from, uint64_t to) -> uint64_t {
                        uint64_t at = from;
                        ssize_t r;
                        while (at < to && (r = pread(fd, dst + at, to - at, (off_t)at)) > 0) at += (uint64_t)r;
                        return at - from;
                    };
                    if (parts >= 2) {
                        std::vector<uint64_t> part_got(parts, 0);
                        sliced_up.assign(parts, 0);
                        std::vector<std::thread> pool;
                        const uint64_t per = (f.text_len + parts - 1) / parts;
                        const uint64_t base = (uint64_t)(dst - s.slab);      // where the text starts in the slab
                        auto slice_job = [&](unsigned t) {
                            const uint64_t a = per * t, z = std::min<uint64_t>(f.text_len, per * (t + 1));
                            part_got[t] = take(a, z);
                            if (part_got[t] == z - a) { copy_up(s, base + a, base + z); sliced_up[t] = 1; }
                        };
                        for (unsigned t = 1; t < parts; ++t) pool.emplace_back(slice_job, t);
                        slice_job(0);
                        for (auto& th : pool) th.join();
                        for (uint64_t g : part_got) got += g;
                    } else got = take(0, f.text_len);
                    close(fd);
                }
                if (got != f.text_len) {
                    // the file contributes NOTHING: its whole region -- the forced '>' and what was read included -- becomes empty
                    // lines, so that no header line of it starts a record that the file in front would be given (gpu() derives a
                    // file's records from the next good file's first record)
                    f.rc = SPSP_ERR_IO; f.err = std::string("short read of '") + in_[f.index] + "' (the file changed while it was read)";
                    memset(s.slab + f.off, '\n', (size_t)(f.force_header ? 1 : 0) + f.text_len);
                }
            }
            // the gap up to the next file's tile: newlines (empty lines: no bases, no record)
            const uint64_t end = f.off + (f.force_header ? 1 : 0) + f.text_len;
            const uint64_t next = (end + 1 + kSlabAlign - 1) / kSlabAlign * kSlabAlign;
            memset(s.slab + end, '\n', next - end);
            // the file's region to the device: everything, or what its slices have not taken up yet (a file that failed is
            // all newlines by now and goes up whole)
            if (f.rc || sliced_up.empty()) copy_up(s, f.off, next);
            else {
                const uint64_t text0 = f.off + (f.force_header ? 1 : 0), per = (f.text_len + sliced_up.size() - 1) / sliced_up.size();
                copy_up(s, f.off, text0);
                for (size_t t = 0; t < sliced_up.size(); ++t)
                    if (!sliced_up[t]) copy_up(s, text0 + per * t, text0 + std::min<uint64_t>(f.text_len, per * (t + 1)));
                copy_up(s, end, next);
            }
        }
        add_time(read_s_, now_s() - t0);
        trace(s, 'f', t0);
        if (s.left.fetch_sub(1) == 1) push([this, &s]() { gpu(s); });
    }

    // stage 3, per batch: one copy, one ingest, one scan, one gather
    void gpu(PipeSlot& s) {
        const double t_gpu = now_s();
        int rc = SPSP_OK;
        spsp_ctx* ctx = s.ctx;
        free(s.compact); free(s.coff); free(s.kflags); s.compact = nullptr; s.coff = nullptr; s.kflags = nullptr;
        s.sk.clear(); s.rec_off.clear();
        auto run = [&]() -> int {
            if (s.total == 0) return SPSP_OK;
            SPSP_HIP(hipSetDevice(s.device));
            double t0 = now_s(), t1;
            int r;
            if ((r = ctx->i_text.reserve((size_t)s.total + 64))) return r;
            if (!s.d_text || s.d_text != ctx->i_text.as<uint8_t>() || s.copy_failed.exchange(false))                // (else: the fill tasks queued the copies)
                SPSP_HIP(hipMemcpyAsync(ctx->i_text.p, s.slab, (size_t)s.total, hipMemcpyHostToDevice, ctx->stream));
            uint8_t* d_bases = nullptr; uint64_t* d_off = nullptr; uint64_t n_bases = 0; uint32_t n_rec = 0;
            const bool packed = spsp::ingest_packs(&p_);           // the ingest writes the 2-bit words the dense pass reads
            if ((r = spsp::clean_device_impl(ctx, ctx->i_text.as<uint8_t>(), s.total, &d_bases, &n_bases, &d_off, &n_rec, packed))) return r;
            // records in front of every file: the ingest's per-tile record base at the file's first tile, less the file's own
            // first record (counted with the newline in front of its header line, i.e. in the tile before)
            const uint64_t n_tiles = (s.total + kSlabAlign - 1) / kSlabAlign;
            std::vector<uint32_t> rec_base((size_t)n_tiles);
            SPSP_HIP(hipMemcpyAsync(rec_base.data(), ctx->i_recbase.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, ctx->stream));
            s.rec_off.resize((size_t)n_rec + 1);
            SPSP_HIP(hipMemcpyAsync(s.rec_off.data(), d_off, s.rec_off.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            t1 = now_s(); ctx->stages.ingest_s += t1 - t0; trace(s, 'i', t0); t0 = t1;
            spsp_superkmer* d_sk = nullptr; uint64_t n_sk = 0;
            spsp_params ps = p_;
            if (packed) ps.flags |= SPSP_SCAN_PACKED_INPUT;
            if ((r = spsp::scan_device_impl(ctx, &ps, d_bases, n_bases, d_off, n_rec, &d_sk, &n_sk))) return r;   // (its host wait also completes the copies above)
            t1 = now_s(); ctx->stages.scan_s += t1 - t0; trace(s, 's', t0); t0 = t1;
            PipeFile* prev = nullptr;
            for (auto& f : s.files) {
                if (f.rc) continue;
                f.first_rec = rec_base[(size_t)(f.off / kSlabAlign)] - 1;
                if (prev) prev->n_rec = f.first_rec - prev->first_rec;
                prev = &f;
            }
            if (prev) prev->n_rec = n_rec - prev->first_rec;
            s.sk.resize((size_t)n_sk);
            if (n_sk) SPSP_HIP(hipMemcpyAsync(s.sk.data(), d_sk, (size_t)n_sk * sizeof(spsp_superkmer), hipMemcpyDeviceToHost, ctx->stream));
            SPSP_HIP(hipStreamSynchronize(ctx->stream));                      // (the stream's copy above: the files' ranges and the choice below come from it)
            uint64_t places = 0;
            for (const spsp_superkmer& e : s.sk) places += e.len >= p_.k ? e.len - p_.k + 1 : 0;
            const bool dev_build = spsp::build_on_device(places) && n_sk <= 0xfffffff0ull;
            bool built = false;
            if (!dev_build && (r = spsp::gather_superkmers_impl(ctx, d_bases, d_off, d_sk, n_sk, &s.compact, &s.coff, packed))) return r;   // synchronises the stream
            // the stream is in genome order: a file's super-k-mers are those of its records
            size_t at = 0;
            for (auto& f : s.files) {
                if (f.rc) continue;
                while (at < s.sk.size() && s.sk[at].rec < f.first_rec) ++at;
                f.sk0 = at;
                while (at < s.sk.size() && s.sk[at].rec < f.first_rec + f.n_rec) ++at;
                f.sk1 = at;
            }
            t1 = now_s(); ctx->stages.gather_s += t1 - t0; trace(s, 'G', t0); t0 = t1;
            if (dev_build) {
                // the sketch builder on the device, for the whole batch (spsp_build.hip): file j's super-k-mers are [sk0, sk1)
                std::vector<uint32_t> fsk;
                std::vector<size_t> who;
                for (size_t j = 0; j < s.files.size(); ++j) {
                    if (s.files[j].rc) continue;
                    fsk.push_back((uint32_t)s.files[j].sk0);
                    who.push_back(j);
                }
                fsk.push_back((uint32_t)n_sk);
                if (!who.empty()) fsk[0] = 0;
                std::vector<std::string> bodies;
                std::vector<uint64_t> fst;
                r = who.empty() ? SPSP_OK : spsp::sketch_build_device_impl(ctx, &p_, d_bases, packed, d_off, d_sk, n_sk, fsk.data(), (uint32_t)who.size(), &bodies, &fst);
                if (r == SPSP_OK) {
                    for (size_t x = 0; x < who.size(); ++x) {
                        PipeFile& f = s.files[who[x]];
                        f.body.swap(bodies[x]); f.built = true;
                        for (int q = 0; q < 4; ++q) f.dev_stats[q] = fst[4 * x + q];
                    }
                    built = true;
                } else if (r != SPSP_ERR_OVERFLOW) return r;
                else if ((r = spsp::gather_superkmers_impl(ctx, d_bases, d_off, d_sk, n_sk, &s.compact, &s.coff, packed))) return r;   // too many places: the host builder
                t1 = now_s(); add_time(build_s_, t1 - t0); t0 = t1;
            }
            if (!built && p_.abundance > 1 && n_sk) {
                // -a > 1: every k-mer occurrence of the batch counted in ONE device pass, file by file (the reference's index is per
                // file: one Subsampler per file, SubSampler.cpp:787) -- a GPU job per file was ~0.35 ms of launches and waits each
                std::vector<uint32_t> seg;
                uint64_t occ = 0;
                size_t q = 0;
                for (auto& f : s.files) {
                    if (f.rc) continue;
                    for (; q < f.sk0; ++q) occ += s.sk[q].len >= p_.k ? s.sk[q].len - p_.k + 1 : 0;
                    f.occ0 = occ;
                    seg.push_back(seg.empty() ? 0u : (uint32_t)f.sk0);
                }
                seg.push_back((uint32_t)n_sk);
                uint64_t n_occ = 0;
                if ((r = spsp::abundance_flags_impl(ctx, &p_, d_sk, n_sk, &s.kflags, &n_occ, seg.data(), (uint32_t)seg.size() - 1))) return r;
                t1 = now_s(); ctx->stages.scan_s += t1 - t0; t0 = t1;
            }
            if (p_.flags & SPSP_SCAN_STATS) {
                // print_stat's count of ALL super-k-mers (SubSampler.cpp:429-430,451-452) is a per-file figure.  One counting pass
                // over the batch's records, a total per file (a launch per 5 Mbp file filled a third of the chip: 7 ms each, the
                // largest part of a batch of six in a `sub_sampler` process); the count of a record does not depend on what lies
                // in front of it (spsp_stats.hip: every record starts from a fresh rescan)
                std::vector<uint32_t> frec;
                std::vector<size_t> who;
                for (size_t j = 0; j < s.files.size(); ++j) {
                    if (s.files[j].rc || s.files[j].n_rec == 0) continue;
                    frec.push_back(s.files[j].first_rec);
                    who.push_back(j);
                }
                if (!who.empty()) {
                    frec[0] = 0;                                        // (records in front of the first good file: there are none)
                    std::vector<uint64_t> tot(who.size(), 0);
                    if ((r = spsp::count_superkmers_impl(ctx, &p_, d_bases, n_bases, d_off, n_rec, tot.data(), packed, 0, frec.data(), (uint32_t)frec.size()))) return r;
                    for (size_t x = 0; x < who.size(); ++x) s.files[who[x]].total_superkmers = tot[x];
                }
                ctx->stages.scan_s += now_s() - t0;
                trace(s, 'S', t0);
            }
            return SPSP_OK;
        };
        rc = run();
        if (rc == SPSP_ERR_OVERFLOW) {
            // together the files exceed a limit of ONE job (32-bit offsets of the gathered super-k-mers at -s near 1, say):
            // one job per file on this slot's context instead, the way spsp_sketch_file would have run them
            for (auto& f : s.files) {
                if (f.rc) continue;
                f.rc = spsp_sketch_file(ctx, &p_, rate_, in_[f.index], out_[f.index], &f.st);
                if (f.rc) f.err = spsp_last_error();
                f.done = true;
            }
            rc = SPSP_OK;
        }
        if (rc) { const std::string e = spsp_last_error(); for (auto& f : s.files) if (!f.rc) { f.rc = rc; f.err = e; } }
        s.left.store((int)s.files.size());
        trace(s, 'g', t_gpu);
        for (size_t j = 0; j < s.files.size(); ++j) push([this, &s, j]() { finish(s, j); });
    }

    // stage 4, per file: handle_superkmer + emission (SubSampler.cpp:243-302, 458-504), gzip -9, write, report
    void finish(PipeSlot& s, size_t j) {
        PipeFile& f = s.files[j];
        const double t_fin = now_s();
        uint8_t* payload = nullptr; uint64_t plen = 0;
        if (!f.rc && !f.done) {
            double t0 = now_s();
            std::vector<spsp_superkmer> mine(s.sk.begin() + (ptrdiff_t)f.sk0, s.sk.begin() + (ptrdiff_t)f.sk1);
            for (auto& e : mine) e.rec -= f.first_rec;
            static const uint64_t no_rec[1] = {0};
            if (f.built) {
                // built on the device: the counters of the stream and the header line are the host's
                f.rc = spsp::sketch_stream_stats(&p_, f.n_rec ? s.rec_off.data() + f.first_rec : no_rec, f.n_rec, mine.data(), mine.size(), &f.st);
                if (!f.rc) {
                    f.st.actual_minimizer_number = f.dev_stats[0]; f.st.seen_kmers_at_reconstruction = f.dev_stats[1];
                    f.st.seen_superkmers_at_reconstruction = f.dev_stats[2]; f.st.seen_max_superkmers_at_reconstruction = f.dev_stats[3];
                    std::string head;
                    spsp::sketch_header_line(p_.k, p_.m, f.st.selected_kmer_number, rate_, head);
                    plen = head.size() + f.body.size();
                    payload = (uint8_t*)malloc((size_t)plen + 1);
                    if (!payload) { spsp::set_error("out of host memory"); f.rc = SPSP_ERR_NOMEM; }
                    else { memcpy(payload, head.data(), head.size()); memcpy(payload + head.size(), f.body.data(), f.body.size()); }
                    std::string().swap(f.body);
                }
            } else
            f.rc = spsp::sketch_build_core(&p_, rate_, f.n_rec ? s.rec_off.data() + f.first_rec : no_rec, f.n_rec, mine.data(), mine.size(), nullptr, s.compact,
                                           s.coff ? s.coff + f.sk0 : nullptr, &payload, &plen, &f.st, s.kflags ? s.kflags + f.occ0 : nullptr);
            if (f.rc) f.err = spsp_last_error();
            if (!f.rc && (p_.flags & SPSP_SCAN_STATS)) { f.st.total_superkmer_number = f.total_superkmers; f.st.total_kmer_number = f.st.read_kmer; }
            double t1 = now_s();
            add_time(build_s_, t1 - t0);
            if (!f.rc) {
                f.rc = spsp_write_gz_host(out_[f.index], payload, plen, 9);  // level 9: SubSampler.cpp:326
                if (f.rc) f.err = spsp_last_error();
                add_time(gzip_s_, now_s() - t1);
            }
            free(payload);
        }
        {   // critical section `cout` of the reference (:791): one file's report at a time
            std::lock_guard<std::mutex> g(report_m_);
            if (f.rc) ++failed_;
            ++done_files_;
            // one worker: the reference's single thread prints a file's name, sketches it, prints its statistics, then takes
            // the next file -- name and report alternate, in list order (this worker runs the batch's files in that order)
            if (cb_ && threads_ == 1) cb_(user_, f.index, 0, SPSP_OK, nullptr, nullptr);
            if (cb_) cb_(user_, f.index, 1, f.rc, &f.st, f.rc ? f.err.c_str() : nullptr);
        }
        trace(s, 'w', t_fin);
        if (s.left.fetch_sub(1) == 1) {
            std::lock_guard<std::mutex> g(m_);
            --batches_in_flight_;
            form_batch(s);
            cv_.notify_all();
        }
    }

    void add_time(double& acc, double dt) { std::lock_guard<std::mutex> g(time_m_); acc += dt; }
    // SPSP_DEBUG_PIPE_TRACE: every task's begin and end (ms since the call began), by slot and stage, printed when the call ends
    struct TraceEv { int slot; char stage; double t0, t1; };
    void trace(const PipeSlot& s, char stage, double t0) {
        if (!trace_on_) return;
        int slot = -1;
        for (size_t j = 0; j < slots_.size(); ++j) if (slots_[j].get() == &s) slot = (int)j;
        std::lock_guard<std::mutex> g(time_m_);
        trace_.push_back(TraceEv{slot, stage, (t0 - t_begin_) * 1e3, (now_s() - t_begin_) * 1e3});
    }
    bool trace_on_ = getenv("SPSP_DEBUG_PIPE_TRACE") != nullptr;
    double t_begin_ = 0;
    std::vector<TraceEv> trace_;

    std::vector<int> devices_;
    spsp_params p_;
    double rate_;
    const char* const* in_;
    const char* const* out_;
    uint32_t n_, threads_;
    spsp_file_callback cb_;
    void* user_;
    std::vector<uint64_t> sizes_;
    uint64_t budget_ = 0;
    std::vector<std::unique_ptr<PipeSlot>> slots_;
    std::mutex m_, report_m_, time_m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    int running_ = 0, batches_in_flight_ = 0;
    uint32_t next_ = 0, failed_ = 0;
    uint64_t done_files_ = 0;
    double read_s_ = 0, build_s_ = 0, gzip_s_ = 0, setup_s_ = 0, slab_s_ = 0;
    int fatal_rc_ = 0;
    std::string fatal_err_;
};

}  // namespace

void spsp_sketch_files_release(int device) { FilePipeline::release_idle(device); }

int spsp_sketch_files_multi(const int* devices, uint32_t n_dev, const spsp_params* p, double rate, const char* const* fasta_paths, const char* const* out_paths,
                            uint32_t n, uint32_t threads, spsp_file_callback cb, void* user, spsp_stage_times* times) {
    if (!devices || n_dev == 0 || n_dev > 64 || !p || (n && (!fasta_paths || !out_paths))) { set_error("NULL argument (or not 1..64 devices)"); return SPSP_ERR_ARG; }
    const std::vector<int> device(devices, devices + n_dev);
    const int rc0 = spsp::check_params(p);
    if (rc0) return rc0;
    if (threads == 0) threads = 1;
    static const bool per_worker = getenv("SPSP_FILES_PER_WORKER") != nullptr;   // A/B switch: one GPU job per file
    static const bool abund_per_file = getenv("SPSP_DEBUG_ABUND_PER_FILE") != nullptr;   // A/B: -a > 1 as one GPU job per file (the form until round 5)
    if (per_worker || (p->abundance > 1 && abund_per_file) || n == 0)
        return sketch_files_per_worker(device, p, rate, fasta_paths, out_paths, n, threads, cb, user, times);
    if (times) memset(times, 0, sizeof *times);
    FilePipeline pipe(device, p, rate, fasta_paths, out_paths, n, threads, cb, user);
    return pipe.run(times);
}

int spsp_sketch_files(int device, const spsp_params* p, double rate, const char* const* fasta_paths, const char* const* out_paths,
                      uint32_t n, uint32_t threads, spsp_file_callback cb, void* user, spsp_stage_times* times) {
    return spsp_sketch_files_multi(&device, 1, p, rate, fasta_paths, out_paths, n, threads, cb, user, times);
}


}  // extern "C"
