// spsp_abi.hip -- context lifecycle, error reporting and the host-buffer forms
// of the GPU entry points declared in include/spsp.h.
#include <cstdlib>
#include <cstring>
#include <vector>

#include <mutex>
#include <utility>
#include <vector>

#include "spsp_internal.h"

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

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
    return (e == hipErrorOutOfMemory) ? SPSP_ERR_NOMEM : SPSP_ERR_HIP;
}

// A buffer that is outgrown is not freed on the spot: hipFree waits for the whole device and works off whatever the runtime
// has deferred -- 19 ms measured for ONE free in front of a 40 ms decode of 10 000 sketch files -- and buffers grow
// geometrically, so the outgrown ones add up to less than what is in use.  They are kept in a list and freed when a context
// is destroyed, or when the list passes 16 GiB.
namespace {
struct Retired { std::mutex mu; std::vector<std::pair<void*, size_t>> list; size_t bytes = 0; };
Retired& retired() { static Retired r; return r; }
void retire(void* p, size_t bytes) {
    Retired& r = retired();
    std::vector<std::pair<void*, size_t>> now;
    {
        std::lock_guard<std::mutex> g(r.mu);
        r.list.emplace_back(p, bytes);
        r.bytes += bytes;
        if (r.bytes > (16ull << 30)) { now.swap(r.list); r.bytes = 0; }
    }
    for (auto& q : now) (void)hipFree(q.first);
}
}  // namespace
void devbuf_flush_retired() {
    Retired& r = retired();
    std::vector<std::pair<void*, size_t>> now;
    { std::lock_guard<std::mutex> g(r.mu); now.swap(r.list); r.bytes = 0; }
    for (auto& q : now) (void)hipFree(q.first);
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap && p) return SPSP_OK;
    static const bool dbg = getenv("SPSP_DEBUG_ALLOC_TIMES") != nullptr;   // analysis: what growing a buffer costs (stderr)
    const double t0 = dbg ? now_s() : 0.0;
    const size_t had = cap;
    if (p) { retire(p, cap); p = nullptr; cap = 0; }
    const double t1 = dbg ? now_s() : 0.0;
    size_t want = bytes + bytes / 4 + 256;  // headroom so batch loops settle quickly
    hipError_t e = hipMalloc(&p, want);
    if (dbg) fprintf(stderr, "[spsp alloc] %zu -> %zu bytes: free %.2f ms, malloc %.2f ms\n", had, want, (t1 - t0) * 1e3, (now_s() - t1) * 1e3);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return SPSP_ERR_NOMEM;
    }
    cap = want;
    return SPSP_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
}

}  // namespace spsp

using namespace spsp;

int spsp_ctx::ev_begin(int kind) {
    if (!(timing_mask & (1u << kind))) return SPSP_OK;   // (scatter and group share one bit)
    if (timing_every > 1 && (ev_seq[kind]++ % timing_every) != 0) { ev_open[kind] = false; return SPSP_OK; }   // sampled: spsp_timing_sample
    EventLog& L = evlog[kind];
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!L.spare.empty()) { ev = L.spare.back(); L.spare.pop_back(); }
    else { SPSP_HIP(hipEventCreate(&ev.first)); SPSP_HIP(hipEventCreate(&ev.second)); }
    L.used.push_back(ev);
    ev_open[kind] = true;
    SPSP_HIP(hipEventRecord(ev.first, stream));
    return SPSP_OK;
}
// a bracket whose two events the caller attaches to a kernel dispatch itself (hipExtLaunchKernelGGL): nothing is
// recorded on the stream here.  false = this region is not timed (timing off, or sampled out)
bool spsp_ctx::ev_pair(int kind, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!(timing_mask & (1u << kind))) return false;
    if (timing_every > 1 && (ev_seq[kind]++ % timing_every) != 0) return false;
    EventLog& L = evlog[kind];
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!L.spare.empty()) { ev = L.spare.back(); L.spare.pop_back(); }
    else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return false;
    L.used.push_back(ev);
    *start = ev.first; *stop = ev.second;
    return true;
}
int spsp_ctx::ev_end(int kind) {
    // closes the bracket opened by the matching ev_begin only: timing may have been switched on, off or read
    // between the two calls (the comparison's bracket spans API calls)
    if (!ev_open[kind] || evlog[kind].used.empty()) { ev_open[kind] = false; return SPSP_OK; }
    ev_open[kind] = false;
    SPSP_HIP(hipEventRecord(evlog[kind].used.back().second, stream));
    return SPSP_OK;
}

extern "C" {

int spsp_timing_enable(spsp_ctx* ctx, int on) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    // SPSP_TIME_*: bit 0 dense kernel, 1 scan pipeline, 2 accumulate kernel, 3 compare pipeline
    ctx->timing_mask = (uint32_t)on & 15u;
    if (on & SPSP_TIME_PARTS) ctx->timing_mask |= (1u << kEvScatter) | (1u << kEvGroup);
    ctx->timing = ctx->timing_mask != 0;
    return SPSP_OK;
}

int spsp_timing_sample(spsp_ctx* ctx, uint32_t every) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    ctx->timing_every = every ? every : 1;
    for (int kind = 0; kind < kEvKinds; ++kind) ctx->ev_seq[kind] = 0;
    return SPSP_OK;
}

int spsp_timing_read(spsp_ctx* ctx, spsp_timing* out) {
    if (!ctx || !out) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    double ms[kEvKinds]; uint64_t cnt[kEvKinds];
    for (int kind = 0; kind < kEvKinds; ++kind) {
        ms[kind] = 0; cnt[kind] = 0;
        EventLog& L = ctx->evlog[kind];
        // a bracket whose end has not been recorded yet (a comparison between its begin and end calls) stays open
        const size_t done = L.used.size() - ((ctx->ev_open[kind] && !L.used.empty()) ? 1 : 0);
        for (size_t i = 0; i < done; ++i) {
            auto& ev = L.used[i];
            float t = 0;
            SPSP_HIP(hipEventElapsedTime(&t, ev.first, ev.second));
            ms[kind] += t; ++cnt[kind];
            L.spare.push_back(ev);
        }
        L.used.erase(L.used.begin(), L.used.begin() + done);
    }
    out->dense_ms = ms[kEvDense]; out->dense_launches = cnt[kEvDense];
    out->scan_ms = ms[kEvScan]; out->scan_calls = cnt[kEvScan];
    out->accumulate_ms = ms[kEvAccumulate]; out->accumulate_launches = cnt[kEvAccumulate];
    out->compare_ms = ms[kEvCompare]; out->compare_calls = cnt[kEvCompare];
    out->scatter_ms = ms[kEvScatter]; out->scatter_launches = cnt[kEvScatter];
    out->group_ms = ms[kEvGroup]; out->group_launches = cnt[kEvGroup];
    return SPSP_OK;
}

const char* spsp_last_error(void) { return g_err.c_str(); }
const char* spsp_version(void) { return "spsp-mi355x 0.1 (gfx950)"; }

int spsp_create(int device, void* hip_stream, spsp_ctx** out) {
    if (!out) { set_error("out is NULL"); return SPSP_ERR_ARG; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); libspsp has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return SPSP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { set_error("device %d out of range (0..%d)", device, count - 1); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPSP_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libspsp carries gfx950 code objects only", device, prop.gcnArchName);
        return SPSP_ERR_NO_DEVICE;
    }
    spsp_ctx* c = new spsp_ctx();
    c->device = device;
    c->n_cu = c->n_cu_device = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
        c->own_stream = true;
    }
    e = hipHostMalloc((void**)&c->h_scalar, 16 * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) { if (c->own_stream) (void)hipStreamDestroy(c->stream); delete c; return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__); }
    memset(c->h_scalar, 0, 16 * sizeof(uint64_t));
    *out = c;
    return SPSP_OK;
}

int spsp_device_count(void) {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { set_error("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e)); return 0; }
    int usable = 0;
    for (int d = 0; d < count; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable; else break;
    }
    if (!usable) set_error("no gfx950 device among the %d visible", count);
    return usable;
}

int spsp_stream_create_cus(int device, uint32_t first_cu, uint32_t n_cu, void** hip_stream) {
    if (!hip_stream) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    *hip_stream = nullptr;
    SPSP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPSP_HIP(hipGetDeviceProperties(&prop, device));
    const uint32_t total = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    if (n_cu == 0 || first_cu >= total || n_cu > total - first_cu) {
        set_error("CU range [%u, %u) outside the device's %u compute units", first_cu, first_cu + n_cu, total);
        return SPSP_ERR_ARG;
    }
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (uint32_t c = first_cu; c < first_cu + n_cu; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    SPSP_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    *hip_stream = (void*)s;
    return SPSP_OK;
}

int spsp_stream_destroy(int device, void* hip_stream) {
    if (!hip_stream) return SPSP_OK;
    SPSP_HIP(hipSetDevice(device));
    SPSP_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    SPSP_HIP(hipStreamDestroy((hipStream_t)hip_stream));
    return SPSP_OK;
}

int spsp_set_cu_count(spsp_ctx* ctx, uint32_t n_cu, uint32_t dense_blocks_per_cu) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (ctx->scan_job.pending) { set_error("a scan is pending on this context"); return SPSP_ERR_ARG; }
    if (n_cu > (uint32_t)ctx->n_cu_device) { set_error("the device has %d compute units", ctx->n_cu_device); return SPSP_ERR_ARG; }
    if (dense_blocks_per_cu > 2) { set_error("at most two dense workgroups fit a CU"); return SPSP_ERR_ARG; }
    ctx->n_cu = n_cu ? (int)n_cu : ctx->n_cu_device;
    ctx->dense_blocks_per_cu = dense_blocks_per_cu ? (int)dense_blocks_per_cu : 1;
    return SPSP_OK;
}

void spsp_destroy(spsp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->tail_stream) (void)hipStreamSynchronize(c->tail_stream);
    compare_job_drop(c);
    if (c->h_skoff) (void)hipHostFree(c->h_skoff);
    for (auto& g : c->h_read_regions) free(g.p);
    if (c->h_text) (void)hipHostFree(c->h_text);
    if (c->h_keys) (void)hipHostFree(c->h_keys);
    if (c->keys_done) (void)hipEventDestroy(c->keys_done);
    if (c->dense_done) (void)hipEventDestroy(c->dense_done);
    if (c->tail_event) (void)hipEventDestroy(c->tail_event);
    if (c->scan_done) (void)hipEventDestroy(c->scan_done);
    if (c->compare_done) (void)hipEventDestroy(c->compare_done);
    DevBuf* bufs[] = {&c->bases, &c->rec_off, &c->bitmap, &c->tile_count, &c->tile_off, &c->hits, &c->emit_count,
                      &c->scan_tmp, &c->d_scalar, &c->seg_a, &c->seg_b, &c->wave_hits, &c->wave_cnt, &c->packed, &c->unpacked, &c->st_count, &c->st_open, &c->st_total, &c->st_over, &c->filter, &c->bloom, &c->pairtab, &c->c_min, &c->c_lo, &c->c_hi, &c->c_table,
                      &c->c_owner, &c->c_rowid, &c->c_row, &c->x_cnt, &c->x_off, &c->x_begin, &c->x_end, &c->x_tot, &c->dc_text, &c->dc_desc, &c->dc_mn, &c->dc_lo, &c->dc_hi, &c->dc_meta, &c->dc_walk, &c->bl_hist, &c->bl_keys, &c->bl_vals, &c->bl_meta, &c->bl_lo, &c->bl_hi, &c->bl_pmin, &c->bl_pb, &c->bl_slot, &c->bl_first, &c->bl_codes, &c->bl_text, &c->bl_outs, &c->bl_out, &c->a_cnt, &c->a_off, &c->a_mn, &c->a_lo, &c->a_hi, &c->a_slot, &c->a_slot_of, &c->a_flags, &c->a_seg, &c->b_mn, &c->b_lo, &c->b_hi, &c->b_table, &c->b_tiles, &c->b_seg, &c->m_send, &c->m_recv, &c->m_cells, &c->m_mn, &c->m_lo, &c->m_hi, &c->c_matrix, &c->c_inter, &c->c_flags, &c->c_slot_lo, &c->c_slot_hi, &c->c_slot_mn, &c->c_part_cnt, &c->c_recs, &c->c_where, &c->c_lref, &c->c_filter, &c->c_bits, &c->c_sig, &c->c_order, &c->c_multi, &c->scan_blocks,
                      &c->c_skoff, &c->i_text, &c->i_tiles, &c->i_entry, &c->i_outoff, &c->i_recbase, &c->i_lens, &c->i_dst,
                      &c->i_compact};
    for (DevBuf* b : bufs) b->release();
    devbuf_flush_retired();
    for (int kind = 0; kind < kEvKinds; ++kind) {
        for (auto* v : {&c->evlog[kind].used, &c->evlog[kind].spare})
            for (auto& ev : *v) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    }
    if (c->h_scalar) (void)hipHostFree(c->h_scalar);
    if (c->own_tail_stream && c->tail_stream) { (void)hipStreamSynchronize(c->tail_stream); (void)hipStreamDestroy(c->tail_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void spsp_free(void* p) { free(p); }

int spsp_copy_to_host(spsp_ctx* ctx, void* dst, const void* d_src, uint64_t bytes) {
    if (!ctx || (bytes && (!dst || !d_src))) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (!bytes) return SPSP_OK;
    SPSP_HIP(hipSetDevice(ctx->device));
    SPSP_HIP(hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    SPSP_HIP(hipStreamSynchronize(ctx->stream));
    return SPSP_OK;
}
int spsp_scan(spsp_ctx* ctx, const spsp_params* p, const uint8_t* bases, const uint64_t* rec_off, uint32_t n_rec,
              spsp_superkmer** out, uint64_t* n_out) {
    if (!ctx || !out || !n_out) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    *out = nullptr; *n_out = 0;
    int rc = check_params(p);
    if (rc) return rc;
    if (p->flags & SPSP_SCAN_PACKED_INPUT) { set_error("SPSP_SCAN_PACKED_INPUT is for the device forms (spsp_scan_device...)"); return SPSP_ERR_ARG; }
    if (n_rec == 0) return SPSP_OK;
    if (!bases || !rec_off) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (rec_off[0] != 0) { set_error("rec_off[0] must be 0"); return SPSP_ERR_ARG; }
    for (uint32_t r = 0; r < n_rec; ++r)
        if (rec_off[r + 1] < rec_off[r]) { set_error("rec_off must be non-decreasing"); return SPSP_ERR_ARG; }
    const uint64_t n = rec_off[n_rec];
    if (n < p->k) return SPSP_OK;
    SPSP_HIP(hipSetDevice(ctx->device));
    if ((rc = ctx->bases.reserve((size_t)n + 64))) return rc;
    if ((rc = ctx->rec_off.reserve((size_t)(n_rec + 1) * 8))) return rc;
    SPSP_HIP(hipMemcpyAsync(ctx->bases.p, bases, n, hipMemcpyHostToDevice, ctx->stream));
    SPSP_HIP(hipMemcpyAsync(ctx->rec_off.p, rec_off, (size_t)(n_rec + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    spsp_superkmer* d_out = nullptr;
    uint64_t cnt = 0;
    rc = scan_device_impl(ctx, p, ctx->bases.as<uint8_t>(), n, ctx->rec_off.as<uint64_t>(), n_rec, &d_out, &cnt);
    if (rc) return rc;
    if (cnt == 0) { SPSP_HIP(hipStreamSynchronize(ctx->stream)); return SPSP_OK; }
    spsp_superkmer* h = (spsp_superkmer*)malloc((size_t)cnt * sizeof(spsp_superkmer));
    if (!h) { set_error("out of host memory"); return SPSP_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(h, d_out, (size_t)cnt * sizeof(spsp_superkmer), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(h); return hip_fail(e, "copy back super-k-mers", __FILE__, __LINE__); }
    *out = h; *n_out = cnt;
    return SPSP_OK;
}

int spsp_scan_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                     const void* d_rec_off, uint32_t n_rec, void** d_out, uint64_t* n_out) {
    if (!ctx || !d_out || !n_out) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    spsp_superkmer* o = nullptr;
    int rc = scan_device_impl(ctx, p, (const uint8_t*)d_bases, n_bases, (const uint64_t*)d_rec_off, n_rec, &o, n_out);
    *d_out = o;
    return rc;
}

int spsp_pack_bases_device(spsp_ctx* ctx, const void* d_bases, uint64_t n_bases, void** d_packed) {
    if (!ctx || !d_packed || (n_bases && !d_bases)) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    uint32_t* o = nullptr;
    const int rc = pack_bases_impl(ctx, (const uint8_t*)d_bases, n_bases, &o);
    *d_packed = o;
    return rc;
}

int spsp_scan_device_begin(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                           const void* d_rec_off, uint32_t n_rec) {
    if (!ctx || !p) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return scan_begin_impl(ctx, p, (const uint8_t*)d_bases, n_bases, (const uint64_t*)d_rec_off, n_rec);
}

int spsp_scan_device_end(spsp_ctx* ctx, void** d_out, uint64_t* n_out) {
    if (!ctx || !d_out || !n_out) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    spsp_superkmer* o = nullptr;
    int rc = scan_end_impl(ctx, &o, n_out);
    *d_out = o;
    return rc;
}

int spsp_scan_tail_stream(spsp_ctx* ctx, int tail, void* hip_stream) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (ctx->scan_job.pending) { set_error("a scan is pending on this context"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    if (ctx->tail_stream) {
        SPSP_HIP(hipStreamSynchronize(ctx->tail_stream));
        if (ctx->own_tail_stream) (void)hipStreamDestroy(ctx->tail_stream);
        ctx->tail_stream = nullptr; ctx->own_tail_stream = false;
    }
    if (!tail) return SPSP_OK;
    if (hip_stream) { ctx->tail_stream = (hipStream_t)hip_stream; return SPSP_OK; }
    SPSP_HIP(hipStreamCreateWithFlags(&ctx->tail_stream, hipStreamNonBlocking));
    ctx->own_tail_stream = true;
    return SPSP_OK;
}

int spsp_wait_dense(spsp_ctx* waiter, spsp_ctx* scanner) {
    if (!waiter || !scanner) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (waiter->device != scanner->device) { set_error("both contexts must be on the same device"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(waiter->device));
    if (scanner->dense_marker && waiter->stream != scanner->stream)
        SPSP_HIP(hipStreamWaitEvent(waiter->stream, scanner->dense_marker, 0));
    return SPSP_OK;
}

int spsp_wait_stream(spsp_ctx* waiter, spsp_ctx* other) {
    if (!waiter || !other) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (waiter->device != other->device) { set_error("both contexts must be on the same device"); return SPSP_ERR_ARG; }
    if (waiter->stream == other->stream) return SPSP_OK;
    SPSP_HIP(hipSetDevice(waiter->device));
    if (!other->tail_event) SPSP_HIP(hipEventCreateWithFlags(&other->tail_event, hipEventDisableTiming));
    SPSP_HIP(hipEventRecord(other->tail_event, other->stream));
    SPSP_HIP(hipStreamWaitEvent(waiter->stream, other->tail_event, 0));
    return SPSP_OK;
}

// what a context has learnt from earlier comparisons (a good row order of the input's own, lists for most records, parts that
// spilled, the filter's pass rate) only changes WHICH kernels the next comparison queues, never its result; a caller that
// times or profiles comparisons of different collections on one context can start from a clean slate
int spsp_compare_forget(spsp_ctx* ctx) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    ctx->learnt_on = 0; ctx->order_quiet = 0; ctx->multi_quiet = 0; ctx->spill_expect = 0; ctx->filter_ratio = 1.0;
    return SPSP_OK;
}

int spsp_compare_keys_unordered(spsp_ctx* ctx, int on) {
    if (!ctx) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    ctx->keys_unordered = on != 0;
    return SPSP_OK;
}

int spsp_scan_output_wait(spsp_ctx* scanner, spsp_ctx* reader) {
    if (!scanner || !reader) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    if (scanner->device != reader->device) { set_error("both contexts must be on the same device"); return SPSP_ERR_ARG; }
    if (!reader->keys_done) return SPSP_OK;                 // nothing of that kind was ever queued on `reader`
    SPSP_HIP(hipSetDevice(scanner->device));
    // the output buffer is written by the scan's LAST stage (the write pass of the cluster replay): only the stream that
    // runs the sparse stages has to wait, the dense pass in front of them does not
    SPSP_HIP(hipStreamWaitEvent(scanner->sparse_stream(), reader->keys_done, 0));
    return SPSP_OK;
}

int spsp_scan_hits_device(spsp_ctx* ctx, const spsp_params* p, const void* d_bases, uint64_t n_bases,
                          uint64_t* n_hits) {
    if (!ctx || !n_hits) { set_error("NULL argument"); return SPSP_ERR_ARG; }
    SPSP_HIP(hipSetDevice(ctx->device));
    return scan_hits_impl(ctx, p, (const uint8_t*)d_bases, n_bases, n_hits);
}

}  // extern "C"
