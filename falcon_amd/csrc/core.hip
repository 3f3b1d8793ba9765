// Context, error reporting and the two host-side helpers (a1 get_dim, a3 hash table).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "hash.h"
#include "scan.h"
#include "ivf.h"
#include <algorithm>

#include <chrono>
namespace fal {
// FALCON_TRACE_ALLOC=1: every device allocation of the library (scratch slots, pool blocks) with its size and duration on stderr
// -- what a cold pass (falcon.main(): one pass per charge in a fresh process) pays before its first kernel
static bool trace_alloc() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FALCON_TRACE_ALLOC");
        v = e && e[0] && e[0] != '0';
    }
    return v == 1;
}
static hipError_t traced_malloc(void** p, size_t bytes, const char* what, int id) {
    if (!trace_alloc()) return hipMalloc(p, bytes);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipMalloc(p, bytes);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[falcon alloc] %s %d: %.1f MB in %.2f ms\n", what, id, bytes / 1048576.0, ms);
    return e;
}
static std::vector<const void*>& warm_kernels() {
    static std::vector<const void*> v;        // (function-local: filled by other units' static initialisers)
    return v;
}
WarmReg::WarmReg(const void* kernel) { warm_kernels().push_back(kernel); }
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace fal

void fal_ctx::release_retired() {
    if (retired.empty()) return;
    (void)hipStreamSynchronize(stream);
    for (void* p : retired) (void)hipFree(p);
    retired.clear();
    retired_held.clear();
}

int fal_ctx::reserve(int slot, size_t bytes, void** out) {
    fal::Scratch& s = scratch[slot];
    if (bytes > s.cap) {
        if (s.ptr) {
            if (debug_poison && call_depth > 0 && slot_epoch[slot] == call_epoch) {
                fal::set_error("scratch slot %d grows (%zu -> %zu bytes) after it was reserved earlier in the same call: "
                               "a pointer into its old block would be stale", slot, s.cap, bytes);
                return FAL_EINTERNAL;
            }
            if (call_depth > 0) {
                retired.push_back(s.ptr);        // kernels of this call may still use it; freed when the next call begins
                // a block whose slot was reserved earlier in THIS call may still be addressed by a pointer on the host
                retired_held.push_back(slot_epoch[slot] == call_epoch);
            } else {
                FAL_CHECK_HIP(hipStreamSynchronize(stream));
                FAL_CHECK_HIP(hipFree(s.ptr));
            }
            s.ptr = nullptr;
            s.cap = 0;
        }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = fal::traced_malloc(&s.ptr, want, "slot", slot);
        if (e == hipErrorOutOfMemory && !retired.empty()) {
            // Out of memory with retired blocks around: the ones nobody can hold a pointer into any more (their slot had been
            // released, or was last reserved by an earlier call) go once the stream has drained; a block retired from a slot
            // this call had reserved stays -- CallScope's contract -- and the call fails with FAL_ENOMEM instead.
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);
            size_t kept = 0;
            for (size_t i = 0; i < retired.size(); ++i) {
                if (retired_held[i]) {
                    retired[kept] = retired[i];
                    retired_held[kept++] = true;
                } else {
                    (void)hipFree(retired[i]);
                }
            }
            retired.resize(kept);
            retired_held.resize(kept);
            e = hipMalloc(&s.ptr, want);
            if (e == hipErrorOutOfMemory) {
                (void)hipGetLastError();
                fal::set_error("out of device memory: scratch slot %d needs %zu bytes (%zu retired block(s) of this call are still "
                               "referenced and cannot be freed before it returns)", slot, want, kept);
                return FAL_ENOMEM;
            }
        }
        FAL_CHECK_HIP(e);
        s.cap = want;
        ++slot_gen[slot];
        if (debug_poison) FAL_CHECK_HIP(hipMemsetAsync(s.ptr, 0xFF, want, stream));
    }
    slot_epoch[slot] = call_epoch;
    *out = s.ptr;
    return FAL_OK;
}

int fal_ctx::pool_alloc(size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    int best = -1;
    for (size_t i = 0; i < pool.size(); ++i)
        if (!pool[i].used && pool[i].cap >= bytes && (best < 0 || pool[i].cap < pool[best].cap)) best = (int)i;
    if (best >= 0 && pool[best].cap <= 2 * bytes + (1 << 20)) {
        pool[best].used = true;
        *out = pool[best].ptr;
        if (debug_poison) FAL_CHECK_HIP(hipMemsetAsync(pool[best].ptr, 0xFF, pool[best].cap, stream));
        return FAL_OK;
    }
    void* p = nullptr;
    const size_t cap = bytes + bytes / 16 + 256;
    FAL_CHECK_HIP(fal::traced_malloc(&p, cap, "pool", (int)pool.size()));
    pool.push_back({p, cap, true});
    if (debug_poison) FAL_CHECK_HIP(hipMemsetAsync(p, 0xFF, cap, stream));
    *out = p;
    return FAL_OK;
}

void fal_ctx::pool_free(void* ptr) {
    if (!ptr) return;
    for (auto& b : pool)
        if (b.ptr == ptr) {
            b.used = false;      // stream-ordered reuse: every user enqueues on this context's stream
            return;
        }
}

void fal_ctx::stage_reset(int stage) { timers[stage].used = 0; }

int fal_ctx::pinned_reserve(size_t bytes, void** out) {
    if (bytes > pinned_cap) {
        if (pinned) FAL_CHECK_HIP(hipHostFree(pinned));
        pinned = nullptr;
        pinned_cap = 0;
        FAL_CHECK_HIP(hipHostMalloc(&pinned, bytes + 4096, hipHostMallocDefault));
        pinned_cap = bytes + 4096;
    }
    *out = pinned;
    return FAL_OK;
}

int fal_ctx::upload(void* dst, const void* src, size_t bytes) {
    constexpr size_t kArena = 32u << 20;
    if (bytes == 0) return FAL_OK;
    if (bytes > kArena / 4) {                       // big tables: plain (host-synchronous) copy
        FAL_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return FAL_OK;
    }
    if (!arena) FAL_CHECK_HIP(hipHostMalloc((void**)&arena, kArena, hipHostMallocDefault));
    size_t off = (arena_off + 63) & ~(size_t)63;
    if (off + bytes > kArena) {                     // ring wrapped: earlier copies out of the arena must have landed
        FAL_CHECK_HIP(hipStreamSynchronize(stream));
        off = 0;
    }
    memcpy(arena + off, src, bytes);
    FAL_CHECK_HIP(hipMemcpyAsync(dst, arena + off, bytes, hipMemcpyHostToDevice, stream));
    arena_off = off + bytes;
    return FAL_OK;
}

int fal_ctx::stage_begin(int stage, hipEvent_t* stop_out, hipStream_t on) {
    StageTimer& t = timers[stage];
    if (t.used == t.ev.size()) {
        hipEvent_t a, b;
        FAL_CHECK_HIP(hipEventCreate(&a));
        FAL_CHECK_HIP(hipEventCreate(&b));
        t.ev.push_back({a, b});
    }
    FAL_CHECK_HIP(hipEventRecord(t.ev[t.used].first, on ? on : stream));
    *stop_out = t.ev[t.used].second;
    t.used++;
    return FAL_OK;
}

int fal_ctx::stage_end(hipEvent_t stop, hipStream_t on) {
    FAL_CHECK_HIP(hipEventRecord(stop, on ? on : stream));
    return FAL_OK;
}

extern "C" {

int fal_version(void) { return 100; }

const char* fal_last_error(void) { return fal::g_err; }

int fal_device_count(int* count) {
    FAL_REQUIRE(count, FAL_EINVAL, "fal_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *count = n;
    return FAL_OK;
}

int fal_ctx_create(int device, void* stream, int own_stream, fal_ctx** out) {
    FAL_REQUIRE(out, FAL_EINVAL, "fal_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fal::set_error("fal_ctx_create: no HIP device visible (this library has no CPU fallback)");
        return FAL_ENODEV;
    }
    FAL_REQUIRE(device >= 0 && device < n, FAL_EINVAL, "fal_ctx_create: device %d out of range [0,%d)", device, n);
    FAL_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FAL_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fal::set_error("fal_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return FAL_ENODEV;
    }
    fal_ctx* c = new fal_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->persistent_wgs = c->num_cus;
    if (const char* e = getenv("FALCON_PERSISTENT_WGS")) {        // experiment switch (NOTES.md round 4: CU shares of two streams)
        const int v = atoi(e);
        if (v >= 8 && v <= c->num_cus) c->persistent_wgs = v;
    }
    if (!own_stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            fal::set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return FAL_EHIP;
        }
        c->own_stream = true;
    }
    if (hipHostMalloc((void**)&c->fb_host, 64, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&c->zero_dev, 64) != hipSuccess || hipMemset(c->zero_dev, 0, 64) != hipSuccess) {
        fal::set_error("fal_ctx_create: cannot allocate the context's counters");
        fal_ctx_destroy(c);
        return FAL_ENOMEM;
    }
    memset(c->fb_host, 0, 64);
    const char* dbg = getenv("FALCON_DEBUG_POISON");
    c->debug_poison = dbg && dbg[0] && dbg[0] != '0';
    *out = c;
    return FAL_OK;
}

int fal_ctx_destroy(fal_ctx* c) {
    if (!c) return FAL_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->scratch)
        if (s.ptr) (void)hipFree(s.ptr);
    for (void* p : c->retired) (void)hipFree(p);
    for (auto& b : c->pool) (void)hipFree(b.ptr);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->fb_host) (void)hipHostFree(c->fb_host);
    if (c->zero_dev) (void)hipFree(c->zero_dev);
    if (c->arena) (void)hipHostFree(c->arena);
    for (auto& t : c->timers)
        for (auto& p : t.ev) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FAL_OK;
}

// Sizes the scratch a pass over n spectra will ask for and loads the kernels' code objects, so that the FIRST pass of a fresh
// process -- the only pass falcon.main() makes per charge (reference falcon.py:153-193) -- does not stop between its kernels for
// them.  What is sized: the slots whose size follows from n alone (sort buffers, the tail's and the graph's per-row arrays, the
// flat scan's hand-off up to its cap); slots that depend on the bucket structure keep growing on demand (a device allocation
// costs ~30 us on this driver: FALCON_TRACE_ALLOC=1).  Safe to call at any time; never shrinks anything.
int fal_ctx_plan(fal_ctx* c, int64_t n, int low_dim, int k_ann, int n_probe, int64_t batch_size) {
    fal::CallScope _call(c);
    FAL_REQUIRE(c && n >= 0 && low_dim >= 1 && k_ann >= 1 && n_probe >= 1 && batch_size >= 1, FAL_EINVAL, "fal_ctx_plan: bad argument");
    FAL_CHECK_HIP(hipSetDevice(c->device));
    for (const void* k : fal::warm_kernels()) {
        hipFuncAttributes attr;
        FAL_CHECK_HIP(hipFuncGetAttributes(&attr, k));           // (loads the unit's code object on this device)
    }
    // the pinned upload ring of the job tables (32 MB of page-locked host memory: milliseconds to get)
    if (!c->arena) FAL_CHECK_HIP(hipHostMalloc((void**)&c->arena, (size_t)32u << 20, hipHostMallocDefault));
    if (n == 0) return FAL_OK;
    void* p = nullptr;
    using namespace fal;
    const size_t nn = (size_t)n;
    // sort by precursor / list sorts: keys + values, in and out (sortutil.hip, ivf.hip)
    FAL_TRY(c->reserve(SLOT_SORT, 24 * nn + 4096, &p));
    FAL_TRY(c->reserve(SLOT_SORT2, 12 * nn + 4096, &p));
    // the flat scan's hand-off: one [32, ceil32(n_b)] block per query tile, batches of at most the cap (search.hip); a bucket holds
    // at most batch_size rows -- the whole job's need is bounded by n * min(n, batch_size) floats, the cap (2 GiB) usually binds
    {
        const char* e = getenv("FALCON_SIMS_MB");
        size_t cap = (e ? (size_t)atoll(e) : 2048) * 1024 * 1024;
        const double need = 4.0 * (double)nn * (double)std::min<int64_t>(n, std::min<int64_t>(batch_size, 4096));
        if (need < (double)cap) cap = (size_t)need;
        FAL_TRY(c->reserve(SLOT_SIMS, cap + sizeof(float) * kSimsSlack, &p));
    }
    // per-row arrays of the graph stages (graph.hip, tail.hip): labels, parents, segment tables
    for (int slot : {SLOT_DB, SLOT_DB2, SLOT_TAIL, SLOT_TAIL2, SLOT_FIN, SLOT_FIN2}) FAL_TRY(c->reserve(slot, 8 * nn + 4096, &p));
    for (int slot = 0; slot < 32; ++slot) c->release(slot);      // (no pointer is held: the passes may still grow them)
    return FAL_OK;
}

// Gives the context's cached device memory back to the driver: every scratch slot, every pool block no index holds, the retired
// blocks.  For a process that runs jobs of very different sizes one after the other (bench.py between its configurations; a
// falcon.main() that has finished a charge partition of 50 M spectra and goes on to one of 10 k).  The stream is drained first.
int fal_ctx_trim(fal_ctx* c) {
    fal::CallScope _call(c);
    FAL_REQUIRE(c, FAL_EINVAL, "fal_ctx_trim: NULL ctx");
    FAL_CHECK_HIP(hipStreamSynchronize(c->stream));
    c->release_retired();
    for (auto& s : c->scratch)
        if (s.ptr) {
            (void)hipFree(s.ptr);
            s.ptr = nullptr;
            s.cap = 0;
        }
    size_t kept = 0;
    for (size_t i = 0; i < c->pool.size(); ++i) {
        if (c->pool[i].used) c->pool[kept++] = c->pool[i];
        else (void)hipFree(c->pool[i].ptr);
    }
    c->pool.resize(kept);
    return FAL_OK;
}

int fal_ctx_sync(fal_ctx* c) {
    fal::CallScope _call(c);
    FAL_REQUIRE(c, FAL_EINVAL, "fal_ctx_sync: NULL ctx");
    FAL_CHECK_HIP(hipStreamSynchronize(c->stream));
    return FAL_OK;
}

int fal_ctx_enable_timing(fal_ctx* c, int on) {
    FAL_REQUIRE(c, FAL_EINVAL, "NULL ctx");
    c->timing = on != 0;
    return FAL_OK;
}

int fal_ctx_counter(fal_ctx* c, int which, int64_t* value) {
    FAL_REQUIRE(c && value && which >= 0 && which < 10, FAL_EINVAL, "fal_ctx_counter: bad argument");
    *value = c->counters[which];
    if (which == 5) *value = c->fb_host ? (int64_t)c->fb_host[0] + (int64_t)c->fb_host[2] : 0;   // fallback queries of the last prefiltered search: flat + IVF buckets (after a sync)
    return FAL_OK;
}

int fal_ctx_stage_ms(fal_ctx* c, int stage, float* ms, int64_t* launches) {
    fal::CallScope _call(c);
    FAL_REQUIRE(c && ms && stage >= 0 && stage < fal::kNumStages, FAL_EINVAL, "fal_ctx_stage_ms: bad argument");
    FAL_CHECK_HIP(hipStreamSynchronize(c->stream));
    float total = 0.f;
    auto& t = c->timers[stage];
    for (size_t i = 0; i < t.used; ++i) {
        float x = 0.f;
        FAL_CHECK_HIP(hipEventElapsedTime(&x, t.ev[i].first, t.ev[i].second));
        total += x;
    }
    *ms = total;
    if (launches) *launches = (int64_t)t.used;
    return FAL_OK;
}

// a1 -- reference spectrum.py:172-199: all float32 (numba signature f4,f4,f4 -> u4,f4,f4).
int fal_get_dim(float min_mz, float max_mz, float bin_size, uint32_t* dim, float* start_dim, float* end_dim) {
    FAL_REQUIRE(dim && start_dim && end_dim, FAL_EINVAL, "fal_get_dim: NULL output");
    FAL_REQUIRE(bin_size > 0.f && max_mz >= min_mz, FAL_EINVAL, "fal_get_dim: need bin_size > 0 and max_mz >= min_mz");
    volatile float s = min_mz - fmodf(min_mz, bin_size);
    volatile float t = max_mz + bin_size;
    volatile float e = t - fmodf(max_mz, bin_size);
    volatile float w = e - s;
    volatile float q = w / bin_size;
    *start_dim = s;
    *end_dim = e;
    *dim = (uint32_t)ceilf(q);
    return FAL_OK;
}

// a3 -- README.md:124-131: MurmurHash3_x86_32 of the int32 bin index, seed, unsigned, mod low_dim.
int fal_hash_lookup(uint32_t n_bins, uint32_t low_dim, uint32_t seed, uint32_t* out_table) {
    FAL_REQUIRE(out_table && low_dim > 0, FAL_EINVAL, "fal_hash_lookup: bad argument");
    for (uint32_t i = 0; i < n_bins; ++i) out_table[i] = fal::murmur3_32(i, seed) % low_dim;
    return FAL_OK;
}

}  // extern "C"
