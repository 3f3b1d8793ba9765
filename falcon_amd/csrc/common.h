// Internal header of libfalcon_hip.so (gfx950 only).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/falcon_hip.h"

namespace fal {

void set_error(const char* fmt, ...);

#define FAL_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            fal::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,        \
                           hipGetErrorString(_e));                                   \
            return (_e == hipErrorOutOfMemory) ? FAL_ENOMEM : FAL_EHIP;              \
        }                                                                            \
    } while (0)

#define FAL_REQUIRE(cond, code, ...)                                                 \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fal::set_error(__VA_ARGS__);                                             \
            return (code);                                                           \
        }                                                                            \
    } while (0)

#define FAL_TRY(expr)                                                                \
    do {                                                                             \
        int _r = (expr);                                                             \
        if (_r != FAL_OK) return _r;                                                 \
    } while (0)

constexpr int kNumStages = 9;
enum Stage { ST_VECTORIZE = 0, ST_BUILD = 1, ST_COARSE = 2, ST_SCAN = 3, ST_SELECT = 4,
             ST_FILTER = 5, ST_DBSCAN = 6, ST_TAIL = 7,
             ST_KERNEL = 8 };   // the launches of the cosine kernel alone (dense_kernel / scan16_kernel / list16_kernel / ivf_list4_kernel;
                                // also counted in ST_SCAN): the per-launch duration behind bench.py's roofline

// A device buffer that only ever grows; lives in the context.
struct Scratch {
    void* ptr = nullptr;
    size_t cap = 0;
};

}  // namespace fal

struct fal_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* pinned = nullptr;               // small pinned host staging buffer (single-sync readbacks)
    size_t pinned_cap = 0;
    int pinned_reserve(size_t bytes, void** out);
    // host -> device copy of a small table through a pinned ring: truly asynchronous (a pageable source makes
    // hipMemcpyAsync wait for everything queued on the stream before it)
    unsigned char* arena = nullptr;
    size_t arena_off = 0;
    int upload(void* dst, const void* src, size_t bytes);
    int num_cus = 256;
    int persistent_wgs = 256;             // workgroups of the persistent one-per-CU kernels (dense4 / dense_tiny4): num_cus, or a
                                          // share of them when another context's stream is meant to run beside this one
    bool timing = false;
    // per-stage accumulated event pairs for the LAST call of that stage
    struct StageTimer {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        size_t used = 0;
    } timers[fal::kNumStages];
    fal::Scratch scratch[32];
    int32_t* fb_host = nullptr;           // pinned, 16 words, zeroed at creation: [0] / [2] fallback queries of the last prefiltered
                                          // search (flat / IVF buckets), [1] ambiguous rows of the last k-means pass
    int32_t* zero_dev = nullptr;          // 16 zero words on the device (stream-ordered resets of fb_host)
    int64_t counters[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // caching device allocator for per-call objects (index arrays): blocks are recycled, never
    // returned to the driver before the context dies (hipMalloc / hipFree cost ~100 us each and
    // hipFree synchronises the device)
    struct PoolBlock { void* ptr; size_t cap; bool used; };
    std::vector<PoolBlock> pool;
    int pool_alloc(size_t bytes, void** out);
    void pool_free(void* ptr);

    // Scratch slots (`reserve`): a pointer handed out stays valid until the public call that obtained it returns -- a slot
    // that has to grow retires its old block instead of freeing it; retired blocks are freed when the next public call
    // begins (fal::CallScope).  FALCON_DEBUG_POISON=1: every grown slot and every pool block is filled with 0xFF before use
    // (reads of scratch nobody wrote show up as NaN / -1 instead of a previous test's plausible data), and a slot that
    // grows after it was already reserved inside the same public call fails with FAL_EINTERNAL (the earlier pointer would
    // silently address the old block).
    bool debug_poison = false;
    int call_depth = 0;
    uint64_t call_epoch = 1;
    uint64_t slot_epoch[32] = {};
    uint32_t slot_gen[32] = {};
    std::vector<void*> retired;
    std::vector<bool> retired_held;       // parallel to `retired`: the slot had been reserved in the call that retired the block
    void release_retired();
    int reserve(int slot, size_t bytes, void** out);
    // the caller holds no pointer into the slot any more (kernels already enqueued keep their block: it is retired, not freed,
    // if the slot grows): the next reserve of this call may grow it without tripping the debug check
    void release(int slot) { slot_epoch[slot] = 0; }
    void stage_reset(int stage);
    int stage_begin(int stage, hipEvent_t* stop_out, hipStream_t on = nullptr);
    int stage_end(hipEvent_t stop, hipStream_t on = nullptr);
};

namespace fal {

// RAII-less helper: time one kernel (or a group) when ctx->timing is on.
struct StageScope {
    fal_ctx* c;
    hipEvent_t stop = nullptr;
    bool on;
    hipStream_t s;
    StageScope(fal_ctx* ctx, int stage, hipStream_t stream = nullptr, bool enable = true)
        : c(ctx), on(ctx->timing && enable), s(stream) {
        if (on) c->stage_begin(stage, &stop, s);
    }
    ~StageScope() {
        if (on && stop) c->stage_end(stop, s);
    }
};

// First statement of every public entry point that uses a context: scratch pointers obtained below it stay valid until it
// ends (see fal_ctx::reserve).  Entry points call each other; only the outermost scope counts.
struct CallScope {
    fal_ctx* c;
    explicit CallScope(fal_ctx* ctx) : c(ctx) {
        if (c && c->call_depth++ == 0) {
            ++c->call_epoch;
            if (!c->retired.empty()) c->release_retired();
        }
    }
    ~CallScope() {
        if (c) --c->call_depth;
    }
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// One kernel of every translation unit, registered at load time: hipcc's runtime loads a translation unit's code object when one
// of its kernels is first used on a device (a few ms each, ~20 units) -- fal_ctx_plan touches them all up front
// (hipFuncGetAttributes) so that the first pass of a fresh process does not pay the loads between its kernels.
struct WarmReg {
    explicit WarmReg(const void* kernel);
};
#define FAL_WARM_KERNEL(...) static const ::fal::WarmReg _fal_warm_reg((const void*)(__VA_ARGS__))

// LDS hand-off between the lanes of ONE wave: every lane's LDS writes in front, the reads of other lanes' data behind.
// LDS operations of a wave execute in order, so no hardware wait is needed beyond the release fence's; what is also needed
// is that hipcc keeps the reads behind the barrier.  Rounds 1-2 wrote `fence(release) + wave_barrier` only -- a release orders
// nothing that FOLLOWS it, so a later LDS read of another lane's data could legally be scheduled above that lane's write.  No
// miscompiled instance was found, but the pattern sits in 17 places that decide parity; the empty asm with a memory clobber pins
// every memory access on its side of the barrier at no cost.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ---- two 16-bit keys per register: packed VALU operations ----------------------------------------------------------------
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t x, uint32_t y) {           // two 16-bit lanes, wrapping
    return __builtin_bit_cast(uint32_t, (ushort2v)(__builtin_bit_cast(ushort2v, x) - __builtin_bit_cast(ushort2v, y)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, (ushort2v)(__builtin_bit_cast(ushort2v, x) + __builtin_bit_cast(ushort2v, y)));
}
// (inline asm: hipcc has no packed selection for the saturating subtraction and turns min(sat(x - y), 1) into two compares, two
// selects and a byte permute per word)
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t x, uint32_t y) {       // max(x - y, 0) per 16-bit lane
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t pk_min(uint32_t x, uint32_t y) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t x, uint32_t y) {
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// ---- reductions over the 64 lanes on the DPP network (VALU only: no ds_bpermute, no LDS round trip per step) ----------------
// Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8), then the row results handed up: row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3.  Lane L ends with the reduction of lanes 0 .. L; lane 63 with the whole wave's.
#define FAL_DPP_SCAN(v, OP, IDENT)                                                                  \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x111, 0xF, 0xF, false));                       \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x112, 0xF, 0xF, false));                       \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x114, 0xF, 0xF, false));                       \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x118, 0xF, 0xF, false));                       \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x142, 0xA, 0xF, false));                       \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x143, 0xC, 0xF, false));
__device__ __forceinline__ int fal_add_i(int a, int b) { return a + b; }
__device__ __forceinline__ int fal_min_i(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int fal_max_i(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int wave_prefix_sum(int v) {         // inclusive prefix sum over the lanes
    FAL_DPP_SCAN(v, fal_add_i, 0)
    return v;
}
__device__ __forceinline__ int wave_min(int v) {                // (signed compare: callers pass values below 2^31)
    FAL_DPP_SCAN(v, fal_min_i, 0x7FFFFFFF)
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t u) {  // unsigned, any value: order-preserving flip into the signed range
    int v = (int)(u ^ 0x80000000u);
    FAL_DPP_SCAN(v, fal_min_i, 0x7FFFFFFF)
    return (uint32_t)__builtin_amdgcn_readlane(v, 63) ^ 0x80000000u;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t u) {
    int v = (int)(u ^ 0x80000000u);
    FAL_DPP_SCAN(v, fal_max_i, (int)0x80000000)
    return (uint32_t)__builtin_amdgcn_readlane(v, 63) ^ 0x80000000u;
}

// sum over the 16 lanes of a DPP row (a quarter wave), every lane of the row gets it: four rotations inside the row
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);            // row_ror:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);            // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);            // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);            // row_ror:8
    return v;
}

// inclusive prefix sum inside the 16 lanes of a DPP row: row_shr 1, 2, 4, 8 with zeros shifted in (bound_ctrl)
__device__ __forceinline__ int row16_prefix_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    return v;
}

// Workgroup barrier behind LDS-DMA (`global_load_lds`): s_barrier does not wait for a wave's outstanding DMA, and hipcc's own
// s_waitcnt insertion for the builtin lost the wait on a loop back-edge (round 3, assign.hip).  Every wave drains its vector
// memory queue, then arrives: past the barrier every wave's DMA issued before it has landed in LDS.
#define FAL_DMA_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

// One LDS-DMA instruction: lane l's 16 bytes at `g` (per lane) land at `lds` (wave-uniform) + 16 l.  Inline asm, not
// `__builtin_amdgcn_global_load_lds`: for a DMA it knows of hipcc either guards later LDS reads with `s_waitcnt vmcnt(0)` (the
// overlap with the next chunk's loads is gone) or, across a loop back-edge, forgets the wait altogether (see above).  With the
// asm form the compiler knows nothing; FAL_DMA_BARRIER / hand-counted vmcnt do ALL the waiting.
__device__ __forceinline__ void lds_dma16(const void* g, const void* lds) {
    const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) const void*)lds);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

}  // namespace fal
