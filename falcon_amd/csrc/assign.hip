// a6: k-means / final list assignment for buckets with few lists -- centroids resident, rows streamed ONCE
// through LDS and shared by four waves (reference spec: Faiss IVF train/add, README.md:132-136; convention
// arg-max inner product, ties -> lowest id).
//
// The row-resident form (dense_kernel<., ARGMAX>) loads a 32-row tile and then meets only n_list / 32 chunks
// of centroids -- 4 for a 128-list bucket -- so the tile load and the pipeline fill are never amortised
// (measured 65 TFLOP/s).  Here a 4-wave workgroup owns (a segment of a bucket's rows) x (up to 128 of its
// centroids): every wave keeps one tile of 32 centroids in registers for the whole segment; the rows
// arrive 32 at a time by direct global->LDS loads (`global_load_lds_dwordx4`, no staging registers) in
// MFMA-operand order -- step j of the chunk is the 1 KB line {lane l: float4 j of row l&31, k-half l>>5},
// so a wave reads its B operand with one conflict-free ds_read_b128 per step -- double-buffered: chunk
// c+1 lands while the 200 MFMAs of chunk c run.  HBM sees every row once per 128 centroids.
// The inner product of a (row, centroid) pair is the same k-ordered fmaf chain as in every other cosine
// kernel (simtile.h), so assignments are bit-identical to the row-resident form and to the oracle.
// Arg-max: per wave over its 32 centroids in registers, across waves (and across the centroid groups of a
// bucket with more than 128 lists) by a 64-bit atomic max on (sortable sim << 32 | ~id) in global memory:
// order-independent, ties -> lowest id.
#include <math.h>
#include <algorithm>
#include "common.h"
#include "ivf.h"
#include "scan.h"
#include "simtile.h"

namespace fal {


template <int DH4>
__global__ __launch_bounds__(256, 1) void assign_kernel(const float* __restrict__ X, const float* __restrict__ Cn, int d,
                                                        const AssignJob* __restrict__ jobs, int64_t n_jobs,
                                                        unsigned long long* __restrict__ keys) {
    // two chunks of 32 rows in operand order: SEPARATE objects, each phase of the 2x-unrolled loop names its own
    // (hipcc tells an in-flight LDS-DMA into one object from ds_reads of another; with one object, or a runtime
    // buffer index, it drains the DMA queue before every chunk's first read and the overlap is gone)
    __shared__ float4 sbuf0[DH4 * 64];
    __shared__ float4 sbuf1[DH4 * 64];
    // a contiguous run of jobs per XCD (neighbouring jobs stream neighbouring rows)
    const int64_t per_xcd = (n_jobs + 7) / 8;
    const int64_t ji = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || ji >= n_jobs) return;
    const AssignJob job = jobs[ji];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int dh = d >> 1, dh4 = dh >> 2;
    const int nr = job.nrows;
    const int n_tiles = (job.ncent + 31) >> 5;
    const bool active = w < n_tiles;                        // waves without a centroid tile only help loading

    float q[DH4 * 4];            // this lane's k-half of centroid r of tile w
    {
        const int cr = min(32 * min(w, n_tiles - 1) + r, job.ncent - 1);
        load_half_row<DH4>(q, Cn + (job.cent0 + cr) * d + (int64_t)h * dh, dh4);
    }

    // chunk loader: the four waves share the DH4 steps of a chunk (wave w issues steps w, w + 4, ...)
    auto issue = [&](int c0, float4* buf) {
        const float4* rowp = reinterpret_cast<const float4*>(X + (job.row0 + min(c0 + r, nr - 1)) * d + (int64_t)h * dh);
#pragma unroll
        for (int jj = 0; jj < (DH4 + 3) / 4; ++jj) {
            const int j = 4 * jj + w;
            if (j < DH4) lds_dma16(rowp + min(j, dh4 - 1), buf + j * 64);      // padded steps re-read the last real one
        }
    };

    f32x16 prev;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = -INFINITY;
    int prev_c0 = 0;
    // epilogue of the PREVIOUS chunk, run in the middle of the current chunk's MFMAs (the first one is a dummy
    // carrying -inf: below any real similarity)
    auto epilogue = [&]() {
        float best = -INFINITY;
        int bid = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 32 * w + mfma32_row(i, h);          // ascending in i: "s > best" alone keeps the lowest id
            const float s = c < job.ncent ? prev[i] : -INFINITY;
            const bool up = s > best;
            best = up ? s : best;
            bid = up ? c : bid;
        }
        const float ob = __shfl_xor(best, 32, 64);
        const int oc = __shfl_xor(bid, 32, 64);
        const bool take = ob > best || (ob == best && oc < bid);
        best = take ? ob : best;
        bid = take ? oc : bid;
        const unsigned long long key = ((unsigned long long)f32_sortable(best) << 32) | (uint32_t)~(uint32_t)(job.id_base + bid);
        // lanes past the segment end streamed a copy of its last row, so their key IS that row's
        // (straight to global memory: an LDS atomic here makes hipcc drain the in-flight LDS-DMA first)
        if (h == 0) atomicMax(keys + job.row0 + min(prev_c0 + r, nr - 1), key);
    };

    // one chunk: 200 MFMAs against the LDS-resident rows, the previous chunk's epilogue in the middle
    auto compute = [&](const float4* buf, int c0) {
        constexpr int kRing = 4, kMid = DH4 / 2;
        const float4* sb = buf + lane;
        float4 ring[kRing];
#pragma unroll
        for (int j = 0; j < kRing; ++j) ring[j] = sb[j * 64];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int j = 0; j < DH4; ++j) {
            const float4 a = ring[j % kRing];
            if (j + kRing < DH4) ring[j % kRing] = sb[(j + kRing) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 0], a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 1], a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 2], a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 3], a.w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read, then this step's four MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j == kMid) epilogue();
        }
        prev = acc;
        prev_c0 = c0;
    };

    // A barrier does NOT wait for LDS-DMA: every wave drains its own queue (vmcnt(0): the chunk issued one phase ago, and the
    // epilogue's atomics) BEFORE it arrives, so past the barrier all four waves' parts of the chunk have landed.  Written
    // out in asm because hipcc's own tracking of `global_load_lds` dropped the wait on the loop's back-edge (round 3: a bare
    // s_barrier at the loop header, the build's rows raced their DMA; tests/test_kernel_resources.py now lints the ISA).
    issue(0, sbuf0);
    for (int c0 = 0; c0 < nr; c0 += 64) {
        FAL_DMA_BARRIER();        // chunk c0 has landed; sbuf1 is free again
        if (c0 + 32 < nr) issue(c0 + 32, sbuf1);
        if (active) compute(sbuf0, c0);
        if (c0 + 32 >= nr) break;
        FAL_DMA_BARRIER();
        if (c0 + 64 < nr) issue(c0 + 64, sbuf0);
        if (active) compute(sbuf1, c0 + 32);
    }
    if (active) epilogue();
}

// Buckets with one or two centroid tiles (n_list <= 64): a 4-wave workgroup would idle two or three of its waves.
// One wave per (row segment, 32-centroid tile) instead, rows streamed row-per-lane straight from L2 / HBM with the
// pinned load ring of simtile.h -- with so few tiles the rows are re-read at most twice.
template <int DH4>
__global__ __launch_bounds__(64, 1) void assign_wave_kernel(const float* __restrict__ X, const float* __restrict__ Cn, int d,
                                                            const AssignJob* __restrict__ jobs, int64_t n_jobs,
                                                            unsigned long long* __restrict__ keys) {
    const int64_t per_xcd = (n_jobs + 7) / 8;
    const int64_t ji = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || ji >= n_jobs) return;
    const AssignJob job = jobs[ji];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int dh = d >> 1, dh4 = dh >> 2;
    const int nr = job.nrows;
    float q[DH4 * 4];            // this lane's k-half of centroid min(r, ncent - 1) of the tile
    load_half_row<DH4>(q, Cn + (job.cent0 + min(r, job.ncent - 1)) * d + (int64_t)h * dh, dh4);
    CandStream<DH4> cs;
    const float* cur = X + (job.row0 + min(r, nr - 1)) * d + (int64_t)h * dh;
    cs.prime(cur, dh4);
    f32x16 prev;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = -INFINITY;
    int prev_c0 = 0;
    auto epilogue = [&]() {      // as in assign_kernel; every lane issues the atomic (both halves hold the same key)
        float best = -INFINITY;
        int bid = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = mfma32_row(i, h);
            const float s = c < job.ncent ? prev[i] : -INFINITY;
            const bool up = s > best;
            best = up ? s : best;
            bid = up ? c : bid;
        }
        const float ob = __shfl_xor(best, 32, 64);
        const int oc = __shfl_xor(bid, 32, 64);
        const bool take = ob > best || (ob == best && oc < bid);
        best = take ? ob : best;
        bid = take ? oc : bid;
        const unsigned long long key = ((unsigned long long)f32_sortable(best) << 32) | (uint32_t)~(uint32_t)(job.id_base + bid);
        atomicMax(keys + job.row0 + min(prev_c0 + r, nr - 1), key);
    };
    for (int c0 = 0; c0 < nr; c0 += 32) {
        const float* nxt = X + (job.row0 + min(c0 + 32 + r, nr - 1)) * d + (int64_t)h * dh;
        const f32x16 acc = cs.template dot<true>(q, cur, nxt, dh4, epilogue);
        prev = acc;
        prev_c0 = c0;
        cur = nxt;
    }
    epilogue();
}

__global__ void assign_unpack_kernel(const unsigned long long* __restrict__ keys, int64_t n, int32_t* __restrict__ assign) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        if (k) assign[i] = (int32_t)~(uint32_t)k;    // rows no job covered (flat buckets, many-list buckets) keep their value
    }
}

int launch_assign(fal_ctx* ctx, int stage, const float* X, const float* centroids, int d, const AssignJob* jobs, int64_t n_jobs,
                  int64_t n_wave_jobs, int64_t n, unsigned long long* keys, int32_t* assign) {
    // jobs[0, n_jobs): 4-wave shared-stream jobs; jobs[n_jobs, n_jobs + n_wave_jobs): one-wave jobs (one centroid tile each)
    if (n_jobs + n_wave_jobs <= 0 || n <= 0) return FAL_OK;
    const int dh4 = d / 8;
    const int64_t per_xcd = (n_jobs + 7) / 8, per_xcd_w = (n_wave_jobs + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX && per_xcd_w * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED,
                "too many assignment jobs in one launch");
    StageScope ts(ctx, stage);
    FAL_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)n, ctx->stream));
    dim3 grid((unsigned)(per_xcd * 8)), block(256), grid_w((unsigned)(per_xcd_w * 8)), block_w(64);
#define FAL_LAUNCH_ASSIGN(DH4)                                                                                          \
    do {                                                                                                                \
        if (n_jobs > 0)                                                                                                 \
            hipLaunchKernelGGL((assign_kernel<DH4>), grid, block, 0, ctx->stream, X, centroids, d, jobs, n_jobs, keys); \
        if (n_wave_jobs > 0)                                                                                            \
            hipLaunchKernelGGL((assign_wave_kernel<DH4>), grid_w, block_w, 0, ctx->stream, X, centroids, d, jobs + n_jobs, \
                               n_wave_jobs, keys);                                                                      \
    } while (0)
    if (dh4 <= 8) FAL_LAUNCH_ASSIGN(8);
    else if (dh4 <= 16) FAL_LAUNCH_ASSIGN(16);
    else if (dh4 <= 32) FAL_LAUNCH_ASSIGN(32);
    else if (dh4 <= 50) FAL_LAUNCH_ASSIGN(50);
    else if (dh4 <= 64) FAL_LAUNCH_ASSIGN(64);
    else {
        set_error("float32 assignment supports low_dim <= 512 (got %d)", d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_ASSIGN
    FAL_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(assign_unpack_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 8192)), dim3(256), 0, ctx->stream,
                       keys, n, assign);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::assign_kernel<8>);      // (fal_ctx_plan: this unit's code object is loaded up front)
