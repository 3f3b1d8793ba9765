// f16-MFMA cosine scan: flat buckets, candidates staged through LDS and shared by four waves.
//
// Serves two precisions of the SAME kernel:
//   PLANES = 1  rows are low_dim float16 (BASELINE config 5: low_dim = 800 fp16).
//   PLANES = 2  rows are the hi/lo float16 split of the float32 vectors (FAL_DTYPE_SPLIT16:
//               x ~= hi + lo/2048).  sim = hi.hi + (hi.lo + lo.hi) / 2048, three f16 MFMAs per
//               k-step with float32 accumulation: error ~3e-7 absolute on unit vectors (inside
//               north_star's 1e-5), at 3/16 of the fp32 matrix-pipe cycles.
//
// Structure (one workgroup = 4 waves = 128 queries of one bucket, 1 workgroup per CU):
//   * every wave keeps ITS 32 queries in registers for the whole tile (200 VGPRs / lane);
//   * the bucket's rows stream through a double-buffered LDS tile of 32 rows (row stride padded
//     by 16 B: conflict-free ds_read_b128), filled with fully coalesced 16 B/lane global loads
//     issued one chunk ahead (registers -> LDS after the MFMAs of the current chunk);
//   * v_mfma_f32_32x32x16_f16, operands: lane (r, h) supplies row r, k = h*d/2 + 8*step + 0..7
//     for both sides (same slot->k map on both operands => plain inner product);
//   * results go to the same [32-query tile][ceil32(nc)] sims layout the fp32 kernel writes, so
//     the select kernel does not care which scan produced them.
#include <hip/hip_fp16.h>
#include <math.h>
#include "common.h"
#include "scan.h"

namespace fal {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// XCD-list lookup with 128-query tiles (see find_job_xcd in simtile.h)
__device__ __forceinline__ bool find_job_xcd128(const DenseJob* __restrict__ jobs, int n_jobs, unsigned bid,
                                                int* job_index, int* local_tile) {
    const int x = bid & 7;
    const int64_t i = bid >> 3;
    const int cnt = (n_jobs - x + 7) >> 3;
    if (cnt <= 0) return false;
    int lo = 0, hi = cnt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[x + 8 * mid].xtile0 <= i) lo = mid; else hi = mid - 1;
    }
    const DenseJob& j = jobs[x + 8 * lo];
    const int64_t lt = i - j.xtile0;
    if (lt >= (j.nq + 127) / 128) return false;
    *job_index = x + 8 * lo;
    *local_tile = (int)lt;
    return true;
}

template <int STEPS, int PLANES>
__global__ __launch_bounds__(256, 1) void scan16_kernel(const __half* __restrict__ Xs, const DenseJob* __restrict__ jobs,
                                                        int n_jobs, float* __restrict__ sims, int64_t sims_base,
                                                        float* __restrict__ sink) {
    // every geometry value is a compile-time constant: static register indexing, cheap address math
    constexpr int D = STEPS * 16;                   // low_dim
    constexpr int DH = D / 2;
    constexpr int ROW_HALVES = PLANES * D;          // halves per row
    constexpr int RB16 = ROW_HALVES / 8;            // 16-byte pieces per row
    constexpr int RS = ROW_HALVES * 2 + 16;         // LDS row stride in bytes (padded: conflict-free b128 reads)
    constexpr int PIECES = 32 * RB16;               // per 32-row chunk
    constexpr int kStage = (PIECES + 255) / 256;    // 16-B pieces per thread per chunk
    constexpr int NB = STEPS < 8 / PLANES ? STEPS : 8 / PLANES;   // LDS operand ring: steps in flight ahead of the MFMAs
    extern __shared__ __align__(16) unsigned char lds[];
    int ji, T;
    if (!find_job_xcd128(jobs, n_jobs, blockIdx.x, &ji, &T)) return;
    const DenseJob job = jobs[ji];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int nc = job.nc, ncp = (nc + 31) & ~31;
    const int tile32 = 4 * T + w;                   // this wave's 32-query tile inside the job
    const int nqw = min(32, job.nq - 32 * tile32);  // <= 0: the wave only helps with staging
    // Waves without queries (last tile of a bucket) still stage candidates and run the same
    // instruction stream; their results go to a scratch row at the end of the buffer.  Keeping the
    // loop free of branches lets the compiler count loads / stores / LDS reads exactly.
    const bool active = nqw > 0;

    // ---- queries -> registers ---------------------------------------------------------------
    half8 q[PLANES][STEPS];
    {
        const int64_t qrow = active ? job.q_row0 + 32 * (int64_t)tile32 + min(r, nqw - 1) : job.q_row0;
#pragma unroll
        for (int p = 0; p < PLANES; ++p) {
            const half8* src = reinterpret_cast<const half8*>(Xs + qrow * ROW_HALVES + p * D + h * DH);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) q[p][s] = src[s];
        }
    }
    float* out = active ? sims + (job.obase - sims_base) + (int64_t)(32 * tile32) * ncp + r : sink + lane;
    const int ostride = active ? ncp : 0;
    // Symmetry (queries == candidates of the same bucket; sim(i,j) == sim(j,i) bit for bit because both
    // are the same products summed in the same order): the workgroup only visits chunks at or above
    // its first query tile, each wave only computes chunks at or above its own tile and writes every
    // off-diagonal block a second time, transposed.
    const int c_first_wg = 128 * T;                 // first chunk (candidate row) the workgroup needs
    const int c_first = 32 * tile32;                // first chunk this wave computes
    float* outT = sims + (job.obase - sims_base) + (int64_t)r * ncp + 32 * tile32 + 4 * h;   // used only when `mine`
    const int tstride = active ? ncp : 0;

    // ---- candidate staging: chunk c -> registers (coalesced 16 B/lane) -> LDS buffer c & 1 -----
    const __half* cbase = Xs + job.c_row0 * (int64_t)ROW_HALVES;
    // 13 named registers instead of an array: a long-lived array is demoted to scratch memory by hipcc
    uint4 st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11, st12;
#define FAL_FOR_STAGE(M) M(0, st0) M(1, st1) M(2, st2) M(3, st3) M(4, st4) M(5, st5) M(6, st6) M(7, st7) M(8, st8) \
    M(9, st9) M(10, st10) M(11, st11) M(12, st12)
    static_assert(kStage <= 13, "staging registers");
#define FAL_LOAD_ONE(I, R)                                                                             \
    if constexpr (I < kStage) {                                                                        \
        const int idx = min((int)threadIdx.x + 256 * I, PIECES - 1); /* surplus threads repeat a piece */ \
        const int row = idx / RB16, col = idx - row * RB16;                                            \
        R = reinterpret_cast<const uint4*>(cbase + (int64_t)min(stage_c0 + row, nc - 1) * ROW_HALVES)[col]; \
    }
#define FAL_STORE_ONE(I, R)                                                                            \
    if constexpr (I < kStage) {                                                                        \
        const int idx = min((int)threadIdx.x + 256 * I, PIECES - 1);                                   \
        const int row = idx / RB16, col = idx - row * RB16;                                            \
        *reinterpret_cast<uint4*>(lds + (size_t)stage_buf * 32 * RS + row * RS + col * 16) = R;        \
    }
#define FAL_STAGE_LOAD(C0) { const int stage_c0 = (C0); FAL_FOR_STAGE(FAL_LOAD_ONE) }
#define FAL_STAGE_STORE(BUF) { const int stage_buf = (BUF); FAL_FOR_STAGE(FAL_STORE_ONE) }
    // chunks are walked from the last one DOWN to the workgroup's first (tiles of a bucket running
    // together then read the same chunk at the same time and stop at their own diagonal)
    const int c_last = ((nc - 1) >> 5) << 5;
    FAL_STAGE_LOAD(c_last)
    FAL_STAGE_STORE(0)
    __syncthreads();

    int buf = 0;
    for (int c0 = c_last; c0 >= c_first_wg; c0 -= 32) {
        FAL_STAGE_LOAD(max(c0 - 32, 0))             // next chunk, in flight during the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, kStage, 0);              // ... so issue them FIRST
        const unsigned char* rowp = lds + (size_t)buf * 32 * RS + r * RS + h * DH * 2;
        half8 rh[NB], rl[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            rh[s] = *reinterpret_cast<const half8*>(rowp + s * 16);
            if (PLANES == 2) rl[s] = *reinterpret_cast<const half8*>(rowp + D * 2 + s * 16);
        }
        // pin the whole operand ring in front of the first MFMA (otherwise the scheduler issues the reads one step ahead
        // only and every MFMA waits for an LDS round trip)
        __builtin_amdgcn_sched_group_barrier(0x100, NB * PLANES, 0);
        f32x16 acc_hh, acc_x;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc_hh[i] = 0.f;
            acc_x[i] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const half8 ch = rh[s % NB];
            const half8 cl = rl[s % NB];
            if (s + NB < STEPS) {
                rh[s % NB] = *reinterpret_cast<const half8*>(rowp + (s + NB) * 16);
                if (PLANES == 2) rl[s % NB] = *reinterpret_cast<const half8*>(rowp + D * 2 + (s + NB) * 16);
            }
            acc_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[0][s], ch, acc_hh, 0, 0, 0);
            if (PLANES == 2) {
                acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[0][s], cl, acc_x, 0, 0, 0);
                acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[PLANES - 1][s], ch, acc_x, 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, PLANES, 0);          // LDS reads of step s + NB
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * PLANES - 1, 0);  // MFMAs of step s
        }
        // D[query][candidate]: this lane owns candidate c0 + r, registers = 16 query rows.
        // Chunks below this wave's diagonal are another tile's job: their (computed) values go to the sink.
        const bool mine = active && c0 >= c_first;
        float* o = mine ? out : sink + lane;
        float* ot = mine ? outT + (int64_t)c0 * tstride : sink + 4 * lane;      // 16-byte aligned either way
        const int tg = mine ? 8 : 0;
        const int os = mine ? ostride : 0, ocol = mine ? c0 : 0;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = (PLANES == 2) ? acc_hh[i] + acc_x[i] * (1.0f / 2048.0f) : acc_hh[i];
            o[mfma32_row(i, h) * os + ocol] = v[i];
        }
        // transposed copy: registers 4g .. 4g+3 are four consecutive columns (query rows 8g + 4h + 0..3)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(ot + tg * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        __builtin_amdgcn_sched_group_barrier(0x040, 20, 0);
        FAL_STAGE_STORE(buf ^ 1)                    // buffer buf^1 was last read before the previous barrier
        __builtin_amdgcn_sched_group_barrier(0x200, kStage, 0);
        __syncthreads();
        buf ^= 1;
    }
#undef FAL_STAGE_LOAD
#undef FAL_STAGE_STORE
#undef FAL_LOAD_ONE
#undef FAL_STORE_ONE
#undef FAL_FOR_STAGE
}

int launch_scan16(fal_ctx* ctx, int planes, const void* Xs, int d, const DenseJob* jobs, int n_jobs, int64_t list_tiles,
                  float* sims, int64_t sims_base, float* sink) {
    if (n_jobs <= 0 || list_tiles <= 0) return FAL_OK;
    FAL_REQUIRE(d % 16 == 0, FAL_EUNSUPPORTED, "f16 scan needs low_dim %% 16 == 0 (got %d)", d);
    const int steps = d / 16;
    FAL_REQUIRE(steps * planes <= 50, FAL_EUNSUPPORTED, "f16 scan supports low_dim*planes <= 800 (got %d x %d)", d, planes);
    FAL_REQUIRE(list_tiles * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    const size_t lds = (size_t)2 * 32 * ((size_t)planes * d * 2 + 16);
    dim3 grid((unsigned)(list_tiles * 8)), block(256);
    const __half* X = reinterpret_cast<const __half*>(Xs);
    StageScope ts(ctx, ST_SCAN);
    StageScope tk(ctx, ST_KERNEL);
#define FAL_LAUNCH16(S, P)                                                                                        \
    do {                                                                                                          \
        FAL_CHECK_HIP(hipFuncSetAttribute((const void*)scan16_kernel<S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds));                                                             \
        hipLaunchKernelGGL((scan16_kernel<S, P>), grid, block, lds, ctx->stream, X, jobs, n_jobs, sims, sims_base, sink); \
    } while (0)
    // the kernel is instantiated for exact step counts (no padded k-steps): low_dim in {64,128,256,400,512,800}
    bool ok = true;
    if (planes == 2) {
        switch (steps) {
            case 4: FAL_LAUNCH16(4, 2); break;
            case 8: FAL_LAUNCH16(8, 2); break;
            case 16: FAL_LAUNCH16(16, 2); break;
            case 25: FAL_LAUNCH16(25, 2); break;
            default: ok = false;
        }
    } else {
        switch (steps) {
            case 4: FAL_LAUNCH16(4, 1); break;
            case 8: FAL_LAUNCH16(8, 1); break;
            case 16: FAL_LAUNCH16(16, 1); break;
            case 25: FAL_LAUNCH16(25, 1); break;
            case 32: FAL_LAUNCH16(32, 1); break;
            case 50: FAL_LAUNCH16(50, 1); break;
            default: ok = false;
        }
    }
    FAL_REQUIRE(ok, FAL_EUNSUPPORTED, "f16 scan: low_dim %d with %d plane(s) has no instantiation", d, planes);
#undef FAL_LAUNCH16
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::scan16_kernel<25, 1>);      // (fal_ctx_plan: this unit's code object is loaded up front)
