// a7 fine scan for IVF buckets: every query against the vectors of its n_probe lists.
#include <math.h>
#include "common.h"
#include "scan.h"

namespace fal {

// ------------------------------------------------------------------------------------------------
// List-major fine scan.  A query-major kernel that scans the UNION of the probe sets of 32
// neighbouring queries was built first and dropped: on hashed spectra neighbouring queries do not
// probe the same lists (measured: 32 neighbours probe 101 of 128 lists = 6.3x wasted MFMA work).
// The production path inverts the probe table instead (search.hip): for every list, the queries
// that probe it.  No masks beyond the list tail, MFMA waste = padding of the list length to 128.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Shared-stream form (production): a 4-wave workgroup owns up to four 32-row slices of ONE list -- one slice
// resident per wave -- and the queries probing the list pass through LDS once for all of them
// (`global_load_lds_dwordx4` into MFMA-operand order, double-buffered; structure and the hipcc traps it
// avoids: assign.hip).  A typical list (~70 rows = 3 slices) used to stream its ~1,000 gathered query rows
// three times from L2/HBM, one wave per slice; here once.  Same fmaf chain, same stores: bit-identical sims.
// Tiles of this kernel = groups of 4 slices (ListScanArgs::group_shift = 7).
// ------------------------------------------------------------------------------------------------

// MODE 1 / 2 (rows of 513..800 columns, `--low_dim` 401..800 in float32): one launch per K-HALF, as in dense4_kernel (scan.hip)
// -- a.d = the columns one pass sees (half the row), the physical rows are 2 a.d floats wide, pass k_off = 0 takes the columns
// [0, d/2) + [d, 3d/2), pass k_off = d/2 the rest and (MODE 2) starts every block from the sums the first pass stored.
template <int DH4, int MODE>
__global__ __launch_bounds__(256, 1) void ivf_list4_kernel(ListScanArgs a, int k_off) {
    __shared__ float4 sbuf0[DH4 * 64];       // separate objects, named per phase of the 2x-unrolled loop (assign.hip)
    __shared__ float4 sbuf1[DH4 * 64];
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    const int64_t lt = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd) return;
    const int64_t t = a.tile_begin + lt;
    if (t >= a.ltile_off[a.list_end]) return;
    int64_t lo = a.list_begin, hi = a.list_end - 1;       // last list with ltile_off <= t
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (a.ltile_off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t L = lo;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int slice = 4 * (int)(t - a.ltile_off[L]) + w;  // this wave's 32-row slice of the list
    const int d = a.d, dh = d >> 1, dh4 = dh >> 2;
    const int rstride = MODE == 0 ? d : 2 * d;            // floats between two rows of Xl
    const int hoff = (MODE == 0 ? dh : d) * h + k_off;    // where this lane's k-half of a row starts
    const int64_t l_row0 = a.list_off[L];
    const int l_rows = (int)(a.list_off[L + 1] - l_row0);
    const int nrow = min(32, l_rows - 32 * slice);        // rows of this slice (<= 0: the wave only helps loading)
    const int64_t e0 = a.inv_off[L];
    const int nq = (int)(a.inv_off[L + 1] - e0);          // queries probing the list
    if (nq <= 0 || l_rows <= 0) return;
    const bool active = nrow > 0;

    float q[DH4 * 4];                                      // the resident operand: list row 32*slice + r
    load_half_row<DH4>(q, a.Xl + (l_row0 + min(32 * slice + min(r, max(nrow, 1) - 1), l_rows - 1)) * rstride + hoff, dh4);
    const int pos = 32 * slice + r;                        // position inside the list = offset inside a query's segment
    const bool rvalid = active && r < nrow;

    // per-lane metadata of a chunk's query r: its row in Xl and where its sims for this list start.  Ordinary loads,
    // always issued one iteration before their first use (which then sits behind a barrier: no extra wait)
    auto q_row = [&](int c0) -> int32_t { return a.inv_q[e0 + min(c0 + r, nq - 1)]; };
    auto q_dest = [&](int c0) -> int64_t { return a.inv_dest[e0 + min(c0 + r, nq - 1)]; };     // raw: no arithmetic on it yet
    // the first USE of a prefetched value must sit behind the next barrier (which drains the memory queue anyway):
    // touched earlier, hipcc waits vmcnt(0) on the spot and the LDS-DMA just issued drains with it
    auto pin = [](int32_t& x, int64_t& y) { asm volatile("" : "+v"(x), "+v"(y)); };
    auto issue = [&](int32_t row, float4* buf) {
        const float4* rowp = reinterpret_cast<const float4*>(a.Xl + (int64_t)row * rstride + hoff);
#pragma unroll
        for (int jj = 0; jj < (DH4 + 3) / 4; ++jj) {
            const int j = 4 * jj + w;
            if (j < DH4) lds_dma16(rowp + min(j, dh4 - 1), buf + j * 64);
        }
    };

    f32x16 prev;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = 0.f;
    int prev_c0 = 0;
    uint32_t dest_prev = (uint32_t)(q_dest(0) - a.sims_base);
    // D[query][list row]: lane = list row (column), registers = 16 streamed queries; every store instruction
    // writes 32 consecutive floats of ONE query's segment for this list.  (First call: zeros into chunk 0's slots,
    // overwritten by the real chunk-0 epilogue later in program order.)
    auto epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qr = mfma32_row(i, h);
            const uint32_t dest = (uint32_t)__shfl((int)dest_prev, qr, 64);      // lane qr holds query prev_c0 + qr
            float* p = (rvalid && prev_c0 + qr < nq) ? a.sims + dest + pos : a.sink + lane;
            *p = prev[i];
        }
        __builtin_amdgcn_sched_group_barrier(0x040, 16, 0);
    };
    auto compute = [&](const float4* buf, int c0) {
        constexpr int kRing = 4, kMid = DH4 / 2;
        const float4* sb = buf + lane;
        float4 ring[kRing];
#pragma unroll
        for (int j = 0; j < kRing; ++j) ring[j] = sb[j * 64];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (MODE == 2) {
            // the sums of the first K-half: this chunk's own slots (read before the previous chunk's epilogue below stores
            // anything; the staged path is not the timed one -- the wait for these loads drains the row DMA just issued)
            const uint32_t dcur = (uint32_t)(q_dest(c0) - a.sims_base);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qr = mfma32_row(i, h);
                const uint32_t dest = (uint32_t)__shfl((int)dcur, qr, 64);
                if (rvalid && c0 + qr < nq) acc[i] = a.sims[dest + pos];
            }
        }
#pragma unroll
        for (int j = 0; j < DH4; ++j) {
            const float4 s = ring[j % kRing];
            if (j + kRing < DH4) ring[j % kRing] = sb[(j + kRing) * 64];
            // streamed queries are the A operand, the resident list rows B: D[query][list row]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.x, q[4 * j + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.y, q[4 * j + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.z, q[4 * j + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s.w, q[4 * j + 3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j == kMid) epilogue();
        }
        prev = acc;
        prev_c0 = c0;
    };

    issue(q_row(0), sbuf0);
    int32_t row_next = q_row(32);
    int64_t dest_raw = q_dest(0);            // of the chunk whose epilogue runs next
    for (int c0 = 0; c0 < nq; c0 += 64) {
        {
            FAL_DMA_BARRIER();    // chunk c0 has landed (every wave drained its own DMA queue first); sbuf1 is free again
            pin(row_next, dest_raw);
            if (c0 + 32 < nq) issue(row_next, sbuf1);
            dest_prev = (uint32_t)(dest_raw - a.sims_base);
            row_next = q_row(c0 + 64);
            dest_raw = q_dest(c0);
            if (active) compute(sbuf0, c0);
        }
        if (c0 + 32 >= nq) break;
        {
            FAL_DMA_BARRIER();
            pin(row_next, dest_raw);
            if (c0 + 64 < nq) issue(row_next, sbuf0);
            dest_prev = (uint32_t)(dest_raw - a.sims_base);
            row_next = q_row(c0 + 96);
            dest_raw = q_dest(c0 + 32);
            if (active) compute(sbuf1, c0 + 32);
        }
    }
    if (active) {
        dest_prev = (uint32_t)(dest_raw - a.sims_base);
        epilogue();
    }
}

int launch_list_scan(fal_ctx* ctx, const ListScanArgs& a_in) {
    if (a_in.n_tiles_max <= 0) return FAL_OK;
    ListScanArgs a = a_in;
    const bool split = a.d > 512;
    if (split) a.d /= 2;                                   // (the columns one K-half pass sees)
    const int dh4 = a.d / 8;
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    dim3 grid((unsigned)(per_xcd * 8)), block(256);
    StageScope ts(ctx, ST_SCAN);
    StageScope tk(ctx, ST_KERNEL);
#define FAL_LAUNCH_LIST(DH4) hipLaunchKernelGGL((ivf_list4_kernel<DH4, 0>), grid, block, 0, ctx->stream, a, 0)
    if (split) {
        if (a_in.d > 800 || a_in.d % 16 != 0) {
            set_error("float32 scan supports low_dim <= 512 and 513..800 in steps of 16 (got %d)", a_in.d);
            return FAL_EUNSUPPORTED;
        }
        hipLaunchKernelGGL((ivf_list4_kernel<50, 1>), grid, block, 0, ctx->stream, a, 0);
        hipLaunchKernelGGL((ivf_list4_kernel<50, 2>), grid, block, 0, ctx->stream, a, a.d / 2);
    } else if (dh4 <= 8) FAL_LAUNCH_LIST(8);
    else if (dh4 <= 16) FAL_LAUNCH_LIST(16);
    else if (dh4 <= 32) FAL_LAUNCH_LIST(32);
    else if (dh4 <= 50) FAL_LAUNCH_LIST(50);
    else FAL_LAUNCH_LIST(64);
#undef FAL_LAUNCH_LIST
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::ivf_list4_kernel<50, 0>);      // (fal_ctx_plan: this unit's code object is loaded up front)
