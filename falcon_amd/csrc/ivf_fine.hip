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
// that probe it.  No LDS, no masks beyond the list tail, MFMA waste = padding of the list length
// to 32 only.
// ------------------------------------------------------------------------------------------------
template <int DH4>
__global__ __launch_bounds__(64, 1) void ivf_list_kernel(ListScanArgs a) {
    // One wave = one 32-row slice of one list, RESIDENT in registers, against the stream of all
    // queries that probe the list (gathered rows, 32 per chunk).  A list is probed by ~n_probe/n_list
    // of the bucket (~1,000 queries = ~34 chunks), so the prologue is amortised like in the flat scan.
    // contiguous run of tiles per XCD: a list's slices and neighbouring lists share query rows in L2
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    const int64_t lt = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd) return;
    const int64_t t = a.tile_begin + lt;
    if (t >= a.ltile_off[a.list_end]) return;
    int64_t lo = a.list_begin, hi = a.list_end - 1;       // last list with ltile_off <= t
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (a.ltile_off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t L = lo;
    const int slice = (int)(t - a.ltile_off[L]);          // 32-row slice of the list
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int d = a.d, dh = d >> 1, dh4 = dh >> 2;
    const int64_t l_row0 = a.list_off[L];
    const int l_rows = (int)(a.list_off[L + 1] - l_row0);
    const int nrow = min(32, l_rows - 32 * slice);        // rows of this slice
    const int64_t e0 = a.inv_off[L];
    const int nq = (int)(a.inv_off[L + 1] - e0);          // queries probing the list
    if (nrow <= 0 || nq <= 0) return;

    float q[DH4 * 4];                                      // the resident operand: list row 32*slice + r
    load_half_row<DH4>(q, a.Xl + (l_row0 + 32 * slice + min(r, nrow - 1)) * d + (int64_t)h * dh, dh4);
    const int pos = 32 * slice + r;                        // position inside the list = offset inside a query's segment
    const bool rvalid = r < nrow;

    CandStream<DH4> cs;                                    // the stream: gathered query rows
    auto qrow = [&](int c0) -> const float* {
        const int64_t e = e0 + min(c0 + r, nq - 1);
        return a.Xl + (int64_t)a.inv_q[e] * d + (int64_t)h * dh;
    };
    const float* cur = qrow(0);
    cs.prime(cur, dh4);
    f32x16 prev;
    int prev_c0 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = 0.f;
    // D[query][list row]: lane = list row (column), registers = 16 streamed queries; every store
    // instruction writes 32 consecutive floats of ONE query's segment for this list
    auto epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qi = prev_c0 + mfma32_row(i, h);
            const int64_t dest = a.inv_dest[e0 + min(qi, nq - 1)] - a.sims_base;
            float* p = (rvalid && qi < nq) ? a.sims + dest + pos : a.sink + lane;
            *p = prev[i];
        }
        __builtin_amdgcn_sched_group_barrier(0x040, 16, 0);
    };
    for (int c0 = 0; c0 < nq; c0 += 32) {
        const float* nxt = qrow(c0 + 32);
        // resident rows are the B operand: D[streamed query][list row], column = lane & 31 = list row
        const f32x16 acc = cs.template dot<false>(q, cur, nxt, dh4, epilogue);
        prev = acc;
        prev_c0 = c0;
        cur = nxt;
    }
    epilogue();
}

int launch_list_scan(fal_ctx* ctx, const ListScanArgs& a) {
    if (a.n_tiles_max <= 0) return FAL_OK;
    const int dh4 = a.d / 8;
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    dim3 grid((unsigned)(per_xcd * 8)), block(64);
    StageScope ts(ctx, ST_SCAN);
#define FAL_LAUNCH_LIST(DH4) hipLaunchKernelGGL(ivf_list_kernel<DH4>, grid, block, 0, ctx->stream, a)
    if (dh4 <= 8) FAL_LAUNCH_LIST(8);
    else if (dh4 <= 16) FAL_LAUNCH_LIST(16);
    else if (dh4 <= 32) FAL_LAUNCH_LIST(32);
    else if (dh4 <= 50) FAL_LAUNCH_LIST(50);
    else if (dh4 <= 64) FAL_LAUNCH_LIST(64);
    else {
        set_error("float32 scan supports low_dim <= 512 (got %d)", a.d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_LIST
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
