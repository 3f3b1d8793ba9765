// a7 fine scan for IVF buckets: every query against the vectors of its n_probe lists.
//
// One wave = 32 consecutive queries in (bucket, list, row) order.  Neighbouring queries sit in
// the same inverted list and probe nearly the same lists, so the wave scans the UNION of its
// queries' probe sets once (rows gathered straight from L2 into VGPRs, one candidate row per
// lane, see simtile.h) and each query keeps only the candidates of lists it really probes:
// the result is exactly the per-query n_probe search, at ~1/32 of the row traffic.
//
// Per-wave LDS: a bitmap over the bucket's lists (union), the union's list ids and row
// prefix, and the transposed per-query probe / prefix tables.
#include <math.h>
#include "common.h"
#include "scan.h"

namespace fal {

__device__ __forceinline__ int wave_excl_scan_i32(int v, int lane, int* total) {
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

template <int DH4>
__global__ __launch_bounds__(64, 1) void ivf_fine_kernel(FineArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int np = a.n_probe;
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(smem);                 // bm_words
    int32_t* ulist = reinterpret_cast<int32_t*>(bitmap + a.bm_words);     // u_cap
    int32_t* uoff = ulist + a.u_cap;                                      // u_cap + 1
    int32_t* pq = uoff + a.u_cap + 1;                                     // [np][32] probed list
    int32_t* preq = pq + np * 32;                                         // [np][32] stream prefix

    const int64_t per_xcd = (a.n_tiles + 7) / 8;
    const int64_t lt_all = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || lt_all >= a.n_tiles) return;
    const int64_t t = a.tile_begin + lt_all;
    const DenseJob job = a.jobs[find_job(a.jobs, a.n_jobs, t)];
    const int lt = (int)(t - job.tile0);
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nq_t = min(32, job.nq - 32 * lt);
    const int d = a.d, dh = d >> 1, dh4 = dh >> 2;
    const int64_t p0 = job.q_row0 + 32 * (int64_t)lt;     // first query (list-order position)
    const int64_t list0 = job.c_row0;                     // global id of the bucket's list 0
    const int n_list = job.nc;
    const int64_t* __restrict__ loff = a.list_off + list0;

    float q[DH4 * 4];
    load_half_row<DH4>(q, a.Xl + (p0 + min(r, nq_t - 1)) * d + (int64_t)h * dh, dh4);

    // ---- union of the probe sets ---------------------------------------------------------
    const int words = (n_list + 31) >> 5;
    for (int w = lane; w < words; w += 64) bitmap[w] = 0;
    __syncthreads();
    const bool qvalid = r < nq_t;
    for (int j = h; j < np; j += 2) {
        int l = -1;
        if (qvalid) l = a.probes[(p0 + r) * np + j];
        pq[j * 32 + r] = l;
        int sz = 0;
        if (l >= 0) {
            atomicOr(&bitmap[l >> 5], 1u << (l & 31));
            sz = (int)(loff[l + 1] - loff[l]);
        }
        preq[j * 32 + r] = sz;
    }
    __syncthreads();
    if (h == 0) {   // exclusive prefix over the probe order, per query
        int run = 0;
        for (int j = 0; j < np; ++j) {
            const int sz = preq[j * 32 + r];
            preq[j * 32 + r] = run;
            run += sz;
        }
    }
    // enumerate set bits -> ulist (ascending) ; sizes -> uoff
    int U = 0;
    for (int w0 = 0; w0 < words; w0 += 64) {
        const int w = w0 + lane;
        uint32_t bits = w < words ? bitmap[w] : 0u;
        int tot;
        int at = U + wave_excl_scan_i32(__popc(bits), lane, &tot);
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            if (at < a.u_cap) ulist[at] = (w << 5) + b;
            ++at;
        }
        U += tot;
    }
    U = min(U, a.u_cap);   // cannot exceed 32 * n_probe by construction
    __syncthreads();
    int R = 0;
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        int sz = 0;
        if (u < U) {
            const int l = ulist[u];
            sz = (int)(loff[l + 1] - loff[l]);
        }
        int tot;
        const int ex = wave_excl_scan_i32(sz, lane, &tot);
        if (u < U) uoff[u] = R + ex;
        R += tot;
    }
    if (lane == 0) uoff[U] = R;
    __syncthreads();

    const int64_t simrow = qvalid ? (a.q_sim_off[32 * t + r] - a.sims_base) : 0;   // tile-order slot
    float* __restrict__ orow = a.sims + simrow;

    // ---- stream the union's rows, 32 per step ----------------------------------------------
    int ucur = 0;   // wave-uniform: list containing stream row c0
    // this lane's candidate row (as the MFMA A operand) of the chunk starting at stream row c0
    auto row_ptr = [&](int c0, int uhint) -> const float* {
        const int s = min(c0 + r, R - 1);
        int u = uhint;
        while (u + 1 < U && uoff[u + 1] <= s) ++u;
        const int64_t crow = loff[ulist[u]] + (s - uoff[u]);
        return a.Xl + crow * d + (int64_t)h * dh;
    };
    CandStream<DH4> cs;
    const float* cur = R > 0 ? row_ptr(0, 0) : a.Xl;
    if (R > 0) cs.prime(cur, dh4);
    for (int c0 = 0; c0 < R; c0 += 32) {
        while (ucur + 1 < U && uoff[ucur + 1] <= c0) ++ucur;
        const float* nxt = (c0 + 32 < R) ? row_ptr(c0 + 32, ucur) : cur;
        const f32x16 acc = cs.template dot<false>(q, cur, nxt, dh4, [] {});
        cur = nxt;
        // epilogue: lane = query r; registers = candidates c0 + mfma32_row(i, h)
        const int cend = min(c0 + 32, R);
        for (int us = ucur; us < U && uoff[us] < cend; ++us) {      // wave-uniform segment loop
            const int lo = max(c0, uoff[us]), hi = min(cend, uoff[us + 1]);
            if (lo >= hi) continue;
            const int l = ulist[us];
            int dest = -1;
            for (int j = 0; j < np; ++j)
                if (pq[j * 32 + r] == l) dest = preq[j * 32 + r];
            if (dest >= 0) {
                const int shift = dest - uoff[us];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int sc = c0 + mfma32_row(i, h);
                    if (sc >= lo && sc < hi) orow[sc + shift] = acc[i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// List-major fine scan.  The union-of-probes kernel above only pays off when neighbouring queries
// probe the same lists; on hashed spectra they do not (measured: 32 neighbouring queries probe 101 of
// 128 lists), so the production path inverts the probe table instead: for every list, the queries
// that probe it.  One wave = one list x 32 of those queries: the 32 query rows are GATHERED into
// registers (row-per-lane loads), the list's rows stream through the load ring, and every lane
// (= query) stores its 16 results per chunk at its own precomputed destination.  No LDS, no masks
// beyond the list tail, MFMA waste = padding of the list length to 32 only.
// ------------------------------------------------------------------------------------------------
template <int DH4>
__global__ __launch_bounds__(64, 1) void ivf_list_kernel(ListScanArgs a) {
    // contiguous run of tiles per XCD: tiles cost about the same, and a list's tiles share its rows in L2
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    const int64_t lt = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd) return;
    const int64_t t = a.tile_begin + lt;
    if (t >= a.ltile_off[a.list_end]) return;
    // last list with ltile_off <= t
    int64_t lo = a.list_begin, hi = a.list_end - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (a.ltile_off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t L = lo;
    const int64_t it = t - a.ltile_off[L];                 // tile inside the list
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int d = a.d, dh = d >> 1, dh4 = dh >> 2;
    const int64_t e0 = a.inv_off[L] + 32 * it;
    const int nq = (int)min<int64_t>(32, a.inv_off[L + 1] - e0);
    const int64_t c_row0 = a.list_off[L];
    const int nc = (int)(a.list_off[L + 1] - c_row0);
    if (nq <= 0 || nc <= 0) return;

    const bool qvalid = r < nq;
    const int64_t e = e0 + min(r, nq - 1);
    float q[DH4 * 4];
    load_half_row<DH4>(q, a.Xl + (int64_t)a.inv_q[e] * d + (int64_t)h * dh, dh4);
    float* orow = qvalid ? a.sims + (a.inv_dest[e] - a.sims_base) : a.sink + lane;
    const int ovalid = qvalid ? nc : 0;                   // candidates >= ovalid go to the sink

    CandStream<DH4> cs;
    const float* cur = a.Xl + (c_row0 + min(r, nc - 1)) * d + (int64_t)h * dh;
    cs.prime(cur, dh4);
    f32x16 prev;
    int prev_c0 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = 0.f;
    // lane = query, registers = 16 candidates of the chunk (D[cand][query]); masked slots -> sink
    auto epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = prev_c0 + mfma32_row(i, h);
            float* p = c < ovalid ? orow + c : a.sink + lane;
            *p = prev[i];
        }
        __builtin_amdgcn_sched_group_barrier(0x040, 16, 0);
    };
    for (int c0 = 0; c0 < nc; c0 += 32) {
        const float* nxt = a.Xl + (c_row0 + min(c0 + 32 + r, nc - 1)) * d + (int64_t)h * dh;
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

int launch_fine(fal_ctx* ctx, const FineArgs& a) {
    if (a.n_tiles <= 0) return FAL_OK;
    const int dh4 = a.d / 8;
    const int64_t per_xcd = (a.n_tiles + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    const size_t lds = sizeof(uint32_t) * a.bm_words + sizeof(int32_t) * (2 * (size_t)a.u_cap + 1) +
                       sizeof(int32_t) * 64 * (size_t)a.n_probe;
    FAL_REQUIRE(lds <= 160 * 1024, FAL_EUNSUPPORTED, "IVF fine scan needs %zu B of LDS (n_probe %d, lists %d): too large",
                lds, a.n_probe, a.bm_words * 32);
    dim3 grid((unsigned)(per_xcd * 8)), block(64);
    StageScope ts(ctx, ST_SCAN);
#define FAL_LAUNCH_FINE(DH4)                                                                                 \
    do {                                                                                                     \
        if (lds > 64 * 1024)                                                                                 \
            FAL_CHECK_HIP(hipFuncSetAttribute((const void*)ivf_fine_kernel<DH4>,                             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(ivf_fine_kernel<DH4>, grid, block, lds, ctx->stream, a);                          \
    } while (0)
    if (dh4 <= 8) FAL_LAUNCH_FINE(8);
    else if (dh4 <= 16) FAL_LAUNCH_FINE(16);
    else if (dh4 <= 32) FAL_LAUNCH_FINE(32);
    else if (dh4 <= 50) FAL_LAUNCH_FINE(50);
    else if (dh4 <= 64) FAL_LAUNCH_FINE(64);
    else {
        set_error("float32 scan supports low_dim <= 512 (got %d)", a.d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_FINE
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
