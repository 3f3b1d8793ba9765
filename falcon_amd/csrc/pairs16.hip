// IVF prefilter path, the exact part WITHOUT the band: only the (query, candidate) pairs that can matter are evaluated.
//
// band_kernel<., IVF> evaluates the whole precursor window of a 32-query tile on the fp32 matrix cores (240 rows at
// BASELINE configs[2]'s bucket density, 870 at configs[3]'s) although a query keeps ~5 of them: the cost grows with the
// window, the need does not.  Here:
//   window16_kernel   the precursor window [lo, hi) of every query as a range of sorted rows (binary searches inside the bucket,
//      slightly widened: the exact tolerance tests run on the candidates below);
//   kept16_kernel   16 lanes per query walk the window.  A per-query table in LDS maps a list of the bucket to where its
//      keys start in the query's key stream (or "not probed"): a window row that lies in a probed list has its key looked up --
//      stream offset of the list + the row's position inside the list -- and stays if the key is not certainly below the k-th
//      best (select16_kernel left that threshold) and the exact tolerance tests pass;
//   pairs16_kernel   the exact similarity of every kept pair by the k-ordered fmaf chain (bit-identical to the matrix-core
//      chain), one pair per lane, the pairs of a 32-query tile compacted so that the lanes are full;
//   resolve_kernel (fused.hip, ivf = 2)   thresholds on the exact values, the k-th key for ambiguous queries, sort, store.
//
// Reference: README.md:107-113, 137-142; see ivf16.hip for the error bound and the rest of the path.
#include <math.h>
#include <limits.h>
#include <algorithm>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "fused.h"
#include "ivf16.h"

namespace fal {

constexpr int32_t kNotProbed = INT_MIN;

__global__ void window16_kernel(const DenseJob* __restrict__ jobs, int n_jobs, int64_t n_tiles, const float* __restrict__ pmz,
                                double tol, int is_da, int2* __restrict__ win) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t t = g >> 5;
    if (t >= n_tiles) return;
    const DenseJob job = jobs[find_job(jobs, n_jobs, t)];
    const int ql = 32 * (int)(t - job.tile0) + (int)(g & 31);
    if (ql >= job.nq) return;
    const int64_t b0 = job.q_row0, b1 = job.q_row0 + job.nq;    // the bucket's sorted rows
    const int64_t row = b0 + ql;
    const double q = (double)pmz[row];
    double lob, hib;
    if (is_da) {
        lob = q - tol - 1e-3;
        hib = q + tol + 1e-3;
    } else {
        const double tt = tol * 1e-6;
        lob = q * (1.0 - 1.01 * tt - 2e-6);
        hib = tt < 0.5 ? q * (1.0 + 1.01 * tt / (1.0 - tt) + 2e-6) : INFINITY;
    }
    int64_t lo = b0, hi = row;                                   // first row with pmz >= lob (rows are sorted by precursor m/z)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((double)pmz[mid] < lob) lo = mid + 1; else hi = mid;
    }
    const int64_t wlo = lo;
    lo = row;
    hi = b1;                                                     // first row with pmz > hib
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((double)pmz[mid] <= hib) lo = mid + 1; else hi = mid;
    }
    win[row] = make_int2((int)wlo, (int)lo);
}

// one workgroup = the 32 queries of a sorted-row tile, 16 lanes each
__global__ __launch_bounds__(512) void kept16_kernel(Kept16Args a) {
    extern __shared__ int32_t tab_all[];                         // [32][tab_stride]: list -> (key-stream offset - first position), or kNotProbed
    const int tid = threadIdx.x, ql = tid >> 4, sub = tid & 15, lane = tid & 63, sh = 16 * (lane >> 4);
    const int64_t t = a.tile_begin + blockIdx.x;
    const DenseJob job = a.jobs[a.tile_job[blockIdx.x]];
    const int lt = (int)(t - job.tile0);
    const int nl = job.nc, np = a.n_probe;
    const int64_t row0 = job.q_row0, lbase = job.c_row0;
    const bool live = 32 * lt + ql < job.nq;
    const int64_t row = row0 + min(32 * lt + ql, job.nq - 1);
    const int64_t p = a.pos_of_row[row];                         // list-order position of the query
    int32_t* tq = tab_all + ql * a.tab_stride;
    for (int i = sub; i < nl; i += 16) tq[i] = kNotProbed;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    {
        const int32_t* pr = a.probes + p * np;
        int run = 0;
        for (int j0 = 0; j0 < np; j0 += 16) {                    // stream offset of every probed list: prefix sum inside the lane group
            const int j = j0 + sub;
            const int32_t l = j < np ? pr[j] : -1;
            int64_t b = 0, e = 0;
            if (l >= 0) {
                b = a.list_off[lbase + l];
                e = a.list_off[lbase + l + 1];
            }
            const int len = (int)(e - b);
            int incl = len;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int o = __shfl_up(incl, off, 16);
                if (sub >= off) incl += o;
            }
            if (l >= 0) tq[l] = (int32_t)((int64_t)(run + incl - len) - b);
            run += __shfl(incl, 15, 16);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int64_t lp = p - row0;                                 // the query's tile-order slot: its keys start at q_sim_off[slot]
    const uint16_t* krow = a.keys + (a.q_sim_off[32 * (job.tile0 + (lp >> 5)) + (lp & 31)] - a.keys_base);
    const int2 sel = a.gsel[row];
    const int2 wn = a.win[row];
    const float qmz = a.pmz[row];
    const bool use_rt = a.rt != nullptr && a.rt_tol >= 0.0;
    const float qrt = use_rt ? a.rt[row] : 0.f;
    const float tol_f = a.tol_f, rt_f = a.rt_f;
    uint32_t* gk = a.gkept_id + row * FAL_FUSED_KEEP;
    int kc = 0;
    bool amb = false;
    constexpr int U = 2;                                         // window rows per lane and step: their loads are in flight together
    for (int64_t c0 = live ? wn.x : 0; c0 < (live ? wn.y : 0); c0 += 16 * U) {
        int ls[U], ps[U];
        float mz[U], rt[U];
        bool inw[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const int64_t c = c0 + 16 * x + sub;
            inw[x] = c < wn.y;
            const int64_t cc = inw[x] ? c : row;
            ls[x] = a.assign[cc];
            ps[x] = a.pos_of_row[cc];
            mz[x] = a.pmz[cc];
            rt[x] = use_rt ? a.rt[cc] : 0.f;
        }
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const int64_t c = c0 + 16 * x + sub;
            const int32_t off1 = inw[x] ? tq[ls[x]] : kNotProbed;
            const bool member = off1 != kNotProbed && c != row;
            const int u = member ? (int)krow[(int64_t)off1 + ps[x]] + 1 : 0;
            const float diff = qmz - mz[x];                      // mass_diff(query, neighbour): the arithmetic of filter_kernel
            const float xx = a.is_da ? diff : diff / mz[x];
            bool ok = member && u >= sel.x && fabsf(xx) <= tol_f;
            if (use_rt) ok = ok && fabsf(qrt - rt[x]) <= rt_f;
            amb = amb || (ok && u <= sel.y);
            const uint32_t gm = (uint32_t)(__ballot(ok) >> sh) & 0xFFFFu;
            if (ok) {
                const int at = kc + __popc(gm & ((1u << sub) - 1u));
                if (at < FAL_FUSED_KEEP) gk[at] = (uint32_t)c;
            }
            kc += __popc(gm);
        }
    }
    const bool q_amb = ((uint32_t)(__ballot(amb) >> sh) & 0xFFFFu) != 0u;
    if (live && sub == 0) {
        a.gkcnt[row * 2] = min(kc, FAL_FUSED_KEEP / 2) | (q_amb ? 0x100 : 0) | (kc > FAL_FUSED_KEEP ? 0x200 : 0);
        a.gkcnt[row * 2 + 1] = min(max(kc - FAL_FUSED_KEEP / 2, 0), FAL_FUSED_KEEP / 2);
    }
}

// exact similarity of every kept pair: the pairs of a 32-query tile in consecutive lanes, one k-ordered fmaf chain per lane
// (bit-identical to the matrix-core chain).  A lane's candidate row is a gathered row: with a row per lane a load instruction
// touches 64 cache lines for 16 bytes each and the texture addresser, not the memory, sets the pace (94 us per chain measured).
// So the candidates' bytes are fetched COALESCED -- an instruction covers 16 rows x 64 bytes: lane L fetches piece L & 3 of the
// row of pair 16g + (L >> 2) -- and transposed through LDS (pair slots of 176 bytes: conflict-free b128 reads) into the lanes
// that own the pairs.  The query rows are shared by neighbouring lanes (a wave's 64 pairs span a dozen queries): read directly.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void pairs16_kernel(FusedArgs a, int d) {
    constexpr int kSlot = 176;                                   // bytes per pair: 64 (low half) + 16 + 64 (high half) + 32
    constexpr int U = 4;                                         // 16-byte steps of either half per batch
    __shared__ int32_t off[33];
    __shared__ __attribute__((aligned(16))) unsigned char tbuf[4][64 * kSlot];
    int ji, lt;
    if (!find_job_xcd(a.jobs32, a.n_jobs32, blockIdx.x, &ji, &lt)) return;
    const DenseJob job = a.jobs32[ji];
    const int nqw = min(32, job.nc - 32 * lt);
    const int64_t row_t = job.q_row0 + 32 * lt;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 64) {                                      // counts of the tile's queries, prefix sum by the first wave
        const int q = threadIdx.x;
        int c = 0;
        if (q < nqw) {
            const int i0 = a.gkcnt[(row_t + q) * 2], i1 = a.gkcnt[(row_t + q) * 2 + 1];
            if (((i0 | i1) & 0x200) == 0) c = (i0 & 0xFF) + (i1 & 0xFF);
        }
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (q >= o) incl += v;
        }
        if (q < 32) off[q + 1] = incl;
        if (q == 0) off[0] = 0;
    }
    __syncthreads();
    const int P = off[32];
    const int dh4 = d >> 3;
    const float4* X4 = reinterpret_cast<const float4*>(a.X);
    unsigned char* tb = tbuf[w];
    for (int i0 = 64 * w; i0 < P; i0 += 256) {                   // (wave-uniform: the lanes of a wave work together)
        const int i = i0 + lane;
        const bool live = i < P;
        const int ic = min(i, P - 1);
        int q = 0;
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) q = (q + s < 32 && off[q + s] <= ic) ? q + s : q;
        const int j = ic - off[q];
        const int64_t row = row_t + q;
        const uint32_t id = a.gkept_id[row * FAL_FUSED_KEEP + j];
        int64_t idT[4];                                          // candidate row of the pair this lane FETCHES for in instruction g
#pragma unroll
        for (int g = 0; g < 4; ++g) idT[g] = (int64_t)(uint32_t)__shfl((int)id, 16 * g + (lane >> 2), 64) * (2 * dh4);
        const float4* qp = X4 + row * (2 * dh4);
        float acc = 0.f;
        for (int j0 = 0; j0 < dh4; j0 += U) {
            const int pc = min(j0 + (lane & 3), dh4 - 1);
            float4 sl[4], sh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sl[g] = X4[idT[g] + pc];
                sh[g] = X4[idT[g] + dh4 + pc];
            }
            float4 ql[U], qh[U];
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const int jj = min(j0 + t, dh4 - 1);
                ql[t] = qp[jj];
                qh[t] = qp[dh4 + jj];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* slot = tb + (16 * g + (lane >> 2)) * kSlot + (lane & 3) * 16;
                *reinterpret_cast<float4*>(slot) = sl[g];
                *reinterpret_cast<float4*>(slot + 80) = sh[g];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const unsigned char* mine = tb + lane * kSlot;
#pragma unroll
            for (int t = 0; t < U; ++t) {
                if (j0 + t < dh4) {                              // (wave-uniform)
                    const float4 cl = *reinterpret_cast<const float4*>(mine + t * 16);
                    const float4 ch = *reinterpret_cast<const float4*>(mine + 80 + t * 16);
                    acc = __builtin_fmaf(ql[t].x, cl.x, acc);
                    acc = __builtin_fmaf(qh[t].x, ch.x, acc);
                    acc = __builtin_fmaf(ql[t].y, cl.y, acc);
                    acc = __builtin_fmaf(qh[t].y, ch.y, acc);
                    acc = __builtin_fmaf(ql[t].z, cl.z, acc);
                    acc = __builtin_fmaf(qh[t].z, ch.z, acc);
                    acc = __builtin_fmaf(ql[t].w, cl.w, acc);
                    acc = __builtin_fmaf(qh[t].w, ch.w, acc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        if (live) a.gkept_u[row * FAL_FUSED_KEEP + j] = max(f32_sortable(acc), 1u);
    }
}

int launch_windows16(fal_ctx* ctx, const DenseJob* jobs, int n_jobs, int64_t n_tiles, const float* pmz, double tol, int is_da,
                     void* win) {
    if (n_tiles <= 0) return FAL_OK;
    StageScope ts(ctx, ST_COARSE);
    hipLaunchKernelGGL(window16_kernel, dim3((unsigned)ceil_div(n_tiles * 32, 256)), dim3(256), 0, ctx->stream, jobs, n_jobs, n_tiles,
                       pmz, tol, is_da, reinterpret_cast<int2*>(win));
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_kept16(fal_ctx* ctx, const Kept16Args& a_in, int64_t n_tiles) {
    if (n_tiles <= 0) return FAL_OK;
    Kept16Args a = a_in;
    int32_t* tj = nullptr;           // (the tile -> job table launch_select16 left in the slot for the same tiles)
    FAL_TRY(ctx->reserve(SLOT_TILEJOB, sizeof(int32_t) * 16, (void**)&tj));
    a.tile_job = tj;
    const size_t lds = sizeof(int32_t) * 32 * (size_t)a.tab_stride;
    FAL_CHECK_HIP(hipFuncSetAttribute((const void*)kept16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    StageScope ts(ctx, ST_SELECT);
    hipLaunchKernelGGL(kept16_kernel, dim3((unsigned)n_tiles), dim3(512), lds, ctx->stream, a);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_pairs16(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles32) {
    if (list_tiles32 <= 0) return FAL_OK;
    StageScope ts(ctx, ST_SCAN);
    hipLaunchKernelGGL(pairs16_kernel, dim3((unsigned)(list_tiles32 * 8)), dim3(256), 0, ctx->stream, a, d);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
