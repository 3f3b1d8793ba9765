// The cosine kernels: fp32-MFMA query x candidate inner products (dense form: k-means assign,
// coarse quantiser, flat-bucket scan) and the wavefront top-k select.
//
// Spec: reference README.md:107-113, 134-142 (Faiss IVF build / n_probe search; no code in the
// snapshot).  See simtile.h for the tile algorithm, DESIGN.md for the roofline.
#include <math.h>
#include "common.h"
#include "simtile.h"
#include "scan.h"

namespace fal {

// ---------------------------------------------------------------------------------------------
// dense tile kernel
// ---------------------------------------------------------------------------------------------
template <int DH4, int EPI>
__global__ __launch_bounds__(64, 1) void dense_kernel(
    const float* __restrict__ Q, const float* __restrict__ Cm, int d, const DenseJob* __restrict__ jobs,
    int n_jobs, int64_t tile_begin, int64_t n_tiles, float* __restrict__ sims, int64_t sims_base,
    int32_t* __restrict__ assign, int xcd_lists) {
    DenseJob job;
    int lt;
    if (xcd_lists) {
        int ji;
        if (!find_job_xcd(jobs, n_jobs, blockIdx.x, &ji, &lt)) return;
        job = jobs[ji];
    } else {
        // contiguous run of tiles per XCD (neighbouring tiles scan the same bucket -> shared L2 lines)
        const int64_t per_xcd = (n_tiles + 7) / 8;
        const int64_t lt_all = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per_xcd || lt_all >= n_tiles) return;
        const int64_t t = tile_begin + lt_all;
        job = jobs[find_job(jobs, n_jobs, t)];
        lt = (int)(t - job.tile0);
    }
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nq_t = min(32, job.nq - 32 * lt);
    const int dh = d >> 1, dh4 = dh >> 2;

    float q[DH4 * 4];
    {
        const int64_t qrow = job.q_row0 + 32 * (int64_t)lt + min(r, nq_t - 1);
        load_half_row<DH4>(q, Q + qrow * d + (int64_t)h * dh, dh4);
    }
    float best = -INFINITY;
    int bestc = 0x7fffffff;
    const int nc = job.nc;
    float* out = nullptr;
    // the [32, ncp] block of this tile (ncp = nc rounded up to 32): every store below is in bounds,
    // so the epilogue has no branches and the compiler can count its stores in vmcnt exactly
    const int ncp = (nc + 31) & ~31;
    if (EPI == EPI_STORE) out = sims + (job.obase - sims_base) + (int64_t)(32 * lt) * ncp + r;

    CandStream<DH4> cs;
    const float* cur = Cm + (job.c_row0 + min(r, nc - 1)) * d + (int64_t)h * dh;
    cs.prime(cur, dh4);
    f32x16 prev;                 // the previous chunk's result; its epilogue runs inside this chunk
    // before the first chunk "previous" is a dummy: zeros land in chunk 0's slots and are overwritten
    // by the real chunk-0 epilogue later in program order; -inf never wins the arg-max.  Keeping the
    // epilogue unconditional keeps its stores out of branches (exact vmcnt bookkeeping).
    int prev_c0 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = (EPI == EPI_STORE) ? 0.f : -INFINITY;
    auto epilogue = [&]() {
        if (EPI == EPI_STORE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) out[mfma32_row(i, h) * ncp + prev_c0] = prev[i];
            __builtin_amdgcn_sched_group_barrier(0x040, 16, 0);   // keep the 16 stores HERE in the pipeline
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = prev_c0 + mfma32_row(i, h);
                const float s = prev[i];
                if (c < nc && s > best) {
                    best = s;
                    bestc = c;
                }
            }
        }
    };
    for (int c0 = 0; c0 < nc; c0 += 32) {
        const float* nxt = Cm + (job.c_row0 + min(c0 + 32 + r, nc - 1)) * d + (int64_t)h * dh;
        const f32x16 acc = (EPI == EPI_STORE) ? cs.template dot<true>(q, cur, nxt, dh4, epilogue)
                                              : cs.template dot<false>(q, cur, nxt, dh4, epilogue);
        prev = acc;
        prev_c0 = c0;
        cur = nxt;
    }
    epilogue();
    if (EPI == EPI_ARGMAX) {
        const float ob = __shfl_xor(best, 32, 64);
        const int oc = __shfl_xor(bestc, 32, 64);
        if (ob > best || (ob == best && oc < bestc)) {
            best = ob;
            bestc = oc;
        }
        if (h == 0 && r < nq_t) assign[job.q_row0 + 32 * (int64_t)lt + r] = bestc;
    }
}

template <int EPI>
static int launch_dense_t(fal_ctx* ctx, int stage, const float* Q, const float* Cm, int d, const DenseJob* jobs,
                          int n_jobs, int64_t tile_begin, int64_t n_tiles, float* sims, int64_t sims_base,
                          int32_t* assign, int64_t xcd_list_tiles) {
    if (n_tiles <= 0) return FAL_OK;
    const int dh4 = d / 8;
    const int xcd_lists = xcd_list_tiles > 0;
    // XCD-list mode: n_tiles is unused, the grid is 8 x (longest list)
    const int64_t per_xcd = xcd_lists ? xcd_list_tiles : (n_tiles + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    dim3 grid((unsigned)(per_xcd * 8)), block(64);
    StageScope ts(ctx, stage);
#define FAL_LAUNCH_DENSE(DH4)                                                                              \
    hipLaunchKernelGGL((dense_kernel<DH4, EPI>), grid, block, 0, ctx->stream, Q, Cm, d, jobs, n_jobs,      \
                       tile_begin, n_tiles, sims, sims_base, assign, xcd_lists)
    if (dh4 <= 8) FAL_LAUNCH_DENSE(8);
    else if (dh4 <= 16) FAL_LAUNCH_DENSE(16);
    else if (dh4 <= 32) FAL_LAUNCH_DENSE(32);
    else if (dh4 <= 50) FAL_LAUNCH_DENSE(50);
    else if (dh4 <= 64) FAL_LAUNCH_DENSE(64);
    else {
        set_error("float32 scan supports low_dim <= 512 (got %d)", d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_DENSE
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_dense(fal_ctx* ctx, int stage, int epi, const float* Q, const float* Cm, int d, const DenseJob* jobs,
                 int n_jobs, int64_t tile_begin, int64_t n_tiles, float* sims, int64_t sims_base, int32_t* assign,
                 int64_t xcd_list_tiles) {
    if (epi == EPI_STORE)
        return launch_dense_t<EPI_STORE>(ctx, stage, Q, Cm, d, jobs, n_jobs, tile_begin, n_tiles, sims, sims_base,
                                         assign, xcd_list_tiles);
    return launch_dense_t<EPI_ARGMAX>(ctx, stage, Q, Cm, d, jobs, n_jobs, tile_begin, n_tiles, sims, sims_base, assign,
                                      xcd_list_tiles);
}

// ---------------------------------------------------------------------------------------------
// wavefront top-k select: one wave per query over that query's row of sims
// ---------------------------------------------------------------------------------------------
// Keys are (sortable sim, id); order = sim descending, then id ascending.  A round holds up to
// 1024 keys in registers (16 per lane): the survivors of earlier rounds plus new candidates.
// The k-th largest sim is found by a 32-step bitwise binary search whose counts are wave ballots
// (v_cmp + s_bcnt1, no LDS); boundary ties are resolved by a second search over ids.  Survivors
// are compacted into LDS by ballot-prefix ranks; the final <= k keys are rank-sorted.
constexpr int kSelRegs = 16;

__device__ __forceinline__ int wave_count(bool p) { return __popcll(__ballot(p)); }

template <int MODE>
__global__ __launch_bounds__(64) void select_kernel(SelectArgs a) {
    __shared__ uint32_t sel_u[FAL_MAX_K_ANN];
    __shared__ uint32_t sel_id[FAL_MAX_K_ANN];
    __shared__ int64_t seg_off[FAL_MAX_N_PROBE + 1];   // MODE_IVF: stream offset of each probed list
    __shared__ int64_t seg_src[FAL_MAX_N_PROBE];       // MODE_IVF: perm position of each probed list
    const int lane = threadIdx.x;
    const int k = a.k;

    // ---- locate this query ---------------------------------------------------------------
    const float* row = nullptr;   // this query's sims
    int64_t nc = 0;               // number of candidates
    int64_t out_row = 0;
    int64_t id0 = 0;
    if (MODE == MODE_DENSE) {
        const int64_t t = a.tile_begin + (blockIdx.x >> 5);
        const int ql = blockIdx.x & 31;
        const DenseJob job = a.jobs[find_job(a.jobs, a.n_jobs, t)];
        const int lt = (int)(t - job.tile0);
        if (32 * lt + ql >= job.nq) return;
        nc = job.nc;
        row = a.sims + (job.obase - a.sims_base) + (int64_t)(32 * lt + ql) * ((nc + 31) & ~31);
        out_row = job.q_row0 + 32 * (int64_t)lt + ql;
        id0 = a.ids_are_rows ? job.c_row0 : 0;
    } else {
        const int64_t t = a.tile_begin + (blockIdx.x >> 5);
        const int ql = blockIdx.x & 31;
        const DenseJob job = a.jobs[find_job(a.jobs, a.n_jobs, t)];
        const int lt = (int)(t - job.tile0);
        if (32 * lt + ql >= job.nq) return;
        const int64_t p = job.q_row0 + 32 * (int64_t)lt + ql;   // query position in list order
        const int np = a.n_probe;
        const int32_t* pr = a.probes + p * np;
        const int64_t lbase = job.c_row0;              // global id of the bucket's list 0
        if (lane == 0) {
            int64_t off = 0;
            for (int j = 0; j < np; ++j) {
                const int32_t l = pr[j];
                seg_off[j] = off;
                if (l >= 0) {
                    const int64_t b = a.list_off[lbase + l], e = a.list_off[lbase + l + 1];
                    seg_src[j] = b;
                    off += e - b;
                } else {
                    seg_src[j] = 0;
                }
            }
            seg_off[np] = off;
        }
        __syncthreads();
        nc = seg_off[np];
        row = a.sims + (a.q_sim_off[32 * t + ql] - a.sims_base);   // tile-order slot
        out_row = a.perm[p];
    }

    uint32_t u[kSelRegs], id[kSelRegs];
    int carry = 0;
    int64_t pos = 0;
    bool first = true;
    while (first || pos < nc) {
        first = false;
        // ---- fill: slot s = i*64 + lane; the first `carry` slots come from LDS ---------------
        const int64_t fresh = min<int64_t>(nc - pos, 64 * kSelRegs - carry);
#pragma unroll
        for (int i = 0; i < kSelRegs; ++i) {
            const int s = i * 64 + lane;
            u[i] = 0;
            id[i] = 0xFFFFFFFFu;
            if (s < carry) {
                u[i] = sel_u[s];
                id[i] = sel_id[s];
            } else if (s - carry < fresh) {
                const int64_t pp = pos + (s - carry);
                u[i] = max(f32_sortable(row[pp]), 1u);
                if (MODE == MODE_DENSE) {
                    id[i] = (uint32_t)(id0 + pp);
                } else {
                    int lo = 0, hi = a.n_probe - 1;      // last segment with seg_off <= pp
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (seg_off[mid] <= pp) lo = mid; else hi = mid - 1;
                    }
                    id[i] = (uint32_t)a.perm[seg_src[lo] + (pp - seg_off[lo])];
                }
            }
        }
        const int m = carry + (int)fresh;
        pos += fresh;
        __syncthreads();   // all reads of sel_* done before they are rewritten

        uint32_t T = 1, I = 0xFFFFFFFFu;
        if (m > k) {
            // k-th largest sortable sim
            T = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t c = T | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < kSelRegs; ++i) cnt += wave_count(u[i] >= c);
                if (cnt >= k) T = c;
            }
            int gt = 0, eq = 0;
#pragma unroll
            for (int i = 0; i < kSelRegs; ++i) {
                gt += wave_count(u[i] > T);
                eq += wave_count(u[i] == T);
            }
            const int need = k - gt;
            if (eq > need) {
                // smallest I with count(u == T && id <= I) >= need
                uint32_t lo = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t c = lo | (1u << bit);      // test: are there >= need ids < c ?
                    int cnt = 0;
#pragma unroll
                    for (int i = 0; i < kSelRegs; ++i) cnt += wave_count(u[i] == T && id[i] < c);
                    if (cnt < need) lo = c;
                }
                I = lo;   // largest value with count(id < I) < need  =>  count(id <= I) >= need
            }
        }
        // ---- compact survivors into LDS --------------------------------------------------
        int base = 0;
#pragma unroll
        for (int i = 0; i < kSelRegs; ++i) {
            const bool keep = (u[i] > T) || (u[i] == T && u[i] != 0 && id[i] <= I);
            const uint64_t mask = __ballot(keep);
            if (keep) {
                const int w = base + __popcll(mask & ((1ull << lane) - 1ull));
                sel_u[w] = u[i];
                sel_id[w] = id[i];
            }
            base += __popcll(mask);
        }
        carry = base;
        __syncthreads();
    }

    // ---- rank sort the survivors and write the row ---------------------------------------
    float* osim = a.out_sim + out_row * k;
    int32_t* oidx = a.out_idx + out_row * k;
    for (int e = lane; e < k; e += 64) {
        if (e >= carry) {
            osim[e] = -INFINITY;
            oidx[e] = -1;
        }
    }
    for (int e = lane; e < carry; e += 64) {
        const uint32_t mu = sel_u[e], mi = sel_id[e];
        int rank = 0;
        for (int j = 0; j < carry; ++j) {
            const uint32_t ou = sel_u[j], oi = sel_id[j];
            rank += (ou > mu) || (ou == mu && oi < mi);
        }
        osim[rank] = sortable_f32(mu);
        oidx[rank] = (int32_t)mi;
    }
}

int launch_select(fal_ctx* ctx, int stage, int mode, const SelectArgs& a, int64_t n_blocks) {
    if (n_blocks <= 0) return FAL_OK;
    FAL_REQUIRE(a.k >= 1 && a.k <= FAL_MAX_K_ANN, FAL_EUNSUPPORTED, "k must be in [1, %d]", FAL_MAX_K_ANN);
    FAL_REQUIRE(n_blocks < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many queries in one select launch");
    StageScope ts(ctx, stage);
    if (mode == MODE_DENSE)
        hipLaunchKernelGGL(select_kernel<MODE_DENSE>, dim3((unsigned)n_blocks), dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(select_kernel<MODE_IVF>, dim3((unsigned)n_blocks), dim3(64), 0, ctx->stream, a);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
