// The cosine kernels: fp32-MFMA query x candidate inner products (dense form: k-means assign,
// coarse quantiser, flat-bucket scan) and the wavefront top-k select.
//
// Spec: reference README.md:107-113, 134-142 (Faiss IVF build / n_probe search; no code in the
// snapshot).  See simtile.h for the tile algorithm, DESIGN.md for the roofline.
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include "simtile.h"
#include "scan.h"
#include "select.h"
#include "ivf.h"

namespace fal {

// ---------------------------------------------------------------------------------------------
// dense tile kernel
// ---------------------------------------------------------------------------------------------
template <int DH4, int EPI>
__global__ __launch_bounds__(64, 1) void dense_kernel(
    const float* __restrict__ Q, const float* __restrict__ Cm, int d, const DenseJob* __restrict__ jobs,
    int n_jobs, int64_t tile_begin, int64_t n_tiles, float* __restrict__ sims, int64_t sims_base,
    int32_t* __restrict__ assign, int xcd_lists, int symmetric) {
    DenseJob job;
    int lt;
    int ji = 0;
    if (xcd_lists) {
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

    // symmetric mode (a flat bucket scanned against itself): sim(i, j) == sim(j, i) BIT FOR BIT (same
    // k-ordered fmaf chain, commutative products), so this tile only computes the blocks on and above
    // its diagonal and writes every off-diagonal block twice, once transposed.  Half the MFMA work.
    const int c_first = (EPI == EPI_STORE && symmetric) ? 32 * lt : 0;
    float* outT = nullptr;      // transposed target: rows = this lane's candidate, columns = this tile's queries
    if (EPI == EPI_STORE && symmetric) outT = sims + (job.obase - sims_base) + (int64_t)r * ncp + 32 * lt + 4 * h;
    // Chunks are walked from the LAST one down to c_first: tiles of a bucket that start together then
    // read the same chunk at the same time (shared L2 lines) and simply stop at their own diagonal;
    // walking upwards from the diagonal would spread the running tiles over the whole bucket
    // (measured: 2.84 vs 2.60 ms of scan per 1 M spectra; alternating the direction per bucket: 2.68).
    // (The arg-max form keeps the ascending walk: with ids arriving in ascending order "s > best" alone breaks
    //  ties towards the lowest id; the longer tie test made hipcc demote the query registers to scratch, 3.5x
    //  slower k-means.)
    const int c_last = ((nc - 1) >> 5) << 5;
    constexpr bool kDown = EPI == EPI_STORE;
    const int c_begin = kDown ? c_last : 0, c_step = kDown ? -32 : 32;
    const int n_chunks = (c_last - c_first) / 32 + 1;
    CandStream<DH4> cs;
    const float* cur = Cm + (job.c_row0 + min(c_begin + r, nc - 1)) * d + (int64_t)h * dh;
    cs.prime(cur, dh4);
    f32x16 prev;                 // the previous chunk's result; its epilogue runs inside this chunk
    // before the first chunk "previous" is a dummy: zeros land in chunk 0's slots and are overwritten
    // by the real chunk-0 epilogue later in program order; -inf never wins the arg-max.  Keeping the
    // epilogue unconditional keeps its stores out of branches (exact vmcnt bookkeeping).
    int prev_c0 = c_begin;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = (EPI == EPI_STORE) ? 0.f : -INFINITY;
    auto epilogue = [&]() {
        if (EPI == EPI_STORE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) out[mfma32_row(i, h) * ncp + prev_c0] = prev[i];
            if (symmetric) {
                // registers 4g .. 4g+3 hold query rows 8g + 4h + 0..3 = four consecutive columns of the
                // transposed block: one 16-byte store each (the diagonal block is rewritten with itself)
                float* t = outT + (int64_t)prev_c0 * ncp;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(t + 8 * g) = make_float4(prev[4 * g], prev[4 * g + 1], prev[4 * g + 2], prev[4 * g + 3]);
                __builtin_amdgcn_sched_group_barrier(0x040, 20, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x040, 16, 0);   // keep the 16 stores HERE in the pipeline
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = prev_c0 + mfma32_row(i, h);
                const float s = prev[i];
                if (c < nc && s > best) {   // ids arrive in ascending order: ties keep the lowest id
                    best = s;
                    bestc = c;
                }
            }
        }
    };
    for (int ci = 0, c0 = c_begin; ci < n_chunks; ++ci, c0 += c_step) {
        const float* nxt = Cm + (job.c_row0 + min(max(c0 + c_step, 0) + r, nc - 1)) * d + (int64_t)h * dh;   // (clamped prefetch)
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
    // a flat bucket against itself (XCD-list mode is only used for that): exploit the symmetry
    const int symmetric = (EPI == EPI_STORE && xcd_list_tiles > 0 && Q == Cm) ? 1 : 0;
    const int dh4 = d / 8;
    const int xcd_lists = xcd_list_tiles > 0;
    // XCD-list mode: n_tiles is unused, the grid is 8 x (longest list)
    const int64_t per_xcd = xcd_lists ? xcd_list_tiles : (n_tiles + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
 dim3 grid((unsigned)(per_xcd * 8)), block(64);
    StageScope ts(ctx, stage);
    StageScope tk(ctx, ST_KERNEL, nullptr, stage == ST_SCAN);      // (a second event pair for the fine scan only)
#define FAL_LAUNCH_DENSE(DH4)                                                                              \
    hipLaunchKernelGGL((dense_kernel<DH4, EPI>), grid, block, 0, ctx->stream, Q, Cm, d, jobs, n_jobs,      \
                       tile_begin, n_tiles, sims, sims_base, assign, xcd_lists, symmetric)
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

template <int MODE, bool FUSE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == MODE_DENSE ? 8 : 3, 8))) void select_kernel(SelectArgs a) {
    __shared__ uint32_t sel_u[kSelBuf];       // the selected set (+ room for one streamed chunk of a long row)
    __shared__ uint32_t sel_id[kSelBuf];
    __shared__ int64_t seg_off[FAL_MAX_N_PROBE + 1];   // MODE_IVF: stream offset of each probed list
    __shared__ int64_t seg_src[FAL_MAX_N_PROBE];       // MODE_IVF: perm position of each probed list
    const int lane = threadIdx.x;
    const int k = a.k;

    // ---- locate this query ---------------------------------------------------------------
    SelQuery qy{nullptr, 0, 0};
    int64_t out_row = 0;
    const int64_t t = a.tile_begin + (blockIdx.x >> 5);
    const int ql = blockIdx.x & 31;
    const DenseJob job = a.jobs[a.tile_job[blockIdx.x >> 5]];
    const int lt = (int)(t - job.tile0);
    if (32 * lt + ql >= job.nq) return;
    if (MODE == MODE_DENSE) {
        qy.nc = job.nc;
        qy.row = a.sims + (job.obase - a.sims_base) + (int64_t)(32 * lt + ql) * ((job.nc + 31) & ~31);
        out_row = job.q_row0 + 32 * (int64_t)lt + ql;
        qy.id0 = a.ids_are_rows ? job.c_row0 : 0;
    } else {
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
        qy.nc = seg_off[np];
        qy.row = a.sims + (a.q_sim_off[32 * t + ql] - a.sims_base);   // tile-order slot
        out_row = a.perm[p];
    }

    int carry;
    if (qy.nc <= 128) carry = select_rounds<MODE, 2>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 256) carry = select_rounds<MODE, 4>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 384) carry = select_rounds<MODE, 6>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 512) carry = select_rounds<MODE, 8>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 768) carry = select_rounds<MODE, 12>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else carry = select_rounds<MODE, 16>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);

    if constexpr (FUSE) {
        // the survivors of the filter go to the tail of the set's buffers (free once the selection is done)
        filter_sort_store(a, sel_u, sel_id, sel_u + FAL_MAX_K_ANN, sel_id + FAL_MAX_K_ANN, carry, out_row, lane);
    } else {
        float* osim = a.out_sim + out_row * k;
        int32_t* oidx = a.out_idx + out_row * k;
        if (k <= 64) sort_and_store<1>(sel_u, sel_id, carry, k, lane, osim, oidx);
        else if (k <= 128) sort_and_store<2>(sel_u, sel_id, carry, k, lane, osim, oidx);
        else sort_and_store<4>(sel_u, sel_id, carry, k, lane, osim, oidx);
    }
}

__global__ void tile_job_kernel(const DenseJob* __restrict__ jobs, int n_jobs, int64_t tile_begin, int64_t n_tiles,
                                int32_t* __restrict__ tile_job) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_tiles) tile_job[i] = find_job(jobs, n_jobs, tile_begin + i);
}

int launch_select(fal_ctx* ctx, int stage, int mode, const SelectArgs& a_in, int64_t n_blocks, hipStream_t on) {
    if (n_blocks <= 0) return FAL_OK;
    SelectArgs a = a_in;
    {
        // tile -> job once per tile instead of a binary search in every query's workgroup
        const int64_t n_tiles = n_blocks / 32;
        int32_t* tj = nullptr;
        FAL_TRY(ctx->reserve(SLOT_TILEJOB, sizeof(int32_t) * (size_t)std::max<int64_t>(n_tiles, 1 << 16), (void**)&tj));
        hipLaunchKernelGGL(tile_job_kernel, dim3((unsigned)ceil_div(n_tiles, 256)), dim3(256), 0, on ? on : ctx->stream, a.jobs,
                           a.n_jobs, a.tile_begin, n_tiles, tj);
        a.tile_job = tj;
    }
    FAL_REQUIRE(a.k >= 1 && a.k <= FAL_MAX_K_ANN, FAL_EUNSUPPORTED, "k must be in [1, %d]", FAL_MAX_K_ANN);
    FAL_REQUIRE(n_blocks < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many queries in one select launch");
    hipStream_t st = on ? on : ctx->stream;
    StageScope ts(ctx, stage, st);
    const bool fuse = a.nb_idx != nullptr;
    if (fuse) FAL_REQUIRE(a.f_pmz && a.nb_dist && a.f_keep >= 1 && a.f_keep <= FAL_MAX_K_ANN, FAL_EINVAL, "fused filter: bad arguments");
    if (mode == MODE_DENSE && fuse)
        hipLaunchKernelGGL((select_kernel<MODE_DENSE, true>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    else if (mode == MODE_DENSE)
        hipLaunchKernelGGL((select_kernel<MODE_DENSE, false>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    else if (fuse)
        hipLaunchKernelGGL((select_kernel<MODE_IVF, true>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((select_kernel<MODE_IVF, false>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
