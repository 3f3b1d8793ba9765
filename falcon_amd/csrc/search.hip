// a7: n_probe search of every row against its bucket's index, top-k_ann by (sim desc, id asc).
//
// Flat buckets (one list): dense tile scan of the bucket against itself + wavefront select.
// IVF buckets: coarse quantiser (dense scan vs the bucket's centroids + select k = n_probe),
// candidate-count prefix, fine scan over the union of probed lists, select.
// The sims of a batch of buckets live in one scratch buffer (default 2 GiB, FALCON_SIMS_MB): big
// enough that a launch holds many tiles per wave slot (short launches lose more to their tail than
// the buffer's HBM round trip costs; measured 160 MiB: 4.5 ms, 1 GiB: 3.5 ms of scan at 1 M spectra).
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "util.h"
#include "fused.h"
#include "ivf16.h"
#include "coarse16.h"

namespace fal {

// total candidates of each query = sum of the sizes of its probed lists (0 where probes are -1),
// stored at the query's TILE-ORDER slot 32 * tile + lane (unused lanes of a last tile stay 0)
template <int NPV>      // probes per query / 4 loaded as 16-byte pieces in front of the list-size gathers (0: any n_probe)
__global__ void probe_totals_kernel(const int32_t* __restrict__ probes, int np, const DenseJob* __restrict__ jobs,
                                    int n_jobs, int64_t n_tiles, const int64_t* __restrict__ list_off,
                                    int64_t* __restrict__ totals) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t t = g >> 5;
    int64_t tot = 0;
    if (t < n_tiles) {
        const DenseJob job = jobs[find_job(jobs, n_jobs, t)];
        const int lt = (int)(t - job.tile0), ql = (int)(g & 31);
        if (32 * lt + ql < job.nq) {
            const int64_t p = job.q_row0 + 32 * (int64_t)lt + ql;
            const int64_t* lo = list_off + job.c_row0;
            if (NPV > 0) {
                int4 pr[NPV > 0 ? NPV : 1];
                const int4* src = reinterpret_cast<const int4*>(probes + p * np);
#pragma unroll
                for (int v = 0; v < NPV; ++v) pr[v] = src[v];
                int64_t part[NPV > 0 ? 4 * NPV : 1];
#pragma unroll
                for (int v = 0; v < NPV; ++v) {
                    const int l[4] = {pr[v].x, pr[v].y, pr[v].z, pr[v].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) part[4 * v + c] = l[c] >= 0 ? lo[l[c] + 1] - lo[l[c]] : 0;
                }
#pragma unroll
                for (int j = 0; j < 4 * NPV; ++j) tot += part[j];
            } else {
                for (int j = 0; j < np; ++j) {
                    const int l = probes[p * np + j];
                    if (l >= 0) tot += lo[l + 1] - lo[l];
                }
            }
            totals[g] = tot;      // slot g = 32 * tile + lane (tile order, padded)
        }
    }
    // the largest candidate count of the launch, behind the padded slots (one atomic per wave)
    int64_t m = tot;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (int64_t)__shfl_xor((long long)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(reinterpret_cast<unsigned long long*>(totals + 32 * n_tiles + 1), (unsigned long long)m);
}

// ---- inverted probe table: for every list, the queries that probe it ---------------------------
__global__ void probe_hist_kernel(const int32_t* __restrict__ probes, int np, const DenseJob* __restrict__ jobs,
                                  int n_jobs, int64_t n_tiles, int32_t* __restrict__ cnt) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t t = g >> 5;
    if (t >= n_tiles) return;
    const DenseJob job = jobs[find_job(jobs, n_jobs, t)];
    const int lt = (int)(t - job.tile0), ql = (int)(g & 31);
    if (32 * lt + ql >= job.nq) return;
    const int64_t p = job.q_row0 + 32 * (int64_t)lt + ql;
    for (int j = 0; j < np; ++j) {
        const int l = probes[p * np + j];
        if (l >= 0) atomicAdd(&cnt[job.c_row0 + l], 1);
    }
}

// tiles of a list = groups of 2^shift rows of the LIST (each streams all queries probing the list); a list
// nobody probes needs none
__global__ void list_tiles_kernel(const int32_t* __restrict__ cnt, const int64_t* __restrict__ list_off, int64_t n,
                                  int shift, int32_t* __restrict__ tiles) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rows = list_off[i + 1] - list_off[i];
        tiles[i] = (cnt[i] > 0 && rows > 0) ? (int32_t)((rows + (1 << shift) - 1) >> shift) : 0;
    }
}

// inv_q[e] = query position, inv_dest[e] = where that query's sims for this list start
// (= its sims offset + the sizes of the lists it probes before this one)
__global__ void probe_scatter_kernel(const int32_t* __restrict__ probes, int np, const DenseJob* __restrict__ jobs,
                                     int n_jobs, int64_t n_tiles, const int64_t* __restrict__ list_off,
                                     const int64_t* __restrict__ q_sim_off, const int64_t* __restrict__ inv_off,
                                     int32_t* __restrict__ cursor, int32_t* __restrict__ inv_q,
                                     int64_t* __restrict__ inv_dest) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t t = g >> 5;
    if (t >= n_tiles) return;
    const DenseJob job = jobs[find_job(jobs, n_jobs, t)];
    const int lt = (int)(t - job.tile0), ql = (int)(g & 31);
    if (32 * lt + ql >= job.nq) return;
    const int64_t p = job.q_row0 + 32 * (int64_t)lt + ql;
    int64_t dest = q_sim_off[g];
    for (int j = 0; j < np; ++j) {
        const int l = probes[p * np + j];
        if (l < 0) continue;
        const int64_t G = job.c_row0 + l;
        const int64_t e = inv_off[G] + atomicAdd(&cursor[G], 1);
        inv_q[e] = (int32_t)p;
        inv_dest[e] = dest;
        dest += list_off[G + 1] - list_off[G];
    }
}

// entries per list (the histogram of the probe table), one workgroup per bucket with the counters in LDS
__global__ __launch_bounds__(1024) void probe_hist_bucket_kernel(const int32_t* __restrict__ probes, int np,
                                                                 const DenseJob* __restrict__ jobs, int32_t* __restrict__ cnt) {
    extern __shared__ int32_t hist[];
    const DenseJob job = jobs[blockIdx.x];
    const int nl = job.nc;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t e = threadIdx.x; e < (int64_t)job.nq * np; e += blockDim.x) {
        const int l = probes[job.q_row0 * np + e];
        if (l >= 0) atomicAdd(&hist[l], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nl; i += blockDim.x) cnt[job.c_row0 + i] = hist[i];
}

// probe_hist_bucket_kernel + probe_totals_kernel in one pass over the probes (round 5; buckets of up to 4,096 lists): a thread per
// query, the bucket's list sizes and the histogram in LDS -- a query's candidate total is the sum of its probed lists' sizes
// (16 LDS reads instead of 32 dependent 8-byte gathers from global memory), stored at its tile-order slot; the launch's largest
// total behind the padded slots as probe_totals_kernel leaves it.
template <int NPV>
__global__ __launch_bounds__(1024) void probe_hist_totals_bucket_kernel(const int32_t* __restrict__ probes, int np,
                                                                        const DenseJob* __restrict__ jobs,
                                                                        const int64_t* __restrict__ list_off, int64_t n_tiles,
                                                                        int32_t* __restrict__ cnt, int64_t* __restrict__ totals) {
    extern __shared__ int32_t lds32[];
    const DenseJob job = jobs[blockIdx.x];
    const int nl = job.nc;
    int32_t* hist = lds32;                // [nl]
    int32_t* len = lds32 + nl;            // [nl] rows of every list
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        hist[i] = 0;
        len[i] = (int32_t)(list_off[job.c_row0 + i + 1] - list_off[job.c_row0 + i]);
    }
    __syncthreads();
    int64_t mx = 0;
    for (int ql = threadIdx.x; ql < job.nq; ql += blockDim.x) {
        const int64_t p = job.q_row0 + ql;
        int64_t tot = 0;
        auto take = [&](int l) {
            if (l < 0) return;
            atomicAdd(&hist[l], 1);
            tot += len[l];
        };
        if (NPV > 0) {
            int4 pr[NPV > 0 ? NPV : 1];
            const int4* src = reinterpret_cast<const int4*>(probes + p * np);
#pragma unroll
            for (int v = 0; v < NPV; ++v) pr[v] = src[v];
#pragma unroll
            for (int v = 0; v < NPV; ++v) { take(pr[v].x); take(pr[v].y); take(pr[v].z); take(pr[v].w); }
        } else {
            for (int j = 0; j < np; ++j) take(probes[p * np + j]);
        }
        totals[32 * (job.tile0 + (ql >> 5)) + (ql & 31)] = tot;
        mx = max(mx, tot);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (int64_t)__shfl_xor((long long)mx, off, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(reinterpret_cast<unsigned long long*>(totals + 32 * n_tiles + 1), (unsigned long long)mx);
    __syncthreads();
    for (int i = threadIdx.x; i < nl; i += blockDim.x) cnt[job.c_row0 + i] = hist[i];
}

// The same table with every list's entries in ASCENDING QUERY ORDER (to within one chunk of 1,024 queries): one workgroup per
// bucket walks the bucket's queries chunk by chunk, cursors in LDS.  The fine-scan kernels stream a list's queries in table
// order; the ~64 lists of a bucket that run concurrently on an XCD then sweep the bucket's rows together and share them in L2
// (with the arrival order of global atomics the gathers came from HBM: 4.2 TB/s for 128 GB at 10 M spectra).
// Nothing the scatter of a query needs is loaded inside its probe loop: the bucket's list sizes and table offsets sit in LDS,
// the query's probes arrive as 16-byte pieces before the first of them is used (the first version paid three dependent global
// round trips per probe: probe -> list_off pair / inv_off, 32 times per query and chunk).
template <int NPV>      // probes per query / 4 held in registers (0: any n_probe, the plain loop)
__global__ __launch_bounds__(1024) void probe_scatter_bucket_kernel(const int32_t* __restrict__ probes, int np,
                                                                    const DenseJob* __restrict__ jobs,
                                                                    const int64_t* __restrict__ list_off,
                                                                    const int64_t* __restrict__ q_sim_off,
                                                                    const int64_t* __restrict__ inv_off, int32_t* __restrict__ inv_q,
                                                                    int64_t* __restrict__ inv_dest, const int32_t* __restrict__ perm,
                                                                    int32_t* __restrict__ inv_row, int lds_lists) {
    extern __shared__ int64_t lds64[];
    const DenseJob job = jobs[blockIdx.x];
    const int nl = job.nc;
    // lds_lists >= the launch's largest list count: 12 bytes of LDS per list (table position + size); otherwise (buckets with
    // thousands of lists: 12 bytes each would not fit) only a 4-byte cursor per list, offsets and sizes from global memory
    const bool full = nl <= lds_lists;
    int64_t* at = lds64;                                     // [nl] where the list's next table entry goes
    int32_t* len = reinterpret_cast<int32_t*>(lds64 + (full ? nl : 0));   // [nl] rows of the list | (else) entries written so far
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        if (full) {
            const int64_t b = list_off[job.c_row0 + i], e = list_off[job.c_row0 + i + 1];
            len[i] = (int32_t)(e - b);
            at[i] = inv_off[job.c_row0 + i];
        } else {
            len[i] = 0;
        }
    }
    __syncthreads();
    for (int q0 = 0; q0 < job.nq; q0 += blockDim.x) {
        const int ql = q0 + threadIdx.x;
        if (ql < job.nq) {
            const int64_t p = job.q_row0 + ql;
            int64_t dest = q_sim_off[32 * (job.tile0 + (ql >> 5)) + (ql & 31)];
            const int32_t row = inv_row ? perm[p] : 0;       // (the f16 list scan gathers its queries by sorted row)
            auto put = [&](int l) {
                if (l < 0) return;
                int64_t e;
                int32_t rows;
                if (full) {
                    e = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(&at[l]), 1ull);
                    rows = len[l];
                } else {
                    const int64_t G = job.c_row0 + l;
                    e = inv_off[G] + atomicAdd(&len[l], 1);
                    rows = (int32_t)(list_off[G + 1] - list_off[G]);
                }
                inv_q[e] = (int32_t)p;
                inv_dest[e] = dest;
                if (inv_row) inv_row[e] = row;
                dest += rows;
            };
            if (NPV > 0) {
                int4 pr[NPV > 0 ? NPV : 1];
                const int4* src = reinterpret_cast<const int4*>(probes + p * np);
#pragma unroll
                for (int v = 0; v < NPV; ++v) pr[v] = src[v];
#pragma unroll
                for (int v = 0; v < NPV; ++v) { put(pr[v].x); put(pr[v].y); put(pr[v].z); put(pr[v].w); }
            } else {
                for (int j = 0; j < np; ++j) put(probes[p * np + j]);
            }
        }
        __syncthreads();                                     // chunk after chunk: the order inside a list follows the queries
    }
}

// lists per bucket up to which the probe table kernel keeps 12 bytes per list in LDS (48 KB)
static int probe_lds_lists() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FALCON_PROBE_LDS_LISTS");
        v = e ? std::max(0, std::min(4096, atoi(e))) : 4096;
    }
    return v;
}

static size_t sims_capacity_floats() {
    static size_t cap = 0;
    if (!cap) {
        const char* e = getenv("FALCON_SIMS_MB");
        size_t mb = e ? (size_t)atoll(e) : 2048;
        if (mb < 16) mb = 16;
        cap = mb * 1024 * 1024 / sizeof(float);
    }
    return cap;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::probe_totals_kernel<4>);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

namespace {
struct NeighborFilter {          // a8 fused into the final selection (scan.h SelectArgs::nb_idx)
    const float* pmz;
    const float* rt;
    double tol, rt_tol;
    int is_da, keep;
    int32_t* nb_idx;
    float* nb_dist;
    int32_t* nb_count;
};

void set_filter(SelectArgs& sa, const NeighborFilter* nf) {
    if (!nf) return;
    sa.f_pmz = nf->pmz; sa.f_rt = nf->rt; sa.f_tol = nf->tol; sa.f_rt_tol = nf->rt_tol;
    sa.f_is_da = nf->is_da; sa.f_keep = nf->keep; sa.nb_idx = nf->nb_idx; sa.nb_dist = nf->nb_dist;
    sa.nb_count = nf->nb_count;
}
}  // namespace

static int search_impl(fal_ctx* ctx, const fal_ivf* ivf, int n_probe, int k_ann, float* sim, int32_t* idx,
                       const NeighborFilter* nf) {
    FAL_REQUIRE(ctx && ivf, FAL_EINVAL, "fal_ivf_search_topk: NULL ctx/ivf");
    FAL_REQUIRE(k_ann >= 1 && k_ann <= FAL_MAX_K_ANN, FAL_EUNSUPPORTED, "fal_ivf_search_topk: k_ann must be in [1, %d]", FAL_MAX_K_ANN);
    FAL_REQUIRE(n_probe >= 1 && n_probe <= FAL_MAX_N_PROBE, FAL_EUNSUPPORTED, "fal_ivf_search_topk: n_probe must be in [1, %d]", FAL_MAX_N_PROBE);
    if (ivf->n == 0) return FAL_OK;
    FAL_REQUIRE(nf || (sim && idx), FAL_EINVAL, "fal_ivf_search_topk: NULL output");
    hipStream_t st = ctx->stream;
    const int d = ivf->d;
    const int64_t n_buckets = (int64_t)ivf->n_list.size();
    ctx->stage_reset(ST_COARSE);
    ctx->stage_reset(ST_SCAN);
    ctx->stage_reset(ST_KERNEL);
    ctx->stage_reset(ST_SELECT);
    // fallback queries of this call (fal_ctx_counter 5): reset in stream order -- a previous call's asynchronous copy into
    // the same pinned words may still be in flight
    FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host, ctx->zero_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host + 2, ctx->zero_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    const size_t cap = sims_capacity_floats();

    // ---- job tables ------------------------------------------------------------------------
    // Flat buckets are visited in order of decreasing size and dealt to the 8 XCD lists round-robin
    // (simtile.h, XCD-list mode): the longest tiles start first, every XCD gets the same mix.
    std::vector<int64_t> border;
    {
        // counting sort by size (stable, descending): O(buckets + max size), this runs on the critical path
        int64_t max_sz = 0, n_flat = 0;
        for (int64_t b = 0; b < n_buckets; ++b) {
            const int64_t nb = ivf->bucket_off[b + 1] - ivf->bucket_off[b];
            if (ivf->n_list[b] == 1 && nb > 0) {
                max_sz = std::max(max_sz, nb);
                ++n_flat;
            }
        }
        std::vector<int64_t> start((size_t)max_sz + 2, 0);
        for (int64_t b = 0; b < n_buckets; ++b) {
            const int64_t nb = ivf->bucket_off[b + 1] - ivf->bucket_off[b];
            if (ivf->n_list[b] == 1 && nb > 0) ++start[(size_t)(max_sz - nb) + 1];
        }
        for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
        border.resize((size_t)n_flat);
        for (int64_t b = 0; b < n_buckets; ++b) {
            const int64_t nb = ivf->bucket_off[b + 1] - ivf->bucket_off[b];
            if (ivf->n_list[b] == 1 && nb > 0) border[(size_t)start[(size_t)(max_sz - nb)]++] = b;
        }
    }
    std::vector<DenseJob> flat, coarse;    // coarse doubles as the IVF tile table
    flat.reserve(border.size());
    // Flat buckets with the top-k kept on chip (fused.hip): needs the float16 prefilter rows and the fused a8 output
    const bool fused = nf && ivf->Xpre && ivf->X && fused_supports(d) && !border.empty() &&
                       (ivf->bucket_off[border[0] + 1] - ivf->bucket_off[border[0]]) < 65536;
    if (fused) {
        // two tables over the same buckets (sorted by decreasing size, dealt to the 8 XCD lists round-robin): 32-query
        // tiles for the exact band / resolve kernels, 128-query tiles for the prefilter of buckets with more than k rows
        std::vector<DenseJob> j32, j128;
        j32.reserve(border.size());
        int64_t xt32[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xt128[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t pairs = 0;
        for (size_t j = 0; j < border.size(); ++j) {
            const int64_t b = border[j];
            const int64_t row0 = ivf->bucket_off[b], nb = ivf->bucket_off[b + 1] - row0;
            const int x = (int)(j & 7);
            j32.push_back({row0, row0, 0, 0, (int32_t)nb, (int32_t)nb, xt32[x]});
            xt32[x] += ceil_div(nb, 32);
            if (nb > k_ann) {                                 // (sizes are non-increasing: a prefix)
                const int y = (int)(j128.size() & 7);
                j128.push_back({row0, row0, 0, 0, (int32_t)nb, (int32_t)nb, xt128[y]});
                xt128[y] += ceil_div(nb, 128);
            }
            pairs += nb * nb;
        }
        DenseJob* jd = nullptr;
        FAL_TRY(ctx->reserve(SLOT_JOBS, sizeof(DenseJob) * (j32.size() + j128.size() + 1), (void**)&jd));
        FAL_TRY(ctx->upload(jd, j32.data(), sizeof(DenseJob) * j32.size()));
        if (!j128.empty()) FAL_TRY(ctx->upload(jd + j32.size(), j128.data(), sizeof(DenseJob) * j128.size()));
        FusedArgs fa{};
        fa.X = ivf->X; fa.X16 = reinterpret_cast<const __half*>(ivf->Xpre);
        fa.jobs32 = jd; fa.n_jobs32 = (int)j32.size(); fa.jobs128 = jd + j32.size(); fa.n_jobs128 = (int)j128.size();
        fa.k = k_ann; fa.pmz = nf->pmz; fa.rt = nf->rt; fa.tol = nf->tol; fa.rt_tol = nf->rt_tol; fa.is_da = nf->is_da;
        fa.keep = nf->keep; fa.nb_idx = nf->nb_idx; fa.nb_dist = nf->nb_dist; fa.nb_count = nf->nb_count;
        FAL_TRY(launch_fused(ctx, fa, d, ivf->n, *std::max_element(xt128, xt128 + 8), *std::max_element(xt32, xt32 + 8),
                             j32[0].nc));
        ctx->counters[0] = pairs;
        ctx->counters[4] = 0;
        ctx->counters[2] = 1;
        ctx->counters[3] = 0;
        border.clear();                       // nothing left for the staged flat path
    }
    // a batch = jobs [j0, j1): [j0, jm) go to the f16-MFMA kernel (128-query tiles), [jm, jk) to the shared-stream fp32 kernel
    // (groups of four 32-query tiles), [jk, j1) to the one-wave fp32 kernel
    // (round 6: [js, jk) -- buckets of 33 .. small_max rows -- to the ONE-WAVE fp32 kernel, a wave per 32-query tile with no staircase
    //  to wait in: dense4_kernel's four lockstep waves idle half of their steps on buckets of two or three tiles)
    struct FlatBatch { size_t j0, jm, js, jk, j1; int64_t tiles, list_tiles16, list_tiles32, list_tiles1, floats; };
    std::vector<FlatBatch> flat_batches;
    size_t need_flat = 0;
    const bool have16 = ivf->X16 != nullptr;
    const bool have4 = ivf->X && dense4_supports(d);       // the shared-stream fp32 kernels (scan.hip)
    static const int small_max_env = [] {
        const char* e = getenv("FALCON_DENSE1_MAX");
        return e ? atoi(e) : 0;
    }();
    const int64_t small_max = (have4 && d <= 512) ? std::max(32, small_max_env) : 32;      // (no one-wave fp32 kernel beyond 512 columns)
    const int64_t thr16 = ivf->X ? 64 : 0;                 // without float32 rows everything takes the f16 kernel; buckets below it: the fp32
                                                           // matrix kernels up to low_dim 512, exact chains on the vector ALU beyond
    FAL_REQUIRE(have16 || ivf->X || border.empty(), FAL_EINVAL, "fal_ivf_search_topk: the index has no vectors to scan");
    {
        FlatBatch cur{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int64_t xt16[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xt32[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xt1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto close = [&]() {
            if (cur.j1 > cur.j0) {
                cur.list_tiles16 = *std::max_element(xt16, xt16 + 8);
                cur.list_tiles32 = *std::max_element(xt32, xt32 + 8);
                cur.list_tiles1 = *std::max_element(xt1, xt1 + 8);
                flat_batches.push_back(cur);
                need_flat = std::max(need_flat, (size_t)cur.floats);
            }
            cur = FlatBatch{cur.j1, cur.j1, cur.j1, cur.j1, cur.j1, 0, 0, 0, 0, 0};
            std::fill(xt16, xt16 + 8, 0);
            std::fill(xt32, xt32 + 8, 0);
            std::fill(xt1, xt1 + 8, 0);
        };
        for (int64_t b : border) {
            const int64_t row0 = ivf->bucket_off[b], nb = ivf->bucket_off[b + 1] - row0;
            const int64_t tiles = ceil_div(nb, 32), floats = tiles * 32 * ((nb + 31) & ~31ll);
            if (cur.floats + floats > (int64_t)cap && cur.j1 > cur.j0) close();
            const bool use16 = have16 && nb >= thr16;      // sizes are non-increasing: the f16 part is a prefix ...
            const bool use4 = !use16 && have4 && nb > small_max;  // ... and the shared-stream part follows it; then the small ones
            const bool use1 = !use16 && !use4 && have4 && nb > 32;      // ... the one-wave part; then the one-block buckets
            int64_t xtile0;
            if (use16) {
                const int x = (int)((cur.j1 - cur.j0) & 7);
                xtile0 = xt16[x];
                xt16[x] += ceil_div(nb, 128);
                cur.jm = cur.js = cur.jk = cur.j1 + 1;
            } else if (use4) {
                xtile0 = 0;                                  // (launch_dense4 lists the groups of these jobs itself)
                cur.js = cur.jk = cur.j1 + 1;
            } else if (use1) {
                const int x = (int)((cur.j1 - cur.js) & 7);
                xtile0 = xt1[x];
                xt1[x] += tiles;
                cur.jk = cur.j1 + 1;
            } else {
                const int x = (int)((cur.j1 - cur.jk) & 7);
                xtile0 = xt32[x];
                xt32[x] += tiles;
            }
            // obase / tile0 / xtile0 are relative to the batch
            flat.push_back({row0, row0, cur.floats, cur.tiles, (int32_t)nb, (int32_t)nb, xtile0});
            cur.tiles += tiles;
            cur.floats += floats;
            cur.j1++;
        }
        close();
    }
    int64_t ivf_tiles = 0, coarse_o = 0;
    int max_n_list = 0;
    for (int64_t b = 0; b < n_buckets; ++b) {
        const int64_t row0 = ivf->bucket_off[b], nb = ivf->bucket_off[b + 1] - row0;
        if (nb == 0 || ivf->n_list[b] == 1) continue;
        const int64_t tiles = ceil_div(nb, 32);
        coarse.push_back({row0, ivf->list_base[b], coarse_o, ivf_tiles, (int32_t)nb, ivf->n_list[b], 0});
        ivf_tiles += tiles;
        coarse_o += tiles * 32 * (((int64_t)ivf->n_list[b] + 31) & ~31ll);
        max_n_list = std::max(max_n_list, ivf->n_list[b]);
    }
    // batches of whole tiles whose sims fit the buffer: [tile_begin, tile_end)
    auto make_batches = [&](const std::vector<DenseJob>& jobs, int64_t n_tiles, std::vector<int64_t>& cuts, size_t& need) {
        cuts.assign(1, 0);
        need = 0;
        int64_t used = 0;
        for (const DenseJob& j : jobs) {
            const int64_t tiles = ceil_div(j.nq, 32), per_tile = 32 * (((int64_t)j.nc + 31) & ~31ll);
            for (int64_t t = 0; t < tiles;) {
                int64_t room = ((int64_t)cap - used) / per_tile;
                if (room <= 0 && used > 0) {
                    cuts.push_back(j.tile0 + t);
                    used = 0;
                    continue;
                }
                if (room <= 0) room = 1;   // a single tile larger than the buffer: grow the buffer
                const int64_t take = std::min(room, tiles - t);
                used += take * per_tile;
                need = std::max(need, (size_t)used);
                t += take;
            }
        }
        if (cuts.back() != n_tiles) cuts.push_back(n_tiles);
    };
    auto obase_of_tile = [&](const std::vector<DenseJob>& jobs, int64_t t) -> int64_t {
        // jobs sorted by tile0
        size_t lo = 0, hi = jobs.size() - 1;
        while (lo < hi) {
            size_t mid = (lo + hi + 1) / 2;
            if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
        }
        return jobs[lo].obase + (t - jobs[lo].tile0) * 32 * (((int64_t)jobs[lo].nc + 31) & ~31ll);
    };

    std::vector<int64_t> coarse_cuts;
    size_t need_coarse = 0;
    if (!coarse.empty()) make_batches(coarse, ivf_tiles, coarse_cuts, need_coarse);

    DenseJob *flat_dev = nullptr, *coarse_dev = nullptr;
    if (!flat.empty()) {
        FAL_TRY(ctx->reserve(SLOT_JOBS, sizeof(DenseJob) * flat.size(), (void**)&flat_dev));
        FAL_TRY(ctx->upload(flat_dev, flat.data(), sizeof(DenseJob) * flat.size()));
    }
    if (!coarse.empty()) {
        FAL_TRY(ctx->reserve(SLOT_JOBS2, sizeof(DenseJob) * coarse.size(), (void**)&coarse_dev));
        FAL_TRY(ctx->upload(coarse_dev, coarse.data(), sizeof(DenseJob) * coarse.size()));
    }
    float* sims = nullptr;
    const size_t sims_floats = std::max(need_flat, need_coarse);
    FAL_TRY(ctx->reserve(SLOT_SIMS, sizeof(float) * (sims_floats + kSimsSlack), (void**)&sims));   // slack: scan16 sink, select over-reads
    // ---- A. flat buckets ---------------------------------------------------------------------
    for (size_t bi = 0; bi < flat_batches.size(); ++bi) {
        const FlatBatch& fb = flat_batches[bi];
        float* buf = sims;
        const DenseJob* jb = flat_dev + fb.j0;
        const int nj = (int)(fb.j1 - fb.j0);
        if (fb.jm > fb.j0)
            FAL_TRY(launch_scan16(ctx, ivf->x16_planes, ivf->X16, d, jb, (int)(fb.jm - fb.j0), fb.list_tiles16, buf, 0,
                                  buf + sims_floats));
        // (flat buckets keep their rows' positions in list order: the sorted rows serve)
        if (fb.js > fb.jm)
            FAL_TRY(launch_dense4(ctx, ivf->X, d, flat_dev + fb.jm, flat.data() + fb.jm, (int)(fb.js - fb.jm), buf, 0));
        if (fb.jk > fb.js)
            FAL_TRY(launch_dense(ctx, ST_SCAN, EPI_STORE, ivf->X, ivf->X, d, flat_dev + fb.js, (int)(fb.jk - fb.js), 0, fb.tiles, buf, 0,
                                 nullptr, fb.list_tiles1));
        if (fb.j1 > fb.jk && d > 512)
            FAL_TRY(launch_flat_exact_small(ctx, ivf->X, d, flat_dev + fb.jk, (int)(fb.j1 - fb.jk), buf, 0));
        else if (fb.j1 > fb.jk && have4)
            FAL_TRY(launch_dense_tiny4(ctx, ivf->X, d, flat_dev + fb.jk, (int)(fb.j1 - fb.jk), buf, 0));
        else if (fb.j1 > fb.jk)
            FAL_TRY(launch_dense(ctx, ST_SCAN, EPI_STORE, ivf->X, ivf->X, d, flat_dev + fb.jk, (int)(fb.j1 - fb.jk), 0,
                                 fb.tiles, buf, 0, nullptr, fb.list_tiles32));
        SelectArgs sa{};
        sa.sims = buf; sa.sims_base = 0; sa.k = k_ann; sa.out_sim = sim; sa.out_idx = idx;
        sa.jobs = jb; sa.n_jobs = nj; sa.tile_begin = 0; sa.ids_are_rows = 1;
        set_filter(sa, nf);
        FAL_TRY(launch_select(ctx, ST_SELECT, MODE_DENSE, sa, fb.tiles * 32));
    }
    if (!fused) {
        ctx->counters[0] = 0;
        ctx->counters[4] = 0;                  // inner products the matrix cores actually computed (incl. padding)
    }
    for (const FlatBatch& fb : flat_batches)
        for (size_t j = fb.j0; j < fb.j1; ++j) {
            const int64_t ch = ceil_div(flat[j].nc, 32);
            // fp32 kernel: blocks on and above the diagonal only (symmetric); f16 kernel: full square
            ctx->counters[4] += 1024 * (ch * (ch + 1) / 2);   // both kernels: blocks on/above the diagonal (f16: + up to 3 per 128-row tile)
        }
    for (const DenseJob& j : flat) ctx->counters[0] += (int64_t)j.nq * j.nc;
    ctx->counters[7] = ctx->counters[8] = 0;               // the one-block buckets (dense_tiny4_kernel: outside stage 8)
    if (have4)
        for (const FlatBatch& fb : flat_batches)
            for (size_t j = fb.jk; j < fb.j1; ++j) {
                ctx->counters[7] += (int64_t)flat[j].nq * flat[j].nc;
                ctx->counters[8] += 1024;
            }
    ctx->counters[1] = 0;
    if (!fused) {
        ctx->counters[2] = (int64_t)flat_batches.size();
        ctx->counters[3] = (int64_t)(sizeof(float) * std::max(need_flat, need_coarse));
    }
    if (coarse.empty()) return FAL_OK;   // (job tables went through the pinned upload ring: nothing to wait for)
    for (const DenseJob& j : coarse) ctx->counters[1] += (int64_t)j.nq * j.nc;

    // ---- B. IVF buckets: coarse quantiser ------------------------------------------------------
    const int np = std::min(n_probe, max_n_list);
    int32_t* probes = nullptr;
    float* probe_sim = nullptr;
    int64_t *totals = nullptr, *q_sim_off = nullptr;
    FAL_TRY(ctx->reserve(SLOT_PROBES, sizeof(int32_t) * (size_t)ivf->n * np, (void**)&probes));
    FAL_TRY(ctx->reserve(SLOT_PROBE_SIM, sizeof(float) * (size_t)ivf->n * np, (void**)&probe_sim));
    const int64_t n_slots = ivf_tiles * 32;
    FAL_TRY(ctx->reserve(SLOT_QOFF, sizeof(int64_t) * (size_t)(n_slots + 1), (void**)&q_sim_off));
    FAL_TRY(ctx->reserve(SLOT_MISC, sizeof(int64_t) * (size_t)(n_slots + 2), (void**)&totals));
    FAL_CHECK_HIP(hipMemsetAsync(totals, 0, sizeof(int64_t) * (size_t)(n_slots + 2), st));
    // the final k-means pass left the (row, centroid) similarities behind as 16-bit keys: no second scan (coarse16.hip)
    const bool from_keys = ivf->ckeys != nullptr && ivf->X != nullptr &&
                           (ivf->ckeys_stride <= 512 || (ivf->ckeys_stride <= 2048 && ivf->sp_cols != nullptr));
    if (from_keys) {
        Coarse16Args ca{ivf->ckeys, ivf->ckeys_stride, ivf->X, ivf->centroids, d, coarse_dev, (int)coarse.size(), ivf_tiles, nullptr,
                        ivf->perm, np, probes};
        ca.sp_cols = ivf->sp_cols;
        ca.sp_vals = ivf->sp_vals;
        FAL_TRY(launch_coarse16(ctx, ca));
    }
    if (!from_keys) FAL_TRY(ivf_ensure_xl(ctx, ivf));
    for (size_t bi = 0; !from_keys && bi + 1 < coarse_cuts.size(); ++bi) {
        const int64_t t0 = coarse_cuts[bi], t1 = coarse_cuts[bi + 1];
        const int64_t base = obase_of_tile(coarse, t0);
        FAL_TRY(launch_dense(ctx, ST_COARSE, EPI_STORE, ivf->Xl, ivf->centroids, d, coarse_dev, (int)coarse.size(), t0,
                             t1 - t0, sims, base, nullptr));
        SelectArgs sa{};
        sa.sims = sims; sa.sims_base = base; sa.k = np; sa.out_sim = probe_sim; sa.out_idx = probes;
        sa.jobs = coarse_dev; sa.n_jobs = (int)coarse.size(); sa.tile_begin = t0; sa.ids_are_rows = 0;
        FAL_TRY(launch_select(ctx, ST_COARSE, MODE_DENSE, sa, (t1 - t0) * 32));
    }
    // (buckets of up to 4,096 lists: the candidate totals come out of the probe histogram's pass below, one kernel for both)
    const bool fused_totals = max_n_list <= 4096;
    if (!fused_totals) {
        StageScope ts(ctx, ST_COARSE);
        const dim3 tg((unsigned)ceil_div(n_slots, 256));
        if (np == 16)
            hipLaunchKernelGGL(probe_totals_kernel<4>, tg, dim3(256), 0, st, probes, np, coarse_dev, (int)coarse.size(), ivf_tiles,
                               ivf->list_off, totals);
        else if (np == 32)
            hipLaunchKernelGGL(probe_totals_kernel<8>, tg, dim3(256), 0, st, probes, np, coarse_dev, (int)coarse.size(), ivf_tiles,
                               ivf->list_off, totals);
        else
            hipLaunchKernelGGL(probe_totals_kernel<0>, tg, dim3(256), 0, st, probes, np, coarse_dev, (int)coarse.size(), ivf_tiles,
                               ivf->list_off, totals);
        FAL_TRY(device_scan_i64(ctx, totals, n_slots, q_sim_off, SLOT_MISC2));      // (millions of slots: multi-block)
    }
    // fine-scan tiles: groups of four 32-row list slices for the 4-wave shared-stream kernels (ivf_fine.hip, ivf16.hip)
    const int group_shift = 7;
    // ---- inverted probe table (list -> queries probing it) ------------------------------------
    const int64_t TL = ivf->total_lists;
    const int64_t n_pairs_max = ivf->n * (int64_t)np;
    int32_t *cnt = nullptr, *inv_q = nullptr;
    int64_t *inv_off = nullptr, *inv_dest = nullptr;
    FAL_TRY(ctx->reserve(SLOT_INVCNT, sizeof(int32_t) * (size_t)(3 * TL + 3) + sizeof(int64_t) * (size_t)(2 * TL + 4), (void**)&inv_off));
    int64_t* ltile_off = inv_off + (TL + 2);
    cnt = reinterpret_cast<int32_t*>(ltile_off + (TL + 2));
    int32_t* cursor = cnt + (TL + 1);
    int32_t* ltiles = cursor + (TL + 1);
    // (the probe table by sorted row as well when the f16 list scan may follow: ordered table + float16 rows attached)
    const bool want_rows = nf && ivf->X16pre && ivf->pos_of_row && max_n_list <= 16384;
    FAL_TRY(ctx->reserve(SLOT_INV, (sizeof(int64_t) + (want_rows ? 2 : 1) * sizeof(int32_t)) * (size_t)n_pairs_max + 1024, (void**)&inv_dest));   // (slack: list16_kernel reads whole chunks of row ids)
    inv_q = reinterpret_cast<int32_t*>(inv_dest + n_pairs_max);
    int32_t* inv_row = want_rows ? inv_q + n_pairs_max + 64 : nullptr;
    FAL_CHECK_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(2 * TL + 2), st));     // cnt, cursor
    {
        StageScope ts(ctx, ST_COARSE);
        const unsigned pg = (unsigned)ceil_div(n_slots, 256);
        if (fused_totals) {
            const dim3 hg((unsigned)coarse.size());
            const size_t hl = 2 * sizeof(int32_t) * (size_t)max_n_list;
            if (np == 16)
                hipLaunchKernelGGL(probe_hist_totals_bucket_kernel<4>, hg, dim3(1024), hl, st, probes, np, coarse_dev, ivf->list_off,
                                   ivf_tiles, cnt, totals);
            else if (np == 32)
                hipLaunchKernelGGL(probe_hist_totals_bucket_kernel<8>, hg, dim3(1024), hl, st, probes, np, coarse_dev, ivf->list_off,
                                   ivf_tiles, cnt, totals);
            else
                hipLaunchKernelGGL(probe_hist_totals_bucket_kernel<0>, hg, dim3(1024), hl, st, probes, np, coarse_dev, ivf->list_off,
                                   ivf_tiles, cnt, totals);
            FAL_TRY(device_scan_i64(ctx, totals, n_slots, q_sim_off, SLOT_MISC2));
        } else if (max_n_list <= 16384)
            hipLaunchKernelGGL(probe_hist_bucket_kernel, dim3((unsigned)coarse.size()), dim3(1024), sizeof(int32_t) * (size_t)max_n_list,
                               st, probes, np, coarse_dev, cnt);
        else
            hipLaunchKernelGGL(probe_hist_kernel, dim3(pg), dim3(256), 0, st, probes, np, coarse_dev, (int)coarse.size(),
                               ivf_tiles, cnt);
        FAL_TRY(device_scan_i32(ctx, cnt, TL, inv_off, SLOT_MISC2));
        const dim3 tg((unsigned)std::min<int64_t>(ceil_div(TL, 256), 1024));
        hipLaunchKernelGGL(list_tiles_kernel, tg, dim3(256), 0, st, cnt, ivf->list_off, TL, group_shift, ltiles);
        FAL_TRY(device_scan_i32(ctx, ltiles, TL, ltile_off, SLOT_MISC2));
        if (max_n_list <= 16384) {
            // (dynamic LDS stays at or below 64 KB: 12 bytes per list up to 4,096 lists, else 4)
            const int lds_lists = max_n_list <= probe_lds_lists() ? max_n_list : 0;
            const size_t lds = lds_lists ? (sizeof(int64_t) + sizeof(int32_t)) * (size_t)max_n_list : sizeof(int32_t) * (size_t)max_n_list;
            const dim3 sg((unsigned)coarse.size());
            if (np == 16)
                hipLaunchKernelGGL(probe_scatter_bucket_kernel<4>, sg, dim3(1024), lds, st, probes, np, coarse_dev, ivf->list_off,
                                   q_sim_off, inv_off, inv_q, inv_dest, ivf->perm, inv_row, lds_lists);
            else if (np == 32)
                hipLaunchKernelGGL(probe_scatter_bucket_kernel<8>, sg, dim3(1024), lds, st, probes, np, coarse_dev, ivf->list_off,
                                   q_sim_off, inv_off, inv_q, inv_dest, ivf->perm, inv_row, lds_lists);
            else
                hipLaunchKernelGGL(probe_scatter_bucket_kernel<0>, sg, dim3(1024), lds, st, probes, np, coarse_dev, ivf->list_off,
                                   q_sim_off, inv_off, inv_q, inv_dest, ivf->perm, inv_row, lds_lists);
        } else
            hipLaunchKernelGGL(probe_scatter_kernel, dim3(pg), dim3(256), 0, st, probes, np, coarse_dev, (int)coarse.size(),
                               ivf_tiles, ivf->list_off, q_sim_off, inv_off, cursor, inv_q, inv_dest);
    }
    FAL_CHECK_HIP(hipGetLastError());
    // per-tile sims prefix and per-list tile prefix back to the host to cut bucket-sized batches
    std::vector<int64_t> qoff((size_t)ivf_tiles + 1), lt_host((size_t)TL + 1);
    FAL_CHECK_HIP(hipMemcpy2DAsync(qoff.data(), sizeof(int64_t), q_sim_off, 32 * sizeof(int64_t), sizeof(int64_t),
                                   (size_t)ivf_tiles + 1, hipMemcpyDeviceToHost, st));
    FAL_CHECK_HIP(hipMemcpyAsync(lt_host.data(), ltile_off, sizeof(int64_t) * (size_t)(TL + 1), hipMemcpyDeviceToHost, st));
    int64_t max_total = 0;                                   // the most candidates any query has
    FAL_CHECK_HIP(hipMemcpyAsync(&max_total, totals + n_slots + 1, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    FAL_CHECK_HIP(hipStreamSynchronize(st));
    // batches of whole IVF buckets (every list of a bucket touches queries all over the bucket).  The float16 path's batches hold
    // 2-byte keys: the same BYTES as a batch of float32 sims = twice the candidates (fewer, larger launches of the three per-batch
    // kernels: -1.3 ms per 10 M pass; smaller batches -- down to the 256 MB of the memory-side cache -- only lose, NOTES r6)
    const bool key_path = want_rows && ivf->X && ivf16_supports(d) && max_total <= 4096;
    const int64_t cap_fine = key_path ? (int64_t)std::min<size_t>(2 * cap, ((size_t)1 << 31) - 4 * kSimsSlack) : (int64_t)cap;
    struct IvfBatch { size_t j0, j1; };
    std::vector<IvfBatch> ivf_batches;
    size_t need_fine = 0;
    {
        size_t j0 = 0;
        for (size_t j = 0; j < coarse.size(); ++j) {
            const int64_t t_end = coarse[j].tile0 + ceil_div(coarse[j].nq, 32);
            const int64_t span = qoff[(size_t)t_end] - qoff[(size_t)coarse[j0].tile0];
            if (span > cap_fine && j > j0) {
                ivf_batches.push_back({j0, j});
                j0 = j;
            }
            need_fine = std::max(need_fine, (size_t)(qoff[(size_t)t_end] - qoff[(size_t)coarse[j0].tile0]));
        }
        ivf_batches.push_back({j0, coarse.size()});
    }
    ctx->counters[0] += qoff[(size_t)ivf_tiles];
    ctx->counters[2] += (int64_t)ivf_batches.size();
    ctx->counters[3] = std::max<int64_t>(ctx->counters[3], (int64_t)(sizeof(float) * need_fine));
    FAL_REQUIRE(need_fine + kSimsSlack < ((size_t)1 << 32), FAL_EUNSUPPORTED, "sims batch too large (lower FALCON_SIMS_MB)");
    // ---- C'. fine scan with the float16 prefilter (ivf16.hip): f16-MFMA scan to 16-bit keys, k-th key per query, exact tail
    // (select16_kernel holds at most 4,096 keys of a query in registers; coarser indexes keep the exact staged scan rather
    // than sending every query through the exact fallback)
    if (key_path) {
        // the IVF buckets in sorted-row order, 32-query tiles, sorted by decreasing size and dealt to the 8 XCD lists
        std::vector<size_t> ord(coarse.size());
        for (size_t j = 0; j < ord.size(); ++j) ord[j] = j;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return coarse[x].nq > coarse[y].nq; });
        std::vector<DenseJob> j32(coarse.size());
        int64_t xt32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t j = 0; j < ord.size(); ++j) {
            const DenseJob& cj = coarse[ord[j]];
            const int x = (int)(j & 7);
            j32[j] = {cj.q_row0, cj.c_row0, 0, 0, cj.nq, cj.nq, xt32[x]};
            xt32[x] += ceil_div(cj.nq, 32);
        }
        DenseJob* jd = nullptr;
        const int64_t tiles32 = *std::max_element(xt32, xt32 + 8);
        const size_t jd_bytes = (sizeof(DenseJob) * j32.size() + 15) & ~(size_t)15;
        FAL_TRY(ctx->reserve(SLOT_JOBS3, jd_bytes + sizeof(int32_t) * (size_t)(8 * tiles32), (void**)&jd));
        FAL_TRY(ctx->upload(jd, j32.data(), sizeof(DenseJob) * j32.size()));
        int32_t* tj32 = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(jd) + jd_bytes);
        FAL_TRY(launch_tile_job32(ctx, jd, (int)j32.size(), tiles32, tj32));
        FusedArgs fa{};
        fa.X = ivf->X; fa.jobs32 = jd; fa.n_jobs32 = (int)j32.size(); fa.tile_job32 = tj32;
        fa.k = k_ann; fa.pmz = nf->pmz; fa.rt = nf->rt; fa.tol = nf->tol; fa.rt_tol = nf->rt_tol; fa.is_da = nf->is_da;
        fa.keep = nf->keep; fa.nb_idx = nf->nb_idx; fa.nb_dist = nf->nb_dist; fa.nb_count = nf->nb_count;
        fa.ivf = 1; fa.assign = ivf->assign; fa.pos_of_row = ivf->pos_of_row; fa.probes = probes; fa.n_probe = np;
        fa.mask_words = (max_n_list + 31) / 32; fa.list_off = ivf->list_off; fa.perm = ivf->perm;
        fa.sp_cols = ivf->sp_cols; fa.sp_vals = ivf->sp_vals; fa.rows_f16 = ivf->rows_f16;
        FAL_TRY(fused_prepare(ctx, &fa, ivf->n));
        // the exact part: only the (query, candidate) pairs that can matter (pairs16.hip)
        int2* gsel = nullptr;
        float* pmz_l = nullptr;
        fa.ivf = 2;
        FAL_TRY(ctx->reserve(SLOT_WIN, (sizeof(int2) + sizeof(float)) * (size_t)ivf->n, (void**)&gsel));
        pmz_l = reinterpret_cast<float*>(gsel + ivf->n);
        FAL_TRY(launch_gather_pmz(ctx, nf->pmz, ivf->perm, ivf->n, pmz_l));
        uint16_t* keys = nullptr;
        // (list16_kernel addresses a batch's keys by 32-bit byte offsets from the buffer's base)
        FAL_REQUIRE(need_fine + 2 * kSimsSlack < ((size_t)1 << 31), FAL_EUNSUPPORTED, "key batch too large (lower FALCON_SIMS_MB)");
        ctx->release(SLOT_SIMS);        // the flat / coarse stage's sims are dead (its launches are enqueued)
        FAL_TRY(ctx->reserve(SLOT_SIMS, sizeof(uint16_t) * (need_fine + 2 * kSimsSlack), (void**)&keys));
        int64_t max_cand = 0;
        for (int64_t t = 0; t < ivf_tiles; ++t) max_cand = std::max(max_cand, qoff[(size_t)t + 1] - qoff[(size_t)t]);
        for (const IvfBatch& bt : ivf_batches) {
            const DenseJob &first = coarse[bt.j0], &last = coarse[bt.j1 - 1];
            const int64_t t0 = first.tile0, t1 = last.tile0 + ceil_div(last.nq, 32);
            const int64_t L0 = first.c_row0, L1 = last.c_row0 + last.nc;
            const int64_t base = qoff[(size_t)t0];
            List16Args la{};
            la.X16 = reinterpret_cast<const __half*>(ivf->X16pre); la.perm = ivf->perm;
            la.sq16 = ivf->rows_many ? nullptr : ivf->sq16;      // (queries as 256-byte sparse records: list16s.hip)
            la.d = d; la.list_off = ivf->list_off; la.inv_off = inv_off;
            la.ltile_off = ltile_off; la.inv_row = inv_row; la.inv_dest = inv_dest; la.list_begin = L0; la.list_end = L1;
            la.tile_begin = lt_host[(size_t)L0]; la.n_tiles_max = lt_host[(size_t)L1] - lt_host[(size_t)L0];
            la.keys = keys; la.keys_base = base; la.sink = keys + need_fine; la.n_rows = ivf->n;
            FAL_TRY(launch_list16(ctx, la));
            Kept16Args ka{};
            ka.keys = keys; ka.keys_base = base; ka.jobs = coarse_dev; ka.tile_begin = t0; ka.n_probe = np;
            ka.probes = probes; ka.list_off = ivf->list_off; ka.q_sim_off = q_sim_off; ka.perm = ivf->perm; ka.pmz_l = pmz_l;
            ka.rt = nf->rt; ka.tol = nf->tol; ka.rt_tol = nf->rt_tol; ka.tol_f = fa.tol_f; ka.rt_f = fa.rt_f; ka.is_da = nf->is_da;
            ka.gsel = gsel; ka.gkept_id = fa.gkept_id; ka.gkcnt = fa.gkcnt;
            ka.n_tiles = t1 - t0;
            Select16Args sa{};
            sa.keys = keys; sa.keys_base = base; sa.k = k_ann; sa.jobs = coarse_dev; sa.n_jobs = (int)coarse.size();
            sa.tile_begin = t0; sa.q_sim_off = q_sim_off;
            sa.perm = ivf->perm; sa.thr = fa.thr; sa.gmem_v = fa.gmem_v; sa.gmem_id = fa.gmem_id;
            sa.max_keys = max_total;
            sa.gsel = gsel;
            // FALCON_KEPT16=fused (A/B switch, read per launch; single-pass searches -- no query above 2,048 keys): kept16 as the
            // tail of the selection kernel, four queries per wave (select16k_kernel)
            const char* ke = getenv("FALCON_KEPT16");
            const bool fuse_kept = max_total + 7 <= 2048 && ke && !strcmp(ke, "fused");
            sa.fuse_kept = fuse_kept ? 1 : 0;
            sa.kept = ka;
            FAL_TRY(launch_select16(ctx, sa, t1 - t0));
            if (!fuse_kept) FAL_TRY(launch_kept16(ctx, ka, t1 - t0));
        }
        FAL_TRY(launch_pairs16(ctx, fa, d, *std::max_element(xt32, xt32 + 8)));
        FAL_TRY(launch_fused_ivf_tail(ctx, fa, d, *std::max_element(xt32, xt32 + 8), max_cand));
        FAL_CHECK_HIP(hipStreamSynchronize(st));
        return FAL_OK;
    }
    ctx->release(SLOT_SIMS);        // the flat / coarse stage's sims are dead (its launches are enqueued)
    FAL_TRY(ctx->reserve(SLOT_SIMS, sizeof(float) * (need_fine + kSimsSlack), (void**)&sims));
    FAL_TRY(ivf_ensure_xl(ctx, ivf));
    for (const IvfBatch& bt : ivf_batches) {
        const DenseJob &first = coarse[bt.j0], &last = coarse[bt.j1 - 1];
        const int64_t t0 = first.tile0, t1 = last.tile0 + ceil_div(last.nq, 32);
        const int64_t L0 = first.c_row0, L1 = last.c_row0 + last.nc;        // lists of these buckets (natural order)
        const int64_t base = qoff[(size_t)t0];
        ListScanArgs la{};
        la.Xl = ivf->Xl; la.d = d; la.list_off = ivf->list_off; la.inv_off = inv_off; la.ltile_off = ltile_off;
        la.inv_q = inv_q; la.inv_dest = inv_dest; la.list_begin = L0; la.list_end = L1; la.group_shift = group_shift;
        la.tile_begin = lt_host[(size_t)L0]; la.n_tiles_max = lt_host[(size_t)L1] - lt_host[(size_t)L0];
        la.sims = sims; la.sims_base = base; la.sink = sims + need_fine;
        FAL_TRY(launch_list_scan(ctx, la));
        SelectArgs sa{};
        sa.sims = sims; sa.sims_base = base; sa.k = k_ann; sa.out_sim = sim; sa.out_idx = idx;
        sa.jobs = coarse_dev; sa.n_jobs = (int)coarse.size(); sa.tile_begin = t0;
        sa.n_probe = np; sa.probes = probes; sa.list_off = ivf->list_off; sa.q_sim_off = q_sim_off; sa.perm = ivf->perm;
        set_filter(sa, nf);
        FAL_TRY(launch_select(ctx, ST_SELECT, MODE_IVF, sa, (t1 - t0) * 32));
    }
    FAL_CHECK_HIP(hipStreamSynchronize(st));
    return FAL_OK;
}

extern "C" int fal_ivf_search_topk(fal_ctx* ctx, const fal_ivf* ivf, int n_probe, int k_ann, float* sim, int32_t* idx) {
    fal::CallScope _call(ctx);
    return search_impl(ctx, ivf, n_probe, k_ann, sim, idx, nullptr);
}

extern "C" int fal_ivf_search_neighbors(fal_ctx* ctx, const fal_ivf* ivf, int n_probe, int k_ann,
                                        const float* precursor_mz_sorted, const float* rt_sorted, double tol, int tol_is_da,
                                        double rt_tol, int n_neighbors, int32_t* nb_idx, float* nb_dist,
                                        int32_t* nb_count) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && ivf, FAL_EINVAL, "fal_ivf_search_neighbors: NULL ctx/ivf");
    FAL_REQUIRE(n_neighbors >= 1 && n_neighbors <= FAL_MAX_K_ANN, FAL_EUNSUPPORTED,
                "fal_ivf_search_neighbors: n_neighbors must be in [1, %d]", FAL_MAX_K_ANN);
    if (ivf->n == 0) return FAL_OK;
    FAL_REQUIRE(precursor_mz_sorted && nb_idx && nb_dist, FAL_EINVAL, "fal_ivf_search_neighbors: NULL array");
    NeighborFilter nf{precursor_mz_sorted, rt_tol >= 0.0 ? rt_sorted : nullptr, tol, rt_tol, tol_is_da, n_neighbors, nb_idx, nb_dist,
                      nb_count};
    ctx->stage_reset(ST_FILTER);
    return search_impl(ctx, ivf, n_probe, k_ann, nullptr, nullptr, &nf);
}
