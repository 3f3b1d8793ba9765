// f4: hierarchical clustering of the (re-scored) neighbour graph -- the clustering the reference snapshot ships:
// `fcluster(fastcluster.linkage(pdist, linkage), distance_threshold, "distance")` on the exact cosine distances of a
// block (reference cluster.py:283-290), here on the sparse neighbour graph with "missing pair = distance 1"
// (the reference's own convention for pairs below min_matched_peaks, cluster.py:621-626).
//
// Cut at a threshold t < 1 a flat cluster can only join spectra of one connected component of the graph of edges with
// d <= t (single: the components themselves; complete / average: a merge across two components has height 1).  So:
//   1. lock-free union-find over the stored edges with d <= t (both directions) -> components, root = lowest row;
//   2. single linkage: done.  complete / average: one wave per component of >= 2 rows -- dense matrix of the component
//      (d(i, j) = the smaller of the stored directions, 1 where neither is stored), naive agglomeration with the
//      Lance-Williams update in float64 (scipy's formulas), always merging the pair with the smallest height (ties:
//      lowest (a, b)), until the smallest height exceeds t;
//   3. clusters of one row become noise (-1: the reference's _postprocess_cluster drops groups < 2, cluster.py:441-454),
//      the others are numbered by their lowest row -- the interface of the DBSCAN stage (a9), so a10..a12 follow
//      unchanged.
// fastcluster is not available and scipy's tie order is not specified: PARITY UNPINNED for exact ties (the partition is
// identical whenever merge heights are distinct; tests/test_gpu_linkage.py against scipy.cluster.hierarchy).
#include <math.h>
#include <algorithm>
#include "common.h"
#include "ivf.h"
#include "util.h"

namespace fal {

// Connected groups of up to kLinkageWaveMax rows are agglomerated one wave per group (O(m^3 / 64): the whole matrix is
// searched for every merge); larger ones by a 1,024-thread workgroup that keeps every row's nearest partner cached
// (`lk_agglomerate_big_kernel`: O(m^2)-ish, the same merges in the same order).  There is no size cap (round 3 refused groups of
// more than 2,048 rows; the reference runs fastcluster on whole blocks of up to batch_size rows, cluster.py:277-290): a group is
// bounded by its bucket, a bucket by batch_size, and the m x m float64 matrix (8.6 GB at 32,768 rows) is the memory the
// reference's own condensed matrix takes twice over.
constexpr int kLinkageWaveMax = 256;

__device__ __forceinline__ int32_t lk_find(int32_t* parent, int32_t x) {
    int32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const int32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = g;
    }
    return x;
}

// the final pass's find: read-only (graph.hip, uf_find_final: halving stores of other threads could overwrite a row's root)
__device__ __forceinline__ int32_t lk_find_final(const int32_t* parent, int32_t x) {
    int32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

__device__ __forceinline__ void lk_union(int32_t* parent, int32_t a, int32_t b) {
    while (true) {
        a = lk_find(parent, a);
        b = lk_find(parent, b);
        if (a == b) return;
        if (a < b) { const int32_t t = a; a = b; b = t; }     // hook the larger root under the smaller
        if (atomicCAS(&parent[a], a, b) == a) return;
    }
}

__global__ void lk_init_kernel(int32_t* __restrict__ parent, int32_t* __restrict__ count, int32_t* __restrict__ rep, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        parent[i] = (int32_t)i;
        count[i] = 0;
        rep[i] = -1;
    }
}

__global__ void lk_edges_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist, int64_t n, int k,
                                float t, int32_t* __restrict__ parent) {
    const int64_t total = n * k;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / k;
        const int32_t j = nb_idx[e];
        if (j >= 0 && (int64_t)j < n && (int64_t)j != i && nb_dist[e] <= t) lk_union(parent, (int32_t)i, j);
    }
}

__global__ void lk_roots_kernel(int32_t* __restrict__ parent, int32_t* __restrict__ count, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t root = lk_find_final(parent, (int32_t)i);
        __hip_atomic_store(&parent[i], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&count[root], 1);
    }
}

// per row: 1 where the row is the root of a component of >= 2 rows; sizes / squared sizes for the member and matrix offsets
// (+ the list of the components beyond the one-wave form: big[0] = how many, big[1..] = their roots, any order)
__global__ void lk_comp_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ count, int64_t n,
                               int32_t* __restrict__ is_comp, int64_t* __restrict__ msz, int64_t* __restrict__ msq,
                               int32_t* __restrict__ big) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool c = parent[i] == (int32_t)i && count[i] >= 2;
        is_comp[i] = c;
        msz[i] = c ? count[i] : 0;
        msq[i] = c ? (int64_t)count[i] * count[i] : 0;
        if (c && count[i] > kLinkageWaveMax) big[1 + atomicAdd(&big[0], 1)] = (int32_t)i;
    }
}

__global__ void lk_fill_kernel(double* __restrict__ D, int64_t cells) {
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x) D[c] = 1.0;
}

__global__ void lk_scatter_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ count, int64_t n,
                                  const int64_t* __restrict__ moff, int32_t* __restrict__ cursor, int32_t* __restrict__ mem,
                                  const int64_t* __restrict__ comp_rank, int32_t* __restrict__ comp_root) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t root = parent[i];
        if (count[root] < 2) continue;
        mem[moff[root] + atomicAdd(&cursor[root], 1)] = (int32_t)i;
        if (root == (int32_t)i) comp_root[comp_rank[i]] = root;
    }
}

// single linkage: the component is the cluster
__global__ void lk_single_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ count, int64_t n,
                                 int32_t* __restrict__ rep) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        rep[i] = count[parent[i]] >= 2 ? parent[i] : -1;
}

// one wave per component: method 1 = complete, 2 = average
__global__ __launch_bounds__(64) void lk_agglomerate_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist,
                                                            int64_t n, int k, double t, int method, const int32_t* __restrict__ parent,
                                                            const int32_t* __restrict__ count, const int32_t* __restrict__ comp_root,
                                                            const int64_t* __restrict__ moff, const int64_t* __restrict__ qoff,
                                                            const int32_t* __restrict__ mem, int32_t* __restrict__ mem_sorted,
                                                            int32_t* __restrict__ lidx, int32_t* __restrict__ act,
                                                            int32_t* __restrict__ sz, int32_t* __restrict__ cl,
                                                            double* __restrict__ Dall, int32_t* __restrict__ rep) {
    const int lane = threadIdx.x;
    const int32_t root = comp_root[blockIdx.x];
    const int m = count[root];
    if (m > kLinkageWaveMax) return;                        // lk_agglomerate_big_kernel's
    const int64_t mo = moff[root];
    const int32_t* mu = mem + mo;
    int32_t* ms = mem_sorted + mo;
    int32_t* a_act = act + mo;
    int32_t* a_sz = sz + mo;
    int32_t* a_cl = cl + mo;
    double* D = Dall + qoff[root];
    // members in ascending row order (rank by counting); local index of every member row
    for (int x = lane; x < m; x += 64) {
        const int32_t rx = mu[x];
        int rank = 0;
        for (int y = 0; y < m; ++y) rank += mu[y] < rx;
        ms[rank] = rx;
        lidx[rx] = rank;
        a_act[x] = 1;
        a_sz[x] = 1;
        a_cl[x] = x;
    }
    // (D arrives filled with 1.0: missing pair = distance 1, lk_fill_kernel)
    __threadfence_block();
    __syncthreads();
    for (int a = 0; a < m; ++a) {
        const int64_t row = ms[a];
        for (int s = lane; s < k; s += 64) {
            const int32_t j = nb_idx[row * k + s];
            if (j >= 0 && (int64_t)j < n && (int64_t)j != row && parent[j] == root) D[(int64_t)a * m + lidx[j]] = (double)nb_dist[row * k + s];
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int64_t c = lane; c < (int64_t)m * m; c += 64) {                          // d(i, j) = the smaller stored direction
        const int a = (int)(c / m), b = (int)(c % m);
        if (a < b) {
            const double v = fmin(D[c], D[(int64_t)b * m + a]);
            D[c] = v;
            D[(int64_t)b * m + a] = v;
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int step = 0; step < m - 1; ++step) {
        double bv = INFINITY;
        int ba = 0x7fffffff, bb = 0x7fffffff;
        for (int a = lane; a < m; a += 64) {
            if (!a_act[a]) continue;
            for (int b = a + 1; b < m; ++b) {
                if (!a_act[b]) continue;
                const double v = D[(int64_t)a * m + b];
                if (v < bv || (v == bv && (a < ba || (a == ba && b < bb)))) {
                    bv = v;
                    ba = a;
                    bb = b;
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int oa = __shfl_xor(ba, off, 64), ob = __shfl_xor(bb, off, 64);
            if (ov < bv || (ov == bv && (oa < ba || (oa == ba && ob < bb)))) {
                bv = ov;
                ba = oa;
                bb = ob;
            }
        }
        if (!(bv <= t)) break;                                                      // fcluster(Z, t, "distance")
        const double sa = (double)a_sz[ba], sb = (double)a_sz[bb];
        for (int c = lane; c < m; c += 64) {
            if (!a_act[c] || c == ba || c == bb) continue;
            const double dac = D[(int64_t)ba * m + c], dbc = D[(int64_t)bb * m + c];
            const double nv = method == 1 ? fmax(dac, dbc) : (sa * dac + sb * dbc) / (sa + sb);   // scipy's Lance-Williams forms
            D[(int64_t)ba * m + c] = nv;
            D[(int64_t)c * m + ba] = nv;
        }
        for (int x = lane; x < m; x += 64)
            if (a_cl[x] == bb) a_cl[x] = ba;
        __threadfence_block();
        __syncthreads();
        if (lane == 0) {
            a_act[bb] = 0;
            a_sz[ba] += a_sz[bb];
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int x = lane; x < m; x += 64) {
        const int c = a_cl[x];                              // representative = the cluster's lowest member (a < b in every merge)
        rep[ms[x]] = a_sz[c] >= 2 ? ms[c] : -1;
    }
}


// Groups beyond the one-wave form: one workgroup of 16 waves per group.  The same agglomeration -- always the pair with the
// smallest height, ties -> lowest (a, b), Lance-Williams updates in float64 -- with the search made incremental: every active
// row a keeps its nearest partner among the active b > a (`nnv`, `nni`: smallest value, ties -> lowest b), so the global
// minimum with the (value, a, b) tie order is a reduction over m cached entries; after a merge only the rows whose cached
// partner was one of the merged pair are searched again (a wave per row), the others are updated in place (a new value can
// only tie the cached one: complete and average linkage are reducible, d(a + b, c) >= min(d(a, c), d(b, c))).
__global__ __launch_bounds__(1024) void lk_agglomerate_big_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist,
                                                                  int64_t n, int k, double t, int method, const int32_t* __restrict__ parent,
                                                                  const int32_t* __restrict__ count, const int32_t* __restrict__ big,
                                                                  const int64_t* __restrict__ moff, const int64_t* __restrict__ qoff,
                                                                  const int32_t* __restrict__ mem, int32_t* __restrict__ mem_sorted,
                                                                  int32_t* __restrict__ lidx, int32_t* __restrict__ act,
                                                                  int32_t* __restrict__ sz, int32_t* __restrict__ cl,
                                                                  int32_t* __restrict__ nni_all, int32_t* __restrict__ todo_all,
                                                                  double* __restrict__ nnv_all, double* __restrict__ Dall,
                                                                  int32_t* __restrict__ rep) {
    constexpr int T = 1024, W = T / 64;
    __shared__ double s_v[W];
    __shared__ int32_t s_a[W];
    __shared__ double s_bv;
    __shared__ int32_t s_ba, s_bb, s_ntodo;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int32_t root = big[1 + blockIdx.x];
    const int m = count[root];
    const int64_t mo = moff[root];
    const int32_t* mu = mem + mo;
    int32_t* ms = mem_sorted + mo;
    int32_t* a_act = act + mo;
    int32_t* a_sz = sz + mo;
    int32_t* a_cl = cl + mo;
    int32_t* nni = nni_all + mo;
    int32_t* todo = todo_all + mo;
    double* nnv = nnv_all + mo;
    double* D = Dall + qoff[root];
    auto sync = [&]() {
        __threadfence_block();
        __syncthreads();
    };
    for (int x = tid; x < m; x += T) {                      // members in ascending row order (rank by counting)
        const int32_t rx = mu[x];
        int rank = 0;
        for (int y = 0; y < m; ++y) rank += mu[y] < rx;
        ms[rank] = rx;
        lidx[rx] = rank;
        a_act[x] = 1;
        a_sz[x] = 1;
        a_cl[x] = x;
    }
    sync();
    for (int64_t e = tid; e < (int64_t)m * k; e += T) {     // stored edges inside the group (D arrives filled with 1.0)
        const int a = (int)(e / k), s = (int)(e % k);
        const int64_t row = ms[a];
        const int32_t j = nb_idx[row * k + s];
        if (j >= 0 && (int64_t)j < n && (int64_t)j != row && parent[j] == root) D[(int64_t)a * m + lidx[j]] = (double)nb_dist[row * k + s];
    }
    sync();
    for (int64_t c = tid; c < (int64_t)m * m; c += T) {     // d(i, j) = the smaller stored direction
        const int a = (int)(c / m), b = (int)(c % m);
        if (a < b) {
            const double v = fmin(D[c], D[(int64_t)b * m + a]);
            D[c] = v;
            D[(int64_t)b * m + a] = v;
        }
    }
    sync();
    // nearest active partner b > a of row a: smallest value, ties -> lowest b (one wave per row)
    auto search_row = [&](int a) {
        double bv = INFINITY;
        int bb = 0x7fffffff;
        const double* Da = D + (int64_t)a * m;
        for (int b = a + 1 + lane; b < m; b += 64) {
            if (!a_act[b]) continue;
            const double v = Da[b];
            if (v < bv) {                                   // (b ascends inside a lane: "<" keeps the lowest b)
                bv = v;
                bb = b;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int ob = __shfl_xor(bb, off, 64);
            if (ov < bv || (ov == bv && ob < bb)) {
                bv = ov;
                bb = ob;
            }
        }
        if (lane == 0) {
            nnv[a] = bv;
            nni[a] = bb;
        }
    };
    for (int a = w; a < m; a += W) search_row(a);
    sync();
    for (int step = 0; step < m - 1; ++step) {
        double bv = INFINITY;
        int ba = 0x7fffffff;
        for (int a = tid; a < m; a += T) {
            if (!a_act[a]) continue;
            const double v = nnv[a];
            if (v < bv) {                                   // (a ascends inside a thread)
                bv = v;
                ba = a;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int oa = __shfl_xor(ba, off, 64);
            if (ov < bv || (ov == bv && oa < ba)) {
                bv = ov;
                ba = oa;
            }
        }
        if (lane == 0) {
            s_v[w] = bv;
            s_a[w] = ba;
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < W; ++i)
                if (s_v[i] < bv || (s_v[i] == bv && s_a[i] < ba)) {
                    bv = s_v[i];
                    ba = s_a[i];
                }
            s_bv = bv;
            s_ba = ba;
            s_bb = ba < m ? nni[ba] : 0x7fffffff;
            s_ntodo = 0;
        }
        __syncthreads();
        bv = s_bv;
        ba = s_ba;
        const int bb = s_bb;
        if (!(bv <= t)) break;                              // fcluster(Z, t, "distance")
        const double sa = (double)a_sz[ba], sb = (double)a_sz[bb];
        for (int c = tid; c < m; c += T) {
            if (a_cl[c] == bb) a_cl[c] = ba;
            if (!a_act[c] || c == ba || c == bb) continue;
            const double dac = D[(int64_t)ba * m + c], dbc = D[(int64_t)bb * m + c];
            const double nv = method == 1 ? fmax(dac, dbc) : (sa * dac + sb * dbc) / (sa + sb);   // scipy's Lance-Williams forms
            D[(int64_t)ba * m + c] = nv;
            D[(int64_t)c * m + ba] = nv;
            if (c < bb) {                                   // rows above bb never had ba or bb among their partners b > c
                const int32_t p = nni[c];
                if (p == bb || (p == ba && c < ba)) todo[atomicAdd(&s_ntodo, 1)] = c;
                else if (c < ba && (nv < nnv[c] || (nv == nnv[c] && ba < p))) {
                    nnv[c] = nv;
                    nni[c] = ba;
                }
            }
        }
        sync();
        if (tid == 0) {
            a_act[bb] = 0;
            a_sz[ba] += a_sz[bb];
            todo[s_ntodo++] = ba;
        }
        sync();
        const int nt = s_ntodo;
        for (int i = w; i < nt; i += W) search_row(todo[i]);
        sync();
    }
    for (int x = tid; x < m; x += T) {
        const int c = a_cl[x];                              // representative = the cluster's lowest member (a < b in every merge)
        rep[ms[x]] = a_sz[c] >= 2 ? ms[c] : -1;
    }
}

__global__ void lk_isrep_kernel(const int32_t* __restrict__ rep, int64_t n, int32_t* __restrict__ is_rep) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        is_rep[i] = rep[i] == (int32_t)i;
}

__global__ void lk_label_kernel(const int32_t* __restrict__ rep, const int64_t* __restrict__ rank, int64_t n,
                                int32_t* __restrict__ labels) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        labels[i] = rep[i] >= 0 ? (int32_t)rank[rep[i]] : -1;
}

// labels + cluster count on the device (the interface of dbscan_dev); synchronises once for the scratch sizes of the
// complete / average forms
int linkage_dev(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float t, int method,
                int32_t* labels, int64_t** d_count_out) {
    hipStream_t st = ctx->stream;
    int32_t* buf = nullptr;
    int64_t* rank = nullptr;
    FAL_TRY(ctx->reserve(SLOT_DB, sizeof(int32_t) * (size_t)n * 4, (void**)&buf));
    FAL_TRY(ctx->reserve(SLOT_DB2, sizeof(int64_t) * (size_t)(n + 1), (void**)&rank));
    int32_t *parent = buf, *count = buf + n, *rep = buf + 2 * n, *flag = buf + 3 * n;
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16);
    const int egrid = (int)std::min<int64_t>(ceil_div(n * k, 256), (int64_t)ctx->num_cus * 32);
    ctx->stage_reset(ST_DBSCAN);
    StageScope ts(ctx, ST_DBSCAN);
    hipLaunchKernelGGL(lk_init_kernel, dim3(grid), dim3(256), 0, st, parent, count, rep, n);
    hipLaunchKernelGGL(lk_edges_kernel, dim3(egrid), dim3(256), 0, st, nb_idx, nb_dist, n, k, t, parent);
    hipLaunchKernelGGL(lk_roots_kernel, dim3(grid), dim3(256), 0, st, parent, count, n);
    if (method == 0) {
        hipLaunchKernelGGL(lk_single_kernel, dim3(grid), dim3(256), 0, st, parent, count, n, rep);
    } else {
        int64_t *msz = nullptr, *moff = nullptr, *qoff = nullptr, *crank = nullptr;
        FAL_TRY(ctx->reserve(SLOT_TAIL, sizeof(int64_t) * (size_t)(4 * (n + 1)), (void**)&msz));
        int64_t* msq = msz + (n + 1);
        moff = msq + (n + 1);
        qoff = moff + (n + 1);
        const int64_t big_cap = n / kLinkageWaveMax + 2;    // components of more than kLinkageWaveMax rows: at most n / that many
        FAL_TRY(ctx->reserve(SLOT_TAIL2, sizeof(int64_t) * (size_t)(n + 1) + sizeof(int32_t) * (size_t)(big_cap + 2), (void**)&crank));
        int32_t* big = reinterpret_cast<int32_t*>(crank + n + 1);           // (behind the n + 1 words of the scan)
        FAL_CHECK_HIP(hipMemsetAsync(big, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(lk_comp_kernel, dim3(grid), dim3(256), 0, st, parent, count, n, flag, msz, msq, big);
        FAL_TRY(device_scan_i32(ctx, flag, n, crank, SLOT_DB3));
        FAL_TRY(device_scan_i64(ctx, msz, n, moff, SLOT_DB3));
        FAL_TRY(device_scan_i64(ctx, msq, n, qoff, SLOT_DB3));
        int64_t tot[3] = {0, 0, 0};
        int32_t n_big = 0;
        FAL_CHECK_HIP(hipMemcpyAsync(&n_big, big, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        FAL_CHECK_HIP(hipMemcpyAsync(&tot[0], crank + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        FAL_CHECK_HIP(hipMemcpyAsync(&tot[1], moff + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        FAL_CHECK_HIP(hipMemcpyAsync(&tot[2], qoff + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        FAL_CHECK_HIP(hipStreamSynchronize(st));
        const int64_t n_comp = tot[0], n_mem = tot[1], n_sq = tot[2];
        if (n_comp > 0) {
            int32_t* ibuf = nullptr;
            double* D = nullptr;
            FAL_TRY(ctx->reserve(SLOT_TAIL3, sizeof(int32_t) * (size_t)(7 * n_mem + 2 * n + n_comp + 16), (void**)&ibuf));
            const int rc = ctx->reserve(SLOT_TAIL4, sizeof(double) * (size_t)(n_sq + n_mem + 16), (void**)&D);
            if (rc != FAL_OK) {
                set_error("hierarchical clustering (complete / average): the connected groups within the distance threshold need "
                          "%.1f GB of pair distances (sum of squared group sizes %lld); single linkage or DBSCAN take the same graph",
                          8e-9 * (double)n_sq, (long long)n_sq);
                return rc;
            }
            int32_t *mem = ibuf, *mem_sorted = mem + n_mem, *act = mem_sorted + n_mem, *sz = act + n_mem, *cl = sz + n_mem;
            int32_t *nni = cl + n_mem, *todo = nni + n_mem;
            int32_t *lidx = todo + n_mem, *cursor = lidx + n, *comp_root = cursor + n;
            double* nnv = D + n_sq;
            FAL_CHECK_HIP(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)n, st));
            hipLaunchKernelGGL(lk_fill_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n_sq, 256), (int64_t)ctx->num_cus * 32)), dim3(256), 0, st,
                               D, n_sq);
            hipLaunchKernelGGL(lk_scatter_kernel, dim3(grid), dim3(256), 0, st, parent, count, n, moff, cursor, mem, crank, comp_root);
            hipLaunchKernelGGL(lk_agglomerate_kernel, dim3((unsigned)n_comp), dim3(64), 0, st, nb_idx, nb_dist, n, k, (double)t, method,
                               parent, count, comp_root, moff, qoff, mem, mem_sorted, lidx, act, sz, cl, D, rep);
            if (n_big > 0)
                hipLaunchKernelGGL(lk_agglomerate_big_kernel, dim3((unsigned)n_big), dim3(1024), 0, st, nb_idx, nb_dist, n, k, (double)t,
                                   method, parent, count, big, moff, qoff, mem, mem_sorted, lidx, act, sz, cl, nni, todo, nnv, D, rep);
        }
    }
    hipLaunchKernelGGL(lk_isrep_kernel, dim3(grid), dim3(256), 0, st, rep, n, flag);
    FAL_TRY(device_scan_i32(ctx, flag, n, rank, SLOT_DB3));
    hipLaunchKernelGGL(lk_label_kernel, dim3(grid), dim3(256), 0, st, rep, rank, n, labels);
    FAL_CHECK_HIP(hipGetLastError());
    *d_count_out = rank + n;
    // the agglomeration's work arrays are dead on the host side (their kernels are enqueued): the tail stage reuses the slots
    for (int slot : {SLOT_TAIL, SLOT_TAIL2, SLOT_TAIL3, SLOT_TAIL4, SLOT_DB3}) ctx->release(slot);
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::lk_init_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

extern "C" int fal_linkage_cluster(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float threshold,
                                   int method, int32_t* labels, int64_t* n_clusters) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && k >= 1 && n < (int64_t)INT32_MAX, FAL_EINVAL, "fal_linkage_cluster: bad argument");
    FAL_REQUIRE(method >= 0 && method <= 2, FAL_EINVAL, "fal_linkage_cluster: method must be 0 (single), 1 (complete) or 2 (average)");
    FAL_REQUIRE(threshold < 1.0f, FAL_EUNSUPPORTED, "fal_linkage_cluster: the threshold must be below 1 (the distance of a missing pair)");
    if (n_clusters) *n_clusters = 0;
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(nb_idx && nb_dist && labels, FAL_EINVAL, "fal_linkage_cluster: NULL array");
    int64_t* d_count = nullptr;
    FAL_TRY(linkage_dev(ctx, nb_idx, nb_dist, n, k, threshold, method, labels, &d_count));
    if (n_clusters) {
        FAL_CHECK_HIP(hipMemcpyAsync(n_clusters, d_count, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return FAL_OK;
}
