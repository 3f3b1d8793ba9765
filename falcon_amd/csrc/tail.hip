// a10 precursor / RT refinement of the DBSCAN clusters, a11 medoids, a12 label globalisation.
//
// Reference: falcon/cluster/cluster.py:362-455 (_postprocess_cluster), 458-509 (_linkage),
// 512-553 (_get_cluster_medoids), 556-590 + 144-155 (global labels), with the flat cut
// scipy.cluster.hierarchy.fcluster(Z, t, "distance") restated (cluster numbering included, it
// matters for the m/z x RT combination of cluster.py:418-429).
//
// One wave per DBSCAN cluster.  Members are the cluster's rows in ascending (precursor-sorted)
// row order.  All per-cluster working arrays live in a global scratch slab at the cluster's own
// offsets (clusters are tiny on average, so this stays in L2); every loop is strided over the
// 64 lanes, so one huge cluster costs O(m^2 / 64), not O(m^2).
#include <math.h>
#include "common.h"
#include "ivf.h"
#include "util.h"

namespace fal {

struct RefineScratch {
    float* val;      // [n] the values being linked (m/z, then RT) in member order
    int32_t* ord;    // [n] member index by ascending value (stable)
    float* smin;     // [n] current segments, left to right
    float* smax;
    int32_t* sid;    // linkage node id of the segment (leaf = member index, merged = m + it)
    int32_t* zl;     // [n] linkage rows: left child, right child, max height in subtree
    int32_t* zr;
    double* zmd;
    int32_t* t_a;    // [n] flat cluster number of each member from m/z
    int32_t* t_b;    // [n] ... from RT
    int32_t* stack;  // [n] traversal stack / scratch
    int32_t* visit;  // [n] traversal visited flags / scratch
};

__device__ __forceinline__ double link_dist(float hi, float lo, bool ppm) {
    const float diff = hi - lo;                           // cluster.py:488 (float32)
    return ppm ? (double)(diff / lo) * 1e6 : (double)diff;   // cluster.py:489-490
}

// Complete-linkage dendrogram of the m values val[0..m) (cluster.py:458-509) and its flat cut at
// threshold t numbered like scipy's fcluster (minus 1), into T[0..m).
// exact = false: stop at the first merge above t and number the segments left to right (same
// partition; the numbering is then irrelevant to the caller).
__device__ void flat_cut_1d(const RefineScratch& S, int64_t o, int m, double t, bool ppm, bool exact,
                            int32_t* __restrict__ T, int lane) {
    const float* val = S.val + o;
    int32_t* ord = S.ord + o;
    float *smin = S.smin + o, *smax = S.smax + o;
    int32_t *sid = S.sid + o, *zl = S.zl + o, *zr = S.zr + o, *stack = S.stack + o, *visit = S.visit + o;
    double* zmd = S.zmd + o;
    // stable argsort by counting
    for (int e = lane; e < m; e += 64) {
        const float v = val[e];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const float w = val[j];
            rank += (w < v) || (w == v && j < e);
        }
        ord[rank] = e;
    }
    __syncthreads();
    for (int s = lane; s < m; s += 64) {
        const int e = ord[s];
        smin[s] = smax[s] = val[e];
        sid[s] = e;
    }
    __syncthreads();
    int nseg = m, n_merge = 0;
    for (int it = 0; it < m - 1; ++it) {
        // leftmost minimum of dist(s) = smax[s+1] - smin[s]  (cluster.py:486-492)
        double best = INFINITY;
        int bs = 0x7fffffff;
        for (int s = lane; s < nseg - 1; s += 64) {
            const double dd = link_dist(smax[s + 1], smin[s], ppm);
            if (dd < best) {
                best = dd;
                bs = s;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int os = __shfl_xor(bs, off, 64);
            if (ob < best || (ob == best && os < bs)) {
                best = ob;
                bs = os;
            }
        }
        if (bs == 0x7fffffff) bs = 0;           // all distances NaN: take the first pair like np.inf init would not; keep going
        if (!exact && !(best <= t)) break;
        if (lane == 0) {
            const int l = sid[bs], r = sid[bs + 1];
            double md = best;
            if (l >= m) md = fmax(md, zmd[l - m]);
            if (r >= m) md = fmax(md, zmd[r - m]);
            zl[it] = l;
            zr[it] = r;
            zmd[it] = md;
            smax[bs] = smax[bs + 1];
            sid[bs] = m + it;
        }
        __syncthreads();
        // delete segment bs+1: shift the tail left by one, 64 at a time in ascending order
        for (int s0 = bs + 1; s0 < nseg - 1; s0 += 64) {
            const int s = s0 + lane;
            float a = 0.f, b = 0.f;
            int c = 0;
            const bool on = s < nseg - 1;
            if (on) {
                a = smin[s + 1];
                b = smax[s + 1];
                c = sid[s + 1];
            }
            __syncthreads();
            if (on) {
                smin[s] = a;
                smax[s] = b;
                sid[s] = c;
            }
            __syncthreads();
        }
        --nseg;
        ++n_merge;
    }
    if (!exact) {
        // Surviving segments are contiguous runs of the sorted order; number them left to right.
        // The run length of a segment = number of leaves under its linkage node.
        if (lane == 0) {
            for (int it = 0; it < n_merge; ++it) {
                const int l = zl[it], r = zr[it];
                stack[it] = (l >= m ? stack[l - m] : 1) + (r >= m ? stack[r - m] : 1);
            }
            int p = 0;
            for (int s = 0; s < nseg; ++s) {
                const int id = sid[s];
                const int sz = id >= m ? stack[id - m] : 1;
                for (int q = 0; q < sz; ++q) visit[p++] = s;
            }
        }
        __syncthreads();
        for (int p = lane; p < m; p += 64) T[ord[p]] = visit[p];
        __syncthreads();
        return;
    }
    // scipy _hierarchy.cluster_monocrit(Z, MD, T, cutoff, n): depth-first from the root, left
    // child first; a subtree whose max height <= cutoff becomes one flat cluster; numbers are
    // handed out in visiting order.
    if (lane == 0) {
        for (int i = 0; i < m - 1; ++i) visit[i] = 0;
        int k = 0, n_cluster = 0, leader = -1;
        stack[0] = 2 * m - 2;
        while (k >= 0) {
            const int root = stack[k] - m;
            const int lc = zl[root], rc = zr[root];
            if (leader == -1 && zmd[root] <= t) {
                leader = root;
                ++n_cluster;
            }
            if (lc >= m && !visit[lc - m]) {
                visit[lc - m] = 1;
                stack[++k] = lc;
                continue;
            }
            if (rc >= m && !visit[rc - m]) {
                visit[rc - m] = 1;
                stack[++k] = rc;
                continue;
            }
            if (lc < m) {
                if (leader == -1) ++n_cluster;
                T[lc] = n_cluster - 1;
            }
            if (rc < m) {
                if (leader == -1) ++n_cluster;
                T[rc] = n_cluster - 1;
            }
            if (leader == root) leader = -1;
            --k;
        }
    }
    __syncthreads();
}

// ascending in-place sort of the m member rows of one cluster (rank by counting through `tmp`)
__device__ __forceinline__ void sort_rows(int32_t* __restrict__ rows, int32_t* __restrict__ tmp, int m, int lane) {
    for (int e = lane; e < m; e += 64) tmp[e] = rows[e];
    __syncthreads();
    for (int e = lane; e < m; e += 64) {
        const int32_t v = tmp[e];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += tmp[j] < v;
        rows[rank] = v;
    }
    __syncthreads();
}

// one wave per DBSCAN cluster c (grid-stride; the cluster count lives on the device):
// rows[seg[c] .. seg[c+1]) are its members, scattered in arbitrary order by member_scatter_kernel.
__global__ __launch_bounds__(64) void refine_kernel(int32_t* __restrict__ rows, const int64_t* __restrict__ seg,
                                                    const int64_t* __restrict__ d_count,
                                                    const float* __restrict__ mz, const float* __restrict__ rt,
                                                    double tol, int is_da, double rt_tol, RefineScratch S,
                                                    int32_t* __restrict__ sub, int32_t* __restrict__ n_sub) {
    const int lane = threadIdx.x;
    const int64_t C = *d_count;
    const bool use_rt = rt != nullptr && rt_tol >= 0.0;
    // Clusters of up to kLdsMembers members (nearly all of them) keep every work array in LDS: the phases below are
    // chains of write -> barrier -> read, each a ~1-2 us round trip through global memory but ~0.1 us through LDS
    // (measured: the kernel is latency-bound, 0.45 -> see DESIGN).  Larger clusters use the global scratch arrays.
    constexpr int kLdsMembers = 64;
    __shared__ __align__(8) unsigned char lds[kLdsMembers * (8 + 11 * 4)];
    RefineScratch L;
    {
        L.zmd = reinterpret_cast<double*>(lds);
        float* f = reinterpret_cast<float*>(lds + 8 * kLdsMembers);
        L.val = f;
        L.smin = f + kLdsMembers;
        L.smax = f + 2 * kLdsMembers;
        int32_t* q = reinterpret_cast<int32_t*>(f + 3 * kLdsMembers);
        L.ord = q;
        L.sid = q + kLdsMembers;
        L.zl = q + 2 * kLdsMembers;
        L.zr = q + 3 * kLdsMembers;
        L.t_a = q + 4 * kLdsMembers;
        L.t_b = q + 5 * kLdsMembers;
        L.stack = q + 6 * kLdsMembers;
        L.visit = q + 7 * kLdsMembers;
    }
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int64_t o = seg[c];
        const int m = (int)(seg[c + 1] - o);
        int32_t* out = sub + o;
        __syncthreads();
        if (m < 2) {                                   // cluster.py:399-402 (min_samples = 2)
            for (int e = lane; e < m; e += 64) out[e] = -1;
            if (lane == 0) n_sub[c] = 0;
            continue;
        }
        // Round 5: the common case first -- every member within the tolerance of every other.  The root of the 1-D complete-
        // linkage dendrogram is the pair (largest, smallest value) and its height link_dist(max, min) bounds every other
        // merge (float32 difference and division are monotone), so fcluster(Z, t, "distance") returns ONE flat cluster
        // exactly when link_dist(max, min) <= t (cluster.py:433-435: every member keeps the DBSCAN cluster) -- the same
        // arithmetic, no margin.  Two loads per member, two wave reductions; the sort, the dendrogram and the numbering
        // (~1,500 instructions per cluster: the kernel was instruction-bound) run only for clusters that do split.
        {
            float vmin = INFINITY, vmax = -INFINITY, rmin = INFINITY, rmax = -INFINITY;
            bool bad = false;
            for (int e = lane; e < m; e += 64) {
                const int32_t rr = rows[o + e];
                const float v = mz[rr];
                bad = bad || !(v == v);
                vmin = fminf(vmin, v);
                vmax = fmaxf(vmax, v);
                if (use_rt) {
                    const float u = rt[rr];
                    bad = bad || !(u == u);
                    rmin = fminf(rmin, u);
                    rmax = fmaxf(rmax, u);
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                vmin = fminf(vmin, __shfl_xor(vmin, off, 64));
                vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
                rmin = fminf(rmin, __shfl_xor(rmin, off, 64));
                rmax = fmaxf(rmax, __shfl_xor(rmax, off, 64));
            }
            bool one = !__any(bad) && link_dist(vmax, vmin, !is_da) <= tol;
            if (use_rt) one = one && link_dist(rmax, rmin, false) <= rt_tol;
            if (one) {                                 // (uniform over the wave)
                for (int e = lane; e < m; e += 64) out[e] = 0;
                if (lane == 0) n_sub[c] = 1;
                continue;
            }
        }
        const bool small = m <= kLdsMembers;
        const RefineScratch& W = small ? L : S;        // work arrays of this cluster ...
        const int64_t wo = small ? 0 : o;              // ... and where they start
        sort_rows(rows + o, W.stack + wo, m, lane);    // members in ascending (precursor-sorted) row order
        int32_t* A = W.t_a + wo;
        for (int e = lane; e < m; e += 64) W.val[wo + e] = mz[rows[o + e]];
        __syncthreads();
        flat_cut_1d(W, wo, m, tol, !is_da, use_rt, A, lane);
        if (use_rt) {
            int32_t* B = W.t_b + wo;
            for (int e = lane; e < m; e += 64) W.val[wo + e] = rt[rows[o + e]];
            __syncthreads();
            flat_cut_1d(W, wo, m, rt_tol, false, true, B, lane);
            // cluster.py:423-429: np.unique(a * 2 + b * 3, return_inverse=True)[1]
            int32_t* V = W.stack + wo;
            int32_t* first = W.visit + wo;
            for (int e = lane; e < m; e += 64) V[e] = A[e] * 2 + B[e] * 3;
            __syncthreads();
            for (int e = lane; e < m; e += 64) {
                bool f = true;
                for (int j = 0; j < e && f; ++j) f = V[j] != V[e];
                first[e] = f;
            }
            __syncthreads();
            for (int e = lane; e < m; e += 64) {
                int r = 0;
                for (int j = 0; j < m; ++j) r += first[j] && V[j] < V[e];
                A[e] = r;
            }
            __syncthreads();
        }
        // number of flat clusters = max + 1
        int mx = 0;
        for (int e = lane; e < m; e += 64) mx = max(mx, A[e]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
        const int n_flat = mx + 1;
        if (n_flat == 1) {                              // cluster.py:433-435
            for (int e = lane; e < m; e += 64) out[e] = 0;
            if (lane == 0) n_sub[c] = 1;
            continue;
        }
        if (n_flat == m) {                              // cluster.py:436-439
            for (int e = lane; e < m; e += 64) out[e] = -1;
            if (lane == 0) n_sub[c] = 0;
            continue;
        }
        // cluster.py:441-454: groups with < 2 members -> -1, the others numbered by first occurrence
        int32_t* fo = W.ord + wo;     // first occurrence (member index) of the member's group, -1 if the group is too small
        for (int e = lane; e < m; e += 64) {
            int cnt = 0, f = -1;
            for (int j = 0; j < m; ++j) {
                if (A[j] == A[e]) {
                    if (f < 0) f = j;
                    ++cnt;
                }
            }
            fo[e] = cnt >= 2 ? f : -1;
        }
        __syncthreads();
        int total = 0;
        for (int e = lane; e < m; e += 64) {
            int id = -1;
            if (fo[e] >= 0) {
                id = 0;
                for (int j = 0; j < fo[e]; ++j) id += fo[j] == j;     // kept groups that start earlier
            }
            out[e] = id;
            total += fo[e] == e;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
        if (lane == 0) n_sub[c] = total;
    }
}

// members of every DBSCAN cluster, grouped: histogram -> exclusive scan -> scatter
__global__ void label_hist_kernel(const int32_t* __restrict__ labels, int64_t n, int32_t* __restrict__ counts) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (labels[i] >= 0) atomicAdd(&counts[labels[i]], 1);
}

__global__ void member_scatter_kernel(const int32_t* __restrict__ labels, int64_t n, const int64_t* __restrict__ seg,
                                      int32_t* __restrict__ cursor, int32_t* __restrict__ rows) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t l = labels[i];
        if (l >= 0) rows[seg[l] + atomicAdd(&cursor[l], 1)] = (int32_t)i;
    }
}

// one wave per cluster: final label = (clusters kept by earlier DBSCAN clusters) + sub id
__global__ __launch_bounds__(64) void relabel_kernel(const int32_t* __restrict__ rows, const int64_t* __restrict__ seg,
                                                     const int64_t* __restrict__ d_count,
                                                     const int32_t* __restrict__ sub, const int64_t* __restrict__ base,
                                                     int32_t* __restrict__ labels) {
    const int64_t C = *d_count;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int64_t o = seg[c];
        const int m = (int)(seg[c + 1] - o);
        for (int e = threadIdx.x; e < m; e += 64) labels[rows[o + e]] = sub[o + e] >= 0 ? (int32_t)(base[c] + sub[o + e]) : -1;
    }
}

// ------------------------------------------------------------------------------------------
// a11 / a12
// ------------------------------------------------------------------------------------------
__global__ void cluster_size_kernel(const int32_t* __restrict__ labels, int64_t n, int32_t* __restrict__ size,
                                    const int64_t* __restrict__ row_order, int32_t* __restrict__ noise_by_dataset_row) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t l = labels[i];
        if (l >= 0) atomicAdd(&size[l], 1);
        noise_by_dataset_row[row_order[i]] = l < 0;
    }
}

// score_i = sum (float32, stored order) of dist to stored same-cluster neighbours + 1.0 per
// member the row does not store (cluster.py:536-550 on the sparse graph); argmin per cluster
// with ties to the lowest row, via a 64-bit atomic min of (score bits, row).
__global__ __launch_bounds__(256) void medoid_score_kernel(const int32_t* __restrict__ labels, int64_t n,
                                                           const int32_t* __restrict__ nb_idx,
                                                           const float* __restrict__ nb_dist, int k,
                                                           const int32_t* __restrict__ extent,
                                                           const int32_t* __restrict__ size,
                                                           unsigned long long* __restrict__ best) {
    // one wave per row: coalesced row read, same-cluster slots found by ballot, then their
    // distances added ONE BY ONE in slot order (float32) so the sum equals the oracle's
    const int lane = threadIdx.x & 63;
    for (int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int32_t l = labels[i];
        if (l < 0) continue;
        float s = 0.f;
        int same = 0;
        const int ext = extent ? extent[i] : k;           // (a9's row extents: nothing is stored beyond)
        for (int c0 = 0; c0 < ext; c0 += 64) {
            const int c = c0 + lane;
            int32_t j = -1;
            float dv = 0.f;
            if (c < ext) {
                j = nb_idx[i * k + c];
                dv = nb_dist[i * k + c];
            }
            const bool in = j >= 0 && (int64_t)j < n && (int64_t)j != i && labels[j] == l;
            uint64_t mask = __ballot(in);
            same += __popcll(mask);
            while (mask) {
                const int b = __ffsll((unsigned long long)mask) - 1;
                mask &= mask - 1;
                s += __shfl(dv, b, 64);
            }
        }
        if (lane == 0) {
            s += (float)(size[l] - 1 - same);
            const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(uint32_t)i;
            atomicMin(&best[l], key);
        }
    }
}

// the same for rows of known extent (a9 hands them over): one THREAD per row walks the few stored slots in order -- a wave
// per row leaves most of its lanes idle on rows of a handful of neighbours
__global__ __launch_bounds__(256) void medoid_score_rows_kernel(const int32_t* __restrict__ labels, int64_t n,
                                                                const int32_t* __restrict__ nb_idx,
                                                                const float* __restrict__ nb_dist, int k,
                                                                const int32_t* __restrict__ extent,
                                                                const int32_t* __restrict__ size,
                                                                unsigned long long* __restrict__ best) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t l = labels[i];
        if (l < 0) continue;
        float s = 0.f;
        int same = 0;
        const int ext = min(extent[i], k);
        for (int c = 0; c < ext; ++c) {
            const int32_t j = nb_idx[i * k + c];
            if (j >= 0 && (int64_t)j < n && (int64_t)j != i && labels[j] == l) {
                s += nb_dist[i * k + c];                        // (slot order, float32: the oracle's sum)
                ++same;
            }
        }
        s += (float)(size[l] - 1 - same);
        const unsigned long long key = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(uint32_t)i;
        atomicMin(&best[l], key);
    }
}

__global__ void finalize_kernel(const int32_t* __restrict__ labels, int64_t n, const int64_t* __restrict__ d_count,
                                const int64_t* __restrict__ row_order, const int64_t* __restrict__ noise_rank,
                                const unsigned long long* __restrict__ best, int32_t* __restrict__ labels_out,
                                int32_t* __restrict__ medoids_out) {
    const int64_t n_clusters = *d_count;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ds = row_order[i];
        const int32_t l = labels[i];
        if (l >= 0) {
            labels_out[ds] = l;
        } else {
            const int64_t r = n_clusters + noise_rank[ds];
            labels_out[ds] = (int32_t)r;
            medoids_out[r] = (int32_t)ds;
        }
        if (i < n_clusters) medoids_out[i] = (int32_t)row_order[(uint32_t)(best[i] & 0xFFFFFFFFull)];
    }
}

}  // namespace fal
FAL_WARM_KERNEL(fal::refine_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

// a10 on device-resident counts: *d_count_in DBSCAN clusters in, *d_count_out refined clusters out
int fal::refine_dev(fal_ctx* ctx, int32_t* labels, int64_t n, const float* mz, const float* rt, double tol, int is_da,
                    double rt_tol, const int64_t* d_count_in, int64_t** d_count_out) {
    hipStream_t st = ctx->stream;
    // sized by n: a DBSCAN cluster of this library may hold ONE member (a core row whose only eps-neighbours are
    // non-core borders that join a lower core), and the staged ABI accepts any cluster count <= n
    const int64_t cmax = n + 1;
    int32_t *counts = nullptr, *rows = nullptr;
    int64_t *seg = nullptr, *base = nullptr;
    unsigned char* slab = nullptr;
    FAL_TRY(ctx->reserve(SLOT_TAIL, sizeof(int32_t) * (size_t)(3 * cmax + 2 * n) + 64, (void**)&counts));
    int32_t* cursor = counts + cmax;
    int32_t* n_sub = cursor + cmax;
    rows = n_sub + cmax;
    int32_t* sub = rows + n;
    FAL_TRY(ctx->reserve(SLOT_TAIL2, sizeof(int64_t) * (size_t)(2 * (cmax + 2)), (void**)&seg));
    base = seg + (cmax + 2);
    const size_t per = 4 * 11 + 8;   // 11 4-byte arrays + 1 double array
    FAL_TRY(ctx->reserve(SLOT_TAIL4, per * (size_t)n + 256, (void**)&slab));
    RefineScratch S;
    S.zmd = reinterpret_cast<double*>(slab);
    float* f = reinterpret_cast<float*>(slab + 8 * (size_t)n);
    S.val = f;            S.smin = f + n;       S.smax = f + 2 * n;
    int32_t* q = reinterpret_cast<int32_t*>(f + 3 * n);
    S.ord = q;            S.sid = q + n;        S.zl = q + 2 * n;     S.zr = q + 3 * n;
    S.t_a = q + 4 * n;    S.t_b = q + 5 * n;    S.stack = q + 6 * n;  S.visit = q + 7 * n;
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16);
    const int cgrid = (int)std::min<int64_t>(cmax, (int64_t)ctx->num_cus * 32);
    FAL_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)(3 * cmax), st));   // counts, cursor, n_sub
    {
        StageScope ts(ctx, ST_TAIL);
        hipLaunchKernelGGL(label_hist_kernel, dim3(grid), dim3(256), 0, st, labels, n, counts);
        FAL_TRY(device_scan_i32(ctx, counts, cmax, seg, SLOT_TAIL3));
        hipLaunchKernelGGL(member_scatter_kernel, dim3(grid), dim3(256), 0, st, labels, n, seg, cursor, rows);
        hipLaunchKernelGGL(refine_kernel, dim3(cgrid), dim3(64), 0, st, rows, seg, d_count_in, mz,
                           rt_tol >= 0.0 ? rt : nullptr, tol, is_da, rt_tol, S, sub, n_sub);
        FAL_TRY(device_scan_i32(ctx, n_sub, cmax, base, SLOT_TAIL3));
        hipLaunchKernelGGL(relabel_kernel, dim3(cgrid), dim3(64), 0, st, rows, seg, d_count_in, sub, base, labels);
    }
    FAL_CHECK_HIP(hipGetLastError());
    *d_count_out = base + cmax;     // total of n_sub
    return FAL_OK;
}

// a11 + a12 on a device-resident cluster count; the number of noise rows is left at *d_noise_out
int fal::finalize_dev(fal_ctx* ctx, const int32_t* labels_sorted, int64_t n, const int64_t* d_count,
                      const int64_t* row_order, const int32_t* nb_idx, const float* nb_dist, int k,
                      int32_t* labels_out, int32_t* medoids_out, int64_t** d_noise_out, const int32_t* extent) {
    hipStream_t st = ctx->stream;
    const int64_t cmax = n + 1;               // fal_finalize accepts any n_clusters <= n (single-member clusters included)
    int32_t *size = nullptr, *noise = nullptr;
    unsigned long long* best = nullptr;
    int64_t* rank = nullptr;
    FAL_TRY(ctx->reserve(SLOT_FIN, sizeof(int32_t) * (size_t)(n + cmax + 2), (void**)&size));
    noise = size + cmax + 1;
    FAL_TRY(ctx->reserve(SLOT_FIN2, sizeof(unsigned long long) * (size_t)(cmax + 1), (void**)&best));
    FAL_TRY(ctx->reserve(SLOT_FIN3, sizeof(int64_t) * (size_t)(n + 1), (void**)&rank));
    FAL_CHECK_HIP(hipMemsetAsync(size, 0, sizeof(int32_t) * (size_t)(cmax + 1), st));
    FAL_CHECK_HIP(hipMemsetAsync(best, 0xFF, sizeof(unsigned long long) * (size_t)(cmax + 1), st));
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16);
    {
        StageScope ts(ctx, ST_TAIL);
        hipLaunchKernelGGL(cluster_size_kernel, dim3(grid), dim3(256), 0, st, labels_sorted, n, size, row_order, noise);
        if (extent)
            hipLaunchKernelGGL(medoid_score_rows_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 32)),
                               dim3(256), 0, st, labels_sorted, n, nb_idx, nb_dist, k, extent, size, best);
        else
            hipLaunchKernelGGL(medoid_score_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 4), (int64_t)ctx->num_cus * 64)), dim3(256), 0, st, labels_sorted, n, nb_idx, nb_dist, k, extent, size, best);
        FAL_TRY(device_scan_i32(ctx, noise, n, rank, SLOT_TAIL3));
        hipLaunchKernelGGL(finalize_kernel, dim3(grid), dim3(256), 0, st, labels_sorted, n, d_count, row_order, rank,
                           best, labels_out, medoids_out);
    }
    FAL_CHECK_HIP(hipGetLastError());
    *d_noise_out = rank + n;
    return FAL_OK;
}

extern "C" {

int fal_refine_clusters(fal_ctx* ctx, int32_t* labels, int64_t n, const float* precursor_mz_sorted,
                        const float* rt_sorted, double tol, int tol_is_da, double rt_tol, int64_t* n_clusters) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n_clusters && n >= 0 && n < (int64_t)INT32_MAX, FAL_EINVAL, "fal_refine_clusters: bad argument");
    const int64_t C = *n_clusters;
    FAL_REQUIRE(C >= 0 && C <= n, FAL_EINVAL, "fal_refine_clusters: *n_clusters must hold the DBSCAN cluster count");
    if (n == 0 || C == 0) {
        *n_clusters = 0;
        return FAL_OK;
    }
    FAL_REQUIRE(labels && precursor_mz_sorted, FAL_EINVAL, "fal_refine_clusters: NULL array");
    int64_t *d_in = nullptr, *d_out = nullptr;
    FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(int64_t) * 4, (void**)&d_in));
    FAL_CHECK_HIP(hipMemcpyAsync(d_in, &C, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    ctx->stage_reset(ST_TAIL);
    FAL_TRY(refine_dev(ctx, labels, n, precursor_mz_sorted, rt_sorted, tol, tol_is_da, rt_tol, d_in, &d_out));
    FAL_CHECK_HIP(hipMemcpyAsync(n_clusters, d_out, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return FAL_OK;
}

int fal_finalize(fal_ctx* ctx, const int32_t* labels_sorted, int64_t n, int64_t n_clusters, const int64_t* row_order,
                 const int32_t* nb_idx, const float* nb_dist, int k, int32_t* labels_out, int32_t* medoids_out,
                 int64_t* n_labels) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && n_clusters >= 0 && n_clusters <= n && k >= 1, FAL_EINVAL, "fal_finalize: bad argument");
    if (n_labels) *n_labels = 0;
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(labels_sorted && row_order && nb_idx && nb_dist && labels_out && medoids_out, FAL_EINVAL,
                "fal_finalize: NULL array");
    int64_t *d_c = nullptr, *d_noise = nullptr;
    FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(int64_t) * 4, (void**)&d_c));
    FAL_CHECK_HIP(hipMemcpyAsync(d_c, &n_clusters, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    FAL_TRY(finalize_dev(ctx, labels_sorted, n, d_c, row_order, nb_idx, nb_dist, k, labels_out, medoids_out, &d_noise));
    if (n_labels) {
        int64_t n_noise = 0;
        FAL_CHECK_HIP(hipMemcpyAsync(&n_noise, d_noise, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        *n_labels = n_clusters + n_noise;
    }
    return FAL_OK;
}

// a9 + a10 + a11 + a12 in one call with every intermediate count on the device: one host
// synchronisation (for the two output counts) instead of three.
static int cluster_graph_impl(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float eps, int method,
                              const float* precursor_mz_sorted, const float* rt_sorted, double tol, int tol_is_da,
                              double rt_tol, const int64_t* row_order, int32_t* labels_sorted_scratch, int32_t* labels_out,
                              int32_t* medoids_out, int64_t* n_clusters, int64_t* n_labels, const int32_t* nb_count = nullptr) {
    FAL_REQUIRE(ctx && n >= 0 && k >= 1 && n < (int64_t)INT32_MAX && n_clusters && n_labels, FAL_EINVAL,
                "fal_cluster_graph: bad argument");
    *n_clusters = *n_labels = 0;
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(nb_idx && nb_dist && precursor_mz_sorted && row_order && labels_sorted_scratch && labels_out && medoids_out,
                FAL_EINVAL, "fal_cluster_graph: NULL array");
    int64_t *d_db = nullptr, *d_cl = nullptr, *d_noise = nullptr;
    const int32_t* extent = nullptr;                      // a9's row extents (SLOT_DB: not touched by a10 / a11)
    ctx->stage_reset(ST_TAIL);
    if (method < 0) FAL_TRY(dbscan_dev(ctx, nb_idx, nb_dist, n, k, eps, labels_sorted_scratch, &d_db, &extent, nb_count));
    else FAL_TRY(linkage_dev(ctx, nb_idx, nb_dist, n, k, eps, method, labels_sorted_scratch, &d_db));
    FAL_TRY(refine_dev(ctx, labels_sorted_scratch, n, precursor_mz_sorted, rt_sorted, tol, tol_is_da, rt_tol, d_db, &d_cl));
    FAL_TRY(finalize_dev(ctx, labels_sorted_scratch, n, d_cl, row_order, nb_idx, nb_dist, k, labels_out, medoids_out,
                         &d_noise, extent));
    int64_t h[2] = {0, 0};
    FAL_CHECK_HIP(hipMemcpyAsync(&h[0], d_cl, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FAL_CHECK_HIP(hipMemcpyAsync(&h[1], d_noise, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *n_clusters = h[0];
    *n_labels = h[0] + h[1];
    return FAL_OK;
}

int fal_cluster_graph(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float eps,
                      const float* precursor_mz_sorted, const float* rt_sorted, double tol, int tol_is_da,
                      double rt_tol, const int64_t* row_order, int32_t* labels_sorted_scratch, int32_t* labels_out,
                      int32_t* medoids_out, int64_t* n_clusters, int64_t* n_labels) {
    fal::CallScope _call(ctx);
    return cluster_graph_impl(ctx, nb_idx, nb_dist, n, k, eps, -1, precursor_mz_sorted, rt_sorted, tol, tol_is_da, rt_tol,
                              row_order, labels_sorted_scratch, labels_out, medoids_out, n_clusters, n_labels);
}

int fal_cluster_graph_counted(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, const int32_t* nb_count, int64_t n, int k,
                              float eps, const float* precursor_mz_sorted, const float* rt_sorted, double tol, int tol_is_da,
                              double rt_tol, const int64_t* row_order, int32_t* labels_sorted_scratch, int32_t* labels_out,
                              int32_t* medoids_out, int64_t* n_clusters, int64_t* n_labels) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(nb_count || n == 0, FAL_EINVAL, "fal_cluster_graph_counted: NULL nb_count");
    return cluster_graph_impl(ctx, nb_idx, nb_dist, n, k, eps, -1, precursor_mz_sorted, rt_sorted, tol, tol_is_da, rt_tol,
                              row_order, labels_sorted_scratch, labels_out, medoids_out, n_clusters, n_labels, nb_count);
}

int fal_cluster_graph_linkage(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float threshold,
                              int method, const float* precursor_mz_sorted, const float* rt_sorted, double tol, int tol_is_da,
                              double rt_tol, const int64_t* row_order, int32_t* labels_sorted_scratch, int32_t* labels_out,
                              int32_t* medoids_out, int64_t* n_clusters, int64_t* n_labels) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(method >= 0 && method <= 2, FAL_EINVAL, "fal_cluster_graph_linkage: method must be 0 (single), 1 (complete) or 2 (average)");
    FAL_REQUIRE(threshold < 1.0f, FAL_EUNSUPPORTED, "fal_cluster_graph_linkage: the threshold must be below 1 (the distance of a missing pair)");
    return cluster_graph_impl(ctx, nb_idx, nb_dist, n, k, threshold, method, precursor_mz_sorted, rt_sorted, tol, tol_is_da, rt_tol,
                              row_order, labels_sorted_scratch, labels_out, medoids_out, n_clusters, n_labels);
}

}  // extern "C"
