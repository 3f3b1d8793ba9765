// a8 neighbour filter + distance, a9 DBSCAN(min_samples = 2) on the sparse neighbour graph.
//
// Reference: tolerance arithmetic as cluster.py:190-195 uses spectrum_utils.mass_diff;
// distance = 1 - sim (cluster.py:626) clamped to [0, 1] (similarity.py:78); min_samples = 2
// (cluster.py:66); DBSCAN itself is README.md:143-146.  HBM-bound integer/graph work: one pass
// over the [n, k_ann] search result, two passes over the [n, k] neighbour lists.
#include <math.h>
#include "common.h"
#include "ivf.h"
#include "util.h"

namespace fal {

// ------------------------------------------------------------------------------------------
// a8: one wave per row; the row is already sorted by descending similarity
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_kernel(const float* __restrict__ sim, const int32_t* __restrict__ idx,
                                                     int64_t n, int k_ann, const float* __restrict__ pmz,
                                                     const float* __restrict__ rt, double tol, int is_da,
                                                     double rt_tol, int k, int32_t* __restrict__ nb_idx,
                                                     float* __restrict__ nb_dist) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (row >= n) return;
    const float qmz = pmz[row];
    const float qrt = rt ? rt[row] : 0.f;
    int kept = 0;
    for (int c0 = 0; c0 < k_ann && kept < k; c0 += 64) {
        const int c = c0 + lane;
        bool ok = false;
        int32_t j = -1;
        float s = 0.f;
        if (c < k_ann) {
            j = idx[row * k_ann + c];
            s = sim[row * k_ann + c];
            if (j >= 0 && (int64_t)j != row) {
                const float nmz = pmz[j];
                const float diff = qmz - nmz;     // mass_diff(query, neighbour)
                const double md = is_da ? (double)diff : (double)(diff / nmz) * 1e6;
                ok = fabs(md) <= tol;
                if (ok && rt && rt_tol >= 0.0) ok = fabs((double)(qrt - rt[j])) <= rt_tol;
            }
        }
        const uint64_t mask = __ballot(ok);
        const int pos = kept + __popcll(mask & ((1ull << lane) - 1ull));
        if (ok && pos < k) {
            nb_idx[row * k + pos] = j;
            nb_dist[row * k + pos] = fminf(fmaxf(1.0f - s, 0.f), 1.f);
        }
        kept += __popcll(mask);
    }
    kept = min(kept, k);
    for (int c = kept + lane; c < k; c += 64) {
        nb_idx[row * k + c] = -1;
        nb_dist[row * k + c] = INFINITY;
    }
}

// ------------------------------------------------------------------------------------------
// a9
// ------------------------------------------------------------------------------------------
// (ids outside [0, n) are treated as "no neighbour": an uninitialised neighbour array must not send the union-find walking
//  through foreign memory -- it hung a GPU box for its whole time limit once)
constexpr int kEdgeThreads = 8;      // threads per row of the counted core pass and the edges pass
__device__ __forceinline__ bool edge_ok(int32_t j, float dist, int64_t i, float eps, int64_t n) {
    return j >= 0 && (int64_t)j < n && (int64_t)j != i && dist <= eps;
}

// core(i) <=> row i stores a neighbour within eps (the point itself is the other sample).
// One wave per row: 64 consecutive slots per load (coalesced), any() by ballot.
// Also leaves extent[i] = one past the row's last stored neighbour: rows come front-packed from a8 with a handful of
// neighbours in n_neighbors slots, and the later passes over the graph (edges here, the medoid scores in tail.hip) read
// only that far instead of 8 n_neighbors bytes of every row.
__global__ __launch_bounds__(256) void dbscan_core_kernel(const int32_t* __restrict__ nb_idx,
                                                          const float* __restrict__ nb_dist, int64_t n, int k,
                                                          float eps, int32_t* __restrict__ core,
                                                          int32_t* __restrict__ parent, int32_t* __restrict__ border_src,
                                                          int32_t* __restrict__ extent) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        bool c = false;
        int ext = 0;
        for (int s0 = 0; s0 < k; s0 += 64) {
            const int s = s0 + lane;
            const int32_t j = s < k ? nb_idx[i * k + s] : -1;
            const bool ok = s < k && edge_ok(j, nb_dist[i * k + s], i, eps, n);
            c = c || __ballot(ok) != 0;
            const uint64_t stored = __ballot(j >= 0);
            if (stored) ext = s0 + 64 - __clzll((unsigned long long)stored);
        }
        if (lane == 0) {
            core[i] = c;
            parent[i] = (int32_t)i;
            border_src[i] = 0x7fffffff;
            extent[i] = ext;
        }
    }
}

__device__ __forceinline__ int32_t uf_find(int32_t* parent, int32_t x) {
    // path halving; parent links only ever point to smaller ids
    int32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const int32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = g;
    }
    return x;
}

// The find of the FINAL pass (every row -> its root, after all unions): no path halving.  With halving, a thread that has
// loaded parent[i] and parent[parent[i]] may store that grandparent into parent[i] AFTER the row's own thread has stored
// the root there -- parent[i] then names an ancestor that is not the root and the row gets the label slot of a non-root
// (found by a 3,000-row chain in tests/test_gpu_linkage.py; shallow trees make it rare, not impossible).  Read-only finds
// race with nothing: whatever a concurrent reader sees in parent[i] (the old link or the root) is an ancestor.
__device__ __forceinline__ int32_t uf_find_final(const int32_t* parent, int32_t x) {
    int32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

// hook the larger root under the smaller one (so every component's root is its lowest core row)
__device__ __forceinline__ void uf_union(int32_t* parent, int32_t a, int32_t b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int32_t t = a; a = b; b = t; }     // a > b
        const int32_t old = atomicCAS(&parent[a], a, b);
        if (old == a) return;
    }
}

// dbscan_core_kernel for front-packed rows of known length (`count` = a8's nb_count): eight threads per row look at the
// stored neighbours only, the row's verdict is the OR over their lanes
__global__ __launch_bounds__(256) void dbscan_core_counted_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist,
                                                                  const int32_t* __restrict__ count, int64_t n, int k, float eps,
                                                                  int32_t* __restrict__ core, int32_t* __restrict__ parent,
                                                                  int32_t* __restrict__ border_src) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const int64_t total = (n + 7) / 8 * 64;                  // whole waves: every lane takes part in the ballot
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e >> 3;
        bool c = false;
        if (i < n) {
            const int ext = min(count[i], k);
            for (int s = sub; s < ext && !c; s += 8) c = edge_ok(nb_idx[i * k + s], nb_dist[i * k + s], i, eps, n);
        }
        const uint64_t any = __ballot(c);
        if (sub == 0 && i < n) {
            core[i] = ((any >> (lane & ~7)) & 0xffull) != 0;
            parent[i] = (int32_t)i;
            border_src[i] = 0x7fffffff;
        }
    }
}

// eight threads per row, one per stored edge (slots s, s + 8, ... below the row's extent): core->core edges are united,
// core->border edges vote for the border point's lowest-index core in-neighbour.  (One thread per SLOT read 8 n_neighbors
// bytes of every core row: 0.39 ms per 1 M spectra; one wave per row left most lanes idle behind four dependent loads.)
__global__ __launch_bounds__(256) void dbscan_edges_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist,
                                                           int64_t n, int k, float eps, const int32_t* __restrict__ core,
                                                           const int32_t* __restrict__ extent, int32_t* __restrict__ parent,
                                                           int32_t* __restrict__ border_src) {
    const int64_t total = n * kEdgeThreads;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / kEdgeThreads;
        const int ext = core[i] ? min(extent[i], k) : 0;        // (a caller's nb_count may exceed k: clamp like the core / medoid passes)
        for (int s = (int)(e % kEdgeThreads); s < ext; s += kEdgeThreads) {
            const int32_t j = nb_idx[i * k + s];
            if (!edge_ok(j, nb_dist[i * k + s], i, eps, n)) continue;
            if (core[j]) uf_union(parent, (int32_t)i, j);
            else atomicMin(&border_src[j], (int32_t)i);
        }
    }
}

__global__ void dbscan_roots_kernel(const int32_t* __restrict__ core, int32_t* __restrict__ parent, int64_t n,
                                    int32_t* __restrict__ is_root) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t r = 0;
        if (core[i]) {
            const int32_t root = uf_find_final(parent, (int32_t)i);
            __hip_atomic_store(&parent[i], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r = root == (int32_t)i;
        }
        is_root[i] = r;
    }
}

__global__ void dbscan_label_kernel(const int32_t* __restrict__ core, const int32_t* __restrict__ parent,
                                    const int32_t* __restrict__ border_src, const int64_t* __restrict__ root_rank,
                                    int64_t n, int32_t* __restrict__ labels) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int32_t lab = -1;
        if (core[i]) lab = (int32_t)root_rank[parent[i]];
        else if (border_src[i] != 0x7fffffff) lab = (int32_t)root_rank[parent[border_src[i]]];
        labels[i] = lab;
    }
}

// ------------------------------------------------------------------------------------------
// ELL -> CSR of the neighbour lists (the payload of the multi-GPU exchange):
// one wave per row, entries keep their order, ids are shifted to global rows
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nb_count_kernel(const int32_t* __restrict__ nb_idx, int64_t n, int k,
                                                       int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
        int c = 0;
        for (int j = lane; j < k + lane; j += 64) {           // wave-uniform trip count
            const bool valid = j < k && nb_idx[row * k + j] >= 0;
            c += __popcll(__ballot(valid));
        }
        if (lane == 0) count[row] = c;
    }
}

// rows known to be front-packed with indptr[row + 1] - indptr[row] entries: only that prefix is read
__global__ __launch_bounds__(256) void nb_pack_prefix_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist,
                                                             int64_t n, int k, int64_t id_offset,
                                                             const int64_t* __restrict__ id_map,
                                                             const int64_t* __restrict__ indptr, int32_t* __restrict__ out_idx,
                                                             float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
        const int64_t base = indptr[row];
        const int cnt = (int)(indptr[row + 1] - base);
        for (int j = lane; j < cnt; j += 64) {
            const int32_t id = nb_idx[row * k + j];
            out_idx[base + j] = (int32_t)((id_map ? id_map[id] : (int64_t)id) + id_offset);
            out_dist[base + j] = nb_dist[row * k + j];
        }
    }
}

__global__ void nb_chain_kernel(const int64_t* __restrict__ local, int64_t n, int64_t* __restrict__ indptr) {
    const int64_t base = indptr[0];                           // written by the previous segment (0 for the first)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        indptr[i + 1] = base + local[i + 1];
}

__global__ __launch_bounds__(256) void nb_pack_kernel(const int32_t* __restrict__ nb_idx, const float* __restrict__ nb_dist,
                                                      int64_t n, int k, int64_t id_offset,
                                                      const int64_t* __restrict__ id_map,
                                                      const int64_t* __restrict__ indptr, int32_t* __restrict__ out_idx,
                                                      float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
        int64_t base = indptr[row];
        if (indptr[row + 1] == base) continue;                // wave-uniform
        for (int j = lane; j < k + lane; j += 64) {
            const int32_t id = j < k ? nb_idx[row * k + j] : -1;
            const bool valid = id >= 0;
            const unsigned long long m = __ballot(valid);
            if (valid) {
                const int64_t o = base + __popcll(m & ((1ull << lane) - 1ull));
                out_idx[o] = (int32_t)((id_map ? id_map[id] : (int64_t)id) + id_offset);
                out_dist[o] = nb_dist[row * k + j];
            }
            base += __popcll(m);
        }
    }
}

}  // namespace fal
FAL_WARM_KERNEL(fal::filter_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

extern "C" {

int fal_filter_neighbors(fal_ctx* ctx, const float* sim, const int32_t* idx, int64_t n, int k_ann,
                         const float* precursor_mz_sorted, const float* rt_sorted, double tol, int tol_is_da,
                         double rt_tol, int n_neighbors, int32_t* nb_idx, float* nb_dist) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && k_ann >= 1 && n_neighbors >= 1, FAL_EINVAL, "fal_filter_neighbors: bad argument");
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(sim && idx && precursor_mz_sorted && nb_idx && nb_dist, FAL_EINVAL, "fal_filter_neighbors: NULL array");
    ctx->stage_reset(ST_FILTER);
    StageScope ts(ctx, ST_FILTER);
    hipLaunchKernelGGL(filter_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, ctx->stream, sim, idx, n, k_ann,
                       precursor_mz_sorted, rt_tol >= 0.0 ? rt_sorted : nullptr, tol, tol_is_da, rt_tol, n_neighbors,
                       nb_idx, nb_dist);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int fal_neighbors_to_csr(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, const int32_t* nb_count, int64_t n, int k,
                         int64_t id_offset, int64_t row0, int64_t* indptr_out, int32_t* idx_out, float* dist_out) {
    fal::CallScope _call(ctx);
    return fal_neighbors_to_csr_mapped(ctx, nb_idx, nb_dist, nb_count, n, k, nullptr, id_offset, row0, indptr_out, idx_out,
                                       dist_out);
}

int fal_neighbors_to_csr_mapped(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, const int32_t* nb_count, int64_t n,
                                int k, const int64_t* id_map, int64_t id_offset, int64_t row0, int64_t* indptr_out,
                                int32_t* idx_out, float* dist_out) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && k >= 1 && row0 >= 0, FAL_EINVAL, "fal_neighbors_to_csr: bad argument");
    FAL_REQUIRE(indptr_out, FAL_EINVAL, "fal_neighbors_to_csr: NULL indptr");
    if (row0 == 0) FAL_CHECK_HIP(hipMemsetAsync(indptr_out, 0, sizeof(int64_t), ctx->stream));
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(nb_idx && nb_dist && idx_out && dist_out, FAL_EINVAL, "fal_neighbors_to_csr: NULL array");
    int32_t* count = nullptr;
    int64_t* local = nullptr;
    FAL_TRY(ctx->reserve(SLOT_DB, sizeof(int32_t) * (size_t)n, (void**)&count));
    FAL_TRY(ctx->reserve(SLOT_DB2, sizeof(int64_t) * (size_t)(n + 1), (void**)&local));
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 4), (int64_t)ctx->num_cus * 64);
    if (nb_count) {
        count = const_cast<int32_t*>(nb_count);          // rows are front-packed and their lengths known (fused a7+a8)
    } else {
        hipLaunchKernelGGL(nb_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, nb_idx, n, k, count);
    }
    FAL_TRY(device_scan_i32(ctx, count, n, local, SLOT_DB3));
    // indptr_out[row0 + 1 + i] = indptr_out[row0] + local[i + 1]: segments chain on the device
    hipLaunchKernelGGL(nb_chain_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, ctx->stream,
                       local, n, indptr_out + row0);
    if (nb_count)
        hipLaunchKernelGGL(nb_pack_prefix_kernel, dim3(grid), dim3(256), 0, ctx->stream, nb_idx, nb_dist, n, k, id_offset,
                           id_map, indptr_out + row0, idx_out, dist_out);
    else
        hipLaunchKernelGGL(nb_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, nb_idx, nb_dist, n, k, id_offset,
                           id_map, indptr_out + row0, idx_out, dist_out);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // extern "C"

// a9 with the cluster count left on the device at *d_count_out (no host synchronisation)
int fal::dbscan_dev(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float eps,
                    int32_t* labels, int64_t** d_count_out, const int32_t** extent_out, const int32_t* nb_count) {
    int32_t* buf = nullptr;
    int64_t* rank = nullptr;
    FAL_TRY(ctx->reserve(SLOT_DB, sizeof(int32_t) * (size_t)n * 5, (void**)&buf));
    FAL_TRY(ctx->reserve(SLOT_DB2, sizeof(int64_t) * (size_t)(n + 1), (void**)&rank));
    int32_t *core = buf, *parent = buf + n, *border = buf + 2 * n, *is_root = buf + 3 * n, *extent = buf + 4 * n;
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16);
    const unsigned wgrid = (unsigned)std::min<int64_t>(ceil_div(n, 4), (int64_t)ctx->num_cus * 64);      // a wave per row
    ctx->stage_reset(ST_DBSCAN);
    {
        StageScope ts(ctx, ST_DBSCAN);
        if (nb_count) {
            // front-packed rows of known length: nothing beyond nb_count[i] is read, here or by the later passes
            const unsigned cgrid = (unsigned)std::min<int64_t>(ceil_div(n * kEdgeThreads, 256), (int64_t)ctx->num_cus * 32);
            hipLaunchKernelGGL(dbscan_core_counted_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, nb_idx, nb_dist, nb_count, n, k,
                               eps, core, parent, border);
            extent = const_cast<int32_t*>(nb_count);
        } else {
            hipLaunchKernelGGL(dbscan_core_kernel, dim3(wgrid), dim3(256), 0, ctx->stream, nb_idx, nb_dist, n, k, eps, core, parent,
                               border, extent);
        }
        const unsigned egrid = (unsigned)std::min<int64_t>(ceil_div(n * kEdgeThreads, 256), (int64_t)ctx->num_cus * 32);
        hipLaunchKernelGGL(dbscan_edges_kernel, dim3(egrid), dim3(256), 0, ctx->stream, nb_idx, nb_dist, n, k, eps, core, extent,
                           parent, border);
        hipLaunchKernelGGL(dbscan_roots_kernel, dim3(grid), dim3(256), 0, ctx->stream, core, parent, n, is_root);
        FAL_TRY(device_scan_i32(ctx, is_root, n, rank, SLOT_DB3));
        hipLaunchKernelGGL(dbscan_label_kernel, dim3(grid), dim3(256), 0, ctx->stream, core, parent, border, rank, n, labels);
    }
    FAL_CHECK_HIP(hipGetLastError());
    *d_count_out = rank + n;
    if (extent_out) *extent_out = extent;
    return FAL_OK;
}

extern "C" int fal_dbscan(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float eps,
                          int32_t* labels, int64_t* n_clusters) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && k >= 1 && n < (int64_t)INT32_MAX, FAL_EINVAL, "fal_dbscan: bad argument");
    if (n_clusters) *n_clusters = 0;
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(nb_idx && nb_dist && labels, FAL_EINVAL, "fal_dbscan: NULL array");
    int64_t* d_count = nullptr;
    FAL_TRY(fal::dbscan_dev(ctx, nb_idx, nb_dist, n, k, eps, labels, &d_count));
    if (n_clusters) {
        FAL_CHECK_HIP(hipMemcpyAsync(n_clusters, d_count, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return FAL_OK;
}
