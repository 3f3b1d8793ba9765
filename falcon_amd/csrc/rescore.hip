// f4: exact re-scoring of the ANN neighbour lists with the matched-peak cosine the reference snapshot
// ships (similarity.py:17-80 `cosine_fast`; its use cluster.py:593-639: dist = 1 - sim, sim = 0 when
// fewer than min_matches peaks match).
//
// One lane per stored (query, neighbour) pair.  The reference fills a dense cost matrix and calls
// scipy's linear_sum_assignment on it; here the same optimum is reached without the matrix: both peak
// lists are sorted, so the pairs inside the fragment tolerance form runs ("components": consecutive
// query peaks whose windows chain through shared neighbour peaks) and the assignment decomposes into
// one small problem per component -- almost always 1x1, solved in place; the general case runs the
// Hungarian algorithm on the component (<= kMaxComp peaks a side, per-lane scratch).  Window arithmetic
// is the reference's: `peak_mz - tol` in float64, `abs(peak_mz - other_mz)` in float32 against the
// float64 tolerance; pair costs are float32 products; the positive pair scores are summed in query-peak
// order in float64.  Latency/L2-bound gather work (peaks of a bucket's spectra are shared by its rows).
#include <math.h>
#include <algorithm>
#include "common.h"
#include "ivf.h"

namespace fal {

constexpr int kMaxComp = 32;

struct PeakLists {
    const float* amz;
    const float* ait;
    const float* bmz;
    const float* bit;
};

// weight of (component row r, column q): the float32 product inside the row's window, else 0
__device__ __forceinline__ float comp_w(const PeakLists& s, const int* rp, const int* rs, const int* re, int r, int q) {
    return (q >= rs[r] && q < re[r]) ? s.ait[rp[r]] * s.bit[q] : 0.f;
}

// maximum-weight assignment of one component (rows = query peaks rp[0..nr), columns [qs, qe)); adds the
// positive pair scores in row order.  Returns false when the component is larger than kMaxComp.
__device__ bool solve_component(const PeakLists& s, const int* rp, const int* rs, const int* re, int nr, int qs, int qe,
                                double* score, int* n_match) {
    const int nc = qe - qs;
    if (nr == 1) {                                    // one query peak: its best partner
        float best = 0.f;
        for (int q = rs[0]; q < re[0]; ++q) best = fmaxf(best, s.ait[rp[0]] * s.bit[q]);
        if (best > 0.f) {
            *score += (double)best;
            *n_match += 1;
        }
        return true;
    }
    if (nr > kMaxComp || nc > kMaxComp) return false;
    // Hungarian algorithm (potentials, O(n^2 m)), minimising -w; n = the smaller side
    const bool tr = nr > nc;                          // transposed: "rows" of the solver are the columns
    const int n = tr ? nc : nr, m = tr ? nr : nc;
    double u[kMaxComp + 1], v[kMaxComp + 1], minv[kMaxComp + 1];
    int p[kMaxComp + 1], way[kMaxComp + 1];
    bool used[kMaxComp + 1];
    for (int j = 0; j <= m; ++j) {
        v[j] = 0.0;
        p[j] = 0;
    }
    for (int i = 0; i <= n; ++i) u[i] = 0.0;
    auto cost = [&](int i, int j) -> double {         // 1-based solver indices
        const int r = tr ? j - 1 : i - 1, q = qs + (tr ? i - 1 : j - 1);
        return -(double)comp_w(s, rp, rs, re, r, q);
    };
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        for (int j = 0; j <= m; ++j) {
            minv[j] = INFINITY;
            used[j] = false;
        }
        do {
            used[j0] = true;
            const int i0 = p[j0];
            double delta = INFINITY;
            int j1 = 0;
            for (int j = 1; j <= m; ++j) {
                if (!used[j]) {
                    const double cur = cost(i0, j) - u[i0] - v[j];
                    if (cur < minv[j]) {
                        minv[j] = cur;
                        way[j] = j0;
                    }
                    if (minv[j] < delta) {
                        delta = minv[j];
                        j1 = j;
                    }
                }
            }
            for (int j = 0; j <= m; ++j) {
                if (used[j]) {
                    u[p[j]] += delta;
                    v[j] -= delta;
                } else {
                    minv[j] -= delta;
                }
            }
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    // p[j] = solver row assigned to solver column j.  Sum in query-peak (component row) order.
    if (!tr) {
        int col_of[kMaxComp];
        for (int r = 0; r < nr; ++r) col_of[r] = -1;
        for (int j = 1; j <= m; ++j)
            if (p[j] != 0) col_of[p[j] - 1] = qs + j - 1;
        for (int r = 0; r < nr; ++r) {
            const float w = col_of[r] >= 0 ? comp_w(s, rp, rs, re, r, col_of[r]) : 0.f;
            if (w > 0.f) {
                *score += (double)w;
                *n_match += 1;
            }
        }
    } else {
        for (int j = 1; j <= m; ++j) {                // solver column j = component row j - 1
            const float w = p[j] != 0 ? comp_w(s, rp, rs, re, j - 1, qs + p[j] - 1) : 0.f;
            if (w > 0.f) {
                *score += (double)w;
                *n_match += 1;
            }
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void rescore_kernel(const int32_t* __restrict__ nb_idx, float* __restrict__ nb_dist, int64_t n,
                                                      int k, const float* __restrict__ mz, const float* __restrict__ intensity,
                                                      const int64_t* __restrict__ indptr, const int64_t* __restrict__ order,
                                                      double tol, int min_matches, int32_t* __restrict__ err) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n * k; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = nb_idx[t];
        if (j < 0) continue;
        const int64_t a = order[t / k], b = order[j];
        const int64_t a0 = indptr[a], b0 = indptr[b];
        const int na = (int)(indptr[a + 1] - a0), nb = (int)(indptr[b + 1] - b0);
        PeakLists s{mz + a0, intensity + a0, mz + b0, intensity + b0};
        double score = 0.0;
        int n_match = 0;
        bool ok = true;
        if (na > 0 && nb > 0) {
            int rp[kMaxComp], rs[kMaxComp], re[kMaxComp];
            int nr = 0, qs = 0, qe = 0, o = 0;
            for (int p = 0; p < na; ++p) {                                   // similarity.py:45-63
                const float pm = s.amz[p];
                while (o < nb - 1 && (double)pm - tol > (double)s.bmz[o]) ++o;
                int q = o;
                while (q < nb && (double)fabsf(pm - s.bmz[q]) <= tol) ++q;
                if (q == o) continue;                                        // nothing inside this peak's window
                if (nr > 0 && o >= qe) {                                     // window starts past the component: close it
                    ok = solve_component(s, rp, rs, re, nr, qs, qe, &score, &n_match) && ok;
                    nr = 0;
                }
                if (nr == 0) qs = o;
                if (nr < kMaxComp) {
                    rp[nr] = p;
                    rs[nr] = o;
                    re[nr] = q;
                }
                ++nr;
                qe = nr == 1 ? q : max(qe, q);
            }
            if (nr > 0) ok = solve_component(s, rp, rs, re, nr, qs, qe, &score, &n_match) && ok;
        }
        if (!ok) atomicExch(err, 1);
        double sim = fmax(0.0, fmin(score, 1.0));                            // similarity.py:78
        if (n_match < min_matches) sim = 0.0;                                // cluster.py:624-626
        nb_dist[t] = (float)(1.0 - sim);
    }
}

}  // namespace fal
FAL_WARM_KERNEL(fal::rescore_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

extern "C" int fal_rescore_neighbors(fal_ctx* ctx, const int32_t* nb_idx, float* nb_dist, int64_t n, int k, const float* mz,
                                     const float* intensity, const int64_t* indptr, const int64_t* row_order,
                                     double fragment_tol, int min_matches) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && k >= 1 && fragment_tol >= 0.0, FAL_EINVAL, "fal_rescore_neighbors: bad argument");
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(nb_idx && nb_dist && indptr && row_order, FAL_EINVAL, "fal_rescore_neighbors: NULL array");
    int32_t* err = nullptr;
    FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(int32_t), (void**)&err));
    FAL_CHECK_HIP(hipMemsetAsync(err, 0, sizeof(int32_t), ctx->stream));
    ctx->stage_reset(ST_FILTER);
    {
        StageScope ts(ctx, ST_FILTER);
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n * k, 256), (int64_t)ctx->num_cus * 32);
        hipLaunchKernelGGL(rescore_kernel, dim3(grid), dim3(256), 0, ctx->stream, nb_idx, nb_dist, n, k, mz, intensity, indptr,
                           row_order, fragment_tol, min_matches, err);
        FAL_CHECK_HIP(hipGetLastError());
    }
    int32_t h = 0;
    FAL_CHECK_HIP(hipMemcpyAsync(&h, err, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    FAL_REQUIRE(h == 0, FAL_EUNSUPPORTED,
                "fal_rescore_neighbors: more than %d peaks of one spectrum chain inside the fragment tolerance", kMaxComp);
    return FAL_OK;
}
