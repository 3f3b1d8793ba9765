// Internal interface of fused.hip: flat-bucket cosine scan with the top-k kept on chip (f16-MFMA prefilter,
// exact fp32 refinement of the precursor window, exact resolution of the k-th key, exact fallback for the rows the
// chip cannot hold).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"
#include "simtile.h"

#define FAL_FUSED_MEM 40      /* members (candidates inside the threshold bin) kept per query: 20 per lane half */
#define FAL_FUSED_KEEP 64     /* selected window candidates kept per query: 32 per lane half */

namespace fal {

struct QThr {                    // per query, approx_kernel -> band_kernel / resolve_kernel
    float L, U;                  // the exact k-th best similarity lies in [L, U]
    float T, eps;                // k-th best approximate similarity, bound of |approx - exact| around it
    int bstar, nabove;           // its histogram bin, candidates in higher bins
    int mc;                      // members of half 0 | half 1 << 16
    int flags;                   // bit 1: exact fallback
};

struct FusedArgs {
    const float* X;              // [n, d] float32 rows, precursor-sorted
    const __half* X16;           // [n, d] the same rows rounded to float16 (prefilter only)
    // flat buckets sorted by decreasing size, q_row0 == c_row0, nq == nc; xtile0 = tiles of earlier jobs of the same XCD list
    const DenseJob* jobs128;     // buckets with more than k rows, 128-query tiles (approx_kernel)
    int n_jobs128;
    const DenseJob* jobs32;      // every flat bucket, 32-query tiles (band_kernel, resolve_kernel, fallback)
    int n_jobs32;
    const int32_t* tile_job32;   // optional [grid]: job of workgroup `bid` of a 32-query-tile launch (-1: none) -- find_job_xcd's
                                 // binary search (seven dependent loads at 600 buckets) done once per search (launch_tile_job32)
    int k;                       // n_neighbors_ann
    const float* pmz;            // [n] precursor m/z by sorted row
    const float* rt;             // [n] or nullptr
    double tol, rt_tol;
    float tol_f, rt_f;           // float32 forms of the tolerance tests (launch_fused)
    int is_da;
    int keep;                    // n_neighbors
    int32_t* nb_idx;             // [n, keep]
    float* nb_dist;              // [n, keep]
    int32_t* nb_count;           // [n] or nullptr
    // hand-off between the kernels, indexed by sorted row (launch_fused)
    QThr* thr;
    float* gmem_v;               // [n, FAL_FUSED_MEM] approximate values of the members (half 0 | half 1)
    uint32_t* gmem_id;           // [n, FAL_FUSED_MEM] bucket-local candidate
    uint32_t* gkept_u;           // [n, FAL_FUSED_KEEP] sortable exact similarity of the kept window candidates
    uint32_t* gkept_id;          // [n, FAL_FUSED_KEEP] sorted row | 0x80000000 (ambiguous)
    int32_t* gkcnt;              // [n, 2] kept per half | 0x100 ambiguous | 0x200 overflow
    int32_t* fb_list;            // (row, job) pairs of the queries left to the exact fallback
    int32_t* fb_count;
    int fb_cap;
    // IVF buckets (ivf16.hip): the candidates of a query are the rows of its probed lists; thr / gmem_* come from
    // select16_kernel instead of approx_kernel
    int ivf;                     // 1: jobs32 are IVF buckets (c_row0 = global id of the bucket's list 0); 2: the same, and the kept
                                 // candidates + their exact similarities come from pairs16.hip instead of band_kernel
    const int32_t* assign;       // [n] bucket-local list of every sorted row
    const int32_t* pos_of_row;   // [n] sorted row -> list-order position
    const int32_t* probes;       // [n, n_probe] by list-order position (-1 = none)
    int n_probe;
    int mask_words;              // 32-bit words of one query's probe mask (>= max n_list / 32)
    const int64_t* list_off;     // [total_lists + 1]
    const int32_t* perm;         // [n] list-order position -> sorted row
    const uint16_t* sp_cols;     // [n, 64] the rows' sparse form (ivf.h), or nullptr: pairs16 then reads the dense rows
    const float* sp_vals;
    int rows_f16;                // 1: every row component is a float16 value (pairs16s keeps its query tile as float16 in LDS)
};

bool fused_supports(int d);
// n_rows = rows of the index (the hand-off buffers are indexed by sorted row); list_tiles* = tiles of the longest XCD list
int launch_fused(fal_ctx* ctx, const FusedArgs& a, int d, int64_t n_rows, int64_t list_tiles128, int64_t list_tiles32,
                 int max_nc);
// IVF buckets: hand-off buffers and the fallback list (before select16_kernel runs), then the exact tail --
// band_kernel over the precursor windows (candidates outside the query's probed lists masked out), resolve_kernel, and the
// exact fallback over the probed lists.  max_cand = upper bound of the candidates of one query.
// the job and local tile of workgroup `bid` of a launch over the 32-query tiles (XCD lists: simtile.h find_job_xcd)
__device__ __forceinline__ bool find_job32(const FusedArgs& a, unsigned bid, int* job_index, int* local_tile) {
    if (a.tile_job32 == nullptr) return find_job_xcd(a.jobs32, a.n_jobs32, bid, job_index, local_tile);
    const int j = a.tile_job32[bid];
    if (j < 0) return false;
    *job_index = j;
    *local_tile = (int)((int64_t)(bid >> 3) - a.jobs32[j].xtile0);
    return true;
}
// table[bid] for bid < 8 * list_tiles32 (FusedArgs::tile_job32)
int launch_tile_job32(fal_ctx* ctx, const DenseJob* jobs32, int n_jobs32, int64_t list_tiles32, int32_t* table);
int fused_prepare(fal_ctx* ctx, FusedArgs* a, int64_t n_rows);
int launch_fused_ivf_tail(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles32, int64_t max_cand);

}  // namespace fal
