// Internal interface of fused.hip: flat-bucket cosine scan with the top-k kept on chip (f16-MFMA prefilter,
// exact fp32 refinement of the precursor window, exact fallback for the rows the chip cannot hold).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"
#include "simtile.h"

namespace fal {

struct FusedArgs {
    const float* X;              // [n, d] float32 rows, precursor-sorted
    const __half* X16;           // [n, d] the same rows rounded to float16 (prefilter only)
    const DenseJob* jobs;        // flat buckets sorted by decreasing size; xtile0 = 128-query tiles of earlier jobs of
    int n_jobs;                  // the same XCD list (jobs j, j + 8, ...); q_row0 == c_row0, nq == nc
    int k;                       // n_neighbors_ann
    const float* pmz;            // [n] precursor m/z by sorted row
    const float* rt;             // [n] or nullptr
    double tol, rt_tol;
    int is_da;
    int keep;                    // n_neighbors
    int32_t* nb_idx;             // [n, keep]
    float* nb_dist;              // [n, keep]
    int32_t* nb_count;           // [n] or nullptr
    int32_t* fb_list;            // (row, job) pairs of the queries left to the exact fallback
    int32_t* fb_count;
    int fb_cap;
    unsigned long long* stamps;  // FALCON_FUSED_DBG bit 128: per-workgroup phase time stamps [grid][10] (s_memtime)
    int dbg;                     // FALCON_FUSED_DBG: phase-skipping bits for timing experiments (results invalid when set)
};

bool fused_supports(int d);
// list_tiles = 128-query tiles of the longest XCD list; max_nc = largest bucket
int launch_fused(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles, int max_nc);

}  // namespace fal
