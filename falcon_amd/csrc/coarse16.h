// Internal interface of coarse16.hip: the coarse quantiser of the search from the 16-bit keys of the final k-means pass.
#pragma once
#include "common.h"
#include "simtile.h"

namespace fal {

struct Coarse16Args {
    const uint16_t* ckeys;       // [n, stride] by sorted row
    int stride;                  // 128 x groups (<= 2,048)
    const float* X;              // [n, d] float32 rows, sorted order (exact re-evaluation of close calls)
    const float* C;              // [total_lists, d] centroids
    int d;
    const DenseJob* jobs;        // IVF tile table (q_row0 = first list-order position of the bucket, c_row0 = its list 0, nc = n_list)
    int n_jobs;
    int64_t n_tiles;
    const int32_t* tile_job;     // (filled by launch_coarse16)
    const int32_t* perm;         // [n] list-order position -> sorted row
    int np;                      // probes per query
    int32_t* probes;             // [n, np] by list-order position: the chosen bucket-local lists, -1 behind them
    int32_t* ovf_count;          // (launch_coarse16) queries with more members than the wave-level kernel holds
    int32_t* ovf_list;
    int ovf_cap;
    const uint16_t* sp_cols;     // [n, 64] the rows' sparse form (ivf.h): the exact chains of the close calls walk the query's
    const float* sp_vals;        // <= 64 entries against the dense centroid instead of low_dim terms (nullptr: dense chains)
};

int launch_coarse16(fal_ctx* ctx, const Coarse16Args& a);

}  // namespace fal
