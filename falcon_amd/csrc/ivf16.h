// Internal interface of ivf16.hip: IVF fine scan with a float16 prefilter (list-major f16-MFMA scan to 16-bit keys, k-th key
// per query, exact tail in fused.hip).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"
#include "simtile.h"
#include "fused.h"

namespace fal {

struct List16Args {              // cf. ListScanArgs (scan.h); tiles = groups of four 32-row list slices
    const __half* Xl16;          // float16 rows in (bucket, list, row) order
    int d;
    const int64_t* list_off;
    const int64_t* inv_off;
    const int64_t* ltile_off;
    const int32_t* inv_q;
    const int64_t* inv_dest;     // element index in `keys` where that query's segment for this list starts
    int64_t list_begin, list_end;
    int64_t tile_begin, n_tiles_max;
    uint16_t* keys;              // round(approximate similarity * 65535)
    int64_t keys_base;
    uint16_t* sink;              // >= 64 entries of scratch for masked stores
    int64_t n_rows;              // rows of Xl16 (row ids read past a list's end are clamped)
};

struct Select16Args {
    const uint16_t* keys;
    int64_t keys_base;
    int k;
    const DenseJob* jobs;        // IVF tile table (q_row0 = first list-order position of the bucket, c_row0 = its list 0)
    int n_jobs;
    int64_t tile_begin;
    const int32_t* tile_job;     // (filled by launch_select16)
    const int64_t* q_sim_off;
    const int32_t* perm;
    QThr* thr;                   // hand-off (fused.h), by sorted row
    float* gmem_v;               // members: approximate value ...
    uint32_t* gmem_id;           // ... and position in the query's key stream (resolve_kernel: -> row where needed)
    int64_t n_tiles;             // (filled by launch_select16)
    int64_t max_keys;            // the most keys any query of the search has (picks the kernel form)
    int32_t* big_count;          // queries with more than 2,048 keys: second pass with 64 keys per lane
    int32_t* big_list;
    int big_cap;
};

bool ivf16_supports(int d);
int launch_gather16(fal_ctx* ctx, const void* X16, const int32_t* perm, int64_t n, int d, void* out, int32_t* pos_of_row);
int launch_list16(fal_ctx* ctx, const List16Args& a);
int launch_select16(fal_ctx* ctx, const Select16Args& a, int64_t n_tiles);

}  // namespace fal
