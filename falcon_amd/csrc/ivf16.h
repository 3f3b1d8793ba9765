// Internal interface of ivf16.hip: IVF fine scan with a float16 prefilter (list-major f16-MFMA scan to 16-bit keys, k-th key
// per query, exact tail in fused.hip).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"
#include "simtile.h"
#include "fused.h"

namespace fal {

struct List16Args {              // cf. ListScanArgs (scan.h); tiles = groups of four 32-row list slices
    const __half* X16;           // float16 rows in sorted-row order
    const uint16_t* sq16;        // optional [n][64 x u16 column | 64 x f16 value]: the rows' sparse records (ivf.h) -- the queries are then
                                 // gathered as these 256 bytes and expanded in LDS (list16s.hip) instead of as dense rows
    const int32_t* perm;         // [n] list-order position -> sorted row (the resident rows of a list)
    int d;
    const int64_t* list_off;
    const int64_t* inv_off;
    const int64_t* ltile_off;
    const int32_t* inv_row;      // [entries] sorted row of the query of every probe-table entry
    const int64_t* inv_dest;     // element index in `keys` where that query's segment for this list starts
    int64_t list_begin, list_end;
    int64_t tile_begin, n_tiles_max;
    uint16_t* keys;              // round(approximate similarity * 65535)
    int64_t keys_base;
    uint16_t* sink;              // >= 64 entries of scratch for masked stores
    int64_t n_rows;              // rows of X16 (row ids read past a list's end are clamped)
};

struct Kept16Args {              // pairs16.hip
    const uint16_t* keys;
    int64_t keys_base;
    const DenseJob* jobs;        // IVF tile table (tile order = list-order positions)
    int64_t tile_begin, n_tiles;
    const int32_t* tile_job;     // (launch_kept16: the table launch_select16 built for the same tiles)
    int n_probe;
    const int32_t* probes;
    const int64_t* list_off;
    const int64_t* q_sim_off;
    const int32_t* perm;         // [n] list-order position -> sorted row
    const float* pmz_l;          // [n] precursor m/z by list-order position (ascending inside a list)
    const float* rt;             // [n] by sorted row, or nullptr
    double tol, rt_tol;
    float tol_f, rt_f;
    int is_da;
    const int2* gsel;            // [n] by sorted row (select16_kernel)
    uint32_t* gkept_id;          // [n, FAL_FUSED_KEEP] kept candidates (sorted rows), entries 0 .. count - 1
    int32_t* gkcnt;              // [n, 2] count (first 32 | the rest) | 0x100 ambiguous | 0x200 more than the hand-off holds
};

struct Select16Args {
    const uint16_t* keys;
    int64_t keys_base;
    int k;
    const DenseJob* jobs;        // IVF tile table (q_row0 = first list-order position of the bucket, c_row0 = its list 0)
    int n_jobs;
    int64_t tile_begin;
    const int32_t* tile_job;     // (filled by launch_select16)
    const int32_t* tile_p0;      // (filled by launch_select16) list-order position of a tile's first query
    const int64_t* q_sim_off;
    const int32_t* perm;
    QThr* thr;                   // hand-off (fused.h), by sorted row
    float* gmem_v;               // members: approximate value ...
    uint32_t* gmem_id;           // ... and position in the query's key stream (resolve_kernel: -> row where needed)
    int2* gsel;                  // optional [n] by sorted row: (smallest key + 1 a window candidate must reach to stay in the
                                 // race, largest key + 1 that is still ambiguous) -- for kept16_kernel (pairs16.hip)
    int64_t n_tiles;             // (filled by launch_select16)
    int64_t max_keys;            // the most keys any query of the search has (picks the kernel form)
    const int64_t* tile_l0;      // (filled by launch_select16) global id of list 0 of a tile's bucket
    int fuse_kept;               // 1: kept16_query (kept16.h) runs as the tail of every query's selection (single-pass searches)
    Kept16Args kept;             // ... with these arguments (the caller then skips launch_kept16)
    int32_t* big_count;          // queries with more than 2,048 keys: second pass with 64 keys per lane
    int32_t* big_list;
    int big_cap;
};


bool ivf16_supports(int d);
int launch_gather_pmz(fal_ctx* ctx, const float* pmz, const int32_t* perm, int64_t n, float* out);
int launch_kept16(fal_ctx* ctx, const Kept16Args& a, int64_t n_tiles);        // right after launch_select16 on the same tiles
int launch_pairs16(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles32);
int launch_pos_of_row(fal_ctx* ctx, const int32_t* perm, int64_t n, int32_t* pos_of_row);
int launch_list16(fal_ctx* ctx, const List16Args& a);
int launch_list16r(fal_ctx* ctx, const List16Args& a);      // list16r.hip (called by launch_list16)
int launch_list16s(fal_ctx* ctx, const List16Args& a);      // list16s.hip (called by launch_list16)
int launch_select16(fal_ctx* ctx, const Select16Args& a, int64_t n_tiles);

}  // namespace fal
