// Internal interface of scan.hip / ivf_fine.hip (launchers shared with ivf.hip).
#pragma once
#include "common.h"
#include "simtile.h"

namespace fal {

enum { EPI_STORE = 0, EPI_ARGMAX = 1 };
enum { MODE_DENSE = 0, MODE_IVF = 1 };

// every sims buffer is allocated with this many floats of slack after its last row: the select kernel
// reads whole 64*R-key rounds without bounds checks, the f16 scan parks idle waves' stores there
constexpr size_t kSimsSlack = 2048;

struct SelectArgs {
    const float* sims;       // sims buffer of the current batch
    int64_t sims_base;       // float index the buffer starts at
    int k;                   // how many to keep
    float* out_sim;          // [rows, k]
    int32_t* out_idx;        // [rows, k]
    // MODE_DENSE: queries are the rows of dense tiles [tile_begin, ...)
    const DenseJob* jobs;
    int n_jobs;
    int64_t tile_begin;
    const int32_t* tile_job; // [tiles of the launch] job of every tile (filled by launch_select)
    int ids_are_rows;        // ids = job.c_row0 + position (else position)
    // MODE_IVF: queries are the rows of IVF tiles (jobs: q_row0 = first list-order position of
    // the bucket, c_row0 = global id of its list 0, nc = its n_list)
    int n_probe;
    const int32_t* probes;       // [n, n_probe] bucket-local list ids (-1 = none)
    const int64_t* list_off;     // [total_lists + 1] positions in list order
    const int64_t* q_sim_off;    // [32 * tiles] where the sims of tile-order slot 32*t+lane start
    const int32_t* perm;         // [n] list-order position -> sorted row
    // a8 fused into the selection (nb_idx != nullptr): the k selected candidates are filtered by the
    // precursor / RT tolerance, only the survivors are sorted, and the neighbour lists are written
    // instead of out_sim / out_idx
    const float* f_pmz;          // [n] precursor m/z by sorted row
    const float* f_rt;           // [n] or nullptr
    double f_tol, f_rt_tol;
    int f_is_da;
    int f_keep;                  // n_neighbors
    int32_t* nb_idx;             // [rows, f_keep]
    float* nb_dist;              // [rows, f_keep]
    int32_t* nb_count;           // [rows] stored neighbours per row, or nullptr
};

// the shared-stream form of the symmetric flat scan (scan.hip): jobs sorted by decreasing size, job j on XCD list j % 8
bool dense4_supports(int d);
int launch_dense4(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, const DenseJob* jobs_host, int n_jobs, float* sims,
                  int64_t sims_base);
// the same scan with the query tile split along K between two waves per tile: two waves per SIMD (dense4ab.hip; low_dim 400)
bool dense4ab_supports(int d);
int launch_dense4ab(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, const DenseJob* jobs_host, int n_jobs, float* sims,
                    int64_t sims_base);
int launch_dense_tiny4(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, int n_jobs, float* sims, int64_t sims_base);
// flat buckets of fewer than 64 rows at low_dim > 512: exact fmaf chains on the vector ALU (scan.hip)
int launch_flat_exact_small(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, int n_jobs, float* sims, int64_t sims_base);
// xcd_list_tiles > 0 selects XCD-list mode (simtile.h): `jobs` = the launch's jobs sorted by
// decreasing size with xtile0 filled in, xcd_list_tiles = tiles of the longest of the 8 lists.
int launch_dense(fal_ctx* ctx, int stage, int epi, const float* Q, const float* Cm, int d, const DenseJob* jobs,
                 int n_jobs, int64_t tile_begin, int64_t n_tiles, float* sims, int64_t sims_base, int32_t* assign,
                 int64_t xcd_list_tiles = 0);
// Shared-stream list assignment (assign.hip): job = (a segment of a bucket's rows) x (up to 128 of its centroids)
constexpr int kAssignSeg = 2048;      // rows per segment
constexpr int kAssignGroup = 128;     // centroids per job: one 32-centroid tile per wave
constexpr int kAssignMergeLists = 2048;     // lists per bucket up to which the float16 assignment runs in groups of 128 + a merge
                                            // (16 groups = 64 subgroups of 32; round 4 stopped at 512 and sent a 1,024-list
                                            // bucket -- `--batch_size 65536` on 45 k-row windows -- to the exact fp32 kernels:
                                            // build 584 ms instead of 77 per 10 M spectra, profiles/NOTES.md r5)
struct AssignJob {
    int64_t row0;        // first row of the segment (sorted rows = rows of X)
    int64_t cent0;       // global row of the group's first centroid
    int32_t nrows;       // rows in the segment (<= kAssignSeg)
    int32_t ncent;       // centroids in the group (1..kAssignGroup)
    int32_t id_base;     // bucket-local list id of the group's first centroid
    int32_t part0;       // merge / group jobs (assign16.hip): position of the segment's first row among the rows of ALL merge
                         // buckets = the row index of the partial arrays (0 elsewhere)
};
// keys: u64[n] scratch (cleared here); assign[i] is written for every row a job covers
// jobs[0, n_jobs): 4-wave jobs (<= 128 centroids each); jobs[n_jobs, n_jobs + n_wave_jobs): one-wave jobs (<= 32 centroids)
int launch_assign(fal_ctx* ctx, int stage, const float* X, const float* centroids, int d, const AssignJob* jobs, int64_t n_jobs,
                  int64_t n_wave_jobs, int64_t n, unsigned long long* keys, int32_t* assign);
// Assignment with a float16 prefilter (assign16.hip): buckets with <= 128 lists in one job per row segment, buckets with
// <= kAssignMergeLists lists in groups of 128 + a merge over the groups; identical results.  Job table layout: see launch_assign16
bool assign16_supports(int d);
int launch_cvt_f16(fal_ctx* ctx, const float* in, void* out, int64_t count);
int launch_assign16(fal_ctx* ctx, int stage, const void* X16, const float* X, const void* C16, const float* Cn, int d,
                    const AssignJob* jobs, int64_t n_single, int64_t n_merge, int64_t n_group, int64_t n_rows, int32_t* assign,
                    uint16_t* ckeys = nullptr, int ckeys_stride = 0, const uint16_t* sp_cols = nullptr,
                    const float* sp_vals = nullptr, int merge_max_lists = 4 * kAssignGroup, int64_t merge_rows = 0);
// List-major IVF fine scan (ivf_fine.hip): tile = (one inverted list, 32 of the queries that probe it)
struct ListScanArgs {
    const float* Xl;             // vectors in (bucket, list, row) order
    int d;
    const int64_t* list_off;     // [total_lists + 1] rows of every list (positions in Xl)
    const int64_t* inv_off;      // [total_lists + 1] entries of every list in the inverted probe table
    const int64_t* ltile_off;    // [total_lists + 1] tiles of every list, prefix (tile = 2^group_shift list rows)
    int group_shift;             // 7: tiles = 4-wave groups of four 32-row list slices (ivf_list4_kernel)
    const int32_t* inv_q;        // [pairs] query position (row of Xl)
    const int64_t* inv_dest;     // [pairs] float index in `sims` where that query's sims for this list start
    int64_t list_begin, list_end;   // lists of this launch
    int64_t tile_begin, n_tiles_max;   // first tile of list_begin; upper bound on the tiles of the launch
    float* sims;
    int64_t sims_base;
    float* sink;                 // >= 64 floats of scratch for masked stores
};
int launch_list_scan(fal_ctx* ctx, const ListScanArgs& a);
int launch_select(fal_ctx* ctx, int stage, int mode, const SelectArgs& a, int64_t n_blocks, hipStream_t on = nullptr);
// f16-MFMA flat scan (scan16.hip): jobs sorted by decreasing size, xtile0 = 128-query tiles of earlier
// jobs of the same XCD list; list_tiles = longest list.  planes = 1 (f16 rows) or 2 (hi/lo split).
// sink: >= 256 floats (16-byte aligned) of scratch that idle waves store to.
int launch_scan16(fal_ctx* ctx, int planes, const void* Xs, int d, const DenseJob* jobs, int n_jobs, int64_t list_tiles,
                  float* sims, int64_t sims_base, float* sink);

}  // namespace fal
