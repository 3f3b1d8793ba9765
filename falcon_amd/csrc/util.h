// Small shared launch helpers (implemented in sortutil.hip).
#pragma once
#include "common.h"

namespace fal {
// out[0..n) = exclusive prefix of in (int32 flags/counts), out[n] = total (all on device)
int device_scan_i32(fal_ctx* ctx, const int32_t* in, int64_t n, int64_t* out, int scratch_slot);
int device_scan_i64(fal_ctx* ctx, const int64_t* in, int64_t n, int64_t* out, int scratch_slot);   // same, int64 input
// stable LSD radix sort of (uint32 key, int32 value) pairs on bits [0, end_bit)
int sort_pairs_u32_i32(fal_ctx* ctx, const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout,
                       int64_t n, int end_bit, int scratch_slot);
// a9 / a10 / a11+a12 with every count left on the device (graph.hip, tail.hip)
// (*extent_out: per row, one past its last stored neighbour -- valid until SLOT_DB is reserved again)
int dbscan_dev(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float eps, int32_t* labels,
               int64_t** d_count_out, const int32_t** extent_out = nullptr, const int32_t* nb_count = nullptr);
// f4: hierarchical clustering of the neighbour graph cut at t (linkage.hip): method 0 single, 1 complete, 2 average
int linkage_dev(fal_ctx* ctx, const int32_t* nb_idx, const float* nb_dist, int64_t n, int k, float t, int method, int32_t* labels,
                int64_t** d_count_out);
int refine_dev(fal_ctx* ctx, int32_t* labels, int64_t n, const float* mz, const float* rt, double tol, int is_da,
               double rt_tol, const int64_t* d_count_in, int64_t** d_count_out);
int finalize_dev(fal_ctx* ctx, const int32_t* labels_sorted, int64_t n, const int64_t* d_count,
                 const int64_t* row_order, const int32_t* nb_idx, const float* nb_dist, int k, int32_t* labels_out,
                 int32_t* medoids_out, int64_t** d_noise_out, const int32_t* extent = nullptr);
}  // namespace fal
