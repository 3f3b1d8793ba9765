// TEMPORARY: entry points not implemented yet return FAL_EUNSUPPORTED (never a CPU fallback).
#include "common.h"
#define NOTYET(name) fal::set_error(name ": not implemented yet"); return FAL_EUNSUPPORTED
extern "C" {
int fal_precursor_splits(fal_ctx*, const float*, int64_t, double, int, int64_t, double, int, int64_t*, int64_t, int64_t*) { NOTYET("fal_precursor_splits"); }
int fal_filter_neighbors(fal_ctx*, const float*, const int32_t*, int64_t, int, const float*, const float*, double, int, double, int, int32_t*, float*) { NOTYET("fal_filter_neighbors"); }
int fal_dbscan(fal_ctx*, const int32_t*, const float*, int64_t, int, float, int32_t*, int64_t*) { NOTYET("fal_dbscan"); }
int fal_refine_clusters(fal_ctx*, int32_t*, int64_t, const float*, const float*, double, int, double, int64_t*) { NOTYET("fal_refine_clusters"); }
int fal_finalize(fal_ctx*, const int32_t*, int64_t, int64_t, const int64_t*, const int32_t*, const float*, int, int32_t*, int32_t*, int64_t*) { NOTYET("fal_finalize"); }
int fal_sort_by_precursor(fal_ctx*, const float*, int64_t, int64_t*, float*) { NOTYET("fal_sort_by_precursor"); }
int fal_gather_f32(fal_ctx*, const float*, const int64_t*, int64_t, float*) { NOTYET("fal_gather_f32"); }
}
