// Small shared launch helpers (implemented in sortutil.hip).
#pragma once
#include "common.h"

namespace fal {
// out[0..n) = exclusive prefix of in (int32 flags/counts), out[n] = total (all on device)
int device_scan_i32(fal_ctx* ctx, const int32_t* in, int64_t n, int64_t* out, int scratch_slot);
// stable LSD radix sort of (uint32 key, int32 value) pairs on bits [0, end_bit)
int sort_pairs_u32_i32(fal_ctx* ctx, const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout,
                       int64_t n, int end_bit, int scratch_slot);
}  // namespace fal
