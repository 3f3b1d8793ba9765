// Plumbing around the hot path: stable sort by precursor m/z (reference cluster.py:73-85),
// precursor-m/z bucket boundaries (cluster.py:159-209), gathers, device-wide scans.
// The radix sort is rocPRIM's (AMD's own primitive library); everything else is hand-written.
#include <cstring>
#include <math.h>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdlib.h>
#include "common.h"
#include "ivf.h"
#include "util.h"

namespace fal {

// ------------------------------------------------------------------------------------------
// device-wide exclusive scan of int32 values (flags): out[i] = sum_{j<i} in[j]; *total_dev = sum
// ------------------------------------------------------------------------------------------
constexpr int kScanBlock = 1024;

template <class T>
__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(const T* __restrict__ in, int64_t n,
                                                                     int64_t* __restrict__ block_sums) {
    __shared__ int64_t ws[kScanBlock / 64];
    const int64_t i = blockIdx.x * (int64_t)kScanBlock + threadIdx.x;
    int64_t v = i < n ? in[i] : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t s = 0;
        for (int j = 0; j < kScanBlock / 64; ++j) s += ws[j];
        block_sums[blockIdx.x] = s;
    }
}

template <class T>
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const T* __restrict__ in, int64_t n,
                                                                const int64_t* __restrict__ block_off,
                                                                int64_t* __restrict__ out) {
    __shared__ int64_t ws[kScanBlock / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = blockIdx.x * (int64_t)kScanBlock + threadIdx.x;
    const int64_t v = i < n ? in[i] : 0;
    int64_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) ws[w] = x;
    __syncthreads();
    int64_t pre = block_off[blockIdx.x];
    for (int j = 0; j < w; ++j) pre += ws[j];
    if (i < n) out[i] = pre + x - v;
}

// the same with the scan of the block sums folded in: a block adds up the sums of the blocks in front of it itself (a few
// thousand L2-resident values), the last block writes the total -- two launches per scan instead of three and a copy
template <class T>
__global__ __launch_bounds__(kScanBlock) void scan_apply_fused_kernel(const T* __restrict__ in, int64_t n,
                                                                      const int64_t* __restrict__ block_sums,
                                                                      int64_t* __restrict__ out) {
    __shared__ int64_t ws[kScanBlock / 64];
    __shared__ int64_t front_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t part = 0;
    for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += kScanBlock) part += block_sums[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) ws[w] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t f = 0;
        for (int j = 0; j < kScanBlock / 64; ++j) f += ws[j];
        front_s = f;
    }
    __syncthreads();
    const int64_t front = front_s;
    __syncthreads();
    const int64_t i = blockIdx.x * (int64_t)kScanBlock + threadIdx.x;
    const int64_t v = i < n ? in[i] : 0;
    int64_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) ws[w] = x;
    __syncthreads();
    int64_t pre = front;
    for (int j = 0; j < w; ++j) pre += ws[j];
    if (i < n) out[i] = pre + x - v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanBlock - 1) out[n] = pre + x;      // the total
}

// out[0..n) exclusive prefix of in[0..n) ; out[n] = total.  tmp: ceil(n/1024)+1 int64 (x2).
template <class T>
static int device_scan_t(fal_ctx* ctx, const T* in, int64_t n, int64_t* out, int scratch_slot) {
    if (n <= 0) {
        FAL_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(int64_t), ctx->stream));
        return FAL_OK;
    }
    const int64_t nb = ceil_div(n, kScanBlock);
    int64_t* tmp = nullptr;
    FAL_TRY(ctx->reserve(scratch_slot, sizeof(int64_t) * (size_t)(2 * nb + 2), (void**)&tmp));
    int64_t* sums = tmp;
    int64_t* offs = tmp + nb + 1;
    hipLaunchKernelGGL(scan_block_sums_kernel<T>, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, in, n, sums);
    if (nb <= 4096) {
        hipLaunchKernelGGL(scan_apply_fused_kernel<T>, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, in, n, sums, out);
        FAL_CHECK_HIP(hipGetLastError());
        return FAL_OK;
    }
    FAL_TRY(launch_exclusive_scan(ctx, sums, nb, offs));
    hipLaunchKernelGGL(scan_apply_kernel<T>, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, in, n, offs, out);
    // total
    FAL_CHECK_HIP(hipMemcpyAsync(out + n, offs + nb, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int device_scan_i32(fal_ctx* ctx, const int32_t* in, int64_t n, int64_t* out, int scratch_slot) {
    return device_scan_t<int32_t>(ctx, in, n, out, scratch_slot);
}

int device_scan_i64(fal_ctx* ctx, const int64_t* in, int64_t n, int64_t* out, int scratch_slot) {
    return device_scan_t<int64_t>(ctx, in, n, out, scratch_slot);
}

// ------------------------------------------------------------------------------------------
// sort / gather
// ------------------------------------------------------------------------------------------
__global__ void iota_i64_kernel(int64_t* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = i;
}

__global__ void gather_f32_kernel(const float* __restrict__ src, const int64_t* __restrict__ order, int64_t n,
                                  float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = src[order[i]];
}

int sort_pairs_u32_i32(fal_ctx* ctx, const uint32_t* kin, uint32_t* kout, const int32_t* vin, int32_t* vout,
                       int64_t n, int end_bit, int scratch_slot) {
    // Onesweep radix passes from 128 k keys on (rocprim's default merge sort below 1 M keys takes ~20 launches)
    using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 131072>;
    size_t bytes = 0;
    FAL_CHECK_HIP(rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0, end_bit, ctx->stream));
    void* tmp = nullptr;
    FAL_TRY(ctx->reserve(scratch_slot, bytes, &tmp));
    FAL_CHECK_HIP(rocprim::radix_sort_pairs<SortConfig>(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0, end_bit, ctx->stream));
    return FAL_OK;
}

// ------------------------------------------------------------------------------------------
// a5: flags where a new bucket may start
// ------------------------------------------------------------------------------------------
// bit 0: reference gap  mass_diff(mz[i], mz[i-1]) > tol  (cluster.py:186-197; float32 difference and
//        division, float64 product with 10**6 -- the typing numba gives spectrum_utils.mass_diff)
// bit 1: fixed-window cut  floor(mz[i] / w) != floor(mz[i-1] / w)   [build rule]
__global__ void split_flags_kernel(const float* __restrict__ mz, int64_t n, double tol, int is_da, double window,
                                   int32_t* __restrict__ flag) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int f = 0;
        if (i > 0) {
            const float a = mz[i], b = mz[i - 1];
            const float diff = a - b;
            const double md = is_da ? (double)diff : (double)(diff / b) * 1e6;
            if (md > tol) f |= 1;
            if (window > 0.0 && floor((double)a / window) != floor((double)b / window)) f |= 2;
        }
        flag[i] = f;
    }
}

__global__ void compact_flags_kernel(const int32_t* __restrict__ flag, const int64_t* __restrict__ pos, int64_t n,
                                     int64_t* __restrict__ out_idx, int32_t* __restrict__ out_flag) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (flag[i]) {
            out_idx[pos[i]] = i;
            out_flag[pos[i]] = flag[i];
        }
    }
}

// ---- multi-GPU front: whole precursor windows are the unit dealt to ranks -------------------------------------
__device__ __forceinline__ int64_t window_of(float mz, double interval, int64_t n_windows) {
    // slot of the window floor(mz / interval) (the arithmetic of split_flags_kernel) in the table of n_windows deal units:
    // windows beyond the table WRAP AROUND (w mod n_windows) instead of collapsing into the last slot -- with a small
    // mz_interval (0.05 -> the table ends at 819 m/z) a clamp sent most of the dataset to one unit = one rank;
    // wrapped, a slot holds whole windows w, w + n_windows, ... : still a valid deal unit (buckets never cross a window),
    // and the load stays spread.  (NaN / negative -> slot 0)
    // (+inf, or a quotient too large for the integer cast: the last slot -- fmod(inf, n) is NaN and its cast undefined)
    const double w = floor((double)mz / interval);
    if (!(w >= 0.0)) return 0;
    if (!(w < 9.0e15)) return n_windows - 1;
    return w < (double)n_windows ? (int64_t)w : (int64_t)fmod(w, (double)n_windows);
}

// the partitions of one job (precursor charges) are counted by ONE pair of launches: blockIdx.y = partition
constexpr int kWindowParts = 64;
struct WindowParts {
    const float* mz[kWindowParts];
    int64_t n[kWindowParts];
};

// lo_hi[2 p] = smallest, lo_hi[2 p + 1] = largest window of partition p that holds a spectrum (one atomic pair per wave)
__global__ void window_range_kernel(WindowParts parts, double interval, int64_t n_windows, int32_t* __restrict__ lo_hi) {
    const float* __restrict__ mz = parts.mz[blockIdx.y];
    const int64_t n = parts.n[blockIdx.y];
    lo_hi += 2 * blockIdx.y;
    int32_t lo = INT32_MAX, hi = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t w = (int32_t)window_of(mz[i], interval, n_windows);
        lo = min(lo, w);
        hi = max(hi, w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && lo <= hi) {
        atomicMin(&lo_hi[0], lo);
        atomicMax(&lo_hi[1], hi);
    }
}

// counts[w] += spectra of window w.  The windows in use span a few hundred to a few thousand counters -- a few dozen cache
// lines: counted by global atomics directly, 10 M increments serialise on those lines (measured 2 ms at 10 M spectra).  Every
// workgroup counts its share of the spectra in LDS first (when the span fits) and adds only its non-zero bins.
constexpr int kWindowBins = 12288;
__global__ __launch_bounds__(1024) void window_counts_kernel(WindowParts parts, double interval, int64_t n_windows,
                                                             const int32_t* __restrict__ lo_hi, int32_t* __restrict__ counts) {
    __shared__ int32_t bins[kWindowBins];
    const float* __restrict__ mz = parts.mz[blockIdx.y];
    const int64_t n = parts.n[blockIdx.y];
    lo_hi += 2 * blockIdx.y;
    counts += n_windows * blockIdx.y;
    const int32_t lo = lo_hi[0], span = lo_hi[1] - lo_hi[0] + 1;
    const bool local = span >= 1 && span <= kWindowBins;
    if (local) {
        for (int i = threadIdx.x; i < span; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t w = (int32_t)window_of(mz[i], interval, n_windows);
        if (local) atomicAdd(&bins[w - lo], 1); else atomicAdd(&counts[w], 1);
    }
    if (local) {
        __syncthreads();
        for (int i = threadIdx.x; i < span; i += blockDim.x)
            if (bins[i]) atomicAdd(&counts[lo + i], bins[i]);
    }
}

__global__ void window_flags_kernel(const float* __restrict__ mz, int64_t n, double interval, int64_t n_windows,
                                    const int32_t* __restrict__ owner, int rank, int32_t* __restrict__ flag) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flag[i] = owner[window_of(mz[i], interval, n_windows)] == rank;
}

__global__ void window_compact_kernel(const float* __restrict__ mz, const int32_t* __restrict__ flag, const int64_t* __restrict__ pos,
                                      int64_t n, int64_t* __restrict__ rows_out, float* __restrict__ mz_out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (flag[i]) {
            rows_out[pos[i]] = i;
            mz_out[pos[i]] = mz[i];
        }
}

__global__ void nonzero_i32_kernel(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i] != 0;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::iota_i64_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

extern "C" {

int fal_sort_by_precursor(fal_ctx* ctx, const float* precursor_mz, int64_t n, int64_t* order_out, float* mz_sorted_out) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0, FAL_EINVAL, "fal_sort_by_precursor: bad argument");
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(precursor_mz && order_out && mz_sorted_out, FAL_EINVAL, "fal_sort_by_precursor: NULL array");
    int64_t* iota = nullptr;
    FAL_TRY(ctx->reserve(SLOT_SORT, sizeof(int64_t) * (size_t)n, (void**)&iota));
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(iota_i64_kernel, dim3(grid), dim3(256), 0, ctx->stream, iota, n);
    size_t bytes = 0;
    // LSD radix sort is stable: equal m/z keep dataset order (the oracle sorts with kind="stable").  rocprim's default takes its
    // merge sort below 1 M keys (~20 launches; 0.173 ms for the 700 k keys of a BASELINE configs[1] partition); with the limit
    // at 128 k keys the Onesweep radix passes (histogram + scan + 4 passes) run instead: 0.129 ms (tools/sort_ab.py)
    using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 131072>;
    void* tmp = nullptr;
    FAL_CHECK_HIP(rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, precursor_mz, mz_sorted_out, iota, order_out, (size_t)n, 0,
                                                        32, ctx->stream));
    FAL_TRY(ctx->reserve(SLOT_SORT2, bytes, &tmp));
    FAL_CHECK_HIP(rocprim::radix_sort_pairs<SortConfig>(tmp, bytes, precursor_mz, mz_sorted_out, iota, order_out, (size_t)n, 0, 32,
                                                        ctx->stream));
    return FAL_OK;
}

int fal_gather_f32(fal_ctx* ctx, const float* src, const int64_t* order, int64_t n, float* out) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0, FAL_EINVAL, "fal_gather_f32: bad argument");
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(src && order && out, FAL_EINVAL, "fal_gather_f32: NULL array");
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
    hipLaunchKernelGGL(gather_f32_kernel, dim3(grid), dim3(256), 0, ctx->stream, src, order, n, out);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int fal_window_counts(fal_ctx* ctx, const float* const* precursor_mz, const int64_t* n, int n_parts, double mz_interval,
                      int64_t n_windows, int32_t* counts, int32_t* counts_host) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n_parts >= 0 && mz_interval > 0.0 && n_windows >= 1 && (n_parts == 0 || (precursor_mz && n && counts)),
                FAL_EINVAL, "fal_window_counts: bad argument");
    if (n_parts == 0) return FAL_OK;
    FAL_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)n_windows * n_parts, ctx->stream));
    int32_t* lo_hi_all = nullptr;
    FAL_TRY(ctx->reserve(SLOT_MISC, sizeof(int32_t) * 2 * (size_t)(n_parts + kWindowParts), (void**)&lo_hi_all));
    for (int p0 = 0; p0 < n_parts; p0 += kWindowParts) {
        int32_t* lo_hi = lo_hi_all + 2 * p0;
        const int np = std::min(kWindowParts, n_parts - p0);
        WindowParts parts;
        int32_t init[2 * kWindowParts];
        int64_t n_max = 0;
        for (int p = 0; p < kWindowParts; ++p) {
            parts.mz[p] = p < np ? precursor_mz[p0 + p] : nullptr;
            parts.n[p] = p < np ? n[p0 + p] : 0;
            FAL_REQUIRE(parts.n[p] >= 0 && (parts.n[p] == 0 || parts.mz[p]), FAL_EINVAL, "fal_window_counts: NULL array");
            n_max = std::max(n_max, parts.n[p]);
            init[2 * p] = INT32_MAX;
            init[2 * p + 1] = 0;
        }
        FAL_TRY(ctx->upload(lo_hi, init, sizeof(init)));
        if (n_max == 0) continue;
        // the CUs are shared by the partitions of the launch
        const int per = std::max(1, (2 * ctx->num_cus + np - 1) / np);
        const dim3 grid((unsigned)std::min<int64_t>(ceil_div(n_max, 1024), per), (unsigned)np);
        int32_t* c = counts + (int64_t)p0 * n_windows;
        hipLaunchKernelGGL(window_range_kernel, grid, dim3(1024), 0, ctx->stream, parts, mz_interval, n_windows, lo_hi);
        hipLaunchKernelGGL(window_counts_kernel, grid, dim3(1024), 0, ctx->stream, parts, mz_interval, n_windows, lo_hi, c);
        FAL_CHECK_HIP(hipGetLastError());
    }
    if (counts_host) {
        FAL_CHECK_HIP(hipMemcpyAsync(counts_host, counts, sizeof(int32_t) * (size_t)n_windows * n_parts, hipMemcpyDeviceToHost,
                                     ctx->stream));
        FAL_CHECK_HIP(hipMemcpyAsync(counts_host + n_windows * n_parts, lo_hi_all, sizeof(int32_t) * 2 * (size_t)n_parts,
                                     hipMemcpyDeviceToHost, ctx->stream));
    }
    return FAL_OK;
}

int fal_window_select(fal_ctx* ctx, const float* precursor_mz, int64_t n, double mz_interval, int64_t n_windows,
                      const int32_t* owner, int rank, int64_t* rows_out, float* mz_out, int64_t* count) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && mz_interval > 0.0 && n_windows >= 1 && owner && count, FAL_EINVAL, "fal_window_select: bad argument");
    *count = 0;
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(precursor_mz && rows_out && mz_out, FAL_EINVAL, "fal_window_select: NULL array");
    int32_t* flag = nullptr;
    int64_t* pos = nullptr;
    FAL_TRY(ctx->reserve(SLOT_MISC, sizeof(int32_t) * (size_t)n, (void**)&flag));
    FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(int64_t) * (size_t)(n + 1), (void**)&pos));
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(window_flags_kernel, dim3(grid), dim3(256), 0, ctx->stream, precursor_mz, n, mz_interval, n_windows, owner, rank, flag);
    FAL_TRY(device_scan_i32(ctx, flag, n, pos, SLOT_SORT));
    hipLaunchKernelGGL(window_compact_kernel, dim3(grid), dim3(256), 0, ctx->stream, precursor_mz, flag, pos, n, rows_out, mz_out);
    FAL_CHECK_HIP(hipGetLastError());
    int64_t* h = nullptr;
    FAL_TRY(ctx->pinned_reserve(sizeof(int64_t), (void**)&h));
    FAL_CHECK_HIP(hipMemcpyAsync(h, pos + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *count = *h;
    return FAL_OK;
}

int fal_precursor_splits(fal_ctx* ctx, const float* mz, int64_t n, double tol, int tol_is_da, int64_t batch_size,
                         double mz_interval, int chunk_last, int64_t* splits_out, int64_t max_splits,
                         int64_t* n_splits) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && splits_out && n_splits && n >= 0 && batch_size >= 1 && max_splits >= 2, FAL_EINVAL,
                "fal_precursor_splits: bad argument");
    *n_splits = 0;
    std::vector<int64_t> gap_idx;
    std::vector<int32_t> gap_flag;
    if (n > 1) {
        FAL_REQUIRE(mz, FAL_EINVAL, "fal_precursor_splits: NULL mz");
        int32_t *flag = nullptr, *nz = nullptr, *cflag = nullptr;
        int64_t *pos = nullptr, *cidx = nullptr;
        FAL_TRY(ctx->reserve(SLOT_MISC, sizeof(int32_t) * (size_t)n * 2, (void**)&flag));
        nz = flag + n;
        FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(int64_t) * (size_t)(n + 1), (void**)&pos));
        const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 4096);
        hipLaunchKernelGGL(split_flags_kernel, dim3(grid), dim3(256), 0, ctx->stream, mz, n, tol, tol_is_da, mz_interval, flag);
        hipLaunchKernelGGL(nonzero_i32_kernel, dim3(grid), dim3(256), 0, ctx->stream, flag, n, nz);
        FAL_TRY(device_scan_i32(ctx, nz, n, pos, SLOT_SORT));
        // compact on the device right away, then ONE synchronisation: the count and the first K entries
        // come back through a pinned buffer (a second readback only if there are more than K gaps)
        FAL_TRY(ctx->reserve(SLOT_SORT2, (sizeof(int64_t) + sizeof(int32_t)) * (size_t)n, (void**)&cidx));
        cflag = reinterpret_cast<int32_t*>(cidx + n);
        hipLaunchKernelGGL(compact_flags_kernel, dim3(grid), dim3(256), 0, ctx->stream, flag, pos, n, cidx, cflag);
        const int64_t K = std::min<int64_t>(n, 65536);
        unsigned char* pin = nullptr;
        FAL_TRY(ctx->pinned_reserve(sizeof(int64_t) * (size_t)(K + 1) + sizeof(int32_t) * (size_t)K, (void**)&pin));
        int64_t* h_total = reinterpret_cast<int64_t*>(pin);
        int64_t* h_idx = h_total + 1;
        int32_t* h_flag = reinterpret_cast<int32_t*>(h_idx + K);
        FAL_CHECK_HIP(hipMemcpyAsync(h_total, pos + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipMemcpyAsync(h_idx, cidx, sizeof(int64_t) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipMemcpyAsync(h_flag, cflag, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        const int64_t total = *h_total;
        if (total > 0) {
            gap_idx.resize(total);
            gap_flag.resize(total);
            const int64_t first = std::min(total, K);
            memcpy(gap_idx.data(), h_idx, sizeof(int64_t) * (size_t)first);
            memcpy(gap_flag.data(), h_flag, sizeof(int32_t) * (size_t)first);
            if (total > K) {
                FAL_CHECK_HIP(hipMemcpyAsync(gap_idx.data() + K, cidx + K, sizeof(int64_t) * (size_t)(total - K), hipMemcpyDeviceToHost, ctx->stream));
                FAL_CHECK_HIP(hipMemcpyAsync(gap_flag.data() + K, cflag + K, sizeof(int32_t) * (size_t)(total - K), hipMemcpyDeviceToHost, ctx->stream));
                FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            }
        }
    }
    // reference cluster.py:183-208 over the gap positions (host; a few entries per bucket)
    std::vector<int64_t> splits(1, 0);
    auto chunk_block = [&](int64_t block) {
        const int64_t n_chunks = (block + batch_size - 1) / batch_size;
        const int64_t chunk = block / n_chunks;
        for (int64_t c = 0; c < block % n_chunks; ++c) splits.push_back(splits.back() + chunk + 1);
        for (int64_t c = 0; c < n_chunks - (block % n_chunks); ++c) splits.push_back(splits.back() + chunk);
    };
    // [build rule] fixed windows: the reference's rule runs INSIDE every window floor(mz / mz_interval) -- a window boundary
    // ends a block like a gap does, and the chunk rule applies to the window's last block too -- so the buckets of a window
    // depend on that window's spectra alone (what lets GPUs take whole windows without seeing the rest of the dataset).
    // Without windows (mz_interval = 0) this is cluster.py:183-208 unchanged (+ the chunked last block).
    for (size_t g = 0; g < gap_idx.size(); ++g) {
        const int64_t i = gap_idx[g], block = i - splits.back();
        const bool window_end = (gap_flag[g] & 2) != 0;      // = the end of the array the reference's loop sees
        if (block < batch_size || (window_end && !chunk_last)) splits.push_back(i);
        else chunk_block(block);
    }
    if (chunk_last && n - splits.back() >= batch_size) chunk_block(n - splits.back());   // [build rule]
    if (splits.back() != n || n == 0) splits.push_back(n);
    std::vector<int64_t> all(splits);
    FAL_REQUIRE((int64_t)all.size() <= max_splits, FAL_EINVAL, "fal_precursor_splits: %zu boundaries do not fit max_splits %lld",
                all.size(), (long long)max_splits);
    memcpy(splits_out, all.data(), sizeof(int64_t) * all.size());
    *n_splits = (int64_t)all.size();
    return FAL_OK;
}

}  // extern "C"
