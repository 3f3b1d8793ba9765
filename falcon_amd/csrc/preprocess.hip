// f1: spectrum preprocessing on the device -- the step immediately in front of the hot path
// (reference falcon/cluster/spectrum.py:73-169 `process_spectrum`, validity check spectrum.py:27-52,
// normalisation spectrum.py:55-70).  Same CSR layout as the path's input (spectrum.py:284-296).
//
// One wave per spectrum, two kernels with a device-wide scan of the surviving peak counts between them:
//   flags : m/z range cut (135) -> precursor-peak removal (139-149) -> base-peak intensity filter and
//           top-`max_peaks_used` selection (151-155), the validity check after every step; one keep-byte
//           per raw peak in scratch, peaks re-read from L1/L2 (a spectrum is a few KB);
//   emit  : compaction in m/z order, intensity scaling (157: root / log / rank), L2 normalisation (158).
// HBM-bound byte/integer work: 12 B read per raw peak (+ re-reads that stay in cache), 8 B written per
// kept peak.  Arithmetic conventions are the oracle's (`oracle.process_spectra`): m/z comparisons in
// float64, the intensity threshold in float32, sqrt / log2 in float64 rounded to float32, the norm in
// the fixed float64 tree of vectorize.hip.  spectrum_utils 0.3.5 itself is absent: PARITY UNPINNED.
#include <math.h>
#include <algorithm>
#include "common.h"
#include "ivf.h"
#include "util.h"
#include "simtile.h"

namespace fal {

constexpr double kProton = 1.0072766;

struct PrepParams {
    int min_peaks;
    double min_mz_range;
    double mz_min, mz_max;
    int has_min, has_max;
    double rm_tol;            // < 0: keep the precursor peak
    float min_intensity;      // fraction of the base peak
    int filter_intensity;     // min_intensity or max_peaks given
    int max_peaks;            // 0: unlimited
    int scaling;              // 0 none, 1 root, 2 log, 3 rank
};

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// count, min and max m/z of the peaks flagged 1 (spectrum.py:27-52 needs len, mz[0], mz[-1])
__device__ __forceinline__ bool still_valid(const double* __restrict__ mz, const uint8_t* __restrict__ flag, int64_t b,
                                            int64_t e, int lane, const PrepParams& p, int* count_out) {
    int cnt = 0;
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t j0 = b; j0 < e; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool k = j < e && flag[j];
        if (k) {
            lo = fmin(lo, mz[j]);
            hi = fmax(hi, mz[j]);
        }
        cnt += __popcll(__ballot(k));
    }
    lo = wave_min_f64(lo);
    hi = wave_max_f64(hi);
    *count_out = cnt;
    return cnt >= max(p.min_peaks, 1) && (hi - lo) >= p.min_mz_range;
}

__global__ __launch_bounds__(256) void prep_flags_kernel(const double* __restrict__ mz, const float* __restrict__ intensity,
                                                         const int64_t* __restrict__ indptr, int64_t n,
                                                         const double* __restrict__ pmz, const int32_t* __restrict__ charge,
                                                         PrepParams p, uint8_t* __restrict__ flag,
                                                         int32_t* __restrict__ count, int32_t* __restrict__ valid) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t b = indptr[i], e = indptr[i + 1];
        int cnt = 0;
        bool ok = true;
        // ---- m/z range (spectrum.py:135) ------------------------------------------------------
        for (int64_t j0 = b; j0 < e; j0 += 64) {
            const int64_t j = j0 + lane;
            if (j < e) {
                const double m = mz[j];
                flag[j] = (!p.has_min || m >= p.mz_min) && (!p.has_max || m <= p.mz_max);
            }
        }
        __threadfence_block();
        ok = still_valid(mz, flag, b, e, lane, p, &cnt);
        // ---- precursor peak(s) (spectrum.py:139-149): every charge state z..1 of the neutral mass -------
        if (ok && p.rm_tol >= 0.0) {
            const int z = charge[i] != 0 ? abs(charge[i]) : 1;
            const double neutral = (pmz[i] - kProton) * (double)z;
            for (int64_t j0 = b; j0 < e; j0 += 64) {
                const int64_t j = j0 + lane;
                if (j < e && flag[j]) {
                    const double m = mz[j];
                    bool keep = true;
                    for (int c = z; c >= 1; --c) keep = keep && fabs(m - (neutral / (double)c + kProton)) > p.rm_tol;
                    flag[j] = keep;
                }
            }
            __threadfence_block();
            ok = still_valid(mz, flag, b, e, lane, p, &cnt);
        }
        // ---- intensity filter (spectrum.py:151-155) -----------------------------------------------------
        if (ok && p.filter_intensity) {
            float mx = -INFINITY;
            for (int64_t j0 = b; j0 < e; j0 += 64) {
                const int64_t j = j0 + lane;
                if (j < e && flag[j]) mx = fmaxf(mx, intensity[j]);
            }
            mx = wave_max_f32(mx);
            const float thr = p.min_intensity * mx;
            int kept = 0;
            for (int64_t j0 = b; j0 < e; j0 += 64) {
                const int64_t j = j0 + lane;
                bool k = false;
                if (j < e && flag[j]) {
                    k = intensity[j] >= thr;
                    flag[j] = k;
                }
                kept += __popcll(__ballot(k));
            }
            __threadfence_block();
            if (p.max_peaks > 0 && kept > p.max_peaks) {
                // the max_peaks most intense; equal intensities keep the lower index (= lower m/z).
                // Largest T with count(key >= T) >= max_peaks, by bits (keys: order-preserving uint of the float)
                uint32_t T = 0;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t c = T | (1u << bit);
                    int cn = 0;
                    for (int64_t j0 = b; j0 < e; j0 += 64) {
                        const int64_t j = j0 + lane;
                        const bool k = j < e && flag[j] && f32_sortable(intensity[j]) >= c;
                        cn += __popcll(__ballot(k));
                    }
                    if (cn >= p.max_peaks) T = c;
                }
                int gt = 0;
                for (int64_t j0 = b; j0 < e; j0 += 64) {
                    const int64_t j = j0 + lane;
                    gt += __popcll(__ballot(j < e && flag[j] && f32_sortable(intensity[j]) > T));
                }
                int need = p.max_peaks - gt;                    // how many of the keys == T are kept, in index order
                for (int64_t j0 = b; j0 < e; j0 += 64) {
                    const int64_t j = j0 + lane;
                    const bool live = j < e && flag[j];
                    const uint32_t key = live ? f32_sortable(intensity[j]) : 0u;
                    const bool tie = live && key == T;
                    const uint64_t tm = __ballot(tie);
                    const int before = __popcll(tm & ((1ull << lane) - 1ull));
                    if (live) flag[j] = key > T || (tie && before < need);
                    need = max(need - (int)__popcll(tm), 0);      // (__popcll is unsigned: keep the subtraction signed)
                }
                __threadfence_block();
            }
            ok = still_valid(mz, flag, b, e, lane, p, &cnt);
        }
        if (lane == 0) {
            count[i] = ok ? cnt : 0;
            valid[i] = ok ? 1 : 0;
        }
    }
}

__global__ __launch_bounds__(256) void prep_emit_kernel(const double* __restrict__ mz, const float* __restrict__ intensity,
                                                        const int64_t* __restrict__ indptr, int64_t n, PrepParams p,
                                                        const uint8_t* __restrict__ flag, const int32_t* __restrict__ count,
                                                        const int64_t* __restrict__ out_indptr, float* __restrict__ raw,
                                                        float* __restrict__ out_mz, float* __restrict__ out_it) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int K = count[i];
        if (K == 0) continue;
        const int64_t b = indptr[i], e = indptr[i + 1], ob = out_indptr[i];
        // ---- compaction, m/z order kept ---------------------------------------------------------
        int base = 0;
        for (int64_t j0 = b; j0 < e; j0 += 64) {
            const int64_t j = j0 + lane;
            const bool k = j < e && flag[j];
            const uint64_t m = __ballot(k);
            if (k) {
                const int w = base + __popcll(m & ((1ull << lane) - 1ull));
                out_mz[ob + w] = (float)mz[j];
                raw[ob + w] = intensity[j];
            }
            base += __popcll(m);
        }
        __threadfence();
        // ---- scaling (spectrum.py:157) ----------------------------------------------------------
        const float* rw = raw + ob;
        float* oi = out_it + ob;
        const int max_rank = p.max_peaks > 0 ? p.max_peaks : K;
        for (int q = lane; q < K; q += 64) {
            const float x = rw[q];
            float v = x;
            if (p.scaling == 1) {
                v = (float)sqrt((double)x);
            } else if (p.scaling == 2) {
                v = (float)log2((double)(1.0f + x));
            } else if (p.scaling == 3) {
                int r = 1;                                    // rank in ascending order, ties by position
                for (int f = 0; f < K; ++f) {
                    const float y = rw[f];
                    r += (y < x || (y == x && f < q)) ? 1 : 0;
                }
                v = (float)(max_rank - (K - r));
            }
            oi[q] = v;
        }
        __threadfence();
        // ---- L2 norm in the fixed float64 tree (lane l owns elements 256p + 4l + c), float32 divide --------
        double acc = 0.0;
        for (int p0 = 0; p0 < K; p0 += 256) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int q = p0 + 4 * lane + c;
                const double x = q < K ? (double)oi[q] : 0.0;
                acc = acc + x * x;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        const float nrm = (float)sqrt(acc);
        for (int q = lane; q < K; q += 64) oi[q] = oi[q] / nrm;
    }
}

}  // namespace fal
FAL_WARM_KERNEL(fal::prep_flags_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

using namespace fal;

extern "C" int fal_process_spectra(fal_ctx* ctx, const double* mz, const float* intensity, const int64_t* indptr, int64_t n,
                                   int64_t nnz, const double* precursor_mz, const int32_t* precursor_charge, int min_peaks,
                                   double min_mz_range, double mz_min, double mz_max, double remove_precursor_tol,
                                   double min_intensity, int max_peaks_used, int scaling, int32_t* valid_out,
                                   int64_t* out_indptr, float* out_mz, float* out_intensity) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && n >= 0 && nnz >= 0, FAL_EINVAL, "fal_process_spectra: bad argument");
    FAL_REQUIRE(scaling >= 0 && scaling <= 3 && max_peaks_used >= 0 && min_peaks >= 0, FAL_EINVAL,
                "fal_process_spectra: bad option");
    FAL_REQUIRE(out_indptr, FAL_EINVAL, "fal_process_spectra: NULL out_indptr");
    if (n == 0) {
        FAL_CHECK_HIP(hipMemsetAsync(out_indptr, 0, sizeof(int64_t), ctx->stream));
        return FAL_OK;
    }
    FAL_REQUIRE(indptr && precursor_mz && precursor_charge && valid_out, FAL_EINVAL, "fal_process_spectra: NULL array");
    FAL_REQUIRE(nnz == 0 || (mz && intensity && out_mz && out_intensity), FAL_EINVAL, "fal_process_spectra: NULL peaks");
    PrepParams p{};
    p.min_peaks = min_peaks;
    p.min_mz_range = min_mz_range;
    p.has_min = !isnan(mz_min);
    p.has_max = !isnan(mz_max);
    p.mz_min = mz_min;
    p.mz_max = mz_max;
    p.rm_tol = remove_precursor_tol;
    p.filter_intensity = min_intensity >= 0.0 || max_peaks_used > 0;
    p.min_intensity = min_intensity >= 0.0 ? (float)min_intensity : 0.0f;
    p.max_peaks = max_peaks_used;
    p.scaling = scaling;
    uint8_t* flag = nullptr;
    int32_t* count = nullptr;
    float* raw = nullptr;
    FAL_TRY(ctx->reserve(SLOT_MISC, (size_t)nnz + 64, (void**)&flag));
    FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(int32_t) * (size_t)n, (void**)&count));
    FAL_TRY(ctx->reserve(SLOT_TAIL, sizeof(float) * ((size_t)nnz + 64), (void**)&raw));
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n, 4), (int64_t)ctx->num_cus * 64);
    hipLaunchKernelGGL(prep_flags_kernel, dim3(grid), dim3(256), 0, ctx->stream, mz, intensity, indptr, n, precursor_mz,
                       precursor_charge, p, flag, count, valid_out);
    FAL_CHECK_HIP(hipGetLastError());
    FAL_TRY(device_scan_i32(ctx, count, n, out_indptr, SLOT_SORT));
    hipLaunchKernelGGL(prep_emit_kernel, dim3(grid), dim3(256), 0, ctx->stream, mz, intensity, indptr, n, p, flag, count,
                       out_indptr, raw, out_mz, out_intensity);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}
