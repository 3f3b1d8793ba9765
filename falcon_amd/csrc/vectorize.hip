// a2 + a3: CSR peak lists -> dense feature-hashed, L2-normalised vectors.
//
// Reference: falcon/cluster/spectrum.py:250-296 (_to_vector: bin index) and 202-247
// (to_vector: projection + normalize_L2), hashing spec README.md:124-131.
//
// One 64-lane wave (= one workgroup) per spectrum, grid-stride.  HBM-bound:
// 8 B/peak in, low_dim*4 (or *2) B/spectrum out.
//   * lane p of a 64-peak chunk owns peak p: f64 bin index (bit-exact with the reference's
//     float64 arithmetic), MurmurHash3 in-register (exact integer ops, no table gather);
//   * colliding peaks are added in PEAK ORDER: each round every pending lane posts its lane id
//     with an LDS atomic-min on tag[h]; the lowest lane per hash bin wins, adds, and clears the
//     tag.  That makes the float32 sums independent of hardware atomic ordering and equal to
//     the oracle's sequential adds bit for bit;
//   * the sum of squares runs in float64 in a fixed tree (lane l owns elements 256p+4l+c, then
//     an xor butterfly), so the norm is reproducible too; scale = (float)(1.0 / sqrt(nr)) with the
//     sqrt and the divide in float64 (both correctly rounded on gfx950; the f32 v_sqrt is not);
//   * rows leave as 16 B/lane (f32) or 8 B/lane (f16) coalesced stores straight from LDS.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "hash.h"

namespace {

constexpr int kMaxPasses = FAL_MAX_LOW_DIM / 256;

// lane ^ X exchange of a 32-bit value: DPP quad_perm for X = 1, 2 (no LDS), ds_swizzle in bit-mask mode for X = 4, 8, 16 (no address
// register), ds_bpermute for 32 (select.h lane_xor)
template <int X>
__device__ __forceinline__ uint32_t vec_lane_xor(uint32_t v) {
    if (X == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    if (X == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    if (X == 4 || X == 8 || X == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (X << 10));
    return (uint32_t)__shfl_xor((int)v, X, 64);
}
template <int X>
__device__ __forceinline__ double vec_lane_xor_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = vec_lane_xor<X>((uint32_t)b), hi = vec_lane_xor<X>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the butterfly sum of the 64 lanes' doubles, levels 32, 16, 8, 4, 2, 1 (the oracle's tree: the order is part of the result).  The
// generic __shfl_xor cost five address instructions, two ds_bpermute and a full LDS round trip per level -- six dependent round trips
// per row in a kernel whose VALU instructions x 4 cycles are 0.67 of its SIMD cycles (round 6 counters)
__device__ __forceinline__ double wave_xor_sum(double v) {
    v += vec_lane_xor_f64<32>(v);
    v += vec_lane_xor_f64<16>(v);
    v += vec_lane_xor_f64<8>(v);
    v += vec_lane_xor_f64<4>(v);
    v += vec_lane_xor_f64<2>(v);
    v += vec_lane_xor_f64<1>(v);
    return v;
}

__device__ __forceinline__ bool bin_of(float m, double min_mz, double bin_size, uint32_t n_bins, int32_t* bin) {
    // spectrum.py:291: floor((mz - min_mz) / bin_size) with mz promoted to float64
    double q = floor(__ddiv_rn((double)m - min_mz, bin_size));
    if (q >= 0.0 && q < (double)n_bins) {
        *bin = (int32_t)q;
        return true;
    }
    return false;
}

template <int OUT_F16>
__global__ __launch_bounds__(64) void vectorize_kernel(
    const float* __restrict__ mz, const float* __restrict__ inten, const int64_t* __restrict__ indptr,
    const int64_t* __restrict__ row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
    uint32_t d, uint32_t hash_mod, uint32_t seed, int normalize, void* __restrict__ out, void* __restrict__ out2, int chunk) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t passes = (d + 255) / 256;
    float* acc = reinterpret_cast<float*>(smem);                 // passes*256 floats
    uint32_t* tag = reinterpret_cast<uint32_t*>(acc + passes * 256);  // d entries
    const uint32_t lane = threadIdx.x;

    for (uint32_t i = lane; i < passes * 256; i += 64) acc[i] = 0.f;
    for (uint32_t i = lane; i < d; i += 64) tag[i] = 0xFFFFFFFFu;
    __syncthreads();

    // A wave takes `chunk` CONSECUTIVE output rows at a time: their row_order / indptr entries come by one coalesced load and one
    // gather for the whole chunk (lane i = row i of the chunk) instead of two dependent round trips per row, and the first 64
    // peaks of row i + 1 are in flight while row i is hashed and written -- per row the kernel used to wait for a chain of three
    // dependent loads (order -> indptr -> peaks), which, not the bytes, set its time (0.45 of the HBM roof).
    for (int64_t c0 = (int64_t)blockIdx.x * chunk; c0 < n; c0 += (int64_t)gridDim.x * chunk) {
    const int n_rows = (int)min<int64_t>(chunk, n - c0);
    int64_t beg_l = 0, end_l = 0;
    if ((int)lane < n_rows) {
        const int64_t s = row_order ? row_order[c0 + lane] : c0 + lane;
        beg_l = indptr[s];
        end_l = indptr[s + 1];
    }
    auto row_range = [&](int i, int64_t* b, int64_t* e) {
        *b = ((int64_t)__builtin_amdgcn_readlane((int)(beg_l >> 32), i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)beg_l, i);
        *e = ((int64_t)__builtin_amdgcn_readlane((int)(end_l >> 32), i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)end_l, i);
    };
    float mz_n = 0.f, x_n = 0.f;                                 // the next row's first 64 peaks (lane = peak)
    {
        int64_t b0, e0;
        row_range(0, &b0, &e0);
        if (b0 + lane < e0) {
            mz_n = mz[b0 + lane];
            x_n = inten[b0 + lane];
        }
    }
    for (int ri = 0; ri < n_rows; ++ri) {
        const int64_t r = c0 + ri;
        int64_t beg, end;
        row_range(ri, &beg, &end);
        const float mz_c = mz_n, x_c = x_n;
        if (ri + 1 < n_rows) {                                   // (uniform)
            int64_t b1, e1;
            row_range(ri + 1, &b1, &e1);
            mz_n = 0.f;
            x_n = 0.f;
            if (b1 + lane < e1) {
                mz_n = mz[b1 + lane];
                x_n = inten[b1 + lane];
            }
        }
        for (int64_t p0 = beg; p0 < end; p0 += 64) {
            const int64_t p = p0 + lane;
            bool pending = false;
            uint32_t h = 0;
            float x = 0.f;
            if (p < end) {
                int32_t b;
                const bool first = p0 == beg;                    // (uniform) the prefetched peaks; later chunks of a long spectrum: loaded here
                x = first ? x_c : inten[p];
                if (bin_of(first ? mz_c : mz[p], min_mz, bin_size, n_bins, &b)) {
                    pending = true;
                    h = fal::murmur3_32((uint32_t)b, seed) % hash_mod;      // (columns hash_mod .. d - 1 of a padded row stay zero)
                }
            }
            while (__any(pending)) {
                if (pending) atomicMin(&tag[h], lane);
                __syncthreads();
                const bool win = pending && tag[h] == lane;
                __syncthreads();
                if (win) {
                    acc[h] += x;
                    tag[h] = 0xFFFFFFFFu;
                    pending = false;
                }
                __syncthreads();
            }
        }
        // ---- norm (fixed-order float64 tree) and write-out ------------------------------
        float4 v[kMaxPasses];
        double part = 0.0;
#pragma unroll
        for (int p = 0; p < kMaxPasses; ++p) {
            if ((uint32_t)p < passes) {
                v[p] = *reinterpret_cast<const float4*>(&acc[256 * p + 4 * lane]);
                *reinterpret_cast<float4*>(&acc[256 * p + 4 * lane]) = make_float4(0.f, 0.f, 0.f, 0.f);
                part += (double)v[p].x * (double)v[p].x;
                part += (double)v[p].y * (double)v[p].y;
                part += (double)v[p].z * (double)v[p].z;
                part += (double)v[p].w * (double)v[p].w;
            }
        }
        float inv = 1.f;
        if (normalize) {
            const double nr = wave_xor_sum(part);
            inv = nr > 0.0 ? (float)__ddiv_rn(1.0, __dsqrt_rn(nr)) : 0.f;
        }
#pragma unroll
        for (int p = 0; p < kMaxPasses; ++p) {
            const uint32_t e = 256 * p + 4 * lane;
            if ((uint32_t)p < passes && e < d) {
                float4 o = make_float4(v[p].x * inv, v[p].y * inv, v[p].z * inv, v[p].w * inv);
                if (OUT_F16 == 2) {
                    // hi/lo split of the float32 value: x ~= hi + lo / 2048 to ~2^-22 relative
                    const __half h0 = __float2half_rn(o.x), h1 = __float2half_rn(o.y), h2 = __float2half_rn(o.z),
                                 h3 = __float2half_rn(o.w);
                    const __half2 ha = __halves2half2(h0, h1), hb = __halves2half2(h2, h3);
                    const __half2 la = __floats2half2_rn((o.x - __half2float(h0)) * 2048.f, (o.y - __half2float(h1)) * 2048.f);
                    const __half2 lb = __floats2half2_rn((o.z - __half2float(h2)) * 2048.f, (o.w - __half2float(h3)) * 2048.f);
                    uint2 ph, pl;
                    ph.x = *reinterpret_cast<const uint32_t*>(&ha);
                    ph.y = *reinterpret_cast<const uint32_t*>(&hb);
                    pl.x = *reinterpret_cast<const uint32_t*>(&la);
                    pl.y = *reinterpret_cast<const uint32_t*>(&lb);
                    __half* rowp = reinterpret_cast<__half*>(out) + r * (int64_t)(2 * d);
                    *reinterpret_cast<uint2*>(rowp + e) = ph;
                    *reinterpret_cast<uint2*>(rowp + d + e) = pl;
                } else if (OUT_F16) {
                    __half2 a = __floats2half2_rn(o.x, o.y), b = __floats2half2_rn(o.z, o.w);
                    uint2 pk;
                    pk.x = *reinterpret_cast<uint32_t*>(&a);
                    pk.y = *reinterpret_cast<uint32_t*>(&b);
                    // OUT_F16 == 3: the float32 row to `out` AND its float16 rounding to `out2` (one pass over the peaks);
                    // OUT_F16 == 4: float16 vectors (BASELINE config 5): the rounding to `out2`, and to `out` the float32 IMAGE
                    // of the rounded values (what the exact float32 kernels of the index work on)
                    *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(OUT_F16 >= 3 ? out2 : out) + r * (int64_t)d + e) = pk;
                    if (OUT_F16 == 4) o = make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
                    if (OUT_F16 >= 3) *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + r * (int64_t)d + e) = o;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + r * (int64_t)d + e) = o;
                }
            }
        }
        __syncthreads();
    }
    }
}

__global__ void to_vector_indices_kernel(const float* __restrict__ mz, int64_t nnz, double min_mz,
                                         double bin_size, int32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (int32_t)floor(__ddiv_rn((double)mz[i] - min_mz, bin_size));
}

FAL_WARM_KERNEL(vectorize_kernel<0>);
}  // namespace

extern "C" {

int fal_to_vector_indices(fal_ctx* ctx, const float* mz, int64_t nnz, double min_mz, double bin_size,
                          int32_t* out_indices) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && (nnz == 0 || (mz && out_indices)) && nnz >= 0 && bin_size > 0, FAL_EINVAL,
                "fal_to_vector_indices: bad argument");
    if (nnz == 0) return FAL_OK;
    int grid = (int)std::min<int64_t>(fal::ceil_div(nnz, 256), (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(to_vector_indices_kernel, dim3(grid), dim3(256), 0, ctx->stream, mz, nnz, min_mz, bin_size,
                       out_indices);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

static int vectorize_impl(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                          const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                          uint32_t hash_mod, uint32_t row_width, uint32_t seed, int normalize, int out_dtype, void* out, void* out2);

int fal_vectorize(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                  const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                  uint32_t low_dim, uint32_t seed, int normalize, int out_dtype, void* out) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(out_dtype == FAL_DTYPE_F32 || out_dtype == FAL_DTYPE_F16 || out_dtype == FAL_DTYPE_SPLIT16, FAL_EINVAL,
                "fal_vectorize: bad out_dtype");
    return vectorize_impl(ctx, mz, intensity, indptr, row_order, n, min_mz, bin_size, n_bins, low_dim, low_dim, seed, normalize, out_dtype,
                          out, nullptr);
}

int fal_vectorize_pair(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                       const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                       uint32_t low_dim, uint32_t seed, int normalize, float* out_f32, void* out_f16) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(n == 0 || out_f16, FAL_EINVAL, "fal_vectorize_pair: NULL array");
    return vectorize_impl(ctx, mz, intensity, indptr, row_order, n, min_mz, bin_size, n_bins, low_dim, low_dim, seed, normalize, -3,
                          out_f32, out_f16);
}

int fal_vectorize_f16_image(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                            const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                            uint32_t low_dim, uint32_t seed, int normalize, float* out_f32_image, void* out_f16) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(n == 0 || (out_f16 && out_f32_image), FAL_EINVAL, "fal_vectorize_f16_image: NULL array");
    return vectorize_impl(ctx, mz, intensity, indptr, row_order, n, min_mz, bin_size, n_bins, low_dim, low_dim, seed, normalize, -4,
                          out_f32_image, out_f16);
}

int fal_row_width(uint32_t low_dim, uint32_t* row_width) {
    FAL_REQUIRE(row_width, FAL_EINVAL, "fal_row_width: NULL");
    static const uint32_t widths[] = {64, 128, 256, 400, 800};
    for (uint32_t w : widths)
        if (low_dim >= 1 && low_dim <= w) {
            *row_width = w;
            return FAL_OK;
        }
    fal::set_error("low_dim must be in [1, 800] (got %u): the widest instantiation of the cosine kernels holds 800 columns", low_dim);
    return FAL_EUNSUPPORTED;
}

int fal_vectorize_rows(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr, const int64_t* row_order,
                       int64_t n, double min_mz, double bin_size, uint32_t n_bins, uint32_t low_dim, uint32_t row_width,
                       uint32_t seed, int normalize, int out_mode, void* out, void* out2) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(out_mode >= FAL_DTYPE_F32 && out_mode <= FAL_OUT_F16_IMAGE, FAL_EINVAL, "fal_vectorize_rows: bad out_mode");
    FAL_REQUIRE(n == 0 || out_mode < FAL_OUT_F32_F16 || out2, FAL_EINVAL, "fal_vectorize_rows: NULL second output");
    const int code = out_mode == FAL_OUT_F32_F16 ? -3 : out_mode == FAL_OUT_F16_IMAGE ? -4 : out_mode;
    return vectorize_impl(ctx, mz, intensity, indptr, row_order, n, min_mz, bin_size, n_bins, low_dim, row_width, seed, normalize, code,
                          out, out2);
}

// low_dim = the hash modulus (README.md:114-117: any positive integer); row_width = columns of an output row, a multiple of 8
// >= low_dim: the columns behind low_dim are zero
static int vectorize_impl(fal_ctx* ctx, const float* mz, const float* intensity, const int64_t* indptr,
                          const int64_t* row_order, int64_t n, double min_mz, double bin_size, uint32_t n_bins,
                          uint32_t hash_mod, uint32_t row_width, uint32_t seed, int normalize, int out_dtype, void* out, void* out2) {
    FAL_REQUIRE(ctx, FAL_EINVAL, "fal_vectorize: NULL ctx");
    FAL_REQUIRE(n >= 0 && bin_size > 0 && n_bins > 0, FAL_EINVAL, "fal_vectorize: bad sizes");
    FAL_REQUIRE(row_width >= 8 && row_width <= FAL_MAX_LOW_DIM && row_width % 8 == 0, FAL_EUNSUPPORTED,
                "fal_vectorize: the row width must be a multiple of 8 in [8, %d] (got %u)", FAL_MAX_LOW_DIM, row_width);
    FAL_REQUIRE(hash_mod >= 1 && hash_mod <= row_width, FAL_EINVAL, "fal_vectorize: low_dim %u must be in [1, row width %u]", hash_mod,
                row_width);
    if (n == 0) return FAL_OK;
    FAL_REQUIRE(indptr && out, FAL_EINVAL, "fal_vectorize: NULL array");
    ctx->stage_reset(fal::ST_VECTORIZE);
    const uint32_t passes = (row_width + 255) / 256;
    const size_t lds = (size_t)(passes * 256 + row_width) * 4;
    // rows per wave and turn: 16 at most (measured: 4..16 equal, 64 loses the balance again), fewer while the waves (32 per CU)
    // would get less than four turns each
    int chunk = 16;
    while (chunk > 1 && n / chunk < (int64_t)ctx->num_cus * 32 * 4) chunk >>= 1;
    const int grid = (int)std::min<int64_t>(fal::ceil_div(n, chunk), (int64_t)ctx->num_cus * 32);
    {
        fal::StageScope t(ctx, fal::ST_VECTORIZE);
        if (out_dtype == -4)
            hipLaunchKernelGGL(vectorize_kernel<4>, dim3(grid), dim3(64), lds, ctx->stream, mz, intensity, indptr,
                               row_order, n, min_mz, bin_size, n_bins, row_width, hash_mod, seed, normalize, out, out2, chunk);
        else if (out_dtype == -3)
            hipLaunchKernelGGL(vectorize_kernel<3>, dim3(grid), dim3(64), lds, ctx->stream, mz, intensity, indptr,
                               row_order, n, min_mz, bin_size, n_bins, row_width, hash_mod, seed, normalize, out, out2, chunk);
        else if (out_dtype == FAL_DTYPE_SPLIT16)
            hipLaunchKernelGGL(vectorize_kernel<2>, dim3(grid), dim3(64), lds, ctx->stream, mz, intensity, indptr,
                               row_order, n, min_mz, bin_size, n_bins, row_width, hash_mod, seed, normalize, out, out2, chunk);
        else if (out_dtype == FAL_DTYPE_F16)
            hipLaunchKernelGGL(vectorize_kernel<1>, dim3(grid), dim3(64), lds, ctx->stream, mz, intensity, indptr,
                               row_order, n, min_mz, bin_size, n_bins, row_width, hash_mod, seed, normalize, out, out2, chunk);
        else
            hipLaunchKernelGGL(vectorize_kernel<0>, dim3(grid), dim3(64), lds, ctx->stream, mz, intensity, indptr,
                               row_order, n, min_mz, bin_size, n_bins, row_width, hash_mod, seed, normalize, out, out2, chunk);
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // extern "C"
