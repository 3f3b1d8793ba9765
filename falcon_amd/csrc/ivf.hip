// a6 IVF build (deterministic k-means + inverted lists) and a7 search orchestration.
//
// Reference: README.md:132-142 is the only statement of this stage in the snapshot (Faiss
// IndexIVFFlat over inner product, un-vendored dependency setup.cfg:25); the conventions are the
// build's own and are restated on the CPU in oracle/falcon_oracle.py (ivf_train / ivf_build /
// ivf_search).  Everything is batched over ALL buckets of a charge partition: one launch per
// step, never one launch per bucket.
#include <algorithm>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "ivf16.h"
#include "util.h"

namespace fal {

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
struct BucketDev {       // one per IVF (n_list > 1) bucket
    int64_t row0;        // first sorted row
    int64_t list0;       // global id of its list 0
    int32_t n;           // rows
    int32_t n_list;
    int64_t wave0;       // index of its first (bucket, list) wave in per-list launches
};

__device__ __forceinline__ int find_bucket(const BucketDev* __restrict__ b, int nb, int64_t w) {
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (b[mid].wave0 <= w) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// init: centroid i of a bucket = row floor(i * n / n_list)   (one wave per centroid)
__global__ __launch_bounds__(64) void kmeans_init_kernel(const float* __restrict__ X, int d,
                                                         const BucketDev* __restrict__ bk, int nb,
                                                         float* __restrict__ C) {
    const BucketDev b = bk[find_bucket(bk, nb, blockIdx.x)];
    const int i = (int)(blockIdx.x - b.wave0);
    const int64_t src = b.row0 + ((int64_t)i * b.n) / b.n_list;
    const float4* s = reinterpret_cast<const float4*>(X + src * d);
    float4* o = reinterpret_cast<float4*>(C + (b.list0 + i) * d);
    for (int e = threadIdx.x; e < d / 4; e += 64) o[e] = s[e];
}

__device__ __forceinline__ double wave_xor_sum_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// k-means update from the SPARSE form of the rows.  A vectorised spectrum has at most as many non-zero components as peaks
// (~50 of 400), and adding a zero never changes a float32 sum (the accumulators start at +0.0), so summing a list's members
// over their non-zero components only -- in the same row order -- gives the same centroid bit for bit while reading 384 B per
// member instead of 4 d.  The members of a list come from a stable sort of the rows by list (row order inside a list).
// ------------------------------------------------------------------------------------------

// rows of the IVF buckets -> (column, value) entries; blocks of 256 rows, seg_off = first block of every bucket
__global__ __launch_bounds__(256) void sparsify_rows_kernel(const float* __restrict__ X, int d, const BucketDev* __restrict__ bk,
                                                            const int64_t* __restrict__ seg_off, int nb,
                                                            uint16_t* __restrict__ cols, float* __restrict__ vals,
                                                            uint16_t* __restrict__ sq16, int32_t* __restrict__ neg_flag) {
    __shared__ uint16_t sc[4][kSparseW];
    __shared__ float sv[4][kSparseW];
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_off[mid] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const BucketDev b = bk[lo];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r_first = (int)(blockIdx.x - seg_off[lo]) * 256;
    const int r_end = min(b.n, r_first + 256);
    bool neg = false;            // a component that is negative or not finite: the float16 prefilters' error bound needs rows >= 0
    bool many = false;           // a row with more than kSparseW non-zero components (kept dense: list16s_kernel cannot take the index)
    bool wide = false;           // a component that is not a float16 value (float16 VECTORS, config 5: none -- pairs16.hip then keeps
                                 // its query tile in LDS as float16, exactly)
    for (int rl = r_first + w; rl < r_end; rl += 4) {
        const int64_t r = b.row0 + rl;
        sc[w][lane] = kColPad;
        sv[w][lane] = 0.f;
        // entries in the order of the exact similarity chains (column 0, d/2, 1, d/2 + 1, ...: dense.hip / pairs16.hip), so that
        // a chain over a row's entries visits its non-zero terms in the order the dense chain does
        const float4* row = reinterpret_cast<const float4*>(X + r * d);
        const int dh4 = d >> 3;
        int base = 0;
        for (int e0 = 0; e0 < dh4; e0 += 64) {
            const int e = e0 + lane;
            const float4 lo = e < dh4 ? row[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 hi = e < dh4 ? row[dh4 + e] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float xs[8] = {lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, lo.w, hi.w};
            int c = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c += (int)(xs[j] != 0.f);
                neg |= !(xs[j] >= 0.f) || xs[j] > 65504.f;
                wide |= __half2float(__float2half_rn(xs[j])) != xs[j];
            }
            const int pre = wave_prefix_sum(c);                  // (DPP: the generic shuffles were six LDS round trips per row)
            const int tot = __builtin_amdgcn_readlane(pre, 63);
            if (base + tot <= kSparseW) {
                int pos = base + pre - c;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (xs[j] != 0.f) {
                        sc[w][pos] = (uint16_t)((j & 1) * (d >> 1) + 4 * e + (j >> 1));
                        sv[w][pos] = xs[j];
                        ++pos;
                    }
            }
            base += tot;
        }
        if (base > kSparseW && lane == 0) sc[w][0] = kColDense;
        many |= base > kSparseW;
        cols[r * kSparseW + lane] = sc[w][lane];
        vals[r * kSparseW + lane] = sv[w][lane];
        if (sq16) {
            // the row as the query record of list16s_kernel: 64 columns, then the 64 values rounded to float16 (the float16 row's
            // own components: vectorize.hip rounds the same float32 values the same way) -- 256 contiguous bytes
            sq16[r * (2 * kSparseW) + lane] = sc[w][lane];
            sq16[r * (2 * kSparseW) + kSparseW + lane] = __half_as_ushort(__float2half_rn(sv[w][lane]));
        }
    }
    if (__ballot(neg) != 0ull && lane == 0) atomicOr(neg_flag, 1);
    if (__ballot(wide) != 0ull && lane == 0) atomicOr(neg_flag + 1, 1);
    if (__ballot(many) != 0ull && lane == 0) atomicOr(neg_flag + 2, 1);
}

// global list of every sorted row (bucket by binary search on the bucket table) = the key of the stable sort by list
__global__ void list_key_kernel(const int32_t* __restrict__ assign, const int64_t* __restrict__ bucket_off,
                                const int64_t* __restrict__ list_base, int n_buckets, int64_t n, uint32_t* __restrict__ keys) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_buckets - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (bucket_off[mid] <= i) lo = mid; else hi = mid - 1;
        }
        keys[i] = (uint32_t)(list_base[lo] + assign[i]);
    }
}

// list_off[L] = first position of list L in the sorted keys (lower bound), list_off[total] = n
__global__ void list_bounds_kernel(const uint32_t* __restrict__ keys_sorted, int64_t n, int64_t total, int64_t* __restrict__ list_off) {
    for (int64_t L = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; L <= total; L += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys_sorted[mid] < L) lo = mid + 1; else hi = mid;
        }
        list_off[L] = lo;
    }
}

// Stable sort of an IVF bucket's rows by list, one wave per bucket: what the k-means passes need after every assignment (the
// members of a list in row order).  A global radix sort of all n (list, row) pairs -- rocPRIM, 3 + 2 kernels per call, 11 calls
// per build -- does the same; here a bucket's rows are counted (LDS histogram of its <= kBucketSortMaxLists lists), the list
// offsets follow from a wave scan, and the rows are placed chunk by chunk IN ORDER: lanes with the same list find each other
// with one ballot per list-id bit, rank = earlier peers in the chunk, base = the list's running cursor in LDS.  Identical output
// (perm, list_off of the bucket's lists); flat buckets keep what the first, global sort of the build wrote.
constexpr int kBucketSortMaxLists = 2048;

__global__ __launch_bounds__(64) void bucket_list_sort_kernel(const int32_t* __restrict__ assign, const BucketDev* __restrict__ bk, int nb,
                                                              int32_t* __restrict__ perm, int64_t* __restrict__ list_off) {
    __shared__ uint32_t cur[kBucketSortMaxLists];
    if ((int)blockIdx.x >= nb) return;
    const BucketDev b = bk[blockIdx.x];
    const int lane = threadIdx.x;
    const int nl = b.n_list;
    for (int l = lane; l < nl; l += 64) cur[l] = 0u;
    wave_lds_sync();
    constexpr int U = 8;                                    // chunks whose ids are fetched together (the loop is one wave's
    const int32_t* ids = assign + b.row0;                   //  chain of dependent steps: latency, not bytes)
    for (int i0 = 0; i0 < b.n; i0 += 64 * U) {
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 64 * u + lane;
            id[u] = i < b.n ? ids[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (id[u] >= 0) atomicAdd(&cur[id[u]], 1u);
    }
    wave_lds_sync();
    // exclusive prefix over the lists (64 at a time): cur[l] = first position of list l inside the bucket
    uint32_t carry = 0u;
    for (int l0 = 0; l0 < nl; l0 += 64) {
        const int l = l0 + lane;
        const uint32_t c = l < nl ? cur[l] : 0u;
        uint32_t pre = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(pre, off, 64);
            if (lane >= off) pre += y;
        }
        const uint32_t tot = __shfl(pre, 63, 64);
        if (l < nl) {
            cur[l] = carry + pre - c;
            list_off[b.list0 + l] = b.row0 + (int64_t)(carry + pre - c);
        }
        carry += tot;
    }
    wave_lds_sync();
    int bits = 1;
    while ((1 << bits) < nl) ++bits;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int i0 = 0; i0 < b.n; i0 += 64 * U) {
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 64 * u + lane;
            id[u] = i < b.n ? ids[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = id[u] >= 0;
            if (__ballot(live) == 0ull) break;
            unsigned long long peers = __ballot(live);
            for (int t = 0; t < bits; ++t) {
                const unsigned long long m = __ballot(live && ((id[u] >> t) & 1));
                peers &= ((id[u] >> t) & 1) ? m : ~m;
            }
            const uint32_t base = live ? cur[id[u]] : 0u;                 // every peer reads the cursor before ...
            wave_lds_sync();
            if (live && (peers & lt) == 0ull) cur[id[u]] = base + (uint32_t)__popcll(peers);      // ... the first of them advances it
            wave_lds_sync();
            if (live) perm[b.row0 + base + __popcll(peers & lt)] = (int32_t)(b.row0 + i0 + 64 * u + lane);
        }
    }
}

// update: one wave per (bucket, list); the members (sorted rows, in row order) are added one after the other into the wave's
// LDS accumulators, eight members' entries in flight.  Normalisation: the fixed-order float64 tree of the vectorise kernel.  Empty lists keep their centroid.
__global__ __launch_bounds__(64) void centroid_update_kernel(const uint16_t* __restrict__ cols, const float* __restrict__ vals,
                                                             const float* __restrict__ X, int d,
                                                             const int32_t* __restrict__ rows_sorted,
                                                             const int64_t* __restrict__ list_off,
                                                             const BucketDev* __restrict__ bk, int nb, float* __restrict__ C,
                                                             __half* __restrict__ C16) {
    __shared__ float acc[FAL_MAX_LOW_DIM];
    const BucketDev b = bk[find_bucket(bk, nb, blockIdx.x)];
    const int li = (int)(blockIdx.x - b.wave0);
    const int lane = threadIdx.x;
    const int64_t o0 = list_off[b.list0 + li];
    const int cnt = (int)(list_off[b.list0 + li + 1] - o0);
    if (cnt == 0) return;   // keep the previous centroid
    for (int c = lane; c < d; c += 64) acc[c] = 0.f;
    constexpr int G = 8;
    for (int base = 0; base < cnt; base += 64) {
        const int m_in = min(64, cnt - base);
        const int my = lane < m_in ? rows_sorted[o0 + base + lane] : 0;
        for (int g = 0; g < m_in; g += G) {
            int c[G];
            float v[G];
            int64_t m[G];
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const bool ok = g + t < m_in;
                m[t] = __shfl(my, ok ? g + t : 0, 64);
                c[t] = ok ? (int)cols[m[t] * kSparseW + lane] : (int)kColPad;
                v[t] = ok ? vals[m[t] * kSparseW + lane] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < G; ++t) {
                if (g + t >= m_in) break;
                if (__shfl(c[t], 0, 64) == (int)kColDense) {
                    const float* row = X + m[t] * d;
                    for (int e = lane; e < d; e += 64) acc[e] += row[e];
                } else if (c[t] != (int)kColPad) {
                    acc[c[t]] += v[t];      // LDS operations of a wave execute in order: member after member
                }
            }
        }
    }
    constexpr int P = FAL_MAX_LOW_DIM / 256;
    const int passes = (d + 255) / 256;
    float4 r[P];
    double part = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int e = 64 * p + lane;
        r[p] = (p < passes && e < d / 4) ? make_float4(acc[4 * e], acc[4 * e + 1], acc[4 * e + 2], acc[4 * e + 3])
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < passes) {
            part += (double)r[p].x * (double)r[p].x;
            part += (double)r[p].y * (double)r[p].y;
            part += (double)r[p].z * (double)r[p].z;
            part += (double)r[p].w * (double)r[p].w;
        }
    }
    const double nr = wave_xor_sum_d(part);
    const float inv = nr > 0.0 ? (float)__ddiv_rn(1.0, __dsqrt_rn(nr)) : 0.f;
    float4* o = reinterpret_cast<float4*>(C + (b.list0 + li) * d);
    uint2* o16 = C16 ? reinterpret_cast<uint2*>(C16 + (b.list0 + li) * d) : nullptr;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int e = 64 * p + lane;
        if (p < passes && e < d / 4) {
            const float4 v = make_float4(r[p].x * inv, r[p].y * inv, r[p].z * inv, r[p].w * inv);
            o[e] = v;
            if (o16) {                           // the float16 copy the next assignment pass reads (cvt_f16_kernel's rounding)
                const __half2 a = __floats2half2_rn(v.x, v.y), bb = __floats2half2_rn(v.z, v.w);
                uint2 pk;
                pk.x = *reinterpret_cast<const uint32_t*>(&a);
                pk.y = *reinterpret_cast<const uint32_t*>(&bb);
                o16[e] = pk;
            }
        }
    }
}

// exclusive scan of int64 counts -> offsets (single workgroup; n up to a few hundred thousand)
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int64_t* __restrict__ in, int64_t n,
                                                              int64_t* __restrict__ out) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n ? in[i] : 0;
        int64_t x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int64_t pre = carry_s;
        for (int j = 0; j < w; ++j) pre += wsum[j];
        if (i < n) out[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

__global__ void iota_i32_kernel(int32_t* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (int32_t)i;
}

// Xl[p] = X[perm[p]]   (one wave per row, 16 B per lane)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ X, const int32_t* __restrict__ perm,
                                                          int64_t n, int d, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t p = blockIdx.x * 4ll + (threadIdx.x >> 6); p < n; p += (int64_t)gridDim.x * 4) {
        const float4* s = reinterpret_cast<const float4*>(X + (int64_t)perm[p] * d);
        float4* o = reinterpret_cast<float4*>(out + p * d);
        for (int e = lane; e < d / 4; e += 64) o[e] = s[e];
    }
}

int ivf_ensure_xl(fal_ctx* ctx, const fal_ivf* civf) {
    fal_ivf* ivf = const_cast<fal_ivf*>(civf);              // (a cache inside the index handle)
    if (ivf->Xl != nullptr || ivf->n == 0) return FAL_OK;
    FAL_REQUIRE(ivf->X != nullptr, FAL_EINVAL, "the index has no float32 rows");
    FAL_TRY(ctx->pool_alloc(sizeof(float) * (size_t)ivf->n * ivf->d, (void**)&ivf->Xl_owned));
    StageScope ts(ctx, ST_SCAN);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(ivf->n, 4), ctx->num_cus * 16)), dim3(256), 0,
                       ctx->stream, ivf->X, ivf->perm, ivf->n, ivf->d, ivf->Xl_owned);
    FAL_CHECK_HIP(hipGetLastError());
    ivf->Xl = ivf->Xl_owned;
    return FAL_OK;
}

int launch_exclusive_scan(fal_ctx* ctx, const int64_t* in, int64_t n, int64_t* out) {
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, in, n, out);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal

using namespace fal;

// ------------------------------------------------------------------------------------------
// build
// ------------------------------------------------------------------------------------------

namespace fal {
// any float16 component that is negative (not -0), infinite or NaN -> *flag = 1   (8 halves per lane and step)
__global__ void rows_sign16_kernel(const uint4* __restrict__ x, int64_t n8, int32_t* __restrict__ flag) {
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = x[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = w[j] & 0xFFFFu, hi = w[j] >> 16;
            bad |= lo > 0x8000u || (lo & 0x7FFFu) >= 0x7C00u || hi > 0x8000u || (hi & 0x7FFFu) >= 0x7C00u;
        }
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace fal
FAL_WARM_KERNEL(fal::kmeans_init_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)

extern "C" {

int fal_ivf_destroy(fal_ivf* ivf) {
    fal::CallScope _call(ivf ? ivf->ctx : nullptr);
    if (!ivf) return FAL_OK;
    void* ptrs[] = {ivf->Xl_owned, ivf->centroids, ivf->assign, ivf->perm, ivf->list_off, ivf->counts, ivf->bk_dev,
                    ivf->pos_of_row, ivf->ckeys, ivf->sp_cols, ivf->sp_vals, ivf->sq16, ivf->neg_dev};
    for (void* p : ptrs)
        if (p && ivf->ctx) ivf->ctx->pool_free(p);     // recycled in stream order, no device sync
    delete ivf;
    return FAL_OK;
}

int fal_ivf_attach_f16(fal_ivf* ivf, const void* X16, int planes) {
    fal::CallScope _call(ivf ? ivf->ctx : nullptr);
    FAL_REQUIRE(ivf && X16 && (planes == 1 || planes == 2), FAL_EINVAL, "fal_ivf_attach_f16: bad argument");
    {
        const int d = ivf->d;
        const bool ok = planes == 2 ? (d == 64 || d == 128 || d == 256 || d == 400)
                                    : (d == 64 || d == 128 || d == 256 || d == 400 || d == 512 || d == 800);
        FAL_REQUIRE(ok, FAL_EUNSUPPORTED, "fal_ivf_attach_f16: low_dim %d with %d plane(s) is not instantiated "
                    "(planes 1: 64/128/256/400/512/800, planes 2: 64/128/256/400)", d, planes);
    }
    ivf->X16 = X16;
    ivf->x16_planes = planes;
    return FAL_OK;
}

int fal_ivf_attach_prefilter(fal_ivf* ivf, const void* X16) { return fal_ivf_attach_prefilter_ex(ivf, X16, 1); }

int fal_ivf_attach_prefilter_ex(fal_ivf* ivf, const void* X16, int which) {
    fal::CallScope _call(ivf ? ivf->ctx : nullptr);
    FAL_REQUIRE(ivf && X16, FAL_EINVAL, "fal_ivf_attach_prefilter: NULL argument");
    FAL_REQUIRE(ivf->X, FAL_EINVAL, "fal_ivf_attach_prefilter: the index has no float32 rows to refine with");
    FAL_REQUIRE(which >= 1 && which <= 3, FAL_EINVAL, "fal_ivf_attach_prefilter_ex: which must be 1 (flat buckets), 2 (IVF buckets) or 3");
    fal_ctx* ctx = ivf->ctx;
    if ((which & 1) && ivf->n > 0) {
        // the precondition of the prefilter (header): no negative / non-finite component.  One pass over the float16 rows
        // (2 d bytes per row) and one synchronisation; rows that fail it leave the flat buckets to the exact staged scan.
        int32_t* flag = nullptr;
        FAL_TRY(ctx->reserve(SLOT_MISC, sizeof(int64_t), (void**)&flag));
        FAL_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int64_t), ctx->stream));
        const int64_t n8 = ivf->n * (int64_t)ivf->d / 8;
        hipLaunchKernelGGL(rows_sign16_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n8, 256), (int64_t)ctx->num_cus * 32)), dim3(256),
                           0, ctx->stream, reinterpret_cast<const uint4*>(X16), n8, flag);
        FAL_CHECK_HIP(hipGetLastError());
        FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host + 5, flag, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ivf->Xpre = ctx->fb_host[5] == 0 ? X16 : nullptr;
        if (!ivf->Xpre) ctx->counters[6] |= 2;
    }
    if ((which & 2) && ivf->n_ivf_buckets > 0 && ivf->rows_signed < 0) {
        FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host + 6, ivf->neg_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ivf->rows_signed = ctx->fb_host[6] != 0 ? 1 : 0;
        ivf->rows_f16 = ctx->fb_host[7] == 0 ? 1 : 0;
        ctx->counters[6] |= ivf->rows_signed;
    }
    if ((which & 2) && ivf->n_ivf_buckets > 0 && !ivf->X16pre && ivf->rows_signed == 0) {
        FAL_REQUIRE(ivf16_supports(ivf->d), FAL_EUNSUPPORTED,
                    "fal_ivf_attach_prefilter_ex: the IVF prefilter is instantiated for low_dim 64, 128, 256, 400 (got %d)", ivf->d);
        // (round 2 made a list-order copy of the float16 rows here: 16 GB of traffic at 10 M spectra for rows the scan gathers
        // at random inside a bucket either way; now the scan reads X16[perm[position]] and only the inverse of perm is made)
        FAL_TRY(ctx->pool_alloc(sizeof(int32_t) * (size_t)ivf->n, (void**)&ivf->pos_of_row));
        FAL_TRY(launch_pos_of_row(ctx, ivf->perm, ivf->n, ivf->pos_of_row));
        ivf->X16pre = X16;
    }
    return FAL_OK;
}

int fal_ivf_total_lists(const fal_ivf* ivf, int64_t* total_lists) {
    FAL_REQUIRE(ivf && total_lists, FAL_EINVAL, "fal_ivf_total_lists: NULL");
    *total_lists = ivf->total_lists;
    return FAL_OK;
}

int fal_ivf_build(fal_ctx* ctx, const float* X, int64_t n, int low_dim, const int64_t* bucket_off,
                  int64_t n_buckets, const int32_t* n_list, int kmeans_iters, fal_ivf** out) {
    fal::CallScope _call(ctx);
    return fal_ivf_build_x16(ctx, X, nullptr, n, low_dim, bucket_off, n_buckets, n_list, kmeans_iters, out);
}

int fal_ivf_build_x16(fal_ctx* ctx, const float* X, const void* X16, int64_t n, int low_dim, const int64_t* bucket_off,
                      int64_t n_buckets, const int32_t* n_list, int kmeans_iters, fal_ivf** out) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && out, FAL_EINVAL, "fal_ivf_build: NULL ctx/out");
    *out = nullptr;
    FAL_REQUIRE(n >= 0 && n < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "fal_ivf_build: n must be < 2^31 per partition");
    FAL_REQUIRE(low_dim >= 8 && low_dim <= FAL_MAX_LOW_DIM && low_dim % 8 == 0, FAL_EUNSUPPORTED,
                "fal_ivf_build: low_dim must be a multiple of 8 in [8, %d]", FAL_MAX_LOW_DIM);
    FAL_REQUIRE(n_buckets >= 0 && (n_buckets == 0 || (bucket_off && n_list)), FAL_EINVAL, "fal_ivf_build: NULL bucket arrays");
    FAL_REQUIRE(kmeans_iters >= 0, FAL_EINVAL, "fal_ivf_build: kmeans_iters < 0");
    {
        bool any_ivf = false;
        for (int64_t b = 0; b < n_buckets; ++b) any_ivf |= n_list[b] > 1;
        FAL_REQUIRE(n == 0 || X || !any_ivf, FAL_EINVAL, "fal_ivf_build: X may only be NULL when every bucket is flat");
    }
    if (n_buckets > 0) {
        FAL_REQUIRE(bucket_off[0] == 0 && bucket_off[n_buckets] == n, FAL_EINVAL,
                    "fal_ivf_build: bucket_off must start at 0 and end at n");
    } else {
        FAL_REQUIRE(n == 0, FAL_EINVAL, "fal_ivf_build: no buckets but n > 0");
    }
    ctx->counters[6] = 0;      // bit 0: rows of an indexed bucket, bit 1: rows handed to the flat prefilter have negative / non-finite
                               // components -> this index runs without float16 prefilters (fal_ctx_counter 6)
    fal_ivf* ivf = new fal_ivf();
    ivf->ctx = ctx;
    ivf->n = n;
    ivf->d = low_dim;
    ivf->X = X;
    ivf->bucket_off.assign(bucket_off, bucket_off + n_buckets + 1 * (n_buckets > 0));
    if (n_buckets == 0) ivf->bucket_off.assign(1, 0);
    ivf->n_list.assign(n_list, n_list + n_buckets);
    ivf->list_base.resize(n_buckets + 1);
    std::vector<BucketDev> bk;
    int64_t total = 0, waves = 0;
    for (int64_t b = 0; b < n_buckets; ++b) {
        const int64_t nb = bucket_off[b + 1] - bucket_off[b];
        if (nb < 0 || n_list[b] < 1 || n_list[b] > FAL_MAX_N_LIST || (nb > 0 && n_list[b] > nb) || nb > INT32_MAX) {
            set_error("fal_ivf_build: bucket %lld: size %lld, n_list %d invalid", (long long)b, (long long)nb, n_list[b]);
            delete ivf;
            return FAL_EINVAL;
        }
        ivf->list_base[b] = total;
        if (n_list[b] > 1) {
            bk.push_back({bucket_off[b], total, (int32_t)nb, n_list[b], waves});
            waves += n_list[b];
        }
        total += n_list[b];
    }
    ivf->list_base[n_buckets] = total;
    ivf->total_lists = total;
    ivf->n_ivf_buckets = (int)bk.size();
    ivf->ivf_waves = waves;
    hipStream_t st = ctx->stream;
    int rc = FAL_OK;
    auto fail = [&](int code) {
        fal_ivf_destroy(ivf);
        return code;
    };
#define B_TRY(e) do { rc = (e); if (rc != FAL_OK) return fail(rc); } while (0)
#define B_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error("%s:%d %s: %s", __FILE__, __LINE__, #e, hipGetErrorString(_e)); return fail(_e == hipErrorOutOfMemory ? FAL_ENOMEM : FAL_EHIP); } } while (0)

    B_TRY(ctx->pool_alloc(sizeof(int32_t) * (size_t)n, (void**)&ivf->perm));
    B_TRY(ctx->pool_alloc(sizeof(int32_t) * (size_t)n, (void**)&ivf->assign));
    B_TRY(ctx->pool_alloc(sizeof(int64_t) * (size_t)(total + 1), (void**)&ivf->list_off));
    B_TRY(ctx->pool_alloc(sizeof(int64_t) * (size_t)(total + 1), (void**)&ivf->counts));
    B_TRY(ctx->pool_alloc(sizeof(float) * (size_t)total * low_dim, (void**)&ivf->centroids));
    ctx->stage_reset(ST_BUILD);

    // counts: flat buckets hold all their rows in their single list
    {
        std::vector<int64_t> cnt_host(total + 1, 0), qlb;
        for (int64_t b = 0; b < n_buckets; ++b)
            if (n_list[b] == 1) cnt_host[ivf->list_base[b]] = bucket_off[b + 1] - bucket_off[b];
        B_TRY(ctx->upload(ivf->counts, cnt_host.data(), sizeof(int64_t) * (total + 1)));
    }
    B_HIP(hipMemsetAsync(ivf->assign, 0, sizeof(int32_t) * (size_t)n, st));
    B_HIP(hipMemsetAsync(ivf->centroids, 0, sizeof(float) * (size_t)total * low_dim, st));
    if (n > 0) {
        hipLaunchKernelGGL(iota_i32_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, st,
                           ivf->perm, n);
        B_HIP(hipGetLastError());
    }
    if (!bk.empty()) {
        B_TRY(ctx->pool_alloc(sizeof(BucketDev) * bk.size(), (void**)&ivf->bk_dev));
        B_TRY(ctx->upload(ivf->bk_dev, bk.data(), sizeof(BucketDev) * bk.size()));
        const BucketDev* bkd = (const BucketDev*)ivf->bk_dev;
        const int nbk = (int)bk.size();
        {
            StageScope ts(ctx, ST_BUILD);
            hipLaunchKernelGGL(kmeans_init_kernel, dim3((unsigned)waves), dim3(64), 0, st, X, low_dim, bkd, nbk,
                               ivf->centroids);
        B_HIP(hipGetLastError());
        }
        // The rows' sparse form (k-means update: centroid_update_kernel; exact chains of the kept pairs: pairs16.hip), made before
        // anything else because the same pass over the float32 rows tells whether any component is negative (or not finite):
        // the float16 prefilters' error bound |f16-MFMA - exact| <= 1.3e-3 v + 2e-6 holds for NON-NEGATIVE rows only, so with
        // such rows the build and the searches of this index use the exact float32 kernels (same results, slower).
        uint16_t* sp_cols = nullptr;
        float* sp_vals = nullptr;
        uint32_t *key_in = nullptr, *key_out = nullptr;
        int32_t* iota = nullptr;
        int64_t *boff_dev = nullptr, *lbase_dev = nullptr;
        int end_bit = 1;
        while (end_bit < 32 && (1ll << end_bit) < total) ++end_bit;
        {
            std::vector<int64_t> seg_off(bk.size() + 1, 0), tab(2 * (size_t)(n_buckets + 1));
            for (size_t i = 0; i < bk.size(); ++i) seg_off[i + 1] = seg_off[i] + ceil_div(bk[i].n, 256);
            for (int64_t b = 0; b <= n_buckets; ++b) {
                tab[b] = bucket_off[b];
                tab[n_buckets + 1 + b] = ivf->list_base[b];
            }
            int64_t* seg_dev = nullptr;
            void* sortbuf = nullptr;
            B_TRY(ctx->reserve(SLOT_MISC, sizeof(int64_t) * (seg_off.size() + 2), (void**)&seg_dev));
            B_TRY(ctx->upload(seg_dev, seg_off.data(), sizeof(int64_t) * seg_off.size()));
            int32_t* neg_dev = reinterpret_cast<int32_t*>(seg_dev + seg_off.size());
            B_HIP(hipMemsetAsync(neg_dev, 0, 2 * sizeof(int64_t), st));
            B_TRY(ctx->reserve(SLOT_MISC2, sizeof(int64_t) * tab.size(), (void**)&boff_dev));
            B_TRY(ctx->upload(boff_dev, tab.data(), sizeof(int64_t) * tab.size()));
            lbase_dev = boff_dev + n_buckets + 1;
            B_TRY(ctx->pool_alloc(sizeof(uint16_t) * (size_t)n * kSparseW, (void**)&ivf->sp_cols));
            B_TRY(ctx->pool_alloc(sizeof(float) * (size_t)n * kSparseW, (void**)&ivf->sp_vals));
            sp_cols = ivf->sp_cols;
            sp_vals = ivf->sp_vals;
            // (with float16 rows at hand the searches may run the float16 list scan: its queries arrive as 256-byte sparse records)
            // (rows of up to 400 columns: at 800 the dense gather is the faster form, list16s.hip)
            if (X16 != nullptr && low_dim <= 400) B_TRY(ctx->pool_alloc(sizeof(uint16_t) * (size_t)n * 2 * kSparseW, (void**)&ivf->sq16));
            B_TRY(ctx->reserve(SLOT_SORT2, 3 * sizeof(uint32_t) * (size_t)n, &sortbuf));
            key_in = (uint32_t*)sortbuf;
            key_out = key_in + n;
            iota = (int32_t*)(key_out + n);
            {
                StageScope ts(ctx, ST_BUILD);
                hipLaunchKernelGGL(sparsify_rows_kernel, dim3((unsigned)seg_off.back()), dim3(256), 0, st, X, low_dim, bkd, seg_dev,
                                   nbk, sp_cols, sp_vals, ivf->sq16, neg_dev);
                B_HIP(hipGetLastError());
                hipLaunchKernelGGL(iota_i32_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, st, iota, n);
                B_HIP(hipGetLastError());
            }
            // the one synchronisation of the build (only when float16 rows were handed in: ~20 us of an empty queue against the
            // tens of ms of the k-means passes that follow)
            if (X16 != nullptr) {
                B_HIP(hipMemcpyAsync(ctx->fb_host + 6, neg_dev, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                B_HIP(hipStreamSynchronize(st));
                ivf->rows_signed = ctx->fb_host[6] != 0 ? 1 : 0;
                ivf->rows_f16 = ctx->fb_host[7] == 0 ? 1 : 0;
                ivf->rows_many = ctx->fb_host[8] != 0 ? 1 : 0;
                ctx->counters[6] |= ivf->rows_signed;
            } else {
                ivf->rows_signed = -1;        // not known on the host yet: fal_ivf_attach_prefilter_ex reads it if it needs it
                B_TRY(ctx->pool_alloc(sizeof(int32_t) * 2, (void**)&ivf->neg_dev));
                B_HIP(hipMemcpyAsync(ivf->neg_dev, neg_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            }
        }
        // Buckets with few lists (<= kAssignMaxLists) are assigned by the shared-stream kernel (assign.hip): jobs =
        // (row segment) x (group of <= 128 centroids).  Buckets with many lists keep the row-resident kernel, whose
        // 32-row tile is amortised over their many centroid chunks.
        constexpr int kAssignMaxLists = 512;
        std::vector<AssignJob> ajobs;
        std::vector<DenseJob> djobs;
        int64_t dtiles = 0;
        // rows per job: long segments amortise the resident centroid tiles, but with only a few indexed buckets long
        // segments leave most CUs idle -- aim for >= 4 workgroups per CU, between 256 and kAssignSeg rows
        int64_t work_rows = 0;
        for (const BucketDev& b : bk)
            if (b.n_list > 64 && b.n_list <= kAssignMaxLists) work_rows += (int64_t)b.n * ceil_div(b.n_list, kAssignGroup);
        const int64_t seg = std::min<int64_t>(kAssignSeg, std::max<int64_t>(256, ceil_div(work_rows, (int64_t)ctx->num_cus * 4 * 32) * 32));
        // buckets with one or two centroid tiles take one-wave jobs (a 4-wave workgroup would idle most of its waves)
        std::vector<AssignJob> wjobs;
        int64_t wave_rows = 0;
        for (const BucketDev& b : bk)
            if (b.n_list <= 64) wave_rows += (int64_t)b.n * ceil_div(b.n_list, 32);
        const int64_t wseg = std::min<int64_t>(kAssignSeg, std::max<int64_t>(256, ceil_div(wave_rows, (int64_t)ctx->num_cus * 16 * 32) * 32));
        // buckets with <= 128 lists and float16 rows at hand: the prefiltered assignment (assign16.hip; identical results)
        const bool use16 = X16 != nullptr && assign16_supports(low_dim) && ivf->rows_signed == 0;
        std::vector<AssignJob> hjobs, mjobs, gjobs;    // single-group jobs; merge jobs and their group jobs (129..kAssignMergeLists lists)
        int merge_max_lists = 0;
        int64_t merge_rows = 0;                        // rows of the merge buckets: the partial arrays' rows (AssignJob::part0)
        for (const BucketDev& b : bk) {
            if (use16 && b.n_list <= kAssignGroup) {
                for (int64_t s0 = 0; s0 < b.n; s0 += kAssignSeg)
                    hjobs.push_back({b.row0 + s0, b.list0, (int32_t)std::min<int64_t>(kAssignSeg, b.n - s0), b.n_list, 0, 0});
            } else if (use16 && b.n_list <= kAssignMergeLists) {
                merge_max_lists = std::max(merge_max_lists, (int)b.n_list);
                for (int64_t s0 = 0; s0 < b.n; s0 += kAssignSeg) {
                    const int32_t nr = (int32_t)std::min<int64_t>(kAssignSeg, b.n - s0);
                    mjobs.push_back({b.row0 + s0, b.list0, nr, b.n_list, 0, (int32_t)merge_rows});
                    for (int t0 = 0; t0 < b.n_list; t0 += kAssignGroup)      // the groups of a segment next to each other: they
                        gjobs.push_back({b.row0 + s0, b.list0 + t0, nr, std::min(kAssignGroup, b.n_list - t0), t0, (int32_t)merge_rows});   // share its rows in L2
                    merge_rows += nr;
                }
            } else if (b.n_list <= 64) {
                for (int64_t s0 = 0; s0 < b.n; s0 += wseg)
                    for (int t0 = 0; t0 < b.n_list; t0 += 32)
                        wjobs.push_back({b.row0 + s0, b.list0 + t0, (int32_t)std::min<int64_t>(wseg, b.n - s0),
                                         std::min(32, b.n_list - t0), t0, 0});
            } else if (b.n_list <= kAssignMaxLists) {
                for (int64_t s0 = 0; s0 < b.n; s0 += seg)
                    for (int t0 = 0; t0 < b.n_list; t0 += kAssignGroup)
                        ajobs.push_back({b.row0 + s0, b.list0 + t0, (int32_t)std::min<int64_t>(seg, b.n - s0),
                                         std::min(kAssignGroup, b.n_list - t0), t0, 0});
            } else {
                djobs.push_back({b.row0, b.list0, 0, dtiles, b.n, b.n_list, 0});
                dtiles += ceil_div(b.n, 32);
            }
        }
        AssignJob* ajobs_dev = nullptr;
        DenseJob* djobs_dev = nullptr;
        unsigned long long* keys = nullptr;
        const int64_t n_wide = (int64_t)ajobs.size(), n_wave = (int64_t)wjobs.size();
        ajobs.insert(ajobs.end(), wjobs.begin(), wjobs.end());
        if (!ajobs.empty()) {
            B_TRY(ctx->reserve(SLOT_JOBS2, sizeof(AssignJob) * ajobs.size(), (void**)&ajobs_dev));
            B_TRY(ctx->upload(ajobs_dev, ajobs.data(), sizeof(AssignJob) * ajobs.size()));
            B_TRY(ctx->reserve(SLOT_SIMS, sizeof(unsigned long long) * (size_t)n, (void**)&keys));
        }
        if (!djobs.empty()) {
            B_TRY(ctx->reserve(SLOT_JOBS, sizeof(DenseJob) * djobs.size(), (void**)&djobs_dev));
            B_TRY(ctx->upload(djobs_dev, djobs.data(), sizeof(DenseJob) * djobs.size()));
        }
        AssignJob* hjobs_dev = nullptr;
        void* C16 = nullptr;
        const int64_t n_single = (int64_t)hjobs.size(), n_merge = (int64_t)mjobs.size(), n_group = (int64_t)gjobs.size();
        hjobs.insert(hjobs.end(), mjobs.begin(), mjobs.end());
        hjobs.insert(hjobs.end(), gjobs.begin(), gjobs.end());
        if (!hjobs.empty()) {
            B_TRY(ctx->reserve(SLOT_INV, sizeof(AssignJob) * hjobs.size(), (void**)&hjobs_dev));
            B_TRY(ctx->upload(hjobs_dev, hjobs.data(), sizeof(AssignJob) * hjobs.size()));
            B_TRY(ctx->reserve(SLOT_INVCNT, sizeof(uint16_t) * (size_t)total * low_dim + 64, &C16));
        }
        // stable sort of the rows by (global) list: perm = rows in list order, list_off = the lists' boundaries
        auto sort_by_list = [&]() -> int {
            hipLaunchKernelGGL(list_key_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 8192)), dim3(256), 0, st, ivf->assign,
                               boff_dev, lbase_dev, (int)n_buckets, n, key_in);
            FAL_CHECK_HIP(hipGetLastError());
            FAL_TRY(sort_pairs_u32_i32(ctx, key_in, key_out, iota, ivf->perm, n, end_bit, SLOT_SORT));
            hipLaunchKernelGGL(list_bounds_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total + 1, 256), 4096)), dim3(256), 0, st,
                               key_out, n, total, ivf->list_off);
            FAL_CHECK_HIP(hipGetLastError());
            return FAL_OK;
        };
        // after the first (global) sort every later one only re-orders the rows inside the IVF buckets: one wave per bucket
        int max_nl_all = 0;
        for (const BucketDev& b : bk) max_nl_all = std::max(max_nl_all, (int)b.n_list);
        bool sorted_once = false;
        auto sort_buckets = [&]() -> int {
            if (!sorted_once || max_nl_all > kBucketSortMaxLists) {
                sorted_once = true;
                return sort_by_list();
            }
            hipLaunchKernelGGL(bucket_list_sort_kernel, dim3((unsigned)nbk), dim3(64), 0, st, ivf->assign, bkd, nbk, ivf->perm, ivf->list_off);
            FAL_CHECK_HIP(hipGetLastError());
            return FAL_OK;
        };
        for (int it = 0; it <= kmeans_iters; ++it) {
            if (!hjobs.empty()) {
                if (it == 0) {                   // (later passes: centroid_update_kernel writes the float16 copy of what it changes)
                    StageScope ts(ctx, ST_BUILD);
                    B_TRY(launch_cvt_f16(ctx, ivf->centroids, C16, total * low_dim));
                }
                // the final pass leaves its approximate similarities behind for the coarse quantiser (coarse16.hip)
                if (it == kmeans_iters && djobs.empty() && ajobs.empty()) {
                    int max_nl = 0;
                    for (const BucketDev& b : bk) max_nl = std::max(max_nl, (int)b.n_list);
                    ivf->ckeys_stride = kAssignGroup * (int)ceil_div(max_nl, kAssignGroup);
                    // n x stride keys: ONE 2,048-list bucket sizes them for every row of the partition (41 GB at 10 M rows).  Beyond
                    // a budget (FALCON_CKEYS_MB, default 32 GB), or when the allocation fails, the index is built without them and the
                    // coarse quantiser scans in float32 (search.hip `from_keys`: the same probes, slower).
                    const char* ke = getenv("FALCON_CKEYS_MB");
                    const size_t budget = (ke ? (size_t)atoll(ke) : (size_t)32768) << 20;
                    const size_t key_bytes = sizeof(uint16_t) * (size_t)n * ivf->ckeys_stride;
                    if (key_bytes > budget || ctx->pool_alloc(key_bytes, (void**)&ivf->ckeys) != FAL_OK) {
                        ivf->ckeys = nullptr;
                        (void)hipGetLastError();
                    }
                }
                B_TRY(launch_assign16(ctx, ST_BUILD, X16, X, C16, ivf->centroids, low_dim, hjobs_dev, n_single, n_merge, n_group, n,
                                      ivf->assign, it == kmeans_iters ? ivf->ckeys : nullptr, ivf->ckeys_stride, sp_cols, sp_vals,
                                      merge_max_lists, merge_rows));
            }
            if (!djobs.empty())
                B_TRY(launch_dense(ctx, ST_BUILD, EPI_ARGMAX, X, ivf->centroids, low_dim, djobs_dev, (int)djobs.size(), 0,
                                   dtiles, nullptr, 0, ivf->assign));
            if (n_wide + n_wave > 0)
                B_TRY(launch_assign(ctx, ST_BUILD, X, ivf->centroids, low_dim, ajobs_dev, n_wide, n_wave, n, keys, ivf->assign));
            if (it == kmeans_iters) break;   // final assignment against the final centroids
            StageScope ts(ctx, ST_BUILD);
            B_TRY(sort_buckets());
            hipLaunchKernelGGL(centroid_update_kernel, dim3((unsigned)waves), dim3(64), 0, st, sp_cols, sp_vals, X, low_dim,
                               ivf->perm, ivf->list_off, bkd, nbk, ivf->centroids, reinterpret_cast<__half*>(C16));
            B_HIP(hipGetLastError());
        }
        StageScope ts(ctx, ST_BUILD);
        B_TRY(sort_buckets());       // (the sparse rows stay with the index: pairs16.hip evaluates its exact chains over them)
        // the float32 rows in list order are made on demand (fal_ivf_ensure_xl: the staged fine scan and the staged coarse scan read
        // them; the default path -- coarse quantiser from the build's keys, float16 prefilter -- does not: 16 GB at 10 M spectra)
        ivf->Xl = nullptr;
    } else {
        // every bucket is flat: the list offsets ARE the bucket offsets -- known on the host, uploaded, no kernel (a one-workgroup
        // scan here sat 585 us in the two-stream trace of the headline, waiting for a CU behind the other partition's matrix kernel)
        std::vector<int64_t> off_host((size_t)total + 1, 0);
        for (int64_t b = 0; b < n_buckets; ++b) off_host[(size_t)ivf->list_base[b] + 1] = bucket_off[b + 1];
        for (int64_t g = 1; g <= total; ++g) off_host[(size_t)g] = std::max(off_host[(size_t)g], off_host[(size_t)g - 1]);
        B_TRY(ctx->upload(ivf->list_off, off_host.data(), sizeof(int64_t) * (size_t)(total + 1)));
        ivf->Xl = X;
    }
    B_HIP(hipGetLastError());
#undef B_TRY
#undef B_HIP
    *out = ivf;
    return FAL_OK;
}

int fal_ivf_export(fal_ctx* ctx, const fal_ivf* ivf, float* centroids, int32_t* assign, int32_t* perm,
                   int64_t* list_off) {
    fal::CallScope _call(ctx);
    FAL_REQUIRE(ctx && ivf, FAL_EINVAL, "fal_ivf_export: NULL");
    hipStream_t st = ctx->stream;
    if (centroids)
        FAL_CHECK_HIP(hipMemcpyAsync(centroids, ivf->centroids, sizeof(float) * (size_t)ivf->total_lists * ivf->d,
                                     hipMemcpyDeviceToDevice, st));
    if (assign) FAL_CHECK_HIP(hipMemcpyAsync(assign, ivf->assign, sizeof(int32_t) * (size_t)ivf->n, hipMemcpyDeviceToDevice, st));
    if (perm) FAL_CHECK_HIP(hipMemcpyAsync(perm, ivf->perm, sizeof(int32_t) * (size_t)ivf->n, hipMemcpyDeviceToDevice, st));
    if (list_off)
        FAL_CHECK_HIP(hipMemcpyAsync(list_off, ivf->list_off, sizeof(int64_t) * (size_t)(ivf->total_lists + 1),
                                     hipMemcpyDeviceToDevice, st));
    return FAL_OK;
}

}  // extern "C"
