// IVF prefilter path, the exact part WITHOUT the band: only the (query, candidate) pairs that can matter are evaluated.
//
// band_kernel<., IVF> evaluates the whole precursor window of a 32-query tile on the fp32 matrix cores (240 rows at
// BASELINE configs[2]'s bucket density, 870 at configs[3]'s) although a query keeps ~5 of them: the cost grows with the
// window, the need does not.  Here:
//   kept16_kernel   16 lanes per query, one per probed list.  Inside a list the rows keep their precursor order, so the part of
//      a probed list inside the query's precursor window is a contiguous range of list positions (two binary searches on the
//      precursor m/z in list order) and its keys are contiguous in the query's key stream.  A row stays if its key is not
//      certainly below the k-th best (select16_kernel left that threshold) and the exact tolerance tests pass;
//   pairs16_kernel   the exact similarity of every kept pair by the k-ordered fmaf chain (bit-identical to the matrix-core
//      chain), one pair per lane, the pairs of a 32-query tile compacted so that the lanes are full;
//   resolve_kernel (fused.hip, ivf = 2)   thresholds on the exact values, the k-th key for ambiguous queries, sort, store.
//
// Reference: README.md:107-113, 137-142; see ivf16.hip for the error bound and the rest of the path.
#include <math.h>
#include <type_traits>
#include <limits.h>
#include <algorithm>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "fused.h"
#include "ivf16.h"
#include "kept16.h"

namespace fal {


// pmz_l[pos] = pmz[perm[pos]]: precursor m/z in list order (inside a list the rows keep their sorted order: ascending)
__global__ void gather_pmz_kernel(const float* __restrict__ pmz, const int32_t* __restrict__ perm, int64_t n, float* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = pmz[perm[i]];
}

// kept16_kernel: kept16_query (kept16.h) for four queries per wave -- the form the searches with a second select16 pass use
// (queries of more than 2,048 keys: n_probe 32); the common case runs the same code as the tail of select16_kernel.
__global__ __launch_bounds__(256) void kept16_kernel(Kept16Args a) {
    const int tid = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * 16 + (tid >> 4);     // tile-order slot (tile-order = list-order positions)
    const int64_t t = a.tile_begin + (g >> 5);
    const int ql = (int)(g & 31);
    bool live = (g >> 5) < a.n_tiles;
    DenseJob job{};
    if (live) job = a.jobs[a.tile_job[g >> 5]];
    const int lt = (int)(t - job.tile0);
    live = live && 32 * lt + ql < job.nq;
    const int64_t p = live ? job.q_row0 + 32 * (int64_t)lt + ql : 0;      // the query's list-order position
    const int64_t row = live ? a.perm[p] : 0;                    // ... and sorted row
    const uint16_t* krow = a.keys + (a.q_sim_off[live ? 32 * t + ql : 32 * a.tile_begin] - a.keys_base);
    const int2 sel = live ? a.gsel[row] : make_int2(INT32_MAX, INT32_MIN);
    kept16_query(a, live, p, row, job.c_row0, krow, sel, tid);
}

// exact similarity of every kept pair: the pairs of a 32-query tile in consecutive lanes, one k-ordered fmaf chain per lane
// (bit-identical to the matrix-core chain).  A lane's candidate row is a gathered row: with a row per lane a load instruction
// touches 64 cache lines for 16 bytes each and the texture addresser, not the memory, sets the pace (94 us per chain measured).
// So the candidates' bytes are fetched COALESCED -- an instruction covers 16 rows x 64 bytes: lane L fetches piece L & 3 of the
// row of pair 16g + (L >> 2) -- and transposed through LDS (pair slots of 176 bytes: conflict-free b128 reads) into the lanes
// that own the pairs.  The query rows are shared by neighbouring lanes (a wave's 64 pairs span a dozen queries): read directly.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void pairs16_kernel(FusedArgs a, int d) {
    constexpr int kSlot = 176;                                   // bytes per pair: 64 (low half) + 16 + 64 (high half) + 32
    constexpr int U = 4;                                         // 16-byte steps of either half per batch
    __shared__ int32_t off[33];
    __shared__ __attribute__((aligned(16))) unsigned char tbuf[4][64 * kSlot];
    int ji, lt;
    if (!find_job32(a, blockIdx.x, &ji, &lt)) return;
    const DenseJob job = a.jobs32[ji];
    const int nqw = min(32, job.nc - 32 * lt);
    const int64_t row_t = job.q_row0 + 32 * lt;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 64) {                                      // counts of the tile's queries, prefix sum by the first wave
        const int q = threadIdx.x;
        int c = 0;
        if (q < nqw) {
            const int i0 = a.gkcnt[(row_t + q) * 2], i1 = a.gkcnt[(row_t + q) * 2 + 1];
            if (((i0 | i1) & 0x200) == 0) c = (i0 & 0xFF) + (i1 & 0xFF);
        }
        const int incl = wave_prefix_sum(c);
        if (q < 32) off[q + 1] = incl;
        if (q == 0) off[0] = 0;
    }
    __syncthreads();
    const int P = off[32];
    const int dh4 = d >> 3;
    const float4* X4 = reinterpret_cast<const float4*>(a.X);
    unsigned char* tb = tbuf[w];
    for (int i0 = 64 * w; i0 < P; i0 += 256) {                   // (wave-uniform: the lanes of a wave work together)
        const int i = i0 + lane;
        const bool live = i < P;
        const int ic = min(i, P - 1);
        int q = 0;
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) q = (q + s < 32 && off[q + s] <= ic) ? q + s : q;
        const int j = ic - off[q];
        const int64_t row = row_t + q;
        const uint32_t id = a.gkept_id[row * FAL_FUSED_KEEP + j];
        int64_t idT[4];                                          // candidate row of the pair this lane FETCHES for in instruction g
#pragma unroll
        for (int g = 0; g < 4; ++g) idT[g] = (int64_t)(uint32_t)__shfl((int)id, 16 * g + (lane >> 2), 64) * (2 * dh4);
        const float4* qp = X4 + row * (2 * dh4);
        float acc = 0.f;
        for (int j0 = 0; j0 < dh4; j0 += U) {
            const int pc = min(j0 + (lane & 3), dh4 - 1);
            float4 sl[4], sh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sl[g] = X4[idT[g] + pc];
                sh[g] = X4[idT[g] + dh4 + pc];
            }
            float4 ql[U], qh[U];
#pragma unroll
            for (int t = 0; t < U; ++t) {
                const int jj = min(j0 + t, dh4 - 1);
                ql[t] = qp[jj];
                qh[t] = qp[dh4 + jj];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned char* slot = tb + (16 * g + (lane >> 2)) * kSlot + (lane & 3) * 16;
                *reinterpret_cast<float4*>(slot) = sl[g];
                *reinterpret_cast<float4*>(slot + 80) = sh[g];
            }
            wave_lds_sync();
            const unsigned char* mine = tb + lane * kSlot;
#pragma unroll
            for (int t = 0; t < U; ++t) {
                if (j0 + t < dh4) {                              // (wave-uniform)
                    const float4 cl = *reinterpret_cast<const float4*>(mine + t * 16);
                    const float4 ch = *reinterpret_cast<const float4*>(mine + 80 + t * 16);
                    acc = __builtin_fmaf(ql[t].x, cl.x, acc);
                    acc = __builtin_fmaf(qh[t].x, ch.x, acc);
                    acc = __builtin_fmaf(ql[t].y, cl.y, acc);
                    acc = __builtin_fmaf(qh[t].y, ch.y, acc);
                    acc = __builtin_fmaf(ql[t].z, cl.z, acc);
                    acc = __builtin_fmaf(qh[t].z, ch.z, acc);
                    acc = __builtin_fmaf(ql[t].w, cl.w, acc);
                    acc = __builtin_fmaf(qh[t].w, ch.w, acc);
                }
            }
            wave_lds_sync();
        }
        if (live) a.gkept_u[row * FAL_FUSED_KEEP + j] = max(f32_sortable(acc), 1u);
    }
}

// The same chains over the rows' SPARSE form (ivf.h: <= 64 (column, value) entries per row, in chain order).  A term whose
// candidate component is zero leaves the accumulator as it is (q * 0 = +-0, and the accumulator, starting at +0.0, is never -0.0
// unless every earlier product underflowed below 2^-150), so the chain over the candidate's entries alone gives the same bits
// while a pair costs 384 bytes and <= 64 fmaf instead of 4 d bytes and d fmaf.  The tile's 32 query rows are expanded into LDS
// once (from their own sparse form); a lane reads the query component of an entry from there.  The candidates' entries are
// fetched coalesced, 16 entries of 64 pairs at a time (an instruction covers 16 pairs x 64 bytes of values / 32 pairs x 32 bytes
// of columns), and transposed through LDS into the lanes that own the pairs.  Rows with more than 64 non-zeros (entry 0 =
// kColDense) take the dense chain.
// QH: the query tile in LDS as float16 -- only when every component IS a float16 value (float16 vectors: FusedArgs.rows_f16,
// checked by the build's pass over the rows), so the conversion back is exact and the chains keep their bits; at low_dim 800 the
// tile then takes 51 KB instead of 102 and two workgroups share a CU.
template <bool QH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pairs16s_kernel(FusedArgs a, int d) {
    constexpr int kSlot = 112;                                   // bytes per pair and step: 16 columns + 16 values + 16
    using QT = typename std::conditional<QH, __half, float>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QT* qs = reinterpret_cast<QT*>(smem);                        // [32][d]
    unsigned char* tbuf = smem + (size_t)32 * d * sizeof(QT);    // [4][64 * kSlot]
    int32_t* off = reinterpret_cast<int32_t*>(tbuf + 4 * 64 * kSlot);   // [33]
    int ji, lt;
    if (!find_job32(a, blockIdx.x, &ji, &lt)) return;
    const DenseJob job = a.jobs32[ji];
    const int nqw = min(32, job.nc - 32 * lt);
    const int64_t row_t = job.q_row0 + 32 * lt;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 64) {
        const int q = threadIdx.x;
        int c = 0;
        if (q < nqw) {
            const int i0 = a.gkcnt[(row_t + q) * 2], i1 = a.gkcnt[(row_t + q) * 2 + 1];
            if (((i0 | i1) & 0x200) == 0) c = (i0 & 0xFF) + (i1 & 0xFF);
        }
        const int incl = wave_prefix_sum(c);
        if (q < 32) off[q + 1] = incl;
        if (q == 0) off[0] = 0;
    }
    // the sparse rows of this wave's queries: all loads in flight before the accumulator rows are cleared
    int qc[8];
    float qv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int64_t row = row_t + min(w + 4 * t, nqw - 1);
        qc[t] = a.sp_cols[row * kSparseW + lane];
        qv[t] = a.sp_vals[row * kSparseW + lane];
    }
    {
        float4* z = reinterpret_cast<float4*>(qs);
        for (int i = threadIdx.x; i < (int)(8 * d * sizeof(QT) / 4); i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int P = off[32];
    if (P == 0) return;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int q = w + 4 * t;
        if (q < nqw) {
            if (__shfl(qc[t], 0, 64) == (int)kColDense) {
                const int64_t row = row_t + q;
                for (int e = lane; e < d; e += 64) qs[q * d + e] = (QT)a.X[row * d + e];
            } else if (qc[t] != (int)kColPad) {
                qs[q * d + qc[t]] = (QT)qv[t];
            }
        }
    }
    __syncthreads();
    const int dh4 = d >> 3;
    unsigned char* tb = tbuf + w * 64 * kSlot;
    for (int i0 = 64 * w; i0 < P; i0 += 256) {                   // (wave-uniform: the lanes of a wave work together)
        const int i = i0 + lane;
        const bool live = i < P;
        const int ic = min(i, P - 1);
        int q = 0;
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) q = (q + s < 32 && off[q + s] <= ic) ? q + s : q;
        const int j = ic - off[q];
        const int64_t row = row_t + q;
        const uint32_t id = a.gkept_id[row * FAL_FUSED_KEEP + j];
        int64_t idV[4], idC[2];                                  // rows this lane FETCHES for: values (16 B of 64), columns (16 B of 32)
#pragma unroll
        for (int g = 0; g < 4; ++g) idV[g] = (int64_t)(uint32_t)__shfl((int)id, 16 * g + (lane >> 2), 64) * kSparseW + 4 * (lane & 3);
#pragma unroll
        for (int g = 0; g < 2; ++g) idC[g] = (int64_t)(uint32_t)__shfl((int)id, 32 * g + (lane >> 1), 64) * kSparseW + 8 * (lane & 1);
        const QT* qrow = qs + q * d;
        float acc = 0.f;
        bool dense = false;
        // every entry of the 64 candidates in flight at once (four steps of 16 entries)
#define FAL_LD_V(g, c) (*reinterpret_cast<const float4*>(a.sp_vals + idV[g] + 16 * (c)))
#define FAL_LD_C(g, c) (*reinterpret_cast<const uint4*>(a.sp_cols + idC[g] + 16 * (c)))
        const float4 a00 = FAL_LD_V(0, 0), a01 = FAL_LD_V(1, 0), a02 = FAL_LD_V(2, 0), a03 = FAL_LD_V(3, 0);
        const uint4 b00 = FAL_LD_C(0, 0), b01 = FAL_LD_C(1, 0);
        const float4 a10 = FAL_LD_V(0, 1), a11 = FAL_LD_V(1, 1), a12 = FAL_LD_V(2, 1), a13 = FAL_LD_V(3, 1);
        const uint4 b10 = FAL_LD_C(0, 1), b11 = FAL_LD_C(1, 1);
        const float4 a20 = FAL_LD_V(0, 2), a21 = FAL_LD_V(1, 2), a22 = FAL_LD_V(2, 2), a23 = FAL_LD_V(3, 2);
        const uint4 b20 = FAL_LD_C(0, 2), b21 = FAL_LD_C(1, 2);
        const float4 a30 = FAL_LD_V(0, 3), a31 = FAL_LD_V(1, 3), a32 = FAL_LD_V(2, 3), a33 = FAL_LD_V(3, 3);
        const uint4 b30 = FAL_LD_C(0, 3), b31 = FAL_LD_C(1, 3);
#undef FAL_LD_V
#undef FAL_LD_C
        bool more = true;                                        // (wave-uniform) some row has entries in this step
        unsigned char* wv = tb + (lane >> 2) * kSlot + 32 + (lane & 3) * 16;      // this lane's piece of pair (lane >> 2) + 16 g
        unsigned char* wc = tb + (lane >> 1) * kSlot + (lane & 1) * 16;           // ... of pair (lane >> 1) + 32 g
        auto step = [&](float4 s0, float4 s1, float4 s2, float4 s3, uint4 t0, uint4 t1, bool first) {
            if (!more) return;
            *reinterpret_cast<float4*>(wv) = s0;
            *reinterpret_cast<float4*>(wv + 16 * kSlot) = s1;
            *reinterpret_cast<float4*>(wv + 32 * kSlot) = s2;
            *reinterpret_cast<float4*>(wv + 48 * kSlot) = s3;
            *reinterpret_cast<uint4*>(wc) = t0;
            *reinterpret_cast<uint4*>(wc + 32 * kSlot) = t1;
            wave_lds_sync();
            const unsigned char* mine = tb + lane * kSlot;
            const uint4 c0 = *reinterpret_cast<const uint4*>(mine), c1 = *reinterpret_cast<const uint4*>(mine + 16);
            const float4 v0 = *reinterpret_cast<const float4*>(mine + 32), v1 = *reinterpret_cast<const float4*>(mine + 48);
            const float4 v2 = *reinterpret_cast<const float4*>(mine + 64), v3 = *reinterpret_cast<const float4*>(mine + 80);
            if (first && (c0.x & 0xFFFFu) == (uint32_t)kColDense) dense = true;
            auto term = [&](uint32_t col, float val) {
                const bool on = col < (uint32_t)kColDense;
                const float qv = (float)qrow[on ? col : 0u];
                const float nx = __builtin_fmaf(qv, val, acc);
                acc = on ? nx : acc;
            };
            term(c0.x & 0xFFFFu, v0.x); term(c0.x >> 16, v0.y); term(c0.y & 0xFFFFu, v0.z); term(c0.y >> 16, v0.w);
            term(c0.z & 0xFFFFu, v1.x); term(c0.z >> 16, v1.y); term(c0.w & 0xFFFFu, v1.z); term(c0.w >> 16, v1.w);
            term(c1.x & 0xFFFFu, v2.x); term(c1.x >> 16, v2.y); term(c1.y & 0xFFFFu, v2.z); term(c1.y >> 16, v2.w);
            term(c1.z & 0xFFFFu, v3.x); term(c1.z >> 16, v3.y); term(c1.w & 0xFFFFu, v3.z); term(c1.w >> 16, v3.w);
            wave_lds_sync();
            // entries are packed: a row whose last entry of this step is unused has none in the next
            more = __ballot((c1.w >> 16) < (uint32_t)kColDense) != 0ull;
        };
        step(a00, a01, a02, a03, b00, b01, true);
        step(a10, a11, a12, a13, b10, b11, false);
        step(a20, a21, a22, a23, b20, b21, false);
        step(a30, a31, a32, a33, b30, b31, false);
        if (__ballot(dense) != 0ull) {                           // (rare) rows kept dense: the dense chain
            if (dense) {
                const float4* cp = reinterpret_cast<const float4*>(a.X) + (int64_t)id * (2 * dh4);
                auto q4 = [&](int e4) -> float4 {
                    return make_float4((float)qrow[4 * e4], (float)qrow[4 * e4 + 1], (float)qrow[4 * e4 + 2], (float)qrow[4 * e4 + 3]);
                };
                acc = 0.f;
                for (int jj = 0; jj < dh4; ++jj) {
                    const float4 cl = cp[jj], chh = cp[dh4 + jj], ql = q4(jj), qh = q4(dh4 + jj);
                    acc = __builtin_fmaf(ql.x, cl.x, acc);
                    acc = __builtin_fmaf(qh.x, chh.x, acc);
                    acc = __builtin_fmaf(ql.y, cl.y, acc);
                    acc = __builtin_fmaf(qh.y, chh.y, acc);
                    acc = __builtin_fmaf(ql.z, cl.z, acc);
                    acc = __builtin_fmaf(qh.z, chh.z, acc);
                    acc = __builtin_fmaf(ql.w, cl.w, acc);
                    acc = __builtin_fmaf(qh.w, chh.w, acc);
                }
            }
        }
        if (live) a.gkept_u[row * FAL_FUSED_KEEP + j] = max(f32_sortable(acc), 1u);
    }
}

int launch_gather_pmz(fal_ctx* ctx, const float* pmz, const int32_t* perm, int64_t n, float* out) {
    if (n <= 0) return FAL_OK;
    StageScope ts(ctx, ST_COARSE);
    hipLaunchKernelGGL(gather_pmz_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->num_cus * 16)), dim3(256), 0,
                       ctx->stream, pmz, perm, n, out);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_kept16(fal_ctx* ctx, const Kept16Args& a_in, int64_t n_tiles) {
    if (n_tiles <= 0) return FAL_OK;
    Kept16Args a = a_in;
    int32_t* tj = nullptr;           // (the tile -> job table launch_select16 left in the slot for the same tiles)
    ctx->release(SLOT_TILEJOB);        // a launcher-local table: the previous launcher's pointer is dead
    FAL_TRY(ctx->reserve(SLOT_TILEJOB, sizeof(int32_t) * 16, (void**)&tj));
    a.tile_job = tj;
    a.n_tiles = n_tiles;
    StageScope ts(ctx, ST_SELECT);
    hipLaunchKernelGGL(kept16_kernel, dim3((unsigned)(n_tiles * 2)), dim3(256), 0, ctx->stream, a);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_pairs16(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles32) {
    if (list_tiles32 <= 0) return FAL_OK;
    StageScope ts(ctx, ST_SCAN);
    if (a.sp_cols != nullptr && a.sp_vals != nullptr && d <= 832) {        // (32 query rows of d floats + 28 KB must fit 160 KB of LDS)
        const bool qh = a.rows_f16 == 1 && d > 512;              // (below, two workgroups fit a CU with a float32 tile as well)
        const size_t lds = (size_t)32 * d * (qh ? 2 : 4) + 4 * 64 * 112 + 34 * 4;
        // (per launch: the attribute is per device, and two partition threads may launch at once)
        const void* fn = qh ? (const void*)pairs16s_kernel<true> : (const void*)pairs16s_kernel<false>;
        FAL_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(96 * 1024, lds)));
        if (qh) hipLaunchKernelGGL(pairs16s_kernel<true>, dim3((unsigned)(list_tiles32 * 8)), dim3(256), lds, ctx->stream, a, d);
        else hipLaunchKernelGGL(pairs16s_kernel<false>, dim3((unsigned)(list_tiles32 * 8)), dim3(256), lds, ctx->stream, a, d);
    } else {
        hipLaunchKernelGGL(pairs16_kernel, dim3((unsigned)(list_tiles32 * 8)), dim3(256), 0, ctx->stream, a, d);
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::gather_pmz_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)
