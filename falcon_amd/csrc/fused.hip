// a7 + a8 for FLAT buckets in one kernel: the cosine scan with the top-k kept on chip.
//
// What the staged path does (scan.hip): fp32-MFMA similarity of every (query, candidate) pair of the bucket ->
// [32, nc] block in HBM -> select kernel: k_ann best per query -> precursor / RT filter -> sort -> neighbour
// lists.  The result only needs (i) the exact similarities of the candidates inside the query's precursor
// window -- a contiguous band next to the diagonal, because the bucket's rows are sorted by precursor m/z -- and
// (ii) the k_ann-th best key of the row, to decide which of them made the top-k_ann.  So this kernel
//
//   1. scans the whole bucket on the f16 matrix cores (float16 copies of the rows, float32 accumulation:
//      1/16 of the fp32 matrix-pipe cycles) and builds a 256-bin histogram of the APPROXIMATE similarities of
//      every query in LDS; the bin b* that holds the k_ann-th best approximate value follows from a suffix sum;
//   2. scans a second time and collects the candidates whose approximate value lies in (a slightly widened) b*:
//      the "members", <= 64 per query, in LDS.  The k_ann-th best approximate value T~ is selected among them;
//      |approx - exact| <= eps(value) (bound below) puts the exact k_ann-th best value inside [T~ - eps, T~ + eps];
//   3. computes the EXACT similarities of the precursor window with the fp32 matrix cores (same k-ordered
//      fmaf chain as dense_kernel: bit-identical values) and classifies every in-tolerance candidate:
//      above T~ + eps -> selected, below T~ - eps -> not selected, in between -> resolved exactly: the few
//      members within 2 eps of T~ are re-evaluated exactly (VALU fmaf chain, the same bits) and the true
//      k_ann-th key (value, id) decides;
//   4. sorts the selected in-tolerance candidates and writes the neighbour lists.
//
// Nothing but the neighbour lists leaves the CU: no [n, nc] similarity hand-off through HBM.  The output is
// BIT-IDENTICAL to fal_ivf_search_topk -> fal_filter_neighbors (tests/test_gpu_search.py).  Queries the
// on-chip structures cannot hold (more than 64 members -- e.g. hundreds of identical spectra --, more than 48
// selected window candidates) are appended to a fallback list and redone by fused_fallback_kernel: exact row
// by VALU fmaf chains + the staged path's own selection code.
//
// Error bound of step 1 (non-negative unit vectors x, y; ^ = rounded to float16): each factor carries a relative
// error <= 2^-11 (normal range) or an absolute one <= 2^-25 (below 2^-14), products of float16 values are exact
// in float32, the MFMA's float32 accumulation of d <= 1024 terms is within d * 2^-23 of the exact sum, and
// the exact path's own fmaf chain within d * 2^-24:  |approx - exact| <= 1.3e-3 * approx + 2e-6.
//
// Reference: README.md:107-113, 137-142 (n_neighbors_ann nearest neighbours, then the precursor filter keeps
// n_neighbors); no code in the snapshot.
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "scan.h"
#include "select.h"
#include "ivf.h"
#include "fused.h"

namespace fal {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr float kEpsRel = 1.3e-3f;
constexpr float kEpsAbs = 2e-6f;
constexpr int kMemCap = 64;          // members (candidates inside the threshold bin) kept per query
constexpr int kKeepCap = 48;         // selected window candidates kept per query
constexpr int kHistStride = 516;     // bytes per query: 256 x u16 + 4 (lane = query reads stay conflict-free)
constexpr int kWaveHist = 32 * kHistStride;                     // 16,512 B: histogram, later the member lists
static_assert(32 * kMemCap * 8 <= kWaveHist, "member lists alias the histograms");
constexpr int kWaveSmall = 2048;     // per-wave per-query scalars + the sort staging of phase E
constexpr int kKeptBytes = 4 * 32 * kKeepCap * 8;               // kept lists of the four waves (alias the staging buffers)
__host__ __device__ constexpr int region_a_bytes(int d) { return 2 * 32 * (d * 2 + 16) > kKeptBytes ? 2 * 32 * (d * 2 + 16) : kKeptBytes; }

__device__ __forceinline__ int rowoff16(int i) { return (i & 3) + 8 * (i >> 2); }

// the exact similarity on the vector ALU: the k-ordered fmaf chain of simtile.h, bit for bit
__device__ __forceinline__ float exact_dot(const float* __restrict__ a, const float* __restrict__ b, int d) {
    const int dh4 = d >> 3;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float acc = 0.f;
    for (int j = 0; j < dh4; ++j) {
        const float4 al = a4[j], ah = a4[dh4 + j], bl = b4[j], bh = b4[dh4 + j];
        acc = __builtin_fmaf(al.x, bl.x, acc);
        acc = __builtin_fmaf(ah.x, bh.x, acc);
        acc = __builtin_fmaf(al.y, bl.y, acc);
        acc = __builtin_fmaf(ah.y, bh.y, acc);
        acc = __builtin_fmaf(al.z, bl.z, acc);
        acc = __builtin_fmaf(ah.z, bh.z, acc);
        acc = __builtin_fmaf(al.w, bl.w, acc);
        acc = __builtin_fmaf(ah.w, bh.w, acc);
    }
    return acc;
}

__device__ __forceinline__ bool find_job_xcd128f(const DenseJob* __restrict__ jobs, int n_jobs, unsigned bid,
                                                 int* job_index, int* local_tile) {
    const int x = bid & 7;
    const int64_t i = bid >> 3;
    const int cnt = (n_jobs - x + 7) >> 3;
    if (cnt <= 0) return false;
    int lo = 0, hi = cnt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[x + 8 * mid].xtile0 <= i) lo = mid; else hi = mid - 1;
    }
    const DenseJob& j = jobs[x + 8 * lo];
    const int64_t lt = i - j.xtile0;
    if (lt >= (j.nq + 127) / 128) return false;
    *job_index = x + 8 * lo;
    *local_tile = (int)lt;
    return true;
}

__device__ __forceinline__ void push_fallback(const FusedArgs& a, int64_t row, int job) {
    const int at = atomicAdd(a.fb_count, 1);
    if (at < a.fb_cap) {
        a.fb_list[2 * at] = (int32_t)row;
        a.fb_list[2 * at + 1] = job;
    }
}

template <int STEPS>
__global__ __launch_bounds__(256, 1) void fused_kernel(FusedArgs a) {
    constexpr int D = STEPS * 16, DH = D / 2, DH4 = D / 8;
    constexpr int RB16 = D / 8;                     // 16-byte pieces per float16 row
    constexpr int RS = D * 2 + 16;                  // LDS row stride (padded: conflict-free b128 reads)
    constexpr int PIECES = 32 * RB16;
    constexpr int kStage = (PIECES + 255) / 256;
    constexpr int NB = STEPS < 4 ? STEPS : 4;
    constexpr int kStageBytes = region_a_bytes(D);   // staging buffers, later the kept lists
    extern __shared__ __align__(16) unsigned char lds[];
    int ji, T;
    if (!find_job_xcd128f(a.jobs, a.n_jobs, blockIdx.x, &ji, &T)) return;
    const DenseJob job = a.jobs[ji];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int nc = job.nc, k = a.k;
    const int64_t row0 = job.q_row0;
    const int tile32 = 4 * T + w;
    const int nqw = min(32, nc - 32 * tile32);      // <= 0: the wave only helps with staging
    const bool active = nqw > 0;
    const bool need_thr = nc > k;                   // otherwise every candidate is among the k best
    const int qbase = 32 * tile32;                  // first query of this wave inside the bucket

    unsigned char* whist = lds + kStageBytes + w * kWaveHist;                  // histogram / member lists
    unsigned char* wsmall = lds + kStageBytes + 4 * kWaveHist + w * kWaveSmall;
    float* q_lo = reinterpret_cast<float*>(wsmall);          // [32] member interval
    float* q_hi = q_lo + 32;
    float* q_L = q_hi + 32;                                  // [32] exact k-th value lies in [L, U]
    float* q_U = q_L + 32;
    float* q_T = q_U + 32;                                   // [32] k-th best approximate value
    float* q_eps = q_T + 32;
    int* q_bstar = reinterpret_cast<int*>(q_eps + 32);       // [32]
    int* q_nabove = q_bstar + 32;
    int* q_mcnt = q_nabove + 32;
    int* q_kcnt = q_mcnt + 32;
    int* q_flag = q_kcnt + 32;                               // bit 0: ambiguous candidate present, bit 1: fallback
    uint32_t* s_u = reinterpret_cast<uint32_t*>(q_flag + 32);   // [64] sort staging
    uint32_t* s_lo = s_u + 64;
    static_assert((11 * 32 + 128) * 4 <= kWaveSmall, "per-wave scalars");
    uint32_t* mem_id = reinterpret_cast<uint32_t*>(whist);              // [32][kMemCap]
    float* mem_v = reinterpret_cast<float*>(whist + 32 * kMemCap * 4);  // [32][kMemCap]
    uint32_t* kept_u = reinterpret_cast<uint32_t*>(lds + w * (kKeptBytes / 4));    // [32][kKeepCap] (after the passes)
    uint32_t* kept_id = kept_u + 32 * kKeepCap;

    if (lane < 32) {
        q_L[lane] = -INFINITY;
        q_U[lane] = -INFINITY;
        q_flag[lane] = 0;
        q_mcnt[lane] = 0;
        q_kcnt[lane] = 0;
    }

    const int dbg = a.dbg;
    if (need_thr && !(dbg & 32)) {
        // ================= approximate passes on the f16 matrix cores ==============================================
        const __half* X16 = a.X16;
        half8 q[STEPS];
        {
            const int64_t qrow = row0 + (active ? qbase + min(r, nqw - 1) : 0);
            const half8* src = reinterpret_cast<const half8*>(X16 + qrow * D + h * DH);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) q[s] = src[s];
        }
        for (int e = lane; e < kWaveHist / 16; e += 64) reinterpret_cast<uint4*>(whist)[e] = make_uint4(0, 0, 0, 0);
        const __half* cbase = X16 + row0 * (int64_t)D;
        uint4 st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11, st12;
#define FAL_FOR_STAGE(M) M(0, st0) M(1, st1) M(2, st2) M(3, st3) M(4, st4) M(5, st5) M(6, st6) M(7, st7) M(8, st8) \
    M(9, st9) M(10, st10) M(11, st11) M(12, st12)
        static_assert(kStage <= 13, "staging registers");
#define FAL_LOAD_ONE(I, R)                                                                             \
    if constexpr (I < kStage) {                                                                        \
        const int idx = min((int)threadIdx.x + 256 * I, PIECES - 1);                                   \
        const int row = idx / RB16, col = idx - row * RB16;                                            \
        R = reinterpret_cast<const uint4*>(cbase + (int64_t)min(stage_c0 + row, nc - 1) * D)[col];     \
    }
#define FAL_STORE_ONE(I, R)                                                                            \
    if constexpr (I < kStage) {                                                                        \
        const int idx = min((int)threadIdx.x + 256 * I, PIECES - 1);                                   \
        const int row = idx / RB16, col = idx - row * RB16;                                            \
        *reinterpret_cast<uint4*>(lds + (size_t)stage_buf * 32 * RS + row * RS + col * 16) = R;        \
    }
#define FAL_STAGE_LOAD(C0) { const int stage_c0 = (C0); FAL_FOR_STAGE(FAL_LOAD_ONE) }
#define FAL_STAGE_STORE(BUF) { const int stage_buf = (BUF); FAL_FOR_STAGE(FAL_STORE_ONE) }
        // one pass over all candidate chunks; EPI(acc, c0) consumes the 32x32 block D[query][candidate]
#define FAL_PASS(EPI)                                                                                   \
    {                                                                                                   \
        __syncthreads();                                                                                \
        FAL_STAGE_LOAD(0)                                                                               \
        FAL_STAGE_STORE(0)                                                                              \
        __syncthreads();                                                                                \
        int buf = 0;                                                                                    \
        for (int c0 = 0; c0 < nc; c0 += 32) {                                                           \
            FAL_STAGE_LOAD(min(c0 + 32, nc - 1))                                                        \
            __builtin_amdgcn_sched_group_barrier(0x020, kStage, 0);                                     \
            const unsigned char* rowp = lds + (size_t)buf * 32 * RS + r * RS + h * DH * 2;              \
            half8 rh[NB];                                                                               \
            _Pragma("unroll") for (int s = 0; s < NB; ++s) rh[s] = *reinterpret_cast<const half8*>(rowp + s * 16); \
            f32x16 acc;                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[i] = 0.f;                                \
            _Pragma("unroll") for (int s = 0; s < STEPS; ++s) {                                         \
                const half8 ch = rh[s % NB];                                                            \
                if (s + NB < STEPS) rh[s % NB] = *reinterpret_cast<const half8*>(rowp + (s + NB) * 16); \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(q[s], ch, acc, 0, 0, 0);                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            }                                                                                           \
            EPI                                                                                         \
            FAL_STAGE_STORE(buf ^ 1)                                                                    \
            __syncthreads();                                                                            \
            buf ^= 1;                                                                                   \
        }                                                                                               \
    }

        // ---- pass 1: histogram of the approximate similarities (bin = floor(256 v), 255 = everything above) ----
        unsigned char* hbase = whist + (4 * h) * kHistStride;
        FAL_PASS({
            const bool cval = c0 + r < nc && !(dbg & 1);
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {
                const float v = fmaxf(acc[i], 0.f);
                const uint32_t b = min(255u, (uint32_t)(v * 256.f));
                if (cval)
                    atomicAdd(reinterpret_cast<unsigned*>(hbase + rowoff16(i) * kHistStride + ((b >> 1) << 2)),
                              (b & 1) ? 0x10000u : 1u);
            }
        })
        // ---- the bin of the k-th best approximate value: suffix sums from the top, lane = query ------------------
        {
            const unsigned* hrow = reinterpret_cast<const unsigned*>(whist + r * kHistStride);
            int cum = 0, bstar = -1, nabove = 0;
            for (int j = 127; j >= 0; --j) {
                const unsigned wv = hrow[j];
                const int chi = (int)(wv >> 16), clo = (int)(wv & 0xFFFFu);
                if (bstar < 0) {
                    if (cum + chi >= k) {
                        bstar = 2 * j + 1;
                        nabove = cum;
                    } else if (cum + chi + clo >= k) {
                        bstar = 2 * j;
                        nabove = cum + chi;
                    }
                    cum += chi + clo;
                }
                if (__ballot(bstar < 0) == 0ull) break;
            }
            if (bstar < 0) bstar = 0;                       // (cannot happen: the row holds nc > k values)
            const float binlo = (float)bstar * (1.f / 256.f);
            const float binhi = bstar == 255 ? 1.0625f : (float)(bstar + 1) * (1.f / 256.f);
            const float e0 = kEpsRel * binhi + kEpsAbs;
            const float m = 2.5f * e0;
            if (lane < 32) {
                q_lo[r] = binlo - m;
                q_hi[r] = bstar == 255 ? INFINITY : binhi + m;
                q_eps[r] = kEpsRel * (binhi + m) + kEpsAbs;  // bounds |approx - exact| of every value <= binhi + m
                q_bstar[r] = bstar;
                q_nabove[r] = nabove;
            }
        }
        __syncthreads();
        // ---- pass 2: the members of the (widened) threshold bin -> LDS lists (the histograms' memory) -------------
        float lo_r[16], hi_r[16];
        int cnt[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            lo_r[i] = q_lo[4 * h + rowoff16(i)];
            hi_r[i] = q_hi[4 * h + rowoff16(i)];
            cnt[i] = 0;
        }
        const uint32_t below = (1u << r) - 1u;
        FAL_PASS({
            const bool cval = c0 + r < nc && !(dbg & 2);
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {
                const float v = fmaxf(acc[i], 0.f);
                const bool hit = cval && v >= lo_r[i] && v <= hi_r[i];
                const unsigned long long mk = __ballot(hit);
                if (mk) {
                    const uint32_t m32 = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                    const int pos = cnt[i] + __popc(m32 & below);
                    if (hit && pos < kMemCap) {
                        const int ql = 4 * h + rowoff16(i);
                        mem_id[ql * kMemCap + pos] = (uint32_t)(c0 + r);
                        mem_v[ql * kMemCap + pos] = v;
                    }
                    cnt[i] += __popc(m32);
                }
            }
        })
#undef FAL_PASS
#undef FAL_STAGE_LOAD
#undef FAL_STAGE_STORE
#undef FAL_LOAD_ONE
#undef FAL_STORE_ONE
#undef FAL_FOR_STAGE
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) q_mcnt[4 * h + rowoff16(i)] = cnt[i];
        }
        __syncthreads();
        // ---- T~ = the (k - n_above)-th best approximate value inside bin b*; exact k-th value in [T~ - eps, T~ + eps] --
        for (int ql = 0; ql < ((dbg & 4) ? 0 : 32); ++ql) {
            const int mc = q_mcnt[ql];
            if (mc > kMemCap) {                              // too many values share the bin: exact fallback
                if (lane == 0) q_flag[ql] = 2;
                continue;
            }
            const int bstar = q_bstar[ql], need = k - q_nabove[ql];
            const float v = lane < mc ? mem_v[ql * kMemCap + lane] : 0.f;
            const bool inbin = lane < mc && (int)min(255u, (uint32_t)(v * 256.f)) == bstar;
            const uint32_t u = inbin ? max(__float_as_uint(v), 1u) : 0u;     // non-negative floats: bit order = value order
            const int have = __popcll(__ballot(inbin));
            if (have < need || need < 1) {                   // (defensive: cannot happen when the lists are complete)
                if (lane == 0) q_flag[ql] = 2;
                continue;
            }
            uint32_t Tb = 0;
            for (int bit = 30; bit >= 0; --bit) {
                const uint32_t c = Tb | (1u << bit);
                const int n = __popcll(__ballot(u >= c));
                if (n >= need) Tb = c;
                if (n == need) break;
            }
            uint32_t tv = u >= Tb ? u : 0xFFFFFFFFu;         // smallest member at or above the threshold = the need-th best
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) tv = min(tv, (uint32_t)__shfl_xor((int)tv, off, 64));
            if (lane == 0) {
                const float Tv = __uint_as_float(tv == 1u ? 0u : tv);
                const float e = q_eps[ql];
                q_T[ql] = Tv;
                q_L[ql] = Tv - e;
                q_U[ql] = Tv + e;
            }
        }
    }
    __syncthreads();                 // the staging buffers become the kept lists; the per-query scalars are final

    // ================= exact similarities of the precursor window on the fp32 matrix cores ============================
    const int dh4 = DH4;
    if (active) {
        const float* X = a.X;
        float qf[DH4 * 4];
        load_half_row<DH4>(qf, X + (row0 + qbase + min(r, nqw - 1)) * D + (int64_t)h * DH, dh4);
        const float* pm = a.pmz + row0;
        const float* rtp = a.rt ? a.rt + row0 : nullptr;
        const bool use_rt = rtp != nullptr && a.rt_tol >= 0.0;
        // candidate range that can pass the tolerance for ANY query of the tile (slightly widened; the exact test
        // decides below): rows are sorted by precursor m/z
        int wlo, whi;
        {
            const double qf_first = (double)pm[qbase], qf_last = (double)pm[qbase + nqw - 1];
            double lob, hib;
            if (a.is_da) {
                lob = qf_first - a.tol - 1e-3;
                hib = qf_last + a.tol + 1e-3;
            } else {
                const double t = a.tol * 1e-6;
                lob = qf_first * (1.0 - 1.01 * t - 2e-6);
                hib = t < 0.5 ? qf_last * (1.0 + 1.01 * t / (1.0 - t) + 2e-6) : INFINITY;
            }
            int lo = 0, hi = nc;                             // first c with pm[c] >= lob
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((double)pm[mid] < lob) lo = mid + 1; else hi = mid;
            }
            wlo = lo;
            lo = wlo;
            hi = nc;                                         // first c with pm[c] > hib
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((double)pm[mid] <= hib) lo = mid + 1; else hi = mid;
            }
            whi = lo;
        }
        float qmz[16], qrt[16], Lr[16], Ur[16];
        int kc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ql = 4 * h + rowoff16(i);
            qmz[i] = pm[qbase + min(ql, nqw - 1)];
            qrt[i] = use_rt ? rtp[qbase + min(ql, nqw - 1)] : 0.f;
            Lr[i] = q_L[ql];
            Ur[i] = q_U[ql];
            kc[i] = 0;
        }
        const uint32_t below = (1u << r) - 1u;
        const int n_chunks = (dbg & 8) ? 0 : (whi - wlo + 31) >> 5;
        CandStream<DH4> cs;
        auto crow = [&](int c0) -> const float* { return X + (row0 + min(c0 + r, nc - 1)) * D + (int64_t)h * DH; };
        const float* cur = crow(wlo);
        cs.prime(cur, dh4);
        f32x16 prev;
        int prev_c0 = -1;                                    // -1: nothing to consume yet
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[i] = 0.f;
        auto epilogue = [&]() {
            if (prev_c0 < 0) return;                         // wave-uniform
            const int c = prev_c0 + r;
            const bool cval = c < whi;
            const int cc = min(c, nc - 1);
            const float nmz = pm[cc];
            const float nrt = use_rt ? rtp[cc] : 0.f;
            const uint32_t cid = (uint32_t)(row0 + cc);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ql = 4 * h + rowoff16(i);
                const float s = prev[i];
                const float diff = qmz[i] - nmz;             // mass_diff(query, neighbour): the arithmetic of filter_kernel
                const double md = a.is_da ? (double)diff : (double)(diff / nmz) * 1e6;
                bool ok = fabs(md) <= a.tol;
                if (use_rt) ok = ok && fabs((double)(qrt[i] - nrt)) <= a.rt_tol;
                ok = ok && cval && ql < nqw && cc != qbase + ql;
                const bool sel = ok && s >= Lr[i];           // below L: certainly not among the k best
                const bool amb = sel && s <= Ur[i];          // inside [L, U]: decided exactly in phase E
                const unsigned long long mk = __ballot(sel);
                if (mk) {
                    const uint32_t m32 = h ? (uint32_t)(mk >> 32) : (uint32_t)mk;
                    const int pos = kc[i] + __popc(m32 & below);
                    if (sel && pos < kKeepCap) {
                        kept_u[ql * kKeepCap + pos] = max(f32_sortable(s), 1u);
                        kept_id[ql * kKeepCap + pos] = cid | (amb ? 0x80000000u : 0u);
                    }
                    kc[i] += __popc(m32);
                    if (amb) atomicOr(&q_flag[ql], 1);
                }
            }
        };
        for (int ci = 0, c0 = wlo; ci < n_chunks; ++ci, c0 += 32) {
            const float* nxt = crow(c0 + 32);
            const f32x16 acc = cs.template dot<true>(qf, cur, nxt, dh4, epilogue);
            prev = acc;
            prev_c0 = c0;
            cur = nxt;
        }
        epilogue();
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) q_kcnt[4 * h + rowoff16(i)] = kc[i];
        }
    }
    __syncthreads();

    // ================= per query: exact resolution of ambiguous candidates, sort, neighbour lists =====================
    if (!active || (dbg & 16)) return;
    for (int ql = 0; ql < nqw; ++ql) {
        const int64_t row = row0 + qbase + ql;
        int flag = q_flag[ql];
        const int kcn = q_kcnt[ql];
        if (kcn > kKeepCap) flag |= 2;
        uint32_t uT = 0, iT = 0xFFFFFFFFu;                  // exact k-th key (only when an ambiguous candidate exists)
        if (!(flag & 2) && (flag & 1)) {
            const int mc = q_mcnt[ql];
            const float Tv = q_T[ql], e = q_eps[ql];
            const int bstar = q_bstar[ql];
            const bool have = lane < mc;
            const float v = have ? mem_v[ql * kMemCap + lane] : 0.f;
            const uint32_t mid = have ? mem_id[ql * kMemCap + lane] : 0u;
            const bool inE = have && fabsf(v - Tv) <= 2.f * e;
            const int n_bin_above = __popcll(__ballot(have && (int)min(255u, (uint32_t)(v * 256.f)) > bstar));
            const int n_hi = __popcll(__ballot(have && v > Tv + 2.f * e));
            const int nH = q_nabove[ql] - n_bin_above + n_hi;          // candidates certainly above the k-th value
            const int need = k - nH;
            const unsigned long long em = __ballot(inE);
            if (need < 1 || need > __popcll(em)) {
                flag |= 2;
            } else {
                float s = 0.f;
                if (inE) s = exact_dot(a.X + row * D, a.X + (row0 + mid) * D, D);
                const uint32_t u = inE ? max(f32_sortable(s), 1u) : 0u;
                const uint32_t id = (uint32_t)(row0 + mid);
                int rank = 0;                                // members of E with a better key
                unsigned long long rest = em;
                while (rest) {
                    const int j = __ffsll((unsigned long long)rest) - 1;
                    rest &= rest - 1;
                    const uint32_t uj = (uint32_t)__shfl((int)u, j, 64), idj = (uint32_t)__shfl((int)id, j, 64);
                    if (uj > u || (uj == u && idj < id)) ++rank;
                }
                const unsigned long long pick = __ballot(inE && rank == need - 1);
                const int src = __ffsll((unsigned long long)pick) - 1;
                uT = (uint32_t)__shfl((int)u, src, 64);
                iT = (uint32_t)__shfl((int)id, src, 64);
            }
        }
        if (flag & 2) {
            if (lane == 0 && !(dbg & 64)) push_fallback(a, row, ji);
            continue;
        }
        // compact the selected candidates (ambiguous ones against the exact k-th key) and sort them
        bool keepit = false;
        uint32_t u = 0, id = 0;
        if (lane < kcn) {
            u = kept_u[ql * kKeepCap + lane];
            id = kept_id[ql * kKeepCap + lane];
            keepit = true;
            if (id & 0x80000000u) {
                id &= 0x7FFFFFFFu;
                keepit = u > uT || (u == uT && id <= iT);
            }
        }
        const unsigned long long km = __ballot(keepit);
        if (keepit) {
            const int at = __popcll(km & ((1ull << lane) - 1ull));
            s_u[at] = u;
            s_lo[at] = ~id;
        }
        const int c = __popcll(km);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (a.nb_count && lane == 0) a.nb_count[row] = min(c, a.keep);
        sort_and_store_nb<1>(s_u, s_lo, c, a.keep, lane, a.nb_idx + row * a.keep, a.nb_dist + row * a.keep);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------------
// exact fallback: one wave per listed query.  The whole similarity row by VALU fmaf chains (bit-identical to the
// matrix-core chain) into a private scratch row, then the staged path's selection + filter + sort.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fused_fallback_kernel(FusedArgs a, SelectArgs sa, float* __restrict__ scratch,
                                                            int64_t scratch_stride, int d) {
    __shared__ uint32_t sel_u[kSelBuf];
    __shared__ uint32_t sel_id[kSelBuf];
    const int lane = threadIdx.x;
    const int total = min(*a.fb_count, a.fb_cap);
    float* row_s = scratch + (int64_t)blockIdx.x * scratch_stride;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t row = a.fb_list[2 * t];
        const DenseJob job = a.jobs[a.fb_list[2 * t + 1]];
        const int nc = job.nc;
        const float* qp = a.X + row * d;
        for (int c = lane; c < nc; c += 64) row_s[c] = exact_dot(qp, a.X + (job.c_row0 + c) * d, d);
        __threadfence();                                     // the row is re-read through this CU's L1
        __syncthreads();
        SelQuery qy{row_s, nc, job.c_row0};
        int carry;
        if (nc <= 128) carry = select_rounds<MODE_DENSE, 2>(sa, qy, sa.k, lane, sel_u, sel_id, nullptr, nullptr);
        else if (nc <= 256) carry = select_rounds<MODE_DENSE, 4>(sa, qy, sa.k, lane, sel_u, sel_id, nullptr, nullptr);
        else if (nc <= 512) carry = select_rounds<MODE_DENSE, 8>(sa, qy, sa.k, lane, sel_u, sel_id, nullptr, nullptr);
        else carry = select_rounds<MODE_DENSE, 16>(sa, qy, sa.k, lane, sel_u, sel_id, nullptr, nullptr);
        filter_sort_store(sa, sel_u, sel_id, sel_u + FAL_MAX_K_ANN, sel_id + FAL_MAX_K_ANN, carry, row, lane);
        __syncthreads();
    }
}

int launch_fused(fal_ctx* ctx, const FusedArgs& a_in, int d, int64_t list_tiles, int max_nc) {
    if (a_in.n_jobs <= 0 || list_tiles <= 0) return FAL_OK;
    FusedArgs a = a_in;
    {
        const char* e = getenv("FALCON_FUSED_DBG");
        a.dbg = e ? atoi(e) : 0;
    }
    FAL_REQUIRE(list_tiles * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    FAL_REQUIRE(max_nc < 65536, FAL_EUNSUPPORTED, "fused scan: buckets must hold fewer than 65,536 rows");
    const int steps = d / 16;
    // fallback list + its scratch rows
    const int fb_grid = ctx->num_cus * 8;
    const int64_t stride = (((int64_t)max_nc + 63) & ~63ll) + (int64_t)kSimsSlack;
    int32_t* fb = nullptr;
    float* scratch = nullptr;
    const int fb_cap = 1 << 22;
    FAL_TRY(ctx->reserve(SLOT_FUSED, sizeof(int32_t) * (size_t)(2 * fb_cap + 16), (void**)&fb));
    FAL_TRY(ctx->reserve(SLOT_FUSED2, sizeof(float) * (size_t)fb_grid * (size_t)stride, (void**)&scratch));
    a.fb_count = fb;
    a.fb_list = fb + 16;
    a.fb_cap = fb_cap;
    FAL_CHECK_HIP(hipMemsetAsync(fb, 0, sizeof(int32_t) * 16, ctx->stream));
    const size_t lds = (size_t)region_a_bytes(d) + 4 * kWaveHist + 4 * kWaveSmall;
    dim3 grid((unsigned)(list_tiles * 8)), block(256);
    {
        StageScope ts(ctx, ST_SCAN);
#define FAL_LAUNCH_FUSED(S)                                                                                       \
    do {                                                                                                          \
        FAL_CHECK_HIP(hipFuncSetAttribute((const void*)fused_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((fused_kernel<S>), grid, block, lds, ctx->stream, a);                                  \
    } while (0)
        switch (steps) {
            case 4: FAL_LAUNCH_FUSED(4); break;
            case 8: FAL_LAUNCH_FUSED(8); break;
            case 16: FAL_LAUNCH_FUSED(16); break;
            case 25: FAL_LAUNCH_FUSED(25); break;
            default:
                set_error("fused scan: low_dim %d has no instantiation (64, 128, 256, 400)", d);
                return FAL_EUNSUPPORTED;
        }
#undef FAL_LAUNCH_FUSED
        FAL_CHECK_HIP(hipGetLastError());
    }
    {
        StageScope ts(ctx, ST_SELECT);
        SelectArgs sa{};
        sa.k = a.k;
        sa.f_pmz = a.pmz; sa.f_rt = a.rt; sa.f_tol = a.tol; sa.f_rt_tol = a.rt_tol; sa.f_is_da = a.is_da;
        sa.f_keep = a.keep; sa.nb_idx = a.nb_idx; sa.nb_dist = a.nb_dist; sa.nb_count = a.nb_count;
        hipLaunchKernelGGL(fused_fallback_kernel, dim3((unsigned)fb_grid), dim3(64), 0, ctx->stream, a, sa, scratch, stride, d);
        FAL_CHECK_HIP(hipGetLastError());
    }
    // the number of fallback queries of this call: fal_ctx_counter(5) after a sync (pinned target: truly asynchronous)
    if (!ctx->fb_host) FAL_CHECK_HIP(hipHostMalloc((void**)&ctx->fb_host, 64, hipHostMallocDefault));
    FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host, fb, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return FAL_OK;
}

bool fused_supports(int d) { return d == 64 || d == 128 || d == 256 || d == 400; }

}  // namespace fal
