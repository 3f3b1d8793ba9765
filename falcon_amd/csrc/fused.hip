// a7 + a8 for FLAT buckets with the top-k kept on chip: no [queries, candidates] similarity matrix in HBM.
//
// What the staged path does (scan.hip): fp32-MFMA similarity of every (query, candidate) pair of the bucket ->
// [32, nc] block in HBM -> select kernel: k_ann best per query -> precursor / RT filter -> sort -> neighbour
// lists.  The result only needs (i) the exact similarities of the candidates inside the query's precursor
// window -- a contiguous band next to the diagonal, because the bucket's rows are sorted by precursor m/z -- and
// (ii) the k_ann-th best key of the row, to decide which of them made the top-k_ann.  Three kernels:
//
//   approx_kernel  (f16 matrix cores; buckets with more than k_ann rows)  scans the whole bucket twice on float16
//      copies of the rows (float32 accumulation, 1/16 of the fp32 matrix-pipe cycles).  Pass 1 builds a 256-bin
//      histogram of the APPROXIMATE similarities of every query in LDS (fine bins below 0.25 where the k-th value of
//      spectra lies, coarse ones above); a suffix sum gives the bin b* of the k_ann-th best approximate value.  Pass 2
//      collects the candidates whose approximate value lies in (a slightly widened) b* -- the "members", a handful per
//      query -- and selects the k_ann-th best approximate value T~ among them.  |approx - exact| <= eps(value) (bound
//      below) puts the exact k_ann-th best value inside [L, U] = [T~ - eps, T~ + eps].  Out: 32 B of thresholds and the
//      member list per query.
//   band_kernel  (fp32 matrix cores)  computes the EXACT similarities of the precursor window (the same k-ordered fmaf
//      chain as dense_kernel: bit-identical values), applies the precursor / RT tolerance and keeps the candidates
//      that are not certainly outside the top-k_ann (s >= L); those inside [L, U] are marked ambiguous.
//   resolve_kernel  (one wave per query, full occupancy)  decides ambiguous candidates exactly -- the few members within
//      2 eps of T~ are re-evaluated by VALU fmaf chains (the same bits), the true k_ann-th key (value, id) follows by
//      counting --, sorts the survivors and writes the neighbour lists.
//
// The hand-off between the kernels is ~0.9 kB per query (thresholds, members, kept candidates) instead of 4 B per
// (query, candidate) pair.  The output is BIT-IDENTICAL to fal_ivf_search_topk -> fal_filter_neighbors
// (tests/test_gpu_fused.py).  Queries the on-chip structures cannot hold (more than 32 members per lane half -- e.g.
// hundreds of identical spectra --, more than 24 kept window candidates per half) go to fused_fallback_kernel: exact
// row by VALU fmaf chains + the staged path's own selection code.
//
// Error bound of the prefilter (non-negative unit vectors x, y; ^ = rounded to float16): each factor carries a relative
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
#include <type_traits>
#include "fused.h"

namespace fal {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr float kEpsRel = 1.3e-3f;
constexpr float kEpsAbs = 2e-6f;
constexpr int kBins = 160;           // 128 fine bins below 0.25 + 32 coarse ones
constexpr int kHistStride = kBins * 2 + 4;   // bytes per query: u16 counters + 4 (lane = query reads stay conflict-free)
constexpr int kWaveHist = 11008;     // >= 32 * kHistStride: histogram, later the member lists
constexpr int kWaveSmall = 1280;     // per-wave per-query scalars
constexpr int kMemHalf = FAL_FUSED_MEM / 2;      // members kept per (query, lane half)
constexpr int kMemSlot = kMemHalf + 1;           // ... plus the dump slot the branch-free append writes to when full
constexpr int kMemStride = 43;                   // dwords per query row (odd: lane = query accesses are conflict-free)
constexpr int kKeepHalf = FAL_FUSED_KEEP / 2;    // selected window candidates kept per (query, lane half)
static_assert(2 * 32 * kMemStride * 4 <= kWaveHist && 32 * kHistStride <= kWaveHist && 2 * kMemSlot <= kMemStride, "member lists alias the histograms");

__device__ __forceinline__ int rowoff16(int i) { return (i & 3) + 8 * (i >> 2); }

// histogram bin of an approximate similarity: 128 bins of 1/512 below 0.25 (where the k-th best value of hashed spectra
// lies: a handful of members per bin), 32 bins of 3/128 above
__device__ __forceinline__ uint32_t bin_of(float v) {
    const uint32_t lo = (uint32_t)(v * 512.f);
    const uint32_t hi = min((uint32_t)kBins - 1u, 128u + (uint32_t)(fmaxf(v - 0.25f, 0.f) * (128.f / 3.f)));
    return v < 0.25f ? lo : hi;
}
__device__ __forceinline__ float bin_lo(int b) { return b < 128 ? (float)b * (1.f / 512.f) : 0.25f + (float)(b - 128) * (3.f / 128.f); }
__device__ __forceinline__ float bin_hi(int b) {
    return b < 128 ? (float)(b + 1) * (1.f / 512.f) : (b == kBins - 1 ? 1.0625f : 0.25f + (float)(b - 127) * (3.f / 128.f));
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


// ------------------------------------------------------------------------------------------------------------
// approx_kernel: workgroup = 4 waves = 128 queries of one bucket; lane (r, h) serves query r of its wave throughout --
// MFMA results are D[candidate][query] (column = lane & 31), so a lane's 16 accumulator registers hold 16 candidates of
// ITS query and every per-query structure is lane-private (the two halves of a query keep separate sub-lists)
// ------------------------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(256, 2) void approx_kernel(FusedArgs a) {
    constexpr int D = STEPS * 16, DH = D / 2;
    constexpr int RB16 = D / 8;                     // 16-byte pieces per float16 row
    constexpr int RS = D * 2 + 16;                  // LDS row stride (padded: conflict-free b128 reads)
    constexpr int PIECES = 32 * RB16;
    constexpr int kStage = (PIECES + 255) / 256;
    constexpr int NB = STEPS < 4 ? STEPS : 4;       // LDS operand reads in flight ahead of the MFMAs (two waves per SIMD)
    constexpr int kStageBytes = 32 * RS;            // ONE staging buffer: two workgroups share a CU and fill each other's gaps
    static_assert(kStage <= 7, "staging registers");
    extern __shared__ __align__(16) unsigned char lds[];
    int ji, T;
    if (!find_job_xcd128f(a.jobs128, a.n_jobs128, blockIdx.x, &ji, &T)) return;
    const DenseJob job = a.jobs128[ji];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int nc = job.nc, k = a.k;
    const int64_t row0 = job.q_row0;
    const int tile32 = 4 * T + w;
    const int nqw = min(32, nc - 32 * tile32);      // <= 0: the wave only helps with staging
    const bool active = nqw > 0;
    const int qbase = 32 * tile32;                  // first query of this wave inside the bucket

    unsigned char* whist = lds + kStageBytes + w * kWaveHist;                  // histogram, then the member lists
    unsigned char* wsmall = lds + kStageBytes + 4 * kWaveHist + w * kWaveSmall;
    float* q_lo = reinterpret_cast<float*>(wsmall);          // [32] member interval
    float* q_hi = q_lo + 32;
    float* q_eps = q_hi + 32;
    float* q_T = q_eps + 32;                                 // [32] k-th best approximate value
    int* q_bstar = reinterpret_cast<int*>(q_T + 32);         // [32]
    int* q_nabove = q_bstar + 32;
    int* q_mcnt = q_nabove + 32;                             // [64] members per (query, half)
    int* q_flag = q_mcnt + 64;                               // [64] bit 1: fallback
    static_assert((6 * 32 + 2 * 64) * 4 <= kWaveSmall, "per-wave scalars");
    float* mem_v = reinterpret_cast<float*>(whist);                            // [32][kMemStride]: half h at [h * kMemSlot ...)
    uint32_t* mem_id = reinterpret_cast<uint32_t*>(whist + 32 * kMemStride * 4);
    q_flag[lane] = 0;
    q_mcnt[lane] = 0;
    if (lane < 32) q_T[lane] = 0.f;

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
    // candidate chunks travel global -> registers -> LDS two chunks ahead of their use: two register sets
    uint4 sa0, sa1, sa2, sa3, sa4, sa5, sa6, sb0, sb1, sb2, sb3, sb4, sb5, sb6;
#define FAL_FOR_A(M) M(0, sa0) M(1, sa1) M(2, sa2) M(3, sa3) M(4, sa4) M(5, sa5) M(6, sa6)
#define FAL_FOR_B(M) M(0, sb0) M(1, sb1) M(2, sb2) M(3, sb3) M(4, sb4) M(5, sb5) M(6, sb6)
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
#define FAL_LOAD(SET, C0) { const int stage_c0 = min((C0), nc - 1); FAL_FOR_##SET(FAL_LOAD_ONE) }
#define FAL_STORE(SET, BUF) { const int stage_buf = (BUF); FAL_FOR_##SET(FAL_STORE_ONE) }
    // One pass over all candidate chunks.  acc = D[candidate][query] of a chunk; its epilogue (16 values per lane)
    // runs one chunk LATER, spread over the MFMA steps of the next chunk: the matrix pipe holds the issue port for 8 of
    // its 32 cycles and the branch-free epilogue pieces fill the rest.  piece(i, v, c, valid) consumes value i.
    constexpr int kPiecesPerStep = (16 + STEPS - 1) / STEPS;
    auto run_pass = [&](auto&& piece) {
        __syncthreads();
        FAL_LOAD(A, 0)
        FAL_STORE(A, 0)
        FAL_LOAD(A, 32)
        FAL_LOAD(B, 64)
        __syncthreads();
        f32x16 prev;
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[i] = 0.f;
        int prev_c0 = nc;                                // "no previous chunk": every candidate index is invalid
        auto chunk = [&]() {
            const unsigned char* rowp = lds + r * RS + h * DH * 2;
            half8 rh[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) rh[s] = *reinterpret_cast<const half8*>(rowp + s * 16);
            // pin the whole operand ring in front of the first MFMA: without this group the scheduler satisfies the
            // "one read per step" pattern below by issuing the reads one at a time and every MFMA waits for an LDS
            // round trip
            __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const half8 ch = rh[s % NB];
                if (s + NB < STEPS) rh[s % NB] = *reinterpret_cast<const half8*>(rowp + (s + NB) * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, q[s], acc, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < kPiecesPerStep; ++t) {
                    const int i = s * kPiecesPerStep + t;
                    if (i < 16) {
                        const int c = prev_c0 + rowoff16(i) + 4 * h;
                        piece(i, fmaxf(prev[i], 0.f), c, c < nc);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // this step's MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // the operand read for step s + NB
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);    // a piece of the previous chunk's epilogue
                __builtin_amdgcn_sched_group_barrier(0x200, 2 * kPiecesPerStep, 0);
            }
            return acc;
        };
        // chunk cc0 is in LDS, register set A holds cc0 + 32, B holds cc0 + 64 (two chunks of loads in flight)
        for (int cc0 = 0; cc0 < nc; cc0 += 64) {
            {
                const f32x16 acc = chunk();
                prev = acc;
                prev_c0 = cc0;
                __syncthreads();                         // every wave is done reading the buffer
                FAL_STORE(A, 0)
                FAL_LOAD(A, cc0 + 96)
                __syncthreads();
            }
            if (cc0 + 32 >= nc) break;
            {
                const f32x16 acc = chunk();
                prev = acc;
                prev_c0 = cc0 + 32;
                __syncthreads();
                FAL_STORE(B, 0)
                FAL_LOAD(B, cc0 + 128)
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {                   // the last chunk's epilogue
            const int c = prev_c0 + rowoff16(i) + 4 * h;
            piece(i, fmaxf(prev[i], 0.f), c, c < nc);
        }
    };

    // ---- pass 1: histogram of the approximate similarities ---------------------------------------------------------
    unsigned char* hrow_b = whist + r * kHistStride;
    run_pass([&](int, float v, int, bool valid) {
        const uint32_t b = bin_of(v);
        const uint32_t inc = (b & 1) ? 0x10000u : 1u;
        atomicAdd(reinterpret_cast<unsigned*>(hrow_b + ((b >> 1) << 2)), valid ? inc : 0u);
    });
    // ---- the bin of the k-th best approximate value: suffix sums from the top, lane = query (reads in batches of 8:
    //      one wave per SIMD, every dependent LDS round trip is exposed) ------------------------------------------------
    {
        const unsigned* hrow = reinterpret_cast<const unsigned*>(whist + r * kHistStride);
        int cum = 0, bstar = -1, nabove = 0;
        for (int j0 = kBins / 2 - 1; j0 >= 0; j0 -= 8) {
            unsigned wv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) wv[t] = hrow[j0 - t];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = j0 - t;
                const int chi = (int)(wv[t] >> 16), clo = (int)(wv[t] & 0xFFFFu);
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
            }
        }
        if (bstar < 0) bstar = 0;                           // (cannot happen: the row holds nc > k values)
        const float binlo = bin_lo(bstar), binhi = bin_hi(bstar);
        const float e0 = kEpsRel * binhi + kEpsAbs;
        const float m = 2.5f * e0;
        if (lane < 32) {
            q_lo[r] = binlo - m;
            q_hi[r] = bstar == kBins - 1 ? INFINITY : binhi + m;
            q_eps[r] = kEpsRel * (binhi + m) + kEpsAbs;      // bounds |approx - exact| of every value <= binhi + m
            q_bstar[r] = bstar;
            q_nabove[r] = nabove;
        }
    }
    __syncthreads();
    // ---- pass 2: the members of the (widened) threshold bin -> lane-private LDS lists (the histograms' memory).
    //      Branch-free: every value is written at the list's end, the end only advances on a hit (slot kMemHalf
    //      is the dump slot of a full list) -----------------------------------------------------------------------------
    {
        const float lo_q = q_lo[r], hi_q = q_hi[r];
        int cnt = 0;
        float* mv = mem_v + r * kMemStride + h * kMemSlot;
        uint32_t* mi = mem_id + r * kMemStride + h * kMemSlot;
        run_pass([&](int, float v, int c, bool valid) {
            const bool hit = valid && v >= lo_q && v <= hi_q;
            const int at = min(cnt, kMemHalf);
            mv[at] = v;
            mi[at] = (uint32_t)c;
            cnt += hit ? 1 : 0;
        });
        q_mcnt[r * 2 + h] = cnt;
    }
#undef FAL_LOAD
#undef FAL_STORE
#undef FAL_LOAD_ONE
#undef FAL_STORE_ONE
#undef FAL_FOR_A
#undef FAL_FOR_B
    __syncthreads();
    // ---- T~ = the (k - n_above)-th best approximate value inside bin b*: lane (r, h) tries the members of ITS half as
    //      pivots and ranks each against all members of the query (a handful); exact k-th value in [T~ - eps, T~ + eps] ----
    {
        const int m0 = q_mcnt[r * 2], m1 = q_mcnt[r * 2 + 1];
        const int bstar = q_bstar[r], need = k - q_nabove[r];
        const float bhi = bin_hi(bstar);
        const float* mvq = mem_v + r * kMemStride;
        bool found = false;
        if (m0 > kMemHalf || m1 > kMemHalf || need < 1) {
            q_flag[r * 2 + h] = 2;                           // too many values share the bin: exact fallback
        } else {
            const int mine = h ? m1 : m0;
            // members with value >= bhi lie above the bin (margin): they count in n_above, not here
            auto better_cnt = [&](const float ve, const int e) -> int {
                int better = 0;                              // in-bin members with a larger value (ties: half 0 first, lower slot first)
                for (int j0 = 0; j0 < m0; j0 += 8) {
                    float vj[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) vj[t] = mvq[min(j0 + t, kMemHalf)];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int j = j0 + t;
                        const bool inb = j < m0 && bin_of(vj[t]) == (uint32_t)bstar;
                        better += (inb && (vj[t] > ve || (vj[t] == ve && (h == 1 || j < e)))) ? 1 : 0;
                    }
                }
                for (int j0 = 0; j0 < m1; j0 += 8) {
                    float vj[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) vj[t] = mvq[kMemSlot + min(j0 + t, kMemHalf)];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int j = j0 + t;
                        const bool inb = j < m1 && bin_of(vj[t]) == (uint32_t)bstar;
                        better += (inb && (vj[t] > ve || (vj[t] == ve && h == 1 && j < e))) ? 1 : 0;
                    }
                }
                return better;
            };
            (void)bhi;
            for (int e = 0; e < mine; ++e) {
                const float ve = mvq[h * kMemSlot + e];
                if (bin_of(ve) != (uint32_t)bstar) continue;
                if (better_cnt(ve, e) == need - 1) {
                    q_T[r] = ve;
                    found = true;
                }
            }
        }
        // exactly one half finds the pivot; if neither does the lists are incomplete (defensive): fallback
        const unsigned long long fm = __ballot(found);
        const bool any = ((fm >> r) & 1ull) || ((fm >> (r + 32)) & 1ull);
        if (!any) q_flag[r * 2 + h] = 2;
    }
    wave_lds_sync();
    // ---- hand-off: thresholds and member lists of the wave's queries --------------------------------------------------
    if (!active) return;
    if (lane < nqw) {
        QThr t;
        const float Tv = q_T[lane], e = q_eps[lane];
        t.L = Tv - e;
        t.U = Tv + e;
        t.T = Tv;
        t.eps = e;
        t.bstar = q_bstar[lane];
        t.nabove = q_nabove[lane];
        t.mc = min(q_mcnt[lane * 2], kMemHalf) | (min(q_mcnt[lane * 2 + 1], kMemHalf) << 16);
        t.flags = (q_flag[lane * 2] | q_flag[lane * 2 + 1]) & 2;
        a.thr[row0 + qbase + lane] = t;
    }
    {
        float* gv = a.gmem_v + (row0 + qbase) * (int64_t)FAL_FUSED_MEM;
        uint32_t* gi = a.gmem_id + (row0 + qbase) * (int64_t)FAL_FUSED_MEM;
        for (int e = lane; e < nqw * FAL_FUSED_MEM; e += 64) {
            const int qq = e / FAL_FUSED_MEM, slot = e % FAL_FUSED_MEM;
            const int src = qq * kMemStride + (slot < kMemHalf ? slot : kMemSlot + slot - kMemHalf);
            gv[e] = mem_v[src];
            gi[e] = mem_id[src];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// band_kernel: one wave = 32 queries of one bucket; exact similarities of the precursor window on the fp32 matrix cores
// ------------------------------------------------------------------------------------------------------------
// IVF = true (ivf16.hip): the bucket has an index; a window candidate only counts if its list is one of the query's probed
// lists (a bit mask per query in LDS, built from the probe table), thresholds come from select16_kernel for every query
template <int DH4, bool IVF>
__global__ __launch_bounds__(64, 1) void band_kernel(FusedArgs a) {
    constexpr int D = DH4 * 8, DH = D / 2;
    constexpr int kStride = FAL_FUSED_KEEP + 1;               // dwords per query row of the kept lists (odd: conflict-free)
    __shared__ uint32_t kept_u[32 * kStride];
    __shared__ uint32_t kept_id[32 * kStride];
    extern __shared__ uint32_t pmask[];                       // IVF: [32][mask_words] probed lists of the tile's queries
    int ji, lt;
    if (!find_job32(a, blockIdx.x, &ji, &lt)) return;
    const DenseJob job = a.jobs32[ji];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nc = job.nc;
    const int64_t row0 = job.q_row0;
    const int qbase = 32 * lt;
    const int nqw = min(32, nc - qbase);
    const bool need_thr = IVF || nc > a.k;
    const int dh4 = DH4;
    const float* X = a.X;
    const int W = a.mask_words;
    if constexpr (IVF) {
        for (int e = threadIdx.x; e < 32 * W; e += 64) pmask[e] = 0u;
        __syncthreads();
        if (r < nqw) {
            const int64_t p = a.pos_of_row[row0 + qbase + r];
            for (int j = h; j < a.n_probe; j += 2) {
                const int l = a.probes[p * a.n_probe + j];
                if (l >= 0) atomicOr(&pmask[r * W + (l >> 5)], 1u << (l & 31));
            }
        }
        __syncthreads();
    }
    float qf[DH4 * 4];
    load_half_row<DH4>(qf, X + (row0 + qbase + min(r, nqw - 1)) * D + (int64_t)h * DH, dh4);
    const float* pm = a.pmz + row0;
    const float* rtp = a.rt ? a.rt + row0 : nullptr;
    const bool use_rt = rtp != nullptr && a.rt_tol >= 0.0;
    // candidate range that can pass the tolerance for ANY query of the tile (slightly widened; the exact test decides
    // below): rows are sorted by precursor m/z
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
        // rows are sorted: scan outwards from the tile 64 rows at a time (one load + ballot per step, typically one step)
        // instead of a binary search whose every probe is a dependent global round trip
        wlo = qbase;
        for (;;) {
            const int c = wlo - 1 - lane;
            const bool in = c >= 0 && (double)pm[max(c, 0)] >= lob;
            const int cnt = __popcll(__ballot(in));          // sorted => the lanes that pass are a prefix
            wlo -= cnt;
            if (cnt < 64) break;
        }
        whi = qbase + nqw;
        for (;;) {
            const int c = whi + lane;
            const bool in = c < nc && (double)pm[min(c, nc - 1)] <= hib;
            const int cnt = __popcll(__ballot(in));
            whi += cnt;
            if (cnt < 64) break;
        }
    }
    const int ql = r;
    const bool qvalid = ql < nqw;
    const int64_t qrow = row0 + qbase + min(ql, nqw - 1);
    const float qmz = pm[qbase + min(ql, nqw - 1)];
    const float qrt = use_rt ? rtp[qbase + min(ql, nqw - 1)] : 0.f;
    float Lq = -INFINITY, Uq = -INFINITY;
    if (need_thr) {
        const QThr t = a.thr[qrow];
        Lq = t.L;
        Uq = t.U;
    }
    // the tolerance tests in float32: |x| <= tol_f is EXACTLY fabs((double)x * scale) <= tol of filter_kernel (tol_f is the
    // largest float32 for which the float64 form holds; the product with a positive constant is monotone)
    const float tol_f = a.tol_f, rt_f = a.rt_f;
    int kc = 0;
    bool amb_any = false;
    uint32_t* ku = kept_u + ql * kStride + h * kKeepHalf;
    uint32_t* kid = kept_id + ql * kStride + h * kKeepHalf;
    const int n_chunks = (whi - wlo + 31) >> 5;
    CandStream<DH4> cs;
    auto crow = [&](int c0) -> const float* { return X + (row0 + min(c0 + r, nc - 1)) * D + (int64_t)h * DH; };
    const float* cur = crow(wlo);
    cs.prime(cur, dh4);
    for (int ci = 0, c0 = wlo; ci < n_chunks; ++ci, c0 += 32) {
        const float* nxt = crow(c0 + 32);
        // candidate metadata of this lane's 16 rows: issued before the MFMA chain, used after it
        float nmz[16], nrt[16];
        int nls[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cc = min(c0 + rowoff16(i) + 4 * h, nc - 1);
            nmz[i] = pm[cc];
            nrt[i] = use_rt ? rtp[cc] : 0.f;
            nls[i] = IVF ? a.assign[row0 + cc] : 0;
        }
        const f32x16 acc = cs.template dot<false>(qf, cur, nxt, dh4, [] {});
        cur = nxt;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + rowoff16(i) + 4 * h;
            const float s = acc[i];
            const float diff = qmz - nmz[i];                 // mass_diff(query, neighbour): the arithmetic of filter_kernel
            const float x = a.is_da ? diff : diff / nmz[i];
            bool ok = fabsf(x) <= tol_f;
            if (use_rt) ok = ok && fabsf(qrt - nrt[i]) <= rt_f;
            ok = ok && c < whi && qvalid && c != qbase + ql;
            if constexpr (IVF) ok = ok && ((pmask[ql * W + (nls[i] >> 5)] >> (nls[i] & 31)) & 1u) != 0u;
            if (ok && s >= Lq) {                             // below L: certainly not among the k best
                const bool amb = s <= Uq;                    // inside [L, U]: decided exactly by resolve_kernel
                if (kc < kKeepHalf) {
                    ku[kc] = max(f32_sortable(s), 1u);
                    kid[kc] = (uint32_t)(row0 + c) | (amb ? 0x80000000u : 0u);
                }
                ++kc;
                amb_any = amb_any || amb;
            }
        }
    }
    if (qvalid) a.gkcnt[(row0 + qbase + ql) * 2 + h] = min(kc, kKeepHalf) | (amb_any ? 0x100 : 0) | (kc > kKeepHalf ? 0x200 : 0);
    __syncthreads();
    {
        uint32_t* gu = a.gkept_u + (row0 + qbase) * (int64_t)FAL_FUSED_KEEP;
        uint32_t* gi = a.gkept_id + (row0 + qbase) * (int64_t)FAL_FUSED_KEEP;
        for (int e = lane; e < nqw * FAL_FUSED_KEEP; e += 64) {
            const int qq = e / FAL_FUSED_KEEP, slot = e % FAL_FUSED_KEEP;
            gu[e] = kept_u[qq * kStride + slot];
            gi[e] = kept_id[qq * kStride + slot];
        }
    }
}

__device__ __forceinline__ void push_fallback(const FusedArgs& a, int64_t row, int job) {
    const int at = atomicAdd(a.fb_count, 1);
    if (at < a.fb_cap) {
        a.fb_list[2 * at] = (int32_t)row;
        a.fb_list[2 * at + 1] = job;
    }
}

// ------------------------------------------------------------------------------------------------------------
// resolve_kernel: one wave per query (4 per workgroup, 8 queries each per 32-query tile): ambiguous candidates against
// the exact k-th key, sort, neighbour lists
// ------------------------------------------------------------------------------------------------------------
// (6 / 7 / 8 waves per SIMD = 80 / 72 / 64 registers: no difference, round 6)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void resolve_kernel(FusedArgs a, int d) {
    __shared__ uint2 s_key_all[4 * 64];                        // kept candidates: (x = ~id, y = sortable similarity)
    __shared__ int32_t seg_off_all[4][FAL_MAX_N_PROBE + 1];      // IVF: key-stream offset / first position of every probed list
    __shared__ int64_t seg_src_all[4][FAL_MAX_N_PROBE];
    int ji, lt;
    if (!find_job32(a, blockIdx.x, &ji, &lt)) return;
    const DenseJob job = a.jobs32[ji];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint2* s_key = s_key_all + 64 * w;
    const int nc = job.nc, k = a.k;
    const int64_t row0 = job.q_row0;
    const int nqw = min(32, nc - 32 * lt);
    const bool need_thr = a.ivf || nc > k;
    // Two passes over the wave's queries: the first settles every query without an ambiguous candidate (94 % of them) and leaves
    // the others to the second.  As ONE loop the exact chains' addresses and masks (loop invariants of the heavy branch) were
    // hoisted in front of it and spilled there, at the kernel's 80-register cap: 24 dwords per lane = 6 KB of scratch writes per
    // wave = 768 B per query -- 7.7 GB of the 12.9 GB this kernel wrote per 10 M pass for 5.1 GB of neighbour rows (round 6).
    // A query's hand-off: every load of the common path, issued together (one memory round trip, not four) -- and, in the first
    // pass, one query AHEAD of its use (the wave's eight queries were eight round trips in a row).
    struct Hand { int i0, i1; uint32_t ku, ki; QThr t; };
    auto fetch = [&](int ql) __attribute__((always_inline)) -> Hand {
        const int64_t row = row0 + 32 * lt + ql;
        Hand h{};
        h.i0 = a.gkcnt[row * 2];
        h.i1 = a.gkcnt[row * 2 + 1];
        h.ku = a.gkept_u[row * FAL_FUSED_KEEP + lane];
        h.ki = a.gkept_id[row * FAL_FUSED_KEEP + lane];
        if (need_thr) h.t = a.thr[row];
        return h;
    };
    auto one = [&](int ql, const Hand& hand, auto heavy) __attribute__((always_inline)) -> bool {
        constexpr bool HEAVY = decltype(heavy)::value;
        const int64_t row = row0 + 32 * lt + ql;
        const int i0 = hand.i0, i1 = hand.i1;
        const uint32_t ku_l = hand.ku, ki_l = hand.ki;
        const QThr t = hand.t;
        const int k0 = i0 & 0xFF, k1 = i1 & 0xFF;
        bool fb = ((i0 | i1) & 0x200) != 0;
        bool amb = ((i0 | i1) & 0x100) != 0;
        int why = fb ? 2 : 0;                                // (reason counters of the fallback list: fb_count[1..3])
        if (need_thr) {
            if (!fb && (t.flags & 2)) why = 1;
            fb = fb || (t.flags & 2);
        }
        uint32_t ki_x = ki_l;
        bool drop = false;
        if (a.ivf == 2) {
            // pairs16.hip handed over every window candidate whose KEY is not certainly below the k-th best, with its exact
            // similarity: below L certainly not among the k best, inside [L, U] ambiguous
            const int half0 = lane >= kKeepHalf, idx0 = lane - half0 * kKeepHalf;
            const bool mine0 = lane < FAL_FUSED_KEEP && idx0 < (half0 ? k1 : k0) && !fb;
            const float s = sortable_f32(ku_l);
            const bool ok = mine0 && s >= t.L;
            const bool amb_l = ok && s <= t.U;
            drop = mine0 && !ok;
            ki_x |= amb_l ? 0x80000000u : 0u;
            const bool amb_x = __ballot(amb_l) != 0ull;
            if (amb_x && !amb && !fb) {                      // (cannot happen: an exact value inside [L, U] has its key inside the
                fb = true;                                   //  ambiguous range, which kept16_kernel flags)
                why = 3;
            }
            amb = amb_x;
        }
        uint32_t uT = 0, iT = 0xFFFFFFFFu;                  // exact k-th key (only when an ambiguous candidate exists)
        if (!HEAVY && !fb && amb) return true;               // (second pass)
        if (HEAVY && !fb && amb) {
            const int m0 = t.mc & 0xFFFF, m1 = t.mc >> 16;
            const bool have = lane < kMemHalf ? lane < m0 : (lane - kMemHalf) < m1;
            const float v = have ? a.gmem_v[row * FAL_FUSED_MEM + lane] : 0.f;
            uint32_t mid = have ? a.gmem_id[row * FAL_FUSED_MEM + lane] : 0u;
            if (a.ivf) {
                // select16_kernel handed the members over as positions in the query's key stream (probe order, list order
                // inside): stream offsets of the probed lists by a wave prefix sum, then position -> list row -> sorted row
                int32_t* seg_off = seg_off_all[w];
                int64_t* seg_src = seg_src_all[w];
                const int np = a.n_probe;
                const int32_t* pr = a.probes + (int64_t)a.pos_of_row[row] * np;
                int run = 0;
                for (int j0 = 0; j0 < np; j0 += 64) {
                    const int j = j0 + lane;
                    const int32_t l = j < np ? pr[j] : -1;
                    int64_t b = 0, e = 0;
                    if (l >= 0) {
                        b = a.list_off[job.c_row0 + l];
                        e = a.list_off[job.c_row0 + l + 1];
                    }
                    const int len = (int)(e - b);
                    int incl = len;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += o;
                    }
                    if (j < np) {
                        seg_off[j] = run + incl - len;
                        seg_src[j] = b;
                    }
                    run += __shfl(incl, 63, 64);
                }
                wave_lds_sync();
                if (have) {
                    const int pp = (int)mid;
                    int lo = 0, hi = np - 1;                     // last probe slot with seg_off <= pp
                    while (lo < hi) {
                        const int md = (lo + hi + 1) >> 1;
                        if (seg_off[md] <= pp) lo = md; else hi = md - 1;
                    }
                    mid = (uint32_t)a.perm[seg_src[lo] + (pp - seg_off[lo])] - (uint32_t)row0;
                }
                wave_lds_sync();
            }
            const bool inE = have && fabsf(v - t.T) <= 2.f * t.eps;
            const int n_bin_above = __popcll(__ballot(have && (int)bin_of(v) > t.bstar));
            const int n_hi = __popcll(__ballot(have && v > t.T + 2.f * t.eps));
            const int need = k - (t.nabove - n_bin_above + n_hi);       // rank of the k-th key among the members of E
            const unsigned long long em = __ballot(inE);
            if (need < 1 || need > __popcll(em)) {
                fb = true;
                why = 3;
            } else {
                float s = 0.f;
                if (a.sp_cols) {
                    // the query's sparse form across the wave (lane e = entry e), every member lane gathers its own row at the
                    // query's columns (simtile.h: sparse_row_chain -- the bits of exact_dot)
                    const uint32_t qc = a.sp_cols[row * kSparseW + lane];
                    const float qv = a.sp_vals[row * kSparseW + lane];
                    const bool sparse = (uint32_t)__builtin_amdgcn_readlane((int)qc, 0) != (uint32_t)kColDense;
                    const int n_ent = __popcll(__ballot(qc < (uint32_t)kColDense));
                    if (sparse) {
                        const float sv = sparse_row_chain(qc, qv, n_ent, a.X + (row0 + (inE ? mid : 0u)) * d);
                        s = inE ? sv : 0.f;
                    } else if (inE) {
                        s = exact_dot(a.X + row * d, a.X + (row0 + mid) * d, d);
                    }
                } else if (inE) {
                    s = exact_dot(a.X + row * d, a.X + (row0 + mid) * d, d);
                }
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
        if (fb) {
            if (lane == 0) {
                push_fallback(a, row, ji);
                atomicAdd(a.fb_count + why, 1);
            }
            return false;
        }
        // the kept candidates of the two lane halves, ambiguous ones against the exact k-th key; sort; store
        bool keepit = false;
        uint32_t u = 0, id = 0;
        const int half = lane >= kKeepHalf, idx = lane - half * kKeepHalf;
        if (lane < FAL_FUSED_KEEP && idx < (half ? k1 : k0) && !drop) {
            u = ku_l;
            id = ki_x;
            keepit = true;
            if (id & 0x80000000u) {
                id &= 0x7FFFFFFFu;
                keepit = u > uT || (u == uT && id <= iT);
            }
        }
        const unsigned long long km = __ballot(keepit);
        if (keepit) {
            const int at = __popcll(km & ((1ull << lane) - 1ull));
            s_key[at] = make_uint2(~id, u);
        }
        const int c = __popcll(km);
        wave_lds_sync();
        if (a.nb_count && lane == 0) a.nb_count[row] = min(c, a.keep);
        rank_or_sort_and_store_nb(s_key, c, a.keep, lane, a.nb_idx + row * a.keep, a.nb_dist + row * a.keep);
        wave_lds_sync();
        return false;
    };
    uint32_t later = 0;
    if (w >= nqw) return;
    Hand cur = fetch(w);
    for (int ql = w; ql < nqw; ql += 4) {
        Hand nxt = cur;
        if (ql + 4 < nqw) nxt = fetch(ql + 4);
        if (one(ql, cur, std::false_type{})) later |= 1u << (ql >> 2);
        cur = nxt;
    }
    later = (uint32_t)__builtin_amdgcn_readfirstlane((int)later);
    if (later == 0) return;
    asm volatile("" ::: "memory");
    for (int ql = w; ql < nqw; ql += 4)
        if ((later >> (ql >> 2)) & 1u) one(ql, fetch(ql), std::true_type{});
}

static void launch_resolve(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles32) {
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)(list_tiles32 * 8)), dim3(256), 0, ctx->stream, a, d);
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
        const DenseJob job = a.jobs32[a.fb_list[2 * t + 1]];
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

// the same for a query of an IVF bucket: the exact similarities of the rows of its probed lists, in the order of the staged
// path's sims segment (probe order, list order inside), then the staged path's MODE_IVF selection
__global__ __launch_bounds__(64) void ivf_fallback_kernel(FusedArgs a, SelectArgs sa, float* __restrict__ scratch,
                                                          int64_t scratch_stride, int d) {
    __shared__ uint32_t sel_u[kSelBuf];
    __shared__ uint32_t sel_id[kSelBuf];
    __shared__ int64_t seg_off[FAL_MAX_N_PROBE + 1];
    __shared__ int64_t seg_src[FAL_MAX_N_PROBE];
    const int lane = threadIdx.x;
    const int total = min(*a.fb_count, a.fb_cap);
    float* row_s = scratch + (int64_t)blockIdx.x * scratch_stride;
    const int np = a.n_probe;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t row = a.fb_list[2 * t];
        const DenseJob job = a.jobs32[a.fb_list[2 * t + 1]];
        const int32_t* pr = a.probes + (int64_t)a.pos_of_row[row] * np;
        if (lane == 0) {
            int64_t off = 0;
            for (int j = 0; j < np; ++j) {
                const int32_t l = pr[j];
                seg_off[j] = off;
                seg_src[j] = 0;
                if (l >= 0) {
                    const int64_t b = a.list_off[job.c_row0 + l], e = a.list_off[job.c_row0 + l + 1];
                    seg_src[j] = b;
                    off += e - b;
                }
            }
            seg_off[np] = off;
        }
        __syncthreads();
        const int64_t nc = min<int64_t>(seg_off[np], scratch_stride - (int64_t)kSimsSlack);
        const float* qp = a.X + row * d;
        for (int j = 0; j < np; ++j) {
            const int64_t o = seg_off[j], len = min<int64_t>(seg_off[j + 1], nc) - o;
            for (int64_t i = lane; i < len; i += 64)            // (list-order position -> sorted row: no list-order copy of X needed)
                row_s[o + i] = exact_dot(qp, a.X + (int64_t)a.perm[seg_src[j] + i] * d, d);
        }
        __threadfence();
        __syncthreads();
        SelQuery qy{row_s, nc, 0};
        int carry;
        if (nc <= 128) carry = select_rounds<MODE_IVF, 2>(sa, qy, sa.k, lane, sel_u, sel_id, seg_off, seg_src);
        else if (nc <= 256) carry = select_rounds<MODE_IVF, 4>(sa, qy, sa.k, lane, sel_u, sel_id, seg_off, seg_src);
        else if (nc <= 512) carry = select_rounds<MODE_IVF, 8>(sa, qy, sa.k, lane, sel_u, sel_id, seg_off, seg_src);
        else carry = select_rounds<MODE_IVF, 16>(sa, qy, sa.k, lane, sel_u, sel_id, seg_off, seg_src);
        filter_sort_store(sa, sel_u, sel_id, sel_u + FAL_MAX_K_ANN, sel_id + FAL_MAX_K_ANN, carry, row, lane);
        __syncthreads();
    }
}

// largest float32 y >= 0 with (double)y * scale <= tol (-1 when there is none): the float32 form of the tolerance tests
static float float_le_bound(double tol, double scale) {
    if (!(tol >= 0.0)) return -1.f;
    uint32_t lo = 0, hi = 0x7F7FFFFFu;                       // bit patterns of non-negative finite floats are ordered
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        float y;
        memcpy(&y, &mid, 4);
        if ((double)y * scale <= tol) lo = mid; else hi = mid - 1;
    }
    float y;
    memcpy(&y, &lo, 4);
    return y;
}

__global__ void tile_job32_kernel(const DenseJob* __restrict__ jobs, int n_jobs, int64_t grid, int32_t* __restrict__ table) {
    const int64_t bid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (bid >= grid) return;
    int ji = -1, lt = 0;
    if (!find_job_xcd(jobs, n_jobs, (unsigned)bid, &ji, &lt)) ji = -1;
    table[bid] = ji;
}

int launch_tile_job32(fal_ctx* ctx, const DenseJob* jobs32, int n_jobs32, int64_t list_tiles32, int32_t* table) {
    const int64_t grid = list_tiles32 * 8;
    if (grid <= 0) return FAL_OK;
    hipLaunchKernelGGL(tile_job32_kernel, dim3((unsigned)ceil_div(grid, 256)), dim3(256), 0, ctx->stream, jobs32, n_jobs32, grid, table);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int fused_prepare(fal_ctx* ctx, FusedArgs* ap, int64_t n_rows) {
    FusedArgs& a = *ap;
    a.tol_f = float_le_bound(a.tol, a.is_da ? 1.0 : 1e6);
    a.rt_f = float_le_bound(a.rt_tol, 1.0);
    // hand-off buffers, indexed by sorted row
    unsigned char* hand = nullptr;
    const size_t per_row = sizeof(QThr) + (size_t)FAL_FUSED_MEM * 8 + (size_t)FAL_FUSED_KEEP * 8 + 8;
    FAL_TRY(ctx->reserve(SLOT_FUSED3, per_row * (size_t)n_rows + 256, (void**)&hand));
    a.thr = reinterpret_cast<QThr*>(hand);
    a.gmem_v = reinterpret_cast<float*>(a.thr + n_rows);
    a.gmem_id = reinterpret_cast<uint32_t*>(a.gmem_v + n_rows * FAL_FUSED_MEM);
    a.gkept_u = a.gmem_id + n_rows * FAL_FUSED_MEM;
    a.gkept_id = a.gkept_u + n_rows * FAL_FUSED_KEEP;
    a.gkcnt = reinterpret_cast<int32_t*>(a.gkept_id + n_rows * FAL_FUSED_KEEP);
    // fallback list: resolve_kernel pushes a query at most once, so n_rows entries can never overflow
    int32_t* fb = nullptr;
    const int fb_cap = (int)std::max<int64_t>(n_rows, 1);
    FAL_TRY(ctx->reserve(SLOT_FUSED, sizeof(int32_t) * (size_t)(2 * fb_cap + 16), (void**)&fb));
    a.fb_count = fb;
    a.fb_list = fb + 16;
    a.fb_cap = fb_cap;
    FAL_CHECK_HIP(hipMemsetAsync(fb, 0, sizeof(int32_t) * 16, ctx->stream));
    return FAL_OK;
}

int launch_fused_ivf_tail(fal_ctx* ctx, const FusedArgs& a, int d, int64_t list_tiles32, int64_t max_cand) {
    if (a.n_jobs32 <= 0 || list_tiles32 <= 0) return FAL_OK;
    FAL_REQUIRE(list_tiles32 * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    StageScope ts(ctx, ST_SELECT);
    launch_resolve(ctx, a, d, list_tiles32);
    FAL_CHECK_HIP(hipGetLastError());
    // exact fallback over the probed lists
    const int64_t stride = ((max_cand + 63) & ~63ll) + (int64_t)kSimsSlack;
    const int fb_grid = (int)std::max<int64_t>(64, std::min<int64_t>(ctx->num_cus * 16, ((int64_t)1 << 28) / stride));
    float* scratch = nullptr;
    FAL_TRY(ctx->reserve(SLOT_FUSED2, sizeof(float) * (size_t)fb_grid * (size_t)stride, (void**)&scratch));
    SelectArgs sa{};
    sa.k = a.k;
    sa.n_probe = a.n_probe; sa.perm = a.perm;
    sa.f_pmz = a.pmz; sa.f_rt = a.rt; sa.f_tol = a.tol; sa.f_rt_tol = a.rt_tol; sa.f_is_da = a.is_da;
    sa.f_keep = a.keep; sa.nb_idx = a.nb_idx; sa.nb_dist = a.nb_dist; sa.nb_count = a.nb_count;
    hipLaunchKernelGGL(ivf_fallback_kernel, dim3((unsigned)fb_grid), dim3(64), 0, ctx->stream, a, sa, scratch, stride, d);
    FAL_CHECK_HIP(hipGetLastError());
    FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host + 2, a.fb_count, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return FAL_OK;
}

int launch_fused(fal_ctx* ctx, const FusedArgs& a_in, int d, int64_t n_rows, int64_t list_tiles128, int64_t list_tiles32,
                 int max_nc) {
    if (a_in.n_jobs32 <= 0 || list_tiles32 <= 0) return FAL_OK;
    FusedArgs a = a_in;
    FAL_REQUIRE(list_tiles32 * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    FAL_REQUIRE(max_nc < 65536, FAL_EUNSUPPORTED, "fused scan: buckets must hold fewer than 65,536 rows");
    FAL_TRY(fused_prepare(ctx, &a, n_rows));
    int32_t* fb = a.fb_count;
    // scratch rows of the fallback
    const int fb_grid = ctx->num_cus * 16;
    const int64_t stride = (((int64_t)max_nc + 63) & ~63ll) + (int64_t)kSimsSlack;
    float* scratch = nullptr;
    FAL_TRY(ctx->reserve(SLOT_FUSED2, sizeof(float) * (size_t)fb_grid * (size_t)stride, (void**)&scratch));
    const int steps = d / 16;
    {
        StageScope ts(ctx, ST_SCAN);
        if (a.n_jobs128 > 0 && list_tiles128 > 0) {
            const size_t lds = (size_t)32 * ((size_t)d * 2 + 16) + 4 * kWaveHist + 4 * kWaveSmall;
            dim3 grid((unsigned)(list_tiles128 * 8)), block(256);
#define FAL_LAUNCH_APPROX(S)                                                                                      \
    do {                                                                                                          \
        FAL_CHECK_HIP(hipFuncSetAttribute((const void*)approx_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((approx_kernel<S>), grid, block, lds, ctx->stream, a);                                 \
    } while (0)
            switch (steps) {
                case 4: FAL_LAUNCH_APPROX(4); break;
                case 8: FAL_LAUNCH_APPROX(8); break;
                case 16: FAL_LAUNCH_APPROX(16); break;
                case 25: FAL_LAUNCH_APPROX(25); break;
                default:
                    set_error("fused scan: low_dim %d has no instantiation (64, 128, 256, 400)", d);
                    return FAL_EUNSUPPORTED;
            }
#undef FAL_LAUNCH_APPROX
            FAL_CHECK_HIP(hipGetLastError());
        }
    }
    {
        StageScope ts(ctx, ST_SCAN);
        dim3 grid((unsigned)(list_tiles32 * 8)), block(64);
        switch (steps) {
            case 4: hipLaunchKernelGGL((band_kernel<8, false>), grid, block, 0, ctx->stream, a); break;
            case 8: hipLaunchKernelGGL((band_kernel<16, false>), grid, block, 0, ctx->stream, a); break;
            case 16: hipLaunchKernelGGL((band_kernel<32, false>), grid, block, 0, ctx->stream, a); break;
            case 25: hipLaunchKernelGGL((band_kernel<50, false>), grid, block, 0, ctx->stream, a); break;
            default: return FAL_EUNSUPPORTED;
        }
        FAL_CHECK_HIP(hipGetLastError());
    }
    {
        StageScope ts(ctx, ST_SELECT);
        launch_resolve(ctx, a, d, list_tiles32);
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
    FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host, fb, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return FAL_OK;
}

bool fused_supports(int d) { return d == 64 || d == 128 || d == 256 || d == 400; }

}  // namespace fal
FAL_WARM_KERNEL(fal::resolve_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)
