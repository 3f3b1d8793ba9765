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
constexpr int kWaveHist = 17408;                                // >= 32 * kHistStride: histogram, later member lists / output rows
constexpr int kWaveSmall = 2560;     // per-wave per-query scalars + the sort staging of phase E
constexpr int kKeptBytes = 4 * 2 * 32 * (kKeepCap + 1) * 4;       // kept lists of the four waves (alias the staging buffers)
__host__ __device__ constexpr int region_a_bytes(int d) { return 2 * 32 * (d * 2 + 16) > kKeptBytes ? 2 * 32 * (d * 2 + 16) : kKeptBytes; }

__device__ __forceinline__ int rowoff16(int i) { return (i & 3) + 8 * (i >> 2); }

// the exact similarity on the vector ALU: the k-ordered fmaf chain of simtile.h, bit for bit
__device__ __forceinline__ float exact_dot(const float* __restrict__ a, const float* __restrict__ b, int d) {
    const int dh4 = d >> 3;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float acc = 0.f;
    constexpr int U = 4;                       // 16 loads in flight per lane: the chain itself is latency-bound otherwise
    int j = 0;
    for (; j + U <= dh4; j += U) {
        float4 al[U], ah[U], bl[U], bh[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
            al[t] = a4[j + t];
            ah[t] = a4[dh4 + j + t];
            bl[t] = b4[j + t];
            bh[t] = b4[dh4 + j + t];
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
            acc = __builtin_fmaf(al[t].x, bl[t].x, acc);
            acc = __builtin_fmaf(ah[t].x, bh[t].x, acc);
            acc = __builtin_fmaf(al[t].y, bl[t].y, acc);
            acc = __builtin_fmaf(ah[t].y, bh[t].y, acc);
            acc = __builtin_fmaf(al[t].z, bl[t].z, acc);
            acc = __builtin_fmaf(ah[t].z, bh[t].z, acc);
            acc = __builtin_fmaf(al[t].w, bl[t].w, acc);
            acc = __builtin_fmaf(ah[t].w, bh[t].w, acc);
        }
    }
    for (; j < dh4; ++j) {
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

// LDS per wave while the passes run: histogram [32 queries][256 x u16 (+4 B pad)]; afterwards the member lists
// [32 queries][2 halves][32] (value f32 + candidate u32, row stride 65 dwords); in phase E the sorted output rows.
constexpr int kMemHalf = kMemCap / 2;            // members kept per (query, lane half)
constexpr int kMemSlot = kMemHalf + 1;           // ... plus the dump slot the branch-free append writes to when full
constexpr int kMemStride = 67;                   // dwords per query row (odd: lane = query accesses are conflict-free)
constexpr int kKeepHalf = kKeepCap / 2;          // selected window candidates kept per (query, lane half)
constexpr int kKeepStride = kKeepCap + 1;        // dwords per query row of the kept lists
static_assert(2 * 32 * kMemStride * 4 <= kWaveHist, "member lists alias the histograms");
static_assert(4 * 2 * 32 * kKeepStride * 4 <= kKeptBytes + 4096, "kept lists");

template <int STEPS>
__global__ __launch_bounds__(256, 1) void fused_kernel(FusedArgs a) {
    constexpr int D = STEPS * 16, DH = D / 2, DH4 = D / 8;
    constexpr int RB16 = D / 8;                     // 16-byte pieces per float16 row
    constexpr int RS = D * 2 + 16;                  // LDS row stride (padded: conflict-free b128 reads)
    constexpr int PIECES = 32 * RB16;
    constexpr int kStage = (PIECES + 255) / 256;
    constexpr int NB = STEPS < 8 ? STEPS : 8;         // LDS operand reads in flight ahead of the MFMAs (8 passes each)
    constexpr int kStageBytes = region_a_bytes(D);   // staging buffers, later the kept lists
    static_assert(kStage <= 7, "staging registers");
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
    const int dbg = a.dbg;
    unsigned long long* stamp = ((dbg & 128) && threadIdx.x == 0) ? a.stamps + (size_t)blockIdx.x * 10 : nullptr;
#define FAL_STAMP(I) { if (stamp) stamp[I] = __builtin_amdgcn_s_memtime(); }
    FAL_STAMP(0)
    if (stamp) { stamp[8] = (unsigned long long)nc; stamp[9] = 0; }
    // lane (r, h) serves query r of the wave throughout: MFMA results are D[candidate][query] (column = lane & 31),
    // so its 16 accumulator registers hold 16 candidates of ITS query -- every per-query structure is lane-private
    // (the two halves of a query keep separate sub-lists), no ballots, no cross-lane traffic in the epilogues

    unsigned char* whist = lds + kStageBytes + w * kWaveHist;                  // histogram / member lists / output rows
    unsigned char* wsmall = lds + kStageBytes + 4 * kWaveHist + w * kWaveSmall;
    float* q_lo = reinterpret_cast<float*>(wsmall);          // [32] member interval
    float* q_hi = q_lo + 32;
    float* q_L = q_hi + 32;                                  // [32] exact k-th value lies in [L, U]
    float* q_U = q_L + 32;
    float* q_T = q_U + 32;                                   // [32] k-th best approximate value
    float* q_eps = q_T + 32;
    int* q_bstar = reinterpret_cast<int*>(q_eps + 32);       // [32]
    int* q_nabove = q_bstar + 32;
    int* q_mcnt = q_nabove + 32;                             // [64] members per (query, half)
    int* q_kcnt = q_mcnt + 64;                               // [64] kept per (query, half)
    int* q_flag = q_kcnt + 64;                               // [64] bit 0: ambiguous candidate present, bit 1: fallback
    uint32_t* q_uT = reinterpret_cast<uint32_t*>(q_flag + 64);   // [32] exact k-th key of ambiguous queries
    uint32_t* q_iT = q_uT + 32;
    int* e_q = reinterpret_cast<int*>(q_iT + 32);            // [32] the exact chunk of phase D': query, candidate, value
    int* e_c = e_q + 32;
    float* e_v = reinterpret_cast<float*>(e_c + 32);
    static_assert((8 * 32 + 3 * 64 + 2 * 32 + 3 * 32) * 4 <= kWaveSmall, "per-wave scalars");
    float* mem_v = reinterpret_cast<float*>(whist);                            // [32][kMemStride]: half h at [h * kMemSlot ...)
    uint32_t* mem_id = reinterpret_cast<uint32_t*>(whist + 32 * kMemStride * 4);
    uint32_t* kept_u = reinterpret_cast<uint32_t*>(lds + w * (2 * 32 * kKeepStride * 4));   // [32][kKeepStride] (after the passes)
    uint32_t* kept_id = kept_u + 32 * kKeepStride;

    q_flag[lane] = 0;
    q_mcnt[lane] = 0;
    q_kcnt[lane] = 0;
    if (lane < 32) {
        q_L[lane] = -INFINITY;
        q_U[lane] = -INFINITY;
        q_uT[lane] = 0u;
        q_iT[lane] = 0xFFFFFFFFu;
    }

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
            auto chunk = [&](const int bufcur) {
                const unsigned char* rowp = lds + (size_t)bufcur * 32 * RS + r * RS + h * DH * 2;
                half8 rh[NB];
#pragma unroll
                for (int s = 0; s < NB; ++s) rh[s] = *reinterpret_cast<const half8*>(rowp + s * 16);
                // pin the whole operand ring in front of the first MFMA: without this group the scheduler satisfies the
                // "one read per step" pattern below by issuing the reads one at a time -- each MFMA then waits for an LDS
                // round trip (measured: 6.4k instead of ~1k cycles per chunk)
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
            for (int cc0 = 0; cc0 < nc; cc0 += 64) {
                {
                    const f32x16 acc = chunk(0);
                    prev = acc;
                    prev_c0 = cc0;
                    FAL_STORE(A, 1)                          // set A holds chunk cc0 + 32 ...
                    FAL_LOAD(A, cc0 + 96)                    // ... and now fetches chunk cc0 + 96
                    __syncthreads();
                }
                if (cc0 + 32 >= nc) break;
                {
                    const f32x16 acc = chunk(1);
                    prev = acc;
                    prev_c0 = cc0 + 32;
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

        // ---- pass 1: histogram of the approximate similarities (bin = floor(256 v), 255 = everything above) ----
        unsigned char* hrow_b = whist + r * kHistStride;
        run_pass([&](int, float v, int, bool valid) {
            const uint32_t b = min(255u, (uint32_t)(v * 256.f));
            const uint32_t inc = (b & 1) ? 0x10000u : 1u;
            atomicAdd(reinterpret_cast<unsigned*>(hrow_b + ((b >> 1) << 2)), (valid && !(dbg & 1)) ? inc : 0u);
        });
        FAL_STAMP(1)
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
        FAL_STAMP(2)
        // ---- pass 2: the members of the (widened) threshold bin -> lane-private LDS lists (the histograms' memory).
        //      Branch-free: every value is written at the list's end, the end only advances on a hit (slot kMemHalf
        //      is the dump slot of a full list) -------------------------------------------------------------------------
        {
            const float lo_q = q_lo[r], hi_q = q_hi[r];
            int cnt = 0;
            float* mv = mem_v + r * kMemStride + h * kMemSlot;
            uint32_t* mi = mem_id + r * kMemStride + h * kMemSlot;
            run_pass([&](int, float v, int c, bool valid) {
                const bool hit = valid && v >= lo_q && v <= hi_q && !(dbg & 2);
                const int at = min(cnt, kMemHalf);
                mv[at] = v;
                mi[at] = (uint32_t)c;
                cnt += hit ? 1 : 0;
            });
            q_mcnt[r * 2 + h] = cnt;
        }
#undef FAL_PASS
#undef FAL_CHUNK
#undef FAL_LOAD
#undef FAL_STORE
#undef FAL_LOAD_ONE
#undef FAL_STORE_ONE
#undef FAL_FOR_A
#undef FAL_FOR_B
        __syncthreads();
        FAL_STAMP(3)
        // ---- T~ = the (k - n_above)-th best approximate value inside bin b*: lane (r, h) tries the members of ITS half as
        //      pivots and ranks each against all members of the query; exact k-th value in [T~ - eps, T~ + eps] -------------
        if (!(dbg & 4)) {
            const int m0 = q_mcnt[r * 2], m1 = q_mcnt[r * 2 + 1];
            const int bstar = q_bstar[r], need = k - q_nabove[r];
            const float* mvq = mem_v + r * kMemStride;
            bool found = false;
            if (m0 > kMemHalf || m1 > kMemHalf || need < 1) {
                q_flag[r * 2 + h] = 2;                       // too many values share the bin: exact fallback
            } else {
                const int mine = h ? m1 : m0;
                for (int e = 0; e < mine; ++e) {
                    const float ve = mvq[h * kMemSlot + e];
                    if ((int)min(255u, (uint32_t)(ve * 256.f)) != bstar) continue;
                    int better = 0;                          // in-bin members with a larger value (ties: lower slot first)
                    for (int j = 0; j < m0; ++j) {
                        const float vj = mvq[j];
                        const bool inb = (int)min(255u, (uint32_t)(vj * 256.f)) == bstar;
                        better += (inb && (vj > ve || (vj == ve && (h == 1 || j < e)))) ? 1 : 0;
                    }
                    for (int j = 0; j < m1; ++j) {
                        const float vj = mvq[kMemSlot + j];
                        const bool inb = (int)min(255u, (uint32_t)(vj * 256.f)) == bstar;
                        better += (inb && (vj > ve || (vj == ve && h == 1 && j < e))) ? 1 : 0;
                    }
                    if (better == need - 1) {
                        const float eq = q_eps[r];
                        q_T[r] = ve;
                        q_L[r] = ve - eq;
                        q_U[r] = ve + eq;
                        found = true;
                    }
                }
            }
            // exactly one half finds the pivot; if neither does the lists are incomplete (defensive): fallback
            const unsigned long long fm = __ballot(found);
            const bool any = ((fm >> r) & 1ull) || ((fm >> (r + 32)) & 1ull);
            if (!any) q_flag[r * 2 + h] = 2;
        }
    }
    __syncthreads();                 // the staging buffers become the kept lists; the per-query scalars are final
    FAL_STAMP(4)

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
        const int ql = r;
        const bool qvalid = ql < nqw;
        const float qmz = pm[qbase + min(ql, nqw - 1)];
        const float qrt = use_rt ? rtp[qbase + min(ql, nqw - 1)] : 0.f;
        const float Lq = q_L[ql], Uq = q_U[ql];
        int kc = 0;
        bool amb_any = false;
        uint32_t* ku = kept_u + ql * kKeepStride + h * kKeepHalf;
        uint32_t* kid = kept_id + ql * kKeepStride + h * kKeepHalf;
        const int n_chunks = (dbg & 8) ? 0 : (whi - wlo + 31) >> 5;
        CandStream<DH4> cs;
        auto crow = [&](int c0) -> const float* { return X + (row0 + min(c0 + r, nc - 1)) * D + (int64_t)h * DH; };
        const float* cur = crow(wlo);
        cs.prime(cur, dh4);
        for (int ci = 0, c0 = wlo; ci < n_chunks; ++ci, c0 += 32) {
            const float* nxt = crow(c0 + 32);
            // candidate metadata of this lane's 16 rows: issued before the MFMA chain, used after it
            float nmz[16], nrt[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cc = min(c0 + rowoff16(i) + 4 * h, nc - 1);
                nmz[i] = pm[cc];
                nrt[i] = use_rt ? rtp[cc] : 0.f;
            }
            const f32x16 acc = cs.template dot<false>(qf, cur, nxt, dh4, [] {});
            cur = nxt;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = c0 + rowoff16(i) + 4 * h;
                const float s = acc[i];
                const float diff = qmz - nmz[i];             // mass_diff(query, neighbour): the arithmetic of filter_kernel
                const double md = a.is_da ? (double)diff : (double)(diff / nmz[i]) * 1e6;
                bool ok = fabs(md) <= a.tol;
                if (use_rt) ok = ok && fabs((double)(qrt - nrt[i])) <= a.rt_tol;
                ok = ok && c < whi && qvalid && c != qbase + ql;
                if (ok && s >= Lq) {                         // below L: certainly not among the k best
                    const bool amb = s <= Uq;                // inside [L, U]: decided exactly in phase E
                    if (kc < kKeepHalf) {
                        ku[kc] = max(f32_sortable(s), 1u);
                        kid[kc] = (uint32_t)(row0 + c) | (amb ? 0x80000000u : 0u);
                    }
                    ++kc;
                    amb_any = amb_any || amb;
                }
            }
        }
        FAL_STAMP(5)
        if (stamp) stamp[9] = (unsigned long long)n_chunks;
        q_kcnt[ql * 2 + h] = kc;
        if (amb_any) q_flag[ql * 2 + h] |= 1;
        if (kc > kKeepHalf) q_flag[ql * 2 + h] |= 2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---- ambiguous candidates (exact similarity inside [L, U]): the exact k-th key of their queries.  The members
        //      within 2 eps of T~ of ALL ambiguous queries of the wave (a few per query) form ONE more chunk for the fp32
        //      matrix cores -- gathered rows, the same fmaf chain -- instead of latency-bound scalar chains -------------
        if (need_thr && !(dbg & 16)) {
            int fq = q_flag[r * 2] | q_flag[r * 2 + 1];
            bool amb_q = (fq & 1) && !(fq & 2) && qvalid;
            const int mcnt = min(q_mcnt[r * 2 + h], kMemHalf);
            const float Tq = q_T[r], e2 = 2.f * q_eps[r];
            const float* mvh = mem_v + r * kMemStride + h * kMemSlot;
            const uint32_t* mih = mem_id + r * kMemStride + h * kMemSlot;
            int total = 0;
            const unsigned long long lt = (1ull << lane) - 1ull;
            for (int j = 0; j < kMemHalf; ++j) {
                const bool live = amb_q && j < mcnt;
                if (__ballot(live) == 0ull) break;
                const bool pred = live && fabsf(mvh[j] - Tq) <= e2;
                const unsigned long long mk = __ballot(pred);
                if (mk) {
                    const int pos = total + __popcll(mk & lt);
                    if (pred && pos < 32) {
                        e_q[pos] = r;
                        e_c[pos] = (int)mih[j];
                    }
                    total += __popcll(mk);
                }
            }
            if (total > 32) {                                // more pairs than one chunk holds (rare): exact fallback
                if (amb_q) q_flag[r * 2 + h] |= 2;
                amb_q = false;
                total = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (total > 0) {                                 // wave-uniform
                const float* er = X + (row0 + e_c[min(r, total - 1)]) * D + (int64_t)h * DH;
                cs.prime(er, dh4);
                const f32x16 acc = cs.template dot<false>(qf, er, er, dh4, [] {});
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int slot = rowoff16(i) + 4 * h;
                    if (slot < total && e_q[slot] == r) e_v[slot] = acc[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                unsigned long long todo = __ballot(lane < 32 && amb_q);
                while (todo) {
                    const int qa = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const int m0 = min(q_mcnt[qa * 2], kMemHalf), m1 = min(q_mcnt[qa * 2 + 1], kMemHalf);
                    const float Tv = q_T[qa], ee = q_eps[qa];
                    const int bstar = q_bstar[qa];
                    // lanes 0..31 = member slots of half 0, lanes 32..63 = half 1: counts certainly above the k-th value
                    const bool have = lane < 32 ? lane < m0 : (lane - 32) < m1;
                    const float v = have ? mem_v[qa * kMemStride + (lane < 32 ? lane : kMemSlot + lane - 32)] : 0.f;
                    const int n_bin_above = __popcll(__ballot(have && (int)min(255u, (uint32_t)(v * 256.f)) > bstar));
                    const int n_hi = __popcll(__ballot(have && v > Tv + 2.f * ee));
                    const int need = k - (q_nabove[qa] - n_bin_above + n_hi);
                    // lanes 0..31 = slots of the exact chunk
                    const bool mine = lane < total && e_q[lane & 31] == qa;
                    const unsigned long long em = __ballot(mine);
                    if (need < 1 || need > __popcll(em)) {
                        if (lane == 0) q_flag[qa * 2] |= 2;
                        continue;
                    }
                    const uint32_t u = mine ? max(f32_sortable(e_v[lane & 31]), 1u) : 0u;
                    const uint32_t id = (uint32_t)(row0 + e_c[lane & 31]);
                    int rank = 0;                            // pairs of this query with a better key
                    unsigned long long rest = em;
                    while (rest) {
                        const int j = __ffsll((unsigned long long)rest) - 1;
                        rest &= rest - 1;
                        const uint32_t uj = (uint32_t)__shfl((int)u, j, 64), idj = (uint32_t)__shfl((int)id, j, 64);
                        if (uj > u || (uj == u && idj < id)) ++rank;
                    }
                    const unsigned long long pick = __ballot(mine && rank == need - 1);
                    const int src = __ffsll((unsigned long long)pick) - 1;
                    const uint32_t uT = (uint32_t)__shfl((int)u, src, 64), iT = (uint32_t)__shfl((int)id, src, 64);
                    if (lane == 0) {
                        q_uT[qa] = uT;
                        q_iT[qa] = iT;
                    }
                }
            }
        }
    }
    __syncthreads();
    FAL_STAMP(6)
    if (!active || (dbg & 16)) return;

    // ================= rank the selected candidates of every query (lane-private), stage the rows, write them ============
    const int keep = a.keep;
    const int scols = min(keep, 64);                         // staged columns (the kept lists hold <= 48 entries)
    uint32_t* out_i = reinterpret_cast<uint32_t*>(whist);    // [32][scols] ids, then [32][scols] distances
    float* out_d = reinterpret_cast<float*>(whist) + 32 * scols;
    for (int e = lane; e < 32 * scols; e += 64) {
        out_i[e] = 0xFFFFFFFFu;                              // -1
        out_d[e] = INFINITY;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    {
        const int ql = r;
        const int f = q_flag[ql * 2] | q_flag[ql * 2 + 1];
        const int k0 = min(q_kcnt[ql * 2], kKeepHalf), k1 = min(q_kcnt[ql * 2 + 1], kKeepHalf);
        const uint32_t uT = q_uT[ql], iT = q_iT[ql];
        const uint32_t* kuq = kept_u + ql * kKeepStride;
        const uint32_t* kiq = kept_id + ql * kKeepStride;
        // an entry survives unless it was ambiguous and lost against the exact k-th key
        auto alive = [&](uint32_t u, uint32_t idf) -> bool {
            if (!(idf & 0x80000000u)) return true;
            const uint32_t id = idf & 0x7FFFFFFFu;
            return u > uT || (u == uT && id <= iT);
        };
        int total = 0;
        if (ql < nqw && !(f & 2)) {
            const int mine = h ? k1 : k0;
            for (int e = 0; e < mine; ++e) {
                const uint32_t ue = kuq[h * kKeepHalf + e], ie = kiq[h * kKeepHalf + e];
                if (!alive(ue, ie)) continue;
                const uint32_t ide = ie & 0x7FFFFFFFu;
                int rank = 0;                                // surviving entries with a better key (sim desc, id asc)
                for (int j = 0; j < k0; ++j) {
                    const uint32_t uj = kuq[j], ij = kiq[j];
                    rank += (alive(uj, ij) && (uj > ue || (uj == ue && (ij & 0x7FFFFFFFu) < ide))) ? 1 : 0;
                }
                for (int j = 0; j < k1; ++j) {
                    const uint32_t uj = kuq[kKeepHalf + j], ij = kiq[kKeepHalf + j];
                    rank += (alive(uj, ij) && (uj > ue || (uj == ue && (ij & 0x7FFFFFFFu) < ide))) ? 1 : 0;
                }
                if (rank < scols) {
                    out_i[ql * scols + rank] = ide;
                    out_d[ql * scols + rank] = fminf(fmaxf(1.0f - sortable_f32(ue), 0.f), 1.f);
                }
                ++total;
            }
        }
        const int other = __shfl(total, lane ^ 32, 64);
        if (a.nb_count && h == 0 && ql < nqw && !(f & 2)) a.nb_count[row0 + qbase + ql] = min(total + other, keep);
        if (h == 0 && ql < nqw && (f & 2) && !(dbg & 64)) push_fallback(a, row0 + qbase + ql, ji);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // rows of the wave's queries are contiguous in the output arrays
    {
        int32_t* gi = a.nb_idx + (row0 + qbase) * (int64_t)keep;
        float* gd = a.nb_dist + (row0 + qbase) * (int64_t)keep;
        for (int qq = 0; qq < nqw; ++qq)
            for (int col = lane; col < keep; col += 64) {
                gi[qq * keep + col] = col < scols ? (int32_t)out_i[qq * scols + col] : -1;
                gd[qq * keep + col] = col < scols ? out_d[qq * scols + col] : INFINITY;
            }
    }
    FAL_STAMP(7)
#undef FAL_STAMP
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
    const int fb_grid = ctx->num_cus * 16;
    const int64_t stride = (((int64_t)max_nc + 63) & ~63ll) + (int64_t)kSimsSlack;
    int32_t* fb = nullptr;
    float* scratch = nullptr;
    const int fb_cap = 1 << 22;
    FAL_TRY(ctx->reserve(SLOT_FUSED, sizeof(int32_t) * (size_t)(2 * fb_cap + 16), (void**)&fb));
    FAL_TRY(ctx->reserve(SLOT_FUSED2, sizeof(float) * (size_t)fb_grid * (size_t)stride, (void**)&scratch));
    a.stamps = nullptr;
    if (a.dbg & 128) {
        FAL_TRY(ctx->reserve(SLOT_MISC2, sizeof(unsigned long long) * 10 * (size_t)(list_tiles * 8), (void**)&a.stamps));
        FAL_CHECK_HIP(hipMemsetAsync(a.stamps, 0, sizeof(unsigned long long) * 10 * (size_t)(list_tiles * 8), ctx->stream));
        ctx->counters[6] = (int64_t)(uintptr_t)a.stamps;
        ctx->counters[7] = list_tiles * 8;
    }
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
