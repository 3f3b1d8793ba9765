// Shared device pieces of the cosine kernels (a6 assign, a7 coarse + fine scan).
//
// One 64-lane wave owns a tile of 32 query rows, kept ENTIRELY IN REGISTERS for the whole
// tile (d/2 floats per lane: lane (r = l&31, h = l>>5) holds query r's k-half h), and streams
// candidate rows 32 at a time straight from global memory / L2 into VGPRs with 16-byte loads
// (each lane reads ITS OWN candidate row, so candidates may be any gather of rows).  The inner
// products run on the fp32 matrix cores: v_mfma_f32_32x32x2_f32, one chain of d/2 MFMAs per
// 32x32 block.  The k-slot of MFMA step kk is {k = kk (lanes 0-31), k = d/2 + kk (lanes 32-63)};
// since both operands use the same slot->k map the result is the plain inner product, summed
// as the chain  fma(a[d/2+kk] , b[d/2+kk], fma(a[kk], b[kk], acc)),  kk = 0 .. d/2-1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Which output row (0..31) of a 32x32 MFMA result lives in accumulator register `reg` of a
// lane in half `h` (guide: row = (reg & 3) + 8 * (reg >> 2) + 4 * h; column = lane & 31).
__device__ __forceinline__ int mfma32_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Load one lane's k-half of a row into registers; slots beyond the real half are zero.
template <int DH4>
__device__ __forceinline__ void load_half_row(float (&q)[DH4 * 4], const float* __restrict__ row_half, int dh4) {
    const float4* p = reinterpret_cast<const float4*>(row_half);
#pragma unroll
    for (int j = 0; j < DH4; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < dh4) v = p[j];
        q[4 * j + 0] = v.x;
        q[4 * j + 1] = v.y;
        q[4 * j + 2] = v.z;
        q[4 * j + 3] = v.w;
    }
}

// Candidate stream of one lane: an NB-deep ring of 16-byte loads that stays in flight ACROSS
// 32-candidate chunks (the loads of chunk i+1 are issued while chunk i's MFMAs run), so the
// matrix pipe never waits for an L2 round trip.  hipcc would otherwise serialise load -> wait ->
// 4 MFMAs because the query registers leave it little room.
// The ring depth divides DH4 so that slot (j % kRing) means the same step in every chunk.
template <int DH4>
struct CandStream {
    static constexpr int kRing = (DH4 % 8 == 0) ? 8 : 10;   // (a 25-deep ring was measured: no gain)
    static_assert(DH4 % kRing == 0, "ring depth must divide the number of 16-byte steps per row half");
    float4 ring[kRing];

    __device__ __forceinline__ void prime(const float* __restrict__ row_half, int dh4) {
        const float4* p = reinterpret_cast<const float4*>(row_half);
#pragma unroll
        for (int j = 0; j < kRing; ++j) ring[j] = p[j < dh4 ? j : dh4 - 1];
    }

    // acc[32x32] = rows(a-side) x rows(b-side)^T over the wave's k-slots for the chunk whose
    // row is already streaming; `next_half` = this lane's row of the NEXT chunk (any valid row
    // when there is none).
    // QUERY_IS_A = true : D[query][cand]  (lane: candidate = lane&31, 16 query rows in registers)
    // QUERY_IS_A = false: D[cand][query]  (lane: query = lane&31, 16 candidate rows in registers)
    // `mid()` runs once, kMidStep steps into the chunk: callers put the PREVIOUS chunk's epilogue
    // there, so its stores / VALU work overlap this chunk's MFMAs and are long retired before any
    // load issued after them is waited on (gfx950's vmcnt counts stores and loads together).
    static constexpr int kMidStep = (DH4 > 24) ? 12 : DH4 / 2;

    template <bool QUERY_IS_A, class Mid>
    __device__ __forceinline__ f32x16 dot(const float (&q)[DH4 * 4], const float* __restrict__ cur_half,
                                          const float* __restrict__ next_half, int dh4, Mid&& mid) {
        const float4* pc = reinterpret_cast<const float4*>(cur_half);
        const float4* pn = reinterpret_cast<const float4*>(next_half);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int j = 0; j < DH4; ++j) {
            // padded slots re-read the last real float4: q is zero there, so they add nothing
            const float4 a = ring[j % kRing];
            const int jj = j + kRing;
            if (jj < DH4) ring[j % kRing] = pc[jj < dh4 ? jj : dh4 - 1];
            else ring[j % kRing] = pn[(jj - DH4) < dh4 ? (jj - DH4) : dh4 - 1];
            if (QUERY_IS_A) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 0], a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 1], a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 2], a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 3], a.w, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q[4 * j + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q[4 * j + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q[4 * j + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q[4 * j + 3], acc, 0, 0, 0);
            }
            // pin the schedule: one 16-byte load, then this step's four MFMAs (keeps kRing loads
            // in flight instead of letting the scheduler sink them next to their use)
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j == kMidStep) mid();
        }
        // The last MFMA needs 16 passes before its accumulators may be read.  hipcc's hazard recognizer counts the wait
        // states along the fall-through path only: with a taken branch right behind the chain (band_kernel<16>) the
        // first v_accvgpr_read came 12 states after the MFMA and saw the sum WITHOUT the last k-slot pair (found by
        // tests/test_gpu_fused.py at low_dim 128).  24 explicit wait states cost 0.2 % of a chunk.
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc));      // (tied to acc: cannot move in front of the last MFMA)
        return acc;
    }
};

// The exact chain of (row, other vector) over the ROW's sparse form (ivf.h: <= 64 (column, value) entries in chain order, column
// 0xFFFF = unused, 0xFFFE in entry 0 = read the dense row) held across the wave (lane e = entry e: every lane of the wave
// evaluates the SAME row against its own dense vector `c` -- a centroid, a member row): the entries come as scalars (v_readlane), the lane gathers c[column] --
// 16 independent loads per batch, four batches at most.  Zero row components contribute nothing to the dense chain (0 * c = +0,
// centroids are non-negative), so the bits are those of exact_dot (coarse16.hip / pairs16.hip use the same fact).
__device__ __forceinline__ float sparse_row_chain(uint32_t col_lane, float val_lane, int n_ent, const float* __restrict__ c) {
    float acc = 0.f;
    const int vbits = __float_as_int(val_lane);
#pragma unroll
    for (int e0 = 0; e0 < 64; e0 += 16) {
        if (e0 >= n_ent) break;                                      // (wave-uniform: one row per wave)
        float cv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)col_lane, e0 + t);
            cv[t] = c[ce < 0xFFFEu ? ce : 0u];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)col_lane, e0 + t);
            const float ve = ce < 0xFFFEu ? __int_as_float(__builtin_amdgcn_readlane(vbits, e0 + t)) : 0.f;
            acc = __builtin_fmaf(ve, cv[t], acc);
        }
    }
    return acc;
}

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

// float -> uint32 whose unsigned order equals the float order (and back).
__device__ __forceinline__ uint32_t f32_sortable(float f) {
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(uint32_t u) {
    uint32_t b = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(b);
}

// One unit of dense work: the queries [q_row0, q_row0+nq) of array Q against the candidates
// [c_row0, c_row0+nc) of array C.  Tiles of 32 queries; tile0 = index of the job's first tile
// in the launch; obase = where the job's sims start (float index): one [32, ceil32(nc)] block per tile.
struct DenseJob {
    int64_t q_row0;
    int64_t c_row0;
    int64_t obase;
    int64_t tile0;
    int32_t nq;
    int32_t nc;
    int64_t xtile0;   // XCD-list mode: tiles of earlier jobs of the same XCD list (jobs j, j+8, j+16 ...)
};

// binary search: last job whose tile0 <= t
__device__ __forceinline__ int find_job(const DenseJob* __restrict__ jobs, int n_jobs, int64_t t) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// XCD-list mode.  Workgroups are dealt round-robin over the 8 XCDs in blockIdx order, so a launch
// is only as fast as its most loaded XCD.  Jobs (buckets) arrive sorted by decreasing size and job
// j of the launch belongs to XCD list j % 8: every XCD gets the same mix of sizes (LPT-like) while
// all tiles of a bucket stay on ONE XCD (its rows are fetched into one L2 only).  Block b serves
// tile (b >> 3) of list (b & 7).  Returns false when that list is shorter.
__device__ __forceinline__ bool find_job_xcd(const DenseJob* __restrict__ jobs, int n_jobs, unsigned bid,
                                             int* job_index, int* local_tile) {
    const int x = bid & 7;
    const int64_t i = bid >> 3;
    const int cnt = (n_jobs - x + 7) >> 3;          // jobs x, x+8, ... < n_jobs
    if (cnt <= 0) return false;
    int lo = 0, hi = cnt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[x + 8 * mid].xtile0 <= i) lo = mid; else hi = mid - 1;
    }
    const DenseJob& j = jobs[x + 8 * lo];
    const int64_t lt = i - j.xtile0;
    if (lt >= (j.nq + 31) / 32) return false;
    *job_index = x + 8 * lo;
    *local_tile = (int)lt;
    return true;
}

}  // namespace fal
