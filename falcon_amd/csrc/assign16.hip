// a6 with a float16 prefilter: k-means / final list assignment for buckets with <= 128 lists.
//
// The exact assignment (assign.hip) runs the fp32 matrix cores at 2 * d * n_list flop per row and iteration -- the largest
// single cost of the IVF regime (10 M spectra: 11 passes x 1.02 TFLOP).  A row's list is the arg-max of its inner
// products with the bucket's centroids; the arg-max only needs EXACT values where the two best are close.  So:
//   1. assign16_kernel: float16 copies of the rows (fal_vectorize FAL_DTYPE_F16) against float16 copies of the
//      centroids on the f16 matrix cores (float32 accumulation, 1/16 of the fp32 cycles): best and runner-up per row.
//      |approx - exact| <= eps(v) = 1.3e-3 v + 2e-6 (bound derived in fused.hip): if runner-up < best - 2.2 eps(best) the
//      approximate winner IS the exact arg-max and is written; otherwise the row goes to a list;
//   2. the listed rows (a few per cent) by the exact k-ordered fmaf chain (VALU, bit-identical to the matrix-core chain), arg-max
//      with ties -> lowest id: assign_exact_pairs_kernel for rows with at most one contender (approximate value within the
//      bound of the best) per 32-centroid group -- almost all of them: just the contenders, four lanes per row --,
//      assign_exact_rows_kernel against all centroids for the rest.
// The assignment -- and with it every centroid, list and search result -- is IDENTICAL to the exact kernels'
// (tests/test_gpu_search.py builds both and compares bit for bit).  Reference: README.md:132-136 (Faiss IVF train / add).
#include <hip/hip_fp16.h>
#include <math.h>
#include <algorithm>
#include "common.h"
#include "ivf.h"
#include "scan.h"
#include "simtile.h"

namespace fal {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr float kA16EpsRel = 1.3e-3f, kA16EpsAbs = 2e-6f;
constexpr uint32_t kAmbMerged = 0x80000000u;          // amb_list entry [4]: a row of a merged (> 128-list) bucket, see assign_exact_rows_kernel

__global__ void cvt_f16_kernel(const float* __restrict__ in, __half* __restrict__ out, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
        reinterpret_cast<__half2*>(out)[2 * i] = a;
        reinterpret_cast<__half2*>(out)[2 * i + 1] = b;
    }
}

struct Assign16Args {
    const __half* X16;       // [n, d] float16 rows (sorted order)
    const __half* C16;       // [total_lists, d] float16 centroids
    const AssignJob* jobs;   // (row segment) x (ALL lists of the bucket, <= 128)
    int64_t n_jobs;
    int32_t* assign;         // [n]
    int32_t* amb_list;       // (row, job, 4 contender ids, full-group mask) left to assign_exact_rows_kernel: the 32 centroids of
                             // every group in the mask + the contenders (mask 0: ALL centroids of the bucket)
    int32_t* amb_count;      // [0] entries of amb_list, [1] entries of pair_list
    int amb_cap;
    int32_t* pair_list;      // (row, job, <= 4 contender ids packed in bytes, 0xFF = none): rows with one contender per
                             // 32-centroid group at most -- the exact kernel evaluates just those
    uint16_t* ckeys;         // optional [n, ckeys_stride]: round(approximate similarity * 65535) of every (row, centroid) pair -- the
    int ckeys_stride;        // final pass leaves them for the coarse quantiser of the search (coarse16.hip)
    // buckets with more than 128 lists (<= kAssignMergeLists): one job per (row segment, group of 128 centroids) leaves the best /
    // runner-up / best id of its four 32-centroid subgroups per row (PARTIAL), assign16_merge_kernel decides over all subgroups
    // of the row
    float* part_b;           // [n_sub, n] best value of subgroup sg = 4 * group + wave (n_sub = the launch's most lists / 32)
    float* part_s;           // [n_sub, n] its runner-up
    int32_t* part_id;        // [n_sub, n] bucket-local id of its best
    int64_t n;               // rows of the merge buckets (stride of the partial arrays; a row's index: job.part0 + row - job.row0)
    const AssignJob* mjobs;  // merge jobs: (row segment) x (ALL lists of the bucket); the exact kernels' entries refer to these
    int64_t n_mjobs;
    const uint16_t* sp_cols; // optional [n, 64] the rows' sparse form (ivf.h): assign_exact_rows_kernel walks a row's entries
    const float* sp_vals;    // against the dense centroids instead of low_dim terms
};


__device__ __forceinline__ int a16_rowoff(int i) { return (i & 3) + 8 * (i >> 2); }

template <int STEPS, bool KEYS, bool PARTIAL>
__global__ __launch_bounds__(256, STEPS > 32 ? 1 : 2) void assign16_kernel(Assign16Args a) {
    constexpr int D = STEPS * 16, DH = D / 2;
    constexpr int RB16 = D / 8;
    constexpr int RS = D * 2 + 16;
    constexpr int PIECES = 32 * RB16;
    constexpr int kStage = (PIECES + 255) / 256;
    constexpr int NB = STEPS < 4 ? STEPS : 4;
    static_assert(kStage <= 13, "staging registers");      // (low_dim 800: 13 x 16 bytes per thread and chunk, one workgroup per CU)
    __shared__ __align__(16) unsigned char stage[2 * 32 * RS];
    __shared__ float r_best[2][4][32];       // per chunk parity, wave, row: best / runner-up value and best id
    __shared__ float r_second[2][4][32];
    __shared__ int r_id[2][4][32];
    const int64_t per_xcd = (a.n_jobs + 7) / 8;
    const int64_t ji = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || ji >= a.n_jobs) return;
    const AssignJob job = a.jobs[ji];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int nr = job.nrows;
    const int ncw = min(32, job.ncent - 32 * w);                 // centroids of this wave's tile (<= 0: only helps staging)
    half8 c16[STEPS];
    {
        const int cr = min(32 * w + min(r, max(ncw, 1) - 1), job.ncent - 1);
        const half8* src = reinterpret_cast<const half8*>(a.C16 + (job.cent0 + cr) * (int64_t)D + h * DH);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) c16[s] = src[s];
    }
    const __half* rbase = a.X16 + job.row0 * (int64_t)D;
    uint4 sa0, sa1, sa2, sa3, sa4, sa5, sa6, sa7, sa8, sa9, sa10, sa11, sa12;
#define FAL_FOR_A(M) M(0, sa0) M(1, sa1) M(2, sa2) M(3, sa3) M(4, sa4) M(5, sa5) M(6, sa6) M(7, sa7) M(8, sa8) M(9, sa9) \
    M(10, sa10) M(11, sa11) M(12, sa12)
#define FAL_LOAD_ONE(I, R)                                                                             \
    if constexpr (I < kStage) {                                                                        \
        const int idx = min((int)threadIdx.x + 256 * I, PIECES - 1);                                   \
        const int row = idx / RB16, col = idx - row * RB16;                                            \
        R = reinterpret_cast<const uint4*>(rbase + (int64_t)min(stage_c0 + row, nr - 1) * D)[col];     \
    }
#define FAL_STORE_ONE(I, R)                                                                            \
    if constexpr (I < kStage) {                                                                        \
        const int idx = min((int)threadIdx.x + 256 * I, PIECES - 1);                                   \
        const int row = idx / RB16, col = idx - row * RB16;                                            \
        *reinterpret_cast<uint4*>(stage + (size_t)stage_buf * 32 * RS + row * RS + col * 16) = R;      \
    }
#define FAL_LOAD(C0) { const int stage_c0 = min((C0), nr - 1); FAL_FOR_A(FAL_LOAD_ONE) }
#define FAL_STORE(BUF) { const int stage_buf = (BUF); FAL_FOR_A(FAL_STORE_ONE) }
    // decision for the 32 rows of a finished chunk: lanes 0..31 of the duty wave combine the four waves' results
    auto decide = [&](int par, int c0) {
        if (lane >= 32 || c0 + lane >= nr) return;
        if constexpr (PARTIAL) {                                     // this group's four subgroup summaries of the row
            const int64_t row = job.row0 + c0 + lane;
            const int g = job.id_base >> 7;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const int64_t at = (int64_t)(4 * g + ww) * a.n + (job.part0 + c0 + lane);
                a.part_b[at] = r_best[par][ww][lane];
                a.part_s[at] = r_second[par][ww][lane];
                a.part_id[at] = job.id_base + r_id[par][ww][lane];
            }
            return;
        }
        float best = -INFINITY, second = -INFINITY;
        int bid = 0x7fffffff;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float b = r_best[par][ww][lane], s2 = r_second[par][ww][lane];
            const int id = r_id[par][ww][lane];
            if (b > best || (b == best && id < bid)) {
                second = fmaxf(second, best);
                best = b;
                bid = id;
            } else {
                second = fmaxf(second, b);
            }
            second = fmaxf(second, s2);
        }
        const int64_t row = job.row0 + c0 + lane;
        const float eps = kA16EpsRel * best + kA16EpsAbs;
        const float thr = best - 2.2f * eps;
        if (second < thr) {
            a.assign[row] = job.id_base + bid;
        } else {
            // too close to call in float16: exact re-evaluation of the contenders -- the centroids whose approximate value
            // reaches thr.  A 32-centroid group whose runner-up stays below thr contributes its best only; if a runner-up
            // reaches thr too, the group's other members are unknown here: all centroids are re-evaluated
            uint32_t pk[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};             // four contender ids, 16 bits each (0xFFFF = none)
            bool full = false;
            uint32_t full_groups = 0;                                // groups whose runner-up reaches thr too
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const float b = r_best[par][ww][lane], s2 = r_second[par][ww][lane];
                if (b >= thr) pk[ww >> 1] = (pk[ww >> 1] & ~(0xFFFFu << (16 * (ww & 1)))) | ((uint32_t)r_id[par][ww][lane] << (16 * (ww & 1)));
                full = full || s2 >= thr;
                full_groups |= s2 >= thr ? (1u << ww) : 0u;
            }
            if (!full) {
                const int at = atomicAdd(a.amb_count + 1, 1);
                if (at < a.amb_cap) {
                    a.pair_list[4 * at] = (int32_t)row;
                    a.pair_list[4 * at + 1] = (int32_t)ji;
                    a.pair_list[4 * at + 2] = (int32_t)pk[0];
                    a.pair_list[4 * at + 3] = (int32_t)pk[1];
                } else {
                    full = true;
                }
            }
            if (full) {
                // only the groups whose runner-up reaches thr can hide further candidates: their 32 centroids + the other
                // groups' contenders are re-evaluated (all centroids of the bucket: 3.2 ms and 20 GB per 10 M build)
                const int at = atomicAdd(a.amb_count, 1);
                if (at < a.amb_cap) {
                    a.amb_list[5 * at] = (int32_t)row;
                    a.amb_list[5 * at + 1] = (int32_t)ji;
                    a.amb_list[5 * at + 2] = (int32_t)pk[0];
                    a.amb_list[5 * at + 3] = (int32_t)pk[1];
                    a.amb_list[5 * at + 4] = (int32_t)full_groups;
                }
            }
        }
    };
    FAL_LOAD(0)
    FAL_STORE(0)
    FAL_LOAD(32)
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < nr; c0 += 32) {
        const unsigned char* rowp = stage + (size_t)buf * 32 * RS + r * RS + h * DH * 2;
        half8 rh[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) rh[s] = *reinterpret_cast<const half8*>(rowp + s * 16);
        __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const half8 rv = rh[s % NB];
            if (s + NB < STEPS) rh[s % NB] = *reinterpret_cast<const half8*>(rowp + (s + NB) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(c16[s], rv, acc, 0, 0, 0);      // D[centroid][row]: lane = row
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (KEYS) {                                        // (the final pass; four keys per 8-byte store, no branch: rows
            //  past the segment re-write the last row's keys with the same values)
            uint16_t* kp = a.ckeys + (job.row0 + min(c0 + r, nr - 1)) * (int64_t)a.ckeys_stride + job.id_base + 32 * w + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t lo2 = 0, hi2 = 0;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const uint32_t key = __float2uint_rn(__builtin_amdgcn_fmed3f(acc[4 * g + x], 0.f, 1.f) * 65535.f);
                    if (x < 2) lo2 |= key << (16 * x); else hi2 |= key << (16 * (x - 2));
                }
                *reinterpret_cast<uint2*>(kp + 8 * g) = make_uint2(lo2, hi2);
            }
        }
        // this lane's row against its 16 centroids (of the wave's 32): best, runner-up
        float best = -INFINITY, second = -INFINITY;
        int bid = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = a16_rowoff(i) + 4 * h;
            const float v = ci < ncw ? acc[i] : -INFINITY;
            if (v > best) {                                          // (ids ascend with i: an equal later value is a runner-up)
                second = best;
                best = v;
                bid = 32 * w + ci;
            } else {
                second = fmaxf(second, v);
            }
        }
        {
            const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
            const int oi = __shfl_xor(bid, 32, 64);
            if (ob > best || (ob == best && oi < bid)) {
                second = fmaxf(best, os);
                best = ob;
                bid = oi;
            } else {
                second = fmaxf(second, ob);
            }
        }
        const int par = (c0 >> 5) & 1;
        if (h == 0) {
            r_best[par][w][r] = best;
            r_second[par][w][r] = second;
            r_id[par][w][r] = bid;
        }
        FAL_STORE(buf ^ 1)                       // chunk c0 + 32
        FAL_LOAD(c0 + 64)
        __syncthreads();
        if (w == ((c0 >> 5) & 3)) decide(par, c0);                   // (the duty rotates over the waves)
        buf ^= 1;
    }
#undef FAL_LOAD
#undef FAL_STORE
#undef FAL_LOAD_ONE
#undef FAL_STORE_ONE
#undef FAL_FOR_A
}

// the rows the prefilter could not decide and could not name the contenders of: exact arg-max over the candidates that are left
// -- all 32 centroids of every group in the entry's mask + the named contenders of the other groups (mask 0: every centroid of
// the bucket).  Anything else has an approximate value below thr = best - 2.2 eps and cannot be the exact arg-max.
__global__ __launch_bounds__(64) void assign_exact_rows_kernel(Assign16Args a, const float* __restrict__ X, const float* __restrict__ Cn,
                                                               int d) {
    const int lane = threadIdx.x;
    const int total = min(*a.amb_count, a.amb_cap);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t row = a.amb_list[5 * t];
        const AssignJob job = a.jobs[a.amb_list[5 * t + 1]];
        const uint32_t pk[2] = {(uint32_t)a.amb_list[5 * t + 2], (uint32_t)a.amb_list[5 * t + 3]};
        const uint32_t fg = (uint32_t)a.amb_list[5 * t + 4];
        float best = -INFINITY;
        int bid = 0x7fffffff;
        uint32_t col_lane = kColDense;
        float val_lane = 0.f;
        if (a.sp_cols) {
            col_lane = a.sp_cols[row * kSparseW + lane];
            val_lane = a.sp_vals[row * kSparseW + lane];
        }
        const bool sparse = (uint32_t)__builtin_amdgcn_readlane((int)col_lane, 0) != (uint32_t)kColDense;    // (else: > 64 non-zeros)
        const int n_ent = __popcll(__ballot(col_lane < (uint32_t)kColDense));
        // Every chain runs under FULL EXEC: sparse_row_chain fetches the row's entries with v_readlane from lanes 0 .. n_ent - 1,
        // and a lane that a divergent branch had switched off would hand it an undefined register (the loops below are
        // wave-uniform, a lane without a candidate evaluates centroid 0 and drops the result).
        auto eval = [&](int c, bool valid) {
            const float* cp = Cn + (job.cent0 + (valid ? c : 0)) * d;
            const float s = sparse ? sparse_row_chain(col_lane, val_lane, n_ent, cp) : exact_dot(X + row * d, cp, d);
            if (valid && (s > best || (s == best && c < bid))) {
                best = s;
                bid = c;
            }
        };
        if (fg == kAmbMerged) {
            // a row of a bucket with more than 128 lists that the merge could not decide (round 5; before: every centroid of the
            // bucket, 1,024 chains per row): the row's subgroup summaries are still in the partial arrays -- thr from the best
            // of them (the merge kernel's own formula), then only the subgroups whose RUNNER-UP reaches thr are evaluated in
            // full (their other members are unknown), the others contribute their best if it reaches thr.  Any centroid left
            // out has an approximate value below thr and cannot be the exact arg-max.
            const int nsg = (job.ncent + 31) >> 5;
            float pb = -INFINITY, ps = -INFINITY;
            int pid = 0;
            if (lane < nsg) {
                const int64_t at = (int64_t)lane * a.n + (job.part0 + (row - job.row0));
                pb = a.part_b[at];
                ps = a.part_s[at];
                pid = a.part_id[at];
            }
            float top = pb;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) top = fmaxf(top, __shfl_xor(top, off, 64));
            const float thr = top - 2.2f * (kA16EpsRel * top + kA16EpsAbs);
            unsigned long long fm = __ballot(lane < nsg && ps >= thr);
            const unsigned long long sm = __ballot(lane < nsg && pb >= thr && !(ps >= thr));
            while (fm) {                                                 // two full subgroups per round: lanes 0-31 / 32-63
                const int g0 = __ffsll((long long)fm) - 1;
                fm &= fm - 1;
                const int g1 = fm ? __ffsll((long long)fm) - 1 : -1;
                fm &= fm ? fm - 1 : 0ull;
                const int g = lane < 32 ? g0 : g1;
                const int c = g >= 0 ? 32 * g + (lane & 31) : -1;
                eval(c, c >= 0 && c < job.ncent);
            }
            const int n_single = __popcll(sm);                           // (<= 64: one lane each)
            if (n_single > 0) {
                const int src = __fns64(sm, 0, lane + 1);               // the lane of the (lane + 1)-th single subgroup, or -1
                const int c = __shfl(pid, src < 0 ? 0 : src, 64) - job.id_base;
                eval(c, src >= 0 && c >= 0 && c < job.ncent);
            }
        } else if (fg == 0u) {
            for (int c0 = 0; c0 < job.ncent; c0 += 64) eval(c0 + lane, c0 + lane < job.ncent);
        } else {
            const int nfull = 32 * __popc(fg);
            for (int i0 = 0; i0 < nfull + 4; i0 += 64) {
                const int i = i0 + lane;
                int c = -1;
                if (i < nfull) {
                    int g = 0, k = i >> 5;                             // the k-th set bit of the mask
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const bool on = (fg >> b) & 1u;
                        g = (on && k == 0) ? b : g;
                        k -= on ? 1 : 0;
                    }
                    c = 32 * g + (i & 31);
                } else if (i < nfull + 4) {
                    const int slot = i - nfull;
                    const uint32_t id = (pk[slot >> 1] >> (16 * (slot & 1))) & 0xFFFFu;
                    c = id == 0xFFFFu ? -1 : (int)id;
                }
                eval(c, c >= 0 && c < job.ncent);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bid, off, 64);
            if (ob > best || (ob == best && oi < bid)) {
                best = ob;
                bid = oi;
            }
        }
        if (lane == 0) a.assign[row] = job.id_base + bid;
    }
}

// the rows with at most one contender per 32-centroid group: four lanes per row, one exact chain each
__global__ __launch_bounds__(256) void assign_exact_pairs_kernel(Assign16Args a, const float* __restrict__ X, const float* __restrict__ Cn,
                                                                 int d) {
    const int total = min(a.amb_count[1], a.amb_cap);
    const int slot = threadIdx.x & 3;
    for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 2; e < (((int64_t)total + 15) & ~15ll);
         e += ((int64_t)gridDim.x * blockDim.x) >> 2) {
        const bool live = e < total;
        const int64_t row = live ? a.pair_list[4 * e] : 0;
        const AssignJob job = a.jobs[live ? a.pair_list[4 * e + 1] : 0];
        const uint32_t packed = live ? (uint32_t)a.pair_list[4 * e + 2 + (slot >> 1)] : 0xFFFFFFFFu;
        int bid = (int)((packed >> (16 * (slot & 1))) & 0xFFFFu);
        float best = -INFINITY;
        if (bid != 0xFFFF) best = exact_dot(X + row * d, Cn + (job.cent0 + bid) * d, d);
        else bid = 0x7fffffff;
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {                    // (ties -> lowest id)
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bid, off, 64);
            if (ob > best || (ob == best && oi < bid)) {
                best = ob;
                bid = oi;
            }
        }
        if (live && slot == 0) a.assign[row] = job.id_base + bid;
    }
}

// buckets with more than 128 lists: the decision over ALL 32-centroid subgroups of a row.  One thread per row of a merge job
// (row segment x all lists of the bucket): the four best subgroup winners and the largest value among everything else (the
// other winners and every subgroup's runner-up -- any other centroid of a subgroup lies at or below its runner-up).  Winner
// certain if everything but the best stays below thr; else the winners that reach thr are the contenders (exact pairs), unless
// "everything else" reaches thr too: then the contenders cannot be named and all centroids are re-evaluated.
__global__ __launch_bounds__(256) void assign16_merge_kernel(Assign16Args a, int job_index0) {
    const int64_t per_xcd = (a.n_mjobs + 7) / 8;
    const int64_t ji = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || ji >= a.n_mjobs) return;
    const AssignJob job = a.mjobs[ji];
    const int nsg = (job.ncent + 31) >> 5;
    for (int i = threadIdx.x; i < job.nrows; i += blockDim.x) {
        const int64_t row = job.row0 + i;
        float tv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int ti[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        float rest = -INFINITY;
        for (int sg = 0; sg < nsg; ++sg) {
            const int64_t at = (int64_t)sg * a.n + (job.part0 + i);
            float v = a.part_b[at];
            int id = a.part_id[at];
            rest = fmaxf(rest, a.part_s[at]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {                            // insert into the sorted four (value desc, id asc)
                if (v > tv[k] || (v == tv[k] && id < ti[k])) {
                    const float ov = tv[k];
                    const int oi = ti[k];
                    tv[k] = v;
                    ti[k] = id;
                    v = ov;
                    id = oi;
                }
            }
            rest = fmaxf(rest, v);                                   // what fell out of the four
        }
        const float eps = kA16EpsRel * tv[0] + kA16EpsAbs;
        const float thr = tv[0] - 2.2f * eps;
        if (fmaxf(tv[1], rest) < thr) {
            a.assign[row] = ti[0];
        } else if (rest < thr) {
            uint32_t pk[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tv[k] >= thr) pk[k >> 1] = (pk[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | ((uint32_t)ti[k] << (16 * (k & 1)));
            const int at = atomicAdd(a.amb_count + 1, 1);
            bool ok = at < a.amb_cap;
            if (ok) {
                a.pair_list[4 * at] = (int32_t)row;
                a.pair_list[4 * at + 1] = (int32_t)(job_index0 + ji);
                a.pair_list[4 * at + 2] = (int32_t)pk[0];
                a.pair_list[4 * at + 3] = (int32_t)pk[1];
            } else {
                const int at2 = atomicAdd(a.amb_count, 1);
                if (at2 < a.amb_cap) {
                    a.amb_list[5 * at2] = (int32_t)row;
                    a.amb_list[5 * at2 + 1] = (int32_t)(job_index0 + ji);
                    a.amb_list[5 * at2 + 4] = 0;
                }
            }
        } else {
            const int at = atomicAdd(a.amb_count, 1);
            if (at < a.amb_cap) {
                a.amb_list[5 * at] = (int32_t)row;
                a.amb_list[5 * at + 1] = (int32_t)(job_index0 + ji);
                a.amb_list[5 * at + 4] = (int32_t)kAmbMerged;         // (assign_exact_rows_kernel: consult the partial arrays)
            }
        }
    }
}

bool assign16_supports(int d) { return d == 64 || d == 128 || d == 256 || d == 400 || d == 800; }

int launch_cvt_f16(fal_ctx* ctx, const float* in, void* out, int64_t count) {
    if (count <= 0) return FAL_OK;
    const int64_t n4 = count / 4;
    hipLaunchKernelGGL(cvt_f16_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n4, 256), (int64_t)ctx->num_cus * 16)), dim3(256), 0,
                       ctx->stream, in, reinterpret_cast<__half*>(out), n4);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

// jobs: device table of `n_jobs` jobs, each covering ALL (<= 128) lists of its bucket
// jobs: device table [single jobs (n_single): (row segment) x (ALL <= 128 lists of the bucket)] [merge jobs (n_merge): (row
// segment) x (all 129..kAssignMergeLists lists)] [group jobs (n_group): (row segment) x (128 of those lists), the groups of a segment next
// to each other]
int launch_assign16(fal_ctx* ctx, int stage, const void* X16, const float* X, const void* C16, const float* Cn, int d,
                    const AssignJob* jobs, int64_t n_single, int64_t n_merge, int64_t n_group, int64_t n_rows, int32_t* assign,
                    uint16_t* ckeys, int ckeys_stride, const uint16_t* sp_cols, const float* sp_vals, int merge_max_lists,
                    int64_t merge_rows) {
    if (n_single + n_merge <= 0) return FAL_OK;
    // work lists of the exact kernels: a row enters at most one of them once per pass, so n_rows entries never overflow
    int32_t* amb = nullptr;
    const int amb_cap = (int)std::max<int64_t>(n_rows, 1);
    FAL_TRY(ctx->reserve(SLOT_FUSED, sizeof(int32_t) * (size_t)(9 * (size_t)amb_cap + 16), (void**)&amb));
    FAL_CHECK_HIP(hipMemsetAsync(amb, 0, sizeof(int32_t) * 16, ctx->stream));
    Assign16Args a{};
    a.X16 = reinterpret_cast<const __half*>(X16); a.C16 = reinterpret_cast<const __half*>(C16);
    a.jobs = jobs; a.n_jobs = n_single; a.assign = assign; a.amb_list = amb + 16; a.amb_count = amb; a.amb_cap = amb_cap;
    a.pair_list = amb + 16 + 5 * (size_t)amb_cap; a.ckeys = ckeys; a.ckeys_stride = ckeys_stride; a.n = std::max<int64_t>(merge_rows, 1);
    a.sp_cols = sp_cols; a.sp_vals = sp_vals;
    StageScope ts(ctx, stage);
    const dim3 block(256);
#define FAL_LAUNCH_A16(S)                                                                                  \
    do {                                                                                                   \
        if (partial && ckeys) hipLaunchKernelGGL((assign16_kernel<S, true, true>), grid, block, 0, ctx->stream, a); \
        else if (partial) hipLaunchKernelGGL((assign16_kernel<S, false, true>), grid, block, 0, ctx->stream, a); \
        else if (ckeys) hipLaunchKernelGGL((assign16_kernel<S, true, false>), grid, block, 0, ctx->stream, a); \
        else hipLaunchKernelGGL((assign16_kernel<S, false, false>), grid, block, 0, ctx->stream, a);       \
    } while (0)
    auto run = [&](bool partial) -> int {
        const dim3 grid((unsigned)(((a.n_jobs + 7) / 8) * 8));
        switch (d / 16) {
            case 4: FAL_LAUNCH_A16(4); break;
            case 8: FAL_LAUNCH_A16(8); break;
            case 16: FAL_LAUNCH_A16(16); break;
            case 25: FAL_LAUNCH_A16(25); break;
            case 50: FAL_LAUNCH_A16(50); break;
            default: set_error("assign16: low_dim %d has no instantiation", d); return FAL_EUNSUPPORTED;
        }
        return FAL_OK;
    };
    if (n_single > 0) FAL_TRY(run(false));
    if (n_merge > 0) {
        float* part = nullptr;
        const size_t n_sub = 4 * (size_t)std::max(1, (merge_max_lists + kAssignGroup - 1) / kAssignGroup);      // four per group job, whole groups
        FAL_REQUIRE(merge_max_lists <= kAssignMergeLists, FAL_EINTERNAL, "assign16: a merge job with %d lists", merge_max_lists);
        // (sized by the rows of the MERGE buckets -- AssignJob::part0 --, not by the partition's: one 2,048-list bucket among 10 M
        // rows asked for 7.7 GB)
        const size_t n_part = (size_t)a.n;
        FAL_TRY(ctx->reserve(SLOT_PROBE_SIM, sizeof(float) * 3 * n_sub * n_part, (void**)&part));
        a.part_b = part; a.part_s = part + n_sub * n_part; a.part_id = reinterpret_cast<int32_t*>(part + 2 * n_sub * n_part);
        a.jobs = jobs + n_single + n_merge; a.n_jobs = n_group;
        FAL_TRY(run(true));
        a.mjobs = jobs + n_single; a.n_mjobs = n_merge;
        a.jobs = jobs;                       // the exact kernels' entries index [single | merge] jobs
        hipLaunchKernelGGL(assign16_merge_kernel, dim3((unsigned)(((n_merge + 7) / 8) * 8)), block, 0, ctx->stream, a, (int)n_single);
    }
#undef FAL_LAUNCH_A16
    a.jobs = jobs;
    hipLaunchKernelGGL(assign_exact_pairs_kernel, dim3((unsigned)(ctx->num_cus * 8)), dim3(256), 0, ctx->stream, a, X, Cn, d);
    hipLaunchKernelGGL(assign_exact_rows_kernel, dim3((unsigned)(ctx->num_cus * 16)), dim3(64), 0, ctx->stream, a, X, Cn, d);
    FAL_CHECK_HIP(hipGetLastError());
    FAL_CHECK_HIP(hipMemcpyAsync(ctx->fb_host + 1, amb, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));   // counter(6'): last pass
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::cvt_f16_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)
