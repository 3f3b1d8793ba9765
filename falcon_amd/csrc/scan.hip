// The cosine kernels: fp32-MFMA query x candidate inner products (dense form: k-means assign,
// coarse quantiser, flat-bucket scan) and the wavefront top-k select.
//
// Spec: reference README.md:107-113, 134-142 (Faiss IVF build / n_probe search; no code in the
// snapshot).  See simtile.h for the tile algorithm, DESIGN.md for the roofline.
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include "simtile.h"
#include "scan.h"
#include "ivf.h"

namespace fal {

// ---------------------------------------------------------------------------------------------
// dense tile kernel
// ---------------------------------------------------------------------------------------------
template <int DH4, int EPI>
__global__ __launch_bounds__(64, 1) void dense_kernel(
    const float* __restrict__ Q, const float* __restrict__ Cm, int d, const DenseJob* __restrict__ jobs,
    int n_jobs, int64_t tile_begin, int64_t n_tiles, float* __restrict__ sims, int64_t sims_base,
    int32_t* __restrict__ assign, int xcd_lists, int symmetric) {
    DenseJob job;
    int lt;
    int ji = 0;
    if (xcd_lists) {
        if (!find_job_xcd(jobs, n_jobs, blockIdx.x, &ji, &lt)) return;
        job = jobs[ji];
    } else {
        // contiguous run of tiles per XCD (neighbouring tiles scan the same bucket -> shared L2 lines)
        const int64_t per_xcd = (n_tiles + 7) / 8;
        const int64_t lt_all = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per_xcd || lt_all >= n_tiles) return;
        const int64_t t = tile_begin + lt_all;
        job = jobs[find_job(jobs, n_jobs, t)];
        lt = (int)(t - job.tile0);
    }
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nq_t = min(32, job.nq - 32 * lt);
    const int dh = d >> 1, dh4 = dh >> 2;

    float q[DH4 * 4];
    {
        const int64_t qrow = job.q_row0 + 32 * (int64_t)lt + min(r, nq_t - 1);
        load_half_row<DH4>(q, Q + qrow * d + (int64_t)h * dh, dh4);
    }
    float best = -INFINITY;
    int bestc = 0x7fffffff;
    const int nc = job.nc;
    float* out = nullptr;
    // the [32, ncp] block of this tile (ncp = nc rounded up to 32): every store below is in bounds,
    // so the epilogue has no branches and the compiler can count its stores in vmcnt exactly
    const int ncp = (nc + 31) & ~31;
    if (EPI == EPI_STORE) out = sims + (job.obase - sims_base) + (int64_t)(32 * lt) * ncp + r;

    // symmetric mode (a flat bucket scanned against itself): sim(i, j) == sim(j, i) BIT FOR BIT (same
    // k-ordered fmaf chain, commutative products), so this tile only computes the blocks on and above
    // its diagonal and writes every off-diagonal block twice, once transposed.  Half the MFMA work.
    const int c_first = (EPI == EPI_STORE && symmetric) ? 32 * lt : 0;
    float* outT = nullptr;      // transposed target: rows = this lane's candidate, columns = this tile's queries
    if (EPI == EPI_STORE && symmetric) outT = sims + (job.obase - sims_base) + (int64_t)r * ncp + 32 * lt + 4 * h;
    // Chunks are walked from the LAST one down to c_first: tiles of a bucket that start together then
    // read the same chunk at the same time (shared L2 lines) and simply stop at their own diagonal;
    // walking upwards from the diagonal would spread the running tiles over the whole bucket
    // (measured: 2.84 vs 2.60 ms of scan per 1 M spectra; alternating the direction per bucket: 2.68).
    // (The arg-max form keeps the ascending walk: with ids arriving in ascending order "s > best" alone breaks
    //  ties towards the lowest id; the longer tie test made hipcc demote the query registers to scratch, 3.5x
    //  slower k-means.)
    const int c_last = ((nc - 1) >> 5) << 5;
    constexpr bool kDown = EPI == EPI_STORE;
    const int c_begin = kDown ? c_last : 0, c_step = kDown ? -32 : 32;
    const int n_chunks = (c_last - c_first) / 32 + 1;
    CandStream<DH4> cs;
    const float* cur = Cm + (job.c_row0 + min(c_begin + r, nc - 1)) * d + (int64_t)h * dh;
    cs.prime(cur, dh4);
    f32x16 prev;                 // the previous chunk's result; its epilogue runs inside this chunk
    // before the first chunk "previous" is a dummy: zeros land in chunk 0's slots and are overwritten
    // by the real chunk-0 epilogue later in program order; -inf never wins the arg-max.  Keeping the
    // epilogue unconditional keeps its stores out of branches (exact vmcnt bookkeeping).
    int prev_c0 = c_begin;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = (EPI == EPI_STORE) ? 0.f : -INFINITY;
    auto epilogue = [&]() {
        if (EPI == EPI_STORE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) out[mfma32_row(i, h) * ncp + prev_c0] = prev[i];
            if (symmetric) {
                // registers 4g .. 4g+3 hold query rows 8g + 4h + 0..3 = four consecutive columns of the
                // transposed block: one 16-byte store each (the diagonal block is rewritten with itself)
                float* t = outT + (int64_t)prev_c0 * ncp;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(t + 8 * g) = make_float4(prev[4 * g], prev[4 * g + 1], prev[4 * g + 2], prev[4 * g + 3]);
                __builtin_amdgcn_sched_group_barrier(0x040, 20, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x040, 16, 0);   // keep the 16 stores HERE in the pipeline
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = prev_c0 + mfma32_row(i, h);
                const float s = prev[i];
                if (c < nc && s > best) {   // ids arrive in ascending order: ties keep the lowest id
                    best = s;
                    bestc = c;
                }
            }
        }
    };
    for (int ci = 0, c0 = c_begin; ci < n_chunks; ++ci, c0 += c_step) {
        const float* nxt = Cm + (job.c_row0 + min(max(c0 + c_step, 0) + r, nc - 1)) * d + (int64_t)h * dh;   // (clamped prefetch)
        const f32x16 acc = (EPI == EPI_STORE) ? cs.template dot<true>(q, cur, nxt, dh4, epilogue)
                                              : cs.template dot<false>(q, cur, nxt, dh4, epilogue);
        prev = acc;
        prev_c0 = c0;
        cur = nxt;
    }
    epilogue();
    if (EPI == EPI_ARGMAX) {
        const float ob = __shfl_xor(best, 32, 64);
        const int oc = __shfl_xor(bestc, 32, 64);
        if (ob > best || (ob == best && oc < bestc)) {
            best = ob;
            bestc = oc;
        }
        if (h == 0 && r < nq_t) assign[job.q_row0 + 32 * (int64_t)lt + r] = bestc;
    }
}

template <int EPI>
static int launch_dense_t(fal_ctx* ctx, int stage, const float* Q, const float* Cm, int d, const DenseJob* jobs,
                          int n_jobs, int64_t tile_begin, int64_t n_tiles, float* sims, int64_t sims_base,
                          int32_t* assign, int64_t xcd_list_tiles) {
    if (n_tiles <= 0) return FAL_OK;
    // a flat bucket against itself (XCD-list mode is only used for that): exploit the symmetry
    static const bool no_sym = getenv("FALCON_NO_SYMMETRY") != nullptr;
    const int symmetric = (EPI == EPI_STORE && xcd_list_tiles > 0 && Q == Cm && !no_sym) ? 1 : 0;
    const int dh4 = d / 8;
    const int xcd_lists = xcd_list_tiles > 0;
    // XCD-list mode: n_tiles is unused, the grid is 8 x (longest list)
    const int64_t per_xcd = xcd_lists ? xcd_list_tiles : (n_tiles + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
 dim3 grid((unsigned)(per_xcd * 8)), block(64);
    StageScope ts(ctx, stage);
#define FAL_LAUNCH_DENSE(DH4)                                                                              \
    hipLaunchKernelGGL((dense_kernel<DH4, EPI>), grid, block, 0, ctx->stream, Q, Cm, d, jobs, n_jobs,      \
                       tile_begin, n_tiles, sims, sims_base, assign, xcd_lists, symmetric)
    if (dh4 <= 8) FAL_LAUNCH_DENSE(8);
    else if (dh4 <= 16) FAL_LAUNCH_DENSE(16);
    else if (dh4 <= 32) FAL_LAUNCH_DENSE(32);
    else if (dh4 <= 50) FAL_LAUNCH_DENSE(50);
    else if (dh4 <= 64) FAL_LAUNCH_DENSE(64);
    else {
        set_error("float32 scan supports low_dim <= 512 (got %d)", d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_DENSE
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_dense(fal_ctx* ctx, int stage, int epi, const float* Q, const float* Cm, int d, const DenseJob* jobs,
                 int n_jobs, int64_t tile_begin, int64_t n_tiles, float* sims, int64_t sims_base, int32_t* assign,
                 int64_t xcd_list_tiles) {
    if (epi == EPI_STORE)
        return launch_dense_t<EPI_STORE>(ctx, stage, Q, Cm, d, jobs, n_jobs, tile_begin, n_tiles, sims, sims_base,
                                         assign, xcd_list_tiles);
    return launch_dense_t<EPI_ARGMAX>(ctx, stage, Q, Cm, d, jobs, n_jobs, tile_begin, n_tiles, sims, sims_base, assign,
                                      xcd_list_tiles);
}

// ---------------------------------------------------------------------------------------------
// wavefront top-k select: one wave per query over that query's row of sims
// ---------------------------------------------------------------------------------------------
// Keys are (sortable sim, id); order = sim descending, then id ascending.  The first round holds up to
// 64*R keys in registers (R per lane; R = 2 ... 16 chosen per query from its candidate count); longer rows
// stream the rest against the running k-th best value (select_rounds).  The k-th largest sim is found by a bitwise
// binary search whose counts are wave ballots (v_cmp + s_bcnt1, no LDS) and which stops as soon as
// a threshold splits off exactly k keys; boundary ties are resolved by a second search over ids.
// Survivors are compacted into LDS by ballot-prefix ranks; the final <= k keys are sorted by an
// in-register bitonic network over the wave (shuffles, no LDS).
__device__ __forceinline__ int wave_count(bool p) { return __popcll(__ballot(p)); }

struct SelQuery {
    const float* row;     // this query's sims
    int64_t nc;           // number of candidates
    int64_t id0;          // MODE_DENSE: id = id0 + position
};

// The first round of the selection: up to R*64 keys from the query's sims row (slot s = i*64 + lane).  Keep the k
// best: threshold by bitwise search with ballot counts (early exit when a threshold isolates exactly k keys),
// ties at the threshold by id, survivors compacted into LDS.  Returns how many were kept.
//  * loads are unconditional and unclamped (the sims buffer has kSimsSlack floats of slack), so the R
//    loads of a round are in flight together with immediate offsets;
//  * MODE_DENSE ids are implicit (id0 + stream position): no id registers, none written until compaction.
template <int MODE, int R>
__device__ __forceinline__ int select_round(const SelectArgs& a, const SelQuery& qy, int k, int lane, int fresh,
                                            uint32_t* sel_u, uint32_t* sel_id, const int64_t* seg_off,
                                            const int64_t* seg_src) {
    uint32_t u[R];
    const float* rl = qy.row + lane;
    float fv[R];
#pragma unroll
    for (int i = 0; i < R; ++i) fv[i] = rl[i * 64];
    // ids.  MODE_DENSE: implicit (id0 + stream position).  MODE_IVF: the id of stream position pp is
    // perm[list-order position of pp] -- a segment search plus a gather -- so it is resolved LAZILY: only for
    // the k survivors after the rounds (select_rounds), and here only in the rare tie-at-the-threshold path.
    const uint32_t id_lane = (uint32_t)(qy.id0 + lane);
    auto real_id = [&](int64_t pp) -> uint32_t {
        pp = min<int64_t>(pp, qy.nc - 1);
        int lo = 0, hi = a.n_probe - 1;              // last segment with seg_off <= pp
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_off[mid] <= pp) lo = mid; else hi = mid - 1;
        }
        return (uint32_t)a.perm[seg_src[lo] + (pp - seg_off[lo])];
    };
    auto id_of = [&](int i) -> uint32_t {
        return MODE == MODE_DENSE ? id_lane + (uint32_t)(i * 64) : real_id(i * 64 + lane);
    };
#pragma unroll
    for (int i = 0; i < R; ++i) u[i] = (i * 64 + lane < fresh) ? max(f32_sortable(fv[i]), 1u) : 0u;
    const int m = fresh;

    uint32_t T = 1, I = 0xFFFFFFFFu;
    if (m > k) {
        // largest T with count(key >= T) >= k
        T = 0;
        bool exact = false;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t c = T | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) cnt += wave_count(u[i] >= c);
            if (cnt >= k) T = c;
            if (cnt == k) {
                exact = true;
                break;
            }
        }
        if (!exact) {
            int gt = 0, eq = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                gt += wave_count(u[i] > T);
                eq += wave_count(u[i] == T);
            }
            const int need = k - gt;
            if (eq > need) {
                uint32_t lo = 0;      // largest value with count(key == T && id < lo) < need
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t c = lo | (1u << bit);
                    int cnt = 0;
#pragma unroll
                    for (int i = 0; i < R; ++i) cnt += wave_count(u[i] == T && id_of(i) < c);
                    if (cnt < need) lo = c;
                }
                I = lo;
            }
        }
    }
    // ---- compact survivors into LDS ------------------------------------------------------
    int base = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        // (I is wave-uniform; all-ones = no tie-break in force: ids need not be resolved)
        const bool keep = u[i] != 0 && ((u[i] > T) || (u[i] == T && (I == 0xFFFFFFFFu || id_of(i) <= I)));
        const uint64_t mask = __ballot(keep);
        if (keep) {
            const int w = base + __popcll(mask & ((1ull << lane) - 1ull));
            sel_u[w] = u[i];
            // MODE_IVF: the stream position, flagged; select_rounds turns the survivors' positions into ids
            sel_id[w] = MODE == MODE_DENSE ? id_of(i) : (0x80000000u | (uint32_t)(i * 64 + lane));
        }
        base += __popcll(mask);
    }
    return base;
}

// MODE_IVF: survivors kept as flagged stream positions -> real ids (k / 64 gathers per lane, all in flight together)
template <int E = FAL_MAX_K_ANN / 64>
__device__ __forceinline__ void resolve_ids(const SelectArgs& a, const SelQuery& qy, int carry, int lane, uint32_t* sel_id,
                                            const int64_t* seg_off, const int64_t* seg_src) {
    uint32_t v[E];
    int64_t at[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        v[j] = 0u;
        at[j] = 0;
        if (j * 64 < carry) {                            // wave-uniform: registers beyond the set cost nothing
            const int e = j * 64 + lane;
            v[j] = e < carry ? sel_id[e] : 0u;
            const int64_t pp = min<int64_t>((int64_t)(v[j] & 0x7FFFFFFFu), qy.nc - 1);
            int lo = 0, hi = a.n_probe - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (seg_off[mid] <= pp) lo = mid; else hi = mid - 1;
            }
            at[j] = seg_src[lo] + (pp - seg_off[lo]);
        }
    }
    uint32_t g[E];
#pragma unroll
    for (int j = 0; j < E; ++j) g[j] = (j * 64 < carry) ? (uint32_t)a.perm[at[j]] : 0u;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int e = j * 64 + lane;
        if (e < carry && (v[j] & 0x80000000u)) sel_id[e] = g[j];
    }
}

// entries the selected set may hold in LDS while a long row streams by (k survivors + one 256-key chunk); kept small:
// the kernel is latency-bound and LDS is what limits the waves per CU
constexpr int kSelBuf = FAL_MAX_K_ANN + 256;

// cut the set in LDS (cnt > k entries with explicit ids) back to its k best; *T = the k-th best key
template <int EC>
__device__ __forceinline__ int reselect(uint32_t* sel_u, uint32_t* sel_id, int cnt, int k, int lane, uint32_t* T_out) {
    uint32_t u[EC], id[EC];
#pragma unroll
    for (int j = 0; j < EC; ++j) {
        const int e = j * 64 + lane;
        u[j] = e < cnt ? sel_u[e] : 0u;
        id[j] = sel_id[e];
    }
    __syncthreads();                 // all reads done before the set is rewritten
    uint32_t T = 0, I = 0xFFFFFFFFu;
    bool exact = false;
    for (int bit = 31; bit >= 0; --bit) {                  // largest T with count(key >= T) >= k
        const uint32_t c = T | (1u << bit);
        int n = 0;
#pragma unroll
        for (int j = 0; j < EC; ++j) n += wave_count(u[j] >= c);
        if (n >= k) T = c;
        if (n == k) {
            exact = true;
            break;
        }
    }
    if (!exact) {
        int gt = 0, eq = 0;
#pragma unroll
        for (int j = 0; j < EC; ++j) {
            gt += wave_count(u[j] > T);
            eq += wave_count(u[j] == T);
        }
        const int need = k - gt;
        if (eq > need) {
            uint32_t lo = 0;          // largest value with count(key == T && id < lo) < need
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t c = lo | (1u << bit);
                int n = 0;
#pragma unroll
                for (int j = 0; j < EC; ++j) n += wave_count(u[j] == T && id[j] < c);
                if (n < need) lo = c;
            }
            I = lo;
        }
    }
    int base = 0;
#pragma unroll
    for (int j = 0; j < EC; ++j) {
        const bool keep = u[j] != 0 && ((u[j] > T) || (u[j] == T && id[j] <= I));
        const uint64_t mask = __ballot(keep);
        if (keep) {
            const int w = base + __popcll(mask & ((1ull << lane) - 1ull));
            sel_u[w] = u[j];
            sel_id[w] = id[j];
        }
        base += __popcll(mask);
    }
    __syncthreads();
    *T_out = T;
    return base;
}

template <int MODE, int R>
__device__ __forceinline__ int select_rounds(const SelectArgs& a, const SelQuery& qy, int k, int lane,
                                             uint32_t* sel_u, uint32_t* sel_id, const int64_t* seg_off,
                                             const int64_t* seg_src) {
    int fresh = (int)min<int64_t>(qy.nc, 64 * R);
    int carry = select_round<MODE, R>(a, qy, k, lane, fresh, sel_u, sel_id, seg_off, seg_src);
    __syncthreads();
    if (MODE != MODE_DENSE) {
        resolve_ids(a, qy, carry, lane, sel_id, seg_off, seg_src);
        __syncthreads();
    }
    if constexpr (R == 16) {
        // More than 1,024 candidates: stream the rest.  After the first round the k-th best value T is known; a later
        // key can only matter if it beats T, and with candidates in no particular order ever fewer do (~k ln(nc/1024)
        // in total).  So a chunk costs its loads, one compare per key and a ballot per register; only survivors are
        // appended to the set in LDS, and the set is cut back to k (tightening T) when it outgrows its buffer.
        constexpr int RS = 4;                              // keys per lane per chunk (256 per chunk)
        if (qy.nc > 64 * R) {
            uint32_t T = 0xFFFFFFFFu;                      // the smallest kept key = k-th best so far (carry == k here)
            for (int e = lane; e < carry; e += 64) T = min(T, sel_u[e]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) T = min(T, (uint32_t)__shfl_xor((int)T, off, 64));
            int cnt = carry;
            for (int64_t pos = fresh; pos < qy.nc; pos += 64 * RS) {
                const int nf = (int)min<int64_t>(qy.nc - pos, 64 * RS);
                const float* rl = qy.row + pos + lane;
                float fv[RS];
#pragma unroll
                for (int i = 0; i < RS; ++i) fv[i] = rl[i * 64];
#pragma unroll
                for (int i = 0; i < RS; ++i) {
                    const uint32_t u = (i * 64 + lane < nf) ? max(f32_sortable(fv[i]), 1u) : 0u;
                    // MODE_DENSE: ids grow with the position, an equal key further on loses the tie.  MODE_IVF: ids are
                    // arbitrary, equal keys stay in the race until ids are resolved.
                    const bool in = MODE == MODE_DENSE ? u > T : u >= T;
                    const uint64_t mask = __ballot(in);
                    if (mask) {                            // wave-uniform
                        if (in) {
                            const int wpos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                            sel_u[wpos] = u;
                            sel_id[wpos] = MODE == MODE_DENSE ? (uint32_t)(qy.id0 + pos + i * 64 + lane)
                                                              : (0x80000000u | (uint32_t)(pos + i * 64 + lane));
                        }
                        cnt += __popcll(mask);
                    }
                }
                __syncthreads();
                if (cnt > kSelBuf - 64 * RS || pos + 64 * RS >= qy.nc) {      // no room for another chunk, or the end
                    if (MODE != MODE_DENSE) {
                        resolve_ids<kSelBuf / 64>(a, qy, cnt, lane, sel_id, seg_off, seg_src);
                        __syncthreads();
                    }
                    if (cnt > k) cnt = reselect<kSelBuf / 64>(sel_u, sel_id, cnt, k, lane, &T);
                }
            }
            carry = cnt;
        }
    }
    return carry;
}

// lane ^ X exchange without LDS traffic where the hardware allows: DPP quad_perm for X = 1, 2;
// ds_swizzle (bit-mask mode, no address VGPR, no memory) for X = 4, 8, 16; ds_bpermute for 32.
template <int X>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if (X == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    if (X == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    if (X == 4 || X == 8 || X == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (X << 10));
    return (uint32_t)__shfl_xor((int)v, X, 64);
}

// one compare-exchange level of the bitonic network at distance STRIDE inside blocks of SIZE.
// Element index = lane * E + reg, so strides below E stay inside a lane's registers.
template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_level(uint32_t (&hi)[E], uint32_t (&lo)[E], int lane) {
    if constexpr (STRIDE < E) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            if ((r & STRIDE) == 0) {
                const bool desc = ((lane * E + r) & SIZE) == 0;
                const bool a_lt_b = hi[r] < hi[r | STRIDE] || (hi[r] == hi[r | STRIDE] && lo[r] < lo[r | STRIDE]);
                if (a_lt_b == desc) {
                    const uint32_t th = hi[r], tl = lo[r];
                    hi[r] = hi[r | STRIDE];
                    lo[r] = lo[r | STRIDE];
                    hi[r | STRIDE] = th;
                    lo[r | STRIDE] = tl;
                }
            }
        }
    } else {
        constexpr int X = STRIDE / E;
        const bool lower = (lane & X) == 0;          // this lane holds the lower index of the pair
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t oh = lane_xor<X>(hi[r]), ol = lane_xor<X>(lo[r]);
            const bool desc = ((lane * E + r) & SIZE) == 0;
            const bool mine_lt = hi[r] < oh || (hi[r] == oh && lo[r] < ol);
            const bool mine_gt = hi[r] > oh || (hi[r] == oh && lo[r] > ol);
            // descending: the lower index keeps the larger key
            if ((lower == desc) ? mine_lt : mine_gt) {
                hi[r] = oh;
                lo[r] = ol;
            }
        }
    }
}

template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge(uint32_t (&hi)[E], uint32_t (&lo)[E], int lane) {
    bitonic_level<E, SIZE, STRIDE>(hi, lo, lane);
    if constexpr (STRIDE > 1) bitonic_merge<E, SIZE, STRIDE / 2>(hi, lo, lane);
}

template <int E, int SIZE>
__device__ __forceinline__ void bitonic_build(uint32_t (&hi)[E], uint32_t (&lo)[E], int lane) {
    if constexpr (SIZE > 2) bitonic_build<E, SIZE / 2>(hi, lo, lane);
    bitonic_merge<E, SIZE, SIZE / 2>(hi, lo, lane);
}

// descending sort of 64*E keys (hi = sortable sim, lo = ~id), E per lane at index lane*E + reg;
// empty slots are (0, 0) and sink to the end
template <int E>
__device__ __forceinline__ void sort_and_store(const uint32_t* sel_u, const uint32_t* sel_id, int carry, int k, int lane,
                                               float* __restrict__ osim, int32_t* __restrict__ oidx) {
    uint32_t hi[E], lo[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = lane * E + r;
        hi[r] = e < carry ? sel_u[e] : 0u;
        lo[r] = e < carry ? ~sel_id[e] : 0u;
    }
    bitonic_build<E, 64 * E>(hi, lo, lane);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = lane * E + r;
        if (e < k) {
            const bool valid = e < carry;
            osim[e] = valid ? sortable_f32(hi[r]) : -INFINITY;
            oidx[e] = valid ? (int32_t)~lo[r] : -1;
        }
    }
}

// a8 on the selected set (graph.hip filter_kernel is the staged form): drop self and neighbours outside the
// precursor / RT tolerance, sort the SURVIVORS by (similarity desc, id asc) with the smallest network that
// holds them, keep the first f_keep, dist = clip(1 - sim, 0, 1).  Typically a handful of the k_ann
// candidates survive, so this replaces a 128-key sort + a second kernel by a 16/32-key sort.
template <int SIZE>
__device__ __forceinline__ void sort_small(uint32_t& hi, uint32_t& lo, int lane) {
    uint32_t h[1] = {hi}, l[1] = {lo};
    bitonic_build<1, SIZE>(h, l, lane);
    hi = h[0];
    lo = l[0];
}

template <int E>
__device__ __forceinline__ void sort_and_store_nb(const uint32_t* f_u, const uint32_t* f_lo, int c, int keep, int lane,
                                                  int32_t* __restrict__ onb, float* __restrict__ odist) {
    uint32_t hi[E], lo[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = lane * E + r;
        hi[r] = e < c ? f_u[e] : 0u;
        lo[r] = e < c ? f_lo[e] : 0u;
    }
    if constexpr (E == 1) {
        if (c <= 2) sort_small<2>(hi[0], lo[0], lane);
        else if (c <= 4) sort_small<4>(hi[0], lo[0], lane);
        else if (c <= 8) sort_small<8>(hi[0], lo[0], lane);
        else if (c <= 16) sort_small<16>(hi[0], lo[0], lane);
        else if (c <= 32) sort_small<32>(hi[0], lo[0], lane);
        else sort_small<64>(hi[0], lo[0], lane);
    } else {
        bitonic_build<E, 64 * E>(hi, lo, lane);
    }
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int e = lane * E + r;
        if (e < keep) {
            const bool valid = e < c;
            onb[e] = valid ? (int32_t)~lo[r] : -1;
            odist[e] = valid ? fminf(fmaxf(1.0f - sortable_f32(hi[r]), 0.f), 1.f) : INFINITY;
        }
    }
    for (int e = 64 * E + lane; e < keep; e += 64) {
        onb[e] = -1;
        odist[e] = INFINITY;
    }
}

__device__ __forceinline__ void filter_sort_store(const SelectArgs& a, const uint32_t* sel_u, const uint32_t* sel_id,
                                                  uint32_t* f_u, uint32_t* f_lo, int carry, int64_t row, int lane) {
    const float qmz = a.f_pmz[row];
    const bool use_rt = a.f_rt != nullptr && a.f_rt_tol >= 0.0;
    const float qrt = use_rt ? a.f_rt[row] : 0.f;
    int c = 0;
    for (int e0 = 0; e0 < carry; e0 += 64) {
        const int e = e0 + lane;
        bool ok = false;
        uint32_t u = 0, id = 0;
        if (e < carry) {
            u = sel_u[e];
            id = sel_id[e];
            if ((int64_t)id != row) {
                const float nmz = a.f_pmz[id];
                const float diff = qmz - nmz;     // mass_diff(query, neighbour), the arithmetic of filter_kernel
                const double md = a.f_is_da ? (double)diff : (double)(diff / nmz) * 1e6;
                ok = fabs(md) <= a.f_tol;
                if (ok && use_rt) ok = fabs((double)(qrt - a.f_rt[id])) <= a.f_rt_tol;
            }
        }
        const uint64_t mask = __ballot(ok);
        if (ok) {
            const int w = c + __popcll(mask & ((1ull << lane) - 1ull));
            f_u[w] = u;
            f_lo[w] = ~id;
        }
        c += __popcll(mask);
    }
    __syncthreads();
    int32_t* onb = a.nb_idx + row * a.f_keep;
    float* odist = a.nb_dist + row * a.f_keep;
    if (a.nb_count && lane == 0) a.nb_count[row] = min(c, a.f_keep);
    if (c <= 64) sort_and_store_nb<1>(f_u, f_lo, c, a.f_keep, lane, onb, odist);
    else if (c <= 128) sort_and_store_nb<2>(f_u, f_lo, c, a.f_keep, lane, onb, odist);
    else sort_and_store_nb<4>(f_u, f_lo, c, a.f_keep, lane, onb, odist);
}

template <int MODE, bool FUSE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == MODE_DENSE ? 8 : 3, 8))) void select_kernel(SelectArgs a) {
    __shared__ uint32_t sel_u[kSelBuf];       // the selected set (+ room for one streamed chunk of a long row)
    __shared__ uint32_t sel_id[kSelBuf];
    __shared__ int64_t seg_off[FAL_MAX_N_PROBE + 1];   // MODE_IVF: stream offset of each probed list
    __shared__ int64_t seg_src[FAL_MAX_N_PROBE];       // MODE_IVF: perm position of each probed list
    const int lane = threadIdx.x;
    const int k = a.k;

    // ---- locate this query ---------------------------------------------------------------
    SelQuery qy{nullptr, 0, 0};
    int64_t out_row = 0;
    const int64_t t = a.tile_begin + (blockIdx.x >> 5);
    const int ql = blockIdx.x & 31;
    const DenseJob job = a.jobs[a.tile_job[blockIdx.x >> 5]];
    const int lt = (int)(t - job.tile0);
    if (32 * lt + ql >= job.nq) return;
    if (MODE == MODE_DENSE) {
        qy.nc = job.nc;
        qy.row = a.sims + (job.obase - a.sims_base) + (int64_t)(32 * lt + ql) * ((job.nc + 31) & ~31);
        out_row = job.q_row0 + 32 * (int64_t)lt + ql;
        qy.id0 = a.ids_are_rows ? job.c_row0 : 0;
    } else {
        const int64_t p = job.q_row0 + 32 * (int64_t)lt + ql;   // query position in list order
        const int np = a.n_probe;
        const int32_t* pr = a.probes + p * np;
        const int64_t lbase = job.c_row0;              // global id of the bucket's list 0
        if (lane == 0) {
            int64_t off = 0;
            for (int j = 0; j < np; ++j) {
                const int32_t l = pr[j];
                seg_off[j] = off;
                if (l >= 0) {
                    const int64_t b = a.list_off[lbase + l], e = a.list_off[lbase + l + 1];
                    seg_src[j] = b;
                    off += e - b;
                } else {
                    seg_src[j] = 0;
                }
            }
            seg_off[np] = off;
        }
        __syncthreads();
        qy.nc = seg_off[np];
        qy.row = a.sims + (a.q_sim_off[32 * t + ql] - a.sims_base);   // tile-order slot
        out_row = a.perm[p];
    }

    int carry;
    if (qy.nc <= 128) carry = select_rounds<MODE, 2>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 256) carry = select_rounds<MODE, 4>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 384) carry = select_rounds<MODE, 6>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 512) carry = select_rounds<MODE, 8>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else if (qy.nc <= 768) carry = select_rounds<MODE, 12>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);
    else carry = select_rounds<MODE, 16>(a, qy, k, lane, sel_u, sel_id, seg_off, seg_src);

    if constexpr (FUSE) {
        // the survivors of the filter go to the tail of the set's buffers (free once the selection is done)
        filter_sort_store(a, sel_u, sel_id, sel_u + FAL_MAX_K_ANN, sel_id + FAL_MAX_K_ANN, carry, out_row, lane);
    } else {
        float* osim = a.out_sim + out_row * k;
        int32_t* oidx = a.out_idx + out_row * k;
        if (k <= 64) sort_and_store<1>(sel_u, sel_id, carry, k, lane, osim, oidx);
        else if (k <= 128) sort_and_store<2>(sel_u, sel_id, carry, k, lane, osim, oidx);
        else sort_and_store<4>(sel_u, sel_id, carry, k, lane, osim, oidx);
    }
}

__global__ void tile_job_kernel(const DenseJob* __restrict__ jobs, int n_jobs, int64_t tile_begin, int64_t n_tiles,
                                int32_t* __restrict__ tile_job) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_tiles) tile_job[i] = find_job(jobs, n_jobs, tile_begin + i);
}

int launch_select(fal_ctx* ctx, int stage, int mode, const SelectArgs& a_in, int64_t n_blocks, hipStream_t on) {
    if (n_blocks <= 0) return FAL_OK;
    SelectArgs a = a_in;
    {
        // tile -> job once per tile instead of a binary search in every query's workgroup
        const int64_t n_tiles = n_blocks / 32;
        int32_t* tj = nullptr;
        FAL_TRY(ctx->reserve(SLOT_TILEJOB, sizeof(int32_t) * (size_t)std::max<int64_t>(n_tiles, 1 << 16), (void**)&tj));
        hipLaunchKernelGGL(tile_job_kernel, dim3((unsigned)ceil_div(n_tiles, 256)), dim3(256), 0, on ? on : ctx->stream, a.jobs,
                           a.n_jobs, a.tile_begin, n_tiles, tj);
        a.tile_job = tj;
    }
    FAL_REQUIRE(a.k >= 1 && a.k <= FAL_MAX_K_ANN, FAL_EUNSUPPORTED, "k must be in [1, %d]", FAL_MAX_K_ANN);
    FAL_REQUIRE(n_blocks < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many queries in one select launch");
    hipStream_t st = on ? on : ctx->stream;
    StageScope ts(ctx, stage, st);
    const bool fuse = a.nb_idx != nullptr;
    if (fuse) FAL_REQUIRE(a.f_pmz && a.nb_dist && a.f_keep >= 1 && a.f_keep <= FAL_MAX_K_ANN, FAL_EINVAL, "fused filter: bad arguments");
    if (mode == MODE_DENSE && fuse)
        hipLaunchKernelGGL((select_kernel<MODE_DENSE, true>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    else if (mode == MODE_DENSE)
        hipLaunchKernelGGL((select_kernel<MODE_DENSE, false>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    else if (fuse)
        hipLaunchKernelGGL((select_kernel<MODE_IVF, true>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((select_kernel<MODE_IVF, false>), dim3((unsigned)n_blocks), dim3(64), 0, st, a);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
