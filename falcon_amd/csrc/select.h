// Device pieces of the wavefront top-k select and of the fused a8 filter (shared by scan.hip's select_kernel and by
// fused.hip's prefilter kernel and its exact fallback).
#pragma once
#include <math.h>
#include "common.h"
#include "simtile.h"
#include "scan.h"

namespace fal {

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
        // T = the k-th largest key, by radix levels over LDS histograms (sel_u is free until the compaction below): the
        // bits all keys share are skipped, then 8 bits per level -- R LDS atomics and one suffix sum over the lanes per
        // level instead of R ballot counts per BIT.  A level whose bin holds exactly the keys still wanted ends the
        // search: T = the prefix so far is then a threshold with exactly k keys at or above it.
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            mn = min(mn, u[i] ? u[i] : 0xFFFFFFFFu);
            mx = max(mx, u[i]);
        }
        mn = wave_min_u32(mn);                                  // (DPP network: no LDS round trips)
        mx = wave_max_u32(mx);
        int shift = 32 - __clz((int)(mn ^ mx));                 // low bits in which the keys differ (0: all keys equal)
        if (mn == mx) shift = 0;
        uint32_t prefix = shift >= 32 ? 0u : (mx >> shift) << shift;
        int kk = k, eq = m;                                     // keys still wanted at or below the prefix; keys equal to T
        bool exact = false;
        uint32_t* hist = sel_u;
        while (shift > 0 && !exact) {
            const int wbits = min(8, shift);
            const int hi = shift;                               // bits [hi, 32) are fixed by the prefix
            shift -= wbits;
            const uint32_t dmask = (1u << wbits) - 1u;
            const uint32_t himask = hi >= 32 ? 0u : ~0u << hi;
            *reinterpret_cast<uint4*>(hist + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (u[i] != 0u && (u[i] & himask) == prefix)
                    __hip_atomic_fetch_add(&hist[(u[i] >> shift) & dmask], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            wave_lds_sync();
            const uint4 cv = *reinterpret_cast<const uint4*>(hist + 4 * lane);      // bins 4 lane .. 4 lane + 3
            const int c[4] = {(int)cv.x, (int)cv.y, (int)cv.z, (int)cv.w};
            const int own = c[0] + c[1] + c[2] + c[3];
            const int pre = wave_prefix_sum(own);
            const int suf = __builtin_amdgcn_readlane(pre, 63) - pre + own;        // keys in this lane's bins and in all higher ones
            const int L = 63 - __clzll((unsigned long long)__ballot(suf >= kk));    // the lane whose bins hold the kk-th largest
            int acc = suf - own, bin = 4 * lane, cj = c[0];
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (acc + c[j] >= kk) { bin = 4 * lane + j; cj = c[j]; break; }
                acc += c[j];
            }
            acc = __builtin_amdgcn_readlane(acc, L);            // (L is wave-uniform)
            bin = __builtin_amdgcn_readlane(bin, L);
            cj = __builtin_amdgcn_readlane(cj, L);
            kk -= acc;
            prefix |= (uint32_t)bin << shift;
            exact = cj == kk;
            eq = cj;
            wave_lds_sync();
        }
        T = prefix;
        if (!exact) {
            const int need = kk;                                // keys equal to T that are still wanted (ties: lowest ids)
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
            static_assert(RS == 4, "the 16-byte chunk load holds four keys per lane");
            const bool wide = ((reinterpret_cast<uintptr_t>(qy.row) + 4 * (uintptr_t)fresh) & 15) == 0;      // (wave-uniform)
            for (int64_t pos = fresh; pos < qy.nc; pos += 64 * RS) {
                const int nf = (int)min<int64_t>(qy.nc - pos, 64 * RS);
                // a 16-byte load per lane where the row allows it (one load instruction per chunk instead of four: the long rows
                // of the exhaustive float16 configuration spend their time here); key i of a lane is then chunk position
                // 4 lane + i instead of 64 i + lane
                float fv[RS];
                if (wide) {
                    const float4 v4 = *reinterpret_cast<const float4*>(qy.row + pos + 4 * lane);
                    fv[0] = v4.x; fv[1] = v4.y; fv[2] = v4.z; fv[3] = v4.w;
                } else {
                    const float* rl = qy.row + pos + lane;
#pragma unroll
                    for (int i = 0; i < RS; ++i) fv[i] = rl[i * 64];
                }
#pragma unroll
                for (int i = 0; i < RS; ++i) {
                    const int cp = wide ? 4 * lane + i : i * 64 + lane;      // position inside the chunk
                    const uint32_t u = (cp < nf) ? max(f32_sortable(fv[i]), 1u) : 0u;
                    // MODE_DENSE: ids grow with the position, an equal key further on loses the tie.  MODE_IVF: ids are
                    // arbitrary, equal keys stay in the race until ids are resolved.
                    const bool in = MODE == MODE_DENSE ? u > T : u >= T;
                    const uint64_t mask = __ballot(in);
                    if (mask) {                            // wave-uniform
                        if (in) {
                            const int wpos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                            sel_u[wpos] = u;
                            sel_id[wpos] = MODE == MODE_DENSE ? (uint32_t)(qy.id0 + pos + cp)
                                                              : (0x80000000u | (uint32_t)(pos + cp));
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

// The same for <= 64 survivors held as 64-bit keys (x = ~id, y = sortable similarity) in LDS.  Up to 16 of them (the usual case:
// ~10 per query) are RANKED instead of sorted -- every lane counts the keys above its own with broadcast reads, 2 VALU + 1 LDS
// instruction per key against ~9 per stage of the 10-stage network: resolve_kernel is VALU-bound (0.77 of the SIMDs' VALU
// cycles, round 6 counters) and the sort was a third of its instructions.  Keys are unique (ids are), so ranks are the sorted
// positions: identical output.
__device__ __forceinline__ void rank_or_sort_and_store_nb(const uint2* keys, int c, int keep, int lane, int32_t* __restrict__ onb,
                                                          float* __restrict__ odist) {
    const uint2 mine = lane < c ? keys[lane] : make_uint2(0u, 0u);
    if (c <= 16) {
        const unsigned long long me = ((unsigned long long)mine.y << 32) | mine.x;
        int rank = 0;
        for (int j = 0; j < c; ++j) {
            const uint2 kj = keys[j];
            rank += ((((unsigned long long)kj.y << 32) | kj.x) > me) ? 1 : 0;
        }
        if (lane < c && rank < keep) {
            onb[rank] = (int32_t)~mine.x;
            odist[rank] = fminf(fmaxf(1.0f - sortable_f32(mine.y), 0.f), 1.f);
        }
        for (int e = c + lane; e < keep; e += 64) {
            onb[e] = -1;
            odist[e] = INFINITY;
        }
        return;
    }
    uint32_t hi = mine.y, lo = mine.x;
    if (c <= 32) sort_small<32>(hi, lo, lane);
    else sort_small<64>(hi, lo, lane);
    if (lane < keep) {
        const bool valid = lane < c;
        onb[lane] = valid ? (int32_t)~lo : -1;
        odist[lane] = valid ? fminf(fmaxf(1.0f - sortable_f32(hi), 0.f), 1.f) : INFINITY;
    }
    for (int e = 64 + lane; e < keep; e += 64) {
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


}  // namespace fal
