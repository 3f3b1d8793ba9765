// a7 coarse quantiser from the keys the index build left behind.
//
// The final k-means pass (assign16.hip) already computes every (row, centroid) similarity of a bucket on the f16 matrix cores
// -- against the FINAL centroids, which is exactly what the coarse quantiser of the search needs.  It leaves them as 16-bit
// keys round(v * 65535) ([n, 128] by sorted row).  Here: the n_probe best centroids of every query from those keys, exact where
// it matters.  With T~ the n_probe-th largest key and |key / 65535 - exact| <= e(v) = 1.3e-3 v + 1.2e-5 (ivf16.hip), a key above
// T~ + 2e is certainly among the n_probe best, one below T~ - 2e certainly not; the keys in between ("members") decide the
// rest.  If there are exactly as many members as open places they all are in; otherwise (two centroids about equally far: a
// few per cent of the queries) the members are re-evaluated by the exact k-ordered fmaf chain and ranked by (similarity
// descending, list id ascending) -- the order of the staged coarse scan + select.  The probe SET is identical to the staged
// path's; its order inside a query is arbitrary (nothing depends on it: every consumer reads the same table).
//
// 16 lanes per query (8 keys each), 4 queries per wave: the counts of the bitwise threshold search are 16-bit slices of one
// ballot.  Replaces a [n, n_list] fp32-MFMA scan + a wavefront select (13 + 6 ms at 10 M spectra).
//
// Reference: README.md:107-113 (n_probe lists per query); faiss IndexIVFFlat's quantizer->search is a dependency of the
// reference, not in the snapshot.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "coarse16.h"

namespace fal {

// KPL keys per lane (8: buckets with <= 128 lists, 32: <= 512), 16 lanes per query, QPW queries per workgroup
// (the workgroup-level form: every query keeps room for ALL its keys as members.  It serves the queries coarse16w_kernel
// hands over -- more than 16 members: many equal similarities.)
template <int KPL, int QPW>
__global__ __launch_bounds__(16 * QPW) void coarse16_kernel(Coarse16Args a) {
    constexpr int kMem = 16 * KPL;                         // members a query can hold (every key, in the worst case)
    constexpr int kThreads = 16 * QPW;
    __shared__ float m_val[QPW][kMem];
    __shared__ int32_t m_id[QPW][kMem];
    __shared__ int32_t q_cnt[QPW];
    __shared__ int64_t q_row[QPW], q_cbase[QPW];
    const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 4, sub = lane & 15, qw = tid >> 4;
    const int64_t n_items = (int64_t)min(*a.ovf_count, a.ovf_cap);
  for (int64_t item0 = (int64_t)blockIdx.x * QPW; item0 < n_items; item0 += (int64_t)gridDim.x * QPW) {
    const int64_t item = item0 + qw;
    const int64_t g = item < n_items ? (int64_t)a.ovf_list[item] : (int64_t)-1;   // tile-order slot of the group's query
    const int64_t t = g >> 5;
    const int ql = (int)(g & 31);
    bool live = g >= 0 && t < a.n_tiles;
    DenseJob job{};
    if (live) job = a.jobs[a.tile_job[t]];
    const int lt = (int)(t - job.tile0);
    live = live && 32 * lt + ql < job.nq;
    const int64_t p = live ? job.q_row0 + 32 * (int64_t)lt + ql : 0;
    const int64_t row = live ? a.perm[p] : 0;
    const int nl = live ? job.nc : 0;
    const int np = a.np;
    // key j of 16-byte piece (16 q + sub) of the row: list id (16 q + sub) * 8 + j -- a group reads 256 contiguous bytes per q
    uint32_t u[KPL];
    auto id_of = [&](int x) -> int { return (16 * (x >> 3) + sub) * 8 + (x & 7); };
#pragma unroll
    for (int q = 0; q < KPL / 8; ++q) {
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (live && (16 * q + sub) * 8 < nl) raw = *reinterpret_cast<const uint4*>(a.ckeys + row * (int64_t)a.stride + (16 * q + sub) * 8);
        const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) u[8 * q + j] = (id_of(8 * q + j) < nl) ? ((wv[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) + 1u : 0u;
    }
    const int want = min(np, nl);                                     // probes of this query
    const int sh = 16 * grp;
    auto gcount = [&](bool pred) -> int { return __popc((uint32_t)(__ballot(pred) >> sh) & 0xFFFFu); };
    // T = the want-th largest key of the group (largest T with count(u >= T) >= want)
    uint32_t T = 0;
    for (int bit = 16; bit >= 0; --bit) {
        const uint32_t c = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) cnt += gcount(u[j] >= c);
        if (cnt >= want && want > 0) T = c;
    }
    const float Tv = (float)(max(T, 1u) - 1u) * (1.f / 65535.f);
    const float e = 1.3e-3f * Tv + 1.2e-5f;
    const int delta = 2 * ((int)ceilf(e * 65535.f) + 1) + 2;
    int n_hi = 0, n_mem = 0;
    uint32_t hi = 0, mem = 0;                                         // bit j: key j of this lane
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int df = (int)u[j] - (int)T;
        const bool h1 = u[j] != 0u && df > delta;
        const bool m1 = u[j] != 0u && df <= delta && df >= -delta;
        hi |= h1 ? (1u << j) : 0u;
        mem |= m1 ? (1u << j) : 0u;
        n_hi += gcount(h1);
        n_mem += gcount(m1);
    }
    const int need = want - n_hi;                                     // places left for the members (1 .. n_mem when want > 0)
    const bool ambiguous = want > 0 && n_mem > need;
    // Exact values for the members of the ambiguous queries.  A chain is serial (400 dependent fmaf), so the members of all
    // queries of the workgroup are pooled: item i of the pool goes to thread i -- normally they all fit the first wave, and
    // the others do not run a chain at all (one per wave for two or three busy lanes was most of this kernel's time).
    if (sub == 0) {
        q_cnt[qw] = ambiguous ? n_mem : 0;
        q_row[qw] = row;
        q_cbase[qw] = job.c_row0;
    }
    {
        int base = 0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const bool m1 = ((mem >> j) & 1u) && ambiguous;
            const uint32_t gm = (uint32_t)(__ballot(m1) >> sh) & 0xFFFFu;
            if (m1) m_id[qw][base + __popc(gm & ((1u << sub) - 1u))] = id_of(j);
            base += __popc(gm);
        }
    }
    __syncthreads();
    {
        int off[QPW + 1];
        off[0] = 0;
#pragma unroll
        for (int k = 0; k < QPW; ++k) off[k + 1] = off[k] + q_cnt[k];
        for (int i = tid; i < off[QPW]; i += kThreads) {
            int k = 0;
#pragma unroll
            for (int kk = 1; kk < QPW; ++kk) k = off[kk] <= i ? kk : k;
            const int m = i - off[k];
            m_val[k][m] = exact_dot(a.X + q_row[k] * a.d, a.C + (q_cbase[k] + m_id[k][m]) * (int64_t)a.d, a.d);
        }
    }
    __syncthreads();
    if (ambiguous) {
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            if (!((mem >> j) & 1u)) continue;
            const int me = id_of(j);
            float mine = 0.f;
            for (int i = 0; i < n_mem; ++i) mine = m_id[qw][i] == me ? m_val[qw][i] : mine;
            int rank = 0;
            for (int i = 0; i < n_mem; ++i) {
                const float v = m_val[qw][i];
                const int id = m_id[qw][i];
                rank += (v > mine || (v == mine && id < me)) ? 1 : 0;
            }
            if (rank >= need) mem &= ~(1u << j);
        }
    }
    // the probe table row: the chosen lists (any order: nothing depends on it), -1 behind them
    int32_t* out = a.probes + p * np;
    int base = 0;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const bool in = live && want > 0 && (((hi | mem) >> j) & 1u);
        const uint32_t gm = (uint32_t)(__ballot(in) >> sh) & 0xFFFFu;
        if (in) out[base + __popc(gm & ((1u << sub) - 1u))] = id_of(j);
        base += __popc(gm);
    }
    if (live)
        for (int i = want + sub; i < np; i += 16) out[i] = -1;
    __syncthreads();
  }
}

// The production form: ONE WAVE serves 16 queries.  Four rounds of four queries do the threshold search and the
// classification; the members of all ambiguous queries of the wave then share ONE round of exact chains (a chain is 400
// dependent fmaf behind memory latency, ~30 us: with a chain round per 4 queries -- or per 16 queries but with three of four
// waves of a workgroup waiting for it -- the kernel spent its time there), then the four rounds rank and emit.  A query with
// more than 16 members (many equal similarities) is handed to the workgroup-level kernel.
// The exact chain of a (query, centroid) pair over the query's sparse form: `cols` / `vals` = the query row's <= 64 entries in
// chain order (LDS; unused entries: column kColPad), `c` = the dense centroid.  A term whose query component is zero leaves the
// accumulator as it is (0 * c = +0, centroids of non-negative rows are non-negative), so the chain over the entries has the
// bits of the dense chain (pairs16.hip uses the same fact the other way round) -- at 64 independent 4-byte gathers in four
// batches instead of 2 x low_dim / 4 dependent 16-byte loads in low_dim / 32 batches: the round of chains is what the kernel
// waits for.  No early exit (a divergent trip count here is what hipcc miscompiled in round 3): padded entries multiply c[0] by 0.
__device__ __forceinline__ float coarse_sparse_chain(const uint16_t* cols, const float* vals, const float* __restrict__ c) {
    float acc = 0.f;
#pragma unroll
    for (int e0 = 0; e0 < kSparseW; e0 += 16) {
        uint32_t col[16];
        float qv[16], cv[16];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const uint2 c2 = *reinterpret_cast<const uint2*>(cols + e0 + 4 * t4);
            const float4 v4 = *reinterpret_cast<const float4*>(vals + e0 + 4 * t4);
            col[4 * t4] = c2.x & 0xFFFFu; col[4 * t4 + 1] = c2.x >> 16; col[4 * t4 + 2] = c2.y & 0xFFFFu; col[4 * t4 + 3] = c2.y >> 16;
            qv[4 * t4] = v4.x; qv[4 * t4 + 1] = v4.y; qv[4 * t4 + 2] = v4.z; qv[4 * t4 + 3] = v4.w;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const bool on = col[t] < (uint32_t)kColDense;
            cv[t] = c[on ? col[t] : 0u];
            qv[t] = on ? qv[t] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_fmaf(qv[t], cv[t], acc);
    }
    return acc;
}

template <int KPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KPL == 8 ? 5 : 3, 8))) void coarse16w_kernel(Coarse16Args a) {
    constexpr int kCap = 16;
    __shared__ float m_val[16][kCap];
    __shared__ int32_t m_id[16][kCap];
    __shared__ int32_t q_cnt[16];
    __shared__ int64_t q_row[16], q_cbase[16];
    __shared__ __attribute__((aligned(16))) uint16_t sp_c[8][kSparseW];       // the sparse rows of eight queries at a time
    __shared__ __attribute__((aligned(16))) float sp_v[8][kSparseW];
    const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
    const int sh = 16 * grp;
    const int np = a.np;
    auto gcount = [&](bool pred) -> int { return __popc((uint32_t)(__ballot(pred) >> sh) & 0xFFFFu); };
    auto id_of = [&](int x) -> int { return (16 * (x >> 3) + sub) * 8 + (x & 7); };
    // The rounds' results wait in LDS for the rank-and-emit phase (per lane: the masks of its keys above / within the bound; per
    // query: the counts).  Held in registers across the round of exact chains they were spilled: 49 dwords per lane at the
    // 96-register cap = 12.5 KB of scratch written (and read back) per wave -- 7.8 GB of the 9.1 GB this kernel wrote per 10 M
    // pass for 1.3 GB of probe lists (round 6, profiles/NOTES.md).
    __shared__ uint32_t s_hi[4][64], s_mem[4][64];
    __shared__ int32_t q_need[16], q_nmem[16], q_flag[16];                    // flag: 1 emit, 2 ambiguous
    // The wave's 16 queries are consecutive slots of ONE tile: one job lookup, one load of their sorted rows (lane = query), and --
    // with 8 keys per lane -- the keys of all four rounds in flight before the first is worked on (the first form paid the chain
    // tile -> job -> perm -> keys once per round).
    const int64_t g0 = (int64_t)blockIdx.x * 16;
    const int64_t t_w = g0 >> 5;
    const bool live_w = t_w < a.n_tiles;
    DenseJob job{};
    if (live_w) job = a.jobs[a.tile_job[t_w]];
    const int lt = (int)(t_w - job.tile0);
    const int ql0 = (int)(g0 & 31);
    const int n_live = live_w ? max(0, min(16, job.nq - (32 * lt + ql0))) : 0;      // queries 0 .. n_live - 1 of the wave exist
    const int64_t p0 = job.q_row0 + 32 * (int64_t)lt + ql0;
    const int row_lane = lane < n_live ? a.perm[p0 + lane] : 0;
    auto keys_of = [&](int rd, int q) -> uint4 {
        const int qi = 4 * rd + grp;
        const int64_t rw = (int64_t)(uint32_t)__shfl(row_lane, qi, 64);
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (qi < n_live && (16 * q + sub) * 8 < job.nc) raw = *reinterpret_cast<const uint4*>(a.ckeys + rw * (int64_t)a.stride + (16 * q + sub) * 8);
        return raw;
    };
    uint4 ahead[KPL == 8 ? 4 : 1];
    if constexpr (KPL == 8) {
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) ahead[rd] = keys_of(rd, 0);
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int qi = 4 * rd + grp;
        const int64_t g = g0 + qi;                                   // tile-order slot of this 16-lane group's query
        const bool live = qi < n_live;
        const int64_t p = live ? p0 + qi : 0;
        const int64_t row = live ? (int64_t)(uint32_t)__shfl(row_lane, qi, 64) : 0;
        const int nl = live ? job.nc : 0;
        uint32_t u[KPL], pw[KPL / 2];
#pragma unroll
        for (int q = 0; q < KPL / 8; ++q) {
            uint4 raw;
            if constexpr (KPL == 8) raw = ahead[rd]; else raw = keys_of(rd, q);
            const uint32_t wv[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) u[8 * q + j] = (id_of(8 * q + j) < nl) ? ((wv[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) + 1u : 0u;
            // the same keys packed, two per register, lists the bucket does not have as key 0 (below every rank asked for)
            const int nv = nl - (16 * q + sub) * 8;                  // keys of this piece that exist
#pragma unroll
            for (int m = 0; m < 4; ++m)
                pw[4 * q + m] = wv[m] & ((2 * m < nv ? 0xFFFFu : 0u) | (2 * m + 1 < nv ? 0xFFFF0000u : 0u));
        }
        const int wnt = min(np, nl);
        // T = the wnt-th largest key + 1 (0: no list), bit by bit: the keys at or above the candidate are counted lane-locally
        // with the packed 16-bit operations (two keys per instruction), then once over the query's 16 lanes on the DPP network --
        // the first form paid a compare, a shifted ballot and a count per KEY and bit and was VALU-bound on exactly that.
        uint32_t T = 0;
        {
            uint32_t Tk = 0;
            for (int bit = 15; bit >= 0; --bit) {
                const uint32_t c = Tk | (1u << bit);
                const uint32_t cm1 = (c - 1u) * 0x00010001u;
                uint32_t acc = 0u;
#pragma unroll
                for (int m = 0; m < KPL / 2; ++m) acc = pk_add(acc, pk_min(pk_sub_sat(pw[m], cm1), 0x00010001u));
                const int cnt = row16_sum((int)((acc & 0xFFFFu) + (acc >> 16)));
                Tk = cnt >= wnt ? c : Tk;
            }
            T = wnt > 0 ? Tk + 1u : 0u;
        }
        const float Tv = (float)(max(T, 1u) - 1u) * (1.f / 65535.f);
        const float e = 1.3e-3f * Tv + 1.2e-5f;
        const int delta = 2 * ((int)ceilf(e * 65535.f) + 1) + 2;
        int n_hi = 0, n_mem = 0;
        uint32_t h = 0, m = 0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int df = (int)u[j] - (int)T;
            const bool h1 = u[j] != 0u && df > delta;
            const bool m1 = u[j] != 0u && df <= delta && df >= -delta;
            h |= h1 ? (1u << j) : 0u;
            m |= m1 ? (1u << j) : 0u;
            n_hi += gcount(h1);
            n_mem += gcount(m1);
        }
        const int nd = wnt - n_hi;
        bool ambiguous = wnt > 0 && n_mem > nd;
        bool handed = false;
        if (ambiguous && n_mem > kCap) {                             // too many members for the wave's lists
            if (sub == 0) {
                const int at = atomicAdd(a.ovf_count, 1);
                if (at < a.ovf_cap) a.ovf_list[at] = (int32_t)g;
            }
            handed = true;
            ambiguous = false;
        }
        if (sub == 0) {
            q_cnt[qi] = ambiguous ? n_mem : 0;
            q_row[qi] = row;
            q_cbase[qi] = job.c_row0;
            q_need[qi] = nd;
            q_nmem[qi] = n_mem;
            q_flag[qi] = ((live && wnt > 0 && !handed) ? 1 : 0) | (ambiguous ? 2 : 0);
        }
        int base = 0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const bool m1 = ((m >> j) & 1u) && ambiguous;
            const uint32_t gm = (uint32_t)(__ballot(m1) >> sh) & 0xFFFFu;
            if (m1) m_id[qi][base + __popc(gm & ((1u << sub) - 1u))] = id_of(j);
            base += __popc(gm);
        }
        s_hi[rd][lane] = h;
        s_mem[rd][lane] = m;
        if (live && !handed)
            for (int i = wnt + sub; i < np; i += 16) a.probes[p * np + i] = -1;
    }
    wave_lds_sync();
    {
        int off[17];                                                 // (wave-uniform: scalar registers)
        off[0] = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) off[k + 1] = off[k] + __builtin_amdgcn_readfirstlane(q_cnt[k]);
        if (a.sp_cols == nullptr) {
            for (int i = lane; i < off[16]; i += 64) {
                int k = 0;
#pragma unroll
                for (int kk = 1; kk < 16; ++kk) k = off[kk] <= i ? kk : k;
                const int mm = i - off[k];
                m_val[k][mm] = exact_dot(a.X + q_row[k] * a.d, a.C + (q_cbase[k] + m_id[k][mm]) * (int64_t)a.d, a.d);
            }
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int lo = off[8 * half], hi = off[8 * half + 8];
                if (hi > lo) {                                   // (wave-uniform)
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {             // the rows of this half's ambiguous queries: one round trip
                        const int k = 8 * half + kk;
                        const bool want = q_cnt[k] > 0;
                        sp_c[kk][lane] = want ? a.sp_cols[q_row[k] * kSparseW + lane] : kColPad;
                        sp_v[kk][lane] = want ? a.sp_vals[q_row[k] * kSparseW + lane] : 0.f;
                    }
                    wave_lds_sync();
                    for (int i = lo + lane; i < hi; i += 64) {
                        int k = 8 * half;
#pragma unroll
                        for (int kk = 1; kk < 8; ++kk) k = off[8 * half + kk] <= i ? 8 * half + kk : k;
                        const int mm = i - off[k];
                        const float* c = a.C + (q_cbase[k] + m_id[k][mm]) * (int64_t)a.d;
                        float v;
                        if (sp_c[k - 8 * half][0] == kColDense) v = exact_dot(a.X + q_row[k] * a.d, c, a.d);     // (more than 64 non-zeros)
                        else v = coarse_sparse_chain(sp_c[k - 8 * half], sp_v[k - 8 * half], c);
                        m_val[k][mm] = v;
                    }
                    wave_lds_sync();
                }
            }
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int qi = 4 * rd + grp;
        uint32_t m = s_mem[rd][lane];
        const uint32_t h = s_hi[rd][lane];
        const int flag = q_flag[qi], n_mem = q_nmem[qi], nd = q_need[qi];
        if (flag & 2) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                if (!((m >> j) & 1u)) continue;
                const int me = id_of(j);
                float mine = 0.f;
                for (int i = 0; i < n_mem; ++i) mine = m_id[qi][i] == me ? m_val[qi][i] : mine;
                int rank = 0;
                for (int i = 0; i < n_mem; ++i) {
                    const float v = m_val[qi][i];
                    const int id = m_id[qi][i];
                    rank += (v > mine || (v == mine && id < me)) ? 1 : 0;
                }
                if (rank >= nd) m &= ~(1u << j);
            }
        }
        int32_t* out = a.probes + (qi < n_live ? p0 + qi : 0) * np;
        int base = 0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const bool in = (flag & 1) && (((h | m) >> j) & 1u);
            const uint32_t gm = (uint32_t)(__ballot(in) >> sh) & 0xFFFFu;
            if (in) out[base + __popc(gm & ((1u << sub) - 1u))] = id_of(j);
            base += __popc(gm);
        }
    }
}

// Buckets with 129..512 lists: the same algorithm with ALL 64 lanes on one query (8 keys per lane), sixteen queries per wave one
// after the other.  The 16-lanes-per-query form holds 32 keys per lane there (168 registers: three waves per SIMD) and pays
// 17 x 32 compare + ballot steps per round of four queries; here a lane holds 8 keys, the per-query state of the two phases
// lives in LDS, the wave runs at eight per SIMD, and the round of exact chains is still one per sixteen queries.
// G > 1 (round 5): 8 G keys per lane for buckets with up to 512 G lists (1,024 / 2,048: `--batch_size 65536` on windows of
// 40 k+ spectra, SURVEY 8d's C4 row) -- key (g, j) of a lane is list 512 g + 8 lane + j, one 16-byte load per g.
template <int G>
__device__ __forceinline__ void coarse16x_body(const Coarse16Args& a) {
    constexpr int kCap = 16;
    constexpr int KX = 8 * G;                                            // keys per lane
    typedef typename std::conditional<G == 1, uint8_t, uint32_t>::type MaskT;
    __shared__ float m_val[16][kCap];
    __shared__ int32_t m_id[16][kCap];
    __shared__ int32_t q_cnt[16], q_need[16], q_nmem[16], q_want[16], q_flag[16];      // flag: 1 emit, 2 ambiguous
    __shared__ int64_t q_row[16], q_cbase[16], q_pos[16];
    __shared__ MaskT s_hi[16][64], s_mem[16][64];                        // per query and lane: which of the lane's 8 G keys
    __shared__ uint32_t hist[320];
    __shared__ __attribute__((aligned(16))) uint16_t sp_c[8][kSparseW];
    __shared__ __attribute__((aligned(16))) float sp_v[8][kSparseW];
    const int lane = threadIdx.x;
    const int np = a.np;
    auto wcnt = [&](bool pred) -> int { return __popcll(__ballot(pred)); };
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // The wave's 16 queries are consecutive slots of ONE tile: its job is looked up once, the 16 sorted rows come by one load
    // (lane = query), and query qi + 1's keys are in flight while query qi is worked on -- the first form paid four dependent
    // round trips (tile -> job -> perm -> keys) per query, sixteen times in a row per wave.
    const int64_t g0 = (int64_t)blockIdx.x * 16;
    const int64_t t = g0 >> 5;
    const bool live_w = t < a.n_tiles;
    DenseJob job{};
    if (live_w) job = a.jobs[a.tile_job[t]];
    const int lt = (int)(t - job.tile0);
    const int ql0 = (int)(g0 & 31);
    const int n_live = live_w ? max(0, min(16, job.nq - (32 * lt + ql0))) : 0;      // queries 0 .. n_live - 1 of the wave exist
    const int64_t p0 = job.q_row0 + 32 * (int64_t)lt + ql0;
    const int row_lane = lane < n_live ? a.perm[p0 + lane] : 0;
    const int nl_w = n_live > 0 ? job.nc : 0;
    struct KeyRegs { uint4 v[G]; };
    auto id_of = [&](int x) -> int { return 512 * (x >> 3) + lane * 8 + (x & 7); };      // list of the lane's key x = 8 g + j
    auto keys_of = [&](int qi) -> KeyRegs {
        KeyRegs raw;
        const int64_t rw = (int64_t)(uint32_t)__builtin_amdgcn_readlane(row_lane, qi);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            raw.v[g] = make_uint4(0, 0, 0, 0);
            if (qi < n_live && 512 * g + lane * 8 < nl_w)
                raw.v[g] = *reinterpret_cast<const uint4*>(a.ckeys + rw * (int64_t)a.stride + 512 * g + lane * 8);
        }
        return raw;
    };
    KeyRegs raw_next = keys_of(0);
    for (int qi = 0; qi < 16; ++qi) {
        const int64_t g = g0 + qi;                                       // tile-order slot of the query
        const bool live = qi < n_live;
        const int64_t p = live ? p0 + qi : 0;
        const int64_t row = live ? (int64_t)(uint32_t)__builtin_amdgcn_readlane(row_lane, qi) : 0;
        const int nl = live ? job.nc : 0;
        uint32_t u[KX];
        {
            const KeyRegs raw = raw_next;
            if (qi + 1 < 16) raw_next = keys_of(qi + 1);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t wv[4] = {raw.v[g].x, raw.v[g].y, raw.v[g].z, raw.v[g].w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    u[8 * g + j] = (id_of(8 * g + j) < nl) ? ((wv[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) + 1u : 0u;
            }
        }
        const int wnt = min(np, nl);
        // T = the wnt-th largest u: two histogram levels in LDS (high byte, then the low byte inside the bin that holds it), as in
        // select16_kernel -- 2 x 8 LDS atomics + two suffix sums over the lanes instead of 17 x 8 compare-and-count steps
        uint32_t T = 0;
        if (wnt > 0) {
            auto level = [&](auto bin_of_key, int kk, int* above) -> int {
                hist[lane] = 0u; hist[lane + 64] = 0u; hist[lane + 128] = 0u; hist[lane + 192] = 0u; hist[lane + 256] = 0u;
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < KX; ++i) {
                    const int b = bin_of_key(u[i]);
                    if (b >= 0) __hip_atomic_fetch_add(&hist[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                wave_lds_sync();
                uint32_t c[5];                                            // bins 5 lane .. 5 lane + 4
#pragma unroll
                for (int jj = 0; jj < 5; ++jj) c[jj] = hist[5 * lane + jj];
                const int own = (int)(c[0] + c[1] + c[2] + c[3] + c[4]);
                const int pre = wave_prefix_sum(own);                     // (DPP network)
                const int suf = __builtin_amdgcn_readlane(pre, 63) - pre + own;      // keys in this lane's bins and all higher ones
                const unsigned long long reach = __ballot(suf >= kk);     // (a prefix of the lanes: suf falls with the lane)
                const int L = 63 - __clzll(reach);
                int acc = suf - own, bin = 5 * lane;
#pragma unroll
                for (int jj = 4; jj >= 0; --jj) {
                    if (acc + (int)c[jj] >= kk) { bin = 5 * lane + jj; break; }
                    acc += (int)c[jj];
                }
                *above = __builtin_amdgcn_readlane(acc, L);
                const int res = __builtin_amdgcn_readlane(bin, L);
                wave_lds_sync();
                return res;
            };
            int above1 = 0, above2 = 0;
            const int b1 = level([&](uint32_t key) -> int { return key ? (int)min(key >> 8, 255u) : -1; }, wnt, &above1);
            const int b2 = level([&](uint32_t key) -> int { return (key && (int)min(key >> 8, 255u) == b1) ? (int)key - (b1 << 8) : -1; },
                                 wnt - above1, &above2);
            T = (uint32_t)((b1 << 8) + b2);
        }
        const float Tv = (float)(max(T, 1u) - 1u) * (1.f / 65535.f);
        const float e = 1.3e-3f * Tv + 1.2e-5f;
        const int delta = 2 * ((int)ceilf(e * 65535.f) + 1) + 2;
        int n_hi = 0, n_mem = 0;
        uint32_t h = 0, m = 0;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const int df = (int)u[j] - (int)T;
            const bool h1 = u[j] != 0u && df > delta;
            const bool m1 = u[j] != 0u && df <= delta && df >= -delta;
            h |= h1 ? (1u << j) : 0u;
            m |= m1 ? (1u << j) : 0u;
            n_hi += wcnt(h1);
            n_mem += wcnt(m1);
        }
        const int nd = wnt - n_hi;
        bool ambiguous = wnt > 0 && n_mem > nd;
        bool handed = false;
        if (ambiguous && n_mem > kCap) {                                 // too many members for the wave's lists
            if (lane == 0) {
                const int at = atomicAdd(a.ovf_count, 1);
                if (at < a.ovf_cap) a.ovf_list[at] = (int32_t)g;
            }
            handed = true;
            ambiguous = false;
        }
        if (lane == 0) {
            q_cnt[qi] = ambiguous ? n_mem : 0;
            q_row[qi] = row;
            q_cbase[qi] = job.c_row0;
            q_need[qi] = nd;
            q_nmem[qi] = n_mem;
            q_want[qi] = wnt;
            q_pos[qi] = p;
            q_flag[qi] = ((live && wnt > 0 && !handed) ? 1 : 0) | (ambiguous ? 2 : 0);
        }
        s_hi[qi][lane] = (MaskT)h;
        s_mem[qi][lane] = (MaskT)m;
        int base = 0;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const bool m1 = ((m >> j) & 1u) && ambiguous;
            const unsigned long long gm = __ballot(m1);
            if (m1) m_id[qi][base + __popcll(gm & lt_mask)] = id_of(j);
            base += __popcll(gm);
        }
        if (live && !handed)
            for (int i = wnt + lane; i < np; i += 64) a.probes[p * np + i] = -1;
    }
    wave_lds_sync();
    {
        int off[17];
        off[0] = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) off[k + 1] = off[k] + q_cnt[k];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int lo = off[8 * half], hi = off[8 * half + 8];
            if (hi > lo) {                                               // (wave-uniform)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int k = 8 * half + kk;
                    const bool want = q_cnt[k] > 0;
                    sp_c[kk][lane] = want ? a.sp_cols[q_row[k] * kSparseW + lane] : kColPad;
                    sp_v[kk][lane] = want ? a.sp_vals[q_row[k] * kSparseW + lane] : 0.f;
                }
                wave_lds_sync();
                for (int i = lo + lane; i < hi; i += 64) {
                    int k = 8 * half;
#pragma unroll
                    for (int kk = 1; kk < 8; ++kk) k = off[8 * half + kk] <= i ? 8 * half + kk : k;
                    const int mm = i - off[k];
                    const float* c = a.C + (q_cbase[k] + m_id[k][mm]) * (int64_t)a.d;
                    float v;
                    if (sp_c[k - 8 * half][0] == kColDense) v = exact_dot(a.X + q_row[k] * a.d, c, a.d);
                    else v = coarse_sparse_chain(sp_c[k - 8 * half], sp_v[k - 8 * half], c);
                    m_val[k][mm] = v;
                }
                wave_lds_sync();
            }
        }
    }
    wave_lds_sync();
    for (int qi = 0; qi < 16; ++qi) {
        const int flag = q_flag[qi];
        if (!(flag & 1)) continue;                                       // (wave-uniform)
        uint32_t m = s_mem[qi][lane];
        const uint32_t h = s_hi[qi][lane];
        if (flag & 2) {
            const int nm = q_nmem[qi], need = q_need[qi];
#pragma unroll
            for (int j = 0; j < KX; ++j) {
                if (!((m >> j) & 1u)) continue;
                const int me = id_of(j);
                float mine = 0.f;
                for (int i = 0; i < nm; ++i) mine = m_id[qi][i] == me ? m_val[qi][i] : mine;
                int rank = 0;
                for (int i = 0; i < nm; ++i) {
                    const float v = m_val[qi][i];
                    const int id = m_id[qi][i];
                    rank += (v > mine || (v == mine && id < me)) ? 1 : 0;
                }
                if (rank >= need) m &= ~(1u << j);
            }
        }
        int32_t* out = a.probes + q_pos[qi] * np;
        int base = 0;
#pragma unroll
        for (int j = 0; j < KX; ++j) {
            const bool in = ((h | m) >> j) & 1u;
            const unsigned long long gm = __ballot(in);
            if (in) out[base + __popcll(gm & lt_mask)] = id_of(j);
            base += __popcll(gm);
        }
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void coarse16x_kernel(Coarse16Args a) { coarse16x_body<1>(a); }
__global__ __launch_bounds__(64) void coarse16x2_kernel(Coarse16Args a) { coarse16x_body<2>(a); }      // <= 1,024 lists
__global__ __launch_bounds__(64) void coarse16x4_kernel(Coarse16Args a) { coarse16x_body<4>(a); }      // <= 2,048 lists

// Buckets with more than 512 lists: the queries coarse16x{2,4}_kernel hands over (more than 16 keys within the bound of the
// n_probe-th: many equal similarities -- rare).  One 256-thread workgroup per query, thread t holds the keys of lists t,
// t + 256, ...; the n_probe-th largest key by the bitwise search with workgroup-wide counts, EVERY member re-evaluated by the
// exact chain and ranked against all members (value descending, list id ascending: the staged order), the chosen lists
// appended through an LDS cursor (any order: nothing depends on it).
template <int KPT>      // keys per thread: 256 KPT >= the launch's most lists
__global__ __launch_bounds__(256) void coarse16_big_kernel(Coarse16Args a) {
    constexpr int kMem = 256 * KPT;
    __shared__ float m_val[kMem];
    __shared__ int32_t m_id[kMem];
    __shared__ int32_t s_cnt[4], s_nmem, s_out;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n_items = (int64_t)min(*a.ovf_count, a.ovf_cap);
    auto block_count = [&](bool pred) -> int {                          // (every thread calls it the same number of times)
        const int c = __popcll(__ballot(pred));
        __syncthreads();
        if (lane == 0) s_cnt[wv] = c;
        __syncthreads();
        return s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    };
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int64_t g = a.ovf_list[item];                             // tile-order slot of the query
        const int64_t t = g >> 5;
        const int ql = (int)(g & 31);
        bool live = t < a.n_tiles;
        DenseJob job{};
        if (live) job = a.jobs[a.tile_job[t]];
        const int lt = (int)(t - job.tile0);
        live = live && 32 * lt + ql < job.nq;
        const int64_t p = live ? job.q_row0 + 32 * (int64_t)lt + ql : 0;
        const int64_t row = live ? a.perm[p] : 0;
        const int nl = live ? job.nc : 0;
        const int np = a.np;
        uint32_t u[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int id = tid + 256 * k;
            u[k] = id < nl ? (uint32_t)a.ckeys[row * (int64_t)a.stride + id] + 1u : 0u;
        }
        const int want = min(np, nl);
        uint32_t T = 0;
        for (int bit = 16; bit >= 0; --bit) {
            const uint32_t c = T | (1u << bit);
            int mine = 0;
#pragma unroll
            for (int k = 0; k < KPT; ++k) mine += u[k] >= c ? 1 : 0;
            int cnt = 0;                                                // sum of `mine` over the workgroup: KPT + 1 one-bit rounds
#pragma unroll
            for (int b = 0; (1 << b) <= KPT; ++b) cnt += block_count((mine >> b) & 1) << b;
            if (cnt >= want && want > 0) T = c;
        }
        const float Tv = (float)(max(T, 1u) - 1u) * (1.f / 65535.f);
        const float e = 1.3e-3f * Tv + 1.2e-5f;
        const int delta = 2 * ((int)ceilf(e * 65535.f) + 1) + 2;
        if (tid == 0) { s_nmem = 0; s_out = 0; }
        __syncthreads();
        int32_t* out = a.probes + p * np;
        int n_hi_mine = 0;
        uint32_t memb = 0;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int df = (int)u[k] - (int)T;
            const bool h1 = u[k] != 0u && df > delta;
            const bool m1 = u[k] != 0u && df <= delta && df >= -delta;
            n_hi_mine += h1 ? 1 : 0;
            if (h1 && live && want > 0) out[atomicAdd(&s_out, 1)] = tid + 256 * k;      // certainly among the n_probe best
            if (m1) {
                memb |= 1u << k;
                const int at = atomicAdd(&s_nmem, 1);
                m_id[at] = tid + 256 * k;
            }
        }
        int n_hi = 0;
#pragma unroll
        for (int b = 0; (1 << b) <= KPT; ++b) n_hi += block_count((n_hi_mine >> b) & 1) << b;
        __syncthreads();
        const int n_mem = s_nmem;
        const int need = want - n_hi;                                   // places left for the members
        for (int i = tid; i < n_mem; i += 256)
            m_val[i] = exact_dot(a.X + row * a.d, a.C + (job.c_row0 + m_id[i]) * (int64_t)a.d, a.d);
        __syncthreads();
        if (live && want > 0) {
#pragma unroll
            for (int k = 0; k < KPT; ++k) {
                if (!((memb >> k) & 1u)) continue;
                const int me = tid + 256 * k;
                float mine = 0.f;
                for (int i = 0; i < n_mem; ++i) mine = m_id[i] == me ? m_val[i] : mine;
                int rank = 0;
                for (int i = 0; i < n_mem; ++i) {
                    const float v = m_val[i];
                    const int id = m_id[i];
                    rank += (v > mine || (v == mine && id < me)) ? 1 : 0;
                }
                if (rank < need) out[atomicAdd(&s_out, 1)] = me;
            }
        }
        if (live)
            for (int i = want + tid; i < np; i += 256) out[i] = -1;
        __syncthreads();
    }
}

__global__ void tile_job_c16_kernel(const DenseJob* __restrict__ jobs, int n_jobs, int64_t n_tiles, int32_t* __restrict__ tile_job) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_tiles) tile_job[i] = find_job(jobs, n_jobs, i);
}

int launch_coarse16(fal_ctx* ctx, const Coarse16Args& a_in) {
    if (a_in.n_tiles <= 0) return FAL_OK;
    Coarse16Args a = a_in;
    int32_t* tj = nullptr;
    ctx->release(SLOT_TILEJOB);        // a launcher-local table: the previous launcher's pointer is dead
    FAL_TRY(ctx->reserve(SLOT_TILEJOB, sizeof(int32_t) * (size_t)(std::max<int64_t>(a.n_tiles, 1 << 16) + 32 * a.n_tiles + 64), (void**)&tj));
    StageScope ts(ctx, ST_COARSE);
    hipLaunchKernelGGL(tile_job_c16_kernel, dim3((unsigned)ceil_div(a.n_tiles, 256)), dim3(256), 0, ctx->stream, a.jobs, a.n_jobs,
                       a.n_tiles, tj);
    a.tile_job = tj;
    FAL_REQUIRE(a.n_tiles * 4 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many queries in one coarse launch");
    // the queries the wave-level kernel hands over (count in front), behind the tile table
    a.ovf_count = tj + std::max<int64_t>(a.n_tiles, 1 << 16);
    a.ovf_list = a.ovf_count + 16;
    a.ovf_cap = (int)std::min<int64_t>(32 * a.n_tiles, INT32_MAX);
    FAL_CHECK_HIP(hipMemsetAsync(a.ovf_count, 0, sizeof(int32_t), ctx->stream));
    const unsigned list_grid = (unsigned)std::min<int64_t>(a.n_tiles * 4, (int64_t)ctx->num_cus * 16);
    if (a.stride <= 128) {
        // (five waves per SIMD with 28 dwords per lane of spills left beats four with 8: coarse stage 6.5 vs 6.85 ms per 10 M)
        hipLaunchKernelGGL((coarse16w_kernel<8>), dim3((unsigned)(a.n_tiles * 2)), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL((coarse16_kernel<8, 16>), dim3(list_grid), dim3(256), 0, ctx->stream, a);
    } else if (a.stride > 512) {
        // 513..2,048 lists per bucket (`--batch_size 65536` on 40 k+-row windows): 16 / 32 keys per lane, workgroup-level overflow
        FAL_REQUIRE(a.sp_cols && a.stride <= 2048, FAL_EINTERNAL, "coarse16: stride %d without the rows' sparse form", a.stride);
        const unsigned big_grid = (unsigned)std::min<int64_t>(a.n_tiles * 32, (int64_t)ctx->num_cus * 8);
        if (a.stride <= 1024) {
            hipLaunchKernelGGL(coarse16x2_kernel, dim3((unsigned)(a.n_tiles * 2)), dim3(64), 0, ctx->stream, a);
            hipLaunchKernelGGL((coarse16_big_kernel<4>), dim3(big_grid), dim3(256), 0, ctx->stream, a);
        } else {
            hipLaunchKernelGGL(coarse16x4_kernel, dim3((unsigned)(a.n_tiles * 2)), dim3(64), 0, ctx->stream, a);
            hipLaunchKernelGGL((coarse16_big_kernel<8>), dim3(big_grid), dim3(256), 0, ctx->stream, a);
        }
    } else {
        if (a.sp_cols) hipLaunchKernelGGL(coarse16x_kernel, dim3((unsigned)(a.n_tiles * 2)), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL((coarse16w_kernel<32>), dim3((unsigned)(a.n_tiles * 2)), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL((coarse16_kernel<32, 8>), dim3(list_grid), dim3(128), 0, ctx->stream, a);
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::coarse16x_kernel);      // (fal_ctx_plan: this unit's code object is loaded up front)
