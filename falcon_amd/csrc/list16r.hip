// list16r_kernel: the fine scan of indexed buckets on the f16 matrix cores, second form (round 5).  Same inputs, same key for
// every (query, list row) pair at the same place as list16_kernel (ivf16.hip) -- the operand roles and the k-slot map of
// every MFMA step are the same, so the keys are BIT-IDENTICAL -- but the residency is transposed:
//
//   list16_kernel   the list's rows live in REGISTERS (one 32-row slice per wave), the probing queries stream through a
//                   shared LDS ring, the four waves step in lockstep (one barrier per 32-query chunk).  A list holds ~70
//                   rows: wave 3 has no rows, wave 2 mostly padding, and a wave's step is the SUM of its DMA issue time and
//                   its matrix work (profiles/NOTES.md r3, r4: 0.15 of the f16 peak, no variant of that shape moved it).
//   list16r_kernel  the list's rows (<= 128, float16) are loaded ONCE into LDS and are every wave's B operand; each wave
//                   walks ITS OWN 32-query chunks of the list's probe stream (chunk c belongs to wave c mod 4) with the
//                   queries as the A operand in registers: no workgroup barrier after the prologue, no idle wave (every
//                   wave has queries whatever the list's size), no metadata ring.  The bytes in flight live in REGISTERS,
//                   not in LDS: a chunk's 32 rows are fetched row-contiguously (one 16-byte piece per lane and row: whole
//                   cache lines, one TA instruction per row) into 128 staging registers while the previous chunk is
//                   multiplied, then turned into the MFMA's row-per-lane layout through a 13 KB wave-private LDS tile
//                   (two halves of 16 rows).  Plain loads and stores only, and a FIXED number of them per loop iteration
//                   (the loop is specialised on the tile's slice count, the last chunk is peeled): hipcc counts every
//                   VMEM operation itself and its `s_waitcnt vmcnt(N)` are exact, not vmcnt(0).
//
// Reference: README.md:107-113, 137-142 (n_probe lists per query, n_neighbors_ann neighbours); faiss IndexIVFFlat is a
// dependency of the reference, not in the snapshot.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "scan.h"
#include "ivf16.h"

namespace fal {

typedef _Float16 half8r __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 is a struct whose copies are
                                                                 // memcpys -- an array of it stays in scratch)

struct List16rTile {                 // what the prologue found out about the workgroup's tile
    int64_t e0;                      // first probe-table entry of the list
    int nq, n_chunks;                // queries probing the list, 32-query chunks
    int l_rows, row0;                // rows of the list, first row of this tile
    uint32_t row_max;
};

// One wave, its chunks c = w, w + 4, ... of the list's probe stream against the NS resident 32-row slices.
template <int STEPS, int NS>
__device__ __forceinline__ void list16r_body(const List16Args& a, const List16rTile& T, const unsigned char* lrows,
                                             unsigned char* tb, int32_t* dm, int w, int lane) {
    constexpr int D = STEPS * 16, DH = D / 2, PIECES = D / 8, RS = D * 2 + 16;
    constexpr int NB = STEPS < 4 ? STEPS : 4;              // LDS operand reads in flight ahead of the MFMAs
    constexpr int kMid = STEPS / 2;
    const int r = lane & 31, h = lane >> 5;
    const int pl = min(lane, PIECES - 1);                  // (lanes past the row's last piece re-read it: no EXEC games around the loads)
    const uint64_t xbase = reinterpret_cast<uint64_t>(a.X16);
    // per chunk and lane one dword of metadata: lanes 0-31 the sorted row of query r, lanes 32-63 its destination (the low
    // dword of the element index in `keys` where the query's segment for this list starts)
    auto load_meta = [&](int cc) __attribute__((always_inline)) -> int32_t {
        const int64_t e = T.e0 + min(32 * cc + r, T.nq - 1);
        return h ? reinterpret_cast<const int32_t*>(a.inv_dest + e)[0] : a.inv_row[e];
    };
    u32x4 stage[32];
    // the 32 rows of a chunk: lane j holds row j's id -> its address (vector arithmetic, once), then per row two v_readlane
    // and one load with a scalar base
    auto issue_rows = [&](int32_t ids) __attribute__((always_inline)) {
        const uint64_t ra = xbase + (uint64_t)min((uint32_t)ids, T.row_max) * (uint64_t)(D * 2);
        const int ra_lo = (int)(uint32_t)ra, ra_hi = (int)(uint32_t)(ra >> 32);
        // eight rows at a time: their 16 scalars first, then the 8 loads (a load right behind the v_readlane that wrote its
        // base costs five wait states each time)
#pragma unroll
        for (int j0 = 0; j0 < 32; j0 += 8) {
            uint64_t b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                b[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(ra_hi, j0 + j) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readlane(ra_lo, j0 + j);
            __builtin_amdgcn_sched_barrier(0);
            // (GLOBAL pointers: a pointer made from an integer is a flat one, and flat loads return out of order -- every wait
            // behind them becomes vmcnt(0) lgkmcnt(0))
#pragma unroll
            for (int j = 0; j < 8; ++j) stage[j0 + j] = reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(b[j])[pl];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    half8r A[STEPS];                                       // lane (r, h): query r of the chunk, k-half h (list16_kernel's stream operand)
    auto transpose_half = [&](auto half_c) __attribute__((always_inline)) {
        constexpr int half = decltype(half_c)::value;
        if (lane < PIECES) {
#pragma unroll
            for (int j = 0; j < 16; ++j) *reinterpret_cast<u32x4*>(tb + j * RS + lane * 16) = stage[16 * half + j];
        }
        wave_lds_sync();
        if ((r >> 4) == half) {
            const unsigned char* src = tb + (r & 15) * RS + h * (DH * 2);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) A[s] = *reinterpret_cast<const half8r*>(src + s * 16);
        }
        wave_lds_sync();
    };
    auto transpose = [&]() __attribute__((always_inline)) {
        transpose_half(std::integral_constant<int, 0>{});
        transpose_half(std::integral_constant<int, 1>{});
    };
    const uint32_t base_lo = (uint32_t)a.keys_base;
    unsigned char* keys_b = reinterpret_cast<unsigned char*>(a.keys);
    const uint32_t sink_off = 2u * ((uint32_t)(a.sink - a.keys) + (uint32_t)lane);

    // The keys of the last finished (chunk, slice) wait in `pk`, two per register (v_cvt_pknorm_u16_f32), with their
    // destinations, and leave in the middle of the NEXT chain: 16 two-byte stores, each 32 consecutive keys of one query's
    // segment per half wave (D[query][list row]: lane = list row, registers = 16 queries).
    uint32_t pk[8];
    int4 p_md[4];
    uint32_t p_k2 = 0;
    int p_left = 0;                                        // (<= 0: nothing pending / every store goes to the sink)
    bool p_valid = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) pk[j] = 0u;
#pragma unroll
    for (int g = 0; g < 4; ++g) p_md[g] = make_int4(0, 0, 0, 0);
    auto store_pending = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q0 = (i & 3) + 8 * (i >> 2);
            const int4 m4 = p_md[i >> 2];
            const int32_t mdq = (i & 3) == 0 ? m4.x : (i & 3) == 1 ? m4.y : (i & 3) == 2 ? m4.z : m4.w;
            uint32_t off = ((uint32_t)mdq << 1) + p_k2;
            off = (p_valid && q0 < p_left) ? off : sink_off;
            const uint32_t v = pk[i >> 1];
            *reinterpret_cast<uint16_t*>(keys_b + (size_t)off) = (uint16_t)((i & 1) ? (v >> 16) : (v & 0xFFFFu));
        }
    };
    // chunk c (queries in A, destinations in m_cur's upper half) against the resident slices
    auto compute = [&](int c, int32_t m_cur) __attribute__((always_inline)) {
        if (h) dm[r] = m_cur;
        wave_lds_sync();
        int4 mdv[4];                                       // lane (r, h): the destinations of queries 4 h + {0..3} + 8 g
#pragma unroll
        for (int g = 0; g < 4; ++g) mdv[g] = *reinterpret_cast<const int4*>(dm + 4 * h + 8 * g);
        wave_lds_sync();
        const int left = T.nq - 32 * c - 4 * h;            // queries q0 < left of this chunk exist
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const unsigned char* lb = lrows + (32 * sl + r) * RS + h * (DH * 2);
            // (a scheduling region per chain: the group barriers below pick "any LDS read" / "any MFMA" of their region, and
            // with the NS chains of a chunk in one region they took each other's reads -- the ring collapsed to one register
            // and every MFMA waited for its own operand with lgkmcnt(0))
            __builtin_amdgcn_sched_barrier(0);
            half8r ring[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) ring[j] = *reinterpret_cast<const half8r*>(lb + j * 16);
            __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const half8r b = ring[s % NB];
                if (s + NB < STEPS) ring[s % NB] = *reinterpret_cast<const half8r*>(lb + (s + NB) * 16);
                // queries are the A operand, list rows B: D[query][list row]
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], b, acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (s == kMid) store_pending();
            }
            asm volatile("s_nop 15" : "+a"(acc));          // MFMA -> accumulator read behind a taken branch (simtile.h)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const us2 two = __builtin_amdgcn_cvt_pknorm_u16(acc[2 * j], acc[2 * j + 1]);
                pk[j] = (uint32_t)two.x | ((uint32_t)two.y << 16);
            }
            const int pos = T.row0 + 32 * sl + r;          // position inside the list = offset inside a query's segment
            p_valid = pos < T.l_rows;
            p_k2 = 2u * ((uint32_t)pos - base_lo);
            p_left = left;
#pragma unroll
            for (int g = 0; g < 4; ++g) p_md[g] = mdv[g];
        }
    };

    int c = w;
    int32_t m_cur = load_meta(c);
    int32_t m_nxt = load_meta(c + 4);
    issue_rows(m_cur);
    transpose();
    while (c + 4 < T.n_chunks) {
        issue_rows(m_nxt);                                 // in flight while chunk c is multiplied
        const int32_t m_nn = load_meta(c + 8);
        compute(c, m_cur);
        transpose();
        m_cur = m_nxt;
        m_nxt = m_nn;
        c += 4;
    }
    compute(c, m_cur);
    store_pending();
}

template <int STEPS>
__global__ __launch_bounds__(256, 1) void list16r_kernel(List16Args a) {
    constexpr int D = STEPS * 16, PIECES = D / 8, RS = D * 2 + 16;
    static_assert(PIECES <= 64, "one load instruction per row");
    static_assert(128 * RS + 4 * 16 * RS + 4 * 32 * 4 <= 160 * 1024, "the tile, the staging rows and the metadata must fit one CU's 160 KB of LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lrows[128 * RS];       // the tile's list rows (B operand)
    __shared__ __attribute__((aligned(16))) unsigned char tbuf_all[4][16 * RS];  // per wave: 16 query rows on their way to A
    __shared__ __attribute__((aligned(16))) int32_t dmeta[4][32];                // per wave: the chunk's 32 destinations
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    const int64_t lt = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd) return;
    const int64_t t = a.tile_begin + lt;
    if (t >= a.ltile_off[a.list_end]) return;
    int64_t lo = a.list_begin, hi = a.list_end - 1;        // last list with ltile_off <= t
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (a.ltile_off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t L = lo;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t l_row0 = a.list_off[L];
    List16rTile T;
    T.l_rows = (int)(a.list_off[L + 1] - l_row0);
    T.row0 = 128 * (int)(t - a.ltile_off[L]);              // this tile's first row of the list
    const int t_rows = min(128, T.l_rows - T.row0);        // (whole workgroup: the same)
    T.e0 = a.inv_off[L];
    T.nq = (int)(a.inv_off[L + 1] - T.e0);                 // queries probing the list
    if (T.nq <= 0 || t_rows <= 0) return;
    const int n_slices = (t_rows + 31) >> 5;
    T.row_max = (uint32_t)(a.n_rows - 1);
    T.n_chunks = (T.nq + 31) >> 5;
    const u32x4* Xp = reinterpret_cast<const u32x4*>(a.X16);

    // ---- prologue: the list rows of the tile -> LDS (row-contiguous; rows past the tile's end are zero) -------------------
    // eight rows per wave and turn: their ids by one coalesced load, the rows by one 16-byte piece per lane
    for (int i0 = 8 * w; i0 < 32 * n_slices; i0 += 32) {
        const int my = i0 + (lane & 7);
        const int32_t rid = my < t_rows ? a.perm[l_row0 + T.row0 + my] : 0;
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t row = min((uint32_t)__builtin_amdgcn_readlane(rid, j), T.row_max);
            v[j] = u32x4{0u, 0u, 0u, 0u};
            if (lane < PIECES && i0 + j < t_rows) v[j] = Xp[(int64_t)row * PIECES + lane];
        }
        if (lane < PIECES) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(lrows + (i0 + j) * RS + lane * 16) = v[j];
        }
    }
    __syncthreads();                                       // the only workgroup barrier
    if (w >= T.n_chunks) return;
    switch (n_slices) {
        case 1: list16r_body<STEPS, 1>(a, T, lrows, tbuf_all[w], dmeta[w], w, lane); break;
        case 2: list16r_body<STEPS, 2>(a, T, lrows, tbuf_all[w], dmeta[w], w, lane); break;
        case 3: list16r_body<STEPS, 3>(a, T, lrows, tbuf_all[w], dmeta[w], w, lane); break;
        default: list16r_body<STEPS, 4>(a, T, lrows, tbuf_all[w], dmeta[w], w, lane); break;
    }
}

int launch_list16r(fal_ctx* ctx, const List16Args& a) {
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8)), block(256);
    switch (a.d / 16) {
        case 4: hipLaunchKernelGGL((list16r_kernel<4>), grid, block, 0, ctx->stream, a); break;
        case 8: hipLaunchKernelGGL((list16r_kernel<8>), grid, block, 0, ctx->stream, a); break;
        case 16: hipLaunchKernelGGL((list16r_kernel<16>), grid, block, 0, ctx->stream, a); break;
        case 25: hipLaunchKernelGGL((list16r_kernel<25>), grid, block, 0, ctx->stream, a); break;
        default:
            set_error("list16r: low_dim %d has no instantiation (64, 128, 256, 400)", a.d);
            return FAL_EUNSUPPORTED;
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::list16r_kernel<25>);      // (fal_ctx_plan: this unit's code object is loaded up front)
