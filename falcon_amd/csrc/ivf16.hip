// a7 + a8 for IVF buckets with a float16 prefilter: the fine scan on the f16 matrix cores, the exact work only where it
// decides something.  Output BIT-IDENTICAL to the staged path (ivf_fine.hip + select_kernel<MODE_IVF, true>).
//
// The staged path computes the exact float32 similarity of every (query, probed-list row) pair on the fp32 matrix cores
// (16 passes per k-pair: the dominant cost at 10 M spectra) and selects the k_ann best per query.  What the neighbour lists
// need is (i) the exact similarities of the few candidates inside the query's precursor window and (ii) the k_ann-th best
// key of the query, to decide which of them are among the k_ann best.  So:
//
//   list16_kernel   list-major like ivf_list4_kernel -- a 4-wave workgroup keeps up to 128 rows of ONE list resident, the
//      queries probing the list stream through LDS (global_load_lds, double-buffered) -- but on float16 copies of the rows
//      with v_mfma_f32_32x32x16_f16 (1/16 of the matrix-pipe cycles).  Every approximate similarity leaves as a 16-bit
//      fixed-point key (round(v * 65535): half the bytes of a float, finer than float16 above 0.016) into the query's
//      segment, laid out exactly like the staged path's sims.
//   select16_kernel   one wave per query: all keys of the query in registers, the k_ann-th largest key T~ by a bitwise search
//      with ballot counts; |key / 65535 - exact| <= e(T~) (bound below) puts the exact k_ann-th best value inside
//      [T~ - e, T~ + e].  Out: the thresholds, how many keys lie certainly above, and the few candidates within 2e of T~
//      ("members": the only ones whose exact value can decide the k_ann-th key), as positions in the query's key stream.
//   band_kernel<., IVF> / resolve_kernel / ivf_fallback_kernel (fused.hip)   exact similarities of the precursor window on the
//      fp32 matrix cores (candidates outside the query's probed lists masked out), ambiguous candidates against the exact
//      k_ann-th key, sort, neighbour lists; queries the hand-off cannot hold take the exact fallback.
//
// Error bound: |approx - exact| <= 1.3e-3 * approx + 2e-6 (fused.hip), the key adds 0.5 / 65535 + rounding < 7.7e-6:
//   |key / 65535 - exact| <= e(v) = 1.3e-3 * v + 1.2e-5.
// With T~ the k-th largest key value: at least k candidates have exact >= T~ - e(T~) and at least N - k + 1 have
// exact <= T~ + e(T~), so the exact k-th best value lies in [L, U] = [T~ - e, T~ + e]; a key above T~ + 2e (+ the growth of e
// over that distance, one key unit of slack) is certainly above U, one below T~ - 2e certainly below L.
//
// Reference: README.md:107-113, 137-142 (n_probe lists per query, n_neighbors_ann neighbours, precursor filter); faiss
// IndexIVFFlat is a dependency of the reference, not in the snapshot.
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "fused.h"
#include "ivf16.h"
#include "kept16.h"

namespace fal {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));



// ------------------------------------------------------------------------------------------------------------
// list16_kernel: tile = (one inverted list, 128 of its rows); see ivf_list4_kernel for the structure.  Two things bound the
// first version of this kernel, both in the gather of the streamed query rows (a chunk of 32 rows = 0.33 us of matrix work):
//   * request count: with a row per lane, a DMA instruction touched 64 different cache lines for 16 bytes each (the other
//     pieces of a line only hit if it survived in the 32 kB L1 until the step that wanted them; with ~100 kB of chunks in
//     flight per CU it did not).  Now ONE ROW PER INSTRUCTION: lanes 0 .. d/8-1 fetch the row's consecutive 16-byte pieces
//     (800 B = 6.25 lines), the row lands row-major in LDS, rows padded to 2d + 16 bytes (conflict-free b128 operand reads);
//   * latency: THREE stream buffers, the DMA of chunk c + 2 is issued while chunk c is computed.  vmcnt is an in-order counter
//     and hipcc answers any VMEM instruction it cannot count with vmcnt(0), so the loop's memory traffic is spelled out: per
//     step and wave exactly 1 metadata DMA (row ids and destinations of chunk c + 3, into an 8-slot LDS ring), 8 row DMAs and
//     16 key stores; DMAs and waits in inline asm (the compiler never sees a load result it would have to wait for); before
//     the barrier of step c + 1, vmcnt(24) -- everything but step c's row DMAs and stores -- has retired chunk c + 1 and the
//     metadata of chunk c + 3.
// ------------------------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(256, STEPS > 32 ? 1 : 2) void list16_kernel(List16Args a) {
    constexpr int D = STEPS * 16, DH = D / 2;
    constexpr int RD = (D / 8 + 63) / 64;                 // DMA instructions per row (64 x 16 bytes each): 2 for low_dim 800
    constexpr int kRowOps = 8 * RD;                       // row DMAs per step and wave
    constexpr int NB = STEPS < 4 ? STEPS : 4;             // LDS operand reads in flight ahead of the MFMAs
    constexpr int RS = D * 2 + 16;                        // LDS row stride in bytes
    __shared__ __attribute__((aligned(16))) unsigned char sbuf0[32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char sbuf1[32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char sbuf2[32 * RS];
    __shared__ int32_t meta[8][64];                       // per chunk: [32, 64) destination (low dword) of query r ([0, 32): its row)
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    const int64_t lt = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd) return;
    const int64_t t = a.tile_begin + lt;
    if (t >= a.ltile_off[a.list_end]) return;
    int64_t lo = a.list_begin, hi = a.list_end - 1;       // last list with ltile_off <= t
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (a.ltile_off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t L = lo;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = 4 * (int)(t - a.ltile_off[L]) + w;  // this wave's 32-row slice of the list
    const int64_t l_row0 = a.list_off[L];
    const int l_rows = (int)(a.list_off[L + 1] - l_row0);
    const int nrow = min(32, l_rows - 32 * slice);        // rows of this slice (<= 0: the wave only helps loading)
    const int64_t e0 = a.inv_off[L];
    const int nq = (int)(a.inv_off[L + 1] - e0);          // queries probing the list
    if (nq <= 0 || l_rows <= 0) return;
    const bool active = nrow > 0;

    half8 q[STEPS];                                        // the resident operand: list row 32*slice + r, k-half h
    {
        const int64_t rr = a.perm[l_row0 + min(32 * slice + min(r, max(nrow, 1) - 1), l_rows - 1)];
        const half8* src = reinterpret_cast<const half8*>(a.X16 + rr * D + h * DH);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) q[s] = src[s];
        // The resident rows must be COMPLETE here, as far as hipcc's wait-count bookkeeping goes: left pending, their first
        // use sits inside the stream loop and hipcc guards every MFMA of every chunk with `s_waitcnt vmcnt(24 - s)` for them
        // -- counts that know nothing of the loop's hand-counted DMAs and stores and therefore drained the whole pipeline
        // (down to vmcnt(0)) once per 32-query chunk: an HBM round trip per 0.33 us of matrix work (round 2's 25 ms).
#pragma unroll
        for (int s = 0; s < STEPS; ++s) asm volatile("" : "+v"(q[s]));
    }
    const int pos = 32 * slice + r;                        // position inside the list = offset inside a query's segment
    const bool rvalid = active && r < nrow;
    const uint32_t base_lo = (uint32_t)a.keys_base;

    auto lds_addr = [](const void* p) -> uint32_t {
        return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
    };
    // metadata of chunk c (queries c*32 + r): lanes of the lower half fetch the row id, the upper half the destination
    auto issue_meta = [&](int c) {
        const int64_t e = e0 + min(32 * c + r, nq - 1);
        const void* g = h ? (const void*)(a.inv_dest + e) : (const void*)(a.inv_row + e);
        const uint32_t l = lds_addr(&meta[c & 7][0]);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l) : "memory", "m0");
    };
    // rows of a chunk served by this wave: 8w .. 8w + 7; their ids come from the metadata ring (landed a step ago), one
    // v_readlane each.  Chunks past the end re-load the last rows (their metadata is clamped); ids are clamped to valid rows.
    const uint32_t row_max = (uint32_t)(a.n_rows - 1);
    auto issue_rows = [&](int c, const unsigned char* buf) {
        const uint32_t lb = lds_addr(buf);
        const int32_t ids = meta[c & 7][r];
        uint32_t rows[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rows[i] = min((uint32_t)__builtin_amdgcn_readlane(ids, 8 * w + i), row_max);
#pragma unroll
        for (int part = 0; part < RD; ++part) {
            if (lane + 64 * part < D / 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const void* g = reinterpret_cast<const uint4*>(a.X16 + (int64_t)rows[i] * D) + lane + 64 * part;
                    const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane(
                        (int)(lb + (uint32_t)(8 * w + i) * (uint32_t)RS + (uint32_t)(1024 * part)));
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
                }
            }
        }
    };
    // The keys of the last finished chunk wait here, two per register (v_cvt_pknorm_u16_f32: round(clamp(v, 0, 1) * 65535),
    // within half a key step of v like key16 -- probed on the GPU, tools/probes/pknorm.hip), and leave during the NEXT
    // chunk's matrix work.  D[query][list row]: lane = list row (column), registers = 16 streamed queries; every store
    // instruction writes 32 consecutive keys of ONE query's segment for this list.
    // The epilogue used to be 250 VALU instructions per chunk and wave (float -> key conversions one value at a time, 64-bit
    // pointer arithmetic and pointer selects per store) against 25 MFMAs: the kernel was bound by issuing them.  Now: 8
    // conversions, and per store one shift-add (32-bit byte offset from the keys base in SGPRs) + one select.
    uint32_t pk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pk[j] = 0u;
    int prev_c = 0;                                        // chunk whose keys sit in `pk`
    const uint32_t k2 = 2u * ((uint32_t)pos - base_lo);    // byte offset of this lane's column inside a query's segment, less the base
    const uint32_t sink_off = 2u * ((uint32_t)(a.sink - a.keys) + (uint32_t)lane);
    // (the 16 destinations of a chunk are read from the ring as four 16-byte pieces in front of the matrix work: read one by
    // one between the stores -- which the stores' "memory" clobbers enforce -- every store waited for an LDS round trip with
    // lgkmcnt(0), draining the operand ring with it)
    int4 mdv[4];
    auto load_dest = [&]() {
        const int32_t* md = &meta[prev_c & 7][32 + 4 * h];
#pragma unroll
        for (int g = 0; g < 4; ++g) mdv[g] = *reinterpret_cast<const int4*>(md + 8 * g);
    };
    auto epilogue = [&]() {
        const int left = nq - 32 * prev_c - 4 * h;         // queries q0 < left of this chunk exist (all 32, except in a list's last chunk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q0 = (i & 3) + 8 * (i >> 2);
            const int4 m4 = mdv[i >> 2];
            const int32_t mdq = (i & 3) == 0 ? m4.x : (i & 3) == 1 ? m4.y : (i & 3) == 2 ? m4.z : m4.w;
            uint32_t off = ((uint32_t)mdq << 1) + k2;
            off = (rvalid && q0 < left) ? off : sink_off;
            if (i & 1) asm volatile("global_store_short_d16_hi %0, %1, %2" ::"v"(off), "v"(pk[i >> 1]), "s"(a.keys) : "memory");
            else asm volatile("global_store_short %0, %1, %2" ::"v"(off), "v"(pk[i >> 1]), "s"(a.keys) : "memory");
        }
    };
    auto compute = [&](const unsigned char* buf, int c) {
        constexpr int kMid = STEPS / 2;
        const unsigned char* sb = buf + r * RS + h * (DH * 2);
        half8 ring[NB];
        load_dest();
#pragma unroll
        for (int j = 0; j < NB; ++j) ring[j] = *reinterpret_cast<const half8*>(sb + j * 16);
        __builtin_amdgcn_sched_group_barrier(0x100, NB + 4, 0);  // the destinations and the whole ring in front of the first MFMA (fused.hip)
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const half8 ch = ring[s % NB];
            if (s + NB < STEPS) ring[s % NB] = *reinterpret_cast<const half8*>(sb + (s + NB) * 16);
            // streamed queries are the A operand, the resident list rows B: D[query][list row]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, q[s], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (s == kMid) epilogue();
        }
        asm volatile("s_nop 15" : "+a"(acc));              // MFMA -> accumulator read behind a taken branch (simtile.h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const us2 two = __builtin_amdgcn_cvt_pknorm_u16(acc[2 * j], acc[2 * j + 1]);
            pk[j] = (uint32_t)two.x | ((uint32_t)two.y << 16);
        }
        prev_c = c;
    };
    // step c: chunk c (in CUR) is computed; the row DMA of chunk c + 2 goes to FILL (free since the barrier: its last reader
    // was the computation of chunk c - 1); the metadata DMA of chunk c + 5 goes to the ring LAST.  Issue order per step and
    // wave: kRowOps row DMAs, 16 key stores (inside the computation), 1 metadata DMA.  vmcnt counts in order, so the wait in
    // front of step c + 1 -- which needs the rows of chunk c + 1, issued first thing in step c - 1 -- may leave everything
    // issued after them outstanding: (16 + 1) + (kRowOps + 16 + 1) operations.  That keeps the metadata of chunk c + 5 (read
    // by the row DMAs of step c + 3: two whole steps later, 2 (kRowOps + 17) > the allowance) and the key stores off the
    // critical path.  (Round 2 issued the metadata FIRST and waited with vmcnt(kRowOps + 16): every step then waited for a
    // metadata fetch issued one step earlier -- an HBM round trip per 0.33 us of matrix work; it, not the gather or the
    // matrix pipe, set the kernel's time: profiles/NOTES.md.)  The first two steps have less in front of them and wait
    // strictly.
    constexpr int kAllow = (16 + 1) + (kRowOps + 16 + 1), kAllowIdle = 1 + (kRowOps + 1);
#define FAL_STEP(CUR, FILL, C, STRICT)                                                                     \
    {                                                                                                      \
        /* wait and barrier in ONE asm per arm: no path reaches a barrier without its wait (tests/isa_lint.py) */ \
        if (STRICT) {                                                                                      \
            if (active) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kRowOps + 16 + 1) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kRowOps + 1) : "memory");            \
        } else {                                                                                           \
            if (active) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kAllow) : "memory");          \
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kAllowIdle) : "memory");             \
        }                                                                                                  \
        issue_rows((C) + 2, FILL);                                                                         \
        if (active) compute(CUR, C);                                                                       \
        issue_meta((C) + 5);                                                                               \
    }
    issue_meta(0);
    issue_meta(1);
    issue_meta(2);
    issue_meta(3);
    issue_meta(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_rows(0, sbuf0);
    issue_rows(1, sbuf1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRowOps) : "memory");       // chunk 0 has landed (the first step's own wait is a no-op)
    const int n_chunks = (nq + 31) >> 5;
    // (strict waits: in front of step 1 only the operations of step 0 lie behind the rows of chunk 1, in front of step 2 those
    // of steps 0 and 1 behind the rows of chunk 2: the steady-state allowance starts with step 2)
    FAL_STEP(sbuf0, sbuf2, 0, true)
    if (1 < n_chunks) {
        FAL_STEP(sbuf1, sbuf0, 1, true)
        if (2 < n_chunks) {
            FAL_STEP(sbuf2, sbuf1, 2, false)
            for (int c = 3; c < n_chunks; c += 3) {
                FAL_STEP(sbuf0, sbuf2, c, false)
                if (c + 1 >= n_chunks) break;
                FAL_STEP(sbuf1, sbuf0, c + 1, false)
                if (c + 2 >= n_chunks) break;
                FAL_STEP(sbuf2, sbuf1, c + 2, false)
            }
        }
    }
#undef FAL_STEP
    // (outstanding DMAs target this workgroup's LDS: the hardware holds the allocation until they retire)
    if (active) {
        load_dest();
        epilogue();
    }
}

// ------------------------------------------------------------------------------------------------------------
// select16_kernel: one wave per query (4 per workgroup)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wcount(bool p) { return __popcll(__ballot(p)); }

constexpr int kSelect16Copies = 1;       // copies of the high-byte histogram (select16_body)
// A value loaded early and needed late: the empty asm is its first use, so hipcc's wait for the load sits HERE and not where its
// scheduler would have hoisted the first arithmetic on it (right behind the load, in front of every later load).
__device__ __forceinline__ int late_use(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// One query: its keys stay PACKED, two 16-bit keys per register (R keys = R / 2 registers per lane), and are counted with the
// packed 16-bit VALU operations; slots outside the query's stream hold key 0 ("pads": they sit below every rank that is asked
// for -- nc > k -- and are taken out of the one count they can enter).  The first version unpacked every key into its own
// register and tested its position per key: 44 VALU instructions per key and the kernel was VALU-bound.
//  1. T = the k-th largest key (two 256-bin histogram levels in LDS);  e = bound of |key / 65535 - exact| around T, in key units
//  2. thresholds for the exact tail: [L, U] holds the exact k-th best similarity; n_hi = keys certainly above U
//  3. "members": the keys within 2e of T, as (value, stream position) -- resolve_kernel turns the positions into rows for the
//     few queries that turn out to have an ambiguous window candidate (no probe table / list offsets / perm gathers here)
template <int R>
__device__ __forceinline__ int2 select16_body(const Select16Args& a, const uint16_t* __restrict__ row, int nc, int k, int lane,
                                              int out_row_loaded, uint32_t* hist) {
    constexpr int P = R / 8, W = R / 2;
    // Keys in 16-byte pieces (8 per load instruction and lane): the stream is read from the 16-byte boundary in front of it,
    // `lead` keys early; key t of piece (j, lane) is stream position 8 (64 j + lane) + t - lead; word 4 j + m holds t = 2 m
    // (low half) and 2 m + 1.
    nc = __builtin_amdgcn_readfirstlane(nc);                                // (one query per wave: the masks below are scalar work)
    const int lead = __builtin_amdgcn_readfirstlane((int)((reinterpret_cast<uintptr_t>(row) & 15) >> 1));
    const int span = nc + lead;
    uint32_t w[W];
    {
        const uint4* base = reinterpret_cast<const uint4*>(row - lead);
        uint4 raw[P];
#pragma unroll
        for (int j = 0; j < P; ++j) raw[j] = base[64 * j + lane];           // unclamped: the buffer has kSimsSlack floats of slack
#pragma unroll
        for (int j = 0; j < P; ++j) { w[4 * j] = raw[j].x; w[4 * j + 1] = raw[j].y; w[4 * j + 2] = raw[j].z; w[4 * j + 3] = raw[j].w; }
        // pads -> 0.  Behind the stream: piece p = 64 j + lane is whole for p < span / 8, holds span % 8 keys for p == span / 8
        // (both wave-uniform per group), nothing behind; in front: the first `lead` keys of piece 0.
        const int q_all = span >> 3, r = span & 7;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int q = q_all - 64 * j;
            if (q < 64) {                                                    // wave-uniform (one group per query, two with R = 64)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t part = (2 * m < r ? 0xFFFFu : 0u) | (2 * m + 1 < r ? 0xFFFF0000u : 0u);
                    w[4 * j + m] &= lane < q ? 0xFFFFFFFFu : (lane == q ? part : 0u);
                }
            }
        }
        if (lead) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t keep = (2 * m >= lead ? 0xFFFFu : 0u) | (2 * m + 1 >= lead ? 0xFFFF0000u : 0u);
                w[m] &= lane == 0 ? keep : 0xFFFFFFFFu;
            }
        }
    }
    const int n_pads = 64 * R - nc;
    auto pos_of = [&](int i) -> int { return 8 * (64 * (i >> 3) + lane) + (i & 7) - lead; };
    // T = the k-th largest key (nc > k).  Two histogram levels in LDS -- 256 bins of the key's high byte, then the low byte
    // inside the bin that holds the k-th key; after the atomics: bins 4 lane .. 4 lane + 3 per lane, suffix sums over the lanes.
    uint32_t T;
    {
        // C copies of every bin (copy = lane % C, a bin's copies side by side): the keys of a query crowd into a few bins of
        // the high byte (most similarities are small), and LDS atomics of one instruction to ONE address are taken one after
        // the other -- with a single copy they were all of the kernel's time.
        auto clear = [&](auto copies) {
            constexpr int C = decltype(copies)::value;
#pragma unroll
            for (int q = 0; q < C; ++q) *reinterpret_cast<uint4*>(&hist[4 * (C * lane + q)]) = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_sync();
        };
        auto find = [&](auto copies, int kk, int* above) -> int {           // -> bin of the kk-th largest, *above = keys in higher bins
            constexpr int C = decltype(copies)::value;
            wave_lds_sync();
            int c[4];                                                         // bins 4 lane .. 4 lane + 3
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = 0;
#pragma unroll
                for (int q = 0; q < C; q += (C >= 4 ? 4 : C)) {
                    if (C >= 4) {
                        const uint4 v = *reinterpret_cast<const uint4*>(&hist[(4 * lane + j) * C + q]);
                        c[j] += (int)(v.x + v.y + v.z + v.w);
                    } else if (C == 2) {
                        const uint2 v = *reinterpret_cast<const uint2*>(&hist[(4 * lane + j) * C + q]);
                        c[j] += (int)(v.x + v.y);
                    } else c[j] += (int)hist[4 * lane + j];
                }
            }
            const int own = c[0] + c[1] + c[2] + c[3];
            const int pre = wave_prefix_sum(own);
            const int suf = __builtin_amdgcn_readlane(pre, 63) - pre + own;   // keys in this lane's bins and all higher ones
            const unsigned long long reach = __ballot(suf >= kk);             // (a prefix of the lanes: suf falls with the lane)
            const int L = 63 - __clzll(reach);
            int acc = suf - own, bin = 4 * lane;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (acc + c[j] >= kk) { bin = 4 * lane + j; break; }
                acc += c[j];
            }
            *above = __builtin_amdgcn_readlane(acc, L);
            const int res = __builtin_amdgcn_readlane(bin, L);
            wave_lds_sync();
            return res;
        };
        constexpr int C1 = kSelect16Copies;
        const std::integral_constant<int, C1> many{};
        const std::integral_constant<int, 1> one{};
        int above1 = 0, above2 = 0;
        clear(many);
        // A lower bound of T first, so that the crowd of small keys stays out of the histogram: every lane's second largest
        // key has two keys of that lane at or above it, so the smallest of the 64 has 128 >= k (packed running top-2 per
        // 16-bit half, then across the halves and, on the DPP network, across the lanes).
        uint32_t LB = 0u;
        if (k <= 128) {
            uint32_t m1 = 0u, m2 = 0u;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint32_t t = pk_min(m1, w[i]);
                m1 = pk_max(m1, w[i]);
                m2 = pk_max(m2, t);
            }
            const uint32_t a1 = m1 & 0xFFFFu, b1h = m1 >> 16, a2 = m2 & 0xFFFFu, b2 = m2 >> 16;
            LB = (uint32_t)wave_min((int)max(min(a1, b1h), max(a2, b2)));
        }
        uint32_t* mine = hist + (lane & (C1 - 1));
#pragma unroll
        for (int i = 0; i < W; ++i) {
            if ((w[i] & 0xFFFFu) >= LB)
                __hip_atomic_fetch_add(&mine[((w[i] >> 8) & 0xFFu) * C1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((w[i] >> 16) >= LB)
                __hip_atomic_fetch_add(&mine[(w[i] >> 24) * C1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const int b1 = find(many, k, &above1);
        clear(one);
        const uint32_t pat = (uint32_t)b1 * 0x01000100u;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const uint32_t x = w[i] ^ pat;                                    // a key of bin b1: its high byte is 0 now
            const bool lo = (x & 0xFF00u) == 0u, hi = x < 0x01000000u;
            if (__ballot(lo || hi)) {                                         // wave-uniform; the bin holds a few keys of the query
                if (lo) __hip_atomic_fetch_add(&hist[x & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (hi) __hip_atomic_fetch_add(&hist[(x >> 16) & 0xFFu], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        const int b2 = find(one, k - above1, &above2);
        T = (uint32_t)((b1 << 8) + b2);                                      // (as a key; the hand-off counts in key + 1)
    }
    const float Tv = (float)T * (1.f / 65535.f);
    const float e = 1.3e-3f * Tv + 1.2e-5f;
    const int delta_e = (int)ceilf(e * 65535.f) + 1;
    const int delta = 2 * delta_e + 2;
    // n_hi = keys > T + delta; members = keys in [T - delta, T + delta]: per word  min(sat(x - hi), 1)  and, with y = x - lo
    // (wrapping: keys below lo turn large),  f = min(sat(y - (hi - lo)), 1) = 1 for keys OUTSIDE the range
    const uint32_t c_hi = (uint32_t)min((int)T + delta, 65535) * 0x00010001u;
    const int lo_key = max((int)T - delta, 0);
    const uint32_t c_lo = (uint32_t)lo_key * 0x00010001u;
    const uint32_t c_w = c_hi - c_lo;                                        // (no borrow between the halves: hi >= lo)
    uint32_t f[W], acc_hi = 0u, acc_out = 0u;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        acc_hi = pk_add(acc_hi, pk_min(pk_sub_sat(w[i], c_hi), 0x00010001u));
        f[i] = pk_min(pk_sub_sat(pk_sub(w[i], c_lo), c_w), 0x00010001u);
        acc_out = pk_add(acc_out, f[i]);
    }
    int n_hi, n_mem;
    {
        const int both = (int)((acc_hi & 0xFFFFu) + (acc_hi >> 16)) | ((int)((acc_out & 0xFFFFu) + (acc_out >> 16)) << 16);
        const int tot = __builtin_amdgcn_readlane(wave_prefix_sum(both), 63);      // (each sum <= 64 R <= 4,096)
        n_hi = tot & 0xFFFF;
        n_mem = 64 * R - (tot >> 16) - (lo_key == 0 ? n_pads : 0);          // pads (key 0) are inside the range iff it reaches 0
    }
    const int64_t out_row = late_use(out_row_loaded);
    QThr t;
    t.T = Tv;
    t.L = ((float)T - (float)delta_e) * (1.f / 65535.f);
    t.U = ((float)T + (float)delta_e) * (1.f / 65535.f);
    t.eps = 0.5f * ((float)delta + 0.5f) * (1.f / 65535.f);     // resolve_kernel: members are |v - T| <= 2 eps
    t.bstar = 1 << 30;                                           // (no histogram bins here: nothing "above the bin")
    t.nabove = n_hi;
    t.flags = 0;
    if (n_mem > FAL_FUSED_MEM) {
        t.flags = 2;                                             // e.g. hundreds of identical spectra: exact fallback
        n_mem = 0;
    }
    t.mc = min(n_mem, FAL_FUSED_MEM / 2) | (max(n_mem - FAL_FUSED_MEM / 2, 0) << 16);
    const int2 sel = make_int2((int)T + 1 - delta, (int)T + 1 + delta);      // (kept16: what a window candidate's key must reach | is ambiguous up to)
    if (lane == 0) {
        a.thr[out_row] = t;
        if (a.gsel) a.gsel[out_row] = sel;
    }
    if (n_mem == 0) return sel;
    float* gv = a.gmem_v + out_row * FAL_FUSED_MEM;
    uint32_t* gi = a.gmem_id + out_row * FAL_FUSED_MEM;
    int base = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        if (__ballot(f[i] != 0x00010001u)) {                     // wave-uniform: some lane holds a member in this word
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int slot_i = 8 * (i >> 2) + 2 * (i & 3) + h, pos = pos_of(slot_i);
                const bool in = ((f[i] >> (16 * h)) & 0xFFFFu) == 0u && pos >= 0 && pos < nc;
                const unsigned long long mask = __ballot(in);
                if (in) {
                    const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                    gv[slot] = (float)((w[i] >> (16 * h)) & 0xFFFFu) * (1.f / 65535.f);
                    gi[slot] = (uint32_t)pos;                    // stream position (resolve_kernel: -> row)
                }
                base += __popcll(mask);
            }
        }
    }
    return sel;
}

__device__ __forceinline__ int2 select16_trivial(const Select16Args& a, int out_row_loaded, int flags, int lane) {
    const int64_t out_row = late_use(out_row_loaded);
    if (lane == 0) {
        QThr t0{};
        t0.L = t0.U = -INFINITY;
        t0.bstar = 1 << 30;
        t0.flags = flags;
        a.thr[out_row] = t0;
        if (a.gsel) a.gsel[out_row] = make_int2(INT32_MIN, INT32_MIN);      // every window candidate stays, none is ambiguous
    }
    return make_int2(INT32_MIN, INT32_MIN);
}

// BIG = false: one wave per query slot of the launch's tiles; queries with more keys than 32 per lane are appended to
// a.big_list.  BIG = true: the listed queries, 64 keys per lane (more than 4,096 keys: exact fallback)
// WIDE: the first pass also holds 40 keys per lane (a search whose queries have up to 2,560 keys: BASELINE configs[3]'s n_probe
// = 32 over ~70-row lists -- most of them would otherwise take the second pass at one wave per SIMD); the narrow form keeps
// the common n_probe = 16 case at its register count
template <bool BIG, bool WIDE>
__global__ __launch_bounds__(256) void select16_kernel(Select16Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[4][256 * kSelect16Copies];      // a wave's key histogram (select16_body)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* hist = hist_all[w];
    const int n_big = BIG ? min(*a.big_count, a.big_cap) : 0;
    for (int64_t item = (int64_t)blockIdx.x * 4 + w;; item += (int64_t)gridDim.x * 4) {
        int64_t slot;
        if (BIG) {
            if (item >= n_big) return;
            slot = a.big_list[item];
        } else {
            slot = 32 * a.tile_begin + item;
            if (item >= 32 * a.n_tiles) return;
        }
        // Two round trips in front of the keys, not five: the offsets and the tile's first list-order position (one table entry,
        // tile_job16_kernel) travel together, then perm[p] next to the key loads (it is needed for the stores only).
        const int64_t t = slot >> 5;
        const int64_t o0 = a.q_sim_off[slot], o1 = a.q_sim_off[slot + 1];
        const int32_t p0 = late_use(a.tile_p0[t - a.tile_begin]);            // (used here: hipcc would sink the load behind the wait for o0)
        const int nc = (int)(o1 - o0);
        if (nc > 0) {                                            // (0: padding slot of a bucket's last tile)
            const int out_row = a.perm[(int64_t)p0 + (slot & 31)];           // query position in list order -> sorted row; used late
            const uint16_t* row = a.keys + (o0 - a.keys_base);
            const int k = a.k;
            const int span = nc + (int)((reinterpret_cast<uintptr_t>(row) & 15) >> 1);      // keys from the 16-byte boundary in front
            int2 sel = make_int2(INT32_MAX, INT32_MIN);
            if (nc <= k) sel = select16_trivial(a, out_row, 0, lane);  // every candidate is among the k best
            else if (!BIG) {
                if (span <= 512) sel = select16_body<8>(a, row, nc, k, lane, out_row, hist);
                else if (span <= 1024) sel = select16_body<16>(a, row, nc, k, lane, out_row, hist);
                else if (span <= 1536) sel = select16_body<24>(a, row, nc, k, lane, out_row, hist);
                else if (span <= 2048) sel = select16_body<32>(a, row, nc, k, lane, out_row, hist);
                else if (WIDE && span <= 2560) sel = select16_body<40>(a, row, nc, k, lane, out_row, hist);
                else if (lane == 0) {
                    const int at = atomicAdd(a.big_count, 1);
                    if (at < a.big_cap) a.big_list[at] = (int32_t)slot;
                }
            } else {
                if (span <= 4096) select16_body<64>(a, row, nc, k, lane, out_row, hist);
                else select16_trivial(a, out_row, 2, lane);      // more keys than the registers hold: exact fallback
            }
            (void)sel;
        }
        if (!BIG) return;
    }
}

// select16_kernel<false, false> + kept16_kernel in one (round 6, `FALCON_KEPT16=fused`): a wave takes FOUR consecutive query
// slots of a tile, brackets their k-th keys one after the other (select16_body), then runs kept16_query once with its four
// 16-lane groups on the four queries -- the thresholds stay in registers, and the window gathers of kept16 (latency) sit in the
// same kernel as the selections (VALU).  Single-pass searches only (no query above 2,048 keys).
__global__ __launch_bounds__(256) void select16k_kernel(Select16Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[4][256 * kSelect16Copies];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4;
    uint32_t* hist = hist_all[w];
    const int64_t g = (int64_t)blockIdx.x * 4 + w;              // group of four slots
    if (g >= 8 * a.n_tiles) return;
    const int64_t slot0 = 32 * a.tile_begin + 4 * g;
    const int64_t t = slot0 >> 5;
    const int32_t p0 = a.tile_p0[t - a.tile_begin];
    const int64_t l0 = a.tile_l0[t - a.tile_begin];
    int64_t off[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) off[i] = a.q_sim_off[slot0 + i];
    bool live_g = false;
    int64_t p_g = 0, row_g = 0;
    const uint16_t* krow_g = a.keys;
    int2 sel_g = make_int2(INT32_MAX, INT32_MIN);
#pragma unroll 1
    for (int qi = 0; qi < 4; ++qi) {
        const int64_t slot = slot0 + qi;
        const int64_t o0 = qi == 0 ? off[0] : qi == 1 ? off[1] : qi == 2 ? off[2] : off[3];
        const int64_t o1 = qi == 0 ? off[1] : qi == 1 ? off[2] : qi == 2 ? off[3] : off[4];
        const int nc = (int)(o1 - o0);
        if (nc <= 0) continue;                                   // (padding slot of a bucket's last tile)
        const int out_row = a.perm[(int64_t)p0 + (slot & 31)];
        const uint16_t* row = a.keys + (o0 - a.keys_base);
        const int k = a.k;
        const int span = nc + (int)((reinterpret_cast<uintptr_t>(row) & 15) >> 1);
        int2 sel;
        if (nc <= k) sel = select16_trivial(a, out_row, 0, lane);
        else if (span <= 512) sel = select16_body<8>(a, row, nc, k, lane, out_row, hist);
        else if (span <= 1024) sel = select16_body<16>(a, row, nc, k, lane, out_row, hist);
        else if (span <= 1536) sel = select16_body<24>(a, row, nc, k, lane, out_row, hist);
        else sel = select16_body<32>(a, row, nc, k, lane, out_row, hist);        // (span <= 2,048: the launcher's condition)
        if (grp == qi) {
            live_g = true;
            p_g = (int64_t)p0 + (slot & 31);
            row_g = out_row;
            krow_g = row;
            sel_g = sel;
        }
    }
    kept16_query(a.kept, live_g, p_g, row_g, l0, krow_g, sel_g, lane);
}

__global__ void tile_job16_kernel(const DenseJob* __restrict__ jobs, int n_jobs, int64_t tile_begin, int64_t n_tiles,
                                  int32_t* __restrict__ tile_job, int32_t* __restrict__ tile_p0, int64_t* __restrict__ tile_l0) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_tiles) {
        const int j = find_job(jobs, n_jobs, tile_begin + i);
        tile_job[i] = j;
        tile_p0[i] = (int32_t)(jobs[j].q_row0 + 32 * (tile_begin + i - jobs[j].tile0));      // list-order position of the tile's query 0
        tile_l0[i] = jobs[j].c_row0;                                                        // global id of its bucket's list 0
    }
}

// pos_of_row[perm[p]] = p
__global__ void pos_of_row_kernel(const int32_t* __restrict__ perm, int64_t n, int32_t* __restrict__ pos_of_row) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        pos_of_row[perm[p]] = (int32_t)p;
}

bool ivf16_supports(int d) { return d == 64 || d == 128 || d == 256 || d == 400 || d == 800; }

int launch_pos_of_row(fal_ctx* ctx, const int32_t* perm, int64_t n, int32_t* pos_of_row) {
    if (n <= 0) return FAL_OK;
    StageScope ts(ctx, ST_BUILD);
    hipLaunchKernelGGL(pos_of_row_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), ctx->num_cus * 16)), dim3(256), 0,
                       ctx->stream, perm, n, pos_of_row);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

// FALCON_LIST16 = "r": list16r_kernel (list16r.hip: the list's rows in LDS, every wave its own query chunks; low_dim <= 400)
// instead of list16_kernel (rows in registers, shared query ring, four waves in lockstep).  Same keys, bit for bit
// (tests/test_gpu_ivf16.py); measured SLOWER (profiles/NOTES.md r5: both forms issue ~900 instructions per 32-query chunk
// for its 65 MFMAs, and the lockstep form runs them on two waves per SIMD), so the lockstep form stays the default.  Read at
// every launch: tests and tools/list16_ab.py switch it inside one process.
// FALCON_LIST16 = "d": the dense-gather form even where the sparse records exist (list16s.hip is the default then: A/B switch)
static int list16_form() {
    const char* e = getenv("FALCON_LIST16");
    return (e && !strcmp(e, "r")) ? 1 : (e && !strcmp(e, "d")) ? 2 : 0;
}

int launch_list16(fal_ctx* ctx, const List16Args& a_in) {
    if (a_in.n_tiles_max <= 0) return FAL_OK;
    const int64_t per_xcd = (a_in.n_tiles_max + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
    dim3 grid((unsigned)(per_xcd * 8)), block(256);
    const List16Args& a = a_in;
    StageScope ts(ctx, ST_SCAN);
    StageScope tk(ctx, ST_KERNEL);
    if (a.d <= 400 && list16_form() == 1) return launch_list16r(ctx, a);
    if (a.sq16 != nullptr && a.d <= 400 && list16_form() == 0) return launch_list16s(ctx, a);      // (measured slower at 800 columns)
    switch (a.d / 16) {
        case 4: hipLaunchKernelGGL((list16_kernel<4>), grid, block, 0, ctx->stream, a); break;
        case 8: hipLaunchKernelGGL((list16_kernel<8>), grid, block, 0, ctx->stream, a); break;
        case 16: hipLaunchKernelGGL((list16_kernel<16>), grid, block, 0, ctx->stream, a); break;
        case 25: hipLaunchKernelGGL((list16_kernel<25>), grid, block, 0, ctx->stream, a); break;
        case 50: hipLaunchKernelGGL((list16_kernel<50>), grid, block, 0, ctx->stream, a); break;
        default:
            set_error("IVF prefilter: low_dim %d has no instantiation (64, 128, 256, 400)", a.d);
            return FAL_EUNSUPPORTED;
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

int launch_select16(fal_ctx* ctx, const Select16Args& a_in, int64_t n_tiles) {
    if (n_tiles <= 0) return FAL_OK;
    Select16Args a = a_in;
    a.n_tiles = n_tiles;
    // tile -> job table, then the list of queries with more than 2,048 keys (count in front)
    int32_t* tj = nullptr;
    ctx->release(SLOT_TILEJOB);        // a launcher-local table: the previous launcher's pointer is dead
    FAL_TRY(ctx->reserve(SLOT_TILEJOB, sizeof(int32_t) * (size_t)(std::max<int64_t>(n_tiles, 1 << 16) + 35 * n_tiles + 64 + 4), (void**)&tj));
    a.big_count = tj + std::max<int64_t>(n_tiles, 1 << 16);
    a.big_list = a.big_count + 16;
    int32_t* tp0 = a.big_list + 32 * n_tiles;
    int64_t* tl0 = reinterpret_cast<int64_t*>(tj + ((std::max<int64_t>(n_tiles, 1 << 16) + 33 * n_tiles + 16 + 1) & ~(int64_t)1));      // (8-byte aligned)
    a.big_cap = (int)std::min<int64_t>(32 * n_tiles, INT32_MAX);
    FAL_REQUIRE(n_tiles * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many queries in one select launch");
    StageScope ts(ctx, ST_SELECT);
    FAL_CHECK_HIP(hipMemsetAsync(a.big_count, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(tile_job16_kernel, dim3((unsigned)ceil_div(n_tiles, 256)), dim3(256), 0, ctx->stream, a.jobs, a.n_jobs,
                       a.tile_begin, n_tiles, tj, tp0, tl0);
    a.tile_job = tj;
    a.tile_p0 = tp0;
    a.tile_l0 = tl0;
    // (a query's keys are read from the 16-byte boundary in front of its stream: up to 7 keys more than it has)
    const bool two_pass = a.max_keys + 7 > 2048;
    if (two_pass) a.fuse_kept = 0;                        // (kept16_kernel follows the second pass: launch_kept16)
    a.kept.tile_job = tj;
    if (two_pass) hipLaunchKernelGGL((select16_kernel<false, true>), dim3((unsigned)(n_tiles * 8)), dim3(256), 0, ctx->stream, a);
    else if (a.fuse_kept) hipLaunchKernelGGL(select16k_kernel, dim3((unsigned)(n_tiles * 2)), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((select16_kernel<false, false>), dim3((unsigned)(n_tiles * 8)), dim3(256), 0, ctx->stream, a);
    if (two_pass)
        hipLaunchKernelGGL((select16_kernel<true, false>), dim3((unsigned)(ctx->num_cus * 4)), dim3(256), 0, ctx->stream, a);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::list16_kernel<25>);      // (fal_ctx_plan: this unit's code object is loaded up front)
