// list16s_kernel: list16_kernel (ivf16.hip) with the probing queries gathered in SPARSE form.
//
// list16_kernel fetches every query that probes a list as its dense float16 row: 2 d bytes (800 B at low_dim 400, 1,600 at
// 800) per (query, list) pair although a hashed spectrum holds at most 50-64 non-zero components -- 128 GB of logical gathers per
// 10 M pass, and the kernel runs at the rate a CU gathers rows from L2 / Infinity Cache (28 GB/s per CU; MI355X_MICROARCH.md
// "Indexed rows: gather into LDS": 33 from the Infinity Cache, 66-73 from L2): what bounds it is gathered BYTES per pair, not
// the schedule (profiles/NOTES.md r3-r5: occupancy, residency, instruction count all measured and ruled out).
//
// Here a query arrives as the 256-byte record the index build leaves per row (fal_ivf::sq16: 64 x u16 column | 64 x f16 value,
// unused entries = column 0xFFFF): FOUR rows per `global_load_lds_dwordx4` (16 lanes x 16 B each), 2 row DMAs per step and wave
// instead of 8-16, 3.1x / 6.3x fewer gathered bytes.  The wave that fetched 8 rows of a chunk expands them itself into the
// chunk's dense operand tile in LDS (the tile list16_kernel's MFMAs read: row-major, stride 2 d + 16 B): it first writes zeros
// where it scattered two chunks ago (the offsets wait in 4 registers per tile), then one `ds_write_b16` per entry -- a lane owns
// 8 entries of one row.  Staging and expansion are wave-private (no barrier between DMA and expansion); the one barrier per
// step hands the finished operand tile (expanded during the previous step) to the four waves' MFMAs and frees the other tile.
//
//   step c:   wait (own DMA of chunk c + 1) + barrier
//             DMA   chunk c + 3 -> stage[c % 3]                  (2 instructions per wave)
//             expand chunk c + 1: stage[(c + 1) % 3] -> tile[(c + 1) & 1]
//             compute chunk c from tile[c & 1]   (STEPS MFMAs; the 16 key stores of chunk c - 1 in the middle)
//             metadata DMA of chunk c + 6
//
// Measured (tools/list16_ab.py, 2.5 M spectra at the 10 M job's density, alternating passes; profiles/NOTES.md r6): 5.85 ms against
// 6.05 for the dense gather (128 lists per bucket), 10.6 against 12.05 with 256 lists and n_probe 32 (buckets beyond an XCD's
// L2), but 12.2 against 11.0 at low_dim 800 (one workgroup per CU either way; the expansion is serial work of the one wave a
// SIMD has) -- so the dense gather keeps low_dim 800 (launch_list16).  Gathered bytes were NOT what bounds the kernel: with 4x
// fewer row DMAs the step is as long as before; knocking out key stores (wrong results, timing only) says the 16 two-byte key
// stores per step and wave are worth 20 % (8 stores: -12.5 %, 2 stores: -20 %) -- the next lever, and it needs 16-byte
// aligned (padded) key segments so that a store instruction can carry 8 keys per lane.
//
// The expanded tile holds exactly the float16 row (the record keeps every non-zero component; an index with a row of more than
// 64 non-zeros does not take this kernel: launch_list16), the MFMA chain and the key conversion are list16_kernel's: the keys are
// bit-identical (tests/test_gpu_ivf16.py, tests/test_gpu_regimes.py, tests/test_gpu_stress.py run both forms).
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "scan.h"
#include "ivf.h"
#include "ivf16.h"

namespace fal {

typedef _Float16 half8s __attribute__((ext_vector_type(8)));

// KN: timing experiments with WRONG results (FALCON_L16_KNOCK, tools/list16_ab.py): 1 = no key stores, 2 = four 8-byte stores per
// lane at addresses rounded down to 8 bytes (the store shape a padded key stream would allow, without the transposition),
// 3 = 2 + a wave-private LDS transposition in quarter rounds in front of them
template <int STEPS, int KN = 0>
__global__ __launch_bounds__(256, STEPS > 32 ? 1 : (KN == 7 ? 3 : 2)) void list16s_kernel(List16Args a) {
    constexpr int kStores = (KN == 0 || KN == 4 || KN == 5) ? 16 : (KN == 1 || KN == 6 || KN == 7) ? 0 : 4;      // (4: no expansion, 5: no record DMAs, 6: neither, no stores)
    constexpr int D = STEPS * 16, DH = D / 2;
    constexpr int kRowOps = (KN == 5 || KN == 6 || KN == 7) ? 0 : 2;  // row DMAs per step and wave (4 records of 256 B each)
    constexpr int NB = STEPS < 4 ? STEPS : 4;             // LDS operand reads in flight ahead of the MFMAs
    constexpr int RS = D * 2 + 16;                        // LDS row stride of the operand tiles in bytes (the last 16: dump slot)
    __shared__ __attribute__((aligned(16))) unsigned char tile0[32 * RS];
    // (KN == 7: the bare matrix loop of KN == 6 with ONE operand tile and no stages -- 28 KB of LDS: does a third workgroup per CU help it?)
    __shared__ __attribute__((aligned(16))) unsigned char tile1[KN == 7 ? 16 : 32 * RS];
    __shared__ __attribute__((aligned(16))) unsigned char stage0[KN == 7 ? 16 : 32 * 256];
    __shared__ __attribute__((aligned(16))) unsigned char stage1[KN == 7 ? 16 : 32 * 256];
    __shared__ __attribute__((aligned(16))) unsigned char stage2[KN == 7 ? 16 : 32 * 256];
    __shared__ int32_t meta[8][64];                       // per chunk: [0, 32) sorted row of query r, [32, 64) its destination (low dword)
    __shared__ __attribute__((aligned(16))) unsigned char xpose[KN == 3 ? 4 * 576 : 16];
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
    const int nrow = min(32, l_rows - 32 * slice);        // rows of this slice (<= 0: the wave only fetches and expands)
    const int64_t e0 = a.inv_off[L];
    const int nq = (int)(a.inv_off[L + 1] - e0);          // queries probing the list
    if (nq <= 0 || l_rows <= 0) return;
    const bool active = nrow > 0;

    half8s q[STEPS];                                       // the resident operand: list row 32*slice + r, k-half h (dense float16 row)
    {
        const int64_t rr = a.perm[l_row0 + min(32 * slice + min(r, max(nrow, 1) - 1), l_rows - 1)];
        const half8s* src = reinterpret_cast<const half8s*>(a.X16 + rr * D + h * DH);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) q[s] = src[s];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) asm volatile("" : "+v"(q[s]));      // (complete HERE: ivf16.hip list16_kernel)
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
    // the sparse records of this wave's 8 rows of chunk c (rows 8 w .. 8 w + 7) into ITS part of a stage: two instructions of
    // four records each -- lane l fetches piece (l & 15) of the record of row 8 w + 4 i + (l >> 4).  Chunks past the end re-load
    // the last rows (their metadata is clamped); ids are clamped to valid rows.
    const uint32_t row_max = (uint32_t)(a.n_rows - 1);
    auto issue_rows = [&](int c, const unsigned char* st) {
        const uint32_t lb = lds_addr(st) + (uint32_t)(8 * w) * 256u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t row = min((uint32_t)meta[c & 7][8 * w + 4 * i + (lane >> 4)], row_max);
            const void* g = a.sq16 + (int64_t)row * 128 + 8 * (lane & 15);
            const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lb + 1024u * (uint32_t)i));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
        }
    };
    // expansion of this wave's 8 rows of a chunk: stage -> operand tile.  Lane l owns entries 8 (l & 7) .. + 7 of row 8 w + (l >> 3).
    // `prev`: byte offsets (inside the wave's 8 rows of this tile) of what the lane scattered here two chunks ago, two per register.
    const int xrow = lane >> 3, xpiece = lane & 7;
    auto expand = [&](const unsigned char* st, unsigned char* tile, uint32_t (&prev)[4]) {
        unsigned char* reg = tile + (8 * w) * RS;
        const unsigned char* rec = st + (8 * w + xrow) * 256 + xpiece * 16;
        const uint4 cols = *reinterpret_cast<const uint4*>(rec);
        const uint4 vals = *reinterpret_cast<const uint4*>(rec + 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint16_t*>(reg + (prev[j] & 0xFFFFu)) = 0;
            *reinterpret_cast<uint16_t*>(reg + (prev[j] >> 16)) = 0;
        }
        const uint32_t cw[4] = {cols.x, cols.y, cols.z, cols.w}, vw[4] = {vals.x, vals.y, vals.z, vals.w};
        const uint32_t rbase = (uint32_t)xrow * (uint32_t)RS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t o0 = rbase + 2u * min(cw[j] & 0xFFFFu, (uint32_t)D);      // unused entries (0xFFFF) -> the row's dump slot
            const uint32_t o1 = rbase + 2u * min(cw[j] >> 16, (uint32_t)D);
            *reinterpret_cast<uint16_t*>(reg + o0) = (uint16_t)(vw[j] & 0xFFFFu);
            *reinterpret_cast<uint16_t*>(reg + o1) = (uint16_t)(vw[j] >> 16);
            prev[j] = o0 | (o1 << 16);
        }
    };
    uint32_t pk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pk[j] = 0u;
    int prev_c = 0;                                        // chunk whose keys sit in `pk`
    const uint32_t k2 = 2u * ((uint32_t)pos - base_lo);    // byte offset of this lane's column inside a query's segment, less the base
    const uint32_t sink_off = 2u * ((uint32_t)(a.sink - a.keys) + (uint32_t)lane);
    int4 mdv[4];
    auto load_dest = [&]() {
        const int32_t* md = &meta[prev_c & 7][32 + 4 * h];
#pragma unroll
        for (int g = 0; g < 4; ++g) mdv[g] = *reinterpret_cast<const int4*>(md + 8 * g);
    };
    auto epilogue = [&]() {
        const int left = nq - 32 * prev_c - 4 * h;         // queries q0 < left of this chunk exist (all 32, except in a list's last chunk)
        if (KN == 1 || KN == 6 || KN == 7) return;
        if (KN >= 2) {
            const int32_t* md = &meta[prev_c & 7][32];
            unsigned char* xp = xpose + (KN == 3 ? 576 * w : 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi = 8 * g + (lane >> 3);
                uint32_t off = ((((uint32_t)md[qi] << 1) + 2u * ((uint32_t)(32 * slice) - base_lo)) & ~7u) + 8u * (uint32_t)(lane & 7);
                off = (active && 4 * (lane & 7) < nrow && qi < nq - 32 * prev_c) ? off : (sink_off & ~7u);
                uint2 v = make_uint2(pk[2 * g], pk[2 * g + 1]);
                if (KN == 3) {
                    // D[query (i & 3) + 4 h of this round][list row r] -> [8 queries][32 rows + pad] of 2 bytes
                    unsigned char* wp = xp + (4 * h) * 72 + 2 * r;
                    *reinterpret_cast<uint16_t*>(wp) = (uint16_t)(v.x & 0xFFFFu);
                    *reinterpret_cast<uint16_t*>(wp + 72) = (uint16_t)(v.x >> 16);
                    *reinterpret_cast<uint16_t*>(wp + 144) = (uint16_t)(v.y & 0xFFFFu);
                    *reinterpret_cast<uint16_t*>(wp + 216) = (uint16_t)(v.y >> 16);
                    v = *reinterpret_cast<const uint2*>(xp + (lane >> 3) * 72 + 8 * (lane & 7));
                }
                asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(a.keys) : "memory");
            }
            return;
        }
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
    uint32_t prev0[4], prev1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) prev0[j] = prev1[j] = (uint32_t)(2 * D) | ((uint32_t)(2 * D) << 16);     // (the dump slot of row 0)
    auto compute = [&](const unsigned char* buf, int c) {
        constexpr int kMid = STEPS / 2;
        const unsigned char* sb = buf + r * RS + h * (DH * 2);
        half8s ring[NB];
        load_dest();
#pragma unroll
        for (int j = 0; j < NB; ++j) ring[j] = *reinterpret_cast<const half8s*>(sb + j * 16);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const half8s ch = ring[s % NB];
            if (s + NB < STEPS) ring[s % NB] = *reinterpret_cast<const half8s*>(sb + (s + NB) * 16);
            // streamed queries are the A operand, the resident list rows B: D[query][list row]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, q[s], acc, 0, 0, 0);
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
    // VM operations per step and wave, in issue order: kRowOps row DMAs, 16 key stores (active waves), 1 metadata DMA.  The wait in
    // front of step c needs the wave's OWN records of chunk c + 1 (issued first thing in step c - 2): everything issued after them
    // may stay in flight -- (16 + 1) of step c - 2 and (kRowOps + 16 + 1) of step c - 1.  The metadata read by step c's row DMAs
    // (chunk c + 3: issued last in step c - 3) is older than that.  Steps 0 and 1 have less behind them.
    constexpr int kAllow = (kStores + 1) + (kRowOps + kStores + 1), kAllowIdle = 1 + (kRowOps + 1);
    constexpr int kFirst = kRowOps + kStores + 1, kFirstIdle = kRowOps + 1;
    // WAIT: 0 = step 0 (records of chunk 1: only chunk 2's DMAs behind them), 1 = step 1, 2 = steady state
#define FAL_STEP_S(C, STG_FILL, STG_NEXT, TILE_CUR, TILE_NEXT, PREV_NEXT, WAIT)                                \
    {                                                                                                          \
        /* wait and barrier in ONE asm per arm (tests/isa_lint.py); lgkmcnt(0): this wave's expansion has landed */ \
        if ((WAIT) == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kRowOps) : "memory");  \
        else if ((WAIT) == 1) {                                                                                \
            if (active) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kFirst) : "memory");   \
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kFirstIdle) : "memory");      \
        } else {                                                                                               \
            if (active) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kAllow) : "memory");   \
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kAllowIdle) : "memory");      \
        }                                                                                                      \
        if (kRowOps) issue_rows((C) + 3, STG_FILL);                                                            \
        if (KN != 4 && KN != 6 && KN != 7 && (C) + 1 < n_chunks) expand(STG_NEXT, TILE_NEXT, PREV_NEXT);       \
        if (active) compute(KN == 7 ? tile0 : TILE_CUR, C);                                                    \
        issue_meta((C) + 6);                                                                                   \
    }
    const int n_chunks = (nq + 31) >> 5;
    issue_meta(0);
    issue_meta(1);
    issue_meta(2);
    issue_meta(3);
    issue_meta(4);
    issue_meta(5);
    // both operand tiles start as zeros: every wave clears its own 8 rows of each
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int i = lane; i < 8 * RS / 16; i += 64) {
            *reinterpret_cast<uint4*>(tile0 + (8 * w) * RS + 16 * i) = z;
            if (KN != 7) *reinterpret_cast<uint4*>(tile1 + (8 * w) * RS + 16 * i) = z;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kRowOps) {
        issue_rows(0, stage0);
        issue_rows(1, stage1);
        issue_rows(2, stage2);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kRowOps) : "memory");   // the records of chunk 0 have landed
    if (KN != 7) expand(stage0, tile0, prev0);
    FAL_STEP_S(0, stage0, stage1, tile0, tile1, prev1, 0)
    if (1 < n_chunks) {
        FAL_STEP_S(1, stage1, stage2, tile1, tile0, prev0, 1)
        for (int c = 2; c < n_chunks; c += 6) {
            FAL_STEP_S(c, stage2, stage0, tile0, tile1, prev1, 2)
            if (c + 1 >= n_chunks) break;
            FAL_STEP_S(c + 1, stage0, stage1, tile1, tile0, prev0, 2)
            if (c + 2 >= n_chunks) break;
            FAL_STEP_S(c + 2, stage1, stage2, tile0, tile1, prev1, 2)
            if (c + 3 >= n_chunks) break;
            FAL_STEP_S(c + 3, stage2, stage0, tile1, tile0, prev0, 2)
            if (c + 4 >= n_chunks) break;
            FAL_STEP_S(c + 4, stage0, stage1, tile0, tile1, prev1, 2)
            if (c + 5 >= n_chunks) break;
            FAL_STEP_S(c + 5, stage1, stage2, tile1, tile0, prev0, 2)
        }
    }
#undef FAL_STEP_S
    // (outstanding DMAs target this workgroup's LDS: the hardware holds the allocation until they retire)
    if (active) {
        load_dest();
        epilogue();
    }
}

int launch_list16s(fal_ctx* ctx, const List16Args& a) {
    const int64_t per_xcd = (a.n_tiles_max + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8)), block(256);
    switch (a.d / 16) {
        case 4: hipLaunchKernelGGL((list16s_kernel<4>), grid, block, 0, ctx->stream, a); break;
        case 8: hipLaunchKernelGGL((list16s_kernel<8>), grid, block, 0, ctx->stream, a); break;
        case 16: hipLaunchKernelGGL((list16s_kernel<16>), grid, block, 0, ctx->stream, a); break;
        case 25: {
            const char* ke = getenv("FALCON_L16_KNOCK");      // timing experiments (wrong results)
            // (honoured only together with FALCON_TIMING_EXPERIMENTS=1: a stray variable must not cost a production run its results)
            const char* te = getenv("FALCON_TIMING_EXPERIMENTS");
            const int kn = (ke && te && te[0] == '1') ? atoi(ke) : 0;
            if (kn == 1) hipLaunchKernelGGL((list16s_kernel<25, 1>), grid, block, 0, ctx->stream, a);
            else if (kn == 2) hipLaunchKernelGGL((list16s_kernel<25, 2>), grid, block, 0, ctx->stream, a);
            else if (kn == 3) hipLaunchKernelGGL((list16s_kernel<25, 3>), grid, block, 0, ctx->stream, a);
            else if (kn == 4) hipLaunchKernelGGL((list16s_kernel<25, 4>), grid, block, 0, ctx->stream, a);
            else if (kn == 5) hipLaunchKernelGGL((list16s_kernel<25, 5>), grid, block, 0, ctx->stream, a);
            else if (kn == 6) hipLaunchKernelGGL((list16s_kernel<25, 6>), grid, block, 0, ctx->stream, a);
            else if (kn == 7) hipLaunchKernelGGL((list16s_kernel<25, 7>), grid, block, 0, ctx->stream, a);
            else hipLaunchKernelGGL((list16s_kernel<25>), grid, block, 0, ctx->stream, a);
            break;
        }
        case 50: hipLaunchKernelGGL((list16s_kernel<50>), grid, block, 0, ctx->stream, a); break;
        default:
            set_error("list16s: low_dim %d has no instantiation (64, 128, 256, 400, 800)", a.d);
            return FAL_EUNSUPPORTED;
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
FAL_WARM_KERNEL(fal::list16s_kernel<25>);      // (fal_ctx_plan: this unit's code object is loaded up front)
