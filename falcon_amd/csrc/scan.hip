// The cosine kernels: fp32-MFMA query x candidate inner products (dense form: k-means assign,
// coarse quantiser, flat-bucket scan) and the wavefront top-k select.
//
// Spec: reference README.md:107-113, 134-142 (Faiss IVF build / n_probe search; no code in the
// snapshot).  See simtile.h for the tile algorithm, DESIGN.md for the roofline.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "common.h"
#include "simtile.h"
#include "scan.h"
#include "select.h"
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

// ---------------------------------------------------------------------------------------------
// dense tile kernel, shared-stream form (flat buckets of more than 32 rows scanned against themselves)
//
// dense_kernel's four waves of a CU each stream their own candidates with one 16-byte load per lane and row: 64 cache
// lines per load instruction, 3,200 line look-ups per wave and 32-candidate chunk -- the texture addresser, not the matrix
// pipe, sets the pace (measured at 1 M spectra: 1.31 ms per launch; with the loads compiled out 0.83; with the same bytes
// read 1 KB per instruction 0.98).  Here the four waves of a workgroup own four CONSECUTIVE 32-query tiles of one bucket and
// share ONE candidate stream: a chunk's 32 rows arrive in LDS once, by row-contiguous LDS-DMA (`global_load_lds_dwordx4`:
// 1 KB of one row per instruction, 8 lines), double-buffered; rows sit at a stride of 4 d + 16 bytes so that the MFMA
// operand reads (lane = candidate row, 16 bytes of its k-half) are conflict-free.  The stream runs UPWARDS from the group's
// own four diagonal chunks -- which are the query rows of its four tiles: a wave takes its queries into registers from LDS
// when its chunk comes by and then meets every later chunk (blocks on and above the diagonal; every off-diagonal block is
// stored twice, once transposed, as in dense_kernel).  Same k-ordered chain per block (simtile.h), same stores: the sims
// are bit-identical.  A wave idles until the stream reaches its diagonal (the 4 x 4 corner of blocks costs 16 slots for 10
// blocks).  Measured (s_memtime per workgroup): 16.6 k cycles per chunk against 12.8 k of MFMA issue.
// ---------------------------------------------------------------------------------------------
//
// MODE 1 / 2 (rows of more than 400 columns: `--low_dim` 401..800 in float32): one launch per K-HALF of rows that are 2 d floats
// wide.  A pass sees the "virtual" row [k_off, k_off + d/2) ++ [d + k_off, d + k_off + d/2) of every physical row (k_off = 0, then
// d/2) -- two contiguous segments, fetched by the same row DMAs -- so its MFMA step kk multiplies the columns (k_off + kk,
// d + k_off + kk): pass 0 then pass 1 walk the k-ordered chain of the 2 d-wide row (simtile.h) front to back.  MODE 2 starts
// every block's accumulators from what pass 0 stored (the block's own [32, 32] slots: loaded one chunk ahead, before the next
// chunk's row DMAs are issued, so the hand-counted waits see them complete) instead of zero: the same chain, bit for bit,
// with the query tile still in registers (200 of them; 400 do not exist).
template <int DH4, int MODE>
__global__ __launch_bounds__(256, 1) void dense4_kernel(const float* __restrict__ X, int d, int k_off, const DenseJob* __restrict__ jobs,
                                                        float* __restrict__ sims, int64_t sims_base,
                                                        int32_t* __restrict__ cursors, const int32_t* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_lds[];
    __shared__ int32_t next_item;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int dh = d >> 1, dh4 = dh >> 2;
    const int row_bytes = d * 4, rs = row_bytes + 16;
    const int rstride = MODE == 0 ? d : 2 * d;               // floats between two rows of X
    unsigned char* const buf0 = stream_lds;
    unsigned char* const buf1 = stream_lds + 32 * rs;
    const unsigned char* const my0 = buf0 + r * rs + h * (dh * 4);      // this lane's operand row in either buffer
    const unsigned char* const my1 = buf1 + r * rs + h * (dh * 4);
    constexpr int kParts = (DH4 * 32 + 1023) / 1024;
    constexpr int kPieces = 8 * kParts;
    constexpr int kStores = 20;                              // VM operations of one block's stores

    // a chunk's 32 rows into LDS: wave w brings rows 8 w .. 8 w + 7, one KB of one row per instruction = one "piece".
    // (inline asm, not the builtin: hipcc guards every later LDS read with `s_waitcnt vmcnt(0)` for a DMA it knows of,
    //  which drains the stores and the next chunk's loads once per operand read; `chunk_barrier` below does the waiting)
    struct Target {
        const float* rows;       // the bucket's first row
        int nc, c0;              // its rows, the chunk's first row
        unsigned char* buf;
        bool valid;
    };
    auto issue_piece = [&](int k, const Target& t) {
        const int p = k / 8, row = 8 * w + (k & 7);
        if (p * 1024 + lane * 16 < row_bytes) {
            const int vb = p * 1024 + lane * 16;             // byte of the (virtual) row this lane brings
            const float* rowp = t.rows + (int64_t)min(t.c0 + row, t.nc - 1) * rstride;
            const unsigned char* src = MODE == 0 ? reinterpret_cast<const unsigned char*>(rowp) + vb
                                                 : reinterpret_cast<const unsigned char*>(rowp + k_off + (vb >= 2 * d ? d : 0)) +
                                                       (vb >= 2 * d ? vb - 2 * d : vb);
            const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane(
                (int)((uint32_t)(size_t)(__attribute__((address_space(3))) const void*)t.buf + (uint32_t)(row * rs + p * 1024)));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
        }
    };
    auto issue = [&](const Target& t) {
#pragma unroll
        for (int k = 0; k < kPieces; ++k) issue_piece(k, t);
    };
    // Wait for this wave's share of the chunk about to be read, then meet the others.  VM operations retire in order:
    // `after` = what the wave has issued SINCE that chunk's loads (one or two blocks' stores) may stay in flight.
    // (Three buffers with the loads two chunks ahead were measured: 17.6 k cycles per chunk against 16.6 k -- the chunk is
    //  not waiting for its rows.  So was a rolled loop over a runtime buffer index: 17.9 k.)
    auto chunk_barrier = [&](int after) {
        if (after >= 2 * kStores) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * kStores) : "memory");
        else if (after >= kStores) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(kStores) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    };

    // Persistent workgroups, one per CU, that pull (bucket, group) items themselves: as one workgroup per item the launch
    // left every CU idle for 19 us on average between two workgroups (median 7, 90th percentile 53: one 100 KB workgroup per
    // CU, dispatched in order -- measured per CU with s_memtime).  Items of XCD list x (jobs x, x + 8, ...: simtile.h; the
    // host lists them in `table`: 9 offsets, then (job, group) pairs from word 16) are taken by the workgroups that RUN on
    // XCD x (its L2 holds the bucket), in order, by a cursor per list; a workgroup whose own list is exhausted helps the
    // next ones.  The cursor runs two items ahead: while item `cur` is computed the next one is known -- its first chunk is
    // loaded under cur's last chunk -- and the fetch of the one after that is in flight.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int turn = 0; turn < 8; ++turn) {
    const int xl = (int)((xcc + turn) & 7);
    const int n_items = table[xl + 1] - table[xl];
    const int32_t* const items = table + 16 + 2 * table[xl];
    FAL_DMA_BARRIER();                                       // (nothing of the previous list is in flight past here)
    if (tid == 0) next_item = atomicAdd(&cursors[xl], 1);
    __syncthreads();
    int cur = next_item;
    __syncthreads();
    if (tid == 0) next_item = atomicAdd(&cursors[xl], 1);
    __syncthreads();
    int nxt = next_item;
    bool preloaded = false, prev_stored = false;
    int par = 0;                                             // the buffer of the item's first chunk
    while (cur < n_items) {
    int fetched = 0;
    if (tid == 0) fetched = atomicAdd(&cursors[xl], 1);      // (consumed at the end of the item)
    const int g = items[2 * cur + 1];
    const DenseJob job = jobs[items[2 * cur]];
    const bool have_next = nxt < n_items;
    Target first_of_next{X, 1, 0, nullptr, false};
    if (have_next) {
        const int nj = items[2 * nxt], ng = items[2 * nxt + 1];
        first_of_next = Target{X + jobs[nj].c_row0 * rstride, jobs[nj].nc, 128 * ng, nullptr, true};
    }
    const float* const rows = X + job.c_row0 * rstride;
    const int lt = 4 * g + w;                                // this wave's 32-query tile of the bucket
    const int nc = job.nc, ncp = (nc + 31) & ~31;
    const bool active = 32 * lt < job.nq;

    float* const out = sims + (job.obase - sims_base) + (int64_t)(32 * lt) * ncp + r;
    float* const outT = sims + (job.obase - sims_base) + (int64_t)r * ncp + 32 * lt + 4 * h;
    const int c_last = ((nc - 1) >> 5) << 5;
    const int c_stop = 32 * lt;                              // this wave's diagonal chunk = its own query rows
    const int c_first = 128 * g;                             // the group's first diagonal
    const int n_chunks = (c_last - c_first) / 32 + 1;

    float q[DH4 * 4];
    f32x16 prev;
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[i] = 0.f;
    int prev_c0 = min(c_stop, c_last);
    // MODE 2: the accumulators the NEXT block starts from = what the first K-half's pass stored in the block's slots
    f32x16 init;
    auto load_init = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) init[i] = out[mfma32_row(i, h) * ncp + min(c0, c_last)];
    };
    // the 20 stores of a finished block (first call: zeros into the diagonal chunk's slots, overwritten by the real
    // epilogue later in program order): 16 rows of the block, then the transposed block in four 16-byte columns
    auto store_piece = [&](int k) {
        if (k < 16) {
            out[mfma32_row(k, h) * ncp + prev_c0] = prev[k];
        } else {
            const int gq = k - 16;
            *reinterpret_cast<float4*>(outT + (int64_t)prev_c0 * ncp + 8 * gq) =
                make_float4(prev[4 * gq], prev[4 * gq + 1], prev[4 * gq + 2], prev[4 * gq + 3]);
        }
    };
    auto epilogue = [&]() {
#pragma unroll
        for (int k = 0; k < kStores; ++k) store_piece(k);
    };
    // One chunk: DH4 steps of one 16-byte operand read + four MFMAs.  The wave's share of the NEXT chunk's DMA goes out one
    // piece per step in the first half, the previous block's stores one per step in the second half.
    auto compute = [&](const unsigned char* lb, int c0, const Target& next) {
        constexpr int kRing = 4, kHalf = DH4 / 2;
        constexpr int kDmaPer = (kPieces + kHalf - 1) / kHalf, kStPer = (kStores + (DH4 - kHalf) - 1) / (DH4 - kHalf);
        auto ld = [&](int j) { return *reinterpret_cast<const float4*>(lb + 16 * (j < dh4 ? j : dh4 - 1)); };
        float4 ring[kRing];
#pragma unroll
        for (int j = 0; j < kRing; ++j) ring[j] = ld(j);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = MODE == 2 ? init[i] : 0.f;
        if (MODE == 2) load_init(c0 + 32);                   // (VM loads in front of this chunk's row DMAs and stores)
#pragma unroll
        for (int j = 0; j < DH4; ++j) {
            const float4 s = ring[j % kRing];
            if (j + kRing < DH4) ring[j % kRing] = ld(j + kRing);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 0], s.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 1], s.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 2], s.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 3], s.w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (j < kHalf) {
                if (next.valid) {
#pragma unroll
                    for (int k = j * kDmaPer; k < (j + 1) * kDmaPer && k < kPieces; ++k) issue_piece(k, next);
                }
            } else {
#pragma unroll
                for (int k = (j - kHalf) * kStPer; k < (j - kHalf + 1) * kStPer && k < kStores; ++k) store_piece(k);
                __builtin_amdgcn_sched_group_barrier(0x040, kStPer, 0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc));       // (simtile.h: the last MFMA's passes before the first read)
        prev = acc;
        prev_c0 = c0;
    };
    // The group's stream starts at its own four diagonal chunks -- which ARE the query rows of its four tiles: a wave takes
    // its 32 rows into registers from LDS when its chunk comes by (no second pass over them; as one 16-byte load per lane and
    // row straight from memory the 51 KB of a tile cost 15 us of every workgroup), then meets every later chunk.
    auto take_queries = [&](const unsigned char* lb) {
#pragma unroll
        for (int j = 0; j < DH4; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < dh4) v = *reinterpret_cast<const float4*>(lb + 16 * j);
            q[4 * j + 0] = v.x;
            q[4 * j + 1] = v.y;
            q[4 * j + 2] = v.z;
            q[4 * j + 3] = v.w;
        }
    };

    // chunk ci sits in buffer (par + ci) & 1; behind the first chunk's loads (issued under the previous item's last chunk)
    // the wave has stored that item's last two blocks
    int after = 0;
    if (preloaded) after = prev_stored ? 2 * kStores : 0;
    else issue(Target{rows, nc, c_first, par ? buf1 : buf0, true});
    bool odd = par != 0;
    int ci = 0, c0 = c_first;
    for (;;) {
        if (!odd) {
            chunk_barrier(after);                            // chunk ci has landed in buf0, buf1 is free again
            Target next = ci + 1 < n_chunks ? Target{rows, nc, c0 + 32, buf1, true} : first_of_next;
            next.buf = buf1;
            if (active && c0 == c_stop) {
                if (MODE == 2) load_init(c0);
                take_queries(my0);
            }
            if (active && c0 >= c_stop) {
                compute(my0, c0, next);
                after = kStores;
            } else {
                if (next.valid) issue(next);
                after = 0;
            }
            ++ci;
            c0 += 32;
            if (ci >= n_chunks) break;
        }
        odd = false;
        chunk_barrier(after);                                // chunk ci has landed in buf1, buf0 is free again
        Target next = ci + 1 < n_chunks ? Target{rows, nc, c0 + 32, buf0, true} : first_of_next;
        next.buf = buf0;
        if (active && c0 == c_stop) {
            if (MODE == 2) load_init(c0);
            take_queries(my1);
        }
        if (active && c0 >= c_stop) {
            compute(my1, c0, next);
            after = kStores;
        } else {
            if (next.valid) issue(next);
            after = 0;
        }
        ++ci;
        c0 += 32;
        if (ci >= n_chunks) break;
    }
    if (active) epilogue();
    // the cursor value fetched at the top becomes the item after next
    if (tid == 0) next_item = fetched;
    // (the next item's first chunk stays in flight across this barrier -- it only hands `next_item` over; the chunk is
    //  waited for by the next item's first chunk_barrier.  The tag exempts it from tests/isa_lint.py's DMA-barrier rule.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier ; dma-ok: cursor hand-off only" ::: "memory");
    const int nxt2 = next_item;
    par = (par + n_chunks) & 1;
    preloaded = have_next;
    prev_stored = active;
    cur = nxt;
    nxt = nxt2;
    }
    }
}

// Buckets of up to 32 rows: one 32 x 32 block each, query rows = candidate rows = the SAME registers (lane (r, h) holds row r's
// k-half h for either operand).  A persistent workgroup takes four such buckets at a time, one per wave; the rows arrive by
// row-contiguous LDS-DMA (two buckets per round: the staging area is dense4_kernel's) and each wave runs its chain with
// A = B = its registers.  (As one-wave workgroups of dense_kernel these buckets cost 66 us per tile: 15 us of
// lane-strided row loads, the dispatch of 14,000 workgroups, one chunk of work.)
template <int DH4>
__global__ __launch_bounds__(256, 1) void dense_tiny4_kernel(const float* __restrict__ X, int d, const DenseJob* __restrict__ jobs,
                                                             int n_jobs, float* __restrict__ sims, int64_t sims_base,
                                                             int32_t* __restrict__ cursor) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_lds[];
    __shared__ int32_t next_item;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int dh = d >> 1, dh4 = dh >> 2;
    const int row_bytes = d * 4, rs = row_bytes + 16;
    constexpr int kParts = (DH4 * 32 + 1023) / 1024;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)stream_lds;
    for (;;) {
        __syncthreads();
        if (tid == 0) next_item = atomicAdd(cursor, 1);
        __syncthreads();
        const int j0 = 4 * next_item;
        if (j0 >= n_jobs) break;
        const bool active = j0 + w < n_jobs;
        float q[DH4 * 4];
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (round) __syncthreads();                      // (the first round's rows have been taken)
            // 64 staging rows = two buckets of up to 32 rows; wave w brings staging rows 16 w .. 16 w + 15
            const int jb = j0 + 2 * round + (w >> 1);        // the bucket of this wave's 16 staging rows
            if (jb < n_jobs) {
                const int64_t row0 = jobs[jb].c_row0;
                const int nb = jobs[jb].nq;
#pragma unroll
                for (int p = 0; p < kParts; ++p) {
                    if (p * 1024 + lane * 16 < row_bytes) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int sr = 16 * w + i;                                    // staging row
                            const unsigned char* src = reinterpret_cast<const unsigned char*>(X + (row0 + min(sr & 31, nb - 1)) * d) +
                                                       p * 1024 + lane * 16;
                            const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)(sr * rs + p * 1024)));
                            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if ((w >> 1) == round && active) {
                const unsigned char* lb = stream_lds + ((w & 1) * 32 + r) * rs + h * (dh * 4);
#pragma unroll
                for (int j = 0; j < DH4; ++j) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < dh4) v = *reinterpret_cast<const float4*>(lb + 16 * j);
                    q[4 * j + 0] = v.x;
                    q[4 * j + 1] = v.y;
                    q[4 * j + 2] = v.z;
                    q[4 * j + 3] = v.w;
                }
            }
        }
        if (!active) continue;
        const DenseJob job = jobs[j0 + w];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int j = 0; j < DH4 * 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[j], q[j], acc, 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc));       // (simtile.h: the last MFMA's passes before the first read)
        float* const out = sims + (job.obase - sims_base) + r;   // one [32, 32] block (nc <= 32)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[mfma32_row(i, h) * 32] = acc[i];
    }
}


bool dense4_supports(int d) { return d % 8 == 0 && d >= 32 && (d <= 400 || (d > 512 && d <= 800 && d % 16 == 0)); }
// (low_dim 512: the 64-step form needs scratch -> dense_kernel; 513..800: two K-half passes of the 50-step form, MODE 1 / 2)

// jobs_host = the host copy of `jobs` (sorted by decreasing size: job j belongs to XCD list j % 8)
int launch_dense4(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, const DenseJob* jobs_host, int n_jobs, float* sims,
                  int64_t sims_base) {
    if (n_jobs <= 0) return FAL_OK;
    {
        const char* e = getenv("FALCON_DENSE4");              // "ab": the two-waves-per-SIMD form (dense4ab.hip), read per launch
        if (e && !strcmp(e, "ab") && dense4ab_supports(d)) return launch_dense4ab(ctx, X, d, jobs, jobs_host, n_jobs, sims, sims_base);
    }
    const bool split = d > 512;
    const int dv = split ? d / 2 : d;                        // columns one pass sees
    const int dh4 = dv / 8;
    // the item table: 9 offsets (in items), then the (job, 128-row group) pairs of XCD list 0, 1, ... from word 16
    std::vector<int32_t> table(16, 0);
    for (int x = 0; x < 8; ++x) {
        for (int j = x; j < n_jobs; j += 8)
            for (int g = 0; g < (jobs_host[j].nq + 127) / 128; ++g) {
                table.push_back(j);
                table.push_back(g);
            }
        table[x + 1] = (int32_t)((table.size() - 16) / 2);
    }
    const int64_t n_items = table[8];
    int32_t* table_dev = nullptr;
    FAL_TRY(ctx->reserve(SLOT_ITEMS, sizeof(int32_t) * table.size(), (void**)&table_dev));
    FAL_TRY(ctx->upload(table_dev, table.data(), sizeof(int32_t) * table.size()));
    const size_t lds = (size_t)2 * 32 * (dv * 4 + 16);
    int32_t* cursors = nullptr;
    FAL_TRY(ctx->reserve(SLOT_CURSORS, sizeof(int32_t) * 16, (void**)&cursors));
    dim3 grid((unsigned)std::min<int64_t>(n_items, ctx->persistent_wgs)), block(256);
    StageScope ts(ctx, ST_SCAN);
    StageScope tk(ctx, ST_KERNEL);
#define FAL_LAUNCH_DENSE4(DH4, MODE, KOFF)                                                                          \
    do {                                                                                                            \
        FAL_CHECK_HIP(hipMemsetAsync(cursors, 0, sizeof(int32_t) * 8, ctx->stream));                                \
        FAL_CHECK_HIP(hipFuncSetAttribute((const void*)dense4_kernel<DH4, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds));                                                               \
        hipLaunchKernelGGL((dense4_kernel<DH4, MODE>), grid, block, lds, ctx->stream, X, dv, KOFF, jobs, sims, sims_base, cursors, \
                           table_dev);                                                                              \
    } while (0)
    if (split) {
        if (dh4 > 50 || d % 16 != 0) {
            set_error("dense4 supports low_dim <= 400 and 513..800 in steps of 16 (got %d)", d);
            return FAL_EUNSUPPORTED;
        }
        FAL_LAUNCH_DENSE4(50, 1, 0);                         // columns [0, d/4) + [d/2, 3d/4) ...
        FAL_LAUNCH_DENSE4(50, 2, dv / 2);                    // ... then [d/4, d/2) + [3d/4, d), from the first pass's sums
    } else if (dh4 <= 8) FAL_LAUNCH_DENSE4(8, 0, 0);
    else if (dh4 <= 16) FAL_LAUNCH_DENSE4(16, 0, 0);
    else if (dh4 <= 32) FAL_LAUNCH_DENSE4(32, 0, 0);
    else if (dh4 <= 50) FAL_LAUNCH_DENSE4(50, 0, 0);
    else {
        set_error("dense4 supports low_dim <= 400 and 513..800 in steps of 16 (got %d)", d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_DENSE4
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

// Flat buckets of fewer than 64 rows beyond the fp32 matrix kernels' low_dim (512): the exact k-ordered fmaf chain on the vector
// ALU, one wave per bucket, a lane per candidate -- the oracle's bits (simtile.h: exact_dot) where the f16 matrix cores are
// within 2e-6 only.  Round 5: the two-or-three-row buckets at the ends of a float16 / low_dim 800 job's precursor range
// (BASELINE configs[4]) were the one place where the neighbour distances were not bit-identical.  Same block layout as the
// other flat kernels: one [32, ceil32(nc)] block per 32-query tile.
__global__ __launch_bounds__(64) void flat_exact_small_kernel(const float* __restrict__ X, int d, const DenseJob* __restrict__ jobs,
                                                              int n_jobs, float* __restrict__ sims, int64_t sims_base) {
    const int lane = threadIdx.x;
    for (int j = blockIdx.x; j < n_jobs; j += gridDim.x) {
        const DenseJob job = jobs[j];
        const int nc = job.nc, W = (nc + 31) & ~31;
        float* const out = sims + (job.obase - sims_base);
        for (int i = 0; i < job.nq; ++i)
            for (int c = lane; c < nc; c += 64)
                out[(int64_t)(i >> 5) * 32 * W + (i & 31) * W + c] = exact_dot(X + (job.q_row0 + i) * (int64_t)d, X + (job.c_row0 + c) * (int64_t)d, d);
    }
}

int launch_flat_exact_small(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, int n_jobs, float* sims, int64_t sims_base) {
    if (n_jobs <= 0) return FAL_OK;
    StageScope ts(ctx, ST_SCAN);
    hipLaunchKernelGGL(flat_exact_small_kernel, dim3((unsigned)std::min<int64_t>(n_jobs, (int64_t)ctx->num_cus * 16)), dim3(64), 0,
                       ctx->stream, X, d, jobs, n_jobs, sims, sims_base);
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

// jobs of up to 32 rows each (any order)
int launch_dense_tiny4(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, int n_jobs, float* sims, int64_t sims_base) {
    if (n_jobs <= 0) return FAL_OK;
    const int dh4 = d / 8;
    const size_t lds = (size_t)2 * 32 * (d * 4 + 16);
    int32_t* cursors = nullptr;
    FAL_TRY(ctx->reserve(SLOT_CURSORS, sizeof(int32_t) * 16, (void**)&cursors));
    cursors += 8;                                            // (the first eight are dense4_kernel's, possibly still in use)
    dim3 grid((unsigned)std::min<int64_t>(ceil_div(n_jobs, 4), ctx->persistent_wgs)), block(256);
    StageScope ts(ctx, ST_SCAN);                             // (not ST_KERNEL: that stage times the dominant kernel alone)
    FAL_CHECK_HIP(hipMemsetAsync(cursors, 0, sizeof(int32_t), ctx->stream));
#define FAL_LAUNCH_TINY4(DH4)                                                                                       \
    do {                                                                                                            \
        FAL_CHECK_HIP(hipFuncSetAttribute((const void*)dense_tiny4_kernel<DH4>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)lds));                                                               \
        hipLaunchKernelGGL((dense_tiny4_kernel<DH4>), grid, block, lds, ctx->stream, X, d, jobs, n_jobs, sims, sims_base, \
                           cursors);                                                                                \
    } while (0)
    if (dh4 <= 8) FAL_LAUNCH_TINY4(8);
    else if (dh4 <= 16) FAL_LAUNCH_TINY4(16);
    else if (dh4 <= 32) FAL_LAUNCH_TINY4(32);
    else if (dh4 <= 50) FAL_LAUNCH_TINY4(50);
    else {
        set_error("dense_tiny4 supports low_dim <= 400 (got %d)", d);
        return FAL_EUNSUPPORTED;
    }
#undef FAL_LAUNCH_TINY4
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

template <int EPI>
static int launch_dense_t(fal_ctx* ctx, int stage, const float* Q, const float* Cm, int d, const DenseJob* jobs,
                          int n_jobs, int64_t tile_begin, int64_t n_tiles, float* sims, int64_t sims_base,
                          int32_t* assign, int64_t xcd_list_tiles) {
    if (n_tiles <= 0) return FAL_OK;
    // a flat bucket against itself (XCD-list mode is only used for that): exploit the symmetry
    const int symmetric = (EPI == EPI_STORE && xcd_list_tiles > 0 && Q == Cm) ? 1 : 0;
    const int dh4 = d / 8;
    const int xcd_lists = xcd_list_tiles > 0;
    // XCD-list mode: n_tiles is unused, the grid is 8 x (longest list)
    const int64_t per_xcd = xcd_lists ? xcd_list_tiles : (n_tiles + 7) / 8;
    FAL_REQUIRE(per_xcd * 8 < (int64_t)INT32_MAX, FAL_EUNSUPPORTED, "too many tiles in one launch");
 dim3 grid((unsigned)(per_xcd * 8)), block(64);
    StageScope ts(ctx, stage);
    StageScope tk(ctx, ST_KERNEL, nullptr, stage == ST_SCAN);      // (a second event pair for the fine scan only)
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

// (Four queries per workgroup, taken in turn -- a test of whether the dispatch of 700,000 one-wave workgroups bounds the flat
//  select of 1 M spectra: 0.54 -> 0.80 ms per launch.  It does not; the waves in flight do.)
template <int MODE, bool FUSE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == MODE_DENSE ? 8 : 3, 8))) void select_kernel(SelectArgs a) {
    __shared__ uint32_t sel_u[kSelBuf];       // the selected set (+ room for one streamed chunk of a long row)
    __shared__ uint32_t sel_id[kSelBuf];
    __shared__ int64_t seg_off[FAL_MAX_N_PROBE + 1];   // MODE_IVF: stream offset of each probed list
    __shared__ int64_t seg_src[FAL_MAX_N_PROBE];       // MODE_IVF: perm position of each probed list
    const int lane = threadIdx.x;
    const int k = a.k;
    const unsigned qblock = blockIdx.x;

    // ---- locate this query ---------------------------------------------------------------
    SelQuery qy{nullptr, 0, 0};
    int64_t out_row = 0;
    const int64_t t = a.tile_begin + (qblock >> 5);
    const int ql = qblock & 31;
    const DenseJob job = a.jobs[a.tile_job[qblock >> 5]];
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
        ctx->release(SLOT_TILEJOB);        // a launcher-local table: the previous launcher's pointer is dead
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
FAL_WARM_KERNEL(fal::dense4_kernel<50, 0>);      // (fal_ctx_plan: this unit's code object is loaded up front)
