// dense4ab_kernel: dense4_kernel (scan.hip) with TWO waves per SIMD.
//
// dense4_kernel keeps a wave's 32-row query tile in registers: 200 of them at low_dim 400, 436 in all -- ONE wave per SIMD.  A
// wave issues in order, so every instruction it has to issue between two MFMAs (operand reads, row DMAs, block stores, their
// addresses) and every `s_waitcnt` idles the matrix pipe: 16.6 k cycles per chunk against 12.8 k of MFMA issue, the pipes busy
// 56.7 % of the CUs' cycles (profiles/r6_pmc_dense4_counters.txt).  Registers are what forbids a second wave, so the query tile
// is SPLIT ALONG K between two waves: wave A of a tile holds the columns [0, d/4) + [d/2, 3d/4) of its 32 queries, wave B the
// rest -- 100 registers each -- and a block's k-ordered chain (simtile.h) runs front to back through both: A computes the first
// half of the chain of chunk s and leaves the 32 x 32 accumulators in LDS, B picks them up one step later, finishes the chain
// and stores the block (twice: the symmetric trick of dense4_kernel).  While B works on chunk s - 1, A is already in chunk s:
// the two waves of a SIMD (waves w and w + 4 of the workgroup share one) always have independent MFMA chains in flight.
// The candidate stream is the same (row-contiguous LDS-DMA, one chunk ahead), cut into an A part and a B part of every row
// (two 400-byte segments each); bit-identical to dense4_kernel (the same chain, the same stores).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "simtile.h"
#include "scan.h"
#include "ivf.h"

namespace fal {

template <int DQ>      // 16-byte steps per role and row half: d = 16 DQ (25: low_dim 400)
__global__ __launch_bounds__(512, 1) void dense4ab_kernel(const float* __restrict__ X, const DenseJob* __restrict__ jobs,
                                                          float* __restrict__ sims, int64_t sims_base,
                                                          int32_t* __restrict__ cursors, const int32_t* __restrict__ table, int knock) {
    constexpr int D = 16 * DQ;                     // columns of a row
    constexpr int HB = 16 * DQ;                    // bytes of one role's share of a row half (D / 4 floats)
    constexpr int PB = 2 * HB;                     // bytes of a row's part (both halves)
    constexpr int PRS = PB + 16;                   // LDS stride of a part row (conflict-free b128 operand reads)
    constexpr int kLanes = PB / 16;                // DMA lanes per part row (50)
    static_assert(kLanes <= 64, "one DMA instruction per part row");
    constexpr int kStores = 20;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int32_t next_item;
    auto part = [&](int P, int parity) -> unsigned char* { return lds + (2 * P + parity) * (32 * PRS); };      // [role][parity]
    float* const hand = reinterpret_cast<float*>(lds + 128 * PRS);                                     // [tile][parity][4][64][4]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int tw = wv & 3, role = wv >> 2;

    auto lds_addr = [](const void* p) -> uint32_t { return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p; };
    // one part (role P) of chunk rows [c0, c0 + 32) into `dst`: wave wv brings rows 4 wv .. 4 wv + 3, one instruction per row
    auto issue_part = [&](int P, const float* rows, int nc, int c0, unsigned char* dst) {
        if (lane < kLanes) {
            const int hh = lane >= DQ, j = lane - hh * DQ;
            const uint32_t l0 = lds_addr(dst);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * wv + i;
                const unsigned char* src = reinterpret_cast<const unsigned char*>(rows + (int64_t)min(c0 + row, nc - 1) * D) +
                                           hh * (2 * HB) + P * HB + 16 * j;
                const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + (uint32_t)(row * PRS)));
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory", "m0");
            }
        }
    };

    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // (knock & 8: cycle stamps -- time at the step barrier / in the step's body / steps taken, per role, summed into cursors[16..])
    unsigned long long t_bar = 0, t_body = 0, n_steps = 0, n_work = 0;
    bool stored = false;                                     // this wave's last VM operations are a block's stores
    for (int turn = 0; turn < 8; ++turn) {
        const int xl = (int)((xcc + turn) & 7);
        const int n_items = table[xl + 1] - table[xl];
        const int32_t* const items = table + 16 + 2 * table[xl];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0) next_item = atomicAdd(&cursors[xl], 1);
        __syncthreads();
        int cur = next_item;
        __syncthreads();
        while (cur < n_items) {
            int fetched = 0;
            if (tid == 0) fetched = atomicAdd(&cursors[xl], 1);      // the item after this one (used at the end)
            const int g = items[2 * cur + 1];
            const DenseJob job = jobs[items[2 * cur]];
            const float* const rows = X + job.c_row0 * D;
            const int lt = 4 * g + tw;                           // this wave pair's 32-query tile of the bucket
            const int nc = job.nc, ncp = (nc + 31) & ~31;
            const bool active = 32 * lt < job.nq;
            float* const out = sims + (job.obase - sims_base) + (int64_t)(32 * lt) * ncp + r;
            float* const outT = sims + (job.obase - sims_base) + (int64_t)r * ncp + 32 * lt + 4 * h;
            const int c_last = ((nc - 1) >> 5) << 5;
            const int c_stop = 32 * lt, c_first = 128 * g;
            const int n_chunks = (c_last - c_first) / 32 + 1;
            float q[4 * DQ];                                     // this wave's K-share of its tile's queries (taken from the stream)
            // (nothing of the previous item is in flight past its last barrier except a B wave's stores)
            issue_part(0, rows, nc, c_first, part(0, 0));
            for (int s = 0; s <= n_chunks; ++s) {
                // this wave's row DMAs of the previous step (or of the line above) have landed; everybody's accumulators of the
                // previous step are in LDS; a B wave's 20 stores, issued behind its DMAs, may stay in flight
                const unsigned long long ts0 = (knock & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
                if (stored) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kStores) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                stored = false;
                const unsigned long long ts1 = (knock & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
                if (knock & 8) { t_bar += ts1 - ts0; ++n_steps; }
                if (!(knock & 2) || s < 2) {                     // (knock: timing experiments only, tools/dense4_ab.py)
                    if (s + 1 < n_chunks) issue_part(0, rows, nc, c_first + 32 * (s + 1), part(0, (s + 1) & 1));
                    if (s < n_chunks) issue_part(1, rows, nc, c_first + 32 * s, part(1, s & 1));
                }
                const int sc = role ? s - 1 : s;                 // the chunk this wave works on in this step
                const int c0 = c_first + 32 * sc;
                if (sc < 0 || sc >= n_chunks || !active || c0 < c_stop) continue;      // (wave-uniform)
                ++n_work;
                const unsigned char* lb = part(role, sc & 1) + r * PRS + h * HB;
                if (c0 == c_stop) {                              // the tile's own rows come by: its queries
#pragma unroll
                    for (int j = 0; j < DQ; ++j) {
                        const float4 v = *reinterpret_cast<const float4*>(lb + 16 * j);
                        q[4 * j + 0] = v.x; q[4 * j + 1] = v.y; q[4 * j + 2] = v.z; q[4 * j + 3] = v.w;
                    }
                }
                float* const hp = hand + ((tw * 2 + (sc & 1)) * 4) * 256 + 4 * lane;      // [quad j][lane][4]
                f32x16 acc;
                if (role == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                } else {
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        const float4 v = *reinterpret_cast<const float4*>(hp + 256 * jq);
                        acc[4 * jq + 0] = v.x; acc[4 * jq + 1] = v.y; acc[4 * jq + 2] = v.z; acc[4 * jq + 3] = v.w;
                    }
                }
                constexpr int kRing = 4;
                float4 ring[kRing];
#pragma unroll
                for (int j = 0; j < kRing; ++j) ring[j] = *reinterpret_cast<const float4*>(lb + 16 * j);
#pragma unroll
                for (int j = 0; j < DQ; ++j) {
                    const float4 sv = ring[j % kRing];
                    if (j + kRing < DQ && !(knock & 4)) ring[j % kRing] = *reinterpret_cast<const float4*>(lb + 16 * (j + kRing));
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 0], sv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 1], sv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 2], sv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * j + 3], sv.w, acc, 0, 0, 0);
                }
                asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc));       // (simtile.h: the last MFMA's passes before the first read)
                if (role == 0) {
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq)
                        *reinterpret_cast<float4*>(hp + 256 * jq) = make_float4(acc[4 * jq], acc[4 * jq + 1], acc[4 * jq + 2], acc[4 * jq + 3]);
                } else if (!(knock & 1)) {
                    // the finished block, twice: 16 rows of the block, then the transposed block in four 16-byte columns
#pragma unroll
                    for (int k = 0; k < 16; ++k) out[mfma32_row(k, h) * ncp + c0] = acc[k];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<float4*>(outT + (int64_t)c0 * ncp + 8 * gq) =
                            make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
                    stored = true;
                }
                if (knock & 8) t_body += __builtin_amdgcn_s_memtime() - ts1;
            }
            if (tid == 0) next_item = fetched;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stored = false;
            cur = next_item;
            __syncthreads();
        }
    }
    if ((knock & 8) && lane == 0 && tw == 0) {
        unsigned long long* c64 = reinterpret_cast<unsigned long long*>(cursors + 16);
        atomicAdd(c64 + 4 * role + 0, t_bar);
        atomicAdd(c64 + 4 * role + 1, t_body);
        atomicAdd(c64 + 4 * role + 2, n_steps);
        atomicAdd(c64 + 4 * role + 3, n_work);
    }
}
FAL_WARM_KERNEL(dense4ab_kernel<25>);

bool dense4ab_supports(int d) { return d == 400; }

int launch_dense4ab(fal_ctx* ctx, const float* X, int d, const DenseJob* jobs, const DenseJob* jobs_host, int n_jobs, float* sims,
                    int64_t sims_base) {
    if (n_jobs <= 0) return FAL_OK;
    FAL_REQUIRE(dense4ab_supports(d), FAL_EUNSUPPORTED, "dense4ab: low_dim %d has no instantiation (400)", d);
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
    constexpr int DQ = 25;
    const size_t lds = (size_t)128 * (2 * 16 * DQ + 16) + 4 * 2 * 4 * 256 * sizeof(float);
    int32_t* cursors = nullptr;
    FAL_TRY(ctx->reserve(SLOT_CURSORS, sizeof(int32_t) * 64, (void**)&cursors));
    dim3 grid((unsigned)std::min<int64_t>(n_items, ctx->persistent_wgs)), block(512);
    StageScope ts(ctx, ST_SCAN);
    StageScope tk(ctx, ST_KERNEL);
    FAL_CHECK_HIP(hipMemsetAsync(cursors, 0, sizeof(int32_t) * 8, ctx->stream));
    FAL_CHECK_HIP(hipFuncSetAttribute((const void*)dense4ab_kernel<DQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char* ke = getenv("FALCON_AB_KNOCK");             // timing experiments (wrong results): 1 no stores, 2 no row DMAs, 4 no operand reads
    const char* te = getenv("FALCON_TIMING_EXPERIMENTS");   // (the knock-outs are honoured only together with it)
    const int knock = (ke && te && te[0] == '1') ? atoi(ke) : 0;
    if (knock & 8) FAL_CHECK_HIP(hipMemsetAsync(cursors + 16, 0, sizeof(int32_t) * 32, ctx->stream));
    hipLaunchKernelGGL((dense4ab_kernel<DQ>), grid, block, lds, ctx->stream, X, jobs, sims, sims_base, cursors, table_dev, knock);
    if (knock & 8) {
        unsigned long long c[8];
        FAL_CHECK_HIP(hipMemcpyAsync(c, cursors + 16, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
        FAL_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        for (int r = 0; r < 2; ++r)
            fprintf(stderr, "[dense4ab] role %c: per workgroup: steps %.0f (with work %.0f), s_memtime ticks per step: barrier %.0f, body %.0f\n",
                    r ? 'B' : 'A', (double)c[4 * r + 2] / grid.x, (double)c[4 * r + 3] / grid.x, (double)c[4 * r] / (double)c[4 * r + 2],
                    (double)c[4 * r + 1] / (double)std::max<unsigned long long>(c[4 * r + 3], 1));
    }
    FAL_CHECK_HIP(hipGetLastError());
    return FAL_OK;
}

}  // namespace fal
