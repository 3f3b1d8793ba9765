// a7: n_probe search of every row against its bucket's index, top-k_ann by (sim desc, id asc).
//
// Flat buckets (one list): dense tile scan of the bucket against itself + wavefront select.
// IVF buckets: coarse quantiser (dense scan vs the bucket's centroids + select k = n_probe),
// candidate-count prefix, fine scan over the union of probed lists, select.
// The sims of a batch of tiles live in one scratch buffer sized to stay inside the 256 MiB
// Infinity Cache, so the scan -> select hand-off does not travel through HBM.
#include <algorithm>
#include <stdlib.h>
#include "common.h"
#include "scan.h"
#include "ivf.h"

namespace fal {

// total candidates of each query = sum of the sizes of its probed lists (0 where probes are -1)
__global__ void probe_totals_kernel(const int32_t* __restrict__ probes, int np, const DenseJob* __restrict__ jobs,
                                    int n_jobs, int64_t n_tiles, const int64_t* __restrict__ list_off,
                                    int64_t* __restrict__ totals) {
    const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t t = g >> 5;
    if (t >= n_tiles) return;
    const DenseJob job = jobs[find_job(jobs, n_jobs, t)];
    const int lt = (int)(t - job.tile0), ql = (int)(g & 31);
    if (32 * lt + ql >= job.nq) return;
    const int64_t p = job.q_row0 + 32 * (int64_t)lt + ql;
    int64_t tot = 0;
    for (int j = 0; j < np; ++j) {
        const int l = probes[p * np + j];
        if (l >= 0) tot += list_off[job.c_row0 + l + 1] - list_off[job.c_row0 + l];
    }
    totals[p] = tot;
}

static size_t sims_capacity_floats() {
    static size_t cap = 0;
    if (!cap) {
        const char* e = getenv("FALCON_SIMS_MB");
        size_t mb = e ? (size_t)atoll(e) : 160;
        if (mb < 16) mb = 16;
        cap = mb * 1024 * 1024 / sizeof(float);
    }
    return cap;
}

}  // namespace fal

using namespace fal;

extern "C" int fal_ivf_search_topk(fal_ctx* ctx, const fal_ivf* ivf, int n_probe, int k_ann, float* sim, int32_t* idx) {
    FAL_REQUIRE(ctx && ivf, FAL_EINVAL, "fal_ivf_search_topk: NULL ctx/ivf");
    FAL_REQUIRE(k_ann >= 1 && k_ann <= FAL_MAX_K_ANN, FAL_EUNSUPPORTED, "fal_ivf_search_topk: k_ann must be in [1, %d]", FAL_MAX_K_ANN);
    FAL_REQUIRE(n_probe >= 1 && n_probe <= FAL_MAX_N_PROBE, FAL_EUNSUPPORTED, "fal_ivf_search_topk: n_probe must be in [1, %d]", FAL_MAX_N_PROBE);
    if (ivf->n == 0) return FAL_OK;
    FAL_REQUIRE(sim && idx, FAL_EINVAL, "fal_ivf_search_topk: NULL output");
    hipStream_t st = ctx->stream;
    const int d = ivf->d;
    const int64_t n_buckets = (int64_t)ivf->n_list.size();
    ctx->stage_reset(ST_COARSE);
    ctx->stage_reset(ST_SCAN);
    ctx->stage_reset(ST_SELECT);
    const size_t cap = sims_capacity_floats();

    // ---- job tables ------------------------------------------------------------------------
    std::vector<DenseJob> flat, coarse;    // coarse doubles as the IVF tile table
    int64_t flat_tiles = 0, ivf_tiles = 0, flat_o = 0, coarse_o = 0;
    int max_n_list = 0;
    for (int64_t b = 0; b < n_buckets; ++b) {
        const int64_t row0 = ivf->bucket_off[b], nb = ivf->bucket_off[b + 1] - row0;
        if (nb == 0) continue;
        const int64_t tiles = ceil_div(nb, 32);
        if (ivf->n_list[b] == 1) {
            flat.push_back({row0, row0, flat_o, flat_tiles, (int32_t)nb, (int32_t)nb});
            flat_tiles += tiles;
            flat_o += tiles * 32 * nb;
        } else {
            coarse.push_back({row0, ivf->list_base[b], coarse_o, ivf_tiles, (int32_t)nb, ivf->n_list[b]});
            ivf_tiles += tiles;
            coarse_o += tiles * 32 * (int64_t)ivf->n_list[b];
            max_n_list = std::max(max_n_list, ivf->n_list[b]);
        }
    }
    // batches of whole tiles whose sims fit the buffer: [tile_begin, tile_end)
    auto make_batches = [&](const std::vector<DenseJob>& jobs, int64_t n_tiles, std::vector<int64_t>& cuts, size_t& need) {
        cuts.assign(1, 0);
        need = 0;
        int64_t used = 0;
        for (const DenseJob& j : jobs) {
            const int64_t tiles = ceil_div(j.nq, 32), per_tile = 32 * (int64_t)j.nc;
            for (int64_t t = 0; t < tiles;) {
                int64_t room = ((int64_t)cap - used) / per_tile;
                if (room <= 0 && used > 0) {
                    cuts.push_back(j.tile0 + t);
                    used = 0;
                    continue;
                }
                if (room <= 0) room = 1;   // a single tile larger than the buffer: grow the buffer
                const int64_t take = std::min(room, tiles - t);
                used += take * per_tile;
                need = std::max(need, (size_t)used);
                t += take;
            }
        }
        if (cuts.back() != n_tiles) cuts.push_back(n_tiles);
    };
    auto obase_of_tile = [&](const std::vector<DenseJob>& jobs, int64_t t) -> int64_t {
        // jobs sorted by tile0
        size_t lo = 0, hi = jobs.size() - 1;
        while (lo < hi) {
            size_t mid = (lo + hi + 1) / 2;
            if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
        }
        return jobs[lo].obase + (t - jobs[lo].tile0) * 32 * (int64_t)jobs[lo].nc;
    };

    std::vector<int64_t> flat_cuts, coarse_cuts;
    size_t need_flat = 0, need_coarse = 0;
    if (!flat.empty()) make_batches(flat, flat_tiles, flat_cuts, need_flat);
    if (!coarse.empty()) make_batches(coarse, ivf_tiles, coarse_cuts, need_coarse);

    DenseJob *flat_dev = nullptr, *coarse_dev = nullptr;
    if (!flat.empty()) {
        FAL_TRY(ctx->reserve(SLOT_JOBS, sizeof(DenseJob) * flat.size(), (void**)&flat_dev));
        FAL_CHECK_HIP(hipMemcpyAsync(flat_dev, flat.data(), sizeof(DenseJob) * flat.size(), hipMemcpyHostToDevice, st));
    }
    if (!coarse.empty()) {
        FAL_TRY(ctx->reserve(SLOT_JOBS2, sizeof(DenseJob) * coarse.size(), (void**)&coarse_dev));
        FAL_CHECK_HIP(hipMemcpyAsync(coarse_dev, coarse.data(), sizeof(DenseJob) * coarse.size(), hipMemcpyHostToDevice, st));
    }
    float* sims = nullptr;
    FAL_TRY(ctx->reserve(SLOT_SIMS, sizeof(float) * std::max(need_flat, need_coarse), (void**)&sims));

    // ---- A. flat buckets ---------------------------------------------------------------------
    for (size_t bi = 0; bi + 1 < flat_cuts.size(); ++bi) {
        const int64_t t0 = flat_cuts[bi], t1 = flat_cuts[bi + 1];
        const int64_t base = obase_of_tile(flat, t0);
        FAL_TRY(launch_dense(ctx, ST_SCAN, EPI_STORE, ivf->Xl, ivf->Xl, d, flat_dev, (int)flat.size(), t0, t1 - t0, sims,
                             base, nullptr));
        SelectArgs sa{};
        sa.sims = sims; sa.sims_base = base; sa.k = k_ann; sa.out_sim = sim; sa.out_idx = idx;
        sa.jobs = flat_dev; sa.n_jobs = (int)flat.size(); sa.tile_begin = t0; sa.ids_are_rows = 1;
        FAL_TRY(launch_select(ctx, ST_SELECT, MODE_DENSE, sa, (t1 - t0) * 32));
    }
    ctx->counters[0] = 0;
    for (const DenseJob& j : flat) ctx->counters[0] += (int64_t)j.nq * j.nc;
    ctx->counters[1] = 0;
    ctx->counters[2] = (int64_t)(flat_cuts.empty() ? 0 : flat_cuts.size() - 1);
    ctx->counters[3] = (int64_t)(sizeof(float) * std::max(need_flat, need_coarse));
    if (coarse.empty()) {
        FAL_CHECK_HIP(hipStreamSynchronize(st));   // job vectors die with this frame
        return FAL_OK;
    }
    for (const DenseJob& j : coarse) ctx->counters[1] += (int64_t)j.nq * j.nc;

    // ---- B. IVF buckets: coarse quantiser ------------------------------------------------------
    const int np = std::min(n_probe, max_n_list);
    int32_t* probes = nullptr;
    float* probe_sim = nullptr;
    int64_t *totals = nullptr, *q_sim_off = nullptr;
    FAL_TRY(ctx->reserve(SLOT_PROBES, sizeof(int32_t) * (size_t)ivf->n * np, (void**)&probes));
    FAL_TRY(ctx->reserve(SLOT_PROBE_SIM, sizeof(float) * (size_t)ivf->n * np, (void**)&probe_sim));
    FAL_TRY(ctx->reserve(SLOT_QOFF, sizeof(int64_t) * (size_t)(ivf->n + 1), (void**)&q_sim_off));
    FAL_TRY(ctx->reserve(SLOT_MISC, sizeof(int64_t) * (size_t)(ivf->n + 1), (void**)&totals));
    FAL_CHECK_HIP(hipMemsetAsync(totals, 0, sizeof(int64_t) * (size_t)(ivf->n + 1), st));
    for (size_t bi = 0; bi + 1 < coarse_cuts.size(); ++bi) {
        const int64_t t0 = coarse_cuts[bi], t1 = coarse_cuts[bi + 1];
        const int64_t base = obase_of_tile(coarse, t0);
        FAL_TRY(launch_dense(ctx, ST_COARSE, EPI_STORE, ivf->Xl, ivf->centroids, d, coarse_dev, (int)coarse.size(), t0,
                             t1 - t0, sims, base, nullptr));
        SelectArgs sa{};
        sa.sims = sims; sa.sims_base = base; sa.k = np; sa.out_sim = probe_sim; sa.out_idx = probes;
        sa.jobs = coarse_dev; sa.n_jobs = (int)coarse.size(); sa.tile_begin = t0; sa.ids_are_rows = 0;
        FAL_TRY(launch_select(ctx, ST_COARSE, MODE_DENSE, sa, (t1 - t0) * 32));
    }
    {
        StageScope ts(ctx, ST_COARSE);
        hipLaunchKernelGGL(probe_totals_kernel, dim3((unsigned)ceil_div(ivf_tiles * 32, 256)), dim3(256), 0, st, probes, np,
                           coarse_dev, (int)coarse.size(), ivf_tiles, ivf->list_off, totals);
        FAL_TRY(launch_exclusive_scan(ctx, totals, ivf->n, q_sim_off));
    }
    // candidate prefix back to the host to cut the fine scan into buffer-sized batches
    std::vector<int64_t> qoff((size_t)ivf->n + 1);
    FAL_CHECK_HIP(hipMemcpyAsync(qoff.data(), q_sim_off, sizeof(int64_t) * qoff.size(), hipMemcpyDeviceToHost, st));
    FAL_CHECK_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> fine_cuts(1, 0);
    size_t need_fine = 0;
    {
        int64_t batch_first_p = -1, used = 0;
        for (const DenseJob& j : coarse) {
            const int64_t tiles = ceil_div(j.nq, 32);
            for (int64_t t = 0; t < tiles; ++t) {
                const int64_t pa = j.q_row0 + 32 * t, pb = std::min<int64_t>(pa + 32, j.q_row0 + j.nq);
                const int64_t sz = qoff[pb] - qoff[pa];
                if (batch_first_p < 0) batch_first_p = pa;
                // a batch's sims span [qoff[first p], qoff[last p]) -- flat rows in between add nothing
                const int64_t span = qoff[pb] - qoff[batch_first_p];
                if (span > (int64_t)cap && used > 0) {
                    fine_cuts.push_back(j.tile0 + t);
                    batch_first_p = pa;
                    used = 0;
                }
                used += sz;
                need_fine = std::max(need_fine, (size_t)(qoff[pb] - qoff[batch_first_p]));
            }
        }
        if (fine_cuts.back() != ivf_tiles) fine_cuts.push_back(ivf_tiles);
    }
    ctx->counters[0] += qoff[(size_t)ivf->n];
    ctx->counters[2] += (int64_t)fine_cuts.size() - 1;
    ctx->counters[3] = std::max<int64_t>(ctx->counters[3], (int64_t)(sizeof(float) * need_fine));
    FAL_TRY(ctx->reserve(SLOT_SIMS, sizeof(float) * std::max<size_t>(need_fine, 16), (void**)&sims));
    auto first_p_of_tile = [&](int64_t t) -> int64_t {
        size_t lo = 0, hi = coarse.size() - 1;
        while (lo < hi) {
            size_t mid = (lo + hi + 1) / 2;
            if (coarse[mid].tile0 <= t) lo = mid; else hi = mid - 1;
        }
        return coarse[lo].q_row0 + 32 * (t - coarse[lo].tile0);
    };
    for (size_t bi = 0; bi + 1 < fine_cuts.size(); ++bi) {
        const int64_t t0 = fine_cuts[bi], t1 = fine_cuts[bi + 1];
        const int64_t base = qoff[first_p_of_tile(t0)];
        FineArgs fa{};
        fa.Xl = ivf->Xl; fa.d = d; fa.jobs = coarse_dev; fa.n_jobs = (int)coarse.size();
        fa.tile_begin = t0; fa.n_tiles = t1 - t0; fa.n_probe = np; fa.probes = probes;
        fa.list_off = ivf->list_off; fa.q_sim_off = q_sim_off; fa.sims = sims; fa.sims_base = base;
        fa.bm_words = (max_n_list + 31) / 32;
        fa.u_cap = std::min(32 * np, max_n_list);
        FAL_TRY(launch_fine(ctx, fa));
        SelectArgs sa{};
        sa.sims = sims; sa.sims_base = base; sa.k = k_ann; sa.out_sim = sim; sa.out_idx = idx;
        sa.jobs = coarse_dev; sa.n_jobs = (int)coarse.size(); sa.tile_begin = t0;
        sa.n_probe = np; sa.probes = probes; sa.list_off = ivf->list_off; sa.q_sim_off = q_sim_off; sa.perm = ivf->perm;
        FAL_TRY(launch_select(ctx, ST_SELECT, MODE_IVF, sa, (t1 - t0) * 32));
    }
    FAL_CHECK_HIP(hipStreamSynchronize(st));
    return FAL_OK;
}
