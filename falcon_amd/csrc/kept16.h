// kept16: which window candidates of a query stay in the race (shared by kept16_kernel, pairs16.hip, and the fused tail of
// select16_kernel, ivf16.hip).
#pragma once
#include "common.h"
#include "simtile.h"
#include "fused.h"
#include "ivf16.h"

namespace fal {

// 16 lanes per query, one lane per probed list (two rounds for 32 probes ...).  Inside a list the rows are sorted by
// precursor m/z, so the part of the list inside the query's (slightly widened) precursor window is a contiguous range of list
// positions -- two binary searches -- and its keys are contiguous in the query's key stream.  A row stays if its key is not
// certainly below the k-th best (`sel` = select16_kernel's thresholds: x = smallest key + 1 a window candidate must reach, y =
// largest key + 1 that is still ambiguous) and the exact tolerance tests pass.
// Every lane of a 16-lane group passes the SAME query (live, p = its list-order position, row = its sorted row, lbase = global id
// of its bucket's list 0, krow = its key stream); groups of a wave are independent (a group with live = false does nothing).
__device__ __forceinline__ void kept16_query(const Kept16Args& a, bool live, int64_t p, int64_t row, int64_t lbase,
                                             const uint16_t* krow, int2 sel, int tid) {
    const int sub = tid & 15, lane = tid & 63, sh = 16 * (lane >> 4);
    const int np = a.n_probe;
    const float qmz = a.pmz_l[p];
    const bool use_rt = a.rt != nullptr && a.rt_tol >= 0.0;
    const float qrt = use_rt ? a.rt[row] : 0.f;
    const float tol_f = a.tol_f, rt_f = a.rt_f;
    float lob, hib;                                               // conservative float32 bounds of the window (exact tests below)
    {
        const double q = (double)qmz;
        double lo, hi;
        if (a.is_da) {
            lo = q - a.tol - 1e-3;
            hi = q + a.tol + 1e-3;
        } else {
            const double tt = a.tol * 1e-6;
            lo = q * (1.0 - 1.01 * tt - 2e-6);
            hi = tt < 0.5 ? q * (1.0 + 1.01 * tt / (1.0 - tt) + 2e-6) : INFINITY;
        }
        lob = (float)lo;
        lob = (double)lob > lo ? __uint_as_float(__float_as_uint(lob) - 1u) : lob;       // round down (positive values)
        hib = (float)hi;
        hib = (double)hib < hi ? __uint_as_float(__float_as_uint(hib) + 1u) : hib;       // round up
    }
    uint32_t* gk = a.gkept_id + row * FAL_FUSED_KEEP;
    const int32_t* pr = a.probes + p * np;
    int kc = 0, run = 0;
    bool amb = false;
    for (int j0 = 0; j0 < np; j0 += 16) {
        const int j = j0 + sub;
        const int32_t l = (live && j < np) ? pr[j] : -1;
        int64_t b = 0, e = 0;
        if (l >= 0) {
            b = a.list_off[lbase + l];
            e = a.list_off[lbase + l + 1];
        }
        const int len = (int)(e - b);
        const int incl = row16_prefix_sum(len);                  // (DPP inside the query's 16 lanes: no LDS round trips)
        const int seg = run + incl - len;                        // where this list's keys start in the query's stream
        run += row16_sum(len);
        // first position with pmz >= lob
        int64_t lo = b, hi = e;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.pmz_l[mid] < lob) lo = mid + 1; else hi = mid;
        }
        const int64_t wa = lo;
        // first position above the window.  The window holds a few rows (tens of ppm of a bucket's m/z range), so the end is
        // looked for from its start -- rows 0, 1, 3, 7, ... of the remainder, then a binary search inside the last doubling:
        // 1-4 loads instead of log2(len); the kernel's pace is set by the cache lines its gathers touch
        {
            const int64_t rem = e - wa;
            int64_t bound = 1;                                   // rows [wa, wa + bound / 2) are known to be inside
            while (bound - 1 < rem && a.pmz_l[wa + bound - 1] <= hib) bound <<= 1;
            lo = wa + (bound >> 1);
            hi = wa + (bound - 1 < rem ? bound - 1 : rem);
        }
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.pmz_l[mid] <= hib) lo = mid + 1; else hi = mid;
        }
        // The window rows of the group's 16 lists, FLATTENED over its lanes: a lane per row, 16 rows per step (a lane per LIST
        // walked max-window-length steps with a third of the lanes busy -- the walk was two thirds of this kernel).  Row t of
        // the group belongs to the list whose exclusive prefix of window lengths is the last one <= t.
        const int wlen = (int)(lo - wa);
        const int wincl = row16_prefix_sum(wlen);
        const int wtotal = row16_sum(wlen);
        const int wexcl = wincl - wlen;
        const int wa_lo = (int)(uint32_t)wa, wa_hi = (int)(wa >> 32);
        const int rel = (int)(wa - b) + seg;                     // key-stream position of the window's first row
        for (int t0 = 0; t0 < wtotal; t0 += 16) {                // (uniform inside the group; groups of a wave may differ)
            const int t = t0 + sub;
            int L = 0;
#pragma unroll
            for (int s = 8; s >= 1; s >>= 1) {
                const int cand = L + s;
                const int ex = __shfl(wexcl, min(cand, 15), 16);
                L = (cand < 16 && ex <= t) ? cand : L;
            }
            const int o = t - __shfl(wexcl, L, 16);             // offset inside list L's window
            const int64_t c = (((int64_t)__shfl(wa_hi, L, 16) << 32) | (uint32_t)__shfl(wa_lo, L, 16)) + o;
            const int kpos = __shfl(rel, L, 16) + o;
            const bool in = t < wtotal;
            bool ok = in && c != p;
            int u = 0;
            uint32_t id = 0;
            if (ok) {                                            // precursor, key and row: three loads in flight
                const float nmz = a.pmz_l[c];
                u = (int)krow[kpos] + 1;
                id = (uint32_t)a.perm[c];
                const float diff = qmz - nmz;                    // mass_diff(query, neighbour): the arithmetic of filter_kernel
                const float xx = a.is_da ? diff : diff / nmz;
                ok = u >= sel.x && fabsf(xx) <= tol_f;
            }
            if (ok && use_rt) ok = fabsf(qrt - a.rt[id]) <= rt_f;
            amb = amb || (ok && u <= sel.y);
            const uint32_t gm = (uint32_t)(__ballot(ok) >> sh) & 0xFFFFu;
            if (ok) {
                const int at = kc + __popc(gm & ((1u << sub) - 1u));
                if (at < FAL_FUSED_KEEP) gk[at] = id;
            }
            kc += __popc(gm);
        }
    }
    const bool q_amb = ((uint32_t)(__ballot(amb) >> sh) & 0xFFFFu) != 0u;
    if (live && sub == 0) {
        a.gkcnt[row * 2] = min(kc, FAL_FUSED_KEEP / 2) | (q_amb ? 0x100 : 0) | (kc > FAL_FUSED_KEEP ? 0x200 : 0);
        a.gkcnt[row * 2 + 1] = min(max(kc - FAL_FUSED_KEEP / 2, 0), FAL_FUSED_KEEP / 2);
    }
}

}  // namespace fal
