/* CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/falcon_oracle.py).
 *
 * float32 inner products summed in the order the HIP cosine kernels use
 * (falcon_amd/csrc/simtile.h): with dh = d / 2,
 *
 *     acc = 0
 *     for kk in 0 .. dh-1:
 *         acc = fmaf(a[kk],      b[kk],      acc)
 *         acc = fmaf(a[dh + kk], b[dh + kk], acc)
 *
 * i.e. MFMA step kk of v_mfma_f32_32x32x2_f32 multiplies k-slot 0 = k (lanes 0-31) and
 * k-slot 1 = dh + k (lanes 32-63) and accumulates them in slot order with one rounding per
 * fused multiply-add.  fmaf() is correctly rounded, so this reproduces the kernels bit for
 * bit (tests/test_gpu_search.py, tests/test_gpu_pipeline.py assert exact equality).
 *
 * The reference snapshot has no code for this stage (Faiss IndexIVFFlat inner product,
 * README.md:132-142; setup.cfg:25): the summation order is the build's own convention.
 *
 * Built by __graft_entry__.build() -> oracle/_build/libkordered.so  (gcc -O3 -mavx2 -mfma -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#define JT 64 /* outputs per register tile: independent chains hide the fma latency */

/* threads per call: 0 = all (OpenMP default); the all-cores CPU baseline of bench.py runs one bucket per pool
 * thread and sets this to 1 */
static int g_threads = 0;
void fo_set_threads(int n) { g_threads = n; }
static int nthreads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

/* out[i * nb + j] = <A[i], B[j]> in the kernel's k order.  A: [na, d], B: [nb, d], row-major. */
int fo_sims_kordered(const float* A, int64_t na, const float* B, int64_t nb, int d, float* out) {
    if (d <= 0 || (d & 1)) return -1;
    const int dh = d / 2;
    const int64_t nbp = (nb + JT - 1) / JT * JT;
    float* Bt = (float*)calloc((size_t)d * (size_t)nbp + 1, sizeof(float)); /* Bt[k][j] */
    if (!Bt) return -2;
    for (int64_t j = 0; j < nb; ++j)
        for (int k = 0; k < d; ++k) Bt[(size_t)k * nbp + j] = B[(size_t)j * d + k];
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads())
    for (int64_t i = 0; i < na; ++i) {
        const float* a = A + (size_t)i * d;
        for (int64_t j0 = 0; j0 < nbp; j0 += JT) {
            float acc[JT];
            for (int j = 0; j < JT; ++j) acc[j] = 0.0f;
            for (int kk = 0; kk < dh; ++kk) {
                const float a0 = a[kk], a1 = a[dh + kk];
                const float* b0 = Bt + (size_t)kk * nbp + j0;
                const float* b1 = Bt + (size_t)(dh + kk) * nbp + j0;
                for (int j = 0; j < JT; ++j) acc[j] = fmaf(a0, b0[j], acc[j]);
                for (int j = 0; j < JT; ++j) acc[j] = fmaf(a1, b1[j], acc[j]);
            }
            const int64_t lim = nb - j0 < JT ? nb - j0 : JT;
            for (int64_t j = 0; j < lim; ++j) out[(size_t)i * nb + j0 + j] = acc[j];
        }
    }
    free(Bt);
    return 0;
}

/* arg-max over j of <A[i], B[j]> (ties -> lowest j): k-means / final list assignment. */
int fo_argmax_kordered(const float* A, int64_t na, const float* B, int64_t nb, int d, int32_t* out) {
    if (d <= 0 || (d & 1) || nb <= 0) return -1;
    const int dh = d / 2;
    const int64_t nbp = (nb + JT - 1) / JT * JT;
    float* Bt = (float*)calloc((size_t)d * (size_t)nbp + 1, sizeof(float));
    if (!Bt) return -2;
    for (int64_t j = 0; j < nb; ++j)
        for (int k = 0; k < d; ++k) Bt[(size_t)k * nbp + j] = B[(size_t)j * d + k];
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads())
    for (int64_t i = 0; i < na; ++i) {
        const float* a = A + (size_t)i * d;
        float best = -INFINITY;
        int32_t bj = 0;
        for (int64_t j0 = 0; j0 < nbp; j0 += JT) {
            float acc[JT];
            for (int j = 0; j < JT; ++j) acc[j] = 0.0f;
            for (int kk = 0; kk < dh; ++kk) {
                const float a0 = a[kk], a1 = a[dh + kk];
                const float* b0 = Bt + (size_t)kk * nbp + j0;
                const float* b1 = Bt + (size_t)(dh + kk) * nbp + j0;
                for (int j = 0; j < JT; ++j) acc[j] = fmaf(a0, b0[j], acc[j]);
                for (int j = 0; j < JT; ++j) acc[j] = fmaf(a1, b1[j], acc[j]);
            }
            const int64_t lim = nb - j0 < JT ? nb - j0 : JT;
            for (int64_t j = 0; j < lim; ++j)
                if (acc[j] > best) { best = acc[j]; bj = (int32_t)(j0 + j); }
        }
        out[i] = bj;
    }
    free(Bt);
    return 0;
}

/* k best of one row by (sim descending, id ascending): partial selection sort over a heap-free
 * scan -- rows are short (<= a few thousand candidates).  out_* hold k entries (pad: -inf / -1). */
static void topk_row(const float* s, const int64_t* ids, int64_t n, int k, float* out_s, int64_t* out_i) {
    /* insertion into a sorted window of size k */
    int m = 0;
    for (int64_t c = 0; c < n; ++c) {
        const float v = s[c];
        const int64_t id = ids ? ids[c] : c;
        if (m == k) {
            const float lv = out_s[k - 1];
            const int64_t li = out_i[k - 1];
            if (!(v > lv || (v == lv && id < li))) continue;
        }
        int p = m < k ? m : k - 1;
        while (p > 0 && (out_s[p - 1] < v || (out_s[p - 1] == v && out_i[p - 1] > id))) {
            out_s[p] = out_s[p - 1];
            out_i[p] = out_i[p - 1];
            --p;
        }
        out_s[p] = v;
        out_i[p] = id;
        if (m < k) ++m;
    }
    for (int p = m; p < k; ++p) { out_s[p] = -INFINITY; out_i[p] = -1; }
}

/* n_probe search of every row of one bucket against the bucket's own index (oracle ivf_search):
 * X [n, d]; probes [n, np] list ids (-1 = none); perm/off = inverted lists (rows of X);
 * sim/idx [n, k] out, ids = base + row. */
int fo_ivf_search(const float* X, int64_t n, int d, const int32_t* probes, int np, const int64_t* perm,
                  const int64_t* off, int k, int64_t base, float* sim, int32_t* idx) {
    if (d <= 0 || (d & 1)) return -1;
    const int dh = d / 2;
    int rc = 0;
#pragma omp parallel num_threads(nthreads())
    {
        int64_t cap = 1024;
        float* s = (float*)malloc(sizeof(float) * cap);
        int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * cap);
        float* os = (float*)malloc(sizeof(float) * k);
        int64_t* oi = (int64_t*)malloc(sizeof(int64_t) * k);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < n; ++i) {
            int64_t m = 0;
            for (int p = 0; p < np; ++p) {
                const int32_t l = probes[i * np + p];
                if (l < 0) continue;
                for (int64_t e = off[l]; e < off[l + 1]; ++e) {
                    if (m == cap) {
                        cap *= 2;
                        s = (float*)realloc(s, sizeof(float) * cap);
                        ids = (int64_t*)realloc(ids, sizeof(int64_t) * cap);
                    }
                    ids[m++] = perm[e];
                }
            }
            const float* q = X + (size_t)i * d;
            for (int64_t c = 0; c < m; ++c) {
                const float* b = X + (size_t)ids[c] * d;
                float acc = 0.0f;
                for (int kk = 0; kk < dh; ++kk) {
                    acc = fmaf(q[kk], b[kk], acc);
                    acc = fmaf(q[dh + kk], b[dh + kk], acc);
                }
                s[c] = acc;
            }
            topk_row(s, ids, m, k, os, oi);
            for (int p = 0; p < k; ++p) {
                sim[i * k + p] = os[p];
                idx[i * k + p] = oi[p] < 0 ? -1 : (int32_t)(oi[p] + base);
            }
        }
        free(s); free(ids); free(os); free(oi);
    }
    return rc;
}

/* row-wise top-k of a dense [n, m] similarity matrix, ids = column (coarse quantiser, exhaustive search) */
int fo_topk_rows(const float* S, int64_t n, int64_t m, int k, float* sim, int32_t* idx) {
#pragma omp parallel num_threads(nthreads())
    {
        float* os = (float*)malloc(sizeof(float) * k);
        int64_t* oi = (int64_t*)malloc(sizeof(int64_t) * k);
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < n; ++i) {
            topk_row(S + (size_t)i * m, 0, m, k, os, oi);
            for (int p = 0; p < k; ++p) {
                sim[i * k + p] = os[p];
                idx[i * k + p] = (int32_t)oi[p];
            }
        }
        free(os); free(oi);
    }
    return 0;
}

/* Probe helper (tools/probe_mfma_order.py): the same inner product under other candidate summation
 * orders, to establish on the hardware which one v_mfma_f32_32x32x2_f32 implements.
 *   mode 0: fma(a1*b1, fma(a0*b0, acc))            (k-slot 0 then 1: the order fo_sims_kordered uses)
 *   mode 1: fma(a0*b0, fma(a1*b1, acc))            (k-slot 1 then 0)
 *   mode 2: acc + fma(a0, b0, a1*b1)               (pair summed first, then added)
 *   mode 3: acc + fma(a1, b1, a0*b0)
 *   mode 4: plain k = 0..d-1 sequential fma        (no half interleave)
 */
int fo_sims_mode(const float* A, int64_t na, const float* B, int64_t nb, int d, int mode, float* out) {
    const int dh = d / 2;
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads())
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j) {
            const float* a = A + (size_t)i * d;
            const float* b = B + (size_t)j * d;
            float acc = 0.0f;
            if (mode == 4) {
                for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
            } else {
                for (int kk = 0; kk < dh; ++kk) {
                    const float a0 = a[kk], b0 = b[kk], a1 = a[dh + kk], b1 = b[dh + kk];
                    if (mode == 0) acc = fmaf(a1, b1, fmaf(a0, b0, acc));
                    else if (mode == 1) acc = fmaf(a0, b0, fmaf(a1, b1, acc));
                    else if (mode == 2) { volatile float p = a1 * b1; float t = fmaf(a0, b0, p); volatile float s = acc + t; acc = s; }
                    else { volatile float p = a0 * b0; float t = fmaf(a1, b1, p); volatile float s = acc + t; acc = s; }
                }
            }
            out[(size_t)i * nb + j] = acc;
        }
    return 0;
}
