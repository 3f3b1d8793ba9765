// The index handle behind the opaque `fal_ivf` of the C ABI.
#pragma once
#include <vector>
#include "common.h"

namespace fal {
// the sparse form of a row (fal_ivf::sp_cols / sp_vals)
constexpr int kSparseW = 64;                 // entries per row (one per lane)
constexpr uint16_t kColPad = 0xFFFF;         // unused entry
constexpr uint16_t kColDense = 0xFFFE;       // in entry 0: the row has more than kSparseW non-zeros, read the dense row

// context scratch slots
enum { SLOT_JOBS = 0, SLOT_SIMS = 1, SLOT_PROBES = 2, SLOT_PROBE_SIM = 3, SLOT_QOFF = 4, SLOT_JOBS2 = 5,
       SLOT_MISC = 6, SLOT_MISC2 = 7, SLOT_SORT = 8, SLOT_SORT2 = 9, SLOT_TAIL = 10, SLOT_TAIL2 = 11,
       SLOT_TAIL3 = 12, SLOT_TAIL4 = 13, SLOT_DB = 14, SLOT_DB2 = 15, SLOT_DB3 = 16, SLOT_FIN = 17, SLOT_FIN2 = 18,
       SLOT_FIN3 = 19, SLOT_INV = 20, SLOT_INVCNT = 21, SLOT_SIMS2 = 22, SLOT_TILEJOB = 23, SLOT_FUSED = 24, SLOT_FUSED2 = 25, SLOT_FUSED3 = 26, SLOT_JOBS3 = 27, SLOT_WIN = 28, SLOT_CURSORS = 29, SLOT_ITEMS = 30 };
// out[0..n] = exclusive prefix sums of in[0..n) (out has n + 1 entries)
int launch_exclusive_scan(fal_ctx* ctx, const int64_t* in, int64_t n, int64_t* out);
}
struct fal_ivf;
namespace fal {
int ivf_ensure_xl(fal_ctx* ctx, const fal_ivf* ivf);      // Xl = X[perm] (float32 rows in list order), made once on demand
}

struct fal_ivf {
    fal_ctx* ctx = nullptr;
    int64_t n = 0;
    int d = 0;
    const float* X = nullptr;        // caller's vectors, precursor-sorted rows (borrowed)
    const float* Xl = nullptr;       // the same rows in (bucket, list, row) order (== X when all flat; else made on demand: ivf_ensure_xl)
    float* Xl_owned = nullptr;
    const void* X16 = nullptr;       // optional f16 rows (sorted order) for the flat scan: [n, planes, d]
    int x16_planes = 0;
    const void* Xpre = nullptr;      // optional float16 copy of X used ONLY as the prefilter of the fused flat scan
    // The float16 prefilters are exact only for rows without negative (or non-finite) components (their error bound is relative
    // to the similarity).  1 = some row of an IVF bucket has one (found by the build's pass over the rows): the build and the
    // searches use the exact kernels; 0 = none; -1 = not read back yet (neg_dev holds the flag on the device)
    int rows_signed = 0;
    int rows_f16 = 0;                // 1 = every component of the indexed rows is a float16 value (float16 vectors: X is the image of X16)
    int32_t* neg_dev = nullptr;
    const void* X16pre = nullptr;    // float16 rows (sorted order) the IVF fine scan gathers its list rows and queries from
                                     // (ivf16.hip), borrowed: fal_ivf_attach_prefilter_ex(which & 2)
    int32_t* pos_of_row = nullptr;   // [n] sorted row -> list-order position (with X16pre), owned
    int ckeys_stride = 0;            // columns of ckeys: 128 x (groups of the bucket with the most lists)
    uint16_t* ckeys = nullptr;       // [n, ckeys_stride] approximate (row, centroid) similarities of the final k-means pass as 16-bit keys,
                                     // by sorted row: the coarse quantiser reads them instead of scanning again; only when EVERY
                                     // indexed bucket went through the float16 assignment (<= 512 lists), owned
    std::vector<int64_t> bucket_off; // host, n_buckets + 1
    std::vector<int32_t> n_list;     // host, per bucket
    std::vector<int64_t> list_base;  // host, global id of each bucket's list 0
    int64_t total_lists = 0;
    int n_ivf_buckets = 0;
    int64_t ivf_waves = 0;
    float* centroids = nullptr;      // [total_lists, d]
    int32_t* assign = nullptr;       // [n] bucket-local list of each sorted row
    int32_t* perm = nullptr;         // [n] list-order position -> sorted row
    // the rows of the IVF buckets in sparse form, [n, 64] each (entry = column, value; in the order of the exact similarity chains;
    // column 0xFFFF = unused entry, 0xFFFE in entry 0 = more than 64 non-zeros, use the dense row); owned
    uint16_t* sp_cols = nullptr;
    float* sp_vals = nullptr;
    // the same rows as the 256-byte query records of list16s_kernel: [n][64 x u16 column | 64 x f16 value] (made when the build was
    // given float16 rows); owned.  rows_many = 1: some row has more than 64 non-zeros -- the dense-gather scan then serves the index
    uint16_t* sq16 = nullptr;
    int rows_many = 0;
    int64_t* list_off = nullptr;     // [total_lists + 1] list-order positions
    int64_t* counts = nullptr;       // [total_lists + 1] list sizes (flat buckets)
    void* bk_dev = nullptr;          // BucketDev[n_ivf_buckets]
};
