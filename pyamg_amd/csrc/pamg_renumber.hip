// pamg_renumber.hip -- host-side renumbering of the unknowns of an INTERIOR level (no device code in this file).
//
// The unknowns of level l >= 1 are the solver's own: the caller never sees a level-l vector.  The aggregation numbers them along the
// fine rows, so the neighbours of a coarse unknown in the other two grid directions sit a plane of aggregates away and a row range of
// A_l gathers x through ~1 000 distinct columns per 1 536 entries.  Numbering the unknowns blob by blob (the blobs being the hierarchy's
// OWN aggregates of the next levels: pyamg_amd/hierarchy.py renumber_levels) brings that to ~300 and A_1's product from 0.209 to
// 0.161 ms on the 256^3 hierarchy (profiles/r06_microbench_renumber_*.json).
//
// Arithmetic: rows are moved and columns renamed; the entries of a row keep their STORED order, so every row sum adds the same products
// in the same order as the reference's loop over the original operator (amg_core/linalg.h / scipy's csr_matvec) -- results are the
// original results, permuted, bit for bit.
#include <cstring>

#include "pamg_common.h"

using namespace pamg;

extern "C" {

// B = rows of A in the order row_old_of_new (row i of B = row row_old_of_new[i] of A; NULL = unchanged), columns renamed through
// col_new_of_old (NULL = unchanged).  Bp [nrows + 1], Bj / Bx [nnz] are the caller's.
int pamg_csr_renumber(int dtype, int64_t nrows, int64_t ncols, const int32_t *Ap, const int32_t *Aj, const void *Ax, const int32_t *row_old_of_new,
                      const int32_t *col_new_of_old, int32_t *Bp, int32_t *Bj, void *Bx)
{
    if (nrows < 0 || ncols < 0 || !Ap || !Bp || (dtype != PAMG_F64 && dtype != PAMG_F32)) return PAMG_E_ARG;
    const int64_t nnz = Ap[nrows];
    if (nnz < 0 || (nnz > 0 && (!Aj || !Ax || !Bj || !Bx))) return PAMG_E_ARG;
    const size_t ts = tsize(dtype);
    Bp[0] = 0;
    for (int64_t i = 0; i < nrows; ++i) {
        const int64_t r = row_old_of_new ? row_old_of_new[i] : i;
        if (r < 0 || r >= nrows) return PAMG_E_ARG;
        Bp[i + 1] = Bp[i] + (Ap[r + 1] - Ap[r]);
    }
    if (Bp[nrows] != nnz) return PAMG_E_ARG;                 // row_old_of_new is not a permutation
    int bad = 0;
    host_parallel(nrows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t r = row_old_of_new ? row_old_of_new[i] : i;
            const int64_t s = Ap[r], len = Ap[r + 1] - s, d = Bp[i];
            memcpy((char *)Bx + (size_t)d * ts, (const char *)Ax + (size_t)s * ts, (size_t)len * ts);
            if (col_new_of_old) {
                for (int64_t k = 0; k < len; ++k) {
                    const int32_t c = Aj[s + k];
                    if (c < 0 || c >= ncols) { __atomic_store_n(&bad, 1, __ATOMIC_RELAXED); Bj[d + k] = 0; continue; }
                    Bj[d + k] = col_new_of_old[c];
                }
            } else {
                memcpy(Bj + d, Aj + s, (size_t)len * sizeof(int32_t));
            }
        }
    }, 1 << 16);
    return bad ? PAMG_E_ARG : PAMG_OK;
}

// out[i] = the column of the entry of largest magnitude of row i (the first one on ties; -1 for an empty row): for a smoothed-aggregation
// prolongator this is the aggregate the unknown belongs to, for a classical one the strongest C point
int pamg_csr_row_argmax_abs(int dtype, int64_t nrows, const int32_t *Ap, const int32_t *Aj, const void *Ax, int32_t *out)
{
    if (nrows < 0 || !Ap || !out || (dtype != PAMG_F64 && dtype != PAMG_F32)) return PAMG_E_ARG;
    if (Ap[nrows] > 0 && (!Aj || !Ax)) return PAMG_E_ARG;
    host_parallel(nrows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            double best = -1.0;
            int32_t at = -1;
            for (int64_t p = Ap[i]; p < Ap[i + 1]; ++p) {
                const double v = dtype == PAMG_F64 ? ((const double *)Ax)[p] : (double)((const float *)Ax)[p];
                const double m = v < 0 ? -v : v;
                if (m > best) { best = m; at = Aj[p]; }
            }
            out[i] = at;
        }
    }, 1 << 16);
    return PAMG_OK;
}

// host threads the library's planners count on: hardware threads, affinity mask and cgroup quota (csrc/pamg_host_threads.h).  fresh != 0 evaluates
// the environment again (tests); 0 returns what the planners use (evaluated once per process)
int pamg_host_cpus(int fresh) { return (int)(fresh ? host_cpus_now() : host_cpus()); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// Rows of a CSR / BSR operator sorted by column, in place, on the host threads: what scipy's sort_indices() does (the reference calls
// it on every Galerkin product before it reads the diagonal, util/utils.py:583 -- SciPy's SpGEMM leaves the rows in first-touch
// order), 63 M entries in 0.13 s on one core there.  Columns inside a row are distinct after a product, so the result is unique.
// block = R * C values per stored entry (1 for CSR).
#include <algorithm>
#include <numeric>
#include <vector>

extern "C" int pamg_csr_sort_rows(int dtype, int64_t nrows, const int32_t *Ap, int32_t *Aj, void *Ax, int block)
{
    if (nrows < 0 || !Ap || block < 1 || (dtype != PAMG_F64 && dtype != PAMG_F32)) return PAMG_E_ARG;
    if (Ap[nrows] > 0 && (!Aj || !Ax)) return PAMG_E_ARG;
    const size_t eb = tsize(dtype) * (size_t)block;
    host_parallel(nrows, [&](int64_t lo, int64_t hi) {
        std::vector<int> perm;
        std::vector<int32_t> cj;
        std::vector<unsigned char> cx;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t s = Ap[i];
            const int len = (int)(Ap[i + 1] - s);
            if (len < 2 || std::is_sorted(Aj + s, Aj + s + len)) continue;
            perm.resize((size_t)len);
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return Aj[s + a] < Aj[s + b]; });
            cj.assign(Aj + s, Aj + s + len);
            cx.assign((const unsigned char *)Ax + (size_t)s * eb, (const unsigned char *)Ax + (size_t)(s + len) * eb);
            for (int k = 0; k < len; ++k) {
                Aj[s + k] = cj[(size_t)perm[(size_t)k]];
                memcpy((unsigned char *)Ax + (size_t)(s + k) * eb, cx.data() + (size_t)perm[(size_t)k] * eb, eb);
            }
        }
    }, 1 << 14);
    return PAMG_OK;
}
