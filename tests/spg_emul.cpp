// TEST INFRASTRUCTURE (CPU): replays the order-exact sparse product of pyamg_amd/csrc/pamg_setup.hip task by task, the
// way spg_kernel / spg_long_kernel / spg_reorder_kernel consume the host plan (csrc/pamg_spg_plan.h): keys, batches of a
// long row cut by the prefix of in-window products, per-column accumulators continued across batches, first-touch
// sequence numbers, SciPy's emission order (reverse first touch; forward and whole blocks in true-block mode), exact
// zeros squeezed out.  tests/test_setup_host.py compares its output with SciPy's `A @ B`, array for array, without a GPU.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../pyamg_amd/csrc/pamg_spg_plan.h"

using namespace pamg;

namespace {
constexpr int BLK = 256;
struct Out { int col; double val; int seq; };
}

extern "C" {

// returns 0, or 2 = a row of B longer than SPL_CAP meets a long row (PAMG_E_UNSUPPORTED on the device).
// Cp: [m + 1]; Cj / Cx: capacity cap (>= the number of products is always enough); stats: [0] tasks, [1] long rows,
// [2] window tasks, [3] exact zeros squeezed out
int spg_emul_f64(int m, int n, const int *Ap, const int *Aj, const double *Ax, const int *Bp, const int *Bj, const double *Bx,
                 int col_block, int keep, int *Cp, int *Cj, double *Cx, int64_t cap, int64_t *stats)
{
    (void)n;
    std::vector<int> nprod((size_t)m), lohi;
    int maxlen = 0;
    bool any_long = false;
    for (int i = 0; i < m; ++i) {
        long long c = 0;
        int lo = INT_MAX, hi = -1;
        for (int e = Ap[i]; e < Ap[i + 1]; ++e) {
            const int k = Aj[e];
            c += Bp[k + 1] - Bp[k];
            for (int p = Bp[k]; p < Bp[k + 1]; ++p) { lo = std::min(lo, Bj[p]); hi = std::max(hi, Bj[p]); }
        }
        nprod[(size_t)i] = (int)std::min<long long>(c, INT_MAX);
        if (nprod[(size_t)i] > SPG_CAP2) { lohi.push_back(lo); lohi.push_back(hi); any_long = true; }
    }
    std::vector<SpgTask> tasks;
    spg_plan(m, nprod, lohi, tasks);
    if (any_long) {
        // the device checks B's longest row against the batch capacity before it plans windows
        int rowsB = 0;
        for (int i = 0; i < m; ++i) for (int e = Ap[i]; e < Ap[i + 1]; ++e) rowsB = std::max(rowsB, Aj[e] + 1);
        for (int k = 0; k < rowsB; ++k) maxlen = std::max(maxlen, Bp[k + 1] - Bp[k]);
        if (maxlen > SPL_CAP) return 2;
    }
    std::vector<std::vector<Out>> rows((size_t)m);
    std::vector<char> is_long((size_t)m, 0);
    int64_t nwin = 0, nlong = 0;
    for (const SpgTask &t : tasks) {
        if (spg_whole(t)) {
            // ---- spg_kernel
            std::vector<unsigned long long> K;
            std::vector<double> V;
            unsigned seq = 0;
            for (int r = t.row0; r < t.row1; ++r)
                for (int e = Ap[r]; e < Ap[r + 1]; ++e) {
                    const int k = Aj[e];
                    for (int p = Bp[k]; p < Bp[k + 1]; ++p, ++seq) {
                        K.push_back(((unsigned long long)(r - t.row0) << 53) | ((unsigned long long)(unsigned)Bj[p] << 22) | seq);
                        V.push_back(Ax[e] * Bx[p]);
                    }
                }
            if (seq > (unsigned)(t.row1 - t.row0 == 1 ? SPG_CAP2 : SPG_CAP)) return 3;
            std::vector<int> idx(K.size());
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return K[(size_t)a] < K[(size_t)b]; });
            std::vector<unsigned long long> K2;
            std::vector<double> V2;
            for (size_t p = 0; p < idx.size();) {
                const unsigned long long g = K[(size_t)idx[p]] >> 22;
                double s = 0.0;
                size_t q = p;
                while (q < idx.size() && (K[(size_t)idx[q]] >> 22) == g) { s += V[(size_t)idx[q]]; ++q; }
                const unsigned long long col = g & 0x7FFFFFFFull;
                long long first = (long long)(K[(size_t)idx[p]] & 0x3FFFFFull) - (col_block > 1 ? (long long)(col % (unsigned)col_block) : 0ll);
                if (first < 0) first = 0;
                K2.push_back(((g >> 31) << 53) | ((keep ? (unsigned long long)first : 0x3FFFFFull - (unsigned long long)first) << 31) | col);
                V2.push_back(s);
                p = q;
            }
            std::vector<int> id2(K2.size());
            for (size_t i = 0; i < id2.size(); ++i) id2[i] = (int)i;
            std::sort(id2.begin(), id2.end(), [&](int a, int b) { return K2[(size_t)a] < K2[(size_t)b]; });
            for (int i : id2) {
                const int r = t.row0 + (int)(K2[(size_t)i] >> 53);
                rows[(size_t)r].push_back(Out{(int)(K2[(size_t)i] & 0x7FFFFFFFull), V2[(size_t)i], 0});
            }
            continue;
        }
        // ---- spg_long_kernel: one row, one window of columns
        ++nwin;
        const int row = t.row0, w0 = t.col0, w1 = t.col1;
        if (!is_long[(size_t)row]) { is_long[(size_t)row] = 1; ++nlong; }
        std::vector<double> hval((size_t)SPL_WIN, 0.0);
        std::vector<int> hseq((size_t)SPL_WIN, -1);
        int gtotal = 0;
        int e0 = Ap[row];
        const int eEnd = Ap[row + 1];
        while (e0 < eEnd) {
            const int cand = std::min(BLK, eEnd - e0);
            std::vector<int> off((size_t)BLK + 1, 0), foff((size_t)BLK + 1, 0);
            for (int tI = 0; tI < BLK; ++tI) {
                int len = 0, flen = 0;
                if (tI < cand) {
                    const int k = Aj[e0 + tI];
                    flen = Bp[k + 1] - Bp[k];
                    for (int p = Bp[k]; p < Bp[k + 1]; ++p) len += (Bj[p] >= w0 && Bj[p] < w1) ? 1 : 0;
                }
                off[(size_t)tI + 1] = off[(size_t)tI] + len;
                foff[(size_t)tI + 1] = foff[(size_t)tI] + flen;
            }
            int take = BLK;
            if (off[BLK] > SPL_CAP) {
                int lo = 0, hi = BLK;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[(size_t)mid] <= SPL_CAP) lo = mid; else hi = mid; }
                take = lo;
                if (take == 0) return 2;
            }
            std::vector<unsigned long long> K;
            std::vector<double> V;
            for (int tI = 0; tI < std::min(take, cand); ++tI) {
                const int e = e0 + tI, k = Aj[e];
                for (int p = Bp[k]; p < Bp[k + 1]; ++p)
                    if (Bj[p] >= w0 && Bj[p] < w1) {
                        K.push_back(((unsigned long long)(unsigned)(Bj[p] - w0) << 32) | (unsigned)(gtotal + foff[(size_t)tI] + (p - Bp[k])));
                        V.push_back(Ax[e] * Bx[p]);
                    }
            }
            std::vector<int> idx(K.size());
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return K[(size_t)a] < K[(size_t)b]; });
            for (size_t p = 0; p < idx.size();) {
                const unsigned c = (unsigned)(K[(size_t)idx[p]] >> 32);
                if (hseq[c] < 0) hseq[c] = (int)(unsigned)(K[(size_t)idx[p]] & 0xFFFFFFFFull);
                double s = hval[c];
                size_t q = p;
                while (q < idx.size() && (unsigned)(K[(size_t)idx[q]] >> 32) == c) { s += V[(size_t)idx[q]]; ++q; }
                hval[c] = s;
                p = q;
            }
            gtotal += take == BLK ? foff[BLK] : foff[(size_t)take];
            e0 += take;                                  // like the kernel: beyond eEnd only after the last batch
        }
        for (int c = 0; c < SPL_WIN; ++c) {
            if (hseq[(size_t)c] < 0) continue;
            const int tt = std::max(0, hseq[(size_t)c] - (col_block > 1 ? (w0 + c) % col_block : 0));
            rows[(size_t)row].push_back(Out{w0 + c, hval[(size_t)c], keep ? INT_MAX - tt : tt});
        }
    }
    // ---- spg_reorder_kernel on the finished long rows, then the zero squeeze
    int64_t zeros = 0, w = 0;
    Cp[0] = 0;
    for (int r = 0; r < m; ++r) {
        std::vector<Out> &R = rows[(size_t)r];
        if (is_long[(size_t)r])
            std::sort(R.begin(), R.end(), [](const Out &a, const Out &b) { return a.seq != b.seq ? a.seq > b.seq : a.col < b.col; });
        for (const Out &o : R) {
            if (!keep && o.val == 0.0) { ++zeros; continue; }
            if (w >= cap) return 4;
            Cj[w] = o.col; Cx[w] = o.val; ++w;
        }
        Cp[r + 1] = (int)w;
    }
    if (stats) { stats[0] = (int64_t)tasks.size(); stats[1] = nlong; stats[2] = nwin; stats[3] = zeros; }
    return 0;
}

int spg_emul_limits(int *out)
{
    out[0] = SPG_CAP; out[1] = SPG_ROWS; out[2] = SPL_CAP; out[3] = SPL_WIN; out[4] = SPG_SLICE;
    return 0;
}

}  // extern "C"
