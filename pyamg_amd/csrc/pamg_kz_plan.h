// pamg_kz_plan.h -- host-side layout of the LANE-PARALLEL ("fast order") Kaczmarz-type sweeps: gauss_seidel_ne and gauss_seidel_nr
// (plain C++, no HIP: the CPU suite compiles this header with g++ and replays the plan, tests/kz_emul.cpp).
//
// The sequential loops (amg_core::gauss_seidel_ne, relaxation.h:875-904: for every row i of A  d = (b_i - <a_i, x>) Dinv_i omega,
// x += a_i d;  gauss_seidel_nr, :939-975: for every column i of A  d = <a_i, r> (Dinv_i omega), x_i += d, r -= d a_i) read AND
// rewrite a vector v (x resp. the running residual r) at the indices of the line (row / column) they visit.  Two lines that share
// an index are ordered; lines that share none are independent.  The sweep keeps the reference's ORDER OF LINES -- the same
// dependency DAG, hence the same iterates in exact arithmetic -- and gives up only the association inside a line's dot product:
// L lanes of a wave share a line, every lane holds K of its entries, a butterfly adds the lanes (agrees with the reference to
// rounding, like the fast order of the Gauss-Seidel sweeps; the order-exact kernels stay for order = 'exact').
//
// Hand-off.  v is rewritten many times per sweep, so "the published value is the flag" of the Gauss-Seidel sweeps is not enough:
// every index of v has a 16-byte SLOT {value, version}; version = how many lines of this sweep have rewritten the index so far.
// A line knows for each of its entries which version it must see (= the number of EARLIER lines of the sweep that hold the index:
// `ver` below); it polls the slots of its entries until every version is the expected one -- from then on it is the only line
// allowed to touch those indices --, forms the dot product from the values it has just read, and writes {new value, version + 1}
// back with one 16-byte store per entry.  One memory round trip per dependency level, as in the Gauss-Seidel sweeps.
//
// Layout ("groups": the work of one wave; RPW = 64 / L lines each; levels padded to whole groups):
//   idx [(g * K + k) * 64 + lane]    index of v | NONE (bit 30: padding)
//   vals[(g * K + k) * 64 + lane]    the entry
//   ver [(g * K + k) * 64 + lane]    version the line must see at this index
//   line[g * RPW + r]                the line (row of the operator handed in), -1 for a dummy
// with lane = r * L + i and the line's entries e = 0, 1, ... (storage order) at k = e / L, i = e % L.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pamg_lane_plan.h"

namespace pamg {

constexpr int KZL_NONE = 0x40000000;
constexpr int KZL_MASK = 0x3FFFFFFF;

struct KzLanePlan {
    int L = 0, K = 0, RPW = 0;
    int nlevels = 0;
    int64_t ngroups = 0, nslots = 0;
    std::vector<int> idx, ver, line;
    std::vector<unsigned char> vals;
    int64_t max_level_groups = 0;
    int max_version = 0;
};

// Lp / Lj / Lx: CSR of the operator whose ROWS are the lines (A for gauss_seidel_ne, A^T for gauss_seidel_nr), ncols = length of v.
// Returns 0, or 1 when the form does not apply (lines too long, a line holding an index twice, padding too wasteful).
inline int build_kz_lane_plan(int nrows, int ncols, const int *Lp, const int *Lj, const unsigned char *Lx, int tsize, int start, int stop, int step,
                              KzLanePlan &P)
{
    P = KzLanePlan();
    if (step == 0) return 1;
    const long span = (long)stop - start;
    if (span % step != 0 || span / step < 0) return 1;
    const int m = (int)(span / step);
    if (m <= 0 || ncols > KZL_MASK) return 1;
    if (start < 0 || start >= nrows || start + (long)(m - 1) * step < 0 || start + (long)(m - 1) * step >= nrows) return 1;
    // dependency levels over shared indices (get_line_schedule in pamg_matrix.hip: one "highest level so far" per index)
    std::vector<int> seen((size_t)ncols, -1), lvl((size_t)m), cnt((size_t)ncols, 0), mark((size_t)ncols, -1);
    int nl = 0, maxlen = 0;
    int64_t total = 0;
    for (int t = 0; t < m; ++t) {
        const int i = start + t * step;
        int lv = 0;
        for (int p = Lp[i]; p < Lp[i + 1]; ++p) {
            const int j = Lj[p];
            if (j < 0 || j >= ncols) return 1;
            if (mark[(size_t)j] == t) return 1;                      // an index twice in one line: the slot protocol has one reader / writer per line and index
            mark[(size_t)j] = t;
            lv = std::max(lv, seen[(size_t)j] + 1);
        }
        for (int p = Lp[i]; p < Lp[i + 1]; ++p) seen[(size_t)Lj[p]] = lv;
        lvl[(size_t)t] = lv;
        nl = std::max(nl, lv + 1);
        maxlen = std::max(maxlen, Lp[i + 1] - Lp[i]);
        total += Lp[i + 1] - Lp[i];
    }
    P.nlevels = nl;
    int K = 0;
    const int L = lane_geometry(maxlen, 0, K);
    if (!L) return 1;
    const int RPW = 64 / L;
    P.L = L; P.K = K; P.RPW = RPW;
    std::vector<int64_t> lptr((size_t)nl + 1, 0), lgrp((size_t)nl + 1, 0);
    for (int t = 0; t < m; ++t) lptr[(size_t)lvl[(size_t)t] + 1]++;
    for (int l = 0; l < nl; ++l) {
        const int64_t w = (lptr[(size_t)l + 1] + RPW - 1) / RPW;
        P.max_level_groups = std::max(P.max_level_groups, w);
        lgrp[(size_t)l + 1] = lgrp[(size_t)l] + w;
        lptr[(size_t)l + 1] += lptr[(size_t)l];
    }
    P.ngroups = lgrp[(size_t)nl];
    P.nslots = P.ngroups * K * 64;
    if ((int64_t)K * L * m > 4 * total + (int64_t)8 * L * m || P.nslots >= ((int64_t)1 << 33) || P.ngroups >= ((int64_t)1 << 30)) return 1;
    P.idx.assign((size_t)P.nslots, KZL_NONE);
    P.ver.assign((size_t)P.nslots, 0);
    P.vals.assign((size_t)P.nslots * tsize, 0);
    P.line.assign((size_t)P.ngroups * RPW, -1);
    // the versions follow the SWEEP order (t), the slots the level order: one pass in sweep order fills both
    std::vector<int64_t> cur(lptr.begin(), lptr.end() - 1);
    for (int t = 0; t < m; ++t) {
        const int i = start + t * step, l = lvl[(size_t)t];
        const int64_t q = cur[(size_t)l]++ - lptr[(size_t)l];             // position of the line inside its level
        const int64_t g = lgrp[(size_t)l] + q / RPW;
        const int r = (int)(q % RPW);
        P.line[(size_t)(g * RPW + r)] = i;
        int e = 0;
        for (int p = Lp[i]; p < Lp[i + 1]; ++p, ++e) {
            const int j = Lj[p];
            const size_t s = (size_t)((g * K + e / L) * 64 + r * L + e % L);
            P.idx[s] = j;
            P.ver[s] = cnt[(size_t)j];
            std::memcpy(&P.vals[s * tsize], Lx + (size_t)p * tsize, (size_t)tsize);
            P.max_version = std::max(P.max_version, ++cnt[(size_t)j]);
        }
    }
    return 0;
}

}  // namespace pamg
