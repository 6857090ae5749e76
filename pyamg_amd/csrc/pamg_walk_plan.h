// pamg_walk_plan.h -- host-side layout of the LINE-WALK form of the fast-order Gauss-Seidel / SOR sweep (plain C++, no HIP:
// the CPU suite compiles this header with g++ and replays the plan, tests/walk_emul.cpp).
//
// Where the lane form (pamg_lane_plan.h) stops.  With one row per wave the lane-parallel sweep is bound by the hand-off
// alone: ~1 us per dependency LEVEL OF ROWS, and the level structure of an operator whose rows are numbered along lines
// (the coarse operators of smoothed aggregation on a grid: aggregates are numbered in the order of their roots, line after
// line) is deep because every row waits for the row before it.  Here a wave takes a whole LINE -- a run of consecutively
// visited rows each of which has its predecessor among its early operands -- and walks it row after row: the predecessor's
// new value never leaves the wave (a register), so along the line a dependency costs four dependent flops instead of a trip
// through memory; only operands from OTHER lines are polled in the hand-off buffer.  The rows and their order, every product
// and every hand-off are the lane form's (64 lanes share a row, K entries per lane, DPP butterfly, x 1/a_tt): the reference's
// sweep (amg_core::gauss_seidel, relaxation.h:48-76; sor :116-145) up to rounding.  Lines are ordered by their dependency
// level over the LINE graph and dealt out statically (wave w: lines w, w + W, ...): a row only waits for rows of lines with a
// lower level or for earlier rows of its own line -- deadlock-free with all waves resident.
//
// Layout (scheduled row q = position in line-major order; lane l of the wave that owns the line):
//   cols [(q * K + k) * 64 + l]   column | EARLY (bit 31: poll the hand-off buffer) | NONE (bit 30: padding); the diagonal and
//                                 the in-line predecessor are not stored here
//   vals [(q * K + k) * 64 + l]   a_ij
//   rid  [q]                      original row | NODIAG (bit 30: no / zero diagonal: row left untouched, still published)
//   rdiag[q]                      1 / a_tt
//   afwd [q]                      a_{t,t-1}: coefficient of the predecessor's NEW value (0: first row of a line / not coupled)
//   line_row[L], line_row[L + 1]  scheduled rows of line L (lines in level order)
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pamg {

constexpr int WALK_KMAX = 4;
constexpr int WALK_EARLY = (int)0x80000000u;
constexpr int WALK_NONE = 0x40000000;
constexpr int WALK_MASK = 0x3FFFFFFF;
constexpr int WALK_NODIAG = 0x40000000;
constexpr int WALK_MAXLINE = 256;             // rows per line (longer runs are cut: the next piece polls its predecessor)

struct WalkPlan {
    int K = 0;
    int64_t nrows = 0, nlines = 0;
    int nlevels = 0;
    int64_t max_level_lines = 0, n_early = 0, n_forward = 0;
    std::vector<int> cols, rid, line_row;
    std::vector<unsigned char> vals, rdiag, afwd;
};

// 0 = built; 1 = the form does not apply (rows with more than WALK_KMAX * 64 other entries, lines shorter than 8 rows on
// average: the caller keeps the lane form)
inline int build_walk_plan(int n, const int *Ap, const int *Aj, const unsigned char *Ax, int tsize, int row_start, int row_stop, int row_step,
                           WalkPlan &P)
{
    P = WalkPlan();
    if (row_step == 0) return 1;
    const int64_t span = (int64_t)row_stop - row_start;
    if (span % row_step != 0 || span / row_step <= 0) return 1;
    const int64_t m = span / row_step;
    if (row_start < 0 || row_start >= n || row_start + (m - 1) * row_step < 0 || row_start + (m - 1) * row_step >= n) return 1;
    auto row_of = [&](int64_t t) { return (int)(row_start + t * row_step); };
    auto vis = [&](int j) -> int64_t {
        const int64_t d = (int64_t)j - row_start;
        if (d % row_step != 0) return -1;
        const int64_t t = d / row_step;
        return (t >= 0 && t < m) ? t : -1;
    };
    // lines: runs of visits each coupled to the one before, at most WALK_MAXLINE rows
    std::vector<int64_t> lstart;                               // visit index of each line's first row
    std::vector<int64_t> line_of((size_t)m);
    int maxother = 1;
    {
        int64_t t0 = 0;
        lstart.push_back(0);
        for (int64_t t = 0; t < m; ++t) {
            const int i = row_of(t), prev = t > 0 ? row_of(t - 1) : -1;
            bool has_prev = false;
            int c = 0;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                if (Aj[p] == i) continue;
                if (prev >= 0 && Aj[p] == prev) has_prev = true;
                ++c;
            }
            if (t > 0 && (!has_prev || t - t0 >= WALK_MAXLINE)) { t0 = t; lstart.push_back(t); }
            line_of[(size_t)t] = (int64_t)lstart.size() - 1;
            maxother = std::max(maxother, c);
        }
    }
    const int64_t nl = (int64_t)lstart.size();
    if (nl * 8 > m) return 1;
    const int K = (maxother + 63) / 64;
    if (K > WALK_KMAX) return 1;
    lstart.push_back(m);
    // levels over the line graph
    std::vector<int> llevel((size_t)nl, 0);
    int maxl = 0;
    for (int64_t L = 0; L < nl; ++L) {
        int lv = 0;
        for (int64_t t = lstart[(size_t)L]; t < lstart[(size_t)L + 1]; ++t) {
            const int i = row_of(t);
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int j = Aj[p];
                if (j == i || j < 0 || j >= n) continue;
                const int64_t tj = vis(j);
                if (tj < 0 || tj >= t) continue;
                const int64_t Lj = line_of[(size_t)tj];
                if (Lj != L) lv = std::max(lv, llevel[(size_t)Lj] + 1);
            }
        }
        llevel[(size_t)L] = lv;
        maxl = std::max(maxl, lv);
    }
    P.nlevels = maxl + 1;
    std::vector<int64_t> lorder((size_t)nl);
    {
        std::vector<int64_t> cnt((size_t)maxl + 2, 0);
        for (int64_t L = 0; L < nl; ++L) cnt[(size_t)llevel[(size_t)L] + 1]++;
        for (int l = 0; l <= maxl; ++l) { P.max_level_lines = std::max(P.max_level_lines, cnt[(size_t)l + 1]); cnt[(size_t)l + 1] += cnt[(size_t)l]; }
        for (int64_t L = 0; L < nl; ++L) lorder[(size_t)cnt[(size_t)llevel[(size_t)L]]++] = L;
    }
    P.K = K; P.nrows = m; P.nlines = nl;
    P.cols.assign((size_t)m * K * 64, WALK_NONE);
    P.vals.assign((size_t)m * K * 64 * tsize, 0);
    P.rid.assign((size_t)m, 0);
    P.rdiag.assign((size_t)m * tsize, 0);
    P.afwd.assign((size_t)m * tsize, 0);
    P.line_row.assign((size_t)nl + 1, 0);
    int64_t q = 0;
    for (int64_t s = 0; s < nl; ++s) {
        const int64_t L = lorder[(size_t)s];
        P.line_row[(size_t)s] = (int)q;
        for (int64_t t = lstart[(size_t)L]; t < lstart[(size_t)L + 1]; ++t, ++q) {
            const int i = row_of(t);
            const int prev = (t > lstart[(size_t)L]) ? row_of(t - 1) : -1;        // forwarded only inside the line
            const unsigned char *dptr = nullptr, *pptr = nullptr;
            int e = 0;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int j = Aj[p];
                if (j == i) { dptr = Ax + (size_t)p * tsize; continue; }              // last stored diagonal wins
                if (j == prev && !pptr) { pptr = Ax + (size_t)p * tsize; continue; }  // (a duplicate entry keeps the slot path: it polls)
                const size_t slot = (size_t)((q * K + e / 64) * 64 + e % 64);
                ++e;
                if (j < 0 || j >= n) continue;
                const int64_t tj = vis(j);
                const bool early = tj >= 0 && tj < t;
                P.cols[slot] = j | (early ? WALK_EARLY : 0);
                std::memcpy(&P.vals[slot * tsize], Ax + (size_t)p * tsize, (size_t)tsize);
                if (early) ++P.n_early;
            }
            bool nod = true;
            if (tsize == 8) {
                double d = 0.0;
                if (dptr) std::memcpy(&d, dptr, 8);
                nod = !(d != 0.0);
                const double rd = nod ? 0.0 : 1.0 / d;
                std::memcpy(&P.rdiag[(size_t)q * 8], &rd, 8);
            } else {
                float d = 0.f;
                if (dptr) std::memcpy(&d, dptr, 4);
                nod = !(d != 0.f);
                const float rd = nod ? 0.f : 1.f / d;
                std::memcpy(&P.rdiag[(size_t)q * 4], &rd, 4);
            }
            if (pptr) { std::memcpy(&P.afwd[(size_t)q * tsize], pptr, (size_t)tsize); ++P.n_forward; }
            P.rid[(size_t)q] = i | (nod ? WALK_NODIAG : 0);
        }
    }
    P.line_row[(size_t)nl] = (int)q;
    return 0;
}

}  // namespace pamg
