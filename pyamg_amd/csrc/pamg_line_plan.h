// pamg_line_plan.h -- host-side layout of the LINE-SCAN ("fast order") Gauss-Seidel / SOR sweep for banded operators in
// their natural order -- the fine levels of structured-grid problems (plain C++, no HIP: the CPU suite compiles this header
// with g++ and replays the plan, tests/line_emul.cpp).
//
// The idea.  In a sweep over consecutive rows (amg_core::gauss_seidel, relaxation.h:48-76, row_step = +-1) row t usually
// needs the NEW value of the row visited just before it -- the x-neighbour of a grid stencil -- which makes a grid line one
// long dependency chain and the level schedule of the whole grid 3 n levels deep with little work per level.  But along
// such a run the update is a first-order linear recurrence
//     x_t = B_t + A_t x_{t-1},   B_t = (b_t - sum over the OTHER entries) / a_tt,   A_t = - a_{t,t-1} / a_tt,
// and a linear recurrence is a SCAN: 64 consecutive rows are finished by one wave in log2(64) combining steps
// ((A2, B2) o (A1, B1) = (A2 A1, B2 + A2 B1)), exactly the same algebra evaluated in another association -- the reference's
// iterates up to rounding, like the lane-parallel row sums of pamg_lane.hip (fast order, tune key 24 = 1).  What remains
// sequential is the dependency between LINES (a line waits for the lines its other early operands live in): 2 n levels of
// n lines on an n^3 grid instead of 3 n levels, and -- the point -- a whole line of work per hand-off instead of a plane
// diagonal's worth of scattered rows.
//
// Layout.  The visit order is cut into CHUNKS of at most 64 consecutive rows (one row per lane, one wave per chunk step)
// such that inside a chunk a row's only early operand from the chunk itself is its immediate predecessor; consecutive
// chunks whose first row is coupled to its predecessor form a LINE, processed by ONE wave chunk after chunk (the running
// value is carried in a register).  Lines are sorted by their dependency level over the line graph; wave w takes lines
// w, w + W, ...  Per chunk g and lane l (row = row0[g] + l * step):
//   cols [(g * K + k) * 64 + l]   the row's entries other than the diagonal and the in-line predecessor:
//                                 column | EARLY (bit 31: poll the hand-off buffer) | NONE (bit 30: padding)
//   vals [(g * K + k) * 64 + l]   a_ij
//   rdiag[g * 64 + l]             1 / a_tt        (0 with the NODIAG flag: row left untouched, relaxation.h:72-74)
//   acoef[g * 64 + l]             - a_{t,t-1} / a_tt  (0: not coupled to the predecessor / first row of a line)
//   meta [g] = {row0, rows | NODIAG mask is kept per lane in the sign of rdiag's companion array `flag`}
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pamg {

constexpr int LINE_KMAX = 8;
constexpr int LINE_EARLY = (int)0x80000000u;
constexpr int LINE_NONE = 0x40000000;
constexpr int LINE_MASK = 0x3FFFFFFF;

struct LinePlan {
    int K = 0, step = 1;
    int64_t nchunks = 0, nlines = 0;
    int nlevels = 0;                          // levels of the line graph
    std::vector<int> cols;                    // nchunks * K * 64
    std::vector<unsigned char> vals;          // nchunks * K * 64 values
    std::vector<unsigned char> rdiag, acoef;  // nchunks * 64 values
    std::vector<unsigned char> nodiag;        // nchunks * 64: 1 = no (or zero) diagonal
    std::vector<int> row0, cnt, gate;         // per chunk: first row, rows, gate operand (column, or -1)
    std::vector<int> line_chunk;              // [nlines + 1] chunk range of each line, lines in level order
    int64_t n_early = 0, n_old = 0, max_level_lines = 0;
};

// Build from the CSR pattern (Ap, Aj), values Ax (tsize bytes each) and the sweep start / stop / step (|step| = 1).
// Returns 0, or 1 when the form does not apply (|step| != 1, a row with more than LINE_KMAX other entries, chunks that come
// out too short to pay: the caller keeps its other schedulers).
inline int build_line_plan(int n, const int *Ap, const int *Aj, const unsigned char *Ax, int tsize, int row_start, int row_stop, int row_step,
                           LinePlan &P)
{
    P = LinePlan();
    if (row_step != 1 && row_step != -1) return 1;
    const int64_t m = ((int64_t)row_stop - row_start) / row_step;
    if (m <= 0 || ((int64_t)row_stop - row_start) % row_step != 0) return 1;
    if (row_start < 0 || row_start >= n || row_start + (m - 1) * row_step < 0 || row_start + (m - 1) * row_step >= n) return 1;
    P.step = row_step;
    auto vis = [&](int j) -> int64_t {                         // visit index of row j, or -1
        const int64_t t = ((int64_t)j - row_start) * row_step; // step = +-1
        return (t >= 0 && t < m) ? t : -1;
    };
    auto row_of = [&](int64_t t) { return (int)(row_start + t * row_step); };
    // other entries per row (not diagonal, not the immediate predecessor) -> K
    int K = 1;
    for (int64_t t = 0; t < m; ++t) {
        const int i = row_of(t), prev = t > 0 ? row_of(t - 1) : -1;
        int c = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) c += (Aj[p] != i && Aj[p] != prev);
        if (c > LINE_KMAX) return 1;
        K = std::max(K, c);
    }
    P.K = K;
    // chunks: at most 64 consecutive visits; a row may need, from inside its chunk, only its immediate predecessor
    std::vector<int64_t> cstart;                               // visit index of each chunk's first row
    std::vector<char> coupled;                                 // chunk's first row is coupled to its predecessor -> same line as the chunk before
    {
        int64_t t0 = 0;
        cstart.push_back(0);
        coupled.push_back(0);
        for (int64_t t = 1; t < m; ++t) {
            const int i = row_of(t), prev = row_of(t - 1);
            bool brk = (t - t0) >= 64, has_prev = false;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int j = Aj[p];
                if (j == i || j < 0 || j >= n) continue;
                if (j == prev) { has_prev = true; continue; }
                const int64_t tj = vis(j);
                if (tj >= t0 && tj < t) brk = true;            // an early operand inside the chunk other than the predecessor
            }
            if (!has_prev) brk = true;                         // not coupled to its predecessor: a new line starts here (it can run beside the old one)
            if (brk) { t0 = t; cstart.push_back(t); coupled.push_back(has_prev ? 1 : 0); }
        }
    }
    const int64_t nch = (int64_t)cstart.size();
    if (nch * 16 > m) return 1;                                // chunks shorter than 16 rows on average: no banded structure to speak of
    cstart.push_back(m);
    // lines (in visit order) and their levels over the line graph
    std::vector<int64_t> line_of_chunk((size_t)nch);
    std::vector<int64_t> lfirst;                               // first chunk of each line (visit order)
    for (int64_t g = 0; g < nch; ++g) {
        if (g == 0 || !coupled[(size_t)g]) lfirst.push_back(g);
        line_of_chunk[(size_t)g] = (int64_t)lfirst.size() - 1;
    }
    const int64_t nl = (int64_t)lfirst.size();
    lfirst.push_back(nch);
    std::vector<int64_t> chunk_of_visit((size_t)m);
    for (int64_t g = 0; g < nch; ++g)
        for (int64_t t = cstart[(size_t)g]; t < cstart[(size_t)g + 1]; ++t) chunk_of_visit[(size_t)t] = g;
    std::vector<int> llevel((size_t)nl, 0);
    int maxl = 0;
    for (int64_t L = 0; L < nl; ++L) {
        int lv = 0;
        for (int64_t g = lfirst[(size_t)L]; g < lfirst[(size_t)L + 1]; ++g)
            for (int64_t t = cstart[(size_t)g]; t < cstart[(size_t)g + 1]; ++t) {
                const int i = row_of(t);
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                    const int j = Aj[p];
                    if (j == i || j < 0 || j >= n) continue;
                    const int64_t tj = vis(j);
                    if (tj < 0 || tj >= t) continue;
                    const int64_t Lj = line_of_chunk[(size_t)chunk_of_visit[(size_t)tj]];
                    if (Lj != L) lv = std::max(lv, llevel[(size_t)Lj] + 1);
                }
            }
        llevel[(size_t)L] = lv;
        maxl = std::max(maxl, lv);
    }
    P.nlevels = maxl + 1;
    // lines in level order (visit order inside a level); chunks renumbered line after line
    std::vector<int64_t> lorder((size_t)nl);
    {
        std::vector<int64_t> cntl((size_t)maxl + 2, 0);
        for (int64_t L = 0; L < nl; ++L) cntl[(size_t)llevel[(size_t)L] + 1]++;
        for (int l = 0; l <= maxl; ++l) { P.max_level_lines = std::max(P.max_level_lines, cntl[(size_t)l + 1]); cntl[(size_t)l + 1] += cntl[(size_t)l]; }
        for (int64_t L = 0; L < nl; ++L) lorder[(size_t)cntl[(size_t)llevel[(size_t)L]]++] = L;
    }
    P.nlines = nl;
    P.nchunks = nch;
    P.line_chunk.assign((size_t)nl + 1, 0);
    P.cols.assign((size_t)nch * K * 64, LINE_NONE);
    P.vals.assign((size_t)nch * K * 64 * tsize, 0);
    P.rdiag.assign((size_t)nch * 64 * tsize, 0);
    P.acoef.assign((size_t)nch * 64 * tsize, 0);
    P.nodiag.assign((size_t)nch * 64, 0);
    P.row0.assign((size_t)nch, 0); P.cnt.assign((size_t)nch, 0); P.gate.assign((size_t)nch, -1);
    int64_t gnew = 0;
    for (int64_t q = 0; q < nl; ++q) {
        const int64_t L = lorder[(size_t)q];
        P.line_chunk[(size_t)q] = (int)gnew;
        for (int64_t g = lfirst[(size_t)L]; g < lfirst[(size_t)L + 1]; ++g, ++gnew) {
            const int64_t t0 = cstart[(size_t)g], t1 = cstart[(size_t)g + 1];
            P.row0[(size_t)gnew] = row_of(t0);
            P.cnt[(size_t)gnew] = (int)(t1 - t0);
            for (int64_t t = t0; t < t1; ++t) {
                const int lane = (int)(t - t0);
                const int i = row_of(t), prev = t > 0 ? row_of(t - 1) : -1;
                const unsigned char *dptr = nullptr, *pptr = nullptr;
                int k = 0;
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                    const int j = Aj[p];
                    if (j == i) { dptr = Ax + (size_t)p * tsize; continue; }          // last stored diagonal wins
                    // the predecessor's coefficient goes into the recurrence -- only when the two rows share a LINE (the first row of a
                    // line polls its predecessor like any other early operand); a duplicate entry keeps the slot path
                    if (j == prev && !pptr && (t > t0 || coupled[(size_t)g])) { pptr = Ax + (size_t)p * tsize; continue; }
                    const size_t s = (size_t)((gnew * K + k) * 64 + lane);
                    ++k;
                    if (j < 0 || j >= n) continue;
                    const int64_t tj = vis(j);
                    const bool early = tj >= 0 && tj < t;
                    P.cols[s] = j | (early ? LINE_EARLY : 0);
                    std::memcpy(&P.vals[s * tsize], Ax + (size_t)p * tsize, (size_t)tsize);
                    if (early) { ++P.n_early; P.gate[(size_t)gnew] = j; } else ++P.n_old;
                }
                const size_t rs = (size_t)(gnew * 64 + lane);
                if (tsize == 8) {
                    double d = 0.0, ap = 0.0;
                    if (dptr) std::memcpy(&d, dptr, 8);
                    if (pptr) std::memcpy(&ap, pptr, 8);
                    const bool nod = !(d != 0.0);
                    const double rd = nod ? 0.0 : 1.0 / d, ac = nod ? 0.0 : -ap * rd;
                    P.nodiag[rs] = nod;
                    std::memcpy(&P.rdiag[rs * 8], &rd, 8);
                    std::memcpy(&P.acoef[rs * 8], &ac, 8);
                } else {
                    float d = 0.f, ap = 0.f;
                    if (dptr) std::memcpy(&d, dptr, 4);
                    if (pptr) std::memcpy(&ap, pptr, 4);
                    const bool nod = !(d != 0.f);
                    const float rd = nod ? 0.f : 1.f / d, ac = nod ? 0.f : -ap * rd;
                    P.nodiag[rs] = nod;
                    std::memcpy(&P.rdiag[rs * 4], &rd, 4);
                    std::memcpy(&P.acoef[rs * 4], &ac, 4);
                }
            }
        }
    }
    P.line_chunk[(size_t)nl] = (int)gnew;
    return 0;
}

}  // namespace pamg
