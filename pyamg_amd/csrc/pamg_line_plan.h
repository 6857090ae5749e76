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
#include "pamg_host_threads.h"
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "pamg_plan_vec.h"

namespace pamg {

constexpr int LINE_KMAX = 8;
constexpr int LINE_EARLY = (int)0x80000000u;
constexpr int LINE_NONE = 0x40000000;
constexpr int LINE_MASK = 0x3FFFFFFF;

struct LinePlan {
    int K = 0, step = 1;
    int64_t nchunks = 0, nlines = 0;
    int nlevels = 0;                          // levels of the line graph
    PlanVec<int> cols;                        // nchunks * K * 64
    PlanVec<unsigned char> vals;              // nchunks * K * 64 values
    PlanVec<unsigned char> rdiag, acoef;      // nchunks * 64 values
    PlanVec<unsigned char> nodiag;            // nchunks * 64: 1 = no (or zero) diagonal
    std::vector<int> row0, cnt, gate;         // per chunk: first row, rows, gate operand (column, or -1)
    std::vector<unsigned char> coupled;       // per chunk: its first row continues the line of the chunk before (used by the device fill)
    std::vector<int> line_chunk;              // [nlines + 1] chunk range of each line, lines in level order
    int64_t n_early = 0, n_old = 0, max_level_lines = 0;
};

// the planner's passes over the rows / lines run on a few threads (the 256^3 fine level: 117 M entries, 1.7 s on one)
template <typename F>
inline void line_parallel(int64_t n, F fn, int64_t grain = 1 << 15)
{
    unsigned nt = std::max(1u, std::min(32u, pamg::host_cpus()));
    nt = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nt, n / std::max<int64_t>(grain, 1)));
    if (nt <= 1) { fn((int64_t)0, n, 0); return; }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (unsigned k = 0; k < nt; ++k) {
        const int64_t lo = (int64_t)k * per, hi = std::min<int64_t>(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=, &fn] { fn(lo, hi, (int)k); });
    }
    for (auto &t : th) t.join();
}

// Build from the CSR pattern (Ap, Aj), values Ax (tsize bytes each) and the sweep start / stop / step (|step| = 1).
// Returns 0, or 1 when the form does not apply (|step| != 1, a row with more than LINE_KMAX other entries, chunks that come
// out too short to pay: the caller keeps its other schedulers).
// fill = false: only the structure (chunks, lines, levels, row0 / cnt / coupled per chunk) -- cols, vals, rdiag, acoef, nodiag and
// gate are then written by the device from the resident CSR arrays (line_fill_kernel, pamg_line.hip), Ax may be null.
inline int build_line_plan(int n, const int *Ap, const int *Aj, const unsigned char *Ax, int tsize, int row_start, int row_stop, int row_step,
                           LinePlan &P, bool fill = true)
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
    // per visit: other entries (not diagonal, not the immediate predecessor) -> K; coupled to the predecessor?; the latest early
    // operand other than the predecessor (what may force a chunk boundary)
    std::vector<char> has_prev((size_t)m, 0);
    std::vector<int> near((size_t)m, -1);
    std::atomic<int> Kmax(1), over(0);
    line_parallel(m, [&](int64_t lo, int64_t hi, int) {
        int kloc = 1;
        for (int64_t t = lo; t < hi; ++t) {
            const int i = row_of(t), prev = t > 0 ? row_of(t - 1) : -1;
            int c = 0;
            int64_t nr = -1;
            bool hp = false;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int j = Aj[p];
                c += (j != i);
                if (j == i || j < 0 || j >= n) continue;
                if (j == prev && !hp) { hp = true; continue; }
                const int64_t tj = vis(j);                         // (a DUPLICATE of the predecessor's entry is an early operand like any other:
                if (tj >= 0 && tj < t) nr = std::max(nr, tj);      //  it forces a chunk boundary, so it is polled from the chunk before)
            }
            // slots = entries other than the diagonal and the FIRST entry of the predecessor (a duplicate of it keeps a slot; a row
            // with a predecessor entry is always in its predecessor's line, see the chunk rule below)
            if (hp) --c;
            if (c > LINE_KMAX) { over = 1; return; }
            kloc = std::max(kloc, c);
            has_prev[(size_t)t] = hp;
            near[(size_t)t] = (int)nr;
        }
        int cur = Kmax.load();
        while (kloc > cur && !Kmax.compare_exchange_weak(cur, kloc)) {}
    });
    if (over.load()) return 1;
    const int K = Kmax.load();
    P.K = K;
    // chunks: at most 64 consecutive visits; a row may need, from inside its chunk, only its immediate predecessor
    std::vector<int64_t> cstart;                               // visit index of each chunk's first row
    std::vector<char> coupled;                                 // chunk's first row is coupled to its predecessor -> same line as the chunk before
    {
        int64_t t0 = 0;
        cstart.push_back(0);
        coupled.push_back(0);
        for (int64_t t = 1; t < m; ++t) {
            // 64 rows, an early operand inside the chunk other than the predecessor, or not coupled to its predecessor: a new chunk
            // (the last: a new line starts here -- it can run beside the old one)
            const bool brk = (t - t0) >= 64 || near[(size_t)t] >= t0 || !has_prev[(size_t)t];
            if (brk) { t0 = t; cstart.push_back(t); coupled.push_back(has_prev[(size_t)t] ? 1 : 0); }
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
    std::vector<int> line_of_visit((size_t)m);
    line_parallel(nch, [&](int64_t lo, int64_t hi, int) {
        for (int64_t g = lo; g < hi; ++g)
            for (int64_t t = cstart[(size_t)g]; t < cstart[(size_t)g + 1]; ++t) line_of_visit[(size_t)t] = (int)line_of_chunk[(size_t)g];
    }, 1 << 10);
    // the lines a line waits for: found in parallel (a handful per line on a grid), levels by one light pass in visit order
    constexpr int NPRED = 12;
    std::vector<int> pred((size_t)nl * NPRED, -1);
    std::vector<char> many((size_t)nl, 0);
    line_parallel(nl, [&](int64_t lo, int64_t hi, int) {
        for (int64_t L = lo; L < hi; ++L) {
            int *pl = &pred[(size_t)L * NPRED];
            int np = 0;
            for (int64_t t = cstart[(size_t)lfirst[(size_t)L]]; t < cstart[(size_t)lfirst[(size_t)L + 1]] && !many[(size_t)L]; ++t) {
                const int i = row_of(t);
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                    const int j = Aj[p];
                    if (j == i || j < 0 || j >= n) continue;
                    const int64_t tj = vis(j);
                    if (tj < 0 || tj >= t) continue;
                    const int Lj = line_of_visit[(size_t)tj];
                    if (Lj == (int)L) continue;
                    bool seen = false;
                    for (int q = np - 1; q >= 0 && !seen; --q) seen = pl[q] == Lj;
                    if (seen) continue;
                    if (np == NPRED) { many[(size_t)L] = 1; break; }
                    pl[np++] = Lj;
                }
            }
        }
    }, 1 << 6);
    std::vector<int> llevel((size_t)nl, 0);
    int maxl = 0;
    for (int64_t L = 0; L < nl; ++L) {
        int lv = 0;
        if (!many[(size_t)L]) {
            const int *pl = &pred[(size_t)L * NPRED];
            for (int q = 0; q < NPRED && pl[q] >= 0; ++q) lv = std::max(lv, llevel[(size_t)pl[q]] + 1);
        } else {
            for (int64_t t = cstart[(size_t)lfirst[(size_t)L]]; t < cstart[(size_t)lfirst[(size_t)L + 1]]; ++t) {
                const int i = row_of(t);
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                    const int j = Aj[p];
                    if (j == i || j < 0 || j >= n) continue;
                    const int64_t tj = vis(j);
                    if (tj < 0 || tj >= t) continue;
                    const int64_t Lj = line_of_visit[(size_t)tj];
                    if (Lj != L) lv = std::max(lv, llevel[(size_t)Lj] + 1);
                }
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
    P.row0.assign((size_t)nch, 0); P.cnt.assign((size_t)nch, 0); P.gate.assign((size_t)nch, -1); P.coupled.assign((size_t)nch, 0);
    {
        int64_t gnew = 0;
        for (int64_t q = 0; q < nl; ++q) {
            const int64_t L = lorder[(size_t)q];
            P.line_chunk[(size_t)q] = (int)gnew;
            for (int64_t g = lfirst[(size_t)L]; g < lfirst[(size_t)L + 1]; ++g, ++gnew) {
                P.row0[(size_t)gnew] = row_of(cstart[(size_t)g]);
                P.cnt[(size_t)gnew] = (int)(cstart[(size_t)g + 1] - cstart[(size_t)g]);
                P.coupled[(size_t)gnew] = (unsigned char)coupled[(size_t)g];
            }
        }
        P.line_chunk[(size_t)nl] = (int)gnew;
    }
    if (!fill) return 0;
    plan_fill(P.cols, (size_t)nch * K * 64, (int)LINE_NONE);
    plan_fill(P.vals, (size_t)nch * K * 64 * tsize, (unsigned char)0);
    plan_fill(P.rdiag, (size_t)nch * 64 * tsize, (unsigned char)0);
    plan_fill(P.acoef, (size_t)nch * 64 * tsize, (unsigned char)0);
    plan_fill(P.nodiag, (size_t)nch * 64, (unsigned char)0);
    std::atomic<int64_t> n_early(0), n_old(0);
    line_parallel(nl, [&](int64_t qlo, int64_t qhi, int) {
        int64_t ne = 0, no = 0;
        for (int64_t q = qlo; q < qhi; ++q) {
            const int64_t L = lorder[(size_t)q];
            int64_t gnew = P.line_chunk[(size_t)q];
            for (int64_t g = lfirst[(size_t)L]; g < lfirst[(size_t)L + 1]; ++g, ++gnew) {
                const int64_t t0 = cstart[(size_t)g], t1 = cstart[(size_t)g + 1];
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
                        if (early) { ++ne; P.gate[(size_t)gnew] = j; } else ++no;
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
        n_early += ne;
        n_old += no;
    }, 1 << 6);
    P.n_early = n_early.load();
    P.n_old = n_old.load();
    return 0;
}

}  // namespace pamg
