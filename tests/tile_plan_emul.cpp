// TEST INFRASTRUCTURE (CPU): replays a tile plan (pyamg_amd/csrc/pamg_tile_plan.h) exactly the way
// gs_tile_kernel consumes it -- LDS ring per tile with wrap-around, global hand-off buffer with the
// sentinel, publish flags, OLD values fetched one step ahead -- under two interleavings of the tiles
// (level order; greedy: every tile runs as far ahead as its operands allow).  Used by
// tests/test_tile_plan.py to pin the host logic bit-for-bit against the oracle without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../pyamg_amd/csrc/pamg_tile_plan.h"

using namespace pamg;

namespace {
const uint64_t SENT = 0x7FF8DEADBEEF5A5Aull;
inline bool is_sent(double v) { uint64_t b; std::memcpy(&b, &v, 8); return b == SENT; }
inline double sent() { double v; std::memcpy(&v, &SENT, 8); return v; }

struct Emul {
    const TilePlan &P;
    const double *Ax;      // operator values (original order)
    const double *b;
    std::vector<double> &x;          // live vector
    const std::vector<double> *snap;  // snapshot for OLD reads or nullptr
    std::vector<double> xs;
    std::vector<std::vector<double>> ring;
    std::vector<std::vector<double>> oldv;   // per tile: OLD operands of the NEXT step (prefetched)
    std::vector<int> next;                   // per tile: next step to run
    int epi; double omega;
    int hazards = 0;

    Emul(const TilePlan &P_, const double *Ax_, const double *b_, std::vector<double> &x_, const std::vector<double> *snap_,
         int epi_, double omega_)
        : P(P_), Ax(Ax_), b(b_), x(x_), snap(snap_), epi(epi_), omega(omega_)
    {
        xs.assign(x.size(), sent());
        ring.assign(P.G, std::vector<double>((size_t)P.W, std::nan("")));
        oldv.resize(P.G);
        next.resize(P.G);
        for (int k = 0; k < P.G; ++k) { next[k] = P.tile_step[k]; if (next[k] < P.tile_step[k + 1]) prefetch(k, next[k]); }
    }
    double old_value(int j) const { return snap ? (*snap)[j] : x[j]; }
    void prefetch(int k, int s)
    {
        const TileStep &st = P.steps[s];
        oldv[k].assign((size_t)(st.p1 - st.p0), 0.0);
        for (int q = st.p0; q < st.p1; ++q) {
            const int c = P.Aj[q];
            if (c >= 0 && !(c & TP_DIAG)) oldv[k][q - st.p0] = old_value(c & TP_MASK);
        }
    }
    bool ready(int k) const
    {
        const int s = next[k];
        if (s >= P.tile_step[k + 1]) return false;
        const TileStep &st = P.steps[s];
        for (int q = st.p0; q < st.p1; ++q) {
            const int c = P.Aj[q];
            if (c < 0 && !(c & TP_DIAG) && is_sent(xs[c & TP_MASK])) return false;
        }
        return true;
    }
    void run(int k)
    {
        const int s = next[k];
        const TileStep &st = P.steps[s];
        const int base = P.steps[P.tile_step[k]].r0;
        const std::vector<double> cur_old = oldv[k];
        // the kernel issues the gathers of step s+1 BEFORE it consumes step s
        if (s + 1 < P.tile_step[k + 1]) prefetch(k, s + 1);
        std::vector<double> prod((size_t)(st.p1 - st.p0));
        for (int q = st.p0; q < st.p1; ++q) {
            const int c = P.Aj[q];
            const bool early = c < 0, dg = (c & TP_DIAG) != 0;
            double xv;
            if (early && dg) xv = ring[k][(size_t)(c & (P.W - 1))];
            else if (early) { xv = xs[c & TP_MASK]; if (is_sent(xv)) hazards++; }
            else if (dg) { prod[q - st.p0] = 0.0; continue; }
            else xv = cur_old[q - st.p0];
            if (std::isnan(xv) && !(early && dg)) {}
            prod[q - st.p0] = Ax[P.src[q]] * xv;
        }
        std::vector<double> newv((size_t)(st.r1 - st.r0));
        for (int r = st.r0; r < st.r1; ++r) {
            const int row = P.rid[r] & TP_MASK;
            double d = 0.0;
            for (int q = P.Ap[r]; q < P.Ap[r + 1]; ++q)
                if ((P.Aj[q] & TP_DIAG) && P.Aj[q] >= 0) d = Ax[P.src[q]];
            double sum = (epi == 1) ? b[row] : 0.0;
            for (int q = P.Ap[r]; q < P.Ap[r + 1]; ++q) {
                if (epi == 1) sum -= prod[q - st.p0];
                else sum += prod[q - st.p0];
            }
            const double xo = old_value(row);
            double v;
            if (epi == 0) v = (b[row] - sum) / d;
            else if (epi == 1) v = sum / d;
            else v = omega * ((b[row] - sum) / d) + (1.0 - omega) * xo;
            if (!(d != 0.0)) v = xo;
            newv[r - st.r0] = v;
        }
        for (int r = st.r0; r < st.r1; ++r) {
            const int row = P.rid[r] & TP_MASK;
            ring[k][(size_t)((r - base) & (P.W - 1))] = newv[r - st.r0];
            x[row] = newv[r - st.r0];
            if (P.rid[r] < 0) xs[row] = newv[r - st.r0];
        }
        next[k] = s + 1;
    }
};
}  // namespace

extern "C" {

// x is swept in place.  policy 0: steps in (level, tile) order; 1: greedy, lowest tile first; 2: greedy, highest
// tile first.  stats: [0] tiles, [1] steps, [2] levels, [3] local early entries, [4] global early entries,
// [5] publishing rows, [6] hazards (sentinel consumed), [7] stuck (deadlock in the replay)
int tile_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                        int row_stop, int row_step, int G, int W, int cap, int max_rows, int epi, double omega,
                        int use_snapshot, int policy, int64_t *stats)
{
    TilePlan P;
    if (build_tile_plan(n, Ap, Aj, row_start, row_stop, row_step, G, W, cap, max_rows, P)) return 1;
    std::vector<double> xv(x, x + n), snap;
    if (use_snapshot) snap = xv;
    Emul E(P, Ax, b, xv, use_snapshot ? &snap : nullptr, epi, omega);
    const int nsteps = (int)P.steps.size();
    int done = 0, stuck = 0;
    if (policy == 0) {
        std::vector<int> ord((size_t)nsteps);
        for (int s = 0; s < nsteps; ++s) ord[s] = s;
        std::vector<int> tile_of((size_t)nsteps);
        for (int k = 0; k < P.G; ++k) for (int s = P.tile_step[k]; s < P.tile_step[k + 1]; ++s) tile_of[s] = k;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return P.step_level[a] < P.step_level[c]; });
        for (int s : ord) {
            const int k = tile_of[s];
            if (E.next[k] != s || !E.ready(k)) { stuck = 1; break; }
            E.run(k);
            ++done;
        }
    } else {
        while (done < nsteps) {
            bool any = false;
            for (int kk = 0; kk < P.G; ++kk) {
                const int k = policy == 1 ? kk : P.G - 1 - kk;
                while (E.ready(k)) { E.run(k); ++done; any = true; }
            }
            if (!any) { stuck = 1; break; }
        }
    }
    std::memcpy(x, xv.data(), sizeof(double) * (size_t)n);
    if (stats) {
        stats[0] = P.G; stats[1] = nsteps; stats[2] = P.nlevels; stats[3] = P.n_local; stats[4] = P.n_global;
        stats[5] = P.n_publish; stats[6] = E.hazards; stats[7] = stuck;
    }
    return 0;
}

}  // extern "C"
