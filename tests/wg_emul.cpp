// wg_emul.cpp -- CPU replay of the workgroup-resident fast-order sweep (test infrastructure, not product code).
// Builds the layout with the product's own planner (pyamg_amd/csrc/pamg_wg_plan.h) and consumes it the way gs_wg_kernel does:
// every tile keeps the x values of its rows in an array of its own ("LDS"), walks its rounds in order, a round's waves add their K
// products per lane, the lanes of a row by the XOR butterfly, the row is finished with (b - sum) * rdiag, written over its old
// value in the tile's array, stored to x and -- if another tile reads it -- published in the sentinel-filled hand-off buffer.
// The tiles are advanced round-robin in the adversarial order (the LAST tile first); a tile whose next round needs a value an
// earlier tile has not published yet "polls" (is skipped); a full turn without progress is a deadlock (error 20).
// Checked on the way: an in-tile operand of a lower dependency level has been written, one of a higher level has not (the
// barrier between levels is what orders them: error 31 / 32), two connected rows never share a level (33); an OLD operand
// outside the tile is still old when it is read unless a snapshot is used (13); no product in padding or dummy rows (11);
// every visited row is done exactly once (14, 15); PUBLISH is set on every row another tile reads (34).
#include "../pyamg_amd/csrc/pamg_tile_plan.h"
#include "../pyamg_amd/csrc/pamg_wg_plan.h"
#include <cmath>
#include <cstdio>

using namespace pamg;

extern "C" int wg_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                                 int row_stop, int row_step, int max_tile_rows, int force_tiles, int sor, double omega, int snapshot,
                                 long long *stats)
{
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (sweep_levels(n, Ap, Aj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    if (m == 0) return 0;
    WgPlan P;
    if (build_wg_plan(n, Ap, Aj, reinterpret_cast<const unsigned char *>(Ax), 8, row_start, row_step, m, nl, vis, lvl, max_tile_rows, P, force_tiles)) return 2;
    const int L = P.L, K = P.K, RPW = P.RPW, G = P.G;
    stats[0] = L; stats[1] = K; stats[2] = G; stats[3] = P.nrounds; stats[4] = P.n_intile; stats[5] = P.n_cross; stats[6] = P.n_old; stats[7] = P.n_publish;
    const double *vals = reinterpret_cast<const double *>(P.vals.data());
    const double *rd = reinterpret_cast<const double *>(P.rdiag.data());
    std::vector<double> xs((size_t)n), xold;
    std::vector<char> pub((size_t)n, 0), written((size_t)n, 0);
    if (snapshot) xold.assign(x, x + n);
    const double *xsrc = snapshot ? xold.data() : x;
    // the tiles' own arrays, loaded with the old values of their rows
    std::vector<std::vector<double>> xt((size_t)G);
    std::vector<std::vector<int>> row_at((size_t)G);
    for (int k = 0; k < G; ++k) {
        const int t0 = P.tile_vis0[(size_t)k], t1 = P.tile_vis0[(size_t)k + 1];
        xt[(size_t)k].resize((size_t)(t1 - t0));
        row_at[(size_t)k].resize((size_t)(t1 - t0));
        for (int t = t0; t < t1; ++t) { xt[(size_t)k][(size_t)(t - t0)] = x[row_start + t * row_step]; row_at[(size_t)k][(size_t)(t - t0)] = row_start + t * row_step; }
    }
    int64_t rows_done = 0;
    auto run_round = [&](int k, int64_t round) -> int {
        // may the round run?  (every CROSS operand published)
        for (int w = 0; w < WG_NW; ++w) {
            const int64_t g = round * WG_NW + w;
            for (int kk = 0; kk < K; ++kk)
                for (int lane = 0; lane < 64; ++lane) {
                    const int c = P.cols[(size_t)((g * K + kk) * 64 + lane)];
                    if (!(c & WG_INTILE) && !(c & WG_NONE) && (c & WG_CROSS) && !pub[(size_t)(c & WG_MASK)]) return -1;
                }
        }
        struct Out { int row, lp; double v; bool upd, publish; };
        std::vector<Out> outs;
        for (int w = 0; w < WG_NW; ++w) {
            const int64_t g = round * WG_NW + w;
            double lane_sum[64];
            for (int lane = 0; lane < 64; ++lane) {
                double s = 0.0;
                const int rid = P.rid[(size_t)(g * RPW + lane / L)];
                const int irow = rid < 0 ? -1 : (rid & WG_MASK);
                for (int kk = 0; kk < K; ++kk) {
                    const size_t e = (size_t)((g * K + kk) * 64 + lane);
                    const int c = P.cols[e];
                    if (!(c & WG_INTILE) && (c & WG_NONE)) continue;
                    if (irow < 0) return 11;
                    double xv;
                    if (c & WG_INTILE) {
                        const int lp = c & WG_MASK;
                        if (lp < 0 || lp >= (int)xt[(size_t)k].size()) return 30;
                        const int j = row_at[(size_t)k][(size_t)lp];
                        if (lvl[j] < lvl[irow] && !written[(size_t)j]) return 31;     // a lower level not done before the barrier
                        if (lvl[j] > lvl[irow] && written[(size_t)j]) return 32;      // a higher level done early
                        if (lvl[j] == lvl[irow]) return 33;                           // connected rows on one level
                        xv = xt[(size_t)k][(size_t)lp];
                    } else if (c & WG_CROSS) {
                        const int j = c & WG_MASK;
                        if (!pub[(size_t)j]) return 12;
                        xv = xs[(size_t)j];
                    } else {
                        const int j = c & WG_MASK;
                        if (!snapshot && written[(size_t)j]) return 13;
                        xv = xsrc[j];
                    }
                    s = s + vals[e] * xv;
                }
                lane_sum[lane] = s;
            }
            for (int step = 1; step < L; step *= 2) {
                double t[64];
                for (int lane = 0; lane < 64; ++lane) t[lane] = lane_sum[lane] + lane_sum[lane ^ step];
                for (int lane = 0; lane < 64; ++lane) lane_sum[lane] = t[lane];
            }
            for (int r = 0; r < RPW; ++r) {
                const int rid = P.rid[(size_t)(g * RPW + r)];
                if (rid < 0) continue;
                const int row = rid & WG_MASK, lp = P.lpos[(size_t)(g * RPW + r)];
                const bool upd = !(rid & WG_NODIAG);
                const double xo = xt[(size_t)k][(size_t)lp];
                if (row_at[(size_t)k][(size_t)lp] != row) return 35;
                double v = (b[row] - lane_sum[r * L]) * rd[(size_t)(g * RPW + r)];
                if (sor) v = omega * v + (1.0 - omega) * xo;
                if (!upd) v = xo;
                outs.push_back({row, lp, v, upd, (rid & WG_PUBLISH) != 0});
            }
        }
        // the round's stores land after its reads (rows of one level never read each other)
        for (const Out &o : outs) {
            if (written[(size_t)o.row] || pub[(size_t)o.row]) return 14;
            xt[(size_t)k][(size_t)o.lp] = o.v;
            written[(size_t)o.row] = 1;
            if (o.upd) x[o.row] = o.v;
            if (o.publish) { xs[(size_t)o.row] = o.v; pub[(size_t)o.row] = 1; }
            ++rows_done;
        }
        return 0;
    };
    std::vector<int64_t> next((size_t)G);
    for (int k = 0; k < G; ++k) next[(size_t)k] = P.tile_round[(size_t)k];
    int64_t left = P.nrounds;
    while (left > 0) {
        bool progress = false;
        for (int k = G - 1; k >= 0; --k) {
            while (next[(size_t)k] < P.tile_round[(size_t)k + 1]) {
                const int rc = run_round(k, next[(size_t)k]);
                if (rc > 0) return rc;
                if (rc < 0) break;
                ++next[(size_t)k]; --left; progress = true;
            }
        }
        if (!progress) return 20;
    }
    if (rows_done != m) return 15;
    // PUBLISH covers every cross-tile read (a missing flag would have deadlocked above); the reverse: nothing publishes for nobody
    return 0;
}
