// lane_emul.cpp -- CPU replay of the lane-parallel fast-order sweep (test infrastructure, not product code).
// Builds the layout with the product's own planner (pyamg_amd/csrc/pamg_lane_plan.h) and consumes it the way
// gs_lane_kernel does: group after group, lane l of a group adds its K products, the lanes of a row are added by the
// XOR butterfly (1, 2, 4, ...), the head lane finishes the row with (b - sum) * rdiag, publishes into the
// sentinel-filled hand-off buffer and stores x.  Checks on the way what the device relies on: every EARLY operand has
// been published by a group with a SMALLER number (deadlock freedom of the static assignment; with `waves` > 0 the groups are
// run by that many waves taking w, w + W, ..., visited in the adversarial order: the wave furthest ahead first -- a round without
// progress is a deadlock), every OLD operand is still old when it is read (write-after-read safety on structurally symmetric
// patterns, else a snapshot is used), dummy rows / padding slots carry no product.
#include "../pyamg_amd/csrc/pamg_tile_plan.h"
#include "../pyamg_amd/csrc/pamg_lane_plan.h"
#include <cmath>
#include <cstdio>

using namespace pamg;

extern "C" int lane_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                                   int row_stop, int row_step, int want_L, int sor, double omega, int snapshot, long long *stats,
                                   int waves)
{
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (sweep_levels(n, Ap, Aj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    if (m == 0) return 0;
    LanePlan P;
    if (build_lane_plan(n, Ap, Aj, reinterpret_cast<const unsigned char *>(Ax), 8, row_start, row_step, m, nl, vis, lvl, want_L, P)) return 2;
    const int L = P.L, K = P.K, RPW = P.RPW;
    stats[0] = L; stats[1] = K; stats[2] = P.ngroups; stats[3] = P.n_slots; stats[4] = P.n_early; stats[5] = P.n_old; stats[6] = nl; stats[7] = 0;
    std::vector<double> xs((size_t)n), xold;
    std::vector<char> pub((size_t)n, 0), written((size_t)n, 0);
    const double *rd = reinterpret_cast<const double *>(P.rdiag.data());
    const double *vals = reinterpret_cast<const double *>(P.vals.data());
    if (snapshot) xold.assign(x, x + n);
    const double *xsrc = snapshot ? xold.data() : x;
    int64_t rows_done = 0;
    auto run_group = [&](int64_t g, bool may_wait) -> int {
        if (may_wait) {
            for (int lane = 0; lane < 64; ++lane)
                for (int k = 0; k < K; ++k) {
                    const int c = P.cols[(size_t)((g * K + k) * 64 + lane)];
                    if (!(c & LANE_NONE) && (c & LANE_EARLY) && !pub[(size_t)(c & LANE_MASK)]) return -1;       // still polling
                }
        }
        double lane_sum[64];
        for (int lane = 0; lane < 64; ++lane) {
            double s = 0.0;
            const int rid = P.rid[(size_t)(g * RPW + lane / L)];
            for (int k = 0; k < K; ++k) {
                const size_t e = (size_t)((g * K + k) * 64 + lane);
                const int c = P.cols[e];
                if (c & LANE_NONE) continue;
                if (rid < 0) return 11;                                   // an entry in a dummy row
                const int col = c & LANE_MASK;
                double xv;
                if (c & LANE_EARLY) {
                    if (!pub[(size_t)col]) return 12;                     // producer has a larger group number: deadlock on the device
                    xv = xs[(size_t)col];
                } else {
                    if (!snapshot && written[(size_t)col]) return 13;     // an old value was overwritten before it was read
                    xv = xsrc[col];
                }
                const double pr = vals[e] * xv;
                s = s + pr;
            }
            lane_sum[lane] = s;
        }
        for (int step = 1; step < L; step *= 2) {                          // XOR butterfly: every lane ends with the total
            double t[64];
            for (int lane = 0; lane < 64; ++lane) t[lane] = lane_sum[lane] + lane_sum[lane ^ step];
            for (int lane = 0; lane < 64; ++lane) lane_sum[lane] = t[lane];
        }
        // all rows of a group publish "at once": operands were read above, before any store of this group
        for (int r = 0; r < RPW; ++r) {
            const int rid = P.rid[(size_t)(g * RPW + r)];
            if (rid < 0) continue;
            const int row = rid & LANE_MASK;
            const bool upd = !(rid & LANE_NODIAG);
            double v = (b[row] - lane_sum[r * L]) * rd[(size_t)(g * RPW + r)];
            if (sor) v = omega * v + (1.0 - omega) * xsrc[row];
            if (!upd) v = xsrc[row];
            if (pub[(size_t)row]) return 14;                              // a row scheduled twice
            xs[(size_t)row] = v; pub[(size_t)row] = 1;
            if (upd) { x[row] = v; written[(size_t)row] = 1; }
            ++rows_done;
        }
        return 0;
    };
    if (waves <= 0) {
        for (int64_t g = 0; g < P.ngroups; ++g) { const int rc = run_group(g, false); if (rc) return rc; }
    } else {
        // `waves` waves, wave w takes groups w, w + waves, ...; visited LAST to first (the adversarial order: the waves that are furthest
        // ahead get the first chance and must wait); a full round without progress is a deadlock (error 20)
        std::vector<int64_t> next((size_t)waves);
        for (int w = 0; w < waves; ++w) next[(size_t)w] = w;
        int64_t left = P.ngroups;
        while (left > 0) {
            bool progress = false;
            for (int w = waves - 1; w >= 0; --w) {
                int64_t &g = next[(size_t)w];
                if (g >= P.ngroups) continue;
                const int rc = run_group(g, true);
                if (rc > 0) return rc;
                if (rc == 0) { g += waves; --left; progress = true; }
            }
            if (!progress) return 20;
        }
    }
    if (rows_done != m) return 15;
    return 0;
}
