// blane_emul.cpp -- CPU replay of the lane-parallel fast-order block Gauss-Seidel sweep (test infrastructure, not product code).
// Builds the layout with the product's own planner (pyamg_amd/csrc/pamg_blane_plan.h) and consumes it the way bsr_lane_kernel does:
// `waves` waves take the groups w, w + W, ... and are visited in the adversarial order (the wave furthest ahead first; a group whose
// early operands are not all published "polls": skipped this turn; a turn without progress is a deadlock, error 20); lane l forms
// the bs partial sums of its K blocks (block by block, c = 0 .. bs - 1 inside a block row), the lanes of a block row are added by the
// XOR butterfly component by component, component r of the new x_i is sum_c Dinv_i[r][c] (b_i[c] - s[c]) in the order c = 0 .. bs - 1.
// Checked: early operands come from groups with smaller numbers (12), old operands are still old (13) unless a snapshot is used,
// no product in padding (11), every block row once (14, 15); a group whose gate is not published waits for it (a gate that belonged to a LATER
// group would stop the replay: error 20).
#include "../pyamg_amd/csrc/pamg_blane_plan.h"
#include <cmath>
#include <cstdio>

using namespace pamg;

extern "C" int blane_emul_sweep_f64(int n_brow, int bs, const int *bAp, const int *bAj, const double *bAx, const double *Dinv, double *x, const double *b,
                                    int row_start, int row_stop, int row_step, int waves, int snapshot, long long *stats)
{
    BlanePlan P;
    if (build_blane_plan(n_brow, bAp, bAj, reinterpret_cast<const unsigned char *>(bAx), 8, bs, row_start, row_stop, row_step, P)) return 2;
    const int L = P.L, K = P.K, RPW = P.RPW, bs2 = bs * bs;
    stats[0] = L; stats[1] = K; stats[2] = P.ngroups; stats[3] = P.nslots; stats[4] = P.n_early; stats[5] = P.n_old; stats[6] = P.nlevels;
    const double *vals = reinterpret_cast<const double *>(P.vals.data());
    const int n = n_brow * bs;
    std::vector<double> xs((size_t)n), xold;
    std::vector<char> pub((size_t)n_brow, 0), written((size_t)n_brow, 0);
    if (snapshot) xold.assign(x, x + n);
    const double *xsrc = snapshot ? xold.data() : x;
    int64_t rows_done = 0;
    int64_t gate_waits = 0;
    auto run_group = [&](int64_t g) -> int {
        // the gate: an early operand of an EARLIER group (or of one of this group's operands), never of a later one
        const int gt = P.gate[(size_t)g];
        if (gt >= 0 && !pub[(size_t)gt]) { ++gate_waits; return -1; }
        for (int k = 0; k < K; ++k)
            for (int lane = 0; lane < 64; ++lane) {
                const int c = P.cols[(size_t)((g * K + k) * 64 + lane)];
                if (!(c & LANE_NONE) && (c & LANE_EARLY) && !pub[(size_t)(c & LANE_MASK)]) return -1;
            }
        std::vector<double> acc((size_t)64 * bs, 0.0);
        for (int lane = 0; lane < 64; ++lane) {
            const int rid = P.rid[(size_t)(g * RPW + lane / L)];
            for (int k = 0; k < K; ++k) {
                const size_t sl = (size_t)((g * K + k) * 64 + lane);
                const int c = P.cols[sl];
                if (c & LANE_NONE) continue;
                if (rid < 0) return 11;
                const int j = c & LANE_MASK;
                const double *xj;
                if (c & LANE_EARLY) { if (!pub[(size_t)j]) return 12; xj = &xs[(size_t)j * bs]; }
                else { if (!snapshot && written[(size_t)j]) return 13; xj = &xsrc[(size_t)j * bs]; }
                for (int r = 0; r < bs; ++r) {
                    double a = acc[(size_t)lane * bs + r];
                    for (int cc = 0; cc < bs; ++cc) a = a + vals[((size_t)(g * K + k) * bs2 + (size_t)(r * bs + cc)) * 64 + lane] * xj[cc];
                    acc[(size_t)lane * bs + r] = a;
                }
            }
        }
        for (int r = 0; r < bs; ++r)
            for (int step = 1; step < L; step *= 2) {
                double t[64];
                for (int lane = 0; lane < 64; ++lane) t[lane] = acc[(size_t)lane * bs + r] + acc[(size_t)(lane ^ step) * bs + r];
                for (int lane = 0; lane < 64; ++lane) acc[(size_t)lane * bs + r] = t[lane];
            }
        for (int rr = 0; rr < RPW; ++rr) {
            const int i = P.rid[(size_t)(g * RPW + rr)];
            if (i < 0) continue;
            if (pub[(size_t)i]) return 14;
            double nv[8];
            for (int r = 0; r < bs; ++r) {
                double s = 0.0;
                for (int cc = 0; cc < bs; ++cc) s = s + Dinv[(size_t)i * bs2 + r * bs + cc] * (b[(size_t)i * bs + cc] - acc[(size_t)(rr * L) * bs + cc]);
                nv[r] = s;
            }
            for (int r = 0; r < bs; ++r) { xs[(size_t)i * bs + r] = nv[r]; x[(size_t)i * bs + r] = nv[r]; }
            pub[(size_t)i] = 1; written[(size_t)i] = 1;
            ++rows_done;
        }
        return 0;
    };
    if (waves < 1) waves = 1;
    std::vector<int64_t> next((size_t)waves);
    for (int w = 0; w < waves; ++w) next[(size_t)w] = w;
    int64_t left = P.ngroups;
    while (left > 0) {
        bool progress = false;
        for (int w = waves - 1; w >= 0; --w) {
            int64_t &g = next[(size_t)w];
            if (g >= P.ngroups) continue;
            const int rc = run_group(g);
            if (rc > 0) return rc;
            if (rc == 0) { g += waves; --left; progress = true; }
        }
        if (!progress) return 20;
    }
    stats[7] = gate_waits;
    const long span = (long)row_stop - row_start;
    if (rows_done != span / row_step) return 15;
    return 0;
}
