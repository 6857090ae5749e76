// CPU replay of the MERGED lane-parallel Gauss-Seidel sweep (pyamg_amd/csrc/pamg_lanem_plan.h): the plan is consumed the way
// gs_lanem_kernel consumes it -- group after group (one row per wave: 64 lanes, K_g slots per lane, XOR butterfly, (b - sum) * (1 / a_ii),
// early operands from the hand-off buffer, old operands from the SNAPSHOT of x, b operands from b) -- and asserts what the device relies on:
// producers have smaller group numbers AND an earlier super-level, no product in padding, every visited row exactly once.
// Test infrastructure only (tests/test_lanem_plan.py); also the statistics probe used to choose s (tools/lanem_stats.py).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../pyamg_amd/csrc/pamg_lanem_plan.h"
#include "../pyamg_amd/csrc/pamg_tile_plan.h"

using namespace pamg;

// stats[16]: 0 nsuper, 1 nlevels, 2 groups, 3 units, 4 early, 5 old, 6 b operands, 7 direct entries, 8 max_len, 9 closed by length,
//            10 closed by growth, 11 widest super-level, 12 rows with K = 1, 13 K = 2, 14 K = 3, 15 K >= 4;  gstat[1] = max accepted growth
extern "C" int lanem_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                                    int row_stop, int row_step, int s_max, double growth_cap, int len_cap, long long *stats, double *gstat,
                                    int waves, int plan_only, int rpw)
{
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (sweep_levels(n, Ap, Aj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    if (m == 0) return 0;
    LaneMPlan P;
    if (build_lanem_plan(n, Ap, Aj, Ax, row_start, row_step, m, nl, vis, lvl, s_max, growth_cap, P, len_cap, rpw ? rpw : 1)) return 2;
    stats[0] = P.nsuper; stats[1] = nl; stats[2] = P.nrows; stats[3] = P.n_units; stats[4] = P.n_early; stats[5] = P.n_old; stats[6] = P.n_b;
    stats[7] = P.n_direct; stats[8] = P.max_len; stats[9] = P.closed_by_length; stats[10] = P.closed_by_growth; stats[11] = P.max_super_groups;
    for (int k = 12; k < 16; ++k) stats[k] = 0;
    for (int64_t g = 0; g < P.ngroups; ++g) stats[11 + std::min<int>(4, P.K[(size_t)g])]++;          // K >= 4 in the last bin
    gstat[0] = P.max_growth;
    if (plan_only) return 0;
    std::vector<double> xs((size_t)n), xold(x, x + n);
    std::vector<char> pub((size_t)n, 0);
    std::vector<int> sup_pub((size_t)n, -1);
    int64_t rows_done = 0;
    // structure checks the kernel's address arithmetic relies on
    for (int64_t g = 0; g + 1 < P.ngroups; ++g) {
        if (P.unit[(size_t)g] + P.K[(size_t)g] != P.unit[(size_t)g + 1]) return 30;
        if (P.super_of[(size_t)g] > P.super_of[(size_t)g + 1]) return 31;
    }
    for (int s = 0; s < P.nsuper; ++s)
        for (int64_t g = P.super_grp[(size_t)s]; g < P.super_grp[(size_t)s + 1]; ++g) if (P.super_of[(size_t)g] != s) return 32;
    const int RPW = P.rpw, LPR = 64 / RPW;
    if (P.nrows != m) return 34;
    auto run_group = [&](int64_t g, bool may_wait) -> int {
        const int K = P.K[(size_t)g];
        const size_t s0 = (size_t)P.unit[(size_t)g] * 64;
        if (K < 1 || K > LANEM_KMAX) return 33;
        if (may_wait) {
            for (int e = 0; e < K * 64; ++e) {
                const int c = P.cols[s0 + (size_t)e];
                if (!(c & LANE_NONE) && (c & LANE_EARLY) && !pub[(size_t)(c & LANEM_MASK)]) return -1;            // still polling
            }
            const int gt = P.gate[(size_t)g];
            if (gt >= 0 && !pub[(size_t)gt]) return 16;                // every operand is there but the gate is not: the gate is not an ancestor
        }
        double lane_sum[64];
        for (int lane = 0; lane < 64; ++lane) {
            const int rid = P.rid[(size_t)(g * RPW + lane / LPR)];
            double s = 0.0;
            for (int k = 0; k < K; ++k) {
                const size_t e = s0 + (size_t)k * 64 + (size_t)lane;
                const int c = P.cols[e];
                if (c & LANE_NONE) { if (P.vals[e] != 0.0) return 11; continue; }
                if (rid < 0 || (rid & LANE_NODIAG)) return 17;          // a dummy slot / an untouched row carries no operands
                const int col = c & LANEM_MASK;
                double xv;
                if (c & LANE_EARLY) {
                    if (c & LANEM_BSRC) return 18;
                    if (!pub[(size_t)col]) return 12;                   // producer has a larger group number: deadlock on the device
                    if (sup_pub[(size_t)col] >= P.super_of[(size_t)g]) return 19;   // polled operands come from EARLIER super-levels
                    xv = xs[(size_t)col];
                } else if (c & LANEM_BSRC) xv = b[col];
                else xv = xold[(size_t)col];
                const double pr = P.vals[e] * xv;
                s = s + pr;
            }
            lane_sum[lane] = s;
        }
        for (int step = 1; step < LPR; step *= 2) {
            double t[64];
            for (int lane = 0; lane < 64; ++lane) t[lane] = lane_sum[lane] + lane_sum[lane ^ step];
            for (int lane = 0; lane < 64; ++lane) lane_sum[lane] = t[lane];
        }
        for (int r = 0; r < RPW; ++r) {                                  // the rows of a group publish "at once": operands were read above
            const int rid = P.rid[(size_t)(g * RPW + r)];
            if (rid < 0) continue;
            const int row = rid & LANE_MASK;
            const bool upd = !(rid & LANE_NODIAG);
            double v = (b[row] - lane_sum[r * LPR]) * P.rdiag[(size_t)(g * RPW + r)];
            if (!upd) v = xold[(size_t)row];
            if (pub[(size_t)row]) return 14;
            xs[(size_t)row] = v; pub[(size_t)row] = 1; sup_pub[(size_t)row] = P.super_of[(size_t)g];
            if (upd) x[row] = v;
            ++rows_done;
        }
        return 0;
    };
    if (waves <= 0) {
        for (int64_t g = 0; g < P.ngroups; ++g) { const int rc = run_group(g, false); if (rc) return rc; }
    } else {
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
