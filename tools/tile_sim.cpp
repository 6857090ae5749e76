// DESIGN AID (not product code): timing model of the tiled order-exact sweep on the CPU.  Builds the tile plan
// (pyamg_amd/csrc/pamg_tile_plan.h) for an operator and replays it with a simple cost model: a tile runs its steps in
// order, a step costs c0 + c1 * (longest row), an operand produced by another tile becomes usable `hop` after its
// step ended.  Returns the makespan and a few statistics -- used to compare partitions without GPU time.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef SIM_NOLIMIT
#define PAMG_TILE_SIM_LIMITS 1
#endif
#include "../pyamg_amd/csrc/pamg_tile_plan.h"

using namespace pamg;

extern "C" {
// out: [0] makespan us, [1] steps, [2] levels, [3] critical-path crossings, [4] sum of step costs on the busiest tile,
//      [5] tiles, [6] global early entries, [7] local early entries, [8] ideal = levels * mean step cost
int tile_sim(int n, const int *Ap, const int *Aj, int row_start, int row_stop, int row_step, int G, int W, int cap, int max_rows,
             double c0, double c1, double c2, double hop, int mode, double *out, const int *row_tile)
{
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (sweep_levels(n, Ap, Aj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    if (G <= 0) G = std::max(1, (int)((long)m / std::max(1L, 7L * nl)));
    TilePlan P;
    if (build_tile_plan_from(n, Ap, Aj, row_start, row_step, m, nl, vis, lvl, G, W, cap, max_rows, P, mode, row_tile)) return 2;
    const int ns = (int)P.steps.size();
    // stored row -> step
    std::vector<int> step_of((size_t)m, 0), tile_of_step((size_t)ns, 0);
    for (int k = 0; k < P.G; ++k)
        for (int s = P.tile_step[k]; s < P.tile_step[k + 1]; ++s) {
            tile_of_step[s] = k;
            for (int r = P.steps[s].r0; r < P.steps[s].r1; ++r) step_of[r] = s;
        }
    std::vector<int> pos((size_t)n, -1);
    for (int r = 0; r < m; ++r) pos[P.rid[r] & TP_MASK] = r;
    // steps must be simulated in an order compatible with dependencies: by (level, tile)
    std::vector<int> ord((size_t)ns);
    for (int s = 0; s < ns; ++s) ord[s] = s;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return P.step_level[a] < P.step_level[b]; });
    std::vector<double> end((size_t)ns, 0.0), tile_t((size_t)P.G, 0.0), busy((size_t)P.G, 0.0);
    std::vector<int> cross((size_t)ns, 0);      // crossings on the longest path ending at this step
    std::vector<int> tile_last((size_t)P.G, -1);
    double span = 0, cost_sum = 0;
    for (int s : ord) {
        const TileStep &st = P.steps[s];
        const int k = tile_of_step[s];
        double start = tile_t[k];
        int cr = tile_last[k] >= 0 ? cross[tile_last[k]] : 0;
        int maxlen = 0;
        for (int r = st.r0; r < st.r1; ++r) {
            maxlen = std::max(maxlen, P.Ap[r + 1] - P.Ap[r]);
            for (int q = P.Ap[r]; q < P.Ap[r + 1]; ++q) {
                const int c = P.Aj[q];
                if (c < 0 && !(c & TP_DIAG)) {
                    const int ps = step_of[pos[c & TP_MASK]];
                    if (end[ps] + hop > start) { start = end[ps] + hop; cr = cross[ps] + 1; }
                }
            }
        }
        const double c = c0 + c1 * maxlen + c2 * ((st.p1 - st.p0) / 64);
        end[s] = start + c;
        tile_t[k] = end[s];
        busy[k] += c;
        cost_sum += c;
        cross[s] = cr;
        tile_last[k] = s;
        span = std::max(span, end[s]);
    }
    if (getenv("TILE_SIM_VERBOSE")) {
        for (int k = 0; k < P.G; k += std::max(1, P.G / 16)) {
            const int sa = P.tile_step[k], sb = P.tile_step[k + 1];
            if (sa >= sb) continue;
            int waits = 0; double waited = 0;
            for (int s2 = sa + 1; s2 < sb; ++s2) {
                const double c = end[s2] - end[s2 - 1];
                const double own = c0 + c1 * 30;
                if (c > own * 1.5) { waits++; waited += c - own; }
            }
            printf("  tile %d: steps %d levels %d..%d rows %d first-start %.1f end %.1f busy %.1f waits %d waited %.1f\n", k, sb - sa,
                   P.step_level[sa], P.step_level[sb - 1], P.steps[sb - 1].r1 - P.steps[sa].r0, end[sa], end[sb - 1], busy[k], waits, waited);
        }
    }
    int crit = 0;
    for (int s = 0; s < ns; ++s) if (end[s] == span) crit = cross[s];
    double bmax = 0;
    for (double b : busy) bmax = std::max(bmax, b);
    out[0] = span; out[1] = ns; out[2] = nl; out[3] = crit; out[4] = bmax; out[5] = P.G; out[6] = (double)P.n_global;
    out[7] = (double)P.n_local; out[8] = nl * (cost_sum / std::max(1, ns));
    return 0;
}
}
