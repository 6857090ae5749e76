// pamg_lane_plan.h -- host-side layout of the LANE-PARALLEL ("fast order") Gauss-Seidel / SOR sweep (plain C++, no
// HIP: the CPU suite compiles this header with g++ and replays the plan, tests/lane_emul.cpp).
//
// What stays and what goes.  The sweep keeps the reference's ORDER OF ROWS (amg_core::gauss_seidel,
// relaxation.h:48-76; sor_gauss_seidel :116-145; bsr_gauss_seidel with 1x1 blocks :185-266): row i uses the NEW
// values of the connected rows visited before it and the OLD values of the others, exactly as the sequential loop
// does -- same dependency DAG, same iterates in exact arithmetic.  What goes is the order of the additions INSIDE a
// row sum and the IEEE division: L lanes share a row, every lane adds its K products, a butterfly adds the lanes, and
// the row is finished with (b - sum) * (1 / a_ii).  Results agree with the reference to rounding (a few ulp per
// sweep; tests hold 1e-13 per sweep and the north star's 1e-10 on residual norms), not bit for bit -- the
// order-exact schedulers remain available (tune key 24 = 0).
//
// Layout ("groups").  The rows of one dependency level are cut into groups of RPW = 64 / L rows; a group is the work
// of ONE wave.  Every level is padded to whole groups with dummy rows, every row to K * L entry slots, so the
// address of everything follows from the group number alone (no descriptors, no row pointers):
//   cols [(g * K + k) * 64 + lane]   column | EARLY (bit 31: poll the hand-off buffer) | NONE (bit 30: padding, no product)
//   vals [(g * K + k) * 64 + lane]   a_ij
//   rid  [g * RPW + r]               original row (-1: dummy row of the padding) | NODIAG (bit 30: the row has no or a
//                                    zero diagonal: it is left untouched, relaxation.h:72-74, but still publishes its value)
//   rdiag[g * RPW + r]               1 / a_ii
//   gate [g]                         "gate" operand of the group: the column of its early operand with the HIGHEST dependency
//                                    level among those at least two levels below the group's own (-1: none).  A wave that runs
//                                    ahead polls this one value until the sweep is one level away, and only then all its
//                                    operands: polling traffic of ~1 instead of ~3 dependency levels per group.
// (on the device rid, gate and rdiag of a slot row travel as ONE 16-byte record, pamg_lane.hip: one request instead of three)
// with lane = r * L + i and the row's off-diagonal entries e = 0, 1, ... (storage order) at k = e / L, i = e % L.
// Diagonal entries are not stored at all (every stored a_ii is skipped by the reference's sum; the last one is the
// diagonal, relaxation.h:64-69).
#pragma once
#include "pamg_host_threads.h"
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <atomic>
#include <vector>

#include "pamg_plan_vec.h"

namespace pamg {

constexpr int LANE_KMAX = 4;                  // entry slots per lane the kernels are built for
constexpr int LANE_EARLY = (int)0x80000000u;
constexpr int LANE_NONE = 0x40000000;
constexpr int LANE_MASK = 0x3FFFFFFF;
constexpr int LANE_NODIAG = 0x40000000;       // in rid[]

struct LanePlan {
    int L = 0, K = 0, RPW = 0;
    int64_t ngroups = 0;
    int nlevels = 0;
    int max_offdiag = 0;
    PlanVec<int> cols;
    PlanVec<unsigned char> vals;              // ngroups * K * 64 values of tsize bytes
    std::vector<int> rid;
    std::vector<unsigned char> rdiag;         // ngroups * RPW values of tsize bytes
    std::vector<int> gate;                    // [ngroups] column of the group's latest early operand from a level <= own level - 2, or -1
    std::vector<int64_t> level_grp;           // [nlevels + 1] group range of each dependency level
    int64_t n_early = 0, n_old = 0, n_slots = 0;
    int64_t max_level_groups = 0;             // groups of the widest dependency level
};

template <typename F>
inline void lane_parallel(int64_t n, F fn, int64_t grain = 4096)
{
    const unsigned hw = std::max(1u, std::min(48u, pamg::host_cpus()));
    const int nt = (n < 2 * grain) ? 1 : (int)std::min<int64_t>(hw, n / grain);
    if (nt <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}

// lanes per row for rows with at most `maxlen` off-diagonal entries: the smallest power of two in [4, 64] that needs at
// most LANE_KMAX slots per lane (in-lane adds are cheaper than butterfly steps, and more rows per wave mean fewer
// polling waves); 0 = rows too long for this form.  want_L != 0 asks for at least that width (tuning).
inline int lane_geometry(int maxlen, int want_L, int &K)
{
    for (int L = 4; L <= 64; L *= 2) {
        if (want_L && L < want_L) continue;
        const int k = std::max(1, (maxlen + L - 1) / L);
        if (k <= LANE_KMAX) { K = k; return L; }
    }
    K = 0;
    return 0;
}

// Build the layout from a finished analysis of the sweep (vis = visit index or -1, lvl = dependency level of every
// visited row: sweep_levels in pamg_tile_plan.h), m visited rows, nl levels.  Ax: the operator's values (tsize bytes
// each).  Returns 0, or 1 when the rows are too long / the padding too wasteful (caller keeps the exact schedulers).
inline int build_lane_plan(int n, const int *Ap, const int *Aj, const unsigned char *Ax, int tsize, int row_start, int row_step,
                           int m, int nl, const std::vector<int> &vis, const std::vector<int> &lvl, int want_L, LanePlan &P, bool fill = true)
{
    // fill = false: the structure, the row of every (group, slot row), the gates and the statistics -- cols, vals, rdiag and the
    // NODIAG flags are then written by the device from the resident CSR arrays (lane_fill_kernel, pamg_lane.hip); Ax may be null
    P = LanePlan();
    P.nlevels = nl;
    if (m <= 0 || nl <= 0) return 1;
    if (n > LANE_MASK) return 1;
    // rows in level order, visit order inside a level
    std::vector<int64_t> lptr((size_t)nl + 1, 0);
    for (int t = 0; t < m; ++t) lptr[(size_t)lvl[row_start + (int64_t)t * row_step] + 1]++;
    for (int l = 0; l < nl; ++l) lptr[l + 1] += lptr[l];
    std::vector<int> order((size_t)m);
    {
        std::vector<int64_t> cur(lptr.begin(), lptr.end() - 1);
        for (int t = 0; t < m; ++t) {
            const int i = row_start + t * row_step;
            order[(size_t)cur[(size_t)lvl[i]]++] = i;
        }
    }
    int maxlen = 0;
    int64_t total = 0;
    {
        std::atomic<int> amax(0);
        std::atomic<int64_t> atot(0);
        lane_parallel(m, [&](int64_t t0, int64_t t1) {
            int ml = 0;
            int64_t tl = 0;
            for (int64_t t = t0; t < t1; ++t) {
                const int i = order[(size_t)t];
                int c = 0;
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) c += Aj[p] != i;
                ml = std::max(ml, c);
                tl += c;
            }
            atot += tl;
            int cur = amax.load();
            while (ml > cur && !amax.compare_exchange_weak(cur, ml)) {}
        });
        maxlen = amax.load();
        total = atot.load();
    }
    P.max_offdiag = maxlen;
    int K = 0;
    const int L = lane_geometry(maxlen, want_L, K);
    if (!L) return 1;
    const int RPW = 64 / L;
    P.L = L; P.K = K; P.RPW = RPW;
    P.level_grp.assign((size_t)nl + 1, 0);
    for (int l = 0; l < nl; ++l) {
        const int64_t w = (lptr[l + 1] - lptr[l] + RPW - 1) / RPW;
        P.level_grp[l + 1] = P.level_grp[l] + w;
        P.max_level_groups = std::max(P.max_level_groups, w);
    }
    P.ngroups = P.level_grp[nl];
    P.n_slots = P.ngroups * K * 64;
    // padding inside the rows (a few long rows set K for everybody): give up when the rows' slots exceed 4x the entries
    // (+ 8 per row: short rows are fine); the padding of every level to whole groups is at most one group per level
    if ((int64_t)K * L * m > 4 * total + (int64_t)8 * L * m || P.n_slots >= ((int64_t)1 << 33)) return 1;
    if (P.ngroups >= ((int64_t)1 << 30)) return 1;
    if (fill) {
        plan_fill(P.cols, (size_t)P.n_slots, (int)LANE_NONE);
        plan_fill(P.vals, (size_t)P.n_slots * tsize, (unsigned char)0);
        P.rdiag.assign((size_t)P.ngroups * RPW * tsize, 0);
    }
    P.rid.assign((size_t)P.ngroups * RPW, -1);
    P.gate.assign((size_t)P.ngroups, -1);
    std::vector<int> gate_lvl((size_t)P.ngroups, -1);
    // latest early operand of every visited row (-1: none): where a group has no operand two levels down, an operand OF one of its
    // operands serves as the gate (stencils: every early operand of a row sits exactly one level below it)
    std::vector<int> best_dep((size_t)n, -1);
    lane_parallel(m, [&](int64_t t0, int64_t t1) {
        for (int64_t t = t0; t < t1; ++t) {
            const int i = row_start + (int)t * row_step, ti = vis[i];
            int bl = -1;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int j = Aj[p];
                if (j == i || j < 0 || j >= n || vis[j] < 0 || vis[j] >= ti) continue;
                if (lvl[j] > bl) { bl = lvl[j]; best_dep[(size_t)i] = j; }
            }
        }
    });
    std::vector<int64_t> ne((size_t)nl, 0), no((size_t)nl, 0);
    lane_parallel(nl, [&](int64_t l0, int64_t l1) {
        for (int64_t l = l0; l < l1; ++l) {
            const int mylevel = (int)l;
            int64_t e_cnt = 0, o_cnt = 0;
            for (int64_t q = lptr[l]; q < lptr[l + 1]; ++q) {
                const int64_t rel = q - lptr[l];
                const int64_t g = P.level_grp[l] + rel / RPW;
                const int r = (int)(rel % RPW);
                const int i = order[(size_t)q], ti = vis[i];
                int e = 0;
                const unsigned char *dptr = nullptr;
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                    const int j = Aj[p];
                    if (j == i) { if (fill) dptr = Ax + (size_t)p * tsize; continue; }          // last stored diagonal wins
                    const int k = e / L, lane = r * L + e % L;
                    const size_t s = (size_t)((g * K + k) * 64 + lane);
                    ++e;
                    if (j < 0 || j >= n) continue;                                    // not a column of x: no product
                    const bool early = vis[j] >= 0 && vis[j] < ti;
                    if (early) {
                        int cand = j;
                        if (lvl[cand] > mylevel - 2) cand = best_dep[(size_t)j];          // one level down: its own latest operand is two or more down
                        if (cand >= 0 && lvl[cand] <= mylevel - 2 && lvl[cand] > gate_lvl[(size_t)g]) { gate_lvl[(size_t)g] = lvl[cand]; P.gate[(size_t)g] = cand; }
                    }
                    if (fill) {
                        P.cols[s] = j | (early ? LANE_EARLY : 0);
                        std::memcpy(&P.vals[s * tsize], Ax + (size_t)p * tsize, (size_t)tsize);
                    }
                    if (early) ++e_cnt; else ++o_cnt;
                }
                bool nodiag = true;
                if (dptr) {
                    if (tsize == 8) {
                        double d;
                        std::memcpy(&d, dptr, 8);
                        nodiag = !(d != 0.0);
                        const double rd = nodiag ? 0.0 : 1.0 / d;
                        std::memcpy(&P.rdiag[(size_t)(g * RPW + r) * 8], &rd, 8);
                    } else {
                        float d;
                        std::memcpy(&d, dptr, 4);
                        nodiag = !(d != 0.0f);
                        const float rd = nodiag ? 0.0f : 1.0f / d;
                        std::memcpy(&P.rdiag[(size_t)(g * RPW + r) * 4], &rd, 4);
                    }
                }
                P.rid[(size_t)(g * RPW + r)] = fill ? (i | (nodiag ? LANE_NODIAG : 0)) : i;
            }
            ne[(size_t)l] = e_cnt; no[(size_t)l] = o_cnt;
        }
    }, 1);
    for (int l = 0; l < nl; ++l) { P.n_early += ne[(size_t)l]; P.n_old += no[(size_t)l]; }
    return 0;
}

}  // namespace pamg
