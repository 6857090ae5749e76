// pamg_blane_plan.h -- host-side layout of the LANE-PARALLEL ("fast order") block Gauss-Seidel sweep on square-block BSR operators
// (plain C++, no HIP: the CPU suite compiles this header with g++ and replays the plan, tests/blane_emul.cpp).
//
// amg_core::block_gauss_seidel (relaxation.h:1242-1298): for every block row i in sweep order  rsum = b_i - sum_{j != i} A_ij x_j,
// x_i = Dinv_i rsum -- new values of the block rows visited before i, old values of the others.  The fast order keeps the ORDER OF
// BLOCK ROWS (the same dependency DAG over the block graph, the same iterates in exact arithmetic) and gives up the order of the
// additions inside a block row: L lanes of a wave share a block row, every lane holds K of its off-diagonal BLOCKS (bs x bs values,
// its own gemv with x_j in registers), the bs partial sums are added across the lanes, the first bs lanes of the row apply one row of
// Dinv_i each.  Agrees with the reference to rounding (like the scalar fast order); the order-exact block kernels stay for order =
// 'exact', for the BSR point sweep (amg_core::bsr_gauss_seidel) and for block sizes the kernel is not compiled for.
// Hand-off: every block row is written once per sweep, so the sentinel protocol of the scalar sweeps applies component by component
// (xs[i * bs + c]); a consumer takes x_j when none of its bs components is the sentinel any more.
//
// Layout ("groups": the work of one wave; RPW = 64 / L block rows each; levels padded to whole groups):
//   cols [(g * K + k) * 64 + lane]               block column | EARLY (bit 31) | NONE (bit 30: padding)
//   vals [(((g * K + k) * bs^2) + e) * 64 + lane] entry e = r * bs + c of that block (lanes contiguous: every load of a wave is one run)
//   rid  [g * RPW + r]                            block row (-1: dummy)
//   gate [g]                                      "gate" of the group (pamg_lane_plan.h): the block column of its early operand with the highest
//                                                 dependency level among those at least two levels below the group's own (-1: none) -- a wave
//                                                 that runs ahead polls this ONE value until the sweep is a level away, then all its operands
// with lane = r * L + q and the block row's off-diagonal blocks q' = 0, 1, ... (storage order) at k = q' / L, q = q' % L.  Diagonal
// blocks are not stored (block_gauss_seidel skips them: relaxation.h:1268-1271).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pamg_lane_plan.h"
#include "pamg_tile_plan.h"

namespace pamg {

struct BlanePlan {
    int L = 0, K = 0, RPW = 0, bs = 0;
    int nlevels = 0;
    int64_t ngroups = 0, nslots = 0, max_level_groups = 0;
    std::vector<int> cols, rid, gate;
    std::vector<unsigned char> vals;          // nslots * bs^2 values
    int64_t n_early = 0, n_old = 0;
    bool symmetric = true;
};

// bAp / bAj / bAx: the block CSR (n_brow block rows, blocks row-major bs x bs, tsize bytes per value).
// Returns 0, or 1 when the form does not apply.
inline int build_blane_plan(int n_brow, const int *bAp, const int *bAj, const unsigned char *bAx, int tsize, int bs, int row_start, int row_stop, int row_step,
                            BlanePlan &P, int want_L = 0)
{
    P = BlanePlan();
    P.bs = bs;
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (bs < 2 || sweep_levels(n_brow, bAp, bAj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    if (m <= 0 || nl <= 0 || n_brow > LANE_MASK) return 1;
    P.nlevels = nl;
    int maxlen = 0;
    int64_t total = 0;
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step;
        int c = 0;
        for (int p = bAp[i]; p < bAp[i + 1]; ++p) c += bAj[p] != i;
        maxlen = std::max(maxlen, c);
        total += c;
    }
    // as many lanes per block row as it has off-diagonal blocks (a power of two, 4 .. 64): one block per lane (K = 1) wherever a row fits a wave,
    // two where it does not (2 blocks of 6 x 6 per lane are 72 values in registers: the limit); want_L > 0 forces a wider row
    int L = 4;
    while (L < 64 && (L < maxlen || L < want_L)) L *= 2;
    const int K = std::max(1, (maxlen + L - 1) / L);
    if (K > 2) return 1;
    const int RPW = 64 / L, bs2 = bs * bs;
    P.L = L; P.K = K; P.RPW = RPW;
    std::vector<int64_t> lptr((size_t)nl + 1, 0), lgrp((size_t)nl + 1, 0);
    for (int t = 0; t < m; ++t) lptr[(size_t)lvl[row_start + (int64_t)t * row_step] + 1]++;
    for (int l = 0; l < nl; ++l) {
        const int64_t w = (lptr[(size_t)l + 1] + RPW - 1) / RPW;
        P.max_level_groups = std::max(P.max_level_groups, w);
        lgrp[(size_t)l + 1] = lgrp[(size_t)l] + w;
        lptr[(size_t)l + 1] += lptr[(size_t)l];
    }
    P.ngroups = lgrp[(size_t)nl];
    P.nslots = P.ngroups * K * 64;
    if ((int64_t)K * L * m > 4 * total + (int64_t)8 * L * m || P.nslots * bs2 >= ((int64_t)1 << 33) || P.ngroups >= ((int64_t)1 << 30)) return 1;
    P.cols.assign((size_t)P.nslots, LANE_NONE);
    P.vals.assign((size_t)P.nslots * bs2 * tsize, 0);
    P.rid.assign((size_t)P.ngroups * RPW, -1);
    P.gate.assign((size_t)P.ngroups, -1);
    std::vector<int> gate_lvl((size_t)P.ngroups, -1), best_dep((size_t)n_brow, -1);
    // latest early operand of every visited block row: where a group has no operand two levels down, an operand OF one of its operands is the gate
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step;
        int bl = -1;
        for (int p = bAp[i]; p < bAp[i + 1]; ++p) {
            const int j = bAj[p];
            if (j == i || j < 0 || j >= n_brow || vis[j] < 0 || vis[j] >= t) continue;
            if (lvl[j] > bl) { bl = lvl[j]; best_dep[(size_t)i] = j; }
        }
    }
    std::vector<int64_t> cur(lptr.begin(), lptr.end() - 1);
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step, l = lvl[i];
        const int64_t q = cur[(size_t)l]++ - lptr[(size_t)l];
        const int64_t g = lgrp[(size_t)l] + q / RPW;
        const int r = (int)(q % RPW);
        P.rid[(size_t)(g * RPW + r)] = i;
        int e = 0;
        for (int p = bAp[i]; p < bAp[i + 1]; ++p) {
            const int j = bAj[p];
            if (j == i) continue;
            const int k = e / L, lane = r * L + e % L;
            const size_t s = (size_t)((g * K + k) * 64 + lane);
            ++e;
            if (j < 0 || j >= n_brow) continue;
            const bool early = vis[j] >= 0 && vis[j] < t;
            P.cols[s] = j | (early ? LANE_EARLY : 0);
            if (early) {
                ++P.n_early;
                int cand = j;
                if (lvl[cand] > l - 2) cand = best_dep[(size_t)j];
                if (cand >= 0 && lvl[cand] <= l - 2 && lvl[cand] > gate_lvl[(size_t)g]) { gate_lvl[(size_t)g] = lvl[cand]; P.gate[(size_t)g] = cand; }
            } else ++P.n_old;
            for (int ee = 0; ee < bs2; ++ee)
                std::memcpy(&P.vals[(((size_t)(g * K + k) * bs2 + ee) * 64 + lane) * tsize], bAx + ((size_t)p * bs2 + ee) * tsize, (size_t)tsize);
        }
    }
    return 0;
}

}  // namespace pamg
