// pamg_wg_plan.h -- host-side layout of the WORKGROUP-RESIDENT fast-order Gauss-Seidel / SOR sweep (plain C++, no HIP: the
// CPU suite compiles this header with g++ and replays the plan, tests/wg_emul.cpp).
//
// Why.  The lane form (pamg_lane_plan.h) hands every new value over through memory: 0.7 us per dependency level inside one
// XCD's L2, 1.05 us across the chip -- the price of a cross-CU hand-off on this part (MI355X_MICROARCH.md, handoff-1to1), however
// short the arithmetic behind it is made.  Only a hand-off that stays inside ONE CU is cheaper: LDS.  So the
// narrow levels of a hierarchy (tens of rows per dependency level: the SA levels below the first) are swept by a few
// WORKGROUPS that keep the iterate itself in LDS:
//   * the visited rows are cut into G contiguous TILES of the visit order, one persistent workgroup of 16 waves each
//     (one CU); a tile holds the x values of ITS rows in LDS, x_old to begin with, every row's new value written over the old
//     one when the row is done -- Gauss-Seidel in shared memory, no flags, no sentinels, no polling inside a tile;
//   * a tile walks its rows dependency level after dependency level, an LDS-only barrier between levels; the rows of a
//     level are dealt to the waves in ROUNDS of NW x (64 / L) rows, L lanes per row, K entry slots per lane, the row sums by the
//     butterfly of the lane form (same arithmetic: the reference's order of rows -- amg_core::gauss_seidel, relaxation.h:48-76;
//     sor_gauss_seidel :116-145 -- other association inside a row, x (1 / a_ii) instead of the division);
//   * operands outside the tile: a NEW value of an earlier tile is polled in the sentinel-filled hand-off buffer of the
//     lane form (rows that other tiles read publish there); an OLD value (a later tile's row, a row the sweep does not visit)
//     is read from x.  With contiguous tiles a dependency chain crosses tile boundaries G - 1 times per sweep, and an early
//     operand never comes from a LATER tile: tile k only ever waits for tiles < k (deadlock-free with all G workgroups resident).
//
// Layout.  Groups (one wave's work in one round) are numbered tile after tile, level after level, round after round, wave after
// wave; everything follows from the group number:
//   cols [(g * K + k) * 64 + lane]   IN-TILE (bit 31): low bits = position of the operand in the tile's LDS array
//                                    NONE (bit 30): padding, no product
//                                    CROSS (bit 29): new value of an earlier tile: poll xs[column]
//                                    else: column of an OLD value in x
//   vals [(g * K + k) * 64 + lane]   a_ij
//   rec  [g * RPW + r]               {row | NODIAG (bit 30) | PUBLISH (bit 29: another tile reads this row) | -1 for a dummy row,
//                                     position of the row in the tile's LDS array, 1 / a_ii}
//   rflag[round]                     bit 0: the round starts a dependency level of its tile (barrier first); bits 8..15: waves of the round
//                                    that hold a row at all; bits 16..19: entry slots per lane the round's longest row needs (the kernel
//                                    asks memory for nothing beyond either: most rounds of a narrow schedule are far from full)
// with lane = r * L + i and the row's off-diagonal entries e = 0, 1, ... (storage order) at k = e / L, i = e % L.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pamg_lane_plan.h"

namespace pamg {

constexpr int WG_INTILE = (int)0x80000000u;
constexpr int WG_NONE = 0x40000000;
constexpr int WG_CROSS = 0x20000000;
constexpr int WG_MASK = 0x1FFFFFFF;
constexpr int WG_NODIAG = 0x40000000;         // in rec.rid
constexpr int WG_PUBLISH = 0x20000000;        // in rec.rid
constexpr int WG_NW = 16;                     // waves per workgroup
constexpr int WG_MAX_TILES = 8;

struct WgPlan {
    int L = 0, K = 0, RPW = 0, G = 0;
    int tile_rows = 0;                        // LDS values per tile (the largest tile)
    int nlevels = 0;
    int64_t ngroups = 0, nrounds = 0;
    std::vector<int> cols;                    // ngroups * K * 64
    std::vector<unsigned char> vals;          // ngroups * K * 64 values
    std::vector<int> rid, lpos;               // ngroups * RPW
    std::vector<unsigned char> rdiag;         // ngroups * RPW values
    std::vector<int> rflag;                   // nrounds
    std::vector<int> tile_round;              // [G + 1] round range of each tile
    std::vector<int> tile_vis0;               // [G + 1] visit range of each tile
    int64_t n_intile = 0, n_cross = 0, n_old = 0, n_publish = 0;
};

// vis / lvl / m / nl: the analysis of the sweep (sweep_levels, pamg_tile_plan.h).  max_tile_rows: LDS capacity in values.
// Returns 0, or 1 when the form does not apply (rows too long, too many tiles, an index does not fit its field).
inline int build_wg_plan(int n, const int *Ap, const int *Aj, const unsigned char *Ax, int tsize, int row_start, int row_step,
                         int m, int nl, const std::vector<int> &vis, const std::vector<int> &lvl, int max_tile_rows, WgPlan &P, int force_tiles = 0)
{
    P = WgPlan();
    P.nlevels = nl;
    if (m <= 0 || nl <= 0 || max_tile_rows < 64 || n > WG_MASK) return 1;
    int G = (int)(((int64_t)m + max_tile_rows - 1) / max_tile_rows);
    if (force_tiles > G) G = force_tiles;
    if (G > WG_MAX_TILES || G > m) return 1;
    P.G = G;
    P.tile_vis0.assign((size_t)G + 1, 0);
    for (int k = 0; k <= G; ++k) P.tile_vis0[(size_t)k] = (int)((int64_t)m * k / G);
    for (int k = 0; k < G; ++k) P.tile_rows = std::max(P.tile_rows, P.tile_vis0[(size_t)k + 1] - P.tile_vis0[(size_t)k]);
    if (P.tile_rows > max_tile_rows) return 1;
    auto tile_of = [&](int t) { int k = (int)((int64_t)t * G / m); while (t < P.tile_vis0[(size_t)k]) --k; while (t >= P.tile_vis0[(size_t)k + 1]) ++k; return k; };
    int maxlen = 0;
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step;
        int c = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) c += Aj[p] != i;
        maxlen = std::max(maxlen, c);
    }
    int K = 0;
    const int L = lane_geometry(maxlen, 0, K);
    if (!L) return 1;
    const int RPW = 64 / L, RPR = WG_NW * RPW;                       // rows per wave, rows per round
    P.L = L; P.K = K; P.RPW = RPW;
    // rows of every tile by level (visit order inside a level); rounds per (tile, level)
    std::vector<std::vector<int>> rows((size_t)G);                  // per tile: rows in (level, visit) order
    std::vector<std::vector<int>> lcount((size_t)G, std::vector<int>((size_t)nl, 0));
    for (int t = 0; t < m; ++t) lcount[(size_t)tile_of(t)][(size_t)lvl[row_start + t * row_step]]++;
    P.tile_round.assign((size_t)G + 1, 0);
    int64_t nrounds = 0;
    for (int k = 0; k < G; ++k) {
        P.tile_round[(size_t)k] = (int)nrounds;
        for (int l = 0; l < nl; ++l) nrounds += (lcount[(size_t)k][(size_t)l] + RPR - 1) / RPR;
    }
    P.tile_round[(size_t)G] = (int)nrounds;
    if (nrounds >= ((int64_t)1 << 24)) return 1;
    P.nrounds = nrounds;
    P.ngroups = nrounds * WG_NW;
    const int64_t nslots = P.ngroups * K * 64;
    P.cols.assign((size_t)nslots, WG_NONE);
    P.vals.assign((size_t)nslots * tsize, 0);
    P.rid.assign((size_t)P.ngroups * RPW, -1);
    P.lpos.assign((size_t)P.ngroups * RPW, 0);
    P.rdiag.assign((size_t)P.ngroups * RPW * tsize, 0);
    P.rflag.assign((size_t)nrounds, 0);
    // rows read by another tile publish
    std::vector<char> pub((size_t)n, 0);
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step, ki = tile_of(t);
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i || j < 0 || j >= n || vis[j] < 0 || vis[j] >= t) continue;
            if (tile_of(vis[j]) != ki) pub[(size_t)j] = 1;
        }
    }
    for (int k = 0; k < G; ++k) {
        const int t0 = P.tile_vis0[(size_t)k], t1 = P.tile_vis0[(size_t)k + 1];
        // bucket the tile's rows by level
        std::vector<int> lptr((size_t)nl + 1, 0);
        for (int l = 0; l < nl; ++l) lptr[(size_t)l + 1] = lptr[(size_t)l] + lcount[(size_t)k][(size_t)l];
        std::vector<int> order((size_t)(t1 - t0)), cur(lptr.begin(), lptr.end() - 1);
        for (int t = t0; t < t1; ++t) { const int i = row_start + t * row_step; order[(size_t)cur[(size_t)lvl[i]]++] = i; }
        int64_t round = P.tile_round[(size_t)k];
        for (int l = 0; l < nl; ++l) {
            const int cnt = lcount[(size_t)k][(size_t)l];
            if (!cnt) continue;
            P.rflag[(size_t)round] |= 1;                              // barrier before the first round of a level
            for (int rr = 0; rr < (cnt + RPR - 1) / RPR; ++rr) {
                const int here = std::min(RPR, cnt - rr * RPR);       // rows of this round: dealt over the waves first
                P.rflag[(size_t)(round + rr)] |= std::min(here, WG_NW) << 8;
            }
            for (int q = 0; q < cnt; ++q) {
                // rows of a level are dealt round-robin over the waves first (every wave busy), then over the rows of a wave
                const int rr = q / RPR, qq = q % RPR, w = qq % WG_NW, r = qq / WG_NW;
                const int64_t g = (round + rr) * WG_NW + w;
                const int i = order[(size_t)(lptr[(size_t)l] + q)], ti = vis[i];
                int e = 0;
                const unsigned char *dptr = nullptr;
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                    const int j = Aj[p];
                    if (j == i) { dptr = Ax + (size_t)p * tsize; continue; }
                    const int kk = e / L, lane = r * L + e % L;
                    const size_t s = (size_t)((g * K + kk) * 64 + lane);
                    ++e;
                    if (j < 0 || j >= n) continue;
                    const int tj = vis[j];
                    int code;
                    if (tj >= t0 && tj < t1) { code = WG_INTILE | (tj - t0); ++P.n_intile; }
                    else if (tj >= 0 && tj < ti) { code = WG_CROSS | j; ++P.n_cross; }
                    else { code = j; ++P.n_old; }
                    P.cols[s] = code;
                    std::memcpy(&P.vals[s * tsize], Ax + (size_t)p * tsize, (size_t)tsize);
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
                {
                    const int kneed = std::max(1, (e + L - 1) / L);
                    int &f = P.rflag[(size_t)(round + rr)];
                    if (((f >> 16) & 15) < kneed) f = (f & ~(15 << 16)) | (kneed << 16);
                }
                P.rid[(size_t)(g * RPW + r)] = i | (nodiag ? WG_NODIAG : 0) | (pub[(size_t)i] ? WG_PUBLISH : 0);
                P.lpos[(size_t)(g * RPW + r)] = ti - t0;
                if (pub[(size_t)i]) ++P.n_publish;
            }
            round += (cnt + RPR - 1) / RPR;
        }
    }
    return 0;
}

}  // namespace pamg
