// pamg_tile_plan.h -- host-side planning of the TILED order-exact sweep (plain C++, no HIP: the CPU
// test-suite compiles this header with g++ and replays the plan, tests/test_tile_plan.py).
//
// Why tiles.  An order-exact Gauss-Seidel sweep (amg_core::gauss_seidel, relaxation.h:48-76) is a DAG:
// row i runs after every connected row visited before it.  The level-scheduled granular sweep pays
// one cross-workgroup hand-off (>= 1.5 us through L2/HBM) per dependency LEVEL.  Here the visited
// rows are cut into G contiguous chunks of the visit order ("tiles"), ONE persistent workgroup per
// tile.  A workgroup walks the rows of its tile level after level ("steps": rows of one level,
// mutually independent); new values needed by a later step of the SAME tile travel through an LDS
// ring (one LDS round trip per level), only edges that cross tiles use the global hand-off buffer.
// With contiguous chunks of a banded operator tile k depends on tile k-1 (and rarely further back),
// so the tiles run as a skewed pipeline: tile k settles one hand-off latency behind tile k-1 and
// thereafter finds its cross-tile operands already published -- the hand-off latency is paid once
// per tile on the critical path instead of once per level.
//
// Entry codes (32-bit "column" of a scheduled entry):
//   bit31 | bit30   meaning                      low 30 bits
//     0       0     OLD value   x[j]             j
//     1       0     NEW value, global hand-off   j      (poll xs[j])
//     0       1     diagonal (staged as +0)      j
//     1       1     NEW value, same tile         LDS ring slot
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

namespace pamg {

struct TileStep { int r0, r1, p0, p1; };      // stored rows [r0,r1), scheduled entries [p0,p1)  (layout of int4)

struct TilePlan {
    int G = 0;                    // tiles (= persistent workgroups)
    int W = 0;                    // LDS ring slots (power of two)
    int cap = 0;                  // scheduled entries per step (one row longer than cap forms a step of its own)
    int nlevels = 0;
    bool symmetric = true;
    std::vector<int> rid;         // [m]   original row of stored row r | PUBLISH_BIT
    std::vector<int> Ap;          // [m+1] row pointers into the scheduled entry arrays
    std::vector<int> Aj;          // [nnz] entry codes
    std::vector<int> src;         // [nnz] position of the scheduled entry in the operator's own arrays
    std::vector<TileStep> steps;  // all steps, tile after tile
    std::vector<int> tile_step;   // [G+1] step range of each tile
    std::vector<int> step_level;  // [nsteps] dependency level of each step (diagnostics / replay order)
    int64_t n_local = 0, n_global = 0, n_publish = 0;   // statistics: early entries by kind, publishing rows
};

// fn(lo, hi) over [0, n) on a few host threads (planning is a one-time setup cost, but a 16.7M-row level is big)
template <typename F>
inline void tp_parallel(int n, F fn)
{
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int nt = (n < (1 << 15)) ? 1 : (int)hw;
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int lo = (int)((int64_t)n * t / nt), hi = (int)((int64_t)n * (t + 1) / nt);
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}

constexpr int TP_EARLY = (int)0x80000000u;
constexpr int TP_DIAG = 0x40000000;
constexpr int TP_MASK = 0x3FFFFFFF;
constexpr int TP_PUBLISH = (int)0x80000000u;  // in rid[]

// Dependency levels of the sweep i = row_start, row_start+row_step, ... (!= row_stop): level[i] for
// visited rows, vis[i] = visit index or -1.  Row i runs strictly after every connected row visited
// before it and strictly before every connected row visited after it (connection through a stored
// entry in EITHER direction, so structurally non-symmetric patterns are exact too).
inline int sweep_levels(int n, const int *Ap, const int *Aj, int row_start, int row_stop, int row_step,
                        std::vector<int> &vis, std::vector<int> &lvl, int &m_out, int &nlevels)
{
    if (row_step == 0) return 1;
    const long span = (long)row_stop - row_start;
    if (span % row_step != 0 || span / row_step < 0) return 1;
    const int m = (int)(span / row_step);
    m_out = m;
    nlevels = 0;
    vis.assign((size_t)n, -1);
    lvl.assign((size_t)n, 0);
    if (m == 0) return 0;
    const long last = (long)row_start + (long)(m - 1) * row_step;
    if (row_start < 0 || row_start >= n || last < 0 || last >= n) return 1;
    std::vector<int> pend((size_t)n, 0);
    for (int t = 0; t < m; ++t) vis[row_start + t * row_step] = t;
    int maxl = 0;
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step;
        int L = pend[i];
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i || j < 0 || j >= n) continue;
            const int tj = vis[j];
            if (tj >= 0 && tj < t) L = std::max(L, lvl[j] + 1);
        }
        lvl[i] = L;
        maxl = std::max(maxl, L);
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i || j < 0 || j >= n) continue;
            if (vis[j] > t) pend[j] = std::max(pend[j], L + 1);
        }
    }
    nlevels = maxl + 1;
    return 0;
}

// Build the plan from a finished analysis (vis / lvl of sweep_levels, m visited rows, nl levels).  G_want tiles
// (clipped to the number of visited rows), ring of W slots (power of two), at most cap entries and max_rows rows
// per step.
inline int build_tile_plan_from(int n, const int *Ap, const int *Aj, int row_start, int row_step, int m, int nl,
                                const std::vector<int> &vis, const std::vector<int> &lvl, int G_want, int W, int cap,
                                int max_rows, TilePlan &P)
{
    if (W < 64 || (W & (W - 1)) || cap < 2 || max_rows < 1) return 1;
    P = TilePlan();
    P.W = W; P.cap = cap; P.nlevels = nl;
    const int G = std::max(1, std::min(G_want, std::max(1, m)));
    P.G = G;
    // tile of each visited row: contiguous chunks of the visit order, balanced by work (entries + a
    // per-row constant)
    std::vector<int> tile((size_t)n, -1);
    {
        std::vector<int64_t> cum((size_t)m + 1, 0);
        for (int t = 0; t < m; ++t) {
            const int i = row_start + t * row_step;
            cum[t + 1] = cum[t] + (Ap[i + 1] - Ap[i]) + 4;
        }
        const int64_t tot = cum[m];
        std::vector<int64_t> thr((size_t)G);                 // tile k covers work [k*tot/G, (k+1)*tot/G)
        for (int k = 0; k < G; ++k) thr[k] = (int64_t)((__int128)tot * (k + 1) / G);
        int k = 0;
        for (int t = 0; t < m; ++t) {
            while (k + 1 < G && cum[t] >= thr[k]) ++k;
            tile[row_start + t * row_step] = k;
        }
    }
    // stored order: tile-major, inside a tile by (level, visit order)
    std::vector<int> tcount((size_t)G + 1, 0);
    for (int t = 0; t < m; ++t) tcount[tile[row_start + t * row_step] + 1]++;
    for (int k = 0; k < G; ++k) tcount[k + 1] += tcount[k];
    std::vector<int> order((size_t)m);
    // a tile's rows are contiguous in visit order: counting sort of each chunk by level (stable, so the visit
    // order survives inside a level)
    tp_parallel(G, [&](int klo, int khi) {
        std::vector<int> cnt;
        for (int k = klo; k < khi; ++k) {
            const int t0 = tcount[k], t1 = tcount[k + 1];
            if (t0 >= t1) continue;
            int lmin = lvl[row_start + t0 * row_step], lmax = lmin;
            for (int t = t0; t < t1; ++t) {
                const int L = lvl[row_start + t * row_step];
                lmin = std::min(lmin, L); lmax = std::max(lmax, L);
            }
            cnt.assign((size_t)(lmax - lmin) + 2, 0);
            for (int t = t0; t < t1; ++t) cnt[lvl[row_start + t * row_step] - lmin + 1]++;
            for (int l = 0; l <= lmax - lmin; ++l) cnt[l + 1] += cnt[l];
            for (int t = t0; t < t1; ++t) {
                const int i = row_start + t * row_step;
                order[t0 + cnt[lvl[i] - lmin]++] = i;
            }
        }
    });
    std::vector<int> pos((size_t)n, -1);      // stored position of a visited row
    for (int r = 0; r < m; ++r) pos[order[r]] = r;
    P.Ap.assign((size_t)m + 1, 0);
    for (int r = 0; r < m; ++r) P.Ap[r + 1] = P.Ap[r] + (Ap[order[r] + 1] - Ap[order[r]]);
    const int nnz = P.Ap[m];
    // steps
    P.tile_step.assign(1, 0);
    std::vector<int> step_first((size_t)m, 0);   // first stored row of the step a stored row belongs to
    for (int k = 0; k < G; ++k) {
        int r = tcount[k];
        const int rend = tcount[k + 1];
        while (r < rend) {
            const int L = lvl[order[r]];
            TileStep s{r, r, P.Ap[r], P.Ap[r]};
            while (s.r1 < rend && lvl[order[s.r1]] == L && s.r1 - s.r0 < max_rows) {
                const int len = P.Ap[s.r1 + 1] - P.Ap[s.r1];
                if (s.r1 > s.r0 && (s.p1 - s.p0) + len > cap) break;
                s.p1 += len;
                s.r1++;
            }
            for (int q = s.r0; q < s.r1; ++q) step_first[q] = s.r0;
            P.steps.push_back(s);
            P.step_level.push_back(L);
            r = s.r1;
        }
        P.tile_step.push_back((int)P.steps.size());
    }
    // entry codes
    P.Aj.resize((size_t)nnz);
    P.src.resize((size_t)nnz);
    P.rid.assign(order.begin(), order.end());
    std::vector<unsigned char> publish((size_t)m, 0);
    std::atomic<int64_t> n_local(0), n_global(0);
    tp_parallel(m, [&](int rlo, int rhi) {
        int64_t nloc = 0, nglob = 0;
        for (int r = rlo; r < rhi; ++r) {
            const int i = order[r], ti = vis[i], k = tile[i];
            const int first = step_first[r];
            int q = P.Ap[r];
            for (int p = Ap[i]; p < Ap[i + 1]; ++p, ++q) {
                const int j = Aj[p];
                P.src[q] = p;
                if (j == i) { P.Aj[q] = j | TP_DIAG; continue; }
                if (j < 0 || j >= n) { P.Aj[q] = 0 | TP_DIAG; continue; }   // never stored by a valid operator: contributes +0
                const int tj = vis[j];
                if (tj >= 0 && tj < ti) {
                    const int rj = pos[j];
                    // same tile and the ring slot still holds row j's value when this step reads it: slot
                    // (rj - tile base) mod W is next written by stored row rj + W, whose step must not
                    // precede this one
                    if (tile[j] == k && rj + W >= first) {
                        P.Aj[q] = ((rj - tcount[k]) & (W - 1)) | TP_EARLY | TP_DIAG;
                        nloc++;
                    } else {
                        P.Aj[q] = j | TP_EARLY;
                        publish[rj] = 1;          // several threads may store the same 1: benign
                        nglob++;
                    }
                } else {
                    P.Aj[q] = j;
                }
            }
        }
        n_local += nloc; n_global += nglob;
    });
    P.n_local = n_local; P.n_global = n_global;
    for (int r = 0; r < m; ++r)
        if (publish[r]) { P.rid[r] |= TP_PUBLISH; P.n_publish++; }
    return 0;
}

inline int build_tile_plan(int n, const int *Ap, const int *Aj, int row_start, int row_stop, int row_step,
                           int G_want, int W, int cap, int max_rows, TilePlan &P)
{
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (sweep_levels(n, Ap, Aj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    return build_tile_plan_from(n, Ap, Aj, row_start, row_step, m, nl, vis, lvl, G_want, W, cap, max_rows, P);
}

}  // namespace pamg
