// pamg_tile_plan.h -- host-side planning of the TILED order-exact sweep (plain C++, no HIP: the CPU
// test-suite compiles this header with g++ and replays the plan, tests/test_tile_plan.py).
//
// Why tiles.  An order-exact Gauss-Seidel sweep (amg_core::gauss_seidel, relaxation.h:48-76) is a DAG:
// row i runs after every connected row visited before it.  The level-scheduled granular sweep pays
// one cross-workgroup hand-off (>= 1.5 us through L2/HBM) per dependency LEVEL.  Here the visited
// rows are cut into G contiguous chunks of the visit order ("tiles"), ONE persistent workgroup per
// tile.  A workgroup walks the rows of its tile level after level ("steps": at most 64 rows of one
// level, mutually independent); new values needed by a later step of the SAME tile travel through
// an LDS ring (one LDS round trip per level), only edges that cross tiles use the global hand-off
// buffer.  With contiguous chunks of a banded operator tile k depends on tile k-1 (and rarely
// further back), so the tiles run as a skewed pipeline: tile k settles one hand-off latency behind
// tile k-1 -- the hand-off latency is paid once per tile on the critical path instead of once per level.
//
// Device layout ("step blocks", pack_tile_blocks): every step is ONE fixed-size block of 1-KiB chunks
//   [ lists: NCH chunks | entry values T: NV chunks | 64 row records of 16 bytes ]
// so a loader wave streams a tile with LDS-DMA (1 KiB per instruction) and needs no descriptors.
//   lists = header {rows, tile-local index of the first row, old | handoff << 16, local} (16 bytes), then
//           "local" items    entry position | ring slot << 16  4 bytes: x_j is a NEW value of THIS tile (LDS ring)
//           (padded to an even count), then
//           "old" items      {entry position, column j}        8 bytes: x_j is the value from before the sweep
//           "hand-off" items {entry position, column j}        8 bytes: x_j is a NEW value of ANOTHER tile (poll xs[j])
//   so each consumer walks a dense list of exactly its own work (the gather wave the first two, the compute
//   wave the third) instead of classifying every entry.
//   values = a_ij in storage order of the step's rows; the diagonal entry holds +0 (s + (+0) == s for every s
//           the sums can hold), so a row sum is a plain in-order walk.
//   row record = {T diag @0, int (original row | publish bit 31) @8, int (first entry | entries << 16) @12}.
//
// Entry codes of the flat plan (TilePlan::Aj, consumed by the packer):
//   bit31 | bit30   meaning                      low 30 bits
//     0       0     OLD value   x[j]             j
//     1       0     NEW value, global hand-off   j      (poll xs[j])
//     0       1     diagonal (staged as +0)      j
//     1       1     NEW value, same tile         LDS ring slot
#pragma once
#include "pamg_host_threads.h"
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace pamg {

struct TileStep { int r0, r1, p0, p1; };      // stored rows [r0,r1), scheduled entries [p0,p1)  (layout of int4)

struct TilePlan {
    int G = 0;                    // tiles (= persistent workgroups)
    int W = 0;                    // LDS ring slots (power of two)
    int cap = 0;                  // scheduled entries per step (one row longer than cap forms a step of its own)
    std::vector<int> step_old, step_glob, step_loc;   // [nsteps] items per list
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
// grain: items one thread should at least get (rows: 64; whole tiles: 1)
template <typename F>
inline void tp_parallel(int n, F fn, int grain = 64)
{
    const unsigned hw = std::max(1u, std::min(48u, pamg::host_cpus()));
    const int nt = (n < 4 * grain) ? 1 : (int)std::min<unsigned>(hw, (unsigned)(n / grain));
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
inline int tile_list_bytes(int n_old, int n_glob, int n_loc) { return 16 + 4 * ((n_loc + 1) & ~1) + 8 * (n_old + n_glob); }
#ifndef PAMG_TILE_SIM_LIMITS
constexpr int TILE_MAX_OLD = 512, TILE_MAX_GLOB = 256, TILE_MAX_LIST_BYTES = 12288, TILE_MAX_ENTRIES = 2048;   // per step
#else
constexpr int TILE_MAX_OLD = 1 << 20, TILE_MAX_GLOB = 1 << 20, TILE_MAX_LIST_BYTES = 1 << 28, TILE_MAX_ENTRIES = 1 << 20;   // design aid (tools/tile_sim.cpp): no limits
#endif

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

// Is the operator a three-band stencil on a lexicographic grid?  Looks at the backward offsets i - j of a sample of
// rows: the three most frequent ones must be 1 < nx < nx*ny with nx | nx*ny and carry (nearly) all backward entries.
inline bool tile_detect_grid(int n, const int *Ap, const int *Aj, int64_t &nx, int64_t &nxy)
{
    std::vector<std::pair<int64_t, int64_t>> hist;     // (offset, count), tiny
    int64_t total = 0;
    const int stride = std::max(1, n / 4096) | 1;          // odd: does not resonate with power-of-two grid lines
    for (int i = n / 2; i < n; i += stride) {
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int64_t d = (int64_t)i - Aj[p];
            if (d <= 0) continue;
            ++total;
            bool found = false;
            for (auto &h : hist) if (h.first == d) { h.second++; found = true; break; }
            if (!found) { if (hist.size() > 64) return false; hist.push_back({d, 1}); }
        }
    }
    if (hist.size() < 3 || total == 0) return false;
    std::sort(hist.begin(), hist.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
    int64_t d[3] = {hist[0].first, hist[1].first, hist[2].first};
    std::sort(d, d + 3);
    if (hist[0].second + hist[1].second + hist[2].second < total * 95 / 100) return false;
    if (d[0] != 1 || d[1] < 4 || d[2] % d[1] != 0 || d[2] / d[1] < 4) return false;
    nx = d[1]; nxy = d[2];
    return true;
}

// Build the plan from a finished analysis (vis / lvl of sweep_levels, m visited rows, nl levels).  G_want tiles
// (clipped to the number of visited rows), ring of W slots (power of two), at most cap entries and max_rows rows
// per step.
inline int build_tile_plan_from(int n, const int *Ap, const int *Aj, int row_start, int row_step, int m, int nl,
                                const std::vector<int> &vis, const std::vector<int> &lvl, int G_want, int W, int cap,
                                int max_rows, TilePlan &P, int partition = 0, const int *row_tile = nullptr)
{
    if (W < 64 || (W & (W - 1)) || cap < 2 || max_rows < 1) return 1;
    P = TilePlan();
    P.W = W; P.cap = cap; P.nlevels = nl;
    int G = std::max(1, std::min(G_want, std::max(1, m)));
    // tile of each visited row.  partition 0: contiguous chunks of the visit order, balanced by work (entries + a
    // per-row constant).  partition 1: when the operator is a three-band stencil on a lexicographic grid (offsets 1, nx,
    // nx*ny between connected rows), tiles are PENCILS -- all of x, ty lines, tz planes -- so a dependency chain crosses
    // a tile boundary every ty-th step in y and every tz-th step in z instead of at every plane; anything else falls
    // back to partition 0.
    std::vector<int> tile((size_t)n, -1);
    bool pencils = false;
    if (partition == 1 && m >= 512 && (row_step == 1 || row_step == -1) && m == n) {
        int64_t nx = 0, nxy = 0;
        if (tile_detect_grid(n, Ap, Aj, nx, nxy)) {
            const int64_t ny = nxy / nx, nz = ((int64_t)n + nxy - 1) / nxy;
            // lines per tile, split as evenly as the plane counts allow (powers of two)
            const double lines = std::max(1.0, (double)m / G / (double)nx);
            int64_t ty = 1, tz = 1;
            while (ty * tz * 2 <= lines + 0.5) { if (ty <= tz && ty * 2 <= ny) ty *= 2; else if (tz * 2 <= nz) tz *= 2; else if (ty * 2 <= ny) ty *= 2; else break; }
            const int64_t gy = (ny + ty - 1) / ty, gz = (nz + tz - 1) / tz;
            if (gy * gz <= (int64_t)G_want * 2 && gy * gz >= 1) {
                G = (int)(gy * gz);
                tp_parallel(m, [&](int tlo, int thi) {
                    for (int t = tlo; t < thi; ++t) {
                        const int i = row_start + t * row_step;
                        const int64_t p = row_step == 1 ? (int64_t)t : (int64_t)(m - 1 - t);   // lexicographic position of the row
                        const int64_t z = p / nxy, y = (p % nxy) / nx;
                        int64_t k = (z / tz) * gy + (y / ty);
                        if (row_step != 1) k = (int64_t)G - 1 - k;         // tiles numbered along the sweep
                        tile[i] = (int)k;
                    }
                });
                pencils = true;
            }
        }
    }
    // partition 2: the caller supplies the tile of every row (row_tile[i] in [0, G_want)), e.g. pencils in coordinates
    // inherited from a finer grid level
    if (partition == 2 && row_tile) {
        bool ok = true;
        for (int t = 0; t < m && ok; ++t) {
            const int i = row_start + t * row_step;
            if (row_tile[i] < 0 || row_tile[i] >= G_want) ok = false;
            else tile[i] = row_tile[i];
        }
        if (ok) { G = G_want; pencils = true; }
    }
    P.G = G;
    if (!pencils) {
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
    // stored order: tile-major, inside a tile by (level, visit order): two stable counting sorts
    std::vector<int> tcount((size_t)G + 1, 0);
    for (int t = 0; t < m; ++t) tcount[tile[row_start + t * row_step] + 1]++;
    for (int k = 0; k < G; ++k) tcount[k + 1] += tcount[k];
    std::vector<int> order((size_t)m);
    {
        std::vector<int> bytile((size_t)m), cur(tcount.begin(), tcount.end() - 1);
        for (int t = 0; t < m; ++t) {
            const int i = row_start + t * row_step;
            bytile[cur[tile[i]]++] = i;
        }
        tp_parallel(G, [&](int klo, int khi) {
            std::vector<int> cnt;
            for (int k = klo; k < khi; ++k) {
                const int t0 = tcount[k], t1 = tcount[k + 1];
                if (t0 >= t1) continue;
                int lmin = lvl[bytile[t0]], lmax = lmin;
                for (int t = t0; t < t1; ++t) { lmin = std::min(lmin, lvl[bytile[t]]); lmax = std::max(lmax, lvl[bytile[t]]); }
                cnt.assign((size_t)(lmax - lmin) + 2, 0);
                for (int t = t0; t < t1; ++t) cnt[lvl[bytile[t]] - lmin + 1]++;
                for (int l = 0; l <= lmax - lmin; ++l) cnt[l + 1] += cnt[l];
                for (int t = t0; t < t1; ++t) order[t0 + cnt[lvl[bytile[t]] - lmin]++] = bytile[t];
            }
        }, 1);
    }
    std::vector<int> pos((size_t)n, -1);      // stored position of a visited row
    tp_parallel(m, [&](int lo, int hi) { for (int r = lo; r < hi; ++r) pos[order[r]] = r; });
    P.Ap.assign((size_t)m + 1, 0);
    for (int r = 0; r < m; ++r) P.Ap[r + 1] = P.Ap[r] + (Ap[order[r] + 1] - Ap[order[r]]);
    const int nnz = P.Ap[m];
    // entry codes first (the step builder needs the list sizes).  A NEW value of the same tile is served by the
    // LDS ring when its slot (rj - tile base) mod W still holds it: the slot is next written by stored row rj + W,
    // which must not belong to an earlier step than the consumer's -- guaranteed when rj + W >= r (the consumer's
    // own stored position; its step starts at or before r).
    P.Aj.resize((size_t)nnz);
    P.src.resize((size_t)nnz);
    P.rid.assign(order.begin(), order.end());
    std::vector<unsigned char> publish((size_t)m, 0);
    std::vector<int> row_old((size_t)m, 0), row_glob((size_t)m, 0), row_loc((size_t)m, 0);
    std::atomic<int64_t> n_local(0), n_global(0);
    tp_parallel(m, [&](int rlo, int rhi) {
        int64_t nloc = 0, nglob = 0;
        for (int r = rlo; r < rhi; ++r) {
            const int i = order[r], ti = vis[i], k = tile[i];
            int q = P.Ap[r];
            for (int p = Ap[i]; p < Ap[i + 1]; ++p, ++q) {
                const int j = Aj[p];
                P.src[q] = p;
                if (j == i) { P.Aj[q] = j | TP_DIAG; continue; }
                if (j < 0 || j >= n) { P.Aj[q] = 0 | TP_DIAG; continue; }   // never stored by a valid operator: contributes +0
                const int tj = vis[j];
                if (tj >= 0 && tj < ti) {
                    const int rj = pos[j];
                    if (tile[j] == k && rj + W >= r) {
                        P.Aj[q] = ((rj - tcount[k]) & (W - 1)) | TP_EARLY | TP_DIAG;
                        row_loc[r]++;
                    } else {
                        P.Aj[q] = j | TP_EARLY;
                        publish[rj] = 1;          // several threads may store the same 1: benign
                        row_glob[r]++;
                    }
                } else {
                    P.Aj[q] = j;
                    row_old[r]++;
                }
            }
            nloc += row_loc[r]; nglob += row_glob[r];
        }
        n_local += nloc; n_global += nglob;
    });
    // steps: rows of one level of one tile, bounded by rows, entries and list sizes -- every tile on its own, strung together after
    struct TileSteps { std::vector<TileStep> steps; std::vector<int> level, no, ng, nl; };
    std::vector<TileSteps> per((size_t)G);
    tp_parallel(G, [&](int klo, int khi) {
        for (int k = klo; k < khi; ++k) {
            TileSteps &ts = per[(size_t)k];
            int r = tcount[k];
            const int rend = tcount[k + 1];
            while (r < rend) {
                const int L = lvl[order[r]];
                TileStep s{r, r, P.Ap[r], P.Ap[r]};
                int no = 0, ng = 0, nl = 0;
                while (s.r1 < rend && lvl[order[s.r1]] == L && s.r1 - s.r0 < max_rows) {
                    const int len = P.Ap[s.r1 + 1] - P.Ap[s.r1];
                    const int o2 = no + row_old[s.r1], g2 = ng + row_glob[s.r1], l2 = nl + row_loc[s.r1];
                    if (s.r1 > s.r0 && ((s.p1 - s.p0) + len > cap || o2 > TILE_MAX_OLD || g2 > TILE_MAX_GLOB ||
                                        tile_list_bytes(o2, g2, l2) > TILE_MAX_LIST_BYTES))
                        break;
                    s.p1 += len;
                    no = o2; ng = g2; nl = l2;
                    s.r1++;
                }
                ts.steps.push_back(s);
                ts.level.push_back(L);
                ts.no.push_back(no); ts.ng.push_back(ng); ts.nl.push_back(nl);
                r = s.r1;
            }
        }
    }, 1);
    P.tile_step.assign(1, 0);
    for (int k = 0; k < G; ++k) {
        const TileSteps &ts = per[(size_t)k];
        P.steps.insert(P.steps.end(), ts.steps.begin(), ts.steps.end());
        P.step_level.insert(P.step_level.end(), ts.level.begin(), ts.level.end());
        P.step_old.insert(P.step_old.end(), ts.no.begin(), ts.no.end());
        P.step_glob.insert(P.step_glob.end(), ts.ng.begin(), ts.ng.end());
        P.step_loc.insert(P.step_loc.end(), ts.nl.begin(), ts.nl.end());
        P.tile_step.push_back((int)P.steps.size());
    }
    P.n_local = n_local; P.n_global = n_global;
    for (int r = 0; r < m; ++r)
        if (publish[r]) { P.rid[r] |= TP_PUBLISH; P.n_publish++; }
    return 0;
}

inline int build_tile_plan(int n, const int *Ap, const int *Aj, int row_start, int row_stop, int row_step,
                           int G_want, int W, int cap, int max_rows, TilePlan &P, int partition = 0)
{
    std::vector<int> vis, lvl;
    int m = 0, nl = 0;
    if (sweep_levels(n, Ap, Aj, row_start, row_stop, row_step, vis, lvl, m, nl)) return 1;
    return build_tile_plan_from(n, Ap, Aj, row_start, row_step, m, nl, vis, lvl, G_want, W, cap, max_rows, P, partition);
}

// ---- fixed-size step blocks (the device layout)
constexpr int TILE_ROWS = 64;                 // rows per step (one lane of the compute wave each)
struct TileGeom {
    int NCH;                                  // 1-KiB chunks of lists per step
    int NV;                                   // 1-KiB chunks of entry values per step
    int tsize;                                // sizeof(T)
    int val_off() const { return 1024 * NCH; }
    int row_off() const { return 1024 * (NCH + NV); }
    int block_bytes() const { return 1024 * (NCH + NV + 1); }
    int chunks() const { return NCH + NV + 1; }                   // LDS-DMA instructions per step
    int slot_bytes() const { return block_bytes() + 2 * TILE_ROWS * tsize + 32; }   // LDS: block + b[64] + xold[64] + 4 time stamps
    int max_entries() const { return 1024 * NV / tsize; }
};

// smallest geometry whose blocks hold every step of the plan; false = a step does not fit (a row longer than
// TILE_MAX_ENTRIES, or with more old / hand-off operands than one step may carry)
inline bool tile_geometry(const TilePlan &P, int tsize, TileGeom &g, int *max_old = nullptr, int *max_glob = nullptr)
{
    int ment = 0, mlist = 16, mo = 0, mg = 0;
    for (size_t s = 0; s < P.steps.size(); ++s) {
        ment = std::max(ment, P.steps[s].p1 - P.steps[s].p0);
        mlist = std::max(mlist, tile_list_bytes(P.step_old[s], P.step_glob[s], P.step_loc[s]));
        mo = std::max(mo, P.step_old[s]);
        mg = std::max(mg, P.step_glob[s]);
    }
    if (max_old) *max_old = mo;
    if (max_glob) *max_glob = mg;
    if (ment > TILE_MAX_ENTRIES || mo > TILE_MAX_OLD || mg > TILE_MAX_GLOB || mlist > TILE_MAX_LIST_BYTES) return false;
    g.tsize = tsize;
    g.NCH = (mlist + 1023) / 1024;
    g.NV = std::max(1, (ment * tsize + 1023) / 1024);
    return true;
}

// Pack the plan into step blocks.  Ax = the operator's values in its own order (P.src maps scheduled entries to them),
// Aj_op = its column indices (to recognise the diagonal: last stored a_ii wins, relaxation.h:64-74).
template <typename T>
inline int pack_tile_blocks(const TilePlan &P, const TileGeom &g, const T *Ax, const int *Aj_op, std::vector<unsigned char> &out)
{
    if ((int)sizeof(T) != g.tsize) return 1;
    const size_t nsteps = P.steps.size();
    const size_t bb = (size_t)g.block_bytes();
    out.resize(nsteps * bb + 1024);           // + slack: a DMA chunk never reads past the allocation (every block is zeroed by the thread that packs it)
    std::memset(out.data() + nsteps * bb, 0, 1024);
    std::atomic<int> bad(0);
    tp_parallel(P.G, [&](int klo, int khi) {
        for (int k = klo; k < khi; ++k) {
            const int sa = P.tile_step[k], sb = P.tile_step[k + 1];
            if (sa >= sb) continue;
            const int tile_r0 = P.steps[sa].r0;
            for (int s = sa; s < sb; ++s) {
                const TileStep &st = P.steps[s];
                const int nrows = st.r1 - st.r0, nent = st.p1 - st.p0;
                const int no = P.step_old[s], ng = P.step_glob[s], nl = P.step_loc[s];
                unsigned char *blk = out.data() + (size_t)s * bb;
                std::memset(blk, 0, bb);
                if (nrows > TILE_ROWS || nent > g.max_entries() || tile_list_bytes(no, ng, nl) > 1024 * g.NCH) { bad = 1; continue; }
                int *hdr = reinterpret_cast<int *>(blk);
                unsigned *loc_items = reinterpret_cast<unsigned *>(hdr + 4);
                int *old_items = hdr + 4 + ((nl + 1) & ~1), *glob_items = old_items + 2 * no;
                T *vals = reinterpret_cast<T *>(blk + g.val_off());
                unsigned char *rows = blk + g.row_off();
                int io = 0, ig = 0, il = 0;
                for (int r = st.r0; r < st.r1; ++r) {
                    const int i = P.rid[r] & TP_MASK;
                    T d = T(0);
                    for (int q = P.Ap[r]; q < P.Ap[r + 1]; ++q) {
                        const int e = q - st.p0, c = P.Aj[q];
                        const bool early = c < 0, dg = (c & TP_DIAG) != 0;
                        if (dg && !early) {                            // diagonal (or an invalid column): +0
                            if (Aj_op[P.src[q]] == i) d = Ax[P.src[q]];
                            vals[e] = T(0);
                            continue;
                        }
                        vals[e] = Ax[P.src[q]];
                        if (early && dg) loc_items[il++] = (unsigned)e | ((unsigned)(c & TP_MASK) << 16);
                        else if (early) { glob_items[2 * ig] = e; glob_items[2 * ig + 1] = c & TP_MASK; ++ig; }
                        else { old_items[2 * io] = e; old_items[2 * io + 1] = c & TP_MASK; ++io; }
                    }
                    unsigned char *rec = rows + 16 * (size_t)(r - st.r0);
                    std::memcpy(rec, &d, sizeof(T));
                    const int rid = P.rid[r];
                    const int lohi = (P.Ap[r] - st.p0) | ((P.Ap[r + 1] - P.Ap[r]) << 16);
                    std::memcpy(rec + 8, &rid, 4);
                    std::memcpy(rec + 12, &lohi, 4);
                }
                if (io != no || ig != ng || il != nl) bad = 1;
                hdr[0] = nrows; hdr[1] = st.r0 - tile_r0; hdr[2] = no | (ng << 16); hdr[3] = nl;
            }
        }
    }, 1);
    return bad.load();
}

}  // namespace pamg
