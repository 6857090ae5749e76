// pamg_lanem_plan.h -- host-side plan of the MERGED lane-parallel Gauss-Seidel sweep (plain C++, no HIP: the CPU suite
// compiles this header with g++ and replays the plan, tests/lanem_emul.cpp).
//
// Why.  The lane-parallel sweep (pamg_lane_plan.h) keeps the reference's order of rows (amg_core::gauss_seidel,
// relaxation.h:48-76) and pays ONE hand-off through memory per dependency level of that order: level 1 of the 256^3
// SA hierarchy 2 241 levels x 1.01 us = 2.27 ms for 0.82 GB (0.045 of the HBM peak).  The per-hop price is the part's
// (MI355X_MICROARCH.md, handoff-1to1); what is NOT fixed by the reference's order is the NUMBER of hops: s consecutive
// dependency levels can be eliminated algebraically into one "super-level".
//
// The algebra.  With the rows of a group G of s consecutive levels, the sequential sweep computes
//     (D + L_in) x_G = b_G - L_out x_new - U x_old
// (L_in: entries whose column is an earlier row of the SAME group, L_out: earlier rows of earlier groups, U: rows visited
// later -- or never).  L_in is nilpotent (depth < s), so
//     x_G = T_G (b_G - L_out x_new - U x_old),   T_G = (D + L_in)^-1 = sum_{k<s} (-D^-1 L_in)^k D^-1 .
// Row i of that product is again a row of the lane form -- x_i = (b_i - sum_k v_k * operand_k) / a_ii -- with three kinds of
// operands: NEW values of rows of EARLIER groups (polled in the hand-off buffer, as before), OLD values (read from a
// snapshot of x taken before the sweep: a merged row reads old values of rows it is not adjacent to, so the waits no
// longer order those reads before the writes), and entries of b (rows of its own group).  Same iterates in exact
// arithmetic; in floating point the merged sweep differs from the sequential one by rounding (a few 1e-16 measured on SA
// levels; the per-row growth factor sum_r |T_ir| |a_ii| is computed at plan time and a group is CLOSED early -- down to
// s = 1, the unmerged row -- where it exceeds `growth_cap`, e.g. operators that are not diagonally dominant).
// The recursion used below (v = coefficient in "a_ii x_i = b_i - sum v * operand"): for every in-group operand r of row i
// with f = a_ir / a_rr:   v_i[b_r] += f,   v_i[op] -= f * v_r[op] for every operand of the (already merged) row r;
// a row r without a usable diagonal is left untouched by the reference (relaxation.h:72-74): its new value IS its old
// value, so it contributes a_ir to v_i[old x_r].
//
// Layout.  One row per wave (64 lanes share a row; rows of K * 64 operand slots, K = 1..LANEM_KMAX chosen PER ROW): group g
// = the g-th row in (super-level, dependency level, visit) order; its record {row | NODIAG, gate, 1 / a_ii, first 64-slot
// unit, K} gives the address of its slots:
//     cols[(unit + k) * 64 + lane]   column | EARLY (bit 31: poll the hand-off buffer) | NONE (bit 30: padding)
//                                           | BSRC (bit 29: the operand is b[column])
//     vals[(unit + k) * 64 + lane]   v
// A group only waits for groups of EARLIER super-levels (smaller numbers): the static wave assignment g = w, w + W, ...
// stays deadlock-free.  SOR is not merged (its coefficients depend on the relaxation parameter of the call).
#pragma once
#include "pamg_host_threads.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pamg_lane_plan.h"

namespace pamg {

constexpr int LANEM_BSRC = 0x20000000;        // operand read from b
constexpr int LANEM_MASK = 0x1FFFFFFF;        // column of a slot
constexpr int LANEM_KMAX = 8;                 // slots per lane: merged rows of up to 512 operands

struct LaneMPlan {
    int s_max = 0;
    int nlevels = 0;                          // dependency levels of the sweep (hand-offs of the unmerged form)
    int nsuper = 0;                           // super-levels = hand-offs of this form
    int rpw = 1;                              // rows per wave (1: 64 lanes per row; 2: 32 lanes per row, rows of a super-level paired by length)
    int64_t nrows = 0;                        // visited rows
    int64_t ngroups = 0;                      // groups (one wave each): ceil(rows of a super-level / rpw), super-level after super-level
    int64_t n_units = 0;                      // 64-slot units of cols / vals
    std::vector<int> unit;                    // [ngroups] first unit of the group
    std::vector<unsigned char> K;             // [ngroups] units of the group
    std::vector<int> rid;                     // [ngroups * rpw] row | LANE_NODIAG, -1 = dummy slot of an odd super-level
    std::vector<int> gate;                    // [ngroups] gate operand (pamg_lane_plan.h) in terms of super-levels, or -1
    std::vector<double> rdiag;                // [ngroups * rpw] 1 / a_ii (0: no usable diagonal)
    std::vector<int> super_of;                // [ngroups] super-level of the group
    std::vector<int64_t> super_grp;           // [nsuper + 1] group range of each super-level
    PlanVec<int> cols;
    PlanVec<double> vals;
    int64_t n_early = 0, n_old = 0, n_b = 0;  // operands by kind
    int64_t n_direct = 0;                     // off-diagonal entries of the visited rows (what the unmerged form holds)
    int max_len = 0;                          // longest merged row (operands)
    int closed_by_length = 0, closed_by_growth = 0;   // groups closed before s_max levels were in
    double max_growth = 0.0;                  // largest accepted growth factor
    int64_t max_super_groups = 0;             // rows of the widest super-level
};

namespace lanem_detail {

struct Spa {                                  // sparse accumulator over (kind, column); kinds 0 = early, 1 = old, 2 = b
    // a small open-addressing table (a merged row holds a few hundred operands at most): it stays in the L1 of the host core, where dense
    // arrays over all columns cost two cache misses per addition (level 1 of the 256^3 hierarchy: 4.3 s -> see DESIGN 3 round 6)
    static constexpr int CAP = 4096;          // slots; a row may fill half of them, longer rows are "too long" whatever the caller's cap
    std::vector<int> key;
    std::vector<double> val;
    std::vector<int> used;                    // occupied slots in first-touch order
    std::vector<int> touched;                 // the row's codes (column | kind bits), filled by finish()
    bool overflow = false;
    void init(int) { key.assign((size_t)CAP, (int)LANE_NONE); val.assign((size_t)CAP, 0.0); used.clear(); }
    void begin() { for (int h : used) key[(size_t)h] = (int)LANE_NONE; used.clear(); touched.clear(); overflow = false; }
    static int code(int kind, int col) { return col | (kind == 0 ? LANE_EARLY : kind == 2 ? LANEM_BSRC : 0); }
    static int kind_of(int c) { return (c & LANE_EARLY) ? 0 : (c & LANEM_BSRC) ? 2 : 1; }
    void add(int kind, int col, double v)
    {
        const int c = code(kind, col);
        unsigned h = ((unsigned)c * 2654435761u) >> 20;                  // 12 bits
        while (key[h] != (int)LANE_NONE && key[h] != c) h = (h + 1) & (CAP - 1);
        if (key[h] == c) { val[h] += v; return; }
        if ((int)used.size() >= CAP / 2) { overflow = true; return; }
        key[h] = c; val[h] = v; used.push_back((int)h);
    }
    double value_of(int c) const
    {
        unsigned h = ((unsigned)c * 2654435761u) >> 20;
        while (key[h] != c) h = (h + 1) & (CAP - 1);
        return val[h];
    }
    void finish() { touched.clear(); for (int h : used) touched.push_back(key[(size_t)h]); }
};

struct RowRef { int64_t off = 0; int len = 0; int arena = -1; };

}  // namespace lanem_detail

// Build the plan from a finished analysis of the sweep (vis / lvl of sweep_levels; m visited rows in nl levels).  Ax: the
// operator's values (f64).  s_max >= 1 levels per super-level at most.  Returns 0, or 1 when the form does not fit (a row
// longer than LANEM_KMAX * 64 operands even unmerged, index range) -- the caller keeps the unmerged lane form.
inline int build_lanem_plan(int n, const int *Ap, const int *Aj, const double *Ax, int row_start, int row_step, int m, int nl,
                            const std::vector<int> &vis, const std::vector<int> &lvl, int s_max, double growth_cap, LaneMPlan &P,
                            int len_cap = LANEM_KMAX * 64, int rpw = 1)
{
    using namespace lanem_detail;
    const bool timing_ = getenv("PAMG_TIMING") != nullptr;
    auto t_prev_ = std::chrono::steady_clock::now();
    auto lap_ = [&](const char *what) {
        if (!timing_) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pamg timing]     lanem plan: %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_prev_).count());
        t_prev_ = now;
    };
    P = LaneMPlan();
    P.s_max = s_max; P.nlevels = nl;
    if (m <= 0 || nl <= 0 || s_max < 1 || (rpw != 1 && rpw != 2)) return 1;
    if (n > LANEM_MASK) return 1;
    P.rpw = rpw; P.nrows = m;
    len_cap = std::max(1, std::min(len_cap, LANEM_KMAX * (64 / rpw)));
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
    // last stored diagonal of every row (relaxation.h:64-69)
    std::vector<double> diag((size_t)n, 0.0);
    lane_parallel(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i)
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) if (Aj[p] == (int)i) diag[(size_t)i] = Ax[p];
    });
    lap_("levels, order, diagonals");
    // ---- merged rows, window after window of 4 * s_max levels; windows are independent (a merged row only refers to merged rows of
    //      its own window) and are what the host threads share out.  Inside a window the levels are taken greedily: a super-level is
    //      closed when it holds s_max levels, or in front of a level whose merged rows would come out too long / too large
    const int wlev = 4 * s_max;
    const int nblocks = (nl + wlev - 1) / wlev;
    static const unsigned want_threads = [] { const char *e = getenv("PAMG_PLAN_THREADS"); return e ? (unsigned)atoi(e) : 0u; }();
    const unsigned hw = want_threads ? want_threads : std::max(1u, std::min(96u, pamg::host_cpus() / 2u + 1u));      // (two sweep directions are planned at once)
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)hw, (int64_t)nblocks, std::max<int64_t>(1, (int64_t)m / 4096)}));
    std::vector<PlanVec<int>> acode((size_t)nt);
    std::vector<PlanVec<double>> aval((size_t)nt);
    std::vector<RowRef> ref((size_t)n);
    std::vector<unsigned char> starts((size_t)nl, 0);          // level l opens a super-level
    std::vector<int> cl_len((size_t)nt, 0), cl_gr((size_t)nt, 0);
    std::vector<double> mg((size_t)nt, 0.0);
    std::atomic<int> unfit(0), next_blk(0);
    std::vector<double> t_res_((size_t)nt, 0.0), t_row_((size_t)nt, 0.0), t_end_((size_t)nt, 0.0);      // PAMG_TIMING: per-thread phases
    int64_t total_direct = 0;
    for (int t = 0; t < m; ++t) { const int i = row_start + t * row_step; total_direct += Ap[i + 1] - Ap[i]; }
    auto work = [&](int tid) {
        Spa spa;
        spa.init(n);
        PlanVec<int> &ac = acode[(size_t)tid];
        PlanVec<double> &av = aval[(size_t)tid];
        std::vector<std::pair<int, double>> sub;
        const auto tw0_ = std::chrono::steady_clock::now();
        {
            // room for this thread's share of the merged rows up front (a growing vector copies and re-faults what it holds at every doubling);
            // the windows are handed out one by one (below), so the share is an estimate: a thread that outgrows it pays the doubling
            const size_t want = (size_t)((double)total_direct / nt * 1.25 * (s_max >= 4 ? 3.2 : s_max == 3 ? 2.3 : s_max == 2 ? 1.7 : 1.05)) + 4096;
            ac.reserve(want);
            av.reserve(want);
        }
        const auto tw1_ = std::chrono::steady_clock::now();
        // windows differ by an order of magnitude (the sweep's wavefront grows and shrinks): first come, first served
        for (int blk = next_blk.fetch_add(1); blk < nblocks && !unfit.load(); blk = next_blk.fetch_add(1)) {
            const int lb0 = blk * wlev, lb1 = std::min(nl, lb0 + wlev);
            int l0 = lb0;                                     // first level of the open super-level
            starts[(size_t)lb0] = 1;
            for (int l = lb0; l < lb1; ++l) {
                if (l - l0 >= s_max) { l0 = l; starts[(size_t)l] = 1; }      // the open super-level is full
                for (int attempt = 0; attempt < 2; ++attempt) {
                    const size_t mark = ac.size();
                    bool too_long = false, too_large = false;
                    double level_growth = 0.0;
                    for (int64_t q = lptr[l]; q < lptr[l + 1]; ++q) {
                        const int i = order[(size_t)q], ti = vis[i];
                        spa.begin();
                        sub.clear();
                        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                            const int j = Aj[p];
                            if (j == i || j < 0 || j >= n) continue;
                            const bool early = vis[j] >= 0 && vis[j] < ti;
                            if (early && lvl[j] >= l0) sub.emplace_back(j, Ax[p]);
                            else spa.add(early ? 0 : 1, j, Ax[p]);
                        }
                        double growth = 1.0;
                        for (const auto &sr : sub) {
                            const int r = sr.first;
                            const double d = diag[(size_t)r];
                            if (!(d != 0.0)) { spa.add(1, r, sr.second); continue; }     // untouched row: its new value is its old value
                            const double f = sr.second / d;
                            spa.add(2, r, f);
                            const RowRef &rr = ref[(size_t)r];
                            const int *rc = acode[(size_t)rr.arena].data() + rr.off;
                            const double *rv = aval[(size_t)rr.arena].data() + rr.off;
                            for (int e = 0; e < rr.len; ++e) spa.add(Spa::kind_of(rc[e]), rc[e] & LANEM_MASK, -f * rv[e]);
                        }
                        spa.finish();
                        RowRef me;
                        me.arena = tid; me.off = (int64_t)ac.size(); me.len = (int)spa.touched.size();
                        // operands by (kind, column) -- kinds in the order old, b, early: lanes that read neighbouring columns of one array share a
                        // request, and the layout is reproducible.  (Neither the order nor the locality of the gathers moves the sweep much: first-touch
                        // order and this one measured the same on level 1 of the 256^3 hierarchy, early operands FIRST 1.74 -> 1.92 ms, perfectly local
                        // static gathers - 3 %: DESIGN 3, round 6.)
                        std::sort(spa.touched.begin(), spa.touched.end(), [](int x, int y) { return (unsigned)x < (unsigned)y; });
                        for (int c : spa.touched) {
                            const int kind = Spa::kind_of(c);
                            const double v = spa.value_of(c);
                            ac.push_back(c);
                            av.push_back(v);
                            if (kind == 2) growth += std::fabs(v);
                        }
                        if (!std::isfinite(growth)) growth = INFINITY;
                        ref[(size_t)i] = me;
                        if (me.len > len_cap || spa.overflow) too_long = true;
                        if (growth > growth_cap) too_large = true;
                        level_growth = std::max(level_growth, growth);
                    }
                    if (!too_long && !too_large) { mg[(size_t)tid] = std::max(mg[(size_t)tid], level_growth); break; }
                    if (l == l0) {
                        // the rows of this level as they are (no substitution): growth is 1; a row too long for the form ends it
                        if (too_long) unfit.store(1);
                        break;
                    }
                    // close the open super-level in front of this level and take the level again, unmerged
                    ac.resize(mark);
                    av.resize(mark);
                    if (too_long) cl_len[(size_t)tid]++; else cl_gr[(size_t)tid]++;
                    l0 = l;
                    starts[(size_t)l] = 1;
                }
                if (unfit.load()) break;
            }
        }
        if (timing_) {
            const auto tw2_ = std::chrono::steady_clock::now();
            t_res_[(size_t)tid] = std::chrono::duration<double>(tw1_ - tw0_).count();
            t_row_[(size_t)tid] = std::chrono::duration<double>(tw2_ - tw1_).count();
            t_end_[(size_t)tid] = std::chrono::duration<double>(tw2_ - t_prev_).count();
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    if (unfit.load()) return 1;
    if (timing_) {
        double sr = 0, mr = 0, sw = 0, mw = 0, me = 0, mn = 1e9;
        for (int t = 0; t < nt; ++t) { sr += t_res_[(size_t)t]; mr = std::max(mr, t_res_[(size_t)t]); sw += t_row_[(size_t)t]; mw = std::max(mw, t_row_[(size_t)t]); me = std::max(me, t_end_[(size_t)t]); mn = std::min(mn, t_end_[(size_t)t]); }
        fprintf(stderr, "[pamg timing]     lanem plan: %d threads, %d windows: reserve sum %.3f max %.3f s, rows sum %.3f max %.3f s, thread end (since lap) min %.3f max %.3f s\n", nt, nblocks, sr, mr, sw, mw, mn, me);
    }
    lap_("merged rows");
    for (int t = 0; t < nt; ++t) { P.closed_by_length += cl_len[(size_t)t]; P.closed_by_growth += cl_gr[(size_t)t]; P.max_growth = std::max(P.max_growth, mg[(size_t)t]); }
    // ---- super-levels, groups, units
    std::vector<int> sup_of_level((size_t)nl, 0);
    {
        int s = -1;
        for (int l = 0; l < nl; ++l) { if (starts[(size_t)l]) ++s; sup_of_level[(size_t)l] = s; }
        P.nsuper = s + 1;
    }
    // ---- groups: RPW rows of one super-level share a wave (64 / RPW lanes each).  With two rows per wave the rows of a super-level are paired by length
    //      (a pair is padded to the longer row's units; the order of rows INSIDE a super-level is free: they do not depend on each other)
    const int RPW = P.rpw, LPR = 64 / RPW;
    std::vector<int64_t> super_first((size_t)P.nsuper + 1, 0);
    for (int l = 0; l < nl; ++l) super_first[(size_t)sup_of_level[(size_t)l] + 1] = lptr[l + 1];
    auto klen = [&](int i) { const bool nodiag = !(diag[(size_t)i] != 0.0); const int len = nodiag ? 0 : ref[(size_t)i].len; return std::max(1, (len + LPR - 1) / LPR); };
    P.super_grp.assign((size_t)P.nsuper + 1, 0);
    for (int s = 0; s < P.nsuper; ++s) {
        const int64_t rows_s = super_first[s + 1] - super_first[s];
        P.super_grp[(size_t)s + 1] = P.super_grp[(size_t)s] + (rows_s + RPW - 1) / RPW;
        P.max_super_groups = std::max(P.max_super_groups, (rows_s + RPW - 1) / RPW);
    }
    const int64_t G = P.super_grp[(size_t)P.nsuper];
    P.ngroups = G;
    std::vector<int> srow((size_t)G * RPW, -1);                  // the row of every (group, row slot); -1 = dummy
    lane_parallel(P.nsuper, [&](int64_t s0, int64_t s1) {
        std::vector<int> rows;
        for (int64_t s = s0; s < s1; ++s) {
            rows.assign(order.begin() + super_first[(size_t)s], order.begin() + super_first[(size_t)s + 1]);
            if (RPW > 1) std::stable_sort(rows.begin(), rows.end(), [&](int x, int y) { return klen(x) < klen(y); });
            std::copy(rows.begin(), rows.end(), srow.begin() + P.super_grp[(size_t)s] * RPW);
        }
    }, 1);
    P.unit.assign((size_t)G, 0); P.K.assign((size_t)G, 1); P.rid.assign((size_t)G * RPW, -1); P.gate.assign((size_t)G, -1);
    P.rdiag.assign((size_t)G * RPW, 0.0); P.super_of.assign((size_t)G, 0);
    for (int s = 0; s < P.nsuper; ++s)
        for (int64_t g = P.super_grp[(size_t)s]; g < P.super_grp[(size_t)s + 1]; ++g) P.super_of[(size_t)g] = s;
    int64_t units = 0;
    for (int64_t g = 0; g < G; ++g) {
        int k = 1;
        for (int r = 0; r < RPW; ++r) {
            const int i = srow[(size_t)(g * RPW + r)];
            if (i < 0) continue;
            const bool nodiag = !(diag[(size_t)i] != 0.0);
            k = std::max(k, klen(i));
            P.rid[(size_t)(g * RPW + r)] = i | (nodiag ? LANE_NODIAG : 0);
            P.rdiag[(size_t)(g * RPW + r)] = nodiag ? 0.0 : 1.0 / diag[(size_t)i];
            P.max_len = std::max(P.max_len, nodiag ? 0 : ref[(size_t)i].len);
        }
        if (k > LANEM_KMAX) return 1;
        P.unit[(size_t)g] = (int)units;
        P.K[(size_t)g] = (unsigned char)k;
        units += k;
        if (units >= ((int64_t)1 << 27)) return 1;               // the record packs unit * 16 + K into 31 bits
    }
    P.n_units = units;
    // super-level of every visited row, and its latest early operand (the gate rule of pamg_lane_plan.h on super-levels)
    std::vector<int> sup_row((size_t)n, -1), best_dep((size_t)n, -1);
    lane_parallel(m, [&](int64_t q0, int64_t q1) { for (int64_t q = q0; q < q1; ++q) { const int i = order[(size_t)q]; sup_row[(size_t)i] = sup_of_level[(size_t)lvl[i]]; } });
    lane_parallel(m, [&](int64_t q0, int64_t q1) {
        for (int64_t q = q0; q < q1; ++q) {
            const int i = order[(size_t)q];
            if (!(diag[(size_t)i] != 0.0)) continue;
            const RowRef &rr = ref[(size_t)i];
            const int *rc = acode[(size_t)rr.arena].data() + rr.off;
            int bl = -1;
            for (int e = 0; e < rr.len; ++e)
                if (rc[e] & LANE_EARLY) {
                    const int j = rc[e] & LANEM_MASK;
                    if (sup_row[(size_t)j] > bl) { bl = sup_row[(size_t)j]; best_dep[(size_t)i] = j; }
                }
        }
    });
    lap_("groups, units, gates' inputs");
    plan_fill(P.cols, (size_t)units * 64, (int)LANE_NONE);
    plan_fill(P.vals, (size_t)units * 64, 0.0);
    std::atomic<int64_t> ne(0), no(0), nb(0), nd(0);
    lane_parallel(G, [&](int64_t g0, int64_t g1) {
        int64_t e_ = 0, o_ = 0, b_ = 0, d_ = 0;
        for (int64_t g = g0; g < g1; ++g) {
            const int mysup = P.super_of[(size_t)g];
            int gl = -1;
            for (int r = 0; r < RPW; ++r) {
                const int i = srow[(size_t)(g * RPW + r)];
                if (i < 0) continue;
                for (int p = Ap[i]; p < Ap[i + 1]; ++p) d_ += (Aj[p] != i && Aj[p] >= 0 && Aj[p] < n);
                if (P.rid[(size_t)(g * RPW + r)] & LANE_NODIAG) continue;
                const RowRef &rr = ref[(size_t)i];
                const int *rc = acode[(size_t)rr.arena].data() + rr.off;
                const double *rv = aval[(size_t)rr.arena].data() + rr.off;
                for (int e = 0; e < rr.len; ++e) {
                    const size_t sl = ((size_t)P.unit[(size_t)g] + (size_t)(e / LPR)) * 64 + (size_t)(r * LPR + e % LPR);
                    P.cols[sl] = rc[e];
                    P.vals[sl] = rv[e];
                    if (rc[e] & LANE_EARLY) {
                        ++e_;
                        int cand = rc[e] & LANEM_MASK;
                        if (sup_row[(size_t)cand] > mysup - 2) cand = best_dep[(size_t)cand];
                        if (cand >= 0 && sup_row[(size_t)cand] <= mysup - 2 && sup_row[(size_t)cand] > gl) { gl = sup_row[(size_t)cand]; P.gate[(size_t)g] = cand; }
                    } else if (rc[e] & LANEM_BSRC) ++b_;
                    else ++o_;
                }
            }
        }
        ne += e_; no += o_; nb += b_; nd += d_;
    });
    P.n_early = ne.load(); P.n_old = no.load(); P.n_b = nb.load(); P.n_direct = nd.load();
    lap_("slots filled");
    return 0;
}

}  // namespace pamg
