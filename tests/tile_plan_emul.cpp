// TEST INFRASTRUCTURE (CPU): replays a tile plan (pyamg_amd/csrc/pamg_tile_plan.h) exactly the way
// gs_tile_kernel consumes it -- from the packed step blocks (the device layout), LDS ring per tile with
// wrap-around, global hand-off buffer with the sentinel, publish flags, OLD values fetched `look` steps
// ahead of the compute wave -- under three interleavings of the tiles (level order; greedy: every tile
// runs as far ahead as its operands allow, lowest or highest tile first).  Used by
// tests/test_tile_plan.py to pin the host logic bit-for-bit against the oracle without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../pyamg_amd/csrc/pamg_tile_plan.h"

using namespace pamg;

namespace {
const uint64_t SENT = 0x7FF8DEADBEEF5A5Aull;
inline bool is_sent(double v) { uint64_t b; std::memcpy(&b, &v, 8); return b == SENT; }
inline double sent() { double v; std::memcpy(&v, &SENT, 8); return v; }

struct Emul {
    const TilePlan &P;
    TileGeom g;
    std::vector<unsigned char> blocks;       // the DEVICE layout: everything below reads only the blocks, like the kernel
    const double *b;
    std::vector<double> &x;          // live vector
    const std::vector<double> *snap;  // snapshot for OLD reads or nullptr
    std::vector<double> xs;
    std::vector<std::vector<double>> ring;
    struct Pre { std::vector<double> oldv, xo; };
    std::vector<std::vector<Pre>> pre;       // per tile: operands fetched ahead of the compute wave (steps next .. next+look-1)
    std::vector<int> next;                   // per tile: next step to run
    int look;
    int epi; double omega;
    int hazards = 0, bad = 0;

    Emul(const TilePlan &P_, const TileGeom &g_, const double *Ax, const int *Aj_op, const double *b_, std::vector<double> &x_,
         const std::vector<double> *snap_, int epi_, double omega_, int look_)
        : P(P_), g(g_), b(b_), x(x_), snap(snap_), look(look_), epi(epi_), omega(omega_)
    {
        bad = pack_tile_blocks<double>(P, g, Ax, Aj_op, blocks);
        xs.assign(x.size(), sent());
        ring.assign(P.G, std::vector<double>((size_t)P.W, std::nan("")));
        pre.resize(P.G);
        next.resize(P.G);
        for (int k = 0; k < P.G; ++k) {
            next[k] = P.tile_step[k];
            for (int s = next[k]; s < std::min(next[k] + look, P.tile_step[k + 1]); ++s) pre[k].push_back(prefetch(s));
        }
    }
    const unsigned char *blk(int s) const { return blocks.data() + (size_t)s * g.block_bytes(); }
    const int *hdr(int s) const { return reinterpret_cast<const int *>(blk(s)); }
    int n_old(int s) const { return hdr(s)[2] & 0xFFFF; }
    int n_glob(int s) const { return (int)((unsigned)hdr(s)[2] >> 16); }
    const unsigned *loc_items(int s) const { return reinterpret_cast<const unsigned *>(hdr(s) + 4); }
    const int *old_items(int s) const { return hdr(s) + 4 + ((hdr(s)[3] + 1) & ~1); }
    const int *glob_items(int s) const { return old_items(s) + 2 * n_old(s); }
    const double *vals(int s) const { return reinterpret_cast<const double *>(blk(s) + g.val_off()); }
    double old_value(int j) const { return snap ? (*snap)[j] : x[j]; }
    // what the gather wave does ahead of time: OLD operands and the rows' own old values
    Pre prefetch(int s) const
    {
        Pre p;
        const int nrows = hdr(s)[0];
        p.oldv.resize((size_t)n_old(s));
        for (int i = 0; i < n_old(s); ++i) p.oldv[i] = old_value(old_items(s)[2 * i + 1]);
        p.xo.resize((size_t)nrows);
        for (int r = 0; r < nrows; ++r) {
            int rid; std::memcpy(&rid, blk(s) + g.row_off() + 16 * r + 8, 4);
            p.xo[r] = old_value(rid & TP_MASK);
        }
        return p;
    }
    bool ready(int k) const
    {
        const int s = next[k];
        if (s >= P.tile_step[k + 1]) return false;
        for (int i = 0; i < n_glob(s); ++i)
            if (is_sent(xs[glob_items(s)[2 * i + 1]])) return false;
        return true;
    }
    void run(int k)
    {
        const int s = next[k];
        const Pre cur = pre[k].front();
        pre[k].erase(pre[k].begin());
        if (s + look < P.tile_step[k + 1]) pre[k].push_back(prefetch(s + look));
        const int nrows = hdr(s)[0], rbase = hdr(s)[1], nloc = hdr(s)[3];
        const int nent = P.steps[s].p1 - P.steps[s].p0;
        std::vector<double> prod(vals(s), vals(s) + nent);             // a_ij (the diagonal: +0)
        std::vector<int> touched((size_t)nent, 0);
        for (int i = 0; i < n_old(s); ++i) { const int e = old_items(s)[2 * i]; prod[e] = prod[e] * cur.oldv[i]; touched[e]++; }
        for (int i = 0; i < n_glob(s); ++i) {
            const int e = glob_items(s)[2 * i];
            const double xv = xs[glob_items(s)[2 * i + 1]];
            if (is_sent(xv)) hazards++;
            prod[e] = prod[e] * xv; touched[e]++;
        }
        for (int i = 0; i < nloc; ++i) {
            const unsigned it = loc_items(s)[i];
            const int e = (int)(it & 0xFFFFu);
            prod[e] = prod[e] * ring[k][(size_t)(it >> 16)]; touched[e]++;
        }
        for (int e = 0; e < nent; ++e) if (touched[e] > 1) bad++;
        std::vector<double> newv((size_t)nrows);
        for (int r = 0; r < nrows; ++r) {
            const unsigned char *rec = blk(s) + g.row_off() + 16 * r;
            double d; int rid, lohi;
            std::memcpy(&d, rec, 8); std::memcpy(&rid, rec + 8, 4); std::memcpy(&lohi, rec + 12, 4);
            const int row = rid & TP_MASK, lo = lohi & 0xFFFF, len = (int)((unsigned)lohi >> 16);
            double sum = (epi == 1) ? b[row] : 0.0;
            for (int e = lo; e < lo + len; ++e) {
                if (epi == 1) sum -= prod[e];
                else sum += prod[e];
            }
            const double xo = cur.xo[r];
            double v;
            if (epi == 0) v = (b[row] - sum) / d;
            else if (epi == 1) v = sum / d;
            else v = omega * ((b[row] - sum) / d) + (1.0 - omega) * xo;
            if (!(d != 0.0)) v = xo;
            newv[r] = v;
        }
        for (int r = 0; r < nrows; ++r) {
            int rid; std::memcpy(&rid, blk(s) + g.row_off() + 16 * r + 8, 4);
            const int row = rid & TP_MASK;
            ring[k][(size_t)((rbase + r) & (P.W - 1))] = newv[r];
            x[row] = newv[r];
            if (rid < 0) xs[row] = newv[r];
        }
        next[k] = s + 1;
    }
};
}  // namespace

extern "C" {

// x is swept in place.  policy 0: steps in (level, tile) order; 1: greedy, lowest tile first; 2: greedy, highest
// tile first.  stats: [0] tiles, [1] steps, [2] levels, [3] local early entries, [4] global early entries,
// [5] publishing rows, [6] hazards (sentinel consumed), [7] stuck (deadlock in the replay)
int tile_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                        int row_stop, int row_step, int G, int W, int cap, int max_rows, int epi, double omega,
                        int use_snapshot, int policy, int look, int partition, int64_t *stats)
{
    TilePlan P;
    if (build_tile_plan(n, Ap, Aj, row_start, row_stop, row_step, G, W, cap, max_rows, P, partition)) return 1;
    TileGeom geom{0, 0, 8};
    if (!tile_geometry(P, 8, geom)) return 2;
    std::vector<double> xv(x, x + n), snap;
    if (use_snapshot) snap = xv;
    Emul E(P, geom, Ax, Aj, b, xv, use_snapshot ? &snap : nullptr, epi, omega, std::max(1, look));
    if (E.bad) return 3;
    const int nsteps = (int)P.steps.size();
    int done = 0, stuck = 0;
    if (policy == 0) {
        std::vector<int> ord((size_t)nsteps);
        for (int s = 0; s < nsteps; ++s) ord[s] = s;
        std::vector<int> tile_of((size_t)nsteps);
        for (int k = 0; k < P.G; ++k) for (int s = P.tile_step[k]; s < P.tile_step[k + 1]; ++s) tile_of[s] = k;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return P.step_level[a] < P.step_level[c]; });
        for (int s : ord) {
            const int k = tile_of[s];
            if (E.next[k] != s || !E.ready(k)) { stuck = 1; break; }
            E.run(k);
            ++done;
        }
    } else {
        while (done < nsteps) {
            bool any = false;
            for (int kk = 0; kk < P.G; ++kk) {
                const int k = policy == 1 ? kk : P.G - 1 - kk;
                while (E.ready(k)) { E.run(k); ++done; any = true; }
            }
            if (!any) { stuck = 1; break; }
        }
    }
    std::memcpy(x, xv.data(), sizeof(double) * (size_t)n);
    if (stats) {
        stats[0] = P.G; stats[1] = nsteps; stats[2] = P.nlevels; stats[3] = P.n_local; stats[4] = P.n_global;
        stats[5] = P.n_publish; stats[6] = E.hazards + E.bad; stats[7] = stuck;
    }
    return 0;
}

}  // extern "C"
