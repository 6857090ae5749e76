// CPU replay of the persistent Schwarz sweep (pyamg_amd/csrc/pamg_schwarz_plan.h + schwarz_versioned_kernel in pamg_schwarz.hip): the version
// table is consumed the way the kernel consumes it -- G "waves" take the positions w, w + G, ... of the level-sorted order; a wave runs its
// subdomain only when every slot it reads has been written (on the device: it polls), residuals in stored order, the dense product row by row,
// one new slot per member row; x takes the last versions at the end.  The waves are visited in an ADVERSARIAL order (last wave first) and the
// replay asserts what the device relies on: some wave can always move (no deadlock for any G), every slot is written exactly once, a read never
// finds a slot of its own subdomain, version 0 is x as it was before the sweep.  Test infrastructure only (tests/test_schwarz_plan.py).
#include <cstdint>
#include <vector>

#include "../pyamg_amd/csrc/pamg_schwarz_plan.h"

using namespace pamg;

// stats[8]: 0 visited subdomains, 1 dependency levels, 2 widest level, 3 slots, 4 reads, 5 reads of version 0, 6 sweeps over the waves, 7 declined code
extern "C" int schwarz_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, int nsub, const int *Sp, const int *Sj, const int *Tp,
                                      const double *Tx, double *x, const double *b, int start, int stop, int step, int waves, long long max_reads,
                                      long long *stats)
{
    SchwarzLevels g;
    if (schwarz_levels(n, Ap, Aj, nsub, Sp, Sj, start, stop, step, g)) return 1;
    stats[0] = g.m; stats[1] = g.nlevels; stats[2] = g.max_width;
    for (int k = 3; k < 8; ++k) stats[k] = 0;
    if (g.m == 0) return 0;
    // the level-sorted order is a permutation of the visited subdomains, sweep order inside a level
    {
        std::vector<char> hit((size_t)nsub, 0);
        for (int q = 0; q < g.m; ++q) { const int d = g.order[(size_t)q]; if (d < 0 || d >= nsub || hit[(size_t)d]) return 20; hit[(size_t)d] = 1; }
        for (int l = 0; l < g.nlevels; ++l)
            for (int q = g.level_ptr[(size_t)l] + 1; q < g.level_ptr[(size_t)l + 1]; ++q)
                if ((g.order[(size_t)q] - g.order[(size_t)q - 1]) * (step > 0 ? 1 : -1) <= 0) return 21;
    }
    SchwarzVersions V;
    schwarz_versions(n, Ap, Aj, nsub, Sp, Sj, start, step, g, V, max_reads > 0 ? (int64_t)max_reads : ((int64_t)1 << 30));
    stats[7] = V.declined;
    if (!V.ok) return 2;
    stats[3] = V.nslots; stats[4] = V.nreads;
    std::vector<double> xs((size_t)V.nslots, 0.0);
    std::vector<char> written((size_t)V.nslots, 0);
    const int G = std::max(1, std::min(waves > 0 ? waves : 64, g.m));
    std::vector<int> at((size_t)G);
    for (int w = 0; w < G; ++w) at[(size_t)w] = w;
    std::vector<double> r;
    int64_t done = 0;
    while (done < g.m) {
        bool moved = false;
        ++stats[6];
        for (int w = G - 1; w >= 0; --w) {                                         // the wave with the LATEST work first
            const int q = at[(size_t)w];
            if (q >= g.m) continue;
            const int d = g.order[(size_t)q], e0 = V.ebase[(size_t)q];
            const int s0 = Sp[d], size = Sp[d + 1] - s0;
            bool ready = true;
            for (int k = 0; k < size && ready; ++k) {
                const int row = Sj[s0 + k];
                const unsigned char *rv = V.rver.data() + V.roff[(size_t)e0 + k];
                for (int p = Ap[row]; p < Ap[row + 1]; ++p) {
                    const int ver = rv[p - Ap[row]];
                    if (!ver) continue;
                    const int sl = V.vbase[(size_t)Aj[p]] + ver - 1;
                    if (sl < V.vbase[(size_t)Aj[p]] || sl >= V.vbase[(size_t)Aj[p] + 1]) return 10;   // a version the row never gets
                    if (!written[(size_t)sl]) { ready = false; break; }
                }
                const int pc = V.prev[(size_t)e0 + k];
                if (pc >= 0 && !written[(size_t)pc]) ready = false;
            }
            if (!ready) continue;                                                  // on the device: still polling
            r.assign((size_t)size, 0.0);
            for (int k = 0; k < size; ++k) {
                const int row = Sj[s0 + k];
                const unsigned char *rv = V.rver.data() + V.roff[(size_t)e0 + k];
                double rsum = 0.0;
                for (int p = Ap[row]; p < Ap[row + 1]; ++p) {
                    const int ver = rv[p - Ap[row]];
                    if (!ver) ++stats[5];
                    const double xv = ver ? xs[(size_t)(V.vbase[(size_t)Aj[p]] + ver - 1)] : x[Aj[p]];
                    rsum -= Ax[p] * xv;
                }
                rsum += b[row];
                r[(size_t)k] = rsum;
            }
            const double *Tinv = Tx + Tp[d];
            for (int i = 0; i < size; ++i) {
                double s = 0.0;
                for (int k = 0; k < size; ++k) s += Tinv[(size_t)i * size + k] * r[(size_t)k];
                const int row = Sj[s0 + i];
                const int pc = V.prev[(size_t)e0 + i], ws = V.wslot[(size_t)e0 + i];
                if (pc >= 0 ? pc != ws - 1 : (~pc != row || ws != V.vbase[(size_t)row])) return 11;   // versions of a row are consecutive slots
                if (ws < V.vbase[(size_t)row] || ws >= V.vbase[(size_t)row + 1]) return 12;
                if (written[(size_t)ws]) return 13;                                // single assignment
                const double xo = pc < 0 ? x[row] : xs[(size_t)pc];
                xs[(size_t)ws] = xo + s;
                written[(size_t)ws] = 1;
            }
            at[(size_t)w] = q + G;
            ++done;
            moved = true;
        }
        if (!moved) return 14;                                                     // every wave polls for ever: a deadlock on the device
    }
    for (int64_t e = 0; e < V.nslots; ++e) if (!written[(size_t)e]) return 15;
    for (int i = 0; i < n; ++i) {
        const int sl = V.last[(size_t)i];
        if (sl >= 0) { if (sl != V.vbase[(size_t)i + 1] - 1) return 16; x[i] = xs[(size_t)sl]; }
        else if (V.vbase[(size_t)i + 1] != V.vbase[(size_t)i]) return 17;
    }
    return 0;
}
