// pamg_schwarz_plan.h -- host side of the Schwarz sweep schedules (plain C++, no HIP: the CPU replay tests/schwarz_emul.cpp includes it too).
//
// Reference: amg_core::overlapping_schwarz_csr, relaxation.h:1420-1492 -- subdomains one after another; a subdomain forms the residual of its
// rows from the current x, multiplies by its inverted diagonal block and adds the result to x.
//
//  schwarz_levels    the dependency levels of the visited subdomains (d waits for an earlier d' that wrote what d reads or writes, or read what d
//                    writes) and the level-sorted order the device walks.
//  schwarz_versions  the version table of the persistent sweep: a row is updated once by every visited subdomain that holds it; its v-th update of
//                    the sweep writes slot vbase[row] + v - 1 of a hand-off buffer (single assignment), and every read -- every stored entry of every
//                    member row of every subdomain -- is told which version of the column's row the sequential sweep would find there
//                    (one byte: 0 = x as it was before the sweep).  Nothing is overwritten, so a reader can never see a value that is too new,
//                    and x takes the last versions when the sweep is over.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace pamg {

struct SchwarzLevels {
    int m = 0;                        // subdomains visited
    int nlevels = 0;
    int max_width = 0;                // subdomains of the widest level
    std::vector<int> level_ptr;       // [nlevels + 1] offsets into order
    std::vector<int> order;           // [m] subdomains, level after level, sweep order inside a level
};

// 0 = ok, 1 = bad sweep bounds
inline int schwarz_levels(int n, const int *Ap, const int *Aj, int nsub, const int *Sp, const int *Sj, int start, int stop, int step, SchwarzLevels &g)
{
    g = SchwarzLevels();
    g.level_ptr.assign(1, 0);
    if (step == 0) return 1;
    const long span = (long)stop - start;
    if (span % step != 0 || span / step < 0) return 1;
    const int m = (int)(span / step);
    if (m == 0) return 0;
    if (start < 0 || start >= nsub || start + (long)(m - 1) * step < 0 || start + (long)(m - 1) * step >= nsub) return 1;
    std::vector<int> lastW((size_t)n, -1), lastR((size_t)n, -1), lvl((size_t)m, 0);
    int maxl = 0;
    for (int t = 0; t < m; ++t) {
        const int d = start + t * step;
        int L = 0;
        for (int q = Sp[d]; q < Sp[d + 1]; ++q) {
            const int row = Sj[q];
            L = std::max(L, std::max(lastW[(size_t)row], lastR[(size_t)row]) + 1);                 // write after write / write after read
            for (int p = Ap[row]; p < Ap[row + 1]; ++p) L = std::max(L, lastW[(size_t)Aj[p]] + 1);   // read after write
        }
        lvl[(size_t)t] = L;
        maxl = std::max(maxl, L);
        for (int q = Sp[d]; q < Sp[d + 1]; ++q) {
            const int row = Sj[q];
            lastW[(size_t)row] = std::max(lastW[(size_t)row], L);
            for (int p = Ap[row]; p < Ap[row + 1]; ++p) { const int j = Aj[p]; lastR[(size_t)j] = std::max(lastR[(size_t)j], L); }
        }
    }
    g.m = m;
    g.nlevels = maxl + 1;
    g.level_ptr.assign((size_t)g.nlevels + 1, 0);
    for (int t = 0; t < m; ++t) g.level_ptr[(size_t)lvl[(size_t)t] + 1]++;
    for (int l = 0; l < g.nlevels; ++l) g.level_ptr[(size_t)l + 1] += g.level_ptr[(size_t)l];
    g.order.assign((size_t)m, 0);
    std::vector<int> cur(g.level_ptr.begin(), g.level_ptr.end() - 1);
    for (int t = 0; t < m; ++t) g.order[(size_t)cur[(size_t)lvl[(size_t)t]]++] = start + t * step;
    for (int l = 0; l < g.nlevels; ++l) g.max_width = std::max(g.max_width, g.level_ptr[(size_t)l + 1] - g.level_ptr[(size_t)l]);
    return 0;
}

struct SchwarzVersions {
    bool ok = false;                  // false: the schedule runs as one launch per level (why: `declined`)
    int declined = 0;                 // 1 = a row is updated more than 255 times, 2 = tables beyond 31-bit offsets / a gigabyte, 3 = a subdomain lists a row twice
    int64_t nslots = 0;               // E: updates of the sweep = (position, member row) entries
    int64_t nreads = 0;               // R: stored entries of all member rows
    std::vector<int> ebase;           // [m + 1] first entry of every position of the level-sorted order
    std::vector<int> wslot;           // [E] slot this update writes
    std::vector<int> prev;            // [E] the row's value before this update: slot >= 0, or ~row = x itself
    std::vector<int> roff;            // [E] first byte of the row's read versions in rver
    std::vector<unsigned char> rver;  // [R] per stored entry of the row: 0 = x itself, v = version v of the column's row
    std::vector<int> vbase;           // [n + 1] first slot of every row
    std::vector<int> last;            // [n] slot of the row's last version, -1 = not updated by this sweep
};

inline void schwarz_versions(int n, const int *Ap, const int *Aj, int nsub, const int *Sp, const int *Sj, int start, int step, const SchwarzLevels &g,
                             SchwarzVersions &V, int64_t max_reads = (int64_t)1 << 30)
{
    V = SchwarzVersions();
    const int m = g.m;
    std::vector<int> nupd((size_t)n, 0);
    int64_t E = 0, R = 0;
    for (int t = 0; t < m; ++t) {
        const int d = start + t * step;
        for (int q = Sp[d]; q < Sp[d + 1]; ++q) {
            const int row = Sj[q];
            if (++nupd[(size_t)row] > 255) { V.declined = 1; return; }
            R += Ap[row + 1] - Ap[row];
        }
        E += Sp[d + 1] - Sp[d];
    }
    if (E >= ((int64_t)1 << 31) - 1 || R >= max_reads) { V.declined = 2; return; }
    V.vbase.assign((size_t)n + 1, 0);
    V.last.assign((size_t)n, -1);
    for (int i = 0; i < n; ++i) V.vbase[(size_t)i + 1] = V.vbase[(size_t)i] + nupd[(size_t)i];
    for (int i = 0; i < n; ++i) if (nupd[(size_t)i]) V.last[(size_t)i] = V.vbase[(size_t)i + 1] - 1;
    // entries are laid out by POSITION in the level-sorted order (what a wave walks); the versions are counted in SWEEP order
    std::vector<int> pos_of((size_t)nsub, -1);
    for (int q = 0; q < m; ++q) pos_of[(size_t)g.order[(size_t)q]] = q;
    V.ebase.assign((size_t)m + 1, 0);
    for (int q = 0; q < m; ++q) V.ebase[(size_t)q + 1] = V.ebase[(size_t)q] + (Sp[g.order[(size_t)q] + 1] - Sp[g.order[(size_t)q]]);
    V.wslot.assign((size_t)E, 0); V.prev.assign((size_t)E, 0); V.roff.assign((size_t)E, 0);
    {
        int64_t ro = 0;                                           // read offsets: rows of a position one after another
        for (int q = 0; q < m; ++q) {
            const int d = g.order[(size_t)q];
            for (int k = 0; k < Sp[d + 1] - Sp[d]; ++k) {
                const int row = Sj[Sp[d] + k];
                V.roff[(size_t)V.ebase[(size_t)q] + k] = (int)ro;
                ro += Ap[row + 1] - Ap[row];
            }
        }
    }
    V.rver.assign((size_t)std::max<int64_t>(R, 1), 0);
    std::vector<int> cnt((size_t)n, 0), seen((size_t)n, -1);
    for (int t = 0; t < m; ++t) {
        const int d = start + t * step, q = pos_of[(size_t)d];
        const int s0 = Sp[d], size = Sp[d + 1] - s0;
        for (int k = 0; k < size; ++k) {                          // every residual of the subdomain sees the state BEFORE its own updates
            const int row = Sj[s0 + k];
            if (seen[(size_t)row] == t) { V.declined = 3; return; }   // a row listed twice: the reference updates it twice in a row; not this form
            seen[(size_t)row] = t;
            unsigned char *rv = V.rver.data() + V.roff[(size_t)V.ebase[(size_t)q] + k];
            for (int p = Ap[row]; p < Ap[row + 1]; ++p) rv[p - Ap[row]] = (unsigned char)cnt[(size_t)Aj[p]];
        }
        for (int k = 0; k < size; ++k) {
            const int row = Sj[s0 + k];
            const int v = cnt[(size_t)row]++;
            V.wslot[(size_t)V.ebase[(size_t)q] + k] = V.vbase[(size_t)row] + v;
            V.prev[(size_t)V.ebase[(size_t)q] + k] = v == 0 ? ~row : V.vbase[(size_t)row] + v - 1;
        }
    }
    V.nslots = E; V.nreads = R;
    V.ok = true;
}

}  // namespace pamg
