// walk_emul.cpp -- CPU replay of the line-walk fast-order sweep (test infrastructure, not product code).
// Builds the layout with the product's own planner (pyamg_amd/csrc/pamg_walk_plan.h) and consumes it the way gs_walk_kernel
// does: `waves` waves take the lines statically (wave w: lines w, w + waves, ...) and walk them row after row; a row runs only
// when every early operand from OTHER lines (or earlier rows of its own) has been published -- else the wave "polls" (skipped
// this round); the predecessor's new value is forwarded inside the wave.  Per row: lane l adds its K products, XOR butterfly
// over the 64 lanes, v = (b - (s + afwd * v_prev)) * rdiag.  A round without progress is a deadlock (error 20); an old operand
// that has already been overwritten is error 13 (unless a snapshot is used).
#include "../pyamg_amd/csrc/pamg_walk_plan.h"
#include <cmath>

using namespace pamg;

extern "C" int walk_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                                   int row_stop, int row_step, int sor, double omega, int snapshot, int waves, long long *stats)
{
    WalkPlan P;
    if (build_walk_plan(n, Ap, Aj, reinterpret_cast<const unsigned char *>(Ax), 8, row_start, row_stop, row_step, P)) return 2;
    const int K = P.K;
    stats[0] = K; stats[1] = P.nrows; stats[2] = P.nlines; stats[3] = P.nlevels; stats[4] = P.n_early; stats[5] = P.n_forward; stats[6] = P.max_level_lines;
    const double *vals = reinterpret_cast<const double *>(P.vals.data());
    const double *rd = reinterpret_cast<const double *>(P.rdiag.data());
    const double *af = reinterpret_cast<const double *>(P.afwd.data());
    std::vector<double> xs((size_t)n), xold;
    std::vector<char> pub((size_t)n, 0), written((size_t)n, 0);
    if (snapshot) xold.assign(x, x + n);
    const double *xsrc = snapshot ? xold.data() : x;
    if (waves < 1) waves = 1;
    struct WS { int64_t line, q; double vprev; };
    std::vector<WS> ws((size_t)waves);
    for (int w = 0; w < waves; ++w) { ws[(size_t)w].line = w; ws[(size_t)w].q = w < P.nlines ? P.line_row[(size_t)w] : 0; ws[(size_t)w].vprev = 0.0; }
    int64_t left = P.nrows;
    while (left > 0) {
        bool progress = false;
        for (int w = waves - 1; w >= 0; --w) {
            WS &S = ws[(size_t)w];
            while (S.line < P.nlines) {                       // a wave runs on as long as its next row is ready
                const int64_t q = S.q;
                bool ready = true;
                for (int e = 0; e < K * 64 && ready; ++e) {
                    const int c = P.cols[(size_t)(q * K * 64 + e)];
                    if (!(c & WALK_NONE) && (c & WALK_EARLY) && !pub[(size_t)(c & WALK_MASK)]) ready = false;
                }
                if (!ready) break;
                double ls[64];
                for (int l = 0; l < 64; ++l) {
                    double s = 0.0;
                    for (int k = 0; k < K; ++k) {
                        const size_t e = (size_t)((q * K + k) * 64 + l);
                        const int c = P.cols[e];
                        if (c & WALK_NONE) continue;
                        const int col = c & WALK_MASK;
                        double xv;
                        if (c & WALK_EARLY) xv = xs[(size_t)col];
                        else {
                            if (!snapshot && written[(size_t)col]) return 13;
                            xv = xsrc[col];
                        }
                        s = s + vals[e] * xv;
                    }
                    ls[l] = s;
                }
                for (int d = 1; d < 64; d *= 2) { double t2[64]; for (int l = 0; l < 64; ++l) t2[l] = ls[l] + ls[l ^ d]; for (int l = 0; l < 64; ++l) ls[l] = t2[l]; }
                const int rid = P.rid[(size_t)q];
                const int row = rid & WALK_MASK;
                const bool upd = !(rid & WALK_NODIAG);
                if (!snapshot && written[(size_t)row]) return 14;
                double v = (b[row] - (ls[0] + af[(size_t)q] * S.vprev)) * rd[(size_t)q];
                if (sor) v = omega * v + (1.0 - omega) * xsrc[row];
                if (!upd) v = xsrc[row];
                if (pub[(size_t)row]) return 15;
                xs[(size_t)row] = v; pub[(size_t)row] = 1;
                if (upd) { x[row] = v; written[(size_t)row] = 1; }
                S.vprev = v;
                --left; progress = true;
                if (++S.q >= P.line_row[(size_t)S.line + 1]) {
                    S.line += waves; S.vprev = 0.0;
                    if (S.line < P.nlines) S.q = P.line_row[(size_t)S.line];
                }
            }
        }
        if (!progress) return 20;
    }
    return 0;
}
