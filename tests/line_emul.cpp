// line_emul.cpp -- CPU replay of the line-scan fast-order sweep (test infrastructure, not product code).
// Builds the layout with the product's own planner (pyamg_amd/csrc/pamg_line_plan.h) and consumes it the way gs_line_kernel
// does: `waves` waves take the lines statically (wave w: lines w, w + waves, ...), every wave walks the chunks of its line in
// order; a chunk step = for every lane B = (b - sum of the other entries) * rdiag, A = acoef, an inclusive scan of the pairs
// over the lanes (Hillis-Steele: distance 1, 2, 4, ... -- the association of the device's shuffles), x = B + A * carry, publish.
// A chunk runs only when every early operand has been published (else the wave "polls": skipped this round); a round without
// progress is a deadlock (error 20).  Old operands must still be old when read (error 13, unless a snapshot is used).
#include "../pyamg_amd/csrc/pamg_line_plan.h"
#include <cmath>
#include <cstdio>

using namespace pamg;

extern "C" int line_emul_sweep_f64(int n, const int *Ap, const int *Aj, const double *Ax, double *x, const double *b, int row_start,
                                   int row_stop, int row_step, int sor, double omega, int snapshot, int waves, long long *stats)
{
    LinePlan P;
    if (build_line_plan(n, Ap, Aj, reinterpret_cast<const unsigned char *>(Ax), 8, row_start, row_stop, row_step, P)) return 2;
    const int K = P.K;
    stats[0] = K; stats[1] = P.nchunks; stats[2] = P.nlines; stats[3] = P.nlevels; stats[4] = P.n_early; stats[5] = P.n_old; stats[6] = P.max_level_lines;
    const double *vals = reinterpret_cast<const double *>(P.vals.data());
    const double *rd = reinterpret_cast<const double *>(P.rdiag.data());
    const double *ac = reinterpret_cast<const double *>(P.acoef.data());
    std::vector<double> xs((size_t)n), xold;
    std::vector<char> pub((size_t)n, 0), written((size_t)n, 0);
    if (snapshot) xold.assign(x, x + n);
    const double *xsrc = snapshot ? xold.data() : x;
    if (waves < 1) waves = 1;
    struct WaveState { int64_t line, chunk; double carry; };
    std::vector<WaveState> ws((size_t)waves);
    for (int w = 0; w < waves; ++w) { ws[(size_t)w].line = w; ws[(size_t)w].chunk = w < P.nlines ? P.line_chunk[(size_t)w] : 0; ws[(size_t)w].carry = 0.0; }
    int64_t left = P.nchunks, rows_done = 0;
    while (left > 0) {
        bool progress = false;
        for (int w = waves - 1; w >= 0; --w) {               // adversarial order: later lines get the first chance to run ahead
            WaveState &S = ws[(size_t)w];
            if (S.line >= P.nlines) continue;
            const int64_t g = S.chunk;
            const int cnt = P.cnt[(size_t)g], r0 = P.row0[(size_t)g];
            bool ready = true;
            for (int lane = 0; lane < cnt && ready; ++lane)
                for (int k = 0; k < K; ++k) {
                    const int c = P.cols[(size_t)((g * K + k) * 64 + lane)];
                    if (!(c & LINE_NONE) && (c & LINE_EARLY) && !pub[(size_t)(c & LINE_MASK)]) { ready = false; break; }
                }
            if (!ready) continue;
            double A[64], B[64];
            for (int lane = 0; lane < 64; ++lane) { A[lane] = 0.0; B[lane] = 0.0; }
            for (int lane = 0; lane < cnt; ++lane) {
                const int row = r0 + lane * P.step;
                double s = 0.0;
                for (int k = 0; k < K; ++k) {
                    const size_t e = (size_t)((g * K + k) * 64 + lane);
                    const int c = P.cols[e];
                    if (c & LINE_NONE) continue;
                    const int col = c & LINE_MASK;
                    double xv;
                    if (c & LINE_EARLY) xv = xs[(size_t)col];
                    else {
                        if (!snapshot && written[(size_t)col]) return 13;
                        xv = xsrc[col];
                    }
                    s = s + vals[e] * xv;
                }
                const size_t rs = (size_t)(g * 64 + lane);
                if (lane > 0 && !snapshot && written[(size_t)row]) return 14;
                double Bv = (b[row] - s) * rd[rs], Av = ac[rs];
                if (sor) { Bv = omega * Bv + (1.0 - omega) * xsrc[row]; Av = omega * Av; }
                if (P.nodiag[rs]) { Bv = xsrc[row]; Av = 0.0; }
                A[lane] = Av; B[lane] = Bv;
            }
            for (int d = 1; d < 64; d *= 2) {                 // inclusive scan of (A, B) under (A2, B2) o (A1, B1) = (A2 A1, B2 + A2 B1)
                double A2[64], B2[64];
                for (int lane = 0; lane < 64; ++lane) {
                    if (lane >= d) { A2[lane] = A[lane] * A[lane - d]; B2[lane] = B[lane] + A[lane] * B[lane - d]; }
                    else { A2[lane] = A[lane]; B2[lane] = B[lane]; }
                }
                for (int lane = 0; lane < 64; ++lane) { A[lane] = A2[lane]; B[lane] = B2[lane]; }
            }
            double last = 0.0;
            for (int lane = 0; lane < cnt; ++lane) {
                const int row = r0 + lane * P.step;
                const double v = B[lane] + A[lane] * S.carry;
                if (pub[(size_t)row]) return 15;
                xs[(size_t)row] = v; pub[(size_t)row] = 1;
                if (!P.nodiag[(size_t)(g * 64 + lane)]) { x[row] = v; written[(size_t)row] = 1; }
                last = v;
                ++rows_done;
            }
            S.carry = last;
            --left; progress = true;
            if (++S.chunk >= P.line_chunk[(size_t)S.line + 1]) {
                S.line += waves; S.carry = 0.0;
                if (S.line < P.nlines) S.chunk = P.line_chunk[(size_t)S.line];
            }
        }
        if (!progress) return 20;
    }
    return 0;
}
