/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 * CPU restatement, in plain C, of the reference's solve-path kernels.  It is the
 * checker for the HIP path in tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; the product (pyamg_amd/) never loads it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   (1) the reference's own known answers (pyamg/relaxation/tests/test_relaxation.py
 *       :148-197,299-362,808-835; doctest values relaxation.py:136-138,297-299,379-381,
 *       458-460,539-541), committed as tests/golden/known_answers.json, and
 *   (2) outputs of the real reference (oracle/_ref, built from /root/reference by
 *       oracle/build_ref.py) on seeded inputs, committed as tests/golden/ (npz files) by
 *       tests/golden/make_golden.py -- bit-for-bit.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <float.h>
#include <math.h>

#define REAL double
#define FN(x) orc_##x##_f64
#define RSQRT sqrt
#define RFABS fabs
#define REPS DBL_EPSILON
#include "amg_oracle_impl.h"
#undef REAL
#undef FN
#undef RSQRT
#undef RFABS
#undef REPS

#define REAL float
#define FN(x) orc_##x##_f32
#define RSQRT sqrtf
#define RFABS fabsf
#define REPS FLT_EPSILON
#include "amg_oracle_impl.h"
#undef REAL
#undef FN
#undef RSQRT
#undef RFABS
#undef REPS


/* Greedy aggregation of a strength graph.  Follows amg_core standard_aggregation, smoothed_aggregation.h:137-268:
 * pass 1 -- in index order a node none of whose neighbours (nor itself) is taken founds an aggregate of itself and
 * all its neighbours; nodes without neighbours are set aside; pass 2 -- a node still free joins the aggregate of the
 * first neighbour (storage order) that pass 1 placed; pass 3 -- numbers become 0-based, nodes set aside get -1, a node
 * still free founds an aggregate with whichever neighbours read 0 at that moment (a 0 that may by then be the NUMBER of
 * aggregate 0: the reference's quirk, kept).  x: aggregate of every node, y: root nodes; returns their count. */
int orc_standard_aggregation(int n, const int *Ap, const int *Aj, int *x, int *y)
{
    int next = 1;
    for (int i = 0; i < n; ++i) x[i] = 0;
    for (int i = 0; i < n; ++i) {
        if (x[i]) continue;
        int taken = 0, nbrs = 0;
        for (int p = Ap[i]; p < Ap[i + 1] && !taken; ++p)
            if (Aj[p] != i) { nbrs = 1; taken = x[Aj[p]] != 0; }
        if (!nbrs) { x[i] = -n; continue; }
        if (taken) continue;
        x[i] = next;
        y[next - 1] = i;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) x[Aj[p]] = next;
        ++next;
    }
    for (int i = 0; i < n; ++i) {
        if (x[i]) continue;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p)
            if (x[Aj[p]] > 0) { x[i] = -x[Aj[p]]; break; }
    }
    --next;
    for (int i = 0; i < n; ++i) {
        const int v = x[i];
        if (v > 0) x[i] = v - 1;
        else if (v == -n) x[i] = -1;
        else if (v < 0) x[i] = -v - 1;
        else {
            x[i] = next;
            y[next] = i;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p)
                if (x[Aj[p]] == 0) x[Aj[p]] = next;
            ++next;
        }
    }
    return next;
}
