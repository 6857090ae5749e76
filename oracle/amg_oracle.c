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
