/*
 * pyamg_amd.h -- C ABI of the MI355X-native AMG solve-phase engine (libpyamg_amd.so).
 *
 * Drop-in boundary for the solve phase behind pyamg.multilevel.MultilevelSolver.solve()
 * (reference: pyamg/multilevel.py:398-662) and for the amg_core relaxation kernels
 * (reference: pyamg/amg_core/relaxation.h) plus SciPy's sparsetools SpMV that the
 * reference borrows (call sites multilevel.py:545,567,612,614,660).
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Two layers:
 *
 *   Layer 1  "amg_core-compatible": one symbol per (reference function x dtype) with the
 *            reference's exact argument order (pointer followed by its length, scalars by
 *            value -- the convention of amg_core/bindthem.py:77-82).  Pointers are HOST
 *            buffers, exactly what the reference's pybind11 layer hands to amg_core
 *            (relaxation_bind.cpp:11-44); the call stages them through HBM, runs the HIP
 *            kernels and writes x back in place.  The reference returns void; we return a
 *            status.  This is what a maintainer binds to replace `from pyamg import
 *            amg_core` call by call.
 *
 *   Layer 2  "resident engine": opaque handles for operators and for a whole hierarchy that
 *            is shipped to HBM once; vectors are DEVICE pointers; everything is
 *            stream-ordered.  This is what sits behind MultilevelSolver.solve() /
 *            aspreconditioner().
 *
 * Status codes: 0 = ok, > 0 = hipError_t from the runtime, < 0 = PAMG_E_* below.
 * Index type is int32 only and values are f64 or f32 (reference: instantiate.yml:2-6;
 * complex is not on the device path).
 *
 * Arithmetic contract: every kernel accumulates each row's products sequentially in
 * storage order with separate multiply and add (no FMA contraction), i.e. in exactly the
 * order of the reference's scalar loops, so results are bit-identical to the reference
 * on the same inputs (dense coarse solve and norms excepted: last-bit differences).
 */
#ifndef PYAMG_AMD_H
#define PYAMG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAMG_OK              0
#define PAMG_E_ARG          -1   /* bad argument (size/shape/enum)            */
#define PAMG_E_UNSUPPORTED  -2   /* valid in the reference, not on the device */
#define PAMG_E_NODEVICE     -3   /* no HIP device visible                     */
#define PAMG_E_STATE        -4   /* call sequence violated                    */
#define PAMG_E_ALLOC        -5   /* host allocation failed                    */
#define PAMG_E_TIMEOUT      -6   /* a persistent sweep hit its spin bound (a workgroup it waited for never
                                    ran): the vectors it touched are invalid     */
/* -7 = PAMG_E_COMM, declared with the sharded cycle below */

#define PAMG_F64 0
#define PAMG_F32 1

typedef void *pamg_stream_t;               /* hipStream_t, NULL = default stream */
typedef void *pamg_event_t;                /* hipEvent_t                         */
typedef struct pamg_matrix_s *pamg_matrix_t;
typedef struct pamg_solver_s *pamg_solver_t;

/* ------------------------------------------------------------------ runtime plumbing */
const char *pamg_version(void);
/* Layer 1 keeps the last few operators it was handed resident (keyed by the identity of the three host arrays AND a
 * hash of their full contents), so a smoother applied call after call does not re-upload the matrix or redo the
 * dependency analysis of the order-exact sweeps.  PAMG_L1_CACHE=0 in the environment disables it. */
int pamg_l1_cache_clear(void);
int pamg_l1_cache_size(int *entries);
const char *pamg_status_string(int status);
int pamg_device_count(int *count);
/* Measured bandwidth ceiling of the current device: kind 0 = copy c = a (16 bytes per element moved), 1 = triad
 * c = a + s b (24 bytes), n doubles per vector, 16-byte accesses, `reps` timed launches; *gbps = bytes moved / time.
 * bench.py reports it beside the datasheet peak ("also measure an on-device copy/triad ceiling"). */
int pamg_bandwidth_probe(int kind, int64_t n, int reps, double *gbps);
int pamg_set_device(int device);
int pamg_get_device(int *device);
int pamg_device_name(int device, char *buf, int buflen);
int pamg_malloc(void **dptr, size_t bytes);
int pamg_free(void *dptr);
int pamg_memcpy_h2d(void *dst, const void *src, size_t bytes, pamg_stream_t s);
int pamg_memcpy_d2h(void *dst, const void *src, size_t bytes, pamg_stream_t s);
int pamg_memcpy_d2d(void *dst, const void *src, size_t bytes, pamg_stream_t s);
int pamg_memset(void *dst, int byte, size_t bytes, pamg_stream_t s);
int pamg_stream_create(pamg_stream_t *s);
int pamg_stream_destroy(pamg_stream_t s);
int pamg_stream_synchronize(pamg_stream_t s);
int pamg_device_synchronize(void);
int pamg_event_create(pamg_event_t *e);
int pamg_event_destroy(pamg_event_t e);
int pamg_event_record(pamg_event_t e, pamg_stream_t s);
int pamg_event_synchronize(pamg_event_t e);
int pamg_event_elapsed_ms(pamg_event_t start, pamg_event_t stop, float *ms);

/* ------------------------------------------------- Layer 1: amg_core-compatible (HOST) */
/* SciPy sparsetools csr_matvec / bsr_matvec:  Yx += A * Xx  (argument order of
 * scipy/sparse/sparsetools/csr.h csr_matvec, bsr.h bsr_matvec). */
int pamg_csr_matvec_f64(int n_row, int n_col, const int32_t *Ap, const int32_t *Aj,
                        const double *Ax, const double *Xx, double *Yx);
int pamg_csr_matvec_f32(int n_row, int n_col, const int32_t *Ap, const int32_t *Aj,
                        const float *Ax, const float *Xx, float *Yx);
int pamg_bsr_matvec_f64(int n_brow, int n_bcol, int R, int C, const int32_t *Ap,
                        const int32_t *Aj, const double *Ax, const double *Xx, double *Yx);
int pamg_bsr_matvec_f32(int n_brow, int n_bcol, int R, int C, const int32_t *Ap,
                        const int32_t *Aj, const float *Ax, const float *Xx, float *Yx);

/* amg_core::gauss_seidel, relaxation.h:48-56 */
int pamg_gauss_seidel_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                          const double *Ax, int Ax_size, double *x, int x_size,
                          const double *b, int b_size,
                          int32_t row_start, int32_t row_stop, int32_t row_step);
int pamg_gauss_seidel_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                          const float *Ax, int Ax_size, float *x, int x_size,
                          const float *b, int b_size,
                          int32_t row_start, int32_t row_stop, int32_t row_step);
/* amg_core::sor_gauss_seidel, relaxation.h:116-125 */
int pamg_sor_gauss_seidel_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                              const double *Ax, int Ax_size, double *x, int x_size,
                              const double *b, int b_size, int32_t row_start,
                              int32_t row_stop, int32_t row_step, double omega);
int pamg_sor_gauss_seidel_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                              const float *Ax, int Ax_size, float *x, int x_size,
                              const float *b, int b_size, int32_t row_start,
                              int32_t row_stop, int32_t row_step, float omega);
/* amg_core::bsr_gauss_seidel, relaxation.h:185-195 */
int pamg_bsr_gauss_seidel_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                              const double *Ax, int Ax_size, double *x, int x_size,
                              const double *b, int b_size, int32_t row_start,
                              int32_t row_stop, int32_t row_step, int32_t blocksize);
int pamg_bsr_gauss_seidel_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                              const float *Ax, int Ax_size, float *x, int x_size,
                              const float *b, int b_size, int32_t row_start,
                              int32_t row_stop, int32_t row_step, int32_t blocksize);
/* amg_core::jacobi, relaxation.h:309-319 (omega is a 1-element array, as in the reference) */
int pamg_jacobi_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                    const double *Ax, int Ax_size, double *x, int x_size,
                    const double *b, int b_size, double *temp, int temp_size,
                    int32_t row_start, int32_t row_stop, int32_t row_step,
                    const double *omega, int omega_size);
int pamg_jacobi_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                    const float *Ax, int Ax_size, float *x, int x_size,
                    const float *b, int b_size, float *temp, int temp_size,
                    int32_t row_start, int32_t row_stop, int32_t row_step,
                    const float *omega, int omega_size);
/* amg_core::jacobi_indexed, relaxation.h:382-390 (the kernel of cf_jacobi / fc_jacobi) */
int pamg_jacobi_indexed_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                            const double *Ax, int Ax_size, double *x, int x_size,
                            const double *b, int b_size, const int32_t *indices, int indices_size,
                            const double *omega, int omega_size);
int pamg_jacobi_indexed_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                            const float *Ax, int Ax_size, float *x, int x_size,
                            const float *b, int b_size, const int32_t *indices, int indices_size,
                            const float *omega, int omega_size);
/* amg_core::overlapping_schwarz_csr, relaxation.h:1420-1434 (Tx/Tp: the inverted diagonal blocks of the subdomains,
 * row-major, and their offsets; Sj/Sp: the rows of every subdomain, sorted and unique) */
int pamg_overlapping_schwarz_csr_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                                     const double *Ax, int Ax_size, double *x, int x_size,
                                     const double *b, int b_size, const double *Tx, int Tx_size,
                                     const int32_t *Tp, int Tp_size, const int32_t *Sj, int Sj_size,
                                     const int32_t *Sp, int Sp_size, int32_t nsdomains, int32_t nrows,
                                     int32_t row_start, int32_t row_stop, int32_t row_step);
int pamg_overlapping_schwarz_csr_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                                     const float *Ax, int Ax_size, float *x, int x_size,
                                     const float *b, int b_size, const float *Tx, int Tx_size,
                                     const int32_t *Tp, int Tp_size, const int32_t *Sj, int Sj_size,
                                     const int32_t *Sp, int Sp_size, int32_t nsdomains, int32_t nrows,
                                     int32_t row_start, int32_t row_stop, int32_t row_step);
/* amg_core::gauss_seidel_indexed, relaxation.h:736-745: the rows Id[row_start], Id[row_start + row_step], ... in that order,
 * in place (a row may be listed more than once) */
int pamg_gauss_seidel_indexed_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                                  const double *Ax, int Ax_size, double *x, int x_size,
                                  const double *b, int b_size, const int32_t *Id, int Id_size,
                                  int32_t row_start, int32_t row_stop, int32_t row_step);
int pamg_gauss_seidel_indexed_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                                  const float *Ax, int Ax_size, float *x, int x_size,
                                  const float *b, int b_size, const int32_t *Id, int Id_size,
                                  int32_t row_start, int32_t row_stop, int32_t row_step);
/* amg_core::block_jacobi_indexed, relaxation.h:1129-1138 (the kernel of cf_block_jacobi / fc_block_jacobi; Tx = inverse
 * diagonal blocks, indices = block rows) */
int pamg_block_jacobi_indexed_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                                  const double *Ax, int Ax_size, double *x, int x_size,
                                  const double *b, int b_size, const double *Tx, int Tx_size,
                                  const int32_t *indices, int indices_size,
                                  const double *omega, int omega_size, int32_t blocksize);
int pamg_block_jacobi_indexed_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                                  const float *Ax, int Ax_size, float *x, int x_size,
                                  const float *b, int b_size, const float *Tx, int Tx_size,
                                  const int32_t *indices, int indices_size,
                                  const float *omega, int omega_size, int32_t blocksize);
/* amg_core::gauss_seidel_ne relaxation.h:875-884 (Tx = 1/||row||^2), gauss_seidel_nr :939-948 (Ap/Aj/Ax = the CSC
 * arrays of A, z = running residual, Tx = 1/||column||^2; omega by value like the reference's F),
 * jacobi_ne :811-821 (Tx = the row-scaled residual "delta"; full row range only) */
int pamg_gauss_seidel_ne_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                            const double *Ax, int Ax_size, double *x, int x_size, const double *b, int b_size,
                            int32_t row_start, int32_t row_stop, int32_t row_step,
                            const double *Tx, int Tx_size, double omega);
int pamg_gauss_seidel_nr_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                            const double *Ax, int Ax_size, double *x, int x_size, double *z, int z_size,
                            int32_t col_start, int32_t col_stop, int32_t col_step,
                            const double *Tx, int Tx_size, double omega);
int pamg_jacobi_ne_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                      const double *Ax, int Ax_size, double *x, int x_size, const double *b, int b_size,
                      const double *Tx, int Tx_size, double *temp, int temp_size,
                      int32_t row_start, int32_t row_stop, int32_t row_step,
                      const double *omega, int omega_size);
int pamg_gauss_seidel_ne_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                            const float *Ax, int Ax_size, float *x, int x_size, const float *b, int b_size,
                            int32_t row_start, int32_t row_stop, int32_t row_step,
                            const float *Tx, int Tx_size, float omega);
int pamg_gauss_seidel_nr_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                            const float *Ax, int Ax_size, float *x, int x_size, float *z, int z_size,
                            int32_t col_start, int32_t col_stop, int32_t col_step,
                            const float *Tx, int Tx_size, float omega);
int pamg_jacobi_ne_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                      const float *Ax, int Ax_size, float *x, int x_size, const float *b, int b_size,
                      const float *Tx, int Tx_size, float *temp, int temp_size,
                      int32_t row_start, int32_t row_stop, int32_t row_step,
                      const float *omega, int omega_size);
/* amg_core::bsr_jacobi, relaxation.h:472-483 */
int pamg_bsr_jacobi_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                        const double *Ax, int Ax_size, double *x, int x_size,
                        const double *b, int b_size, double *temp, int temp_size,
                        int32_t row_start, int32_t row_stop, int32_t row_step,
                        int32_t blocksize, const double *omega, int omega_size);
int pamg_bsr_jacobi_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                        const float *Ax, int Ax_size, float *x, int x_size,
                        const float *b, int b_size, float *temp, int temp_size,
                        int32_t row_start, int32_t row_stop, int32_t row_step,
                        int32_t blocksize, const float *omega, int omega_size);
/* amg_core::block_jacobi, relaxation.h:1021-1033 (Tx = inverse diagonal blocks) */
int pamg_block_jacobi_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                          const double *Ax, int Ax_size, double *x, int x_size,
                          const double *b, int b_size, const double *Tx, int Tx_size,
                          double *temp, int temp_size, int32_t row_start, int32_t row_stop,
                          int32_t row_step, const double *omega, int omega_size,
                          int32_t blocksize);
int pamg_block_jacobi_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                          const float *Ax, int Ax_size, float *x, int x_size,
                          const float *b, int b_size, const float *Tx, int Tx_size,
                          float *temp, int temp_size, int32_t row_start, int32_t row_stop,
                          int32_t row_step, const float *omega, int omega_size,
                          int32_t blocksize);
/* amg_core::block_gauss_seidel, relaxation.h:1242-1252 */
int pamg_block_gauss_seidel_f64(const int32_t *Ap, int Ap_size, const int32_t *Aj,
                                int Aj_size, const double *Ax, int Ax_size, double *x,
                                int x_size, const double *b, int b_size, const double *Tx,
                                int Tx_size, int32_t row_start, int32_t row_stop,
                                int32_t row_step, int32_t blocksize);
int pamg_block_gauss_seidel_f32(const int32_t *Ap, int Ap_size, const int32_t *Aj,
                                int Aj_size, const float *Ax, int Ax_size, float *x,
                                int x_size, const float *b, int b_size, const float *Tx,
                                int Tx_size, int32_t row_start, int32_t row_stop,
                                int32_t row_step, int32_t blocksize);

/* amg_core::pinv_array, linalg.h:930-1000: every n x n block of AA (m, n, n; HOST) replaced by its pseudo-inverse
 * (one-sided Jacobi SVD per block, linalg.h:546-812) -- what get_block_diag(A, bs, inv_flag=True) (util/utils.py:603-692)
 * runs for the Dinv of the block smoothers.  TransA 'T': row-major blocks (how Python calls it), 'F': column-major.
 * n <= 6 (the reference itself leaves this routine at n >= 7, utils.py:682-687): larger n -> PAMG_E_UNSUPPORTED. */
int pamg_pinv_array_f64(double *AA, int AA_size, int32_t m, int32_t n, char TransA);
int pamg_pinv_array_f32(float *AA, int AA_size, int32_t m, int32_t n, char TransA);

/* amg_core::standard_aggregation, smoothed_aggregation.h:137-268 (bindings :49-75): the greedy aggregation of a
 * strength graph -- x[i] = aggregate of node i (-1: none), y[k] = root node of aggregate k, *naggs = their number (the
 * reference returns it) -- aggregate for aggregate, number for number.  The sequential first pass runs as ONE persistent
 * launch ordered by per-node turn counters (csrc/pamg_aggregate.hip).  Symmetric patterns without duplicate entries
 * (strength-of-connection matrices); anything else: PAMG_E_UNSUPPORTED. */
int pamg_standard_aggregation(int32_t n_row, const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size,
                              int32_t *x, int x_size, int32_t *y, int y_size, int32_t *naggs);
/* amg_core::fit_candidates (real), smoothed_aggregation.h:484-660 (bindings :134-170): Ap / Ai = the CSC arrays of AggOp
 * (the nodes of every aggregate), B = (n_row * K1, K2) candidates; Ax (nnz, K1, K2) receives the orthonormalised blocks
 * in CSC order, R (n_col, K2, K2) the coefficients.  One lane per aggregate, the reference's modified Gram-Schmidt loops. */
int pamg_fit_candidates_f64(int32_t n_row, int32_t n_col, int32_t K1, int32_t K2, const int32_t *Ap, int Ap_size,
                            const int32_t *Ai, int Ai_size, double *Ax, int Ax_size, const double *B, int B_size,
                            double *R, int R_size, double tol);
int pamg_fit_candidates_f32(int32_t n_row, int32_t n_col, int32_t K1, int32_t K2, const int32_t *Ap, int Ap_size,
                            const int32_t *Ai, int Ai_size, float *Ax, int Ax_size, const float *B, int B_size,
                            float *R, int R_size, float tol);
/* aggregation.tentative.fit_candidates (tentative.py:9-152) in one call: AggOp as CSR (Tp / Tj, at most one entry per
 * row), B -> Qx = the (K1, K2) blocks of the tentative prolongator in AggOp's ROW order (its BSR data array) and R; the
 * per-aggregate node lists (what AggOp.tocsc() gives the reference) are formed on the device. */
int pamg_fit_tentative_f64(int32_t n_fine, int32_t n_coarse, int32_t K1, int32_t K2, const int32_t *Tp, const int32_t *Tj,
                           const double *B, double *Qx, double *R, double tol);
int pamg_fit_tentative_f32(int32_t n_fine, int32_t n_coarse, int32_t K1, int32_t K2, const int32_t *Tp, const int32_t *Tj,
                           const float *B, float *Qx, float *R, float tol);

/* SciPy's bsr_transpose / csr_tocsc (sparsetools): B = A^T for A of n_brow x n_bcol blocks, R x C each (R = C = 1: CSR);
 * HOST arrays; Bp[n_bcol + 1], Bi[nblk], Bx[nblk * C * R] receive the arrays `A.T` holds in SciPy -- the blocks of a
 * column in the order of their rows, blocks transposed.  R = P.T of the SA setup (aggregation.py:394-397).
 * PAMG_E_UNSUPPORTED: a column with more than 4096 blocks. */
int pamg_bsr_transpose_f64(int32_t n_brow, int32_t n_bcol, int32_t R, int32_t C, const int32_t *Ap, const int32_t *Aj,
                           const double *Ax, int32_t *Bp, int32_t *Bi, double *Bx);
int pamg_bsr_transpose_f32(int32_t n_brow, int32_t n_bcol, int32_t R, int32_t C, const int32_t *Ap, const int32_t *Aj,
                           const float *Ax, int32_t *Bp, int32_t *Bi, float *Bx);

/* ------------------------------------------------------ Layer 2: resident engine (HBM) */
/* Operator handle: uploads CSR/BSR arrays (HOST pointers) to HBM once and analyses them
 * (row-block plan for the LDS-streamed kernels; dependency-level schedules for the
 * order-exact Gauss-Seidel sweeps are built lazily).
 *   flavour: PAMG_CSR for a reference csr_array, PAMG_BSR for a reference bsr_array --
 *   selects which reference loop's arithmetic order the smoothers reproduce (amg_core
 *   jacobi/gauss_seidel vs bsr_jacobi/bsr_gauss_seidel; SpMV is the same for both).
 *   R x C is the block size ((1,1) for CSR).  */
#define PAMG_CSR 0
#define PAMG_BSR 1
int pamg_matrix_create(pamg_matrix_t *A, int dtype, int flavour, int n_brow, int n_bcol,
                       int R, int C, const int32_t *Ap, const int32_t *Aj, const void *Ax);
int pamg_matrix_destroy(pamg_matrix_t A);
/* info[0]=rows info[1]=cols info[2]=stored scalars info[3]=row blocks info[4]=lds entries
 * per block info[5]=bytes resident in HBM info[6]=fwd GS levels (0 = not analysed)
 * info[7]=bwd GS levels */
int pamg_matrix_info(pamg_matrix_t A, int64_t info[8]);
/* tuning knobs (speed only: every setting computes the same bits).  key 0 = LDS entries per row
 * range, 1 = entries per lane in the staging phase (2; 1 and 4 were measured no better and retired in round 5), 2 = max rows per range (these
 * re-plan the operator); 3 = flow_cap: an order-exact sweep whose schedule averages <= flow_cap/16
 * row ranges per dependency level runs as ONE persistent single-workgroup launch (default 32);
 * 5 = scheduler of the scalar order-exact sweeps: 0 automatic (narrow -> single workgroup, else the
 * granular sweep), 1 one launch per dependency level, 2 granular sync-free sweep (one persistent
 * launch, element-level hand-off, no barriers), 3 single workgroup; block sweeps read 2 as "grid
 * with a barrier per level"; 6 = cap on the persistent grid (0 = automatic); 7 = granular sweep
 * inside one XCD's L2: 0 automatic (small operators), 1 always, 2 never; 8 = streaming flags of the
 * whole-operator kernels: bit 0 non-temporal loads of the operator stream, bit 1 XCD-aware
 * row-range order (9, LDS-staged x windows, was retired in round 5: 4-7 % slower than the direct gather);
 * 5 also accepts 5 = TILED sweep (one persistent workgroup per contiguous chunk of rows; dependency chains
 * stay in LDS, only chunk-crossing edges use the global hand-off); 11 = record time stamps of the granular /
 * tiled sweep (diagnostics, pamg_matrix_gs_profile); 12 = tiles of the tiled sweep (0 = automatic),
 * 13 = its LDS ring slots (0 = automatic, else a power of two, 64..8192), 14 = its entries per step
 * (0 = automatic, <= 1020), 15 = let the automatic choice (key 5 = 0) prefer the tiled sweep, 16 = cap on the
 * steps resident in LDS per tile (0 = automatic), 17 = cap on the steps its loader keeps in flight (-1 = automatic),
 * 18 = tile shapes: 1 (default) pencils on three-band grid stencils, 0 contiguous chunks of the visit order always;
 * 19 = 16-bit windowed column stream of the whole-operator kernels (default 1; 0 = 32-bit columns);
 * 20 = entries per row range of the level schedules of the order-exact sweeps (0 = automatic: key 0's value, 512 where the
 * multi-XCD granular sweep runs SA-like rows; else 64..2048);
 * 21 = 8-bit value codes of the whole-operator kernels (default 1): an operator with at most 256 distinct values
 * (the stencils of pyamg.gallery: 2) streams one byte per value, the kernel looks the value up in an LDS copy of
 * the dictionary -- same bits, same products; needs key 19;
 * 22 = row-gather form of the whole-operator kernels on operators with value codes (default 1 there): lane = row, the j-th
 * entries of 64 consecutive rows in one gather instruction (coalesced on stencils), products summed in storage order in
 * registers; 0 = the LDS-staged kernel on the codes;
 * 23 = row-pattern form where plan_rowpat found a table (see pamg_matrix_row_patterns): 1 (default) the fastest form the rows
 * allow -- the row-MASK kernels when every list is the longest list with entries left out (a constant-coefficient stencil: one
 * mask byte per row, offsets and values are launch constants, no table, no prologue; on a 7-point lattice whose extents fit
 * 64 x 4 x kz tiles the lattice form csr_rowmask3d_kernel), else the table kernel --, 3 the table kernel (one row per lane)
 * always, 4 the linear row-mask kernel instead of the lattice form, 0 off (2, two consecutive rows per lane, measured 28 % slower --
 * profiles/r04_microbench_rowpat_two_rows_per_lane_slower.json -- and retired in round 5);
 * 31 = planes per lane of the lattice form (2 | 4 | 8, default 8); 32 = flags of the row-mask kernels (default 3): bit 0 (the
 * streams touched once -- mask, b, result -- nontemporal) is always on since round 5 and ignored, bit 1 plane-by-plane XCD order (XCD j takes the j-th eighth of
 * every plane), bit 2 XCD-contiguous eighths of the rows instead.
 * NOT speed-only -- 24 = ORDER of the row sums of the scalar Gauss-Seidel / SOR sweeps: 0 (default of a bare operator) =
 * order-exact, every sum runs in storage order with an IEEE division, results are the reference's bit for bit
 * (amg_core/relaxation.h:48-76,116-145,185-266); 1 = FAST order: the same sweep order over the rows (same dependency
 * DAG, same iterates in exact arithmetic), but L lanes of a wave share a row, add their products in parallel (DPP
 * butterfly) and finish with (b - sum) * (1 / a_ii) -- agrees with the reference to rounding (a few ulp per sweep), which
 * is what BASELINE's "residual norms within 1e-10" asks for; schedules the lane form cannot hold (rows with more than
 * 256 off-diagonal entries, padding above 4x) keep the exact kernels.  25 = lanes per row of the fast order (0 = automatic,
 * else 4|8|16|32|64), 26 = its persistent workgroups (0 = automatic), 27 = 1: the fast order also takes wide schedules
 * (>= 2048 rows per dependency level), which the tiled exact sweep keeps by default, 28 = flags of the fast order (bit 0, default
 * on: a wave that runs ahead of the sweep polls ONE gate operand instead of all its operands until the sweep is one
 * dependency level away; bit 3: the same in the line scan, off; bits 1, 2 retired with the slab form in round 5),
 * 30 = line-scan form of the fast order (default 1) where consecutive swept rows are coupled AND
 * enough lines run side by side to beat the lane form by the planner's estimate (3-D grids; not 2-D grids in natural order); 2 = wherever it applies.
 * 33 = MERGED fast order (round 6; f64 Gauss-Seidel): dependency levels of the sweep eliminated algebraically into one super-level at most -- 0 (default)
 * automatic (3 on levels above 131 072 rows, 6 below, off under 12 entries per row or 8 dependency levels), 1 never, 2..8; the sweep then pays one
 * hand-off per super-level; same iterates in exact arithmetic, another association in floating point; groups are closed early where a merged row would
 * exceed 512 operands (256 with two rows per wave) or its growth factor 1e3 (csrc/pamg_lanem_plan.h; pamg_matrix_lanem_info); 34 = its persistent waves
 * as tenths of the rows of an average super-level (default 40); 35 = its rows per wave: 1 (64 lanes per row), 2 (32 lanes per row, rows of a
 * super-level paired by length), 0 (default) = 2 on levels above 131 072 rows.
 * 36 = unused dynamic LDS (bytes) added to the launches of the staged whole-operator kernel: caps its workgroups per CU (a measurement knob: on the SA-level
 * operators the instantiation choice of key 8 bit 5 already sits at the best occupancy, profiles/r06_microbench_sa_ops_lds_pad.json).
 * Key 8, bit 5 (round 6): an operator WITHOUT 8-bit value codes through the kernel instantiation that carries their paths (same arithmetic, another
 * instruction schedule; pamg_matrix_autotune times both).
 * Returns PAMG_E_STATE while a solver holds the operator (captured graphs point into the plans). */
int pamg_matrix_tune(pamg_matrix_t A, int key, int value);
/* n_values = size of the operator's value dictionary when the whole-operator kernels stream 8-bit value codes
 * (tune key 21), 0 when they stream the values themselves. */
int pamg_matrix_value_codes(pamg_matrix_t A, int *n_values);
/* n_patterns = size of the operator's row-pattern table when the whole-operator kernels run in the row-pattern form (tune
 * key 23: square operators with value codes whose rows are mostly one of <= 255 lists of (column - row, value) pairs --
 * constant-coefficient stencils; one byte per such row instead of the row's codes), 0 otherwise. */
int pamg_matrix_row_patterns(pamg_matrix_t A, int *n_patterns);
/* The row-mask form of the row patterns (tune key 23 = 1 or 4 where every list is one list with entries left out):
 * info[0] entries of that list (0 = the operator has no mask form, or another form is selected), [1] rows that walk the CSR
 * arrays, [2] 1 = the lattice kernel runs (64 x 4 x kz tiles), [3] rows per lattice line, [4] rows per plane, [5] planes
 * per lane (key 31), [6] flags (key 32), [7] workgroups launched per whole-operator kernel. */
int pamg_matrix_row_masks(pamg_matrix_t A, long long info[8]);
/* Pick the LDS window (key 0) and streaming flags (key 8) of the whole-operator kernels by timing
 * y = A x on the device with a few candidates (results are bit-identical for every choice; this
 * is speed only).  allow_cap = 0 keeps the LDS window (level schedules depend on it).  Operators
 * below 4M stored entries are left alone.  Synchronises. */
int pamg_matrix_autotune(pamg_matrix_t A, int allow_cap);
/* Diagnostics of the granular sweep (tune key 11): per row range of schedule `which` (0 forward,
 * 1 backward) eight 64-bit words {arrival, gate open, polled, staged, finished (wall clock, 10 ns),
 * XCD id, workgroup id, dependency level}.  out == NULL: only *count.  Synchronises. */
int pamg_matrix_gs_profile(pamg_matrix_t A, int which, long long *out, int64_t capacity, int64_t *count);
/* Plan of the tiled sweep for schedule `which` (built by the first sweep / pamg_solver_finalize):
 * {tiles, ring slots, geometry = code chunks per step | LDS slots << 8 | gather depth << 16 | loader steps in
 * flight << 24, steps, early entries served from LDS, early entries served by the global hand-off, publishing
 * rows, LDS bytes}; all zero when that schedule has no tile plan. */
int pamg_matrix_tile_info(pamg_matrix_t A, int which, int64_t info[8]);
/* Layout of the lane-parallel fast-order sweep (tune key 24) for schedule `which`: {lanes per row, entry slots per lane,
 * groups (one wave each), entry slots, entries that wait for a new value, workgroups of the last launch (the one-XCD form
 * launches 8x what stays), groups of the widest dependency level, bytes}; all zero when that schedule has no lane layout.  pamg_matrix_lane_profile: with tune
 * key 11, per group four 64-bit words {start, last operand seen, published (wall clock, 10 ns), XCD | workgroup << 4}.
 * Block operators (bs > 1): the same fields for the block-row lane form of pamg_matrix_block_gauss_seidel (csrc/pamg_blane.hip: lanes per BLOCK row,
 * BLOCKS per lane, block slots, blocks that wait for a new x_j). */
int pamg_matrix_lane_info(pamg_matrix_t A, int which, int64_t info[8]);
/* Plan of the MERGED lane-parallel sweep (round 6; tune key 33: dependency levels eliminated into one super-level at most -- 0 automatic, 1 never;
 * key 34: waves per average super-level x 10; csrc/pamg_lanem_plan.h) for schedule `which`: {super-levels (hand-offs of this form), dependency
 * levels (hand-offs of the unmerged form), rows, 64-slot operand units, operands polled from earlier super-levels, operands read from the snapshot
 * of x, operands read from b, longest merged row, levels merged at most, groups closed early by row length, closed early by the growth bound,
 * workgroups of the last launch}; *growth (may be NULL) = the largest accepted growth factor sum_r |T_ir| |a_ii|.  All zero when the schedule
 * runs unmerged.  The merged sweep computes the reference's Gauss-Seidel iterates (amg_core::gauss_seidel, relaxation.h:48-76) in another
 * association: equal in exact arithmetic, to rounding in floating point. */
int pamg_matrix_lanem_info(pamg_matrix_t A, int which, int64_t info[12], double *growth);
/* Fast order (tune key 24 = 1) of the BSR POINT sweep (amg_core::bsr_gauss_seidel, relaxation.h:185-266: what relaxation.gauss_seidel runs on a
 * block operator): the same rows in the same order are the scalar Gauss-Seidel sweep of the flattened operator, so the block operator builds a
 * scalar CSR twin of itself with its schedules and sweeps it in the lane-parallel / merged / line-scan form (same iterates to rounding, like every
 * fast-order sweep; order 'exact' keeps the block kernels and the reference's bits).  state: 0 = no twin (exact order, or not swept yet),
 * 1 = the twin carries the point sweeps, 2 = no fast-order form fits the flattened rows: the exact block kernels sweep. */
int pamg_matrix_point_twin(pamg_matrix_t A, int *state);
/* First row (in the merged plan's order) of every super-level, nsuper + 1 values; out == NULL: only *count.  With tune key 11 the merged sweep records per row
 * {arrival | polling rounds << 52, operand slots arrived, all operands present, published} (pamg_matrix_lane_profile, 10 ns ticks). */
int pamg_matrix_lanem_levels(pamg_matrix_t A, int which, int64_t *out, int64_t capacity, int64_t *count);
/* Layout of the lane-parallel fast-order Kaczmarz sweep (tune key 24 = 1 on the operator handed to pamg_matrix_kaczmarz; csrc/pamg_kz_plan.h)
 * of the operator's `which`-th cached line schedule (0 .. 3, in the order the sweep ranges were first used): {lanes per line, entry slots per
 * lane, groups, dependency levels, groups of the widest level, workgroups of the last launch, bytes, 0}; all zero when that schedule has none. */
int pamg_matrix_kz_info(pamg_matrix_t A, int which, int64_t info[8]);
/* Layout of the line-scan fast-order sweep (tune keys 24 / 30: banded operators swept over consecutive rows, i.e. grid
 * stencils in their natural order -- a run of rows each coupled to its predecessor is a first-order linear recurrence, finished
 * 64 rows at a time by a scan) for schedule `which`: {entry slots per row, chunks (<= 64 rows, one wave step each), lines (chained
 * chunks, one wave each), levels of the line graph, entries that wait for a new value, workgroups of the last launch, lines of
 * the widest level, bytes}; all zero when that schedule has no line layout. */
int pamg_matrix_line_info(pamg_matrix_t A, int which, int64_t info[8]);
int pamg_matrix_lane_profile(pamg_matrix_t A, int which, long long *out, int64_t capacity, int64_t *count);
/* Row-subset copy of a CSR operator (rows: HOST list, kept in list order) for the indexed smoothers,
 * and amg_core::jacobi_indexed (relaxation.h:382-427) on it: every listed row of x is relaxed from the
 * OLD x (x, b: DEVICE vectors of the parent's size; work: DEVICE, one value per listed row). */
int pamg_matrix_subset_rows(pamg_matrix_t A, const int32_t *rows, int nrows, pamg_matrix_t *sub);
/* Kaczmarz-type sweeps over the rows of L, order-exact (dependency levels over shared column indices):
 * nr = 0: amg_core::gauss_seidel_ne (relaxation.h:875-904) with L = A, v = x, b, Dinv = 1/||row||^2;
 * nr = 1: amg_core::gauss_seidel_nr (relaxation.h:939-975) with L = CSR of A^T (the CSC arrays of A), v = the
 * running residual z, Dinv = 1/||column||^2, xout = x (b unused).  sweep: PAMG_FORWARD / BACKWARD /
 * SYMMETRIC (forward then backward per iteration, on the SAME v -- the reference's gauss_seidel_nr wrapper
 * refreshes z = b - A x before every directional call: callers that mirror it run directional sweeps). */
int pamg_matrix_kaczmarz(pamg_matrix_t L, int nr, void *v, const void *b, const void *Dinv, double omega,
                         int sweep, int iterations, void *xout, pamg_stream_t s);
int pamg_matrix_jacobi_indexed(pamg_matrix_t sub, void *x, const void *b, double omega, void *work,
                               pamg_stream_t s);
/* *error != 0: a persistent sweep of this operator hit its spin bound (synchronises) */
int pamg_matrix_flow_error(pamg_matrix_t A, int *error);

/* SpMV family (x, y, b, v are DEVICE vectors of the operator's dtype).  mode:           */
#define PAMG_SPMV_SET      0   /* y  = A x                                              */
#define PAMG_SPMV_ACC      1   /* y += A x              (x += P x_c, multilevel.py:660) */
#define PAMG_SPMV_RESID    2   /* y  = b - A x          (multilevel.py:612)             */
#define PAMG_SPMV_AXPBY    3   /* y  = c*v + A x        (relaxation.py:657 Horner step) */
#define PAMG_SPMV_ACC_AXPBY 4  /* y += c*v + A x        (Horner last step fused with x += h) */
int pamg_matrix_spmv(pamg_matrix_t A, int mode, const void *x, const void *b_or_v, double c,
                     void *y, pamg_stream_t s);
/* Row shards (what csrc/pamg_dist.hip drives; exposed for integrators with their own transport): the operator is a block of
 * rows in local numbering [owned columns | halo columns].  pamg_matrix_split_ranges sorts its row ranges into those that read
 * owned columns only (part 1, the interior) and those that read the halo (part 2, the boundary); pamg_matrix_spmv_part runs
 * one part (0 = every range) -- interior while the halo values travel, boundary once they have landed.  Same arithmetic
 * per row as pamg_matrix_spmv: the two parts together write exactly its bits. */
int pamg_matrix_split_ranges(pamg_matrix_t A, int64_t n_owned_cols);
int pamg_matrix_spmv_part(pamg_matrix_t A, int part, int mode, const void *x, const void *b_or_v, double c, void *y,
                          pamg_stream_t stream);
/* ||b - A x||_2^2 without storing the residual; result (one value of dtype f64) written
 * to DEVICE address out_sumsq (multilevel.py:545,567 convergence check). */
int pamg_matrix_resid_sumsq(pamg_matrix_t A, const void *x, const void *b, double *out_sumsq,
                            pamg_stream_t s);

/* Smoothers, in place on DEVICE x (work = DEVICE scratch vector of length n; for the
 * polynomial smoother 2n).  sweep: */
#define PAMG_FORWARD   0
#define PAMG_BACKWARD  1
#define PAMG_SYMMETRIC 2
int pamg_matrix_jacobi(pamg_matrix_t A, void *x, const void *b, void *work, double omega,
                       int iterations, pamg_stream_t s);
/* one out-of-place sweep x_out[0:nrows] = jacobi(x_in).  A may be a row shard of the global
 * operator in local numbering (n_cols >= n_rows, owned columns first: column i of row i is
 * its diagonal), x_in then holds [owned | halo] values -- the multi-GPU building block. */
int pamg_matrix_jacobi_step(pamg_matrix_t A, const void *x_in, const void *b, void *x_out,
                            double omega, pamg_stream_t s);
/* the same for amg_core::block_jacobi (relaxation.h:1021-1090) on a square-block BSR operator or a row shard of
 * one (block rows cut, block columns [owned | halo]); Dinv: DEVICE, the shard's n_brow x bs x bs inverted diagonal
 * blocks.  pamg_matrix_jacobi_step accepts such operators too (point Jacobi on BSR, relaxation.h:472-562). */
int pamg_matrix_block_jacobi_step(pamg_matrix_t A, const void *Dinv, const void *x_in, const void *b, void *x_out,
                                  double omega, pamg_stream_t s);
/* amg_core::block_jacobi_indexed (relaxation.h:1129-1199) on a resident square-block operator: one block-Jacobi step
 * from the old x, taken over for the listed rows only.  idx: DEVICE, the SCALAR indices of the listed block rows
 * (row * blocksize + k); work: DEVICE scratch of x's length. */
int pamg_matrix_block_jacobi_indexed(pamg_matrix_t A, const void *Dinv, void *x, const void *b, const int32_t *idx,
                                     int64_t nidx, double omega, void *work, pamg_stream_t s);
/* gauss_seidel / sor as the reference's Python wrappers run them (relaxation.py:265-346,
 * 100-154, quirks included: 'symmetric' ignores omega, BSR flavour ignores omega). */
int pamg_matrix_gauss_seidel(pamg_matrix_t A, void *x, const void *b, int sweep, double omega,
                             int iterations, pamg_stream_t s);
/* relaxation.polynomial (relaxation.py:585-659); coeffs is a HOST array; x_is_zero != 0
 * asserts x == 0 on entry (the reference tests norm(x) == 0, relaxation.py:649). */
int pamg_matrix_polynomial(pamg_matrix_t A, void *x, const void *b, void *work,
                           const double *coeffs, int ncoeffs, int iterations, int x_is_zero,
                           pamg_stream_t s);
/* true block relaxation with DEVICE Dinv (n_brow x bs x bs) (relaxation.py:423-582) */
int pamg_matrix_block_jacobi(pamg_matrix_t A, void *x, const void *b, void *work,
                             const void *Dinv, double omega, int iterations, pamg_stream_t s);
int pamg_matrix_block_gauss_seidel(pamg_matrix_t A, void *x, const void *b, const void *Dinv,
                                   int sweep, int iterations, pamg_stream_t s);

/* pinv_array on a DEVICE array (m, n, n), in place, stream-ordered */
int pamg_dev_pinv_array(int dtype, void *AA, int64_t m, int n, int transA, pamg_stream_t s);

/* BLAS-1 on DEVICE vectors */
int pamg_vec_sumsq(int dtype, int64_t n, const void *x, double *out_sumsq, pamg_stream_t s);
int pamg_vec_axpy(int dtype, int64_t n, double a, const void *x, void *y, pamg_stream_t s);
int pamg_vec_scale(int dtype, int64_t n, double a, const void *x, void *y, pamg_stream_t s);
/* dst[k] = src[idx[k]], k < n  (halo packing; idx is a DEVICE int32 array) */
/* y = a .* b (element-wise) */
int pamg_vec_mul(int dtype, int64_t n, const void *a, const void *b, void *y, pamg_stream_t s);
int pamg_vec_gather(int dtype, int64_t n, const int32_t *idx, const void *src, void *dst,
                    pamg_stream_t s);

/* Renumbering of an INTERIOR level (host utilities, no device work).  The unknowns of a level l >= 1 are the solver's own -- the reference
 * hands level-l vectors to nobody (multilevel.py:566-662 keeps them inside __solve) -- so the device hierarchy may number them as its
 * gathers like: A_l' = Pi A_l Pi^T, P_{l-1}' = P_{l-1} Pi^T, R_{l-1}' = Pi R_{l-1}, P_l' = Pi P_l, R_l' = R_l Pi^T.  Rows are moved and
 * columns renamed, the entries of a row keep their stored order: every row sum is the reference's, bit for bit.
 * pamg_csr_renumber: row i of B = row row_old_of_new[i] of A (NULL = unchanged), column c becomes col_new_of_old[c] (NULL = unchanged);
 * Bp [nrows + 1], Bj, Bx [nnz] are the caller's HOST arrays.  PAMG_E_ARG when row_old_of_new is not a permutation or a column is out
 * of range.  pamg_csr_row_argmax_abs: out[i] = column of the entry of largest magnitude of row i (first on ties, -1 for an empty row) --
 * the aggregate an unknown falls into on the next level, which is what the blob-by-blob order is built from
 * (pyamg_amd/hierarchy.py renumber_levels). */
int pamg_csr_renumber(int dtype, int64_t nrows, int64_t ncols, const int32_t *Ap, const int32_t *Aj, const void *Ax,
                      const int32_t *row_old_of_new, const int32_t *col_new_of_old, int32_t *Bp, int32_t *Bj, void *Bx);
int pamg_csr_row_argmax_abs(int dtype, int64_t nrows, const int32_t *Ap, const int32_t *Aj, const void *Ax, int32_t *out);
/* Rows of a CSR / BSR operator (HOST arrays) sorted by column in place, on the host threads: scipy's sort_indices(), which the reference
 * runs on every Galerkin product before reading its diagonal (util/utils.py:583).  block = values per stored entry (R * C; 1 for CSR).
 * Stable, so equal columns keep their stored order like SciPy's. */
int pamg_csr_sort_rows(int dtype, int64_t nrows, const int32_t *Ap, int32_t *Aj, void *Ax, int block);
/* Host threads the library's planners count on: min(hardware threads, affinity mask, cgroup CPU quota) -- a container may see 256
 * hardware threads and own 16 cores (PAMG_HOST_THREADS overrides).  fresh != 0 evaluates the environment again instead of the
 * per-process value. */
int pamg_host_cpus(int fresh);

/* Hierarchy / cycle / outer iteration (MultilevelSolver, multilevel.py:17-662).         */
#define PAMG_SMOOTH_NONE        0
#define PAMG_SMOOTH_JACOBI      1
#define PAMG_SMOOTH_GS          2
#define PAMG_SMOOTH_SOR         3
#define PAMG_SMOOTH_POLY        4
#define PAMG_SMOOTH_BLOCK_JACOBI 5
#define PAMG_SMOOTH_BLOCK_GS    6
#define PAMG_SMOOTH_CF_JACOBI   7   /* relaxation.cf_jacobi  relaxation.py:1141-1203: C sweeps, then F sweeps */
#define PAMG_SMOOTH_FC_JACOBI   8   /* relaxation.fc_jacobi  relaxation.py:1206-1268: F sweeps, then C sweeps */
#define PAMG_SMOOTH_GS_NE       9   /* relaxation.gauss_seidel_ne (Kaczmarz)  relaxation.py:815-901  */
#define PAMG_SMOOTH_GS_NR      10   /* relaxation.gauss_seidel_nr             relaxation.py:904-988  */
#define PAMG_SMOOTH_JACOBI_NE  11   /* relaxation.jacobi_ne                   relaxation.py:741-812  */
#define PAMG_SMOOTH_CF_BLOCK_JACOBI 12   /* relaxation.cf_block_jacobi  relaxation.py:1271-1340: C block rows, then F */
#define PAMG_SMOOTH_FC_BLOCK_JACOBI 13   /* relaxation.fc_block_jacobi  relaxation.py:1342-1411: F block rows, then C */
#define PAMG_SMOOTH_SCHWARZ    14   /* relaxation.schwarz  relaxation.py:157-262 (also what strength_based_schwarz runs) */
#define PAMG_SMOOTH_KRYLOV     15   /* a Krylov method as smoother  smoothing.py:794-830 (pamg_solver_set_krylov_smoother) */
#define PAMG_KRYLOV_CG    0         /* krylov/_cg.py   */
#define PAMG_KRYLOV_GMRES 1         /* krylov/_gmres_householder.py (the reference's default orthogonalisation) */
#define PAMG_KRYLOV_CGNE  2         /* krylov/_cgne.py */
#define PAMG_KRYLOV_CGNR  3         /* krylov/_cgnr.py */
#define PAMG_CYCLE_V 0
#define PAMG_CYCLE_W 1
#define PAMG_CYCLE_F 2
#define PAMG_CYCLE_AMLI 3   /* multilevel.py:628-656 (2 A-orthogonalised inner corrections per level) */
int pamg_solver_create(pamg_solver_t *S, int dtype);
int pamg_solver_destroy(pamg_solver_t S);
/* levels are added fine -> coarse; P and R are NULL for the coarsest level.  The solver
 * borrows the handles (caller keeps them alive and destroys them afterwards). */
int pamg_solver_add_level(pamg_solver_t S, pamg_matrix_t A, pamg_matrix_t P, pamg_matrix_t R);
/* which: 0 = presmoother, 1 = postsmoother.  coeffs / Dinv are HOST arrays (copied). */
int pamg_solver_set_smoother(pamg_solver_t S, int level, int which, int kind, int iterations,
                             double omega, int sweep, const double *coeffs, int ncoeffs,
                             const void *Dinv, int blocksize);
/* CF / FC Jacobi (the AIR solver's default F/C relaxation): per outer iteration c_iterations sweeps of
 * amg_core::jacobi_indexed (relaxation.h:382-427) over Cpts and f_iterations over Fpts, in the order the
 * kind names.  Fpts / Cpts: HOST row lists (copied; the solver keeps row-subset copies of the level's
 * operator).  CSR levels only. */
int pamg_solver_set_cf_smoother(pamg_solver_t S, int level, int which, int kind, int iterations,
                                int f_iterations, int c_iterations, double omega, const int32_t *Fpts,
                                int nF, const int32_t *Cpts, int nC);
/* CF / FC block Jacobi on a level with square blocks of `blocksize` >= 2: amg_core::block_jacobi_indexed
 * (relaxation.h:1129-1199) over the C then the F block rows (or F then C).  Dinv: HOST, the inverted diagonal blocks
 * (block rows x blocksize x blocksize, copied); Fpts / Cpts: HOST lists of BLOCK rows (copied). */
int pamg_solver_set_cf_block_smoother(pamg_solver_t S, int level, int which, int kind, int iterations,
                                      int f_iterations, int c_iterations, double omega, const void *Dinv,
                                      int blocksize, const int32_t *Fpts, int nF, const int32_t *Cpts, int nC);
/* Multiplicative overlapping Schwarz (relaxation.schwarz, relaxation.py:157-262 -> amg_core::overlapping_schwarz_csr,
 * relaxation.h:1420-1492) on a resident scalar CSR operator A (borrowed; the reference sweeps lvl.Acsr, whose rows are
 * sorted).  Sp/Sj/Tp/Tx: HOST, copied -- the subdomains' row lists and their inverted diagonal blocks exactly as the
 * reference's schwarz_parameters built them.  The sweep visits subdomains row_start, row_start + row_step, ... like
 * the reference; subdomains of one dependency level run side by side, results are bit-identical.  Subdomains of more
 * than 2048 rows: PAMG_E_UNSUPPORTED. */
typedef struct pamg_schwarz_s *pamg_schwarz_t;
int pamg_schwarz_create(pamg_schwarz_t *out, pamg_matrix_t A, int nsub, const int32_t *Sp, const int32_t *Sj,
                        const int32_t *Tp, const void *Tx);
int pamg_schwarz_destroy(pamg_schwarz_t h);
int pamg_schwarz_sweep(pamg_schwarz_t h, void *x, const void *b, int row_start, int row_stop, int row_step,
                       pamg_stream_t s);
int pamg_schwarz_info(pamg_schwarz_t h, int64_t info[4]);   /* subdomains, largest, dependency levels fwd / bwd */
/* Scheduler of the sweep: 0 (default) = ONE persistent launch per sweep -- every update of a row gets its own slot of a hand-off buffer
 * (version v of row i), every read is told which version the reference's sequential sweep would find, co-resident waves walk the
 * subdomains in level order and poll the slots they read (csrc/pamg_schwarz.hip); 1 = one launch per dependency level (always live;
 * what pamg_solver_solve switches to after a PAMG_E_TIMEOUT, and what a schedule runs as when a row is updated more than 255 times, a
 * subdomain lists a row twice or the version table would exceed a gigabyte).  Same arithmetic, same bits.  PAMG_SCHWARZ_LEVELS=1 forces 1.
 * pamg_schwarz_error: after a synchronising call -- did a wave of a persistent sweep give up waiting (1) since the last query? */
int pamg_schwarz_set_mode(pamg_schwarz_t h, int mode);
int pamg_schwarz_error(pamg_schwarz_t h, int *error);
/* Schwarz as a level's smoother: `iterations` x (forward | backward | forward then backward) sweeps.  Ar: the level's
 * operator as the reference's smoother sees it (lvl.Acsr; NULL = the level operator itself); the solver borrows it. */
int pamg_solver_set_schwarz_smoother(pamg_solver_t S, int level, int which, int iterations, int sweep, pamg_matrix_t Ar,
                                     int nsub, const int32_t *Sp, const int32_t *Sj, const int32_t *Tp, const void *Tx);
/* Normal-equation smoothers (f64/f32 CSR-like levels).  Dinv: HOST vector of the level's size -- 1/||row||^2
 * (GS_NE, JACOBI_NE) or 1/||column||^2 (GS_NR), computed by the caller exactly as the reference's
 * get_diagonal(A, norm_eq=..., inv=True) (util/utils.py:583-598).  At (borrowed handle, kept alive by the caller):
 * GS_NE: NULL; GS_NR: the CSR form of A^T, i.e. the CSC arrays of A (sorted); JACOBI_NE: the same with every
 * value pre-multiplied by omega (the reference multiplies omega * a_ij first, relaxation.h:835).  Ar (GS_NR,
 * borrowed, may be NULL): the level operator with SORTED rows for the residual r = b - A x -- the reference forms it
 * with the CSC matrix, i.e. per row in ascending column order; NULL = the level's own A already is sorted. */
int pamg_solver_set_ne_smoother(pamg_solver_t S, int level, int which, int kind, int iterations, double omega,
                                int sweep, const void *Dinv, pamg_matrix_t At, pamg_matrix_t Ar);
/* A Krylov method as pre- / post-smoother of a level (smoothing.py:794-830: x[:] = cg | gmres | cgne | cgnr (A, b, x0 = x, tol,
 * maxiter[, restart])[0], M = None), and -- set as smoother 0 of the LAST level followed by pamg_solver_set_coarse_relax -- as
 * coarse solver (multilevel.py:752-762: from x = 0).  maxiter = 0 / restart = 0: the method's own default (None in the
 * reference).  At (cgne / cgnr; borrowed, kept alive by the caller): the CSR form of A^H.  The methods' stopping rules read
 * scalars back on the host, so a solver with such a smoother runs its cycles without hipGraph capture. */
int pamg_solver_set_krylov_smoother(pamg_solver_t S, int level, int which, int method, double tol, int maxiter, int restart,
                                    pamg_matrix_t At);
/* coarsest solve x_c = M b_c with HOST row-major M (n_c x n_c); M == NULL: x_c = 0
 * (multilevel.py:717-721, 801-803) */
int pamg_solver_set_coarse_dense(pamg_solver_t S, const void *M, int n_c);
/* coarsest solve = the relaxation method set as smoother `which = 0` of the LAST level, applied from x_c = 0
 * (coarse_solver='gauss_seidel' / 'jacobi' / 'chebyshev' ...: multilevel.py:765-782).  Call after the level's
 * pamg_solver_set_*smoother. */
int pamg_solver_set_coarse_relax(pamg_solver_t S);
/* Coarsest-level solve by a function of the CALLER on the host (multilevel.py:752-762: coarse_solver='bicgstab' | 'cgs' | 'qmr' |
 * 'minres' | ..., :786-788: a callable): the coarse right-hand side (n_c values, host copy) is handed to fn, which fills x and
 * returns 0.  Such solvers are not linear in b, so they cannot be tabulated like 'pinv' / 'splu'; n_c is tiny, the two copies
 * and the synchronisation are the price (cycles of such a solver are not replayed from a hipGraph). */
typedef int (*pamg_coarse_host_fn)(void *user, const void *b_host, void *x_host, int64_t n_c);
int pamg_solver_set_coarse_host(pamg_solver_t S, pamg_coarse_host_fn fn, void *user, int n_c);
int pamg_solver_finalize(pamg_solver_t S);
/* one multigrid cycle on DEVICE x, b of level 0 (multilevel.py:584-662) */
int pamg_solver_cycle(pamg_solver_t S, void *x, const void *b, int cycle, int cycles_per_level,
                      pamg_stream_t s);
/* the accel=None branch of MultilevelSolver.solve (multilevel.py:537-582) on DEVICE x (in:
 * initial guess, out: solution) and b.  residuals: HOST array of maxiter+1 doubles (may be
 * NULL); *n_iter = cycles run; *info = 0 if ||r|| < tol*||b|| was met else n_iter.
 * check_every: 1 = test convergence after every cycle as the reference does; k > 1 = read
 * the norms back every k cycles only (fewer host syncs; may overshoot by < k cycles). */
int pamg_solver_solve(pamg_solver_t S, void *x, const void *b, double tol, int maxiter,
                      int cycle, int cycles_per_level, int check_every, double *residuals,
                      int *n_iter, int *info, pamg_stream_t s);
/* Preconditioned CG with the resident cycle as preconditioner, all vectors on the DEVICE
 * (reference: pyamg/krylov/_cg.py:98-198, stopping criterion 'rr': ||r|| < tol*||b||, as driven
 * by MultilevelSolver.solve(accel='cg'), multilevel.py:479-535).  x: in = initial guess, out =
 * solution.  residuals: HOST array of maxiter+1 doubles or NULL.  *info: 0 converged, -1
 * indefinite operator/preconditioner detected (as the reference), else the iteration count. */
int pamg_solver_pcg(pamg_solver_t S, void *x, const void *b, double tol, int maxiter, int cycle,
                    int cycles_per_level, double *residuals, int *n_iter, int *info, pamg_stream_t s);
/* Flexible GMRES with the resident cycle as preconditioner, all vectors on the DEVICE (reference:
 * pyamg/krylov/_fgmres.py:120-345 as driven by MultilevelSolver.solve(accel='fgmres'),
 * multilevel.py:479-535: Householder reflectors + Givens rotations, the same control flow, stopping
 * rules and residual history).  maxiter / restart <= 0 mean "None".  x: in = initial
 * guess, out = solution.  residuals: HOST array of residuals_cap doubles (may be NULL); *n_res =
 * entries the reference's list would hold.  *info: 0 converged, -1 stagnation, else the number of
 * inner iterations.  Needs (2 * inner iterations + 4) device vectors. */
int pamg_solver_fgmres(pamg_solver_t S, void *x, const void *b, double tol, int maxiter, int restart,
                       int cycle, int cycles_per_level, double *residuals, int residuals_cap, int *n_res,
                       int *n_iter, int *info, pamg_stream_t s);
/* The reference's default GMRES (krylov/_gmres_householder.py, what solve(accel='gmres') and
 * pyamg.solve() on non-symmetric operators run): LEFT-preconditioned, so every norm in `residuals` is a
 * preconditioned-residual norm and the tolerance is relative to ||M b||.  Arguments as pamg_solver_fgmres;
 * needs (inner iterations + 4) device vectors. */
int pamg_solver_gmres(pamg_solver_t S, void *x, const void *b, double tol, int maxiter, int restart,
                      int cycle, int cycles_per_level, double *residuals, int residuals_cap, int *n_res,
                      int *n_iter, int *info, pamg_stream_t s);
/* Same iteration split in three so that callers (benchmarks, device-side Krylov drivers)
 * can run exactly k cycles on the resident state with no staging copies in between:
 * load copies DEVICE x, b into the solver's level-0 buffers; iterate runs k x (cycle +
 * convergence-check residual norm, multilevel.py:558-569) and, if residuals != NULL,
 * writes the k norms ||b - A x|| (HOST, after a stream sync); store copies x back. */
int pamg_solver_load(pamg_solver_t S, const void *x, const void *b, pamg_stream_t s);
int pamg_solver_iterate(pamg_solver_t S, int k, int cycle, int cycles_per_level, double *residuals,
                        pamg_stream_t s);
int pamg_solver_store(pamg_solver_t S, void *x, pamg_stream_t s);
/* stream the solver launches on when the caller passes NULL */
int pamg_solver_stream(pamg_solver_t S, pamg_stream_t *s);
/* use hipGraph replay for the cycle (default 1) */
int pamg_solver_set_graph(pamg_solver_t S, int enable);
/* stats[0]=levels stats[1]=dependency levels of the order-exact schedules stats[2]=HBM bytes resident
 * stats[3]=graphs instantiated stats[4]=times a persistent sweep hit its spin bound (PAMG_E_TIMEOUT: not all of its
 * workgroups were running) and the solver switched to one launch per dependency level -- pamg_solver_solve then runs
 * the solve again from the caller's initial guess and succeeds; pamg_solver_iterate / pcg / (f)gmres report the timeout
 * once (their state is on the device) and are safe from the next call on */
int pamg_solver_stats(pamg_solver_t S, int64_t stats[8]);

/* ------------------------------------------------------------------------------------------------
 * Row-sharded cycle of ONE rank (one process per GPU): MultilevelSolver.__solve (multilevel.py:584-662) and
 * the accel=None loop of .solve (:537-582) on a hierarchy whose fine levels are cut into contiguous row blocks.  The
 * host side (pyamg_amd/dist.py) plans the cut; every operator of a sharded level arrives as a ROW SHARD in local
 * numbering -- columns [owned | halo], the halo grouped by owning rank -- and every level vector is such a buffer.
 * Per operator application one halo exchange of the input vector, overlapped with the row ranges that read owned
 * columns only (second stream).  Smoothers: the row-independent ones (Jacobi, block Jacobi, polynomial); order-exact
 * sweeps do not shard (PAMG_E_UNSUPPORTED).  Below the last sharded level the right-hand side is assembled by an
 * all-reduce of disjoint slices and the remaining cycle runs on every rank with `coarse`.
 * Transport: RCCL (pamg_dist_set_rccl: a communicator of this library's own, built from an id rank 0 obtained with
 * pamg_dist_rccl_unique_id and the host side broadcast), or host callbacks (test rigs: several ranks on one GPU).
 * world = 1 needs none.  Iterates are bit-identical to the single-GPU engine's; the all-reduced norm differs in the
 * last bits. */
#define PAMG_E_COMM         -7   /* RCCL reported an error / a transport callback failed */
typedef struct pamg_dist_s *pamg_dist_t;
/* fill `halo` (DEVICE, halo_count values: the level's halo in plan order) from the peers and hand them the
 * send_count values packed at `send_buf` (DEVICE, plan order); blocking; returns 0 or a status */
typedef int (*pamg_dist_exchange_fn)(void *user, int level, const void *send_buf, int64_t send_count, void *halo,
                                     int64_t halo_count);
/* in-place sum over all ranks of `count` values of dtype (PAMG_F64 / PAMG_F32) at DEVICE address buf; blocking */
typedef int (*pamg_dist_allreduce_fn)(void *user, void *buf, int64_t count, int dtype);
int pamg_dist_create(pamg_dist_t *D, int dtype, int rank, int world);
int pamg_dist_destroy(pamg_dist_t D);
/* sharded levels, fine -> coarse.  A: owned rows x [owned | halo]; P: owned rows x [owned | halo] of the NEXT level;
 * R: owned rows of the next level x [owned | halo] of this one (all borrowed).  Exchange plan in SCALAR units:
 * send_idx[send_off[k] .. send_off[k+1]) = owned-local indices whose values go to send_peer[k]; the halo entries
 * recv_off[k] .. recv_off[k+1] come from recv_peer[k] (HOST arrays, copied). */
int pamg_dist_add_level(pamg_dist_t D, pamg_matrix_t A, pamg_matrix_t P, pamg_matrix_t R, int64_t n_owned, int64_t n_halo,
                        int nsend, const int *send_peer, const int64_t *send_off, const int32_t *send_idx,
                        int nrecv, const int *recv_peer, const int64_t *recv_off);
/* the first replicated level: nc unknowns, of which this rank's R shard produces rows [row0, row0 + n_owned);
 * fill_idx[n_owned + n_halo] (HOST) = global index of every entry of this level's local vector; coarse: the resident
 * solver of the replicated rest of the hierarchy (borrowed) */
int pamg_dist_set_collapse(pamg_dist_t D, pamg_solver_t coarse, int64_t nc, int64_t row0, int64_t n_owned, int64_t n_halo,
                           const int32_t *fill_idx);
/* kind: PAMG_SMOOTH_NONE / JACOBI / POLY / BLOCK_JACOBI; Dinv: HOST, this rank's slice of the inverted diagonal blocks */
int pamg_dist_set_smoother(pamg_dist_t D, int level, int which, int kind, int iterations, double omega, const double *coeffs,
                           int ncoeffs, const void *Dinv, int blocksize);
int pamg_dist_set_callbacks(pamg_dist_t D, pamg_dist_exchange_fn exchange, pamg_dist_allreduce_fn allreduce, void *user);
/* The all-gather form of the halo exchange (the general fallback and the correctness baseline; the default is
 * point to point with the actual neighbours, which moves world x less): every rank contributes its owned part of the level
 * vector padded to count_per_rank values (>= the largest owned part), halo_src[i] (HOST, one per halo value) is the position of
 * halo value i in the gathered vector = owner * count_per_rank + index at the owner.  pamg_dist_set_exchange: 0 = point to
 * point (default), 1 = all-gather (every level that talks must have been given its halo_src); may be switched between
 * iterations.  With RCCL the gather is ONE ncclAllGather per exchange on the comm stream; the host-callback transport of the
 * test rigs forms it with the all-reduce callback (sum of disjoint slices). */
int pamg_dist_set_allgather(pamg_dist_t D, int level, int64_t count_per_rank, const int32_t *halo_src);
int pamg_dist_set_exchange(pamg_dist_t D, int mode);
/* MODEL transport (instead of callbacks / RCCL; before finalize): the rank runs exactly the launches it would run among `world`
 * ranks -- pack, interior ranges, boundary ranges, collapse, replicated tail -- but nothing travels: halos keep what they hold,
 * all-reduces return the rank's own contribution.  What such a run times is the rank's COMPUTE critical path; bench.py adds the
 * wire from the exchange plans (pamg_dist_level_info) with stated xGMI figures -- SURVEY.md 8(e) "availability caveat": no
 * multi-GPU node => N ranks' work on one device + modelled xGMI time.  The iterates of such a run are NOT a solve.
 * pamg_dist_level_info(level; the collapse level = number of sharded levels): {owned values, halo values, exchanges of this
 * level's vectors per iteration (counted while the last iteration was enqueued), peers sent to, peers received from, most
 * values sent to one peer, most values received from one peer, values sent per exchange}. */
int pamg_dist_set_model_transport(pamg_dist_t D);
int pamg_dist_level_info(pamg_dist_t D, int level, int64_t info[8]);
/* One-rank exercise of every RCCL entry point the sharded cycle uses, on the current device, through the table this library
 * binds at run time (ncclGetUniqueId, ncclCommInitRank, grouped ncclSend + ncclRecv to itself on a comm stream ordered against
 * a main stream by events exactly like the halo exchange, ncclAllGather, a one-element ncclAllReduce, ncclCommDestroy) with
 * n float64 values; *max_err = largest |received - sent|.  PAMG_E_UNSUPPORTED: no librccl to bind. */
int pamg_rccl_selftest(int64_t n, double *max_err);
int pamg_rccl_available(void);                             /* PAMG_OK when a librccl could be bound (no communicator is created) */
int pamg_dist_rccl_unique_id(void *id128);                 /* 128 bytes; PAMG_E_UNSUPPORTED: no librccl to bind */
int pamg_dist_set_rccl(pamg_dist_t D, const void *id128);  /* collective: ncclCommInitRank(world, id, rank) */
int pamg_dist_finalize(pamg_dist_t D);
/* use_graph / overlap: 0 or 1, -1 = leave (defaults 1 / 1; RCCL work is captured only with PAMG_DIST_GRAPH=1) */
int pamg_dist_set_options(pamg_dist_t D, int use_graph, int overlap);
int pamg_dist_load(pamg_dist_t D, const void *x_owned, const void *b_owned);     /* DEVICE slices, n_owned values */
int pamg_dist_store(pamg_dist_t D, void *x_owned);
/* k x (V-cycle + all-reduced ||b - A x||): residuals = HOST array of k norms (synchronises), or NULL: k cycles
 * without norms, queued only (pamg_dist_sync waits) */
int pamg_dist_iterate(pamg_dist_t D, int k, double *residuals);
int pamg_dist_resid_norm(pamg_dist_t D, double *norm);
int pamg_dist_sync(pamg_dist_t D);
int pamg_dist_stream(pamg_dist_t D, pamg_stream_t *s);
/* [0] sharded levels [1] transport (0 none, 1 callbacks, 2 RCCL) [2] halo exchanges per iteration [3] of those,
 * overlapped with interior rows [4] iteration replayed from a hipGraph [5] bytes of vectors [6] values sent per
 * exchange round [7] interior row ranges of the fine-level shard */
int pamg_dist_info(pamg_dist_t D, int64_t info[8]);
/* Transport self-test (after pamg_dist_finalize): level `level`'s owned values are set from the HOST array x_owned, one halo
 * exchange runs through the cycle's own code path, the received halo values come back in the HOST array halo_out. */
int pamg_dist_exchange_test(pamg_dist_t D, int level, const void *x_owned, void *halo_out);

/* ------------------------------------------------------------------------------------------------
 * Setup-phase operators: what the reference's smoothed-aggregation setup spends its
 * time in -- approximate_spectral_radius (util/linalg.py:255-370), the prolongation smoother
 * P = T - (omega/rho) D^-1 A T (aggregation/smooth.py:61-207) and the Galerkin product
 * A_c = R @ A @ P (aggregation/aggregation.py:425) -- on device-resident CSR operands (fp64, int32).
 *
 * pamg_csr_t: a plain CSR matrix in HBM.  pamg_csr_matmat is SciPy's csr_matmat: its arithmetic order
 * (for k in row i of A in stored order, for j in row k of B: sums[j] += a_ik * b_kj), its emission order
 * (a row's entries in reverse order of first touch) and its dropping of sums that are exactly zero -- the
 * result is the array `A @ B` produces, entry for entry (the next level's aggregation walks that order).
 * keep_zeros = 1 (with col_block = width of the result's column blocks): the scalar view of a BSR product
 * with true blocks -- bsr_matmat stores whole blocks, zeros included, in FORWARD order of first touch.  pamg_csr_subtract is csr_binop_csr with minus: the canonical
 * merge when both operands have sorted duplicate-free rows, SciPy's general algorithm (and its emission
 * order) otherwise.  PAMG_E_UNSUPPORTED: a row of B longer than 2048 entries meets a row of the product with
 * more than 4096 terms. */
typedef struct pamg_csr_s *pamg_csr_t;
int pamg_csr_create(pamg_csr_t *out, int64_t nrows, int64_t ncols, const int32_t *Ap, const int32_t *Aj,
                    const double *Ax);                       /* HOST arrays */
/* non-owning scalar-CSR view of an fp64 operator (BSR operators: their flattened scalar view, whose row
 * order is SciPy's bsr_matmat accumulation order); A must outlive the view */
int pamg_csr_view(pamg_csr_t *out, pamg_matrix_t A);
int pamg_csr_destroy(pamg_csr_t A);
int pamg_csr_info(pamg_csr_t A, int64_t info[4]);            /* rows, columns, stored entries, owns */
int pamg_csr_download(pamg_csr_t A, int32_t *Ap, int32_t *Aj, double *Ax);   /* HOST arrays */
int pamg_csr_matmat(pamg_csr_t A, pamg_csr_t B, int col_block, int keep_zeros, pamg_csr_t *C);
int pamg_csr_subtract(pamg_csr_t A, pamg_csr_t B, pamg_csr_t *C);
/* A - B for the scalar views of two BSR matrices with R x C blocks: SciPy's bsr_binop_bsr (sparsetools/bsr.h) -- block
 * columns merged per block row, a result block kept when any of its entries is non-zero; ascending block columns when
 * both operands have sorted duplicate-free block rows, reverse order of first touch (A's blocks, then B's) otherwise.
 * PAMG_E_ARG: an operand is not the scalar view of R x C blocks. */
int pamg_csr_subtract_bsr(pamg_csr_t A, pamg_csr_t B, int R, int C, pamg_csr_t *out);
/* strength.py:248-348 for a CSR operator: amg_core::symmetric_strength_of_connection (smoothed_aggregation.h:56-110:
 * |a_ij|^2 >= theta^2 |a_ii| |a_jj|, the diagonal always kept, stored order kept), then magnitudes, every row scaled by
 * the reciprocal of its largest entry */
int pamg_csr_strength_symmetric(pamg_csr_t A, double theta, pamg_csr_t *S);
int pamg_csr_scale(pamg_csr_t A, double alpha);
/* pamg_standard_aggregation on a device-resident pattern (x, y: HOST arrays of A's row count) */
int pamg_csr_standard_aggregation(pamg_csr_t C, int32_t *x, int32_t *y, int32_t *naggs);             /* a_ij <- a_ij * alpha, in place (owning matrices only) */
/* util/utils.py scale_rows (a_ij <- a_ij * d_i, d: HOST, one per row) and `alpha * A` (a_ij <- a_ij * alpha)
 * on a resident fp64 operator, in place (block operators: the scalar view SpMV / Arnoldi / the sparse products read;
 * their block arrays are released -- no block smoothers on it afterwards).  PAMG_E_STATE once a sweep schedule or a
 * solver holds it. */
int pamg_matrix_scale_rows(pamg_matrix_t A, const double *d);
int pamg_matrix_scale_values(pamg_matrix_t A, double alpha);
/* Arnoldi process of util/linalg.py:154-253 (the non-symmetric branch, the only one
 * approximate_spectral_radius uses) on a resident operator: modified Gram-Schmidt, basis in HBM.
 * run: start vector from the HOST (v0_im NULL: real) or, with v0_re NULL, the vector the last combine
 * left on the device; H: HOST, (maxiter+1) x maxiter entries as (re, im) pairs, row-major; *ncols = valid
 * columns (the reference's j + 1); *breakdown_flag as in the reference.  combine: next start vector =
 * V[:, :ncols] @ coef (linalg.py:352), coef_im NULL for a real eigenvector.  vector: download it. */
typedef struct pamg_arnoldi_s *pamg_arnoldi_t;
int pamg_arnoldi_create(pamg_arnoldi_t *out, pamg_matrix_t A, int maxiter);
int pamg_arnoldi_destroy(pamg_arnoldi_t h);
int pamg_arnoldi_run(pamg_arnoldi_t h, const double *v0_re, const double *v0_im, double breakdown, double *H,
                     int *ncols, int *breakdown_flag);
int pamg_arnoldi_combine(pamg_arnoldi_t h, int ncols, const double *coef_re, const double *coef_im);
int pamg_arnoldi_vector(pamg_arnoldi_t h, double *re, double *im, int *planes);

#ifdef __cplusplus
}
#endif
#endif /* PYAMG_AMD_H */
