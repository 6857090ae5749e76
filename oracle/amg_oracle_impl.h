/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called
 * from the product path (pyamg_amd/).  See oracle/README.md.
 *
 * Type-generic body; included twice by amg_oracle.c with
 *     #define REAL   double|float
 *     #define FN(x)  orc_##x##_f64 | orc_##x##_f32
 *
 * Every function is a plain-C restatement of the *semantics* of one function on
 * the reference's solve path; the reference file:line each one
 * follows is cited above it.  Arithmetic is kept in exactly the reference's
 * order (sequential per-row accumulation, no FMA contraction: build with
 * -ffp-contract=off) so that results are bit-identical to the reference on x86-64.
 * Index type is 32-bit int only (reference: instantiate.yml:2-6).
 */

/* y += A x, CSR.  Follows SciPy sparsetools csr_matvec (third-party, not under
 * /root/reference; pinned scipy>=1.11, pyproject.toml:52; installed 1.15.3):
 * the running sum of row i starts from y[i] and adds a_ij*x_j in storage order.
 * Reference call sites: multilevel.py:545,567,612,614,660; relaxation.py:652,657. */
void FN(csr_matvec)(int n_row, const int *Ap, const int *Aj, const REAL *Ax,
                    const REAL *x, REAL *y)
{
    for (int r = 0; r < n_row; ++r) {
        REAL acc = y[r];
        const int hi = Ap[r + 1];
        for (int p = Ap[r]; p < hi; ++p)
            acc += Ax[p] * x[Aj[p]];
        y[r] = acc;
    }
}

/* y += A x, BSR with R x C row-major blocks.  Follows SciPy sparsetools bsr_matvec
 * (+ dense.h gemv): for R==C==1 it IS the CSR loop; otherwise block row i walks its
 * blocks in storage order and, per block, each of the R outputs continues its own
 * running sum over the C block columns. */
void FN(bsr_matvec)(int n_brow, int R, int C, const int *Ap, const int *Aj,
                    const REAL *Ax, const REAL *x, REAL *y)
{
    if (R == 1 && C == 1) { FN(csr_matvec)(n_brow, Ap, Aj, Ax, x, y); return; }
    const long RC = (long)R * C;
    for (int ib = 0; ib < n_brow; ++ib) {
        REAL *yo = y + (long)R * ib;
        for (int p = Ap[ib]; p < Ap[ib + 1]; ++p) {
            const REAL *blk = Ax + RC * p;
            const REAL *xi = x + (long)C * Aj[p];
            for (int r = 0; r < R; ++r) {
                REAL acc = yo[r];
                for (int c = 0; c < C; ++c)
                    acc += blk[(long)r * C + c] * xi[c];
                yo[r] = acc;
            }
        }
    }
}

/* One Gauss-Seidel sweep over rows row_start, row_start+row_step, ... (!= row_stop),
 * in place.  Follows amg_core gauss_seidel, relaxation.h:48-76: off-diagonal products
 * summed in storage order, the LAST stored entry with j==i is the diagonal, a zero or
 * missing diagonal leaves x[i] untouched. */
void FN(gauss_seidel)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x,
                      const REAL *b, int row_start, int row_stop, int row_step)
{
    for (int i = row_start; i != row_stop; i += row_step) {
        REAL offsum = 0, d = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) d = Ax[p];
            else        offsum += Ax[p] * x[j];
        }
        if (d != (REAL)0)
            x[i] = (b[i] - offsum) / d;
    }
}

/* SOR sweep.  Follows amg_core sor_gauss_seidel, relaxation.h:116-145:
 * x_i <- omega*((b_i - offsum)/d) + (1-omega)*x_i, omega a REAL scalar. */
void FN(sor_gauss_seidel)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x,
                          const REAL *b, int row_start, int row_stop, int row_step,
                          REAL omega)
{
    for (int i = row_start; i != row_stop; i += row_step) {
        REAL offsum = 0, d = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) d = Ax[p];
            else        offsum += Ax[p] * x[j];
        }
        if (d != (REAL)0)
            x[i] = omega * ((b[i] - offsum) / d) + (1 - omega) * x[i];
    }
}

/* Weighted Jacobi.  Follows amg_core jacobi, relaxation.h:309-346: snapshot the swept
 * entries of x into temp, then x_i <- (1-w) t_i + w*((b_i - offsum)/d) from temp. */
void FN(jacobi)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *b,
                REAL *temp, int row_start, int row_stop, int row_step, REAL omega)
{
    const REAL one = 1;
    for (int i = row_start; i != row_stop; i += row_step) temp[i] = x[i];
    for (int i = row_start; i != row_stop; i += row_step) {
        REAL offsum = 0, d = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) d = Ax[p];
            else        offsum += Ax[p] * temp[j];
        }
        if (d != (REAL)0)
            x[i] = (one - omega) * temp[i] + omega * ((b[i] - offsum) / d);
    }
}

/* Multiplicative overlapping Schwarz.  Follows amg_core overlapping_schwarz_csr, relaxation.h:1420-1492: per subdomain
 * the residual of its rows from the current x (rsum -= a*x entry by entry, += b last), times the dense inverse of its
 * block (row-major, k ascending), added to x.  work: 2 * (largest subdomain) values. */
void FN(overlapping_schwarz_csr)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *b, const REAL *Tx,
                                 const int *Tp, const int *Sj, const int *Sp, int row_start, int row_stop, int row_step,
                                 REAL *work, int maxsize)
{
    REAL *rsum = work, *dr = work + maxsize;
    for (int d = row_start; d != row_stop; d += row_step) {
        const int size = Sp[d + 1] - Sp[d];
        for (int q = 0; q < size; ++q) {
            const int row = Sj[Sp[d] + q];
            REAL r = 0;
            for (int p = Ap[row]; p < Ap[row + 1]; ++p) r -= Ax[p] * x[Aj[p]];
            r += b[row];
            rsum[q] = r;
        }
        for (int i = 0; i < size; ++i) {
            REAL s = 0;
            for (int k = 0; k < size; ++k) s += Tx[Tp[d] + (long)i * size + k] * rsum[k];
            dr[i] = s;
        }
        for (int q = 0; q < size; ++q) x[Sj[Sp[d] + q]] += dr[q];
    }
}

/* Gauss-Seidel on a list of rows, in list order.  Follows amg_core gauss_seidel_indexed, relaxation.h:736-790. */
void FN(gauss_seidel_indexed)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *b, const int *Id,
                              int row_start, int row_stop, int row_step)
{
    for (int q = row_start; q != row_stop; q += row_step) {
        const int i = Id[q];
        REAL rsum = 0, diag = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) diag = Ax[p];
            else rsum += Ax[p] * x[j];
        }
        if (diag != (REAL)0) x[i] = (b[i] - rsum) / diag;
    }
}

/* Weighted Jacobi on a list of rows.  Follows amg_core jacobi_indexed, relaxation.h:382-427:
 * temp = x (the whole vector), then every listed row is relaxed from temp; a zero / missing
 * diagonal leaves the row untouched (the reference also prints a warning). */
void FN(jacobi_indexed)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, int n, const REAL *b,
                        const int *indices, int n_indices, REAL *temp, REAL omega)
{
    const REAL one = 1;
    for (int i = 0; i < n; ++i) temp[i] = x[i];
    for (int k = 0; k < n_indices; ++k) {
        const int row = indices[k];
        REAL rsum = 0, diag = 0;
        for (int p = Ap[row]; p < Ap[row + 1]; ++p) {
            const int col = Aj[p];
            if (row == col) diag = Ax[p];
            else            rsum += Ax[p] * temp[col];
        }
        if (diag != (REAL)0)
            x[row] = (one - omega) * temp[row] + omega * ((b[row] - rsum) / diag);
    }
}

/* Kaczmarz (Gauss-Seidel on A A^H y = b, x = A^H y).  Follows amg_core gauss_seidel_ne,
 * relaxation.h:875-904: row by row, delta = (b_i - <a_i, x>) * Dinv_i * omega, x += a_i^H delta. */
void FN(gauss_seidel_ne)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *b,
                         int row_start, int row_stop, int row_step, const REAL *Dinv, REAL omega)
{
    for (int i = row_start; i != row_stop; i += row_step) {
        REAL delta = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) delta += Ax[p] * x[Aj[p]];
        delta = (b[i] - delta) * Dinv[i] * omega;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) x[Aj[p]] += Ax[p] * delta;
    }
}

/* Gauss-Seidel on A^H A x = A^H b.  Follows amg_core gauss_seidel_nr, relaxation.h:939-975: the
 * arrays are the CSC form of A (column by column); z is the running residual b - A x. */
void FN(gauss_seidel_nr)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, REAL *z,
                         int col_start, int col_stop, int col_step, const REAL *Dinv, REAL omega)
{
    for (int i = col_start; i != col_stop; i += col_step) {
        REAL delta = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) delta += Ax[p] * z[Aj[p]];
        delta *= (Dinv[i] * omega);
        x[i] += delta;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) z[Aj[p]] -= delta * Ax[p];
    }
}

/* Jacobi on the normal equations.  Follows amg_core jacobi_ne, relaxation.h:811-840: delta is the
 * row-scaled residual handed in by the wrapper; temp accumulates omega * a_ij * delta_i row after row. */
void FN(jacobi_ne)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *delta, REAL *temp,
                   int row_start, int row_stop, int row_step, REAL omega)
{
    for (int i = row_start; i < row_stop; i += row_step) temp[i] = 0;
    for (int i = row_start; i < row_stop; i += row_step)
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) temp[Aj[p]] += omega * Ax[p] * delta[i];
    for (int i = row_start; i < row_stop; i += row_step) x[i] += temp[i];
}

/* dense helper: out[0..bs) = M(bs x bs, row major) * v, each output a fresh running sum
 * starting at 0 (the reference's gemm(...,'F','F','T' overwrite), linalg.h:405-438). */
static void FN(blk_apply)(const REAL *M, const REAL *v, REAL *out, int bs)
{
    for (int r = 0; r < bs; ++r) {
        REAL acc = 0;
        for (int c = 0; c < bs; ++c) acc += M[r * bs + c] * v[c];
        out[r] = acc;
    }
}

/* shared tail of the two BSR *point* smoothers: point sweep inside the diagonal block D
 * in the direction of the outer sweep; src supplies the coupled values (x itself for GS,
 * the snapshot for Jacobi).  relaxation.h:244-259 / 538-556. */
static void FN(bsr_diag_point)(const REAL *D, REAL *res, const REAL *src, REAL *xi,
                               int bs, int dirn, int jac, REAL omega)
{
    const int k0 = dirn > 0 ? 0 : bs - 1, k1 = dirn > 0 ? bs : -1;
    for (int k = k0; k != k1; k += dirn) {
        REAL d = 1;
        for (int kk = k0; kk != k1; kk += dirn) {
            if (kk == k) d = D[k * bs + kk];
            else         res[k] -= D[k * bs + kk] * src[kk];
        }
        if (d != (REAL)0) {
            if (jac) xi[k] = ((REAL)1 - omega) * src[k] + omega * res[k] / d;
            else     xi[k] = res[k] / d;
        }
    }
}

/* Point Gauss-Seidel on a BSR matrix with square bs x bs blocks.  Follows amg_core
 * bsr_gauss_seidel, relaxation.h:185-266: res starts at b_I, every off-diagonal block
 * subtracts its (fresh-summed) block product, then a point sweep inside the diagonal
 * block; a block row without a stored diagonal block is skipped. */
void FN(bsr_gauss_seidel)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x,
                          const REAL *b, int row_start, int row_stop, int row_step, int bs)
{
    REAL res[64], prod[64];
    const int bb = bs * bs, dirn = row_step < 0 ? -1 : 1;
    for (int i = row_start; i != row_stop; i += row_step) {
        long dpos = -1;
        for (int k = 0; k < bs; ++k) res[k] = b[(long)i * bs + k];
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) { dpos = (long)p * bb; continue; }
            FN(blk_apply)(Ax + (long)p * bb, x + (long)j * bs, prod, bs);
            for (int k = 0; k < bs; ++k) res[k] -= prod[k];
        }
        if (dpos >= 0)
            FN(bsr_diag_point)(Ax + dpos, res, x + (long)i * bs, x + (long)i * bs,
                               bs, dirn, 0, (REAL)0);
    }
}

/* Point Jacobi on a BSR matrix.  Follows amg_core bsr_jacobi, relaxation.h:472-562
 * (snapshot of the first |row_stop-row_start|*bs entries, relaxation.h:505-508; the
 * Python wrapper always sweeps every block row forward, relaxation.py:396-420). */
void FN(bsr_jacobi)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *b,
                    REAL *temp, int row_start, int row_stop, int row_step, int bs,
                    REAL omega)
{
    REAL res[64], prod[64];
    const int bb = bs * bs, dirn = row_step < 0 ? -1 : 1;
    long ncopy = (long)(row_stop > row_start ? row_stop - row_start : row_start - row_stop) * bs;
    for (long q = 0; q < ncopy; ++q) temp[q] = x[q];
    for (int i = row_start; i != row_stop; i += row_step) {
        long dpos = -1;
        for (int k = 0; k < bs; ++k) res[k] = b[(long)i * bs + k];
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) { dpos = (long)p * bb; continue; }
            FN(blk_apply)(Ax + (long)p * bb, temp + (long)j * bs, prod, bs);
            for (int k = 0; k < bs; ++k) res[k] -= prod[k];
        }
        if (dpos >= 0)
            FN(bsr_diag_point)(Ax + dpos, res, temp + (long)i * bs, x + (long)i * bs,
                               bs, dirn, 1, omega);
    }
}

/* True block Jacobi with precomputed inverse diagonal blocks Dinv (n_brow,bs,bs).
 * Follows amg_core block_jacobi, relaxation.h:1021-1090. */
void FN(block_jacobi)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, const REAL *b,
                      const REAL *Dinv, REAL *temp, int row_start, int row_stop,
                      int row_step, REAL omega, int bs)
{
    REAL acc[64], v[64];
    const int bb = bs * bs;
    const REAL one = 1;
    for (int i = row_start; i != row_stop; i += row_step)
        for (int k = 0; k < bs; ++k) temp[(long)i * bs + k] = x[(long)i * bs + k];
    for (int i = row_start; i != row_stop; i += row_step) {
        for (int k = 0; k < bs; ++k) acc[k] = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) continue;
            FN(blk_apply)(Ax + (long)p * bb, temp + (long)j * bs, v, bs);
            for (int k = 0; k < bs; ++k) acc[k] += v[k];
        }
        for (int k = 0; k < bs; ++k) acc[k] = b[(long)i * bs + k] - acc[k];
        FN(blk_apply)(Dinv + (long)i * bb, acc, v, bs);
        for (int k = 0; k < bs; ++k)
            x[(long)i * bs + k] = (one - omega) * temp[(long)i * bs + k] + omega * v[k];
    }
}

/* Block Jacobi on a list of block rows.  Follows amg_core block_jacobi_indexed, relaxation.h:1129-1199: the whole x is
 * snapshotted into temp, the listed block rows are updated from the snapshot. */
void FN(block_jacobi_indexed)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x, int n, const REAL *b,
                              const REAL *Dinv, const int *indices, int nidx, REAL *temp, REAL omega, int bs)
{
    REAL acc[64], v[64];
    const int bb = bs * bs;
    const REAL one = 1;
    for (int i = 0; i < n; ++i) temp[i] = x[i];
    for (int q = 0; q < nidx; ++q) {
        const int i = indices[q];
        for (int k = 0; k < bs; ++k) acc[k] = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) continue;
            FN(blk_apply)(Ax + (long)p * bb, temp + (long)j * bs, v, bs);
            for (int k = 0; k < bs; ++k) acc[k] += v[k];
        }
        for (int k = 0; k < bs; ++k) acc[k] = b[(long)i * bs + k] - acc[k];
        FN(blk_apply)(Dinv + (long)i * bb, acc, v, bs);
        for (int k = 0; k < bs; ++k)
            x[(long)i * bs + k] = (one - omega) * temp[(long)i * bs + k] + omega * v[k];
    }
}

/* True block Gauss-Seidel.  Follows amg_core block_gauss_seidel, relaxation.h:1242-1298. */
void FN(block_gauss_seidel)(const int *Ap, const int *Aj, const REAL *Ax, REAL *x,
                            const REAL *b, const REAL *Dinv, int row_start, int row_stop,
                            int row_step, int bs)
{
    REAL acc[64], v[64];
    const int bb = bs * bs;
    for (int i = row_start; i != row_stop; i += row_step) {
        for (int k = 0; k < bs; ++k) acc[k] = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) continue;
            FN(blk_apply)(Ax + (long)p * bb, x + (long)j * bs, v, bs);
            for (int k = 0; k < bs; ++k) acc[k] += v[k];
        }
        for (int k = 0; k < bs; ++k) acc[k] = b[(long)i * bs + k] - acc[k];
        FN(blk_apply)(Dinv + (long)i * bb, acc, x + (long)i * bs, bs);
    }
}


/* ---------------------------------------------------------------------------------------------
 * Pseudo-inverse of every n x n block of AA[m][n][n], in place.  Follows amg_core pinv_array,
 * linalg.h:930-1000: per block a one-sided Jacobi SVD A = U S V^T (svd_jacobi, linalg.h:546-812,
 * column-major factors), the reciprocals of the non-zero singular values, W = S^-1 U^T, block <- V W
 * accumulated from zero over the inner index (gemm, linalg.h:437-458).  TransA = 'T': the blocks are
 * row-major (how util/utils.py:684 calls it for get_block_diag).  n <= 8 here (the reference uses this
 * routine for n < 7 only, utils.py:682).
 * Mixed precision of the reference kept: its literals 1.0 / 2.0 / 50.0 are doubles, so with REAL = float the
 * rotation tangent and cosine are evaluated in double and rounded once. */
static REAL FN(pinv_colnorm)(const REAL *U, int n, int a)
{
    REAL s = 0;
    for (int i = 0; i < n; ++i) s += U[a * n + i] * U[a * n + i];
    return RSQRT(s);
}

static void FN(pinv_svd)(const REAL *A, REAL *U, REAL *V, REAL *S, int n)
{
    const int nn = n * n;
    if (n == 1) {                                          /* linalg.h:559-571 */
        const REAL na = RFABS(A[0]);
        V[0] = 1; S[0] = na;
        U[0] = (na == 0) ? (REAL)1 : A[0] / na;
        return;
    }
    const REAL eps = REPS;
    const int sweepmax = 15 * n > 30 ? 15 * n : 30;
    const REAL tol = RSQRT((REAL)n) * eps;
    int count = 1, sweep = 0;
    for (int i = 0; i < nn; ++i) { V[i] = 0; U[i] = A[i]; }
    for (int i = 0; i < nn; i += n + 1) V[i] = 1;
    for (int j = 0; j < n; ++j) S[j] = eps * FN(pinv_colnorm)(U, n, j);
    while (count > 0 && sweep <= sweepmax) {
        count = n * (n - 1) / 2;
        for (int j = 0; j + 1 < n; ++j)
            for (int k = j + 1; k < n; ++k) {
                const REAL a = FN(pinv_colnorm)(U, n, j), b = FN(pinv_colnorm)(U, n, k);
                REAL d = 0;
                for (int i = 0; i < n; ++i) d += U[j * n + i] * U[k * n + i];
                const REAL nd = RFABS(d), ea = S[j], eb = S[k];
                const int sorted = a >= b, orthog = nd <= tol * a * b;
                if (sorted && (orthog || a < ea || b < eb)) { --count; continue; }
                if (!sorted || (nd == 0 && a == b)) {      /* :651-686 swap with one sign flip */
                    S[j] = eb; S[k] = ea;
                    for (int i = 0; i < n; ++i) { const REAL uj = U[j * n + i], uk = U[k * n + i]; U[j * n + i] = -uk; U[k * n + i] = uj; }
                    for (int i = 0; i < n; ++i) { const REAL vj = V[j * n + i], vk = V[k * n + i]; V[j * n + i] = -vk; V[k * n + i] = vj; }
                    continue;
                }
                const REAL tau = (REAL)((b * b - a * a) / (2.0 * nd));                      /* :694-699 */
                const REAL sg = tau < 0 ? (REAL)-1 : (REAL)1;
                const REAL t = (REAL)(sg / (RFABS(tau) + sqrt(1.0 + tau * tau)));
                const REAL c = (REAL)(1.0 / sqrt(1.0 + t * t));
                const REAL s = d * (t * c / nd), ms = -s, ns = RFABS(s);
                S[j] = RFABS(c) * ea + ns * eb;
                S[k] = ns * ea + RFABS(c) * eb;
                for (int i = 0; i < n; ++i) { const REAL uj = U[j * n + i], uk = U[k * n + i]; U[j * n + i] = uj * c + ms * uk; U[k * n + i] = s * uj + uk * c; }
                for (int i = 0; i < n; ++i) { const REAL vj = V[j * n + i], vk = V[k * n + i]; V[j * n + i] = vj * c + ms * vk; V[k * n + i] = s * vj + vk * c; }
            }
        ++sweep;
    }
    REAL sigma_tol = 0;
    int iszero = n;
    for (int j = 0; j < n; ++j) {                          /* :745-790 */
        const REAL cn = FN(pinv_colnorm)(U, n, j);
        if (j == 0) { const REAL alpha = (REAL)(50.0 / RSQRT(RSQRT(eps))); sigma_tol = alpha * cn * eps; }
        if (cn <= sigma_tol) { --iszero; S[j] = 0; for (int i = 0; i < n; ++i) U[j * n + i] = 0; }
        else { S[j] = cn; for (int i = 0; i < n; ++i) U[j * n + i] = U[j * n + i] / cn; }
    }
    if (iszero == 0) {                                     /* :792-805 */
        for (int i = 0; i < nn; ++i) V[i] = 0;
        for (int i = 0; i < nn; i += n + 1) { V[i] = 1; U[i] = 1; }
    }
}

void FN(pinv_array)(REAL *AA, int m, int n, char TransA)
{
    REAL in[64], U[64], V[64], W[64], S[8];
    if (n < 1 || n > 8) return;
    for (long blk = 0; blk < m; ++blk) {
        REAL *a = AA + blk * n * n;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) in[TransA == 'T' ? c * n + r : r * n + c] = a[r * n + c];
        FN(pinv_svd)(in, U, V, S, n);
        for (int j = 0; j < n; ++j) if (S[j] != 0) S[j] = (REAL)(1.0 / S[j]);
        for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) W[j * n + k] = U[k * n + j] * S[k];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                REAL acc = 0;
                for (int k = 0; k < n; ++k) acc += V[k * n + i] * W[j * n + k];
                a[i * n + j] = acc;
            }
    }
}


/* ---------------------------------------------------------------------------------------------
 * Tentative prolongator blocks.  Follows amg_core fit_candidates_common (real), smoothed_aggregation.h:484-610:
 * Ap / Ai list the nodes of every aggregate (CSC of AggOp); the K1 x K2 blocks of B belonging to aggregate j are
 * copied into Ax in list order, giving a (nodes * K1) x K2 matrix whose columns are orthonormalised left to right by
 * modified Gram-Schmidt -- every norm and inner product a plain running sum down the rows -- R[j] (K2 x K2) taking
 * the coefficients; a column whose norm after orthogonalisation is not above tol * (its norm before) becomes zero. */
void FN(fit_candidates)(int n_col, int K1, int K2, const int *Ap, const int *Ai, REAL *Ax, const REAL *B, REAL *R, REAL tol)
{
    const int bs = K1 * K2;
    for (long k = 0; k < (long)n_col * K2 * K2; ++k) R[k] = 0;
    for (int j = 0; j < n_col; ++j)
        for (int ii = Ap[j]; ii < Ap[j + 1]; ++ii)
            for (int e = 0; e < bs; ++e) Ax[(long)bs * ii + e] = B[(long)bs * Ai[ii] + e];
    for (int j = 0; j < n_col; ++j) {
        REAL *lo = Ax + (long)bs * Ap[j], *hi = Ax + (long)bs * Ap[j + 1], *Rj = R + (long)j * K2 * K2;
        for (int c = 0; c < K2; ++c) {
            REAL before = 0;
            for (REAL *p = lo + c; p < hi; p += K2) before += (*p) * (*p);
            before = RSQRT(before);
            for (int q = 0; q < c; ++q) {
                REAL ip = 0;
                for (REAL *a = lo + q, *b = lo + c; a < hi; a += K2, b += K2) ip += (*a) * (*b);
                for (REAL *a = lo + q, *b = lo + c; a < hi; a += K2, b += K2) *b -= ip * (*a);
                Rj[K2 * q + c] = ip;
            }
            REAL after = 0;
            for (REAL *p = lo + c; p < hi; p += K2) after += (*p) * (*p);
            after = RSQRT(after);
            REAL scale = 0;
            if (after > tol * before) { scale = (REAL)(1.0 / after); Rj[K2 * c + c] = after; }
            else Rj[K2 * c + c] = 0;
            for (REAL *p = lo + c; p < hi; p += K2) *p *= scale;
        }
    }
}
