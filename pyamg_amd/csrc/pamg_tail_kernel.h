// pamg_tail_kernel.h -- cycle_tail_kernel: the small levels of a hierarchy in one launch.  A header of templates and
// inline functions only, so both pamg_matrix.hip (through pamg_kernels.h) and pamg_solver.hip may include it.
#pragma once
#include "pamg_common.h"

namespace pamg {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ------------------------------------------------------------------ the tail of the hierarchy in ONE launch
// Below a few thousand unknowns a level's five kernels (smooth, residual, restrict, prolong, smooth) cost a launch each
// and compute for a microsecond: a third of a 2000^2 Jacobi V-cycle was launch latency.  cycle_tail_kernel runs the whole
// sub-cycle of those levels -- V-cycle, Jacobi / polynomial smoothers, dense coarse solve -- as ONE workgroup walking a
// list of operations recorded at setup, a workgroup barrier between them.  Every operation is the arithmetic of the
// kernel it replaces: a row's products summed in storage order by one lane (csr_stream_kernel's row phase), the same
// epilogue expressions, the dense coarse solve with the same lane-strided partial sums and butterfly.
enum : int { TOP_SPMV = 0, TOP_GEMV, TOP_COPY, TOP_ZERO, TOP_SCALE, TOP_AXPY };

struct TailOp {
    int kind, epi, n, pad;
    const int *Ap, *Aj;
    const void *Ax, *diag;
    const void *x, *b;       // x: gather source / input vector; b: right-hand side or v of the AXPBY forms
    void *y, *z;             // y: destination; z: second vector cleared by TOP_SPMV/EPI_SET (or nullptr)
    double c, omega;
};

constexpr int TAIL_THREADS = 1024;

template <typename T>
__global__ __launch_bounds__(TAIL_THREADS) void cycle_tail_kernel(const TailOp *ops, int nops)
{
    const int tid = (int)threadIdx.x;
    for (int k = 0; k < nops; ++k) {
        const TailOp op = ops[k];
        const T *x = (const T *)op.x, *b = (const T *)op.b;
        T *y = (T *)op.y;
        if (op.kind == TOP_SPMV) {
            const T *Ax = (const T *)op.Ax, *dg = (const T *)op.diag;
            const T cc = (T)op.c, om = (T)op.omega, one = T(1);
            for (int r = tid; r < op.n; r += TAIL_THREADS) {
                const int lo = op.Ap[r], hi = op.Ap[r + 1];
                const int e = op.epi;
                const bool jac = e == EPI_JACOBI || e == EPI_JACOBI_B;
                T s = e == EPI_JACOBI_B ? b[r] : T(0);
                for (int p = lo; p < hi; ++p) {
                    const int j = op.Aj[p];
                    if (jac && j == r) continue;               // the diagonal never enters the sum
                    const T pr = Ax[p] * x[j];
                    if (e == EPI_JACOBI_B) s -= pr; else s += pr;
                }
                if (e == EPI_SET) { y[r] = s; if (op.z) ((T *)op.z)[r] = T(0); }
                else if (e == EPI_ACC) y[r] = y[r] + s;
                else if (e == EPI_RESID) y[r] = b[r] - s;
                else if (e == EPI_AXPBY) { const T t = cc * b[r]; y[r] = t + s; }
                else if (e == EPI_ACC_AXPBY) { const T t = cc * b[r]; const T h = t + s; y[r] = y[r] + h; }
                else if (e == EPI_JACOBI) { const T d = dg[r], xo = x[r]; y[r] = (d != T(0)) ? (one - om) * xo + om * ((b[r] - s) / d) : xo; }
                else if (e == EPI_JACOBI_B) { const T d = dg[r], xo = x[r]; y[r] = (d != T(0)) ? (one - om) * xo + om * s / d : xo; }
            }
        } else if (op.kind == TOP_GEMV) {                         // dense_gemv_kernel: one wave per row
            const T *M = (const T *)op.Ax;
            const int lane = tid & 63;
            for (int row = tid >> 6; row < op.n; row += TAIL_THREADS / 64) {
                double acc = 0.0;
                for (int q = lane; q < op.n; q += 64) acc += (double)M[(size_t)row * op.n + q] * (double)b[q];
                acc = wave_sum(acc);
                if (lane == 0) y[row] = (T)acc;
            }
        } else if (op.kind == TOP_COPY) {
            for (int i = tid; i < op.n; i += TAIL_THREADS) y[i] = x[i];
        } else if (op.kind == TOP_ZERO) {
            for (int i = tid; i < op.n; i += TAIL_THREADS) y[i] = T(0);
        } else if (op.kind == TOP_SCALE) {                        // vec_scale_kernel
            const T a = (T)op.c;
            for (int i = tid; i < op.n; i += TAIL_THREADS) y[i] = a * x[i];
        } else if (op.kind == TOP_AXPY) {                         // vec_axpy_kernel
            const T a = (T)op.c;
            for (int i = tid; i < op.n; i += TAIL_THREADS) { const T t = a * x[i]; y[i] = y[i] + t; }
        }
        __syncthreads();                                          // one workgroup: its own stores are visible to it past the barrier
    }
}


}  // namespace pamg
