// pamg_aggregate.hip -- small dense per-block work of the setup phase, one block per lane:
//   * amg_core::pinv_array (reference pyamg/amg_core/linalg.h:930-1000): the pseudo-inverse of every n x n block of an
//     (m, n, n) array through a one-sided Jacobi SVD (svd_jacobi, linalg.h:546-812) -- what get_block_diag(A, bs,
//     inv_flag=True) (util/utils.py:603-692) runs for bs < 7, i.e. the Dinv of block_jacobi / block_gauss_seidel and of
//     the block-weighted prolongation smoother.
// The arithmetic follows the reference expression by expression (separate multiply and add, IEEE divide and square
// root, the same loop nests), so the blocks come out bit for bit.
#include <limits>

#include "pamg_common.h"

using namespace pamg;

namespace {

constexpr int PN = 6;            // the reference switches to a LAPACK-based routine at n >= 7 (utils.py:682-687)

template <typename T>
struct JacobiSvd {
    // column-major n x n factors, all in registers / scratch of one lane
    T U[PN * PN], V[PN * PN], S[PN];
    int n;

    __device__ T coldot(int a, int b) const
    {
        T s = T(0);
        for (int i = 0; i < n; ++i) s += U[a * n + i] * U[b * n + i];
        return s;
    }
    __device__ T colnorm(int a) const { return sqrt(coldot(a, a)); }

    // linalg.h:546-812 for a square real block held column-major in A
    __device__ void run(const T *A)
    {
        const int nn = n * n;
        if (n == 1) {                                    // :559-571
            const T na = fabs(A[0]);
            V[0] = T(1);
            S[0] = na;
            U[0] = (na == T(0)) ? T(1) : A[0] / na;
            return;
        }
        const T eps = std::numeric_limits<T>::epsilon();
        int count = 1, sweep = 0;
        const int sweepmax = max(15 * n, 30);
        const T tolerance = sqrt((T)n) * eps;
        for (int i = 0; i < nn; ++i) V[i] = T(0);
        for (int i = 0; i < nn; i += n + 1) V[i] = T(1);
        for (int i = 0; i < nn; ++i) U[i] = A[i];
        for (int j = 0; j < n; ++j) S[j] = eps * colnorm(j);                  // column error estimates, :598-603
        while (count > 0 && sweep <= sweepmax) {
            count = n * (n - 1) / 2;
            for (int j = 0; j < n - 1; ++j) {
                for (int k = j + 1; k < n; ++k) {
                    const T a = colnorm(j), b = colnorm(k);
                    const T d = coldot(j, k);
                    const T nd = fabs(d);
                    const T ea = S[j], eb = S[k];
                    const bool sorted = a >= b;
                    const bool orthog = nd <= tolerance * a * b;
                    const bool noisya = a < ea, noisyb = b < eb;
                    if (sorted && (orthog || noisya || noisyb)) {
                        --count;
                    } else if (!sorted || (nd == T(0) && a == b)) {
                        // swap the columns with one sign flip, :651-686
                        S[j] = eb;
                        S[k] = ea;
                        for (int i = 0; i < n; ++i) {
                            const T uj = U[j * n + i], uk = U[k * n + i];
                            U[j * n + i] = -uk;
                            U[k * n + i] = uj;
                        }
                        for (int i = 0; i < n; ++i) {
                            const T vj = V[j * n + i], vk = V[k * n + i];
                            V[j * n + i] = -vk;
                            V[k * n + i] = vj;
                        }
                    } else {
                        // Jacobi rotation, :689-732
                        const T tau = (b * b - a * a) / (T(2) * nd);
                        const T sg = tau < T(0) ? T(-1) : T(1);
                        // the reference's literals are doubles: with T = float these two expressions are evaluated in double
                        // and rounded once (1.0 + tau*tau, 1.0 + t*t); with T = double nothing changes
                        const T t = (T)((double)sg / ((double)fabs(tau) + sqrt(1.0 + (double)(tau * tau))));
                        const T c = (T)(1.0 / sqrt(1.0 + (double)(t * t)));
                        const T s = d * (t * c / nd);
                        const T ms = -s;
                        const T ns = fabs(s);
                        S[j] = fabs(c) * ea + ns * eb;
                        S[k] = ns * ea + fabs(c) * eb;
                        for (int i = 0; i < n; ++i) {
                            const T uj = U[j * n + i], uk = U[k * n + i];
                            U[j * n + i] = uj * c + ms * uk;
                            U[k * n + i] = s * uj + uk * c;
                        }
                        for (int i = 0; i < n; ++i) {
                            const T vj = V[j * n + i], vk = V[k * n + i];
                            V[j * n + i] = vj * c + ms * vk;
                            V[k * n + i] = s * vj + vk * c;
                        }
                    }
                }
            }
            ++sweep;
        }
        // singular values, :745-790
        T sigma_tol = T(0);
        int iszero = n;
        for (int j = 0; j < n; ++j) {
            const T cn = colnorm(j);
            if (j == 0) {
                const T alpha = T(50) / sqrt(sqrt(eps));
                sigma_tol = alpha * cn * eps;
            }
            if (cn <= sigma_tol) {
                --iszero;
                S[j] = T(0);
                for (int i = 0; i < n; ++i) U[j * n + i] = T(0);
            } else {
                S[j] = cn;
                for (int i = 0; i < n; ++i) U[j * n + i] = U[j * n + i] / cn;
            }
        }
        if (iszero == 0) {                                // the zero matrix: U = V = I, :792-805
            for (int i = 0; i < nn; ++i) V[i] = T(0);
            for (int i = 0; i < nn; i += n + 1) V[i] = T(1);
            for (int i = 0; i < nn; i += n + 1) U[i] = T(1);
        }
    }
};

// linalg.h:930-1000.  transA: the blocks are row-major (Python arrays) -> transposed into column-major for the SVD.
template <typename T>
__global__ __launch_bounds__(64) void pinv_array_kernel(T *AA, int64_t m, int n, int transA)
{
    const int64_t blk = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (blk >= m) return;
    T *a = AA + blk * n * n;
    JacobiSvd<T> sv;
    sv.n = n;
    T in[PN * PN], W[PN * PN];
    if (transA) { for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) in[c * n + r] = a[r * n + c]; }
    else { for (int i = 0; i < n * n; ++i) in[i] = a[i]; }
    sv.run(in);
    for (int j = 0; j < n; ++j) if (sv.S[j] != T(0)) sv.S[j] = T(1) / sv.S[j];
    // W(k, j) = S_k^-1 U(j, k), column-major (:967-978)
    for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k) W[j * n + k] = sv.U[k * n + j] * sv.S[k];
    // block <- V W, accumulated from zero over k in order (:983-985, gemm of :437-458), stored row-major
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            T acc = T(0);
            for (int k = 0; k < n; ++k) acc += sv.V[k * n + i] * W[j * n + k];
            a[i * n + j] = acc;
        }
}

int pinv_launch(int dtype, void *d_AA, int64_t m, int n, int transA, hipStream_t s)
{
    if (m == 0) return PAMG_OK;
    const int64_t grid = (m + 63) / 64;
    if (grid > INT32_MAX) return PAMG_E_UNSUPPORTED;
    if (dtype == PAMG_F64) hipLaunchKernelGGL((pinv_array_kernel<double>), dim3((unsigned)grid), dim3(64), 0, s, (double *)d_AA, m, n, transA);
    else hipLaunchKernelGGL((pinv_array_kernel<float>), dim3((unsigned)grid), dim3(64), 0, s, (float *)d_AA, m, n, transA);
    return (int)hipGetLastError();
}

int pinv_host(int dtype, void *AA, int AA_size, int m, int n, char TransA)
{
    if (!AA || m < 0 || n < 1 || (TransA != 'T' && TransA != 'F')) return PAMG_E_ARG;
    if ((int64_t)m * n * n != (int64_t)AA_size) return PAMG_E_ARG;
    if (n > PN) return PAMG_E_UNSUPPORTED;
    if (m == 0) return PAMG_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return PAMG_E_NODEVICE;
    const size_t bytes = (size_t)AA_size * tsize(dtype);
    void *d = nullptr;
    PAMG_HIP(hipMalloc(&d, bytes));
    int st = (int)hipMemcpy(d, AA, bytes, hipMemcpyHostToDevice);
    if (!st) st = pinv_launch(dtype, d, m, n, TransA == 'T' ? 1 : 0, nullptr);
    if (!st) st = (int)hipMemcpy(AA, d, bytes, hipMemcpyDeviceToHost);
    hipFree(d);
    return st;
}

}  // namespace

extern "C" {

int pamg_pinv_array_f64(double *AA, int AA_size, int32_t m, int32_t n, char TransA) { return pinv_host(PAMG_F64, AA, AA_size, m, n, TransA); }
int pamg_pinv_array_f32(float *AA, int AA_size, int32_t m, int32_t n, char TransA) { return pinv_host(PAMG_F32, AA, AA_size, m, n, TransA); }

int pamg_dev_pinv_array(int dtype, void *AA, int64_t m, int n, int transA, pamg_stream_t s)
{
    if ((dtype != PAMG_F64 && dtype != PAMG_F32) || m < 0 || n < 1 || (m > 0 && !AA)) return PAMG_E_ARG;
    if (n > PN) return PAMG_E_UNSUPPORTED;
    return pinv_launch(dtype, AA, m, n, transA ? 1 : 0, (hipStream_t)s);
}

}  // extern "C"
