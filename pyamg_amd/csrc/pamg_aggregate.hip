// pamg_aggregate.hip -- small dense per-block work of the setup phase, one block per lane:
//   * amg_core::pinv_array (reference pyamg/amg_core/linalg.h:930-1000): the pseudo-inverse of every n x n block of an
//     (m, n, n) array through a one-sided Jacobi SVD (svd_jacobi, linalg.h:546-812) -- what get_block_diag(A, bs,
//     inv_flag=True) (util/utils.py:603-692) runs for bs < 7, i.e. the Dinv of block_jacobi / block_gauss_seidel and of
//     the block-weighted prolongation smoother.
// The arithmetic follows the reference expression by expression (separate multiply and add, IEEE divide and square
// root, the same loop nests), so the blocks come out bit for bit.
#include <algorithm>
#include <climits>
#include <limits>
#include <vector>

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

// ================================================================================================================
// amg_core::standard_aggregation (reference pyamg/amg_core/smoothed_aggregation.h:137-268) on a device-resident CSR
// pattern -- the same aggregates, the same numbering, the same C-points, integer for integer.
//
// The reference's first pass is a sequential greedy sweep: vertex i becomes the root of a new aggregate iff, when its
// turn comes, neither i nor any of its neighbours carries a mark; a root marks itself and its neighbours.  What i reads
// (the marks of N[i] = {i} + row(i)) was possibly written by every earlier vertex k with N[k] meeting N[i] -- a dependency
// of distance two.  It is honoured here without ever forming two-hop neighbourhoods: any two members of one N[v] depend
// on each other, so within N[v] the turns go in index order -- every v simply passes a token down its SORTED member list:
// when the member of rank c has had its turn, v hands the token to the member of rank c + 1.  A vertex may take its turn
// once it holds the token of every v in N[i] (pend[i] counts the ones still missing; the smallest member of a list holds
// that list's token from the start).  That is a topological traversal of the dependency graph: rounds of one launch
// each, the vertices that became ready in round r run in round r + 1 (so everything they read was written before their
// launch began -- no fences, no polling), 1 500 rounds on the 256^3 grid, a few microseconds each.
// (Symmetric patterns without duplicate entries: what strength-of-connection matrices are.  Anything else:
// PAMG_E_UNSUPPORTED.)  Passes 2 and 3 of the reference read only what pass 1 wrote (on a symmetric pattern its third
// pass never opens a new aggregate: every vertex is within distance two of a root) and are plain data-parallel kernels
// plus a prefix sum.
namespace {

constexpr int AGG_ISO = INT32_MIN;          // isolated vertex (the reference's x = -n_row)

// Member lists: Ns[Ap[v] + v ...] = {v} + row(v) sorted ascending, nsize[v] members.  flags: bit 0 pattern not symmetric,
// bit 1 duplicate entries.  Isolated vertices (no neighbour but themselves) are marked at once and take no part.
__global__ __launch_bounds__(BLK) void agg_lists_kernel(int n, const int *Ap, const int *Aj, int *Ns, int *nsize, int *mark, unsigned *flag,
                                                        unsigned *nlive)
{
    for (int v = blockIdx.x * BLK + threadIdx.x; v < n; v += gridDim.x * BLK) {
        int *L = Ns + (size_t)Ap[v] + v;
        int m = 0;
        L[m++] = v;
        bool dup = false;
        for (int p = Ap[v]; p < Ap[v + 1]; ++p) {
            const int k = Aj[p];
            if (k == v) { for (int q = p + 1; q < Ap[v + 1]; ++q) dup = dup || Aj[q] == v; continue; }
            int b = m - 1;                                      // insertion sort: rows hold a few tens of entries
            while (b >= 0 && L[b] > k) { L[b + 1] = L[b]; --b; }
            if (b >= 0 && L[b] == k) { dup = true; for (int t = b + 1; t < m; ++t) L[t] = L[t + 1]; continue; }
            L[b + 1] = k;
            ++m;
        }
        nsize[v] = m;
        if (dup) atomicOr(flag, 2u);
        if (m == 1) mark[v] = AGG_ISO;
        else atomicAdd(nlive, 1u);
    }
}

// pend[i] = lists of N[i] whose token i does not hold from the start; symmetry check (i must be a member of every list it reads);
// vertices that hold every token go on the first work list
__global__ __launch_bounds__(BLK) void agg_pend_kernel(int n, const int *Ap, const int *Ns, const int *nsize, int *pend, int *wl, unsigned *wcount,
                                                       unsigned *flag)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const int m = nsize[i];
        if (m == 1) { pend[i] = 0; continue; }
        const int *L = Ns + (size_t)Ap[i] + i;
        int need = 0;
        for (int t = 0; t < m; ++t) {
            const int v = L[t];
            const int *Lv = Ns + (size_t)Ap[v] + v;
            const int mv = nsize[v];
            int lo = 0, hi = mv;                                // i's place in v's list
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (Lv[mid] < i) lo = mid + 1; else hi = mid; }
            if (lo >= mv || Lv[lo] != i) atomicOr(flag, 1u);
            need += lo > 0 ? 1 : 0;
        }
        pend[i] = need;
        if (need == 0) wl[atomicAdd(wcount, 1u)] = i;
    }
}

// one round: the vertices of the incoming work list take their turn (smoothed_aggregation.h:160-188), then every list
// they belong to passes its token on; vertices that now hold all their tokens form the next work list
__global__ __launch_bounds__(BLK) void agg_round_kernel(const int *Ap, const int *Aj, const int *Ns, const int *nsize, int *cnt, int *pend, int *mark,
                                                        const int *wl_in, const unsigned *n_in, int *wl_out, unsigned *n_out, unsigned *n_clear,
                                                        unsigned *ndone)
{
    const unsigned nin = *n_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_clear = 0u;    // the list after next: nobody reads or writes it in this round
    for (unsigned t = blockIdx.x * BLK + threadIdx.x; t < nin; t += gridDim.x * BLK) {
        const int i = wl_in[t];
        const int lo = Ap[i], hi = Ap[i + 1];
        bool free = mark[i] == 0;
        for (int p = lo; p < hi && free; ++p) free = mark[Aj[p]] == 0;
        if (free) {                                             // a new aggregate: the root and its neighbours
            mark[i] = i + 1;
            for (int p = lo; p < hi; ++p) mark[Aj[p]] = i + 1;
        }
        const int *L = Ns + (size_t)lo + i;
        const int m = nsize[i];
        for (int q = 0; q < m; ++q) {
            const int v = L[q];
            const int c = atomicAdd(cnt + v, 1) + 1;            // members of v's list that have had their turn
            if (c < nsize[v]) {
                const int nxt = Ns[(size_t)Ap[v] + v + c];
                if (atomicSub(pend + nxt, 1) == 1) wl_out[atomicAdd(n_out, 1u)] = nxt;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ndone, nin);
}

// pass 2 (smoothed_aggregation.h:190-205): an unmarked vertex joins the aggregate of its first neighbour (storage order)
// that pass 1 put into one; recorded as the negative mark
__global__ __launch_bounds__(BLK) void agg_pass2_kernel(int n, const int *Ap, const int *Aj, int *mark)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        if (mark[i] != 0) continue;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int m = mark[Aj[p]];
            if (m > 0) { mark[i] = -m; break; }
        }
    }
}

constexpr int SCAN_PER = 8;                 // elements per lane in the block scan

// roots per block of BLK * SCAN_PER vertices
__global__ __launch_bounds__(BLK) void agg_count_kernel(int n, const int *mark, int *blocksum)
{
    __shared__ int sm[BLK];
    const long long b0 = (long long)blockIdx.x * BLK * SCAN_PER + (long long)threadIdx.x * SCAN_PER;
    int c = 0;
    for (int e = 0; e < SCAN_PER; ++e) { const long long i = b0 + e; if (i < n && mark[i] == (int)i + 1) ++c; }
    sm[threadIdx.x] = c;
    __syncthreads();
    for (int s = BLK / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) blocksum[blockIdx.x] = sm[0];
}

// rid[i] = number of roots below i (the aggregate number of root i), y[rid[i]] = i for roots
__global__ __launch_bounds__(BLK) void agg_rank_kernel(int n, const int *mark, const int *blockoff, int *rid, int *y)
{
    __shared__ int sm[BLK];
    const long long b0 = (long long)blockIdx.x * BLK * SCAN_PER + (long long)threadIdx.x * SCAN_PER;
    int c = 0;
    for (int e = 0; e < SCAN_PER; ++e) { const long long i = b0 + e; if (i < n && mark[i] == (int)i + 1) ++c; }
    sm[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < BLK; d <<= 1) {                          // inclusive scan over the lanes
        const int t = (int)threadIdx.x >= d ? sm[threadIdx.x - d] : 0;
        __syncthreads();
        sm[threadIdx.x] += t;
        __syncthreads();
    }
    int run = blockoff[blockIdx.x] + sm[threadIdx.x] - c;
    for (int e = 0; e < SCAN_PER; ++e) {
        const long long i = b0 + e;
        if (i >= n) break;
        rid[i] = run;
        if (mark[i] == (int)i + 1) { y[run] = (int)i; ++run; }
    }
}

// pass 3 (smoothed_aggregation.h:210-245) on what passes 1 and 2 left: aggregate numbers from 0, -1 for isolated vertices
__global__ __launch_bounds__(BLK) void agg_final_kernel(int n, const int *mark, const int *rid, int *x, unsigned *flag)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const int m = mark[i];
        if (m == AGG_ISO) x[i] = -1;
        else if (m > 0) x[i] = rid[m - 1];
        else if (m < 0) x[i] = rid[-m - 1];
        else { x[i] = -1; atomicOr(flag, 4u); }                  // would open an aggregate in the reference's third pass
    }
}

int agg_grid(int64_t n, int cap = 4096) { return (int)std::min<int64_t>(cap, std::max<int64_t>(1, (n + BLK - 1) / BLK)); }

// device arrays Ap[n+1], Aj[nnz] -> device x[n], y[n]; *count
int standard_aggregation_device(int n, const int *d_Ap, const int *d_Aj, int64_t nnz, int *d_x, int *d_y, int *count)
{
    *count = 0;
    if (n == 0) return PAMG_OK;
    int *Ns = nullptr, *nsize = nullptr, *cnt = nullptr, *pend = nullptr, *mark = nullptr, *rid = nullptr, *bsum = nullptr, *wl = nullptr;
    unsigned *ctl = nullptr;                                     // [0] flags, [1] live vertices, [2] vertices done, [4..6] work-list sizes
    const int nb = (int)(((int64_t)n + (int64_t)BLK * SCAN_PER - 1) / ((int64_t)BLK * SCAN_PER));
    int st = PAMG_OK;
    auto cleanup = [&]() { hipFree(Ns); hipFree(nsize); hipFree(cnt); hipFree(pend); hipFree(mark); hipFree(rid); hipFree(bsum); hipFree(wl); hipFree(ctl); };
#define AGG_CHECK(expr) do { st = (int)(expr); if (st) { cleanup(); return st; } } while (0)
    AGG_CHECK(hipMalloc((void **)&Ns, sizeof(int) * ((size_t)nnz + (size_t)n + 8)));
    AGG_CHECK(hipMalloc((void **)&nsize, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMalloc((void **)&cnt, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMalloc((void **)&pend, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMalloc((void **)&mark, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMalloc((void **)&rid, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMalloc((void **)&bsum, sizeof(int) * (size_t)(nb + 1)));
    AGG_CHECK(hipMalloc((void **)&wl, sizeof(int) * 3 * (size_t)n));
    AGG_CHECK(hipMalloc((void **)&ctl, 8 * sizeof(unsigned)));
    AGG_CHECK(hipMemset(cnt, 0, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMemset(mark, 0, sizeof(int) * (size_t)n));
    AGG_CHECK(hipMemset(ctl, 0, 8 * sizeof(unsigned)));
    hipLaunchKernelGGL(agg_lists_kernel, dim3(agg_grid(n)), dim3(BLK), 0, 0, n, d_Ap, d_Aj, Ns, nsize, mark, ctl, ctl + 1);
    hipLaunchKernelGGL(agg_pend_kernel, dim3(agg_grid(n)), dim3(BLK), 0, 0, n, d_Ap, (const int *)Ns, (const int *)nsize, pend, wl, ctl + 4, ctl);
    AGG_CHECK(hipGetLastError());
    unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    AGG_CHECK(hipMemcpy(h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (h[0] & 3u) { cleanup(); return PAMG_E_UNSUPPORTED; }     // not symmetric / duplicate entries: the token argument above does not hold
    const unsigned live = h[1];
    // rounds in batches (no host synchronisation inside a batch: every launch reads its list size from the device)
    int dev = 0, cus = 64;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    const int grid = std::max(1, std::min(cus * 4, (n + BLK - 1) / BLK));
    unsigned done = 0;
    int round = 0;
    const int64_t round_cap = 4 * (int64_t)n + 64;               // a traversal needs at most one round per vertex
    while (done < live) {
        const int batch = (int)std::min<int64_t>(256, std::max<int64_t>(8, (int64_t)live - done));
        for (int k = 0; k < batch; ++k, ++round) {
            const int a = round % 3, b = (round + 1) % 3, c = (round + 2) % 3;
            hipLaunchKernelGGL(agg_round_kernel, dim3(grid), dim3(BLK), 0, 0, d_Ap, d_Aj, (const int *)Ns, (const int *)nsize, cnt, pend, mark,
                               (const int *)(wl + (size_t)a * n), (const unsigned *)(ctl + 4 + a), wl + (size_t)b * n, ctl + 4 + b, ctl + 4 + c, ctl + 2);
        }
        AGG_CHECK(hipGetLastError());
        const unsigned before = done;
        AGG_CHECK(hipMemcpy(&done, ctl + 2, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (done < live && (done == before || round > round_cap)) {                                      // no progress: cannot happen on a valid pattern
            unsigned w[8];
            hipMemcpy(w, ctl, sizeof(w), hipMemcpyDeviceToHost);
            fprintf(stderr, "[pamg aggregation] no progress: n %d nnz %lld round %d done %u of %u, list sizes %u %u %u, flags %u\n", n, (long long)nnz, round,
                    done, live, w[4], w[5], w[6], w[0]);
            cleanup();
            return PAMG_E_STATE;
        }
    }
    hipLaunchKernelGGL(agg_pass2_kernel, dim3(agg_grid(n)), dim3(BLK), 0, 0, n, d_Ap, d_Aj, mark);
    hipLaunchKernelGGL(agg_count_kernel, dim3(nb), dim3(BLK), 0, 0, n, (const int *)mark, bsum);
    AGG_CHECK(hipGetLastError());
    std::vector<int> hb((size_t)nb + 1, 0);
    AGG_CHECK(hipMemcpy(hb.data(), bsum, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost));
    int64_t acc = 0;
    for (int b = 0; b < nb; ++b) { const int c = hb[b]; hb[b] = (int)acc; acc += c; }
    *count = (int)acc;
    AGG_CHECK(hipMemcpy(bsum, hb.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(agg_rank_kernel, dim3(nb), dim3(BLK), 0, 0, n, (const int *)mark, (const int *)bsum, rid, d_y);
    hipLaunchKernelGGL(agg_final_kernel, dim3(agg_grid(n)), dim3(BLK), 0, 0, n, (const int *)mark, (const int *)rid, d_x, ctl);
    AGG_CHECK(hipGetLastError());
    AGG_CHECK(hipMemcpy(h, ctl, sizeof(unsigned), hipMemcpyDeviceToHost));
#undef AGG_CHECK
    cleanup();
    if (h[0] & 4u) return PAMG_E_UNSUPPORTED;
    return PAMG_OK;
}

// ================================================================================================================
// amg_core::fit_candidates (smoothed_aggregation.h:484-610): per aggregate the rows of B that belong to it -- a
// (nodes * K1) x K2 dense matrix -- are orthonormalised column by column (modified Gram-Schmidt, every sum over the rows
// in order), R gets the coefficients.  One lane per aggregate, the reference's loops as they are.
// lists: Ap[n_col + 1] / Ai = the nodes of every aggregate (the CSC arrays of AggOp); work: (nnz, K1, K2) in list order;
// sort != 0: the lists were filled in arbitrary order and are sorted first (ascending node = what tocsc() produces);
// out / outpos != nullptr: block ii of the work array is finally copied to out[outpos[Ai[ii]]] (row order of AggOp).
template <typename T>
__global__ __launch_bounds__(64) void fit_candidates_kernel(int n_col, int K1, int K2, const int *Ap, int *Ai, T *work, const T *B, T *R,
                                                            T tol, int sort, T *out, const int *outpos)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_col) return;
    const int BS = K1 * K2;
    const int c0 = Ap[j], c1 = Ap[j + 1];
    if (sort)
        for (int a = c0 + 1; a < c1; ++a) {                       // insertion sort: aggregates hold a few tens of nodes
            const int v = Ai[a];
            int b = a - 1;
            while (b >= c0 && Ai[b] > v) { Ai[b + 1] = Ai[b]; --b; }
            Ai[b + 1] = v;
        }
    T *Rj = R + (size_t)j * K2 * K2;
    for (int e = 0; e < K2 * K2; ++e) Rj[e] = T(0);
    T *W0 = work + (size_t)BS * c0, *W1 = work + (size_t)BS * c1;
    for (int ii = c0; ii < c1; ++ii)
        for (int e = 0; e < BS; ++e) work[(size_t)BS * ii + e] = B[(size_t)BS * Ai[ii] + e];
    for (int bj = 0; bj < K2; ++bj) {
        T norm_j = T(0);
        for (T *p = W0 + bj; p < W1; p += K2) norm_j += (*p) * (*p);
        norm_j = sqrt(norm_j);
        const T threshold_j = tol * norm_j;
        for (int bi = 0; bi < bj; ++bi) {
            T dot = T(0);
            for (T *pi = W0 + bi, *pj = W0 + bj; pi < W1; pi += K2, pj += K2) dot += (*pi) * (*pj);
            for (T *pi = W0 + bi, *pj = W0 + bj; pi < W1; pi += K2, pj += K2) *pj -= dot * (*pi);
            Rj[K2 * bi + bj] = dot;
        }
        norm_j = T(0);
        for (T *p = W0 + bj; p < W1; p += K2) norm_j += (*p) * (*p);
        norm_j = sqrt(norm_j);
        T scale;
        if (norm_j > threshold_j) { scale = (T)(1.0 / norm_j); Rj[K2 * bj + bj] = norm_j; }
        else { scale = T(0); Rj[K2 * bj + bj] = T(0); }
        for (T *p = W0 + bj; p < W1; p += K2) *p *= scale;
    }
    if (out)
        for (int ii = c0; ii < c1; ++ii)
            for (int e = 0; e < BS; ++e) out[(size_t)BS * outpos[Ai[ii]] + e] = work[(size_t)BS * ii + e];
}

// aggregate sizes / node lists from the aggregate number of every node (-1: none)
__global__ __launch_bounds__(BLK) void agglist_count_kernel(int n, const int *Tp, const int *Tj, int *cnt)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        if (Tp[i + 1] > Tp[i]) atomicAdd(cnt + Tj[Tp[i]], 1);
}

__global__ __launch_bounds__(BLK) void agglist_fill_kernel(int n, const int *Tp, const int *Tj, const int *Cp, int *cursor, int *Ci)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK)
        if (Tp[i + 1] > Tp[i]) { const int a = Tj[Tp[i]]; Ci[Cp[a] + atomicAdd(cursor + a, 1)] = i; }
}

template <typename T>
int fit_launch(int n_col, int K1, int K2, const int *Ap, int *Ai, T *work, const T *B, T *R, T tol, int sort, T *out, const int *outpos)
{
    if (n_col == 0) return PAMG_OK;
    hipLaunchKernelGGL((fit_candidates_kernel<T>), dim3((n_col + 63) / 64), dim3(64), 0, 0, n_col, K1, K2, Ap, Ai, work, B, R, tol, sort, out, outpos);
    return (int)hipGetLastError();
}

struct DevBufs {
    std::vector<void *> p;
    ~DevBufs() { for (void *q : p) hipFree(q); }
    template <typename U> int get(U **out, size_t count)
    {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(U), 256));
        if (e != hipSuccess) return (int)e;
        p.push_back(q);
        *out = (U *)q;
        return PAMG_OK;
    }
};

// amg_core.fit_candidates on HOST buffers (the reference's argument order, smoothed_aggregation_bind.cpp:134-170)
template <typename T>
int fit_candidates_host(int n_row, int n_col, int K1, int K2, const int32_t *Ap, int Ap_size, const int32_t *Ai, int Ai_size, T *Ax, int Ax_size,
                        const T *B, int B_size, T *R, int R_size, T tol)
{
    if (n_row < 0 || n_col < 0 || K1 < 1 || K2 < 1 || !Ap || Ap_size != n_col + 1) return PAMG_E_ARG;
    const int64_t nnz = Ap[n_col];
    if (nnz < 0 || nnz != Ai_size || (int64_t)Ax_size != nnz * K1 * K2 || (int64_t)B_size != (int64_t)n_row * K1 * K2 ||
        (int64_t)R_size != (int64_t)n_col * K2 * K2 || (nnz && (!Ai || !Ax || !B)) || (n_col && !R))
        return PAMG_E_ARG;
    for (int64_t k = 0; k < nnz; ++k) if (Ai[k] < 0 || Ai[k] >= n_row) return PAMG_E_ARG;
    if (n_col == 0) return PAMG_OK;
    DevBufs d;
    int *dAp, *dAi;
    T *dW, *dB, *dR;
    PAMG_TRY(d.get(&dAp, (size_t)n_col + 1)); PAMG_TRY(d.get(&dAi, (size_t)nnz));
    PAMG_TRY(d.get(&dW, (size_t)Ax_size)); PAMG_TRY(d.get(&dB, (size_t)B_size)); PAMG_TRY(d.get(&dR, (size_t)R_size));
    PAMG_HIP(hipMemcpy(dAp, Ap, sizeof(int) * ((size_t)n_col + 1), hipMemcpyHostToDevice));
    if (nnz) PAMG_HIP(hipMemcpy(dAi, Ai, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    if (B_size) PAMG_HIP(hipMemcpy(dB, B, sizeof(T) * (size_t)B_size, hipMemcpyHostToDevice));
    PAMG_TRY(fit_launch<T>(n_col, K1, K2, dAp, dAi, dW, dB, dR, tol, 0, nullptr, nullptr));
    if (Ax_size) PAMG_HIP(hipMemcpy(Ax, dW, sizeof(T) * (size_t)Ax_size, hipMemcpyDeviceToHost));
    PAMG_HIP(hipMemcpy(R, dR, sizeof(T) * (size_t)R_size, hipMemcpyDeviceToHost));
    return PAMG_OK;
}

// tentative.fit_candidates fused (aggregation/tentative.py:9-152): AggOp in CSR (at most one entry per row), B (n_fine * K1, K2)
// -> the blocks of the tentative prolongator in AggOp's row order (T.data) and the coarse candidates R (n_coarse * K2, K2)
template <typename T>
int fit_tentative_host(int n_fine, int n_coarse, int K1, int K2, const int32_t *Tp, const int32_t *Tj, const T *B, T *Qx, T *R, T tol)
{
    if (n_fine < 0 || n_coarse < 0 || K1 < 1 || K2 < 1 || !Tp || (n_fine && !B)) return PAMG_E_ARG;
    const int64_t nnz = Tp[n_fine];
    if (nnz < 0 || nnz > n_fine || (nnz && (!Tj || !Qx)) || (n_coarse && !R)) return PAMG_E_ARG;
    for (int i = 0; i < n_fine; ++i) {
        if (Tp[i + 1] - Tp[i] < 0 || Tp[i + 1] - Tp[i] > 1) return PAMG_E_UNSUPPORTED;      // AggOp of an aggregation: a node is in one aggregate
        if (Tp[i + 1] > Tp[i] && (Tj[Tp[i]] < 0 || Tj[Tp[i]] >= n_coarse)) return PAMG_E_ARG;
    }
    if (n_coarse == 0) return PAMG_OK;
    const size_t BS = (size_t)K1 * K2;
    DevBufs d;
    int *dTp, *dTj, *dCp, *dCi, *dCur;
    T *dW, *dB, *dR, *dQ;
    PAMG_TRY(d.get(&dTp, (size_t)n_fine + 1)); PAMG_TRY(d.get(&dTj, (size_t)std::max<int64_t>(nnz, 1)));
    PAMG_TRY(d.get(&dCp, (size_t)n_coarse + 1)); PAMG_TRY(d.get(&dCi, (size_t)std::max<int64_t>(nnz, 1))); PAMG_TRY(d.get(&dCur, (size_t)n_coarse));
    PAMG_TRY(d.get(&dW, (size_t)nnz * BS)); PAMG_TRY(d.get(&dQ, (size_t)nnz * BS));
    PAMG_TRY(d.get(&dB, (size_t)n_fine * BS)); PAMG_TRY(d.get(&dR, (size_t)n_coarse * K2 * K2));
    PAMG_HIP(hipMemcpy(dTp, Tp, sizeof(int) * ((size_t)n_fine + 1), hipMemcpyHostToDevice));
    if (nnz) PAMG_HIP(hipMemcpy(dTj, Tj, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    if (n_fine) PAMG_HIP(hipMemcpy(dB, B, sizeof(T) * (size_t)n_fine * BS, hipMemcpyHostToDevice));
    PAMG_HIP(hipMemset(dCur, 0, sizeof(int) * (size_t)n_coarse));
    if (n_fine) hipLaunchKernelGGL(agglist_count_kernel, dim3(agg_grid(n_fine)), dim3(BLK), 0, 0, n_fine, (const int *)dTp, (const int *)dTj, dCur);
    PAMG_HIP(hipGetLastError());
    std::vector<int> cp((size_t)n_coarse + 1, 0);
    PAMG_HIP(hipMemcpy(cp.data() + 1, dCur, sizeof(int) * (size_t)n_coarse, hipMemcpyDeviceToHost));
    for (int a = 0; a < n_coarse; ++a) cp[(size_t)a + 1] += cp[(size_t)a];
    PAMG_HIP(hipMemcpy(dCp, cp.data(), sizeof(int) * ((size_t)n_coarse + 1), hipMemcpyHostToDevice));
    PAMG_HIP(hipMemset(dCur, 0, sizeof(int) * (size_t)n_coarse));
    if (n_fine) hipLaunchKernelGGL(agglist_fill_kernel, dim3(agg_grid(n_fine)), dim3(BLK), 0, 0, n_fine, (const int *)dTp, (const int *)dTj, (const int *)dCp, dCur, dCi);
    PAMG_HIP(hipGetLastError());
    PAMG_TRY(fit_launch<T>(n_coarse, K1, K2, dCp, dCi, dW, dB, dR, tol, 1, dQ, dTp));          // outpos[node] = Tp[node]: its only entry
    if (nnz) PAMG_HIP(hipMemcpy(Qx, dQ, sizeof(T) * (size_t)nnz * BS, hipMemcpyDeviceToHost));
    PAMG_HIP(hipMemcpy(R, dR, sizeof(T) * (size_t)n_coarse * K2 * K2, hipMemcpyDeviceToHost));
    return PAMG_OK;
}

}  // namespace

extern "C" {

// amg_core::standard_aggregation on HOST arrays (smoothed_aggregation_bind.cpp:49-75): returns the number of aggregates
// through *naggs (the reference returns it); x = aggregate of every node (-1: none), y = the C-points
int pamg_standard_aggregation(int32_t n_row, const int32_t *Ap, int Ap_size, const int32_t *Aj, int Aj_size, int32_t *x, int x_size,
                              int32_t *y, int y_size, int32_t *naggs)
{
    if (n_row < 0 || !Ap || Ap_size != n_row + 1 || x_size < n_row || y_size < n_row || !naggs || (n_row && (!x || !y))) return PAMG_E_ARG;
    const int64_t nnz = Ap[n_row];
    if (nnz < 0 || nnz != Aj_size || (nnz && !Aj) || Ap[0] != 0) return PAMG_E_ARG;
    for (int i = 0; i < n_row; ++i) if (Ap[i + 1] < Ap[i]) return PAMG_E_ARG;
    for (int64_t p = 0; p < nnz; ++p) if (Aj[p] < 0 || Aj[p] >= n_row) return PAMG_E_ARG;
    *naggs = 0;
    if (n_row == 0) return PAMG_OK;
    DevBufs d;
    int *dAp, *dAj, *dx, *dy;
    PAMG_TRY(d.get(&dAp, (size_t)n_row + 1)); PAMG_TRY(d.get(&dAj, (size_t)std::max<int64_t>(nnz, 1)));
    PAMG_TRY(d.get(&dx, (size_t)n_row)); PAMG_TRY(d.get(&dy, (size_t)n_row));
    PAMG_HIP(hipMemcpy(dAp, Ap, sizeof(int) * ((size_t)n_row + 1), hipMemcpyHostToDevice));
    if (nnz) PAMG_HIP(hipMemcpy(dAj, Aj, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    int count = 0;
    PAMG_TRY(standard_aggregation_device(n_row, dAp, dAj, nnz, dx, dy, &count));
    PAMG_HIP(hipMemcpy(x, dx, sizeof(int) * (size_t)n_row, hipMemcpyDeviceToHost));
    if (count) PAMG_HIP(hipMemcpy(y, dy, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost));
    *naggs = count;
    return PAMG_OK;
}

// the same on a device-resident pattern (e.g. the strength matrix pamg_csr_strength_symmetric just produced): x, y HOST
int pamg_csr_standard_aggregation(pamg_csr_t C, int32_t *x, int32_t *y, int32_t *naggs)
{
    if (!C || !naggs) return PAMG_E_ARG;
    CsrArrays a;
    PAMG_TRY(csr_device_arrays(C, &a));
    if (a.m != a.n || a.m > INT32_MAX) return PAMG_E_ARG;
    const int n = (int)a.m;
    *naggs = 0;
    if (n == 0) return PAMG_OK;
    if (!x || !y) return PAMG_E_ARG;
    DevBufs d;
    int *dx, *dy;
    PAMG_TRY(d.get(&dx, (size_t)n)); PAMG_TRY(d.get(&dy, (size_t)n));
    int count = 0;
    PAMG_TRY(standard_aggregation_device(n, a.p, a.j, a.nnz, dx, dy, &count));
    PAMG_HIP(hipMemcpy(x, dx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (count) PAMG_HIP(hipMemcpy(y, dy, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost));
    *naggs = count;
    return PAMG_OK;
}

int pamg_fit_candidates_f64(int32_t n_row, int32_t n_col, int32_t K1, int32_t K2, const int32_t *Ap, int Ap_size, const int32_t *Ai, int Ai_size,
                            double *Ax, int Ax_size, const double *B, int B_size, double *R, int R_size, double tol)
{ return fit_candidates_host<double>(n_row, n_col, K1, K2, Ap, Ap_size, Ai, Ai_size, Ax, Ax_size, B, B_size, R, R_size, tol); }
int pamg_fit_candidates_f32(int32_t n_row, int32_t n_col, int32_t K1, int32_t K2, const int32_t *Ap, int Ap_size, const int32_t *Ai, int Ai_size,
                            float *Ax, int Ax_size, const float *B, int B_size, float *R, int R_size, float tol)
{ return fit_candidates_host<float>(n_row, n_col, K1, K2, Ap, Ap_size, Ai, Ai_size, Ax, Ax_size, B, B_size, R, R_size, tol); }

int pamg_fit_tentative_f64(int32_t n_fine, int32_t n_coarse, int32_t K1, int32_t K2, const int32_t *Tp, const int32_t *Tj, const double *B,
                           double *Qx, double *R, double tol)
{ return fit_tentative_host<double>(n_fine, n_coarse, K1, K2, Tp, Tj, B, Qx, R, tol); }
int pamg_fit_tentative_f32(int32_t n_fine, int32_t n_coarse, int32_t K1, int32_t K2, const int32_t *Tp, const int32_t *Tj, const float *B,
                           float *Qx, float *R, float tol)
{ return fit_tentative_host<float>(n_fine, n_coarse, K1, K2, Tp, Tj, B, Qx, R, tol); }

}  // extern "C"

// ================================================================================================================
// Transpose of a CSR / BSR matrix as SciPy forms it (sparsetools bsr_transpose -> csr_tocsc on the block pattern, blocks
// transposed): the blocks of a column keep the order of their rows and, inside a row, their stored order -- i.e. ascending
// source position.  R = P.T of the SA setup (aggregation.py:394-397) is 0.6 s of serial host code at 256^3.
// count per column (atomics) -> offsets (host prefix sum) -> unordered fill -> every column sorts its few source
// positions (one lane per column) -> blocks copied transposed.
namespace {

__global__ __launch_bounds__(BLK) void tr_count_kernel(int64_t nblk, const int *Aj, int *cnt)
{
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < nblk; p += (int64_t)gridDim.x * BLK) atomicAdd(cnt + Aj[p], 1);
}

__global__ __launch_bounds__(BLK) void tr_fill_kernel(int64_t nblk, const int *Aj, const int *Bp, int *cursor, int *src)
{
    for (int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x; p < nblk; p += (int64_t)gridDim.x * BLK) {
        const int j = Aj[p];
        src[Bp[j] + atomicAdd(cursor + j, 1)] = (int)p;
    }
}

// one lane per column: ascending source positions; the row of a position by bisection of the row pointer
__global__ __launch_bounds__(BLK) void tr_sort_kernel(int n_bcol, int n_brow, const int *Ap, const int *Bp, int *src, int *Bi)
{
    for (int j = blockIdx.x * BLK + threadIdx.x; j < n_bcol; j += gridDim.x * BLK) {
        const int c0 = Bp[j], c1 = Bp[j + 1];
        for (int a = c0 + 1; a < c1; ++a) {
            const int v = src[a];
            int b = a - 1;
            while (b >= c0 && src[b] > v) { src[b + 1] = src[b]; --b; }
            src[b + 1] = v;
        }
        for (int a = c0; a < c1; ++a) {
            const int p = src[a];
            int lo = 0, hi = n_brow;                               // largest i with Ap[i] <= p
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (Ap[mid] <= p) lo = mid; else hi = mid; }
            Bi[a] = lo;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void tr_blocks_kernel(int64_t nblk, int R, int C, const int *src, const T *Ax, T *Bx)
{
    const int64_t RC = (int64_t)R * C;
    for (int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x; e < nblk * RC; e += (int64_t)gridDim.x * BLK) {
        const int64_t q = e / RC;
        const int k = (int)(e - q * RC), c = k / R, r = k - c * R;   // destination block is C x R, row-major: (c, r)
        Bx[e] = Ax[(int64_t)src[q] * RC + (int64_t)r * C + c];
    }
}

template <typename T>
int transpose_host(int n_brow, int n_bcol, int R, int C, const int32_t *Ap, const int32_t *Aj, const T *Ax, int32_t *Bp, int32_t *Bi, T *Bx)
{
    if (n_brow < 0 || n_bcol < 0 || R < 1 || C < 1 || !Ap || !Bp) return PAMG_E_ARG;
    const int64_t nblk = Ap[n_brow];
    if (nblk < 0 || nblk > INT32_MAX || (nblk && (!Aj || !Ax || !Bi || !Bx))) return PAMG_E_ARG;
    for (int j = 0; j <= n_bcol; ++j) Bp[j] = 0;
    if (nblk == 0) return PAMG_OK;
    const size_t RC = (size_t)R * C;
    DevBufs d;
    int *dAp, *dAj, *dBp, *dCur, *dSrc, *dBi;
    T *dAx, *dBx;
    PAMG_TRY(d.get(&dAp, (size_t)n_brow + 1)); PAMG_TRY(d.get(&dAj, (size_t)nblk)); PAMG_TRY(d.get(&dBp, (size_t)n_bcol + 1));
    PAMG_TRY(d.get(&dCur, (size_t)n_bcol + 1)); PAMG_TRY(d.get(&dSrc, (size_t)nblk)); PAMG_TRY(d.get(&dBi, (size_t)nblk));
    PAMG_TRY(d.get(&dAx, (size_t)nblk * RC)); PAMG_TRY(d.get(&dBx, (size_t)nblk * RC));
    PAMG_HIP(hipMemcpy(dAp, Ap, sizeof(int) * ((size_t)n_brow + 1), hipMemcpyHostToDevice));
    PAMG_HIP(hipMemcpy(dAj, Aj, sizeof(int) * (size_t)nblk, hipMemcpyHostToDevice));
    PAMG_HIP(hipMemcpy(dAx, Ax, sizeof(T) * (size_t)nblk * RC, hipMemcpyHostToDevice));
    PAMG_HIP(hipMemset(dCur, 0, sizeof(int) * ((size_t)n_bcol + 1)));
    const int grid = agg_grid(nblk, 8192);
    hipLaunchKernelGGL(tr_count_kernel, dim3(grid), dim3(BLK), 0, 0, nblk, (const int *)dAj, dCur);
    PAMG_HIP(hipGetLastError());
    std::vector<int> cnt((size_t)n_bcol + 1, 0);
    PAMG_HIP(hipMemcpy(cnt.data(), dCur, sizeof(int) * (size_t)n_bcol, hipMemcpyDeviceToHost));
    int mx = 0;
    int64_t acc = 0;
    for (int j = 0; j < n_bcol; ++j) { const int c = cnt[(size_t)j]; mx = std::max(mx, c); Bp[j] = (int)acc; acc += c; }
    Bp[n_bcol] = (int)acc;
    if (mx > 4096) return PAMG_E_UNSUPPORTED;                      // a column of thousands of blocks: the per-column sort is quadratic
    PAMG_HIP(hipMemcpy(dBp, Bp, sizeof(int) * ((size_t)n_bcol + 1), hipMemcpyHostToDevice));
    PAMG_HIP(hipMemset(dCur, 0, sizeof(int) * ((size_t)n_bcol + 1)));
    hipLaunchKernelGGL(tr_fill_kernel, dim3(grid), dim3(BLK), 0, 0, nblk, (const int *)dAj, (const int *)dBp, dCur, dSrc);
    hipLaunchKernelGGL(tr_sort_kernel, dim3(agg_grid(n_bcol)), dim3(BLK), 0, 0, n_bcol, n_brow, (const int *)dAp, (const int *)dBp, dSrc, dBi);
    hipLaunchKernelGGL((tr_blocks_kernel<T>), dim3(agg_grid((int64_t)nblk * RC, 8192)), dim3(BLK), 0, 0, nblk, R, C, (const int *)dSrc, (const T *)dAx, dBx);
    PAMG_HIP(hipGetLastError());
    PAMG_HIP(hipMemcpy(Bi, dBi, sizeof(int) * (size_t)nblk, hipMemcpyDeviceToHost));
    PAMG_HIP(hipMemcpy(Bx, dBx, sizeof(T) * (size_t)nblk * RC, hipMemcpyDeviceToHost));
    return PAMG_OK;
}

}  // namespace

extern "C" {

/* SciPy's bsr_transpose / csr_tocsc (sparsetools): B = A^T for A (n_brow x n_bcol blocks of R x C; R = C = 1: CSR), all
 * arrays HOST; Bp[n_bcol + 1], Bi[nblk], Bx[nblk * C * R] receive what `A.T` holds in SciPy, order included. */
int pamg_bsr_transpose_f64(int32_t n_brow, int32_t n_bcol, int32_t R, int32_t C, const int32_t *Ap, const int32_t *Aj, const double *Ax,
                           int32_t *Bp, int32_t *Bi, double *Bx)
{ return transpose_host<double>(n_brow, n_bcol, R, C, Ap, Aj, Ax, Bp, Bi, Bx); }
int pamg_bsr_transpose_f32(int32_t n_brow, int32_t n_bcol, int32_t R, int32_t C, const int32_t *Ap, const int32_t *Aj, const float *Ax,
                           int32_t *Bp, int32_t *Bi, float *Bx)
{ return transpose_host<float>(n_brow, n_bcol, R, C, Ap, Aj, Ax, Bp, Bi, Bx); }

}  // extern "C"
