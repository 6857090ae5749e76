// pamg_setup.hip -- setup-phase operators on the device, gfx950 only:
//
//   * Arnoldi process for approximate_spectral_radius           (reference: pyamg/util/linalg.py:154-253, :255-370)
//   * sparse matrix-matrix product, row-wise, ORDER-EXACT         (scipy sparsetools csr_matmat, as called by
//                                                                  aggregation.py:425  A = R @ A @ P  and smooth.py:199)
//   * sparse difference P = T - U                                 (scipy csr_binop_csr_canonical, smooth.py:199)
//   * row / value scaling of an operator                          (util/utils.py scale_rows, smooth.py:165-167)
//
// The product is "expand - sort - compress" inside a workgroup: a row range (whole rows, at most SPG_CAP products) is
// expanded into LDS as (row, column, sequence number | product) pairs, sorted by a bitonic network, and every run of
// equal (row, column) is summed IN SEQUENCE ORDER, which is the order SciPy's Gustavson loop adds them (for k in row
// i of A, in stored order: for j in row k of B: sums[j] += a_ik * b_kj).  So every stored value is bit-identical to
// SciPy's; the runs are then put into SciPy's EMISSION order (reverse order of first touch -- csr_matmat walks a linked
// list from its head) and exact zeros are dropped as SciPy drops them, so the result is the array SciPy would have
// produced, entry for entry.  No atomics on values.  Rows with more than SPG_CAP products (coarse Galerkin products:
// tens of thousands) go through spg_long_kernel: dense per-column accumulators over a window of output columns.
#include "pamg_host_threads.h"
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <new>
#include <thread>
#include <mutex>
#include <vector>

#include "pamg_common.h"
#include "pamg_spg_plan.h"

struct pamg_csr_s {
    int64_t m = 0, n = 0, nnz = 0;
    int *d_p = nullptr, *d_j = nullptr;
    double *d_x = nullptr;
    bool owns = true;                 // false: a view of a pamg_matrix_s (which must outlive it)
    std::vector<int> h_p, h_j;        // host copies of the index arrays, fetched when a plan needs them
    int canon = -1;                   // rows sorted by column without duplicates: -1 unknown, 0 no, 1 yes
};

struct pamg_arnoldi_s {
    pamg_matrix_s *A = nullptr;
    int64_t n = 0;
    int maxiter = 0;
    int planes = 0;                   // planes the basis is currently allocated for (1 real, 2 complex)
    double *d_V = nullptr;            // [maxiter + 1][planes][n]
    double *d_v0 = nullptr;           // [2][n] start vector of the next run (combine() writes it)
    int v0_planes = 0;
    double *d_H = nullptr;            // [(maxiter + 1) * maxiter][2]
    double *d_part = nullptr;         // [2 * ARN_GRID] partial sums + [2] reduced
};

namespace pamg {
namespace {

constexpr int ARN_GRID = 2048;

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// exclusive prefix sum of one int per thread over the workgroup (sm: BLK ints); total returned to everyone
__device__ __forceinline__ int block_excl_scan(int v, int *sm, int &total)
{
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int off = 1; off < BLK; off <<= 1) {
        const int t = tid >= off ? sm[tid - off] : 0;
        __syncthreads();
        sm[tid] += t;
        __syncthreads();
    }
    total = sm[BLK - 1];
    const int incl = sm[tid];
    __syncthreads();
    return incl - v;
}

// ---------------------------------------------------------------------------------------------- sparse product
struct SpgArgs {
    const int4 *ranges;               // tasks: {first row, end row, first column, end column}
    const int *ids;                   // task handled by workgroup b = ids[b]
    const int *Ap, *Aj;
    const double *Ax;
    const int *Bp, *Bj;
    const double *Bx;
    int *rowcount;                    // symbolic: stored columns per row (atomic: the windows of a long row share it)
    int *rcount;                      // symbolic: stored entries per task
    const int *obase;                 // numeric: first output entry of each task
    int *Cj;
    double *Cx;
    int *Cseq;                        // numeric, long rows: first-touch sequence number of every stored entry
    unsigned *flags;                  // [0] exact zeros stored, [1] error (a task exceeded its LDS budget: plan bug)
    int cb;                           // width of the result's column blocks (1: scalar; > 1: SciPy's bsr_matmat emits whole blocks)
    int keep;                         // true-block mode (bsr_matmat): exact zeros are stored, blocks in FORWARD order of first touch
};

__global__ __launch_bounds__(BLK) void spg_count_kernel(int m, const int *Ap, const int *Aj, const int *Bp, int *nprod)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        long long c = 0;
        for (int e = Ap[i]; e < Ap[i + 1]; ++e) { const int k = Aj[e]; c += Bp[k + 1] - Bp[k]; }
        nprod[i] = c > INT_MAX ? INT_MAX : (int)c;
    }
}



// bitonic network on (key, value) pairs in LDS; keys are unique, so the order is total
template <bool WITHV>
__device__ __forceinline__ void lds_sort(unsigned long long *K, double *V, int N)
{
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < (N >> 1); i += BLK) {
                const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l | j;
                const bool up = (l & k) == 0;
                const unsigned long long x = K[l], y = K[r];
                if ((x > y) == up) {
                    K[l] = y; K[r] = x;
                    if (WITHV) { const double t = V[l]; V[l] = V[r]; V[r] = t; }
                }
            }
            __syncthreads();
        }
    }
}

// Whole rows (at most SPG_CAP products per task).  Key of a product: local row (11 bits) | column (31) | sequence
// number inside the task (22).
template <bool NUMERIC, int CAP>
__global__ __launch_bounds__(BLK) void spg_kernel(const SpgArgs a)
{
    constexpr int SPG_PER = CAP / BLK;                                             // sorted positions per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *K = reinterpret_cast<unsigned long long *>(smem);
    double *V = reinterpret_cast<double *>(smem + 8 * CAP);                        // numeric only
    unsigned char *misc = smem + (NUMERIC ? 16 : 8) * CAP;
    double *sA = reinterpret_cast<double *>(misc);                                 // [BLK]
    int *sScan = reinterpret_cast<int *>(sA + BLK);                                // [BLK]
    int *sOff = sScan + BLK;                                                       // [BLK + 1]
    int *sB0 = sOff + BLK + 1;                                                     // [BLK]
    int *sRow = sB0 + BLK;                                                         // [BLK]
    int *sAp = sRow + BLK;                                                         // [SPG_ROWS + 1]
    int *sCnt = sAp + SPG_ROWS + 1;                                                // [SPG_ROWS] symbolic
    const int tid = threadIdx.x;
    const int task = a.ids[blockIdx.x];
    const int4 rg = a.ranges[task];
    const int row0 = rg.x, nrows = rg.y - rg.x;
    for (int r = tid; r <= nrows; r += BLK) sAp[r] = a.Ap[row0 + r];
    if (!NUMERIC) for (int r = tid; r < nrows; r += BLK) sCnt[r] = 0;
    __syncthreads();
    const int eBeg = sAp[0], eEnd = sAp[nrows];
    int total = 0;
    bool overflow = false;
    for (int e0 = eBeg; e0 < eEnd; e0 += BLK) {
        const int e = e0 + tid;
        int len = 0, b0 = 0, lrow = 0;
        double av = 0.0;
        if (e < eEnd) {
            const int k = a.Aj[e];
            b0 = a.Bp[k];
            len = a.Bp[k + 1] - b0;
            if (NUMERIC) av = a.Ax[e];
            int lo = 0, hi = nrows;                       // largest r with sAp[r] <= e
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sAp[mid] <= e) lo = mid; else hi = mid; }
            lrow = lo;
        }
        int chunk;
        const int off = block_excl_scan(len, sScan, chunk);
        sOff[tid] = off; sB0[tid] = b0; sRow[tid] = lrow;
        if (NUMERIC) sA[tid] = av;
        if (tid == 0) sOff[BLK] = chunk;
        __syncthreads();
        if (total + chunk > CAP) { overflow = true; break; }
        for (int q = tid; q < chunk; q += BLK) {
            int lo = 0, hi = BLK;                         // largest i with sOff[i] <= q  (empty entries share an offset: take the last)
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sOff[mid] <= q) lo = mid; else hi = mid; }
            const int pb = sB0[lo] + (q - sOff[lo]);
            const unsigned seq = (unsigned)(total + q);
            K[seq] = ((unsigned long long)sRow[lo] << 53) | ((unsigned long long)(unsigned)a.Bj[pb] << 22) | seq;
            if (NUMERIC) V[seq] = sA[lo] * a.Bx[pb];
        }
        total += chunk;
        __syncthreads();
    }
    if (overflow) { if (tid == 0) atomicOr(a.flags + 1, 1u); return; }
    int N = 2;
    while (N < total) N <<= 1;
    for (int q = total + tid; q < N; q += BLK) { K[q] = ~0ull; if (NUMERIC) V[q] = 0.0; }
    __syncthreads();
    lds_sort<NUMERIC>(K, V, N);
    // runs of equal (row, column): heads, ranks, in-order sums
    const int PER = (N + BLK - 1) / BLK;                  // <= SPG_PER
    const int p0 = tid * PER, p1 = min(p0 + PER, total);
    int heads = 0;
    for (int p = p0; p < p1; ++p) heads += (p == 0 || (K[p] >> 22) != (K[p - 1] >> 22)) ? 1 : 0;
    int nheads;
    int rank = block_excl_scan(heads, sScan, nheads);
    if (!NUMERIC) {
        for (int p = p0; p < p1; ++p)
            if (p == 0 || (K[p] >> 22) != (K[p - 1] >> 22)) atomicAdd(&sCnt[(int)(K[p] >> 53)], 1);
        __syncthreads();
        for (int r = tid; r < nrows; r += BLK) if (sCnt[r]) atomicAdd(a.rowcount + row0 + r, sCnt[r]);
        if (tid == 0) a.rcount[task] = nheads;
        return;
    }
    // SciPy emits a row's entries in REVERSE order of first touch (csr_matmat's linked list): second key =
    // row | (max - sequence number of the run's first product) | column, sorted again
    const int ob = a.obase[task];
    unsigned zeros = 0;
    unsigned long long k2[SPG_PER];
    double sv[SPG_PER];
#pragma unroll
    for (int u = 0; u < SPG_PER; ++u) {
        k2[u] = ~0ull; sv[u] = 0.0;
        const int p = p0 + u;
        if (u < PER && p < p1) {
            const unsigned long long g = K[p] >> 22;
            if (p == 0 || g != (K[p - 1] >> 22)) {
                double s = 0.0;                           // sums[j] = 0; sums[j] += v * Bx  (csr_matmat)
                int q = p;
                do { s += V[q]; ++q; } while (q < total && (K[q] >> 22) == g);
                // time of first touch; the columns of one block share their block's (they are touched one after another)
                const unsigned long long col = g & 0x7FFFFFFFull;
                long long first = (long long)(K[p] & 0x3FFFFFull) - (a.cb > 1 ? (long long)(col % (unsigned)a.cb) : 0ll);
                if (first < 0) first = 0;
                // csr_matmat emits in reverse order of first touch, bsr_matmat (true blocks) in forward order
                k2[u] = ((g >> 31) << 53) | ((a.keep ? (unsigned long long)first : 0x3FFFFFull - (unsigned long long)first) << 31) | col;
                sv[u] = s;
                zeros += (s == 0.0 && !a.keep) ? 1u : 0u;
            }
        }
    }
    if (zeros) atomicAdd(a.flags, zeros);
    __syncthreads();                                      // every read of the first ordering is done
#pragma unroll
    for (int u = 0; u < SPG_PER; ++u)
        if (k2[u] != ~0ull) { K[rank] = k2[u]; V[rank] = sv[u]; ++rank; }
    int N2 = 2;
    while (N2 < nheads) N2 <<= 1;
    __syncthreads();
    for (int q = nheads + tid; q < N2; q += BLK) { K[q] = ~0ull; V[q] = 0.0; }
    __syncthreads();
    lds_sort<true>(K, V, N2);
    for (int q = tid; q < nheads; q += BLK) {
        a.Cj[ob + q] = (int)(K[q] & 0x7FFFFFFFull);
        a.Cx[ob + q] = V[q];
    }
}

// ---- long rows (more than SPG_CAP products: the rows of coarse Galerkin products have tens of thousands) ----------
// One task = one row x one window of SPL_WIN output columns, with a dense accumulator per column of the window in LDS.
// The row's products are walked IN SEQUENCE ORDER, a batch of whole left-operand entries at a time (at most SPL_CAP
// products of the window); a batch is sorted by (column, sequence), and the run of each column continues that column's
// accumulator -- one lane per run, so the additions of a column happen in sequence order across batches.  The first
// touch of a column is remembered (its row-wide sequence number) for the reorder pass.
constexpr int SPL_PER = SPL_CAP / BLK;

__global__ __launch_bounds__(BLK) void spg_minmax_kernel(const int *rows, const int *Ap, const int *Aj, const int *Bp, const int *Bj,
                                                         int *lohi)
{
    __shared__ int smin[BLK], smax[BLK];
    const int r = rows[blockIdx.x];
    int lo = INT_MAX, hi = -1;
    for (int e = Ap[r] + threadIdx.x; e < Ap[r + 1]; e += BLK) {
        const int k = Aj[e];
        for (int p = Bp[k]; p < Bp[k + 1]; ++p) { const int j = Bj[p]; lo = min(lo, j); hi = max(hi, j); }
    }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + st]);
            smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { lohi[2 * blockIdx.x] = smin[0]; lohi[2 * blockIdx.x + 1] = smax[0]; }
}

__global__ __launch_bounds__(BLK) void max_rowlen_kernel(int m, const int *Bp, int *out)
{
    int v = 0;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) v = max(v, Bp[i + 1] - Bp[i]);
    if (v) atomicMax(out, v);
}

template <bool NUMERIC>
__global__ __launch_bounds__(BLK) void spg_long_kernel(const SpgArgs a)
{
    __shared__ unsigned long long K[SPL_CAP];
    __shared__ double V[NUMERIC ? SPL_CAP : 1];
    __shared__ double hval[NUMERIC ? SPL_WIN : 1];
    __shared__ int hseq[SPL_WIN];                         // first-touch sequence number, -1 = column not touched
    __shared__ int sScan[BLK], sOff[BLK + 1], sF[BLK];
    const int tid = threadIdx.x;
    const int task = a.ids[blockIdx.x];
    const int4 rg = a.ranges[task];
    const int row = rg.x, w0 = rg.z, w1 = rg.w;
    for (int c = tid; c < SPL_WIN; c += BLK) { hseq[c] = -1; if (NUMERIC) hval[c] = 0.0; }
    __syncthreads();
    const int eBeg = a.Ap[row], eEnd = a.Ap[row + 1];
    int gtotal = 0;                                       // products of the row before the current batch (all windows)
    int e0 = eBeg;
    bool fail = false;
    while (e0 < eEnd) {
        // candidate batch: the next BLK entries; it is cut to the longest prefix whose in-window products fit
        const int e = e0 + tid;
        int len = 0, flen = 0, fb0 = 0;
        if (e < eEnd) {
            const int k = a.Aj[e];
            fb0 = a.Bp[k];
            flen = a.Bp[k + 1] - fb0;
            for (int p = fb0; p < fb0 + flen; ++p) { const int j = a.Bj[p]; len += (j >= w0 && j < w1) ? 1 : 0; }
        }
        int chunk, fchunk;
        const int off = block_excl_scan(len, sScan, chunk);
        const int foff = block_excl_scan(flen, sScan, fchunk);
        sOff[tid] = off; sF[tid] = foff;
        if (tid == 0) sOff[BLK] = chunk;
        __syncthreads();
        int take = BLK;                                   // entries of this batch
        if (chunk > SPL_CAP) {
            int lo = 0, hi = BLK;                         // largest t with sOff[t] <= SPL_CAP, i.e. entries [0, t) fit
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sOff[mid] <= SPL_CAP) lo = mid; else hi = mid; }
            take = lo;
            if (take == 0) { fail = true; break; }        // one row of B with more than SPL_CAP entries in the window
        }
        const int nprod = take == BLK ? chunk : sOff[take];
        const int fprod = take == BLK ? fchunk : sF[take];
        if (tid < take && len > 0) {
            const double av = NUMERIC ? a.Ax[e] : 0.0;
            int w = off;
            for (int p = fb0; p < fb0 + flen; ++p) {
                const int j = a.Bj[p];
                if (j >= w0 && j < w1) {
                    K[w] = ((unsigned long long)(unsigned)(j - w0) << 32) | (unsigned)(gtotal + foff + (p - fb0));
                    if (NUMERIC) V[w] = av * a.Bx[p];
                    ++w;
                }
            }
        }
        int N = 2;
        while (N < nprod) N <<= 1;
        __syncthreads();
        for (int q = nprod + tid; q < N; q += BLK) { K[q] = ~0ull; if (NUMERIC) V[q] = 0.0; }
        __syncthreads();
        lds_sort<NUMERIC>(K, V, N);
        const int PER = (N + BLK - 1) / BLK;              // <= SPL_PER
        const int p0 = tid * PER, p1 = min(p0 + PER, nprod);
        for (int p = p0; p < p1; ++p) {
            const unsigned c = (unsigned)(K[p] >> 32);
            if (p != 0 && c == (unsigned)(K[p - 1] >> 32)) continue;
            if (hseq[c] < 0) hseq[c] = (int)(unsigned)(K[p] & 0xFFFFFFFFull);
            if (NUMERIC) {
                double s = hval[c];                       // 0 at the first touch: sums[j] = 0; sums[j] += ...
                int q = p;
                do { s += V[q]; ++q; } while (q < nprod && (unsigned)(K[q] >> 32) == c);
                hval[c] = s;
            }
        }
        gtotal += fprod;
        e0 += take;
        __syncthreads();
    }
    if (fail) { if (tid == 0) atomicOr(a.flags + 1, 2u); return; }
    // touched columns of the window, by column; the reorder pass puts the finished row into SciPy's order
    constexpr int CPT = SPL_WIN / BLK;
    int cnt = 0;
    for (int c = tid * CPT; c < (tid + 1) * CPT; ++c) cnt += hseq[c] >= 0 ? 1 : 0;
    int ntouched;
    int rank = block_excl_scan(cnt, sScan, ntouched);
    if (!NUMERIC) {
        if (tid == 0) { a.rcount[task] = ntouched; if (ntouched) atomicAdd(a.rowcount + row, ntouched); }
        return;
    }
    const int ob = a.obase[task];
    unsigned zeros = 0;
    for (int c = tid * CPT; c < (tid + 1) * CPT; ++c) {
        if (hseq[c] < 0) continue;
        a.Cj[ob + rank] = w0 + c;
        a.Cx[ob + rank] = hval[c];
        const int t = max(0, hseq[c] - (a.cb > 1 ? (w0 + c) % a.cb : 0));
        a.Cseq[ob + rank] = a.keep ? INT_MAX - t : t;      // the reorder pass sorts descending: forward order for true blocks
        zeros += (hval[c] == 0.0 && !a.keep) ? 1u : 0u;
        ++rank;
    }
    if (zeros) atomicAdd(a.flags, zeros);
}

// Finished long rows: entries into SciPy's order = descending first-touch sequence number.  One workgroup per row, a
// sorting network with same-direction comparisons only on the row's segment in global memory (virtual padding at the
// end, so any length works).
__global__ __launch_bounds__(BLK) void spg_reorder_kernel(const int *rows, const int *Cp, int *Cj, double *Cx, int *Cseq)
{
    const int r = rows[blockIdx.x];
    const int base = Cp[r], G = Cp[r + 1] - base;
    if (G < 2) return;
    int N = 2;
    while (N < G) N <<= 1;
    int *sq = Cseq + base, *cj = Cj + base;
    double *cx = Cx + base;
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool flip = j == (k >> 1);
            for (int i = threadIdx.x; i < (N >> 1); i += BLK) {
                const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int q = flip ? (l ^ (k - 1)) : (l | j);       // first step of a merge: mirrored partner
                if (q < G && l < G) {
                    const int lo = min(l, q), hi = max(l, q);
                    if (sq[lo] < sq[hi] || (sq[lo] == sq[hi] && cj[lo] > cj[hi])) {   // descending first touch; inside a block by column
                        const int t = sq[lo]; sq[lo] = sq[hi]; sq[hi] = t;
                        const int c = cj[lo]; cj[lo] = cj[hi]; cj[hi] = c;
                        const double v = cx[lo]; cx[lo] = cx[hi]; cx[hi] = v;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(BLK) void rownz_kernel(int m, const int *Cp, const double *Cx, int *cnt)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        int c = 0;
        for (int p = Cp[i]; p < Cp[i + 1]; ++p) c += Cx[p] != 0.0 ? 1 : 0;
        cnt[i] = c;
    }
}

__global__ __launch_bounds__(BLK) void compact_kernel(int m, const int *Cp, const int *Cj, const double *Cx, const int *Np,
                                                      int *Nj, double *Nx)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        int w = Np[i];
        for (int p = Cp[i]; p < Cp[i + 1]; ++p)
            if (Cx[p] != 0.0) { Nj[w] = Cj[p]; Nx[w] = Cx[p]; ++w; }
    }
}

// C = A - B on canonical rows (sorted, no duplicates): SciPy's csr_binop_csr_canonical with std::minus -- a result that
// is exactly zero is not stored.  FILL = false counts, FILL = true writes.
template <bool FILL>
__global__ __launch_bounds__(BLK) void sub_kernel(int m, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
                                                  const int *Bj, const double *Bx, int *cnt, const int *Cp, int *Cj, double *Cx)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        int pa = Ap[i], pb = Bp[i];
        const int ea = Ap[i + 1], eb = Bp[i + 1];
        int w = FILL ? Cp[i] : 0;
        while (pa < ea || pb < eb) {
            const int ja = pa < ea ? Aj[pa] : INT_MAX, jb = pb < eb ? Bj[pb] : INT_MAX;
            double r;
            int j;
            if (ja == jb) { r = Ax[pa] - Bx[pb]; j = ja; ++pa; ++pb; }
            else if (ja < jb) { r = Ax[pa] - 0.0; j = ja; ++pa; }
            else { r = 0.0 - Bx[pb]; j = jb; ++pb; }
            if (r != 0.0) {
                if (FILL) { Cj[w] = j; Cx[w] = r; }
                ++w;
            }
        }
        if (!FILL) cnt[i] = w;
    }
}

// C = A - B when an operand is not canonical: SciPy's csr_binop_csr_general.  Per row: A's entries, then B's, are
// accumulated per column (duplicates summed per operand) in a list in order of first touch; the results A_j - B_j that
// are not exactly zero are emitted in REVERSE order of first touch (the linked list is walked from its head).
// Scratch: row i owns slots [Ap[i] + Bp[i], Ap[i+1] + Bp[i+1]).  One lane per row with a linear search: rows are short.
__global__ __launch_bounds__(BLK) void sub_general_kernel(int m, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
                                                          const int *Bj, const double *Bx, int *tj, double *ta, double *tb,
                                                          int *call, int *cnt)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        const int base = Ap[i] + Bp[i];
        int n = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            int k = 0;
            while (k < n && tj[base + k] != j) ++k;
            if (k == n) { tj[base + n] = j; ta[base + n] = 0.0; tb[base + n] = 0.0; ++n; }
            ta[base + k] = ta[base + k] + Ax[p];
        }
        for (int p = Bp[i]; p < Bp[i + 1]; ++p) {
            const int j = Bj[p];
            int k = 0;
            while (k < n && tj[base + k] != j) ++k;
            if (k == n) { tj[base + n] = j; ta[base + n] = 0.0; tb[base + n] = 0.0; ++n; }
            tb[base + k] = tb[base + k] + Bx[p];
        }
        int nz = 0;
        for (int k = 0; k < n; ++k) { const double r = ta[base + k] - tb[base + k]; ta[base + k] = r; nz += r != 0.0 ? 1 : 0; }
        call[i] = n;
        cnt[i] = nz;
    }
}

__global__ __launch_bounds__(BLK) void sub_general_emit_kernel(int m, const int *Ap, const int *Bp, const int *tj, const double *tr,
                                                               const int *call, const int *Cp, int *Cj, double *Cx)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        const int base = Ap[i] + Bp[i];
        int w = Cp[i];
        for (int k = call[i] - 1; k >= 0; --k)
            if (tr[base + k] != 0.0) { Cj[w] = tj[base + k]; Cx[w] = tr[base + k]; ++w; }
    }
}

// C = A - B for the scalar views of two BSR matrices with R x C blocks: SciPy's bsr_binop_bsr_general with minus
// (sparsetools/bsr.h) -- per BLOCK row the blocks of A, then of B, are accumulated per block column (duplicates summed per
// operand) in order of first touch; a result block is kept when ANY of its R*C entries is not exactly zero and the kept
// blocks are emitted in REVERSE order of first touch.  (Its canonical twin, used when both operands have sorted
// duplicate-free block rows, emits ascending columns: emit_sorted.)  One lane per block row; tb[] holds the block
// columns, tv[] the R*C running differences.  Scalar row ib*R + r stores its blocks one after another, C entries each.
__global__ __launch_bounds__(BLK) void sub_bsr_kernel(int mb, int R, int C, const int *Ap, const int *Aj, const double *Ax, const int *Bp,
                                                      const int *Bj, const double *Bx, int *tb, double *tv, int *call, int *cnt,
                                                      unsigned *dup)
{
    for (int ib = blockIdx.x * BLK + threadIdx.x; ib < mb; ib += gridDim.x * BLK) {
        const int a0 = Ap[ib * R], b0 = Bp[ib * R];
        const int na = (Ap[ib * R + 1] - a0) / C, nb = (Bp[ib * R + 1] - b0) / C;
        const int base = (a0 + b0) / (R * C);                              // blocks before this block row in A and B together
        const int64_t vb = (int64_t)base * R * C;
        int n = 0;
        for (int side = 0; side < 2; ++side) {
            const int *Xp = side ? Bp : Ap, *Xj = side ? Bj : Aj;
            const double *Xx = side ? Bx : Ax;
            const int nq = side ? nb : na;
            for (int q = 0; q < nq; ++q) {
                const int j = Xj[Xp[ib * R] + q * C] / C;
                int k = 0;
                while (k < n && (tb[base + k] & 0x3FFFFFFF) != j) ++k;
                if (k == n) {
                    tb[base + n] = j;
                    for (int e = 0; e < R * C; ++e) tv[vb + (int64_t)n * R * C + e] = 0.0;
                    ++n;
                }
                if (side) {
                    // SciPy forms (sum of A's blocks) - (sum of B's blocks): subtracting B's blocks one by one from the
                    // running difference is the same arithmetic as long as B holds a block column once per block row
                    if (tb[base + k] & 0x40000000) atomicOr(dup, 1u);
                    tb[base + k] |= 0x40000000;
                }
                double *dst = tv + vb + (int64_t)k * R * C;
                for (int r = 0; r < R; ++r)
                    for (int c = 0; c < C; ++c) {
                        const double v = Xx[Xp[ib * R + r] + q * C + c];
                        dst[r * C + c] = side ? dst[r * C + c] - v : dst[r * C + c] + v;
                    }
            }
        }
        int kept = 0;
        for (int k = 0; k < n; ++k) {
            bool any = false;
            for (int e = 0; e < R * C; ++e) any = any || tv[vb + (int64_t)k * R * C + e] != 0.0;
            kept += any ? 1 : 0;
        }
        call[ib] = n;
        for (int r = 0; r < R; ++r) cnt[ib * R + r] = kept * C;
    }
}

__global__ __launch_bounds__(BLK) void sub_bsr_emit_kernel(int mb, int R, int C, const int *Ap, const int *Bp, const int *tb, const double *tv,
                                                           const int *call, int emit_sorted, const int *Cp, int *Cj, double *Cx)
{
    for (int ib = blockIdx.x * BLK + threadIdx.x; ib < mb; ib += gridDim.x * BLK) {
        const int base = (Ap[ib * R] + Bp[ib * R]) / (R * C);
        const int64_t vb = (int64_t)base * R * C;
        const int n = call[ib];
        int w = 0;
        int last = -1;
        for (int t = 0; t < n; ++t) {
            int k;
            if (emit_sorted) {                           // next block column above `last`
                k = -1;
                for (int u = 0; u < n; ++u) {
                    const int ju = tb[base + u] & 0x3FFFFFFF;
                    if (ju > last && (k < 0 || ju < (tb[base + k] & 0x3FFFFFFF))) k = u;
                }
                last = tb[base + k] & 0x3FFFFFFF;
            } else {
                k = n - 1 - t;
            }
            bool any = false;
            for (int e = 0; e < R * C; ++e) any = any || tv[vb + (int64_t)k * R * C + e] != 0.0;
            if (!any) continue;
            for (int r = 0; r < R; ++r)
                for (int c = 0; c < C; ++c) {
                    const int o = Cp[ib * R + r] + w * C + c;
                    Cj[o] = (tb[base + k] & 0x3FFFFFFF) * C + c;
                    Cx[o] = tv[vb + (int64_t)k * R * C + r * C + c];
                }
            ++w;
        }
    }
}

// every R consecutive scalar rows hold the same number of entries, a multiple of C, with block-aligned column runs
__global__ __launch_bounds__(BLK) void bsr_shape_kernel(int mb, int R, int C, const int *Ap, const int *Aj, unsigned *flag)
{
    for (int ib = blockIdx.x * BLK + threadIdx.x; ib < mb; ib += gridDim.x * BLK) {
        const int len = Ap[ib * R + 1] - Ap[ib * R];
        bool bad = len % C != 0;
        for (int r = 1; r < R; ++r) bad = bad || (Ap[ib * R + r + 1] - Ap[ib * R + r]) != len;
        int prev = -1;
        bool unsorted = false;
        for (int q = 0; q < len / C && !bad; ++q) {
            const int j0 = Aj[Ap[ib * R] + q * C];
            bad = bad || j0 % C != 0;
            unsorted = unsorted || j0 / C <= prev;
            prev = j0 / C;
        }
        if (bad) atomicOr(flag, 1u);
        if (unsorted) atomicOr(flag, 2u);
    }
}

__global__ __launch_bounds__(BLK) void canonical_kernel(int m, const int *Ap, const int *Aj, unsigned *flag)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        bool bad = false;
        for (int p = Ap[i] + 1; p < Ap[i + 1]; ++p) bad = bad || Aj[p - 1] >= Aj[p];
        if (bad) atomicOr(flag, 1u);
    }
}

// ---- amg_core::symmetric_strength_of_connection (smoothed_aggregation.h:56-110) followed by what strength.py:343-348
// does to its result: magnitudes, every row scaled by the reciprocal of its largest entry
__global__ __launch_bounds__(BLK) void strength_diag_kernel(int m, const int *Ap, const int *Aj, const double *Ax, double *diag)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        double d = 0.0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) if (Aj[p] == i) d += Ax[p];     // duplicates are summed
        diag[i] = fabs(d);
    }
}

template <bool FILL>
__global__ __launch_bounds__(BLK) void strength_kernel(int m, double theta, const int *Ap, const int *Aj, const double *Ax,
                                                       const double *diag, int *cnt, const int *Sp, int *Sj, double *Sx)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        const double eps = theta * theta * diag[i];
        int w = FILL ? Sp[i] : 0;
        double big = 2.2250738585072014e-308;                 // maximum_row_value starts from numeric_limits<F>::min() (ruge_stuben.h:238)
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            const double a = Ax[p];
            if (j == i || a * a >= eps * diag[j]) {
                if (FILL) { Sj[w] = j; Sx[w] = fabs(a); big = big < fabs(a) ? fabs(a) : big; }
                ++w;
            }
        }
        if (!FILL) cnt[i] = w;
        else if (big != 0.0) {
            const double r = 1.0 / big;
            for (int q = Sp[i]; q < w; ++q) Sx[q] = Sx[q] * r;
        }
    }
}

__global__ __launch_bounds__(BLK) void scale_rows_kernel(int m, const int *Ap, double *Ax, double *diag, const double *d)
{
    for (int i = blockIdx.x * BLK + threadIdx.x; i < m; i += gridDim.x * BLK) {
        const double s = d[i];
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) Ax[p] = Ax[p] * s;
        if (diag) diag[i] = diag[i] * s;
    }
}

__global__ __launch_bounds__(BLK) void scale_values_kernel(int64_t n, double alpha, double *x)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) x[i] = x[i] * alpha;
}

// ---------------------------------------------------------------------------------------------- Arnoldi
// vectors are stored by planes: [re | im] (im absent when P == 1)
template <int P>
__global__ __launch_bounds__(BLK) void arn_dot_kernel(int64_t n, const double *v, const double *w, double *partial)
{
    __shared__ double sm[2][BLK / 64];
    double re = 0.0, im = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const double vr = v[i], wr = w[i];
        if (P == 1) re += vr * wr;
        else {
            const double vi = v[n + i], wi = w[n + i];     // conj(v) * w
            re += vr * wr + vi * wi;
            im += vr * wi - vi * wr;
        }
    }
    re = wsum(re); im = wsum(im);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = re; sm[1][wv] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, q = 0.0;
        for (int k = 0; k < BLK / 64; ++k) { r += sm[0][k]; q += sm[1][k]; }
        partial[2 * blockIdx.x] = r; partial[2 * blockIdx.x + 1] = q;
    }
}

// mode 0: out = (re, im); mode 1: out = (sqrt(re), 0)   [norm]; the result also goes to h (an entry of H)
__global__ __launch_bounds__(BLK) void arn_reduce_kernel(const double *partial, int n, int mode, double *out, double *h)
{
    __shared__ double sm[2][BLK / 64];
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < n; i += BLK) { re += partial[2 * i]; im += partial[2 * i + 1]; }
    re = wsum(re); im = wsum(im);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wv] = re; sm[1][wv] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, q = 0.0;
        for (int k = 0; k < BLK / 64; ++k) { r += sm[0][k]; q += sm[1][k]; }
        if (mode == 1) { r = sqrt(r); q = 0.0; }
        out[0] = r; out[1] = q;
        if (h) { h[0] = r; h[1] = q; }
    }
}

// w = w - h * v   (h read from device memory)
template <int P>
__global__ __launch_bounds__(BLK) void arn_axpy_kernel(int64_t n, const double *h, const double *v, double *w)
{
    const double hr = h[0], hi = h[1];
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        if (P == 1) { const double t = hr * v[i]; w[i] = w[i] - t; }
        else {
            const double vr = v[i], vi = v[n + i];
            const double tr = hr * vr - hi * vi, ti = hr * vi + hi * vr;
            w[i] = w[i] - tr;
            w[n + i] = w[n + i] - ti;
        }
    }
}

// w = w / s   (s read from device memory; every plane)
__global__ __launch_bounds__(BLK) void arn_div_kernel(int64_t total, const double *s, double *w)
{
    const double d = s[0];
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLK) w[i] = w[i] / d;
}

struct ArnCoef { double re[32], im[32]; };

// out = sum_k coef_k V_k, k < ncols (complex arithmetic when either side has an imaginary plane)
__global__ __launch_bounds__(BLK) void arn_combine_kernel(int64_t n, int ncols, int pv, int po, const double *V, int64_t stride,
                                                          const ArnCoef c, double *out)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        double re = 0.0, im = 0.0;
        for (int k = 0; k < ncols; ++k) {
            const double vr = V[(int64_t)k * stride + i], vi = pv == 2 ? V[(int64_t)k * stride + n + i] : 0.0;
            re += vr * c.re[k] - vi * c.im[k];
            im += vr * c.im[k] + vi * c.re[k];
        }
        out[i] = re;
        if (po == 2) out[n + i] = im;
    }
}

int grid_for(int64_t n, int cap = 8192) { return (int)std::min<int64_t>(cap, std::max<int64_t>(1, (n + BLK - 1) / BLK)); }

template <typename F>
void par_for(int n, F fn)
{
    const int hw = (int)std::max(1u, std::min(32u, pamg::host_cpus()));
    const int nt = n < (1 << 16) ? 1 : hw;
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int chunk = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo < hi) th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

int ensure_host_index(pamg_csr_s *A)
{
    if (A->h_p.empty()) {
        A->h_p.resize((size_t)A->m + 1);
        PAMG_HIP(hipMemcpy(A->h_p.data(), A->d_p, sizeof(int) * ((size_t)A->m + 1), hipMemcpyDeviceToHost));
    }
    if (A->h_j.size() != (size_t)A->nnz) {
        A->h_j.resize((size_t)A->nnz);
        if (A->nnz) PAMG_HIP(hipMemcpy(A->h_j.data(), A->d_j, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost));
    }
    return PAMG_OK;
}

int new_csr(int64_t m, int64_t n, int64_t nnz, pamg_csr_s **out)
{
    pamg_csr_s *C = new (std::nothrow) pamg_csr_s();
    if (!C) return PAMG_E_ALLOC;
    C->m = m; C->n = n; C->nnz = nnz;
    hipError_t e = hipMalloc((void **)&C->d_p, sizeof(int) * ((size_t)m + 1 + 8));
    if (e == hipSuccess) e = hipMalloc((void **)&C->d_j, sizeof(int) * ((size_t)nnz + 8));
    if (e == hipSuccess) e = hipMalloc((void **)&C->d_x, sizeof(double) * ((size_t)nnz + 8));
    if (e != hipSuccess) { hipFree(C->d_p); hipFree(C->d_j); hipFree(C->d_x); delete C; return (int)e; }
    *out = C;
    return PAMG_OK;
}

// row pointer from per-row counts (host prefix sum; m ints each way -- noise next to the product itself)
int counts_to_ptr(int m, const int *d_cnt, std::vector<int> &hp, int64_t &nnz)
{
    std::vector<int> cnt((size_t)m);
    if (m) PAMG_HIP(hipMemcpy(cnt.data(), d_cnt, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost));
    hp.assign((size_t)m + 1, 0);
    int64_t acc = 0;
    for (int i = 0; i < m; ++i) { acc += cnt[i]; if (acc > INT_MAX) return PAMG_E_UNSUPPORTED; hp[(size_t)i + 1] = (int)acc; }
    nnz = acc;
    return PAMG_OK;
}

size_t spg_lds(bool numeric, int cap = SPG_CAP)
{
    return (size_t)(numeric ? 16 : 8) * cap + sizeof(double) * BLK + sizeof(int) * (size_t)(4 * BLK + 2 + 2 * SPG_ROWS + 2);
}

int matmat(pamg_csr_s *A, pamg_csr_s *B, int col_block, int keep_zeros, pamg_csr_s **out)
{
    if (!A || !B || !out || col_block < 1) return PAMG_E_ARG;
    if (A->n != B->m) return PAMG_E_ARG;
    const int m = (int)A->m;
    // the attribute belongs to the (function, device) pair: remember the devices it was set on
    static std::mutex attr_mu;
    static unsigned long long attr_devs = 0;
    int cur_dev = 0;
    PAMG_HIP(hipGetDevice(&cur_dev));
    std::lock_guard<std::mutex> attr_lk(attr_mu);
    const bool attr_set = cur_dev < 64 && ((attr_devs >> cur_dev) & 1ull);
    if (!attr_set) {
        PAMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spg_kernel<true, SPG_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spg_lds(true)));
        PAMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spg_kernel<false, SPG_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spg_lds(false)));
        PAMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spg_kernel<true, SPG_CAP2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spg_lds(true, SPG_CAP2)));
        PAMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spg_kernel<false, SPG_CAP2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spg_lds(false, SPG_CAP2)));
        if (cur_dev < 64) attr_devs |= 1ull << cur_dev;
    }
    int *d_nprod = nullptr, *d_rowcount = nullptr, *d_rcount = nullptr, *d_obase = nullptr, *d_seq = nullptr, *d_long = nullptr,
        *d_lohi = nullptr, *d_ids = nullptr, *d_cnt2 = nullptr;
    int4 *d_tasks = nullptr;
    unsigned *d_flags = nullptr;
    pamg_csr_s *C = nullptr, *C2 = nullptr;
    int st = PAMG_OK;
    auto cleanup = [&]() {
        hipFree(d_nprod); hipFree(d_rowcount); hipFree(d_rcount); hipFree(d_obase); hipFree(d_tasks); hipFree(d_flags); hipFree(d_cnt2);
        hipFree(d_seq); hipFree(d_long); hipFree(d_lohi); hipFree(d_ids);
    };
#define SPG_CHECK(expr) do { st = (int)(expr); if (st) { cleanup(); if (C) pamg_csr_destroy(C); if (C2) pamg_csr_destroy(C2); return st; } } while (0)
    SPG_CHECK(hipMalloc((void **)&d_nprod, sizeof(int) * ((size_t)m + 1)));
    SPG_CHECK(hipMalloc((void **)&d_rowcount, sizeof(int) * ((size_t)m + 1)));
    SPG_CHECK(hipMalloc((void **)&d_flags, sizeof(unsigned) * 4));
    SPG_CHECK(hipMemset(d_rowcount, 0, sizeof(int) * ((size_t)m + 1)));
    SPG_CHECK(hipMemset(d_flags, 0, sizeof(unsigned) * 4));
    std::vector<int> nprod((size_t)m);
    if (m) {
        hipLaunchKernelGGL(spg_count_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_p, A->d_j, B->d_p, d_nprod);
        SPG_CHECK(hipGetLastError());
        SPG_CHECK(hipMemcpy(nprod.data(), d_nprod, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost));
    }
    std::vector<int> long_rows, lohi;                    // rows handled window by window, and the span of their product columns
    for (int r = 0; r < m; ++r) if (nprod[r] > SPG_CAP2) long_rows.push_back(r);
    const int nlong = (int)long_rows.size();
    if (nlong) {
        for (int r : long_rows) if (nprod[r] == INT_MAX) SPG_CHECK(PAMG_E_UNSUPPORTED);       // sequence numbers are 31-bit
        SPG_CHECK(hipMalloc((void **)&d_long, sizeof(int) * (size_t)nlong));
        SPG_CHECK(hipMalloc((void **)&d_lohi, sizeof(int) * 2 * (size_t)nlong + 16));
        SPG_CHECK(hipMemcpy(d_long, long_rows.data(), sizeof(int) * (size_t)nlong, hipMemcpyHostToDevice));
        for (int off = 0; off < nlong; off += (1 << 22)) {
            hipLaunchKernelGGL(spg_minmax_kernel, dim3(std::min(1 << 22, nlong - off)), dim3(BLK), 0, 0, (const int *)d_long + off, A->d_p, A->d_j, B->d_p,
                               B->d_j, d_lohi + 2 * off);
            SPG_CHECK(hipGetLastError());
        }
        lohi.resize(2 * (size_t)nlong);
        SPG_CHECK(hipMemcpy(lohi.data(), d_lohi, sizeof(int) * 2 * (size_t)nlong, hipMemcpyDeviceToHost));
        // a batch holds whole rows of B: the longest one must fit
        SPG_CHECK(hipMemset(d_lohi, 0, sizeof(int)));
        hipLaunchKernelGGL(max_rowlen_kernel, dim3(grid_for(B->m)), dim3(BLK), 0, 0, (int)B->m, B->d_p, d_lohi);
        SPG_CHECK(hipGetLastError());
        int maxlen = 0;
        SPG_CHECK(hipMemcpy(&maxlen, d_lohi, sizeof(int), hipMemcpyDeviceToHost));
        if (maxlen > SPL_CAP) SPG_CHECK(PAMG_E_UNSUPPORTED);
    }
    std::vector<int4> tasks;
    {
        std::vector<SpgTask> plan;
        spg_plan(m, nprod, lohi, plan);                  // pamg_spg_plan.h
        tasks.reserve(plan.size());
        for (const SpgTask &t : plan) tasks.push_back(make_int4(t.row0, t.row1, t.col0, t.col1));
    }
    const int nt = (int)tasks.size();
    std::vector<int> ids_short, ids_big, ids_long;        // whole rows packed to SPG_CAP / one row of up to SPG_CAP2 products / column windows
    for (int t = 0; t < nt; ++t) {
        const bool whole = tasks[t].w == INT_MAX && tasks[t].z == 0;
        const bool big = whole && tasks[t].y - tasks[t].x == 1 && nprod[(size_t)tasks[t].x] > SPG_CAP;
        (big ? ids_big : whole ? ids_short : ids_long).push_back(t);
    }
    const int ns = (int)ids_short.size(), nb = (int)ids_big.size(), nl = (int)ids_long.size();
    SPG_CHECK(hipMalloc((void **)&d_tasks, sizeof(int4) * ((size_t)nt + 1)));
    SPG_CHECK(hipMalloc((void **)&d_rcount, sizeof(int) * ((size_t)nt + 1)));
    SPG_CHECK(hipMalloc((void **)&d_obase, sizeof(int) * ((size_t)nt + 1)));
    SPG_CHECK(hipMalloc((void **)&d_ids, sizeof(int) * ((size_t)nt + 1)));
    if (nt) SPG_CHECK(hipMemcpy(d_tasks, tasks.data(), sizeof(int4) * (size_t)nt, hipMemcpyHostToDevice));
    if (ns) SPG_CHECK(hipMemcpy(d_ids, ids_short.data(), sizeof(int) * (size_t)ns, hipMemcpyHostToDevice));
    if (nb) SPG_CHECK(hipMemcpy(d_ids + ns, ids_big.data(), sizeof(int) * (size_t)nb, hipMemcpyHostToDevice));
    if (nl) SPG_CHECK(hipMemcpy(d_ids + ns + nb, ids_long.data(), sizeof(int) * (size_t)nl, hipMemcpyHostToDevice));
    SpgArgs a;
    a.ranges = d_tasks; a.ids = d_ids; a.Ap = A->d_p; a.Aj = A->d_j; a.Ax = A->d_x; a.Bp = B->d_p; a.Bj = B->d_j; a.Bx = B->d_x;
    a.rowcount = d_rowcount; a.rcount = d_rcount; a.obase = d_obase; a.Cj = nullptr; a.Cx = nullptr; a.Cseq = nullptr; a.flags = d_flags;
    a.cb = col_block; a.keep = keep_zeros ? 1 : 0;
    SpgArgs ab = a, al = a;
    ab.ids = d_ids + ns;
    al.ids = d_ids + ns + nb;
    // a launch may not exceed 2^32 threads (grid x block): coarse Galerkin products have tens of millions of window
    // tasks (measured: 33 M at 384^3, and a single launch silently ran only the first 2^24), so tasks go out in slices
    auto launch_tasks = [&](bool numeric) -> int {
        constexpr int SLICE = SPG_SLICE;
        for (int off = 0; off < ns; off += SLICE) {
            SpgArgs s1 = a;
            s1.ids = a.ids + off;
            if (numeric) hipLaunchKernelGGL((spg_kernel<true, SPG_CAP>), dim3(std::min(SLICE, ns - off)), dim3(BLK), spg_lds(true), 0, s1);
            else hipLaunchKernelGGL((spg_kernel<false, SPG_CAP>), dim3(std::min(SLICE, ns - off)), dim3(BLK), spg_lds(false), 0, s1);
        }
        for (int off = 0; off < nb; off += SLICE) {
            SpgArgs s1 = ab;
            s1.ids = ab.ids + off;
            if (numeric) hipLaunchKernelGGL((spg_kernel<true, SPG_CAP2>), dim3(std::min(SLICE, nb - off)), dim3(BLK), spg_lds(true, SPG_CAP2), 0, s1);
            else hipLaunchKernelGGL((spg_kernel<false, SPG_CAP2>), dim3(std::min(SLICE, nb - off)), dim3(BLK), spg_lds(false, SPG_CAP2), 0, s1);
        }
        for (int off = 0; off < nl; off += SLICE) {
            SpgArgs s1 = al;
            s1.ids = al.ids + off;
            if (numeric) hipLaunchKernelGGL((spg_long_kernel<true>), dim3(std::min(SLICE, nl - off)), dim3(BLK), 0, 0, s1);
            else hipLaunchKernelGGL((spg_long_kernel<false>), dim3(std::min(SLICE, nl - off)), dim3(BLK), 0, 0, s1);
        }
        return (int)hipGetLastError();
    };
    SPG_CHECK(launch_tasks(false));
    std::vector<int> hp;
    int64_t nnz = 0;
    SPG_CHECK(counts_to_ptr(m, d_rowcount, hp, nnz));
    std::vector<int> rc((size_t)nt), ob((size_t)nt + 1, 0);
    if (nt) SPG_CHECK(hipMemcpy(rc.data(), d_rcount, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
    unsigned flags[4] = {0, 0, 0, 0};
    SPG_CHECK(hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost));
    if (flags[1]) SPG_CHECK(PAMG_E_STATE);
    for (int k = 0; k < nt; ++k) ob[(size_t)k + 1] = ob[k] + rc[k];
    if (ob[nt] != nnz) SPG_CHECK(PAMG_E_STATE);
    SPG_CHECK(new_csr(m, B->n, nnz, &C));
    SPG_CHECK(hipMemcpy(C->d_p, hp.data(), sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice));
    if (nt) SPG_CHECK(hipMemcpy(d_obase, ob.data(), sizeof(int) * (size_t)nt, hipMemcpyHostToDevice));
    if (nl) SPG_CHECK(hipMalloc((void **)&d_seq, sizeof(int) * ((size_t)nnz + 8)));
    a.Cj = ab.Cj = al.Cj = C->d_j; a.Cx = ab.Cx = al.Cx = C->d_x; al.Cseq = d_seq;
    SPG_CHECK(launch_tasks(true));
    for (int off = 0; off < nlong; off += (1 << 22)) {
        hipLaunchKernelGGL(spg_reorder_kernel, dim3((unsigned)std::min(1 << 22, nlong - off)), dim3(BLK), 0, 0, (const int *)d_long + off, (const int *)C->d_p,
                           C->d_j, C->d_x, d_seq);
        SPG_CHECK(hipGetLastError());
    }
    SPG_CHECK(hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost));
    if (flags[1]) SPG_CHECK(PAMG_E_STATE);
    C->h_p = hp;
    if (flags[0]) {
        // SciPy does not store sums that are exactly zero: squeeze them out (order kept)
        SPG_CHECK(hipMalloc((void **)&d_cnt2, sizeof(int) * ((size_t)m + 1)));
        hipLaunchKernelGGL(rownz_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, C->d_p, C->d_x, d_cnt2);
        SPG_CHECK(hipGetLastError());
        std::vector<int> hp2;
        int64_t nnz2 = 0;
        SPG_CHECK(counts_to_ptr(m, d_cnt2, hp2, nnz2));
        SPG_CHECK(new_csr(m, B->n, nnz2, &C2));
        SPG_CHECK(hipMemcpy(C2->d_p, hp2.data(), sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(compact_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, C->d_p, C->d_j, C->d_x, C2->d_p, C2->d_j, C2->d_x);
        SPG_CHECK(hipGetLastError());
        SPG_CHECK(hipDeviceSynchronize());
        C2->h_p = hp2;
        pamg_csr_destroy(C);
        C = C2; C2 = nullptr;
    }
    SPG_CHECK(hipDeviceSynchronize());
#undef SPG_CHECK
    cleanup();
    *out = C;
    return PAMG_OK;
}

// rows sorted by column without duplicates?  (SciPy's has_canonical_format; decides which binop algorithm it runs)
int is_canonical(pamg_csr_s *A, bool &canon)
{
    if (A->canon < 0) {
        unsigned *d_flag = nullptr;
        PAMG_HIP(hipMalloc((void **)&d_flag, sizeof(unsigned)));
        int st = (int)hipMemset(d_flag, 0, sizeof(unsigned));
        if (!st && A->m) {
            hipLaunchKernelGGL(canonical_kernel, dim3(grid_for(A->m)), dim3(BLK), 0, 0, (int)A->m, A->d_p, A->d_j, d_flag);
            st = (int)hipGetLastError();
        }
        unsigned f = 0;
        if (!st) st = (int)hipMemcpy(&f, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost);
        hipFree(d_flag);
        if (st) return st;
        A->canon = f ? 0 : 1;
    }
    canon = A->canon == 1;
    return PAMG_OK;
}

int subtract(pamg_csr_s *A, pamg_csr_s *B, pamg_csr_s **out)
{
    if (!A || !B || !out || A->m != B->m || A->n != B->n) return PAMG_E_ARG;
    const int m = (int)A->m;
    bool ca = false, cb = false;
    PAMG_TRY(is_canonical(A, ca));
    PAMG_TRY(is_canonical(B, cb));
    const bool canonical = ca && cb;
    int *d_cnt = nullptr, *d_all = nullptr, *tj = nullptr;
    double *ta = nullptr, *tb = nullptr;
    pamg_csr_s *C = nullptr;
    int st = PAMG_OK;
    auto cleanup = [&]() { hipFree(d_cnt); hipFree(d_all); hipFree(tj); hipFree(ta); hipFree(tb); };
#define SUB_CHECK(expr) do { st = (int)(expr); if (st) { cleanup(); if (C) pamg_csr_destroy(C); return st; } } while (0)
    SUB_CHECK(hipMalloc((void **)&d_cnt, sizeof(int) * ((size_t)m + 1)));
    if (canonical) {
        hipLaunchKernelGGL((sub_kernel<false>), dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_p, A->d_j, A->d_x, B->d_p, B->d_j, B->d_x,
                           d_cnt, (const int *)nullptr, (int *)nullptr, (double *)nullptr);
    } else {
        const size_t cap = (size_t)A->nnz + (size_t)B->nnz + 8;
        SUB_CHECK(hipMalloc((void **)&d_all, sizeof(int) * ((size_t)m + 1)));
        SUB_CHECK(hipMalloc((void **)&tj, sizeof(int) * cap));
        SUB_CHECK(hipMalloc((void **)&ta, sizeof(double) * cap));
        SUB_CHECK(hipMalloc((void **)&tb, sizeof(double) * cap));
        hipLaunchKernelGGL(sub_general_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_p, A->d_j, A->d_x, B->d_p, B->d_j, B->d_x,
                           tj, ta, tb, d_all, d_cnt);
    }
    SUB_CHECK(hipGetLastError());
    std::vector<int> hp;
    int64_t nnz = 0;
    SUB_CHECK(counts_to_ptr(m, d_cnt, hp, nnz));
    SUB_CHECK(new_csr(m, A->n, nnz, &C));
    SUB_CHECK(hipMemcpy(C->d_p, hp.data(), sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice));
    if (canonical) {
        hipLaunchKernelGGL((sub_kernel<true>), dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_p, A->d_j, A->d_x, B->d_p, B->d_j, B->d_x,
                           (int *)nullptr, C->d_p, C->d_j, C->d_x);
        C->canon = 1;
    } else {
        hipLaunchKernelGGL(sub_general_emit_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_p, B->d_p, (const int *)tj, (const double *)ta,
                           (const int *)d_all, (const int *)C->d_p, C->d_j, C->d_x);
    }
    SUB_CHECK(hipGetLastError());
    SUB_CHECK(hipDeviceSynchronize());
#undef SUB_CHECK
    cleanup();
    C->h_p = hp;
    *out = C;
    return PAMG_OK;
}

// A - B on the scalar views of two BSR matrices with R x C blocks (see sub_bsr_kernel)
int subtract_bsr(pamg_csr_s *A, pamg_csr_s *B, int R, int C, pamg_csr_s **out)
{
    if (!A || !B || !out || A->m != B->m || A->n != B->n || R < 1 || C < 1 || A->m % R || A->n % C) return PAMG_E_ARG;
    if (R == 1 && C == 1) return subtract(A, B, out);
    const int m = (int)A->m, mb = m / R;
    int *d_cnt = nullptr, *d_all = nullptr, *tb = nullptr;
    unsigned *d_flag = nullptr;
    double *tv = nullptr;
    pamg_csr_s *Cm = nullptr;
    int st = PAMG_OK;
    auto cleanup = [&]() { hipFree(d_cnt); hipFree(d_all); hipFree(tb); hipFree(tv); hipFree(d_flag); };
#define SUB_CHECK(expr) do { st = (int)(expr); if (st) { cleanup(); if (Cm) pamg_csr_destroy(Cm); return st; } } while (0)
    // both operands must really be scalar views of R x C blocks; sorted duplicate-free block rows in both -> SciPy's canonical merge
    SUB_CHECK(hipMalloc((void **)&d_flag, 2 * sizeof(unsigned)));
    SUB_CHECK(hipMemset(d_flag, 0, 2 * sizeof(unsigned)));
    if (mb) {
        hipLaunchKernelGGL(bsr_shape_kernel, dim3(grid_for(mb)), dim3(BLK), 0, 0, mb, R, C, A->d_p, A->d_j, d_flag);
        hipLaunchKernelGGL(bsr_shape_kernel, dim3(grid_for(mb)), dim3(BLK), 0, 0, mb, R, C, B->d_p, B->d_j, d_flag + 1);
    }
    unsigned fl[2] = {0, 0};
    SUB_CHECK(hipMemcpy(fl, d_flag, sizeof(fl), hipMemcpyDeviceToHost));
    if ((fl[0] | fl[1]) & 1u) { cleanup(); return PAMG_E_ARG; }
    const int emit_sorted = ((fl[0] | fl[1]) & 2u) ? 0 : 1;
    const size_t nblocks = ((size_t)A->nnz + (size_t)B->nnz) / ((size_t)R * C) + 8;
    SUB_CHECK(hipMalloc((void **)&d_cnt, sizeof(int) * ((size_t)m + 1)));
    SUB_CHECK(hipMalloc((void **)&d_all, sizeof(int) * ((size_t)mb + 1)));
    SUB_CHECK(hipMalloc((void **)&tb, sizeof(int) * nblocks));
    SUB_CHECK(hipMalloc((void **)&tv, sizeof(double) * nblocks * R * C));
    SUB_CHECK(hipMemset(d_flag, 0, sizeof(unsigned)));
    if (mb) hipLaunchKernelGGL(sub_bsr_kernel, dim3(grid_for(mb)), dim3(BLK), 0, 0, mb, R, C, A->d_p, A->d_j, A->d_x, B->d_p, B->d_j, B->d_x,
                               tb, tv, d_all, d_cnt, d_flag);
    SUB_CHECK(hipGetLastError());
    SUB_CHECK(hipMemcpy(fl, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (fl[0]) { cleanup(); return PAMG_E_UNSUPPORTED; }           // duplicate block columns in a block row of B
    std::vector<int> hp;
    int64_t nnz = 0;
    SUB_CHECK(counts_to_ptr(m, d_cnt, hp, nnz));
    SUB_CHECK(new_csr(m, A->n, nnz, &Cm));
    SUB_CHECK(hipMemcpy(Cm->d_p, hp.data(), sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice));
    if (mb) hipLaunchKernelGGL(sub_bsr_emit_kernel, dim3(grid_for(mb)), dim3(BLK), 0, 0, mb, R, C, A->d_p, B->d_p, (const int *)tb, (const double *)tv,
                               (const int *)d_all, emit_sorted, (const int *)Cm->d_p, Cm->d_j, Cm->d_x);
    SUB_CHECK(hipGetLastError());
    SUB_CHECK(hipDeviceSynchronize());
#undef SUB_CHECK
    cleanup();
    Cm->h_p = hp;
    *out = Cm;
    return PAMG_OK;
}

}  // namespace

int csr_device_arrays(pamg_csr_s *A, CsrArrays *out)
{
    if (!A || !out) return PAMG_E_ARG;
    out->m = A->m; out->n = A->n; out->nnz = A->nnz; out->p = A->d_p; out->j = A->d_j; out->x = A->d_x;
    return PAMG_OK;
}
}  // namespace pamg

using namespace pamg;

extern "C" {

int pamg_csr_create(pamg_csr_t *out, int64_t m, int64_t n, const int32_t *Ap, const int32_t *Aj, const double *Ax)
{
    if (!out || !Ap || m < 0 || n < 0 || m > (1 << 30) || n > (1 << 30) || Ap[0] != 0) return PAMG_E_ARG;
    const int64_t nnz = Ap[m];
    if (nnz < 0 || (nnz > 0 && (!Aj || !Ax))) return PAMG_E_ARG;
    {
        std::atomic<int> bad(0);
        host_parallel(m, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) if (Ap[i + 1] < Ap[i]) bad = 1; });
        host_parallel(nnz, [&](int64_t lo, int64_t hi) { int b = 0; for (int64_t p = lo; p < hi; ++p) b |= (Aj[p] < 0) | (Aj[p] >= n); if (b) bad = 1; });
        if (bad.load()) return PAMG_E_ARG;
    }
    pamg_csr_s *C = nullptr;
    PAMG_TRY(new_csr(m, n, nnz, &C));
    int st = (int)hipMemcpy(C->d_p, Ap, sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice);
    if (!st && nnz) st = (int)hipMemcpy(C->d_j, Aj, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice);
    if (!st && nnz) st = (int)hipMemcpy(C->d_x, Ax, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice);
    if (st) { pamg_csr_destroy(C); return st; }
    C->h_p.assign(Ap, Ap + m + 1);                        // the column ids stay on the device (ensure_host_index fetches them if a plan ever asks)
    *out = C;
    return PAMG_OK;
}

int pamg_csr_view(pamg_csr_t *out, pamg_matrix_t A)
{
    if (!out || !A) return PAMG_E_ARG;
    if (A->dtype != PAMG_F64) return PAMG_E_UNSUPPORTED;
    pamg_csr_s *C = new (std::nothrow) pamg_csr_s();
    if (!C) return PAMG_E_ALLOC;
    C->m = A->nrows; C->n = A->ncols; C->nnz = A->nnz;
    C->d_p = A->d_Ap; C->d_j = A->d_Aj; C->d_x = (double *)A->d_Ax;
    C->owns = false;
    *out = C;
    return PAMG_OK;
}

int pamg_csr_destroy(pamg_csr_t A)
{
    if (!A) return PAMG_OK;
    if (A->owns) { hipFree(A->d_p); hipFree(A->d_j); hipFree(A->d_x); }
    delete A;
    return PAMG_OK;
}

int pamg_csr_info(pamg_csr_t A, int64_t info[4])
{
    if (!A || !info) return PAMG_E_ARG;
    info[0] = A->m; info[1] = A->n; info[2] = A->nnz; info[3] = A->owns ? 1 : 0;
    return PAMG_OK;
}

int pamg_csr_download(pamg_csr_t A, int32_t *Ap, int32_t *Aj, double *Ax)
{
    if (!A || !Ap) return PAMG_E_ARG;
    PAMG_HIP(hipMemcpy(Ap, A->d_p, sizeof(int) * ((size_t)A->m + 1), hipMemcpyDeviceToHost));
    if (A->nnz && Aj) PAMG_HIP(hipMemcpy(Aj, A->d_j, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost));
    if (A->nnz && Ax) PAMG_HIP(hipMemcpy(Ax, A->d_x, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToHost));
    return PAMG_OK;
}

int pamg_csr_matmat(pamg_csr_t A, pamg_csr_t B, int col_block, int keep_zeros, pamg_csr_t *C)
{
    return matmat(A, B, col_block, keep_zeros, C);
}

int pamg_csr_scale(pamg_csr_t A, double alpha)
{
    if (!A) return PAMG_E_ARG;
    if (!A->owns) return PAMG_E_STATE;                   // a view: scale the operator it points into instead
    if (A->nnz) hipLaunchKernelGGL(scale_values_kernel, dim3(grid_for(A->nnz)), dim3(BLK), 0, 0, A->nnz, alpha, A->d_x);
    PAMG_HIP(hipGetLastError());
    PAMG_HIP(hipDeviceSynchronize());
    return PAMG_OK;
}

int pamg_csr_subtract(pamg_csr_t A, pamg_csr_t B, pamg_csr_t *C) { return subtract(A, B, C); }
int pamg_csr_subtract_bsr(pamg_csr_t A, pamg_csr_t B, int R, int C, pamg_csr_t *out) { return subtract_bsr(A, B, R, C, out); }

int pamg_csr_strength_symmetric(pamg_csr_t A, double theta, pamg_csr_t *out)
{
    if (!A || !out || A->m != A->n || !(theta >= 0.0)) return PAMG_E_ARG;
    const int m = (int)A->m;
    double *d_diag = nullptr;
    int *d_cnt = nullptr;
    pamg_csr_s *S = nullptr;
    PAMG_HIP(hipMalloc((void **)&d_diag, sizeof(double) * ((size_t)m + 1)));
    int st = (int)hipMalloc((void **)&d_cnt, sizeof(int) * ((size_t)m + 1));
    std::vector<int> hp;
    int64_t nnz = 0;
    if (!st && m) {
        hipLaunchKernelGGL(strength_diag_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_p, A->d_j, A->d_x, d_diag);
        hipLaunchKernelGGL((strength_kernel<false>), dim3(grid_for(m)), dim3(BLK), 0, 0, m, theta, A->d_p, A->d_j, A->d_x, (const double *)d_diag,
                           d_cnt, (const int *)nullptr, (int *)nullptr, (double *)nullptr);
        st = (int)hipGetLastError();
    }
    if (!st) st = counts_to_ptr(m, d_cnt, hp, nnz);
    if (!st) st = new_csr(m, A->n, nnz, &S);
    if (!st) st = (int)hipMemcpy(S->d_p, hp.data(), sizeof(int) * ((size_t)m + 1), hipMemcpyHostToDevice);
    if (!st && m) {
        hipLaunchKernelGGL((strength_kernel<true>), dim3(grid_for(m)), dim3(BLK), 0, 0, m, theta, A->d_p, A->d_j, A->d_x, (const double *)d_diag,
                           (int *)nullptr, (const int *)S->d_p, S->d_j, S->d_x);
        st = (int)hipGetLastError();
    }
    if (!st) st = (int)hipDeviceSynchronize();
    hipFree(d_diag); hipFree(d_cnt);
    if (st) { if (S) pamg_csr_destroy(S); return st; }
    S->h_p = hp;
    *out = S;
    return PAMG_OK;
}

// a block (R x C > 1) operator whose values are rescaled in place keeps its scalar view only -- what SpMV, Arnoldi and the
// sparse products read; the block arrays the block smoothers use would go stale and are released
static void drop_block_view(pamg_matrix_s *A)
{
    if (!A->d_bAp) return;
    hipFree(A->d_bAp); hipFree(A->d_bAj); hipFree(A->d_bAjf); hipFree(A->d_bdiag); hipFree(A->d_bAx); hipFree(A->d_bmeta);
    A->d_bAp = A->d_bAj = A->d_bAjf = A->d_bdiag = nullptr;
    A->d_bAx = nullptr; A->d_bmeta = nullptr; A->bnblk = 0;
}

int pamg_matrix_scale_rows(pamg_matrix_t A, const double *d)
{
    if (!A || !d) return PAMG_E_ARG;
    if (A->dtype != PAMG_F64) return PAMG_E_UNSUPPORTED;
    if (A->borrowed) return PAMG_E_STATE;
    for (int k = 0; k < 4; ++k) if (A->gs[k] || A->ls[k]) return PAMG_E_STATE;   // schedules hold copies of the values
    drop_block_view(A);
    const int m = (int)A->nrows;
    double *dd = nullptr;
    PAMG_HIP(hipMalloc((void **)&dd, sizeof(double) * ((size_t)m + 1)));
    int st = (int)hipMemcpy(dd, d, sizeof(double) * (size_t)m, hipMemcpyHostToDevice);
    if (!st && m) {
        matrix_drop_value_codes(A);
        hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(m)), dim3(BLK), 0, 0, m, A->d_Ap, (double *)A->d_Ax, (double *)A->d_diag, dd);
        st = (int)hipGetLastError();
    }
    if (!st) st = (int)hipDeviceSynchronize();
    hipFree(dd);
    return st;
}

int pamg_matrix_scale_values(pamg_matrix_t A, double alpha)
{
    if (!A) return PAMG_E_ARG;
    if (A->dtype != PAMG_F64) return PAMG_E_UNSUPPORTED;
    if (A->borrowed) return PAMG_E_STATE;
    for (int k = 0; k < 4; ++k) if (A->gs[k] || A->ls[k]) return PAMG_E_STATE;
    drop_block_view(A);
    matrix_drop_value_codes(A);
    if (A->nnz) hipLaunchKernelGGL(scale_values_kernel, dim3(grid_for(A->nnz)), dim3(BLK), 0, 0, A->nnz, alpha, (double *)A->d_Ax);
    if (A->nrows && A->d_diag) hipLaunchKernelGGL(scale_values_kernel, dim3(grid_for(A->nrows)), dim3(BLK), 0, 0, A->nrows, alpha, (double *)A->d_diag);
    PAMG_HIP(hipGetLastError());
    PAMG_HIP(hipDeviceSynchronize());
    return PAMG_OK;
}

// ---------------------------------------------------------------------------------------------- Arnoldi (C ABI)
int pamg_arnoldi_create(pamg_arnoldi_t *out, pamg_matrix_t A, int maxiter)
{
    if (!out || !A || maxiter < 1 || maxiter > 31) return PAMG_E_ARG;
    if (A->dtype != PAMG_F64 || A->nrows != A->ncols) return PAMG_E_UNSUPPORTED;
    pamg_arnoldi_s *h = new (std::nothrow) pamg_arnoldi_s();
    if (!h) return PAMG_E_ALLOC;
    h->A = A; h->n = A->nrows; h->maxiter = (int)std::min<int64_t>(maxiter, A->nrows);
    hipError_t e = hipMalloc((void **)&h->d_v0, sizeof(double) * 2 * (size_t)std::max<int64_t>(1, h->n));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_H, sizeof(double) * 2 * (size_t)(h->maxiter + 1) * (size_t)h->maxiter + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_part, sizeof(double) * (2 * ARN_GRID + 8));
    if (e != hipSuccess) { pamg_arnoldi_destroy(h); return (int)e; }
    A->borrowed++;
    *out = h;
    return PAMG_OK;
}

int pamg_arnoldi_destroy(pamg_arnoldi_t h)
{
    if (!h) return PAMG_OK;
    if (h->A && h->A->borrowed > 0 && h->d_part) h->A->borrowed--;
    hipFree(h->d_V); hipFree(h->d_v0); hipFree(h->d_H); hipFree(h->d_part);
    delete h;
    return PAMG_OK;
}

// One Arnoldi process of up to maxiter steps (util/linalg.py:154-253, the non-symmetric branch -- the only one
// approximate_spectral_radius uses).  Start vector: (v0_re, v0_im) from the host (v0_im may be null), or the vector the
// last pamg_arnoldi_combine left on the device when v0_re is null.  H: host, (maxiter + 1) x maxiter complex entries
// (re, im pairs, row-major).  ncols = columns of H that are valid (j + 1 of the reference), breakdown_flag as there.
int pamg_arnoldi_run(pamg_arnoldi_t h, const double *v0_re, const double *v0_im, double breakdown, double *H, int *ncols,
                     int *breakdown_flag)
{
    if (!h || !H || !ncols || !breakdown_flag) return PAMG_E_ARG;
    const int64_t n = h->n;
    const int m = h->maxiter;
    hipStream_t s = 0;
    int P;
    if (v0_re) {
        P = v0_im ? 2 : 1;
        PAMG_HIP(hipMemcpy(h->d_v0, v0_re, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        if (v0_im) PAMG_HIP(hipMemcpy(h->d_v0 + n, v0_im, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        h->v0_planes = P;
    } else {
        if (!h->v0_planes) return PAMG_E_STATE;
        P = h->v0_planes;
    }
    if (h->planes < P) {
        hipFree(h->d_V); h->d_V = nullptr; h->planes = 0;
        PAMG_HIP(hipMalloc((void **)&h->d_V, sizeof(double) * (size_t)(m + 1) * (size_t)P * (size_t)std::max<int64_t>(1, n)));
        h->planes = P;
    }
    const int64_t stride = (int64_t)P * n;             // basis vector k starts at d_V + k * stride
    const int64_t tot = stride;
    const int g = grid_for(n, ARN_GRID), gt = grid_for(tot);
    double *red = h->d_part + 2 * ARN_GRID;            // reduced scalar (re, im)
    PAMG_HIP(hipMemsetAsync(h->d_H, 0, sizeof(double) * 2 * (size_t)(m + 1) * (size_t)m, s));
    PAMG_HIP(hipMemcpyAsync(h->d_V, h->d_v0, sizeof(double) * (size_t)tot, hipMemcpyDeviceToDevice, s));
#define ARN_DOT(v, w) do { if (P == 1) hipLaunchKernelGGL((arn_dot_kernel<1>), dim3(g), dim3(BLK), 0, s, n, (const double *)(v), (const double *)(w), h->d_part); \
                            else hipLaunchKernelGGL((arn_dot_kernel<2>), dim3(g), dim3(BLK), 0, s, n, (const double *)(v), (const double *)(w), h->d_part); } while (0)
    // v0 /= norm(v0)
    ARN_DOT(h->d_V, h->d_V);
    hipLaunchKernelGGL(arn_reduce_kernel, dim3(1), dim3(BLK), 0, s, (const double *)h->d_part, g, 1, red, (double *)nullptr);
    hipLaunchKernelGGL(arn_div_kernel, dim3(gt), dim3(BLK), 0, s, tot, (const double *)red, h->d_V);
    PAMG_HIP(hipGetLastError());
    for (int j = 0; j < m; ++j) {
        double *vj = h->d_V + (int64_t)j * stride, *w = h->d_V + (int64_t)(j + 1) * stride;
        for (int p = 0; p < P; ++p) PAMG_TRY(stream_launch(h->A, EPI_SET, vj + (int64_t)p * n, nullptr, w + (int64_t)p * n, 0.0, 0.0, nullptr, s));
        for (int i = 0; i <= j; ++i) {
            const double *vi = h->d_V + (int64_t)i * stride;
            double *hij = h->d_H + 2 * ((size_t)i * (size_t)m + (size_t)j);
            ARN_DOT(vi, w);
            hipLaunchKernelGGL(arn_reduce_kernel, dim3(1), dim3(BLK), 0, s, (const double *)h->d_part, g, 0, red, hij);
            if (P == 1) hipLaunchKernelGGL((arn_axpy_kernel<1>), dim3(g), dim3(BLK), 0, s, n, (const double *)hij, vi, w);
            else hipLaunchKernelGGL((arn_axpy_kernel<2>), dim3(g), dim3(BLK), 0, s, n, (const double *)hij, vi, w);
        }
        double *hn = h->d_H + 2 * ((size_t)(j + 1) * (size_t)m + (size_t)j);
        ARN_DOT(w, w);
        hipLaunchKernelGGL(arn_reduce_kernel, dim3(1), dim3(BLK), 0, s, (const double *)h->d_part, g, 1, red, hn);
        hipLaunchKernelGGL(arn_div_kernel, dim3(gt), dim3(BLK), 0, s, tot, (const double *)hn, w);
        PAMG_HIP(hipGetLastError());
    }
#undef ARN_DOT
    PAMG_HIP(hipMemcpyAsync(H, h->d_H, sizeof(double) * 2 * (size_t)(m + 1) * (size_t)m, hipMemcpyDeviceToHost, s));
    PAMG_HIP(hipStreamSynchronize(s));
    // the reference leaves the loop at the first sub-diagonal entry below the breakdown tolerance; the steps run
    // beyond it here touched nothing that is read afterwards
    *breakdown_flag = 0;
    int nc = m;
    for (int j = 0; j < m; ++j) {
        const double sub = H[2 * ((size_t)(j + 1) * (size_t)m + (size_t)j)];
        if (!(sub >= breakdown)) { *breakdown_flag = 1; nc = j + 1; break; }
    }
    *ncols = nc;
    return PAMG_OK;
}

// next start vector (kept on the device) = V[:, :ncols] @ coef   (linalg.py:352: v0 = hstack(V[:-1]) @ evect[:, max_index])
int pamg_arnoldi_combine(pamg_arnoldi_t h, int ncols, const double *coef_re, const double *coef_im)
{
    if (!h || !coef_re || ncols < 1 || ncols > h->maxiter || !h->planes || !h->d_V) return PAMG_E_ARG;
    ArnCoef c;
    bool cplx = h->v0_planes == 2;
    for (int k = 0; k < 32; ++k) { c.re[k] = 0.0; c.im[k] = 0.0; }
    for (int k = 0; k < ncols; ++k) {
        c.re[k] = coef_re[k];
        c.im[k] = coef_im ? coef_im[k] : 0.0;
        if (coef_im) cplx = true;                       // a complex eigenvector makes a complex start vector, whatever its values
    }
    const int pv = h->v0_planes, po = cplx ? 2 : 1;
    const int64_t stride = (int64_t)pv * h->n;
    hipLaunchKernelGGL(arn_combine_kernel, dim3(grid_for(h->n)), dim3(BLK), 0, 0, h->n, ncols, pv, po, (const double *)h->d_V, stride, c, h->d_v0);
    PAMG_HIP(hipGetLastError());
    PAMG_HIP(hipDeviceSynchronize());
    h->v0_planes = po;
    return PAMG_OK;
}

// the start vector pamg_arnoldi_combine left on the device (return_vector=True); *planes = 1 (real) or 2
int pamg_arnoldi_vector(pamg_arnoldi_t h, double *re, double *im, int *planes)
{
    if (!h || !re || !planes || !h->v0_planes) return PAMG_E_ARG;
    PAMG_HIP(hipMemcpy(re, h->d_v0, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost));
    if (h->v0_planes == 2 && im) PAMG_HIP(hipMemcpy(im, h->d_v0 + h->n, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost));
    *planes = h->v0_planes;
    return PAMG_OK;
}

}  // extern "C"
