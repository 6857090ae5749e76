// pamg_schwarz.hip -- multiplicative overlapping Schwarz relaxation (gfx950 only).
//
// Reference: amg_core::overlapping_schwarz_csr, relaxation.h:1420-1492 -- for every subdomain, in order: the residual of
// its rows from the CURRENT x (row sums in stored order, `rsum -= a*x` entry by entry, `+= b` last), times the dense
// inverse of its diagonal block (row-major, sequential over k), added to x.  Subdomains overlap, so the sweep is
// sequential in the reference; here it becomes a dependency DAG over subdomains: d depends on an earlier d' when d'
// wrote something d reads or writes, or read something d writes.  Subdomains of one dependency level touch disjoint
// data and run side by side, in place; levels run one launch after another (inside a captured cycle they are graph
// nodes).  Every subdomain is processed by one wave with the reference's arithmetic order, so results are bit-identical.
#include <algorithm>
#include <new>
#include <vector>

#include "pamg_common.h"

namespace pamg {
struct SchwarzSchedule {
    int start = 0, stop = 0, step = 0;
    int nlevels = 0;
    std::vector<int> level_ptr;       // [nlevels + 1] offsets into d_order
    int *d_order = nullptr;           // subdomains, level after level (sweep order inside a level)
};
}  // namespace pamg

struct pamg_schwarz_s {
    pamg_matrix_s *A = nullptr;       // borrowed: the operator the reference sweeps with (lvl.Acsr: CSR, sorted rows)
    int nsub = 0;
    int max_size = 0;
    std::vector<int> h_Sp, h_Sj;
    int *d_Sp = nullptr, *d_Sj = nullptr, *d_Tp = nullptr;
    void *d_Tx = nullptr;
    pamg::SchwarzSchedule sched[2];
    size_t bytes = 0;
};

namespace pamg {
namespace {

constexpr int SW_THREADS = 64;        // one wave per subdomain
constexpr int SW_MAX = 2048;          // rows per subdomain (LDS: one residual each)

template <typename T>
__global__ __launch_bounds__(SW_THREADS) void schwarz_level_kernel(const int *order, int first, const int *Sp, const int *Sj,
                                                                   const int *Tp, const T *Tx, const int *Ap, const int *Aj,
                                                                   const T *Ax, T *x, const T *b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *r = reinterpret_cast<T *>(smem);
    const int d = order[first + blockIdx.x];
    const int s0 = Sp[d], size = Sp[d + 1] - s0;
    for (int q = threadIdx.x; q < size; q += SW_THREADS) {
        const int row = Sj[s0 + q];
        T rsum = T(0);
        for (int p = Ap[row]; p < Ap[row + 1]; ++p) rsum -= Ax[p] * x[Aj[p]];
        rsum += b[row];
        r[q] = rsum;
    }
    __syncthreads();                                      // every residual is formed before x changes
    const T *Tinv = Tx + Tp[d];
    for (int i = threadIdx.x; i < size; i += SW_THREADS) {
        T s = T(0);
        const T *ti = Tinv + (size_t)i * size;
        for (int k = 0; k < size; ++k) s += ti[k] * r[k];
        const int row = Sj[s0 + i];
        x[row] = x[row] + s;
    }
}

// dependency levels of the subdomains visited in (start, stop, step) order
int build_schedule(pamg_schwarz_s *h, int start, int stop, int step, SchwarzSchedule &g)
{
    const pamg_matrix_s *A = h->A;
    const int n = (int)A->nrows;
    if (step == 0) return PAMG_E_ARG;
    const long span = (long)stop - start;
    if (span % step != 0 || span / step < 0) return PAMG_E_ARG;
    const int m = (int)(span / step);
    g.start = start; g.stop = stop; g.step = step;
    g.level_ptr.assign(1, 0);
    g.nlevels = 0;
    if (m == 0) return PAMG_OK;
    if (start < 0 || start >= h->nsub || start + (long)(m - 1) * step < 0 || start + (long)(m - 1) * step >= h->nsub) return PAMG_E_ARG;
    std::vector<int> lastW((size_t)n, -1), lastR((size_t)n, -1), lvl((size_t)m, 0);
    int maxl = 0;
    for (int t = 0; t < m; ++t) {
        const int d = start + t * step;
        int L = 0;
        for (int q = h->h_Sp[d]; q < h->h_Sp[d + 1]; ++q) {
            const int row = h->h_Sj[q];
            L = std::max(L, std::max(lastW[row], lastR[row]) + 1);                 // write after write / write after read
            for (int p = A->h_Ap[row]; p < A->h_Ap[row + 1]; ++p) L = std::max(L, lastW[A->h_Aj[p]] + 1);   // read after write
        }
        lvl[t] = L;
        maxl = std::max(maxl, L);
        for (int q = h->h_Sp[d]; q < h->h_Sp[d + 1]; ++q) {
            const int row = h->h_Sj[q];
            lastW[row] = std::max(lastW[row], L);
            for (int p = A->h_Ap[row]; p < A->h_Ap[row + 1]; ++p) { const int j = A->h_Aj[p]; lastR[j] = std::max(lastR[j], L); }
        }
    }
    g.nlevels = maxl + 1;
    g.level_ptr.assign((size_t)g.nlevels + 1, 0);
    for (int t = 0; t < m; ++t) g.level_ptr[(size_t)lvl[t] + 1]++;
    for (int l = 0; l < g.nlevels; ++l) g.level_ptr[(size_t)l + 1] += g.level_ptr[l];
    std::vector<int> order((size_t)m), cur(g.level_ptr.begin(), g.level_ptr.end() - 1);
    for (int t = 0; t < m; ++t) order[(size_t)cur[lvl[t]]++] = start + t * step;
    hipFree(g.d_order);
    g.d_order = nullptr;
    PAMG_HIP(hipMalloc((void **)&g.d_order, sizeof(int) * (size_t)m));
    PAMG_HIP(hipMemcpy(g.d_order, order.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
    return PAMG_OK;
}

}  // namespace

int schwarz_sweep(pamg_schwarz_s *h, void *x, const void *b, int start, int stop, int step, hipStream_t s)
{
    if (!h || !x || !b) return PAMG_E_ARG;
    if (start == stop) return PAMG_OK;
    SchwarzSchedule *g = nullptr;
    for (auto &c : h->sched)
        if (c.d_order && c.start == start && c.stop == stop && c.step == step) g = &c;
    if (!g) {
        g = h->sched[0].d_order ? &h->sched[1] : &h->sched[0];
        PAMG_TRY(build_schedule(h, start, stop, step, *g));
    }
    const pamg_matrix_s *A = h->A;
    const size_t lds = (size_t)std::max(1, h->max_size) * tsize(A->dtype);
    for (int l = 0; l < g->nlevels; ++l) {
        const int first = g->level_ptr[l], count = g->level_ptr[l + 1] - first;
        if (count <= 0) continue;
        if (A->dtype == PAMG_F64)
            hipLaunchKernelGGL((schwarz_level_kernel<double>), dim3(count), dim3(SW_THREADS), lds, s, (const int *)g->d_order, first, h->d_Sp, h->d_Sj,
                               h->d_Tp, (const double *)h->d_Tx, A->d_Ap, A->d_Aj, (const double *)A->d_Ax, (double *)x, (const double *)b);
        else
            hipLaunchKernelGGL((schwarz_level_kernel<float>), dim3(count), dim3(SW_THREADS), lds, s, (const int *)g->d_order, first, h->d_Sp, h->d_Sj,
                               h->d_Tp, (const float *)h->d_Tx, A->d_Ap, A->d_Aj, (const float *)A->d_Ax, (float *)x, (const float *)b);
        PAMG_HIP(hipGetLastError());
    }
    return PAMG_OK;
}

// both sweep directions' schedules up front (a captured cycle must not allocate)
int schwarz_prepare(pamg_schwarz_s *h, int sweep)
{
    if (!h) return PAMG_E_ARG;
    if (sweep != PAMG_BACKWARD && !h->sched[0].d_order && h->nsub) PAMG_TRY(build_schedule(h, 0, h->nsub, 1, h->sched[0]));
    if (sweep != PAMG_FORWARD && !h->sched[1].d_order && h->nsub) PAMG_TRY(build_schedule(h, h->nsub - 1, -1, -1, h->sched[1]));
    return PAMG_OK;
}

}  // namespace pamg

using namespace pamg;

extern "C" {

int pamg_schwarz_create(pamg_schwarz_t *out, pamg_matrix_t A, int nsub, const int32_t *Sp, const int32_t *Sj, const int32_t *Tp,
                        const void *Tx)
{
    if (!out || !A || nsub < 0 || !Sp || !Tp || Sp[0] != 0) return PAMG_E_ARG;
    if (A->R != 1 || A->C != 1 || A->nrows != A->ncols) return PAMG_E_UNSUPPORTED;
    const int n = (int)A->nrows;
    const int64_t ns = Sp[nsub];
    if (ns < 0 || (ns > 0 && (!Sj || !Tx))) return PAMG_E_ARG;
    int maxs = 0;
    for (int d = 0; d < nsub; ++d) {
        const int size = Sp[d + 1] - Sp[d];
        if (size < 0 || (int64_t)Tp[d + 1] - Tp[d] != (int64_t)size * size) return PAMG_E_ARG;
        maxs = std::max(maxs, size);
    }
    for (int64_t q = 0; q < ns; ++q) if (Sj[q] < 0 || Sj[q] >= n) return PAMG_E_ARG;
    if (maxs > SW_MAX) return PAMG_E_UNSUPPORTED;
    pamg_schwarz_s *h = new (std::nothrow) pamg_schwarz_s();
    if (!h) return PAMG_E_ALLOC;
    h->A = A; h->nsub = nsub; h->max_size = maxs;
    h->h_Sp.assign(Sp, Sp + nsub + 1);
    h->h_Sj.assign(Sj, Sj + ns);
    const size_t ts = tsize(A->dtype);
    const size_t nt = (size_t)Tp[nsub];
    hipError_t e = hipMalloc((void **)&h->d_Sp, sizeof(int) * ((size_t)nsub + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_Tp, sizeof(int) * ((size_t)nsub + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_Sj, std::max<size_t>(sizeof(int) * (size_t)ns, 256));
    if (e == hipSuccess) e = hipMalloc(&h->d_Tx, std::max<size_t>(nt * ts, 256));
    if (e == hipSuccess) e = hipMemcpy(h->d_Sp, Sp, sizeof(int) * ((size_t)nsub + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_Tp, Tp, sizeof(int) * ((size_t)nsub + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && ns) e = hipMemcpy(h->d_Sj, Sj, sizeof(int) * (size_t)ns, hipMemcpyHostToDevice);
    if (e == hipSuccess && nt) e = hipMemcpy(h->d_Tx, Tx, nt * ts, hipMemcpyHostToDevice);
    if (e != hipSuccess) { pamg_schwarz_destroy(h); return (int)e; }
    h->bytes = sizeof(int) * (2 * ((size_t)nsub + 1) + (size_t)ns) + nt * ts;
    *out = h;
    return PAMG_OK;
}

int pamg_schwarz_destroy(pamg_schwarz_t h)
{
    if (!h) return PAMG_OK;
    hipFree(h->d_Sp); hipFree(h->d_Sj); hipFree(h->d_Tp); hipFree(h->d_Tx);
    for (auto &c : h->sched) hipFree(c.d_order);
    delete h;
    return PAMG_OK;
}

int pamg_schwarz_sweep(pamg_schwarz_t h, void *x, const void *b, int row_start, int row_stop, int row_step, pamg_stream_t s)
{
    return schwarz_sweep(h, x, b, row_start, row_stop, row_step, (hipStream_t)s);
}

int pamg_schwarz_info(pamg_schwarz_t h, int64_t info[4])
{
    if (!h || !info) return PAMG_E_ARG;
    info[0] = h->nsub; info[1] = h->max_size; info[2] = h->sched[0].nlevels; info[3] = h->sched[1].nlevels;
    return PAMG_OK;
}

}  // extern "C"
