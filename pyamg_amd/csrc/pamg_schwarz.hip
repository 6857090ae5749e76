// pamg_schwarz.hip -- multiplicative overlapping Schwarz relaxation (gfx950 only).
//
// Reference: amg_core::overlapping_schwarz_csr, relaxation.h:1420-1492 -- for every subdomain, in order: the residual of
// its rows from the CURRENT x (row sums in stored order, `rsum -= a*x` entry by entry, `+= b` last), times the dense
// inverse of its diagonal block (row-major, sequential over k), added to x.  Subdomains overlap, so the sweep is
// sequential in the reference; here it becomes a dependency DAG over subdomains: d depends on an earlier d' when d'
// wrote something d reads or writes, or read something d writes.  Subdomains of one dependency level touch disjoint
// data and run side by side, in place.  Every subdomain is processed by one wave with the reference's arithmetic order, so
// results are bit-identical.
//
// Round 6: ONE persistent launch per sweep (schwarz_versioned_kernel).  A row is updated once by every subdomain that holds it;
// the plan gives every update its own SLOT (version v of row i = its v-th update of the sweep, single assignment, pre-filled with a
// sentinel) and tells every read which version it wants: the one the reference's sequential sweep would find.  G co-resident waves
// walk the subdomains in level order (wave w takes positions w, w + G, ...) and poll the slots they read until the datum is there
// -- the hand-off of the Gauss-Seidel sweeps: one L2 round trip per dependency level, no counters, no fences, and no write-after-
// read hazard (nothing is overwritten; x itself is only read during the sweep and takes the last versions in a closing launch).
// A first attempt with one completion counter per level (waves wait for the whole level before theirs) measured 11 us per level
// against 5 for the launches: the counter's atomics and the serial chain of static loads behind the wait
// (profiles/r06_microbench_schwarz_level_counters_not_kept.json).  The launch-per-level form stays as the always-live fallback
// (mode 1, and for schedules whose version table would not fit): a wave that polls too long raises the error word,
// pamg_solver_solve switches and reruns.
#include <algorithm>
#include <new>
#include <vector>

#include "pamg_common.h"
#include "pamg_schwarz_plan.h"

namespace pamg {
struct SchwarzSchedule {
    int start = 0, stop = 0, step = 0;
    int nlevels = 0;
    int m = 0;                        // subdomains visited
    int max_width = 0;                // subdomains of the widest level
    std::vector<int> level_ptr;       // [nlevels + 1] offsets into d_order
    int *d_order = nullptr;           // subdomains, level after level (sweep order inside a level)
    // the versioned form (persistent sweep); versioned == false: this schedule runs as one launch per level
    bool versioned = false;
    int64_t nslots = 0;               // updates of the sweep = slots of the hand-off buffer
    int *d_ebase = nullptr;           // [m + 1] first (position, local row) entry of every position of d_order
    int *d_wslot = nullptr;           // [E] slot this update writes
    int *d_prev = nullptr;            // [E] where the row's value before this update is: slot >= 0, or ~row = x itself
    int *d_roff = nullptr;            // [E] first byte of the row's read versions in d_rver
    unsigned char *d_rver = nullptr;  // [R] per stored entry of the row: 0 = x itself, v = version v of the column's row
    int *d_vbase = nullptr;           // [n] first slot of every row
    int *d_last = nullptr;            // [n] slot of the row's last version, -1 = not updated by this sweep
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
    void *d_xs = nullptr;             // hand-off buffer of the persistent sweep: one slot per update (largest schedule)
    int64_t xs_cap = 0;
    unsigned *d_err = nullptr;        // [1] a wave of the persistent sweep gave up waiting
    int mode = 0;                     // 0 = one persistent launch per sweep, 1 = one launch per dependency level
    int occ = 0;                      // co-resident workgroups per CU of the persistent kernel (queried once)
    int cus = 0;
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


// ---- the versioned persistent sweep: see the header
template <typename T> struct SwSentinel;
template <> struct SwSentinel<double> {
    using bits_t = unsigned long long;
    static constexpr bits_t value = 0x7FF8DEADBEEF5A5Aull;
    static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};
template <> struct SwSentinel<float> {
    using bits_t = unsigned int;
    static constexpr bits_t value = 0x7FC5BEEFu;
    static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};

template <typename T>
__global__ __launch_bounds__(256) void schwarz_fill_kernel(T *xs, int64_t n)
{
    using B = typename SwSentinel<T>::bits_t;
    B *p = reinterpret_cast<B *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = SwSentinel<T>::value;
}

// x takes the last version of every row the sweep updated
template <typename T>
__global__ __launch_bounds__(256) void schwarz_close_kernel(const T *xs, const int *last, T *x, int n)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int sl = last[i];
        if (sl >= 0) x[i] = xs[sl];
    }
}

constexpr int SW_CH = 8;              // reads in flight per lane
constexpr int SW_PRE = 8;             // rows of the inverted block kept in registers up to this size

template <typename T>
__global__ __launch_bounds__(SW_THREADS) void schwarz_versioned_kernel(const int *order, const int *ebase, const int *wslot, const int *prev, const int *roff,
                                                                       const unsigned char *rver, const int *vbase, int m, T *xs, unsigned *err, const int *Sp,
                                                                       const int *Sj, const int *Tp, const T *Tx, const int *Ap, const int *Aj, const T *Ax,
                                                                       const T *x, const T *b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *r = reinterpret_cast<T *>(smem);
    using S = SwSentinel<T>;
    for (int q = blockIdx.x; q < m; q += gridDim.x) {
        const int d = order[q], e0 = ebase[q];
        const int s0 = Sp[d], size = Sp[d + 1] - s0;
        if (size <= SW_THREADS) {
            // ---- one lane per row (the usual case: a subdomain is a row's pattern).  Everything that does not depend on another subdomain is
            //      loaded BEFORE the first poll -- the row's entries and version codes, b, the slot numbers, the row of the inverted block (pulled
            //      into L1), and a first look at the row's own previous version -- so that the chain behind the arrival of the last input is
            //      products, one LDS exchange, the small dense product and the store
            const int k = threadIdx.x;
            const bool act = k < size;
            const T *ti = Tx + Tp[d] + (size_t)(act ? k : 0) * size;
            int row = 0, p0 = 0, p1 = 0, pc = -1, ws = 0;
            const unsigned char *rv = rver;
            T bval = T(0), xo = T(0), warm = T(0);
            T tr[SW_PRE];
#pragma unroll
            for (int kk = 0; kk < SW_PRE; ++kk) tr[kk] = T(0);
            if (act) {
                row = Sj[s0 + k];
                p0 = Ap[row]; p1 = Ap[row + 1];
                rv = rver + roff[e0 + k];
                pc = prev[e0 + k];
                ws = wslot[e0 + k];
                bval = b[row];
                xo = pc < 0 ? x[~pc] : __hip_atomic_load(xs + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (size <= SW_PRE) {
#pragma unroll
                    for (int kk = 0; kk < SW_PRE; ++kk) tr[kk] = kk < size ? ti[kk] : T(0);      // the block row in registers
                } else {
                    for (int kk = 0; kk < size; ++kk) warm += ti[kk];                             // ... or at least in L1
                }
            }
            asm volatile("" ::"v"(warm));
            bool fail = false;
            bool xpend = act && pc >= 0 && S::bits(xo) == S::value;     // the row's own previous version is polled along with the operands
            T rsum = T(0);
            for (int p = p0; p < p1 && !fail; p += SW_CH) {
                T a[SW_CH], v[SW_CH];
                const T *ad[SW_CH];
                unsigned pend = 0;
#pragma unroll
                for (int c = 0; c < SW_CH; ++c) {
                    a[c] = T(0); v[c] = T(0); ad[c] = x;
                    if (p + c < p1) {
                        const int j = Aj[p + c];
                        const int ver = rv[p - p0 + c];
                        a[c] = Ax[p + c];
                        ad[c] = ver ? xs + (vbase[j] + ver - 1) : x + j;
                        if (ver) pend |= 1u << c;
                    }
                }
#pragma unroll
                for (int c = 0; c < SW_CH; ++c)
                    if (p + c < p1) v[c] = ((pend >> c) & 1u) ? __hip_atomic_load(ad[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ad[c];
#pragma unroll
                for (int c = 0; c < SW_CH; ++c)
                    if (((pend >> c) & 1u) && S::bits(v[c]) != S::value) pend &= ~(1u << c);
                unsigned spins = 0;
                while (pend) {
#pragma unroll
                    for (int c = 0; c < SW_CH; ++c)
                        if ((pend >> c) & 1u) {
                            v[c] = __hip_atomic_load(ad[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (S::bits(v[c]) != S::value) pend &= ~(1u << c);
                        }
                    if (xpend) {
                        xo = __hip_atomic_load(xs + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        xpend = S::bits(xo) == S::value;
                    }
                    if ((++spins & 255u) == 0 && (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
                }
#pragma unroll
                for (int c = 0; c < SW_CH; ++c)
                    if (p + c < p1) rsum -= a[c] * v[c];                  // stored order, like the reference's loop
            }
            rsum += bval;
            if (act) r[k] = rsum;
            if (__any(fail)) {
                if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __syncthreads();                                  // every residual is formed
            if (act) {
                T sum = T(0);
                if (size <= SW_PRE) {
#pragma unroll
                    for (int kk = 0; kk < SW_PRE; ++kk)
                        if (kk < size) sum += tr[kk] * r[kk];
                } else {
                    for (int kk = 0; kk < size; ++kk) sum += ti[kk] * r[kk];
                }
                if (pc >= 0) {
                    unsigned spins = 0;
                    while (S::bits(xo) == S::value) {
                        if ((++spins & 255u) == 0 && (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
                        xo = __hip_atomic_load(xs + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!fail) __hip_atomic_store(xs + ws, xo + sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__any(fail)) {
                if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __syncthreads();                                  // r is free for the next subdomain
            continue;
        }
        // ---- subdomains of more than 64 rows: lanes loop over the rows
        bool fail = false;
        for (int k = threadIdx.x; k < size && !fail; k += SW_THREADS) {
            const int row = Sj[s0 + k];
            const int p0 = Ap[row], p1 = Ap[row + 1];
            const unsigned char *rv = rver + roff[e0 + k];
            T rsum = T(0);
            for (int p = p0; p < p1 && !fail; p += SW_CH) {
                T a[SW_CH], v[SW_CH];
                const T *ad[SW_CH];
                unsigned pend = 0;
#pragma unroll
                for (int c = 0; c < SW_CH; ++c) {
                    a[c] = T(0); v[c] = T(0); ad[c] = x;
                    if (p + c < p1) {
                        const int j = Aj[p + c];
                        const int ver = rv[p - p0 + c];
                        a[c] = Ax[p + c];
                        ad[c] = ver ? xs + (vbase[j] + ver - 1) : x + j;
                        if (ver) pend |= 1u << c;
                    }
                }
#pragma unroll
                for (int c = 0; c < SW_CH; ++c)
                    if (p + c < p1) v[c] = ((pend >> c) & 1u) ? __hip_atomic_load(ad[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ad[c];
#pragma unroll
                for (int c = 0; c < SW_CH; ++c)
                    if (((pend >> c) & 1u) && S::bits(v[c]) != S::value) pend &= ~(1u << c);
                unsigned spins = 0;
                while (pend) {
#pragma unroll
                    for (int c = 0; c < SW_CH; ++c)
                        if ((pend >> c) & 1u) {
                            v[c] = __hip_atomic_load(ad[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (S::bits(v[c]) != S::value) pend &= ~(1u << c);
                        }
                    if ((++spins & 255u) == 0 && (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
                }
#pragma unroll
                for (int c = 0; c < SW_CH; ++c)
                    if (p + c < p1) rsum -= a[c] * v[c];                  // stored order, like the reference's loop
            }
            rsum += b[row];
            r[k] = rsum;
        }
        if (__any(fail)) {                                    // one wave per workgroup: leave together
            if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __syncthreads();                                      // every residual is formed
        const T *Tinv = Tx + Tp[d];
        for (int i = threadIdx.x; i < size && !fail; i += SW_THREADS) {
            T s = T(0);
            const T *ti = Tinv + (size_t)i * size;
            for (int k = 0; k < size; ++k) s += ti[k] * r[k];
            const int pc = prev[e0 + i];
            T xo;
            if (pc < 0) xo = x[~pc];
            else {
                unsigned spins = 0;
                xo = __hip_atomic_load(xs + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (S::bits(xo) == S::value) {
                    if ((++spins & 255u) == 0 && (spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
                    xo = __hip_atomic_load(xs + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (!fail) __hip_atomic_store(xs + wslot[e0 + i], xo + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__any(fail)) {
            if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __syncthreads();                                      // r is free for the next subdomain
    }
}

// the version table of a sweep (pamg_schwarz_plan.h) on the device.  Leaves g.versioned false (the schedule then runs as one launch per
// level) when the planner declines: a subdomain lists a row twice, a row is updated more than 255 times, tables beyond a gigabyte.
int upload_versions(pamg_schwarz_s *h, SchwarzSchedule &g, const SchwarzLevels &lv)
{
    const pamg_matrix_s *A = h->A;
    const int n = (int)A->nrows, m = g.m;
    g.versioned = false;
    for (void *p : {(void *)g.d_ebase, (void *)g.d_wslot, (void *)g.d_prev, (void *)g.d_roff, (void *)g.d_rver, (void *)g.d_vbase, (void *)g.d_last}) hipFree(p);
    g.d_ebase = g.d_wslot = g.d_prev = g.d_roff = g.d_vbase = g.d_last = nullptr;
    g.d_rver = nullptr;
    SchwarzVersions V;
    schwarz_versions(n, A->h_Ap.data(), A->h_Aj.data(), h->nsub, h->h_Sp.data(), h->h_Sj.data(), g.start, g.step, lv, V);
    if (!V.ok) return PAMG_OK;
    const int64_t E = V.nslots, R = V.nreads;
    auto up = [](auto **dp, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc((void **)dp, std::max<size_t>(bytes, 256));
        if (e == hipSuccess && bytes) e = hipMemcpy(*dp, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    PAMG_HIP(up(&g.d_ebase, V.ebase.data(), sizeof(int) * ((size_t)m + 1)));
    PAMG_HIP(up(&g.d_wslot, V.wslot.data(), sizeof(int) * (size_t)E));
    PAMG_HIP(up(&g.d_prev, V.prev.data(), sizeof(int) * (size_t)E));
    PAMG_HIP(up(&g.d_roff, V.roff.data(), sizeof(int) * (size_t)E));
    PAMG_HIP(up(&g.d_rver, V.rver.data(), (size_t)R));
    PAMG_HIP(up(&g.d_vbase, V.vbase.data(), sizeof(int) * (size_t)n));
    PAMG_HIP(up(&g.d_last, V.last.data(), sizeof(int) * (size_t)n));
    g.nslots = E;
    if (E > h->xs_cap) {
        hipFree(h->d_xs);
        h->d_xs = nullptr;
        PAMG_HIP(hipMalloc(&h->d_xs, std::max<size_t>((size_t)E * tsize(A->dtype), 256)));
        h->xs_cap = E;
    }
    h->bytes += sizeof(int) * (3 * (size_t)E + (size_t)m + 2 * (size_t)n) + (size_t)R;
    g.versioned = true;
    return PAMG_OK;
}

// dependency levels of the subdomains visited in (start, stop, step) order, and the version table of the persistent sweep
int build_schedule(pamg_schwarz_s *h, int start, int stop, int step, SchwarzSchedule &g)
{
    const pamg_matrix_s *A = h->A;
    SchwarzLevels lv;
    if (schwarz_levels((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), h->nsub, h->h_Sp.data(), h->h_Sj.data(), start, stop, step, lv)) return PAMG_E_ARG;
    g.start = start; g.stop = stop; g.step = step;
    g.level_ptr = lv.level_ptr;
    g.nlevels = lv.nlevels; g.m = lv.m; g.max_width = lv.max_width;
    if (lv.m == 0) return PAMG_OK;
    hipFree(g.d_order);
    g.d_order = nullptr;
    PAMG_HIP(hipMalloc((void **)&g.d_order, sizeof(int) * (size_t)lv.m));
    PAMG_HIP(hipMemcpy(g.d_order, lv.order.data(), sizeof(int) * (size_t)lv.m, hipMemcpyHostToDevice));
    if (!h->d_err) {
        PAMG_HIP(hipMalloc((void **)&h->d_err, 256));
        PAMG_HIP(hipMemset(h->d_err, 0, 256));
    }
    PAMG_TRY(upload_versions(h, g, lv));
    return PAMG_OK;
}

// workgroups of the persistent sweep: all of them must be running at once -- the occupancy the runtime reports for the kernel with
// this handle's LDS, one short of it per CU (another kernel's tail may still hold a slot), at most eight per CU, never more than
// the widest level
template <typename T>
int persistent_grid(pamg_schwarz_s *h, const SchwarzSchedule &g, size_t lds)
{
    if (!h->occ) {
        int dev = 0, occ = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, schwarz_versioned_kernel<T>, SW_THREADS, lds) != hipSuccess || occ < 1) return 0;
        h->occ = occ; h->cus = p.multiProcessorCount;
    }
    const int per_cu = std::max(1, std::min(8, h->occ - 1));
    // waves AHEAD of the running front have the static part of their subdomain (row lists, offsets, operator entries, version codes: a chain
    // of half a dozen dependent loads) behind them when their inputs arrive: two level widths of them (512^2 Poisson, widest level 256: 256 / 512 / 1 024 / 2 048 waves
    // 3.02 / 2.63 / 2.80 / 3.8 us per level)
    static const int want = [] { const char *e = getenv("PAMG_SCHWARZ_WAVES"); return e ? atoi(e) : 0; }();
    const int ahead = want > 0 ? want : 2 * g.max_width;
    return std::max(1, std::min({per_cu * h->cus, std::max(ahead, 256), g.m}));
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
    static const int force_levels = [] { const char *e = getenv("PAMG_SCHWARZ_LEVELS"); return (e && *e && *e != '0') ? 1 : 0; }();
    if (h->mode == 0 && !force_levels && g->nlevels > 1 && g->versioned) {
        const int grid = A->dtype == PAMG_F64 ? persistent_grid<double>(h, *g, lds) : persistent_grid<float>(h, *g, lds);
        if (grid > 0) {
            const int fill_grid = (int)std::min<int64_t>(4096, (g->nslots + 255) / 256), close_grid = (int)std::min<int64_t>(4096, ((int64_t)A->nrows + 255) / 256);
            if (A->dtype == PAMG_F64) {
                hipLaunchKernelGGL((schwarz_fill_kernel<double>), dim3(fill_grid), dim3(256), 0, s, (double *)h->d_xs, g->nslots);
                hipLaunchKernelGGL((schwarz_versioned_kernel<double>), dim3(grid), dim3(SW_THREADS), lds, s, (const int *)g->d_order, (const int *)g->d_ebase,
                                   (const int *)g->d_wslot, (const int *)g->d_prev, (const int *)g->d_roff, (const unsigned char *)g->d_rver, (const int *)g->d_vbase,
                                   g->m, (double *)h->d_xs, h->d_err, h->d_Sp, h->d_Sj, h->d_Tp, (const double *)h->d_Tx, A->d_Ap, A->d_Aj, (const double *)A->d_Ax,
                                   (const double *)x, (const double *)b);
                hipLaunchKernelGGL((schwarz_close_kernel<double>), dim3(close_grid), dim3(256), 0, s, (const double *)h->d_xs, (const int *)g->d_last, (double *)x, (int)A->nrows);
            } else {
                hipLaunchKernelGGL((schwarz_fill_kernel<float>), dim3(fill_grid), dim3(256), 0, s, (float *)h->d_xs, g->nslots);
                hipLaunchKernelGGL((schwarz_versioned_kernel<float>), dim3(grid), dim3(SW_THREADS), lds, s, (const int *)g->d_order, (const int *)g->d_ebase,
                                   (const int *)g->d_wslot, (const int *)g->d_prev, (const int *)g->d_roff, (const unsigned char *)g->d_rver, (const int *)g->d_vbase,
                                   g->m, (float *)h->d_xs, h->d_err, h->d_Sp, h->d_Sj, h->d_Tp, (const float *)h->d_Tx, A->d_Ap, A->d_Aj, (const float *)A->d_Ax,
                                   (const float *)x, (const float *)b);
                hipLaunchKernelGGL((schwarz_close_kernel<float>), dim3(close_grid), dim3(256), 0, s, (const float *)h->d_xs, (const int *)g->d_last, (float *)x, (int)A->nrows);
            }
            PAMG_HIP(hipGetLastError());
            return PAMG_OK;
        }
    }
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

// did a wave of a persistent sweep give up waiting since the last call?  (after a synchronising entry point; clears the word)
int schwarz_error(pamg_schwarz_s *h, bool *error)
{
    *error = false;
    if (!h || !h->d_err) return PAMG_OK;
    unsigned w = 0;
    PAMG_HIP(hipMemcpy(&w, h->d_err, sizeof(w), hipMemcpyDeviceToHost));
    if (w) {
        *error = true;
        PAMG_HIP(hipMemset(h->d_err, 0, sizeof(unsigned)));
    }
    return PAMG_OK;
}

// from now on: one launch per dependency level (no wave ever waits for another)
void schwarz_level_launches(pamg_schwarz_s *h) { if (h) h->mode = 1; }

// both sweep directions' schedules up front (a captured cycle must not allocate)
int schwarz_prepare(pamg_schwarz_s *h, int sweep)
{
    if (!h) return PAMG_E_ARG;
    if (sweep != PAMG_BACKWARD && !h->sched[0].d_order && h->nsub) PAMG_TRY(build_schedule(h, 0, h->nsub, 1, h->sched[0]));
    if (sweep != PAMG_FORWARD && !h->sched[1].d_order && h->nsub) PAMG_TRY(build_schedule(h, h->nsub - 1, -1, -1, h->sched[1]));
    // the occupancy query of the persistent kernel, outside any capture too
    const size_t lds = (size_t)std::max(1, h->max_size) * tsize(h->A->dtype);
    for (auto &c : h->sched)
        if (c.d_order) { if (h->A->dtype == PAMG_F64) persistent_grid<double>(h, c, lds); else persistent_grid<float>(h, c, lds); }
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
    for (auto &c : h->sched) {
        hipFree(c.d_order); hipFree(c.d_ebase); hipFree(c.d_wslot); hipFree(c.d_prev); hipFree(c.d_roff); hipFree(c.d_rver); hipFree(c.d_vbase); hipFree(c.d_last);
    }
    hipFree(h->d_xs); hipFree(h->d_err);
    delete h;
    return PAMG_OK;
}

int pamg_schwarz_sweep(pamg_schwarz_t h, void *x, const void *b, int row_start, int row_stop, int row_step, pamg_stream_t s)
{
    return schwarz_sweep(h, x, b, row_start, row_stop, row_step, (hipStream_t)s);
}

int pamg_schwarz_set_mode(pamg_schwarz_t h, int mode)
{
    if (!h || (mode != 0 && mode != 1)) return PAMG_E_ARG;
    h->mode = mode;
    return PAMG_OK;
}

int pamg_schwarz_error(pamg_schwarz_t h, int *error)
{
    if (!h || !error) return PAMG_E_ARG;
    PAMG_HIP(hipDeviceSynchronize());
    bool e = false;
    PAMG_TRY(schwarz_error(h, &e));
    *error = e ? 1 : 0;
    return PAMG_OK;
}

int pamg_schwarz_info(pamg_schwarz_t h, int64_t info[4])
{
    if (!h || !info) return PAMG_E_ARG;
    info[0] = h->nsub; info[1] = h->max_size; info[2] = h->sched[0].nlevels; info[3] = h->sched[1].nlevels;
    return PAMG_OK;
}

}  // extern "C"
