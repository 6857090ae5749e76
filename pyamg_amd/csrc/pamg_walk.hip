// pamg_walk.hip -- the LINE-WALK form of the fast-order Gauss-Seidel / SOR sweep (layout and reasoning: pamg_walk_plan.h).
//
// The lane form (pamg_lane.hip) with one row per wave is bound by the hand-off alone, ~1 us per dependency level of ROWS;
// operators whose rows are numbered along lines (the coarse operators of smoothed aggregation on grids) have deep row-level
// schedules because every row waits for the row before it.  Here a wave owns a whole LINE (a run of consecutively visited
// rows each coupled to its predecessor) and walks it: 64 lanes share the row, K entries per lane, DPP butterfly -- the lane
// form's arithmetic -- but the predecessor's new value stays in a register, so along the line a dependency costs
//     v_t = (b_t - (s_t + a_{t,t-1} v_{t-1})) * (1 / a_tt)          (s_t: the butterfly sum of the OTHER products)
// four dependent flops behind v_{t-1}; s_t does not depend on v_{t-1} and is ready before it.  Operands from other lines are
// polled in the sentinel hand-off buffer like everywhere else.  The static operands of the next RING rows and the operands
// (b, x, first poll) of the next AHEAD rows are in flight while a row is finished: the walk costs issue slots, not round
// trips.  Same rows, same order, same products as amg_core::gauss_seidel (relaxation.h:48-76) / sor_gauss_seidel (:116-145):
// the reference's iterates up to rounding (fast order).  Lines are dealt out statically in the order of their level over the
// line graph: deadlock-free with all waves resident.
#include "pamg_common.h"
#include "pamg_walk_plan.h"

namespace pamg {

struct WalkSched {
    int K = 0;
    int64_t nrows = 0, nlines = 0;
    int nlevels = 0;
    int *d_cols = nullptr, *d_rid = nullptr, *d_line_row = nullptr;
    void *d_vals = nullptr, *d_rdiag = nullptr, *d_afwd = nullptr;
    int64_t max_level_lines = 0, n_forward = 0;
    int last_grid = 0;
    size_t bytes = 0;
};

template <typename T> struct WSentinel;
template <> struct WSentinel<double> {
    using bits_t = unsigned long long;
    static constexpr bits_t value = 0x7FF8DEADBEEF5A5Aull;
    static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};
template <> struct WSentinel<float> {
    using bits_t = unsigned int;
    static constexpr bits_t value = 0x7FC5BEEFu;
    static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};

template <typename T>
__global__ __launch_bounds__(BLK) void walk_fill_sentinel_kernel(T *xs, int64_t n)
{
    using B = typename WSentinel<T>::bits_t;
    B *p = reinterpret_cast<B *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) p[i] = WSentinel<T>::value;
}

template <typename T>
struct WalkArgs {
    const int *cols;
    const T *vals;
    const int *rid;
    const T *rdiag, *afwd;
    const int *line_row;
    const T *x;            // OLD values (x itself, or its snapshot for structurally non-symmetric patterns)
    T *y;                  // destination (the live x)
    T *xs;                 // hand-off buffer, sentinel-filled
    const T *b;
    unsigned *err;
    int nlines, nidle;
    T omega;
};

// ---- sum over the 64 lanes; every lane ends up with the total (the butterfly of pamg_lane.hip)
template <int CTRL>
__device__ __forceinline__ double walk_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float walk_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ double walk_swz16(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float walk_swz16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }

template <typename T>
__device__ __forceinline__ T wave_allreduce(T v)
{
    v = v + walk_dpp<0xB1>(v);
    v = v + walk_dpp<0x4E>(v);
    v = v + walk_dpp<0x141>(v);
    v = v + walk_dpp<0x140>(v);
    v = v + walk_swz16(v);
    v = v + __shfl_xor(v, 32);
    return v;
}

constexpr int WALK_WPB = BLK / 64;
constexpr int WALK_RING = 8;                  // rows whose static operands are in registers
constexpr int WALK_AHEAD = 4;                 // rows whose b / x / first poll are in flight

template <typename T, int K>
struct WalkRow {
    int c[K];
    T v[K];
    int rid;
    T rd, af;
    T bv, xo;
    T xv[K];
};

template <typename T, int EPI, int K>
__global__ __launch_bounds__(BLK) void gs_walk_kernel(const WalkArgs<T> a)
{
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int W = (int)gridDim.x * WALK_WPB;
    const int idle = (int)((((unsigned)blockIdx.x * WALK_WPB + (unsigned)wib) * 16u) % (unsigned)a.nidle);
    WalkRow<T, K> R[WALK_RING];
    for (int line = (int)blockIdx.x * WALK_WPB + wib; line < a.nlines; line += W) {
        const int q0 = a.line_row[line], q1 = a.line_row[line + 1];
        auto load_static = [&](WalkRow<T, K> &S, int q) {
            q = min(q, q1 - 1);                                 // unconditional (see pamg_lane.hip): beyond the line the last row again
            const size_t e0 = (size_t)q * (size_t)(K * 64) + (size_t)lane;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                S.c[k] = a.cols[e0 + (size_t)k * 64];
                S.v[k] = a.vals[e0 + (size_t)k * 64];
            }
            S.rid = a.rid[q];
            S.rd = a.rdiag[q];
            S.af = a.afwd[q];
        };
        auto load_dynamic = [&](WalkRow<T, K> &S) {
            const int row = S.rid & WALK_MASK;
            S.bv = a.b[row];
            S.xo = T(0);
            if constexpr (EPI == EPI_SOR) S.xo = a.x[row];
            else if (S.rid & WALK_NODIAG) S.xo = a.x[row];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int c = S.c[k];
                const int col = c & WALK_MASK;
                const T *p = (c & WALK_NONE) ? a.x + idle : ((c & WALK_EARLY) ? a.xs + col : a.x + col);
                S.xv[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        T vprev = T(0);
        auto finish = [&](WalkRow<T, K> &S) {
            unsigned pend = 0;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if ((S.c[k] & WALK_EARLY) && !(S.c[k] & WALK_NONE) && WSentinel<T>::bits(S.xv[k]) == WSentinel<T>::value) pend |= 1u << k;
            unsigned spins = 0;
            while (pend) {
                if (spins) __builtin_amdgcn_s_sleep(1);
                T t[K];
#pragma unroll
                for (int k = 0; k < K; ++k)
                    t[k] = __hip_atomic_load(((pend >> k) & 1u) ? a.xs + (S.c[k] & WALK_MASK) : a.xs + idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if ((pend >> k) & 1u) {
                        S.xv[k] = t[k];
                        if (WSentinel<T>::bits(t[k]) != WSentinel<T>::value) pend &= ~(1u << k);
                    }
                if ((++spins & 1023u) == 0) {
                    if (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            T s = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const T pr = S.v[k] * S.xv[k];
                s = s + ((S.c[k] & WALK_NONE) ? T(0) : pr);
            }
            s = wave_allreduce<T>(s);                          // independent of the predecessor's value
            const T fw = S.af * vprev;
            T v = (S.bv - (s + fw)) * S.rd;
            if constexpr (EPI == EPI_SOR) v = a.omega * v + (T(1) - a.omega) * S.xo;
            const bool upd = !(S.rid & WALK_NODIAG);
            if (!upd) v = S.xo;
            if (lane == 0) {
                const int row = S.rid & WALK_MASK;
                __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (upd) a.y[row] = v;
            }
            vprev = v;
        };
        // prologue: static operands of the first RING rows, then the operands of the first AHEAD rows
#pragma unroll
        for (int u = 0; u < WALK_RING; ++u) load_static(R[u], q0 + u);
#pragma unroll
        for (int u = 0; u < WALK_AHEAD; ++u) load_dynamic(R[u]);
        for (int base = q0; base < q1; base += WALK_RING) {
#pragma unroll
            for (int u = 0; u < WALK_RING; ++u) {
                const int q = base + u;
                load_dynamic(R[(u + WALK_AHEAD) % WALK_RING]);  // row q + AHEAD (its static operands arrived RING - AHEAD rows ago)
                if (q < q1) finish(R[u]);
                load_static(R[u], q + WALK_RING);
            }
        }
    }
}

// ------------------------------------------------------------------ host side
namespace {

template <typename U>
int walk_upload(U **dst, const void *src, size_t bytes, size_t *total)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 256;
    PAMG_HIP(hipMalloc((void **)dst, alloc));
    if (bytes) PAMG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    if (total) *total += alloc;
    return PAMG_OK;
}

template <typename T, int EPI>
const void *walk_kernel_k(int K)
{
    switch (K) {
        case 1: return (const void *)gs_walk_kernel<T, EPI, 1>;
        case 2: return (const void *)gs_walk_kernel<T, EPI, 2>;
        case 3: return (const void *)gs_walk_kernel<T, EPI, 3>;
        case 4: return (const void *)gs_walk_kernel<T, EPI, 4>;
    }
    return nullptr;
}

}  // namespace

void free_walk_part(WalkSched *t)
{
    if (!t) return;
    hipFree(t->d_cols); hipFree(t->d_rid); hipFree(t->d_line_row); hipFree(t->d_vals); hipFree(t->d_rdiag); hipFree(t->d_afwd);
    delete t;
}

size_t walk_part_bytes(const GsSchedule *g) { return (g && g->walk) ? g->walk->bytes : 0; }

bool walk_eligible(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->R == 1 && g->nlevels > 1 && g->d_xs != nullptr && g->nrows >= 65536 && A->max_row_len <= WALK_KMAX * 64 + 2 && !g->walk_unfit;
}

int build_walk_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->walk) return PAMG_OK;
    PhaseTimer pt_("build_walk_part", A->nnz);
    const int ts = (int)tsize(A->dtype);
    std::vector<unsigned char> hAx((size_t)A->nnz * ts);
    if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    WalkPlan P;
    if (build_walk_plan((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), hAx.data(), ts, g->row_start, g->row_stop, g->row_step, P))
        return PAMG_E_ARG;
    WalkSched *t = new (std::nothrow) WalkSched();
    if (!t) return PAMG_E_ALLOC;
    t->K = P.K; t->nrows = P.nrows; t->nlines = P.nlines; t->nlevels = P.nlevels;
    t->max_level_lines = P.max_level_lines; t->n_forward = P.n_forward;
    int st = walk_upload(&t->d_cols, P.cols.data(), P.cols.size() * sizeof(int), &t->bytes);
    if (!st) st = walk_upload(&t->d_vals, P.vals.data(), P.vals.size(), &t->bytes);
    if (!st) st = walk_upload(&t->d_rid, P.rid.data(), P.rid.size() * sizeof(int), &t->bytes);
    if (!st) st = walk_upload(&t->d_rdiag, P.rdiag.data(), P.rdiag.size(), &t->bytes);
    if (!st) st = walk_upload(&t->d_afwd, P.afwd.data(), P.afwd.size(), &t->bytes);
    if (!st) st = walk_upload(&t->d_line_row, P.line_row.data(), P.line_row.size() * sizeof(int), &t->bytes);
    if (st) { free_walk_part(t); return st; }
    g->walk = t;
    g->bytes += t->bytes;                                      // the caller books them on the operator
    return PAMG_OK;
}

static int walk_cus()
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 64;
    return p.multiProcessorCount;
}

template <typename T>
static int walk_launch_t(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    WalkSched *t = g->walk;
    const size_t ts = tsize(A->dtype);
    const int64_t n = A->nrows;
    WalkArgs<T> a;
    a.cols = t->d_cols; a.vals = (const T *)t->d_vals; a.rid = t->d_rid; a.rdiag = (const T *)t->d_rdiag; a.afwd = (const T *)t->d_afwd;
    a.line_row = t->d_line_row;
    a.x = (const T *)x; a.y = (T *)x; a.xs = (T *)g->d_xs; a.b = (const T *)b;
    a.err = g->d_sync + 1;
    a.nlines = (int)t->nlines;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
    a.omega = (T)omega;
    if (!g->symmetric) {
        if (!g->d_xold) return PAMG_E_STATE;
        PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
        a.x = (const T *)g->d_xold;
    }
    const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
    hipLaunchKernelGGL((walk_fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
    PAMG_HIP(hipGetLastError());
    const void *k = epi == EPI_SOR ? walk_kernel_k<T, EPI_SOR>(t->K) : walk_kernel_k<T, EPI_GS>(t->K);
    if (!k) return PAMG_E_ARG;
    static thread_local int cus = 0;
    if (!cus) cus = walk_cus();
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLK, 0) != hipSuccess) nb = 2;
    const int cap = std::max(1, std::min(nb - 1, 8));          // every workgroup must be resident (the query can over-report by one)
    // a line is in flight for ~10 hand-off times: about ten dependency levels of lines
    const int64_t want_waves = std::max<int64_t>(256, 10 * t->max_level_lines);
    int G = (int)std::min<int64_t>((want_waves + WALK_WPB - 1) / WALK_WPB, (int64_t)cap * cus);
    if (A->lane_G > 0) G = std::min(A->lane_G, cap * cus);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, (t->nlines + WALK_WPB - 1) / WALK_WPB));
    t->last_grid = G;
    void *args[] = {(void *)&a};
    PAMG_HIP(hipLaunchKernel(k, dim3(G), dim3(BLK), args, 0, s));
    return PAMG_OK;
}

int walk_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    if (!g->walk) return PAMG_E_STATE;
    if (A->dtype == PAMG_F64) return walk_launch_t<double>(A, g, epi, x, b, omega, s);
    return walk_launch_t<float>(A, g, epi, x, b, omega, s);
}

// info[0..7] = entry slots per lane, rows, lines, levels of the line graph, rows that take their predecessor from a register,
// workgroups of the last launch, lines of the widest level, bytes
int walk_info(const GsSchedule *g, int64_t *info)
{
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (!g || !g->walk) return PAMG_OK;
    const WalkSched *t = g->walk;
    info[0] = t->K; info[1] = t->nrows; info[2] = t->nlines; info[3] = t->nlevels; info[4] = t->n_forward; info[5] = t->last_grid;
    info[6] = t->max_level_lines; info[7] = (int64_t)t->bytes;
    return PAMG_OK;
}

}  // namespace pamg
