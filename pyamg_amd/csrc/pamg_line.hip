// pamg_line.hip -- the LINE-SCAN ("fast order") Gauss-Seidel / SOR sweep for banded operators in their natural order
// (the fine levels of structured-grid problems); layout and derivation in pamg_line_plan.h.
//
// amg_core::gauss_seidel (relaxation.h:48-76) / sor_gauss_seidel (:116-145) / bsr_gauss_seidel with 1x1 blocks (:185-266)
// over consecutive rows: the row visited just before a row is (almost always) one of its early operands, so along a grid line
// the sweep is the first-order linear recurrence x_t = B_t + A_t x_{t-1}.  One wave takes 64 consecutive rows (one per lane),
// forms B_t = (b_t - sum of the OTHER entries) * (1 / a_tt) and A_t = - a_{t,t-1} / a_tt, and finishes the 64 rows with an
// inclusive scan of the pairs (six combining steps) -- the same algebra in another association: the reference's iterates
// up to rounding (fast order, tune key 24 = 1; the order-exact schedulers remain).  A LINE (chunks chained by the coupling)
// belongs to one wave, the running value stays in a register; between lines the hand-off is the sentinel protocol of the
// other persistent sweeps (the published 8-byte value is the flag).  Lines are taken in the order of their dependency level
// over the line graph: wave w takes lines w, w + W, ... -- all W waves co-resident, a line only waits for lines of lower
// levels: deadlock-free.  On an n^3 seven-point grid: 2 n levels of up to n lines of n rows each, instead of 3 n levels of
// scattered rows.
#include "pamg_common.h"
#include "pamg_line_plan.h"

namespace pamg {

struct LineSched {
    int K = 0, step = 1;
    int64_t nchunks = 0, nlines = 0;
    int nlevels = 0;
    int *d_cols = nullptr, *d_row0 = nullptr, *d_cnt = nullptr, *d_gate = nullptr, *d_line_chunk = nullptr;
    void *d_vals = nullptr, *d_rdiag = nullptr, *d_acoef = nullptr;
    unsigned char *d_nodiag = nullptr;
    int64_t n_early = 0, max_level_lines = 0;
    int last_grid = 0;
    int cap = 0;                    // co-resident workgroups per CU of this schedule's kernel (queried once)
    const void *cap_kernel = nullptr;
    size_t bytes = 0;
};

template <typename T> struct LSentinel;
template <> struct LSentinel<double> {
    using bits_t = unsigned long long;
    static constexpr bits_t value = 0x7FF8DEADBEEF5A5Aull;
    static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};
template <> struct LSentinel<float> {
    using bits_t = unsigned int;
    static constexpr bits_t value = 0x7FC5BEEFu;
    static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};

template <typename T>
__global__ __launch_bounds__(BLK) void line_fill_sentinel_kernel(T *xs, int64_t n)
{
    using B = typename LSentinel<T>::bits_t;
    B *p = reinterpret_cast<B *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) p[i] = LSentinel<T>::value;
}

template <typename T>
struct LineArgs {
    const int *cols;
    const T *vals, *rdiag, *acoef;
    const unsigned char *nodiag;
    const int *row0, *cnt, *gate, *line_chunk;
    const T *x;            // OLD values (x itself, or its snapshot for structurally non-symmetric patterns)
    T *y;                  // destination (the live x)
    T *xs;                 // hand-off buffer, sentinel-filled
    const T *b;
    unsigned *err;
    int nlines, step, nidle, use_gate;
    T omega;
};

template <typename T, int K>
struct LineSet {
    int c[K];
    T v[K];
    T rd, ac;
    int nod, row0, cnt, gate;
};

template <typename T, int K>
__device__ __forceinline__ void line_load(const LineArgs<T> &a, int g, LineSet<T, K> &S)
{
    const int lane = threadIdx.x & 63;
    const size_t e0 = (size_t)g * (size_t)(K * 64) + (size_t)lane;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        S.c[k] = a.cols[e0 + (size_t)k * 64];
        S.v[k] = a.vals[e0 + (size_t)k * 64];
    }
    const size_t rs = (size_t)g * 64 + (size_t)lane;
    S.rd = a.rdiag[rs];
    S.ac = a.acoef[rs];
    S.nod = a.nodiag[rs];
    S.row0 = a.row0[g];
    S.cnt = a.cnt[g];
    S.gate = a.gate[g];
}

constexpr int LINE_WPB = BLK / 64;

// ---- lane shifts for the scan: DPP row_shr inside the rows of 16 lanes, row_bcast15 / row_bcast31 across them
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double line_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float line_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ double line_readlane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float line_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// inclusive scan of the pairs (A, B) over the 64 lanes under (A2, B2) o (A1, B1) = (A2 A1, B2 + A2 B1)
template <typename T>
__device__ __forceinline__ void line_scan(T &Av, T &Bv, int lane)
{
#define PAMG_LS_STEP(CTRL, D)                                                        \
    {                                                                                \
        const T Au = line_dpp<CTRL, 0xF>(Av), Bu = line_dpp<CTRL, 0xF>(Bv);          \
        if ((lane & 15) >= D) { Bv = Bv + Av * Bu; Av = Av * Au; }                   \
    }
    PAMG_LS_STEP(0x111, 1) PAMG_LS_STEP(0x112, 2) PAMG_LS_STEP(0x114, 4) PAMG_LS_STEP(0x118, 8)
#undef PAMG_LS_STEP
    {   // rows 1 and 3 take over the total of the row before them (its lane 15)
        const T Au = line_dpp<0x142, 0xA>(Av), Bu = line_dpp<0x142, 0xA>(Bv);
        if (lane & 16) { Bv = Bv + Av * Bu; Av = Av * Au; }
    }
    {   // rows 2 and 3 take over the total of the first half (lane 31)
        const T Au = line_dpp<0x143, 0xC>(Av), Bu = line_dpp<0x143, 0xC>(Bv);
        if (lane & 32) { Bv = Bv + Av * Bu; Av = Av * Au; }
    }
}

template <typename T, int EPI, int K>
__global__ __launch_bounds__(BLK) void gs_line_kernel(const LineArgs<T> a)
{
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int W = (int)gridDim.x * LINE_WPB;
    const int idle = (int)((((unsigned)blockIdx.x * LINE_WPB + (unsigned)wib) * 16u) % (unsigned)a.nidle);
    for (int line = (int)blockIdx.x * LINE_WPB + wib; line < a.nlines; line += W) {
        const int g0 = a.line_chunk[line], g1 = a.line_chunk[line + 1];
        T carry = T(0);
        LineSet<T, K> P, Q;
        line_load<T, K>(a, g0, P);
        // the two operand sets alternate (a copy "cur = next" would wait for the loads it has just issued)
        auto chunk = [&](LineSet<T, K> &S, LineSet<T, K> &N, int g) {
            const bool active = lane < S.cnt;
            const int row = S.row0 + (active ? lane : 0) * a.step;
            // ---- everything that depends on the chunk's static operands, requested at once
            const T bv = a.b[row];
            T xo = T(0);
            if constexpr (EPI == EPI_SOR) xo = a.x[row];
            else if (S.nod) xo = a.x[row];
            T xv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int c = S.c[k];
                const int col = c & LINE_MASK;
                const T *p = ((c & LINE_NONE) || !active) ? a.x + idle : ((c & LINE_EARLY) ? a.xs + col : a.x + col);
                xv[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            line_load<T, K>(a, min(g + 1, g1 - 1), N);         // the next chunk of the line (unconditional: see pamg_lane.hip)
            unsigned pend = 0;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (active && (S.c[k] & LINE_EARLY) && !(S.c[k] & LINE_NONE) && LSentinel<T>::bits(xv[k]) == LSentinel<T>::value) pend |= 1u << k;
            unsigned spins = 0;
            if (a.use_gate && S.gate >= 0 && __builtin_amdgcn_ballot_w64(pend != 0)) {
                const T *gp = a.xs + S.gate;
                while (true) {
                    const T gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (LSentinel<T>::bits(gv) != LSentinel<T>::value) break;
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023u) == 0 && (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
                }
            }
            while (pend) {
                if (spins) __builtin_amdgcn_s_sleep(1);
                T t[K];
#pragma unroll
                for (int k = 0; k < K; ++k)
                    t[k] = __hip_atomic_load(((pend >> k) & 1u) ? a.xs + (S.c[k] & LINE_MASK) : a.xs + idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if ((pend >> k) & 1u) {
                        xv[k] = t[k];
                        if (LSentinel<T>::bits(t[k]) != LSentinel<T>::value) pend &= ~(1u << k);
                    }
                if ((++spins & 1023u) == 0) {
                    if (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            // ---- B and A of the recurrence x_t = B_t + A_t x_{t-1}
            T s = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const T pr = S.v[k] * xv[k];
                s = s + ((S.c[k] & LINE_NONE) ? T(0) : pr);
            }
            T Bv = (bv - s) * S.rd, Av = S.ac;
            if constexpr (EPI == EPI_SOR) { Bv = a.omega * Bv + (T(1) - a.omega) * xo; Av = a.omega * Av; }
            if (S.nod) { Bv = xo; Av = T(0); }
            if (!active) { Bv = T(0); Av = T(0); }
            // ---- inclusive scan over the lanes: (A2, B2) o (A1, B1) = (A2 A1, B2 + A2 B1)
            line_scan<T>(Av, Bv, lane);
            const T v = Bv + Av * carry;
            if (active) {
                __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!S.nod) a.y[row] = v;
            }
            carry = line_readlane(v, S.cnt - 1);
        };
        int g = g0;
        while (true) {
            chunk(P, Q, g);
            if (++g >= g1) break;
            chunk(Q, P, g);
            if (++g >= g1) break;
        }
    }
}

// (Round 5: a WORKGROUP per line -- a wave per chunk, the chunk totals folded through LDS behind one barrier, bit-identical -- was built and measured:
// 0.91 ms with 256 lines in flight, 1.06 with 768, against 0.80 for the walking wave above on the 256^3 fine level.  Four waves poll for a line where
// one did; the polls load the memory system more than the shorter chain saves.  profiles/r05_microbench_line_workgroup_per_line_slower_not_kept.json)
// ------------------------------------------------------------------ the layout, filled on the device
// The chunk structure comes from the host (pattern only: pamg_line_plan.h with fill = false); cols / vals / rdiag / acoef /
// nodiag / gate are written here from the resident CSR arrays -- no download of the values, no upload of the 1.3 GB layout
// (256^3: 1.6 s of host time per sweep direction before).  One wave per chunk, lane = row; the slots follow the host's rule
// (build_line_plan's fill pass, which the CPU suite replays): entries in storage order without the diagonal (last stored
// one wins) and without the FIRST entry of the in-line predecessor.
template <typename T>
__global__ __launch_bounds__(256) void line_fill_kernel(int n, const int *__restrict__ Ap, const int *__restrict__ Aj, const T *__restrict__ Ax, int row_start,
                                                        int row_step, long long m, int K, long long nchunks, const int *__restrict__ row0, const int *__restrict__ cnt,
                                                        const unsigned char *__restrict__ coupled, int *__restrict__ cols, T *__restrict__ vals, T *__restrict__ rdiag,
                                                        T *__restrict__ acoef, unsigned char *__restrict__ nodiag, int *__restrict__ gate, unsigned long long *__restrict__ counters)
{
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    if (g >= nchunks) return;
    const int c = cnt[g], r0 = row0[g];
    int my_gate = -1;
    unsigned ne = 0, no = 0;
    // padding lanes keep the host's defaults
    for (int k = 0; k < K; ++k) { cols[(size_t)((g * K + k) * 64 + lane)] = LINE_NONE; vals[(size_t)((g * K + k) * 64 + lane)] = T(0); }
    T rd = T(0), ac = T(0);
    unsigned char nod = 0;
    if (lane < c) {
        const int i = r0 + lane * row_step;
        const long long t = ((long long)i - row_start) * row_step;
        const int prev = t > 0 ? (int)(row_start + (t - 1) * row_step) : -1;
        const bool in_line = lane > 0 || coupled[g];
        T d = T(0), ap = T(0);
        bool have_p = false;
        int k = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) { d = Ax[p]; continue; }
            if (j == prev && !have_p && in_line) { ap = Ax[p]; have_p = true; continue; }
            const size_t s = (size_t)((g * K + k) * 64 + lane);
            ++k;
            if (j < 0 || j >= n) continue;
            const long long tj = ((long long)j - row_start) * row_step;
            const bool early = tj >= 0 && tj < m && tj < t;
            cols[s] = j | (early ? LINE_EARLY : 0);
            vals[s] = Ax[p];
            if (early) { ++ne; my_gate = j; } else ++no;
        }
        nod = !(d != T(0));
        rd = nod ? T(0) : T(1) / d;
        ac = nod ? T(0) : -ap * rd;
    }
    rdiag[(size_t)(g * 64 + lane)] = rd;
    acoef[(size_t)(g * 64 + lane)] = ac;
    nodiag[(size_t)(g * 64 + lane)] = nod;
    // the chunk's gate: the last early operand in (lane, entry) order = the highest lane that has one
    const unsigned long long has = __ballot(my_gate >= 0);
    int gsel = -1;
    if (has) gsel = __shfl(my_gate, 63 - __builtin_clzll(has));
    // totals (statistics)
    for (int o = 32; o > 0; o >>= 1) { ne += __shfl_down(ne, o); no += __shfl_down(no, o); }
    if (lane == 0) {
        gate[g] = gsel;
        if (ne) atomicAdd(&counters[0], (unsigned long long)ne);
        if (no) atomicAdd(&counters[1], (unsigned long long)no);
    }
}

// ------------------------------------------------------------------ host side
namespace {

template <typename U>
int line_upload(U **dst, const void *src, size_t bytes, size_t *total)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 256;
    PAMG_HIP(hipMalloc((void **)dst, alloc));
    if (bytes && src) PAMG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));      // src = nullptr: allocation only (filled on the device)
    if (total) *total += alloc;
    return PAMG_OK;
}

template <typename T, int EPI>
const void *line_kernel_k(int K)
{
    switch (K) {
        case 1: return (const void *)gs_line_kernel<T, EPI, 1>;
        case 2: return (const void *)gs_line_kernel<T, EPI, 2>;
        case 3: return (const void *)gs_line_kernel<T, EPI, 3>;
        case 4: return (const void *)gs_line_kernel<T, EPI, 4>;
        case 5: return (const void *)gs_line_kernel<T, EPI, 5>;
        case 6: return (const void *)gs_line_kernel<T, EPI, 6>;
        case 7: return (const void *)gs_line_kernel<T, EPI, 7>;
        case 8: return (const void *)gs_line_kernel<T, EPI, 8>;
    }
    return nullptr;
}

}  // namespace

void free_line_part(LineSched *t)
{
    if (!t) return;
    hipFree(t->d_cols); hipFree(t->d_row0); hipFree(t->d_cnt); hipFree(t->d_gate); hipFree(t->d_line_chunk);
    hipFree(t->d_vals); hipFree(t->d_rdiag); hipFree(t->d_acoef); hipFree(t->d_nodiag);
    delete t;
}

size_t line_part_bytes(const GsSchedule *g) { return (g && g->line) ? g->line->bytes : 0; }

bool line_eligible(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->R == 1 && g->nlevels > 1 && g->d_xs != nullptr && (g->row_step == 1 || g->row_step == -1) && g->nrows >= 4096 &&
           A->max_row_len <= LINE_KMAX + 2 && !g->line_unfit;
}

int build_line_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->line) return PAMG_OK;
    PhaseTimer pt_("build_line_part", A->nnz);
    const int ts = (int)tsize(A->dtype);
    // PAMG_LINE_HOST_FILL=1: the whole layout on the host (values downloaded, layout uploaded) -- what the CPU suite replays; the
    // default fills it on the device from the resident arrays (same arrays, tests/test_gpu_kernels.py compares the sweeps)
    const char *hf = getenv("PAMG_LINE_HOST_FILL");
    const bool host_fill = (hf && *hf == '1') || !A->d_Ap || !A->d_Aj || !A->d_Ax;
    PlanVec<unsigned char> hAx;
    if (host_fill) {
        hAx.resize((size_t)A->nnz * ts);
        if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    }
    LinePlan P;
    if (build_line_plan((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), host_fill ? hAx.data() : nullptr, ts, g->row_start, g->row_stop, g->row_step, P, host_fill))
        return PAMG_E_ARG;
    // Is the scan worth it?  Lines of one line level run side by side, a line's chunks one after the other in one wave: the sweep takes
    // about (line levels x chunks per line) chunk steps of 0.39 us (256^3: 511 x 4 steps, 0.79 ms measured), the lane form about
    // max(row dependency levels x 1.05 us hand-off, rows x 0.115 ns of throughput) (256^3: 1.93 ms).  A 3-D grid has hundreds of lines
    // per level; a 2-D grid in its natural order has ONE -- 2000 lines of 32 chunks in a row, 25 ms against the lane form's 4.
    {
        const double line_est = (double)P.nlevels * std::max(1.0, (double)P.nchunks / std::max<int64_t>(1, P.nlines)) * 0.39e-3;
        const double lane_est = std::max((double)g->nlevels * 1.05e-3, (double)g->nrows * 0.115e-6);
        if (A->line_scan < 2 && line_est > lane_est) return PAMG_E_ARG;
    }
    LineSched *t = new (std::nothrow) LineSched();
    if (!t) return PAMG_E_ALLOC;
    t->K = P.K; t->step = P.step; t->nchunks = P.nchunks; t->nlines = P.nlines; t->nlevels = P.nlevels;
    t->n_early = P.n_early; t->max_level_lines = P.max_level_lines;
    int st = line_upload(&t->d_row0, P.row0.data(), P.row0.size() * sizeof(int), &t->bytes);
    if (!st) st = line_upload(&t->d_cnt, P.cnt.data(), P.cnt.size() * sizeof(int), &t->bytes);
    if (!st) st = line_upload(&t->d_line_chunk, P.line_chunk.data(), P.line_chunk.size() * sizeof(int), &t->bytes);
    if (host_fill) {
        if (!st) st = line_upload(&t->d_cols, P.cols.data(), P.cols.size() * sizeof(int), &t->bytes);
        if (!st) st = line_upload(&t->d_vals, P.vals.data(), P.vals.size(), &t->bytes);
        if (!st) st = line_upload(&t->d_rdiag, P.rdiag.data(), P.rdiag.size(), &t->bytes);
        if (!st) st = line_upload(&t->d_acoef, P.acoef.data(), P.acoef.size(), &t->bytes);
        if (!st) st = line_upload(&t->d_nodiag, P.nodiag.data(), P.nodiag.size(), &t->bytes);
        if (!st) st = line_upload(&t->d_gate, P.gate.data(), P.gate.size() * sizeof(int), &t->bytes);
    } else {
        const size_t slots = (size_t)P.nchunks * P.K * 64, rows = (size_t)P.nchunks * 64;
        unsigned char *d_coupled = nullptr;
        unsigned long long *d_cnt2 = nullptr;
        if (!st) st = line_upload(&t->d_cols, nullptr, slots * sizeof(int), &t->bytes);
        if (!st) st = line_upload(&t->d_vals, nullptr, slots * ts, &t->bytes);
        if (!st) st = line_upload(&t->d_rdiag, nullptr, rows * ts, &t->bytes);
        if (!st) st = line_upload(&t->d_acoef, nullptr, rows * ts, &t->bytes);
        if (!st) st = line_upload(&t->d_nodiag, nullptr, rows, &t->bytes);
        if (!st) st = line_upload(&t->d_gate, nullptr, (size_t)P.nchunks * sizeof(int), &t->bytes);
        if (!st) st = line_upload(&d_coupled, P.coupled.data(), P.coupled.size(), nullptr);
        if (!st) st = line_upload(&d_cnt2, nullptr, 16, nullptr);
        if (!st) st = (int)hipMemset(d_cnt2, 0, 16);
        if (!st) {
            const unsigned grid = (unsigned)((P.nchunks + 3) / 4);
            (void)hipGetLastError();                                  // a stale error of an earlier query must not be taken for this launch's
            if (ts == 8)
                hipLaunchKernelGGL((line_fill_kernel<double>), dim3(grid), dim3(256), 0, 0, (int)A->nrows, A->d_Ap, A->d_Aj, (const double *)A->d_Ax, g->row_start, g->row_step,
                                   (long long)(((int64_t)g->row_stop - g->row_start) / g->row_step), P.K, (long long)P.nchunks, t->d_row0, t->d_cnt, d_coupled, t->d_cols,
                                   (double *)t->d_vals, (double *)t->d_rdiag, (double *)t->d_acoef, t->d_nodiag, t->d_gate, d_cnt2);
            else
                hipLaunchKernelGGL((line_fill_kernel<float>), dim3(grid), dim3(256), 0, 0, (int)A->nrows, A->d_Ap, A->d_Aj, (const float *)A->d_Ax, g->row_start, g->row_step,
                                   (long long)(((int64_t)g->row_stop - g->row_start) / g->row_step), P.K, (long long)P.nchunks, t->d_row0, t->d_cnt, d_coupled, t->d_cols,
                                   (float *)t->d_vals, (float *)t->d_rdiag, (float *)t->d_acoef, t->d_nodiag, t->d_gate, d_cnt2);
            st = (int)hipGetLastError();
            unsigned long long hc[2] = {0, 0};
            if (!st) st = (int)hipMemcpy(hc, d_cnt2, 16, hipMemcpyDeviceToHost);      // also the fill's completion
            t->n_early = (int64_t)hc[0];
        }
        hipFree(d_coupled); hipFree(d_cnt2);
    }
    if (st) { free_line_part(t); return st; }
    g->line = t;
    g->bytes += t->bytes;                                      // the caller books them on the operator
    return PAMG_OK;
}

static int line_cus()
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 64;
    return p.multiProcessorCount;
}

template <typename T>
static int line_launch_t(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    LineSched *t = g->line;
    const size_t ts = tsize(A->dtype);
    const int64_t n = A->nrows;
    LineArgs<T> a;
    a.cols = t->d_cols; a.vals = (const T *)t->d_vals; a.rdiag = (const T *)t->d_rdiag; a.acoef = (const T *)t->d_acoef;
    a.nodiag = t->d_nodiag; a.row0 = t->d_row0; a.cnt = t->d_cnt; a.gate = t->d_gate; a.line_chunk = t->d_line_chunk;
    a.x = (const T *)x; a.y = (T *)x; a.xs = (T *)g->d_xs; a.b = (const T *)b;
    a.err = g->d_sync + 1;
    a.nlines = (int)t->nlines; a.step = t->step;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
    a.use_gate = (A->lane_flags & 8) ? 1 : 0;               // measured slower on the 256^3 grid (0.96 vs 0.84 ms): off unless asked for
    a.omega = (T)omega;
    if (!g->symmetric) {
        if (!g->d_xold) return PAMG_E_STATE;
        PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
        a.x = (const T *)g->d_xold;
    }
    const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
    hipLaunchKernelGGL((line_fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
    PAMG_HIP(hipGetLastError());
    const void *k = epi == EPI_SOR ? line_kernel_k<T, EPI_SOR>(t->K) : line_kernel_k<T, EPI_GS>(t->K);
    if (!k) return PAMG_E_ARG;
    static thread_local int cus = 0;
    if (!cus) cus = line_cus();
    if (t->cap <= 0 || t->cap_kernel != k) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLK, 0) != hipSuccess) nb = 2;
        t->cap = std::max(1, std::min(nb - 1, 8));              // every workgroup must be resident (the query can over-report by one)
        t->cap_kernel = k;
    }
    const int cap = t->cap;
    // waves: a few dependency levels of lines in flight (a line that runs ahead waits with its operands in registers)
    const int64_t want_waves = std::max<int64_t>(256, 4 * t->max_level_lines);
    int G = (int)std::min<int64_t>((want_waves + LINE_WPB - 1) / LINE_WPB, (int64_t)cap * cus);
    if (A->lane_G > 0) G = std::min(A->lane_G, cap * cus);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, (t->nlines + LINE_WPB - 1) / LINE_WPB));
    t->last_grid = G;
    void *args[] = {(void *)&a};
    PAMG_HIP(hipLaunchKernel(k, dim3(G), dim3(BLK), args, 0, s));
    return PAMG_OK;
}

int line_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    if (!g->line) return PAMG_E_STATE;
    if (A->dtype == PAMG_F64) return line_launch_t<double>(A, g, epi, x, b, omega, s);
    return line_launch_t<float>(A, g, epi, x, b, omega, s);
}

// info[0..7] = entry slots per row, chunks, lines, levels of the line graph, early entries, workgroups of the last launch,
// lines of the widest level, bytes
int line_info(const GsSchedule *g, int64_t *info)
{
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (!g || !g->line) return PAMG_OK;
    const LineSched *t = g->line;
    info[0] = t->K; info[1] = t->nchunks; info[2] = t->nlines; info[3] = t->nlevels; info[4] = t->n_early; info[5] = t->last_grid;
    info[6] = t->max_level_lines; info[7] = (int64_t)t->bytes;
    return PAMG_OK;
}

}  // namespace pamg
