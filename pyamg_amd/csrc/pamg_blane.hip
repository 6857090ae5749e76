// pamg_blane.hip -- the LANE-PARALLEL ("fast order") block Gauss-Seidel sweep on square-block BSR operators: layout in pamg_blane_plan.h.
//
// Same block rows in the same order as amg_core::block_gauss_seidel (relaxation.h:1242-1298) -- what smoothed_aggregation_solver
// configures by default for BSR operators (aggregation.py:97-99, relaxation/smoothing.py:664-692) -- so the dependency DAG of the
// sequential sweep is kept and the iterates are the reference's up to rounding; the order of the additions inside a block row is
// not: L lanes of a wave share a block row, every lane multiplies its K off-diagonal blocks with their x_j in registers, the bs
// partial sums are added across the lanes, lane q < bs of the row forms component q of Dinv_i (b_i - sum) and publishes it.
// ONE persistent launch per sweep, waves never meet (no LDS, no barrier), the hand-off of pamg_lane.hip component by component:
// the published 8-byte value is the flag (sentinel-filled buffer xs, write-through store -> polled L1-bypassing load), a consumer
// takes x_j when none of its bs components is the sentinel any more.
//   static form : wave w takes groups w, w + W, w + 2W, ... (all W waves co-resident; a group only waits for groups with smaller
//                 numbers -> deadlock-free);
//   one-XCD form: small operators.  The first workgroup to arrive claims its XCD, workgroups elsewhere leave, the rest draw groups
//                 from a ticket counter (two tickets ahead, taken in increasing order by running waves) and publish with ordinary
//                 L2-resident stores.
// The order-exact kernels (bsr_gran / bsr_small / bsr_flow, pamg_kernels.h) stay for order = 'exact', for the BSR point sweep
// (amg_core::bsr_gauss_seidel), for float operators and for block sizes other than 2, 3, 4, 6.
#include "pamg_common.h"
#include "pamg_blane_plan.h"

namespace pamg {

constexpr unsigned long long BL_SENTINEL = 0x7FF8DEADBEEF5A5Aull;     // the pattern of the exact sweeps (pamg_kernels.h)

struct BlaneSched {
    int L = 0, K = 0, RPW = 0, bs = 0, nlevels = 0;
    int64_t ngroups = 0, nslots = 0, max_level_groups = 0, n_early = 0, n_old = 0;
    int *d_cols = nullptr, *d_rid = nullptr, *d_gate = nullptr;
    double *d_vals = nullptr, *d_zero = nullptr;     // d_zero: what the padding slots read (0 x 0: no select behind the products)
    long long *d_prof = nullptr;
    int last_grid = 0;
    int cap = 0;
    const void *cap_kernel = nullptr;
    size_t bytes = 0;
};

struct BlaneArgs {
    const int *cols, *rid, *gate;      // gate: nullptr = not used
    const double *vals;
    const double *x;       // OLD values (x itself, or its snapshot for structurally non-symmetric block patterns)
    double *y;             // destination (the live x)
    double *xs;            // hand-off buffer, sentinel-filled
    const double *b, *Dinv, *zero;
    unsigned *err, *ticket;
    long long *prof;       // nullptr or [ngroups][4] time stamps (tune key 11): group started, last operand seen, published, XCD | workgroup << 4
    int ngroups, nidle;
};

template <int BS, int K>
struct BlaneSet {
    int c[K];
    double v[K][BS * BS];
    int rid, gate;
};

template <int BS, int K>
struct BlaneDyn {
    double bv[BS], dv[BS];
    double xv[K][BS];
    long long t0;
};

template <int CTRL>
__device__ __forceinline__ double bl_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);       // (no `old` operand: every lane has a source, the compiler needs no copy)
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// lane ^ 16 and lane ^ 32 without the LDS path: v_permlane16_swap / v_permlane32_swap (CDNA4) exchange the odd 16-lane rows (the upper half) of
// one copy with the even rows (the lower half) of the other -- afterwards one copy holds the even-row (lower-half) values everywhere, the other the
// odd-row (upper-half) values, and their sum is the XOR-butterfly step (the same two addends in both partner lanes).  ds_swizzle / ds_bpermute cost
// ~100 cycles of latency each and a wait per value: 0.6 us of the 0.64 us between the last operand and the publish with six sums and 64 lanes.
__device__ __forceinline__ double bl_xor16_sum(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double bl_xor32_sum(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// the butterfly of the scalar lane form (pamg_lane.hip: seg_allreduce): the same partners in the same order
template <int L>
__device__ __forceinline__ double bl_allreduce(double v)
{
    v = v + bl_dpp<0xB1>(v);
    v = v + bl_dpp<0x4E>(v);
    if constexpr (L >= 8) v = v + bl_dpp<0x141>(v);
    if constexpr (L >= 16) v = v + bl_dpp<0x140>(v);
    if constexpr (L >= 32) v = bl_xor16_sum(v);
    if constexpr (L >= 64) v = bl_xor32_sum(v);
    return v;
}

__device__ __forceinline__ bool bl_missing(double v) { return (unsigned long long)__double_as_longlong(v) == BL_SENTINEL; }

template <int BS, int L, int K>
__device__ __forceinline__ void blane_load(const BlaneArgs &a, int g, BlaneSet<BS, K> &S)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t slot = (size_t)g * K + k;
        S.c[k] = a.cols[slot * 64 + lane];
        const double *vp = a.vals + slot * (size_t)(BS * BS * 64) + lane;
#pragma unroll
        for (int e = 0; e < BS * BS; ++e) S.v[k][e] = vp[(size_t)e * 64];
    }
    S.rid = a.rid[(size_t)g * (64 / L) + (lane / L)];
    S.gate = a.gate ? a.gate[g] : -1;
}

// first half of a group: everything that depends on its static operands only -- b_i, this lane's row of Dinv_i, the first round of operand
// loads (early blocks from the hand-off buffer, the others from x; both bypass the L1: other CUs write these lines during the launch)
template <int BS, int L, int K>
__device__ __forceinline__ void blane_issue(const BlaneArgs &a, const BlaneSet<BS, K> &S, BlaneDyn<BS, K> &D, int idle)
{
    const int lane = threadIdx.x & 63;
    const int q = lane & (L - 1);
    const int qc = q < BS ? q : 0;
    const size_t row = S.rid < 0 ? 0 : (size_t)S.rid;
    D.t0 = 0;
    if (a.prof && lane == 0) D.t0 = wall_clock64();
#pragma unroll
    for (int c = 0; c < BS; ++c) {
        D.bv[c] = a.b[row * BS + c];
        D.dv[c] = a.Dinv[row * (BS * BS) + (size_t)qc * BS + c];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = S.c[k];
        const size_t col = (size_t)(c & LANE_MASK) * BS;
        const double *p = (c & LANE_NONE) ? a.zero : ((c & LANE_EARLY) ? a.xs + col : a.x + col);
#pragma unroll
        for (int e = 0; e < BS; ++e) D.xv[k][e] = __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int BS, int L, int K, int MODE>
__device__ __forceinline__ void blane_finish(const BlaneArgs &a, const BlaneSet<BS, K> &S, BlaneDyn<BS, K> &D, int g, int idle)
{
    const int lane = threadIdx.x & 63;
    const int q = lane & (L - 1);
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((S.c[k] & LANE_EARLY) && !(S.c[k] & LANE_NONE)) {
            bool miss = false;
#pragma unroll
            for (int e = 0; e < BS; ++e) miss = miss || bl_missing(D.xv[k][e]);
            if (miss) pend |= 1u << k;
        }
    unsigned spins = 0;
    if (S.gate >= 0 && __builtin_amdgcn_ballot_w64(pend != 0)) {
        // a wave that runs ahead: the sweep is still two or more dependency levels away while the gate operand is missing -- the whole wave polls
        // that ONE value (its last component; one request per round) instead of all its operands
        const double *gp = a.xs + (size_t)S.gate * BS + (BS - 1);
        while (true) {
            const double gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!bl_missing(gv)) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 && (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
        }
        spins = 1;
    }
    while (__builtin_amdgcn_ballot_w64(pend != 0)) {
        if (spins) __builtin_amdgcn_s_sleep(1);
        double t[K][BS];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double *p = ((pend >> k) & 1u) ? a.xs + (size_t)(S.c[k] & LANE_MASK) * BS : a.xs + idle;
#pragma unroll
            for (int e = 0; e < BS; ++e) t[k][e] = __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((pend >> k) & 1u) {
                bool miss = false;
#pragma unroll
                for (int e = 0; e < BS; ++e) { D.xv[k][e] = t[k][e]; miss = miss || bl_missing(t[k][e]); }
                if (!miss) pend &= ~(1u << k);
            }
        if ((++spins & 1023u) == 0) {
            // a producer that never comes (not resident / an earlier time-out): give up together, quickly
            if (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
    long long t1 = 0;
    if (a.prof && lane == 0) t1 = wall_clock64();
    double acc[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) acc[r] = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int r = 0; r < BS; ++r) {
#pragma unroll
            for (int c = 0; c < BS; ++c) acc[r] = acc[r] + S.v[k][r * BS + c] * D.xv[k][c];      // padding: 0 x 0 (vals and a.zero)
        }
    }
#pragma unroll
    for (int r = 0; r < BS; ++r) acc[r] = bl_allreduce<L>(acc[r]);
    // lane q < bs of the block row: component q of Dinv_i (b_i - sum)   (relaxation.h:1283-1292)
    double nv = 0.0;
#pragma unroll
    for (int c = 0; c < BS; ++c) nv = nv + D.dv[c] * (D.bv[c] - acc[c]);
    if (q < BS && S.rid >= 0) {
        const size_t o = (size_t)S.rid * BS + q;
        if constexpr (MODE == 1) __hip_atomic_store(a.xs + o, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(a.xs + o, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.y[o] = nv;
    }
    if (a.prof && lane == 0) {
        long long *o = a.prof + (size_t)g * 4;
        o[0] = D.t0; o[1] = t1; o[2] = wall_clock64();
        o[3] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) | (blockIdx.x << 4));
    }
}

constexpr int BLANE_WPB = BLK / 64;

// A wave requests a group's blocks when it arrives at the group -- NOT the next group's while it waits for the current one (the scalar lane form's
// habit, and this kernel's first version): the memory counter is in order, so every poll of the current group would return behind the 9 ... 72 loads of
// the prefetch, and a wave typically arrives only 1.4 us before its last operand (0.26 us at the 10th percentile, stamps of the sweep).  One session,
// profiles/r05_microbench_blane_prefetch_ab.txt: level 0 0.316 -> 0.298 ms, level 1 0.223 -> 0.212.
template <int BS, int L, int K, int MODE>
__global__ __launch_bounds__(BLK) void bsr_lane_kernel(const BlaneArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int idle = (int)((((unsigned)blockIdx.x * BLANE_WPB + (unsigned)wib) * 16u) % (unsigned)a.nidle);
    BlaneSet<BS, K> P;
    BlaneDyn<BS, K> D;
    if constexpr (MODE != 1) {
        const int W = (int)gridDim.x * BLANE_WPB;
        for (int g = (int)blockIdx.x * BLANE_WPB + wib; g < a.ngroups; g += W) {
            blane_load<BS, L, K>(a, g, P);
            blane_issue<BS, L, K>(a, P, D, idle);
            blane_finish<BS, L, K, MODE>(a, P, D, g, idle);
        }
    } else {
        __shared__ int sh_home;
        if (threadIdx.x == 0) {
            const unsigned me = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) + 1u;     // HW_REG_XCC_ID[3:0] + 1
            unsigned home = 0u;
            __hip_atomic_compare_exchange_strong(a.ticket + 1, &home, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_home = (home == 0u || home == me) ? 1 : 0;
        }
        __syncthreads();
        if (!sh_home) return;
        // tickets: lane 0 draws, the wave reads lane 0's register; taken in increasing order by running waves: complete for any placement.
        // The next ticket is drawn while the current group waits (the atomic returns long before the group's operands do).
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int g = (int)__builtin_amdgcn_readfirstlane(tk);
        while (g < a.ngroups) {
            blane_load<BS, L, K>(a, g, P);
            blane_issue<BS, L, K>(a, P, D, idle);
            if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            blane_finish<BS, L, K, MODE>(a, P, D, g, idle);
            g = (int)__builtin_amdgcn_readfirstlane(tk);
        }
    }
}

__global__ __launch_bounds__(BLK) void blane_fill_sentinel_kernel(double *xs, int64_t n)
{
    unsigned long long *p = reinterpret_cast<unsigned long long *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) p[i] = BL_SENTINEL;
}

// ------------------------------------------------------------------ host side
namespace {

template <typename U>
int blane_upload(U **dst, const void *src, size_t bytes, size_t *total)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 256;
    PAMG_HIP(hipMalloc((void **)dst, alloc));
    if (bytes && src) PAMG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    if (total) *total += alloc;
    return PAMG_OK;
}

// two blocks per lane only for whole-wave block rows
template <int BS, int MODE>
const void *blane_kernel_l(int L, int K)
{
    if (K == 2) return L == 64 ? (const void *)bsr_lane_kernel<BS, 64, 2, MODE> : nullptr;
    switch (L) {
        case 8: return (const void *)bsr_lane_kernel<BS, 8, 1, MODE>;
        case 16: return (const void *)bsr_lane_kernel<BS, 16, 1, MODE>;
        case 32: return (const void *)bsr_lane_kernel<BS, 32, 1, MODE>;
        case 64: return (const void *)bsr_lane_kernel<BS, 64, 1, MODE>;
    }
    return nullptr;
}

template <int MODE>
const void *blane_kernel_bs(int bs, int L, int K)
{
    switch (bs) {
        case 2: return blane_kernel_l<2, MODE>(L, K);
        case 3: return blane_kernel_l<3, MODE>(L, K);
        case 4: return blane_kernel_l<4, MODE>(L, K);
        case 6: return blane_kernel_l<6, MODE>(L, K);
    }
    return nullptr;
}

const void *blane_kernel(int bs, int L, int K, int mode) { return mode == 1 ? blane_kernel_bs<1>(bs, L, K) : blane_kernel_bs<0>(bs, L, K); }

}  // namespace

void free_blane_part(BlaneSched *t)
{
    if (!t) return;
    (void)hipFree(t->d_cols); (void)hipFree(t->d_rid); (void)hipFree(t->d_gate); (void)hipFree(t->d_vals); (void)hipFree(t->d_prof); (void)hipFree(t->d_zero);
    delete t;
}

size_t blane_part_bytes(const GsSchedule *g) { return (g && g->blane) ? g->blane->bytes : 0; }

bool blane_eligible(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->dtype == PAMG_F64 && A->R == A->C && (A->R == 2 || A->R == 3 || A->R == 4 || A->R == 6) && g->nlevels > 1 && g->d_xs != nullptr && A->d_bAx != nullptr &&
           !g->blane_unfit;
}

// One-XCD form: the vectors (x, hand-off buffer) and the polling stay inside one XCD's 4 MB L2 (the rule of the scalar lane form) -- AND the blocks
// of a dependency level are few enough for ONE XCD's share of the memory system: a level of the 6 x 6 operator of the elasticity hierarchy
// (82 block rows of 64 slots, 1.5 MB) took 2.9 us through one XCD = its ~550 GB/s, profiles/r05_c5_blane_first_kernel_roofline.txt
static bool blane_one_xcd(const pamg_matrix_s *A, const GsSchedule *g)
{
    if (A->gran_xcd == 1) return true;
    if (A->gran_xcd != 0 || !g->blane) return false;
    const BlaneSched *t = g->blane;
    const int64_t level_bytes = t->nslots * (int64_t)(t->bs * t->bs * 8 + 4) / std::max(1, t->nlevels);
    return A->nrows <= 131072 && g->nrows / std::max(1, g->nlevels) <= 1024 && level_bytes <= 256 * 1024;
}

// built on the first sweep that asks for it (never inside a graph capture: the solver runs its sweeps once before it captures);
// PAMG_E_ARG: the form does not apply, the exact kernels keep the sweep
int build_blane_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->blane) return PAMG_OK;
    PhaseTimer pt_("build_blane_part", A->nnz);
    const size_t nval = (size_t)A->nblocks_b * A->R * A->C;
    std::vector<unsigned char> hAx(nval * 8);
    if (nval) PAMG_HIP(hipMemcpy(hAx.data(), A->d_bAx, nval * 8, hipMemcpyDeviceToHost));
    BlanePlan P;
    // lanes per block row: at least 8 (the kernels are compiled for 8 .. 64); across the chip one block row per wave pays as for the scalar rows
    // once a row fills half a wave (pamg_lane.hip: build_lane_part)
    int want_L = std::max(8, A->lane_L);
    if (build_blane_plan(A->n_brow, A->h_bAp.data(), A->h_bAj.data(), hAx.data(), 8, A->R, g->row_start, g->row_stop, g->row_step, P, want_L)) return PAMG_E_ARG;
    if (!blane_kernel(A->R, P.L, P.K, 0)) return PAMG_E_ARG;
    BlaneSched *t = new (std::nothrow) BlaneSched();
    if (!t) return PAMG_E_ALLOC;
    t->L = P.L; t->K = P.K; t->RPW = P.RPW; t->bs = P.bs; t->nlevels = P.nlevels; t->ngroups = P.ngroups; t->nslots = P.nslots;
    t->max_level_groups = P.max_level_groups; t->n_early = P.n_early; t->n_old = P.n_old;
    int st = blane_upload(&t->d_cols, P.cols.data(), P.cols.size() * sizeof(int), &t->bytes);
    if (!st) st = blane_upload(&t->d_rid, P.rid.data(), P.rid.size() * sizeof(int), &t->bytes);
    if (!st) st = blane_upload(&t->d_gate, P.gate.data(), P.gate.size() * sizeof(int), &t->bytes);
    if (!st) st = blane_upload(&t->d_vals, P.vals.data(), P.vals.size(), &t->bytes);
    if (!st) st = blane_upload(&t->d_zero, nullptr, 256, &t->bytes);
    if (!st) st = (int)hipMemset(t->d_zero, 0, 256);
    if (st) { free_blane_part(t); return st; }
    g->blane = t;
    g->bytes += t->bytes;
    return PAMG_OK;
}

static int blane_grid_cap(BlaneSched *t, const void *k)
{
    if (t->cap > 0 && t->cap_kernel == k) return t->cap;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLK, 0) != hipSuccess) nb = 2;
    nb = std::max(1, std::min(nb - 1, 8));                     // the query can over-report by one per CU (MI355X_MICROARCH.md)
    t->cap = nb; t->cap_kernel = k;
    return nb;
}

int blane_launch(pamg_matrix_s *A, GsSchedule *g, const void *Dinv, void *x, const void *b, hipStream_t s)
{
    BlaneSched *t = g->blane;
    if (!t) return PAMG_E_STATE;
    const int64_t n = A->nrows;
    BlaneArgs a;
    a.cols = t->d_cols; a.rid = t->d_rid; a.vals = t->d_vals;
    a.gate = (A->lane_flags & 1) ? t->d_gate : nullptr;
    a.x = (const double *)x; a.y = (double *)x; a.xs = (double *)g->d_xs; a.b = (const double *)b; a.Dinv = (const double *)Dinv; a.zero = t->d_zero;
    a.err = g->d_sync + 1; a.ticket = g->d_sync + 20;
    a.ngroups = (int)t->ngroups;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n - 8, 1 << 20));
    if (!g->symmetric) {
        // write-after-read hazards are not ordered by the waits: old values come from a snapshot
        if (!g->d_xold) return PAMG_E_STATE;
        PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        a.x = (const double *)g->d_xold;
    }
    if (A->gs_prof && !t->d_prof) {
        PAMG_HIP(hipMalloc((void **)&t->d_prof, (size_t)t->ngroups * 4 * sizeof(long long)));
        PAMG_HIP(hipMemset(t->d_prof, 0, (size_t)t->ngroups * 4 * sizeof(long long)));
    }
    a.prof = A->gs_prof ? t->d_prof : nullptr;
    const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
    hipLaunchKernelGGL(blane_fill_sentinel_kernel, dim3(fgrid), dim3(BLK), 0, s, (double *)g->d_xs, n);
    PAMG_HIP(hipGetLastError());
    const bool xcd = blane_one_xcd(A, g);
    const void *k = blane_kernel(A->R, t->L, t->K, xcd ? 1 : 0);
    if (!k) return PAMG_E_ARG;
    static thread_local int cus = 0;
    if (!cus) cus = device_cus_lane();
    const int cap = blane_grid_cap(t, k);
    const int per_level = (int)((t->ngroups + t->nlevels - 1) / std::max(1, t->nlevels));
    // Look-ahead (a wave that runs ahead waits in its poll loop -- on its gate -- with its blocks in registers): three dependency levels where a
    // wave holds several block rows, six with one block row per wave (its blocks are the longer fetch: 18 KB per 6 x 6 row) -- the elasticity
    // hierarchy, profiles/r05_microbench_blane_grid.json (after the prefetch went): level 0 (342 groups per level) 0.289 ms with 256 workgroups, 0.321 with 512,
    // 0.316 with 128; level 1 (82 per level) 0.210 ms with 128 .. 256, 0.212 with 64, 0.253 with 32
    const int64_t want_waves = std::max<int64_t>(128, (int64_t)(t->RPW == 1 ? 6 : 3) * per_level);
    int G = (int)std::min<int64_t>((want_waves + BLANE_WPB - 1) / BLANE_WPB, (int64_t)cap * cus);
    if (G > cus && G <= cus + cus / 4) G = cus;                 // a few workgroups beyond one per CU double up on some CUs: level 0 of the elasticity hierarchy 0.299 ms with 257, 0.289 with 256
    if (A->lane_G > 0) G = std::min(A->lane_G, cap * cus);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, (t->ngroups + BLANE_WPB - 1) / BLANE_WPB));
    void *args[] = {(void *)&a};
    if (xcd) {
        PAMG_HIP(hipMemsetAsync(g->d_sync + 20, 0, 2 * sizeof(unsigned), s));
        const int Gx = std::max(1, std::min(G, (cus / 8) * cap));
        t->last_grid = 8 * Gx;
        PAMG_HIP(hipLaunchKernel(k, dim3(8 * Gx), dim3(BLK), args, 0, s));
        return PAMG_OK;
    }
    t->last_grid = G;
    PAMG_HIP(hipLaunchKernel(k, dim3(G), dim3(BLK), args, 0, s));
    return PAMG_OK;
}

// info[0..7] = lanes per block row, blocks per lane, groups, block slots, early blocks, workgroups of the last launch, widest level (groups), bytes
int blane_info(const GsSchedule *g, int64_t *info)
{
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (!g || !g->blane) return PAMG_OK;
    const BlaneSched *t = g->blane;
    info[0] = t->L; info[1] = t->K; info[2] = t->ngroups; info[3] = t->nslots; info[4] = t->n_early; info[5] = t->last_grid;
    info[6] = t->max_level_groups; info[7] = (int64_t)t->bytes;
    return PAMG_OK;
}

int blane_profile(const GsSchedule *g, long long *out, int64_t cap, int64_t *n)
{
    *n = 0;
    if (!g || !g->blane || !g->blane->d_prof) return PAMG_OK;
    *n = g->blane->ngroups;
    if (out && cap >= *n) PAMG_HIP(hipMemcpy(out, g->blane->d_prof, (size_t)*n * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    return PAMG_OK;
}

}  // namespace pamg
