// pamg_wg.hip -- the WORKGROUP-RESIDENT fast-order Gauss-Seidel / SOR sweep: layout and rationale in pamg_wg_plan.h.
//
// Same rows in the same order as amg_core::gauss_seidel (relaxation.h:48-76) / sor_gauss_seidel (:116-145) / bsr_gauss_seidel with
// 1x1 blocks (:185-266), the row arithmetic of the lane form (pamg_lane.hip: L lanes share a row, K products per lane, butterfly,
// (b - sum) * (1 / a_ii)) -- but the iterate of a tile lives in LDS and the dependency levels of a tile are separated by an LDS-only
// barrier of its 16 waves instead of a hand-off through memory: 0.2 .. 0.3 us per dependency level where the lane form pays 0.7 (one
// XCD) or 1.05 us (chip).  One workgroup per tile, one CU each; only operands of OTHER tiles travel through memory (sentinel
// hand-off for new values of earlier tiles, plain reads for old values).
//
// A wave's round.  The waves of a tile move in lock step from barrier to barrier, so nothing hides a memory round trip unless it was started
// rounds earlier: the static operands (slots, values, record, and the round's right-hand sides, gathered into slot order by a small kernel before
// the sweep) are requested THREE rounds ahead, the few operands that come from x in memory (old values outside the tile, the first poll of another
// tile's new values) ONE round ahead.  Most rounds of a narrow schedule are far from full: a round's flags say how many waves hold a row at all and
// how many slots per lane its longest row needs, and the kernel asks memory for nothing beyond either (the loads stay unconditional -- a load under
// a branch makes the compiler drain the memory counter at the next wait -- but point at one resident line).
#include "pamg_common.h"
#include "pamg_wg_plan.h"

namespace pamg {

template <typename T> struct WgSentinel;
template <> struct WgSentinel<double> {
    using bits_t = unsigned long long;
    static constexpr bits_t value = 0x7FF8DEADBEEF5A5Aull;
    static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};
template <> struct WgSentinel<float> {
    using bits_t = unsigned int;
    static constexpr bits_t value = 0x7FC5BEEFu;
    static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};

template <typename T>
__global__ __launch_bounds__(BLK) void wg_fill_sentinel_kernel(T *xs, int64_t n)
{
    using B = typename WgSentinel<T>::bits_t;
    B *p = reinterpret_cast<B *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) p[i] = WgSentinel<T>::value;
}

struct alignas(16) WgRec { int rid; int lpos; int rd_lo; int rd_hi; };

struct WgSched {
    int L = 0, K = 0, RPW = 0, G = 0, tile_rows = 0;
    int64_t ngroups = 0, nrounds = 0;
    int *d_cols = nullptr, *d_rflag = nullptr, *d_tile_round = nullptr, *d_tile_vis0 = nullptr;
    void *d_vals = nullptr, *d_bperm = nullptr;
    WgRec *d_rec = nullptr;
    int *d_rid = nullptr;
    int64_t n_intile = 0, n_cross = 0, n_old = 0;
    size_t bytes = 0;
};

template <typename T>
struct WgArgs {
    const int *cols;
    const T *vals;
    const WgRec *rec;
    const int *rflag, *tile_round, *tile_vis0;
    const T *x;            // OLD values outside the tile (x itself, or its snapshot for structurally non-symmetric patterns)
    const T *xin;          // the live x: what the tiles load into LDS
    T *y;                  // destination (the live x)
    T *xs;                 // hand-off buffer between tiles, sentinel-filled (nullptr when there is one tile)
    const T *b;            // right-hand side
    T *bperm;              // the same in slot-row order (wg_gather_b_kernel, before every sweep)
    const int *rid;        // slot row -> row | flags (the plan's array, for the gather)
    unsigned *err;
    int row_start, row_step, nidle;
    int64_t nslotrows;
    T omega;
};

constexpr int WG_THREADS = 64 * WG_NW;

template <int CTRL>
__device__ __forceinline__ double wg_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float wg_dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)); }
__device__ __forceinline__ double wg_swz16(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wg_swz16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }

// the lane form's butterfly (pamg_lane.hip: seg_allreduce), same steps in the same order
template <int L, typename T>
__device__ __forceinline__ T wg_allreduce(T v)
{
    v = v + wg_dpp<0xB1>(v);
    v = v + wg_dpp<0x4E>(v);
    if constexpr (L >= 8) v = v + wg_dpp<0x141>(v);
    if constexpr (L >= 16) v = v + wg_dpp<0x140>(v);
    if constexpr (L >= 32) v = v + wg_swz16(v);
    if constexpr (L >= 64) v = v + __shfl_xor(v, 32);
    return v;
}

template <typename T, int K>
struct WgStat {            // static operands of one group
    int c[K];
    T v[K];
    int rid, lpos;
    T rd, bv;
};
template <typename T, int K>
struct WgDyn {             // operands from x in memory
    T xg[K];
};

// flags of a round (pamg_wg_plan.h): active waves, slots per lane in use
__device__ __forceinline__ int wg_nact(int f) { return (f >> 8) & 255; }
__device__ __forceinline__ int wg_kuse(int f) { return (f >> 16) & 15; }
// what a record's second word carries above the row's LDS position: bit 16 = barrier before this round, bits 17.. = flags of the round three ahead
__device__ __forceinline__ int wg_ahead(int lposword) { return __builtin_amdgcn_readfirstlane(lposword) >> 17 << 8; }     // -> (nact << 8) | (kuse << 16) layout of wg_nact / wg_kuse
__device__ __forceinline__ bool wg_bar(int lposword) { return (__builtin_amdgcn_readfirstlane(lposword) >> 16) & 1; }

template <typename T, int L, int K>
__device__ __forceinline__ void wg_load(const WgArgs<T> &a, int64_t g, int flags, int wib, WgStat<T, K> &S)
{
    const int lane = threadIdx.x & 63;
    const bool on = wib < wg_nact(flags);
    const int kuse = wg_kuse(flags);
    const size_t e0 = (size_t)g * (size_t)(K * 64) + (size_t)lane;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t e = (on && k < kuse) ? e0 + (size_t)k * 64 : (size_t)lane;      // beyond the round's needs: the array's first line, resident
        S.c[k] = a.cols[e];
        S.v[k] = a.vals[e];
        if (!(on && k < kuse)) S.c[k] = WG_NONE;
    }
    // the record is always the group's own (16 bytes per slot row): besides the row it carries the round's barrier flag and the flags of the
    // round three ahead -- what the NEXT static request needs to know before it is issued -- so that no load of its own stands in the pipeline
    const size_t sr = (size_t)g * (size_t)(64 / L) + (size_t)(lane / L);
    const int4 q = reinterpret_cast<const int4 *>(a.rec)[sr];
    S.rid = on ? q.x : -1; S.lpos = q.y;
    if constexpr (sizeof(T) == 8) S.rd = __hiloint2double(q.w, q.z);
    else S.rd = __int_as_float(q.z);
    S.bv = a.bperm[sr];
}

// b in slot-row order (one value per slot row; dummy rows get 0)
template <typename T>
__global__ __launch_bounds__(BLK) void wg_gather_b_kernel(const int *__restrict__ rid, const T *__restrict__ b, T *__restrict__ bperm, int64_t nslotrows)
{
    const int64_t q = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (q >= nslotrows) return;
    const int r = rid[q];
    bperm[q] = r < 0 ? T(0) : b[r & WG_MASK];
}

// ONE: the sweep has one tile -- nobody else writes x during the launch, its old values may come through the L1 like any other load
template <typename T, int K, bool ONE>
__device__ __forceinline__ void wg_issue(const WgArgs<T> &a, const WgStat<T, K> &S, WgDyn<T, K> &D, int idle)
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = S.c[k];
        const bool mem = !(c & WG_INTILE) && !(c & WG_NONE);
        const T *p = !mem ? a.x + idle : ((c & WG_CROSS) ? a.xs + (c & WG_MASK) : a.x + (c & WG_MASK));
        if constexpr (ONE) D.xg[k] = *p;
        else D.xg[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int EPI, int L, int K>
__device__ __forceinline__ void wg_compute(const WgArgs<T> &a, const WgStat<T, K> &S, WgDyn<T, K> &D, T *xt, int idle)
{
    const int lane = threadIdx.x & 63;
    const bool head = (lane & (L - 1)) == 0;
    // new values of earlier tiles that had not arrived a round ago: poll (rare: a tile runs a hand-off behind the one before it)
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = S.c[k];
        if (!(c & WG_INTILE) && !(c & WG_NONE) && (c & WG_CROSS) && WgSentinel<T>::bits(D.xg[k]) == WgSentinel<T>::value) pend |= 1u << k;
    }
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(pend != 0)) {
        T t[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            t[k] = __hip_atomic_load(((pend >> k) & 1u) ? a.xs + (S.c[k] & WG_MASK) : a.x + idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((pend >> k) & 1u) {
                D.xg[k] = t[k];
                if (WgSentinel<T>::bits(t[k]) != WgSentinel<T>::value) pend &= ~(1u << k);
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
        const int c = S.c[k];
        const T xv = (c & WG_INTILE) ? xt[c & WG_MASK] : D.xg[k];
        const T pr = S.v[k] * xv;
        s = s + ((!(c & WG_INTILE) && (c & WG_NONE)) ? T(0) : pr);
    }
    s = wg_allreduce<L, T>(s);
    if (head && S.rid >= 0) {
        const int row = S.rid & WG_MASK;
        const bool upd = !(S.rid & WG_NODIAG);
        const int lp = S.lpos & 0xFFFF;
        const T xo = xt[lp];
        T v = (S.bv - s) * S.rd;
        if constexpr (EPI == EPI_SOR) v = a.omega * v + (T(1) - a.omega) * xo;
        if (!upd) v = xo;
        xt[lp] = v;
        if (S.rid & WG_PUBLISH) __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (upd) a.y[row] = v;
    }
}

template <typename T, int EPI, int L, int K, bool ONE>
__global__ __launch_bounds__(WG_THREADS) void gs_wg_kernel(const WgArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    T *xt = reinterpret_cast<T *>(wg_smem);
    const int tile = (int)blockIdx.x;
    const int wib = threadIdx.x >> 6;
    const int idle = (int)(((unsigned)tile * WG_NW + (unsigned)wib) * 16u % (unsigned)a.nidle);
    const int t0 = a.tile_vis0[tile], t1 = a.tile_vis0[tile + 1];
    for (int p = (int)threadIdx.x; p < t1 - t0; p += WG_THREADS) xt[p] = a.xin[a.row_start + (int64_t)(t0 + p) * a.row_step];
    const int r0 = a.tile_round[tile], r1 = a.tile_round[tile + 1];
    if (r0 >= r1) return;
    // round r: static operands three rounds ahead (S0 .. S3 in turn), memory operands one round ahead (D0, D1 in turn): four rounds per trip of the
    // loop below so that every register set has a compile-time name
    WgStat<T, K> S0, S1, S2, S3;
    WgDyn<T, K> D0, D1;
    auto rr = [&](int r) { return min(r, r1 - 1); };
    auto gidx = [&](int r) { return (int64_t)rr(r) * WG_NW + wib; };
    wg_load<T, L, K>(a, gidx(r0), a.rflag[rr(r0)], wib, S0);
    wg_load<T, L, K>(a, gidx(r0 + 1), a.rflag[rr(r0 + 1)], wib, S1);
    wg_load<T, L, K>(a, gidx(r0 + 2), a.rflag[rr(r0 + 2)], wib, S2);
    wg_issue<T, K, ONE>(a, S0, D0, idle);
    // (the first barrier -- the first round of a tile starts a level -- also covers the load of xt above)
#define PAMG_WG_ROUND(SC, SN, SP, DC, DN)                                                                       \
    {                                                                                                           \
        wg_issue<T, K, ONE>(a, SN, DN, idle);                                     /* memory operands of round r + 1 */ \
        wg_load<T, L, K>(a, gidx(r + 3), wg_ahead(SC.lpos), wib, SP);        /* static operands of round r + 3 */ \
        if (wg_bar(SC.lpos)) wg_barrier();                                                                      \
        wg_compute<T, EPI, L, K>(a, SC, DC, xt, idle);                                                          \
        if (++r >= r1) break;                                                                                   \
    }
    int r = r0;
    while (true) {
        PAMG_WG_ROUND(S0, S1, S3, D0, D1)
        PAMG_WG_ROUND(S1, S2, S0, D1, D0)
        PAMG_WG_ROUND(S2, S3, S1, D0, D1)
        PAMG_WG_ROUND(S3, S0, S2, D1, D0)
    }
#undef PAMG_WG_ROUND
}

// ------------------------------------------------------------------ host side
namespace {

template <typename U>
int wg_upload(U **dst, const void *src, size_t bytes, size_t *total)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 256;
    PAMG_HIP(hipMalloc((void **)dst, alloc));
    if (bytes && src) PAMG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    if (total) *total += alloc;
    return PAMG_OK;
}

template <typename T, int EPI, int L, bool ONE>
const void *wg_kernel_k(int K)
{
    switch (K) {
        case 1: return (const void *)gs_wg_kernel<T, EPI, L, 1, ONE>;
        case 2: return (const void *)gs_wg_kernel<T, EPI, L, 2, ONE>;
        case 3: return (const void *)gs_wg_kernel<T, EPI, L, 3, ONE>;
        case 4: return (const void *)gs_wg_kernel<T, EPI, L, 4, ONE>;
    }
    return nullptr;
}

template <typename T, int EPI, bool ONE>
const void *wg_kernel_l(int L, int K)
{
    switch (L) {
        case 4: return wg_kernel_k<T, EPI, 4, ONE>(K);
        case 8: return wg_kernel_k<T, EPI, 8, ONE>(K);
        case 16: return wg_kernel_k<T, EPI, 16, ONE>(K);
        case 32: return wg_kernel_k<T, EPI, 32, ONE>(K);
        case 64: return wg_kernel_k<T, EPI, 64, ONE>(K);
    }
    return nullptr;
}

}  // namespace

void free_wg_part(WgSched *t)
{
    if (!t) return;
    hipFree(t->d_cols); hipFree(t->d_rflag); hipFree(t->d_tile_round); hipFree(t->d_tile_vis0); hipFree(t->d_vals); hipFree(t->d_rec); hipFree(t->d_bperm); hipFree(t->d_rid);
    delete t;
}

constexpr size_t WG_LDS_BYTES = 144 * 1024;    // of the CU's 160 KB: the tile's x values

// the form pays where a dependency level is a few rounds of one workgroup: narrow schedules of operators whose swept rows fit a few tiles
bool wg_eligible(const pamg_matrix_s *A, const GsSchedule *g)
{
    if (A->R != 1 || g->nlevels <= 1 || !g->d_xs || A->max_row_len > LANE_KMAX * 64 + 1 || g->wg_unfit) return false;
    const int64_t cap = (int64_t)(WG_LDS_BYTES / tsize(A->dtype));
    if (g->nrows > (int64_t)WG_MAX_TILES * cap) return false;
    return g->nrows / std::max(1, g->nlevels) <= 96;
}

int build_wg_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->wg) return PAMG_OK;
    PhaseTimer pt_("build_wg_part", A->nnz);
    const int ts = (int)tsize(A->dtype);
    std::vector<unsigned char> hAx((size_t)A->nnz * ts);
    if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    WgPlan P;
    if (build_wg_plan((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), hAx.data(), ts, g->row_start, g->row_step, (int)g->nrows, g->nlevels,
                      g->h_vis, g->h_lvl, (int)(WG_LDS_BYTES / ts), P, A->wg_tiles))
        return PAMG_E_ARG;
    WgSched *t = new (std::nothrow) WgSched();
    if (!t) return PAMG_E_ALLOC;
    t->L = P.L; t->K = P.K; t->RPW = P.RPW; t->G = P.G; t->tile_rows = P.tile_rows;
    t->ngroups = P.ngroups; t->nrounds = P.nrounds;
    t->n_intile = P.n_intile; t->n_cross = P.n_cross; t->n_old = P.n_old;
    const size_t nrec = (size_t)P.ngroups * P.RPW;
    std::vector<WgRec> rec(nrec);
    for (size_t q = 0; q < nrec; ++q) {
        // second word: LDS position (< 2^16 values per tile) | barrier flag of the group's round << 16 | (active waves, slots in use) of the round three ahead << 17
        const int64_t round = (int64_t)(q / (size_t)P.RPW) / WG_NW;
        int tile = 0;
        while (tile + 1 < P.G && round >= P.tile_round[(size_t)tile + 1]) ++tile;
        const int64_t ahead = std::min<int64_t>(round + 3, (int64_t)P.tile_round[(size_t)tile + 1] - 1);
        const int fa = P.rflag[(size_t)ahead];
        rec[q].rid = P.rid[q];
        rec[q].lpos = P.lpos[q] | ((P.rflag[(size_t)round] & 1) << 16) | (((fa >> 8) & 31) << 17) | (((fa >> 16) & 15) << 25);
        rec[q].rd_lo = rec[q].rd_hi = 0;
        std::memcpy(&rec[q].rd_lo, &P.rdiag[q * (size_t)ts], (size_t)ts);
    }
    int st = wg_upload(&t->d_rec, rec.data(), nrec * sizeof(WgRec), &t->bytes);
    if (!st) st = wg_upload(&t->d_cols, P.cols.data(), P.cols.size() * sizeof(int), &t->bytes);
    if (!st) st = wg_upload(&t->d_vals, P.vals.data(), P.vals.size(), &t->bytes);
    if (!st) st = wg_upload(&t->d_rflag, P.rflag.data(), P.rflag.size() * sizeof(int), &t->bytes);
    if (!st) st = wg_upload(&t->d_tile_round, P.tile_round.data(), P.tile_round.size() * sizeof(int), &t->bytes);
    if (!st) st = wg_upload(&t->d_tile_vis0, P.tile_vis0.data(), P.tile_vis0.size() * sizeof(int), &t->bytes);
    if (!st) st = wg_upload(&t->d_rid, P.rid.data(), P.rid.size() * sizeof(int), &t->bytes);
    if (!st) st = wg_upload(&t->d_bperm, nullptr, nrec * (size_t)ts, &t->bytes);
    if (st) { free_wg_part(t); return st; }
    g->wg = t;
    g->bytes += t->bytes;
    return PAMG_OK;
}

size_t wg_part_bytes(const GsSchedule *g) { return (g && g->wg) ? g->wg->bytes : 0; }

template <typename T>
static int wg_launch_t(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    WgSched *t = g->wg;
    const size_t ts = tsize(A->dtype);
    const int64_t n = A->nrows;
    WgArgs<T> a;
    a.cols = t->d_cols; a.vals = (const T *)t->d_vals; a.rec = t->d_rec;
    a.rflag = t->d_rflag; a.tile_round = t->d_tile_round; a.tile_vis0 = t->d_tile_vis0;
    a.x = (const T *)x; a.xin = (const T *)x; a.y = (T *)x; a.xs = (T *)g->d_xs; a.b = (const T *)b;
    a.bperm = (T *)t->d_bperm; a.rid = t->d_rid; a.nslotrows = t->ngroups * t->RPW;
    a.err = g->d_sync + 1;
    a.row_start = g->row_start; a.row_step = g->row_step;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
    a.omega = (T)omega;
    if (!g->symmetric && (t->n_old > 0 || t->G > 1)) {
        // old values OUTSIDE a tile are read from memory while other tiles write: a snapshot (inside a tile the barriers order everything)
        if (!g->d_xold) return PAMG_E_STATE;
        PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
        a.x = (const T *)g->d_xold;
    }
    if (t->G > 1) {
        const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
        hipLaunchKernelGGL((wg_fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
        PAMG_HIP(hipGetLastError());
    }
    const void *k = t->G == 1 ? (epi == EPI_SOR ? wg_kernel_l<T, EPI_SOR, true>(t->L, t->K) : wg_kernel_l<T, EPI_GS, true>(t->L, t->K))
                              : (epi == EPI_SOR ? wg_kernel_l<T, EPI_SOR, false>(t->L, t->K) : wg_kernel_l<T, EPI_GS, false>(t->L, t->K));
    if (!k) return PAMG_E_ARG;
    hipLaunchKernelGGL((wg_gather_b_kernel<T>), dim3((unsigned)((a.nslotrows + BLK - 1) / BLK)), dim3(BLK), 0, s, a.rid, a.b, a.bperm, a.nslotrows);
    PAMG_HIP(hipGetLastError());
    const size_t lds = std::max<size_t>((size_t)t->tile_rows * ts, 1024);
    if (lds > 48 * 1024) PAMG_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void *args[] = {(void *)&a};
    PAMG_HIP(hipLaunchKernel(k, dim3(t->G), dim3(WG_THREADS), args, lds, s));
    return PAMG_OK;
}

int wg_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    if (!g->wg) return PAMG_E_STATE;
    if (A->dtype == PAMG_F64) return wg_launch_t<double>(A, g, epi, x, b, omega, s);
    return wg_launch_t<float>(A, g, epi, x, b, omega, s);
}

// info[0..7] = lanes per row, slots per lane, tiles, rounds, in-tile operands, operands polled from other tiles, old operands from memory, bytes
int wg_info(const GsSchedule *g, int64_t *info)
{
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (!g || !g->wg) return PAMG_OK;
    const WgSched *t = g->wg;
    info[0] = t->L; info[1] = t->K; info[2] = t->G; info[3] = t->nrounds; info[4] = t->n_intile; info[5] = t->n_cross; info[6] = t->n_old;
    info[7] = (int64_t)t->bytes;
    return PAMG_OK;
}

}  // namespace pamg
