// pamg_lane.hip -- the LANE-PARALLEL ("fast order") Gauss-Seidel / SOR sweep: layout in pamg_lane_plan.h.
//
// Same rows in the same order as amg_core::gauss_seidel (relaxation.h:48-76) / sor_gauss_seidel (:116-145) /
// bsr_gauss_seidel with 1x1 blocks (:185-266) -- the dependency DAG of the sequential sweep is kept, so the iterates
// are the reference's up to rounding -- but nothing of the row sum's order is: L lanes of a wave share a row, each adds
// its K products, the lanes are added across the wave ("wavefront-wide segmented reduction per row"), and the row is
// finished with (b - sum) * (1 / a_ii) instead of the IEEE division.  What that buys: the order-exact sweeps spend
// ~0.9 us per dependency level BEHIND the last hand-off (LDS staging, barrier, 31..70-term in-order add chain, divide).
//
// ONE persistent launch per sweep; a group of 64 / L rows of one dependency level is the work of ONE wave, waves never
// meet (no LDS, no barrier).  The hand-off is the one of the granular exact sweep: the published 8-byte value is the
// flag (sentinel-filled buffer xs, write-through store -> polled L1-bypassing load).
//   static form : wave w takes groups w, w + W, w + 2W, ... (all W waves co-resident; a group only waits for groups
//                 with smaller numbers -> deadlock-free);
//   one-XCD form: small operators (vectors fit one XCD's L2).  The first workgroup to arrive claims its XCD, workgroups
//                 elsewhere leave, the rest draw groups from a ticket counter (two tickets ahead, taken in increasing
//                 order by running waves: complete for any placement) and publish with ordinary L2-resident stores.
// In the ticket form the static operands of a wave's NEXT group are requested before it starts to wait for the current one (the static form stopped
// doing that in round 5: - 1.4 % on level 1 of the 256^3 hierarchy; the ticket form WITHOUT it: level 2 0.705 -> 0.745 ms -- inside one XCD a poll
// costs little and the operands of the next group are the longer wait; profiles/r05_microbench_lane_prefetch_ab.txt).
//
// Round 5: the slab form (one slab of the visit order per XCD, hand-off through the XCD's L2 for operands of the own slab) is gone -- it never
// beat the plain static form (profiles/r04_microbench_lane_exp_{g,h}.json) --, rid / gate / 1 / a_ii of a slot row travel as ONE 16-byte record,
// and three forms of a shorter tail behind the last hand-off were built, measured and removed again (early products added one by one by
// v_readlane in a producer-level slot order; the last two / three early slots polled by every lane at one address each): the butterfly is not
// what a dependency level costs -- profiles/r05_microbench_lane_tail_*_not_kept.json, DESIGN 3.
#include "pamg_common.h"
#include "pamg_lane_plan.h"
#include "pamg_lanem_plan.h"

namespace pamg {

// Sentinel bit patterns of the hand-off buffer (the ones of the exact sweeps, pamg_kernels.h): xs[j] == sentinel  <=>  row j
// has not published its new value in this sweep yet.
template <typename T> struct Sentinel;
template <> struct Sentinel<double> {
    using bits_t = unsigned long long;
    static constexpr bits_t value = 0x7FF8DEADBEEF5A5Aull;
    static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};
template <> struct Sentinel<float> {
    using bits_t = unsigned int;
    static constexpr bits_t value = 0x7FC5BEEFu;
    static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};

template <typename T>
__global__ __launch_bounds__(BLK) void lane_fill_sentinel_kernel(T *xs, int64_t n)
{
    using B = typename Sentinel<T>::bits_t;
    B *p = reinterpret_cast<B *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) p[i] = Sentinel<T>::value;
}

// per slot row (group g, row r of the group): original row | NODIAG, the group's gate operand, 1 / a_ii -- ONE 16-byte request
struct alignas(16) LaneRec { int rid; int gate; int rd_lo; int rd_hi; };

struct LaneSched {
    int L = 0, K = 0, RPW = 0;
    int64_t ngroups = 0;
    int *d_cols = nullptr;
    LaneRec *d_rec = nullptr;
    void *d_vals = nullptr;
    long long *d_prof = nullptr;
    int64_t n_early = 0, n_old = 0, n_slots = 0;
    int64_t max_level_groups = 0;
    int last_grid = 0;              // workgroups of the last launch (diagnostics)
    int cap = 0;                    // co-resident workgroups per CU of this schedule's kernel (queried once)
    const void *cap_kernel = nullptr;
    size_t bytes = 0;
};

template <typename T>
struct LaneArgs {
    const int *cols;
    const T *vals;
    const LaneRec *rec;
    const T *x;            // OLD values (x itself, or its snapshot for structurally non-symmetric patterns)
    T *y;                  // destination (the live x)
    T *xs;                 // hand-off buffer, sentinel-filled
    const T *b;
    unsigned *err;         // spin bound hit
    unsigned *ticket;      // one-XCD form: [0] ticket counter, [1] home XCD + 1
    long long *prof;       // nullptr or [ngroups][4] time stamps
    int ngroups, nidle;
    int use_gate;          // != 0: a wave that runs ahead polls its gate operand first
    T omega;
};

template <typename T, int K>
struct LaneSet {
    int c[K];
    T v[K];
    int rid;
    T rd;
    int gate;
};

// ---- sum over the L lanes that share a row; every lane ends up with the total (bit for bit the same total: an XOR butterfly
//      adds the same two values in both lanes of a pair at every step)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);       // (no `old` operand: every lane has a source, the compiler needs no copy)
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// lane ^ 16 and lane ^ 32 without the LDS path (round 5; ds_swizzle + ds_bpermute before): v_permlane16_swap / v_permlane32_swap (CDNA4) exchange
// the odd 16-lane rows (the upper half) of one copy with the even rows (the lower half) of the other; afterwards one copy holds the even-row
// (lower-half) values everywhere, the other the odd-row (upper-half) ones, and their sum is the butterfly step
template <bool HALF>
__device__ __forceinline__ double swap_sum(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (HALF) {
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    } else {
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    }
}
template <bool HALF>
__device__ __forceinline__ float swap_sum(float v)
{
    const int w = __float_as_int(v);
    if constexpr (HALF) {
        const auto a = __builtin_amdgcn_permlane32_swap(w, w, false, false);
        return __int_as_float(a[0]) + __int_as_float(a[1]);
    } else {
        const auto a = __builtin_amdgcn_permlane16_swap(w, w, false, false);
        return __int_as_float(a[0]) + __int_as_float(a[1]);
    }
}

template <int L, typename T>
__device__ __forceinline__ T seg_allreduce(T v)
{
    v = v + dpp_mov<0xB1>(v);                                   // quad_perm [1,0,3,2]: lane ^ 1
    v = v + dpp_mov<0x4E>(v);                                   // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (L >= 8) v = v + dpp_mov<0x141>(v);            // row_half_mirror: the other quad of the 8
    if constexpr (L >= 16) v = v + dpp_mov<0x140>(v);           // row_mirror: the other half of the 16
    if constexpr (L >= 32) v = swap_sum<false>(v);              // lane ^ 16
    if constexpr (L >= 64) v = swap_sum<true>(v);               // lane ^ 32
    return v;
}

template <typename T> __device__ __forceinline__ T rec_rd(const int4 &q);
template <> __device__ __forceinline__ double rec_rd<double>(const int4 &q) { return __hiloint2double(q.w, q.z); }
template <> __device__ __forceinline__ float rec_rd<float>(const int4 &q) { return __int_as_float(q.z); }

template <typename T, int L, int K>
__device__ __forceinline__ void lane_load(const LaneArgs<T> &a, int g, LaneSet<T, K> &S)
{
    const int lane = threadIdx.x & 63;
    const size_t e0 = (size_t)g * (size_t)(K * 64) + (size_t)lane;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        S.c[k] = a.cols[e0 + (size_t)k * 64];
        S.v[k] = a.vals[e0 + (size_t)k * 64];
    }
    const size_t slot = (size_t)g * (size_t)(64 / L) + (size_t)(lane / L);
    const int4 q = reinterpret_cast<const int4 *>(a.rec)[slot];
    S.rid = q.x;
    S.gate = a.use_gate ? q.y : -1;
    S.rd = rec_rd<T>(q);
}

// one group, first half: request everything that depends on the group's static operands -- b, the row's own old value,
// and the first round of operand loads (early entries from the hand-off buffer, the others from x; both bypass the L1,
// other CUs write these lines during the launch; padding reads a per-wave idle element).  The caller requests the
// NEXT group's static operands right after this, so that waiting for these loads (the memory counter is in-order)
// never waits for those.
template <typename T, int K>
struct LaneDyn {
    T bv, xo;
    T xv[K];
    long long t0;
};

template <typename T, int EPI, int K>
__device__ __forceinline__ void lane_issue(const LaneArgs<T> &a, const LaneSet<T, K> &S, LaneDyn<T, K> &D, int idle)
{
    D.t0 = 0;
    if (a.prof && (threadIdx.x & 63) == 0) D.t0 = wall_clock64();
    const int row = S.rid < 0 ? 0 : (S.rid & LANE_MASK);
    D.bv = a.b[row];
    D.xo = T(0);
    if constexpr (EPI == EPI_SOR) D.xo = a.x[row];
    else if (S.rid & LANE_NODIAG) D.xo = a.x[row];             // (dummy rows have the bit set too: harmless)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = S.c[k];
        const int col = c & LANE_MASK;
        const T *p = (c & LANE_NONE) ? a.x + idle : ((c & LANE_EARLY) ? a.xs + col : a.x + col);
        D.xv[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// a wave that runs ahead: the sweep is still two or more dependency levels away while the gate operand is missing -- the
// whole wave polls that ONE value (one request per round) instead of all its operands
template <typename T>
__device__ __forceinline__ unsigned lane_gate_wait(const LaneArgs<T> &a, int gate)
{
    unsigned spins = 0;
    const T *gp = a.xs + gate;
    while (true) {
        const T gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (Sentinel<T>::bits(gv) != Sentinel<T>::value) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023u) == 0 && (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
    }
    return spins;
}

template <typename T, int EPI, int MODE>
__device__ __forceinline__ void lane_publish(const LaneArgs<T> &a, int rid, T s, T bv, T rd, T xo)
{
    const int row = rid & LANE_MASK;
    const bool upd = !(rid & LANE_NODIAG);
    T v = (bv - s) * rd;
    if constexpr (EPI == EPI_SOR) v = a.omega * v + (T(1) - a.omega) * xo;
    if (!upd) v = xo;
    if constexpr (MODE == 1) __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (upd) a.y[row] = v;
}

// second half, butterfly form: wait for the early operands, row sums across the lanes, publish
template <typename T, int EPI, int L, int K, int MODE>
__device__ __forceinline__ void lane_finish(const LaneArgs<T> &a, const LaneSet<T, K> &S, LaneDyn<T, K> &D, int g, int idle)
{
    const int lane = threadIdx.x & 63;
    const bool head = (lane & (L - 1)) == 0;
    long long t1 = 0;
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((S.c[k] & LANE_EARLY) && !(S.c[k] & LANE_NONE) && Sentinel<T>::bits(D.xv[k]) == Sentinel<T>::value) pend |= 1u << k;
    unsigned spins = 0;
    if (S.gate >= 0 && __builtin_amdgcn_ballot_w64(pend != 0)) spins = lane_gate_wait<T>(a, S.gate);
    while (pend) {
        if (spins) __builtin_amdgcn_s_sleep(1);
        T t[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            t[k] = __hip_atomic_load(((pend >> k) & 1u) ? a.xs + (S.c[k] & LANE_MASK) : a.xs + idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if ((pend >> k) & 1u) {
                D.xv[k] = t[k];
                if (Sentinel<T>::bits(t[k]) != Sentinel<T>::value) pend &= ~(1u << k);
            }
        if ((++spins & 1023u) == 0) {
            // a producer that never comes (not resident / an earlier time-out): give up together, quickly
            if (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    if (a.prof && lane == 0) t1 = wall_clock64();
    T s = T(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const T pr = S.v[k] * D.xv[k];
        s = s + ((S.c[k] & LANE_NONE) ? T(0) : pr);
    }
    s = seg_allreduce<L, T>(s);
    if (head && S.rid >= 0) lane_publish<T, EPI, MODE>(a, S.rid, s, D.bv, S.rd, D.xo);
    if (a.prof && lane == 0) {
        long long *o = a.prof + (size_t)g * 4;
        o[0] = D.t0; o[1] = t1; o[2] = wall_clock64();
        o[3] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) | (blockIdx.x << 4));
    }
}

constexpr int LANE_WPB = BLK / 64;            // waves per workgroup (they never meet)

template <typename T, int EPI, int L, int K, int MODE>
__global__ __launch_bounds__(BLK) void gs_lane_kernel(const LaneArgs<T> a)
{
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int idle = (int)((((unsigned)blockIdx.x * LANE_WPB + (unsigned)wib) * 16u) % (unsigned)a.nidle);
    LaneSet<T, K> P, Q;
    LaneDyn<T, K> D;
    if constexpr (MODE != 1) {
        const int W = (int)gridDim.x * LANE_WPB;
        int g = (int)blockIdx.x * LANE_WPB + wib;
        const int gend = a.ngroups;
        if (g >= gend) return;
        // a group's static operands are requested when the wave arrives at it (round 5; until then the NEXT group's were requested before the wave
        // started to wait for the current one: the in-order memory counter puts every poll behind that prefetch -- level 1 of the 256^3 hierarchy
        // 2.282 -> 2.249 ms in one session, profiles/r05_microbench_lane_prefetch_ab.txt)
        for (; g < gend; g += W) {
            lane_load<T, L, K>(a, g, P);
            lane_issue<T, EPI, K>(a, P, D, idle);
            lane_finish<T, EPI, L, K, MODE>(a, P, D, g, idle);
        }
    } else {
        __shared__ int sh_home;
        if (threadIdx.x == 0) {
            const unsigned me = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) + 1u;     // HW_REG_XCC_ID[3:0] + 1
            unsigned home = 0u;
            __hip_atomic_compare_exchange_strong(a.ticket + 1, &home, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_home = (home == 0u || home == me) ? 1 : 0;      // home holds the previous value
        }
        __syncthreads();
        if (!sh_home) return;
        // tickets: lane 0 draws, the wave reads lane 0's register; two tickets are always ahead of the running group
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int g = (int)__builtin_amdgcn_readfirstlane(tk);
        if (g >= a.ngroups) return;
        lane_load<T, L, K>(a, g, P);
        if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int g2 = (int)__builtin_amdgcn_readfirstlane(tk);
        while (true) {
            unsigned tk3 = 0;
            lane_issue<T, EPI, K>(a, P, D, idle);
            if (g2 < a.ngroups) {
                lane_load<T, L, K>(a, g2, Q);
                if (lane == 0) tk3 = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lane_finish<T, EPI, L, K, MODE>(a, P, D, g, idle);
            if (g2 >= a.ngroups) break;
            g = (int)__builtin_amdgcn_readfirstlane(tk3);
            unsigned tk4 = 0;
            lane_issue<T, EPI, K>(a, Q, D, idle);
            if (g < a.ngroups) {
                lane_load<T, L, K>(a, g, P);
                if (lane == 0) tk4 = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lane_finish<T, EPI, L, K, MODE>(a, Q, D, g2, idle);
            if (g >= a.ngroups) break;
            g2 = (int)__builtin_amdgcn_readfirstlane(tk4);
        }
    }
}


// =================================================================== the MERGED form (round 6; layout and algebra: pamg_lanem_plan.h)
// s consecutive dependency levels of the sweep are eliminated into ONE super-level at plan time: the sweep pays one hand-off per
// super-level instead of one per level.  One row per wave (64 lanes share it), K = 1..8 operand slots per lane chosen PER ROW (the
// record of a group carries the first 64-slot unit and K), three kinds of operands: NEW values of earlier super-levels (polled in xs),
// OLD values (the snapshot of x taken by lanem_prepare_kernel) and entries of b.  f64, Gauss-Seidel only (SOR's merged coefficients
// would depend on the relaxation parameter of the call).
struct alignas(32) LaneMRec { int rid_a; int rid_b; int gate; int unitK; int rda_lo; int rda_hi; int rdb_lo; int rdb_hi; };     // unitK = first unit * 16 + units

struct LaneMSched {
    int64_t ngroups = 0, n_units = 0, nrows = 0;
    int rpw = 1;
    int nsuper = 0, nlevels = 0, s_max = 0, max_len = 0;
    int closed_by_length = 0, closed_by_growth = 0;
    double max_growth = 0.0;
    int64_t n_early = 0, n_old = 0, n_b = 0, max_super_groups = 0;
    int *d_cols = nullptr;
    double *d_vals = nullptr;
    LaneMRec *d_rec = nullptr;
    long long *d_prof = nullptr;
    std::vector<int64_t> super_grp;   // [nsuper + 1] first row (group) of every super-level
    int last_grid = 0;
    int cap = 0;
    const void *cap_kernel = nullptr;
    size_t bytes = 0;
};

struct LaneMArgs {
    const int *cols;
    const double *vals;
    const LaneMRec *rec;
    const double *xold;    // snapshot of x from before the sweep
    double *y;             // the live x
    double *xs;            // hand-off buffer, sentinel-filled
    const double *b;
    unsigned *err;
    unsigned *ticket;
    int ngroups, nidle, use_gate;
    long long *prof;       // nullptr or [ngroups][4] time stamps (tune key 11; the unpipelined kernel, static form)
};

// x -> snapshot, sentinels -> hand-off buffer: the two passes every merged sweep starts with, in one launch
__global__ __launch_bounds__(BLK) void lanem_prepare_kernel(const double *__restrict__ x, double *__restrict__ xold, double *__restrict__ xs, int64_t n)
{
    unsigned long long *p = reinterpret_cast<unsigned long long *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        xold[i] = x[i];
        p[i] = Sentinel<double>::value;
    }
}

// ---- the kernel.  One row at a time per wave: the record of the NEXT row travels while this one waits; the first two 64-slot units of a row are
// held in registers (46 VGPRs: eight waves per SIMD), rows of three and more units (a few per cent) fetch the rest in the tail.
// What round 6 measured on level 1 of the 256^3 hierarchy (2.03 M rows; profiles/r06_microbench_lanem_*, r06_lanem_row_phase_stamps_level1.txt,
// r06_pmc_lane_probe_level1_merged_vs_unmerged.json): the forward sweep goes 2.27 -> 1.80 - 1.85 ms at s = 2 AND at s = 3 and stays there -- it is no
// longer bound by the hand-off: 95 - 98 % of the rows find every early operand in their FIRST poll round (stamps), a row costs its wave 2.2 - 2.9 us
// (slots from HBM 0.6 - 0.7 us, operands 0.6 - 2.0 us growing with the number of waves, tail 0.16 us), and the sweep delivers 1.1 G rows/s from 768
// workgroups on, more waves only slow each other down.  The counters: 45 - 52 L1 -> L2 requests per row (35 unmerged), 24 % of them from memory, average
// latency 390 - 430 cycles.  Built, parity-tested, measured and REMOVED again (git history of this file; numbers in profiles/): every unit of a row in
// registers (92 VGPRs, five waves per SIMD: same floor); a pipelined kernel with three row contexts in flight (slower: 2.0 - 2.3 ms -- every re-poll
// waits for the prefetches, the first poll round is stale); the same with a PUBLISHER wave per workgroup taking the stores out of the compute waves'
// in-order memory queue through an LDS mailbox (2.2 - 2.5 ms); ordinary loads for the static operands (+ 3 - 5 %); the next row's slots requested behind
// the current row's operands (1.86 - 1.99 ms).  An ablation of the first kernel (wrong results by construction): 1.21 ms without waiting for early
// operands, 0.94 without the publishing store as well, 0.73 for slots + early operands alone.
// RPW rows of one super-level per wave (64 / RPW lanes each), NREG units of 64 slots in registers (the rest of a long group is fetched in the tail)
template <int NREG>
struct MCtx {
    int c[NREG];
    double v[NREG], xv[NREG];
    double bv, xo, rd;         // per lane: its row's b, old value, 1 / a_ii
    int rid;                   // per lane: its row | NODIAG, -1 = dummy slot
    int gate, unit, K, g;      // uniform
};

__device__ __forceinline__ void m_rec(const int4 *rp, int g, int gend, int4 &q0, int4 &q1)
{
    const size_t gg = (size_t)(g < gend ? g : gend - 1);
    q0 = rp[2 * gg];
    q1 = rp[2 * gg + 1];
}

template <int RPW, int NREG>
__device__ __forceinline__ void m_slots(const LaneMArgs &a, MCtx<NREG> &C, const int4 &q0, const int4 &q1, int g, int lane)
{
    // record: {row a | NODIAG, row b (RPW = 2; else -1), gate, unit * 16 + K, 1 / a_ii of a (2 words), of b (2 words)}
    C.g = g;
    const int rid_a = __builtin_amdgcn_readfirstlane(q0.x), rid_b = __builtin_amdgcn_readfirstlane(q0.y);
    C.gate = a.use_gate ? __builtin_amdgcn_readfirstlane(q0.z) : -1;
    const int uk = __builtin_amdgcn_readfirstlane(q0.w);
    C.unit = uk >> 4;
    C.K = uk & 15;
    const double rd_a = __hiloint2double(__builtin_amdgcn_readfirstlane(q1.y), __builtin_amdgcn_readfirstlane(q1.x));
    const double rd_b = __hiloint2double(__builtin_amdgcn_readfirstlane(q1.w), __builtin_amdgcn_readfirstlane(q1.z));
    const bool second = RPW == 2 && lane >= 32;
    C.rid = second ? rid_b : rid_a;
    C.rd = second ? rd_b : rd_a;
    const size_t e0 = (size_t)C.unit * 64 + (size_t)lane;
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
        // a group of fewer units reads its last unit again (no load under a branch); the copies are masked in m_gather
        const size_t e = e0 + (size_t)64 * (size_t)(k < C.K ? k : C.K - 1);
        C.c[k] = a.cols[e];
        C.v[k] = a.vals[e];
    }
}

template <int NREG>
__device__ __forceinline__ void m_gather(const LaneMArgs &a, MCtx<NREG> &C, int idle)
{
#pragma unroll
    for (int k = 1; k < NREG; ++k)
        if (k >= C.K) C.c[k] = LANE_NONE;
    const int row = C.rid < 0 ? 0 : (C.rid & LANE_MASK);
    C.bv = a.b[row];
    C.xo = a.xold[row];                                        // used by rows without a diagonal only
    // every operand by an L1-bypassing load: early ones poll the hand-off buffer, static ones read the snapshot of x and b (ordinary loads for
    // the static operands were measured 3 - 5 % slower, profiles/r06_microbench_lanem_plain_loads_for_static_operands_not_kept.json)
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
        int col = C.c[k] & LANEM_MASK;
#ifdef PAMG_LANEM_FAKE_LOCALITY      /* experiment only (wrong results): every static operand next to the row -- what perfect locality of the gathers would buy */
        if (!(C.c[k] & LANE_EARLY)) col = (row & ~63) + (int)(threadIdx.x & 63);
#endif
        const double *p = (C.c[k] & LANE_NONE) ? a.xold + idle : ((C.c[k] & LANE_EARLY) ? a.xs + col : ((C.c[k] & LANEM_BSRC) ? a.b + col : a.xold + col));
        C.xv[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// wait for the group's operands and form x_i of its rows; returns the value a row's head lane publishes
template <int RPW, int NREG>
__device__ __forceinline__ double m_finish(const LaneMArgs &a, MCtx<NREG> &C, int idle, long long *t_ready = nullptr, unsigned *n_spins = nullptr)
{
    using T = double;
    const int lane = threadIdx.x & 63;
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < NREG; ++k)
        if ((C.c[k] & LANE_EARLY) && !(C.c[k] & LANE_NONE) && Sentinel<T>::bits(C.xv[k]) == Sentinel<T>::value) pend |= 1u << k;
    unsigned spins = 0;
    if (C.gate >= 0 && __builtin_amdgcn_ballot_w64(pend != 0)) {
        const T *gp = a.xs + C.gate;
        while (true) {
            const T gv = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (Sentinel<T>::bits(gv) != Sentinel<T>::value) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 && (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
        }
    }
    while (pend) {
        if (spins) __builtin_amdgcn_s_sleep(1);
        T t[NREG];
#pragma unroll
        for (int k = 0; k < NREG; ++k)
            t[k] = __hip_atomic_load(((pend >> k) & 1u) ? a.xs + (C.c[k] & LANEM_MASK) : a.xs + idle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < NREG; ++k)
            if ((pend >> k) & 1u) {
                C.xv[k] = t[k];
                if (Sentinel<T>::bits(t[k]) != Sentinel<T>::value) pend &= ~(1u << k);
            }
        if ((++spins & 1023u) == 0) {
            if (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    if (t_ready) { *t_ready = wall_clock64(); *n_spins = spins; }
    T s = T(0);
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
        const T pr = C.v[k] * C.xv[k];
        s = s + ((C.c[k] & LANE_NONE) ? T(0) : pr);
    }
    if (C.K > NREG) {
        // the units beyond the registers: fetched now, one after the other (groups this long are a few per cent)
        for (int k = NREG; k < C.K; ++k) {
            const size_t e = (size_t)(C.unit + k) * 64 + (size_t)lane;
            const int c = a.cols[e];
            const T v = a.vals[e];
            const int col = c & LANEM_MASK;
            const bool early = (c & LANE_EARLY) && !(c & LANE_NONE);
            const T *p = (c & LANE_NONE) ? a.xold + idle : ((c & LANE_EARLY) ? a.xs + col : ((c & LANEM_BSRC) ? a.b + col : a.xold + col));
            T x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned sp2 = 0;
            while (__builtin_amdgcn_ballot_w64(early && Sentinel<T>::bits(x) == Sentinel<T>::value)) {
                __builtin_amdgcn_s_sleep(1);
                const T t = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (early && Sentinel<T>::bits(x) == Sentinel<T>::value) x = t;
                if ((++sp2 & 1023u) == 0 && (sp2 > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            const T pr = v * x;
            s = s + ((c & LANE_NONE) ? T(0) : pr);
        }
    }
    s = seg_allreduce<64 / RPW, T>(s);
    const bool upd = C.rid >= 0 && !(C.rid & LANE_NODIAG);
    T val = (C.bv - s) * C.rd;
    if (!upd) val = C.xo;
    return val;
}

template <int RPW, int MODE, int NREG>
__device__ __forceinline__ void m_publish(const LaneMArgs &a, const MCtx<NREG> &C, double val)
{
    if (((threadIdx.x & 63) & (64 / RPW - 1)) == 0 && C.rid >= 0) {
        const int row = C.rid & LANE_MASK;
        if constexpr (MODE == 1) __hip_atomic_store(a.xs + row, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(a.xs + row, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(C.rid & LANE_NODIAG)) a.y[row] = val;
    }
}

// NREG = units of a group held in registers: 2 (one row per wave) / 4 (two rows per wave) on the large levels, where eight waves per SIMD matter;
// 8 on the small levels, whose sweeps are bound by their hand-offs -- there the sequential tail of a long row sits on the critical path (level 2 of the
// 256^3 hierarchy at s = 6: 56 % of the rows hold three and more units)
template <int MODE, int RPW, int NREG>
__global__ __launch_bounds__(BLK) void gs_lanem_kernel(const LaneMArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int idle = (int)((((unsigned)blockIdx.x * LANE_WPB + (unsigned)wib) * 16u) % (unsigned)a.nidle);
    const int4 *rp = reinterpret_cast<const int4 *>(a.rec);
    const int gend = a.ngroups;
    MCtx<NREG> X;
    if constexpr (MODE != 1) {
        const int W = (int)gridDim.x * LANE_WPB;
        int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * LANE_WPB + wib);
        if (g >= gend) return;
        int4 q0, q1;
        m_rec(rp, g, gend, q0, q1);
        for (; g < gend; g += W) {
            int4 n0, n1;
            m_rec(rp, g + W, gend, n0, n1);
            if (a.prof) {
                // diagnostics: where a group's time goes (forced waits between the phases: the stamps change the timing a little)
                const long long t0 = wall_clock64();
                m_slots<RPW, NREG>(a, X, q0, q1, g, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const long long t1 = wall_clock64();
                m_gather<NREG>(a, X, idle);
                long long t2 = 0;
                unsigned sp = 0;
                const double val = m_finish<RPW, NREG>(a, X, idle, &t2, &sp);
                m_publish<RPW, MODE, NREG>(a, X, val);
                if (lane == 0) {
                    long long *o = a.prof + (size_t)g * 4;
                    o[0] = t0 | ((long long)(sp > 4095u ? 4095u : sp) << 52);
                    o[1] = t1; o[2] = t2; o[3] = wall_clock64();
                }
            } else {
                m_slots<RPW, NREG>(a, X, q0, q1, g, lane);
                m_gather<NREG>(a, X, idle);
                const double val = m_finish<RPW, NREG>(a, X, idle);
                m_publish<RPW, MODE, NREG>(a, X, val);
            }
            q0 = n0; q1 = n1;
        }
    } else {
        __shared__ int sh_home;
        if (threadIdx.x == 0) {
            const unsigned me = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) + 1u;
            unsigned home = 0u;
            __hip_atomic_compare_exchange_strong(a.ticket + 1, &home, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_home = (home == 0u || home == me) ? 1 : 0;
        }
        __syncthreads();
        if (!sh_home) return;
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int g = (int)__builtin_amdgcn_readfirstlane(tk);
        if (g >= gend) return;
        if (lane == 0) tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int g2 = (int)__builtin_amdgcn_readfirstlane(tk);
        int4 q0, q1;
        m_rec(rp, g, gend, q0, q1);
        while (true) {
            int4 n0, n1;
            m_rec(rp, g2, gend, n0, n1);
            unsigned tk3 = 0;
            if (g2 < gend && lane == 0) tk3 = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m_slots<RPW, NREG>(a, X, q0, q1, g, lane);
            m_gather<NREG>(a, X, idle);
            const double val = m_finish<RPW, NREG>(a, X, idle);
            m_publish<RPW, MODE, NREG>(a, X, val);
            if (g2 >= gend) break;
            g = g2; q0 = n0; q1 = n1;
            g2 = (int)__builtin_amdgcn_readfirstlane(tk3);
        }
    }
}

// ------------------------------------------------------------------ host side
namespace {

template <typename U>
int lane_upload(U **dst, const void *src, size_t bytes, size_t *total)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 256;   // slack: nothing reads past the end, but keep allocations non-empty
    PAMG_HIP(hipMalloc((void **)dst, alloc));
    if (bytes && src) PAMG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));      // src = nullptr: allocation only (filled on the device)
    if (total) *total += alloc;
    return PAMG_OK;
}

template <typename T, int EPI, int L, int MODE>
const void *lane_kernel_k(int K)
{
    switch (K) {
        case 1: return (const void *)gs_lane_kernel<T, EPI, L, 1, MODE>;
        case 2: return (const void *)gs_lane_kernel<T, EPI, L, 2, MODE>;
        case 3: return (const void *)gs_lane_kernel<T, EPI, L, 3, MODE>;
        case 4: return (const void *)gs_lane_kernel<T, EPI, L, 4, MODE>;
    }
    return nullptr;
}

template <typename T, int EPI, int MODE>
const void *lane_kernel_l(int L, int K)
{
    switch (L) {
        case 4: return lane_kernel_k<T, EPI, 4, MODE>(K);
        case 8: return lane_kernel_k<T, EPI, 8, MODE>(K);
        case 16: return lane_kernel_k<T, EPI, 16, MODE>(K);
        case 32: return lane_kernel_k<T, EPI, 32, MODE>(K);
        case 64: return lane_kernel_k<T, EPI, 64, MODE>(K);
    }
    return nullptr;
}

// mode: 0 static across the chip, 1 one-XCD (tickets)
template <typename T>
const void *lane_kernel(int epi, int L, int K, int mode)
{
    // bsr_gauss_seidel with 1x1 blocks computes (b - sum) / a_ii as well: in this form the two are one kernel
    if (epi == EPI_SOR) return mode == 1 ? lane_kernel_l<T, EPI_SOR, 1>(L, K) : lane_kernel_l<T, EPI_SOR, 0>(L, K);
    return mode == 1 ? lane_kernel_l<T, EPI_GS, 1>(L, K) : lane_kernel_l<T, EPI_GS, 0>(L, K);
}

}  // namespace

void free_lane_part(LaneSched *t)
{
    if (!t) return;
    hipFree(t->d_cols); hipFree(t->d_rec); hipFree(t->d_vals); hipFree(t->d_prof);
    delete t;
}

// rows the lane form can hold (LANE_KMAX * 64 off-diagonal entries) -- the padding check is the planner's
bool lane_eligible(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->R == 1 && g->nlevels > 1 && g->d_xs != nullptr && A->max_row_len <= LANE_KMAX * 64 + 1 && !g->lane_unfit;
}

// one-XCD form: the vectors (x, hand-off buffer) and the polling stay inside one XCD's 4 MB L2
static bool lane_one_xcd(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->gran_xcd == 1 || (A->gran_xcd == 0 && A->nrows <= 131072 && g->nrows / std::max(1, g->nlevels) <= 1024);
}

// The layout filled on the device (the default): the host plan (pattern only, build_lane_plan with fill = false) gives the row of every
// (group, slot row) and the gates; this kernel writes cols / vals and the 16-byte slot-row records (row | NODIAG, gate, 1 / a_ii) from the
// resident CSR arrays -- no download of the values, no upload of the padded 12-byte slots (256^3 level 1: 1.5 GB, 1 s of host time per direction).
// One wave per group; the L lanes of a row all walk the row (broadcast loads), lane q keeps entries e = q, q + L, ...: the slot rule
// of the host's fill pass (entries in storage order without the diagonal, the last stored diagonal wins).
template <typename T>
__global__ __launch_bounds__(256) void lane_fill_kernel(int n, const int *__restrict__ Ap, const int *__restrict__ Aj, const T *__restrict__ Ax,
                                                        int row_start, int row_step, long long m, int L, int K, long long ngroups, const int *__restrict__ rid,
                                                        const int *__restrict__ gate, int *__restrict__ cols, T *__restrict__ vals, LaneRec *__restrict__ rec)
{
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    if (g >= ngroups) return;
    const int RPW = 64 / L, r = lane / L, q = lane - r * L;
    const int i = rid[g * RPW + r];
    int cnt = 0;
    T d = T(0);
    if (i >= 0) {
        const long long ti = ((long long)i - row_start) * row_step;
        int e = 0;
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i) { d = Ax[p]; continue; }
            if (e % L == q) {
                const size_t s = (size_t)((g * K + e / L) * 64 + lane);
                if (j < 0 || j >= n) { cols[s] = LANE_NONE; vals[s] = T(0); }
                else {
                    const long long tj = ((long long)j - row_start) * row_step;
                    const bool early = tj >= 0 && tj < m && tj < ti;
                    cols[s] = j | (early ? LANE_EARLY : 0);
                    vals[s] = Ax[p];
                }
            }
            ++e;
        }
        cnt = e;
    }
    // the slots behind the row's entries: padding (slot es = k * L + q of this lane)
    for (int k = 0; k < K; ++k)
        if (k * L + q >= cnt) {
            const size_t s = (size_t)((g * K + k) * 64 + lane);
            cols[s] = LANE_NONE; vals[s] = T(0);
        }
    if (q == 0) {
        const bool nodiag = !(d != T(0));
        const T rd = (i >= 0 && !nodiag) ? T(1) / d : T(0);
        LaneRec R;
        R.rid = (i >= 0 && nodiag) ? (i | LANE_NODIAG) : i;
        R.gate = gate[g];
        if constexpr (sizeof(T) == 8) { R.rd_lo = __double2loint((double)rd); R.rd_hi = __double2hiint((double)rd); }
        else { R.rd_lo = __float_as_int((float)rd); R.rd_hi = 0; }
        rec[g * RPW + r] = R;
    }
}

int build_lane_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->lane) return PAMG_OK;
    PhaseTimer pt_("build_lane_part", A->nnz);
    const int ts = (int)tsize(A->dtype);
    // PAMG_LANE_HOST_FILL=1: the whole layout on the host (what the CPU suite replays); default: filled on the device.  The device
    // fill derives visit indices from (row - row_start) * row_step, which is the host's numbering for unit steps only
    const char *hf = getenv("PAMG_LANE_HOST_FILL");
    const bool host_fill = (hf && *hf == '1') || !A->d_Ap || !A->d_Aj || !A->d_Ax || (g->row_step != 1 && g->row_step != -1);
    PlanVec<unsigned char> hAx;
    if (host_fill) {
        hAx.resize((size_t)A->nnz * ts);
        if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    }
    LanePlan P;
    // Lanes per row.  Operators small enough for the one-XCD form (32 CUs): the fewest lanes that hold a row (most rows per
    // wave).  Across the chip waves are plentiful and the sweep is bound by the hand-off latency per dependency level: ONE
    // ROW PER WAVE where the rows are long enough to fill it (a wave then waits for its own row's operands only, not for the
    // slowest of 4 or 16 rows) -- level 1 of the 256^3 hierarchy, 31 entries per row: 2.65 ms with 16 lanes per row, 2.49
    // with 32, 2.37 with 64 (profiles/r04_microbench_lane_width.json).
    int want_L = A->lane_L;
    if (!want_L && !lane_one_xcd(A, g)) {
        want_L = 4;
        while (want_L < 64 && want_L < A->max_row_len - 1) want_L *= 2;
        if (want_L < 32) want_L = 0;                           // short rows: several rows per wave (stencils)
    }
    if (build_lane_plan((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), host_fill ? hAx.data() : nullptr, ts, g->row_start, g->row_step, (int)g->nrows,
                        g->nlevels, g->h_vis, g->h_lvl, want_L, P, host_fill))
        return PAMG_E_ARG;
    LaneSched *t = new (std::nothrow) LaneSched();
    if (!t) return PAMG_E_ALLOC;
    t->L = P.L; t->K = P.K; t->RPW = P.RPW; t->ngroups = P.ngroups;
    t->n_early = P.n_early; t->n_old = P.n_old; t->n_slots = P.n_slots;
    t->max_level_groups = P.max_level_groups;
    const size_t nrec = (size_t)P.ngroups * P.RPW;
    int st = PAMG_OK;
    if (host_fill) {
        std::vector<LaneRec> rec(nrec);
        for (size_t q = 0; q < nrec; ++q) {
            rec[q].rid = P.rid[q];
            rec[q].gate = P.gate[q / (size_t)P.RPW];
            rec[q].rd_lo = rec[q].rd_hi = 0;
            std::memcpy(&rec[q].rd_lo, &P.rdiag[q * (size_t)ts], (size_t)ts);
        }
        st = lane_upload(&t->d_rec, rec.data(), nrec * sizeof(LaneRec), &t->bytes);
        if (!st) st = lane_upload(&t->d_cols, P.cols.data(), P.cols.size() * sizeof(int), &t->bytes);
        if (!st) st = lane_upload(&t->d_vals, P.vals.data(), P.vals.size(), &t->bytes);
    } else {
        int *d_rid = nullptr, *d_gate = nullptr;
        st = lane_upload(&d_rid, P.rid.data(), P.rid.size() * sizeof(int), nullptr);
        if (!st) st = lane_upload(&d_gate, P.gate.data(), P.gate.size() * sizeof(int), nullptr);
        if (!st) st = lane_upload(&t->d_rec, nullptr, nrec * sizeof(LaneRec), &t->bytes);
        if (!st) st = lane_upload(&t->d_cols, nullptr, (size_t)P.n_slots * sizeof(int), &t->bytes);
        if (!st) st = lane_upload(&t->d_vals, nullptr, (size_t)P.n_slots * ts, &t->bytes);
        if (!st) {
            const unsigned grid = (unsigned)((P.ngroups + 3) / 4);
            (void)hipGetLastError();                                  // a stale error of an earlier query must not be taken for this launch's
            if (ts == 8)
                hipLaunchKernelGGL((lane_fill_kernel<double>), dim3(grid), dim3(256), 0, 0, (int)A->nrows, A->d_Ap, A->d_Aj, (const double *)A->d_Ax, g->row_start, g->row_step,
                                   (long long)g->nrows, P.L, P.K, (long long)P.ngroups, d_rid, d_gate, t->d_cols, (double *)t->d_vals, t->d_rec);
            else
                hipLaunchKernelGGL((lane_fill_kernel<float>), dim3(grid), dim3(256), 0, 0, (int)A->nrows, A->d_Ap, A->d_Aj, (const float *)A->d_Ax, g->row_start, g->row_step,
                                   (long long)g->nrows, P.L, P.K, (long long)P.ngroups, d_rid, d_gate, t->d_cols, (float *)t->d_vals, t->d_rec);
            st = (int)hipGetLastError();
            if (!st) st = (int)hipDeviceSynchronize();
        }
        hipFree(d_rid); hipFree(d_gate);
    }
    if (st) { free_lane_part(t); return st; }
    g->lane = t;
    g->bytes += t->bytes;                                      // the caller books them on the operator
    return PAMG_OK;
}

size_t lane_part_bytes(const GsSchedule *g) { return (g && g->lane) ? g->lane->bytes : 0; }

// ---- merged form, host side
void free_lanem_part(LaneMSched *t)
{
    if (!t) return;
    hipFree(t->d_cols); hipFree(t->d_vals); hipFree(t->d_rec); hipFree(t->d_prof);
    delete t;
}

size_t lanem_part_bytes(const GsSchedule *g) { return (g && g->lanem) ? g->lanem->bytes : 0; }

// levels merged per super-level: tune key 33 (1 = unmerged form, >= 2 = that many), 0 = automatic -- rows long enough to fill a wave
// (the SA coarse levels: >= 12 entries per row on average), f64
int lanem_smax(const pamg_matrix_s *A, const GsSchedule *g)
{
    if (A->dtype != PAMG_F64 || A->R != 1 || g->nlevels < 8) return 1;
    if (A->lane_merge == 1) return 1;
    if (A->lane_merge >= 2) return A->lane_merge;
    const char *e = getenv("PAMG_LANE_MERGE");
    if (e && atoi(e) >= 1) return atoi(e);
    if (A->nnz < 12 * std::max<int64_t>(1, A->nrows)) return 1;
    // large levels: 3 (level 1 of the 256^3 hierarchy: 1.80 ms at 2, 1.76 at 3, 2.14 at 4 -- the sweep is bound by rows per second, longer rows cost);
    // small levels are bound by their hand-offs: 8 (every unit in registers there; level 2, 44.6 K rows: 0.484 ms at 4, 0.447 at 6, 0.440 at 8; level 3,
    // 463 rows: 0.052 / 0.042 / 0.034)
    return A->nrows > 131072 ? 3 : 8;
}

int build_lanem_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->lanem) return PAMG_OK;
    const int s_max = lanem_smax(A, g);
    if (s_max < 2) return PAMG_E_ARG;
    PhaseTimer pt_("build_lanem_part", A->nnz);
    PlanVec<double> hAx;
    hAx.resize((size_t)A->nnz + 1);
    if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost));
    LaneMPlan P;
    // two rows per wave (32 lanes each, rows of a super-level paired by length) on the large levels: a fifth fewer padded slots (1.14 instead of 1.44
    // units per row on level 1 of the 256^3 hierarchy) and 1.74 instead of 1.81 ms -- the sweep there is bound by what the memory system delivers per
    // row, not by waves or hand-offs (profiles/r06_microbench_lanem_rows_per_wave.json); small levels are bound by their hand-offs: a wave that waits
    // for ITS row only (tune key 35: 1 / 2, 0 = this rule)
    const int rpw = A->lanem_rpw ? A->lanem_rpw : (A->nrows > 131072 ? 2 : 1);
    if (build_lanem_plan((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), hAx.data(), g->row_start, g->row_step, (int)g->nrows, g->nlevels, g->h_vis, g->h_lvl,
                         s_max, 1e3, P, LANEM_KMAX * 64, rpw))
        return PAMG_E_ARG;
    hAx = PlanVec<double>();
    // nothing gained (every group closed at once: an operator the growth bound rejects): the unmerged form is the cheaper layout
    if (P.nsuper * 10 > P.nlevels * 9) return PAMG_E_ARG;
    LaneMSched *t = new (std::nothrow) LaneMSched();
    if (!t) return PAMG_E_ALLOC;
    t->ngroups = P.ngroups; t->n_units = P.n_units; t->nrows = P.nrows; t->rpw = P.rpw; t->nsuper = P.nsuper; t->nlevels = P.nlevels; t->s_max = s_max; t->max_len = P.max_len;
    t->closed_by_length = P.closed_by_length; t->closed_by_growth = P.closed_by_growth; t->max_growth = P.max_growth;
    t->n_early = P.n_early; t->n_old = P.n_old; t->n_b = P.n_b; t->max_super_groups = P.max_super_groups;
    t->super_grp = P.super_grp;
    std::vector<LaneMRec> rec((size_t)P.ngroups);
    lane_parallel(P.ngroups, [&](int64_t g0, int64_t g1) {
        for (int64_t q = g0; q < g1; ++q) {
            LaneMRec &R = rec[(size_t)q];
            const size_t q0 = (size_t)q * (size_t)P.rpw;
            const double zero = 0.0;
            R.rid_a = P.rid[q0]; R.rid_b = P.rpw == 2 ? P.rid[q0 + 1] : -1;
            R.gate = P.gate[(size_t)q];
            R.unitK = P.unit[(size_t)q] * 16 + (int)P.K[(size_t)q];
            std::memcpy(&R.rda_lo, &P.rdiag[q0], 8);
            std::memcpy(&R.rdb_lo, P.rpw == 2 ? &P.rdiag[q0 + 1] : &zero, 8);
        }
    });
    int st = lane_upload(&t->d_rec, rec.data(), rec.size() * sizeof(LaneMRec), &t->bytes);
    if (!st) st = lane_upload(&t->d_cols, P.cols.data(), P.cols.size() * sizeof(int), &t->bytes);
    if (!st) st = lane_upload(&t->d_vals, P.vals.data(), P.vals.size() * sizeof(double), &t->bytes);
    if (!st && !g->d_xold) {                                   // the snapshot of x (allocated here: sweeps may run inside a graph capture)
        const size_t xb = ((size_t)A->nrows + 8) * sizeof(double);
        st = (int)hipMalloc(&g->d_xold, xb);
        if (!st) t->bytes += xb;
    }
    if (st) { free_lanem_part(t); return st; }
    g->lanem = t;
    g->bytes += t->bytes;
    return PAMG_OK;
}

int lanem_launch(pamg_matrix_s *A, GsSchedule *g, void *x, const void *b, hipStream_t s)
{
    LaneMSched *t = g->lanem;
    if (!t || !g->d_xold) return PAMG_E_STATE;
    const int64_t n = A->nrows;
    LaneMArgs a;
    a.cols = t->d_cols; a.vals = t->d_vals; a.rec = t->d_rec;
    a.use_gate = (A->lane_flags & 1) ? 1 : 0;
    a.xold = (const double *)g->d_xold; a.y = (double *)x; a.xs = (double *)g->d_xs; a.b = (const double *)b;
    a.err = g->d_sync + 1; a.ticket = g->d_sync + 20;
    a.ngroups = (int)t->ngroups;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
    if (A->gs_prof && !t->d_prof) {
        PAMG_HIP(hipMalloc((void **)&t->d_prof, (size_t)t->ngroups * 4 * sizeof(long long)));
        PAMG_HIP(hipMemset(t->d_prof, 0, (size_t)t->ngroups * 4 * sizeof(long long)));
    }
    a.prof = A->gs_prof ? t->d_prof : nullptr;
    const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
    hipLaunchKernelGGL(lanem_prepare_kernel, dim3(fgrid), dim3(BLK), 0, s, (const double *)x, (double *)g->d_xold, (double *)g->d_xs, n);
    PAMG_HIP(hipGetLastError());
    // the ticket form inside one XCD only for tiny levels: one row per group means one ticket per ROW, and the ticket counter is one address whose
    // atomics serialise (11.4 ns each) -- level 2 of the 256^3 hierarchy (44.6 K rows): 0.58 ms inside one XCD, 0.48 across the chip
    const bool xcd = A->gran_xcd == 1 || (A->gran_xcd == 0 && lane_one_xcd(A, g) && A->nrows <= 8192);
    static const int nreg_env = [] { const char *e = getenv("PAMG_LANEM_NREG"); return e ? atoi(e) : 0; }();     // A/B: 2 / 8 units in registers on one-row-per-wave levels
    const bool all_regs = t->rpw == 1 && (nreg_env ? nreg_env == 8 : A->nrows <= 131072);
    const void *k = t->rpw == 2 ? (xcd ? (const void *)gs_lanem_kernel<1, 2, 4> : (const void *)gs_lanem_kernel<0, 2, 4>)
                  : all_regs    ? (xcd ? (const void *)gs_lanem_kernel<1, 1, 8> : (const void *)gs_lanem_kernel<0, 1, 8>)
                                : (xcd ? (const void *)gs_lanem_kernel<1, 1, 2> : (const void *)gs_lanem_kernel<0, 1, 2>);
    static thread_local int cus = 0;
    if (!cus) cus = device_cus_lane();
    if (!(t->cap > 0 && t->cap_kernel == k)) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLK, 0) != hipSuccess) nb = 2;
        t->cap = std::max(1, std::min(nb - 1, 8));
        t->cap_kernel = k;
    }
    const int cap = t->cap;
    // waves wanted: 4 x the rows of an average super-level (level 2 of the 256^3 hierarchy, 199 rows per super-level at s = 6: 0.487 ms with 115
    // workgroups, 0.459 with 192, 0.473 with 256, 0.52 with 384), capped below
    const int per_level = (int)((t->ngroups + t->nsuper - 1) / std::max(1, t->nsuper));
    const int64_t want_waves = std::max<int64_t>(128, ((int64_t)A->lanem_ahead10 * per_level + 9) / 10);
    const int cwpb = LANE_WPB;
    // three workgroups per CU at most: from 768 workgroups on the sweep delivers what it delivers, more waves only slow each other down
    // (level 1 of the 256^3 hierarchy, s = 3: 1.76 ms with 768 workgroups, 1.88 with 1 024, 2.17 with 1 536, 2.33 with 1 792)
    // (two rows per wave, s = 3: 1.74 ms with 512 workgroups, 1.89 with 768, 2.08 with 1 024: two per CU)
    int G = (int)std::min<int64_t>((want_waves + cwpb - 1) / cwpb, (int64_t)std::min(cap, t->rpw == 2 ? 2 : 3) * cus);
    if (A->lane_G > 0) G = std::min(A->lane_G, cap * cus);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, (t->ngroups + cwpb - 1) / cwpb));
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

// info[0..11] = super-levels, dependency levels, groups (rows), 64-slot units, early / old / b operands, longest merged row, levels merged at most,
//               groups closed early by length / by growth, workgroups of the last launch;  growth = largest accepted growth factor
int lanem_info(const GsSchedule *g, int64_t *info, double *growth)
{
    for (int i = 0; i < 12; ++i) info[i] = 0;
    if (growth) *growth = 0.0;
    if (!g || !g->lanem) return PAMG_OK;
    const LaneMSched *t = g->lanem;
    info[0] = t->nsuper; info[1] = t->nlevels; info[2] = t->nrows; info[3] = t->n_units; info[4] = t->n_early; info[5] = t->n_old; info[6] = t->n_b;
    info[7] = t->max_len; info[8] = t->s_max; info[9] = t->closed_by_length; info[10] = t->closed_by_growth; info[11] = t->last_grid;
    if (growth) *growth = t->max_growth;
    return PAMG_OK;
}


// workgroups of the static form that are certainly co-resident: (occupancy - 1, at most 8) per CU -- the occupancy query
// can over-report by one per CU (MI355X_MICROARCH.md).  Asked once per schedule and kernel, not per sweep.
static int lane_grid_cap(LaneSched *t, const void *k)
{
    if (t->cap > 0 && t->cap_kernel == k) return t->cap;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLK, 0) != hipSuccess) nb = 2;
    nb = std::max(1, std::min(nb - 1, 8));
    t->cap = nb; t->cap_kernel = k;
    return nb;
}

int device_cus_lane()
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 64;
    return p.multiProcessorCount;
}

template <typename T>
static int lane_launch_t(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    LaneSched *t = g->lane;
    const size_t ts = tsize(A->dtype);
    const int64_t n = A->nrows;
    LaneArgs<T> a;
    a.cols = t->d_cols; a.vals = (const T *)t->d_vals; a.rec = t->d_rec;
    a.use_gate = (A->lane_flags & 1) ? 1 : 0;
    a.x = (const T *)x; a.y = (T *)x; a.xs = (T *)g->d_xs; a.b = (const T *)b;
    a.err = g->d_sync + 1; a.ticket = g->d_sync + 20;
    a.ngroups = (int)t->ngroups;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
    a.omega = (T)omega;
    if (!g->symmetric) {
        // write-after-read hazards are not ordered by the waits: old values come from a snapshot
        if (!g->d_xold) return PAMG_E_STATE;
        PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
        a.x = (const T *)g->d_xold;
    }
    if (A->gs_prof && !t->d_prof) {
        PAMG_HIP(hipMalloc((void **)&t->d_prof, (size_t)t->ngroups * 4 * sizeof(long long)));
        PAMG_HIP(hipMemset(t->d_prof, 0, (size_t)t->ngroups * 4 * sizeof(long long)));
    }
    a.prof = A->gs_prof ? t->d_prof : nullptr;
    const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
    hipLaunchKernelGGL((lane_fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
    PAMG_HIP(hipGetLastError());
    const int per_level = (int)((t->ngroups + g->nlevels - 1) / g->nlevels);
    const bool xcd = lane_one_xcd(A, g);
    const void *k = lane_kernel<T>(epi, t->L, t->K, xcd ? 1 : 0);
    if (!k) return PAMG_E_ARG;
    static thread_local int cus = 0;
    if (!cus) cus = device_cus_lane();
    const int cap = lane_grid_cap(t, k);
    // waves wanted: ~4 dependency levels of look-ahead (a wave that runs ahead waits in its poll loop with its operands in
    // registers) -- more waves poll more and the polls load the memory system (measured on the 256^3 hierarchy,
    // profiles/r04_microbench_lane_first.json: level 1, 227 groups per dependency level: 2.67 ms with 512 waves, 2.72 with
    // 1 024, 3.28 with 2 048, 4.71 with 4 096)
    // with one row per wave 2.3 levels: 2.37 ms with 2 048 waves, 2.40 with 3 072, 2.57 with 6 144, 2.46 with 1 536)
    const int64_t want_waves = std::max<int64_t>(128, t->RPW == 1 ? ((int64_t)23 * per_level + 9) / 10 : (int64_t)4 * per_level);
    int G = (int)std::min<int64_t>((want_waves + LANE_WPB - 1) / LANE_WPB, (int64_t)cap * cus);
    if (A->lane_G > 0) G = std::min(A->lane_G, cap * cus);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, (t->ngroups + LANE_WPB - 1) / LANE_WPB));
    void *args[] = {(void *)&a};
    if (xcd) {
        // 8x the wanted grid is launched; the workgroups off the home XCD leave at once
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

int lane_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    if (g->lanem && epi != EPI_SOR && A->dtype == PAMG_F64) return lanem_launch(A, g, x, b, s);
    if (!g->lane) {
        // an SOR sweep on an operator that holds the merged layout only (a bare operator tuned by hand; a solver announces SOR before its
        // schedules are built, pamg_solver.hip: prebuild_schedules): the unmerged layout is built now -- allocations, so not inside a graph capture
        const size_t before = g->bytes;
        PAMG_TRY(build_lane_part(A, g));
        A->bytes += g->bytes - before;
    }
    if (A->dtype == PAMG_F64) return lane_launch_t<double>(A, g, epi, x, b, omega, s);
    return lane_launch_t<float>(A, g, epi, x, b, omega, s);
}

// info[0..7] = lanes per row, slots per lane, groups, entry slots, early entries, workgroups of the last launch, widest level (groups), bytes
int lane_info(const GsSchedule *g, int64_t *info)
{
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (!g || !g->lane) return PAMG_OK;
    const LaneSched *t = g->lane;
    info[0] = t->L; info[1] = t->K; info[2] = t->ngroups; info[3] = t->n_slots; info[4] = t->n_early; info[5] = t->last_grid;
    info[6] = t->max_level_groups; info[7] = (int64_t)t->bytes;
    return PAMG_OK;
}

// first row of every super-level of the merged plan (nsuper + 1 values)
int lanem_levels(const GsSchedule *g, int64_t *out, int64_t cap, int64_t *n)
{
    *n = 0;
    if (!g || !g->lanem) return PAMG_OK;
    *n = (int64_t)g->lanem->super_grp.size();
    if (out && cap >= *n) std::memcpy(out, g->lanem->super_grp.data(), (size_t)*n * sizeof(int64_t));
    return PAMG_OK;
}

int lane_profile(const GsSchedule *g, long long *out, int64_t cap, int64_t *n)
{
    *n = 0;
    if (g && g->lanem && g->lanem->d_prof) {
        // merged form: per ROW {arrival | spins << 52, slots arrived, all operands present, published}
        *n = g->lanem->ngroups;
        if (out && cap >= *n) PAMG_HIP(hipMemcpy(out, g->lanem->d_prof, (size_t)*n * 4 * sizeof(long long), hipMemcpyDeviceToHost));
        return PAMG_OK;
    }
    if (!g || !g->lane || !g->lane->d_prof) return PAMG_OK;
    *n = g->lane->ngroups;
    if (out && cap >= *n) PAMG_HIP(hipMemcpy(out, g->lane->d_prof, (size_t)*n * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    return PAMG_OK;
}

}  // namespace pamg
