// pamg_tile_kernels.h -- the TILED order-exact sweep (gfx950).  Plan: pamg_tile_plan.h.
//
// One persistent workgroup per tile: four compute waves + one store wave.
//   * The compute waves walk the steps of their tile.  A step = rows of ONE dependency level: phase 1,
//     one lane per scheduled entry pair forms the products a_ij * x_j (x_j from the LDS ring for new
//     values of this tile, from the global hand-off buffer xs for new values of other tiles -- polled,
//     the published datum IS the flag -- or from x for old values) and parks them in LDS; phase 2, one
//     lane per row sums its products strictly in storage order, applies the reference's update and
//     writes the new value into the LDS ring.  Nothing on this chain touches global memory except
//     operands that were requested one or two steps earlier:
//       step s+2: entry codes / values / row operands  (static: depend on nothing)
//       step s+1: x_j gathers and first polls, b_i     (addresses come from the codes of step s+1)
//       step s  : consumed
//     three operand sets rotate through registers (the loop is unrolled by three so a prefetch lands in
//     the registers it is consumed from).  Every load is unconditional with a selected address: the
//     compiler can then count its vmcnt exactly and a wait for step s never drains the prefetches.
//   * The store wave owns ALL global stores (x and the write-through publishes to xs): gfx950 retires
//     loads and stores through one counter, so a wave that had stored would stall its next operand
//     wait on the acknowledgement of a write-through store (~1-2 us).  The store wave reads the
//     finished values (and row ids) from the LDS ring one step behind the compute waves; it never
//     issues a load, the compute waves never issue a store.
// Deadlock freedom: every workgroup takes its steps in non-decreasing level order and a step only
// waits for rows of strictly lower levels, so the lexicographically smallest unfinished (level, tile)
// can always run PROVIDED all G workgroups are resident (the host sizes G by the occupancy query with
// a margin; spins are bounded and raise the error flag).
#pragma once
#include "pamg_kernels.h"

namespace pamg {

constexpr int TILE_THREADS = BLK + 64;        // 4 compute waves + the store wave
constexpr int MAXP_TILE = 4;                  // entry pairs per lane and step of the wide variant (2 * MAXP_TILE * BLK entries per step)

template <typename T>
struct TileArgs {
    const int4 *steps;        // [nsteps] {r0, r1, p0, p1}
    const int *tile_step;     // [G+1]
    const int *Ap, *Aj;       // scheduled row pointers / entry codes
    const T *Ax;
    const int *rid;           // original row | publish flag (bit 31)
    const T *diag;
    const T *x;               // OLD values (the live vector, or a snapshot for non-symmetric patterns)
    T *xs;                    // global hand-off buffer (sentinel-filled)
    T *y;                     // destination (the live vector)
    const T *b;
    unsigned *err;
    long long *prof;          // nullptr or [nsteps][4]
    T omega;
    int W;                    // ring slots (power of two)
    int nidle;
    int G;
};

template <typename T, int MAXP>
struct TileSet {
    int4 meta;
    int2 c[MAXP];
    typename Vec2<T>::type v[MAXP];
    T xv[2 * MAXP];
    int lo, hi, rid;
    T d, b, xo;
    int4 nmeta;               // descriptor of the step this set holds NEXT (fetched three steps ahead, into the registers
                              // it is consumed from: a value carried in other registers would be copied at the loop's
                              // back edge, and that copy waits for every load in flight)
};

// entry codes / values / row operands of the step described by S.nmeta; then the descriptor of step `next`
template <typename T, int MAXP>
__device__ __forceinline__ void tile_static(const TileArgs<T> &a, TileSet<T, MAXP> &S, int next)
{
    using T2 = typename Vec2<T>::type;
    const int tid = threadIdx.x;
    int4 meta;                                             // uniform: keep it in scalar registers
    meta.x = __builtin_amdgcn_readfirstlane(S.nmeta.x);
    meta.y = __builtin_amdgcn_readfirstlane(S.nmeta.y);
    meta.z = __builtin_amdgcn_readfirstlane(S.nmeta.z);
    meta.w = __builtin_amdgcn_readfirstlane(S.nmeta.w);
    S.meta = meta;
    const int p1 = meta.w, base = meta.z & ~1;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        const int qq = q < p1 ? q : base;                  // lanes past the end re-read the first pair (valid memory)
        S.c[k] = *reinterpret_cast<const int2 *>(a.Aj + qq);
        S.v[k] = *reinterpret_cast<const T2 *>(a.Ax + qq);
    }
    const int r = meta.x + tid;
    const int rr = r < meta.y ? r : meta.x;
    S.lo = a.Ap[rr];
    S.hi = a.Ap[rr + 1];
    S.rid = a.rid[rr];
    S.d = a.diag[rr];
    S.nmeta = a.steps[next];
}

// x_j gathers / first polls and the row-id dependent operands of the step held by S
template <typename T, int MAXP>
__device__ __forceinline__ void tile_gathers(const TileArgs<T> &a, TileSet<T, MAXP> &S)
{
    const int tid = threadIdx.x;
    const int p0 = S.meta.z, p1 = S.meta.w, base = p0 & ~1;
    const T *idle = a.x + (int)(((unsigned)blockIdx.x * 4u + (unsigned)(tid >> 6)) * 16u) % a.nidle;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int code = h ? S.c[k].y : S.c[k].x;
            const bool valid = (q + h) >= p0 && (q + h) < p1;
            const bool early = code < 0, dg = (code & DIAG_BIT) != 0;
            const T *src = idle;
            if (valid && !dg) src = early ? (const T *)a.xs + (code & COL_MASK) : a.x + (code & COL_MASK);
            S.xv[2 * k + h] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int row = S.rid & COL_MASK;
    S.b = a.b[row];
    S.xo = a.x[row];
}

// one step, compute waves: products -> LDS, barrier, in-order row sums -> LDS ring, barrier
template <typename T, int EPI, int MAXP>
__device__ __forceinline__ void tile_consume(const TileArgs<T> &a, const TileSet<T, MAXP> &S, T *prod, T *ring,
                                             int *ringrid, int2 *shmeta, int tile_r0, int parity)
{
    using T2 = typename Vec2<T>::type;
    const int tid = threadIdx.x;
    const int p0 = S.meta.z, p1 = S.meta.w, base = p0 & ~1;
    const int wmask = a.W - 1;
    T xv[2 * MAXP];
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * k + h;
            const int code = h ? S.c[k].y : S.c[k].x;
            const bool valid = (q + h) >= p0 && (q + h) < p1;
            const bool early = code < 0, dg = (code & DIAG_BIT) != 0;
            T val = S.xv[j];
            if (valid && early && dg) val = ring[code & wmask];                    // new value of this tile
            if (valid && early && !dg && Sentinel<T>::bits(val) == Sentinel<T>::value) pend |= 1u << j;
            xv[j] = val;
        }
    }
    if (pend) {
        // not published yet: poll (rare once the tile has settled behind its producers)
        const T *idle = a.x + (int)(((unsigned)blockIdx.x * 4u + (unsigned)(tid >> 6)) * 16u) % a.nidle;
        unsigned spins = 0;
        while (pend) {
            __builtin_amdgcn_s_sleep(1);
            T t[2 * MAXP];
#pragma unroll
            for (int j = 0; j < 2 * MAXP; ++j) {
                const int code = (j & 1) ? S.c[j >> 1].y : S.c[j >> 1].x;
                const T *src = ((pend >> j) & 1u) ? (const T *)a.xs + (code & COL_MASK) : idle;
                t[j] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < 2 * MAXP; ++j) {
                if (((pend >> j) & 1u) && Sentinel<T>::bits(t[j]) != Sentinel<T>::value) {
                    xv[j] = t[j];
                    pend &= ~(1u << j);
                }
            }
            if (++spins > (1u << 22)) {                    // ~seconds: producer not resident / bug
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        if (q < p1) {
            const int2 cc = S.c[k];
            T2 pr;
            pr.x = ((cc.x & DIAG_BIT) && cc.x >= 0) ? T(0) : S.v[k].x * xv[2 * k];
            pr.y = ((cc.y & DIAG_BIT) && cc.y >= 0) ? T(0) : S.v[k].y * xv[2 * k + 1];
            *reinterpret_cast<T2 *>(prod + (q - base)) = pr;
        }
    }
    if (tid == 0) shmeta[parity] = make_int2(S.meta.x, S.meta.y);    // the store wave's view of this step
    lds_barrier();
    const int r = S.meta.x + tid;
    if (r < S.meta.y) {
        T s = EpiTraits<EPI>::bsr_order ? S.b : T(0);
        row_accumulate<T, EPI>(s, prod, nullptr, S.lo - base, S.hi - base, 0);
        T v;
        if constexpr (EPI == EPI_GS) v = (S.b - s) / S.d;
        else if constexpr (EPI == EPI_GS_B) v = s / S.d;
        else v = a.omega * ((S.b - s) / S.d) + (T(1) - a.omega) * S.xo;
        if (!(S.d != T(0))) v = S.xo;                      // no / zero diagonal: the row keeps its value (relaxation.h:72)
        const int slot = (r - tile_r0) & wmask;
        ring[slot] = v;
        ringrid[slot] = S.rid;
    }
    lds_barrier();
}

// the store wave's half of a step: x (and, for rows with consumers in other tiles, xs) from the ring
template <typename T>
__device__ __forceinline__ void tile_store(const TileArgs<T> &a, const T *ring, const int *ringrid, const int2 m,
                                           int tile_r0)
{
    const int lane = threadIdx.x & 63;
    const int wmask = a.W - 1;
    for (int r = m.x + lane; r < m.y; r += 64) {
        const int slot = (r - tile_r0) & wmask;
        const T v = ring[slot];
        const int id = ringrid[slot];
        const int row = id & COL_MASK;
        a.y[row] = v;
        if (id < 0) __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, int EPI, int MAXP>
__global__ __launch_bounds__(TILE_THREADS) void gs_tile_kernel(const TileArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int CAPT = 2 * MAXP * BLK;
    T *prod = reinterpret_cast<T *>(smem_raw);
    T *ring = prod + (CAPT + 8);
    int *ringrid = reinterpret_cast<int *>(ring + a.W);
    int2 *shmeta = reinterpret_cast<int2 *>(ringrid + a.W);
    const int tile = (int)blockIdx.x;
    if (tile >= a.G) return;
    const int s0 = a.tile_step[tile], s1 = a.tile_step[tile + 1];
    if (s0 >= s1) return;
    const int tile_r0 = a.steps[s0].x;
    if (threadIdx.x >= BLK) {
        // ---- store wave: two barriers per step with the compute waves, then the stores of that step
        const bool t0 = a.prof && (threadIdx.x == BLK);
        for (int s = s0; s < s1; ++s) {
            lds_barrier();
            long long tb1 = 0;
            if (t0) tb1 = wall_clock64();
            lds_barrier();
            long long tb2 = 0;
            if (t0) tb2 = wall_clock64();
            const int2 m = shmeta[(s - s0) & 1];
            tile_store<T>(a, ring, ringrid, m, tile_r0);
            if (t0) {
                long long *o = a.prof + (size_t)s * 4;
                o[0] = tb1; o[1] = tb2; o[2] = (long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF); o[3] = tile;
            }
        }
        return;
    }
    // ---- compute waves
    TileSet<T, MAXP> A0, A1, A2;
    const int last = s1 - 1;
    A0.nmeta = a.steps[s0];
    A1.nmeta = a.steps[min(s0 + 1, last)];
    A2.nmeta = a.steps[min(s0 + 2, last)];
    tile_static<T, MAXP>(a, A0, min(s0 + 3, last));
    tile_static<T, MAXP>(a, A1, min(s0 + 4, last));
    tile_gathers<T, MAXP>(a, A0);
    int s = s0;
    while (true) {
        // A0: static + gathers in flight; A1: static in flight; A2.nmeta: descriptor of step s+2
        tile_gathers<T, MAXP>(a, A1);
        tile_static<T, MAXP>(a, A2, min(s + 5, last));
        tile_consume<T, EPI, MAXP>(a, A0, prod, ring, ringrid, shmeta, tile_r0, (s - s0) & 1);
        if (++s >= s1) break;
        tile_gathers<T, MAXP>(a, A2);
        tile_static<T, MAXP>(a, A0, min(s + 5, last));
        tile_consume<T, EPI, MAXP>(a, A1, prod, ring, ringrid, shmeta, tile_r0, (s - s0) & 1);
        if (++s >= s1) break;
        tile_gathers<T, MAXP>(a, A0);
        tile_static<T, MAXP>(a, A1, min(s + 5, last));
        tile_consume<T, EPI, MAXP>(a, A2, prod, ring, ringrid, shmeta, tile_r0, (s - s0) & 1);
        if (++s >= s1) break;
    }
}

}  // namespace pamg
