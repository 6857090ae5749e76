// pamg_kernels.h -- device kernels (gfx950, wave64).  No MFMA anywhere: the whole path is
// HBM-bound sparse streaming (0.13 flop/byte), so the design rules are coalesced streams,
// LDS staging and enough workgroups in flight to cover HBM latency.
//
// The work-horse is csr_stream_kernel ("LDS-streamed CSR"):
//   phase 1  the workgroup streams its contiguous slice of (Aj, Ax) with fully coalesced
//            loads -- one lane per stored entry, independent of row boundaries -- gathers
//            x[Aj] through L1/L2 and parks the products a_ij*x_j (and, for the smoothers,
//            the column ids) in LDS;
//   phase 2  one lane per row walks its products in LDS *in storage order* and applies the
//            epilogue (SpMV / residual / prolongation-add / Horner step / Jacobi / GS ...).
// Phase 2's strictly sequential per-row summation with separate multiply and add is what
// makes every result bit-identical to the reference's scalar loops; it costs nothing
// because LDS bandwidth is an order of magnitude above the HBM stream that bounds us.
#pragma once
#include "pamg_common.h"
#include "pamg_rowmask_map.h"

namespace pamg {

}  // namespace pamg

namespace pamg {
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
}  // namespace pamg
namespace pamg {

// sum over the workgroup, result valid in thread 0; sm = >= BLK/64 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double *sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < BLK / 64; ++i) r += sm[i];
    }
    return r;
}

template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global
// loads and stores (s_waitcnt vmcnt(0)): inside the granular sweeps that would make every range wait
// for the acknowledgement of its write-through publishing stores and for the prefetch it has just
// issued (measured: 2.6 us per range).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int EPI> struct EpiTraits {
    static constexpr bool need_cols = (EPI == EPI_JACOBI || EPI == EPI_JACOBI_B || EPI == EPI_JACOBI_IDX);   // row phase compares column ids
    static constexpr bool diag_flag = (EPI == EPI_GS || EPI == EPI_GS_B || EPI == EPI_SOR);   // schedule copies flag a_ii
    static constexpr bool perm = (EPI == EPI_GS || EPI == EPI_GS_B || EPI == EPI_SOR || EPI == EPI_JACOBI_IDX);   // stored row r is row rid[r]
    static constexpr bool bsr_order = (EPI == EPI_JACOBI_B || EPI == EPI_GS_B);
};

// x accesses of the persistent sweeps: other workgroups rewrite x inside the same launch, so
// loads must bypass this CU's L1 and stores must write through (agent-scope relaxed atomics
// lower to global_load/store ... sc1; MI355X_MICROARCH.md, "valid forms").
template <int COH, typename T>
__device__ __forceinline__ T ldx(const T *p)
{
    if constexpr (COH == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// Level schedules mark "early" entries -- columns whose row is visited EARLIER in the same
// sweep, i.e. whose NEW value must be used -- with the sign bit of the stored column id.
constexpr int EARLY_BIT = (int)0x80000000u;
// Level schedules also flag the DIAGONAL entries (bit 30): their product is staged as +0, which
// leaves the in-order sum bit-identical to skipping them (s + (+0) == s and s - (+0) == s for
// every s the running sums can hold: one that starts at +0 is never -0), so the row phase of the
// sweeps needs no column ids in LDS and no compare per entry.
constexpr int DIAG_BIT = 0x40000000;
constexpr int COL_MASK = 0x3FFFFFFF;

// Sentinel bit patterns of the hand-off buffer xs (a quiet NaN with a payload no arithmetic
// produces): xs[j] == sentinel  <=>  row j has not published its new value in this sweep yet.
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

// wait for row j's published value: the 8-byte (4-byte) datum IS the flag -- one write-through
// store by the producer, relaxed agent-scope polling here (MI355X_MICROARCH.md, hand-off "R2")
template <typename T>
__device__ __forceinline__ T spin_value(const T *xs, int j, unsigned *err)
{
    T v = __hip_atomic_load(xs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (Sentinel<T>::bits(v) == Sentinel<T>::value) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(xs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > (1u << 22)) {                        // ~seconds: producer not resident / bug
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    return v;
}

#ifdef PAMG_FAKE_RUNTABLE
__shared__ double pamg_fake_xl[1024];
#endif
// gather of one x value in the three flavours of the kernel family:
//   plain (COH = 0): ordinary cached load;  COH = 1: L1-bypassing load (block sweeps with a barrier per level);
//   COH = 2 (granular sweep): early entries spin on the hand-off buffer, the others read x.
template <int COH, typename T>
__device__ __forceinline__ T gather_x(const StreamArgs<T> &a, int c)
{
    if constexpr (COH == 2 || COH == 3) {
        if (c & EARLY_BIT) return spin_value<T>(a.xs, c & COL_MASK, a.err);
        return a.x[c & COL_MASK];
    } else if constexpr (COH == 1) {
        return __hip_atomic_load(a.x + (c & COL_MASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        return a.x[c & COL_MASK];
    }
}

// ---- phase 1 with 16-bit column codes AND 8-bit value codes (operators with at most 256 distinct values: stencils).
// 3 instead of 12 bytes of operator stream per entry: the kernel is no longer bound by the stream but by the gather --
// so the lanes of a wave take CONSECUTIVE entries (one 2-byte and one 1-byte load per entry, each a single cache line per
// wave) and a gather instruction covers 64 consecutive entries = ~9 rows of a stencil = one or two lines of x per band,
// instead of the 3-5 lines per band of the several-entries-per-lane layouts.  NCH independent chains per lane in flight.
// The value is looked up in the LDS copy of the dictionary: the very same bits the 8-byte stream would have delivered.
template <typename T, bool NEEDC, int COH, int NCH>
__device__ __forceinline__ void stage_val8(const StreamArgs<T> &a, int p0, int p1, int base, T *prod, int *cols, const T *vd)
{
    const int4 wb = a.wb;
    for (int p = p0 + (int)threadIdx.x; p < p1; p += NCH * BLK) {
        int pk[NCH];
        unsigned c[NCH], v[NCH];
        int cc[NCH];
        T xv[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) pk[j] = (p + j * BLK < p1) ? p + j * BLK : p;      // out of range: repeat a valid entry
#pragma unroll
        for (int j = 0; j < NCH; ++j) { c[j] = a.Aj16[pk[j]]; v[j] = a.Ax8[pk[j]]; }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const unsigned w = c[j] >> 14;
            cc[j] = (w == 0 ? wb.x : w == 1 ? wb.y : w == 2 ? wb.z : wb.w) + (int)(c[j] & 0x3FFFu);
            xv[j] = gather_x<COH>(a, cc[j]);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (j == 0 || p + j * BLK < p1) {
                prod[pk[j] - base] = vd[v[j]] * xv[j];
                if constexpr (NEEDC) cols[pk[j] - base] = cc[j];
            }
        }
    }
}

// two consecutive entries per lane, 16-bit column codes (window << 14 | offset), values as stored; NT = the operator stream past the
// caches' retention, so that what the L1 keeps is x (1 - 8 % on the SA-level operators, profiles/r05_microbench_sa_ops_nontemporal.json;
// the autotune decides per operator)
template <typename T, bool NEEDC, int COH, bool NT>
__device__ __forceinline__ void stage_pairs16(const StreamArgs<T> &a, int p0, int p1, int base, T *prod, int *cols, const int4 wb)
{
    using T2 = typename Vec2<T>::type;
    typedef T T2n __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
#pragma unroll 2
    for (int q = base + 2 * tid; q < p1; q += 2 * BLK) {
        unsigned pr2;
        T2 vv;
        if constexpr (NT) {
            pr2 = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(a.Aj16 + q));
            const T2n v2 = __builtin_nontemporal_load(reinterpret_cast<const T2n *>(a.Ax + q));
            vv.x = v2.x; vv.y = v2.y;
        } else {
            pr2 = *reinterpret_cast<const unsigned *>(a.Aj16 + q);
            vv = *reinterpret_cast<const T2 *>(a.Ax + q);
        }
        const unsigned c0 = pr2 & 0xFFFFu, c1 = pr2 >> 16;
        const unsigned w0 = c0 >> 14, w1 = c1 >> 14;
        int2 cc;
        cc.x = (w0 == 0 ? wb.x : w0 == 1 ? wb.y : w0 == 2 ? wb.z : wb.w) + (int)(c0 & 0x3FFFu);
        cc.y = (w1 == 0 ? wb.x : w1 == 1 ? wb.y : w1 == 2 ? wb.z : wb.w) + (int)(c1 & 0x3FFFu);
        const bool ok0 = q >= p0, ok1 = q + 1 < p1;
        T x0, x1;
        if (a.flags & 4) {                        // ablation: operator stream only, no gather
#ifdef PAMG_FAKE_RUNTABLE            /* experiment only (wrong results): the operands from an LDS stage of x, as a run-table kernel would read them */
            x0 = (T)pamg_fake_xl[cc.x & 1023]; x1 = (T)pamg_fake_xl[cc.y & 1023];
#else
            x0 = T(cc.x); x1 = T(cc.y);
#endif
        } else {
            x0 = ok0 ? gather_x<COH>(a, cc.x) : T(0);
            x1 = ok1 ? gather_x<COH>(a, cc.y) : T(0);
        }
        T2 pr;
        pr.x = vv.x * x0;
        pr.y = vv.y * x1;
        *reinterpret_cast<T2 *>(prod + (q - base)) = pr;
        if constexpr (NEEDC) *reinterpret_cast<int2 *>(cols + (q - base)) = cc;
    }
}

// the same with 32-bit columns (and the schedules' flag bits)
template <typename T, bool NEEDC, int COH, bool DIAGF, bool NT>
__device__ __forceinline__ void stage_pairs32(const StreamArgs<T> &a, int p0, int p1, int base, T *prod, int *cols)
{
    using T2 = typename Vec2<T>::type;
    typedef int int2n __attribute__((ext_vector_type(2)));
    typedef T T2n __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
#pragma unroll 2
    for (int q = base + 2 * tid; q < p1; q += 2 * BLK) {
        int2 cc;
        T2 vv;
        if constexpr (NT) {
            const int2n c2 = __builtin_nontemporal_load(reinterpret_cast<const int2n *>(a.Aj + q));
            const T2n v2 = __builtin_nontemporal_load(reinterpret_cast<const T2n *>(a.Ax + q));
            cc.x = c2.x; cc.y = c2.y; vv.x = v2.x; vv.y = v2.y;
        } else {
            cc = *reinterpret_cast<const int2 *>(a.Aj + q);
            vv = *reinterpret_cast<const T2 *>(a.Ax + q);
        }
        const bool ok0 = q >= p0, ok1 = q + 1 < p1;
        T x0, x1;
        if (a.flags & 4) {                            // ablation: operator stream only, no gather
#ifdef PAMG_FAKE_RUNTABLE            /* experiment only (wrong results): the operands from an LDS stage of x, as a run-table kernel would read them */
            x0 = (T)pamg_fake_xl[cc.x & 1023]; x1 = (T)pamg_fake_xl[cc.y & 1023];
#else
            x0 = T(cc.x); x1 = T(cc.y);
#endif
        } else {
            x0 = ok0 ? gather_x<COH>(a, cc.x) : T(0);
            x1 = ok1 ? gather_x<COH>(a, cc.y) : T(0);
        }
        T2 pr;
        pr.x = (DIAGF && (cc.x & DIAG_BIT)) ? T(0) : vv.x * x0;
        pr.y = (DIAGF && (cc.y & DIAG_BIT)) ? T(0) : vv.y * x1;
        *reinterpret_cast<T2 *>(prod + (q - base)) = pr;
        if constexpr (NEEDC) {
            cc.x &= COL_MASK;
            cc.y &= COL_MASK;
            *reinterpret_cast<int2 *>(cols + (q - base)) = cc;
        }
    }
}

// ---- phase 1: stage products (and column ids) of entries [p0,p1) into LDS slots [p-base]
template <typename T, bool NEEDC, int NPL, int COH = 0, bool DIAGF = false>
__device__ __forceinline__ void stage_products(const StreamArgs<T> &a, int p0, int p1, int base,
                                               T *prod, int *cols, const T *vd = nullptr)
{
    const int tid = threadIdx.x;
    if constexpr (NPL == 1) {
        int p = p0 + tid;
        for (; p + 3 * BLK < p1; p += 4 * BLK) {      // 4 independent load chains in flight
            const int c0 = a.Aj[p], c1 = a.Aj[p + BLK], c2 = a.Aj[p + 2 * BLK], c3 = a.Aj[p + 3 * BLK];
            const T v0 = a.Ax[p], v1 = a.Ax[p + BLK], v2 = a.Ax[p + 2 * BLK], v3 = a.Ax[p + 3 * BLK];
            const T x0 = gather_x<COH>(a, c0), x1 = gather_x<COH>(a, c1), x2 = gather_x<COH>(a, c2),
                    x3 = gather_x<COH>(a, c3);
            prod[p - base] = (DIAGF && (c0 & DIAG_BIT)) ? T(0) : v0 * x0;
            prod[p - base + BLK] = (DIAGF && (c1 & DIAG_BIT)) ? T(0) : v1 * x1;
            prod[p - base + 2 * BLK] = (DIAGF && (c2 & DIAG_BIT)) ? T(0) : v2 * x2;
            prod[p - base + 3 * BLK] = (DIAGF && (c3 & DIAG_BIT)) ? T(0) : v3 * x3;
            if constexpr (NEEDC) {
                cols[p - base] = c0 & COL_MASK;
                cols[p - base + BLK] = c1 & COL_MASK;
                cols[p - base + 2 * BLK] = c2 & COL_MASK;
                cols[p - base + 3 * BLK] = c3 & COL_MASK;
            }
        }
        for (; p < p1; p += BLK) {
            const int c0 = a.Aj[p];
            const T v0 = a.Ax[p];
            prod[p - base] = (DIAGF && (c0 & DIAG_BIT)) ? T(0) : v0 * gather_x<COH>(a, c0);
            if constexpr (NEEDC) cols[p - base] = c0 & COL_MASK;
        }
    } else if constexpr (NPL == 4) {
        // four consecutive entries per lane: one 16-byte index load, two 16-byte value loads.
        // base is a multiple of 4, arrays are padded, out-of-range slots are masked.
        using T2 = typename Vec2<T>::type;
        for (int q = base + 4 * tid; q < p1; q += 4 * BLK) {
            const int4 cc = *reinterpret_cast<const int4 *>(a.Aj + q);
            const T2 va = *reinterpret_cast<const T2 *>(a.Ax + q);
            const T2 vb = *reinterpret_cast<const T2 *>(a.Ax + q + 2);
            const int c[4] = {cc.x, cc.y, cc.z, cc.w};
            const T v[4] = {va.x, va.y, vb.x, vb.y};
            T pr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (q + j >= p0) && (q + j < p1);
                T xv;
                if (a.flags & 4) xv = T(c[j]);
                else xv = ok ? gather_x<COH>(a, c[j]) : T(0);
                pr[j] = (DIAGF && (c[j] & DIAG_BIT)) ? T(0) : v[j] * xv;
            }
            T2 o0, o1;
            o0.x = pr[0]; o0.y = pr[1]; o1.x = pr[2]; o1.y = pr[3];
            *reinterpret_cast<T2 *>(prod + (q - base)) = o0;
            *reinterpret_cast<T2 *>(prod + (q - base) + 2) = o1;
            if constexpr (NEEDC) {
                int4 cm;
                cm.x = c[0] & COL_MASK; cm.y = c[1] & COL_MASK; cm.z = c[2] & COL_MASK; cm.w = c[3] & COL_MASK;
                *reinterpret_cast<int4 *>(cols + (q - base)) = cm;
            }
        }
    } else {
        // two consecutive entries per lane: 8-byte index loads, 16-byte value loads, 16-byte
        // LDS stores.  base is even, so every pair is naturally aligned; the operator's
        // arrays are padded so the pair straddling p1 stays inside the allocation.
        using T2 = typename Vec2<T>::type;
        const bool nt = (a.flags & 1) != 0;               // stream the operator past the caches
        if (a.Aj16) {
            // 16-bit column stream: every column of this row range lies in one of (up to) four windows of 16 K columns;
            // an entry stores window << 14 | offset.  Two bytes less per entry on the operator stream, same arithmetic.
            const int4 wb = a.wb;
            if (vd) {
                if (a.flags & 16) { stage_val8<T, NEEDC, COH, 6>(a, p0, p1, base, prod, cols, vd); return; }
                // four consecutive entries per lane and step (8 + 4 bytes of operator stream for them), TWO steps in flight:
                // with 3 bytes per entry the kernel is bound by its dependent round trips (codes -> gather), so the codes of
                // both steps are requested before the first gather.  base is a multiple of 4 here.
                for (int q = base + 4 * tid; q < p1; q += 8 * BLK) {
                    const int q2 = q + 4 * BLK;
                    const bool two = q2 < p1;
                    const int qb = two ? q2 : q;
                    const uint2 cwa = *reinterpret_cast<const uint2 *>(a.Aj16 + q);
                    const unsigned vca = *reinterpret_cast<const unsigned *>(a.Ax8 + q);
                    const uint2 cwb = *reinterpret_cast<const uint2 *>(a.Aj16 + qb);
                    const unsigned vcb = *reinterpret_cast<const unsigned *>(a.Ax8 + qb);
                    const unsigned c[8] = {cwa.x & 0xFFFFu, cwa.x >> 16, cwa.y & 0xFFFFu, cwa.y >> 16,
                                           cwb.x & 0xFFFFu, cwb.x >> 16, cwb.y & 0xFFFFu, cwb.y >> 16};
                    int cc[8];
                    T xv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned w = c[j] >> 14;
                        cc[j] = (w == 0 ? wb.x : w == 1 ? wb.y : w == 2 ? wb.z : wb.w) + (int)(c[j] & 0x3FFFu);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // unconditional (no branch around the load): an entry outside the range reads the first column of
                        // the window; its product lands in a slot nobody sums (or is not stored at all)
                        const int e = (j < 4 ? q : qb) + (j & 3);
                        const bool ok = (e >= p0) && (e < p1);
                        xv[j] = gather_x<COH>(a, ok ? cc[j] : wb.x);
                    }
                    T2 o0, o1;
                    o0.x = vd[vca & 0xFFu] * xv[0]; o0.y = vd[(vca >> 8) & 0xFFu] * xv[1];
                    o1.x = vd[(vca >> 16) & 0xFFu] * xv[2]; o1.y = vd[vca >> 24] * xv[3];
                    *reinterpret_cast<T2 *>(prod + (q - base)) = o0;
                    *reinterpret_cast<T2 *>(prod + (q - base) + 2) = o1;
                    if constexpr (NEEDC) {
                        int4 cm;
                        cm.x = cc[0]; cm.y = cc[1]; cm.z = cc[2]; cm.w = cc[3];
                        *reinterpret_cast<int4 *>(cols + (q - base)) = cm;
                    }
                    if (two) {
                        o0.x = vd[vcb & 0xFFu] * xv[4]; o0.y = vd[(vcb >> 8) & 0xFFu] * xv[5];
                        o1.x = vd[(vcb >> 16) & 0xFFu] * xv[6]; o1.y = vd[vcb >> 24] * xv[7];
                        *reinterpret_cast<T2 *>(prod + (q2 - base)) = o0;
                        *reinterpret_cast<T2 *>(prod + (q2 - base) + 2) = o1;
                        if constexpr (NEEDC) {
                            int4 cm;
                            cm.x = cc[4]; cm.y = cc[5]; cm.z = cc[6]; cm.w = cc[7];
                            *reinterpret_cast<int4 *>(cols + (q2 - base)) = cm;
                        }
                    }
                }
                return;
            }
            // the nontemporal choice is made OUTSIDE the loop (round 6): with the run-time test inside the unrolled body (round 5) the loads sat
            // under a uniform branch and the fine-level residual went 0.332 -> 0.361 ms -- loads under a branch make the compiler wait
            // conservatively ("three things the compiler taught us" (a))
            if (nt) stage_pairs16<T, NEEDC, COH, true>(a, p0, p1, base, prod, cols, wb);
            else stage_pairs16<T, NEEDC, COH, false>(a, p0, p1, base, prod, cols, wb);
            return;
        }
        if (nt) stage_pairs32<T, NEEDC, COH, DIAGF, true>(a, p0, p1, base, prod, cols);
        else stage_pairs32<T, NEEDC, COH, DIAGF, false>(a, p0, p1, base, prod, cols);
    }
}

// ---- phase 2 pieces
// Per-row operands that do not depend on phase 1.  They are loaded BEFORE the products are
// staged so that their memory latency overlaps the streaming phase instead of extending the
// dependent chain after the barrier (this matters for the latency-bound per-level launches
// of the Gauss-Seidel family, where one level is a handful of workgroups).
template <typename T>
struct RowPre {
    int lo, hi, row, pos;
    T b, y, xo, d;
};

template <typename T, int EPI, int COH = 0>
__device__ __forceinline__ RowPre<T> row_prefetch(const StreamArgs<T> &a, int r)
{
    RowPre<T> q;
    q.lo = a.Ap[r];
    q.hi = a.Ap[r + 1];
    q.row = EpiTraits<EPI>::perm ? a.rid[r] : r;
    q.pos = r;
    q.b = q.y = q.xo = q.d = T(0);
    if constexpr (EPI >= EPI_JACOBI) q.d = a.diag[r];           // precomputed diagonal of stored row r
    if constexpr (EPI == EPI_RESID || EPI == EPI_AXPBY || EPI == EPI_ACC_AXPBY || EPI == EPI_SUMSQ ||
                  EPI >= EPI_JACOBI)
        q.b = a.b[q.row];
    if constexpr (EPI == EPI_ACC || EPI == EPI_ACC_AXPBY || EPI == EPI_ACCSEQ) q.y = a.y[q.row];
    if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_B || EPI == EPI_SOR || EPI == EPI_JACOBI_IDX) q.xo = ldx<COH>(a.x + q.row);
    return q;
}

// Sequential, storage-order accumulation of one row's products.  The adds stay strictly in
// order (the result is the reference's running sum, bit for bit); everything around them is
// arranged so that the dependent add chain is all that is left on the critical path.  The row
// phase is VALU-issue bound (one lane per row, every instruction costs a full wave slot), so:
// eight consecutive LDS slots are read per batch with constant offsets (reads may run up to
// eight slots past the row's end -- inside the padded window, never used), the next batch is
// in flight while the current one is summed, full batches are summed WITHOUT per-entry
// predicates and the 0..7 leftover entries as a 4 + 2 + 1 decomposition.
template <typename T, int EPI>
__device__ __forceinline__ void row_accumulate(T &s, const T *prod, const int *cols, int lo, int hi, int row)
{
    constexpr int U = 8;
    if (lo >= hi) return;
    if constexpr (EpiTraits<EPI>::need_cols) {
        // Jacobi family on the operator's own arrays: the diagonal is recognised by column id
        T p[U];
        int c[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { p[j] = prod[lo + j]; c[j] = cols[lo + j]; }
        for (int k = lo; k < hi; k += U) {
            T pn[U];
            int cn[U];
            const bool more = k + U < hi;
            if (more) {
#pragma unroll
                for (int j = 0; j < U; ++j) { pn[j] = prod[k + U + j]; cn[j] = cols[k + U + j]; }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (k + j < hi && c[j] != row) {           // diagonal never enters the sum
                    if constexpr (EpiTraits<EPI>::bsr_order) s -= p[j];
                    else s += p[j];
                }
            }
            if (more) {
#pragma unroll
                for (int j = 0; j < U; ++j) { p[j] = pn[j]; c[j] = cn[j]; }
            }
        }
    } else {
        const T *q = prod + lo;
        int n = hi - lo;
        T p[U];
#pragma unroll
        for (int j = 0; j < U; ++j) p[j] = q[j];
        while (n >= U) {
            T pn[U];
#pragma unroll
            for (int j = 0; j < U; ++j) pn[j] = q[U + j];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if constexpr (EpiTraits<EPI>::bsr_order) s -= p[j];
                else s += p[j];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) p[j] = pn[j];
            q += U;
            n -= U;
        }
        if (n & 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (EpiTraits<EPI>::bsr_order) s -= p[j];
                else s += p[j];
            }
            p[0] = p[4]; p[1] = p[5]; p[2] = p[6];
        }
        if (n & 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (EpiTraits<EPI>::bsr_order) s -= p[j];
                else s += p[j];
            }
            p[0] = p[2];
        }
        if (n & 1) {
            if constexpr (EpiTraits<EPI>::bsr_order) s -= p[0];
            else s += p[0];
        }
    }
}

template <typename T, int EPI>
__device__ __forceinline__ T row_init(const RowPre<T> &q)
{
    if constexpr (EpiTraits<EPI>::bsr_order) return q.b;
    else if constexpr (EPI == EPI_ACCSEQ) return q.y;
    else return T(0);
}

template <typename T, int EPI, int COH = 0>
__device__ __forceinline__ void row_finish(const StreamArgs<T> &a, const RowPre<T> &q, T s, double &sq)
{
    const T one = T(1);
    const int row = q.row;
    if constexpr (EPI == EPI_SET || EPI == EPI_ACCSEQ) {
        a.y[row] = s;
        // EPI_SET can clear a second vector of the same length on the way (the cycle's b_c = R r also sets x_c = 0:
        // one launch less per level than a separate fill); the otherwise unused `partial` slot carries its address
        if constexpr (EPI == EPI_SET) { if (a.partial) reinterpret_cast<T *>(a.partial)[row] = T(0); }
    } else if constexpr (EPI == EPI_ACC) {
        a.y[row] = q.y + s;
    } else if constexpr (EPI == EPI_RESID) {
        const T t = q.b - s;
        a.y[row] = t;
        // the polynomial smoother's first Horner step h = c0 * r rides along (relaxation.py:652-653; the same single multiplication as the vector
        // kernel it replaces: bit-identical) -- one pass over two vectors less per smoother application; the unused `partial` slot carries h
        if (a.partial) reinterpret_cast<T *>(a.partial)[row] = a.c * t;
    } else if constexpr (EPI == EPI_AXPBY) {
        const T t = a.c * q.b;
        a.y[row] = t + s;
    } else if constexpr (EPI == EPI_ACC_AXPBY) {
        const T t = a.c * q.b;
        const T h = t + s;
        a.y[row] = q.y + h;
    } else if constexpr (EPI == EPI_SUMSQ) {
        const T t = q.b - s;
        sq += (double)t * (double)t;
    } else if constexpr (EPI == EPI_JACOBI) {
        a.y[row] = (q.d != T(0)) ? (one - a.omega) * q.xo + a.omega * ((q.b - s) / q.d) : q.xo;
    } else if constexpr (EPI == EPI_JACOBI_IDX) {
        // out-of-place (every listed row reads the OLD x, relaxation.h:393-399): y is indexed by position
        a.y[q.pos] = (q.d != T(0)) ? (one - a.omega) * q.xo + a.omega * ((q.b - s) / q.d) : q.xo;
    } else if constexpr (EPI == EPI_JACOBI_B) {
        a.y[row] = (q.d != T(0)) ? (one - a.omega) * q.xo + a.omega * s / q.d : q.xo;
    } else if constexpr (EPI == EPI_GS || EPI == EPI_GS_B || EPI == EPI_SOR) {
        T v;
        bool upd = q.d != T(0);
        if constexpr (EPI == EPI_GS) v = (q.b - s) / q.d;
        else if constexpr (EPI == EPI_GS_B) v = s / q.d;
        else v = a.omega * ((q.b - s) / q.d) + (one - a.omega) * q.xo;
        if constexpr (COH == 2 || COH == 3) {
            // granular sweep: ALWAYS publish (an untouched row publishes its old value), then
            // store x for the kernels that follow.  COH == 3: every participant sits on the same
            // XCD, the shared L2 is the coherence point -> an ordinary (L2-resident) store.
            if (!upd) v = a.x[row];
            if constexpr (COH == 2) __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (upd) a.y[row] = v;
        } else {
            if (upd) a.y[row] = v;
        }
    }
}

// VC = false: an instantiation WITHOUT the 8-bit value-code paths, for operators that carry none (every SA-level operator; the stencils' general
// form).  Round 6, one session (profiles/r06_ab_csr_stream.txt): the run-time tests for value codes in the staged kernel cost the fine-level residual
// of the 256^3 stencil 0.3355 -> 0.3145 ms (0.699 -> 0.745 of the HBM peak by SURVEY 8(d)), 2000^2 0.0649 -> 0.0574 -- what rounds 3 - 5 had lost since
// round 2's 0.313 (together with the nontemporal test inside the unrolled staging loop: 0.3475 -> 0.3307).
template <typename T, int EPI, int NPL, int COH, bool VC = true>
__device__ __forceinline__ void stream_block(const StreamArgs<T> &a, const int4 meta, unsigned char *smem_raw,
                                             double &sq)
{
    constexpr bool NEEDC = EpiTraits<EPI>::need_cols;
    const int cap = a.cap;
    // slots per LDS array: the window + the alignment slack below the first entry + the row phase's batch over-read
    const int slots = cap + ((VC && NPL == 2 && COH == 0 && a.Ax8) ? 16 : 8);
    T *prod = reinterpret_cast<T *>(smem_raw);
    int *cols = reinterpret_cast<int *>(smem_raw + sizeof(T) * (size_t)slots);
    const int tid = threadIdx.x;
    const int r0 = meta.x, r1 = meta.y, p0 = meta.z, p1 = meta.w;
    const T *vd = nullptr;
    int amask = (NPL == 4) ? ~3 : (NPL == 2) ? ~1 : ~0;
    if constexpr (VC && NPL == 2 && COH == 0) {
        if (a.Ax8) {
            // value dictionary -> LDS (behind the products and column ids); read after the barrier below
            T *d = reinterpret_cast<T *>(smem_raw + ((sizeof(T) * (size_t)slots + (NEEDC ? sizeof(int) * (size_t)slots : 0) + 15) & ~(size_t)15));
            if (tid < a.nvd) d[tid] = a.vdict[tid];
            vd = d;
            amask = (a.flags & 16) ? ~0 : ~3;
        }
    }
    if (p1 - p0 <= cap) {
        const int base = p0 & amask;
        int r = r0 + tid;
        RowPre<T> q;
        if (r < r1) q = row_prefetch<T, EPI, COH>(a, r);
#ifdef PAMG_FAKE_RUNTABLE
        if constexpr (COH == 0) {
            // 1 024 values of x (what the ~140 column runs of a range of an SA-level operator hold) by coalesced loads, then the barrier a run-table kernel needs
            const int c0 = (int)(((unsigned)blockIdx.x * 1024u) & ((1u << 20) - 1u));      // (with flag 4 only, on operators with >= 2^20 + 1024 columns: the SA levels of 256^3)
            if (a.flags & 4) { for (int i = tid; i < 1024; i += BLK) pamg_fake_xl[i] = (double)a.x[c0 + i]; }     // (ablation flag 4 selects the experiment)
            __syncthreads();
        }
#endif
        if (vd) __syncthreads();
        stage_products<T, NEEDC, NPL, COH, EpiTraits<EPI>::diag_flag>(a, p0, p1, base, prod, cols, vd);
        __syncthreads();
        if (a.flags & 8) {                                // ablation: no row phase
            if (tid == 0) a.y[r0] = prod[0];
            return;
        }
        while (r < r1) {
            T s = row_init<T, EPI>(q);
            row_accumulate<T, EPI>(s, prod, cols, q.lo - base, q.hi - base, q.row);
            row_finish<T, EPI, COH>(a, q, s, sq);
            r += BLK;
            if (r < r1) q = row_prefetch<T, EPI, COH>(a, r);
        }
    } else {
        // one over-long row (the host plan gives it a row range of its own): stream it
        // through LDS chunk by chunk while lane 0 carries the sequential running sum.
        RowPre<T> q = row_prefetch<T, EPI, COH>(a, r0);
        T s = row_init<T, EPI>(q);
        for (int c0 = p0; c0 < p1; c0 += cap) {
            const int c1 = min(c0 + cap, p1);
            const int base = c0 & amask;
            __syncthreads();
            stage_products<T, NEEDC, NPL, COH, EpiTraits<EPI>::diag_flag>(a, c0, c1, base, prod, cols, vd);
            __syncthreads();
            if (tid == 0) row_accumulate<T, EPI>(s, prod, cols, c0 - base, c1 - base, q.row);
        }
        if (tid == 0) row_finish<T, EPI, COH>(a, q, s, sq);
    }
}

template <typename T, int EPI, int NPL, bool VC = true>
__global__ __launch_bounds__(BLK) void csr_stream_kernel(const StreamArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double sq = 0.0;
    int blk = (int)blockIdx.x;
    if (a.flags & 2) {
        // XCD-aware order: workgroup b is observed to run on XCD b % 8; give each XCD a
        // contiguous chunk of row ranges so neighbouring ranges (which share x) share an L2.
        // Speed only -- any placement computes the same thing.
        const int chunk = (a.nblk + 7) >> 3;
        blk = (blk & 7) * chunk + (blk >> 3);
    }
    const bool live = blk < a.nblk;
    // a launch over a SUBSET of the row ranges (the interior / boundary halves of a row shard, pamg_dist.hip): the
    // launch index picks the range out of a list; everything per range (plan, window bases, partial) keeps its place
    if (live && a.blkmap) blk = a.blkmap[blk];
    if (live) {
        if (NPL == 2 && a.Aj16) {
            StreamArgs<T> aw = a;
            aw.wb = a.wbase[blk];
            stream_block<T, EPI, NPL, 0, VC>(aw, a.blkmeta[blk], smem_raw, sq);
        } else {
            stream_block<T, EPI, NPL, 0, VC>(a, a.blkmeta[blk], smem_raw, sq);
        }
    }
    if constexpr (EPI == EPI_SUMSQ) {
        __syncthreads();                                   // LDS reuse for the reduction
        const double tot = block_sum(sq, reinterpret_cast<double *>(smem_raw));
        if (threadIdx.x == 0 && live) a.partial[blk] = tot;
    }
}

// ---- row-gather form for operators that stream 16-bit column codes and 8-bit value codes.
// The PMC counters of the staged kernel on the 256^3 stencil (profiles/r03_pmc_stall_probe_*.json) show what binds it once the
// operator stream is 3 bytes per entry: the L1 (TCP) sees ~1 access per matrix ENTRY -- with several consecutive entries per
// lane, the 64 lanes of a gather instruction hit 64 different places of x -- and its access rate, not HBM, sets the time.
// Here the range's codes go to LDS with coalesced 16-byte loads (6 KB), and then lane l takes ROW r0 + l: instruction j gathers
// the j-th entry of 64 consecutive rows -- on a stencil 64 consecutive values of x, one 512-byte access -- and the lane adds its
// products in storage order straight from registers: no product array, no second pass.  Same order of additions, same bits.
template <typename T, int EPI>
__global__ __launch_bounds__(BLK) void csr_rowgather_kernel(const StreamArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool SKIPD = EpiTraits<EPI>::need_cols;          // Jacobi family: the diagonal never enters the sum
    const int cap = a.cap, tid = threadIdx.x;
    unsigned short *lc = reinterpret_cast<unsigned short *>(smem_raw);                        // [cap + 16] column codes
    unsigned char *lv = smem_raw + sizeof(unsigned short) * (size_t)(cap + 16);               // [cap + 16] value codes
    T *vd = reinterpret_cast<T *>(smem_raw + ((3 * (size_t)(cap + 16) + 15) & ~(size_t)15)); // dictionary
    double sq = 0.0;
    int blk = (int)blockIdx.x;
    if (a.flags & 2) {
        const int chunk = (a.nblk + 7) >> 3;
        blk = (blk & 7) * chunk + (blk >> 3);
    }
    const bool live = blk < a.nblk;
    if (live) {
        if (a.blkmap) blk = a.blkmap[blk];
        const int4 meta = a.blkmeta[blk];
        const int4 wb = a.wbase[blk];
        const int r0 = meta.x, r1 = meta.y, p0 = meta.z, p1 = meta.w;
        const int base = p0 & ~7;
        RowPre<T> q;
        int r = r0 + tid;
        if (r < r1) q = row_prefetch<T, EPI, 0>(a, r);
        if (tid < a.nvd) vd[tid] = a.vdict[tid];
        for (int g = base + 8 * tid; g < p1; g += 8 * BLK) {
            *reinterpret_cast<uint4 *>(lc + (g - base)) = *reinterpret_cast<const uint4 *>(a.Aj16 + g);
            *reinterpret_cast<uint2 *>(lv + (g - base)) = *reinterpret_cast<const uint2 *>(a.Ax8 + g);
        }
        __syncthreads();
        while (r < r1) {
            T s = row_init<T, EPI>(q);
            const int lo = q.lo - base, hi = q.hi - base;
            for (int j = lo; j < hi; j += 8) {
                int col[8];
                T xv[8], av[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = (j + k < hi) ? j + k : lo;                      // beyond the row: re-read its first entry
                    const unsigned c = lc[e];
                    const unsigned w = c >> 14;
                    col[k] = (w == 0 ? wb.x : w == 1 ? wb.y : w == 2 ? wb.z : wb.w) + (int)(c & 0x3FFFu);
                    av[k] = vd[lv[e]];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = a.x[col[k]];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (j + k < hi && (!SKIPD || col[k] != q.row)) {
                        const T pr = av[k] * xv[k];
                        if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
                        else s += pr;
                    }
                }
            }
            row_finish<T, EPI, 0>(a, q, s, sq);
            r += BLK;
            if (r < r1) q = row_prefetch<T, EPI, 0>(a, r);
        }
    }
    if constexpr (EPI == EPI_SUMSQ) {
        __syncthreads();
        const double tot = block_sum(sq, reinterpret_cast<double *>(smem_raw));
        if (threadIdx.x == 0 && live) a.partial[blk] = tot;
    }
}

// ---- row-pattern form (host plan: plan_rowpat, pamg_matrix.hip) for square operators with value codes whose rows are, nine
// times out of ten, one of <= 255 lists of (column - row, value) pairs -- constant-coefficient stencils.  Lane l takes row
// r0 + l and reads ONE byte: the number of its list; offsets and values come from a table in LDS (lanes of a wave mostly
// share the list: broadcast reads), the gathers x[row + offset] of 64 consecutive rows are 64 consecutive values, the
// products are added in the list's (= the row's storage) order.  Rows numbered 255 are walked through the code arrays like
// csr_rowgather_kernel's rows, straight from global memory.  Per regular row the operator costs 1 byte instead of
// 3 per entry + 4; bit-identical to every other form.
template <typename T, int EPI>
__device__ __forceinline__ RowPre<T> row_prefetch_noptr(const StreamArgs<T> &a, int r)
{
    RowPre<T> q;
    q.lo = q.hi = 0;
    q.row = r;
    q.pos = r;
    q.b = q.y = q.xo = q.d = T(0);
    if constexpr (EPI >= EPI_JACOBI) q.d = a.diag[r];
    if constexpr (EPI == EPI_RESID || EPI == EPI_AXPBY || EPI == EPI_ACC_AXPBY || EPI == EPI_SUMSQ || EPI >= EPI_JACOBI) q.b = a.b[r];
    if constexpr (EPI == EPI_ACC || EPI == EPI_ACC_AXPBY || EPI == EPI_ACCSEQ) q.y = a.y[r];
    if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_B) q.xo = a.x[r];
    return q;
}

template <typename T, int EPI>
__global__ __launch_bounds__(BLK) void csr_rowpat_kernel(const StreamArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool SKIPD = EpiTraits<EPI>::need_cols;          // Jacobi family: the diagonal never enters the sum
    const int tid = threadIdx.x, lmax = a.lmax, np = a.npat;
    int *tl = reinterpret_cast<int *>(smem_raw);               // [256] lengths
    int *to = tl + 256;                                        // [np * lmax] offsets
    T *tv = reinterpret_cast<T *>(smem_raw + (((size_t)(256 + np * lmax) * sizeof(int) + 15) & ~(size_t)15));   // [np * lmax] values
    T *vd = tv + (size_t)np * lmax;                            // value dictionary (irregular rows)
    {
        const int *gl = reinterpret_cast<const int *>(a.ptab);
        const T *gv = reinterpret_cast<const T *>(gl + 256 + np * lmax);
        for (int k = tid; k < 256 + np * lmax; k += BLK) tl[k] = gl[k];
        for (int k = tid; k < np * lmax; k += BLK) tv[k] = gv[k];
        if (tid < a.nvd) vd[tid] = a.vdict[tid];
    }
    double sq = 0.0;
    int blk = (int)blockIdx.x;
    if (a.flags & 2) {
        const int chunk = (a.nblk + 7) >> 3;
        blk = (blk & 7) * chunk + (blk >> 3);
    }
    const bool live = blk < a.nblk;
    if (live && a.blkmap) blk = a.blkmap[blk];
    int4 meta = make_int4(0, 0, 0, 0), wb = make_int4(0, 0, 0, 0);
    if (live) { meta = a.blkmeta[blk]; wb = a.wbase[blk]; }
    const int r0 = meta.x, r1 = meta.y;
    int r = r0 + tid;
    RowPre<T> q;
    unsigned pidv = 255;
    if (r < r1) { pidv = a.pid[r]; q = row_prefetch_noptr<T, EPI>(a, r); }
    __syncthreads();                                           // tables in place
    while (r < r1) {
        T s = row_init<T, EPI>(q);
        if (pidv != 255u) {
            const int len = tl[pidv];
            const int *po = to + pidv * lmax;
            const T *pv = tv + pidv * lmax;
            for (int j = 0; j < len; j += 8) {
                int col[8];
                T xv[8], av[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = (j + k < len) ? j + k : 0;                     // beyond the list: re-read its first entry
                    col[k] = r + po[e];
                    av[k] = pv[e];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = a.x[col[k]];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (j + k < len && (!SKIPD || col[k] != r)) {
                        const T pr = av[k] * xv[k];
                        if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
                        else s += pr;
                    }
                }
            }
        } else {
            // an irregular row (a domain corner beyond the table, a halo row of a shard, a long row): codes from global memory
            const int lo = a.Ap[r], hi = a.Ap[r + 1];
            for (int p = lo; p < hi; ++p) {
                const unsigned c = a.Aj16[p];
                const unsigned w = c >> 14;
                const int col = (w == 0 ? wb.x : w == 1 ? wb.y : w == 2 ? wb.z : wb.w) + (int)(c & 0x3FFFu);
                if (!SKIPD || col != r) {
                    const T pr = vd[a.Ax8[p]] * a.x[col];
                    if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
                    else s += pr;
                }
            }
        }
        row_finish<T, EPI, 0>(a, q, s, sq);
        r += BLK;
        if (r < r1) { pidv = a.pid[r]; q = row_prefetch_noptr<T, EPI>(a, r); }
    }
    if constexpr (EPI == EPI_SUMSQ) {
        __syncthreads();
        const double tot = block_sum(sq, reinterpret_cast<double *>(smem_raw));
        if (threadIdx.x == 0 && live) a.partial[blk] = tot;
    }
}

// Single-workgroup persistent sweep (gs_flow1_kernel): ONE workgroup walks all row ranges of a
// schedule, level after level, with __syncthreads() between them -- the scheduler of choice when
// the levels are so narrow (<= 2 row ranges on average) that there is nothing to share out.
template <typename T>
struct FlowArgs {
    StreamArgs<T> s;          // blkmeta = all row ranges of the schedule, level after level
    const int *level_blk;     // [nlevels+1] row-range offsets of the levels (DEVICE)
    int nlevels;
    unsigned *sync;           // [1] error flag
};

// ---- row-mask form (round 4; host plan: plan_row_masks, pamg_stream_plan.h).  When every list of the row-pattern table is
// the LONGEST list with some entries left out (a constant-coefficient stencil: the boundary rows drop neighbours, offsets and
// values of the ones that stay are the interior row's), a row is described by one byte: bit k = "entry k of the longest list is
// present".  Offsets and values are launch constants (scalar registers), so nothing waits for a table: lane l takes row
// blockIdx * BLK + l, no loop, no workgroup prologue, no barrier, and issues its mask byte, b and ALL gathers
// x[row + offset_k] at once (absent entries gather a clamped address and are left out of the sum) -- the access shape of the
// plain copy that reaches the HBM ceiling (bw_copy_block_kernel<1>).  Products are added in the list's (= the row's storage)
// order: bit-identical to every other form.  Rows with mask 0 (not a sub-list: irregular rows of the pattern plan) walk
// the operator's CSR arrays.
template <typename T>
struct RowMaskArgs {
    const unsigned char *mask;   // [nrows]
    int off[8];
    T val[8];
    int nrows, ncols;            // nrows: END of the rows this launch takes (the operator's rows, or a window's end)
    int row0;                    // first row of this launch (0, or the start of a window of rows: the interior rows of a row shard)
    int xcd_chunk;               // > 0: workgroup b works on rows of chunk (b & 7), so every XCD streams one contiguous eighth
    int xcd_share;               // > 0: workgroups per plane and XCD (plane-by-plane order, see the kernel)
};

// NT: the streams that are touched once (mask, b, the result) bypass the caches' retention (nontemporal), so x stays.  (Round 4 also had
// offsets -1 / +1 by whole-wave DPP shifts instead of gathers: measured neutral -- those gathers hit the L1 --, removed in round 5.)
template <typename T, int EPI, int NU, bool NT>
__global__ __launch_bounds__(BLK) void csr_rowmask_kernel(const StreamArgs<T> a, const RowMaskArgs<T> m)
{
    constexpr bool SKIPD = EpiTraits<EPI>::need_cols;          // Jacobi family: the diagonal never enters the sum
    // workgroups go to the XCDs round robin; share > 0: XCD j = blockIdx & 7 takes the j-th eighth of EVERY plane (plane = the
    // largest offset), planes in order -- the chip works on one plane at a time (one compact window of the HBM) and a row's
    // neighbours one plane up and down were, or will be, gathered through the same XCD's L2
    const int blk = rowmask_linear_block((int)blockIdx.x, m.xcd_chunk, m.xcd_share);
    const int r = m.row0 + blk * BLK + (int)threadIdx.x;
    const int rc = r < m.nrows ? r : m.nrows - 1;
    const unsigned mk = NT ? __builtin_nontemporal_load(m.mask + rc) : m.mask[rc];
    RowPre<T> q;
    if constexpr (NT) {
        q.lo = q.hi = 0;
        q.row = rc;
        q.pos = rc;
        q.b = q.y = q.xo = q.d = T(0);
        if constexpr (EPI >= EPI_JACOBI) q.d = __builtin_nontemporal_load(a.diag + rc);
        if constexpr (EPI == EPI_RESID || EPI == EPI_AXPBY || EPI == EPI_ACC_AXPBY || EPI >= EPI_JACOBI) q.b = __builtin_nontemporal_load(a.b + rc);
        if constexpr (EPI == EPI_ACC || EPI == EPI_ACC_AXPBY || EPI == EPI_ACCSEQ) q.y = __builtin_nontemporal_load(a.y + rc);
        if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_B) q.xo = a.x[rc];
    } else {
        q = row_prefetch_noptr<T, EPI>(a, rc);
    }
    T xv[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        int c = rc + m.off[k];
        c = c < 0 ? 0 : c;
        c = c < m.ncols ? c : m.ncols - 1;
        xv[k] = a.x[c];
    }
    if (r >= m.nrows) return;
    T s = row_init<T, EPI>(q);
    double sq = 0.0;
    // unconditional (mask 0 adds nothing), so that the gathers above do not sink behind a test of the mask byte
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        if (((mk >> k) & 1u) && (!SKIPD || m.off[k] != 0)) {
            const T pr = m.val[k] * xv[k];
            if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
            else s += pr;
        }
    }
    if (!mk) {
        const int lo = a.Ap[r], hi = a.Ap[r + 1];
        for (int p = lo; p < hi; ++p) {
            const int col = a.Aj[p];
            if (!SKIPD || col != r) {
                const T pr = a.Ax[p] * a.x[col];
                if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
                else s += pr;
            }
        }
    }
    if constexpr (NT && EPI == EPI_RESID) {
            const T t = q.b - s;
            __builtin_nontemporal_store(t, a.y + r);
            if (a.partial) __builtin_nontemporal_store(a.c * t, reinterpret_cast<T *>(a.partial) + r);     // h = c0 * r (row_finish)
        }
    else row_finish<T, EPI, 0>(a, q, s, sq);
}

// the residual's sum of squares in the row-mask form: the launch keeps the row ranges of the other forms (one partial per
// range, summed in the same order: the norm carries the table kernel's bits), lane l takes rows r0 + l, r0 + l + BLK, ...
template <typename T, int NU>
__global__ __launch_bounds__(BLK) void csr_rowmask_sumsq_kernel(const StreamArgs<T> a, const RowMaskArgs<T> m)
{
    __shared__ double red[BLK / 64];
    const int blk = (int)blockIdx.x;
    const int4 meta = a.blkmeta[blk];
    double sq = 0.0;
    for (int r = meta.x + (int)threadIdx.x; r < meta.y; r += BLK) {
        const unsigned mk = __builtin_nontemporal_load(m.mask + r);
        const T b = __builtin_nontemporal_load(a.b + r);
        T xv[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            int c = r + m.off[k];
            c = c < 0 ? 0 : c;
            c = c < m.ncols ? c : m.ncols - 1;
            xv[k] = a.x[c];
        }
        T s = T(0);
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            if ((mk >> k) & 1u) {
                const T pr = m.val[k] * xv[k];
                s += pr;
            }
        }
        if (!mk) {
            const int lo = a.Ap[r], hi = a.Ap[r + 1];
            for (int p = lo; p < hi; ++p) {
                const T pr = a.Ax[p] * a.x[a.Aj[p]];
                s += pr;
            }
        }
        const T t = b - s;
        sq += (double)t * (double)t;
    }
    const double tot = block_sum(sq, red);
    if (threadIdx.x == 0) a.partial[blk] = tot;
}

// ---- row-mask form on a lattice (round 4).  csr_rowmask_kernel asks the L2 for five 128-byte lines of x per 16 rows (the row's
// own line and its neighbours' one lattice line and one plane up and down); the counters say it is bound by the misses a
// CU's L1 can keep in flight, not by bytes.  When the longest list is (-P, -L, -1, 0, +1, +L, +P) -- the 7-point
// stencil of an nx x ny x nz lattice, L = nx, P = nx ny -- a workgroup takes a tile of 64 x 4 x KZ rows instead of 256
// consecutive ones: wave w the 64 rows [x0, x0 + 64) of lattice line y0 + w, every lane KZ planes of its (x, y).  The KZ + 2
// values of the lane's column are gathered once and serve as -P / 0 / +P operands of the KZ rows (registers), the +-L lines
// of the four waves are each other's own lines (the CU's L1), so the L2 is asked for (6 KZ + 8) / (4 KZ) lines per lattice
// line instead of 5.  Same masks, same products, same order of additions: bit-identical.
// Host guarantees: L % 64 == 0, (P / L) % 4 == 0, (nrows / P) % KZ == 0, nrows % P == 0.

template <typename T, int EPI, int KZ, bool NT, int WY>
__global__ __launch_bounds__(64 * WY) void csr_rowmask3d_kernel(const StreamArgs<T> a, const RowMaskArgs<T> m, const RowMaskLattice g)
{
    constexpr bool SKIPD = EpiTraits<EPI>::need_cols;
    constexpr bool NEEDB = (EPI == EPI_RESID || EPI == EPI_AXPBY || EPI == EPI_ACC_AXPBY || EPI >= EPI_JACOBI);
    constexpr bool NEEDY = (EPI == EPI_ACC || EPI == EPI_ACC_AXPBY || EPI == EPI_ACCSEQ);
    constexpr bool NEEDJ = (EPI == EPI_JACOBI || EPI == EPI_JACOBI_B);
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int r0 = m.row0 + rowmask_tile_row0(g, KZ, (int)blockIdx.x, wave, lane);
    const int last = m.ncols - 1;
    T xc[KZ + 2], xm1[KZ], xp1[KZ], xmL[KZ], xpL[KZ], bb[KZ], yy[KZ], dd[KZ];
    unsigned mk[KZ];
#pragma unroll
    for (int j = 0; j < KZ + 2; ++j) {
        int c = r0 + (j - 1) * g.P;
        c = c < 0 ? 0 : c;
        c = c < last ? c : last;
        xc[j] = a.x[c];
    }
#pragma unroll
    for (int j = 0; j < KZ; ++j) {
        const int r = r0 + j * g.P;
        mk[j] = NT ? __builtin_nontemporal_load(m.mask + r) : m.mask[r];
        if constexpr (NEEDB) bb[j] = NT ? __builtin_nontemporal_load(a.b + r) : a.b[r];
        if constexpr (NEEDY) yy[j] = NT ? __builtin_nontemporal_load(a.y + r) : a.y[r];
        if constexpr (EPI >= EPI_JACOBI) dd[j] = NT ? __builtin_nontemporal_load(a.diag + r) : a.diag[r];
        int c;
        c = r - g.L; c = c < 0 ? 0 : c; xmL[j] = a.x[c];
        c = r - 1; c = c < 0 ? 0 : c; xm1[j] = a.x[c];
        c = r + 1; c = c < last ? c : last; xp1[j] = a.x[c];
        c = r + g.L; c = c < last ? c : last; xpL[j] = a.x[c];
    }
#pragma unroll
    for (int j = 0; j < KZ; ++j) {
        const int r = r0 + j * g.P;
        RowPre<T> q;
        q.lo = q.hi = 0;
        q.row = r;
        q.pos = r;
        q.b = q.y = q.xo = q.d = T(0);
        if constexpr (NEEDB) q.b = bb[j];
        if constexpr (NEEDY) q.y = yy[j];
        if constexpr (EPI >= EPI_JACOBI) q.d = dd[j];
        if constexpr (NEEDJ) q.xo = xc[j + 1];
        T s = row_init<T, EPI>(q);
        double sq = 0.0;
        const T xs[7] = {xc[j], xmL[j], xm1[j], xc[j + 1], xp1[j], xpL[j], xc[j + 2]};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (((mk[j] >> k) & 1u) && (!SKIPD || k != 3)) {
                const T pr = m.val[k] * xs[k];
                if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
                else s += pr;
            }
        }
        if (!mk[j]) {
            const int lo = a.Ap[r], hi = a.Ap[r + 1];
            for (int p = lo; p < hi; ++p) {
                const int col = a.Aj[p];
                if (!SKIPD || col != r) {
                    const T pr = a.Ax[p] * a.x[col];
                    if constexpr (EpiTraits<EPI>::bsr_order) s -= pr;
                    else s += pr;
                }
            }
        }
        if constexpr (NT && EPI == EPI_RESID) {
            const T t = q.b - s;
            __builtin_nontemporal_store(t, a.y + r);
            if (a.partial) __builtin_nontemporal_store(a.c * t, reinterpret_cast<T *>(a.partial) + r);     // h = c0 * r (row_finish)
        }
        else row_finish<T, EPI, 0>(a, q, s, sq);
    }
}

// ---- software-pipelined row-range processing for the persistent sweeps ------------------
// Everything a row range needs that does NOT depend on other ranges (its slice of Aj/Ax, row
// pointers, row ids, diagonal, right-hand side) is fetched into registers one dependency
// range AHEAD, so that only the x gather -> LDS -> in-order row sum -> store chain remains on
// the critical path.
constexpr int MAXP = 4;        // prefetchable entry pairs per lane (ranges up to 2*MAXP*BLK entries)

template <typename T>
struct RangePre {
    int4 meta;
    int2 c[MAXP];
    typename Vec2<T>::type v[MAXP];
    RowPre<T> q;
    bool has_row, fits;
};

template <typename T, int EPI, int COH>
__device__ __forceinline__ void range_prefetch(const StreamArgs<T> &a, int blk, RangePre<T> &R)
{
    using T2 = typename Vec2<T>::type;
    const int tid = threadIdx.x;
    R.meta = a.blkmeta[blk];
    const int p0 = R.meta.z, p1 = R.meta.w, base = p0 & ~1;
    R.fits = (p1 - base) <= 2 * MAXP * BLK && (R.meta.y - R.meta.x) <= BLK && (p1 - p0) <= a.cap;
    R.has_row = false;
    if (!R.fits) return;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        if (q < p1) {
            R.c[k] = *reinterpret_cast<const int2 *>(a.Aj + q);
            R.v[k] = *reinterpret_cast<const T2 *>(a.Ax + q);
        }
    }
    const int r = R.meta.x + tid;
    R.has_row = r < R.meta.y;
    if (R.has_row) R.q = row_prefetch<T, EPI, COH>(a, r);
}

// phase 1 from prefetched registers (x gather + products into LDS); phase 2 after the caller's
// __syncthreads() via range_finish
template <typename T, int EPI, int COH>
__device__ __forceinline__ void range_stage(const StreamArgs<T> &a, const RangePre<T> &R, unsigned char *smem_raw)
{
    using T2 = typename Vec2<T>::type;
    T *prod = reinterpret_cast<T *>(smem_raw);
    int *cols = reinterpret_cast<int *>(smem_raw + sizeof(T) * (size_t)(a.cap + 8));
    const int tid = threadIdx.x;
    const int p0 = R.meta.z, p1 = R.meta.w, base = p0 & ~1;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        if (q < p1) {
            int2 cc = R.c[k];
            const bool ok0 = q >= p0, ok1 = q + 1 < p1;
            const T x0 = ok0 ? gather_x<COH>(a, cc.x) : T(0);
            const T x1 = ok1 ? gather_x<COH>(a, cc.y) : T(0);
            T2 pr;
            pr.x = (EpiTraits<EPI>::diag_flag && (cc.x & DIAG_BIT)) ? T(0) : R.v[k].x * x0;
            pr.y = (EpiTraits<EPI>::diag_flag && (cc.y & DIAG_BIT)) ? T(0) : R.v[k].y * x1;
            *reinterpret_cast<T2 *>(prod + (q - base)) = pr;
            if constexpr (EpiTraits<EPI>::need_cols) {
                cc.x &= COL_MASK;
                cc.y &= COL_MASK;
                *reinterpret_cast<int2 *>(cols + (q - base)) = cc;
            }
        }
    }
}

template <typename T, int EPI, int COH>
__device__ __forceinline__ void range_finish(const StreamArgs<T> &a, const RangePre<T> &R, unsigned char *smem_raw)
{
    if (!R.has_row) return;
    const T *prod = reinterpret_cast<const T *>(smem_raw);
    const int *cols = reinterpret_cast<const int *>(smem_raw + sizeof(T) * (size_t)(a.cap + 8));
    const int base = R.meta.z & ~1;
    double sq = 0.0;
    T s = row_init<T, EPI>(R.q);
    row_accumulate<T, EPI>(s, prod, cols, R.q.lo - base, R.q.hi - base, R.q.row);
    row_finish<T, EPI, COH>(a, R.q, s, sq);
}


// single-workgroup persistent sweep: ranges taken one after the other, separated by
// __syncthreads(), ordinary cached x accesses (same CU); the static operands of range k+1 are
// fetched while range k is being reduced.
template <typename T, int EPI, int NPL>
__global__ __launch_bounds__(BLK) void gs_flow1_kernel(const FlowArgs<T> g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const StreamArgs<T> &a = g.s;
    double sq = 0.0;
    const int b0 = g.level_blk[0], nb = g.level_blk[g.nlevels];
    RangePre<T> cur, nxt;
    nxt.fits = false; nxt.has_row = false;
    if (b0 < nb) range_prefetch<T, EPI, 0>(a, b0, cur);
    for (int blk = b0; blk < nb; ++blk) {
        // consecutive ranges are either in the same level (independent) or in consecutive
        // levels (dependent): the barrier after each covers both LDS reuse and visibility
        if (cur.fits) {
            range_stage<T, EPI, 0>(a, cur, smem_raw);
            if (blk + 1 < nb) range_prefetch<T, EPI, 0>(a, blk + 1, nxt);
            __syncthreads();
            range_finish<T, EPI, 0>(a, cur, smem_raw);
        } else {
            if (blk + 1 < nb) range_prefetch<T, EPI, 0>(a, blk + 1, nxt);
            stream_block<T, EPI, NPL, 0>(a, a.blkmeta[blk], smem_raw, sq);
        }
        __syncthreads();
        cur = nxt;
    }
}

// ---- granular ("sync-free") order-exact sweep ------------------------------------------------
// ONE persistent launch, no barriers between dependency levels.  Row ranges are taken in schedule
// (= dependency-level) order; an entry that needs the NEW value of an earlier row ("early" entry)
// waits for that row's datum to appear in the hand-off buffer xs (pre-filled with a sentinel: the
// published datum IS the flag, one write-through store -> one polled load per dependency hop),
// every other entry reads the OLD value from a.x.  Write-after-read hazards: with a structurally
// symmetric pattern a later row always waits for all its earlier neighbours, so nobody overwrites
// an old value that is still needed and a.x is the live vector; otherwise the host hands a.x = a
// snapshot of x taken before the sweep.  Deadlock-free because a range only ever waits on ranges
// with smaller ids and every workgroup takes its ranges in increasing order (spins are bounded and
// raise the error flag, which the solver's synchronous entry points report).  Measured
// (profiles/r01_microbench_gs_granular2_*.json): 1.8-2.7 us per dependency level against 4-7 us for
// a grid barrier or a kernel boundary per level.  What made the difference:
//  * the static operands of a workgroup's NEXT range are in registers before it starts to wait,
//    fetched AFTER the publishing stores of the current one and without a dependent address
//    chain (the range descriptor is loaded one range further ahead; b is loaded with the polls);
//  * a lane issues the loads of ALL its entries at once and re-polls only the missing ones, so a
//    poll round costs one memory round trip whatever the number of entries per lane;
//  * the grid is sized to ~8 dependency levels of look-ahead (<= 256 workgroups): idle pollers
//    load the memory system (a progress-counter gate for far-ahead workgroups was measured and
//    dropped -- the grid size does the same for free).
template <typename T>
struct GranArgs {
    StreamArgs<T> s;          // blkmeta = all row ranges of the schedule, level after level; xs = hand-off buffer
    int nblk;
    unsigned *ticket;         // XCD form: [0] ticket counter, [1] home XCD + 1 (both zeroed before the launch)
    long long *prof;          // nullptr or [nblk][8] time stamps (wall clock, 10 ns): diagnostics
};

template <typename T, int EPI>
__device__ __forceinline__ void range_stage_gran(const StreamArgs<T> &a, const RangePre<T> &R, unsigned char *smem_raw)
{
    using T2 = typename Vec2<T>::type;
    T *prod = reinterpret_cast<T *>(smem_raw);
    int *cols = reinterpret_cast<int *>(smem_raw + sizeof(T) * (size_t)(a.cap + 8));
    const int tid = threadIdx.x;
    const int p0 = R.meta.z, p1 = R.meta.w, base = p0 & ~1;
    // Every load below is UNCONDITIONAL with a selected address (entries that need nothing read
    // element 0): loads under divergent branches that write the same register make the compiler
    // drain the memory pipeline between them (write-after-write on the destination), which turned
    // the "batch" into up to eight serial round trips.
    T xv[2 * MAXP];
    int col[2 * MAXP];
    unsigned early = 0, old = 0;
    // the element read by entries that need nothing: one per wave, spread over the memory channels
    // (every idle lane of every waiting workgroup re-reading element 0 would hammer one channel)
    const int idle = (int)(((unsigned)blockIdx.x * 4u + (unsigned)(tid >> 6)) * 16u) % a.nidle;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        const bool v0 = q < p1 && q >= p0, v1 = q + 1 < p1;
        const int c0 = v0 ? R.c[k].x : DIAG_BIT, c1 = v1 ? R.c[k].y : DIAG_BIT;
        col[2 * k] = (c0 & DIAG_BIT) ? 0 : (c0 & COL_MASK);
        col[2 * k + 1] = (c1 & DIAG_BIT) ? 0 : (c1 & COL_MASK);
        if (!(c0 & DIAG_BIT)) { if (c0 & EARLY_BIT) early |= 1u << (2 * k); else old |= 1u << (2 * k); }
        if (!(c1 & DIAG_BIT)) { if (c1 & EARLY_BIT) early |= 1u << (2 * k + 1); else old |= 1u << (2 * k + 1); }
    }
    {
        T ve[2 * MAXP], vo[2 * MAXP];
#pragma unroll
        for (int j = 0; j < 2 * MAXP; ++j)
            ve[j] = __hip_atomic_load(a.xs + (((early >> j) & 1u) ? col[j] : idle), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 2 * MAXP; ++j) vo[j] = a.x[((old >> j) & 1u) ? col[j] : 0];
#pragma unroll
        for (int j = 0; j < 2 * MAXP; ++j) xv[j] = ((early >> j) & 1u) ? ve[j] : (((old >> j) & 1u) ? vo[j] : T(0));
    }
    unsigned pend = early;
    unsigned spins = 0;
    while (true) {
#pragma unroll
        for (int j = 0; j < 2 * MAXP; ++j)
            if (((pend >> j) & 1u) && Sentinel<T>::bits(xv[j]) != Sentinel<T>::value) pend &= ~(1u << j);
        if (!pend) break;
        __builtin_amdgcn_s_sleep(1);
        T t[2 * MAXP];
#pragma unroll
        for (int j = 0; j < 2 * MAXP; ++j)
            t[j] = __hip_atomic_load(a.xs + (((pend >> j) & 1u) ? col[j] : idle), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 2 * MAXP; ++j) xv[j] = ((pend >> j) & 1u) ? t[j] : xv[j];
        if (++spins > (1u << 22)) {                        // ~seconds: producer not resident / bug
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        if (q < p1) {
            int2 cc = R.c[k];
            T2 pr;
            pr.x = (cc.x & DIAG_BIT) ? T(0) : R.v[k].x * xv[2 * k];
            pr.y = (cc.y & DIAG_BIT) ? T(0) : R.v[k].y * xv[2 * k + 1];
            *reinterpret_cast<T2 *>(prod + (q - base)) = pr;
            if constexpr (EpiTraits<EPI>::need_cols) {
                cc.x &= COL_MASK;
                cc.y &= COL_MASK;
                *reinterpret_cast<int2 *>(cols + (q - base)) = cc;
            }
        }
    }
}

// static operands of the row range described by meta (everything but b / the old own value,
// whose addresses depend on the row id: those are loaded with the polls): no load here depends
// on another one, so the issuing wave never waits
template <typename T, int EPI>
__device__ __forceinline__ void range_prefetch_static(const StreamArgs<T> &a, const int4 meta, RangePre<T> &R)
{
    using T2 = typename Vec2<T>::type;
    const int tid = threadIdx.x;
    R.meta = meta;
    const int p0 = meta.z, p1 = meta.w, base = p0 & ~1;
    R.fits = (p1 - base) <= 2 * MAXP * BLK && (meta.y - meta.x) <= BLK && (p1 - p0) <= a.cap;
    R.has_row = false;
    if (!R.fits) return;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = base + 2 * tid + k * 2 * BLK;
        if (q < p1) {
            R.c[k] = *reinterpret_cast<const int2 *>(a.Aj + q);
            R.v[k] = *reinterpret_cast<const T2 *>(a.Ax + q);
        }
    }
    const int r = meta.x + tid;
    R.has_row = r < meta.y;
    R.q.b = R.q.y = R.q.xo = R.q.d = T(0);
    if (R.has_row) {
        R.q.lo = a.Ap[r];
        R.q.hi = a.Ap[r + 1];
        R.q.row = a.rid[r];
        R.q.pos = r;
        R.q.d = a.diag[r];
    }
}

// one row range of the granular sweep, from its prefetched static operands to the publishing stores
template <typename T, int EPI, int C>
__device__ __forceinline__ void gran_step(const GranArgs<T> &g, RangePre<T> &cur, int blk, unsigned char *smem_raw, double &sq)
{
    const StreamArgs<T> &a = g.s;
    const int tid = threadIdx.x;
    long long t0 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (g.prof && tid == 0) t0 = wall_clock64();
    if (cur.fits) {
        if (cur.has_row) {                                 // row-id dependent operands, in flight with the polls
            cur.q.b = a.b[cur.q.row];
            if constexpr (EPI == EPI_SOR) cur.q.xo = a.x[cur.q.row];
        }
        range_stage_gran<T, EPI>(a, cur, smem_raw);
        if (g.prof && tid == 0) t2 = wall_clock64();
        lds_barrier();
        if (g.prof && tid == 0) t3 = wall_clock64();
        range_finish<T, EPI, C>(a, cur, smem_raw);
    } else {
        stream_block<T, EPI, 2, C>(a, a.blkmeta[blk], smem_raw, sq);
    }
    if (g.prof && tid == 0) {
        t4 = wall_clock64();
        long long *o = g.prof + (size_t)blk * 8;
        o[0] = t0; o[1] = t0; o[2] = t2; o[3] = t3; o[4] = t4;
        o[5] = (long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF);
        o[6] = (long long)blockIdx.x;
    }
}

// id of this workgroup's next row range (AFTER the publishing stores of the current one) and the
// fetch of its static operands into `nxt`.  Static form: the range descriptor was loaded one range
// further ahead, so nothing here waits.  One-XCD form: a ticket, then descriptor -> operands.
template <typename T, int EPI, bool XCD>
__device__ __forceinline__ int gran_next(const GranArgs<T> &g, int blk, RangePre<T> &nxt, int4 &meta_nxt, int *sh_next)
{
    const StreamArgs<T> &a = g.s;
    const int G = (int)gridDim.x;
    int nb = blk + G;
    nxt.fits = false; nxt.has_row = false;
    if constexpr (XCD) {
        if (threadIdx.x == 0) *sh_next = (int)__hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_barrier();
        nb = *sh_next;
        if (nb < g.nblk) range_prefetch_static<T, EPI>(a, a.blkmeta[nb], nxt);
    } else {
        if (nb < g.nblk) {
            range_prefetch_static<T, EPI>(a, meta_nxt, nxt);
            if (nb + G < g.nblk) meta_nxt = a.blkmeta[nb + G];
        }
    }
    lds_barrier();                                         // LDS (and sh_next) are reused
    return nb;
}

// XCD = false: workgroup w owns ranges w, w+G, w+2G, ... (all G workgroups co-resident: the host
// keeps G <= the number of CUs).
// XCD = true: the hand-off stays inside ONE XCD's L2 (ordinary stores, L1-bypassing loads) -- for
// operators whose vectors fit that L2.  The first workgroup to arrive claims its XCD (read from
// the hardware register) as the home; every workgroup that runs elsewhere leaves at once, the
// others draw row ranges from a ticket counter.  Tickets go out in increasing order to workgroups
// that are already running, so the sweep completes for ANY placement and residency (at least the
// claiming workgroup takes part); placement decides speed only.
// The two operand sets P and Q alternate (the loop is unrolled by two): the prefetch lands in the
// registers it is consumed from, so nothing ever waits for it except its first use -- a copy
// "cur = nxt" at the end of the body would wait for the loads it just issued (measured: 3 us per
// range, which made the 100-220-range-wide fronts of the 256^3 fine grid service-bound).
template <typename T, int EPI, bool XCD>
__global__ __launch_bounds__(BLK) void gs_gran2_kernel(const GranArgs<T> g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int sh_next;
    constexpr int C = XCD ? 3 : 2;
    const StreamArgs<T> &a = g.s;
    const int G = (int)gridDim.x, tid = threadIdx.x;
    double sq = 0.0;
    RangePre<T> P, Q;
    P.fits = false; P.has_row = false;
    Q.fits = false; Q.has_row = false;
    int blk = (int)blockIdx.x;
    if constexpr (XCD) {
        if (tid == 0) {
            const unsigned me = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF) + 1u;     // HW_REG_XCC_ID[3:0] + 1
            unsigned home = 0u;
            __hip_atomic_compare_exchange_strong(g.ticket + 1, &home, me, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool mine = home == 0u || home == me;      // home holds the previous value
            sh_next = mine ? (int)__hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g.nblk;
        }
        __syncthreads();
        blk = sh_next;
        __syncthreads();
    }
    int4 meta_nxt = make_int4(0, 0, 0, 0);                  // descriptor of the range after the current one (static form)
    if (blk < g.nblk) {
        range_prefetch_static<T, EPI>(a, a.blkmeta[blk], P);
        if constexpr (!XCD) if (blk + G < g.nblk) meta_nxt = a.blkmeta[blk + G];
    }
    while (blk < g.nblk) {
        gran_step<T, EPI, C>(g, P, blk, smem_raw, sq);
        blk = gran_next<T, EPI, XCD>(g, blk, Q, meta_nxt, &sh_next);
        if (blk >= g.nblk) break;
        gran_step<T, EPI, C>(g, Q, blk, smem_raw, sq);
        blk = gran_next<T, EPI, XCD>(g, blk, P, meta_nxt, &sh_next);
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void fill_sentinel_kernel(T *xs, int64_t n)
{
    using B = typename Sentinel<T>::bits_t;
    B *p = reinterpret_cast<B *>(xs);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK)
        p[i] = Sentinel<T>::value;
}

// ------------------------------------------------------------------ BLAS-1 and friends
template <typename T>
__global__ __launch_bounds__(BLK) void vec_sumsq_kernel(const T *x, int64_t n, double *partial)
{
    __shared__ double sm[BLK / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const double v = (double)x[i];
        acc += v * v;
    }
    const double tot = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(BLK) void reduce_final_kernel(const double *partial, int n, double *out)
{
    __shared__ double sm[BLK / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += BLK) acc += partial[i];
    const double tot = block_sum(acc, sm);
    if (threadIdx.x == 0) out[0] = tot;
}

static __global__ __launch_bounds__(BLK) void reduce_mid_kernel(const double *partial, int n, double *mid)
{
    __shared__ double sm[BLK / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) acc += partial[i];
    const double tot = block_sum(acc, sm);
    if (threadIdx.x == 0) mid[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_axpy_kernel(int64_t n, T a, const T *x, T *y)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const T t = a * x[i];
        y[i] = y[i] + t;
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_scale_kernel(int64_t n, T a, const T *x, T *y)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK)
        y[i] = a * x[i];
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_gather_kernel(int64_t n, const int *idx, const T *src, T *dst)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK)
        dst[i] = src[idx[i]];
}

// one line of a Kaczmarz-type sweep (shared by the per-level and the persistent kernels)
template <typename T, bool NR>
__device__ __forceinline__ void kaczmarz_line(int i, const int *Lp, const int *Lj, const T *Lx, T *v, const T *b,
                                              const T *Dinv, T omega, T *xout, int lo_ = 0, int len_ = -1)
{
    const int lo = len_ >= 0 ? lo_ : Lp[i], hi = len_ >= 0 ? lo_ + len_ : Lp[i + 1];
    T s = T(0);
    for (int p = lo; p < hi; ++p) s += Lx[p] * v[Lj[p]];
    if constexpr (NR) {
        const T d = s * (Dinv[i] * omega);
        xout[i] = xout[i] + d;
        for (int p = lo; p < hi; ++p) {
            const T t = d * Lx[p];
            v[Lj[p]] = v[Lj[p]] - t;
        }
    } else {
        const T d = (b[i] - s) * Dinv[i] * omega;
        for (int p = lo; p < hi; ++p) {
            const T t = Lx[p] * d;
            v[Lj[p]] = v[Lj[p]] + t;
        }
    }
}

// One dependency level of a Kaczmarz-type sweep: every listed "line" (a row of A for Gauss-Seidel NE,
// a column of A for Gauss-Seidel NR) is handled by one lane -- in-order dot with the read-modify-write
// vector v, the step, then the in-order scatter update.  Lines of one level share no index of v (host
// schedule), so no atomics and the result is the sequential sweep's, bit for bit.
//   NR = false: amg_core::gauss_seidel_ne (relaxation.h:889-902): d = (b_i - s) * Dinv_i * omega; v[j] += a d
//   NR = true : amg_core::gauss_seidel_nr (relaxation.h:954-973): d = s * (Dinv_i * omega); x_i += d; v[j] -= d a
template <typename T, bool NR>
__global__ __launch_bounds__(BLK) void kaczmarz_level_kernel(const int *lines, int first, int count, const int *Lp,
                                                             const int *Lj, const T *Lx, T *v, const T *b,
                                                             const T *Dinv, T omega, T *xout)
{
    const int k = (int)blockIdx.x * BLK + (int)threadIdx.x;
    if (k >= count) return;
    kaczmarz_line<T, NR>(lines[first + k], Lp, Lj, Lx, v, b, Dinv, omega, xout);
}

// Persistent single-workgroup form (the scheduler for narrow schedules -- a few hundred lines per level: 2-D
// operators, SA coarse levels -- where one launch per level is pure launch latency): ONE launch walks all dependency
// levels with __syncthreads() between them (same-CU visibility, as gs_flow1_kernel), software-pipelined:
// a level is one dependent step; everything a step needs
// that does not depend on earlier steps is in registers before the step starts: the host lays the first KZ entries
// of every scheduled line out as a dense slab in schedule order (so their addresses follow from the position alone),
// and the line id of the level after next is fetched one level further ahead (Dinv / b need it).  What stays between
// two barriers: gather v -> in-order dot -> step -> in-order scatter.  Lines longer than KZ entries continue from the
// operator's arrays; levels wider than the workgroup fall back to kaczmarz_line for the surplus lines.
constexpr int KZ = 8;

template <typename T>
struct KzSched {
    const int *lines;      // [m] line id at scheduled position k
    const int *level_ptr;  // [nlevels+1]
    const int *len, *lo;   // [m] length of the line, offset of its first entry in Lj/Lx
    const int *ej;         // [m*KZ] indices of the first KZ entries (unused slots: 0)
    const T *ea;           // [m*KZ] their values (unused slots: 0)
    int nlevels;
};

template <typename T, bool NR>
__global__ __launch_bounds__(BLK) void kaczmarz_flow1p_kernel(const KzSched<T> g, const int *Lj, const T *Lx, T *v,
                                                              const T *b, const T *Dinv, T omega, T *xout)
{
    const int tid = (int)threadIdx.x;
    struct Pre { int i, len, lo; int j[KZ]; T a[KZ]; T dinv, bb; };
    Pre cur, nxt;
    auto line_id = [&](int lvl) -> int {
        if (lvl >= g.nlevels) return -1;
        const int k = g.level_ptr[lvl] + tid;
        return k < g.level_ptr[lvl + 1] ? g.lines[k] : -1;
    };
    auto fetch = [&](Pre &P, int lvl, int i) {               // i = line_id(lvl), already known
        P.i = i;
        P.len = 0; P.lo = 0; P.dinv = T(0); P.bb = T(0);
        if (i >= 0) {
            const int k = g.level_ptr[lvl] + tid;
            P.len = g.len[k];
            P.lo = g.lo[k];
#pragma unroll
            for (int e = 0; e < KZ; ++e) { P.j[e] = g.ej[(size_t)k * KZ + e]; P.a[e] = g.ea[(size_t)k * KZ + e]; }
            P.dinv = Dinv[i];
            if constexpr (!NR) P.bb = b[i];
        } else {
#pragma unroll
            for (int e = 0; e < KZ; ++e) { P.j[e] = 0; P.a[e] = T(0); }
        }
    };
    int i1 = line_id(0);
    fetch(cur, 0, i1);
    i1 = line_id(1);
    for (int l = 0; l < g.nlevels; ++l) {
        fetch(nxt, l + 1, i1);                               // operands of the next step, in flight during this one
        const int i2 = line_id(l + 2);
        if (cur.i >= 0) {
            T x8[KZ];
#pragma unroll
            for (int e = 0; e < KZ; ++e) x8[e] = v[cur.j[e]];          // slot 0.. beyond len reads v[0]: unused
            T s = T(0);
#pragma unroll
            for (int e = 0; e < KZ; ++e) if (e < cur.len) s += cur.a[e] * x8[e];
            for (int p = cur.lo + KZ; p < cur.lo + cur.len; ++p) s += Lx[p] * v[Lj[p]];
            T d;
            if constexpr (NR) { d = s * (cur.dinv * omega); xout[cur.i] = xout[cur.i] + d; }
            else d = (cur.bb - s) * cur.dinv * omega;
#pragma unroll
            for (int e = 0; e < KZ; ++e)
                if (e < cur.len) {
                    if constexpr (NR) { const T t = d * cur.a[e]; v[cur.j[e]] = v[cur.j[e]] - t; }
                    else { const T t = cur.a[e] * d; v[cur.j[e]] = v[cur.j[e]] + t; }
                }
            for (int p = cur.lo + KZ; p < cur.lo + cur.len; ++p) {
                if constexpr (NR) { const T t = d * Lx[p]; v[Lj[p]] = v[Lj[p]] - t; }
                else { const T t = Lx[p] * d; v[Lj[p]] = v[Lj[p]] + t; }
            }
        }
        // levels wider than the workgroup: the surplus lines, one after the other per lane
        for (int k = g.level_ptr[l] + BLK + tid; k < g.level_ptr[l + 1]; k += BLK)
            kaczmarz_line<T, NR>(g.lines[k], nullptr, Lj, Lx, v, b, Dinv, omega, xout, g.lo[k], g.len[k]);
        __syncthreads();
        cur = nxt;
        i1 = i2;
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_mul_kernel(int64_t n, const T *a, const T *b, T *y)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) y[i] = a[i] * b[i];
}

// dst[idx[i]] = src[idx[i]]: the listed entries of a full-length result (indexed block Jacobi: rows outside the list keep
// their old values)
template <typename T>
__global__ __launch_bounds__(BLK) void vec_copy_indexed_kernel(int64_t n, const int *idx, const T *src, T *dst)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const int k = idx[i];
        dst[k] = src[k];
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_scatter_kernel(int64_t n, const int *idx, const T *src, T *dst)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) dst[idx[i]] = src[i];
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_dot_kernel(const T *x, const T *y, int64_t n, double *partial)
{
    __shared__ double sm[BLK / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK)
        acc += (double)x[i] * (double)y[i];
    const double tot = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// p = beta*p + z   (the reference's `p *= beta; p += z`, krylov/_cg.py:167-168)
template <typename T>
__global__ __launch_bounds__(BLK) void vec_xpby_kernel(int64_t n, T beta, const T *z, T *p)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const T t = p[i] * beta;
        p[i] = t + z[i];
    }
}

// y = y + (sign * num/den) * x with the two scalars read from DEVICE memory (AMLI step sizes:
// the cycle stays capturable, no host round trip for alpha/beta)
template <typename T>
__global__ __launch_bounds__(BLK) void vec_axpy_ratio_kernel(int64_t n, const double *num, const double *den, T sign,
                                                             const T *x, T *y)
{
    const T a = sign * (T)(num[0] / den[0]);
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const T t = a * x[i];
        y[i] = y[i] + t;
    }
}

// max_i |u_i / x_i| over x_i != 0 (stagnation test of the reference's FGMRES, krylov/_fgmres.py:316-322):
// one partial maximum per workgroup, reduced by reduce_max_kernel
template <typename T>
__global__ __launch_bounds__(BLK) void vec_maxratio_kernel(const T *u, const T *x, int64_t n, double *partial)
{
    __shared__ double sm[BLK];
    double m = -1.0;                                        // -1: no x_i != 0 seen
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) {
        const T xi = x[i];
        if (xi != T(0)) m = fmax(m, fabs((double)(u[i] / xi)));
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

static __global__ __launch_bounds__(BLK) void reduce_max_kernel(const double *partial, int n, double *out)
{
    __shared__ double sm[BLK];
    double m = -1.0;
    for (int i = threadIdx.x; i < n; i += BLK) m = fmax(m, partial[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int st = BLK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

template <typename T>
__global__ __launch_bounds__(BLK) void vec_fill_kernel(int64_t n, T v, T *y)
{
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLK) y[i] = v;
}

// x = M b, dense row-major n x n (coarsest-level solve, multilevel.py:717-721): one wave
// per row, lanes stride the row, butterfly sum.
template <typename T>
__global__ __launch_bounds__(BLK) void dense_gemv_kernel(int n, const T *M, const T *b, T *x)
{
    const int row = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    double acc = 0.0;
    for (int k = lane; k < n; k += 64) acc += (double)M[(size_t)row * n + k] * (double)b[k];
    acc = wave_sum(acc);
    if (lane == 0) x[row] = (T)acc;
}

// ------------------------------------------------------------------ true block smoothers
// One lane per block row; blocks are tiny (bs <= 8) so the dense products stay in
// registers.  Arithmetic order follows amg_core::block_jacobi / block_gauss_seidel
// (relaxation.h:1021-1090, 1242-1298) and bsr_jacobi / bsr_gauss_seidel (:472-562, :185-266).

template <typename T>
struct BlockArgs {
    const int *bAp, *bAj;   // block CSR
    const T *Ax;            // blocks, row-major bs x bs
    const int *rid;         // block-row ids of this launch (level order) or nullptr
    const T *Dinv;          // (n_brow, bs, bs) or nullptr for the point variants
    const T *xsrc;          // values coupled through off-diagonal blocks
    T *xdst;
    const T *b;
    T omega;
    int bs, first, count;   // rows [first, first+count) of rid (or of 0..n_brow)
    int dirn;               // +1 forward, -1 backward (point sweep inside the diagonal block)
    T *xs;                  // granular sweep: hand-off buffer (sentinel = not published yet) or nullptr
    int nidle;              // granular sweep: elements of xs that idle lanes may read (>= 1)
    unsigned *err;          // granular sweep: error flag (spin bound hit)
};


template <int COH, typename T>
__device__ __forceinline__ void stxb(T *p, T v)
{
    if constexpr (COH == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// tail of one block row: acc = sum (block Jacobi/GS) or b - sum (BSR point variants) of the
// off-diagonal block products is given; apply Dinv / run the point sweep inside the diagonal
// block (dpos = offset of the diagonal block in Ax or -1) and store.  Arithmetic order follows
// amg_core::block_jacobi / block_gauss_seidel (relaxation.h:1021-1090, 1242-1298) and
// bsr_jacobi / bsr_gauss_seidel (:472-562, :185-266).
template <typename T, int KIND, int COH>
__device__ __forceinline__ void block_row_finish(const BlockArgs<T> &a, const int i, T (&acc)[MAXBS], long dpos)
{
    const int bs = a.bs, bb = bs * bs;
    const T one = T(1);
    T v[MAXBS];
    if constexpr (KIND == BLK_JACOBI || KIND == BLK_GS) {
        for (int k = 0; k < bs; ++k) acc[k] = a.b[(long)i * bs + k] - acc[k];
        const T *Di = a.Dinv + (long)i * bb;
        for (int r = 0; r < bs; ++r) {
            T d = T(0);
            for (int c = 0; c < bs; ++c) d += Di[r * bs + c] * acc[c];
            v[r] = d;
        }
        if constexpr (KIND == BLK_JACOBI) {
            for (int k = 0; k < bs; ++k)
                a.xdst[(long)i * bs + k] = (one - a.omega) * a.xsrc[(long)i * bs + k] + a.omega * v[k];
        } else {
            for (int k = 0; k < bs; ++k) {
                stxb<COH>(a.xdst + (long)i * bs + k, v[k]);
                if constexpr (COH == 2)            // granular sweep: the published datum is the flag
                    __hip_atomic_store(a.xs + (long)i * bs + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        // point sweep inside the diagonal block (untouched when no diagonal block is stored)
        T loc[MAXBS];
        for (int k = 0; k < bs; ++k) loc[k] = ldx<COH>(a.xsrc + (long)i * bs + k);
        if (dpos >= 0) {
            const T *D = a.Ax + dpos;
            const int k0 = a.dirn > 0 ? 0 : bs - 1, k1 = a.dirn > 0 ? bs : -1;
            for (int k = k0; k != k1; k += a.dirn) {
                T d = one;
                for (int kk = k0; kk != k1; kk += a.dirn) {
                    if (kk == k) d = D[k * bs + kk];
                    else acc[k] -= D[k * bs + kk] * loc[kk];
                }
                if (d != T(0)) {
                    if constexpr (KIND == PNT_JACOBI) {
                        a.xdst[(long)i * bs + k] = (one - a.omega) * loc[k] + a.omega * acc[k] / d;
                    } else {
                        loc[k] = acc[k] / d;            // GS: later points see the new value
                        stxb<COH>(a.xdst + (long)i * bs + k, loc[k]);
                    }
                } else if constexpr (KIND == PNT_JACOBI) {
                    a.xdst[(long)i * bs + k] = loc[k];
                }
            }
        } else if constexpr (KIND == PNT_JACOBI) {
            for (int k = 0; k < bs; ++k) a.xdst[(long)i * bs + k] = loc[k];
        }
        if constexpr (COH == 2 && KIND == PNT_GS) {
            // granular sweep: ALWAYS publish the whole block row (an untouched point publishes its old value)
            for (int k = 0; k < bs; ++k)
                __hip_atomic_store(a.xs + (long)i * bs + k, loc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// one block row i, everything by one lane (simple fallback path)
template <typename T, int KIND, int COH>
__device__ __forceinline__ void block_relax_row(const BlockArgs<T> &a, const int i)
{
    const int bs = a.bs, bb = bs * bs;
    T acc[MAXBS], v[MAXBS];
    long dpos = -1;
    if constexpr (KIND == BLK_JACOBI || KIND == BLK_GS) {
        for (int k = 0; k < bs; ++k) acc[k] = T(0);
    } else {
        for (int k = 0; k < bs; ++k) acc[k] = a.b[(long)i * bs + k];
    }
    for (int p = a.bAp[i]; p < a.bAp[i + 1]; ++p) {
        const int j = a.bAj[p];
        if (j == i) { dpos = (long)p * bb; continue; }
        const T *blk = a.Ax + (long)p * bb;
        const T *xj = a.xsrc + (long)j * bs;
        T xv[MAXBS];
        for (int c = 0; c < bs; ++c) xv[c] = ldx<COH>(xj + c);
        for (int r = 0; r < bs; ++r) {
            T d = T(0);
            for (int c = 0; c < bs; ++c) d += blk[r * bs + c] * xv[c];
            v[r] = d;
        }
        if constexpr (KIND == BLK_JACOBI || KIND == BLK_GS) {
            for (int k = 0; k < bs; ++k) acc[k] += v[k];
        } else {
            for (int k = 0; k < bs; ++k) acc[k] -= v[k];
        }
    }
    block_row_finish<T, KIND, COH>(a, i, acc, dpos);
}

// ---- LDS-streamed BSR relaxation ----------------------------------------------------------
// Same two-phase idea as csr_stream_kernel at block granularity: phase 1, one lane per (block,
// block-row r) pair computes the in-order dot of that block row with x_j -- all blocks of the
// row range in parallel, coalesced over the contiguous block values -- and parks it in LDS;
// phase 2, one lane per block row adds the per-block vectors in storage order and runs the
// (tiny, sequential) diagonal-block tail.  A block row with 40 6x6 blocks is 1440 multiply-adds
// that the one-lane-per-row kernel executes serially; here they spread over 240 lanes.
template <typename T>
struct BsrRange {
    const int4 *meta;      // [ranges] {first row, end row, first block, end block} in schedule order
    const int *pAp;        // [rows+1] cumulative block count in schedule order
    const int *pblk;       // [blocks] position of scheduled block q in Ax/bAj (nullptr = identity)
    const int *pbj;        // [blocks] block column of scheduled block q; bit 30: diagonal block, bit 31: early (needs the NEW x_j)
    const int *dpos;       // [rows] position in Ax (block index) of the row's diagonal block (last stored) or -1
    int capv;              // LDS capacity in values
};

template <typename T, int KIND, int COH>
__device__ __forceinline__ void bsr_range(const BlockArgs<T> &a, const BsrRange<T> &g, const int4 m, T *prodv)
{
    const int bs = a.bs, bb = bs * bs;
    const int r0 = m.x, r1 = m.y, q0 = m.z, q1 = m.w;
    const int tid = threadIdx.x;
    const int nent = (q1 - q0) * bs;
    if (nent <= g.capv) {
        for (int e = tid; e < nent; e += BLK) {
            const int q = q0 + e / bs, r = e % bs;
            const int cj = g.pbj[q];
            T d = T(0);
            if (!(cj & DIAG_BIT)) {                        // the diagonal block is staged as +0 (see DIAG_BIT)
                const long p = g.pblk ? g.pblk[q] : q;
                const T *Arow = a.Ax + p * bb + r * bs;
                const T *xj = a.xsrc + (long)(cj & COL_MASK) * bs;
                for (int c = 0; c < bs; ++c) d += Arow[c] * ldx<COH>(xj + c);
            }
            prodv[e] = d;
        }
        __syncthreads();
        for (int r = r0 + tid; r < r1; r += BLK) {
            const int i = a.rid ? a.rid[r] : r;
            T acc[MAXBS];
            const int dq = g.dpos[r];
            const long dpos = dq >= 0 ? (long)dq * bb : -1;
            if constexpr (KIND == BLK_JACOBI || KIND == BLK_GS) {
                for (int k = 0; k < bs; ++k) acc[k] = T(0);
            } else {
                for (int k = 0; k < bs; ++k) acc[k] = a.b[(long)i * bs + k];
            }
            const int qa = g.pAp[r], qb = g.pAp[r + 1];
            for (int q = qa; q < qb; ++q) {                // LDS only: nothing here waits on global memory
                const T *v = prodv + (q - q0) * bs;
                if constexpr (KIND == BLK_JACOBI || KIND == BLK_GS) {
                    for (int k = 0; k < bs; ++k) acc[k] += v[k];
                } else {
                    for (int k = 0; k < bs; ++k) acc[k] -= v[k];
                }
            }
            block_row_finish<T, KIND, COH>(a, i, acc, dpos);
        }
    } else {
        // over-long block row (own range): the simple one-lane path
        if (tid == 0) block_relax_row<T, KIND, COH>(a, a.rid ? a.rid[r0] : r0);
    }
}

template <typename T, int KIND>
__global__ __launch_bounds__(BLK) void bsr_stream_kernel(const BlockArgs<T> a, const BsrRange<T> g, int first)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bsr_range<T, KIND, 0>(a, g, g.meta[first + (int)blockIdx.x], reinterpret_cast<T *>(smem_raw));
}

// persistent order-exact block sweep over row ranges: levels separated by __syncthreads() (one
// workgroup) or by the arrival-counter barrier with coherent x traffic (several workgroups)
// ---- small block levels: ONE workgroup, x and b in LDS (round 4).  bsr_flow_kernel on a level of a few hundred dense-ish block
// rows (C5's 216 block rows of 6x6, ~41 blocks each: nearly every row its own dependency level) spends ~3.4 us per step in
// global round trips (the blocks of the step, x_j, the inverted diagonal block) and in ONE lane adding bs x blocks values.
// Here the iterate lives in LDS, the blocks / inverses of the NEXT range are in registers before the current one is finished,
// and bs lanes share a block row in the second phase (lane k adds the k-th components of the per-block vectors, still block
// after block).  Same products, same order of every sum: bit-identical to bsr_range / block_row_finish (and the reference's
// gemm loops, linalg.h:405-438, relaxation.h:185-266,1242-1298).  No hazards by construction: the ranges run one after the
// other, rows of one dependency level never touch each other's unknowns.
constexpr int SMALL_THREADS = 512;
constexpr int SMALL_NP = 3;                 // prefetched (block, row-of-block) entries per lane

template <typename T, int BSC>
struct SmallPre {
    int cj[SMALL_NP];
    T av[SMALL_NP][BSC ? BSC : MAXBS];
    T dv[BSC ? BSC : MAXBS];                // row k of the inverted diagonal block (BLK_GS) / of the diagonal block (PNT_GS) for this lane's (row, k)
    int row, dq;
};

template <typename T, int KIND, int BSC>
__device__ __forceinline__ void small_prefetch(const BlockArgs<T> &a, const BsrRange<T> &g, const int4 m, SmallPre<T, BSC> &P)
{
    constexpr int MB = BSC ? BSC : MAXBS;
    const int bs = BSC ? BSC : a.bs, bb = bs * bs;
    const int tid = threadIdx.x;
    const int q0 = m.z, nent = (m.w - m.z) * bs;
#pragma unroll
    for (int t = 0; t < SMALL_NP; ++t) {
        const int e = tid + t * SMALL_THREADS;
        P.cj[t] = DIAG_BIT;
        if (e < nent) {
            const int q = q0 + e / bs, r = e % bs;
            const int cj = g.pbj[q];
            P.cj[t] = cj;
            if (!(cj & DIAG_BIT)) {
                const long p = g.pblk ? g.pblk[q] : q;
                const T *Arow = a.Ax + p * bb + r * bs;
#pragma unroll
                for (int c = 0; c < MB; ++c) if (c < bs) P.av[t][c] = Arow[c];
            }
        }
    }
    const int nr = (m.y - m.x) * bs;
    P.row = -1; P.dq = -1;
    if (tid < nr) {
        const int rl = m.x + tid / bs, k = tid % bs;
        const int i = a.rid ? a.rid[rl] : rl;
        P.row = i;
        const int dq = g.dpos[rl];
        P.dq = dq;
        if constexpr (KIND == BLK_GS) {
            const T *Di = a.Dinv + (long)i * bb + k * bs;
#pragma unroll
            for (int c = 0; c < MB; ++c) if (c < bs) P.dv[c] = Di[c];
        }
    }
}

template <typename T, int KIND, int BSC>
__global__ __launch_bounds__(SMALL_THREADS) void bsr_small_kernel(const BlockArgs<T> a, const BsrRange<T> g, int nranges, int n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int MB = BSC ? BSC : MAXBS;
    const int bs = BSC ? BSC : a.bs, bb = bs * bs;
    const int tid = threadIdx.x;
    T *xl = reinterpret_cast<T *>(smem_raw);            // [n] the iterate
    T *bl = xl + n;                                      // [n] right-hand side
    T *prodv = bl + n;                                   // [capv] per (block, row-of-block) dot products of the current range
    T *rs = prodv + g.capv;                              // [SMALL_THREADS] b - sum of the current range's rows (BLK_GS exchange)
    for (int k = tid; k < n; k += SMALL_THREADS) { xl[k] = a.xsrc[k]; bl[k] = a.b[k]; }
    SmallPre<T, BSC> P, Q;
    if (nranges > 0) small_prefetch<T, KIND, BSC>(a, g, g.meta[0], P);
    __syncthreads();
    auto step = [&](const int4 m, SmallPre<T, BSC> &C, SmallPre<T, BSC> &N, int nxt) {
        const int q0 = m.z, nent = (m.w - m.z) * bs;
        // phase 1: dot of row r of block q with x_j, in c order
#pragma unroll
        for (int t = 0; t < SMALL_NP; ++t) {
            const int e = tid + t * SMALL_THREADS;
            if (e < nent) {
                T d = T(0);
                if (!(C.cj[t] & DIAG_BIT)) {
                    const T *xj = xl + (long)(C.cj[t] & COL_MASK) * bs;
#pragma unroll
                    for (int c = 0; c < MB; ++c) if (c < bs) d += C.av[t][c] * xj[c];
                }
                prodv[e] = d;
            }
        }
        for (int e = tid + SMALL_NP * SMALL_THREADS; e < nent; e += SMALL_THREADS) {          // beyond the prefetch window: straight from memory
            const int q = q0 + e / bs, r = e % bs;
            const int cj = g.pbj[q];
            T d = T(0);
            if (!(cj & DIAG_BIT)) {
                const long p = g.pblk ? g.pblk[q] : q;
                const T *Arow = a.Ax + p * bb + r * bs;
                const T *xj = xl + (long)(cj & COL_MASK) * bs;
                for (int c = 0; c < bs; ++c) d += Arow[c] * xj[c];
            }
            prodv[e] = d;
        }
        if (nxt < nranges) small_prefetch<T, KIND, BSC>(a, g, g.meta[nxt], N);        // in flight during the second phase
        lds_barrier();
        // phase 2: lane (row, k) adds the k-th components block after block
        const bool mine = C.row >= 0;
        const int rl = m.x + tid / bs, k = tid % bs;
        T acc = T(0);
        if (mine) {
            const int i = C.row;
            if constexpr (KIND == BLK_GS) acc = T(0);
            else acc = bl[(long)i * bs + k];
            const int qa = g.pAp[rl], qb = g.pAp[rl + 1];
            for (int q = qa; q < qb; ++q) {
                const T v = prodv[(q - q0) * bs + k];
                if constexpr (KIND == BLK_GS) acc += v;
                else acc -= v;
            }
            if constexpr (KIND == BLK_GS) acc = bl[(long)i * bs + k] - acc;
            rs[tid] = acc;
        }
        lds_barrier();
        if (mine) {
            const int i = C.row;
            const int base = tid - k;                                               // lane of component 0 of this block row
            if constexpr (KIND == BLK_GS) {
                T d = T(0);
#pragma unroll
                for (int c = 0; c < MB; ++c) if (c < bs) d += C.dv[c] * rs[base + c];
                xl[(long)i * bs + k] = d;
                a.xdst[(long)i * bs + k] = d;
            } else if (k == 0 && C.dq >= 0) {
                // point sweep inside the diagonal block: sequential over its points (relaxation.h:247-259), one lane per block row
                const T *D = a.Ax + (long)C.dq * bb;
                T loc[MB], ac[MB];
                for (int kk = 0; kk < bs; ++kk) { loc[kk] = xl[(long)i * bs + kk]; ac[kk] = rs[base + kk]; }
                const int k0 = a.dirn > 0 ? 0 : bs - 1, k1 = a.dirn > 0 ? bs : -1;
                for (int kk = k0; kk != k1; kk += a.dirn) {
                    T d = T(1);
                    for (int c = k0; c != k1; c += a.dirn) {
                        if (c == kk) d = D[kk * bs + c];
                        else ac[kk] -= D[kk * bs + c] * loc[c];
                    }
                    if (d != T(0)) {
                        loc[kk] = ac[kk] / d;
                        xl[(long)i * bs + kk] = loc[kk];
                        a.xdst[(long)i * bs + kk] = loc[kk];
                    }
                }
            }
        }
        lds_barrier();
    };
    int blk = 0;
    while (blk < nranges) {
        step(g.meta[blk], P, Q, blk + 1);
        if (++blk >= nranges) break;
        step(g.meta[blk], Q, P, blk + 1);
        ++blk;
    }
}

template <typename T, int KIND, bool COH>
__global__ __launch_bounds__(BLK) void bsr_flow_kernel(const BlockArgs<T> a, const BsrRange<T> g, const int *level_blk,
                                                       int nlevels, unsigned *sync)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *prodv = reinterpret_cast<T *>(smem_raw);
    const int G = (int)gridDim.x;
    int lb = level_blk[0];
    for (int l = 0; l < nlevels; ++l) {
        const int le = level_blk[l + 1];
        for (int blk = lb + (int)blockIdx.x; blk < le; blk += G) {
            bsr_range<T, KIND, COH ? 1 : 0>(a, g, g.meta[blk], prodv);
            __syncthreads();
        }
        lb = le;
        if constexpr (COH) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned target = (unsigned)(l + 1) * (unsigned)G;
                __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) {
                        __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- granular order-exact block sweep -------------------------------------------------------
// The block twin of gs_gran2_kernel: one persistent launch, no barrier between dependency levels.
// A lane owns up to GE (block, row-in-block) pairs of its row range.  It first fetches everything
// that does not depend on other ranges (its row of each block: GE x bs values in registers), then
// loads ONE x value per pair -- element e of the range's gathered x -- polling the hand-off
// buffer for the values of block rows visited earlier in the sweep (batch polling: all pending
// loads per round), parks x in LDS, computes its in-order dots from LDS, and the row phase
// finishes and publishes as in the barrier form.
constexpr int GE = 6;          // pairs per lane: ranges of up to GE * BLK values (the default plan's 1536)

// tail of one block row for the granular sweep: like block_row_finish<.., 2>, but every operand that
// does not depend on other rows was fetched before the wait -- b and the row's own old values in
// registers, the diagonal block (PNT_GS) or its inverse (BLK_GS) through D (LDS or global; nullptr =
// no diagonal block stored)
template <typename T, int KIND, int BS = 0>
__device__ __forceinline__ void block_row_finish_gran(const BlockArgs<T> &a, const int i, T (&acc)[MAXBS], const T *D,
                                                      const T (&breg)[MAXBS], T (&loc)[MAXBS])
{
    // BS > 0: the block size is a compile-time constant -- every loop below unrolls and acc / loc / v stay in
    // registers with static indices (run-time bounds cost a select chain per access); BS == 0: generic
    constexpr int NB = BS > 0 ? BS : MAXBS;
    const int bs = BS > 0 ? BS : a.bs;
    if constexpr (KIND == BLK_GS) {
        T v[MAXBS];
#pragma unroll
        for (int k = 0; k < NB; ++k) if (BS > 0 || k < bs) acc[k] = breg[k] - acc[k];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            if (BS > 0 || r < bs) {
                T d = T(0);
#pragma unroll
                for (int c = 0; c < NB; ++c) if (BS > 0 || c < bs) d += D[r * bs + c] * acc[c];
                v[r] = d;
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (BS > 0 || k < bs) {
                a.xdst[(long)i * bs + k] = v[k];
                __hip_atomic_store(a.xs + (long)i * bs + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        if (D) {
            if (a.dirn > 0) {
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    if (BS > 0 || k < bs) {
                        T d = T(1);
#pragma unroll
                        for (int kk = 0; kk < NB; ++kk) {
                            if (BS > 0 || kk < bs) {
                                if (kk == k) d = D[k * bs + kk];
                                else acc[k] -= D[k * bs + kk] * loc[kk];
                            }
                        }
                        if (d != T(0)) {
                            loc[k] = acc[k] / d;            // later points see the new value
                            a.xdst[(long)i * bs + k] = loc[k];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int k = NB - 1; k >= 0; --k) {
                    if (BS > 0 || k < bs) {
                        T d = T(1);
#pragma unroll
                        for (int kk = NB - 1; kk >= 0; --kk) {
                            if (BS > 0 || kk < bs) {
                                if (kk == k) d = D[k * bs + kk];
                                else acc[k] -= D[k * bs + kk] * loc[kk];
                            }
                        }
                        if (d != T(0)) {
                            loc[k] = acc[k] / d;
                            a.xdst[(long)i * bs + k] = loc[k];
                        }
                    }
                }
            }
        }
        // ALWAYS publish the whole block row (an untouched point publishes its old value)
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (BS > 0 || k < bs) __hip_atomic_store(a.xs + (long)i * bs + k, loc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, int KIND, int BS = 0>
__device__ __forceinline__ void bsr_range_gran(const BlockArgs<T> &a, const BsrRange<T> &g, const int4 m, T *xl, T *prodv, T *dl)
{
    constexpr int NB = BS > 0 ? BS : MAXBS;               // BS > 0: compile-time block size (static loops, registers)
    const int bs = BS > 0 ? BS : a.bs, bb = bs * bs;
    const int r0 = m.x, r1 = m.y, q0 = m.z, q1 = m.w;
    const int tid = threadIdx.x;
    const int nent = (q1 - q0) * bs;
    if (nent <= g.capv && nent <= GE * BLK) {
        // ---- everything that does not depend on other ranges, before the wait
        T Areg[GE][NB];
        int cjs[GE];
#pragma unroll
        for (int k = 0; k < GE; ++k) {
            const int e = tid + k * BLK;
            cjs[k] = DIAG_BIT;
            if (e < nent) {
                const int q = q0 + e / bs, r = e % bs;
                cjs[k] = g.pbj[q];
                const long p = g.pblk ? g.pblk[q] : q;
                const T *Arow = a.Ax + p * bb + r * bs;
#pragma unroll
                for (int c = 0; c < NB; ++c) Areg[k][c] = (BS > 0 || c < bs) ? Arow[c] : T(0);
            }
        }
        const int nrow = r1 - r0;
        const bool dl_ok = nrow * bb <= g.capv;              // diagonal blocks (or inverses) of the range fit the LDS slab
        if (dl_ok) {
            for (int t = tid; t < nrow * bb; t += BLK) {
                const int rr = t / bb, u = t - rr * bb;
                T v = T(0);
                if constexpr (KIND == BLK_GS) {
                    v = a.Dinv[(long)(a.rid ? a.rid[r0 + rr] : r0 + rr) * bb + u];
                } else {
                    const int dq = g.dpos[r0 + rr];
                    if (dq >= 0) v = a.Ax[(long)dq * bb + u];
                }
                dl[t] = v;
            }
        }
        // compile-time block sizes: the block-row sums are formed by BS lanes per block row (lane = (row, component);
        // each still adds its component block after block in storage order), the small dense finish by one lane per row
        int pq0 = 0, pq1 = 0;
        const bool par_ok = KIND == BLK_GS && BS > 0 && nrow * BS <= BLK && nrow * BS <= g.capv;   // (the point sweep's b - v0 - v1 ... cannot be split)
        if (par_ok && tid < nrow * BS) {
            pq0 = g.pAp[r0 + tid / BS];
            pq1 = g.pAp[r0 + tid / BS + 1];
        }
        const int myr = r0 + tid;
        const bool has_row = myr < r1;
        int i = 0, qa = 0, qb = 0, dq = -1;
        T breg[MAXBS], loc[MAXBS];
        if (has_row) {
            i = a.rid ? a.rid[myr] : myr;
            qa = g.pAp[myr]; qb = g.pAp[myr + 1];
            dq = g.dpos[myr];
#pragma unroll
            for (int k = 0; k < MAXBS; ++k) {
                breg[k] = (k < NB && k < bs) ? a.b[(long)i * bs + k] : T(0);
                loc[k] = (KIND == PNT_GS && k < NB && k < bs) ? a.xsrc[(long)i * bs + k] : T(0);
            }
        }
        // ---- the wait: one x value per (block, row-in-block) pair, batch-polled
        // unconditional loads with selected addresses (see range_stage_gran)
        T xv[GE];
        long at[GE];
        unsigned early = 0, old = 0;
        const long idle = (long)((((unsigned)blockIdx.x * 4u + (unsigned)(tid >> 6)) * 16u) % (unsigned)a.nidle);
#pragma unroll
        for (int k = 0; k < GE; ++k) {
            const int e = tid + k * BLK;
            at[k] = 0;
            if (e < nent && !(cjs[k] & DIAG_BIT)) {
                at[k] = (long)(cjs[k] & COL_MASK) * bs + e % bs;
                if (cjs[k] & EARLY_BIT) early |= 1u << k; else old |= 1u << k;
            }
        }
        {
            T ve[GE], vo[GE];
#pragma unroll
            for (int k = 0; k < GE; ++k)
                ve[k] = __hip_atomic_load(a.xs + (((early >> k) & 1u) ? at[k] : idle), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < GE; ++k) vo[k] = a.xsrc[((old >> k) & 1u) ? at[k] : 0];
#pragma unroll
            for (int k = 0; k < GE; ++k) xv[k] = ((early >> k) & 1u) ? ve[k] : (((old >> k) & 1u) ? vo[k] : T(0));
        }
        unsigned pend = early;
        unsigned spins = 0;
        while (true) {
#pragma unroll
            for (int k = 0; k < GE; ++k)
                if (((pend >> k) & 1u) && Sentinel<T>::bits(xv[k]) != Sentinel<T>::value) pend &= ~(1u << k);
            if (!pend) break;
            __builtin_amdgcn_s_sleep(1);
            T t[GE];
#pragma unroll
            for (int k = 0; k < GE; ++k)
                t[k] = __hip_atomic_load(a.xs + (((pend >> k) & 1u) ? at[k] : idle), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < GE; ++k) xv[k] = ((pend >> k) & 1u) ? t[k] : xv[k];
            if (++spins > (1u << 22)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
#pragma unroll
        for (int k = 0; k < GE; ++k) {
            const int e = tid + k * BLK;
            if (e < nent) xl[e] = xv[k];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < GE; ++k) {
            const int e = tid + k * BLK;
            if (e < nent) {
                T d = T(0);
                if (!(cjs[k] & DIAG_BIT)) {
                    const T *xq = xl + (e / bs) * bs;
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        if (BS > 0 || c < bs) d += Areg[k][c] * xq[c];
                }
                prodv[e] = d;
            }
        }
        lds_barrier();
        if (par_ok) {
            if (tid < nrow * BS) {
                const int k = tid % BS;
                T sk = T(0);
                const T *v = prodv + (pq0 - q0) * bs + k;
                for (int q = pq0; q < pq1; ++q, v += bs) sk += *v;       // component k, block after block
                xl[tid] = sk;                                            // the x staging area is free again
            }
            lds_barrier();
        }
        if (has_row) {
            T acc[MAXBS];
#pragma unroll
            for (int k = 0; k < NB; ++k) acc[k] = (KIND == BLK_GS) ? T(0) : breg[k];
            if (par_ok) {
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const T sk = xl[(myr - r0) * NB + k];
                    if constexpr (KIND == BLK_GS) acc[k] = sk;           // 0 + p0 + p1 ... == the running sum from +0
                    else acc[k] = breg[k];
                }
            }
            for (int q = par_ok && KIND == BLK_GS ? qb : qa; q < qb; ++q) {
                const T *v = prodv + (q - q0) * bs;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    if (BS > 0 || k < bs) {
                        if constexpr (KIND == BLK_GS) acc[k] += v[k];
                        else acc[k] -= v[k];
                    }
                }
            }
            const T *D;
            if constexpr (KIND == BLK_GS) D = dl_ok ? dl + (myr - r0) * bb : a.Dinv + (long)i * bb;
            else D = dq < 0 ? nullptr : (dl_ok ? dl + (myr - r0) * bb : a.Ax + (long)dq * bb);
            block_row_finish_gran<T, KIND, BS>(a, i, acc, D, breg, loc);
        }
    } else if (tid == 0) {
        // over-long block row (a range of its own): one lane, block after block
        const int r = r0;
        const int i = a.rid ? a.rid[r] : r;
        T acc[MAXBS], v[MAXBS];
        const int dq = g.dpos[r];
        const long dpos = dq >= 0 ? (long)dq * bb : -1;
        if constexpr (KIND == BLK_GS) {
            for (int k = 0; k < bs; ++k) acc[k] = T(0);
        } else {
            for (int k = 0; k < bs; ++k) acc[k] = a.b[(long)i * bs + k];
        }
        for (int q = g.pAp[r]; q < g.pAp[r + 1]; ++q) {
            const int cj = g.pbj[q];
            if (cj & DIAG_BIT) continue;
            const T *blk = a.Ax + (long)(g.pblk ? g.pblk[q] : q) * bb;
            T xq[MAXBS];
            for (int c = 0; c < bs; ++c) {
                const long at = (long)(cj & COL_MASK) * bs + c;
                xq[c] = (cj & EARLY_BIT) ? spin_value<T>(a.xs, (int)at, a.err) : a.xsrc[at];
            }
            for (int rr = 0; rr < bs; ++rr) {
                T d = T(0);
                for (int c = 0; c < bs; ++c) d += blk[rr * bs + c] * xq[c];
                v[rr] = d;
            }
            if constexpr (KIND == BLK_GS) {
                for (int k = 0; k < bs; ++k) acc[k] += v[k];
            } else {
                for (int k = 0; k < bs; ++k) acc[k] -= v[k];
            }
        }
        block_row_finish<T, KIND, 2>(a, i, acc, dpos);
    }
}

template <typename T, int KIND, int BS = 0>
__global__ __launch_bounds__(BLK) void bsr_gran_kernel(const BlockArgs<T> a, const BsrRange<T> g, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *xl = reinterpret_cast<T *>(smem_raw);
    T *prodv = xl + (g.capv + 8);
    T *dl = prodv + (g.capv + 8);
    for (int blk = (int)blockIdx.x; blk < nblk; blk += (int)gridDim.x) {
        bsr_range_gran<T, KIND, BS>(a, g, g.meta[blk], xl, prodv, dl);
        lds_barrier();                                     // LDS is reused by the next row range
    }
}

}  // namespace pamg
