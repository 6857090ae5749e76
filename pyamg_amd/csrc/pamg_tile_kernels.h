// pamg_tile_kernels.h -- the TILED order-exact sweep (gfx950).  Plan and block layout: pamg_tile_plan.h.
//
// One persistent workgroup of FOUR specialised waves per tile; they meet only in LDS (no s_barrier,
// no global flag):
//   * wave 1, the LOADER: streams the tile's step blocks (entry codes | entry values | row records, one
//     fixed-size block per step) into a ring of D LDS slots with LDS-DMA (global_load_lds_dwordx4, 1 KiB
//     per instruction, no registers), Q steps in flight, counted with explicit s_waitcnt vmcnt.
//   * waves 2 and 3, the GATHERERS (alternating batches of KG steps): walk the two gather lists of a landed step and
//     fetches what the step needs from global memory -- OLD values x[j], NEW values of other tiles from
//     the hand-off buffer xs (polled: the published datum IS the flag), b and the old value of each row --
//     and turns every entry it owns into the finished product a_ij * x_j in place.
//   * wave 0, the COMPUTE wave: touches LDS only.  Per step: entries whose x_j was produced by an earlier
//     step of THIS tile (the step's "local" list) are multiplied with the value from the LDS ring; then one lane per row sums its
//     products strictly in storage order, applies the reference's update, writes the new value to the
//     ring and fires the global stores (x, and xs for rows with consumers in other tiles).  It never
//     loads from global memory, so it never waits on the memory counter: stores are fire-and-forget.
// The dependency chain inside a tile therefore costs one LDS round trip + one in-order row sum per
// level; global latency (operator stream, gathers) is hidden by the ring, the cross-tile hand-off is
// paid once per tile boundary on the critical path.
// Progress words in LDS: landed (loader -> gatherers), ready x 2 (gatherer -> compute), done (compute ->
// loader, frees slots).  Deadlock freedom: every workgroup takes its steps in non-decreasing level order
// and a step only waits for rows of strictly lower levels, so the lexicographically smallest unfinished
// (level, tile) can always run PROVIDED all G workgroups are resident (the host sizes G by the occupancy
// query; spins are bounded, a timeout raises the error flag and the workgroup winds down).
#pragma once
#include "pamg_kernels.h"
#include "pamg_tile_plan.h"

namespace pamg {

#define PAMG_LDS __attribute__((address_space(3)))

typedef int tile_v4i __attribute__((ext_vector_type(4)));   // plain vector types: live in any address space
typedef int tile_v2i __attribute__((ext_vector_type(2)));

constexpr int TILE_THREADS = 256;             // compute wave, loader wave, two gather waves
constexpr unsigned TILE_SPIN_LIMIT = 1u << 21;

template <typename T>
struct TileArgs {
    const unsigned char *blocks;  // step blocks, tile after tile
    const int *tile_step;         // [G+1]
    const T *x;                   // OLD values (the live vector, or a snapshot for non-symmetric patterns)
    T *xs;                        // global hand-off buffer (sentinel-filled)
    T *y;                         // destination (the live vector)
    const T *b;
    unsigned *err;
    long long *prof;              // nullptr or [nsteps][8]: compute {start, end, XCD, tile}, loader issue, gatherer {issue, finish start, ready}
    T omega;
    int W;                        // ring slots (power of two)
    int D;                        // LDS slots (steps resident per tile)
    int Q;                        // steps the loader keeps in flight behind the landed mark
    int NCH, NV;                  // 1-KiB chunks of lists / of entry values per step block
    int G;
    int nidle;
};

template <typename T>
__device__ __forceinline__ PAMG_LDS T *ldsp(unsigned off) { return (PAMG_LDS T *)(size_t)off; }
__device__ __forceinline__ unsigned lds_flag_load(unsigned off)
{
    return __hip_atomic_load(ldsp<unsigned>(off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_store(unsigned off, unsigned v)
{
    __hip_atomic_store(ldsp<unsigned>(off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS writes of this wave are complete (and the compiler keeps memory operations on their side of it)
__device__ __forceinline__ void lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }

// one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane*16 (M0 = wave-uniform
// LDS base; written in the same statement that uses it: the compiler does not preserve M0 around asm)
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// at most n (uniform, 0..63) loads of this wave still in flight
__device__ __forceinline__ void wait_vmcnt_rt(int n)
{
#define PAMG_VM_CASE(k) case k: wait_vmcnt<k>(); break;
#define PAMG_VM_CASE8(k) PAMG_VM_CASE(k) PAMG_VM_CASE(k + 1) PAMG_VM_CASE(k + 2) PAMG_VM_CASE(k + 3) PAMG_VM_CASE(k + 4) PAMG_VM_CASE(k + 5) PAMG_VM_CASE(k + 6) PAMG_VM_CASE(k + 7)
    switch (n) {
        PAMG_VM_CASE8(0) PAMG_VM_CASE8(8) PAMG_VM_CASE8(16) PAMG_VM_CASE8(24) PAMG_VM_CASE8(32) PAMG_VM_CASE8(40) PAMG_VM_CASE8(48)
        PAMG_VM_CASE(56) PAMG_VM_CASE(57) PAMG_VM_CASE(58) PAMG_VM_CASE(59) PAMG_VM_CASE(60) PAMG_VM_CASE(61) PAMG_VM_CASE(62)
        default: wait_vmcnt<63>(); break;
    }
#undef PAMG_VM_CASE8
#undef PAMG_VM_CASE
}

// LDS control words (byte offsets)
constexpr unsigned TC_LANDED = 0, TC_READY = 4 /* and 8: one word per gather wave */, TC_DONE = 12, TC_ABORT = 16, TC_BYTES = 64;

// spin until the progress word exceeds v; false = aborted (another wave timed out) or timed out here
__device__ __forceinline__ bool tile_wait(unsigned flag, int v, unsigned *err)
{
    unsigned spins = 0;
    while ((int)lds_flag_load(flag) <= v) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            if (lds_flag_load(TC_ABORT)) return false;
            if (spins > TILE_SPIN_LIMIT) {
                lds_flag_store(TC_ABORT, 1u);
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// byte offsets inside a slot (= a step block + b[64] + xold[64]); the geometry is uniform per launch
struct SlotGeom {
    int val_off, row_off, bv_off, xo_off, st_off, slot_b, block_b, nb;   // st_off: four time stamps (diagnostics)
};

// ---- gatherer: RO / RG = rounds of 64 "old" / "hand-off" items a step may carry
template <typename T, int RO, int RG>
struct GatherSet {
    int eo[RO];               // entry position (-1: none)
    T vo[RO];                 // OLD value
    int eg[RG], jg[RG];
    T vg[RG];                 // hand-off value
    T bv, xo;
    int nrows;
};

// Every step issues exactly RO + RG + 1 (+1) loads, whatever its list sizes (idle rounds re-read a safe element): the
// compiler can then count the memory counter exactly and a finish never waits for the loads of younger steps.  (Loads
// under uniform branches made it fall back to vmcnt(0): the gather pipeline collapsed to one step in flight.)
template <typename T, int RO, int RG, bool XO>
__device__ __forceinline__ void gather_issue(const TileArgs<T> &a, const SlotGeom &sg, GatherSet<T, RO, RG> &S, unsigned slot, int lane)
{
    const tile_v4i hdr = *ldsp<tile_v4i>(slot);
    const int nrows = hdr.x, no = hdr.z & 0xFFFF, ng = (int)((unsigned)hdr.z >> 16);
    S.nrows = nrows;
    const unsigned oitems = slot + 16u + 4u * (((unsigned)hdr.w + 1u) & ~1u), gitems = oitems + 8u * (unsigned)no;
    const int rid = ldsp<int>(slot + (unsigned)sg.row_off)[4 * (lane < nrows ? lane : 0) + 2];
    const int safe = (int)(((unsigned)blockIdx.x * 64u + (unsigned)lane) % (unsigned)a.nidle);
    tile_v2i io[RO], ig[RG];
#pragma unroll
    for (int r = 0; r < RO; ++r) {
        const int i = r * 64 + lane;
        io[r] = ldsp<tile_v2i>(oitems)[i < no ? i : 0];
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int i = r * 64 + lane;
        ig[r] = ldsp<tile_v2i>(gitems)[i < ng ? i : 0];
    }
#pragma unroll
    for (int r = 0; r < RO; ++r) {
        const bool on = r * 64 + lane < no;
        S.eo[r] = on ? io[r].x : -1;
        S.vo[r] = a.x[on ? io[r].y : safe];
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const bool on = r * 64 + lane < ng;
        S.eg[r] = on ? ig[r].x : -1;
        S.jg[r] = on ? ig[r].y : safe;
        S.vg[r] = __hip_atomic_load(a.xs + S.jg[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int row = rid & COL_MASK;
    S.bv = a.b[row];
    if constexpr (XO) S.xo = a.x[row];
}

template <typename T, int RO, int RG, bool XO>
__device__ __forceinline__ bool gather_finish(const TileArgs<T> &a, const SlotGeom &sg, GatherSet<T, RO, RG> &S, unsigned slot, int lane)
{
    const unsigned vals = slot + (unsigned)sg.val_off;
#pragma unroll
    for (int g = 0; g < RO; g += 4) {
        if (__builtin_amdgcn_ballot_w64(S.eo[g] >= 0)) {     // uniform: the group carries items (LDS work only below)
            T av[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = ldsp<T>(vals)[S.eo[g + r] >= 0 ? S.eo[g + r] : 0];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T p = av[r] * S.vo[g + r];
                if (S.eo[g + r] >= 0) ldsp<T>(vals)[S.eo[g + r]] = p;
            }
        }
    }
    unsigned pend = 0;
#pragma unroll
    for (int r = 0; r < RG; ++r)
        if (S.eg[r] >= 0 && Sentinel<T>::bits(S.vg[r]) == Sentinel<T>::value) pend |= 1u << r;
    if (__builtin_amdgcn_ballot_w64(pend != 0)) {
        // not published yet: poll (the tile settles behind its producers, then this is rare)
        unsigned spins = 0;
        while (__builtin_amdgcn_ballot_w64(pend != 0)) {
            __builtin_amdgcn_s_sleep(1);
            T t[RG];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const T *src = ((pend >> r) & 1u) ? (const T *)a.xs + S.jg[r] : (const T *)a.xs + (int)((unsigned)(blockIdx.x * 64 + lane) % (unsigned)a.nidle);
                t[r] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                if (((pend >> r) & 1u) && Sentinel<T>::bits(t[r]) != Sentinel<T>::value) {
                    S.vg[r] = t[r];
                    pend &= ~(1u << r);
                }
            }
            if ((++spins & 63u) == 0) {
                if (lds_flag_load(TC_ABORT)) return false;
                if (spins > TILE_SPIN_LIMIT) {
                    lds_flag_store(TC_ABORT, 1u);
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < RG; g += 2) {
        if (__builtin_amdgcn_ballot_w64(S.eg[g] >= 0)) {
            T av[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) av[r] = ldsp<T>(vals)[S.eg[g + r] >= 0 ? S.eg[g + r] : 0];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const T p = av[r] * S.vg[g + r];
                if (S.eg[g + r] >= 0) ldsp<T>(vals)[S.eg[g + r]] = p;
            }
        }
    }
    if (lane < S.nrows) {
        ldsp<T>(slot + (unsigned)sg.bv_off)[lane] = S.bv;
        if constexpr (XO) ldsp<T>(slot + (unsigned)sg.xo_off)[lane] = S.xo;
    }
    return true;
}

// ---- compute wave
template <typename T>
struct StepRegs {
    tile_v4i hdr, rec;        // step header {rows, first row (tile-local), old | hand-off << 16, local}; this lane's row record
    T bv, xo;
    unsigned item[2];         // this lane's first two "local" items
};

struct ComputeState {
    unsigned slot;            // LDS address of the current step's slot
    int sl;                   // its index in the ring of slots
    int kb;                   // position of the current step inside its batch (batches alternate between the gather waves)
    unsigned rflag;           // ready word of the gather wave that serves the current step
    bool have;                // the operand registers of the current step were filled ahead of time
};

// everything of a step that depends on nothing computed by this tile (LDS -> registers)
template <typename T, bool XO>
__device__ __forceinline__ void step_operands(const SlotGeom &sg, StepRegs<T> &R, unsigned slot, int lane)
{
    R.hdr = *ldsp<tile_v4i>(slot);
    R.rec = ldsp<tile_v4i>(slot + (unsigned)sg.row_off)[lane];
    R.bv = ldsp<T>(slot + (unsigned)sg.bv_off)[lane];
    if constexpr (XO) R.xo = ldsp<T>(slot + (unsigned)sg.xo_off)[lane];
    R.item[0] = ldsp<unsigned>(slot + 16u)[lane];           // the local list follows the header: no dependent address
    R.item[1] = ldsp<unsigned>(slot + 16u)[64 + lane];
}

// One step of the compute wave.  The dependent chain is: ring read -> multiply -> product write -> in-order row
// sum -> divide -> ring write; the operands of the NEXT step (N) are requested between the row sum and the divide,
// speculatively: they are valid iff the ready word read just before them already covered that step.
template <typename T, int EPI, int KG, bool XO>
__device__ __forceinline__ bool compute_step(const TileArgs<T> &a, const SlotGeom &sg, StepRegs<T> &R, StepRegs<T> &N, ComputeState &st,
                                             int t, int ns, int s0, int tile, int lane, unsigned ring, unsigned slot0)
{
    const unsigned wmask = (unsigned)a.W - 1u;
    const unsigned slot = st.slot;
    if (!st.have) {
        if (!tile_wait(st.rflag, t, a.err)) return false;
        step_operands<T, XO>(sg, R, slot, lane);
    }
    long long tp0 = 0;
    if (a.prof) tp0 = wall_clock64();
    // the next step's gather wave (the other one at a batch boundary): its ready word rides along with the stage reads
    const int nkb = st.kb + 1 == KG ? 0 : st.kb + 1;
    const unsigned nflag = nkb == 0 ? (TC_READY + TC_READY + 4u) - st.rflag : st.rflag;
    const int rdy = (int)lds_flag_load(nflag);
    const int nrows = R.hdr.x, rbase = R.hdr.y, nloc = R.hdr.w;
    const unsigned vals = slot + (unsigned)sg.val_off;
    if (nloc > 0) {
        // products with values this tile produced itself (the ring): a dense list, 64 items per round; the reads of
        // both prefetched rounds go out together (idle lanes re-read entry 0 / slot 0 and write nothing)
        const bool on0 = lane < nloc, on1 = 64 + lane < nloc;
        const unsigned it0 = on0 ? R.item[0] : 0u, it1 = on1 ? R.item[1] : 0u;
        PAMG_LDS T *v0 = ldsp<T>(vals) + (it0 & 0xFFFFu);
        PAMG_LDS T *v1 = ldsp<T>(vals) + (it1 & 0xFFFFu);
        const T a0 = *v0, x0 = ldsp<T>(ring)[it0 >> 16];
        if (nloc > 64) {
            const T a1 = *v1, x1 = ldsp<T>(ring)[it1 >> 16];
            const T p0 = a0 * x0, p1 = a1 * x1;
            if (on0) *v0 = p0;
            if (on1) *v1 = p1;
        } else {
            const T p0 = a0 * x0;
            if (on0) *v0 = p0;
        }
        if (nloc > 128) {
            for (int i = 128 + lane; i < nloc; i += 64) {
                const unsigned it = ldsp<unsigned>(slot + 16u)[i];
                PAMG_LDS T *v = ldsp<T>(vals) + (it & 0xFFFFu);
                *v = *v * ldsp<T>(ring)[it >> 16];
            }
        }
    }
    compiler_fence();                                       // one wave: LDS operations execute in order
    T d, s = T(0);
    if constexpr (sizeof(T) == 8) d = __longlong_as_double(((long long)(unsigned)R.rec.y << 32) | (unsigned)R.rec.x);
    else d = __int_as_float(R.rec.x);
    const int rid = R.rec.z;
    if (lane < nrows) {
        const int lo = R.rec.w & 0xFFFF, len = (int)((unsigned)R.rec.w >> 16);
        s = EpiTraits<EPI>::bsr_order ? R.bv : T(0);
        const T *prod = (const T *)ldsp<T>(vals);           // generic pointer into LDS for row_accumulate
        row_accumulate<T, EPI>(s, prod, nullptr, lo, lo + len, 0);
    }
    compiler_fence();
    // operands of the next step, requested while this one divides
    unsigned nslot = slot + (unsigned)sg.slot_b;
    int nsl = st.sl + 1;
    if (nsl == a.D) { nsl = 0; nslot = slot0; }
    if (t + 1 < ns && rdy > t + 1) step_operands<T, XO>(sg, N, nslot, lane);
    compiler_fence();
    if (lane < nrows) {
        T v;
        if constexpr (EPI == EPI_GS) v = (R.bv - s) / d;
        else if constexpr (EPI == EPI_GS_B) v = s / d;
        else v = a.omega * ((R.bv - s) / d) + (T(1) - a.omega) * R.xo;
        if constexpr (XO) { if (!(d != T(0))) v = R.xo; }   // no / zero diagonal: the row keeps its value (relaxation.h:72)
        ldsp<T>(ring)[(unsigned)(rbase + lane) & wmask] = v;
        const int row = rid & COL_MASK;
        a.y[row] = v;
        if (rid < 0) __hip_atomic_store(a.xs + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    compiler_fence();
    long long stamps[4] = {0, 0, 0, 0};
    if (a.prof && lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) stamps[k] = ldsp<long long>(slot + (unsigned)sg.st_off)[k];
    }
    compiler_fence();
    lds_flag_store(TC_DONE, (unsigned)(t + 1));             // every read of this slot has been consumed: the loader may refill it
    if (a.prof && lane == 0) {
        long long *o = a.prof + (size_t)(s0 + t) * 8;
        o[0] = tp0; o[1] = wall_clock64(); o[2] = (long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF); o[3] = tile;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[4 + k] = stamps[k];
    }
    st.kb = nkb;
    st.rflag = nflag;
    st.have = (t + 1 < ns) && (rdy > t + 1);
    st.slot = nslot;
    st.sl = nsl;
    return true;
}

// ---- the kernel
// RO / RG: rounds of gather items per step, KG = steps the gatherer keeps in registers; XO: the rows' own old values are
// needed (SOR, or an operator with a missing / zero diagonal).  The <4, 2, 4> variants run several workgroups per CU
// (wide levels: bandwidth matters, registers capped for three waves per SIMD).
template <typename T, int EPI, int RO, int RG, int KG, bool XO>
__global__ __launch_bounds__(TILE_THREADS, (RO <= 4 ? 2 : 1)) void gs_tile_kernel(const TileArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tile = (int)blockIdx.x;
    if (tile >= a.G) return;
    const int s0 = a.tile_step[tile];
    const int ns = a.tile_step[tile + 1] - s0;
    if (ns <= 0) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int D = a.D;
    SlotGeom sg;
    sg.val_off = 1024 * a.NCH; sg.row_off = 1024 * (a.NCH + a.NV); sg.block_b = sg.row_off + 1024; sg.nb = a.NCH + a.NV + 1;
    sg.bv_off = sg.block_b; sg.xo_off = sg.block_b + TILE_ROWS * (int)sizeof(T); sg.st_off = sg.block_b + 2 * TILE_ROWS * (int)sizeof(T);
    sg.slot_b = sg.st_off + 32;
    const unsigned ring = TC_BYTES;
    const unsigned slot0 = TC_BYTES + (unsigned)a.W * (unsigned)sizeof(T);
    if (threadIdx.x < 5) lds_flag_store(4u * threadIdx.x, 0u);
    __syncthreads();

    if (wave == 1) {
        // ---- loader
        const unsigned char *g = a.blocks + (size_t)s0 * (size_t)sg.block_b + (size_t)lane * 16;
        const int Q = a.Q, nb = sg.nb;
        int sl = 0;
        for (int t = 0; t < ns; ++t) {
            if (t >= D && !tile_wait(TC_DONE, t - D, a.err)) return;
            const unsigned dst = slot0 + (unsigned)sl * (unsigned)sg.slot_b;
            if (a.prof && lane == 0) ldsp<long long>(dst + (unsigned)sg.st_off)[0] = wall_clock64();
            for (int c = 0; c < nb; ++c) glds16(g + c * 1024, dst + (unsigned)c * 1024u);
            g += sg.block_b;
            if (++sl == D) sl = 0;
            wait_vmcnt_rt(Q * nb);
            if (t + 1 - Q > 0) lds_flag_store(TC_LANDED, (unsigned)(t + 1 - Q));
        }
        wait_vmcnt<0>();
        lds_flag_store(TC_LANDED, (unsigned)ns);
        return;
    }

    if (wave >= 2) {
        // ---- two gatherers, alternating BATCHES of KG steps: all loads of a batch are issued, then its steps are
        // finished in order, and nothing stays in flight across the loop's back edge.  (A rotating pipeline with loads in
        // flight across iterations needs exact memory-counter waits; hipcc falls back to near-complete drains there --
        // seen in the ISA -- so the overlap comes from the second wave instead: while one waits for its batch, the
        // other issues or finishes.)
        const int w = wave - 2;
        const unsigned rflag = TC_READY + 4u * (unsigned)w;
        GatherSet<T, RO, RG> S[KG];
        int sl = (w * KG) % D;                               // slot of the batch's first step
        for (int t0 = w * KG; t0 < ns; t0 += 2 * KG) {
            const bool full = t0 + KG <= ns;
            if (full) {
                int s1 = sl;
#pragma unroll
                for (int k = 0; k < KG; ++k) {
                    if (!tile_wait(TC_LANDED, t0 + k, a.err)) return;
                    const unsigned sa = slot0 + (unsigned)s1 * (unsigned)sg.slot_b;
                    if (a.prof && lane == 0) ldsp<long long>(sa + (unsigned)sg.st_off)[1] = wall_clock64();
                    gather_issue<T, RO, RG, XO>(a, sg, S[k], sa, lane);
                    if (++s1 == D) s1 = 0;
                }
                s1 = sl;
#pragma unroll
                for (int k = 0; k < KG; ++k) {
                    const unsigned sa = slot0 + (unsigned)s1 * (unsigned)sg.slot_b;
                    if (a.prof && lane == 0) ldsp<long long>(sa + (unsigned)sg.st_off)[2] = wall_clock64();
                    if (!gather_finish<T, RO, RG, XO>(a, sg, S[k], sa, lane)) return;
                    if (a.prof && lane == 0) ldsp<long long>(sa + (unsigned)sg.st_off)[3] = wall_clock64();
                    lds_drain();
                    lds_flag_store(rflag, (unsigned)(t0 + k + 1));
                    if (++s1 == D) s1 = 0;
                }
            } else {
                // the tile's last, partial batch: one step at a time
                int s1 = sl;
                for (int t = t0; t < ns; ++t) {
                    if (!tile_wait(TC_LANDED, t, a.err)) return;
                    const unsigned sa = slot0 + (unsigned)s1 * (unsigned)sg.slot_b;
                    if (a.prof && lane == 0) ldsp<long long>(sa + (unsigned)sg.st_off)[1] = wall_clock64();
                    gather_issue<T, RO, RG, XO>(a, sg, S[0], sa, lane);
                    if (a.prof && lane == 0) ldsp<long long>(sa + (unsigned)sg.st_off)[2] = wall_clock64();
                    if (!gather_finish<T, RO, RG, XO>(a, sg, S[0], sa, lane)) return;
                    if (a.prof && lane == 0) ldsp<long long>(sa + (unsigned)sg.st_off)[3] = wall_clock64();
                    lds_drain();
                    lds_flag_store(rflag, (unsigned)(t + 1));
                    if (++s1 == D) s1 = 0;
                }
            }
            sl += (2 * KG) % D;
            if (sl >= D) sl -= D;
        }
        return;
    }

    // ---- compute wave (two steps per trip: the operand registers of step t + 1 are filled while step t divides)
    StepRegs<T> R0, R1;
    ComputeState st;
    st.slot = slot0; st.sl = 0; st.kb = 0; st.rflag = TC_READY; st.have = false;
    for (int t = 0; t < ns; t += 2) {
        if (!compute_step<T, EPI, KG, XO>(a, sg, R0, R1, st, t, ns, s0, tile, lane, ring, slot0)) return;
        if (t + 1 >= ns) break;
        if (!compute_step<T, EPI, KG, XO>(a, sg, R1, R0, st, t + 1, ns, s0, tile, lane, ring, slot0)) return;
    }
}

}  // namespace pamg
