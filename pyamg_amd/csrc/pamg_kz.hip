// pamg_kz.hip -- the LANE-PARALLEL ("fast order") Kaczmarz-type sweeps: layout and protocol in pamg_kz_plan.h.
//
// Same lines in the same order as amg_core::gauss_seidel_ne (relaxation.h:875-904) and gauss_seidel_nr (:939-975) -- what
// pyamg.solve() configures for non-symmetric operators (blackbox.py:112-114) -- but ONE persistent launch per directional sweep
// instead of one launch per dependency level: L lanes of a wave share a line, every lane holds K of its entries, a butterfly adds
// the lanes; every index of the vector the sweep rewrites (x resp. the running residual) has a 16-byte slot {value, version} in a
// side buffer, a line polls the slots of its entries until each shows the version the plan expects (the number of earlier lines of
// the sweep that hold the index), and writes {new value, version + 1} back with one 16-byte store per entry.  Waves take the groups
// w, w + W, ... (all co-resident; a line only waits for lines of lower dependency levels = smaller group numbers: deadlock-free).
// The slots are filled from v before the sweep and copied back after it (two streaming launches).
//
// The 16-byte slot accesses are written in assembly: the compiler offers no agent-coherent 16-byte load (a volatile access
// compiles to flat_load sc0 sc1 with a full drain behind every single one; __hip_atomic_load stops at 8 bytes), and a slot must be
// read and written in ONE access -- value and version travel together, that is the whole protocol.  All K loads of a poll round and
// the wait for them are one asm statement, so the compiler never sees a register whose load is still in flight.
// HARDWARE ASSUMPTION: the AMDGPU memory model promises single-copy atomicity up to 8 bytes only; the protocol relies on what the part
// does with a naturally aligned 16-byte global_load/store_dwordx4: the slots are 16-byte aligned, so an access never straddles an L2 line (128 bytes)
// or channel, and the L2 serves it as ONE request -- reader and writer meet in one L2 bank, a reader sees the slot entirely before or entirely after a
// store.  A torn access would pair a new version with an old value: a wrong iterate, never a hang -- which is what the parity runs of every
// normal-equation hierarchy (10 cycles each, tests and bench legs, ~10^9 slot hand-offs per run) would show and never have.
#include "pamg_common.h"
#include "pamg_kz_plan.h"

namespace pamg {

typedef int kz_i4 __attribute__((ext_vector_type(4)));

struct KzLaneSched {
    int L = 0, K = 0, RPW = 0;
    int64_t ngroups = 0, nslots = 0, max_level_groups = 0;
    int nlevels = 0;
    int *d_idx = nullptr, *d_ver = nullptr, *d_line = nullptr;
    double *d_vals = nullptr;
    kz_i4 *d_slots = nullptr;          // [ncols] {value lo, value hi, version, 0}
    unsigned *d_err = nullptr;
    int last_grid = 0;
    int occ_cap = 0;                   // workgroups per CU the sweep may count on (occupancy query, asked once)
    size_t bytes = 0;
};

struct KzArgs {
    const int *idx, *ver, *line;
    const double *vals;
    kz_i4 *slots;
    const double *b, *Dinv;
    double *xout;
    unsigned *err;
    double omega;
    int ngroups, nidle;
};

__global__ __launch_bounds__(BLK) void kz_pack_kernel(const double *__restrict__ v, kz_i4 *__restrict__ slots, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (j >= n) return;
    const double x = v[j];
    kz_i4 q;
    q.x = __double2loint(x); q.y = __double2hiint(x); q.z = 0; q.w = 0;
    slots[j] = q;
}

__global__ __launch_bounds__(BLK) void kz_unpack_kernel(const kz_i4 *__restrict__ slots, double *__restrict__ v, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (j >= n) return;
    const kz_i4 q = slots[j];
    v[j] = __hiloint2double(q.y, q.x);
}

template <int K>
struct KzSet {
    int j[K], ver[K];
    double a[K];
    int line;
};

template <int L, int K>
__device__ __forceinline__ void kz_load(const KzArgs &a, int g, KzSet<K> &S)
{
    const int lane = threadIdx.x & 63;
    const size_t e0 = (size_t)g * (size_t)(K * 64) + (size_t)lane;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        S.j[k] = a.idx[e0 + (size_t)k * 64];
        S.ver[k] = a.ver[e0 + (size_t)k * 64];
        S.a[k] = a.vals[e0 + (size_t)k * 64];
    }
    S.line = a.line[(size_t)g * (size_t)(64 / L) + (size_t)(lane / L)];
}

// K agent-coherent 16-byte loads and the wait for them: one statement (header comment)
template <int K>
__device__ __forceinline__ void kz_poll(kz_i4 (&q)[K], const kz_i4 *const (&p)[K])
{
    if constexpr (K == 1)
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(q[0]) : "v"(p[0]) : "memory");
    else if constexpr (K == 2)
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1]) : "v"(p[0]), "v"(p[1]) : "memory");
    else if constexpr (K == 3)
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]) : "v"(p[0]), "v"(p[1]), "v"(p[2]) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
}

__device__ __forceinline__ void kz_store(kz_i4 *p, kz_i4 q)
{
    // s_nop: the store is invisible to the compiler's hazard recogniser -- a VMEM store of more than 64 bits needs one wait state before a
    // VALU instruction may overwrite its data registers
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(p), "v"(q) : "memory");
}

template <int CTRL>
__device__ __forceinline__ double kz_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);       // (no `old` operand: every lane has a source, the compiler needs no copy)
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// lane ^ 16 / lane ^ 32 by v_permlane16_swap / v_permlane32_swap (pamg_lane.hip: swap_sum): no LDS path
template <bool HALF>
__device__ __forceinline__ double kz_swap_sum(double v)
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
// the butterfly of the Gauss-Seidel lane form (pamg_lane.hip: seg_allreduce), same steps in the same order
template <int L>
__device__ __forceinline__ double kz_allreduce(double v)
{
    if constexpr (L >= 2) v = v + kz_dpp<0xB1>(v);
    if constexpr (L >= 4) v = v + kz_dpp<0x4E>(v);
    if constexpr (L >= 8) v = v + kz_dpp<0x141>(v);
    if constexpr (L >= 16) v = v + kz_dpp<0x140>(v);
    if constexpr (L >= 32) v = kz_swap_sum<false>(v);
    if constexpr (L >= 64) v = kz_swap_sum<true>(v);
    return v;
}

template <bool NR, int L, int K>
__device__ __forceinline__ void kz_group(const KzArgs &a, const KzSet<K> &S, int idle)
{
    const int lane = threadIdx.x & 63;
    const bool head = (lane & (L - 1)) == 0;
    const int i = S.line;
    const int ic = i < 0 ? 0 : i;
    const double dinv = a.Dinv[ic];
    double bi = 0.0;
    if constexpr (!NR) bi = a.b[ic];
    const kz_i4 *p[K];
    kz_i4 q[K];
    unsigned pend = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool real = !(S.j[k] & KZL_NONE);
        p[k] = a.slots + (real ? (S.j[k] & KZL_MASK) : idle);
        if (real) pend |= 1u << k;
    }
    unsigned spins = 0;
    while (true) {
        kz_poll<K>(q, p);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (((pend >> k) & 1u) && q[k].z == S.ver[k]) pend &= ~(1u << k);
        if (!__builtin_amdgcn_ballot_w64(pend != 0)) break;
        if ((++spins & 1023u) == 0) {
            // a producer that never comes (not every wave resident / an earlier time-out): give up together, quickly
            if (spins > (1u << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    double s = 0.0, xv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        xv[k] = __hiloint2double(q[k].y, q[k].x);
        const double pr = S.a[k] * xv[k];
        s = s + ((S.j[k] & KZL_NONE) ? 0.0 : pr);
    }
    s = kz_allreduce<L>(s);
    double d;
    if constexpr (NR) d = s * (dinv * a.omega);                       // relaxation.h:963-966
    else d = (bi - s) * dinv * a.omega;                               // relaxation.h:897
    if (i >= 0) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (!(S.j[k] & KZL_NONE)) {
                double nv;
                if constexpr (NR) nv = xv[k] - d * S.a[k];             // relaxation.h:972
                else { const double t = S.a[k] * d; nv = xv[k] + t; }  // relaxation.h:900
                kz_i4 w;
                w.x = __double2loint(nv); w.y = __double2hiint(nv); w.z = S.ver[k] + 1; w.w = 0;
                kz_store(a.slots + (S.j[k] & KZL_MASK), w);
            }
        if constexpr (NR) {
            if (head) a.xout[i] = a.xout[i] + d;                      // relaxation.h:969: x_i belongs to line i alone
        }
    }
}

constexpr int KZ_WPB = BLK / 64;

template <bool NR, int L, int K>
__global__ __launch_bounds__(BLK) void kz_lane_kernel(const KzArgs a)
{
    const int wib = threadIdx.x >> 6;
    const int W = (int)gridDim.x * KZ_WPB;
    int g = (int)blockIdx.x * KZ_WPB + wib;
    if (g >= a.ngroups) return;
    const int idle = (int)((((unsigned)blockIdx.x * KZ_WPB + (unsigned)wib) * 4u) % (unsigned)a.nidle);
    // a group's operands are requested when the wave arrives at it (until round 5's last day the NEXT group's were requested before the wave waited
    // for the current one: the in-order memory counter put every poll behind that prefetch; c6n3 15.85 -> 15.60 ms, profiles/r05_bench_c6n3_prefetch_ab.txt)
    KzSet<K> P;
    for (; g < a.ngroups; g += W) {
        kz_load<L, K>(a, g, P);
        kz_group<NR, L, K>(a, P, idle);
    }
}

// ------------------------------------------------------------------ host side
namespace {

template <typename U>
int kz_upload(U **dst, const void *src, size_t bytes, size_t *total)
{
    *dst = nullptr;
    const size_t alloc = std::max<size_t>(bytes, 256) + 256;
    PAMG_HIP(hipMalloc((void **)dst, alloc));
    if (bytes && src) PAMG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    if (total) *total += alloc;
    return PAMG_OK;
}

template <bool NR, int L>
const void *kz_kernel_k(int K)
{
    switch (K) {
        case 1: return (const void *)kz_lane_kernel<NR, L, 1>;
        case 2: return (const void *)kz_lane_kernel<NR, L, 2>;
        case 3: return (const void *)kz_lane_kernel<NR, L, 3>;
        case 4: return (const void *)kz_lane_kernel<NR, L, 4>;
    }
    return nullptr;
}

template <bool NR>
const void *kz_kernel(int L, int K)
{
    switch (L) {
        case 4: return kz_kernel_k<NR, 4>(K);
        case 8: return kz_kernel_k<NR, 8>(K);
        case 16: return kz_kernel_k<NR, 16>(K);
        case 32: return kz_kernel_k<NR, 32>(K);
        case 64: return kz_kernel_k<NR, 64>(K);
    }
    return nullptr;
}

}  // namespace

void free_kz_lane_part(KzLaneSched *t)
{
    if (!t) return;
    hipFree(t->d_idx); hipFree(t->d_ver); hipFree(t->d_line); hipFree(t->d_vals); hipFree(t->d_slots); hipFree(t->d_err);
    delete t;
}

// built with the line schedule (never inside a graph capture); PAMG_E_ARG: the form does not apply, the per-level kernels keep the sweep
int build_kz_lane_part(pamg_matrix_s *Lm, LineSchedule *g)
{
    if (g->kzl || g->kzl_unfit) return PAMG_OK;
    if (Lm->dtype != PAMG_F64 || Lm->R != 1 || Lm->C != 1) { g->kzl_unfit = true; return PAMG_OK; }
    PhaseTimer pt_("build_kz_lane_part", Lm->nnz);
    std::vector<unsigned char> hLx((size_t)Lm->nnz * 8);
    if (Lm->nnz) PAMG_HIP(hipMemcpy(hLx.data(), Lm->d_Ax, (size_t)Lm->nnz * 8, hipMemcpyDeviceToHost));
    KzLanePlan P;
    if (build_kz_lane_plan((int)Lm->nrows, (int)Lm->ncols, Lm->h_Ap.data(), Lm->h_Aj.data(), hLx.data(), 8, g->start, g->stop, g->step, P)) {
        g->kzl_unfit = true;
        return PAMG_OK;
    }
    KzLaneSched *t = new (std::nothrow) KzLaneSched();
    if (!t) return PAMG_E_ALLOC;
    t->L = P.L; t->K = P.K; t->RPW = P.RPW; t->ngroups = P.ngroups; t->nslots = P.nslots; t->nlevels = P.nlevels; t->max_level_groups = P.max_level_groups;
    int st = kz_upload(&t->d_idx, P.idx.data(), P.idx.size() * sizeof(int), &t->bytes);
    if (!st) st = kz_upload(&t->d_ver, P.ver.data(), P.ver.size() * sizeof(int), &t->bytes);
    if (!st) st = kz_upload(&t->d_line, P.line.data(), P.line.size() * sizeof(int), &t->bytes);
    if (!st) st = kz_upload(&t->d_vals, P.vals.data(), P.vals.size(), &t->bytes);
    if (!st) st = kz_upload(&t->d_slots, nullptr, (size_t)(Lm->ncols + 16) * sizeof(kz_i4), &t->bytes);
    if (!st) st = kz_upload(&t->d_err, nullptr, 64, &t->bytes);
    if (!st) st = (int)hipMemset(t->d_err, 0, 64);
    if (st) { free_kz_lane_part(t); return st; }
    g->kzl = t;
    g->bytes += t->bytes;
    Lm->bytes += t->bytes;
    return PAMG_OK;
}

static int kz_cus()
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 64;
    return p.multiProcessorCount;
}

int kz_lane_launch(pamg_matrix_s *Lm, LineSchedule *g, bool nr, void *v, const void *b, const void *Dinv, double omega, void *xout, hipStream_t s)
{
    KzLaneSched *t = g->kzl;
    if (!t) return PAMG_E_STATE;
    const int64_t n = Lm->ncols;
    KzArgs a;
    a.idx = t->d_idx; a.ver = t->d_ver; a.line = t->d_line; a.vals = t->d_vals; a.slots = t->d_slots;
    a.b = (const double *)b; a.Dinv = (const double *)Dinv; a.xout = (double *)xout; a.err = t->d_err; a.omega = omega;
    a.ngroups = (int)t->ngroups;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 16));
    const unsigned vg = (unsigned)((n + BLK - 1) / BLK);
    if (n > 0) hipLaunchKernelGGL(kz_pack_kernel, dim3(vg), dim3(BLK), 0, s, (const double *)v, t->d_slots, n);
    PAMG_HIP(hipGetLastError());
    const void *k = nr ? kz_kernel<true>(t->L, t->K) : kz_kernel<false>(t->L, t->K);
    if (!k) return PAMG_E_ARG;
    static thread_local int cus = 0;
    if (!cus) cus = kz_cus();
    // a few dependency levels of look-ahead; every workgroup must be resident: at most 2 per CU, and never more than the occupancy query
    // allows minus one (asked once per schedule, as the lane / line / block sweeps do)
    if (t->occ_cap == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, BLK, 0) != hipSuccess || nb < 1) nb = 1;
        t->occ_cap = std::max(1, std::min(nb > 1 ? nb - 1 : 1, 2));
    }
    const int percu = t->occ_cap;
    const int per_level = (int)((t->ngroups + t->nlevels - 1) / std::max(1, t->nlevels));
    const int64_t want_waves = std::max<int64_t>(64, (int64_t)4 * per_level);
    int G = (int)std::min<int64_t>((want_waves + KZ_WPB - 1) / KZ_WPB, (int64_t)percu * cus);
    if (Lm->lane_G > 0) G = std::min(Lm->lane_G, percu * cus);
    G = (int)std::max<int64_t>(1, std::min<int64_t>(G, (t->ngroups + KZ_WPB - 1) / KZ_WPB));
    t->last_grid = G;
    void *args[] = {(void *)&a};
    PAMG_HIP(hipLaunchKernel(k, dim3(G), dim3(BLK), args, 0, s));
    if (n > 0) hipLaunchKernelGGL(kz_unpack_kernel, dim3(vg), dim3(BLK), 0, s, (const kz_i4 *)t->d_slots, (double *)v, n);
    return (int)hipGetLastError();
}

// info[0..7] = lanes per line, slots per lane, groups, dependency levels, widest level (groups), workgroups of the last launch, bytes, 0
int kz_lane_info(const LineSchedule *g, int64_t *info)
{
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (!g || !g->kzl) return PAMG_OK;
    const KzLaneSched *t = g->kzl;
    info[0] = t->L; info[1] = t->K; info[2] = t->ngroups; info[3] = t->nlevels; info[4] = t->max_level_groups; info[5] = t->last_grid;
    info[6] = (int64_t)t->bytes;
    return PAMG_OK;
}

bool kz_lane_error(LineSchedule *g)
{
    if (!g || !g->kzl) return false;
    unsigned w = 0;
    if (hipMemcpy(&w, g->kzl->d_err, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return true;
    if (w) hipMemset(g->kzl->d_err, 0, sizeof(unsigned));
    return w != 0;
}

}  // namespace pamg
