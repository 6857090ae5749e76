// pamg_dist.hip -- the row-sharded multigrid cycle of ONE rank (one process per GPU), driven from C++.
//
// What it runs is MultilevelSolver.__solve (reference pyamg/multilevel.py:584-662) and the accel=None loop of
// .solve (:537-582) on a hierarchy whose fine levels are cut into contiguous row blocks, one per rank
// (pyamg_amd/dist.py plans the cut: every operator of a sharded level arrives here as a row shard in local
// numbering [owned | halo]).  Per operator application ONE halo exchange of the input vector:
//
//     main stream :  pack (gather the values peers need)  --ev_pack-->  INTERIOR row ranges  --wait ev_halo-->  BOUNDARY row ranges
//     comm stream :                 wait ev_pack, grouped ncclSend / ncclRecv straight into the halo part of the vector, ev_halo
//
// so the halo traffic over xGMI hides behind the rows that do not need it (matrix_split_ranges: two index lists into
// the operator's existing row-range plan, no data copied).  Below the last sharded level the hierarchy is tiny: its
// right-hand side is assembled on every rank by one all-reduce of disjoint slices and the rest of the cycle runs
// redundantly with the resident single-GPU engine.
//
// Transports: RCCL (production; librccl is bound at run time -- the copy torch already loaded when there is one --
// with a communicator of our own built from an id the host side broadcasts), or two host callbacks (test rigs: several
// ranks on one GPU staged through gloo, where RCCL refuses duplicate devices).  With no peers at all (world = 1) the
// whole iteration is one hipGraph.  Per-row arithmetic is the single-GPU kernels' (entries keep their storage order), so
// the sharded iterates are bit-identical to the single-GPU ones; only the all-reduced norm differs in the last bits.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <new>
#include <string>

#include "pamg_common.h"

using namespace pamg;

namespace {

// ---------------------------------------------------------------------------------- RCCL, bound at run time
struct RcclId { char internal[128]; };
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_F32 = 7, NCCL_F64 = 8, NCCL_SUM = 0;      // rccl.h: ncclFloat32 / ncclFloat64 / ncclSum

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r.h ? &r : nullptr;
    tried = true;
    // the copy the process already holds (torch ships and loads its own librccl) -- two RCCLs in one process is asking for trouble
    std::string loaded;
    {
        std::ifstream maps("/proc/self/maps");
        std::string line;
        while (std::getline(maps, line)) {
            const size_t p = line.find('/');
            if (p != std::string::npos && line.find("librccl.so") != std::string::npos) { loaded = line.substr(p); break; }
        }
    }
    const char *cands[] = {loaded.empty() ? nullptr : loaded.c_str(), getenv("PAMG_RCCL_LIB"), "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char *c : cands) {
        if (!c || !*c) continue;
        r.h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (r.h) break;
    }
    if (!r.h) return nullptr;
#define PAMG_SYM(f, name) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, name)); if (!r.f) { r.h = nullptr; return nullptr; }
    PAMG_SYM(GetUniqueId, "ncclGetUniqueId") PAMG_SYM(CommInitRank, "ncclCommInitRank") PAMG_SYM(CommDestroy, "ncclCommDestroy")
    PAMG_SYM(Send, "ncclSend") PAMG_SYM(Recv, "ncclRecv") PAMG_SYM(AllReduce, "ncclAllReduce") PAMG_SYM(AllGather, "ncclAllGather")
    PAMG_SYM(GroupStart, "ncclGroupStart") PAMG_SYM(GroupEnd, "ncclGroupEnd") PAMG_SYM(GetErrorString, "ncclGetErrorString")
#undef PAMG_SYM
    return &r;
}

#define PAMG_NCCL(expr)                                                                            \
    do {                                                                                           \
        int r__ = (expr);                                                                          \
        if (r__ != 0) {                                                                            \
            fprintf(stderr, "[pamg_dist] %s -> %s\n", #expr, rccl() ? rccl()->GetErrorString(r__) : "?"); \
            return PAMG_E_COMM;                                                                    \
        }                                                                                          \
    } while (0)

struct DSmoother {
    int kind = PAMG_SMOOTH_NONE, iterations = 1, blocksize = 1;
    double omega = 1.0;
    std::vector<double> coeffs;
    void *d_Dinv = nullptr;
};

struct DLevel {
    pamg_matrix_s *A = nullptr, *P = nullptr, *R = nullptr;     // row shards, borrowed; all nullptr on the collapse level
    int64_t n_owned = 0, n_halo = 0;                            // scalars
    std::vector<int> send_peer, recv_peer;
    std::vector<int64_t> send_off, recv_off;                    // [n+1] scalar offsets into send_buf / the halo
    int *d_send_idx = nullptr;
    void *send_buf = nullptr;
    // all-gather form of the exchange ("keep the full all-gather as the general fallback and the correctness baseline"):
    // every rank contributes its owned part padded to ag_count values, the halo is then picked out of the gathered vector
    int64_t ag_count = 0;             // values per rank in the gathered vector (max owned over the ranks); 0 = not set up
    int *d_halo_src = nullptr;        // [n_halo] position of every halo value in the gathered vector
    void *ag_stage = nullptr, *ag_all = nullptr;   // [ag_count], [world * ag_count]
    void *x = nullptr, *xalt = nullptr, *x_home = nullptr, *b = nullptr, *r = nullptr, *h0 = nullptr, *h1 = nullptr;
    DSmoother pre, post;
    int64_t n_ex = 0;                 // exchanges of this level's vectors per iteration, counted while enqueueing (diagnostics / the scaling model)
    int64_t n_local() const { return n_owned + n_halo; }
    bool talks() const { return !send_peer.empty() || !recv_peer.empty(); }
};

}  // namespace

struct pamg_dist_s {
    int dtype = PAMG_F64, rank = 0, world = 1;
    std::vector<DLevel> lv;           // sharded levels, then the collapse level (vectors and halo layout only)
    bool collapse_set = false, finalized = false;
    // collapse level: full vectors of the first replicated level
    int64_t nc = 0, c_row0 = 0;
    int *d_fill = nullptr;            // [n_local of the collapse level] global index of every local entry
    void *bc_full = nullptr, *xc_full = nullptr;
    pamg_solver_s *coarse = nullptr;  // borrowed
    hipStream_t main = nullptr, comm = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    int mode = 0;                     // 0 none (no peers allowed), 1 host callbacks, 2 RCCL, 3 MODEL: the rank's kernels with nobody on the wire (pamg_dist_set_model_transport)
    pamg_dist_exchange_fn cb_exchange = nullptr;
    pamg_dist_allreduce_fn cb_allreduce = nullptr;
    void *cb_user = nullptr;
    void *nccl_comm = nullptr;
    int xmode = 0;                    // halo exchange: 0 = point to point with the actual neighbours (default), 1 = all-gather of the owned parts
    bool overlap = true;              // interior rows while the halo is in flight
    bool use_graph = true;
    bool capturing = false;
    hipGraphExec_t graph[2] = {nullptr, nullptr};   // [with norm]
    double *d_ss = nullptr;           // [0] ||r||^2 (all-reduced), [1] scratch
    double *d_norms = nullptr;
    int norms_cap = 0;
    int64_t n_exchanges = 0, n_overlapped = 0;      // per cycle, counted while enqueueing (diagnostics)
    size_t bytes = 0;
};

namespace {

size_t ts_of(const pamg_dist_s *D) { return tsize(D->dtype); }

int dmalloc(pamg_dist_s *D, void **p, size_t bytes)
{
    bytes = std::max<size_t>(bytes, 256);
    PAMG_HIP(hipMalloc(p, bytes));
    PAMG_HIP(hipMemset(*p, 0, bytes));
    D->bytes += bytes;
    return PAMG_OK;
}

bool graph_ok(const pamg_dist_s *D)
{
    if (!D->use_graph) return false;
    if (D->mode == 1) return false;                          // host callbacks cannot be captured
    if (solver_needs_host_sync(D->coarse)) return false;     // the collapsed tail synchronises with the host (Krylov smoother / coarse solver)
    bool talks = false;
    for (const DLevel &L : D->lv) talks = talks || L.talks();
    if (D->mode == 2 && (talks || D->world > 1)) {
        const char *e = getenv("PAMG_DIST_GRAPH");           // RCCL work inside a hipGraph: opt-in until it has run on a multi-GPU node
        return e && *e == '1';
    }
    return true;
}

// all-reduce (sum) of `count` values at device address buf, in place, ordered after everything on the main stream
int all_reduce(pamg_dist_s *D, void *buf, int64_t count, int dtype)
{
    if (D->world == 1) return PAMG_OK;
    if (D->mode == 2) {
        PAMG_NCCL(rccl()->AllReduce(buf, buf, (size_t)count, dtype == PAMG_F64 ? NCCL_F64 : NCCL_F32, NCCL_SUM, D->nccl_comm, D->main));
        return PAMG_OK;
    }
    if (D->mode == 1) {
        PAMG_HIP(hipStreamSynchronize(D->main));
        return D->cb_allreduce(D->cb_user, buf, count, dtype);
    }
    if (D->mode == 3) return PAMG_OK;                        // model transport: this rank's contribution alone
    return PAMG_E_STATE;
}

// Start the halo exchange of level-l vector v (after everything queued on the main stream).  RCCL: the transfers are
// queued on the comm stream behind ev_pack and ev_halo is recorded behind them.  Callbacks: only the pack is queued;
// finish_exchange() does the blocking part.
int begin_exchange(pamg_dist_s *D, int l, void *v)
{
    DLevel &L = D->lv[l];
    const size_t ts = ts_of(D);
    if (D->xmode == 1) {
        // all-gather form: owned part -> staging (padded), gathered over the ranks on the comm stream, halo picked out of it
        if (!L.ag_count) return PAMG_E_STATE;
        if (L.n_owned) PAMG_HIP(hipMemcpyAsync(L.ag_stage, v, (size_t)L.n_owned * ts, hipMemcpyDeviceToDevice, D->main));
        PAMG_HIP(hipEventRecord(D->ev_pack, D->main));
        if (D->mode == 2) {
            Rccl *R = rccl();
            PAMG_HIP(hipStreamWaitEvent(D->comm, D->ev_pack, 0));
            PAMG_NCCL(R->AllGather(L.ag_stage, L.ag_all, (size_t)L.ag_count, D->dtype == PAMG_F64 ? NCCL_F64 : NCCL_F32, D->nccl_comm, D->comm));
            if (L.n_halo) PAMG_TRY(pamg_vec_gather(D->dtype, L.n_halo, L.d_halo_src, L.ag_all, (char *)v + (size_t)L.n_owned * ts, D->comm));
            PAMG_HIP(hipEventRecord(D->ev_halo, D->comm));
        }
        return PAMG_OK;
    }
    const int64_t ns = L.send_off.empty() ? 0 : L.send_off.back();
    if (ns) PAMG_TRY(pamg_vec_gather(D->dtype, ns, L.d_send_idx, v, L.send_buf, D->main));
    PAMG_HIP(hipEventRecord(D->ev_pack, D->main));
    if (D->mode == 2) {
        Rccl *R = rccl();
        const int dt = D->dtype == PAMG_F64 ? NCCL_F64 : NCCL_F32;
        PAMG_HIP(hipStreamWaitEvent(D->comm, D->ev_pack, 0));
        PAMG_NCCL(R->GroupStart());
        for (size_t k = 0; k < L.recv_peer.size(); ++k)
            PAMG_NCCL(R->Recv((char *)v + (size_t)(L.n_owned + L.recv_off[k]) * ts, (size_t)(L.recv_off[k + 1] - L.recv_off[k]), dt,
                              L.recv_peer[k], D->nccl_comm, D->comm));
        for (size_t k = 0; k < L.send_peer.size(); ++k)
            PAMG_NCCL(R->Send((const char *)L.send_buf + (size_t)L.send_off[k] * ts, (size_t)(L.send_off[k + 1] - L.send_off[k]), dt,
                              L.send_peer[k], D->nccl_comm, D->comm));
        PAMG_NCCL(R->GroupEnd());
        PAMG_HIP(hipEventRecord(D->ev_halo, D->comm));
    }
    return PAMG_OK;
}

int finish_exchange(pamg_dist_s *D, int l, void *v)
{
    DLevel &L = D->lv[l];
    if (D->mode == 2) return (int)hipStreamWaitEvent(D->main, D->ev_halo, 0);
    if (D->mode == 3) return PAMG_OK;                        // model transport: the halo keeps what it holds, the launches are the production ones
    if (D->mode == 1 && D->xmode == 1) {
        // host-callback rigs: the all-gather is the sum of the ranks' slices of an otherwise zero vector (the all-reduce callback)
        const size_t ts = ts_of(D);
        PAMG_HIP(hipMemsetAsync(L.ag_all, 0, (size_t)D->world * (size_t)L.ag_count * ts, D->main));
        if (L.n_owned) PAMG_HIP(hipMemcpyAsync((char *)L.ag_all + (size_t)D->rank * (size_t)L.ag_count * ts, L.ag_stage, (size_t)L.n_owned * ts, hipMemcpyDeviceToDevice, D->main));
        PAMG_HIP(hipStreamSynchronize(D->main));
        PAMG_TRY(D->cb_allreduce(D->cb_user, L.ag_all, (int64_t)D->world * L.ag_count, D->dtype));
        if (L.n_halo) PAMG_TRY(pamg_vec_gather(D->dtype, L.n_halo, L.d_halo_src, L.ag_all, (char *)v + (size_t)L.n_owned * ts, D->main));
        return PAMG_OK;
    }
    if (D->mode == 1) {
        PAMG_HIP(hipEventSynchronize(D->ev_pack));                 // the packed values are in send_buf; later main-stream work keeps running
        const int64_t ns = L.send_off.empty() ? 0 : L.send_off.back();
        return D->cb_exchange(D->cb_user, l, L.send_buf, ns, (char *)v + (size_t)L.n_owned * ts_of(D), L.n_halo);
    }
    return PAMG_E_STATE;
}

// y = epi(M, v, ...) where v is a level-lv vector whose halo must be refreshed first: interior ranges of M run while
// the halo is in flight, boundary ranges behind it.  exchange = false: the halo is known to be current (or all zero).
int xlaunch(pamg_dist_s *D, int lvec, void *v, bool exchange, pamg_matrix_s *M, int epi, const void *b, void *y, double c,
            double omega, double *partial)
{
    hipStream_t s = D->main;
    if (!exchange || !D->lv[lvec].talks()) return M->nrows ? stream_launch(M, epi, v, b, y, c, omega, partial, s) : PAMG_OK;
    D->n_exchanges++;
    D->lv[lvec].n_ex++;
    PAMG_TRY(begin_exchange(D, lvec, v));
    const bool split = D->overlap && M->part_cols >= 0 && M->R == 1 && M->C == 1 && M->nrows > 0;
    if (split) {
        D->n_overlapped++;
        PAMG_TRY(stream_launch_part(M, 1, epi, v, b, y, c, omega, partial, s));
    }
    PAMG_TRY(finish_exchange(D, lvec, v));
    if (M->nrows == 0) return PAMG_OK;
    if (split) return stream_launch_part(M, 2, epi, v, b, y, c, omega, partial, s);
    return stream_launch(M, epi, v, b, y, c, omega, partial, s);
}

// exchange only (block operators: their relaxation kernels take the whole shard at once)
int xonly(pamg_dist_s *D, int l, void *v)
{
    if (!D->lv[l].talks()) return PAMG_OK;
    D->n_exchanges++;
    D->lv[l].n_ex++;
    PAMG_TRY(begin_exchange(D, l, v));
    return finish_exchange(D, l, v);
}

// relaxation.jacobi / block_jacobi / polynomial on a row shard (reference: relaxation.py:349-420, 423-499, 585-659)
int smooth(pamg_dist_s *D, int l, const DSmoother &sm, bool x_zero)
{
    DLevel &L = D->lv[l];
    pamg_matrix_s *A = L.A;
    hipStream_t s = D->main;
    switch (sm.kind) {
        case PAMG_SMOOTH_NONE: return PAMG_OK;
        case PAMG_SMOOTH_JACOBI:
            for (int it = 0; it < sm.iterations; ++it) {
                const bool ex = !(x_zero && it == 0);              // the halo of an all-zero iterate is zero already
                if (A->R > 1) {
                    if (ex) PAMG_TRY(xonly(D, l, L.x));
                    PAMG_TRY(block_jacobi_step(A, PNT_JACOBI, nullptr, L.x, L.xalt, L.b, sm.omega, s));
                } else {
                    PAMG_TRY(xlaunch(D, l, L.x, ex, A, A->flavour == PAMG_BSR ? EPI_JACOBI_B : EPI_JACOBI, L.b, L.xalt, 0.0, sm.omega, nullptr));
                }
                std::swap(L.x, L.xalt);
            }
            return PAMG_OK;
        case PAMG_SMOOTH_BLOCK_JACOBI:
            for (int it = 0; it < sm.iterations; ++it) {
                if (!(x_zero && it == 0)) PAMG_TRY(xonly(D, l, L.x));
                PAMG_TRY(block_jacobi_step(A, BLK_JACOBI, sm.d_Dinv, L.x, L.xalt, L.b, sm.omega, s));
                std::swap(L.x, L.xalt);
            }
            return PAMG_OK;
        case PAMG_SMOOTH_POLY: {
            const int nc = (int)sm.coeffs.size();
            const double *co = sm.coeffs.data();
            for (int it = 0; it < sm.iterations; ++it) {
                const void *res = L.b;
                bool have_h = false;
                if (!(x_zero && it == 0)) {
                    // res = b - A x, and h = c0 res by the same launch when there is a Horner step to follow (row_finish, pamg_kernels.h)
                    PAMG_TRY(xlaunch(D, l, L.x, true, A, EPI_RESID, L.b, L.r, nc > 1 ? co[0] : 0.0, 0.0, nc > 1 ? (double *)L.h0 : nullptr));
                    res = L.r;
                    have_h = nc > 1;
                }
                if (nc == 1) { PAMG_TRY(vec_axpy(D->dtype, L.n_owned, co[0], res, L.x, s)); continue; }
                if (!have_h) PAMG_TRY(vec_scale(D->dtype, L.n_owned, co[0], res, L.h0, s));              // h = c0 res
                void *hc = L.h0, *hn = L.h1;
                for (int k = 1; k < nc - 1; ++k) {
                    PAMG_TRY(xlaunch(D, l, hc, true, A, EPI_AXPBY, res, hn, co[k], 0.0, nullptr));       // h = c res + A h
                    std::swap(hc, hn);
                }
                PAMG_TRY(xlaunch(D, l, hc, true, A, EPI_ACC_AXPBY, res, L.x, co[nc - 1], 0.0, nullptr)); // x += c res + A h
            }
            return PAMG_OK;
        }
    }
    return PAMG_E_UNSUPPORTED;
}

// multilevel.py:584-662 on the sharded levels (V-cycle)
int cycle(pamg_dist_s *D, int l, bool x_zero)
{
    const int ns = (int)D->lv.size() - 1;
    DLevel &L = D->lv[l];
    DLevel &N = D->lv[l + 1];
    hipStream_t s = D->main;
    const size_t ts = ts_of(D);
    PAMG_TRY(smooth(D, l, L.pre, x_zero));
    PAMG_TRY(xlaunch(D, l, L.x, true, L.A, EPI_RESID, L.b, L.r, 0.0, 0.0, nullptr));                     // r = b - A x
    if (l + 1 < ns) {
        PAMG_TRY(xlaunch(D, l, L.r, true, L.R, EPI_SET, nullptr, N.b, 0.0, 0.0, nullptr));               // b_c = R r
        PAMG_HIP(hipMemsetAsync(N.x, 0, (size_t)N.n_local() * ts, s));                                   // x_c = 0, halo included
        PAMG_TRY(cycle(D, l + 1, true));
        PAMG_TRY(xlaunch(D, l + 1, N.x, true, L.P, EPI_ACC, nullptr, L.x, 0.0, 0.0, nullptr));           // x += P x_c
    } else {
        // collapse: every rank contributes its slice of b_c, all ranks run the small rest of the cycle redundantly
        PAMG_HIP(hipMemsetAsync(D->bc_full, 0, (size_t)D->nc * ts, s));
        PAMG_TRY(xlaunch(D, l, L.r, true, L.R, EPI_SET, nullptr, N.b, 0.0, 0.0, nullptr));
        if (N.n_owned)
            PAMG_HIP(hipMemcpyAsync((char *)D->bc_full + (size_t)D->c_row0 * ts, N.b, (size_t)N.n_owned * ts, hipMemcpyDeviceToDevice, s));
        PAMG_TRY(all_reduce(D, D->bc_full, D->nc, D->dtype));
        PAMG_HIP(hipMemsetAsync(D->xc_full, 0, (size_t)D->nc * ts, s));
        PAMG_TRY(solver_cycle_inline(D->coarse, D->xc_full, D->bc_full, PAMG_CYCLE_V, 1, s, !D->capturing));
        if (N.n_local()) PAMG_TRY(pamg_vec_gather(D->dtype, N.n_local(), D->d_fill, D->xc_full, N.x, s));
        PAMG_TRY(xlaunch(D, l + 1, N.x, false, L.P, EPI_ACC, nullptr, L.x, 0.0, 0.0, nullptr));          // x += P x_c (halo filled by the gather)
    }
    PAMG_TRY(smooth(D, l, L.post, false));
    if (L.x != L.x_home) {          // odd number of ping-pong swaps: bring the iterate home (graph replays start and end there)
        PAMG_HIP(hipMemcpyAsync(L.x_home, L.x, (size_t)L.n_owned * ts, hipMemcpyDeviceToDevice, s));
        std::swap(L.x, L.xalt);
    }
    return PAMG_OK;
}

// ||b - A x||^2 over all ranks -> d_ss[0] (multilevel.py:567)
int resid_sumsq(pamg_dist_s *D)
{
    DLevel &L = D->lv[0];
    PAMG_TRY(xlaunch(D, 0, L.x, true, L.A, EPI_SUMSQ, L.b, nullptr, 0.0, 0.0, L.A->d_partial));
    PAMG_TRY(reduce_partials(L.A->d_partial, L.A->nblk, D->d_ss, D->main));
    return all_reduce(D, D->d_ss, 1, PAMG_F64);
}

int enqueue_iteration(pamg_dist_s *D, bool norm)
{
    D->n_exchanges = D->n_overlapped = 0;
    for (DLevel &L_ : D->lv) L_.n_ex = 0;
    PAMG_TRY(cycle(D, 0, false));
    if (norm) PAMG_TRY(resid_sumsq(D));
    return PAMG_OK;
}

int run_iteration(pamg_dist_s *D, bool norm)
{
    if (!graph_ok(D)) return enqueue_iteration(D, norm);
    hipGraphExec_t &ex = D->graph[norm ? 1 : 0];
    if (!ex) {
        hipGraph_t g = nullptr;
        PAMG_HIP(hipStreamBeginCapture(D->main, hipStreamCaptureModeThreadLocal));
        D->capturing = true;
        const int st = enqueue_iteration(D, norm);
        D->capturing = false;
        const hipError_t e = hipStreamEndCapture(D->main, &g);
        if (st != PAMG_OK) { if (g) hipGraphDestroy(g); return st; }
        if (e != hipSuccess) return (int)e;
        PAMG_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        hipGraphDestroy(g);
    }
    return (int)hipGraphLaunch(ex, D->main);
}

void free_smoother(DSmoother &sm) { if (sm.d_Dinv) { hipFree(sm.d_Dinv); sm.d_Dinv = nullptr; } }

}  // namespace

extern "C" {

int pamg_dist_create(pamg_dist_t *out, int dtype, int rank, int world)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return PAMG_E_ARG;
    if (dtype != PAMG_F64 && dtype != PAMG_F32) return PAMG_E_UNSUPPORTED;
    pamg_dist_s *D = new (std::nothrow) pamg_dist_s();
    if (!D) return PAMG_E_ALLOC;
    D->dtype = dtype; D->rank = rank; D->world = world;
    const char *e = getenv("PAMG_DIST_OVERLAP");
    if (e && *e == '0') D->overlap = false;
    *out = D;
    return PAMG_OK;
}

int pamg_dist_destroy(pamg_dist_t D)
{
    if (!D) return PAMG_OK;
    for (int k = 0; k < 2; ++k) if (D->graph[k]) hipGraphExecDestroy(D->graph[k]);
    for (DLevel &L : D->lv) {
        for (pamg_matrix_s *M : {L.A, L.P, L.R}) if (M) M->borrowed--;
        hipFree(L.d_send_idx); hipFree(L.send_buf);
        hipFree(L.d_halo_src); hipFree(L.ag_stage); hipFree(L.ag_all);
        hipFree(L.x_home); hipFree(L.x_home == L.x ? L.xalt : L.x); hipFree(L.b); hipFree(L.r); hipFree(L.h0); hipFree(L.h1);
        free_smoother(L.pre); free_smoother(L.post);
    }
    hipFree(D->d_fill); hipFree(D->bc_full); hipFree(D->xc_full); hipFree(D->d_ss); hipFree(D->d_norms);
    if (D->nccl_comm && rccl()) rccl()->CommDestroy(D->nccl_comm);
    if (D->ev_pack) hipEventDestroy(D->ev_pack);
    if (D->ev_halo) hipEventDestroy(D->ev_halo);
    if (D->main) hipStreamDestroy(D->main);
    if (D->comm) hipStreamDestroy(D->comm);
    delete D;
    return PAMG_OK;
}

static int set_plan(pamg_dist_s *D, DLevel &L, int64_t n_owned, int64_t n_halo, int nsend, const int *send_peer, const int64_t *send_off,
                    const int32_t *send_idx, int nrecv, const int *recv_peer, const int64_t *recv_off)
{
    if (n_owned < 0 || n_halo < 0 || nsend < 0 || nrecv < 0) return PAMG_E_ARG;
    if ((nsend && (!send_peer || !send_off || !send_idx)) || (nrecv && (!recv_peer || !recv_off))) return PAMG_E_ARG;
    L.n_owned = n_owned; L.n_halo = n_halo;
    for (int k = 0; k < nsend; ++k) {
        if (send_peer[k] < 0 || send_peer[k] >= D->world || send_peer[k] == D->rank || send_off[k + 1] < send_off[k]) return PAMG_E_ARG;
        L.send_peer.push_back(send_peer[k]);
    }
    for (int k = 0; k < nrecv; ++k) {
        if (recv_peer[k] < 0 || recv_peer[k] >= D->world || recv_peer[k] == D->rank || recv_off[k + 1] < recv_off[k]) return PAMG_E_ARG;
        L.recv_peer.push_back(recv_peer[k]);
    }
    if (nsend) L.send_off.assign(send_off, send_off + nsend + 1);
    if (nrecv) {
        L.recv_off.assign(recv_off, recv_off + nrecv + 1);
        if (L.recv_off.front() != 0 || L.recv_off.back() != n_halo) return PAMG_E_ARG;
    } else if (n_halo) return PAMG_E_ARG;                       // halo entries nobody would ever fill
    const int64_t ns = nsend ? L.send_off.back() : 0;
    for (int64_t i = 0; i < ns; ++i) if (send_idx[i] < 0 || send_idx[i] >= n_owned) return PAMG_E_ARG;
    const size_t ts = ts_of(D);
    if (ns) {
        PAMG_HIP(hipMalloc((void **)&L.d_send_idx, sizeof(int) * (size_t)ns));
        PAMG_HIP(hipMemcpy(L.d_send_idx, send_idx, sizeof(int) * (size_t)ns, hipMemcpyHostToDevice));
        PAMG_TRY(dmalloc(D, &L.send_buf, (size_t)ns * ts));
    }
    return PAMG_OK;
}

int pamg_dist_add_level(pamg_dist_t D, pamg_matrix_t A, pamg_matrix_t P, pamg_matrix_t R, int64_t n_owned, int64_t n_halo,
                        int nsend, const int *send_peer, const int64_t *send_off, const int32_t *send_idx,
                        int nrecv, const int *recv_peer, const int64_t *recv_off)
{
    if (!D || !A || !P || !R) return PAMG_E_ARG;
    if (D->finalized || D->collapse_set) return PAMG_E_STATE;
    if (A->dtype != D->dtype || P->dtype != D->dtype || R->dtype != D->dtype) return PAMG_E_ARG;
    if (A->nrows != n_owned || A->ncols != n_owned + n_halo || P->nrows != n_owned || R->ncols != n_owned + n_halo) return PAMG_E_ARG;
    if (!D->lv.empty()) {
        const DLevel &F = D->lv.back();
        if (F.P->ncols != n_owned + n_halo || F.R->nrows != n_owned) return PAMG_E_ARG;
    }
    D->lv.emplace_back();
    DLevel &L = D->lv.back();
    L.A = A; L.P = P; L.R = R;
    const int st = set_plan(D, L, n_owned, n_halo, nsend, send_peer, send_off, send_idx, nrecv, recv_peer, recv_off);
    if (st != PAMG_OK) { hipFree(L.d_send_idx); hipFree(L.send_buf); D->lv.pop_back(); return st; }
    A->borrowed++; P->borrowed++; R->borrowed++;
    return PAMG_OK;
}

int pamg_dist_set_collapse(pamg_dist_t D, pamg_solver_t coarse, int64_t nc, int64_t row0, int64_t n_owned, int64_t n_halo,
                           const int32_t *fill_idx)
{
    if (!D || !coarse || nc < 0 || row0 < 0 || n_owned < 0 || n_halo < 0 || row0 + n_owned > nc) return PAMG_E_ARG;
    if (D->finalized || D->collapse_set || D->lv.empty()) return PAMG_E_STATE;
    if ((n_owned + n_halo) && !fill_idx) return PAMG_E_ARG;
    const DLevel &F = D->lv.back();
    if (F.P->ncols != n_owned + n_halo || F.R->nrows != n_owned) return PAMG_E_ARG;
    for (int64_t i = 0; i < n_owned + n_halo; ++i) if (fill_idx[i] < 0 || fill_idx[i] >= nc) return PAMG_E_ARG;
    D->lv.emplace_back();
    DLevel &L = D->lv.back();
    L.n_owned = n_owned; L.n_halo = n_halo;
    D->coarse = coarse; D->nc = nc; D->c_row0 = row0;
    if (n_owned + n_halo) {
        PAMG_HIP(hipMalloc((void **)&D->d_fill, sizeof(int) * (size_t)(n_owned + n_halo)));
        PAMG_HIP(hipMemcpy(D->d_fill, fill_idx, sizeof(int) * (size_t)(n_owned + n_halo), hipMemcpyHostToDevice));
    }
    D->collapse_set = true;
    return PAMG_OK;
}

int pamg_dist_set_smoother(pamg_dist_t D, int level, int which, int kind, int iterations, double omega, const double *coeffs,
                           int ncoeffs, const void *Dinv, int blocksize)
{
    if (!D || level < 0 || level >= (int)D->lv.size() || (which != 0 && which != 1) || iterations < 0) return PAMG_E_ARG;
    if (D->finalized) return PAMG_E_STATE;
    DLevel &L = D->lv[level];
    if (!L.A) return PAMG_E_ARG;
    if (kind != PAMG_SMOOTH_NONE && kind != PAMG_SMOOTH_JACOBI && kind != PAMG_SMOOTH_POLY && kind != PAMG_SMOOTH_BLOCK_JACOBI)
        return PAMG_E_UNSUPPORTED;                                  // order-exact sweeps do not shard
    DSmoother &sm = which == 0 ? L.pre : L.post;
    free_smoother(sm);
    sm = DSmoother();
    sm.kind = kind; sm.iterations = iterations; sm.omega = omega; sm.blocksize = blocksize;
    if (kind == PAMG_SMOOTH_POLY) {
        if (!coeffs || ncoeffs < 1) return PAMG_E_ARG;
        sm.coeffs.assign(coeffs, coeffs + ncoeffs);
    }
    if (kind == PAMG_SMOOTH_JACOBI && L.A->R != L.A->C) return PAMG_E_ARG;
    if (kind == PAMG_SMOOTH_BLOCK_JACOBI) {
        if (!Dinv || blocksize < 2 || L.A->R != blocksize || L.A->C != blocksize) return PAMG_E_ARG;
        const size_t sz = (size_t)L.A->n_brow * blocksize * blocksize * ts_of(D);
        PAMG_HIP(hipMalloc(&sm.d_Dinv, std::max<size_t>(sz, 256)));
        PAMG_HIP(hipMemcpy(sm.d_Dinv, Dinv, sz, hipMemcpyHostToDevice));
        D->bytes += sz;
    }
    return PAMG_OK;
}

int pamg_dist_set_callbacks(pamg_dist_t D, pamg_dist_exchange_fn exchange, pamg_dist_allreduce_fn allreduce, void *user)
{
    if (!D || !exchange || !allreduce) return PAMG_E_ARG;
    if (D->finalized) return PAMG_E_STATE;
    D->mode = 1; D->cb_exchange = exchange; D->cb_allreduce = allreduce; D->cb_user = user;
    return PAMG_OK;
}

// MODEL transport (bench.py's modelled 1 -> 8 GPU curve, SURVEY 8e "availability caveat"): the rank runs exactly the launches it would run
// among `world` ranks -- pack, interior ranges, boundary ranges, collapse, tail -- but nothing travels: halos keep what they hold and the
// all-reduces return the rank's own contribution.  What such a run measures is the rank's COMPUTE critical path; the wire is added from the
// exchange plans (pamg_dist_level_info) with stated link figures.  Results are NOT a solve.
int pamg_dist_set_model_transport(pamg_dist_t D)
{
    if (!D) return PAMG_E_ARG;
    if (D->finalized) return PAMG_E_STATE;
    D->mode = 3;
    return PAMG_OK;
}

/* info of sharded level `level` (or the collapse level = the number of sharded levels): [0] owned values [1] halo values [2] exchanges of this
 * level's vectors per iteration (counted while the last iteration was enqueued) [3] peers sent to [4] peers received from [5] most values
 * sent to one peer [6] most values received from one peer [7] values sent per exchange */
int pamg_dist_level_info(pamg_dist_t D, int level, int64_t info[8])
{
    if (!D || !info || level < 0 || level >= (int)D->lv.size()) return PAMG_E_ARG;
    const DLevel &L = D->lv[level];
    for (int k = 0; k < 8; ++k) info[k] = 0;
    info[0] = L.n_owned; info[1] = L.n_halo; info[2] = L.n_ex;
    info[3] = (int64_t)L.send_peer.size(); info[4] = (int64_t)L.recv_peer.size();
    for (size_t k = 0; k + 1 < L.send_off.size(); ++k) info[5] = std::max<int64_t>(info[5], L.send_off[k + 1] - L.send_off[k]);
    for (size_t k = 0; k + 1 < L.recv_off.size(); ++k) info[6] = std::max<int64_t>(info[6], L.recv_off[k + 1] - L.recv_off[k]);
    info[7] = L.send_off.empty() ? 0 : L.send_off.back();
    return PAMG_OK;
}

int pamg_dist_set_allgather(pamg_dist_t D, int level, int64_t count_per_rank, const int32_t *halo_src)
{
    if (!D || level < 0 || level >= (int)D->lv.size() || count_per_rank < 0) return PAMG_E_ARG;
    if (D->finalized) return PAMG_E_STATE;
    DLevel &L = D->lv[level];
    if (count_per_rank < L.n_owned || (L.n_halo && !halo_src)) return PAMG_E_ARG;
    for (int64_t i = 0; i < L.n_halo; ++i)
        if (halo_src[i] < 0 || (int64_t)halo_src[i] >= (int64_t)D->world * count_per_rank) return PAMG_E_ARG;
    const size_t ts = ts_of(D);
    hipFree(L.d_halo_src); hipFree(L.ag_stage); hipFree(L.ag_all);
    L.d_halo_src = nullptr; L.ag_stage = L.ag_all = nullptr;
    PAMG_TRY(dmalloc(D, &L.ag_stage, (size_t)std::max<int64_t>(count_per_rank, 1) * ts));
    PAMG_TRY(dmalloc(D, &L.ag_all, (size_t)std::max<int64_t>(count_per_rank, 1) * (size_t)D->world * ts));
    if (L.n_halo) {
        PAMG_HIP(hipMalloc((void **)&L.d_halo_src, sizeof(int) * (size_t)L.n_halo));
        PAMG_HIP(hipMemcpy(L.d_halo_src, halo_src, sizeof(int) * (size_t)L.n_halo, hipMemcpyHostToDevice));
    }
    L.ag_count = std::max<int64_t>(count_per_rank, 1);
    return PAMG_OK;
}

int pamg_dist_set_exchange(pamg_dist_t D, int mode)
{
    if (!D || (mode != 0 && mode != 1)) return PAMG_E_ARG;
    if (mode == 1)
        for (const DLevel &L : D->lv)
            if (L.talks() && !L.ag_count) return PAMG_E_STATE;
    if (D->xmode != mode) {
        for (int k = 0; k < 2; ++k) if (D->graph[k]) { hipGraphExecDestroy(D->graph[k]); D->graph[k] = nullptr; }
    }
    D->xmode = mode;
    return PAMG_OK;
}

// One-rank exercise of everything the sharded cycle asks of RCCL, on the current device: a communicator from a fresh id, a
// grouped ncclSend + ncclRecv (to itself) on a comm stream ordered against a main stream by the two events of
// begin_exchange / finish_exchange, a 1-element ncclAllReduce and an ncclAllGather -- through the dlsym'd table, with the
// enum constants this file hard-codes.  *max_err = largest deviation of a received value from the value sent.
int pamg_rccl_selftest(int64_t n, double *max_err)
{
    if (n < 1 || !max_err) return PAMG_E_ARG;
    *max_err = -1.0;
    Rccl *R = rccl();
    if (!R) return PAMG_E_UNSUPPORTED;
    RcclId id;
    PAMG_NCCL(R->GetUniqueId(&id));
    void *comm = nullptr;
    PAMG_NCCL(R->CommInitRank(&comm, 1, id, 0));
    hipStream_t main = nullptr, cs = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    double *src = nullptr, *dst = nullptr, *one = nullptr, *all = nullptr;
    int st = PAMG_OK;
    std::vector<double> h((size_t)n), back((size_t)n, -1.0), back2((size_t)n, -1.0);
    for (int64_t i = 0; i < n; ++i) h[(size_t)i] = 0.25 + (double)(i % 4099) * 1.5;
    double hone = 3.5, hone_back = 0.0;
    auto run = [&]() -> int {
        PAMG_HIP(hipStreamCreateWithFlags(&main, hipStreamNonBlocking));
        PAMG_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        PAMG_HIP(hipEventCreateWithFlags(&ev_pack, hipEventDisableTiming));
        PAMG_HIP(hipEventCreateWithFlags(&ev_halo, hipEventDisableTiming));
        PAMG_HIP(hipMalloc((void **)&src, (size_t)n * 8)); PAMG_HIP(hipMalloc((void **)&dst, (size_t)n * 8));
        PAMG_HIP(hipMalloc((void **)&one, 16)); PAMG_HIP(hipMalloc((void **)&all, (size_t)n * 8));
        PAMG_HIP(hipMemsetAsync(dst, 0xFF, (size_t)n * 8, main));
        PAMG_HIP(hipMemcpyAsync(src, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, main));
        PAMG_HIP(hipMemcpyAsync(one, &hone, 8, hipMemcpyHostToDevice, main));
        PAMG_HIP(hipEventRecord(ev_pack, main));
        PAMG_HIP(hipStreamWaitEvent(cs, ev_pack, 0));
        PAMG_NCCL(R->GroupStart());
        PAMG_NCCL(R->Recv(dst, (size_t)n, NCCL_F64, 0, comm, cs));
        PAMG_NCCL(R->Send(src, (size_t)n, NCCL_F64, 0, comm, cs));
        PAMG_NCCL(R->GroupEnd());
        PAMG_NCCL(R->AllGather(src, all, (size_t)n, NCCL_F64, comm, cs));
        PAMG_HIP(hipEventRecord(ev_halo, cs));
        PAMG_HIP(hipStreamWaitEvent(main, ev_halo, 0));
        PAMG_NCCL(R->AllReduce(one, one, 1, NCCL_F64, NCCL_SUM, comm, main));
        PAMG_HIP(hipMemcpyAsync(back.data(), dst, (size_t)n * 8, hipMemcpyDeviceToHost, main));
        PAMG_HIP(hipMemcpyAsync(back2.data(), all, (size_t)n * 8, hipMemcpyDeviceToHost, main));
        PAMG_HIP(hipMemcpyAsync(&hone_back, one, 8, hipMemcpyDeviceToHost, main));
        PAMG_HIP(hipStreamSynchronize(main));
        return PAMG_OK;
    };
    st = run();
    if (st == PAMG_OK) {
        double e = std::fabs(hone_back - hone);
        for (int64_t i = 0; i < n; ++i) e = std::max(e, std::max(std::fabs(back[(size_t)i] - h[(size_t)i]), std::fabs(back2[(size_t)i] - h[(size_t)i])));
        if (!(e == e)) e = 1e300;                              // NaN (the 0xFF fill survived): nothing arrived
        *max_err = e;
    }
    hipFree(src); hipFree(dst); hipFree(one); hipFree(all);
    if (ev_pack) hipEventDestroy(ev_pack);
    if (ev_halo) hipEventDestroy(ev_halo);
    if (main) hipStreamDestroy(main);
    if (cs) hipStreamDestroy(cs);
    R->CommDestroy(comm);
    return st;
}

int pamg_rccl_available(void) { return rccl() ? PAMG_OK : PAMG_E_UNSUPPORTED; }

int pamg_dist_rccl_unique_id(void *id128)
{
    if (!id128) return PAMG_E_ARG;
    Rccl *R = rccl();
    if (!R) return PAMG_E_UNSUPPORTED;
    RcclId id;
    PAMG_NCCL(R->GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return PAMG_OK;
}

int pamg_dist_set_rccl(pamg_dist_t D, const void *id128)
{
    if (!D || !id128) return PAMG_E_ARG;
    if (D->finalized || D->nccl_comm) return PAMG_E_STATE;
    Rccl *R = rccl();
    if (!R) return PAMG_E_UNSUPPORTED;
    RcclId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    PAMG_NCCL(R->CommInitRank(&D->nccl_comm, D->world, id, D->rank));
    D->mode = 2;
    return PAMG_OK;
}

int pamg_dist_finalize(pamg_dist_t D)
{
    if (!D) return PAMG_E_ARG;
    if (D->finalized) return PAMG_OK;
    if (!D->collapse_set || D->lv.size() < 2) return PAMG_E_STATE;
    bool talks = false;
    for (const DLevel &L : D->lv) talks = talks || L.talks();
    if ((talks || D->world > 1) && D->mode == 0) return PAMG_E_STATE;          // peers, but no transport
    const size_t ts = ts_of(D);
    const int ns = (int)D->lv.size() - 1;
    for (int l = 0; l <= ns; ++l) {
        DLevel &L = D->lv[l];
        const size_t vb = (size_t)L.n_local() * ts;
        PAMG_TRY(dmalloc(D, &L.x, vb));
        PAMG_TRY(dmalloc(D, &L.b, vb));
        L.x_home = L.x;
        if (l == ns) break;
        PAMG_TRY(dmalloc(D, &L.xalt, vb));
        PAMG_TRY(dmalloc(D, &L.r, vb));
        if (L.pre.kind == PAMG_SMOOTH_POLY || L.post.kind == PAMG_SMOOTH_POLY) {
            PAMG_TRY(dmalloc(D, &L.h0, vb));
            PAMG_TRY(dmalloc(D, &L.h1, vb));
        }
        // interior / boundary split of every operator that follows an exchange: A reads level-l vectors, R too; P reads level l+1
        PAMG_TRY(matrix_split_ranges(L.A, L.n_owned));
        PAMG_TRY(matrix_split_ranges(L.R, L.n_owned));
        PAMG_TRY(matrix_split_ranges(L.P, D->lv[l + 1].n_owned));
    }
    PAMG_TRY(dmalloc(D, &D->bc_full, (size_t)D->nc * ts));
    PAMG_TRY(dmalloc(D, &D->xc_full, (size_t)D->nc * ts));
    PAMG_TRY(dmalloc(D, (void **)&D->d_ss, 4 * sizeof(double)));
    PAMG_HIP(hipStreamCreateWithFlags(&D->main, hipStreamNonBlocking));
    PAMG_HIP(hipStreamCreateWithFlags(&D->comm, hipStreamNonBlocking));
    PAMG_HIP(hipEventCreateWithFlags(&D->ev_pack, hipEventDisableTiming));
    PAMG_HIP(hipEventCreateWithFlags(&D->ev_halo, hipEventDisableTiming));
    D->finalized = true;
    return PAMG_OK;
}

int pamg_dist_set_options(pamg_dist_t D, int use_graph, int overlap)
{
    if (!D) return PAMG_E_ARG;
    if (use_graph >= 0) D->use_graph = use_graph != 0;
    if (overlap >= 0) D->overlap = overlap != 0;
    for (int k = 0; k < 2; ++k) if (D->graph[k]) { hipGraphExecDestroy(D->graph[k]); D->graph[k] = nullptr; }
    return PAMG_OK;
}

// x, b: DEVICE pointers to this rank's owned slices (n_owned values each)
int pamg_dist_load(pamg_dist_t D, const void *x_owned, const void *b_owned)
{
    if (!D || !x_owned || !b_owned) return PAMG_E_ARG;
    if (!D->finalized) return PAMG_E_STATE;
    DLevel &L = D->lv[0];
    const size_t vb = (size_t)L.n_owned * ts_of(D);
    PAMG_HIP(hipDeviceSynchronize());                      // the inputs may come from any stream
    PAMG_HIP(hipMemcpyAsync(L.x, x_owned, vb, hipMemcpyDeviceToDevice, D->main));
    PAMG_HIP(hipMemcpyAsync(L.b, b_owned, vb, hipMemcpyDeviceToDevice, D->main));
    return (int)hipStreamSynchronize(D->main);
}

int pamg_dist_store(pamg_dist_t D, void *x_owned)
{
    if (!D || !x_owned) return PAMG_E_ARG;
    if (!D->finalized) return PAMG_E_STATE;
    DLevel &L = D->lv[0];
    PAMG_HIP(hipMemcpyAsync(x_owned, L.x, (size_t)L.n_owned * ts_of(D), hipMemcpyDeviceToDevice, D->main));
    return (int)hipStreamSynchronize(D->main);
}

// k x (V-cycle [+ all-reduced convergence-check norm]); residuals: HOST, k norms, or NULL (no norms are computed).
// Returns after the work is queued when residuals == NULL (pamg_dist_sync waits); otherwise synchronises.
int pamg_dist_iterate(pamg_dist_t D, int k, double *residuals)
{
    if (!D || k < 0) return PAMG_E_ARG;
    if (!D->finalized) return PAMG_E_STATE;
    if (residuals && D->norms_cap < k + 1) {
        if (D->d_norms) hipFree(D->d_norms);
        D->norms_cap = k + 1;
        PAMG_HIP(hipMalloc((void **)&D->d_norms, sizeof(double) * (size_t)D->norms_cap));
    }
    for (int it = 0; it < k; ++it) {
        PAMG_TRY(run_iteration(D, residuals != nullptr));
        if (residuals) PAMG_HIP(hipMemcpyAsync(D->d_norms + it, D->d_ss, sizeof(double), hipMemcpyDeviceToDevice, D->main));
    }
    if (residuals && k > 0) {
        PAMG_HIP(hipMemcpyAsync(residuals, D->d_norms, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost, D->main));
        PAMG_HIP(hipStreamSynchronize(D->main));
        for (int it = 0; it < k; ++it) residuals[it] = std::sqrt(residuals[it]);
    }
    return PAMG_OK;
}

// ||b - A x||_2 of the resident iterate, all ranks (synchronises)
int pamg_dist_resid_norm(pamg_dist_t D, double *norm)
{
    if (!D || !norm) return PAMG_E_ARG;
    if (!D->finalized) return PAMG_E_STATE;
    PAMG_TRY(resid_sumsq(D));
    double h = 0.0;
    PAMG_HIP(hipMemcpyAsync(&h, D->d_ss, sizeof(double), hipMemcpyDeviceToHost, D->main));
    PAMG_HIP(hipStreamSynchronize(D->main));
    *norm = std::sqrt(h);
    return PAMG_OK;
}

int pamg_dist_sync(pamg_dist_t D)
{
    if (!D || !D->finalized) return PAMG_E_STATE;
    PAMG_HIP(hipStreamSynchronize(D->comm));
    return (int)hipStreamSynchronize(D->main);
}

int pamg_dist_stream(pamg_dist_t D, pamg_stream_t *s)
{
    if (!D || !s || !D->finalized) return PAMG_E_STATE;
    *s = (pamg_stream_t)D->main;
    return PAMG_OK;
}

/* Transport self-test: the owned part of level `level`'s vector is set from the HOST array x_owned, ONE halo exchange runs
 * (exactly the code path of the cycle: pack, transfers, wait), the halo part comes back in the HOST array halo_out.  A caller
 * that fills x_owned with global indices can check every received value (pyamg_amd/dist.py does, before the first cycle). */
int pamg_dist_exchange_test(pamg_dist_t D, int level, const void *x_owned, void *halo_out)
{
    if (!D || !D->finalized || level < 0 || level >= (int)D->lv.size() || !x_owned || !halo_out) return PAMG_E_ARG;
    DLevel &L = D->lv[level];
    const size_t ts = ts_of(D);
    PAMG_HIP(hipStreamSynchronize(D->main));
    PAMG_HIP(hipMemcpy(L.x, x_owned, (size_t)L.n_owned * ts, hipMemcpyHostToDevice));
    if (L.n_halo) PAMG_HIP(hipMemset((char *)L.x + (size_t)L.n_owned * ts, 0xFF, (size_t)L.n_halo * ts));
    if (L.talks()) {
        PAMG_TRY(begin_exchange(D, level, L.x));
        PAMG_TRY(finish_exchange(D, level, L.x));
    }
    PAMG_HIP(hipStreamSynchronize(D->main));
    if (D->comm) PAMG_HIP(hipStreamSynchronize(D->comm));
    if (L.n_halo) PAMG_HIP(hipMemcpy(halo_out, (char *)L.x + (size_t)L.n_owned * ts, (size_t)L.n_halo * ts, hipMemcpyDeviceToHost));
    return PAMG_OK;
}

/* info: [0] sharded levels [1] transport mode (0 none, 1 callbacks, 2 RCCL; + 16 when the exchange is the all-gather form) [2] halo exchanges per iteration (cycle + norm)
 * [3] of those, exchanges overlapped with interior rows [4] whole iteration replayed from a hipGraph [5] bytes of vectors
 * [6] values sent per iteration [7] interior row ranges of the fine-level operator (of [3] of pamg_matrix_info) */
int pamg_dist_info(pamg_dist_t D, int64_t info[8])
{
    if (!D || !info) return PAMG_E_ARG;
    for (int k = 0; k < 8; ++k) info[k] = 0;
    info[0] = (int64_t)D->lv.size() - (D->collapse_set ? 1 : 0);
    info[1] = D->mode + 16 * D->xmode;               // + 16: all-gather form of the exchange
    info[2] = D->n_exchanges; info[3] = D->n_overlapped;
    info[4] = D->finalized && graph_ok(D) ? 1 : 0;
    info[5] = (int64_t)D->bytes;
    int64_t sent = 0;
    for (const DLevel &L : D->lv) if (!L.send_off.empty()) sent += L.send_off.back();
    info[6] = sent;
    if (!D->lv.empty() && D->lv[0].A) info[7] = D->lv[0].A->npart[0];
    return PAMG_OK;
}

}  // extern "C"
