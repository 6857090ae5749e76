// pamg_matrix.hip -- operator handle: upload, planning, dependency-level analysis for the
// order-exact Gauss-Seidel family, kernel dispatch.
#include "pamg_host_threads.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <climits>
#include <new>
#include <thread>

#include "pamg_kernels.h"
#include "pamg_tile_kernels.h"
#include "pamg_tile_plan.h"
#include "pamg_stream_plan.h"

using namespace pamg;

namespace {

constexpr int PAD = 8;   // elements of slack after every index/value array (vector tail loads)

// schedules of one operator may be built by several host threads at once (pamg_solver_finalize: forward and backward
// sweeps, every level); the slot table and the byte count are the only shared state
std::mutex g_sched_mu;

template <typename U>
int upload(U **dptr, const U *h, size_t n, size_t *bytes)
{
    const size_t sz = (n + PAD) * sizeof(U);
    PAMG_HIP(hipMalloc((void **)dptr, sz));
    PAMG_HIP(hipMemset(*dptr, 0, sz));
    if (n) PAMG_HIP(hipMemcpy(*dptr, h, n * sizeof(U), hipMemcpyHostToDevice));
    if (bytes) *bytes += sz;
    return PAMG_OK;
}

int upload_raw(void **dptr, const void *h, size_t n, size_t elt, size_t *bytes)
{
    const size_t sz = (n + PAD) * elt;
    PAMG_HIP(hipMalloc(dptr, sz));
    PAMG_HIP(hipMemset(*dptr, 0, sz));
    if (n) PAMG_HIP(hipMemcpy(*dptr, h, n * elt, hipMemcpyHostToDevice));
    if (bytes) *bytes += sz;
    return PAMG_OK;
}

// row ranges: plan_row_ranges (pamg_stream_plan.h) as int4 {r0, r1, p0, p1}
void plan_rows(const int *Ap, int begin, int end, int cap, int max_rows, std::vector<int4> &out)
{
    std::vector<RowRange> rr;
    plan_row_ranges(Ap, begin, end, cap, max_rows, rr);
    for (const RowRange &q : rr) out.push_back(make_int4(q.r0, q.r1, q.p0, q.p1));
}

int lds_bytes(int dtype, int epi, int cap)
{
    const int per = (int)tsize(dtype) + ((epi == EPI_JACOBI || epi == EPI_JACOBI_B || epi == EPI_JACOBI_IDX) ? 4 : 0);   // column ids only for the Jacobi row phase
    return per * (cap + 8) + 64;      // + slack: the row phase reads whole batches of 8 slots
}

// extra LDS of the 8-bit value-code path: 8 more slots per array (entries are staged in groups of eight) + the dictionary
int val8_lds(const pamg_matrix_s *A, int epi)
{
    const int per = (int)tsize(A->dtype) + ((epi == EPI_JACOBI || epi == EPI_JACOBI_B || epi == EPI_JACOBI_IDX) ? 4 : 0);
    return 8 * per + 16 + (int)tsize(A->dtype) * ((A->nvdict + 1) & ~1);
}

template <typename T, int EPI>
int launch_epi(int npl, int grid, int lds, hipStream_t s, const StreamArgs<T> &a)
{
    if (grid <= 0) return PAMG_OK;
    // two entries per lane and staging step (one and four were measured no better and retired in round 5: tune key 1)
    (void)npl;
    // flag bit 5: an operator WITHOUT value codes through the instantiation that carries their paths.  Same arithmetic, another instruction
    // schedule: on the SA-level operators of the 256^3 hierarchy (one session, profiles/r06_microbench_sa_ops_vc_ab.txt) A1's residual runs 0.198 ms
    // there and 0.219 in the lean instantiation, R0 0.214 / 0.227 -- but P0 0.187 / 0.162 and the fine-level stencil 0.3355 / 0.3145: the autotune times both
    if (a.Ax8 || (a.flags & 32)) {
        if (lds > 48 * 1024)
            PAMG_HIP(hipFuncSetAttribute((const void *)csr_stream_kernel<T, EPI, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((csr_stream_kernel<T, EPI, 2, true>), dim3(grid), dim3(BLK), lds, s, a);
    } else {
        // no value codes on this operator: the instantiation without their run-time tests (pamg_kernels.h: stream_block)
        if (lds > 48 * 1024)
            PAMG_HIP(hipFuncSetAttribute((const void *)csr_stream_kernel<T, EPI, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((csr_stream_kernel<T, EPI, 2, false>), dim3(grid), dim3(BLK), lds, s, a);
    }
    return (int)hipGetLastError();
}

// row-pattern form (csr_rowpat_kernel): square operators with value codes and a row-pattern plan
template <typename T>
int launch_rowpat(int epi, int grid, const pamg_matrix_s *A, hipStream_t s, StreamArgs<T> a)
{
    a.pid = A->d_pid; a.ptab = A->d_ptab; a.npat = A->npat; a.lmax = A->pat_lmax;
    const size_t tabs = (((size_t)(256 + A->npat * A->pat_lmax) * sizeof(int) + 15) & ~(size_t)15) + sizeof(T) * ((size_t)A->npat * A->pat_lmax + 256);
    const int lds = (int)std::max(tabs + 16, (size_t)BLK * sizeof(double));
#define PAMG_RP(E) case E: hipLaunchKernelGGL((csr_rowpat_kernel<T, E>), dim3(grid), dim3(BLK), lds, s, a); break;
    switch (epi) {
        PAMG_RP(EPI_SET) PAMG_RP(EPI_ACC) PAMG_RP(EPI_RESID) PAMG_RP(EPI_AXPBY) PAMG_RP(EPI_ACC_AXPBY) PAMG_RP(EPI_SUMSQ)
        PAMG_RP(EPI_ACCSEQ) PAMG_RP(EPI_JACOBI) PAMG_RP(EPI_JACOBI_B)
        default: return 1;
    }
#undef PAMG_RP
    return (int)hipGetLastError();
}

// row-mask form (csr_rowmask_kernel): the whole operator, one row per lane; 1 = not available for this launch
template <typename T>
int launch_rowmask(int epi, const pamg_matrix_s *A, hipStream_t s, const StreamArgs<T> &a, int64_t row0 = 0, int64_t row1 = -1)
{
    if (!A->d_pmask || A->rm_nu < 1 || A->rm_nu > 8) return 1;
    if (row1 < 0) row1 = A->nrows;
    if (row0 < 0 || row1 <= row0 || row1 > A->nrows) return 1;
    const int64_t nwin = row1 - row0;                              // rows of this launch: all of them, or a window (interior rows of a shard)
    RowMaskArgs<T> m;
    m.mask = A->d_pmask;
    for (int k = 0; k < 8; ++k) {
        m.off[k] = A->rm_off[k];
        if constexpr (sizeof(T) == 8) std::memcpy(&m.val[k], &A->rm_val[k], 8);
        else { const unsigned v = (unsigned)A->rm_val[k]; std::memcpy(&m.val[k], &v, 4); }
    }
    m.nrows = (int)row1; m.ncols = (int)A->ncols; m.row0 = (int)row0;
    m.xcd_chunk = 0; m.xcd_share = 0;
    if (epi == EPI_SUMSQ) {                                        // one partial per row range, as in every other form
        if (a.blkmap || !a.partial || !a.blkmeta || nwin != A->nrows) return 1;
        const int nu_ = A->rm_nu <= 3 ? 3 : A->rm_nu <= 5 ? 5 : A->rm_nu <= 7 ? 7 : 8;
        if (nu_ == 3) hipLaunchKernelGGL((csr_rowmask_sumsq_kernel<T, 3>), dim3(a.nblk), dim3(BLK), 0, s, a, m);
        else if (nu_ == 5) hipLaunchKernelGGL((csr_rowmask_sumsq_kernel<T, 5>), dim3(a.nblk), dim3(BLK), 0, s, a, m);
        else if (nu_ == 7) hipLaunchKernelGGL((csr_rowmask_sumsq_kernel<T, 7>), dim3(a.nblk), dim3(BLK), 0, s, a, m);
        else hipLaunchKernelGGL((csr_rowmask_sumsq_kernel<T, 8>), dim3(a.nblk), dim3(BLK), 0, s, a, m);
        return (int)hipGetLastError();
    }
    // a 7-point lattice whose extents fit the 64 x 4 x kz tile: csr_rowmask3d_kernel (a window must start on a plane; its planes
    // per lane: the largest of kz, kz / 2, ... that divides its planes)
    if (A->use_rowpat == 1) {
        RowMaskLattice g;
        int grid3 = 0;
        int kz = A->rowmask_kz;
        const int plane_ = A->rm_nu == 7 ? A->rm_off[6] : 0;
        while (kz > 2 && plane_ > 0 && nwin % plane_ == 0 && (nwin / plane_) % kz != 0) kz >>= 1;
        if ((plane_ <= 0 || row0 % plane_ == 0) && rowmask_lattice_plan(A->rm_nu, A->rm_off, nwin, kz, (A->rowmask_flags & 2) != 0, false, g, grid3)) {
// (eight lattice lines per workgroup, 512 lanes: measured -1 % at 256^3 and +2 % at 512^3, profiles/r04_microbench_rowmask_512.json -- the plan and
// the CPU replay keep the option, the kernels are instantiated for four)
#define PAMG_R3W(E, KZ, NT_) hipLaunchKernelGGL((csr_rowmask3d_kernel<T, E, KZ, NT_, 4>), dim3(grid3), dim3(BLK), 0, s, a, m, g);
#define PAMG_R3K(E, KZ) PAMG_R3W(E, KZ, true)
#define PAMG_R3(E) case E: if (kz == 2) { PAMG_R3K(E, 2) } else if (kz == 4) { PAMG_R3K(E, 4) } else { PAMG_R3K(E, 8) } return (int)hipGetLastError();
            switch (epi) {
                PAMG_R3(EPI_SET) PAMG_R3(EPI_ACC) PAMG_R3(EPI_RESID) PAMG_R3(EPI_AXPBY) PAMG_R3(EPI_ACC_AXPBY)
                PAMG_R3(EPI_ACCSEQ) PAMG_R3(EPI_JACOBI) PAMG_R3(EPI_JACOBI_B)
                default: break;
            }
#undef PAMG_R3
#undef PAMG_R3K
#undef PAMG_R3W
        }
    }
    int grid = (int)((nwin + BLK - 1) / BLK);
    const int plane = A->rm_off[A->rm_nu - 1];                   // the largest offset: rows per plane of a lattice
    if ((A->rowmask_flags & 2) && plane >= 8 * BLK && plane % (8 * BLK) == 0 && nwin % plane == 0) m.xcd_share = plane / (8 * BLK);
    else if (A->rowmask_flags & 4) { m.xcd_chunk = (grid + 7) >> 3; grid = 8 * m.xcd_chunk; }
    const int nu = A->rm_nu <= 3 ? 3 : A->rm_nu <= 5 ? 5 : A->rm_nu <= 7 ? 7 : 8;
    // (the single-use streams -- mask, b, result -- are always nontemporal since round 5: 0.112 -> 0.101 ms on the 256^3 residual; flag bit 0 is ignored)
#define PAMG_RMV(E, N) hipLaunchKernelGGL((csr_rowmask_kernel<T, E, N, true>), dim3(grid), dim3(BLK), 0, s, a, m);
#define PAMG_RM(E) case E: \
        if (nu == 3) { PAMG_RMV(E, 3) } else if (nu == 5) { PAMG_RMV(E, 5) } else if (nu == 7) { PAMG_RMV(E, 7) } else { PAMG_RMV(E, 8) } \
        break;
    switch (epi) {
        PAMG_RM(EPI_SET) PAMG_RM(EPI_ACC) PAMG_RM(EPI_RESID) PAMG_RM(EPI_AXPBY) PAMG_RM(EPI_ACC_AXPBY)
        PAMG_RM(EPI_ACCSEQ) PAMG_RM(EPI_JACOBI) PAMG_RM(EPI_JACOBI_B)
        default: return 1;
    }
#undef PAMG_RMV
#undef PAMG_RM
    return (int)hipGetLastError();
}

// row-gather form (csr_rowgather_kernel): operators streaming value codes, no over-long rows
template <typename T>
int launch_rowgather(int epi, int grid, int cap, int nvd, hipStream_t s, const StreamArgs<T> &a)
{
    const int lds = std::max((int)(((3 * (size_t)(cap + 16) + 15) & ~(size_t)15) + sizeof(T) * (size_t)((nvd + 1) & ~1)), (int)(BLK * sizeof(double)));
#define PAMG_RG(E) case E: hipLaunchKernelGGL((csr_rowgather_kernel<T, E>), dim3(grid), dim3(BLK), lds, s, a); break;
    switch (epi) {
        PAMG_RG(EPI_SET) PAMG_RG(EPI_ACC) PAMG_RG(EPI_RESID) PAMG_RG(EPI_AXPBY) PAMG_RG(EPI_ACC_AXPBY) PAMG_RG(EPI_SUMSQ)
        PAMG_RG(EPI_ACCSEQ) PAMG_RG(EPI_JACOBI) PAMG_RG(EPI_JACOBI_B)
        default: return 1;
    }
#undef PAMG_RG
    return (int)hipGetLastError();
}

template <typename T>
int launch_any(int epi, int npl, int grid, int lds, hipStream_t s, const StreamArgs<T> &a)
{
    switch (epi) {
        case EPI_SET: return launch_epi<T, EPI_SET>(npl, grid, lds, s, a);
        case EPI_ACC: return launch_epi<T, EPI_ACC>(npl, grid, lds, s, a);
        case EPI_RESID: return launch_epi<T, EPI_RESID>(npl, grid, lds, s, a);
        case EPI_AXPBY: return launch_epi<T, EPI_AXPBY>(npl, grid, lds, s, a);
        case EPI_ACC_AXPBY: return launch_epi<T, EPI_ACC_AXPBY>(npl, grid, lds, s, a);
        case EPI_SUMSQ: return launch_epi<T, EPI_SUMSQ>(npl, grid, lds, s, a);
        case EPI_ACCSEQ: return launch_epi<T, EPI_ACCSEQ>(npl, grid, lds, s, a);
        case EPI_JACOBI: return launch_epi<T, EPI_JACOBI>(npl, grid, lds, s, a);
        case EPI_JACOBI_B: return launch_epi<T, EPI_JACOBI_B>(npl, grid, lds, s, a);
        case EPI_GS: return launch_epi<T, EPI_GS>(npl, grid, lds, s, a);
        case EPI_GS_B: return launch_epi<T, EPI_GS_B>(npl, grid, lds, s, a);
        case EPI_SOR: return launch_epi<T, EPI_SOR>(npl, grid, lds, s, a);
        case EPI_JACOBI_IDX: return launch_epi<T, EPI_JACOBI_IDX>(npl, grid, lds, s, a);
    }
    return PAMG_E_ARG;
}

template <typename F>
void parallel_rows(int n, F fn)
{
    // n counts rows or row ranges (of ~1.5 K entries): a few hundred of either are worth a thread
    const unsigned hw = std::max(1u, std::min(48u, pamg::host_cpus()));
    const int nt = (n < 512) ? 1 : (int)std::min<unsigned>(hw, (unsigned)(n / 256));
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int lo = (int)((int64_t)n * t / nt), hi = (int)((int64_t)n * (t + 1) / nt);
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}


// 16-bit column codes for the whole-operator kernels: per row range up to four windows of 16 K columns (greedy over the
// range's sorted columns); an entry becomes window << 14 | (column - window base).  All-or-nothing per operator: one
// range that needs a fifth window keeps the operator on 32-bit columns.
int plan_idx16(pamg_matrix_s *A, const std::vector<int4> &blk)
{
    PhaseTimer pt_("plan_idx16", A->nnz);
    if (A->d_Aj16) { hipFree(A->d_Aj16); A->d_Aj16 = nullptr; }
    if (A->d_wbase) { hipFree(A->d_wbase); A->d_wbase = nullptr; }
    if (A->npl != 2 || A->nnz == 0 || A->d_rowid) return PAMG_OK;
    const int nb = (int)blk.size();
    std::vector<int4> wb((size_t)nb);
    std::vector<unsigned short> code((size_t)A->nnz + 16, 0);
    const int *Aj = A->h_Aj.data();
    std::atomic<int> ok(1);
    parallel_rows(nb, [&](int lo, int hi) {
        std::vector<int> c;
        for (int b = lo; b < hi && ok.load(std::memory_order_relaxed); ++b) {
            int base[4];
            if (!plan_range_windows(Aj, blk[b].z, blk[b].w, base, code.data(), c)) { ok = 0; break; }
            wb[(size_t)b] = make_int4(base[0], base[1], base[2], base[3]);
        }
    });
    if (!ok.load()) return PAMG_OK;
    size_t bytes = 0;
    PAMG_TRY(upload_raw((void **)&A->d_Aj16, code.data(), code.size(), sizeof(unsigned short), &bytes));
    PAMG_TRY(upload_raw((void **)&A->d_wbase, wb.data(), wb.size(), sizeof(int4), &bytes));
    return PAMG_OK;
}

// Row patterns / value codes of the whole-operator kernels: the plans are plain host code (pamg_stream_plan.h, replayed
// on the CPU by tests/stream_emul.cpp); here they are run on the operator's host arrays and shipped.
void drop_rowpat(pamg_matrix_s *A)
{
    if (A->d_pid) { hipFree(A->d_pid); A->d_pid = nullptr; }
    if (A->d_ptab) { hipFree(A->d_ptab); A->d_ptab = nullptr; }
    if (A->d_pmask) { hipFree(A->d_pmask); A->d_pmask = nullptr; }
    A->npat = 0; A->pat_lmax = 0; A->rm_nu = 0;
}

template <typename U>
static int plan_rowpat(pamg_matrix_s *A, const unsigned char *code, const U *dict)
{
    PhaseTimer pt_("plan_rowpat", A->nnz);
    drop_rowpat(A);
    // square operators, and row shards in local numbering [owned | halo]: their owned block sits on the diagonal, so interior
    // rows keep the stencil's lists; rows with halo columns come out as irregular rows
    if (A->ncols < A->nrows || A->R != 1 || A->C != 1 || A->nrows < 4096) return PAMG_OK;
    std::vector<unsigned char> pid;
    std::vector<RowPatKey> keys;
    int lmax = 0;
    if (!plan_row_patterns(A->nrows, A->h_Ap.data(), A->h_Aj.data(), code, sizeof(U), pid, keys, lmax)) return PAMG_OK;
    // device table: [256] lengths | [npat * lmax] offsets | [npat * lmax] values
    const int np_ = (int)keys.size();
    const size_t tab_bytes = 256 * sizeof(int) + (size_t)np_ * lmax * (sizeof(int) + sizeof(U));
    std::vector<unsigned char> tab(tab_bytes + 16, 0);
    int *tl = reinterpret_cast<int *>(tab.data());
    int *to = tl + 256;
    U *tv = reinterpret_cast<U *>(to + (size_t)np_ * lmax);
    for (int q = 0; q < np_; ++q) {
        tl[q] = keys[(size_t)q].len;
        for (int j = 0; j < keys[(size_t)q].len; ++j) {
            to[(size_t)q * lmax + j] = keys[(size_t)q].off[j];
            tv[(size_t)q * lmax + j] = dict[keys[(size_t)q].vc[j]];
        }
    }
    size_t bytes = 0;
    PAMG_TRY(upload_raw((void **)&A->d_pid, pid.data(), pid.size(), 1, &bytes));
    PAMG_TRY(upload_raw(&A->d_ptab, tab.data(), tab.size(), 1, &bytes));
    A->npat = np_;
    A->pat_lmax = lmax;
    // the same rows as masks over the longest list, when the lists allow it (csr_rowmask_kernel)
    RowMaskPlan M;
    std::vector<unsigned char> mask;
    if (A->ncols >= A->nrows && plan_row_masks(A->nrows, pid, keys, M, mask)) {
        PAMG_TRY(upload_raw((void **)&A->d_pmask, mask.data(), mask.size(), 1, &bytes));
        A->rm_nu = M.nu;
        A->rm_walked = M.walked;
        for (int k = 0; k < 8; ++k) { A->rm_off[k] = k < M.nu ? M.off[k] : 0; A->rm_val[k] = k < M.nu ? (unsigned long long)dict[M.vc[k]] : 0ull; }
    }
    A->bytes += bytes;
    return PAMG_OK;
}

// 8-bit value codes for the whole-operator kernels: an operator with at most 256 distinct values (bit patterns: +0 and
// -0, NaN payloads stay apart) -- the stencils of the gallery: 2 values -- streams one byte per value instead of eight;
// the kernel looks the value up in an LDS copy of the dictionary, so the product is formed from the very same bits.
// v: the scalar view's values on the host, in storage order.  Independent of the row-range plan.
template <typename U>
static int plan_val8_t(pamg_matrix_s *A, const U *v)
{
    std::vector<U> dict;
    std::vector<unsigned char> code;
    if (!plan_value_codes(A->nnz, v, dict, code)) return PAMG_OK;
    A->nvdict = (int)dict.size();
    dict.resize(256, 0);
    size_t bytes = 0;
    PAMG_TRY(upload_raw((void **)&A->d_Ax8, code.data(), code.size(), 1, &bytes));
    PAMG_TRY(upload_raw(&A->d_vdict, dict.data(), dict.size(), sizeof(U), &bytes));
    A->bytes += bytes;
    return plan_rowpat<U>(A, code.data(), dict.data());
}

void drop_val8(pamg_matrix_s *A)
{
    if (A->d_Ax8) { hipFree(A->d_Ax8); A->d_Ax8 = nullptr; }
    if (A->d_vdict) { hipFree(A->d_vdict); A->d_vdict = nullptr; }
    A->nvdict = 0;
    drop_rowpat(A);
}

int plan_val8(pamg_matrix_s *A, const void *vals)
{
    PhaseTimer pt_("plan_val8", A->nnz);
    drop_val8(A);
    if (!vals || A->nnz < (1 << 16) || A->d_rowid) return PAMG_OK;      // small operators: nothing to gain
    if (A->dtype == PAMG_F64) return plan_val8_t<uint64_t>(A, (const uint64_t *)vals);
    return plan_val8_t<uint32_t>(A, (const uint32_t *)vals);
}

int replan(pamg_matrix_s *A)
{
    PhaseTimer pt_("replan (incl. idx16)", A->nnz);
    if (A->d_blkmeta) { hipFree(A->d_blkmeta); A->d_blkmeta = nullptr; }
    if (A->d_partial) { hipFree(A->d_partial); A->d_partial = nullptr; }
    for (int k = 0; k < 2; ++k) { if (A->d_part[k]) { hipFree(A->d_part[k]); A->d_part[k] = nullptr; } A->npart[k] = 0; }
    A->part_cols = -1;               // an interior / boundary split refers to the old plan
    std::vector<int4> blk;
    blk.reserve((size_t)A->nnz / std::max(1, A->cap / 2) + 16);
    plan_rows(A->h_Ap.data(), 0, (int)A->nrows, A->cap, A->max_rows, blk);
    A->nblk = (int)blk.size();
    PAMG_TRY(upload(&A->d_blkmeta, blk.data(), blk.size(), nullptr));
    if (A->d_bmeta) { hipFree(A->d_bmeta); A->d_bmeta = nullptr; A->bnblk = 0; }
    if (A->R > 1 && A->R == A->C && !A->h_bAp.empty()) {
        std::vector<int4> bb;
        plan_rows(A->h_bAp.data(), 0, A->n_brow, std::max(1, A->cap / A->R), BLK, bb);
        A->bnblk = (int)bb.size();
        PAMG_TRY(upload(&A->d_bmeta, bb.data(), bb.size(), nullptr));
    }
    PAMG_TRY(plan_idx16(A, blk));
    PAMG_HIP(hipMalloc((void **)&A->d_partial, sizeof(double) * (size_t)(A->nblk + 264)));
    return PAMG_OK;
}

void free_tile_part(TileSched *t);

void free_schedule(GsSchedule *g)
{
    if (!g) return;
    free_tile_part(g->tile);
    free_lane_part(g->lane);
    free_lanem_part(g->lanem);
    free_blane_part(g->blane);
    free_line_part(g->line);
    hipFree(g->d_Ap); hipFree(g->d_Aj); hipFree(g->d_Ax); hipFree(g->d_rid); hipFree(g->d_blkmeta); hipFree(g->d_diag); hipFree(g->d_level_blk); hipFree(g->d_sync); hipFree(g->d_xs); hipFree(g->d_xold); hipFree(g->d_pblk); hipFree(g->d_dpos); hipFree(g->d_prof);
    delete g;
}

// Dependency levels of the sweep i = row_start, row_start+row_step, ... (!= row_stop) over
// the pattern (Ap, Aj) with n rows.  Row i must run strictly after every connected row
// visited before it (it reads that row's NEW value) and strictly before every connected
// row visited after it (it reads that row's OLD value) -- "connected" through a stored
// entry in either direction, so non-symmetric patterns are handled too.  Rows of one level
// never touch each other's unknowns.  Output: `order` = visited rows sorted by level
// (visit order inside a level), `lptr` = [nlevels+1] offsets into order.
int analyse_levels(int n, const int *Ap, const int *Aj, int row_start, int row_stop, int row_step,
                   std::vector<int> &order, std::vector<int> &lptr, std::vector<int> *vis_out = nullptr)
{
    if (row_step == 0) return PAMG_E_ARG;
    const long span = (long)row_stop - row_start;
    if (span % row_step != 0 || span / row_step < 0) return PAMG_E_ARG;
    const int m = (int)(span / row_step);
    order.clear();
    lptr.assign(1, 0);
    if (vis_out) vis_out->assign(n, -1);
    if (m == 0) return PAMG_OK;
    const long last = (long)row_start + (long)(m - 1) * row_step;
    if (row_start < 0 || row_start >= n || last < 0 || last >= n) return PAMG_E_ARG;
    std::vector<int> vis(n, -1), lvl(n, 0), pend(n, 0);
    for (int t = 0; t < m; ++t) vis[row_start + t * row_step] = t;
    int maxl = 0;
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step;
        int L = pend[i];
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i || j < 0 || j >= n) continue;
            const int tj = vis[j];
            if (tj >= 0 && tj < t) L = std::max(L, lvl[j] + 1);
        }
        lvl[i] = L;
        maxl = std::max(maxl, L);
        for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
            const int j = Aj[p];
            if (j == i || j < 0 || j >= n) continue;
            if (vis[j] > t) pend[j] = std::max(pend[j], L + 1);
        }
    }
    const int nl = maxl + 1;
    lptr.assign(nl + 1, 0);
    for (int t = 0; t < m; ++t) lptr[lvl[row_start + t * row_step] + 1]++;
    for (int l = 0; l < nl; ++l) lptr[l + 1] += lptr[l];
    order.resize(m);
    std::vector<int> cur(lptr.begin(), lptr.end() - 1);
    for (int t = 0; t < m; ++t) {
        const int i = row_start + t * row_step;
        order[cur[lvl[i]]++] = i;
    }
    if (vis_out) vis_out->swap(vis);
    return PAMG_OK;
}

// true iff for every stored (i,j), i != j, both swept, (j,i) is stored too
bool pattern_symmetric(int n, const int *Ap, const int *Aj, const std::vector<int> &vis)
{
    std::atomic<bool> ok(true);
    parallel_rows(n, [&](int lo, int hi) {
        for (int i = lo; i < hi && ok.load(std::memory_order_relaxed); ++i) {
            if (vis[i] < 0) continue;
            for (int p = Ap[i]; p < Ap[i + 1]; ++p) {
                const int j = Aj[p];
                if (j == i || j < 0 || j >= n || vis[j] < 0) continue;
                bool found = false;
                for (int q = Ap[j]; q < Ap[j + 1]; ++q)
                    if (Aj[q] == i) { found = true; break; }
                if (!found) { ok.store(false); return; }
            }
        }
    });
    return ok.load();
}

// A scalar schedule starts as the ANALYSIS only (visit index and dependency level of every row, symmetry of the
// swept pattern, the hand-off buffers); the device copies of the operator are built on demand for the scheduler
// that is going to run: the level-permuted copy (build_level_part) or the tile-major copy (build_tile_part).
int new_schedule_scalar(pamg_matrix_s *A, int row_start, int row_stop, int row_step, GsSchedule **out)
{
    GsSchedule *g = new (std::nothrow) GsSchedule();
    if (!g) return PAMG_E_ALLOC;
    PhaseTimer pt_("schedule analysis", A->nnz);
    int m = 0, nl = 0;
    if (sweep_levels((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), row_start, row_stop, row_step, g->h_vis, g->h_lvl, m, nl)) {
        delete g;
        return PAMG_E_ARG;
    }
    g->row_start = row_start; g->row_stop = row_stop; g->row_step = row_step;
    g->nlevels = nl;
    g->nrows = m;
    g->symmetric = pattern_symmetric((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), g->h_vis);
    const size_t ts = tsize(A->dtype);
    const size_t xb = ((size_t)A->nrows + 8) * ts;     // hand-off buffer of the granular / tiled sweeps
    int st = (int)hipMalloc(&g->d_xs, xb);
    if (!st) g->bytes += xb;
    if (!st && !g->symmetric) {                        // + snapshot of x (allocated here: sweeps may run inside a graph capture)
        st = (int)hipMalloc(&g->d_xold, xb);
        if (!st) g->bytes += xb;
    }
    if (!st) st = (int)hipMalloc((void **)&g->d_sync, 2048);
    if (!st) st = (int)hipMemset(g->d_sync, 0, 2048);
    if (st) { free_schedule(g); return st; }
    *out = g;
    return PAMG_OK;
}

// level-permuted copy of the operator: the rows of each dependency level stored contiguously
int build_level_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->has_level_part) return PAMG_OK;
    PhaseTimer pt_("build_level_part", A->nnz);
    const int m = (int)g->nrows;
    const std::vector<int> &vis = g->h_vis;
    std::vector<int> lptr((size_t)g->nlevels + 1, 0), order((size_t)m);
    for (int t = 0; t < m; ++t) lptr[g->h_lvl[g->row_start + t * g->row_step] + 1]++;
    for (int l = 0; l < g->nlevels; ++l) lptr[l + 1] += lptr[l];
    {
        std::vector<int> cur(lptr.begin(), lptr.end() - 1);
        for (int t = 0; t < m; ++t) {
            const int i = g->row_start + t * g->row_step;
            order[cur[g->h_lvl[i]]++] = i;
        }
    }
    std::vector<int> pAp((size_t)m + 1, 0);
    for (int r = 0; r < m; ++r) pAp[r + 1] = pAp[r] + (A->h_Ap[order[r] + 1] - A->h_Ap[order[r]]);
    g->nnz = pAp[m];
    // permute on the host (one-time setup cost): the values are fetched back from HBM once
    std::vector<int> pAj((size_t)g->nnz);
    const size_t ts = tsize(A->dtype);
    std::vector<unsigned char> hAx((size_t)A->nnz * ts), pAx((size_t)g->nnz * ts);
    if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    parallel_rows(m, [&](int lo, int hi) {
        for (int r = lo; r < hi; ++r) {
            const int i = order[r];
            const int len = A->h_Ap[i + 1] - A->h_Ap[i];
            if (len) {
                std::memcpy(&pAj[pAp[r]], &A->h_Aj[A->h_Ap[i]], (size_t)len * sizeof(int));
                std::memcpy(&pAx[(size_t)pAp[r] * ts], &hAx[(size_t)A->h_Ap[i] * ts], (size_t)len * ts);
            }
        }
    });
    // "early" entries (column's row is visited earlier in this sweep: the NEW value is needed)
    // carry the sign bit of the column id; every sweep kernel masks it off, the granular
    // sweep uses it to know which values to wait for
    parallel_rows(m, [&](int lo, int hi) {
        for (int r = lo; r < hi; ++r) {
            const int i = order[r], ti = vis[i];
            for (int p = pAp[r]; p < pAp[r + 1]; ++p) {
                const int j = pAj[p];
                if (j != i && j >= 0 && j < (int)A->nrows && vis[j] >= 0 && vis[j] < ti) pAj[p] = j | (int)0x80000000u;
                else if (j == i) pAj[p] = j | 0x40000000;      // diagonal: staged as +0 (DIAG_BIT)
            }
        }
    });
    // diagonal of every stored row (last stored entry with j == i wins; 0 = none): carried by
    // the schedule so the sweep does not have to chase it after the LDS scan
    std::vector<unsigned char> pdiag((size_t)m * ts, 0);
    parallel_rows(m, [&](int lo, int hi) {
        for (int r = lo; r < hi; ++r)
            for (int p = pAp[r]; p < pAp[r + 1]; ++p)
                if (pAj[p] == (order[r] | 0x40000000)) std::memcpy(&pdiag[(size_t)r * ts], &pAx[(size_t)p * ts], ts);
    });
    std::vector<int4> blk;
    auto plan = [&](int cap_) {
        blk.clear();
        g->level_blk.assign(1, 0);
        for (int l = 0; l < g->nlevels; ++l) {
            plan_rows(pAp.data(), lptr[l], lptr[l + 1], cap_, std::min(A->max_rows, BLK), blk);
            g->level_blk.push_back((int)blk.size());
        }
    };
    int gcap = A->gs_cap > 0 ? A->gs_cap : (A->cap_from_val8 ? 1536 : A->cap);
    plan(gcap);
    if (A->gs_cap == 0 && (A->gs_mode == 0 || A->gs_mode == 2) && gcap > 512 && A->npl == 2) {
        // Where this schedule will run as the multi-XCD granular sweep (neither narrow enough for one workgroup nor small
        // enough for the one-XCD form) on SA-like rows, finer ranges win: a range waits for the slowest of its early
        // entries, so 512-entry ranges (~16 rows of 31) track the dependency graph more closely than 1536-entry ones
        // (measured on level 1 of the 256^3 hierarchy, profiles/r03_microbench_gs_range_geometry.json: 4.48 vs 5.04 ms).
        const int64_t nblk = (int64_t)blk.size();
        const bool narrow = nblk * 16 <= (int64_t)g->nlevels * A->flow_cap;
        const int64_t per_level = (nblk + g->nlevels - 1) / std::max(1, g->nlevels);
        const bool xcd = A->gran_xcd == 1 || (A->gran_xcd == 0 && per_level <= 4 && A->nrows <= 262144);
        if (!narrow && !xcd && g->nnz >= 16 * (int64_t)m) { gcap = 512; plan(gcap); }
    }
    g->cap = gcap;
    size_t bytes = 0;
    int st = upload(&g->d_Ap, pAp.data(), pAp.size(), &bytes);
    if (!st) st = upload(&g->d_Aj, pAj.data(), pAj.size(), &bytes);
    if (!st) st = upload_raw(&g->d_Ax, pAx.data(), (size_t)g->nnz, ts, &bytes);
    if (!st) st = upload_raw(&g->d_diag, pdiag.data(), (size_t)m, ts, &bytes);
    if (!st) st = upload(&g->d_rid, order.data(), order.size(), &bytes);
    if (!st) st = upload(&g->d_blkmeta, blk.data(), blk.size(), &bytes);
    if (!st) st = upload(&g->d_level_blk, g->level_blk.data(), g->level_blk.size(), &bytes);
    g->nblk_total = (int)blk.size();
    g->max_level_blocks = 0;
    for (int l = 0; l < g->nlevels; ++l)
        g->max_level_blocks = std::max(g->max_level_blocks, g->level_blk[l + 1] - g->level_blk[l]);
    if (st) return st;
    g->bytes += bytes;
    { std::lock_guard<std::mutex> lk(g_sched_mu); A->bytes += bytes; }
    g->has_level_part = true;
    return PAMG_OK;
}

void free_tile_part(TileSched *t)
{
    if (!t) return;
    hipFree(t->d_blocks); hipFree(t->d_tile_step); hipFree(t->d_prof);
    delete t;
}

int tile_occupancy(int dtype, int variant, int lds);
// kernel variants by the gather items a step may carry: {rounds of old items, rounds of hand-off items, steps per
// gather batch}; RO + RG + 2 loads per step, a batch stays below the 63 the memory counter can count
constexpr int TILE_VAR[2][3] = {{4, 2, 3}, {8, 4, 3}};

// step blocks of the tiled sweep (pamg_tile_plan.h) and the launch geometry that goes with them
}  // namespace
namespace pamg {
// compute units of the CURRENT device (a process may drive several: remembered per device id)
int device_cus()
{
    static std::mutex mu;
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64;
    std::lock_guard<std::mutex> lk(mu);
    if (!cus[dev]) {
        hipDeviceProp_t p;
        cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 64;
    }
    return cus[dev];
}
}  // namespace pamg
namespace {

int build_tile_part(pamg_matrix_s *A, GsSchedule *g)
{
    if (g->tile) return PAMG_OK;
    PhaseTimer pt_("build_tile_part", A->nnz);
    const int m = (int)g->nrows;
    const int ts = (int)tsize(A->dtype);
    const int cus = device_cus();
    // tiles: about seven rows of every dependency level per tile, at most four workgroups per CU (every workgroup must
    // be resident for the whole launch)
    int G = A->tile_G;
    if (G <= 0) G = m <= 1024 ? 1 : (int)std::max<int64_t>(1, (int64_t)m / std::max<int64_t>(1, (int64_t)7 * g->nlevels));
    G = std::max(1, std::min(G, 2 * cus));
    int cap_user = A->tile_cap;
    TilePlan P;
    TileGeom geom{1, 1, ts};
    int W = 0, D = 0, Q = 0, wide = 0, lds = 0;
    for (int attempt = 0; ; ++attempt) {
        const int wpc = (G + cus - 1) / cus;              // workgroups per CU
        W = A->tile_W > 0 ? A->tile_W : (wpc > 1 ? 512 : 2048);
        // entries per step: fat steps for the single-workgroup-per-CU form (fewer, longer steps amortise the fixed cost of
        // a step; the LDS slots still have to hold a few of them), lean ones when several workgroups share a CU
        int cap = cap_user > 0 ? cap_user : (wpc > 1 ? 512 : 1024);
        cap = std::min(TILE_MAX_ENTRIES, std::max(cap, A->max_row_len));
        {
            PhaseTimer pt2_("  tile plan attempt", m);
            if (build_tile_plan_from((int)A->nrows, A->h_Ap.data(), A->h_Aj.data(), g->row_start, g->row_step, m, g->nlevels,
                                     g->h_vis, g->h_lvl, G, W, cap, TILE_ROWS, P, A->tile_part))
                return PAMG_E_ARG;
        }
        int mo = 0, mg = 0;
        if (!tile_geometry(P, ts, geom, &mo, &mg)) return PAMG_E_ARG;
        wide = (mo <= 256 && mg <= 128) ? 0 : 1;          // kernel variant
        const int KG = TILE_VAR[wide][2];
        if (wide > 0 && wpc > 1 && attempt <= 5) { G = cus; continue; }
        const int budget = std::min(150 * 1024, (160 * 1024) / wpc - 1024);
        const int fixed = 64 + W * ts;
        D = (budget - fixed) / geom.slot_bytes();
        if (A->tile_D > 0) D = std::min(D, A->tile_D);
        D = std::min(D, 32);
        if (D < 2 * KG + 2) {                              // two gather batches + the compute wave's step + one landing: does not fit -> leaner steps, else fewer workgroups per CU
            if (attempt > 5) return PAMG_E_ARG;
            if (cap > 512 && cap > A->max_row_len) { cap_user = std::max(512, cap / 2); continue; }
            if (wpc == 1) return PAMG_E_ARG;
            G = (wpc - 1) * cus;
            continue;
        }
        Q = std::max(0, std::min(std::min(4, 63 / geom.chunks()), D - 2 * KG - 1));
        if (A->tile_Q >= 0) Q = std::min(Q, A->tile_Q);
        lds = fixed + D * geom.slot_bytes();
        // Every workgroup must be resident.  The occupancy query is known to over-report by one where the SGPR file is the
        // limit (7-8 waves per SIMD, MI355X_MICROARCH.md); this kernel is held to two waves per SIMD by its 255 VGPRs and
        // to wpc workgroups by the LDS it asks for, far from that edge -- and a sweep that does not get its workgroups
        // reports a time-out, upon which the solver repeats the iteration with one launch per dependency level.
        const int occ = tile_occupancy(A->dtype, wide, lds);
        if (occ >= wpc || attempt > 5) {
            if (occ < wpc) return PAMG_E_STATE;
            break;
        }
        G = std::max(1, occ) * cus;
    }
    TileSched *t = new (std::nothrow) TileSched();
    if (!t) return PAMG_E_ALLOC;
    t->G = P.G; t->W = W; t->NCH = geom.NCH; t->NV = geom.NV; t->wide = wide; t->D = D; t->Q = Q; t->lds = lds;
    t->nsteps = (int)P.steps.size();
    t->n_local = P.n_local; t->n_global = P.n_global; t->n_publish = P.n_publish;
    for (const TileStep &s : P.steps) t->max_step_entries = std::max<int64_t>(t->max_step_entries, s.p1 - s.p0);
    std::vector<unsigned char> hAx((size_t)A->nnz * ts), blocks;
    {
        PhaseTimer pt2_("  tile values d2h", A->nnz);
        if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    }
    // the rows' own old values are only needed when a diagonal is missing or zero (or by SOR, which always asks)
    {
        std::vector<unsigned char> hd((size_t)A->nrows * ts);
        if (A->nrows && A->d_diag) PAMG_HIP(hipMemcpy(hd.data(), A->d_diag, (size_t)A->nrows * ts, hipMemcpyDeviceToHost));
        t->xo = A->d_diag ? 0 : 1;
        for (int64_t i = 0; i < A->nrows && !t->xo; ++i) {
            if (ts == 8) { double v; std::memcpy(&v, &hd[(size_t)i * 8], 8); if (!(v != 0.0)) t->xo = 1; }
            else { float v; std::memcpy(&v, &hd[(size_t)i * 4], 4); if (!(v != 0.0f)) t->xo = 1; }
        }
    }
    int bad;
    {
        PhaseTimer pt2_("  tile pack", A->nnz);
        if (A->dtype == PAMG_F64) bad = pack_tile_blocks<double>(P, geom, reinterpret_cast<const double *>(hAx.data()), A->h_Aj.data(), blocks);
        else bad = pack_tile_blocks<float>(P, geom, reinterpret_cast<const float *>(hAx.data()), A->h_Aj.data(), blocks);
    }
    if (bad) { delete t; return PAMG_E_ARG; }
    PhaseTimer pt3_("  tile upload", (int64_t)blocks.size());
    int st = upload_raw((void **)&t->d_blocks, blocks.data(), blocks.size(), 1, &t->bytes);
    if (!st) st = upload(&t->d_tile_step, P.tile_step.data(), P.tile_step.size(), &t->bytes);
    if (st) { free_tile_part(t); return st; }
    g->tile = t;
    g->bytes += t->bytes;
    { std::lock_guard<std::mutex> lk(g_sched_mu); A->bytes += t->bytes; }
    return PAMG_OK;
}

// schedule for the block path (bs > 1): block rows in level order plus, per scheduled block,
// its position in the operator's block arrays and its block column (the values stay where
// they are: a block row is >= 0.5 KB of contiguous values, which streams well as it is)
int build_schedule_block(pamg_matrix_s *A, int row_start, int row_stop, int row_step, GsSchedule **out)
{
    std::vector<int> order, lptr, vis;
    PAMG_TRY(analyse_levels(A->n_brow, A->h_bAp.data(), A->h_bAj.data(), row_start, row_stop,
                            row_step, order, lptr, &vis));
    GsSchedule *g = new (std::nothrow) GsSchedule();
    if (!g) return PAMG_E_ALLOC;
    g->row_start = row_start; g->row_stop = row_stop; g->row_step = row_step;
    g->nlevels = (int)lptr.size() - 1;
    g->nrows = (int64_t)order.size();
    g->symmetric = pattern_symmetric(A->n_brow, A->h_bAp.data(), A->h_bAj.data(), vis);
    const int m = (int)order.size();
    std::vector<int> pAp((size_t)m + 1, 0);
    for (int r = 0; r < m; ++r) pAp[r + 1] = pAp[r] + (A->h_bAp[order[r] + 1] - A->h_bAp[order[r]]);
    g->nnz = pAp[m];
    // block columns carry the same two flags as the scalar schedules: bit 31 "early" (the column's
    // block row is visited earlier in this sweep: its NEW values are needed), bit 30 diagonal block
    std::vector<int> pblk((size_t)g->nnz), pbj((size_t)g->nnz), dpos((size_t)m, -1);
    for (int r = 0; r < m; ++r) {
        const int i = order[r], ti = vis[i];
        int q = pAp[r];
        for (int p = A->h_bAp[i]; p < A->h_bAp[i + 1]; ++p, ++q) {
            const int j = A->h_bAj[p];
            pblk[q] = p;
            pbj[q] = j;
            if (j == i) { pbj[q] = j | 0x40000000; dpos[r] = p; }
            else if (j >= 0 && j < A->n_brow && vis[j] >= 0 && vis[j] < ti) pbj[q] = j | (int)0x80000000u;
        }
    }
    std::vector<int4> blk;
    g->level_blk.assign(1, 0);
    for (int l = 0; l < g->nlevels; ++l) {
        plan_rows(pAp.data(), lptr[l], lptr[l + 1], std::max(1, A->cap / A->R), BLK, blk);
        g->level_blk.push_back((int)blk.size());
    }
    g->nblk_total = (int)blk.size();
    for (int l = 0; l < g->nlevels; ++l)
        g->max_level_blocks = std::max(g->max_level_blocks, g->level_blk[l + 1] - g->level_blk[l]);
    for (const int4 &m4 : blk) {
        g->max_range_rows = std::max(g->max_range_rows, m4.y - m4.x);
        g->max_range_blocks = std::max(g->max_range_blocks, m4.w - m4.z);
    }
    int st = upload(&g->d_rid, order.data(), order.size(), &g->bytes);
    if (!st) st = upload(&g->d_Ap, pAp.data(), pAp.size(), &g->bytes);
    if (!st) st = upload(&g->d_pblk, pblk.data(), pblk.size(), &g->bytes);
    if (!st) st = upload(&g->d_Aj, pbj.data(), pbj.size(), &g->bytes);
    if (!st) st = upload(&g->d_dpos, dpos.data(), dpos.size(), &g->bytes);
    if (!st) st = upload(&g->d_blkmeta, blk.data(), blk.size(), &g->bytes);
    if (!st) st = upload(&g->d_level_blk, g->level_blk.data(), g->level_blk.size(), &g->bytes);
    if (!st) {                                  // hand-off buffer of the granular sweep (+ snapshot of x)
        const size_t xb = ((size_t)A->nrows + 8) * tsize(A->dtype);
        st = (int)hipMalloc(&g->d_xs, xb);
        if (!st) g->bytes += xb;
        if (!st && !g->symmetric) {
            st = (int)hipMalloc(&g->d_xold, xb);
            if (!st) g->bytes += xb;
        }
    }
    if (!st) st = (int)hipMalloc((void **)&g->d_sync, 2048);
    if (!st) st = (int)hipMemset(g->d_sync, 0, 2048);
    if (st) { free_schedule(g); return st; }
    g->has_level_part = true;
    *out = g;
    return PAMG_OK;
}

}  // namespace
namespace pamg {
int get_schedule(pamg_matrix_s *A, int row_start, int row_stop, int row_step, GsSchedule **out)
{
    auto find = [&]() -> GsSchedule * {
        for (int k = 0; k < 4; ++k) {
            GsSchedule *g = A->gs[k];
            if (g && g->row_start == row_start && g->row_stop == row_stop && g->row_step == row_step) return g;
        }
        return nullptr;
    };
    {
        std::lock_guard<std::mutex> lk(g_sched_mu);
        if (GsSchedule *g = find()) { *out = g; return PAMG_OK; }
    }
    GsSchedule *g = nullptr;
    const bool block = (A->R > 1);
    PAMG_TRY(block ? build_schedule_block(A, row_start, row_stop, row_step, &g)
                   : new_schedule_scalar(A, row_start, row_stop, row_step, &g));
    std::lock_guard<std::mutex> lk(g_sched_mu);
    if (GsSchedule *other = find()) { free_schedule(g); *out = other; return PAMG_OK; }
    int slot = -1;
    for (int k = 0; k < 4; ++k) if (!A->gs[k]) { slot = k; break; }
    if (slot < 0) { free_schedule(A->gs[3]); slot = 3; }
    A->gs[slot] = g;
    A->bytes += g->bytes;
    *out = g;
    return PAMG_OK;
}
}  // namespace pamg
namespace {

template <typename T>
StreamArgs<T> base_args(const pamg_matrix_s *A, const void *x, const void *b, void *y, double c,
                        double omega, double *partial)
{
    StreamArgs<T> a;
    a.blkmeta = A->d_blkmeta;
    a.Ap = A->d_Ap;
    a.Aj = A->d_Aj;
    a.Ax = (const T *)A->d_Ax;
    a.rid = A->d_rowid;
    a.diag = (const T *)A->d_diag;
    a.x = (const T *)x;
    a.xs = nullptr;
    a.err = nullptr;
    a.b = (const T *)b;
    a.y = (T *)y;
    a.partial = partial;
    a.c = (T)c;
    a.omega = (T)omega;
    a.cap = A->cap;
    a.nblk = A->nblk;
    a.flags = 0;
    a.nidle = 1;
    a.Aj16 = nullptr;            // set by stream_launch only: the sweeps run on permuted copies with their own column codes
    a.wbase = nullptr;
    a.wb = make_int4(0, 0, 0, 0);
    a.blkmap = nullptr;
    a.Ax8 = nullptr;             // set by stream_launch only, with the 16-bit column stream
    a.vdict = nullptr;
    a.nvd = 0;
    a.pid = nullptr;
    a.ptab = nullptr;
    a.npat = a.lmax = 0;
    return a;
}

double *g_scratch = nullptr;     // 1024 + 8 doubles for the public vec_sumsq
int g_scratch_dev = -1;

}  // namespace

namespace pamg {

// the operator's values are about to change in place (setup kernels rescale a resident operator): its value codes go
void matrix_drop_value_codes(pamg_matrix_s *A) { if (A) drop_val8(A); }

// Row ranges of a row shard in local numbering [owned | halo] cut in two: INTERIOR ranges touch owned columns only
// (they can run while the halo is still in flight), BOUNDARY ranges read at least one halo column.  Two index lists
// into the existing plan -- no operator data is copied.
int matrix_split_ranges(pamg_matrix_s *A, int64_t n_owned_cols)
{
    if (!A || n_owned_cols < 0) return PAMG_E_ARG;
    for (int k = 0; k < 2; ++k) { if (A->d_part[k]) { hipFree(A->d_part[k]); A->d_part[k] = nullptr; } A->npart[k] = 0; A->part_row0[k] = A->part_row1[k] = -1; }
    A->part_cols = -1;
    if (A->nblk == 0 || A->h_Ap.empty()) { A->part_cols = n_owned_cols; return PAMG_OK; }
    std::vector<int4> blk;
    blk.reserve((size_t)A->nblk);
    plan_rows(A->h_Ap.data(), 0, (int)A->nrows, A->cap, A->max_rows, blk);
    if ((int)blk.size() != A->nblk) return PAMG_E_STATE;
    std::vector<unsigned char> bnd(blk.size(), 0);
    const int *Aj = A->h_Aj.data();
    const int lim = (int)std::min<int64_t>(n_owned_cols, INT_MAX);
    parallel_rows((int)blk.size(), [&](int lo, int hi) {
        for (int b = lo; b < hi; ++b) {
            unsigned char any = 0;
            for (int p = blk[b].z; p < blk[b].w && !any; ++p) any = Aj[p] >= lim;
            bnd[(size_t)b] = any;
        }
    });
    std::vector<int> part[2];
    for (int b = 0; b < (int)blk.size(); ++b) part[bnd[(size_t)b] ? 1 : 0].push_back(b);
    for (int k = 0; k < 2; ++k) {
        A->npart[k] = (int)part[k].size();
        PAMG_TRY(upload(&A->d_part[k], part[k].data(), part[k].size(), &A->bytes));
        // consecutive ranges = one window of rows (the interior planes of a slab shard): the row-mask kernels take it without a range list
        A->part_row0[k] = A->part_row1[k] = -1;
        if (!part[k].empty() && part[k].back() - part[k].front() + 1 == (int)part[k].size()) {
            A->part_row0[k] = blk[(size_t)part[k].front()].x;
            A->part_row1[k] = blk[(size_t)part[k].back()].y;
        }
    }
    A->part_cols = n_owned_cols;
    return PAMG_OK;
}

int stream_launch(pamg_matrix_s *A, int epi, const void *x, const void *b, void *y, double c,
                  double omega, double *partial, hipStream_t s)
{
    return stream_launch_part(A, 0, epi, x, b, y, c, omega, partial, s);
}

// part: 0 = every row range, 1 = the interior ranges, 2 = the boundary ranges (matrix_split_ranges)
int stream_launch_part(pamg_matrix_s *A, int part, int epi, const void *x, const void *b, void *y, double c,
                       double omega, double *partial, hipStream_t s)
{
    if (part < 0 || part > 2) return PAMG_E_ARG;
    if (part && A->part_cols < 0) return PAMG_E_STATE;
    if (part) {
        const int n = A->npart[part - 1];
        if (n == 0) return PAMG_OK;
        const bool idx16 = A->use_idx16 && A->d_Aj16 && A->npl == 2;
        const bool val8 = idx16 && A->use_val8 && A->d_Ax8;
        const int lds = lds_bytes(A->dtype, epi, A->cap) + (val8 ? val8_lds(A, epi) : 0) + A->lds_pad;
        const bool rowg = val8 && A->use_rowg && A->max_row_len <= A->cap;
        const bool rowp = val8 && A->use_rowpat && A->d_pid;
        if (A->dtype == PAMG_F64) {
            StreamArgs<double> a = base_args<double>(A, x, b, y, c, omega, partial);
            a.flags = A->stream_flags & ~2;
            a.nblk = n; a.blkmap = A->d_part[part - 1];
            if (idx16) { a.Aj16 = A->d_Aj16; a.wbase = A->d_wbase; if (val8) { a.Ax8 = A->d_Ax8; a.vdict = (decltype(a.vdict))A->d_vdict; a.nvd = A->nvdict; } }
            if (rowp && A->use_rowpat == 1 && epi != EPI_SUMSQ && part == 1 && A->part_row0[0] >= 0) {     // the interior rows as one window
                StreamArgs<double> aw = a;
                aw.blkmap = nullptr;
                const int st = launch_rowmask<double>(epi, A, s, aw, A->part_row0[0], A->part_row1[0]);
                if (st != 1) return st;
            }
            if (rowp) { const int st = launch_rowpat<double>(epi, n, A, s, a); if (st != 1) return st; }
            if (rowg) { const int st = launch_rowgather<double>(epi, n, A->cap, A->nvdict, s, a); if (st != 1) return st; }
            return launch_any<double>(epi, A->npl, n, lds, s, a);
        }
        StreamArgs<float> a = base_args<float>(A, x, b, y, c, omega, partial);
        a.flags = A->stream_flags & ~2;
        a.nblk = n; a.blkmap = A->d_part[part - 1];
        if (idx16) { a.Aj16 = A->d_Aj16; a.wbase = A->d_wbase; if (val8) { a.Ax8 = A->d_Ax8; a.vdict = (decltype(a.vdict))A->d_vdict; a.nvd = A->nvdict; } }
        if (rowp && A->use_rowpat == 1 && epi != EPI_SUMSQ && part == 1 && A->part_row0[0] >= 0) {     // the interior rows as one window
            StreamArgs<float> aw = a;
            aw.blkmap = nullptr;
            const int st = launch_rowmask<float>(epi, A, s, aw, A->part_row0[0], A->part_row1[0]);
            if (st != 1) return st;
        }
        if (rowp) { const int st = launch_rowpat<float>(epi, n, A, s, a); if (st != 1) return st; }
        if (rowg) { const int st = launch_rowgather<float>(epi, n, A->cap, A->nvdict, s, a); if (st != 1) return st; }
        return launch_any<float>(epi, A->npl, n, lds, s, a);
    }
    int lds = lds_bytes(A->dtype, epi, A->cap) + A->lds_pad;        // lds_pad (tune key 36): unused LDS that caps the workgroups per CU of the staged kernel
    const int grid = (A->stream_flags & 2) ? 8 * ((A->nblk + 7) / 8) : A->nblk;
    const bool idx16 = A->use_idx16 && A->d_Aj16 && A->npl == 2;
    const bool val8 = idx16 && A->use_val8 && A->d_Ax8;
    if (val8) lds += val8_lds(A, epi);
    const bool rowg = val8 && A->use_rowg && A->max_row_len <= A->cap;
    const bool rowp = val8 && A->use_rowpat && A->d_pid;
    if (A->dtype == PAMG_F64) {
        StreamArgs<double> a = base_args<double>(A, x, b, y, c, omega, partial);
        a.flags = A->stream_flags;
        if (idx16) { a.Aj16 = A->d_Aj16; a.wbase = A->d_wbase; if (val8) { a.Ax8 = A->d_Ax8; a.vdict = (decltype(a.vdict))A->d_vdict; a.nvd = A->nvdict; } }
        if (rowp && (A->use_rowpat == 1 || A->use_rowpat == 4)) { const int st = launch_rowmask<double>(epi, A, s, a); if (st != 1) return st; }
        if (rowp) { const int st = launch_rowpat<double>(epi, grid, A, s, a); if (st != 1) return st; }
        if (rowg) { const int st = launch_rowgather<double>(epi, grid, A->cap, A->nvdict, s, a); if (st != 1) return st; }
        return launch_any<double>(epi, A->npl, grid, lds, s, a);
    }
    StreamArgs<float> a = base_args<float>(A, x, b, y, c, omega, partial);
    a.flags = A->stream_flags;
    if (idx16) { a.Aj16 = A->d_Aj16; a.wbase = A->d_wbase; if (val8) { a.Ax8 = A->d_Ax8; a.vdict = (decltype(a.vdict))A->d_vdict; a.nvd = A->nvdict; } }
    if (rowp && (A->use_rowpat == 1 || A->use_rowpat == 4)) { const int st = launch_rowmask<float>(epi, A, s, a); if (st != 1) return st; }
    if (rowp) { const int st = launch_rowpat<float>(epi, grid, A, s, a); if (st != 1) return st; }
    if (rowg) { const int st = launch_rowgather<float>(epi, grid, A->cap, A->nvdict, s, a); if (st != 1) return st; }
    return launch_any<float>(epi, A->npl, grid, lds, s, a);
}

// grid ceiling of the granular sweep: its workgroups must be co-resident; (occupancy - 1, at most
// 4) per CU -- the occupancy query can over-report by one per CU (MI355X_MICROARCH.md)
template <typename T>
static int gran2_grid(int epi, int lds)
{
    const int cus = device_cus();
    int nb = 0;
    hipError_t e;
    if (epi == EPI_GS) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gs_gran2_kernel<T, EPI_GS, false>, BLK, (size_t)lds);
    else if (epi == EPI_GS_B) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gs_gran2_kernel<T, EPI_GS_B, false>, BLK, (size_t)lds);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gs_gran2_kernel<T, EPI_SOR, false>, BLK, (size_t)lds);
    if (e != hipSuccess) nb = 2;
    nb = std::max(1, std::min(nb - 1, 4));
    return nb * cus;
}

template <typename T, bool XCD>
static int gran2_launch(int epi, int grid, int lds, hipStream_t s, const GranArgs<T> &ga)
{
    switch (epi) {
        case EPI_GS: hipLaunchKernelGGL((gs_gran2_kernel<T, EPI_GS, XCD>), dim3(grid), dim3(BLK), lds, s, ga); break;
        case EPI_GS_B: hipLaunchKernelGGL((gs_gran2_kernel<T, EPI_GS_B, XCD>), dim3(grid), dim3(BLK), lds, s, ga); break;
        case EPI_SOR: hipLaunchKernelGGL((gs_gran2_kernel<T, EPI_SOR, XCD>), dim3(grid), dim3(BLK), lds, s, ga); break;
        default: return PAMG_E_ARG;
    }
    return (int)hipGetLastError();
}

template <typename T, int EPI>
static int flow1_launch(int npl, int lds, hipStream_t s, const FlowArgs<T> &f)
{
    (void)npl;
    hipLaunchKernelGGL((gs_flow1_kernel<T, EPI, 2>), dim3(1), dim3(BLK), lds, s, f);
    return (int)hipGetLastError();
}

// ---- tiled sweep: launch plumbing
// kernel variants (TILE_VAR); with or without the rows' own old values
template <typename T, int EPI>
static const void *tile_kernel_ptr(int variant, int xo)
{
    if (EPI == EPI_SOR) xo = 1;
    if (variant == 0) return xo ? (const void *)gs_tile_kernel<T, EPI, 4, 2, 3, true> : (const void *)gs_tile_kernel<T, EPI, 4, 2, 3, false>;
    return xo ? (const void *)gs_tile_kernel<T, EPI, 8, 4, 3, true> : (const void *)gs_tile_kernel<T, EPI, 8, 4, 3, false>;
}

template <typename T>
static const void *tile_kernel_any(int epi, int wide, int xo)
{
    switch (epi) {
        case EPI_GS: return tile_kernel_ptr<T, EPI_GS>(wide, xo);
        case EPI_GS_B: return tile_kernel_ptr<T, EPI_GS_B>(wide, xo);
        case EPI_SOR: return tile_kernel_ptr<T, EPI_SOR>(wide, xo);
    }
    return nullptr;
}

}  // namespace pamg

namespace {
// workgroups of the tiled sweep one CU holds (every workgroup of a launch must be resident): the minimum over the
// variants a schedule may launch
int tile_occupancy(int dtype, int wide, int lds)
{
    int best = 1 << 30;
    for (int epi : {(int)EPI_GS, (int)EPI_GS_B, (int)EPI_SOR})
        for (int xo = 0; xo < 2; ++xo) {
            const void *k = dtype == PAMG_F64 ? pamg::tile_kernel_any<double>(epi, wide, xo) : pamg::tile_kernel_any<float>(epi, wide, xo);
            if (!k) return 0;
            if (lds > 48 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 0;
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, TILE_THREADS, (size_t)lds) != hipSuccess) return 0;
            best = std::min(best, nb);
        }
    return best;
}
}  // namespace

namespace pamg {

template <typename T>
static int tile_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s)
{
    TileSched *t = g->tile;
    const size_t ts = tsize(A->dtype);
    const int64_t n = A->nrows;
    TileArgs<T> a;
    a.blocks = t->d_blocks; a.tile_step = t->d_tile_step;
    a.x = (const T *)x; a.xs = (T *)g->d_xs; a.y = (T *)x; a.b = (const T *)b;
    a.err = g->d_sync + 1;
    a.omega = (T)omega; a.W = t->W; a.D = t->D; a.Q = t->Q; a.G = t->G; a.NCH = t->NCH; a.NV = t->NV;
    a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
    if (!g->symmetric) {
        // write-after-read hazards are not ordered by the waits: old values come from a snapshot
        if (!g->d_xold) return PAMG_E_STATE;
        PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
        a.x = (const T *)g->d_xold;
    }
    if (A->gs_prof && !t->d_prof) {
        PAMG_HIP(hipMalloc((void **)&t->d_prof, (size_t)t->nsteps * 8 * sizeof(long long)));
        PAMG_HIP(hipMemset(t->d_prof, 0, (size_t)t->nsteps * 8 * sizeof(long long)));
    }
    a.prof = A->gs_prof ? t->d_prof : nullptr;
    if (t->n_publish) {
        const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
        hipLaunchKernelGGL((fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
        PAMG_HIP(hipGetLastError());
    }
    const void *k = tile_kernel_any<T>(epi, t->wide, t->xo);
    if (!k) return PAMG_E_ARG;
    if (t->lds > 48 * 1024) PAMG_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, t->lds));
    void *args[] = {(void *)&a};
    PAMG_HIP(hipLaunchKernel(k, dim3(t->G), dim3(TILE_THREADS), args, (size_t)t->lds, s));
    return PAMG_OK;
}

// Scheduling policy of the order-exact sweeps (every scheduler gives the same bits):
//  * tiled sweep (gs_mode 5; pamg_tile_kernels.h): one persistent workgroup per contiguous chunk of rows, dependency
//    chains stay in LDS, only tile-crossing edges use the global hand-off;
//  * narrow schedules (<= flow_cap/16 row ranges per level on average, default 2): ONE workgroup
//    walks all ranges with __syncthreads() and cached accesses (1.4-2 us per range);
//  * otherwise the granular sweep: no barriers, the published datum is the flag (1.8-2.7 us per
//    dependency level; was 4-7 us with a grid barrier or a kernel boundary per level).  Small
//    operators (vectors fit one XCD's 4 MB L2, <= 4 ranges per level) keep the hand-off inside one
//    XCD; patterns that are not structurally symmetric read old values from a snapshot of x;
//  * one launch per level only as the fallback (oversized LDS window, 1- or 4-entries-per-lane plans).
static bool tile_eligible(const pamg_matrix_s *A, const GsSchedule *g)
{
    return g->nlevels > 1 && A->max_row_len <= TILE_MAX_ENTRIES && g->d_xs != nullptr;
}

// Where the tiled sweep is the automatic choice (measured on MI355X, profiles/r02_microbench_tile_*.json): schedules with
// wide dependency levels (>= 2048 rows per level on average: the fine levels of 3-D problems; 1.9 vs 2.9 ms per sweep on
// 256^3, 0.47 vs 1.0 ms on 128^3) and tiny ones that fit a single tile (<= 1024 rows: one workgroup, every hand-off in
// LDS; 0.20 vs 0.29 ms).  In between (SA coarse levels: ~1000 rows and 30-70 entries per row and level) the in-order row
// sums dominate a step and the granular / single-workgroup schedulers are as fast or faster.
static bool tile_auto(const pamg_matrix_s *A, const GsSchedule *g)
{
    if (A->tile_default) return true;
    if (g->nrows <= 1024) return true;
    return g->nrows / std::max(1, g->nlevels) >= 2048;
}

static bool want_tiles(const pamg_matrix_s *A, const GsSchedule *g)
{
    if (!tile_eligible(A, g) || g->tile_unfit) return false;
    return A->gs_mode == 5 || (A->gs_mode == 0 && tile_auto(A, g));
}

// Fast order (tune key 24 = 1): the lane-parallel sweep wherever the schedule fits its form; wide schedules (the fine levels of
// 3-D problems) stay with the tiled sweep unless tune key 27 says otherwise.
static bool want_lanes(const pamg_matrix_s *A, const GsSchedule *g)
{
    if (A->gs_order != 1 || !lane_eligible(A, g)) return false;
    if (A->gs_mode != 0) return false;                     // an explicitly chosen exact scheduler is honoured
    if (!A->lane_wide && g->nrows > 1024 && g->nrows / std::max(1, g->nlevels) >= 2048 && tile_eligible(A, g) && !g->tile_unfit) return false;
    return true;
}

// Fast order on banded operators in their natural order (grid stencils: the fine levels): the line-scan sweep
static bool want_lines(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->gs_order == 1 && A->gs_mode == 0 && A->line_scan && line_eligible(A, g);
}

int ensure_parts(pamg_matrix_s *A, GsSchedule *g, bool block_gs);
// Fast order of the BSR point sweep (amg_core::bsr_gauss_seidel, relaxation.h:185-266): block rows in sweep order, the points of a block row
// one after another (backwards in a backward sweep), every point with the newest values of everything before it -- that IS the scalar
// Gauss-Seidel sweep over the flattened rows.  The order-exact block kernels reproduce the reference's order of additions (off-diagonal blocks
// first, the diagonal block last); in fast order the sums are reordered anyway, so the sweep runs on a scalar CSR twin of the operator through
// the lane-parallel / merged / line-scan forms.  Built with the schedules (never inside a capture), once per operator.
static bool point_twin_bounds(const pamg_matrix_s *A, const GsSchedule *g, int &r0, int &r1, int &rs)
{
    const int R = A->R;
    if (g->row_step == 1 && g->row_start == 0 && g->row_stop == A->n_brow) { r0 = 0; r1 = A->n_brow * R; rs = 1; return true; }
    if (g->row_step == -1 && g->row_start == A->n_brow - 1 && g->row_stop == -1) { r0 = A->n_brow * R - 1; r1 = -1; rs = -1; return true; }
    return false;
}

int ensure_point_twin(pamg_matrix_s *A, GsSchedule *g)
{
    int r0, r1, rs;
    if (A->gs_order != 1 || A->gs_mode != 0 || A->R != A->C || A->point_twin_unfit || !point_twin_bounds(A, g, r0, r1, rs)) return PAMG_OK;
    static std::mutex twin_mu;                                 // both sweep directions arrive here at once (run_sched_jobs)
    {
        std::lock_guard<std::mutex> lk(twin_mu);
        if (A->point_twin_unfit) return PAMG_OK;
        if (!A->point_twin) {
            PhaseTimer pt_("point twin of a block operator", A->nnz);
            std::vector<unsigned char> hAx((size_t)A->nnz * tsize(A->dtype));
            if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, hAx.size(), hipMemcpyDeviceToHost));
            pamg_matrix_s *T = nullptr;
            const int st = pamg_matrix_create(&T, A->dtype, PAMG_CSR, (int)A->nrows, (int)A->ncols, 1, 1, A->h_Ap.data(), A->h_Aj.data(), hAx.data());
            if (st != PAMG_OK) { A->point_twin_unfit = true; return st == PAMG_E_UNSUPPORTED || st == PAMG_E_ARG ? PAMG_OK : st; }
            T->gs_order = 1;
            T->lane_L = A->lane_L; T->lane_G = A->lane_G; T->lane_flags = A->lane_flags; T->lane_merge = A->lane_merge; T->line_scan = A->line_scan;
            T->lane_wide = A->lane_wide;
            A->point_twin = T;
            { std::lock_guard<std::mutex> lk2(g_sched_mu); A->bytes += T->bytes; }
        }
    }
    GsSchedule *tg = nullptr;
    PAMG_TRY(get_schedule(A->point_twin, r0, r1, rs, &tg));
    PAMG_TRY(ensure_parts(A->point_twin, tg, false));
    if (!tg->lane && !tg->lanem && !tg->line) {
        // the flattened rows fit none of the fast-order forms: the exact block kernels keep the sweep (the twin stays unused until the operator goes)
        std::lock_guard<std::mutex> lk(twin_mu);
        A->point_twin_unfit = true;
    }
    return PAMG_OK;
}

// device copies the scheduler of choice needs (called before any graph capture through ensure_schedule)
int ensure_parts(pamg_matrix_s *A, GsSchedule *g, bool block_gs)
{
    if (A->R > 1) {
        // fast order of the block Gauss-Seidel sweep (the BSR point sweep keeps the exact kernels)
        if (block_gs && want_blanes(A, g) && !g->blane) {
            const size_t before = g->bytes;
            const int st = build_blane_part(A, g);
            if (st == PAMG_OK) { std::lock_guard<std::mutex> lk(g_sched_mu); A->bytes += g->bytes - before; }
            if (st != PAMG_E_ARG) return st;
            g->blane_unfit = true;                             // block rows too long / padding too wasteful
        }
        if (!block_gs) PAMG_TRY(ensure_point_twin(A, g));
        return PAMG_OK;
    }
    if (want_lines(A, g)) {
        const size_t before = g->bytes;
        const int st = build_line_part(A, g);
        if (st == PAMG_OK) { std::lock_guard<std::mutex> lk(g_sched_mu); A->bytes += g->bytes - before; }
        if (st != PAMG_E_ARG) return st;
        g->line_unfit = true;                              // no coupled runs / rows too long: the other schedulers take it
    }
    if (want_lanes(A, g) && g->lanem) return PAMG_OK;     // the merged form is there (a second call fell through to the level-permuted copy: 0.4 s and 0.8 GB per direction of level 1 at 256^3)
    if (want_lanes(A, g) && !g->lanem && !g->lanem_unfit && !g->lane && lanem_smax(A, g) >= 2) {
        // the merged form first (f64 Gauss-Seidel; an SOR smoother on this operator sets tune key 33 = 1 before its schedules are built)
        const size_t before = g->bytes;
        const int st = build_lanem_part(A, g);
        if (st == PAMG_OK) { std::lock_guard<std::mutex> lk(g_sched_mu); A->bytes += g->bytes - before; return PAMG_OK; }
        if (st != PAMG_E_ARG) return st;
        g->lanem_unfit = true;
    }
    if (want_lanes(A, g) && !g->lanem) {
        const size_t before = g->bytes;
        const int st = build_lane_part(A, g);
        if (st == PAMG_OK) { std::lock_guard<std::mutex> lk(g_sched_mu); A->bytes += g->bytes - before; }
        if (st != PAMG_E_ARG) return st;
        g->lane_unfit = true;                              // rows too long / padding too wasteful: the exact schedulers take it
    }
    if (want_tiles(A, g)) {
        const int st = build_tile_part(A, g);
        if (st != PAMG_E_ARG) return st;
        g->tile_unfit = true;                              // not representable as step blocks: the level schedulers take it
    }
    return build_level_part(A, g);
}

template <typename T>
static int gs_sweep_scalar_t(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b,
                             double omega, hipStream_t s)
{
    PAMG_TRY(ensure_parts(A, g));
    if (want_lines(A, g)) return line_launch(A, g, epi, x, b, omega, s);
    if (want_lanes(A, g)) return lane_launch(A, g, epi, x, b, omega, s);
    if (want_tiles(A, g)) return tile_launch<T>(A, g, epi, x, b, omega, s);
    StreamArgs<T> a = base_args<T>(A, x, b, x, 0.0, omega, nullptr);
    a.Ap = g->d_Ap;
    a.Aj = g->d_Aj;
    a.Ax = (const T *)g->d_Ax;
    a.rid = g->d_rid;
    a.diag = (const T *)g->d_diag;
    if (g->cap > 0) a.cap = g->cap;                      // the level-permuted copy carries its own range size
    const int lds = lds_bytes(A->dtype, epi, a.cap);
    const bool can_persist = lds <= 48 * 1024 && g->nlevels > 1 && A->gs_mode != 1;
    const bool narrow = (int64_t)g->nblk_total * 16 <= (int64_t)g->nlevels * A->flow_cap;
    const bool single = can_persist && (A->gs_mode == 3 || ((A->gs_mode == 0 || A->gs_mode == 5) && narrow));
    const bool granular = can_persist && !single && A->npl == 2 && g->d_xs;
    if (single) {
        FlowArgs<T> f;
        f.s = a;
        f.s.blkmeta = g->d_blkmeta;
        f.level_blk = g->d_level_blk;
        f.nlevels = g->nlevels;
        f.sync = g->d_sync;
        switch (epi) {
            case EPI_GS: return flow1_launch<T, EPI_GS>(A->npl, lds, s, f);
            case EPI_GS_B: return flow1_launch<T, EPI_GS_B>(A->npl, lds, s, f);
            case EPI_SOR: return flow1_launch<T, EPI_SOR>(A->npl, lds, s, f);
            default: return PAMG_E_ARG;
        }
    }
    if (granular) {
        const size_t ts = tsize(A->dtype);
        const int64_t n = A->nrows;
        GranArgs<T> ga;
        ga.s = a;
        ga.s.blkmeta = g->d_blkmeta;
        ga.s.xs = (T *)g->d_xs;
        ga.s.err = g->d_sync + 1;
        ga.nblk = g->nblk_total;
        ga.ticket = g->d_sync + 20;
        ga.s.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(A->nrows, 1 << 20));

        if (!g->symmetric) {
            // write-after-read hazards are not ordered by the waits: old values come from a snapshot
            if (!g->d_xold) return PAMG_E_STATE;
            PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
            ga.s.x = (const T *)g->d_xold;
        }
        if (A->gs_prof && !g->d_prof) {
            PAMG_HIP(hipMalloc((void **)&g->d_prof, (size_t)g->nblk_total * 8 * sizeof(long long)));
            PAMG_HIP(hipMemset(g->d_prof, 0, (size_t)g->nblk_total * 8 * sizeof(long long)));
        }
        ga.prof = A->gs_prof ? g->d_prof : nullptr;
        const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
        hipLaunchKernelGGL((fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
        PAMG_HIP(hipGetLastError());
        // grid: enough workgroups to run ~8 dependency levels ahead (they wait in the poll loop with
        // their operands in registers), not more -- idle pollers load the memory system (measured)
        const int per_level = (g->nblk_total + g->nlevels - 1) / g->nlevels;
        int G = std::max(1, std::min(g->nblk_total, gran2_grid<T>(epi, lds)));
        if (A->gran_cap > 0) G = std::min(G, A->gran_cap);
        else G = std::min(G, std::min(256, std::max(32, 8 * per_level)));
        const bool xcd = A->gran_xcd == 1 || (A->gran_xcd == 0 && per_level <= 4 && n <= 262144);
        if (xcd) {
            // 8x the wanted grid is launched; the workgroups off the home XCD leave at once
            PAMG_HIP(hipMemsetAsync(g->d_sync + 20, 0, 2 * sizeof(unsigned), s));
            return gran2_launch<T, true>(epi, 8 * std::min(G, A->gran_cap > 0 ? A->gran_cap : 96), lds, s, ga);
        }
        return gran2_launch<T, false>(epi, G, lds, s, ga);
    }
    for (int l = 0; l < g->nlevels; ++l) {
        a.blkmeta = g->d_blkmeta + g->level_blk[l];
        a.nblk = g->level_blk[l + 1] - g->level_blk[l];
        PAMG_TRY(launch_any<T>(epi, A->npl, a.nblk, lds, s, a));
    }
    return PAMG_OK;
}

// in-place order-exact sweep.  epi: EPI_GS / EPI_GS_B / EPI_SOR for scalar operators; block
// operators (bs > 1) run the BSR point sweep (amg_core::bsr_gauss_seidel).
int gs_sweep(pamg_matrix_s *A, int epi, void *x, const void *b, double omega, int row_start,
             int row_stop, int row_step, hipStream_t s)
{
    GsSchedule *g = nullptr;
    PAMG_TRY(get_schedule(A, row_start, row_stop, row_step, &g));
    if (A->R == 1) {
        return A->dtype == PAMG_F64 ? gs_sweep_scalar_t<double>(A, g, epi, x, b, omega, s)
                                    : gs_sweep_scalar_t<float>(A, g, epi, x, b, omega, s);
    }
    if (A->gs_order == 1 && A->gs_mode == 0 && !A->point_twin && !A->point_twin_unfit) PAMG_TRY(ensure_point_twin(A, g));   // (a solver built it with its schedules)
    if (A->point_twin && !A->point_twin_unfit && A->gs_order == 1 && A->gs_mode == 0) {
        int r0, r1, rs;
        if (point_twin_bounds(A, g, r0, r1, rs)) return gs_sweep(A->point_twin, EPI_GS, x, b, 1.0, r0, r1, rs, s);      // fast order (the BSR flavour ignores omega, relaxation.py:343-346)
    }
    return block_point_sweep(A, g, x, b, row_step < 0 ? -1 : 1, s);      // pamg_block.hip
}

int ensure_schedule(pamg_matrix_s *A, int row_start, int row_stop, int row_step, bool block_gs)
{
    GsSchedule *g = nullptr;
    PAMG_TRY(get_schedule(A, row_start, row_stop, row_step, &g));
    return ensure_parts(A, g, block_gs);
}

// deterministic sum of n partials -> out[0].  Large n goes through 256 intermediate sums
// stored behind the partials (callers allocate n + 264 doubles) so the tail is not one
// workgroup crawling over tens of thousands of values.
int reduce_partials(const double *partial, int n, double *out, hipStream_t s)
{
    if (n > 8192) {
        double *mid = const_cast<double *>(partial) + n;
        hipLaunchKernelGGL(reduce_mid_kernel, dim3(256), dim3(BLK), 0, s, partial, n, mid);
        hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(BLK), 0, s, (const double *)mid, 256, out);
    } else {
        hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(BLK), 0, s, partial, n, out);
    }
    return (int)hipGetLastError();
}

int vec_sumsq(int dtype, int64_t n, const void *x, double *scratch, double *out, hipStream_t s)
{
    const int grid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + BLK - 1) / BLK));
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_sumsq_kernel<double>), dim3(grid), dim3(BLK), 0, s, (const double *)x, n, scratch);
    else
        hipLaunchKernelGGL((vec_sumsq_kernel<float>), dim3(grid), dim3(BLK), 0, s, (const float *)x, n, scratch);
    PAMG_HIP(hipGetLastError());
    return reduce_partials(scratch, grid, out, s);
}

int vec_dot(int dtype, int64_t n, const void *x, const void *y, double *scratch, double *out, hipStream_t s)
{
    const int grid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + BLK - 1) / BLK));
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_dot_kernel<double>), dim3(grid), dim3(BLK), 0, s, (const double *)x, (const double *)y, n, scratch);
    else
        hipLaunchKernelGGL((vec_dot_kernel<float>), dim3(grid), dim3(BLK), 0, s, (const float *)x, (const float *)y, n, scratch);
    PAMG_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(BLK), 0, s, (const double *)scratch, grid, out);
    return (int)hipGetLastError();
}

// one element per lane, no loop: the access shape that reaches the HBM ceiling (6.3 TB/s; a grid-stride loop over 8 192 workgroups
// gets 5.0 -- profiles/r04_microbench_bw_shapes.json); the kernels keep their loops for vectors beyond 2^31 / 4 elements
static int vgrid(int64_t n) { return (int)std::min<int64_t>(1 << 21, std::max<int64_t>(1, (n + BLK - 1) / BLK)); }

int vec_mul(int dtype, int64_t n, const void *a, const void *b, void *y, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    const int grid = vgrid(n);
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_mul_kernel<double>), dim3(grid), dim3(BLK), 0, s, n, (const double *)a, (const double *)b, (double *)y);
    else
        hipLaunchKernelGGL((vec_mul_kernel<float>), dim3(grid), dim3(BLK), 0, s, n, (const float *)a, (const float *)b, (float *)y);
    return (int)hipGetLastError();
}

void free_line_schedule(LineSchedule *g)
{
    if (!g) return;
    hipFree(g->d_lines); hipFree(g->d_level_ptr); hipFree(g->d_len); hipFree(g->d_lo); hipFree(g->d_ej); hipFree(g->d_ea);
    free_kz_lane_part(g->kzl);
    delete g;
}

// Order-exact schedule of a Kaczmarz-type sweep over the rows of L in the order (start, stop, step): level of
// a line = 1 + the highest level among EARLIER lines that touch one of its indices -- found in O(nnz) with one
// "highest level seen so far" per index (the conflict graph L L^T is never formed).
static int get_line_schedule(pamg_matrix_s *L, int start, int stop, int step, LineSchedule **out)
{
    for (int k = 0; k < 4; ++k) {
        LineSchedule *g = L->ls[k];
        if (g && g->start == start && g->stop == stop && g->step == step) {
            *out = g;
            if (L->gs_order == 1) PAMG_TRY(build_kz_lane_part(L, g));      // the fast order was asked for after the schedule was made
            return PAMG_OK;
        }
    }
    if (step == 0) return PAMG_E_ARG;
    const long span = (long)stop - start;
    if (span % step != 0 || span / step < 0) return PAMG_E_ARG;
    const int m = (int)(span / step), n = (int)L->nrows;
    if (m > 0 && (start < 0 || start >= n || start + (long)(m - 1) * step < 0 || start + (long)(m - 1) * step >= n)) return PAMG_E_ARG;
    std::vector<int> seen((size_t)L->ncols, -1), lvl((size_t)std::max(m, 1));
    int nl = 0;
    for (int t = 0; t < m; ++t) {
        const int i = start + t * step;
        int lv = 0;
        for (int p = L->h_Ap[i]; p < L->h_Ap[i + 1]; ++p) lv = std::max(lv, seen[L->h_Aj[p]] + 1);
        for (int p = L->h_Ap[i]; p < L->h_Ap[i + 1]; ++p) seen[L->h_Aj[p]] = lv;
        lvl[t] = lv;
        nl = std::max(nl, lv + 1);
    }
    LineSchedule *g = new (std::nothrow) LineSchedule();
    if (!g) return PAMG_E_ALLOC;
    g->start = start; g->stop = stop; g->step = step; g->nlevels = nl;
    g->level_ptr.assign((size_t)nl + 1, 0);
    for (int t = 0; t < m; ++t) g->level_ptr[lvl[t] + 1]++;
    for (int l = 0; l < nl; ++l) g->level_ptr[l + 1] += g->level_ptr[l];
    std::vector<int> lines((size_t)std::max(m, 1)), cur(g->level_ptr.begin(), g->level_ptr.end() - (nl ? 1 : 0));
    for (int t = 0; t < m; ++t) lines[cur[lvl[t]]++] = start + t * step;
    int st = upload(&g->d_lines, lines.data(), (size_t)m, &g->bytes);
    if (!st) st = upload(&g->d_level_ptr, g->level_ptr.data(), g->level_ptr.size(), &g->bytes);
    if (!st) {
        // dense slab of the first KZ entries of every scheduled line (kaczmarz_flow1p_kernel)
        const size_t ts = tsize(L->dtype);
        std::vector<unsigned char> hAx((size_t)L->nnz * ts), ea((size_t)std::max(m, 1) * KZ * ts, 0);
        if (L->nnz) st = (int)hipMemcpy(hAx.data(), L->d_Ax, (size_t)L->nnz * ts, hipMemcpyDeviceToHost);
        std::vector<int> len((size_t)std::max(m, 1), 0), lo((size_t)std::max(m, 1), 0), ej((size_t)std::max(m, 1) * KZ, 0);
        for (int k = 0; k < m; ++k) {
            const int i = lines[k];
            lo[k] = L->h_Ap[i];
            len[k] = L->h_Ap[i + 1] - L->h_Ap[i];
            for (int e = 0; e < std::min(len[k], KZ); ++e) {
                ej[(size_t)k * KZ + e] = L->h_Aj[lo[k] + e];
                std::memcpy(&ea[((size_t)k * KZ + e) * ts], &hAx[(size_t)(lo[k] + e) * ts], ts);
            }
        }
        if (!st) st = upload(&g->d_len, len.data(), (size_t)m, &g->bytes);
        if (!st) st = upload(&g->d_lo, lo.data(), (size_t)m, &g->bytes);
        if (!st) st = upload(&g->d_ej, ej.data(), (size_t)m * KZ, &g->bytes);
        if (!st) st = upload_raw(&g->d_ea, ea.data(), (size_t)m * KZ, ts, &g->bytes);
    }
    if (st) { free_line_schedule(g); return st; }
    int slot = -1;
    for (int k = 0; k < 4; ++k) if (!L->ls[k]) { slot = k; break; }
    if (slot < 0) { free_line_schedule(L->ls[3]); slot = 3; }
    L->ls[slot] = g;
    L->bytes += g->bytes;
    *out = g;
    if (L->gs_order == 1) PAMG_TRY(build_kz_lane_part(L, g));
    return PAMG_OK;
}

int ensure_line_schedule(pamg_matrix_s *L, int start, int stop, int step)
{
    LineSchedule *g = nullptr;
    return get_line_schedule(L, start, stop, step, &g);
}

// one directional sweep: one launch per dependency level
int kaczmarz_sweep(pamg_matrix_s *L, bool nr, void *v, const void *b, const void *Dinv, double omega, int start, int stop,
                   int step, void *xout, hipStream_t s)
{
    if (!L || !v || !Dinv || (nr ? !xout : !b)) return PAMG_E_ARG;
    if (L->R != 1 || L->C != 1) return PAMG_E_UNSUPPORTED;
    LineSchedule *g = nullptr;
    PAMG_TRY(get_line_schedule(L, start, stop, step, &g));
    // fast order (tune key 24 = 1): ONE persistent launch, lanes share a line, versioned 16-byte slots hand the vector over (pamg_kz.hip)
    if (L->gs_order == 1 && g->kzl && L->gs_mode == 0) return kz_lane_launch(L, g, nr, v, b, Dinv, omega, xout, s);
    // narrow schedules (on average <= 512 lines per level: 2-D operators): ONE persistent workgroup walks the levels
    // (about 2 us per level instead of a launch per level); gs_mode 1 keeps the per-level launches
    const int m_lines = g->level_ptr.empty() ? 0 : g->level_ptr.back();
    if (g->nlevels > 1 && L->gs_mode != 1 && (int64_t)m_lines <= (int64_t)512 * g->nlevels) {
#define PAMG_KF(T, NRV)                                                                                                       \
        {                                                                                                                     \
            KzSched<T> ks;                                                                                                    \
            ks.lines = g->d_lines; ks.level_ptr = g->d_level_ptr; ks.len = g->d_len; ks.lo = g->d_lo; ks.ej = g->d_ej;        \
            ks.ea = (const T *)g->d_ea; ks.nlevels = g->nlevels;                                                              \
            hipLaunchKernelGGL((kaczmarz_flow1p_kernel<T, NRV>), dim3(1), dim3(BLK), 0, s, ks, L->d_Aj, (const T *)L->d_Ax,    \
                               (T *)v, (const T *)b, (const T *)Dinv, (T)omega, (T *)xout);                                   \
        }
        if (L->dtype == PAMG_F64) { if (nr) PAMG_KF(double, true) else PAMG_KF(double, false) }
        else { if (nr) PAMG_KF(float, true) else PAMG_KF(float, false) }
#undef PAMG_KF
        return (int)hipGetLastError();
    }
    for (int l = 0; l < g->nlevels; ++l) {
        const int first = g->level_ptr[l], count = g->level_ptr[l + 1] - first;
        if (count <= 0) continue;
        const int grid = (count + BLK - 1) / BLK;
        if (L->dtype == PAMG_F64) {
            if (nr) hipLaunchKernelGGL((kaczmarz_level_kernel<double, true>), dim3(grid), dim3(BLK), 0, s, g->d_lines, first, count, L->d_Ap, L->d_Aj, (const double *)L->d_Ax, (double *)v, (const double *)b, (const double *)Dinv, omega, (double *)xout);
            else hipLaunchKernelGGL((kaczmarz_level_kernel<double, false>), dim3(grid), dim3(BLK), 0, s, g->d_lines, first, count, L->d_Ap, L->d_Aj, (const double *)L->d_Ax, (double *)v, (const double *)b, (const double *)Dinv, omega, (double *)xout);
        } else {
            if (nr) hipLaunchKernelGGL((kaczmarz_level_kernel<float, true>), dim3(grid), dim3(BLK), 0, s, g->d_lines, first, count, L->d_Ap, L->d_Aj, (const float *)L->d_Ax, (float *)v, (const float *)b, (const float *)Dinv, (float)omega, (float *)xout);
            else hipLaunchKernelGGL((kaczmarz_level_kernel<float, false>), dim3(grid), dim3(BLK), 0, s, g->d_lines, first, count, L->d_Ap, L->d_Aj, (const float *)L->d_Ax, (float *)v, (const float *)b, (const float *)Dinv, (float)omega, (float *)xout);
        }
        PAMG_HIP(hipGetLastError());
    }
    return PAMG_OK;
}

int vec_scatter(int dtype, int64_t n, const int *idx, const void *src, void *dst, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    const int grid = vgrid(n);
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_scatter_kernel<double>), dim3(grid), dim3(BLK), 0, s, n, idx, (const double *)src, (double *)dst);
    else
        hipLaunchKernelGGL((vec_scatter_kernel<float>), dim3(grid), dim3(BLK), 0, s, n, idx, (const float *)src, (float *)dst);
    return (int)hipGetLastError();
}

int vec_copy_indexed(int dtype, int64_t n, const int *idx, const void *src, void *dst, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    const int grid = vgrid(n);
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_copy_indexed_kernel<double>), dim3(grid), dim3(BLK), 0, s, n, idx, (const double *)src, (double *)dst);
    else
        hipLaunchKernelGGL((vec_copy_indexed_kernel<float>), dim3(grid), dim3(BLK), 0, s, n, idx, (const float *)src, (float *)dst);
    return (int)hipGetLastError();
}

// Row-subset copy of a scalar CSR operator for the indexed smoothers (CF/FC Jacobi): the listed rows,
// in list order, stored contiguously so they stream like any other operator; d_rowid maps a stored row
// back to its row in the parent, the diagonal is the parent's (last stored a_ii of that row).
int matrix_row_subset(pamg_matrix_s *A, const int32_t *rows, int nrows, pamg_matrix_s **out)
{
    if (!A || !out || nrows < 0 || (nrows > 0 && !rows)) return PAMG_E_ARG;
    if (A->flavour != PAMG_CSR || A->R != 1 || A->C != 1 || A->nrows != A->ncols) return PAMG_E_UNSUPPORTED;
    const int n = (int)A->nrows;
    for (int r = 0; r < nrows; ++r) if (rows[r] < 0 || rows[r] >= n) return PAMG_E_ARG;
    const size_t ts = tsize(A->dtype);
    std::vector<unsigned char> hAx((size_t)A->nnz * ts);
    if (A->nnz) PAMG_HIP(hipMemcpy(hAx.data(), A->d_Ax, (size_t)A->nnz * ts, hipMemcpyDeviceToHost));
    pamg_matrix_s *B = new (std::nothrow) pamg_matrix_s();
    if (!B) return PAMG_E_ALLOC;
    B->dtype = A->dtype; B->flavour = PAMG_CSR;
    B->n_brow = nrows; B->n_bcol = (int)A->ncols; B->R = 1; B->C = 1;
    B->nrows = nrows; B->ncols = A->ncols;
    B->h_Ap.assign((size_t)nrows + 1, 0);
    for (int r = 0; r < nrows; ++r) B->h_Ap[r + 1] = B->h_Ap[r] + (A->h_Ap[rows[r] + 1] - A->h_Ap[rows[r]]);
    B->nnz = B->h_Ap[nrows];
    B->h_Aj.assign((size_t)B->nnz + 8, 0);
    std::vector<unsigned char> bAx(((size_t)B->nnz + 8) * ts, 0), bdiag(((size_t)nrows + 8) * ts, 0);
    for (int r = 0; r < nrows; ++r) {
        const int i = rows[r], lo = A->h_Ap[i], len = A->h_Ap[i + 1] - lo, at = B->h_Ap[r];
        if (len) {
            std::memcpy(&B->h_Aj[at], &A->h_Aj[lo], (size_t)len * sizeof(int));
            std::memcpy(&bAx[(size_t)at * ts], &hAx[(size_t)lo * ts], (size_t)len * ts);
        }
        for (int p = lo; p < lo + len; ++p)
            if (A->h_Aj[p] == i) std::memcpy(&bdiag[(size_t)r * ts], &hAx[(size_t)p * ts], ts);     // last one wins
    }
    int st = upload(&B->d_Ap, B->h_Ap.data(), B->h_Ap.size(), &B->bytes);
    if (!st) st = upload(&B->d_Aj, B->h_Aj.data(), B->h_Aj.size(), &B->bytes);
    if (!st) st = upload_raw(&B->d_Ax, bAx.data(), (size_t)B->nnz + 8, ts, &B->bytes);
    if (!st) st = upload_raw(&B->d_diag, bdiag.data(), (size_t)nrows + 8, ts, &B->bytes);
    std::vector<int> rid(rows, rows + nrows);
    rid.resize((size_t)nrows + 8, 0);
    if (!st) st = upload(&B->d_rowid, rid.data(), rid.size(), &B->bytes);
    B->h_Aj.resize((size_t)B->nnz);
    B->cap = 1536; B->npl = 2; B->max_rows = 1024;
    if (!st) st = replan(B);
    if (st) { pamg_matrix_destroy(B); return st; }
    *out = B;
    return PAMG_OK;
}

// amg_core::jacobi_indexed (relaxation.h:382-427) on a row-subset operator: new values of the listed
// rows from the OLD x into work (one value per listed row), then scattered into x
int jacobi_indexed(pamg_matrix_s *sub, void *x, const void *b, double omega, void *work, hipStream_t s)
{
    if (!sub || !sub->d_rowid || !x || !b || !work) return PAMG_E_ARG;
    if (sub->nrows == 0) return PAMG_OK;
    PAMG_TRY(stream_launch(sub, EPI_JACOBI_IDX, x, b, work, 0.0, omega, nullptr, s));
    return vec_scatter(sub->dtype, sub->nrows, sub->d_rowid, work, x, s);
}

int vec_maxratio(int dtype, int64_t n, const void *u, const void *x, double *scratch, double *out, hipStream_t s)
{
    const int grid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + BLK - 1) / BLK));
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_maxratio_kernel<double>), dim3(grid), dim3(BLK), 0, s, (const double *)u, (const double *)x, n, scratch);
    else
        hipLaunchKernelGGL((vec_maxratio_kernel<float>), dim3(grid), dim3(BLK), 0, s, (const float *)u, (const float *)x, n, scratch);
    PAMG_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_max_kernel, dim3(1), dim3(BLK), 0, s, (const double *)scratch, grid, out);
    return (int)hipGetLastError();
}


int vec_axpy(int dtype, int64_t n, double a, const void *x, void *y, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_axpy_kernel<double>), dim3(vgrid(n)), dim3(BLK), 0, s, n, a, (const double *)x, (double *)y);
    else
        hipLaunchKernelGGL((vec_axpy_kernel<float>), dim3(vgrid(n)), dim3(BLK), 0, s, n, (float)a, (const float *)x, (float *)y);
    return (int)hipGetLastError();
}

int vec_scale(int dtype, int64_t n, double a, const void *x, void *y, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_scale_kernel<double>), dim3(vgrid(n)), dim3(BLK), 0, s, n, a, (const double *)x, (double *)y);
    else
        hipLaunchKernelGGL((vec_scale_kernel<float>), dim3(vgrid(n)), dim3(BLK), 0, s, n, (float)a, (const float *)x, (float *)y);
    return (int)hipGetLastError();
}

int vec_xpby(int dtype, int64_t n, double beta, const void *z, void *p, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_xpby_kernel<double>), dim3(vgrid(n)), dim3(BLK), 0, s, n, beta, (const double *)z, (double *)p);
    else
        hipLaunchKernelGGL((vec_xpby_kernel<float>), dim3(vgrid(n)), dim3(BLK), 0, s, n, (float)beta, (const float *)z, (float *)p);
    return (int)hipGetLastError();
}

int vec_axpy_ratio(int dtype, int64_t n, const double *num, const double *den, double sign, const void *x, void *y,
                   hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_axpy_ratio_kernel<double>), dim3(vgrid(n)), dim3(BLK), 0, s, n, num, den, sign, (const double *)x, (double *)y);
    else
        hipLaunchKernelGGL((vec_axpy_ratio_kernel<float>), dim3(vgrid(n)), dim3(BLK), 0, s, n, num, den, (float)sign, (const float *)x, (float *)y);
    return (int)hipGetLastError();
}

int vec_fill(int dtype, int64_t n, double v, void *y, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_fill_kernel<double>), dim3(vgrid(n)), dim3(BLK), 0, s, n, v, (double *)y);
    else
        hipLaunchKernelGGL((vec_fill_kernel<float>), dim3(vgrid(n)), dim3(BLK), 0, s, n, (float)v, (float *)y);
    return (int)hipGetLastError();
}

int dense_gemv(int dtype, int n, const void *M, const void *b, void *x, hipStream_t s)
{
    if (n <= 0) return PAMG_OK;
    const int grid = (n + (BLK / 64) - 1) / (BLK / 64);
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((dense_gemv_kernel<double>), dim3(grid), dim3(BLK), 0, s, n, (const double *)M, (const double *)b, (double *)x);
    else
        hipLaunchKernelGGL((dense_gemv_kernel<float>), dim3(grid), dim3(BLK), 0, s, n, (const float *)M, (const float *)b, (float *)x);
    return (int)hipGetLastError();
}

}  // namespace pamg

// =============================================================================== C ABI
namespace pamg {
int sweep_error(pamg_matrix_s *A, bool *error)
{
    *error = false;
    for (int k = 0; k < 4; ++k) {
        GsSchedule *g = A->gs[k];
        if (!g || !g->d_sync) continue;
        unsigned w[2] = {0, 0};
        PAMG_HIP(hipMemcpy(w, g->d_sync, sizeof(w), hipMemcpyDeviceToHost));
        if (w[1]) {
            *error = true;
            PAMG_HIP(hipMemset(g->d_sync + 1, 0, sizeof(unsigned)));
        }
    }
    for (int k = 0; k < 4; ++k) if (A->ls[k] && kz_lane_error(A->ls[k])) *error = true;
    if (A->point_twin) {
        bool e2 = false;
        PAMG_TRY(sweep_error(A->point_twin, &e2));
        if (e2) *error = true;
    }
    return PAMG_OK;
}
}  // namespace pamg

extern "C" {

int pamg_matrix_create(pamg_matrix_t *out, int dtype, int flavour, int n_brow, int n_bcol, int R,
                       int C, const int32_t *Ap, const int32_t *Aj, const void *Ax)
{
    if (!out || !Ap || n_brow < 0 || n_bcol < 0 || R < 1 || C < 1) return PAMG_E_ARG;
    if (dtype != PAMG_F64 && dtype != PAMG_F32) return PAMG_E_UNSUPPORTED;
    if (flavour != PAMG_CSR && flavour != PAMG_BSR) return PAMG_E_ARG;
    if (flavour == PAMG_CSR && (R != 1 || C != 1)) return PAMG_E_ARG;
    if (R > MAXBS || C > MAXBS) return PAMG_E_UNSUPPORTED;
    const int64_t nblk = Ap[n_brow];
    PhaseTimer pt_("matrix_create (total)", nblk * R * C);
    if (nblk < 0 || (nblk > 0 && (!Aj || !Ax)) || Ap[0] != 0) return PAMG_E_ARG;
    // a malformed operator would mean out-of-bounds device accesses and (with the flag bits the schedules put
    // into column ids) wrong dependency analysis: check the structure once, here
    {
        std::atomic<int> bad(0);
        host_parallel(n_brow, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) if (Ap[i + 1] < Ap[i]) bad = 1; });
        host_parallel(nblk, [&](int64_t lo, int64_t hi) { int b = 0; for (int64_t p = lo; p < hi; ++p) b |= (Aj[p] < 0) | (Aj[p] >= n_bcol); if (b) bad = 1; });
        if (bad.load()) return PAMG_E_ARG;
    }
    // column ids carry two flag bits in the level schedules (pamg_kernels.h: EARLY_BIT, DIAG_BIT)
    if ((int64_t)n_brow * R > (1 << 30) || (int64_t)n_bcol * C > (1 << 30) || nblk * R * C > INT32_MAX)
        return PAMG_E_UNSUPPORTED;
    pamg_matrix_s *A = new (std::nothrow) pamg_matrix_s();
    if (!A) return PAMG_E_ALLOC;
    A->dtype = dtype; A->flavour = flavour;
    A->n_brow = n_brow; A->n_bcol = n_bcol; A->R = R; A->C = C;
    A->nrows = (int64_t)n_brow * R; A->ncols = (int64_t)n_bcol * C; A->nnz = nblk * R * C;
    const size_t ts = tsize(dtype);
    int st = PAMG_OK;
    if (R == 1 && C == 1) {
        A->h_Ap.assign(Ap, Ap + n_brow + 1);
        A->h_Aj.assign(Aj, Aj + nblk);
        st = upload(&A->d_Ap, A->h_Ap.data(), A->h_Ap.size(), &A->bytes);
        if (!st) st = upload(&A->d_Aj, A->h_Aj.data(), A->h_Aj.size(), &A->bytes);
        if (!st) st = upload_raw(&A->d_Ax, Ax, (size_t)nblk, ts, &A->bytes);
        if (!st) st = plan_val8(A, Ax);
        if (!st) {
            // diagonal of every row, found exactly like the reference finds it (last stored
            // entry with j == i wins, 0 when absent: relaxation.h:64-74); carried separately so
            // the smoothers do not have to re-read the value stream to fetch it
            std::vector<unsigned char> dg((size_t)n_brow * ts, 0);
            const unsigned char *src = (const unsigned char *)Ax;
            host_parallel(n_brow, [&](int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i)
                    for (int p = Ap[i]; p < Ap[i + 1]; ++p)
                        if (Aj[p] == (int)i) std::memcpy(&dg[(size_t)i * ts], src + (size_t)p * ts, ts);
            }, 1 << 18);
            st = upload_raw(&A->d_diag, dg.data(), (size_t)n_brow, ts, &A->bytes);
        }
    } else {
        // scalar (flattened) CSR view: scalar row ib*R+r holds, block after block in
        // storage order, the C entries of block row r -- the exact summation order of
        // SciPy's bsr_matvec for that output.  Explicit zeros inside blocks are kept.
        A->h_Ap.resize((size_t)A->nrows + 1);
        A->h_Aj.resize((size_t)A->nnz);
        std::vector<unsigned char> flat((size_t)A->nnz * ts);
        const unsigned char *src = (const unsigned char *)Ax;
        int64_t w = 0;
        for (int ib = 0; ib < n_brow; ++ib)
            for (int r = 0; r < R; ++r) {
                A->h_Ap[(size_t)ib * R + r] = (int)w;
                for (int p = Ap[ib]; p < Ap[ib + 1]; ++p) {
                    for (int c = 0; c < C; ++c) A->h_Aj[(size_t)w + c] = Aj[p] * C + c;
                    std::memcpy(&flat[(size_t)w * ts], src + ((size_t)p * R * C + (size_t)r * C) * ts, (size_t)C * ts);
                    w += C;
                }
            }
        A->h_Ap[(size_t)A->nrows] = (int)w;
        st = upload(&A->d_Ap, A->h_Ap.data(), A->h_Ap.size(), &A->bytes);
        if (!st) st = upload(&A->d_Aj, A->h_Aj.data(), A->h_Aj.size(), &A->bytes);
        if (!st) st = upload_raw(&A->d_Ax, flat.data(), (size_t)A->nnz, ts, &A->bytes);
        if (!st) st = plan_val8(A, flat.data());
        if (R == C && !st) {
            // square blocks: keep the block view too (point / block smoothers)
            A->h_bAp.assign(Ap, Ap + n_brow + 1);
            A->h_bAj.assign(Aj, Aj + nblk);
            A->nblocks_b = nblk;
            st = upload(&A->d_bAp, A->h_bAp.data(), A->h_bAp.size(), &A->bytes);
            if (!st) st = upload(&A->d_bAj, A->h_bAj.data(), A->h_bAj.size(), &A->bytes);
            if (!st) st = upload_raw(&A->d_bAx, Ax, (size_t)A->nnz, ts, &A->bytes);
            // streamed block kernels: diagonal blocks flagged in the column ids (their product is staged
            // as +0) and the position of each row's diagonal block (last stored one wins, like the reference)
            std::vector<int> bjf(A->h_bAj), bdiag((size_t)n_brow, -1);
            for (int i = 0; i < n_brow; ++i)
                for (int p = Ap[i]; p < Ap[i + 1]; ++p)
                    if (Aj[p] == i) { bjf[p] = i | 0x40000000; bdiag[i] = p; }
            if (!st) st = upload(&A->d_bAjf, bjf.data(), bjf.size(), &A->bytes);
            if (!st) st = upload(&A->d_bdiag, bdiag.data(), bdiag.size(), &A->bytes);
        }
    }
    if (!st) {
        std::atomic<int> mx(0);
        host_parallel(A->nrows, [&](int64_t lo, int64_t hi) {
            int m = 0;
            for (int64_t i = lo; i < hi; ++i) m = std::max(m, A->h_Ap[i + 1] - A->h_Ap[i]);
            int cur = mx.load();
            while (m > cur && !mx.compare_exchange_weak(cur, m)) {}
        });
        A->max_row_len = mx.load();
    }
    // default plan (measured best on 256^3 Poisson): 1536 staged entries = 12 KB (SpMV) / 18 KB
    // (smoothers, with column ids) of LDS per workgroup -> 8 workgroups = 32 waves per CU
    A->cap = 1536; A->npl = 2; A->max_rows = 1024;
    // with 8-bit value codes the whole-operator kernels stage four entries per lane in two steps that are in flight
    // together: 2048 entries fill both (measured on the 256^3 stencil: 0.242 ms against 0.262 with 1536); the level
    // schedules of the order-exact sweeps keep 1536
    if (A->d_Ax8 && R == 1 && C == 1) {
        // ... and run as the row-gather kernel (lane = row): ranges of 512 rows = two full trips of the 256 lanes, the LDS
        // window sized for them (3 bytes per entry there).  256^3 stencil: 0.179 ms with 512 rows / 3584 entries against
        // 0.193 (256 rows) and 0.199 (2048 entries, ragged second trip); profiles/r03_microbench_spmv_value_codes.json
        // Small operators keep at least ~2048 ranges in the launch (8 per CU): 64 .. 512 rows per range.
        const int64_t avg = (A->nnz + A->nrows - 1) / std::max<int64_t>(1, A->nrows);
        const int64_t rows = std::min<int64_t>(512, std::max<int64_t>(64, ((A->nrows / 2048 + 63) / 64) * 64));
        A->max_rows = (int)rows;
        A->cap = (int)std::min<int64_t>(12288, std::max<int64_t>(512, ((rows * avg + 255) / 256) * 256));
        A->cap_from_val8 = 1;
        A->use_rowg = 1;
    }
    if (!st) st = replan(A);
    if (!st && A->d_Ax8 && !A->d_Aj16) {
        // value codes need the 16-bit column stream; without it (a range with more than four column windows) the operator
        // runs the staged kernel on its own default plan
        drop_val8(A);
        A->cap = 1536; A->max_rows = 1024; A->cap_from_val8 = 0; A->use_rowg = 0;
        st = replan(A);
    }
    if (st) { pamg_matrix_destroy(A); return st; }
    *out = A;
    return PAMG_OK;
}

int pamg_matrix_destroy(pamg_matrix_t A)
{
    if (!A) return PAMG_OK;
    hipFree(A->d_Ap); hipFree(A->d_Aj); hipFree(A->d_Ax); hipFree(A->d_diag); hipFree(A->d_rowid);
    hipFree(A->d_bAp); hipFree(A->d_bAj); hipFree(A->d_bAjf); hipFree(A->d_bdiag); hipFree(A->d_bAx); hipFree(A->d_blkmeta); hipFree(A->d_partial); hipFree(A->d_bmeta); hipFree(A->d_Aj16); hipFree(A->d_wbase); hipFree(A->d_Ax8); hipFree(A->d_vdict); hipFree(A->d_pid); hipFree(A->d_ptab);
    hipFree(A->d_part[0]); hipFree(A->d_part[1]);
    for (int k = 0; k < 4; ++k) free_schedule(A->gs[k]);
    for (int k = 0; k < 4; ++k) pamg::free_line_schedule(A->ls[k]);
    if (A->point_twin) pamg_matrix_destroy(A->point_twin);
    delete A;
    return PAMG_OK;
}

int pamg_matrix_point_twin(pamg_matrix_t A, int *state)
{
    if (!A || !state) return PAMG_E_ARG;
    *state = A->point_twin_unfit ? 2 : A->point_twin ? 1 : 0;
    return PAMG_OK;
}

int pamg_matrix_info(pamg_matrix_t A, int64_t info[8])
{
    if (!A || !info) return PAMG_E_ARG;
    info[0] = A->nrows; info[1] = A->ncols; info[2] = A->nnz; info[3] = A->nblk;
    info[4] = A->cap; info[5] = (int64_t)A->bytes;
    info[6] = info[7] = 0;
    const int n = A->R > 1 ? A->n_brow : (int)A->nrows;
    for (int k = 0; k < 4; ++k) {
        GsSchedule *g = A->gs[k];
        if (!g) continue;
        if (g->row_start == 0 && g->row_stop == n && g->row_step == 1) info[6] = g->nlevels;
        if (g->row_start == n - 1 && g->row_stop == -1 && g->row_step == -1) info[7] = g->nlevels;
    }
    return PAMG_OK;
}

int pamg_matrix_value_codes(pamg_matrix_t A, int *n_values)
{
    if (!A || !n_values) return PAMG_E_ARG;
    *n_values = (A->d_Ax8 && A->use_val8 && A->use_idx16 && A->d_Aj16 && A->npl == 2) ? A->nvdict : 0;
    return PAMG_OK;
}

int pamg_matrix_row_patterns(pamg_matrix_t A, int *n_patterns)
{
    if (!A || !n_patterns) return PAMG_E_ARG;
    *n_patterns = (A->d_pid && A->use_rowpat && A->d_Ax8 && A->use_val8 && A->use_idx16 && A->d_Aj16 && A->npl == 2) ? A->npat : 0;
    return PAMG_OK;
}

int pamg_matrix_row_masks(pamg_matrix_t A, long long info[8])
{
    if (!A || !info) return PAMG_E_ARG;
    for (int k = 0; k < 8; ++k) info[k] = 0;
    int npat = 0;
    pamg_matrix_row_patterns(A, &npat);
    if (!npat || !A->d_pmask || (A->use_rowpat != 1 && A->use_rowpat != 4)) return PAMG_OK;
    info[0] = A->rm_nu; info[1] = A->rm_walked; info[5] = A->rowmask_kz; info[6] = A->rowmask_flags;
    info[7] = (A->nrows + BLK - 1) / BLK;
    RowMaskLattice g;
    int grid3 = 0;
    if (A->use_rowpat == 1 && rowmask_lattice_plan(A->rm_nu, A->rm_off, A->nrows, A->rowmask_kz, (A->rowmask_flags & 2) != 0, false, g, grid3)) {
        info[2] = 1; info[3] = g.L; info[4] = g.P; info[7] = grid3;
    }
    return PAMG_OK;
}

int pamg_matrix_tune(pamg_matrix_t A, int key, int value)
{
    if (!A) return PAMG_E_ARG;
    // a finalised solver's captured graphs point into the schedules and plans this call would free
    if (key == 21) { A->use_val8 = value != 0; return PAMG_OK; }      // read at launch time only: no plan depends on it
    if (key == 22) { A->use_rowg = value != 0; return PAMG_OK; }      // likewise
    if (key == 23) { if (value < 0 || value > 4 || value == 2) return PAMG_E_ARG; A->use_rowpat = value; return PAMG_OK; }
    if (A->borrowed > 0) return PAMG_E_STATE;
    switch (key) {
        case 0: if (value < 64 || value > 12288) return PAMG_E_ARG; A->cap = value & ~3; A->cap_from_val8 = 0; break;
        case 1: if (value != 2) return PAMG_E_ARG; A->npl = 2; return PAMG_OK;      // 1 and 4 entries per lane: measured no better, retired in round 5
        case 2: if (value < 1) return PAMG_E_ARG; A->max_rows = value; break;
        case 3: if (value < 0 || value > 256) return PAMG_E_ARG; A->flow_cap = value; return PAMG_OK;
        case 5: if (value < 0 || value > 5) return PAMG_E_ARG; A->gs_mode = value; return PAMG_OK;
        case 6: if (value < 0) return PAMG_E_ARG; A->gran_cap = value; return PAMG_OK;
        case 7: if (value < 0 || value > 2) return PAMG_E_ARG; A->gran_xcd = value; return PAMG_OK;
        case 8: if (value < 0 || value > 63) return PAMG_E_ARG; A->stream_flags = value; return PAMG_OK;
        case 11: A->gs_prof = value != 0; return PAMG_OK;
        case 12: if (value < 0) return PAMG_E_ARG; A->tile_G = value; break;
        case 13: if (value != 0 && (value < 64 || value > 8192 || (value & (value - 1)))) return PAMG_E_ARG; A->tile_W = value; break;
        case 14: if (value < 0 || value > TILE_MAX_ENTRIES) return PAMG_E_ARG; A->tile_cap = value; break;
        case 15: A->tile_default = value != 0; return PAMG_OK;
        case 16: if (value < 0 || value > 32) return PAMG_E_ARG; A->tile_D = value; break;
        case 17: if (value < -1 || value > 4) return PAMG_E_ARG; A->tile_Q = value; break;
        case 18: if (value < 0 || value > 1) return PAMG_E_ARG; A->tile_part = value; break;
        case 19: A->use_idx16 = value != 0; return PAMG_OK;
        case 20: if (value != 0 && (value < 64 || value > 2048)) return PAMG_E_ARG; A->gs_cap = value & ~3; break;
        case 24: if (value < 0 || value > 1) return PAMG_E_ARG; A->gs_order = value; return PAMG_OK;
        case 25: if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64) return PAMG_E_ARG; A->lane_L = value; break;
        case 26: if (value < 0) return PAMG_E_ARG; A->lane_G = value; return PAMG_OK;
        case 27: if (value < 0 || value > 1) return PAMG_E_ARG; A->lane_wide = value; return PAMG_OK;
        case 33: if (value < 0 || value > 8) return PAMG_E_ARG; A->lane_merge = value; break;
        case 35: if (value < 0 || value > 2) return PAMG_E_ARG; A->lanem_rpw = value; break;
        case 36: if (value < 0 || value > 98304) return PAMG_E_ARG; A->lds_pad = value & ~15; return PAMG_OK;
        case 34: if (value < 1 || value > 400) return PAMG_E_ARG; A->lanem_ahead10 = value; return PAMG_OK;
        case 30:                                               // 2: also where the estimate favours the lane form
            if (value < 0 || value > 2) return PAMG_E_ARG;
            A->line_scan = value;
            for (int k = 0; k < 4; ++k) if (A->gs[k]) A->gs[k]->line_unfit = false;      // a schedule the planner declined on its estimate is asked again
            return PAMG_OK;
        case 31: if (value != 2 && value != 4 && value != 8) return PAMG_E_ARG; A->rowmask_kz = value; return PAMG_OK;
        case 32: if (value < 0 || value > 7) return PAMG_E_ARG; A->rowmask_flags = value; return PAMG_OK;
        case 28: if (value < 0 || value > 15 || (value & 6)) return PAMG_E_ARG; A->lane_flags = value; return PAMG_OK;      // bits 1, 2: retired (slab form, old values through the L1)
        default: return PAMG_E_ARG;
    }
    if (key == 25 || key == 33 || key == 35) {         // lane geometry / merging: drop the lane parts only
        for (int k = 0; k < 4; ++k) {
            GsSchedule *g = A->gs[k];
            if (g) g->lane_unfit = false;
            if (g && g->lane) { const size_t lb = lane_part_bytes(g); A->bytes -= lb; g->bytes -= lb; free_lane_part(g->lane); g->lane = nullptr; }
            if (g) g->lanem_unfit = false;
            if (g && g->lanem) { const size_t lb = lanem_part_bytes(g); A->bytes -= lb; g->bytes -= lb; free_lanem_part(g->lanem); g->lanem = nullptr; }
            if (g) g->blane_unfit = false;
            if (g && g->blane) { const size_t lb = blane_part_bytes(g); A->bytes -= lb; g->bytes -= lb; free_blane_part(g->blane); g->blane = nullptr; }
            if (g) g->line_unfit = false;
        }
        return PAMG_OK;
    }
    if (key >= 12) {                                  // tile plan parameters: drop the tile parts only
        for (int k = 0; k < 4; ++k) {
            GsSchedule *g = A->gs[k];
            if (g) g->tile_unfit = false;
            if (g && g->tile) { A->bytes -= g->tile->bytes; g->bytes -= g->tile->bytes; free_tile_part(g->tile); g->tile = nullptr; }
        }
        return PAMG_OK;
    }
    for (int k = 0; k < 4; ++k) { if (A->gs[k]) A->bytes -= A->gs[k]->bytes; free_schedule(A->gs[k]); A->gs[k] = nullptr; }
    return replan(A);
}

int pamg_matrix_autotune(pamg_matrix_t A, int allow_cap)
{
    if (!A) return PAMG_E_ARG;
    if (A->borrowed > 0) return PAMG_E_STATE;
    if (A->nnz < 4000000 || A->npl != 2) return PAMG_OK;
    const size_t ts = tsize(A->dtype);
    void *x = nullptr, *y = nullptr;
    PAMG_HIP(hipMalloc(&x, (size_t)(A->ncols + 8) * ts));
    if (hipMalloc(&y, (size_t)(A->nrows + 8) * ts) != hipSuccess) { hipFree(x); return (int)hipErrorOutOfMemory; }
    hipMemset(x, 0, (size_t)(A->ncols + 8) * ts);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int cap0 = A->cap, fl0 = A->stream_flags;
    const int caps[2] = {cap0, 512};
    int best_cap = cap0, best_fl = fl0, st = PAMG_OK;
    float best_ms = 1e30f;
    // an operator whose whole-operator launches run in the row-mask form has nothing to choose here: its kernels read neither the LDS
    // window nor the streaming flags, and a noise-picked 512-entry window would only multiply the row ranges of the kernels that still
    // use them (the norm's partials, the shard parts)
    long long rm0[8];
    const bool masked = pamg_matrix_row_masks(A, rm0) == PAMG_OK && rm0[0] > 0;
    // candidates: the LDS window (as planned / 512 entries) x the nontemporal operator stream (flag bit 0) x the kernel instantiation (flag bit 5,
    // launch_epi) x the 16-bit column codes on / off
    // (operators without value codes: fewer bytes, but half-width requests -- the SA-level operators of the 256^3 hierarchy, profiles/
    // r05_microbench_sa_ops_wide_codes_vs_32bit_not_kept.json: P0 0.219 ms on codes, 0.199 on 32-bit columns, R0 0.204 against 0.211).  Two interleaved
    // rounds, a candidate's better time counts, and anything but the plan's own setting has to win by 2 %: a single round of six launches picked the
    // XCD-contiguous range order (bit 1: measured slower on every SA-level operator, r05_microbench_sa_ops_nontemporal.json) for R0 by noise and cost
    // 11 % of that product (r05_c4s_kernel_roofline.txt); bit 1 is left to tune key 8: every candidate keeps the caller's setting of it
    const int idx0 = A->use_idx16;
    const int nidx = (A->d_Aj16 && !A->d_Ax8 && idx0) ? 2 : 1;
    // timed on what the cycle runs most: r = b - A x for square operators (three vectors in flight: the Horner steps and the residual), y = A x for
    // the transfer operators -- on A1 of the 256^3 hierarchy the nontemporal stream won up to 8 % on the residual and nothing on y = A x
    const bool square = A->nrows == A->ncols;
    const int probe = square ? EPI_RESID : EPI_SET;
    void *pb = nullptr;
    if (square) {
        if (hipMalloc(&pb, (size_t)(A->nrows + 8) * ts) != hipSuccess) { hipFree(x); hipFree(y); hipEventDestroy(e0); hipEventDestroy(e1); return (int)hipErrorOutOfMemory; }
        hipMemset(pb, 0, (size_t)(A->nrows + 8) * ts);
    }
    float cand_ms[2][2][2][2];                                  // [window][codes on / off][nontemporal][instantiation]
    for (int q = 0; q < 16; ++q) (&cand_ms[0][0][0][0])[q] = 1e30f;
    const int nvc = A->d_Ax8 ? 1 : 2;                           // operators with value codes have one instantiation
    for (int ci = 0; ci < (allow_cap ? 2 : 1) && st == PAMG_OK && !masked; ++ci) {
        if (caps[ci] != A->cap) { A->cap = caps[ci]; st = replan(A); if (st) break; }      // (one re-plan per window: the rounds interleave the other choices)
        for (int round = 0; round < 2 && st == PAMG_OK; ++round) {
            for (int ix = 0; ix < nidx && st == PAMG_OK; ++ix) {
                A->use_idx16 = ix == 0 ? idx0 : 0;
                for (int fl = 0; fl < 2 && st == PAMG_OK; ++fl)
                    for (int vc = 0; vc < nvc && st == PAMG_OK; ++vc) {
                        A->stream_flags = (fl0 & ~(1 | 32)) | fl | (vc ? 32 : 0);
                        for (int w = 0; w < 2 && st == PAMG_OK; ++w) st = stream_launch(A, probe, x, pb, y, 0.0, 0.0, nullptr, nullptr);
                        hipEventRecord(e0, nullptr);
                        for (int r = 0; r < 6 && st == PAMG_OK; ++r) st = stream_launch(A, probe, x, pb, y, 0.0, 0.0, nullptr, nullptr);
                        hipEventRecord(e1, nullptr);
                        hipEventSynchronize(e1);
                        float ms = 0.f;
                        hipEventElapsedTime(&ms, e0, e1);
                        if (st == PAMG_OK) cand_ms[ci][ix][fl][vc] = std::min(cand_ms[ci][ix][fl][vc], ms);
                    }
            }
        }
    }
    A->use_idx16 = idx0;
    hipFree(pb);
    if (st == PAMG_OK && !masked) {
        best_ms = cand_ms[0][0][fl0 & 1][(fl0 & 32) ? nvc - 1 : 0];
        int best_ix = 0;
        for (int ci = 0; ci < (allow_cap ? 2 : 1); ++ci)
            for (int ix = 0; ix < nidx; ++ix)
                for (int fl = 0; fl < 2; ++fl)
                    for (int vc = 0; vc < nvc; ++vc)
                        if (cand_ms[ci][ix][fl][vc] < best_ms * 0.98f) {
                            best_ms = cand_ms[ci][ix][fl][vc]; best_cap = caps[ci]; best_fl = (fl0 & ~(1 | 32)) | fl | (vc ? 32 : 0); best_ix = ix;
                        }
        if (best_ix == 1) A->use_idx16 = 0;
    }
    A->stream_flags = best_fl;
    if (A->cap != best_cap) { A->cap = best_cap; const int s2 = replan(A); if (st == PAMG_OK) st = s2; }
    // lattice form of the row masks: planes per lane (the tile's working set against the XCD's L2: 8 on 256^2-row planes, fewer on larger ones)
    long long rm[8];
    if (st == PAMG_OK && pamg_matrix_row_masks(A, rm) == PAMG_OK && rm[2]) {
        const int kz0 = A->rowmask_kz;
        int best_kz = kz0;
        float best = 1e30f;
        // timed on the residual r = b - A x (three vectors in flight, like the Horner steps): on 512^2-row planes y = A x alone prefers
        // 8 planes per lane, the residual 4 (profiles/r04_microbench_rowmask_512.json)
        void *bvec = nullptr;
        if (hipMalloc(&bvec, (size_t)(A->nrows + 8) * ts) == hipSuccess) hipMemset(bvec, 0, (size_t)(A->nrows + 8) * ts);
        const int epi_t = bvec ? EPI_RESID : EPI_SET;
        float tmin[9];
        for (int k = 0; k < 9; ++k) tmin[k] = 1e30f;
        for (int round = 0; round < 2 && st == PAMG_OK; ++round)                 // two interleaved rounds, the better time of each candidate counts
            for (int kz = 8; kz >= 2 && st == PAMG_OK; kz >>= 1) {
                A->rowmask_kz = kz;
                if (pamg_matrix_row_masks(A, rm) != PAMG_OK || !rm[2]) continue;
                for (int w = 0; w < 2 && st == PAMG_OK; ++w) st = stream_launch(A, epi_t, x, bvec, y, 0.0, 0.0, nullptr, nullptr);
                hipEventRecord(e0, nullptr);
                for (int r = 0; r < 8 && st == PAMG_OK; ++r) st = stream_launch(A, epi_t, x, bvec, y, 0.0, 0.0, nullptr, nullptr);
                hipEventRecord(e1, nullptr);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (st == PAMG_OK) tmin[kz] = std::min(tmin[kz], ms);
            }
        for (int kz = 8; kz >= 2; kz >>= 1)
            if (tmin[kz] < best * 0.98f) { best = tmin[kz]; best_kz = kz; }
        A->rowmask_kz = best_kz;
        if (bvec) hipFree(bvec);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(x); hipFree(y);
    return st;
}

int pamg_matrix_gs_profile(pamg_matrix_t A, int which, long long *out, int64_t capacity, int64_t *count)
{
    if (!A || which < 0 || which > 3 || !count) return PAMG_E_ARG;
    *count = 0;
    GsSchedule *g = A->gs[which];
    if (g && g->tile && g->tile->d_prof) {
        // tiled sweep: [nsteps][8] = {compute wave: operands of the step seen ready, step done (10 ns ticks), XCD, tile,
        // loader: step issued, gatherer: gathers issued, finish started, ready published}; XCD / tile are replaced by the
        // tile and the step's index inside it
        PAMG_HIP(hipDeviceSynchronize());
        TileSched *t = g->tile;
        *count = t->nsteps;
        if (!out) return PAMG_OK;
        if (capacity < t->nsteps) return PAMG_E_ARG;
        PAMG_HIP(hipMemcpy(out, t->d_prof, (size_t)t->nsteps * 8 * sizeof(long long), hipMemcpyDeviceToHost));
        std::vector<int> ts((size_t)t->G + 1);
        PAMG_HIP(hipMemcpy(ts.data(), t->d_tile_step, ts.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < t->G; ++k)
            for (int q = ts[k]; q < ts[k + 1]; ++q) { out[(size_t)q * 8 + 2] = k; out[(size_t)q * 8 + 3] = q - ts[k]; }
        return PAMG_OK;
    }
    if (!g || !g->d_prof) return PAMG_OK;
    PAMG_HIP(hipDeviceSynchronize());
    *count = g->nblk_total;
    if (!out) return PAMG_OK;
    if (capacity < g->nblk_total) return PAMG_E_ARG;
    PAMG_HIP(hipMemcpy(out, g->d_prof, (size_t)g->nblk_total * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    for (int l = 0; l < g->nlevels; ++l)
        for (int q = g->level_blk[l]; q < g->level_blk[l + 1]; ++q) out[(size_t)q * 8 + 7] = l;
    return PAMG_OK;
}

int pamg_matrix_line_info(pamg_matrix_t A, int which, int64_t info[8])
{
    if (!A || which < 0 || which > 3 || !info) return PAMG_E_ARG;
    return pamg::line_info(A->gs[which], info);
}

int pamg_matrix_lane_info(pamg_matrix_t A, int which, int64_t info[8])
{
    if (!A || which < 0 || which > 3 || !info) return PAMG_E_ARG;
    if (A->R > 1) return pamg::blane_info(A->gs[which], info);      // block operators: the block-row lane form (pamg_blane.hip), same fields per block
    return pamg::lane_info(A->gs[which], info);
}

int pamg_matrix_kz_info(pamg_matrix_t A, int which, int64_t info[8])
{
    if (!A || !info || which < 0 || which > 3) return PAMG_E_ARG;
    return pamg::kz_lane_info(A->ls[which], info);
}

int pamg_matrix_lanem_info(pamg_matrix_t A, int which, int64_t info[12], double *growth)
{
    if (!A || !info || which < 0 || which > 3) return PAMG_E_ARG;
    return pamg::lanem_info(A->gs[which], info, growth);
}

int pamg_matrix_lanem_levels(pamg_matrix_t A, int which, int64_t *out, int64_t capacity, int64_t *count)
{
    if (!A || !count || which < 0 || which > 3) return PAMG_E_ARG;
    return pamg::lanem_levels(A->gs[which], out, capacity, count);
}

int pamg_matrix_lane_profile(pamg_matrix_t A, int which, long long *out, int64_t capacity, int64_t *count)
{
    if (!A || which < 0 || which > 3 || !count) return PAMG_E_ARG;
    PAMG_HIP(hipDeviceSynchronize());
    if (A->R > 1) return pamg::blane_profile(A->gs[which], out, capacity, count);
    return pamg::lane_profile(A->gs[which], out, capacity, count);
}

int pamg_matrix_tile_info(pamg_matrix_t A, int which, int64_t info[8])
{
    if (!A || which < 0 || which > 3 || !info) return PAMG_E_ARG;
    for (int k = 0; k < 8; ++k) info[k] = 0;
    const GsSchedule *g = A->gs[which];
    if (!g || !g->tile) return PAMG_OK;
    const TileSched *t = g->tile;
    info[0] = t->G; info[1] = t->W; info[2] = (t->NCH + t->NV + 1) | (t->D << 8) | (TILE_VAR[t->wide][2] << 16) | (t->Q << 24); info[3] = t->nsteps;
    info[4] = t->n_local; info[5] = t->n_global; info[6] = t->n_publish; info[7] = t->lds;
    return PAMG_OK;
}

int pamg_matrix_kaczmarz(pamg_matrix_t L, int nr, void *v, const void *b, const void *Dinv, double omega, int sweep,
                         int iterations, void *xout, pamg_stream_t s)
{
    if (!L || iterations < 0 || sweep < PAMG_FORWARD || sweep > PAMG_SYMMETRIC) return PAMG_E_ARG;
    const int n = (int)L->nrows;
    if (n == 0) return PAMG_OK;
    for (int it = 0; it < iterations; ++it) {
        if (sweep != PAMG_BACKWARD) PAMG_TRY(pamg::kaczmarz_sweep(L, nr != 0, v, b, Dinv, omega, 0, n, 1, xout, (hipStream_t)s));
        if (sweep != PAMG_FORWARD) PAMG_TRY(pamg::kaczmarz_sweep(L, nr != 0, v, b, Dinv, omega, n - 1, -1, -1, xout, (hipStream_t)s));
    }
    return PAMG_OK;
}

int pamg_vec_mul(int dtype, int64_t n, const void *a, const void *b, void *y, pamg_stream_t s)
{
    if (dtype != PAMG_F64 && dtype != PAMG_F32) return PAMG_E_ARG;
    return pamg::vec_mul(dtype, n, a, b, y, (hipStream_t)s);
}

int pamg_matrix_subset_rows(pamg_matrix_t A, const int32_t *rows, int nrows, pamg_matrix_t *sub)
{
    return pamg::matrix_row_subset(A, rows, nrows, sub);
}

int pamg_matrix_jacobi_indexed(pamg_matrix_t sub, void *x, const void *b, double omega, void *work, pamg_stream_t s)
{
    return pamg::jacobi_indexed(sub, x, b, omega, work, (hipStream_t)s);
}

int pamg_matrix_flow_error(pamg_matrix_t A, int *error)
{
    if (!A || !error) return PAMG_E_ARG;
    *error = 0;
    PAMG_HIP(hipDeviceSynchronize());
    bool e = false;
    PAMG_TRY(pamg::sweep_error(A, &e));
    *error = e ? 1 : 0;
    return PAMG_OK;
}

int pamg_matrix_spmv(pamg_matrix_t A, int mode, const void *x, const void *b_or_v, double c, void *y,
                     pamg_stream_t s)
{
    if (!A || !x || !y) return PAMG_E_ARG;
    int epi;
    switch (mode) {
        case PAMG_SPMV_SET: epi = EPI_SET; break;
        case PAMG_SPMV_ACC: epi = EPI_ACC; break;
        case PAMG_SPMV_RESID: epi = EPI_RESID; break;
        case PAMG_SPMV_AXPBY: epi = EPI_AXPBY; break;
        case PAMG_SPMV_ACC_AXPBY: epi = EPI_ACC_AXPBY; break;
        default: return PAMG_E_ARG;
    }
    if (mode >= PAMG_SPMV_RESID && !b_or_v) return PAMG_E_ARG;
    return stream_launch(A, epi, x, b_or_v, y, c, 0.0, nullptr, (hipStream_t)s);
}

int pamg_matrix_split_ranges(pamg_matrix_t A, int64_t n_owned_cols)
{
    if (!A) return PAMG_E_ARG;
    if (A->borrowed > 0) return PAMG_E_STATE;
    return matrix_split_ranges(A, n_owned_cols);
}

int pamg_matrix_spmv_part(pamg_matrix_t A, int part, int mode, const void *x, const void *b_or_v, double c, void *y, pamg_stream_t s)
{
    if (!A || !x || !y || part < 0 || part > 2) return PAMG_E_ARG;
    int epi;
    switch (mode) {
        case PAMG_SPMV_SET: epi = EPI_SET; break;
        case PAMG_SPMV_ACC: epi = EPI_ACC; break;
        case PAMG_SPMV_RESID: epi = EPI_RESID; break;
        case PAMG_SPMV_AXPBY: epi = EPI_AXPBY; break;
        case PAMG_SPMV_ACC_AXPBY: epi = EPI_ACC_AXPBY; break;
        default: return PAMG_E_ARG;
    }
    if (mode >= PAMG_SPMV_RESID && !b_or_v) return PAMG_E_ARG;
    return stream_launch_part(A, part, epi, x, b_or_v, y, c, 0.0, nullptr, (hipStream_t)s);
}

int pamg_matrix_resid_sumsq(pamg_matrix_t A, const void *x, const void *b, double *out_sumsq,
                            pamg_stream_t s)
{
    if (!A || !x || !b || !out_sumsq) return PAMG_E_ARG;
    PAMG_TRY(stream_launch(A, EPI_SUMSQ, x, b, nullptr, 0.0, 0.0, A->d_partial, (hipStream_t)s));
    return reduce_partials(A->d_partial, A->nblk, out_sumsq, (hipStream_t)s);
}

int pamg_vec_sumsq(int dtype, int64_t n, const void *x, double *out_sumsq, pamg_stream_t s)
{
    if (n < 0 || !out_sumsq || (n > 0 && !x)) return PAMG_E_ARG;
    int dev = 0;
    PAMG_HIP(hipGetDevice(&dev));
    if (!g_scratch || g_scratch_dev != dev) {
        PAMG_HIP(hipMalloc((void **)&g_scratch, sizeof(double) * 1032));
        g_scratch_dev = dev;
    }
    return vec_sumsq(dtype, n, x, g_scratch, out_sumsq, (hipStream_t)s);
}

int pamg_vec_axpy(int dtype, int64_t n, double a, const void *x, void *y, pamg_stream_t s)
{
    return vec_axpy(dtype, n, a, x, y, (hipStream_t)s);
}

int pamg_vec_scale(int dtype, int64_t n, double a, const void *x, void *y, pamg_stream_t s)
{
    return vec_scale(dtype, n, a, x, y, (hipStream_t)s);
}

int pamg_vec_gather(int dtype, int64_t n, const int32_t *idx, const void *src, void *dst, pamg_stream_t s)
{
    if (n < 0 || (n > 0 && (!idx || !src || !dst))) return PAMG_E_ARG;
    if (n == 0) return PAMG_OK;
    const int grid = vgrid(n);
    if (dtype == PAMG_F64)
        hipLaunchKernelGGL((vec_gather_kernel<double>), dim3(grid), dim3(BLK), 0, (hipStream_t)s, n, idx, (const double *)src, (double *)dst);
    else if (dtype == PAMG_F32)
        hipLaunchKernelGGL((vec_gather_kernel<float>), dim3(grid), dim3(BLK), 0, (hipStream_t)s, n, idx, (const float *)src, (float *)dst);
    else return PAMG_E_UNSUPPORTED;
    return (int)hipGetLastError();
}

}  // extern "C"
