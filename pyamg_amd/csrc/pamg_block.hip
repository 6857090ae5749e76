// pamg_block.hip -- dispatch of the square-block (BSR) relaxation kernels: the order-exact block / point sweeps over a level schedule, the
// Jacobi-type steps, the hand-over to the fast-order block sweep (pamg_blane.hip).  Split from pamg_matrix.hip in round 5: the block kernels
// (five compile-time block sizes x four flavours x four schedulers) were a third of that file's compile time.
#include <algorithm>

#include "pamg_kernels.h"

using namespace pamg;

namespace pamg {

bool want_blanes(const pamg_matrix_s *A, const GsSchedule *g)
{
    return A->gs_order == 1 && A->gs_mode == 0 && blane_eligible(A, g);
}

namespace {

template <typename T>
static BlockArgs<T> block_args(pamg_matrix_s *A, const int *rid, const void *Dinv, const void *xsrc, void *xdst,
                               const void *b, double omega, int dirn)
{
    BlockArgs<T> a;
    a.bAp = A->d_bAp; a.bAj = A->d_bAj;
    a.Ax = (const T *)A->d_bAx;                // block-ordered values
    a.rid = rid; a.Dinv = (const T *)Dinv;
    a.xsrc = (const T *)xsrc; a.xdst = (T *)xdst; a.b = (const T *)b;
    a.omega = (T)omega; a.bs = A->R; a.first = 0; a.count = A->n_brow; a.dirn = dirn;
    a.xs = nullptr; a.err = nullptr; a.nidle = 1;
    return a;
}

template <typename T>
static int bsr_stream_launch(int kind, int grid, int lds, hipStream_t s, const BlockArgs<T> &a, const BsrRange<T> &g, int first)
{
    if (grid <= 0) return PAMG_OK;
    switch (kind) {
        case BLK_JACOBI: hipLaunchKernelGGL((bsr_stream_kernel<T, BLK_JACOBI>), dim3(grid), dim3(BLK), lds, s, a, g, first); break;
        case BLK_GS: hipLaunchKernelGGL((bsr_stream_kernel<T, BLK_GS>), dim3(grid), dim3(BLK), lds, s, a, g, first); break;
        case PNT_JACOBI: hipLaunchKernelGGL((bsr_stream_kernel<T, PNT_JACOBI>), dim3(grid), dim3(BLK), lds, s, a, g, first); break;
        case PNT_GS: hipLaunchKernelGGL((bsr_stream_kernel<T, PNT_GS>), dim3(grid), dim3(BLK), lds, s, a, g, first); break;
        default: return PAMG_E_ARG;
    }
    return (int)hipGetLastError();
}

// order-exact block sweep (PNT_GS / BLK_GS) over a level schedule, same policy as the scalar sweeps:
// narrow -> one persistent workgroup; otherwise the granular sweep (no barriers, the published datum
// is the flag; old values from a snapshot when the block pattern is not structurally symmetric);
// gs_mode 1 -> one launch per level, 4 -> persistent grid with a counter barrier per level (kept for
// comparison), 3 -> one workgroup, 2 -> granular whenever the plan allows it
template <typename T>
static int block_sweep_t(pamg_matrix_s *A, GsSchedule *g, int kind, const void *Dinv, void *x, const void *b, int dirn,
                         hipStream_t s)
{
    BlockArgs<T> a = block_args<T>(A, g->d_rid, Dinv, x, x, b, 0.0, dirn);
    a.count = (int)g->nrows;
    BsrRange<T> r;
    r.meta = g->d_blkmeta; r.pAp = g->d_Ap; r.pblk = g->d_pblk; r.pbj = g->d_Aj; r.dpos = g->d_dpos; r.capv = A->cap;
    const size_t ts = tsize(A->dtype);
    const int lds = std::max(64, (int)ts * (A->cap + 8));
    const bool narrow = (int64_t)g->nblk_total * 16 <= (int64_t)g->nlevels * A->flow_cap;
    const bool persist = g->nlevels > 1 && A->gs_mode != 1;
    const bool single = persist && (A->gs_mode == 3 || (A->gs_mode == 0 && narrow));
    const bool granular = persist && !single && A->gs_mode != 4 && g->d_xs && A->cap <= GE * BLK && 3 * lds <= 60 * 1024;
    if (granular) {
        const int64_t n = A->nrows;
        a.xs = (T *)g->d_xs;
        a.err = g->d_sync + 1;
        a.nidle = (int)std::max<int64_t>(1, std::min<int64_t>(n, 1 << 20));
        if (!g->symmetric) {
            if (!g->d_xold) return PAMG_E_STATE;
            PAMG_HIP(hipMemcpyAsync(g->d_xold, x, (size_t)n * ts, hipMemcpyDeviceToDevice, s));
            a.xsrc = (const T *)g->d_xold;
        }
        const int fgrid = (int)std::min<int64_t>(4096, (n + BLK - 1) / BLK);
        hipLaunchKernelGGL((fill_sentinel_kernel<T>), dim3(fgrid), dim3(BLK), 0, s, (T *)g->d_xs, n);
        PAMG_HIP(hipGetLastError());
        const int per_level = (g->nblk_total + g->nlevels - 1) / g->nlevels;
        // every workgroup of the persistent grid must be resident (a workgroup spins on ranges owned by others): cap
        // the grid by what the device holds of THIS kernel with THIS much LDS, one per CU below the query as for the
        // scalar sweep (gran2_grid) -- matters on partitioned devices and smaller parts
        int resident = 256;
        {
            const int cus = device_cus();
            int nb = 0;
            const hipError_t e = kind == PNT_GS
                ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bsr_gran_kernel<T, PNT_GS>, BLK, (size_t)(3 * lds))
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bsr_gran_kernel<T, BLK_GS>, BLK, (size_t)(3 * lds));
            if (e != hipSuccess || nb < 1) nb = 1;
            resident = std::max(1, std::min(nb - 1, 4)) * cus;
            if (nb == 1) resident = cus;
        }
        int G = std::max(1, std::min(g->nblk_total, std::min(256, resident)));
        if (A->gran_cap > 0) G = std::min(G, A->gran_cap);
        else G = std::min(G, std::max(32, 8 * per_level));
        G = std::min(G, resident);
        // compile-time block sizes for the common ones (elasticity: 2, 3, 6; 4), the generic kernel otherwise
#define PAMG_BG(K, B) hipLaunchKernelGGL((bsr_gran_kernel<T, K, B>), dim3(G), dim3(BLK), 3 * lds, s, a, r, g->nblk_total)
#define PAMG_BGS(K)                                                                      \
        switch (A->R) {                                                                  \
            case 2: PAMG_BG(K, 2); break;                                                \
            case 3: PAMG_BG(K, 3); break;                                                \
            case 4: PAMG_BG(K, 4); break;                                                \
            case 6: PAMG_BG(K, 6); break;                                                \
            default: PAMG_BG(K, 0); break;                                               \
        }
        if (kind == PNT_GS) { PAMG_BGS(PNT_GS) } else { PAMG_BGS(BLK_GS) }
#undef PAMG_BGS
#undef PAMG_BG
        return (int)hipGetLastError();
    }
    // small block levels (the iterate fits the LDS next to the range's products): one workgroup, x and b in LDS, the next range's
    // blocks prefetched -- bit-identical to the kernels below (tune key 5 = 3 keeps bsr_flow_kernel for comparison)
    if (single && A->gs_mode == 0 && A->R <= MAXBS) {
        const int64_t n = A->nrows;
        const size_t need = (size_t)(2 * n + r.capv + SMALL_THREADS) * ts + 64;
        if (need <= 60 * 1024 && (int64_t)g->max_range_rows * A->R <= SMALL_THREADS && (int64_t)g->max_range_blocks * A->R <= r.capv) {
#define PAMG_BS(K, B) hipLaunchKernelGGL((bsr_small_kernel<T, K, B>), dim3(1), dim3(SMALL_THREADS), need, s, a, r, g->nblk_total, (int)n)
#define PAMG_BSS(K)                                                                      \
            switch (A->R) {                                                              \
                case 2: PAMG_BS(K, 2); break;                                            \
                case 3: PAMG_BS(K, 3); break;                                            \
                case 4: PAMG_BS(K, 4); break;                                            \
                case 6: PAMG_BS(K, 6); break;                                            \
                default: PAMG_BS(K, 0); break;                                           \
            }
            if (kind == PNT_GS) { PAMG_BSS(PNT_GS) } else { PAMG_BSS(BLK_GS) }
#undef PAMG_BSS
#undef PAMG_BS
            return (int)hipGetLastError();
        }
    }
    int G = 0;
    if (persist) {
        const int wide = std::max(1, std::min(256, g->max_level_blocks));
        if (single) G = 1;
        else if (A->gs_mode == 4 || A->gs_mode == 2) G = A->gran_cap > 0 ? std::min(wide, A->gran_cap) : wide;
        else if (A->flow_cap > 0) { G = wide; if (G > 128) G = 0; }
    }
    if (G > 0) {
        if (G > 1) PAMG_HIP(hipMemsetAsync(g->d_sync, 0, 2048, s));
#define PAMG_BF(K)                                                                                                          \
        if (G == 1) hipLaunchKernelGGL((bsr_flow_kernel<T, K, false>), dim3(1), dim3(BLK), lds, s, a, r, g->d_level_blk, g->nlevels, g->d_sync); \
        else hipLaunchKernelGGL((bsr_flow_kernel<T, K, true>), dim3(G), dim3(BLK), lds, s, a, r, g->d_level_blk, g->nlevels, g->d_sync);
        if (kind == PNT_GS) { PAMG_BF(PNT_GS) } else { PAMG_BF(BLK_GS) }
#undef PAMG_BF
        return (int)hipGetLastError();
    }
    for (int l = 0; l < g->nlevels; ++l)
        PAMG_TRY(bsr_stream_launch<T>(kind, g->level_blk[l + 1] - g->level_blk[l], lds, s, a, r, g->level_blk[l]));
    return PAMG_OK;
}

}  // namespace

// the BSR point sweep (amg_core::bsr_gauss_seidel) of gs_sweep's block operators
int block_point_sweep(pamg_matrix_s *A, GsSchedule *g, void *x, const void *b, int dirn, hipStream_t s)
{
    return A->dtype == PAMG_F64 ? block_sweep_t<double>(A, g, PNT_GS, nullptr, x, b, dirn, s)
                                : block_sweep_t<float>(A, g, PNT_GS, nullptr, x, b, dirn, s);
}

int block_gs_sweep(pamg_matrix_s *A, void *x, const void *b, const void *Dinv, int row_start,
                   int row_stop, int row_step, hipStream_t s)
{
    GsSchedule *g = nullptr;
    PAMG_TRY(get_schedule(A, row_start, row_stop, row_step, &g));
    PAMG_TRY(ensure_parts(A, g, true));
    if (want_blanes(A, g) && g->blane) return blane_launch(A, g, Dinv, x, b, s);
    return A->dtype == PAMG_F64 ? block_sweep_t<double>(A, g, BLK_GS, Dinv, x, b, 1, s)
                                : block_sweep_t<float>(A, g, BLK_GS, Dinv, x, b, 1, s);
}

// one out-of-place Jacobi-type step over all block rows (BLK_JACOBI / PNT_JACOBI)
int block_jacobi_step(pamg_matrix_s *A, int kind, const void *Dinv, const void *xsrc, void *xdst,
                      const void *b, double omega, hipStream_t s)
{
    if (!A->d_bmeta) return PAMG_E_STATE;
    const int lds = std::max(64, (int)tsize(A->dtype) * (A->cap + 8));
    if (A->dtype == PAMG_F64) {
        BsrRange<double> r;
        r.meta = A->d_bmeta; r.pAp = A->d_bAp; r.pblk = nullptr; r.pbj = A->d_bAjf; r.dpos = A->d_bdiag; r.capv = A->cap;
        return bsr_stream_launch<double>(kind, A->bnblk, lds, s, block_args<double>(A, nullptr, Dinv, xsrc, xdst, b, omega, 1), r, 0);
    }
    BsrRange<float> r;
    r.meta = A->d_bmeta; r.pAp = A->d_bAp; r.pblk = nullptr; r.pbj = A->d_bAjf; r.dpos = A->d_bdiag; r.capv = A->cap;
    return bsr_stream_launch<float>(kind, A->bnblk, lds, s, block_args<float>(A, nullptr, Dinv, xsrc, xdst, b, omega, 1), r, 0);
}

}  // namespace pamg
