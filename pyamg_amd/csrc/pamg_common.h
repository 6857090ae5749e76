// pamg_common.h -- shared declarations for libpyamg_amd.so (gfx950 / CDNA4 only).
#pragma once
#include "pamg_host_threads.h"
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pyamg_amd.h"

#define PAMG_STR2(x) #x
#define PAMG_STR(x) PAMG_STR2(x)
#define PAMG_HIP(expr)                                   \
    do {                                                 \
        hipError_t e__ = (expr);                         \
        if (e__ != hipSuccess) return (int)e__;          \
    } while (0)
#define PAMG_TRY(expr)                                   \
    do {                                                 \
        int s__ = (expr);                                \
        if (s__ != PAMG_OK) return s__;                  \
    } while (0)

namespace pamg {

#ifndef PAMG_BLK
#define PAMG_BLK 256
#endif
constexpr int BLK = PAMG_BLK;      // threads per workgroup (256 = 4 wave64)
constexpr int WAVE = 64;

// epilogues of the LDS-streamed CSR kernel (what is done with a finished row sum)
enum : int {
    EPI_SET = 0,        // y = s
    EPI_ACC,            // y = y + s
    EPI_RESID,          // y = b - s
    EPI_AXPBY,          // y = c*v + s
    EPI_ACC_AXPBY,      // y = y + (c*v + s)
    EPI_SUMSQ,          // partial[block] = sum (b - s)^2     (nothing stored)
    EPI_ACCSEQ,         // y = ((y + p0) + p1) + ...  (SciPy csr_matvec's running sum from y)
    EPI_JACOBI,         // amg_core::jacobi            relaxation.h:309-346
    EPI_JACOBI_B,       // amg_core::bsr_jacobi, 1x1   relaxation.h:472-562
    EPI_GS,             // amg_core::gauss_seidel      relaxation.h:48-76
    EPI_GS_B,           // amg_core::bsr_gauss_seidel, 1x1  relaxation.h:185-266
    EPI_SOR,            // amg_core::sor_gauss_seidel  relaxation.h:116-145
    EPI_JACOBI_IDX,     // amg_core::jacobi_indexed    relaxation.h:382-427 (row-subset operator, out[r] = new x[rid[r]])
    EPI_COUNT
};

constexpr int MAXBS = 8;          // largest square block handled by the block smoothers
enum : int { BLK_JACOBI = 0, BLK_GS, PNT_JACOBI, PNT_GS };   // flavours of the block (BSR) relaxation kernels

template <typename T>
struct StreamArgs {
    const int4 *blkmeta; // [nblocks] {first row, end row, first entry, end entry} per workgroup
    const int *Ap;       // row pointer (of the operator or of a level-permuted copy)
    const int *Aj;
    const T *Ax;
    const int *rid;      // original row id of stored row r (permuted copies) or nullptr
    const T *diag;       // diagonal value of stored row r (last stored a_ii; 0 = none/zero)
    const T *x;          // gather source
    T *xs;               // granular sweep: hand-off buffer (sentinel = not published yet)
    unsigned *err;       // granular sweep: error flag (spin bound hit)
    const T *b;          // right-hand side / v
    T *y;                // destination (== x for the in-place GS family)
    double *partial;     // EPI_SUMSQ: one double per workgroup
    T c;                 // EPI_*AXPBY coefficient
    T omega;
    int cap;             // products staged in LDS per workgroup
    int nblk;            // row ranges of this launch
    int flags;           // bit 0: non-temporal operator stream, bit 1: XCD-aware range order, bits 2/3: ablations
    int nidle;           // granular sweep: elements of xs that idle lanes may read (>= 1)
    const unsigned short *Aj16;   // whole-operator kernels: column ids as 16-bit window codes (window << 14 | offset) or nullptr
    const int4 *wbase;            //   per row range: the first column of its (up to four) windows
    int4 wb;                      //   the current range's window bases (set by the kernel)
    const int *blkmap;            // whole-operator kernels: launch index -> row range (nullptr = identity); nblk = ranges launched
    const unsigned char *Ax8;     // whole-operator kernels: values as 8-bit codes into vdict (operators with <= 256 distinct values) or nullptr
    const T *vdict;               //   the distinct values (bit patterns in increasing order), nvd of them
    int nvd;
    const unsigned char *pid;     // whole-operator kernels: row-pattern number per row (255 = irregular row) or nullptr
    const void *ptab;             //   [256] lengths | [npat * lmax] offsets | [npat * lmax] values
    int npat, lmax;
};

// One dependency-level schedule for an order-exact sweep (forward or backward, or a
// general (row_start,row_stop,row_step) sweep): a row-permuted copy of the operator with
// the rows of each level stored contiguously, so every level streams coalesced.
struct GsSchedule {
    int row_start = 0, row_stop = 0, row_step = 0;
    int nlevels = 0;
    int64_t nrows = 0, nnz = 0;
    int *d_Ap = nullptr, *d_Aj = nullptr, *d_rid = nullptr;
    int4 *d_blkmeta = nullptr;
    void *d_Ax = nullptr, *d_diag = nullptr;
    int *d_level_blk = nullptr;      // device copy of level_blk (persistent sweep kernel)
    int *d_pblk = nullptr;           // block schedules: position of scheduled block q in the operator's block arrays
    int *d_dpos = nullptr;           // block schedules: position of each scheduled row's diagonal block (or -1)
    void *d_xs = nullptr;            // granular sweep: hand-off buffer (one value per matrix row)
    void *d_xold = nullptr;          // granular sweep, non-symmetric patterns: snapshot of x (old values)
    long long *d_prof = nullptr;     // granular sweep diagnostics: [nblk_total][8] time stamps
    bool symmetric = false;          // pattern among swept rows is structurally symmetric
    int nblk_total = 0;
    unsigned *d_sync = nullptr;      // [0] barrier arrival counter (block sweeps), [1] error flag, [20..21] ticket counter + home XCD
    int max_level_blocks = 0;
    int max_range_rows = 0, max_range_blocks = 0;   // block schedules: the largest row range (block rows / blocks)
    std::vector<int> level_blk;      // [nlevels+1] workgroup range of each level
    size_t bytes = 0;
    std::vector<int> h_vis, h_lvl;   // scalar schedules: visit index (-1 = not swept) and dependency level of every row
    bool has_level_part = false;     // the level-permuted copy above is built (granular / single-workgroup / per-level schedulers)
    int cap = 0;                     // entries per row range of the level-permuted copy (the LDS window its kernels run with)
    struct TileSched *tile = nullptr; // tiled sweep (pamg_tile_plan.h / pamg_tile_kernels.h), built on demand
    bool tile_unfit = false;         // the tile planner declined this schedule (a step would not fit): other schedulers run it
    struct LineSched *line = nullptr; // line-scan fast-order sweep for banded operators in their natural order (pamg_line_plan.h / pamg_line.hip)
    bool line_unfit = false;         // the line planner declined this schedule (no runs of coupled consecutive rows / rows too long)
    struct LaneSched *lane = nullptr; // lane-parallel "fast order" sweep (pamg_lane_plan.h / pamg_lane.hip), built on demand
    bool lane_unfit = false;         // the lane planner declined this schedule (rows too long / padding too wasteful)
    struct LaneMSched *lanem = nullptr; // MERGED lane-parallel sweep: s dependency levels eliminated into one super-level (pamg_lanem_plan.h / pamg_lane.hip), f64 Gauss-Seidel
    bool lanem_unfit = false;        // its planner declined (a row beyond 256 operands, nothing to merge, growth bound)
    struct BlaneSched *blane = nullptr; // block schedules: lane-parallel fast-order block Gauss-Seidel (pamg_blane_plan.h / pamg_blane.hip), built on demand
    bool blane_unfit = false;        // its planner declined (block rows too long, block size not compiled, not f64)
};

// Device side of a tile plan: the step blocks (pamg_tile_plan.h: one fixed-size block of entry codes, values and
// row records per step, tile after tile) and the launch geometry chosen for them.
struct TileSched {
    int G = 0, W = 0, NCH = 1, NV = 1, wide = 0, xo = 0, D = 0, Q = 0, nsteps = 0, lds = 0;   // wide: kernel variant (0 = small gather lists, several workgroups per CU allowed)
    unsigned char *d_blocks = nullptr;
    int *d_tile_step = nullptr;
    long long *d_prof = nullptr;     // [nsteps][4] diagnostics (tune key 11)
    int64_t n_local = 0, n_global = 0, n_publish = 0, max_step_entries = 0;
    size_t bytes = 0;
};

// Dependency levels of a Kaczmarz-type sweep over the lines (rows) of an operator: two lines conflict when they
// share a column index; lines of one level are pairwise conflict-free.
struct LineSchedule {
    int start = 0, stop = 0, step = 0;
    int nlevels = 0;
    int *d_lines = nullptr;          // line ids, level after level
    int *d_level_ptr = nullptr;      // device copy of level_ptr (persistent kernel)
    int *d_len = nullptr, *d_lo = nullptr, *d_ej = nullptr;   // persistent kernel: per scheduled line its length, first-entry
    void *d_ea = nullptr;            //   offset and a dense slab of its first KZ entries (indices / values), schedule order
    std::vector<int> level_ptr;      // [nlevels+1]
    size_t bytes = 0;
    struct KzLaneSched *kzl = nullptr; // lane-parallel fast-order form of the sweep (pamg_kz_plan.h / pamg_kz.hip), built with the schedule when tune key 24 = 1
    bool kzl_unfit = false;          // its planner declined (lines too long, an index twice in a line, not f64)
};

}  // namespace pamg

struct pamg_matrix_s {
    int dtype = PAMG_F64;
    int flavour = PAMG_CSR;
    int n_brow = 0, n_bcol = 0, R = 1, C = 1;
    // scalar (flattened) CSR view actually resident in HBM
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int *d_Ap = nullptr, *d_Aj = nullptr;
    void *d_Ax = nullptr;
    void *d_diag = nullptr;                   // diagonal of the scalar view (point smoothers)
    int *d_rowid = nullptr;                   // row-subset operators (indexed Jacobi): original row of stored row r
    // block view kept for the true block smoothers (bs > 1): block CSR arrays
    int *d_bAp = nullptr, *d_bAj = nullptr;   // nullptr when R == C == 1
    int *d_bAjf = nullptr, *d_bdiag = nullptr; // block columns with the diagonal flag (bit 30); diagonal block position per block row
    void *d_bAx = nullptr;                    // block-ordered values (square blocks only)
    int64_t nblocks_b = 0;
    int4 *d_bmeta = nullptr;                  // row-range plan over block rows (LDS-streamed BSR relaxation)
    int bnblk = 0;
    // host copies of the index arrays (needed for lazy GS analysis / re-planning)
    std::vector<int> h_Ap, h_Aj;
    std::vector<int> h_bAp, h_bAj;
    // plan for the streamed kernels
    int cap = 1536, npl = 2, max_rows = 1024;
    int flow_cap = 32;               // single-workgroup persistent sweep when a schedule averages <= flow_cap/16 row ranges per level
    unsigned short *d_Aj16 = nullptr; // column ids of the scalar view as 16-bit window codes (csr_stream_kernel reads 2 instead of 4 bytes per entry)
    int4 *d_wbase = nullptr;         //   window bases per row range; both null when some range needs more than four 16 K-column windows
    int use_idx16 = 1;               // tune key 19
    unsigned char *d_Ax8 = nullptr;  // values of the scalar view as 8-bit codes into d_vdict: operators with <= 256 distinct values (stencils)
    void *d_vdict = nullptr;         //   stream 1 instead of 8 bytes per value; needs the 16-bit column stream; null otherwise
    int nvdict = 0;
    int use_val8 = 1;                // tune key 21
    int use_rowg = 0;                // tune key 22: row-gather form of the whole-operator kernels (value-code operators; set with the codes)
    unsigned char *d_pid = nullptr;  // row patterns (plan_rowpat): list number per row, 255 = walk the row through the code arrays
    void *d_ptab = nullptr;          //   [256] lengths | [npat * pat_lmax] column offsets | [npat * pat_lmax] values
    int npat = 0, pat_lmax = 0;
    unsigned char *d_pmask = nullptr;  // row masks (plan_row_masks): which entries of the longest list a row has, 0 = walk the CSR arrays
    int rm_nu = 0, rm_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};   //   the longest list: entries, column offsets,
    unsigned long long rm_val[8] = {0, 0, 0, 0, 0, 0, 0, 0};   //   values (bit patterns, low 32 bits for float)
    long long rm_walked = 0;
    int rowmask_kz = 8;              // planes per lane of csr_rowmask3d_kernel (2, 4, 8)
    int rowmask_flags = 3;           // row-mask kernels: (bit 0: nontemporal b / mask / result -- always on since round 5, ignored) bit 1 plane-by-plane XCD order, bit 2 XCD-contiguous eighths
    int use_rowpat = 1;              // tune key 23: 0 off, 1 the row-mask kernels where the operator has the form, else the table kernel (default), 3 the table kernel always, 4 the linear row-mask kernel instead of the lattice form
    int cap_from_val8 = 0;           // cap was raised to 2048 because the operator streams value codes (level schedules keep 1536)
    int stream_flags = 0;            // StreamArgs::flags for the whole-operator launches
    int lds_pad = 0;                 // extra (unused) dynamic LDS of the staged whole-operator kernel: fewer workgroups per CU   (tune key 36, autotune)
    int gran_xcd = 0;                // granular sweep inside one XCD's L2: 0 auto (small operators), 1 always, 2 never
    int gran_cap = 0;                // granular sweep: cap on the persistent grid (0 = auto)
    int gs_mode = 0;                 // scalar sweep scheduler: 0 auto, 1 one launch per level, 2 granular, 3 single workgroup, 5 tiled
    int tile_G = 0;                  // tiled sweep: tiles (0 = auto)
    int tile_W = 0;                  // tiled sweep: LDS ring slots (power of two; 0 = auto)
    int tile_cap = 0;                // tiled sweep: scheduled entries per step (0 = auto)
    int tile_D = 0, tile_Q = -1;     // tiled sweep: LDS slots per tile / steps in flight behind the landed mark (0 / -1 = auto)
    int tile_part = 1;               // tiled sweep: 1 = pencil tiles when the operator is a three-band grid stencil, 0 = contiguous chunks always
    bool tile_default = false;       // auto mode (gs_mode 0) prefers the tiled sweep everywhere (default: only where it measured faster)
    int max_row_len = 0;             // longest row of the scalar view
    int borrowed = 0;                // solvers holding this operator (tuning is refused while > 0: captured graphs point into the schedules)
    int gs_prof = 0;                 // granular sweep: record per-range time stamps (tune key 11, diagnostics)
    int gs_order = 0;                // tune key 24: 0 = order-exact row sums (bit-identical to the reference), 1 = fast order: lane-parallel row
                                     //   sums and multiplication by 1/a_ii (same sweep order; agrees to rounding) where the schedule fits that form
    int lane_L = 0, lane_G = 0;      // fast order: lanes per row (0 = automatic) / persistent workgroups (0 = automatic)   (tune keys 25, 26)
    int lane_flags = 1;              // fast order: bit 0 = gate operand (a wave that runs ahead polls one value instead of all its operands), bit 3 = gate in the line scan   (tune key 28)
    int line_scan = 1;               // fast order: line-scan sweep where consecutive rows are coupled (grid stencils), tried before the lane form   (tune key 30)
    int lane_merge = 0;              // fast order: dependency levels merged into one super-level at most (0 = automatic, 1 = never, 2..8)   (tune key 33)
    int lanem_rpw = 0;               // merged form: rows per wave (0 = automatic: 2 on levels above 131 072 rows, 1 below)   (tune key 35)
    int lanem_ahead10 = 40;          // merged form: waves launched = this / 10 x the rows of an average super-level   (tune key 34)
    int lane_wide = 0;               // fast order on wide schedules (>= 2048 rows per dependency level): 0 = the tiled exact sweep keeps them, 1 = lane form   (tune key 27)
    int gs_cap = 0;                  // entries per row range of the level schedules (tune key 20; 0 = automatic: `cap`, 512 on the multi-XCD granular sweep of SA-like rows)
    int nblk = 0;
    int *d_part[2] = {nullptr, nullptr};   // row shards (pamg_dist.hip): row ranges that read owned columns only / that read the halo
    int npart[2] = {0, 0};
    int64_t part_cols = -1;          //   owned columns the split was made for (-1 = none)
    int64_t part_row0[2] = {-1, -1}, part_row1[2] = {-1, -1};   //   rows [row0, row1) of a part whose ranges are consecutive (else -1)
    int4 *d_blkmeta = nullptr;
    double *d_partial = nullptr;     // nblk doubles (sum-of-squares partials)
    pamg::GsSchedule *gs[4] = {nullptr, nullptr, nullptr, nullptr};  // fwd, bwd, 2 custom
    // fast order of the BSR POINT sweep (amg_core::bsr_gauss_seidel): the same rows in the same order are the scalar Gauss-Seidel sweep of the
    // flattened view, so a block operator swept in fast order owns a scalar CSR twin of itself that carries the lane / merged / line schedules
    pamg_matrix_s *point_twin = nullptr;
    bool point_twin_unfit = false;   // no fast-order form fits the flattened rows (or the twin could not be built): the exact block kernels sweep
    pamg::LineSchedule *ls[4] = {nullptr, nullptr, nullptr, nullptr};  // Kaczmarz sweeps over this operator's rows
    size_t bytes = 0;
};

struct pamg_schwarz_s;
struct pamg_solver_s;
struct pamg_csr_s;

namespace pamg {
// pamg_schwarz.hip
int schwarz_sweep(pamg_schwarz_s *h, void *x, const void *b, int start, int stop, int step, hipStream_t s);
int schwarz_prepare(pamg_schwarz_s *h, int sweep);
int schwarz_error(pamg_schwarz_s *h, bool *error);
void schwarz_level_launches(pamg_schwarz_s *h);
// launch wrappers implemented in pamg_matrix.hip
int stream_launch(pamg_matrix_s *A, int epi, const void *x, const void *b, void *y, double c,
                  double omega, double *partial, hipStream_t s);
int stream_launch_part(pamg_matrix_s *A, int part, int epi, const void *x, const void *b, void *y, double c,
                       double omega, double *partial, hipStream_t s);
int matrix_split_ranges(pamg_matrix_s *A, int64_t n_owned_cols);
void matrix_drop_value_codes(pamg_matrix_s *A);
int gs_sweep(pamg_matrix_s *A, int epi, void *x, const void *b, double omega, int row_start,
             int row_stop, int row_step, hipStream_t s);
int reduce_partials(const double *partial, int n, double *out, hipStream_t s);
int vec_sumsq(int dtype, int64_t n, const void *x, double *scratch, double *out, hipStream_t s);
int vec_axpy(int dtype, int64_t n, double a, const void *x, void *y, hipStream_t s);
int vec_scale(int dtype, int64_t n, double a, const void *x, void *y, hipStream_t s);
int vec_dot(int dtype, int64_t n, const void *x, const void *y, double *scratch, double *out, hipStream_t s);
int vec_mul(int dtype, int64_t n, const void *a, const void *b, void *y, hipStream_t s);
int kaczmarz_sweep(pamg_matrix_s *L, bool nr, void *v, const void *b, const void *Dinv, double omega, int start, int stop,
                   int step, void *xout, hipStream_t s);
int ensure_line_schedule(pamg_matrix_s *L, int start, int stop, int step);
void free_line_schedule(LineSchedule *g);
int vec_scatter(int dtype, int64_t n, const int *idx, const void *src, void *dst, hipStream_t s);
int vec_copy_indexed(int dtype, int64_t n, const int *idx, const void *src, void *dst, hipStream_t s);   // dst[idx[i]] = src[idx[i]]
int matrix_row_subset(pamg_matrix_s *A, const int32_t *rows, int nrows, pamg_matrix_s **out);   // rows: HOST
int jacobi_indexed(pamg_matrix_s *sub, void *x, const void *b, double omega, void *work, hipStream_t s);
int vec_maxratio(int dtype, int64_t n, const void *u, const void *x, double *scratch, double *out, hipStream_t s);
int vec_xpby(int dtype, int64_t n, double beta, const void *z, void *p, hipStream_t s);
int vec_axpy_ratio(int dtype, int64_t n, const double *num, const double *den, double sign, const void *x, void *y, hipStream_t s);
int vec_fill(int dtype, int64_t n, double v, void *y, hipStream_t s);
int dense_gemv(int dtype, int n, const void *M, const void *b, void *x, hipStream_t s);
int block_gs_sweep(pamg_matrix_s *A, void *x, const void *b, const void *Dinv, int row_start,
                   int row_stop, int row_step, hipStream_t s);
int block_jacobi_step(pamg_matrix_s *A, int kind, const void *Dinv, const void *xsrc, void *xdst,
                      const void *b, double omega, hipStream_t s);
int ensure_schedule(pamg_matrix_s *A, int row_start, int row_stop, int row_step, bool block_gs = false);
struct CsrArrays { int64_t m, n, nnz; const int *p, *j; const double *x; };
int csr_device_arrays(struct ::pamg_csr_s *A, CsrArrays *out);                                  // pamg_setup.hip: the device arrays behind a pamg_csr_t
int solver_cycle_inline(pamg_solver_s *S, void *x, const void *b, int cycle, int cpl, hipStream_t s, bool allow_graph);   // pamg_solver.hip
bool solver_needs_host_sync(const pamg_solver_s *S);                                                                   // pamg_solver.hip
// pamg_line.hip: the line-scan fast-order sweep (banded operators in their natural order)
bool line_eligible(const pamg_matrix_s *A, const GsSchedule *g);
int build_line_part(pamg_matrix_s *A, GsSchedule *g);
void free_line_part(LineSched *t);
size_t line_part_bytes(const GsSchedule *g);
int line_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s);
int line_info(const GsSchedule *g, int64_t *info);
// pamg_kz.hip: the lane-parallel fast-order Kaczmarz sweeps (gauss_seidel_ne / gauss_seidel_nr)
int build_kz_lane_part(pamg_matrix_s *Lm, LineSchedule *g);
void free_kz_lane_part(KzLaneSched *t);
int kz_lane_launch(pamg_matrix_s *Lm, LineSchedule *g, bool nr, void *v, const void *b, const void *Dinv, double omega, void *xout, hipStream_t s);
int kz_lane_info(const LineSchedule *g, int64_t *info);
bool kz_lane_error(LineSchedule *g);
// pamg_block.hip: dispatch of the block kernels
bool want_blanes(const pamg_matrix_s *A, const GsSchedule *g);
int block_point_sweep(pamg_matrix_s *A, GsSchedule *g, void *x, const void *b, int dirn, hipStream_t s);
int get_schedule(pamg_matrix_s *A, int row_start, int row_stop, int row_step, GsSchedule **out);
int ensure_parts(pamg_matrix_s *A, GsSchedule *g, bool block_gs = false);
int device_cus();
// pamg_blane.hip: the lane-parallel fast-order block Gauss-Seidel sweep (BSR, square blocks)
bool blane_eligible(const pamg_matrix_s *A, const GsSchedule *g);
int build_blane_part(pamg_matrix_s *A, GsSchedule *g);
void free_blane_part(BlaneSched *t);
size_t blane_part_bytes(const GsSchedule *g);
int blane_launch(pamg_matrix_s *A, GsSchedule *g, const void *Dinv, void *x, const void *b, hipStream_t s);
int blane_info(const GsSchedule *g, int64_t *info);
int blane_profile(const GsSchedule *g, long long *out, int64_t cap, int64_t *n);
int device_cus_lane();
// pamg_lane.hip: the lane-parallel fast-order sweep
bool lane_eligible(const pamg_matrix_s *A, const GsSchedule *g);
int build_lane_part(pamg_matrix_s *A, GsSchedule *g);
void free_lane_part(LaneSched *t);
size_t lane_part_bytes(const GsSchedule *g);
int lane_launch(pamg_matrix_s *A, GsSchedule *g, int epi, void *x, const void *b, double omega, hipStream_t s);
int lane_info(const GsSchedule *g, int64_t *info);
int lane_profile(const GsSchedule *g, long long *out, int64_t cap, int64_t *n);
int lanem_smax(const pamg_matrix_s *A, const GsSchedule *g);
int build_lanem_part(pamg_matrix_s *A, GsSchedule *g);
void free_lanem_part(struct LaneMSched *t);
size_t lanem_part_bytes(const GsSchedule *g);
int lanem_info(const GsSchedule *g, int64_t *info, double *growth);
int lanem_levels(const GsSchedule *g, int64_t *out, int64_t cap, int64_t *n);
int sweep_error(pamg_matrix_s *A, bool *error);      // spin bound hit since the last call? (caller has synchronised; clears the flag)
inline size_t tsize(int dtype) { return dtype == PAMG_F64 ? 8 : 4; }

// fn(lo, hi) over [0, n) on a few host threads (structure checks and scans over 10^8 stored entries are worth it)
template <typename F>
inline void host_parallel(int64_t n, F fn, int64_t grain = 1 << 20)
{
    const unsigned hw = std::max(1u, std::min(32u, pamg::host_cpus()));
    const int nt = n < 2 * grain ? 1 : (int)std::min<int64_t>(hw, n / grain);
    if (nt <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}

// PAMG_TIMING=1: wall-clock of the host-side phases of upload / analysis / planning on stderr (diagnostics)
struct PhaseTimer {
    const char *name;
    int64_t items;
    timespec t0;
    static bool on() { static const int v = [] { const char *e = getenv("PAMG_TIMING"); return (e && *e && *e != '0') ? 1 : 0; }(); return v != 0; }
    explicit PhaseTimer(const char *n, int64_t k = 0) : name(n), items(k) { if (on()) clock_gettime(CLOCK_MONOTONIC, &t0); }
    ~PhaseTimer()
    {
        if (!on()) return;
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[pamg timing] %-28s %8.3f s  (%lld)\n", name, (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec), (long long)items);
    }
};
}  // namespace pamg
