// TEST INFRASTRUCTURE (CPU): replays the compressed operator streams of pyamg_amd/csrc (host plans: pamg_stream_plan.h) the
// way the whole-operator kernels decode them -- csr_stream_kernel on 16-bit column codes, csr_rowgather_kernel on column
// codes + value codes, csr_rowpat_kernel on row patterns with its irregular rows -- and forms y = A x with each: every
// form must deliver the CSR's own (column, value) pairs in storage order, so the sums carry SciPy's bits.
// tests/test_setup_host.py drives it without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../pyamg_amd/csrc/pamg_stream_plan.h"
#include "../pyamg_amd/csrc/pamg_rowmask_map.h"

using namespace pamg;

extern "C" {

// y16 / y8 / ypat: [n] results of the three forms (NaN where a form does not apply); info: [0] 1 = column windows fit,
// [1] distinct values (0 = more than 256), [2] row-pattern lists (0 = no table), [3] irregular rows, [4] entries whose
// decoded (column, value bits) differ from the CSR's in any form (must be 0), [5] row ranges
int stream_emul_f64(int n, const int *Ap, const int *Aj, const double *Ax, const double *x, int cap, int max_rows,
                    double *y16, double *y8, double *ypat, int64_t *info)
{
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < n; ++i) y16[i] = y8[i] = ypat[i] = nan;
    for (int k = 0; k < 6; ++k) info[k] = 0;
    const int64_t nnz = Ap[n];
    std::vector<RowRange> rr;
    plan_row_ranges(Ap, 0, n, cap, max_rows, rr);
    info[5] = (int64_t)rr.size();
    // ---- column windows per range
    std::vector<unsigned short> c16((size_t)nnz + 16, 0);
    std::vector<int> wb(4 * rr.size(), 0), scratch;
    bool ok16 = true;
    for (size_t b = 0; b < rr.size() && ok16; ++b) ok16 = plan_range_windows(Aj, rr[b].p0, rr[b].p1, &wb[4 * b], c16.data(), scratch);
    info[0] = ok16 ? 1 : 0;
    auto column = [&](size_t b, int64_t p) { const unsigned c = c16[(size_t)p]; return wb[4 * b + (c >> 14)] + (int)(c & 0x3FFFu); };
    int64_t bad = 0;
    if (ok16) {
        for (size_t b = 0; b < rr.size(); ++b)
            for (int r = rr[b].r0; r < rr[b].r1; ++r) {
                double s = 0.0;
                for (int p = Ap[r]; p < Ap[r + 1]; ++p) {
                    const int col = column(b, p);
                    bad += col != Aj[p];
                    s += Ax[p] * x[col];
                }
                y16[r] = s;
            }
    }
    // ---- value codes
    std::vector<uint64_t> dict;
    std::vector<unsigned char> code;
    const bool ok8 = plan_value_codes<uint64_t>(nnz, reinterpret_cast<const uint64_t *>(Ax), dict, code);
    info[1] = ok8 ? (int64_t)dict.size() : 0;
    auto value = [&](int64_t p) { double v; std::memcpy(&v, &dict[code[(size_t)p]], 8); return v; };
    if (ok8 && ok16) {
        for (size_t b = 0; b < rr.size(); ++b)
            for (int r = rr[b].r0; r < rr[b].r1; ++r) {
                double s = 0.0;
                for (int p = Ap[r]; p < Ap[r + 1]; ++p) {
                    const double v = value(p);
                    bad += std::memcmp(&v, &Ax[p], 8) != 0;
                    s += v * x[column(b, p)];
                }
                y8[r] = s;
            }
    }
    // ---- row patterns (square operators with value codes)
    if (ok8 && ok16) {
        std::vector<unsigned char> pid;
        std::vector<RowPatKey> keys;
        int lmax = 0;
        if (plan_row_patterns(n, Ap, Aj, code.data(), 8, pid, keys, lmax)) {
            info[2] = (int64_t)keys.size();
            // the device table: offsets and VALUES per list
            std::vector<int> to(keys.size() * (size_t)lmax, 0);
            std::vector<double> tv(keys.size() * (size_t)lmax, 0.0);
            for (size_t q = 0; q < keys.size(); ++q)
                for (int j = 0; j < keys[q].len; ++j) {
                    to[q * lmax + j] = keys[q].off[j];
                    std::memcpy(&tv[q * lmax + j], &dict[keys[q].vc[j]], 8);
                }
            for (size_t b = 0; b < rr.size(); ++b)
                for (int r = rr[b].r0; r < rr[b].r1; ++r) {
                    double s = 0.0;
                    if (pid[(size_t)r] != RPAT_IRREGULAR) {
                        const size_t q = pid[(size_t)r];
                        if (keys[q].len != Ap[r + 1] - Ap[r]) ++bad;
                        for (int j = 0; j < keys[q].len; ++j) {
                            const int col = r + to[q * lmax + j];
                            const double v = tv[q * lmax + j];
                            bad += (col != Aj[Ap[r] + j]) || std::memcmp(&v, &Ax[Ap[r] + j], 8) != 0;
                            s += v * x[col];
                        }
                    } else {
                        ++info[3];
                        for (int p = Ap[r]; p < Ap[r + 1]; ++p) s += value(p) * x[column(b, p)];
                    }
                    ypat[r] = s;
                }
        }
    }
    info[4] = bad;
    return 0;
}

// The row-mask forms (plan_row_masks + pamg_rowmask_map.h): ylin = the linear form (csr_rowmask_kernel: workgroup -> 256 rows
// under the three workgroup orders), ylat = the lattice form (csr_rowmask3d_kernel's 64 x 4 x kz tiles), both decoded the way
// the kernels decode a row: the longest list under the row's mask, or the CSR arrays for mask 0.
// info: [0] entries of the longest list (0 = no mask form), [1] rows with mask 0, [2] 1 = lattice form applies, [3] decoded
// (column, value) pairs that differ from the CSR's (must be 0), [4] rows not visited exactly once by some order (must be 0),
// [5] L, [6] P, [7] workgroups of the lattice form
int rowmask_emul_f64(int n, const int *Ap, const int *Aj, const double *Ax, const double *x, int kz, double *ylin, double *ylat, int64_t *info)
{
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int i = 0; i < n; ++i) ylin[i] = ylat[i] = nan;
    for (int k = 0; k < 8; ++k) info[k] = 0;
    const int64_t nnz = Ap[n];
    std::vector<uint64_t> dict;
    std::vector<unsigned char> code, pid, mask;
    std::vector<RowPatKey> keys;
    int lmax = 0;
    if (!plan_value_codes<uint64_t>(nnz, reinterpret_cast<const uint64_t *>(Ax), dict, code)) return 0;
    if (!plan_row_patterns(n, Ap, Aj, code.data(), 8, pid, keys, lmax)) return 0;
    RowMaskPlan M;
    if (!plan_row_masks(n, pid, keys, M, mask)) return 0;
    info[0] = M.nu;
    info[1] = M.walked;
    int64_t bad = 0, miss = 0;
    auto row = [&](int r) {
        double s = 0.0;
        const unsigned mk = mask[(size_t)r];
        if (mk) {
            int p = Ap[r];
            for (int k = 0; k < M.nu; ++k)
                if ((mk >> k) & 1u) {
                    double v;
                    std::memcpy(&v, &dict[M.vc[k]], 8);
                    const int col = r + M.off[k];
                    bad += p >= Ap[r + 1] || col != Aj[p] || std::memcmp(&v, &Ax[p], 8) != 0;
                    ++p;
                    s += v * x[col];
                }
            bad += p != Ap[r + 1];
        } else {
            for (int p = Ap[r]; p < Ap[r + 1]; ++p) s += Ax[p] * x[Aj[p]];
        }
        return s;
    };
    // ---- linear form, three workgroup orders
    std::vector<int> seen((size_t)n);
    const int grid0 = (n + RMASK_BLK - 1) / RMASK_BLK;
    const int plane = M.off[M.nu - 1];
    for (int order = 0; order < 3; ++order) {
        int chunk = 0, share = 0, grid = grid0;
        if (order == 1) { chunk = (grid0 + 7) >> 3; grid = 8 * chunk; }
        if (order == 2) {
            if (!(plane >= 8 * RMASK_BLK && plane % (8 * RMASK_BLK) == 0 && n % plane == 0)) continue;
            share = plane / (8 * RMASK_BLK);
        }
        std::fill(seen.begin(), seen.end(), 0);
        for (int b = 0; b < grid; ++b) {
            const int blk = rowmask_linear_block(b, chunk, share);
            for (int t = 0; t < RMASK_BLK; ++t) {
                const int64_t r = (int64_t)blk * RMASK_BLK + t;
                if (r >= n) continue;
                ++seen[(size_t)r];
                ylin[r] = row((int)r);
            }
        }
        for (int r = 0; r < n; ++r) miss += seen[(size_t)r] != 1;
    }
    // ---- lattice form, both workgroup orders
    for (int var = 0; var < 4; ++var) {
        const int slabs = var & 1, eight = var >> 1;
        RowMaskLattice g;
        int grid = 0;
        if (!rowmask_lattice_plan(M.nu, M.off, n, kz, slabs != 0, eight != 0, g, grid)) continue;
        if ((slabs && g.slab == 0) || (eight && g.wy != 8)) continue;
        info[2] = 1; info[5] = g.L; info[6] = g.P; if (!eight) info[7] = grid;
        std::fill(seen.begin(), seen.end(), 0);
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < g.wy; ++w)
                for (int l = 0; l < 64; ++l) {
                    const int r0 = rowmask_tile_row0(g, kz, b, w, l);
                    for (int j = 0; j < kz; ++j) {
                        const int64_t r = (int64_t)r0 + (int64_t)j * g.P;
                        if (r < 0 || r >= n) { ++miss; continue; }
                        ++seen[(size_t)r];
                        ylat[r] = row((int)r);
                    }
                }
        for (int r = 0; r < n; ++r) miss += seen[(size_t)r] != 1;
    }
    info[3] = bad;
    info[4] = miss;
    return 0;
}

}  // extern "C"
