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

}  // extern "C"
