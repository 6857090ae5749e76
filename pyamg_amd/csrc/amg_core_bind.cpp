// amg_core_bind.cpp -- the pybind11 face of Layer 1: a module with the names, argument order and overload behaviour
// of the reference's pyamg.amg_core relaxation bindings (pyamg/amg_core/relaxation_bind.cpp:681-715: one overload per
// value type, every array `.noconvert()` so a dtype mismatch is a TypeError, sizes taken from the arrays) whose
// bodies do nothing but hand the host pointers to the C ABI of libpyamg_amd.so (include/pyamg_amd.h, pamg_*_f32 /
// pamg_*_f64).  Plain g++ builds it (no HIP here); pyamg_amd/amg_core.py is the ctypes twin of the same surface.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <stdexcept>
#include <string>

#include "../../include/pyamg_amd.h"

namespace py = pybind11;

namespace {

using Idx = py::array_t<int32_t, py::array::c_style>;
template <typename T> using Vec = py::array_t<T, py::array::c_style>;

void done(int status, const char *what)
{
    if (status != PAMG_OK) throw std::runtime_error(std::string(what) + ": pyamg_amd status " + std::to_string(status));
}

template <typename A> int len(const A &a) { return (int)a.size(); }

// the C entry points of one value type
template <typename T> struct L1;
#define PAMG_L1(T, S)                                                                      \
    template <> struct L1<T> {                                                             \
        static constexpr auto csr_matvec = pamg_csr_matvec_##S;                            \
        static constexpr auto bsr_matvec = pamg_bsr_matvec_##S;                            \
        static constexpr auto gauss_seidel = pamg_gauss_seidel_##S;                        \
        static constexpr auto sor_gauss_seidel = pamg_sor_gauss_seidel_##S;                \
        static constexpr auto bsr_gauss_seidel = pamg_bsr_gauss_seidel_##S;                \
        static constexpr auto jacobi = pamg_jacobi_##S;                                    \
        static constexpr auto bsr_jacobi = pamg_bsr_jacobi_##S;                            \
        static constexpr auto jacobi_indexed = pamg_jacobi_indexed_##S;                    \
        static constexpr auto gauss_seidel_ne = pamg_gauss_seidel_ne_##S;                  \
        static constexpr auto gauss_seidel_nr = pamg_gauss_seidel_nr_##S;                  \
        static constexpr auto jacobi_ne = pamg_jacobi_ne_##S;                              \
        static constexpr auto block_jacobi = pamg_block_jacobi_##S;                        \
        static constexpr auto block_gauss_seidel = pamg_block_gauss_seidel_##S;            \
        static constexpr auto block_jacobi_indexed = pamg_block_jacobi_indexed_##S;        \
        static constexpr auto gauss_seidel_indexed = pamg_gauss_seidel_indexed_##S;        \
        static constexpr auto overlapping_schwarz_csr = pamg_overlapping_schwarz_csr_##S;  \
        static constexpr auto pinv_array = pamg_pinv_array_##S;                            \
        static constexpr auto fit_candidates = pamg_fit_candidates_##S;                    \
    };
PAMG_L1(double, f64)
PAMG_L1(float, f32)
#undef PAMG_L1

template <typename T>
void bind(py::module_ &m)
{
    using F = L1<T>;
    auto nc = [](const char *n) { return py::arg(n).noconvert(); };
    m.def("csr_matvec", [](int n_row, int n_col, Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &Xx, Vec<T> &Yx) {
        done(F::csr_matvec(n_row, n_col, Ap.data(), Aj.data(), Ax.data(), Xx.data(), Yx.mutable_data()), "csr_matvec");
    }, py::arg("n_row"), py::arg("n_col"), nc("Ap"), nc("Aj"), nc("Ax"), nc("Xx"), nc("Yx"));
    m.def("bsr_matvec", [](int n_brow, int n_bcol, int R, int C, Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &Xx, Vec<T> &Yx) {
        done(F::bsr_matvec(n_brow, n_bcol, R, C, Ap.data(), Aj.data(), Ax.data(), Xx.data(), Yx.mutable_data()), "bsr_matvec");
    }, py::arg("n_brow"), py::arg("n_bcol"), py::arg("R"), py::arg("C"), nc("Ap"), nc("Aj"), nc("Ax"), nc("Xx"), nc("Yx"));
    m.def("gauss_seidel", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, int row_start, int row_stop, int row_step) {
        done(F::gauss_seidel(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                             row_start, row_stop, row_step), "gauss_seidel");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"));
    m.def("sor_gauss_seidel", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, int row_start, int row_stop, int row_step, T omega) {
        done(F::sor_gauss_seidel(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                 row_start, row_stop, row_step, omega), "sor_gauss_seidel");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"), py::arg("omega"));
    m.def("bsr_gauss_seidel", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, int row_start, int row_stop, int row_step, int blocksize) {
        done(F::bsr_gauss_seidel(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                 row_start, row_stop, row_step, blocksize), "bsr_gauss_seidel");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"), py::arg("blocksize"));
    m.def("jacobi", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &temp, int row_start, int row_stop, int row_step, Vec<T> &omega) {
        done(F::jacobi(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                       temp.mutable_data(), len(temp), row_start, row_stop, row_step, omega.data(), len(omega)), "jacobi");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("temp"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"), nc("omega"));
    m.def("bsr_jacobi", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &temp, int row_start, int row_stop, int row_step,
                           int blocksize, Vec<T> &omega) {
        done(F::bsr_jacobi(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                           temp.mutable_data(), len(temp), row_start, row_stop, row_step, blocksize, omega.data(), len(omega)), "bsr_jacobi");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("temp"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"),
       py::arg("blocksize"), nc("omega"));
    m.def("jacobi_indexed", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Idx &indices, Vec<T> &omega) {
        done(F::jacobi_indexed(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                               indices.data(), len(indices), omega.data(), len(omega)), "jacobi_indexed");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("indices"), nc("omega"));
    m.def("gauss_seidel_ne", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, int row_start, int row_stop, int row_step, Vec<T> &Tx, T omega) {
        done(F::gauss_seidel_ne(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                row_start, row_stop, row_step, Tx.data(), len(Tx), omega), "gauss_seidel_ne");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"), nc("Tx"), py::arg("omega"));
    m.def("gauss_seidel_nr", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &z, int col_start, int col_stop, int col_step, Vec<T> &Tx, T omega) {
        done(F::gauss_seidel_nr(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), z.mutable_data(), len(z),
                                col_start, col_stop, col_step, Tx.data(), len(Tx), omega), "gauss_seidel_nr");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("z"), py::arg("col_start"), py::arg("col_stop"), py::arg("col_step"), nc("Tx"), py::arg("omega"));
    m.def("jacobi_ne", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &Tx, Vec<T> &temp, int row_start, int row_stop, int row_step,
                          Vec<T> &omega) {
        done(F::jacobi_ne(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                          Tx.data(), len(Tx), temp.mutable_data(), len(temp), row_start, row_stop, row_step, omega.data(), len(omega)), "jacobi_ne");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("Tx"), nc("temp"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"), nc("omega"));
    m.def("block_jacobi", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &Tx, Vec<T> &temp, int row_start, int row_stop, int row_step,
                             Vec<T> &omega, int blocksize) {
        done(F::block_jacobi(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                             Tx.data(), len(Tx), temp.mutable_data(), len(temp), row_start, row_stop, row_step, omega.data(), len(omega),
                             blocksize), "block_jacobi");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("Tx"), nc("temp"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"),
       nc("omega"), py::arg("blocksize"));
    m.def("overlapping_schwarz_csr", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &Tx, Idx &Tp, Idx &Sj, Idx &Sp, int nsdomains,
                                        int nrows, int row_start, int row_stop, int row_step) {
        done(F::overlapping_schwarz_csr(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                        Tx.data(), len(Tx), Tp.data(), len(Tp), Sj.data(), len(Sj), Sp.data(), len(Sp), nsdomains, nrows,
                                        row_start, row_stop, row_step), "overlapping_schwarz_csr");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("Tx"), nc("Tp"), nc("Sj"), nc("Sp"), py::arg("nsdomains"), py::arg("nrows"),
       py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"));
    m.def("gauss_seidel_indexed", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Idx &Id, int row_start, int row_stop, int row_step) {
        done(F::gauss_seidel_indexed(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                     Id.data(), len(Id), row_start, row_stop, row_step), "gauss_seidel_indexed");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("Id"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"));
    m.def("block_jacobi_indexed", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &Tx, Idx &indices, Vec<T> &omega, int blocksize) {
        done(F::block_jacobi_indexed(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                     Tx.data(), len(Tx), indices.data(), len(indices), omega.data(), len(omega), blocksize), "block_jacobi_indexed");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("Tx"), nc("indices"), nc("omega"), py::arg("blocksize"));
    m.def("block_gauss_seidel", [](Idx &Ap, Idx &Aj, Vec<T> &Ax, Vec<T> &x, Vec<T> &b, Vec<T> &Tx, int row_start, int row_stop, int row_step,
                                   int blocksize) {
        done(F::block_gauss_seidel(Ap.data(), len(Ap), Aj.data(), len(Aj), Ax.data(), len(Ax), x.mutable_data(), len(x), b.data(), len(b),
                                   Tx.data(), len(Tx), row_start, row_stop, row_step, blocksize), "block_gauss_seidel");
    }, nc("Ap"), nc("Aj"), nc("Ax"), nc("x"), nc("b"), nc("Tx"), py::arg("row_start"), py::arg("row_stop"), py::arg("row_step"), py::arg("blocksize"));
    // amg_core.pinv_array (linalg_bind.cpp:12-30): AA ravelled, (m, n, n), TransA 'T' / 'F'
    m.def("pinv_array", [](Vec<T> &AA, int m_, int n, char TransA) {
        done(F::pinv_array(AA.mutable_data(), len(AA), m_, n, TransA), "pinv_array");
    }, nc("AA"), py::arg("m"), py::arg("n"), py::arg("TransA"));
    // amg_core.fit_candidates, real overloads (smoothed_aggregation_bind.cpp:134-170)
    m.def("fit_candidates", [](int n_row, int n_col, int K1, int K2, Idx &Ap, Idx &Ai, Vec<T> &Ax, Vec<T> &B, Vec<T> &R, T tol) {
        done(F::fit_candidates(n_row, n_col, K1, K2, Ap.data(), len(Ap), Ai.data(), len(Ai), Ax.mutable_data(), len(Ax), B.data(), len(B),
                               R.mutable_data(), len(R), tol), "fit_candidates");
    }, py::arg("n_row"), py::arg("n_col"), py::arg("K1"), py::arg("K2"), nc("Ap"), nc("Ai"), nc("Ax"), nc("B"), nc("R"), py::arg("tol"));
}

}  // namespace

PYBIND11_MODULE(_amg_core_pybind, m)
{
    m.doc() = "pybind11 bindings of the MI355X relaxation / SpMV entry points (Layer 1 of include/pyamg_amd.h), "
              "signature-compatible with pyamg.amg_core";
    bind<float>(m);
    bind<double>(m);
    // amg_core.standard_aggregation (smoothed_aggregation_bind.cpp:49-75): returns the number of aggregates
    m.def("standard_aggregation", [](int n_row, Idx &Ap, Idx &Aj, Idx &x, Idx &y) {
        int naggs = 0;
        done(pamg_standard_aggregation(n_row, Ap.data(), len(Ap), Aj.data(), len(Aj), x.mutable_data(), len(x), y.mutable_data(), len(y), &naggs),
             "standard_aggregation");
        return naggs;
    }, py::arg("n_row"), py::arg("Ap").noconvert(), py::arg("Aj").noconvert(), py::arg("x").noconvert(), py::arg("y").noconvert());
    m.def("version", [] { return std::string(pamg_version()); });
}
