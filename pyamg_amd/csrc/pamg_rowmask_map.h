// pamg_rowmask_map.h -- which rows a workgroup / wave / lane of the row-mask kernels takes (csr_rowmask_kernel,
// csr_rowmask3d_kernel in pamg_kernels.h).  Plain C++ shared by the kernels and by tests/stream_emul.cpp, which replays the
// maps on the CPU: every row exactly once, for every order.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PAMG_HD __host__ __device__ __forceinline__
#else
#define PAMG_HD inline
#endif

namespace pamg {

constexpr int RMASK_BLK = 256;   // rows per workgroup of the linear form (= BLK), lanes per workgroup of both

// linear form: workgroup b -> first row / RMASK_BLK.  chunk > 0: XCD-contiguous eighths (grid = 8 * chunk); share > 0:
// plane-by-plane order, XCD j = b & 7 takes the j-th eighth of every plane, share = workgroups per plane and XCD
PAMG_HD int rowmask_linear_block(int b, int chunk, int share)
{
    if (chunk > 0) return (b & 7) * chunk + (b >> 3);
    if (share > 0) {
        const int i = b >> 3, z = i / share, yi = i - z * share;
        return z * (8 * share) + (b & 7) * share + yi;
    }
    return b;
}

// lattice form: the longest list is (-P, -L, -1, 0, +1, +L, +P); a workgroup takes 64 x wy x kz rows (wy waves)
struct RowMaskLattice {
    int L, P;
    int wy;                      // lattice lines (= waves) per workgroup: 4 or 8
    int tiles_x, tiles_y;        // tiles of 64 rows and of wy lattice lines per plane
    int slab;                    // > 0: tiles_y / 8 -- XCD j (blockIdx & 7) takes the j-th eighth of the lines of every plane
};

// true: the lattice form applies (fills g); grid = workgroups to launch
inline bool rowmask_lattice_plan(int nu, const int *off, int64_t nrows, int kz, bool xcd_slabs, bool eight_lines, RowMaskLattice &g, int &grid)
{
    if (nu != 7 || off[3] != 0 || off[2] != -1 || off[4] != 1 || off[5] <= 1 || off[1] != -off[5] || off[6] <= off[5] || off[0] != -off[6]) return false;
    if (kz != 2 && kz != 4 && kz != 8) return false;
    g.L = off[5]; g.P = off[6];
    if (g.L % 64 != 0 || g.P % g.L != 0 || (g.P / g.L) % 4 != 0 || nrows % g.P != 0 || (nrows / g.P) % kz != 0) return false;
    if (nrows / RMASK_BLK / kz > 0x7fffffff / 2) return false;
    g.wy = (eight_lines && (g.P / g.L) % 8 == 0) ? 8 : 4;
    g.tiles_x = g.L / 64; g.tiles_y = (g.P / g.L) / g.wy;
    g.slab = (xcd_slabs && g.tiles_y % 8 == 0) ? g.tiles_y / 8 : 0;
    grid = (int)(nrows / ((int64_t)64 * g.wy * kz));
    return true;
}

// the row of lane `lane` of wave `wave` of workgroup `bid` in its first plane; its other rows are + j * P, j < kz
PAMG_HD int rowmask_tile_row0(const RowMaskLattice &g, int kz, int bid, int wave, int lane)
{
    int tz, ty, tx;
    if (g.slab > 0) {
        const int per = g.slab * g.tiles_x, i = bid >> 3;
        tz = i / per;
        const int t = i - tz * per;
        ty = (bid & 7) * g.slab + t / g.tiles_x;
        tx = t % g.tiles_x;
    } else {
        const int tpp = g.tiles_x * g.tiles_y;
        tz = bid / tpp;
        const int t = bid - tz * tpp;
        ty = t / g.tiles_x;
        tx = t - ty * g.tiles_x;
    }
    return tz * kz * g.P + (ty * g.wy + wave) * g.L + tx * 64 + lane;
}

}  // namespace pamg
