// pamg_spg_plan.h -- host-side task plan of the sparse product (plain C++, no HIP): shared by csrc/pamg_setup.hip and
// the CPU replay tests/spg_emul.cpp.
//
// Tasks in row order: runs of whole rows while their products fit SPG_CAP (and SPG_ROWS rows); a row with up to SPG_CAP2
// products is a whole-row task of its own (the same expand - sort - compress in a workgroup that takes 136 KB of the CU's
// 160 KB of LDS: the rows of the level-1 -> 2 Galerkin products of 3-D problems, 6-8 K products each, which as column
// windows were re-scanned 200-400 times -- 6.4 of the 53 s of the 512^3 setup); a row with more products becomes windows
// of SPL_WIN output columns over the span [lo, hi] of its product columns.
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <vector>

namespace pamg {

constexpr int SPG_CAP = 4096;         // products per whole-row task (LDS: 8 B key + 8 B value each)
constexpr int SPG_CAP2 = 8192;        // products of a single-row task of the big variant (one workgroup per CU)
constexpr int SPG_ROWS = 1024;        // rows per whole-row task (local row: 11 bits of the key)
constexpr int SPL_CAP = 2048;         // products per batch of a long row
constexpr int SPL_WIN = 2048;         // output columns per window of a long row
constexpr int SPG_SLICE = 1 << 22;    // workgroups per launch: a launch must stay below 2^32 threads

struct SpgTask { int row0, row1, col0, col1; };          // col0 = 0, col1 = INT_MAX: whole rows

inline bool spg_whole(const SpgTask &t) { return t.col0 == 0 && t.col1 == INT_MAX; }

// nprod: products per row; lohi: for every row with nprod > SPG_CAP2, in row order, the smallest and largest product column
inline void spg_plan(int m, const std::vector<int> &nprod, const std::vector<int> &lohi, std::vector<SpgTask> &tasks)
{
    tasks.clear();
    tasks.reserve((size_t)m / 64 + 16);
    size_t nl = 0;
    int r = 0;
    while (r < m) {
        if (nprod[r] > SPG_CAP && nprod[r] <= SPG_CAP2) {           // a whole-row task of its own (big variant)
            tasks.push_back(SpgTask{r, r + 1, 0, INT_MAX});
            ++r;
            continue;
        }
        if (nprod[r] > SPG_CAP2) {
            const int lo = lohi[2 * nl], hi = lohi[2 * nl + 1];
            ++nl;
            for (int64_t w = lo; w <= hi; w += SPL_WIN)
                tasks.push_back(SpgTask{r, r + 1, (int)w, (int)std::min<int64_t>(w + SPL_WIN, (int64_t)hi + 1)});
            ++r;
            continue;
        }
        int acc = 0, r1 = r;
        while (r1 < m && r1 - r < SPG_ROWS && nprod[r1] <= SPG_CAP && acc + nprod[r1] <= SPG_CAP) { acc += nprod[r1]; ++r1; }
        tasks.push_back(SpgTask{r, r1, 0, INT_MAX});
        r = r1;
    }
}

}  // namespace pamg
