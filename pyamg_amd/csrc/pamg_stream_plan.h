// pamg_stream_plan.h -- host-side plans of the compressed operator streams (plain C++, no HIP): shared by
// csrc/pamg_matrix.hip and the CPU replay tests/stream_emul.cpp.
//
//   column windows   per row range up to four windows of 16 K columns; an entry stores window << 14 | (column - base)
//   value codes      operators with <= 256 distinct values (bit patterns): one byte per value + the dictionary
//   row patterns     square operators whose rows are mostly one of <= 255 lists of (column - row, value code) pairs:
//                    one byte per row (255 = the row is walked through the code arrays) + the table of lists
//
// Every form delivers the very same (column, value) pairs in the row's storage order: the kernels that read them add the
// same products in the same order.
#pragma once
#include "pamg_host_threads.h"
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace pamg {

template <typename F>
inline void plan_parallel(int64_t n, F fn, int64_t grain)
{
    const unsigned hw = std::max(1u, std::min(64u, pamg::host_cpus()));
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(hw, n / std::max<int64_t>(1, grain)));
    if (nt == 1) { fn((int64_t)0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        th.emplace_back([=] { fn(lo, hi, t); });
    }
    for (auto &x : th) x.join();
}
inline int plan_threads(int64_t n, int64_t grain)
{
    const unsigned hw = std::max(1u, std::min(64u, pamg::host_cpus()));
    return (int)std::max<int64_t>(1, std::min<int64_t>(hw, n / std::max<int64_t>(1, grain)));
}

// ---------------------------------------------------------------------------------------------- row ranges
// Greedy split of rows [begin, end) of a CSR row pointer into workgroup row ranges holding at most `cap` stored entries and
// `max_rows` rows.  A row longer than `cap` gets a range of its own (the kernel streams it in chunks).
struct RowRange { int r0, r1, p0, p1; };

inline void plan_row_ranges(const int *Ap, int begin, int end, int cap, int max_rows, std::vector<RowRange> &out)
{
    int r = begin;
    while (r < end) {
        const int p0 = Ap[r];
        int e = r + 1;
        while (e < end && e - r < max_rows && Ap[e + 1] - p0 <= cap) ++e;
        out.push_back(RowRange{r, e, p0, Ap[e]});
        r = e;
    }
}

// ---------------------------------------------------------------------------------------------- column windows
// Columns of entries [p0, p1) of one row range -> up to four window bases (greedy over the sorted columns) and a 16-bit code
// per entry.  false: the range needs a fifth window.  scratch: reused between calls.
inline bool plan_range_windows(const int *Aj, int p0, int p1, int base[4], unsigned short *code, std::vector<int> &scratch)
{
    base[0] = base[1] = base[2] = base[3] = 0;
    if (p1 <= p0) return true;
    scratch.assign(Aj + p0, Aj + p1);
    std::sort(scratch.begin(), scratch.end());
    int nw = 0;
    for (int v : scratch) {
        if (nw == 0 || v >= base[nw - 1] + 16384) {
            if (nw == 4) return false;
            base[nw++] = v;
        }
    }
    for (int p = p0; p < p1; ++p) {
        const int v = Aj[p];
        int w = nw - 1;
        while (w > 0 && v < base[w]) --w;
        code[p] = (unsigned short)((w << 14) | (v - base[w]));
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- value codes
// v: n values as bit patterns (U = uint64_t for f64, uint32_t for f32).  true: dict holds the <= 256 distinct patterns in
// increasing order and code[p] the position of v[p] in it (code is sized n + pad).
template <typename U>
inline bool plan_value_codes(int64_t n, const U *v, std::vector<U> &dict, std::vector<unsigned char> &code, size_t pad = 16)
{
    dict.clear();
    const int nt = plan_threads(n, 1 << 20);
    std::vector<std::vector<U>> sets((size_t)nt);
    std::atomic<int> over(0);
    plan_parallel(n, [&](int64_t lo, int64_t hi, int t) {
        std::vector<U> &d = sets[(size_t)t];
        U last = 0;
        bool have = false;
        for (int64_t p = lo; p < hi; ++p) {
            const U x = v[p];
            if (have && x == last) continue;
            last = x; have = true;
            auto it = std::lower_bound(d.begin(), d.end(), x);
            if (it != d.end() && *it == x) continue;
            if (d.size() == 256 || over.load(std::memory_order_relaxed)) { over = 1; return; }
            d.insert(it, x);
        }
    }, 1 << 20);
    if (over.load()) return false;
    for (auto &d : sets) dict.insert(dict.end(), d.begin(), d.end());
    std::sort(dict.begin(), dict.end());
    dict.erase(std::unique(dict.begin(), dict.end()), dict.end());
    if (dict.size() > 256 || dict.empty()) { dict.clear(); return false; }
    code.assign((size_t)n + pad, 0);
    plan_parallel(n, [&](int64_t lo, int64_t hi, int) {
        U last = 0;
        unsigned char lc = 0;
        bool have = false;
        for (int64_t p = lo; p < hi; ++p) {
            const U x = v[p];
            if (!have || x != last) {
                last = x; have = true;
                lc = (unsigned char)(std::lower_bound(dict.begin(), dict.end(), x) - dict.begin());
            }
            code[(size_t)p] = lc;
        }
    }, 1 << 20);
    return true;
}

// ---------------------------------------------------------------------------------------------- row patterns
constexpr int RPAT_LMAX = 32;                 // entries per list
constexpr int RPAT_TABLE_BYTES = 24 * 1024;   // LDS the table may take
constexpr int RPAT_IRREGULAR = 255;           // list number of a row that is walked through the code arrays

struct RowPatKey {
    int len;
    int off[RPAT_LMAX];
    unsigned char vc[RPAT_LMAX];
    bool operator==(const RowPatKey &o) const
    {
        return len == o.len && std::memcmp(off, o.off, sizeof(int) * (size_t)len) == 0 && std::memcmp(vc, o.vc, (size_t)len) == 0;
    }
    uint64_t hash() const
    {
        uint64_t h = 1469598103934665603ull ^ (uint64_t)len;
        for (int j = 0; j < len; ++j) {
            h = (h ^ (uint64_t)(uint32_t)off[j]) * 1099511628211ull;
            h = (h ^ (uint64_t)vc[j]) * 1099511628211ull;
        }
        return h;
    }
};

inline bool rowpat_key(const int *Ap, const int *Aj, const unsigned char *code, int64_t r, RowPatKey &k)
{
    const int lo = Ap[r], len = Ap[r + 1] - lo;
    if (len > RPAT_LMAX) return false;
    k.len = len;
    for (int j = 0; j < len; ++j) {
        k.off[j] = Aj[lo + j] - (int)r;
        k.vc[j] = code[(size_t)lo + j];
    }
    return true;
}

// true: pid[r] = number of row r's list (RPAT_IRREGULAR: none), keys = the lists of the table (most frequent first),
// lmax = entries per table row (even).  value_size: bytes per value in the device table (8 or 4) -- it bounds the table.
inline bool plan_row_patterns(int64_t n, const int *Ap, const int *Aj, const unsigned char *code, size_t value_size,
                              std::vector<unsigned char> &pid, std::vector<RowPatKey> &keys, int &lmax, size_t pad = 16)
{
    keys.clear();
    lmax = 0;
    struct Seen { uint64_t h; int64_t count; int64_t row; };
    const int nt = plan_threads(n, 1 << 16);
    std::vector<std::vector<Seen>> seen((size_t)nt);
    std::atomic<int> over(0);
    plan_parallel(n, [&](int64_t lo, int64_t hi, int t) {
        std::vector<Seen> &S = seen[(size_t)t];
        RowPatKey k;
        size_t last = 0;
        for (int64_t r = lo; r < hi; ++r) {
            if (!rowpat_key(Ap, Aj, code, r, k)) continue;                     // a long row: irregular by definition
            const uint64_t h = k.hash();
            if (last < S.size() && S[last].h == h) { ++S[last].count; continue; }
            size_t q = 0;
            while (q < S.size() && S[q].h != h) ++q;
            if (q == S.size()) {
                if (S.size() >= 4096 || over.load(std::memory_order_relaxed)) { over = 1; return; }   // no stencil: too many different rows
                S.push_back(Seen{h, 0, r});
            }
            ++S[q].count;
            last = q;
        }
    }, 1 << 16);
    if (over.load()) return false;
    std::vector<Seen> all;
    for (auto &S : seen)
        for (const Seen &e : S) {
            size_t q = 0;
            while (q < all.size() && all[q].h != e.h) ++q;
            if (q == all.size()) all.push_back(e);
            else { all[q].count += e.count; all[q].row = std::min(all[q].row, e.row); }
            if (all.size() > 16384) return false;
        }
    std::sort(all.begin(), all.end(), [](const Seen &a, const Seen &b) { return a.count != b.count ? a.count > b.count : a.row < b.row; });
    int64_t covered = 0;
    lmax = 1;
    for (const Seen &e : all) {
        if (keys.size() == (size_t)RPAT_IRREGULAR) break;
        RowPatKey k;
        if (!rowpat_key(Ap, Aj, code, e.row, k)) continue;
        const int lm = (std::max(lmax, k.len) + 1) & ~1;
        if ((int64_t)(keys.size() + 1) * lm * (int64_t)(sizeof(int) + value_size) + 1024 > RPAT_TABLE_BYTES) break;
        lmax = lm;
        keys.push_back(k);
        covered += e.count;
    }
    if (keys.empty() || covered * 10 < n * 9) { keys.clear(); return false; }
    lmax = (lmax + 1) & ~1;                           // keeps the value table 8-byte aligned behind the offsets
    std::vector<uint64_t> kh(keys.size());
    for (size_t q = 0; q < keys.size(); ++q) kh[q] = keys[q].hash();
    pid.assign((size_t)n + pad, (unsigned char)RPAT_IRREGULAR);
    plan_parallel(n, [&](int64_t lo, int64_t hi, int) {
        RowPatKey k;
        size_t last = 0;
        for (int64_t r = lo; r < hi; ++r) {
            if (!rowpat_key(Ap, Aj, code, r, k)) continue;
            const uint64_t h = k.hash();
            size_t q = last;
            if (kh[q] != h) { q = 0; while (q < kh.size() && kh[q] != h) ++q; }
            if (q < kh.size() && keys[q] == k) { pid[(size_t)r] = (unsigned char)q; last = q; }   // equal lists, not just equal hashes
        }
    }, 1 << 16);
    return true;
}

// ---------------------------------------------------------------------------------------------- row masks
// The row-mask form of the row patterns (csr_rowmask_kernel): U = the list of the table whose sub-lists cover the most rows (a
// stencil's interior row); a list is expressible when it is U with entries left out -- a subsequence of U, offset AND value code equal at the matched positions, so that walking
// U's entries under the mask adds the row's products in the row's storage order.  mask[r] = bit k set when entry k of U is
// in row r's list; 0 = walk the row through the CSR arrays (irregular rows, rows of lists that are no sub-list, empty rows).
// false: no such form (U longer than 8 entries, or more than a tenth of the rows would walk the CSR arrays).
constexpr int RMASK_MAX = 8;

struct RowMaskPlan {
    int nu = 0;
    int off[RMASK_MAX] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned char vc[RMASK_MAX] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t walked = 0;            // rows with mask 0
};

inline bool plan_row_masks(int64_t n, const std::vector<unsigned char> &pid, const std::vector<RowPatKey> &keys, RowMaskPlan &M,
                           std::vector<unsigned char> &mask, size_t pad = 16)
{
    M = RowMaskPlan();
    mask.clear();
    if (keys.empty() || (int64_t)pid.size() < n) return false;
    // U = the list (of at most RMASK_MAX entries) whose sub-lists cover the most rows
    std::vector<int64_t> hist(256, 0);
    for (int64_t r = 0; r < n; ++r) ++hist[pid[(size_t)r]];
    auto sub_bits = [&](const RowPatKey &K, const RowPatKey &U) -> int {          // K as a mask over U, -1: no sub-list
        int k = 0, bits = 0;
        for (int j = 0; j < K.len; ++j) {
            while (k < U.len && !(U.off[k] == K.off[j] && U.vc[k] == K.vc[j])) ++k;
            if (k == U.len) return -1;
            bits |= 1 << k;
            ++k;
        }
        return bits;
    };
    std::vector<int64_t> cov(keys.size(), -1);
    int64_t best = -1;
    for (size_t c = 0; c < keys.size(); ++c) {
        if (keys[c].len < 1 || keys[c].len > RMASK_MAX) continue;
        cov[c] = 0;
        for (size_t q = 0; q < keys.size(); ++q)
            if (keys[q].len > 0 && sub_bits(keys[q], keys[c]) > 0) cov[c] += hist[q];
        best = std::max(best, cov[c]);
    }
    size_t u = keys.size();                       // within 1 % of the best coverage: the shortest list (fewest gathers per row)
    for (size_t c = 0; c < keys.size(); ++c)
        if (cov[c] >= 0 && cov[c] * 100 >= best * 99 && (u == keys.size() || keys[c].len < keys[u].len)) u = c;
    if (u == keys.size()) return false;
    const RowPatKey &U = keys[u];
    M.nu = U.len;
    for (int k = 0; k < U.len; ++k) { M.off[k] = U.off[k]; M.vc[k] = U.vc[k]; }
    std::vector<int> mask_of(256, 0);
    for (size_t q = 0; q < keys.size(); ++q) {
        const int bits = sub_bits(keys[q], U);
        mask_of[q] = bits > 0 ? bits : 0;
    }
    mask.assign((size_t)n + pad, 0);
    std::atomic<int64_t> walked(0);
    plan_parallel(n, [&](int64_t lo, int64_t hi, int) {
        int64_t w = 0;
        for (int64_t r = lo; r < hi; ++r) {
            const unsigned char mk = (unsigned char)mask_of[pid[(size_t)r]];
            mask[(size_t)r] = mk;
            w += mk == 0;
        }
        walked += w;
    }, 1 << 18);
    M.walked = walked.load();
    if (M.walked * 10 > n) { mask.clear(); return false; }
    return true;
}

}  // namespace pamg
