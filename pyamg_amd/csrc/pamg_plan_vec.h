// pamg_plan_vec.h -- the big arrays of the host-side sweep layouts (plain C++).  A std::vector whose resize() leaves trivially
// constructible elements uninitialised, and a fill on several threads: a 1.5 GB layout is then first touched (page faults) by all
// of them instead of being zero-filled by one.  Large blocks are 2 MB aligned and marked MADV_HUGEPAGE: a hundred planner threads
// first-touching gigabytes in 4 KB pages queue up behind the process's page-table lock (the merged plan of level 1 of the 256^3
// hierarchy spent most of its second in page faults), 2 MB pages are 512 times fewer faults where the kernel grants them.
#pragma once
#include "pamg_host_threads.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include <sys/mman.h>

namespace pamg {

template <typename T>
struct huge_page_allocator {
    using value_type = T;
    huge_page_allocator() = default;
    template <typename U> huge_page_allocator(const huge_page_allocator<U> &) noexcept {}
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        constexpr size_t HUGE = (size_t)2 << 20;
        void *p = nullptr;
        static const bool want_huge = [] { const char *e = getenv("PAMG_PLAN_HUGEPAGES"); return e && *e && *e != '0'; }();
        if (want_huge && bytes >= 2 * HUGE) {
            if (posix_memalign(&p, HUGE, (bytes + HUGE - 1) / HUGE * HUGE) != 0) throw std::bad_alloc();
            madvise(p, (bytes + HUGE - 1) / HUGE * HUGE, MADV_HUGEPAGE);        // a hint: ignored where transparent huge pages are off
        } else {
            p = malloc(std::max<size_t>(bytes, 1));
            if (!p) throw std::bad_alloc();
        }
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) noexcept { free(p); }
    template <typename U> bool operator==(const huge_page_allocator<U> &) const noexcept { return true; }
    template <typename U> bool operator!=(const huge_page_allocator<U> &) const noexcept { return false; }
};

template <typename T, typename A = huge_page_allocator<T>>
struct default_init_allocator : A {
    using A::A;
    template <typename U> struct rebind { using other = default_init_allocator<U, typename std::allocator_traits<A>::template rebind_alloc<U>>; };
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { std::allocator_traits<A>::construct(static_cast<A &>(*this), p, std::forward<Args>(args)...); }
};

template <typename T> using PlanVec = std::vector<T, default_init_allocator<T>>;

template <typename T>
inline void plan_fill(PlanVec<T> &v, size_t n, T value)
{
    v.clear();
    v.resize(n);                                        // uninitialised
    T *p = v.data();
    const size_t grain = (size_t)1 << 22;
    unsigned nt = std::max(1u, std::min(32u, pamg::host_cpus()));
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n / grain));
    if (nt <= 1) { std::fill(p, p + n, value); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned k = 0; k < nt; ++k) {
        const size_t lo = (size_t)k * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=] { std::fill(p + lo, p + hi, value); });
    }
    for (auto &t : th) t.join();
}

}  // namespace pamg
