// pamg_plan_vec.h -- the big arrays of the host-side sweep layouts (plain C++).  A std::vector whose resize() leaves trivially
// constructible elements uninitialised, and a fill on several threads: a 1.5 GB layout is then first touched (page faults) by all
// of them instead of being zero-filled by one.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

namespace pamg {

template <typename T, typename A = std::allocator<T>>
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
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
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
