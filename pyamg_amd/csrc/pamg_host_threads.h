// pamg_host_threads.h -- how many host threads a planner may count on (plain C++).  std::thread::hardware_concurrency() reports the machine;
// a container sees what its cgroup grants.  The GPU boxes of this pool show 256 hardware threads and a quota of 16 cores (cpu.max =
// "1600000 100000"): a planner that started 96 threads per sweep direction there got 16 cores' worth of time slices, and LESS work done than
// 16 threads would (tools/cpu_quota_probe.py: 8 / 16 / 32 / 64 / 128 busy threads = 8.3 / 10.8 / 9.5 / 7.2 / 6.8 x one thread).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <sched.h>

namespace pamg {

// cores a cgroup-v2 "cpu.max" file grants ("<quota> <period>" | "max <period>"), rounded up; 0 = no quota / unreadable
inline unsigned cgroup_quota_cores(const char *path)
{
    long long quota = -1, period = -1;
    if (FILE *f = fopen(path, "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    }
    if (quota > 0 && period > 0) return (unsigned)std::max<long long>(1, (quota + period - 1) / period);
    return 0u;
}

inline unsigned host_cpus_now()
{
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (const char *e = getenv("PAMG_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = std::min(hw, (unsigned)c); }
    const char *v2 = getenv("PAMG_CGROUP_CPU_MAX");                      // (tests point this at a file of their own)
    unsigned q = cgroup_quota_cores(v2 ? v2 : "/sys/fs/cgroup/cpu.max");
    if (!q && !v2) {                                                     // cgroup v1: two files
        long long quota = -1, period = -1;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
        if (quota > 0 && period > 0) q = (unsigned)std::max<long long>(1, (quota + period - 1) / period);
    }
    if (q) hw = std::min(hw, q);
    return std::max(1u, hw);
}

inline unsigned host_cpus()
{
    static const unsigned n = host_cpus_now();
    return n;
}

}  // namespace pamg
