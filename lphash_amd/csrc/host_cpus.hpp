// host_cpus.hpp -- how many host threads are worth starting: the CPUs this process may run on, capped by the container's CPU-time quota
// (cgroup v2 cpu.max, v1 cpu.cfs_quota_us).  Under a quota, threads beyond it only buy throttling: on the round-2 GPU boxes 256 CPUs are
// visible and 16 cores' worth of time is granted -- 16 threads did 133 M oracle lookups/s, 256 threads 112 M (tools/host_probe.py).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <thread>

namespace lphq {
inline int usable_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, std::min(n, CPU_COUNT(&set)));
    double quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64];
        if (fscanf(f, "%63s %lf", q, &period) == 2 && q[0] != 'm') quota = atof(q);
        fclose(f);
    } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lf", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lf", &period) != 1) period = -1;
            fclose(h);
        }
    }
    // twice the quota: the quota is an average over 100-ms periods, short parallel sections finish before it bites, and the sustained probe was
    // fastest there as well (16 threads 133 M/s, 32 threads 153 M/s, 64 threads 125 M/s)
    if (quota > 0 && period > 0) n = std::max(1, std::min(n, 2 * (int)std::ceil(quota / period)));
    return n;
}
}  // namespace lphq
