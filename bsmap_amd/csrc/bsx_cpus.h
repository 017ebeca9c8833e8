// Number of CPUs this process can actually use: the hardware thread count, cut down to the scheduler affinity mask and to
// the cgroup CPU quota (a container that shows 256 hardware threads may be allowed 16 CPUs' worth of time; starting one
// worker per hardware thread there only buys throttling, which also stalls the threads that drive the GPU).
#pragma once
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

inline unsigned bsx_usable_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[32];
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {  // cgroup v1
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min(n, (unsigned)std::max(1LL, (quota + period - 1) / period));
    return n;
}
