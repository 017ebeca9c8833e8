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

// Under a CPU quota far below the CPUs the process may run on (16 of 256 on the GPU boxes) the scheduler lets the threads wander over
// every core of both sockets: caches are cold wherever a thread lands and half of the memory is on the other NUMA node.  Measured on
// the command line's host side (tools/host_numa.sh): format workers 22.8 -> 8.7 CPU-seconds per 33.5 M reads, 12.5 -> 15.3 M reads/s.
// bsx_pin_to_node restricts the process to the first `n_cpus` CPUs of NUMA node `node` that it may use (the node the GPU hangs on);
// threads created afterwards inherit the mask.  Returns the number of CPUs in the new mask, 0 if nothing was changed.
// `skip`: leave out the node's first `skip` usable CPUs (lane k of several on one node takes CPUs [k n, k n + n)).
inline unsigned bsx_pin_to_node(int node, unsigned n_cpus, unsigned skip = 0)
{
    cpu_set_t have, want;
    if (node < 0 || sched_getaffinity(0, sizeof have, &have) != 0) return 0;
    // nothing to gain where the quota does not bite: the mask IS the share (skip == 0), or it is too small to hold this lane's share behind `skip` others'
    // (a mask that the lanes' shares partition exactly still pins its last lane: an unpinned one would float over the other lanes' CPUs)
    if (skip == 0 ? (unsigned)CPU_COUNT(&have) <= n_cpus : (unsigned)CPU_COUNT(&have) < n_cpus + skip) return 0;
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return 0;
    CPU_ZERO(&want);
    unsigned got = 0;
    for (char *q = buf; *q && got < n_cpus;) {   // "0-63,128-191"
        char *e;
        long a = strtol(q, &e, 10), b = a;
        if (e == q) break;
        if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
        for (long c = a; c <= b && got < n_cpus; c++)
            if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET((int)c, &have)) { if (skip) { skip--; continue; } CPU_SET((int)c, &want); got++; }
        q = (*e == ',') ? e + 1 : e;
        if (*e != ',' && *e != '-') break;
    }
    if (got == 0 || sched_setaffinity(0, sizeof want, &want) != 0) return 0;
    return got;
}
