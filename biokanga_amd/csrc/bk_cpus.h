// bk_cpus.h - how many CPUs this process can actually keep busy (no HIP in here: the host front end and its test harnesses include it).
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

namespace bk {

// CPUs this process can actually keep busy: the hardware threads its affinity mask allows (taskset, a container's cpuset), cut down to
// the cgroup's CPU quota when there is one (cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us of v1).  More runnable threads
// than that only buy throttling: on a box that gives a process 16 CPUs' worth of a 256-thread host, a pool sized by the 256 stalls every
// thread of the process in turn (profiles/NOTES.md, round 4).
inline int effective_cpus()
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const long a = CPU_COUNT(&set);
        if (a >= 1 && a < n) n = a;
    }
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64];
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota > 0 && period > 0) {
        const long lim = (long)((quota + period - 1) / period);
        if (lim >= 1 && lim < n) n = lim;
    }
    return (int)n;
}

}  // namespace bk
