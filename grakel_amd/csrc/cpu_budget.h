/* How many runnable host threads this process may have at once -- ONE helper for libgk_hip.so (gram.hip: the widening
 * threads of the host copy) and _gk_ingest.so (ingest.c: the threaded input walks); plain C, header-only.
 *
 * min(online CPUs, affinity mask, CPU quota of the container's cgroup - 2).  The quota is what matters on a shared box
 * (cpu.max "1600000 100000" = 16 CPUs on the MI355X boxes of this project, which show 256 hardware threads): more runnable
 * threads than the quota do not go faster, they get the whole cgroup THROTTLED for the rest of the 100 ms period -- with 32
 * widening threads one call in twenty took 30-36 ms instead of 5 (round 5, profiles/r05_cpu_quota.txt).  Two CPUs of the
 * quota stay free for the calling thread (it spins in hipEventSynchronize) and the runtime's own threads.
 *
 * The result is cached in a word that is only ever written with the same value (relaxed atomics: no data race). */
#ifndef GK_CPU_BUDGET_H
#define GK_CPU_BUDGET_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { int online, affinity, quota_cpus, budget; } gk_cpu_info_t;   /* quota_cpus 0 = no quota */

static inline void gk_cpu_info(gk_cpu_info_t* o) {
    long c = sysconf(_SC_NPROCESSORS_ONLN);
    int n = c > 0 ? (int)c : 1;
    o->online = n;
    o->affinity = n;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a >= 1) o->affinity = a;
        if (a >= 1 && a < n) n = a;
    }
    long long quota = 0, period = 0;
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");                        /* cgroup v2 */
    if (f) {
        char a[64];
        if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0) quota = atoll(a);
        fclose(f);
    } else {                                                               /* cgroup v1 */
        FILE* q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        if (q) { if (fscanf(q, "%lld", &quota) != 1) quota = 0; fclose(q); }
        q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (q) { if (fscanf(q, "%lld", &period) != 1) period = 0; fclose(q); }
    }
    o->quota_cpus = 0;
    if (quota > 0 && period > 0) {
        int k = (int)((quota + period - 1) / period);
        o->quota_cpus = k;
        if (k > 4) k -= 2;
        if (k >= 1 && k < n) n = k;
    }
    o->budget = n;
}

static inline int gk_cpu_budget(void) {
    static int cached = 0;
    int v = __atomic_load_n(&cached, __ATOMIC_RELAXED);
    if (v) return v;
    gk_cpu_info_t o;
    gk_cpu_info(&o);
    __atomic_store_n(&cached, o.budget, __ATOMIC_RELAXED);
    return o.budget;
}
#endif
