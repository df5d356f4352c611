// How fast can host threads WRITE the 800 MB float64 matrix of 10 000 graphs (the widening half of the compact copy,
// gram.hip: gram_copy_out)?  Destination pinned (hipHostMalloc default / non-coherent / NUMA-user) or malloc'd, threads
// unpinned or pinned to the CPUs of one NUMA node, 8..64 threads, non-temporal or plain stores; the source is 200 MB of
// uint16 in pinned memory.     hipcc --offload-arch=gfx950 -O3 -pthread -o hostwrite hostwrite.hip && ./hostwrite
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <chrono>
#include <emmintrin.h>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<int> node_cpus(int node) {
    std::vector<int> out;
    char path[128];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return out;
    char buf[4096];
    if (fgets(buf, sizeof buf, f)) {
        char* p = buf;
        while (*p) {
            int a = (int)strtol(p, &p, 10), b = a;
            if (*p == '-') b = (int)strtol(p + 1, &p, 10);
            for (int c = a; c <= b; ++c) out.push_back(c);
            if (*p == ',') ++p; else break;
        }
    }
    fclose(f);
    return out;
}
static int node_of(void* addr) {         // get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR)
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0, addr, 3) != 0) return -1;
    return node;
}

static void widen(const uint16_t* src, double* dst, size_t n, bool nt) {
    if (nt) for (size_t i = 0; i + 2 <= n; i += 2) _mm_stream_pd(dst + i, _mm_set_pd((double)src[i + 1], (double)src[i]));
    else for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
    _mm_sfence();
}

static double run(const uint16_t* src, double* dst, size_t n, int threads, const std::vector<int>& cpus, bool nt) {
    double best = 1e9;
    for (int it = 0; it < 4; ++it) {
        const double t0 = now();
        std::vector<std::thread> pool;
        for (int w = 0; w < threads; ++w)
            pool.emplace_back([&, w] {
                if (!cpus.empty()) {
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    CPU_SET(cpus[(size_t)w % cpus.size()], &set);
                    sched_setaffinity(0, sizeof set, &set);
                }
                const size_t lo = n * (size_t)w / (size_t)threads & ~(size_t)7, hi = (w + 1 == threads) ? n : (n * (size_t)(w + 1) / (size_t)threads & ~(size_t)7);
                widen(src + lo, dst + lo, hi - lo, nt);
            });
        for (auto& t : pool) t.join();
        const double dt = now() - t0;
        if (dt < best) best = dt;
    }
    return best;
}

int main() {
    const size_t n = (size_t)10000 * 10000;
    uint16_t* src = nullptr;
    if (hipHostMalloc((void**)&src, n * 2, hipHostMallocDefault) != hipSuccess) return 1;
    for (size_t i = 0; i < n; ++i) src[i] = (uint16_t)i;
    int n_nodes = 0;
    while (!node_cpus(n_nodes).empty()) ++n_nodes;
    printf("NUMA nodes: %d; hardware threads %u; source (pinned) on node %d\n", n_nodes, std::thread::hardware_concurrency(), node_of(src));
    struct Dst { const char* name; double* p; };
    std::vector<Dst> dsts;
    double* q = nullptr;
    if (hipHostMalloc((void**)&q, n * 8, hipHostMallocDefault) == hipSuccess) dsts.push_back({"pinned default", q});
    if (hipHostMalloc((void**)&q, n * 8, hipHostMallocNonCoherent) == hipSuccess) dsts.push_back({"pinned non-coherent", q});
    if (hipHostMalloc((void**)&q, n * 8, hipHostMallocNumaUser) == hipSuccess) dsts.push_back({"pinned numa-user", q});
    q = (double*)aligned_alloc(4096, n * 8);
    memset(q, 0, n * 8);
    dsts.push_back({"malloc (touched by main)", q});
    for (auto& d : dsts) {
        memset(d.p, 0, n * 8);
        printf("%-26s node %d\n", d.name, node_of(d.p));
        for (int threads : {8, 16, 32, 64}) {
            printf("   %2d threads:", threads);
            printf("  unpinned nt %.2f ms", run(src, d.p, n, threads, {}, true) * 1e3);
            printf("  plain %.2f", run(src, d.p, n, threads, {}, false) * 1e3);
            for (int node = 0; node < n_nodes && node < 4; ++node)
                printf("  | node%d nt %.2f", node, run(src, d.p, n, threads, node_cpus(node), true) * 1e3);
            printf("\n");
        }
    }
    return 0;
}
