// Device -> host copy rate of an 800 MB matrix into pinned vs pageable host memory, and the cost of pinning
// (grakel_amd.engine.PinnedPool).   hipcc --offload-arch=gfx950 -O3 -o pinbw pinbw.hip && ./pinbw
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)10000 * 10000 * 8;
    void* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    void* pageable = malloc(bytes);
    memset(pageable, 0, bytes);                       // touch the pages
    double t0 = now();
    void* pinned = nullptr;
    if (hipHostMalloc(&pinned, bytes, hipHostMallocDefault) != hipSuccess) return 1;
    printf("hipHostMalloc of %zu MB: %.1f ms (%.3f ms per MB)\n", bytes >> 20, (now() - t0) * 1e3, (now() - t0) * 1e3 / (bytes >> 20));
    for (int kind = 0; kind < 2; ++kind) {
        double best = 1e9;
        for (int it = 0; it < 4; ++it) {
            (void)hipDeviceSynchronize();
            t0 = now();
            (void)hipMemcpy(kind ? pinned : pageable, d, bytes, hipMemcpyDeviceToHost);
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("D2H into %s: %.1f ms  %.1f GB/s\n", kind ? "pinned  " : "pageable", best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
