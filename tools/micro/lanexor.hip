#include <hip/hip_runtime.h>
template <int J> __device__ __forceinline__ int lane_xor(int x) {
    if (J == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
    if (J == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
    if (J == 4) { int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false); return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false); }
    if (J == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);
    if (J == 16) return __builtin_amdgcn_ds_swizzle(x, 0x401F);
    return __shfl_xor(x, 32, 64);
}
__global__ void k(int* out) {
    int x = threadIdx.x * 7 + 3;
    out[threadIdx.x] = lane_xor<1>(x); out[64 + threadIdx.x] = lane_xor<2>(x); out[128 + threadIdx.x] = lane_xor<4>(x);
    out[192 + threadIdx.x] = lane_xor<8>(x); out[256 + threadIdx.x] = lane_xor<16>(x); out[320 + threadIdx.x] = lane_xor<32>(x);
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[384 + threadIdx.x] = r[0]; out[448 + threadIdx.x] = r[1];
#endif
}
int main() {
    int* d; hipMalloc(&d, 512 * 4); hipMemset(d, 0, 2048);
    k<<<1, 64>>>(d); int h[512]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
    int bad = 0; const int J[6] = {1, 2, 4, 8, 16, 32};
    for (int s = 0; s < 6; ++s) for (int i = 0; i < 64; ++i) if (h[s * 64 + i] != (i ^ J[s]) * 7 + 3) { ++bad; if (bad < 10) printf("J=%d lane %d got %d want %d\n", J[s], i, h[s*64+i], (i ^ J[s]) * 7 + 3); }
    printf("bad %d; permlane32_swap: r0[0]=%d r0[32]=%d r1[0]=%d r1[32]=%d\n", bad, h[384], h[384+32], h[448], h[448+32]);
    return bad != 0;
}
