// Development aid: how many wait states a v_readlane needs behind the VALU instruction that wrote its source VGPR on gfx950 (0: reads the old value).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int* out) {
    int bad = 0;
    for (int it = 0; it < 1000; ++it) {
        int y = it * 64 + threadIdx.x, r, t = 0, u = it;
        if (MODE == 0) asm volatile("v_mov_b32 %1, 0\n\ts_nop 4\n\tv_add_u32 %1, %2, %2\n\tv_readlane_b32 %0, %1, 5" : "=s"(r), "+v"(t) : "v"(y));
        if (MODE == 1) asm volatile("v_mov_b32 %1, 0\n\ts_nop 4\n\tv_add_u32 %1, %2, %2\n\ts_add_u32 %3, %3, 1\n\tv_readlane_b32 %0, %1, 5" : "=s"(r), "+v"(t), "+v"(y), "+s"(u) : : "scc");
        if (MODE == 2) asm volatile("v_mov_b32 %1, 0\n\ts_nop 4\n\tv_add_u32 %1, %2, %2\n\tv_xor_b32 %2, %2, %2\n\tv_readlane_b32 %0, %1, 5" : "=s"(r), "+v"(t), "+v"(y));
        if (MODE == 3) asm volatile("v_mov_b32 %1, 0\n\ts_nop 4\n\tv_add_u32 %1, %2, %2\n\ts_nop 0\n\tv_readlane_b32 %0, %1, 5" : "=s"(r), "+v"(t) : "v"(y));
        if (MODE == 4) asm volatile("v_mov_b32 %1, 0\n\ts_nop 4\n\tv_add_u32_dpp %1, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_readlane_b32 %0, %1, 5" : "=s"(r), "+v"(t) : "v"(y));
        if (MODE == 5) asm volatile("v_mov_b32 %1, 0\n\ts_nop 4\n\tv_add_u32_dpp %1, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_add_u32 %3, %3, 1\n\tv_readlane_b32 %0, %1, 5" : "=s"(r), "+v"(t), "+v"(y), "+s"(u) : : "scc");
        const int expect = MODE >= 4 ? (it * 64 + 5) + (it * 64 + 4) : 2 * (it * 64 + 5);
        const int expect2 = MODE >= 4 ? (it * 64 + 5) + (it * 64 + 6) : expect;
        if (r != expect && r != expect2) ++bad;
    }
    if (threadIdx.x == 0) out[0] = bad;
}
int main() {
    int* d; hipMalloc(&d, 64); int h;
    const char* names[] = {"plain VALU, readlane next", "plain VALU, one SALU between", "plain VALU, one VALU between", "plain VALU, s_nop 0 between", "DPP VALU, readlane next", "DPP VALU, one SALU between"};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("%-32s %d of 1000 reads stale\n", names[M], h);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
