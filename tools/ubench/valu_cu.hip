// Development aid: VALU issue rate of a SIMD shared by several waves on gfx950 (WAVES waves per workgroup, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned long long* out, int n, int seed) {
    int a = seed + threadIdx.x, b = seed * 3, c = seed * 5, d = seed * 7, e = seed ^ 9;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) asm volatile(REP32("v_max_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 1) asm volatile(REP32("v_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 2) asm volatile(REP32("v_mad_i32_i24 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 3) asm volatile(REP32("v_max_i32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_min_i32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 4) asm volatile(REP32("v_perm_b32 %0, %0, %1, %2\n\tv_xor_b32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 5) asm volatile(REP32("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + d + e == 0x12345) out[4000] = a;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    const int n = 200;
    const char* names[] = {"VOP2 max/min", "packed i16 add/max", "VOP3 mad/med3", "SDWA max/min", "perm/xor", "fp32 mul/add"};
    auto run = [&](auto kern, int m, int waves) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, d, n, 3);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-20s %2d waves per CU: %.2f cycles per instruction per wave, %.2f cycles per wave-instruction per SIMD\n", names[m], waves, (double)h / n / 64, (double)h / n / 64 / (waves / 4.0));
    };
    run(k<0, 4>, 0, 4); run(k<0, 8>, 0, 8); run(k<0, 12>, 0, 12); run(k<0, 16>, 0, 16);
    run(k<1, 4>, 1, 4); run(k<1, 12>, 1, 12); run(k<1, 16>, 1, 16);
    run(k<2, 4>, 2, 4); run(k<2, 12>, 2, 12);
    run(k<3, 4>, 3, 4); run(k<3, 12>, 3, 12);
    run(k<4, 12>, 4, 12); run(k<5, 12>, 5, 12);
    return 0;
}
