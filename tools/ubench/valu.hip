// Development aid: issue / dependent-issue cost of VALU, SDWA and LDS instructions for ONE wave on gfx950 (cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int n, int seed) {
    __shared__ int buf[256];
    int a = seed + threadIdx.x, b = seed * 3, c = seed * 5, d = seed * 7, e = seed ^ 9;
    buf[threadIdx.x] = a; buf[threadIdx.x + 64] = b;
    __syncthreads();
    unsigned la = (unsigned)(threadIdx.x * 4);
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) asm volatile(REP32("v_max_i32 %0, %0, %1\n\tv_min_i32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));             // 64 dependent
        if (MODE == 1) asm volatile(REP32("v_max_i32 %0, %0, %4\n\tv_min_i32 %1, %1, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));   // 2 independent chains
        if (MODE == 2) asm volatile(REP32("v_max_i32 %0, %0, %4\n\tv_min_i32 %1, %1, %4\n\tv_max_i32 %2, %2, %4\n\tv_min_i32 %3, %3, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));   // 4 independent (128 instrs)
        if (MODE == 3) asm volatile(REP32("v_max_i32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_min_i32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t") : "+v"(a) : "v"(b), "v"(c));   // 64 dependent SDWA
        if (MODE == 4) asm volatile(REP32("v_mad_i32_i24 %0, %0, %1, %2\n\tv_med3_i32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));   // 64 dependent VOP3
        if (MODE == 5) asm volatile(REP32("ds_write_b8 %1, %0\n\tv_max_i32 %0, %0, %2\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "+v"(a) : "v"(la), "v"(b) : "memory");   // 32 x (LDS store + 1 VALU)
        if (MODE == 6) asm volatile(REP32("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(a) : "v"(la) : "memory");           // 32 dependent LDS round trips
        if (MODE == 7) asm volatile(REP32("v_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));       // 64 dependent packed
        if (MODE == 8) asm volatile(REP32("v_cndmask_b32 %0, %0, %1, vcc\n\tv_xor_b32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));    // 64 dependent
        if (MODE == 9) asm volatile(REP32("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));             // 64 dependent fp32
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 0x12345) out[1000] = a;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    const int n = 200;
    const char* names[] = {"64 dependent VOP2 (max/min)", "2 independent chains x32", "4 independent chains x32 (128)", "64 dependent SDWA", "64 dependent VOP3 (mad/med3)",
                           "32 x (ds_write + VALU)", "32 dependent LDS round trips", "64 dependent packed i16", "64 dependent cndmask/xor", "64 dependent fp32 mul/add"};
    const int count[] = {64, 64, 128, 64, 64, 64, 32, 64, 64, 64};
    auto run = [&](auto kern, int m, int grid) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, n, 3);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-34s grid %5d: %.2f cycles per instruction (%d per trip)\n", names[m], grid, (double)h / n / count[m], count[m]);
    };
    for (int grid : {1, 2048}) {
        run(k<0>, 0, grid); run(k<1>, 1, grid); run(k<2>, 2, grid); run(k<3>, 3, grid); run(k<4>, 4, grid);
        run(k<5>, 5, grid); run(k<6>, 6, grid); run(k<7>, 7, grid); run(k<8>, 8, grid); run(k<9>, 9, grid);
    }
    return 0;
}
