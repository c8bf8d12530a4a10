// Development aid: LDS instruction throughput of a full compute unit on gfx950 (12 waves issuing back-to-back, like an LDPC layer phase).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)
template <int MODE>
__global__ __launch_bounds__(768) void k(unsigned long long* out, int n, int seed) {
    extern __shared__ int buf[];
    for (int i = threadIdx.x; i < 32768; i += 768) buf[i] = i * seed;
    __syncthreads();
    unsigned a0 = (threadIdx.x * (MODE == 2 || MODE == 3 ? 4 : 1) + seed * 360) & 0x1ffff, a1 = (a0 + 4001) & 0x1ffff, a2 = (a0 + 9001) & 0x1ffff, a3 = (a0 + 20011) & 0x1ffff;
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) asm volatile(REP16("ds_read_u8_d16 %0, %4\n\tds_read_u8_d16_hi %1, %5\n\tds_read_u8_d16 %2, %6\n\tds_read_u8_d16_hi %3, %7\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        if (MODE == 1) asm volatile(REP16("ds_write_b8 %4, %0\n\tds_write_b8 %5, %1\n\tds_write_b8 %6, %2\n\tds_write_b8 %7, %3\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        if (MODE == 2) asm volatile(REP16("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0 & ~3u), "v"(a1 & ~3u), "v"(a2 & ~3u), "v"(a3 & ~3u) : "memory");
        if (MODE == 3) asm volatile(REP16("ds_write_b32 %4, %0\n\tds_write_b32 %5, %1\n\tds_write_b32 %6, %2\n\tds_write_b32 %7, %3\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0 & ~3u), "v"(a1 & ~3u), "v"(a2 & ~3u), "v"(a3 & ~3u) : "memory");
        if (MODE == 4) asm volatile(REP16("ds_read_u8_d16 %0, %4\n\tv_max_i32 %1, %1, %5\n\tv_max_i32 %2, %2, %5\n\tv_max_i32 %3, %3, %5\n\t") "s_waitcnt lgkmcnt(0)\n\t" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1) : "memory");   // 1 LDS : 3 VALU
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (r0 + r1 + r2 + r3 == 0x12345) out[4000] = r0;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    const int n = 100;
    const char* names[] = {"ds_read_u8_d16 (consecutive bytes)", "ds_write_b8", "ds_read_b32 (consecutive dwords)", "ds_write_b32", "1 ds_read_u8 : 3 VALU"};
    auto run = [&](auto kern, int m, int grid) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(768), 131072, 0, d, n, 3);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%-36s grid %4d: %.1f cycles per trip of 64 instructions per wave, 12 waves -> %.2f cycles per LDS instruction per CU\n", names[m], grid, (double)h / n,
               (double)h / n / ((m == 4 ? 16 : 64) * 12));
    };
    for (int grid : {1, 256}) { run(k<0>, 0, grid); run(k<1>, 1, grid); run(k<2>, 2, grid); run(k<3>, 3, grid); run(k<4>, 4, grid); }
    return 0;
}
