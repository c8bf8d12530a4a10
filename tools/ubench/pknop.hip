// Development aid: what the one wait state the compiler puts behind every packed 16-bit (VOP3P) result costs on gfx950 (s_nop 0 between dependent
// v_pk_* instructions), against the same chain without it (inline asm is not hazard-checked) and against two interleaved chains; WAVES per CU as in valu_cu.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned long long* out, int n, int seed, int* sink) {
    int a = seed + threadIdx.x, b = seed * 3, c = seed * 5, d = seed * 7 + threadIdx.x;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) asm volatile(REP32("v_pk_add_i16 %0, %0, %1\n\tv_pk_max_i16 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 1) asm volatile(REP32("v_pk_add_i16 %0, %0, %1\n\ts_nop 0\n\tv_pk_max_i16 %0, %0, %2\n\ts_nop 0\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 2) asm volatile(REP32("v_pk_add_i16 %0, %0, %2\n\tv_pk_add_i16 %1, %1, %2\n\tv_pk_max_i16 %0, %0, %3\n\tv_pk_max_i16 %1, %1, %3\n\t") : "+v"(a), "+v"(d) : "v"(b), "v"(c));
        if (MODE == 3) asm volatile(REP32("v_pk_add_i16 %0, %0, %1\n\tv_xor_b32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 4) asm volatile(REP32("v_pk_add_i16 %0, %0, %1\n\ts_nop 0\n\tv_xor_b32 %0, %0, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * WAVES * 64 + threadIdx.x] = a + d;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    int* sink; hipMalloc(&sink, 256 * 1024 * 4);
    const int n = 200;
    const char* names[] = {"dependent pk, no nop", "dependent pk + s_nop 0", "two interleaved pk chains", "pk -> xor dependent, no nop", "pk -> nop -> xor"};
    const int cnt[] = {64, 64, 128, 64, 64};
    auto run = [&](auto kern, int m, int waves) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, d, n, 3, sink);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        int s0; hipMemcpy(&s0, sink, 4, hipMemcpyDeviceToHost);
        printf("%-30s %2d waves per CU: %.2f cycles per VALU instruction per wave   (lane 0 result %08x)\n", names[m], waves, (double)h / n / cnt[m], s0);
    };
    run(k<0, 4>, 0, 4); run(k<1, 4>, 1, 4); run(k<2, 4>, 2, 4); run(k<3, 4>, 3, 4); run(k<4, 4>, 4, 4);
    run(k<0, 12>, 0, 12); run(k<1, 12>, 1, 12); run(k<2, 12>, 2, 12);
    return 0;
}
