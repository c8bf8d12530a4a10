// Development aid: VALU THROUGHPUT of a gfx950 SIMD by instruction type, measured by wall clock over the whole chip (hipEvent) -- every SIMD
// holds W waves, each with four independent chains of the instruction under test.  Prints wave-instructions per SIMD per shader cycle, the
// shader clock taken from s_memtime / s_memrealtime inside the same launch.   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_tput valu_tput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int n, int seed) {
    int a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5 ^ threadIdx.x, d = seed * 7 + 1, e = seed ^ 9, f = seed | 0x01010101;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#define BODY(INS) asm volatile(REP16(INS) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f))
        if (MODE == 0) BODY("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %5\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %5\n\t");
        if (MODE == 1) BODY("v_pk_add_i16 %0, %0, %4\n\tv_pk_max_i16 %1, %1, %5\n\tv_pk_min_i16 %2, %2, %4\n\tv_pk_sub_i16 %3, %3, %5 clamp\n\t");
        if (MODE == 2) BODY("v_perm_b32 %0, %0, %4, %5\n\tv_perm_b32 %1, %1, %5, %4\n\tv_perm_b32 %2, %2, %4, %5\n\tv_perm_b32 %3, %3, %5, %4\n\t");
        if (MODE == 3) BODY("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\tv_add_u32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\tv_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\tv_add_u32_sdwa %3, %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t");
        if (MODE == 4) BODY("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %5, %4\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %5, %4\n\t");
        if (MODE == 5) BODY("v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_min_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_min_i32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_min_i32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t");
        if (MODE == 6) BODY("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %5\n\t");
        if (MODE == 7) BODY("v_min3_i32 %0, %0, %4, %5\n\tv_med3_i32 %1, %1, %5, %4\n\tv_alignbit_b32 %2, %2, %4, 16\n\tv_bfe_i32 %3, %3, 8, 8\n\t");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a + b + c + d == 0x12345) out[100000] = a;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 1 << 21);
    const int n = 2000;
    const char* names[] = {"v_xor_b32 (VOP2)", "v_pk_* i16 (VOP3P)", "v_perm_b32 (VOP3)", "v_add_u32_sdwa", "v_fma_f32", "v_min_i32_dpp", "v_add_u32 (VOP2)", "min3/med3/alignbit/bfe"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int m, int wgs_per_cu) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, n, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;      // s_memrealtime ticks at 100 MHz
        const double winstr_per_simd = (double)n * 64 * wgs_per_cu;       // one wave of every workgroup per SIMD (256 threads = 4 waves, one per SIMD)
        printf("%-26s %d waves/SIMD: %.3f ms, in-kernel clock %.2f GHz, %.2f cycles per instruction per wave, %.2f shader cycles per wave-instruction per SIMD\n", names[m], wgs_per_cu, ms, ghz,
               (double)h[0] / n / 64, ms * 1e-3 * ghz * 1e9 / winstr_per_simd);
    };
    for (int w : {1, 2, 4, 6, 8}) { run(k<0>, 0, w); }
    for (int w : {1, 2, 4, 6, 8}) { run(k<1>, 1, w); }
    for (int w : {2, 4, 8}) { run(k<2>, 2, w); run(k<3>, 3, w); run(k<4>, 4, w); run(k<5>, 5, w); run(k<6>, 6, w); run(k<7>, 7, w); }
    return 0;
}
