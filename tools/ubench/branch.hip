// Development aid: what a TAKEN scalar branch costs one wave (the serial per-symbol loops jump over their rare paths several times per
// symbol).   hipcc -O3 --offload-arch=gfx950 -o /tmp/br tools/ubench/branch.hip && /tmp/br
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NB>
__global__ void k(float* out, long long* t, int n, int never) {
    float a = threadIdx.x * 1e-9f;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            a = __builtin_fmaf(a, 1.0000001f, 1e-7f);
            if (b < NB) {
                // a forward branch over code that never runs (uniform condition in an SGPR)
                asm volatile("s_cmp_eq_u32 %1, 0\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n1:" : "+v"(a) : "s"(never) : "scc");
            }
        }
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) t[0] = c1 - c0;
    out[threadIdx.x] = a;
}
int main() {
    float* o; long long* t; long long h0 = 0, h;
    hipMalloc(&o, 1024); hipMalloc(&t, 16);
    const int n = 100000;
#define RUN(NB) hipLaunchKernelGGL(k<NB>, dim3(1), dim3(64), 0, 0, o, t, n, 0); hipDeviceSynchronize(); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); \
    if (NB == 0) h0 = h; printf("%d taken branches per iteration: %.1f cycles per iteration (+%.1f per branch incl. its s_cmp)\n", NB, (double)h / n, NB ? (double)(h - h0) / n / NB : 0.0);
    RUN(0) RUN(1) RUN(2) RUN(4)
    return 0;
}
