// Development aid: shader clock of a LIGHT launch (one wave on one CU) against a chip-filling one: dependent v_fma chain timed with the
// shader cycle counter (clock64) and the constant 100 MHz counter (wall_clock64).   hipcc -O3 --offload-arch=gfx950 -o /tmp/clk tools/ubench/clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(float* out, long long* t, int n) {
    float a = threadIdx.x * 1e-9f, b = 1.0000001f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, 1e-7f);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float* o; long long* t; long long h[2];
    hipMalloc(&o, 4096 * 256 * 4); hipMalloc(&t, 16);
    for (int rep = 0; rep < 3; ++rep)
    for (int blocks : {1, 16, 4096}) {
        const int n = 20000;
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(blocks == 4096 ? 256 : 64), 0, 0, o, t, n);
        hipDeviceSynchronize();
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("blocks %5d: %.2f shader cycles per dependent fma, %.1f ns per fma -> shader clock %.0f MHz\n", blocks, (double)h[0] / (n * 64.0), h[1] * 10.0 / (n * 64.0),
               (double)h[0] / (h[1] * 10.0) * 1000.0);
    }
    return 0;
}
