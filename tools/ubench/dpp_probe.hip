// Development aid: what the DPP row_bcast controls do on gfx950 (lane values printed after each operation).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int v = 100 + threadIdx.x, a, b, c, d;
    asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "=&v"(a) : "v"(v));
    asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "=&v"(b) : "v"(v));
    asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "=&v"(c) : "v"(v));
    asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 4\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "=&v"(d) : "v"(v));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"min bcast15 0xa", "min bcast31 0xc", "both (nop 1)", "both (nop 4)"};
    for (int q = 0; q < 4; ++q) { printf("%s:", names[q]); for (int l = 0; l < 64; l += 5) printf(" %d:%d", l, h[64 * q + l]); printf(" 63:%d\n", h[64 * q + 63]); }
    return 0;
}
