// Development aid: cost of a workgroup barrier / LDS round trip on gfx950 (768-thread workgroup like the LDPC kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int MODE>
__global__ __launch_bounds__(768) void k(unsigned long long* out, int n, int worker) {
    __shared__ volatile int buf[1024];
    const int w = threadIdx.x >> 6;
    int x = threadIdx.x;
    buf[threadIdx.x] = x;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) { lds_barrier(); }
        if (MODE == 1) {   // one wave works between barriers: LDS read -> ~40 VALU -> LDS write
            lds_barrier();
            if (w == worker) {
                int v = buf[(x + 1) & 1023];
#pragma unroll
                for (int q = 0; q < 40; ++q) v = v * 3 + q;
                buf[x & 1023] = v;
                x = v & 1023;
            }
        }
        if (MODE == 2) {   // the same work without barriers (single wave timing)
            if (w == worker) {
                int v = buf[(x + 1) & 1023];
#pragma unroll
                for (int q = 0; q < 40; ++q) v = v * 3 + q;
                buf[x & 1023] = v;
                x = v & 1023;
            }
        }
        if (MODE == 3) {   // worker with s_setprio(3)
            lds_barrier();
            if (w == worker) {
                __builtin_amdgcn_s_setprio(3);
                int v = buf[(x + 1) & 1023];
#pragma unroll
                for (int q = 0; q < 40; ++q) v = v * 3 + q;
                buf[x & 1023] = v;
                x = v & 1023;
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (x == 123456789) out[1] = x;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    const int n = 2000;
    auto run = [&](auto kern, const char* name, int worker, int grid) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(768), 0, 0, d, n, worker);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-34s worker wave %2d grid %3d: %.1f cycles per trip\n", name, worker, grid, (double)h[0] / n);
    };
    for (int grid : {1, 256}) {
        run(k<0>, "barrier only", 0, grid);
        run(k<1>, "barrier + read/40 VALU/write", 0, grid);
        run(k<1>, "barrier + read/40 VALU/write", 5, grid);
        run(k<3>, "same with setprio 3", 5, grid);
        run(k<2>, "read/40 VALU/write, no barrier", 0, grid);
    }
    return 0;
}
