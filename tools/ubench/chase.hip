// Development aid: latency of a DEPENDENT global load of one wave (the PLL's phase-error table lookup sits in the per-symbol chain):
// pointer chase over a table of 16 KB (stays in the CU's vector L1), 256 KB (the table's size: L2) and 64 MB (Infinity Cache / HBM),
// and the same chase through LDS.   hipcc -O3 --offload-arch=gfx950 -o /tmp/chase tools/ubench/chase.hip && /tmp/chase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void chase(const int* __restrict__ tab, int n, int* out, long long* t) {
    int i = threadIdx.x == 0 ? 0 : 1;
    const long long c0 = clock64();
    for (int k = 0; k < n; ++k) i = tab[i];
    const long long c1 = clock64();
    if (threadIdx.x == 0) t[0] = c1 - c0;
    out[threadIdx.x] = i;
}
__global__ void chase_lds(const int* __restrict__ tab, int n, int* out, long long* t) {
    __shared__ int l[4096];
    for (int k = threadIdx.x; k < 4096; k += 64) l[k] = tab[k];
    __syncthreads();
    int i = threadIdx.x == 0 ? 0 : 1;
    const long long c0 = clock64();
    for (int k = 0; k < n; ++k) i = l[i];
    const long long c1 = clock64();
    if (threadIdx.x == 0) t[0] = c1 - c0;
    out[threadIdx.x] = i;
}
int main() {
    int* o; long long* t; long long h;
    hipMalloc(&o, 1024); hipMalloc(&t, 16);
    const int n = 20000;
    for (size_t bytes : {16384ul, 262144ul, 67108864ul}) {
        const size_t m = bytes / 4;
        std::vector<int> tab(m);
        // a random cyclic permutation with 64-byte strides at least (no two consecutive hops in one cache line)
        std::vector<int> order(m / 16);
        for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
        srand(1);
        for (size_t k = order.size() - 1; k > 0; --k) { size_t j = rand() % (k + 1); std::swap(order[k], order[j]); }
        for (size_t k = 0; k < m; ++k) tab[k] = 0;
        for (size_t k = 0; k < order.size(); ++k) tab[(size_t)order[k] * 16] = order[(k + 1) % order.size()] * 16;
        tab[1] = 1;
        int* d; hipMalloc(&d, bytes); hipMemcpy(d, tab.data(), bytes, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, n, o, t); hipDeviceSynchronize(); }
        hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("global, table %8zu KB: %.0f cycles per dependent load\n", bytes / 1024, (double)h / n);
        if (bytes == 16384) {
            hipLaunchKernelGGL(chase_lds, dim3(1), dim3(64), 0, 0, d, n, o, t); hipDeviceSynchronize();
            hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            printf("LDS,    table       16 KB: %.0f cycles per dependent load\n", (double)h / n);
        }
        hipFree(d);
    }
    return 0;
}
