// How much LDS can a small workgroup use and still be co-resident with the two-frame LDPC workgroup (136 208 bytes of LDS, 768 threads,
// 128 VGPRs, one per CU)?  A "hog" kernel of that shape spins for ~40 ms on every CU; probe kernels (64 workgroups x 128 threads) with a
// growing LDS footprint are launched on a second stream 5 ms later and report when their first wave started relative to the hog.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_fit tools/ubench/lds_fit.hip && /tmp/lds_fit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
#include <cstdlib>

__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(4))) void hog(unsigned long long* t0, long long ticks) {
    extern __shared__ char lds[];
    volatile char* p = lds;
    p[threadIdx.x] = 1;
    const unsigned long long s = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *t0 = s;
    while ((long long)(wall_clock64() - s) < ticks) { }
}
__global__ __launch_bounds__(256) void fragmenter(long long ticks) {      // small workgroups that are resident when the hog's workgroups are placed
    extern __shared__ char lds[];
    volatile char* p = lds;
    p[threadIdx.x] = 1;
    const unsigned long long s = wall_clock64();
    while ((long long)(wall_clock64() - s) < ticks) { }
}
__global__ __launch_bounds__(128) void probe(unsigned long long* t) {
    extern __shared__ char lds[];
    volatile char* p = lds;
    p[threadIdx.x] = 1;
    if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64();
}
int main(int argc, char** argv) {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned long long *t0, *t;
    hipMalloc(&t0, 8); hipMalloc(&t, 8 * 64);
    hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long long ticks = 40 * 100000;   // wall_clock64: 100 MHz
    const int hog_lds = 136208;
    hipStream_t c;
    hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    const int frag = argc > 1 ? atoi(argv[1]) : 0;     // LDS bytes of the fragmenter's workgroups (0 = none)
    for (int kb2 = 16; kb2 <= 56; kb2 += 4) {     // probe LDS in 512-byte units
        const int lds = kb2 * 512;
        if (frag) { hipLaunchKernelGGL(fragmenter, dim3(512), dim3(256), frag, c, 2 * 100000LL); usleep(500); }
        hipLaunchKernelGGL(hog, dim3(256), dim3(768), hog_lds, a, t0, ticks);
        usleep(5000);
        hipLaunchKernelGGL(probe, dim3(64), dim3(128), lds, b, t);
        hipDeviceSynchronize();
        unsigned long long h0; std::vector<unsigned long long> v(64);
        hipMemcpy(&h0, t0, 8, hipMemcpyDeviceToHost); hipMemcpy(v.data(), t, 8 * 64, hipMemcpyDeviceToHost);
        double mx = 0, mn = 1e9;
        for (auto x : v) { double ms = (double)(long long)(x - h0) / 1e5; mx = ms > mx ? ms : mx; mn = ms < mn ? ms : mn; }
        printf("probe LDS %6d B: first workgroup starts %.2f ms, last %.2f ms after the hog (hog runs 40 ms) -> %s\n", lds, mn, mx, mx < 35 ? "co-resident" : "WAITS");
    }
    return 0;
}
