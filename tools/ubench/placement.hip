// Where do the waves of small front-end workgroups land beside the decoder?  A "hog" of the half-row decoder's shape (768 threads, 64 VGPRs,
// 53 LDS granules, two workgroups per CU) spins on every CU; probe workgroups of NW waves and ~124 VGPRs (the timing recovery's shape: resolver +
// producer, 9 LDS granules) are launched on a second stream and report HW_ID of every wave: (XCC, SE, CU, SIMD).  Printed: how many probe waves each
// SIMD of a CU got -- a decoder workgroup runs at the pace of its most loaded SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/placement tools/ubench/placement.hip && /tmp/placement [waves per probe workgroup] [probe workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <unistd.h>
#include <cstdlib>

__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(8))) void hog(long long ticks) {
    extern __shared__ char lds[];
    volatile char* p = lds;
    p[threadIdx.x] = 1;
    const unsigned long long s = wall_clock64();
    while ((long long)(wall_clock64() - s) < ticks) { }
}
template <int NT>
__global__ __launch_bounds__(NT) void probe(unsigned* out, long long ticks) {
    extern __shared__ char lds[];
    volatile char* p = lds;
    p[threadIdx.x] = 1;
    // (124 registers: a value per register kept alive across the spin)
    float v[100];
#pragma unroll
    for (int i = 0; i < 100; ++i) v[i] = (float)(threadIdx.x + i);
    const unsigned long long s = wall_clock64();
    while ((long long)(wall_clock64() - s) < ticks) {
#pragma unroll
        for (int i = 0; i < 100; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i]));
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 100; ++i) acc += v[i];
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID: wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
        out[2 * (blockIdx.x * (NT / 64) + (threadIdx.x >> 6))] = hw;
        out[2 * (blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) + 1] = (xcc & 15u) | (acc == 12345.f ? 16u : 0u);
    }
}
int main(int argc, char** argv) {
    const int nw = argc > 1 ? atoi(argv[1]) : 2, nwg = argc > 2 ? atoi(argv[2]) : 512;
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned* out;
    hipMalloc(&out, 8 * nwg * nw);
    hipMemset(out, 0xff, 8 * nwg * nw);
    hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long long ticks = 30 * 100000;   // wall_clock64: 100 MHz
    for (int with_hog = 1; with_hog >= 0; --with_hog) {
        if (with_hog) { hipLaunchKernelGGL(hog, dim3(512), dim3(768), 67744, a, ticks); usleep(3000); }
        const int lds = 9 * 1280 - 64;
        if (nw == 1) hipLaunchKernelGGL(probe<64>, dim3(nwg), dim3(64), lds, b, out, 5 * 100000LL);
        else if (nw == 2) hipLaunchKernelGGL(probe<128>, dim3(nwg), dim3(128), lds, b, out, 5 * 100000LL);
        else hipLaunchKernelGGL(probe<256>, dim3(nwg), dim3(256), lds, b, out, 5 * 100000LL);
        hipDeviceSynchronize();
        std::vector<unsigned> v(2 * nwg * nw);
        hipMemcpy(v.data(), out, 8 * nwg * nw, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> per_cu;      // (xcc, se, sh, cu) -> waves per SIMD
        std::map<int, int> wave_simd[4];                  // wave index in its workgroup -> histogram of SIMDs
        for (int i = 0; i < nwg * nw; ++i) {
            const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 15u;
            const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            auto& c = per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu];
            c.resize(4, 0);
            c[simd]++;
            wave_simd[(i % nw) & 3][(int)simd]++;
        }
        std::map<std::vector<int>, int> patterns;
        int mx_sum = 0, tot = 0;
        for (auto& kv : per_cu) { patterns[kv.second]++; int m = 0; for (int x : kv.second) { m = x > m ? x : m; tot += x; } mx_sum += m; }
        printf("%s: %d probe workgroups x %d waves on %zu CUs; sum over CUs of the most loaded SIMD's waves = %d (even spread would be %.0f)\n", with_hog ? "beside the hog" : "alone", nwg, nw, per_cu.size(),
               mx_sum, tot / 4.0);
        for (int w = 0; w < nw && w < 4; ++w) { printf("  wave %d of a workgroup -> SIMD histogram:", w); for (auto& kv : wave_simd[w]) printf(" %d:%d", kv.first, kv.second); printf("\n"); }
        int shown = 0;
        for (auto& kv : patterns) { if (shown++ < 12) printf("  per-CU pattern [%d %d %d %d] x %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second); }
    }
    return 0;
}
