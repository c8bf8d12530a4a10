// Development aid: the rotating-layout Viterbi (csrc/dvbs_kernels.hip: cc_decode_wave) against a plain CPU restatement, step by step (decision words in the
// rotating layout, then the decoded bits).  build: hipcc -O2 --offload-arch=gfx950 -I../../include -I../../sdrpp-dvbs-demodulator_amd/csrc -o vit_dbg.bin vit_dbg.hip
#include "../../sdrpp-dvbs-demodulator_amd/csrc/dvbs_kernels.hip"
#include <cstdio>
#include <vector>
#include <random>
static int par(int x) { return __builtin_popcount(x) & 1; }
static int h_rotr6(int x, int r) { r %= 6; return ((x >> r) | (x << (6 - r))) & 63; }
__global__ void k(const uint8_t* src, int frame, unsigned long long* dec, uint8_t* dst, int* st) {
    int ss = st[0], biased = st[1];
    s2::cc_decode_wave(src, frame, ss, biased, dec, dst, threadIdx.x);
    if (threadIdx.x == 0) { st[0] = ss; st[1] = biased; }
}
int main(int argc, char** argv) {
    const int frame = argc > 1 ? atoi(argv[1]) : 30, veclen = frame + 6;
    std::mt19937 rng(5);
    std::vector<uint8_t> src(2 * veclen);
    for (auto& b : src) b = rng() & 255;
    uint8_t *d_src, *d_dst; unsigned long long* d_dec; int* d_st;
    hipMalloc(&d_src, src.size()); hipMalloc(&d_dst, frame); hipMalloc(&d_dec, 8 * veclen); hipMalloc(&d_st, 8);
    hipMemcpy(d_src, src.data(), src.size(), hipMemcpyHostToDevice);
    int st[2] = {0, 0}; hipMemcpy(d_st, st, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_src, frame, d_dec, d_dst, d_st);
    hipDeviceSynchronize();
    std::vector<unsigned long long> dec(veclen); std::vector<uint8_t> dst(frame);
    hipMemcpy(dec.data(), d_dec, 8 * veclen, hipMemcpyDeviceToHost); hipMemcpy(dst.data(), d_dst, frame, hipMemcpyDeviceToHost); hipMemcpy(st, d_st, 8, hipMemcpyDeviceToHost);
    // CPU: volk_k7_r2_generic_fixed.h BFLY + renormalize, decisions in state order
    int X[64], Y[64]; for (int i = 0; i < 64; ++i) X[i] = 31;
    std::vector<unsigned long long> ref(veclen);
    int bad = 0;
    for (int t = 0; t < veclen; ++t) {
        unsigned long long d = 0;
        const int y0 = src[2 * t], y1 = src[2 * t + 1];
        for (int i = 0; i < 32; ++i) {
            const int b0 = par((2 * i) & 79) ? 255 : 0, b1 = par((2 * i) & 109) ? 255 : 0;
            const int metric = ((1 + (b0 ^ y0) + (b1 ^ y1)) >> 1) >> 2;
            const int m0 = (X[i] + metric) & 255, m1 = (X[i + 32] + 63 - metric) & 255, m2 = (X[i] + 63 - metric) & 255, m3 = (X[i + 32] + metric) & 255;
            const int d0 = m0 - m1 >= 0, d1 = m2 - m3 >= 0;
            Y[2 * i] = d0 ? m1 : m0; Y[2 * i + 1] = d1 ? m3 : m2;
            d |= ((unsigned long long)d0 << (2 * i)) | ((unsigned long long)d1 << (2 * i + 1));
        }
        int mn = 255; for (int i = 0; i < 64; ++i) mn = Y[i] < mn ? Y[i] : mn;
        for (int i = 0; i < 64; ++i) X[i] = Y[i] - mn;
        // expected word in the rotating layout: bit rotr6(s, t + 1) = decision of state s
        unsigned long long e = 0;
        for (int s = 0; s < 64; ++s) e |= ((d >> s) & 1ull) << h_rotr6(s, (t + 1) % 6);
        ref[t] = d;
        if (e != dec[t] && bad < 6) { printf("step %d (r=%d): expected %016llx got %016llx  xor %016llx\n", t, t % 6, e, dec[t], e ^ dec[t]); ++bad; }
    }
    // chainback
    int end = 0; for (int i = 1; i < 64; ++i) if (X[i] < X[end]) end = i;
    unsigned es = end << 2; int retval = 0, nbad = 0;
    std::vector<uint8_t> out(frame);
    for (int j = frame - 1; j >= 0; --j) { const int kb = (ref[6 + j] >> (es >> 2)) & 1; es = (es >> 1) | (kb << 7); out[j] = kb; if (j == frame - 6) retval = es; }
    for (int j = 0; j < frame; ++j) nbad += out[j] != dst[j];
    printf("frame %d: decision mismatches %d%s, bit mismatches %d, ss %d (expected %d)\n", frame, bad, bad >= 6 ? "+" : "", nbad, st[0], retval >> 2);
    return 0;
}
