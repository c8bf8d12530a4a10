// Development aid: the Viterbi step's wave-minimum chain (csrc/dvbs_kernels.hip, VIT_STEP) with its exact fillers, against a shuffle reduction, on random data.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int variant) {
    int bad = 0, badrow[4] = {0, 0, 0, 0};
    unsigned h = 12345u + threadIdx.x * 2654435761u;
    int wlo = 0, whi = 0, cur = threadIdx.x * 257;
    for (int it = 0; it < 4000; ++it) {
        h = h * 1664525u + 1013904223u;
        int y = (h >> 13) & 255;
        int ref = y;
        for (int o = 32; o > 0; o >>= 1) ref = min(ref, __shfl_xor(ref, o));
        int mu;
        if (variant == 0)
            asm volatile(
                "s_mov_b32 m0, 3\n\t"
                "v_mov_b32 v103, %[y]\n\t"
                "s_and_b64 s[22:23], vcc, exec\n\t"
                "s_andn2_b64 s[20:21], s[20:21], exec\n\t"
                "v_min_u32_dpp v104, v103, v103 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_or_b64 s[22:23], s[22:23], s[20:21]\n\t"
                "v_writelane_b32 %[wlo], s22, m0\n\t"
                "v_min_u32_dpp v104, v104, v104 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_writelane_b32 %[whi], s23, m0\n\t"
                "s_add_u32 m0, m0, 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_readlane_b32 s24, %[cur], m0\n\t"
                "s_and_b32 s25, s24, 0xff\n\t"
                "v_min_u32_dpp v104, v104, v104 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                "s_lshr_b32 s26, s24, 8\n\t"
                "v_xor_b32 v106, s25, %[cur]\n\t"
                "v_min_u32_dpp v104, v104, v104 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "v_xor_b32 v107, s26, %[cur]\n\t"
                "v_add3_u32 v105, v106, v107, 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                "v_readlane_b32 %[mu], v104, 63\n\t"
                : [mu] "=s"(mu), [wlo] "+v"(wlo), [whi] "+v"(whi) : [y] "v"(y), [cur] "v"(cur)
                : "v103", "v104", "v105", "v106", "v107", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "m0", "vcc", "scc");
        else
            asm volatile(
                "v_mov_b32 v103, %[y]\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp v104, v103, v103 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_u32_dpp v104, v104, v104 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_readlane_b32 %[mu], v104, 63\n\t"
                : [mu] "=s"(mu) : [y] "v"(y) : "v103", "v104");
        if (mu != ref) {
            ++bad;
            int r = y == ref ? (threadIdx.x >> 4) : 99;      // which row holds the minimum (first such lane)
            unsigned long long m = __ballot(y == ref);
            int first = __ffsll((long long)m) - 1;
            ++badrow[first >> 4];
        }
    }
    if (threadIdx.x == 0) { out[0] = bad; for (int i = 0; i < 4; ++i) out[1 + i] = badrow[i]; }
}
int main() {
    int* d; hipMalloc(&d, 64);
    for (int v = 0; v < 2; ++v) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
        int h[5]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("variant %d (%s): %d of 4000 minima wrong; minimum was in row 0/1/2/3: %d %d %d %d\n", v, v ? "s_nop fillers" : "the step's fillers", h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
