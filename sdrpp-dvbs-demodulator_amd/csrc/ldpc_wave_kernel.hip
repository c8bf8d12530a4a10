// Layered offset-min-sum LDPC decoder, WAVE-PER-FRAME form (short frames; plan and rationale: ldpc_wave_plan.h).
//
// Replaces the same reference code as ldpc_kernel.hip (bbframe_ldpc.cpp:123-139, xdsopl-ldpc-pabr/layered_decoder.hh:23-133,
// algorithms.hh:206-277), bit-exact.  One wave = one frame, persistent over the batch:
//   * the frame's N int8 posteriors live in the wave's LDS piece (16.2 KB: eight frames per CU and room left for the front end);
//   * a step = 8 rows x 8 lanes; lane (g, l) owns link slots 8 kk + l of row g's check: LW byte reads, the row's two smallest
//     magnitudes and its sign by three DPP butterfly steps inside the 8-lane group, LW byte writes;
//   * check->bit messages: one element of LW bytes per lane and row in a per-wave global workspace (coalesced 8-element records),
//     fetched a chunk of steps ahead together with the step list;
//   * rows that share a bit sit in different steps, in the reference's order; the LDS executes a wave's instructions in order, so
//     nothing else is needed -- no barrier, no conflict bookkeeping.
// Roofline: as ldpc_kernel.hip (nominal algorithmic bytes); the kernel is bound by the instructions of its waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_wave_plan.h"
#include "kernels.h"
#include "ldpc_dev_common.h"

#ifndef LDPC_WAVE_OTHER_SUM
#define LDPC_WAVE_OTHER_SUM 1   // the other-minimum selection as a sum minus a minimum (A/B switch; see ldpc_kernel.hip)
#endif
namespace s2 {

// The code tables are separate `const T* __restrict__` kernel parameters (not struct members): hipcc then proves them invariant and
// fetches the wave-uniform entries with scalar loads.
struct LdpcWaveArgs {
    const int8_t* llr;          // [nframes][N]
    uint8_t* hard;              // [nframes][hard_stride]
    int8_t* post;               // optional [nframes][N]
    int32_t* trials;            // [nframes]
    uint8_t* msg_ws;            // [gridDim.x][R * 8 * element bytes]
    uint32_t* sgn_ws;           // [gridDim.x][SGN_WS_DWORDS]
    unsigned int* work_ctr;
    int nframes, N, K, R, q, nsteps, absent_base, synd_base, max_deg, max_trials, force, hard_stride;
};

#define WQUAD(x_, ctrl) __builtin_amdgcn_update_dpp(0, (x_), (ctrl), 0xf, 0xf, true)
// one v_med3_i32 (left to itself the compiler narrows these clamps to 16-bit saturating arithmetic: two shifts in, one out)
__device__ __forceinline__ int wmed3(int x, int lo, int hi) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi)); return r; }

template <int LW> struct WaveElem { typedef uint32_t type; };
template <> struct WaveElem<1> { typedef uint8_t type; };
template <> struct WaveElem<2> { typedef uint16_t type; };

// LDPCDecoder::bad on the packed sign vectors (ldpc_dev_common.h), one wave, few registers: the windows of four links at a time
__device__ __forceinline__ bool wave_syndromes_bad(int q, int max_deg, const uint32_t* __restrict__ tab, const uint32_t* __restrict__ S, int lane) {
    const int ntask = q * 6;
    bool bad = false;
    for (int t = lane; t < ntask; t += 64) {
        unsigned long long acc = 0;
        for (int k0 = 0; k0 < max_deg + 2; k0 += 4) {
            uint32_t e[4], d0[4], d1[4], d2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) e[i] = (k0 + i < max_deg + 2) ? tab[(k0 + i) * ntask + t] : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint32_t* __restrict__ p = S + (e[i] & 0xffffu); d0[i] = p[0]; d1[i] = p[1]; d2[i] = p[2]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t sh = (e[i] >> 16) & 31u;
                uint32_t lo = __builtin_amdgcn_alignbit(d1[i], d0[i], sh), hi = __builtin_amdgcn_alignbit(d2[i], d1[i], sh);
                lo &= ~((e[i] >> 30) & 1u);
                const uint32_t m = (uint32_t)((int)e[i] >> 31);
                acc ^= ((unsigned long long)(hi & m) << 32) | (lo & m);
            }
        }
        if (t - 6 * (t / 6) == 5) acc &= (1ull << 40) - 1;
        bad |= acc != 0;
    }
    return bad;
}

// LW link slots per lane; slots kk >= ABS_FROM may be absent in some layer (tail of an irregular code, or 8 * LW > the row degree)
template <int LW, int ABS_FROM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void ldpc_wave_kernel(const uint32_t* __restrict__ lanec, const uint16_t* __restrict__ steps,
                                                                                        const uint32_t* __restrict__ layer_end, const uint32_t* __restrict__ ents, LdpcWaveArgs A) {
    extern __shared__ __attribute__((aligned(16))) int8_t wpost[];
    typedef typename WaveElem<LW>::type elem_t;
    constexpr int U = LDPC_WAVE_CHUNK;
    __shared__ int s_next;
    const int lane = threadIdx.x, l8 = lane & 7, g = lane >> 3;
    const int N = A.N, K = A.K, R = A.R, q = A.q;
    int8_t* __restrict__ post = wpost;
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) int8_t*)post;
    elem_t* __restrict__ msg = reinterpret_cast<elem_t*>(A.msg_ws) + (size_t)blockIdx.x * (size_t)R * 8 + l8;
    uint32_t* __restrict__ sgn = A.sgn_ws + (size_t)blockIdx.x * SGN_WS_DWORDS;
    const uint32_t dummy = lbase + (uint32_t)(N + l8);        // a byte behind the frame (the LDS piece is padded): where absent link slots read and write
    const uint16_t* __restrict__ mysteps = steps + g;
    typedef __attribute__((address_space(3))) int8_t lds_i8;

    int f = blockIdx.x;
    while (f < A.nframes) {
        {
            const int8_t* __restrict__ src = A.llr + (size_t)f * N;
            for (int i = lane; i < K / 8; i += 64) reinterpret_cast<uint2*>(post)[i] = reinterpret_cast<const uint2*>(src)[i];
            for (int c = lane; c < R; c += 64) {          // parity LLRs: pty[360*i + jj] = llr[K + q*jj + i]   (layered_decoder.hh:124-126)
                const int jj = c / q, i = c - jj * q;
                post[K + 360 * i + jj] = src[K + c];
            }
        }
        int it = 0, ret = 0;
        bool done = false;
        while (true) {
            const bool check = !A.force || it == A.max_trials;
            if (check) {
                // LDPCDecoder::bad on bit vectors (ldpc_dev_common.h); one wave: the sign bytes go through the L2 (write, wait, read)
                const uint32_t z = sign_pack(post, N, reinterpret_cast<uint8_t*>(sgn), lane, 64);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const bool bad = z != 0 || wave_syndromes_bad(q, A.max_deg, ents + A.synd_base, sgn, lane);
                const bool any = __ballot(bad) != 0;
                if (A.force) { ret = any ? -1 : A.max_trials; done = true; }
                else if (!any) { ret = it; done = true; }
                else if (it == A.max_trials) { ret = -1; done = true; }
            }
            if (done) break;
            // ---- one layered sweep (LDPCDecoder::update): layer by layer, a layer's steps in chunks of U; step list and message elements
            // travel a chunk ahead (the list is padded with empty chunks, so the fetches need no bound test)
            const bool first = (it == 0);
            uint32_t jn[U], jm[U], rn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { jn[u] = mysteps[u * 8]; jm[u] = mysteps[(U + u) * 8]; }
            // (message elements are fetched unconditionally -- a load behind a branch is waited for on the spot --: empty slots read row R-1,
            // the first sweep reads the workspace as it is and ignores it)
#pragma unroll
            for (int u = 0; u < U; ++u) { const uint32_t e = (uint32_t)msg[(size_t)min(jn[u], (uint32_t)(R - 1)) * 8]; rn[u] = first ? 0u : e; }
            int chunk = 0;
            for (int layer = 0; layer < q; ++layer) {
                uint32_t thr[LW], cA[LW], cB[LW], absm = 0;
#pragma unroll
                for (int kk = 0; kk < LW; ++kk) {
                    const uint32_t c = lanec[((size_t)layer * 8 + l8) * LW + kk];
                    thr[kk] = c & 0xffffu; cA[kk] = lbase + (c >> 16); cB[kk] = cA[kk] - 360u;
                }
                if (ABS_FROM < LW) absm = lanec[A.absent_base + layer * 8 + l8];
                uint32_t vkeep[(LW + 1) / 2], gnone[(LW + 1) / 2];       // packed form: per pair of slots, what an absent slot is masked with
#pragma unroll
                for (int p = 0; p < (LW + 1) / 2; ++p) {
                    vkeep[p] = 0xffffffffu; gnone[p] = 0u;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (2 * p + h < LW && ((absm >> (2 * p + h)) & 1u)) { vkeep[p] &= ~(0xffffu << (16 * h)); gnone[p] |= (uint32_t)Q8_NONE << (16 * h); }
                }
                const int cend = (int)layer_end[layer];
                const uint32_t row0 = (uint32_t)(360 * layer);
                const bool l0fix = (layer == 0) && l8 == 1;              // row 0 of layer 0 has no previous parity bit (slot 1)
                for (; chunk < cend; ++chunk) {
                    uint32_t jq[U], rq[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) jq[u] = mysteps[((chunk + 2) * U + u) * 8];
#pragma unroll
                    for (int u = 0; u < U; ++u) { const uint32_t e = (uint32_t)msg[(size_t)min(jm[u], (uint32_t)(R - 1)) * 8]; rq[u] = first ? 0u : e; }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool valid = jn[u] != LDPC_WAVE_NOROW;
                        const uint32_t rowid = valid ? jn[u] : row0;
                        const uint32_t j = rowid - row0;
                        const uint32_t rec = rn[u];
                        uint32_t a[LW];
                        bool ab[LW];
#pragma unroll
                        for (int kk = 0; kk < LW; ++kk) {
                            a[kk] = j + (j >= thr[kk] ? cB[kk] : cA[kk]);
                            ab[kk] = kk >= ABS_FROM && ((absm >> kk) & 1u);
                            if (kk == 0) ab[kk] = ab[kk] || (l0fix && j == 0);
                            if (kk >= ABS_FROM || kk == 0) a[kk] = ab[kk] ? dummy : a[kk];
                        }
                        int x[LW];
#pragma unroll
                        for (int kk = 0; kk < LW; ++kk) x[kk] = *(lds_i8*)(uintptr_t)a[kk];
                        int min0 = 255, min1 = 255, sx = 0;
#define WJOIN(ctrl) do { const int o0 = WQUAD(min0, ctrl), o1 = WQUAD(min1, ctrl); min1 = min(max(min0, o0), min(min1, o1)); min0 = min(min0, o0); sx ^= WQUAD(sx, ctrl); } while (0)
                        if constexpr ((LW & 1) == 0) {
                            // ---- two link slots per packed int16 register ("Q8": the int8 value in the high byte of each half; the 16-bit saturating
                            // add / subtract then IS the int8 saturation of the reference's SIMD lanes -- ldpc_dev_common.h, as in the lane-per-row kernel)
                            constexpr int NPW = LW / 2;
                            s16x2 V[NPW], G[NPW], MIN0 = splat2(Q8_NONE), MIN1 = splat2(Q8_NONE);
                            uint32_t SX = 0;
#pragma unroll
                            for (int p = 0; p < NPW; ++p) {
                                const s16x2 X = from_bits2(__builtin_amdgcn_perm((uint32_t)x[2 * p + 1], (uint32_t)x[2 * p], 0x040c000cu));
                                s16x2 v = sat_sub2(X, rec_pair_dw(rec, 2 * p));
                                const s16x2 av = pmax2(v, sat_sub2(splat2(0), v));
                                s16x2 g = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, av), (u16x2){256, 256}));   // mag_of (high byte)
                                // absent slots (tail of the row, or the missing previous parity bit of row 0 in layer 0): value 0, magnitude "none" (>= 127)
                                if (2 * p + 1 >= ABS_FROM) { v = from_bits2(bits2(v) & vkeep[p]); g = from_bits2(bits2(g) | gnone[p]); }
                                if (p == 0 && layer == 0) { const bool z = l0fix && j == 0; v = from_bits2(bits2(v) & (z ? 0xffff0000u : 0xffffffffu)); g = from_bits2(bits2(g) | (z ? (uint32_t)Q8_NONE : 0u)); }
                                V[p] = v; G[p] = g;
                                if (p == 0) MIN0 = g;
                                else if (p == 1) { MIN1 = pmax2(MIN0, g); MIN0 = pmin2(MIN0, g); }
                                else { MIN1 = pmin2(MIN1, pmax2(MIN0, g)); MIN0 = pmin2(MIN0, g); }
                                SX ^= bits2(v);
                            }
                            const int a0 = MIN0[0] >> 8, b0 = MIN0[1] >> 8, a1 = MIN1[0] >> 8, b1 = MIN1[1] >> 8;
                            min0 = min(a0, b0);
                            min1 = min(max(a0, b0), min(a1, b1));
                            sx = (int)(short)(SX ^ (SX >> 16));
                            WJOIN(0xB1);             // quad_perm [1,0,3,2]
                            WJOIN(0x4E);             // quad_perm [2,3,0,1]
                            WJOIN(0x141);            // row_half_mirror: the other quad of the 8-lane group
                            if (valid) {
                                const int min0c = min(min0, 32), min1c = min(min1, 32);
#if LDPC_WAVE_OTHER_SUM
                                const s16x2 MIN1CB = q8(min1c), SUMCB = q8(min0c + min1c);
#else
                                const s16x2 MIN0B = q8(min0), MIN1CB = q8(min1c), NDB = q8(min0c - min1c);
#endif
                                const uint32_t SXB = ((uint32_t)sx & 0xffffu) * 0x10001u;
                                s16x2 NM[2] = {splat2(0), splat2(0)};
#pragma unroll
                                for (int p = 0; p < NPW; ++p) {
                                    // other = (mag == min0) ? min1 : min0  ==  min1 + (mag != min0) * (min0 - min1), limited to 32 once per row
#if LDPC_WAVE_OTHER_SUM
                                    const s16x2 other = SUMCB - pmin2(G[p], MIN1CB);      // (the same value: a magnitude is the row minimum or at least the second one; ldpc_kernel.hip)
#else
                                    const s16x2 ne = pmin2(G[p] - MIN0B, splat2(1));
                                    const s16x2 other = ne * NDB + MIN1CB;
#endif
                                    const s16x2 neg = from_bits2(SXB ^ bits2(V[p])) >> 15;
                                    const s16x2 nm = pmin2(from_bits2(bits2(other) ^ bits2(neg)) - neg, q8(31));
                                    const uint32_t pn = bits2(sat_add2(V[p], nm)) >> 8;          // new posteriors at bits 7:0 and 23:16
                                    *(lds_i8*)(uintptr_t)a[2 * p] = (int8_t)pn;
                                    *(lds_i8*)(uintptr_t)a[2 * p + 1] = (int8_t)(pn >> 16);
                                    NM[p] = nm;
                                }
                                const uint32_t ro = __builtin_amdgcn_perm(bits2(NM[1]), bits2(NM[0]), 0x07050301u);     // the four high bytes
                                msg[(size_t)rowid * 8] = (elem_t)ro;
                            }
                        } else {
                        int v[LW], mg[LW];
#pragma unroll
                        for (int kk = 0; kk < LW; ++kk) {
                            const int m = (int)__builtin_amdgcn_sbfe((int)rec, 8 * kk, 8);
                            int vv = wmed3(x[kk] - m, -128, 127);
                            int gg = mag_of(vv);
                            if (kk >= ABS_FROM || kk == 0) { vv = ab[kk] ? 0 : vv; gg = ab[kk] ? 127 : gg; }
                            v[kk] = vv; mg[kk] = gg;
                            if (kk == 0) min0 = gg;
                            else if (kk == 1) { min1 = max(min0, gg); min0 = min(min0, gg); }
                            else { min1 = min(min1, max(min0, gg)); min0 = min(min0, gg); }
                            sx ^= vv;
                        }
                        // the row's totals over its 8 lanes: two smallest magnitudes (with multiplicity) and the sign
                        WJOIN(0xB1);             // quad_perm [1,0,3,2]
                        WJOIN(0x4E);             // quad_perm [2,3,0,1]
                        WJOIN(0x141);            // row_half_mirror: the other quad of the 8-lane group
                        if (valid) {
                            uint32_t ro = 0;
#pragma unroll
                            for (int kk = 0; kk < LW; ++kk) {
                                const int other = (mg[kk] == min0) ? min1 : min0;
                                const int neg = (sx ^ v[kk]) >> 31;
                                const int nm = wmed3((other ^ neg) - neg, -32, 31);
                                *(lds_i8*)(uintptr_t)a[kk] = (int8_t)wmed3(v[kk] + nm, -128, 127);
                                ro |= ((uint32_t)nm & 0xffu) << (8 * kk);
                            }
                            msg[(size_t)rowid * 8] = (elem_t)ro;
                        }
                        }
#undef WJOIN
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) { jn[u] = jm[u]; jm[u] = jq[u]; rn[u] = rq[u]; }
                }
            }
            ++it;
        }

        // ---- outputs
        if (lane == 0) A.trials[f] = ret;
        uint8_t* __restrict__ hd = A.hard + (size_t)f * A.hard_stride;
        for (int base = 0; base < K; base += 64) {
            const int idx = base + lane;
            const int neg = (idx < K) ? (post[idx] < 0) : 0;
            unsigned long long b = __ballot(neg);
            b = __builtin_bswap64(__brevll(b));
            if (lane == 0) {
                const int nbytes = min(8, (K - base) / 8);
                // (K / 8 is odd for three of the codes this decoder serves, and a caller's stride need not be a multiple of 8: the 8-byte store only where it is aligned)
                if (nbytes == 8 && ((uintptr_t)(hd + base / 8) & 7u) == 0) *reinterpret_cast<uint2*>(hd + base / 8) = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
                else for (int n = 0; n < nbytes; ++n) hd[base / 8 + n] = (uint8_t)(b >> (8 * n));
            }
        }
        if (A.post) {
            int8_t* __restrict__ dst = A.post + (size_t)f * N;
            for (int i = lane; i < K / 8; i += 64) reinterpret_cast<uint2*>(dst)[i] = reinterpret_cast<const uint2*>(post)[i];
            for (int c = lane; c < R; c += 64) {
                const int jj = c / q, i = c - jj * q;
                dst[K + c] = post[K + 360 * i + jj];
            }
        }
        if (A.work_ctr) {
            if (lane == 0) s_next = (int)(gridDim.x + atomicAdd(A.work_ctr, 1u));
            f = __builtin_amdgcn_readfirstlane(*(volatile int*)&s_next);
        } else {
            f += gridDim.x;
        }
    }
}

template <int LW, int ABS_FROM>
static hipError_t launch_wave(const LdpcDeviceCode& C, const LdpcWaveArgs& A, int grid, size_t lds, hipStream_t stream) {
    auto kern = ldpc_wave_kernel<LW, ABS_FROM>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, C.d_wave_lanec, C.d_wave_steps, C.d_wave_layer_end, C.d_ents, A);
    return hipGetLastError();
}

size_t ldpc_wave_msg_bytes_per_frame(const LdpcDeviceCode& C) {
    const int lw = C.wave_lw;
    return (size_t)C.R * 8 * (lw == 1 ? 1 : lw == 2 ? 2 : 4);
}
size_t ldpc_wave_lds_bytes(const LdpcDeviceCode& C) { return (size_t)((C.N + 8 + 15) / 16) * 16; }

hipError_t ldpc_wave_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force, uint8_t* hard, int hard_stride,
                                   int8_t* post, int32_t* trials, uint8_t* msg_ws, int grid, hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws) {
    LdpcWaveArgs A;
    A.llr = llr; A.hard = hard; A.post = post; A.trials = trials; A.msg_ws = msg_ws; A.sgn_ws = sgn_ws; A.work_ctr = work_ctr;
    A.nframes = nframes; A.N = C.N; A.K = C.K; A.R = C.R; A.q = C.q; A.nsteps = C.wave_nsteps;
    A.absent_base = C.wave_absent_base; A.synd_base = C.synd_base; A.max_deg = C.max_deg; A.max_trials = max_trials; A.force = force; A.hard_stride = hard_stride;
    if (work_ctr) {
        hipError_t e = hipMemsetAsync(work_ctr, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
    }
    const size_t lds = ldpc_wave_lds_bytes(C);
    const int abs_from = C.wave_nl_min / 8;         // first slot index kk that holds a link slot >= the smallest row degree
#define WAVE_CASE(LW_, AF_) if (C.wave_lw == LW_ && abs_from == AF_) return launch_wave<LW_, AF_>(C, A, grid, lds, stream)
    WAVE_CASE(1, 0); WAVE_CASE(1, 1);
    WAVE_CASE(2, 0); WAVE_CASE(2, 1); WAVE_CASE(2, 2);
    WAVE_CASE(3, 0); WAVE_CASE(3, 1); WAVE_CASE(3, 2); WAVE_CASE(3, 3);
    WAVE_CASE(4, 1); WAVE_CASE(4, 2); WAVE_CASE(4, 3); WAVE_CASE(4, 4);
#undef WAVE_CASE
    return hipErrorInvalidValue;
}

}  // namespace s2
