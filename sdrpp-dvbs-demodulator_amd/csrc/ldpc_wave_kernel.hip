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

namespace s2 {

struct LdpcWaveArgs {
    const int8_t* llr;          // [nframes][N]
    uint8_t* hard;              // [nframes][hard_stride]
    int8_t* post;               // optional [nframes][N]
    int32_t* trials;            // [nframes]
    uint8_t* msg_ws;            // [gridDim.x][R * 8 * element bytes]
    uint32_t* sgn_ws;           // [gridDim.x][SGN_WS_DWORDS]
    unsigned int* work_ctr;
    const uint32_t* lanec;      // ldpc_wave_plan.h
    const uint16_t* steps;
    const uint16_t* step_layer;
    const uint32_t* ents;       // the lane-per-row plan's table (syndrome-check part)
    int nframes, N, K, R, q, nsteps, nl_min, absent_base, synd_base, max_trials, force, hard_stride;
};

#define WQUAD(x_, ctrl) __builtin_amdgcn_update_dpp(0, (x_), (ctrl), 0xf, 0xf, true)

template <int LW> struct WaveElem { typedef uint32_t type; };
template <> struct WaveElem<1> { typedef uint8_t type; };
template <> struct WaveElem<2> { typedef uint16_t type; };

template <int LW, int MAXDEG_SYND>
__global__ __launch_bounds__(64) void ldpc_wave_kernel(LdpcWaveArgs A) {
    extern __shared__ __attribute__((aligned(16))) int8_t wpost[];
    typedef typename WaveElem<LW>::type elem_t;
    constexpr int U = LDPC_WAVE_CHUNK;
    __shared__ int s_next;
    const int lane = threadIdx.x, l8 = lane & 7, g = lane >> 3;
    const int N = A.N, K = A.K, R = A.R, q = A.q;
    int8_t* __restrict__ post = wpost;
    elem_t* __restrict__ msg = reinterpret_cast<elem_t*>(A.msg_ws) + (size_t)blockIdx.x * (size_t)R * 8;
    uint32_t* __restrict__ sgn = A.sgn_ws + (size_t)blockIdx.x * SGN_WS_DWORDS;
    const int dummy = N + l8;                     // a byte behind the frame (the LDS piece is padded): where absent link slots write

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
                const bool bad = z != 0 || syndromes_bad<MAXDEG_SYND>(q, A.synd_base, A.ents, sgn, lane, 64);
                const bool any = __ballot(bad) != 0;
                if (A.force) { ret = any ? -1 : A.max_trials; done = true; }
                else if (!any) { ret = it; done = true; }
                else if (it == A.max_trials) { ret = -1; done = true; }
            }
            if (done) break;
            // ---- one layered sweep (LDPCDecoder::update)
            const bool first = (it == 0);
            int cur_layer = -1;
            uint32_t thr[LW], cA[LW], cB[LW], absm = 0;
            // step list and message elements travel a chunk ahead
            uint32_t jn[U], jm[U];
            uint32_t rn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { jn[u] = A.steps[u * 8 + g]; jm[u] = A.steps[(U + u) * 8 + g]; }
#pragma unroll
            for (int u = 0; u < U; ++u) rn[u] = (!first && jn[u] != LDPC_WAVE_NOROW) ? (uint32_t)msg[(size_t)jn[u] * 8 + l8] : 0u;
            for (int s0 = 0; s0 < A.nsteps; s0 += U) {
                uint32_t jq[U], rq[U];
#pragma unroll
                for (int u = 0; u < U; ++u) jq[u] = A.steps[(s0 + 2 * U + u) * 8 + g];
#pragma unroll
                for (int u = 0; u < U; ++u) rq[u] = (!first && jm[u] != LDPC_WAVE_NOROW) ? (uint32_t)msg[(size_t)jm[u] * 8 + l8] : 0u;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int layer = (int)A.step_layer[s0 + u];          // wave-uniform
                    if (layer != cur_layer) {
                        cur_layer = layer;
#pragma unroll
                        for (int kk = 0; kk < LW; ++kk) {
                            const uint32_t c = A.lanec[((size_t)layer * 8 + l8) * LW + kk];
                            thr[kk] = c & 0xffffu; cA[kk] = c >> 16; cB[kk] = (c >> 16) - 360u;
                        }
                        absm = A.lanec[A.absent_base + layer * 8 + l8];
                    }
                    const bool valid = jn[u] != LDPC_WAVE_NOROW;
                    const uint32_t rowid = valid ? jn[u] : (uint32_t)(360 * layer);
                    const int j = (int)rowid - 360 * layer;
                    const uint32_t rec = rn[u];
                    int a[LW], v[LW], mg[LW];
#pragma unroll
                    for (int kk = 0; kk < LW; ++kk) {
                        a[kk] = j + (int)((uint32_t)j >= thr[kk] ? cB[kk] : cA[kk]);
                        if ((kk + 1) * 8 > A.nl_min && ((absm >> kk) & 1u)) a[kk] = dummy;                 // (uniform test first: only tail slots can be absent)
                    }
                    if (layer == 0 && l8 == 1 && j == 0) a[0] = dummy;                                     // row 0 of layer 0 has no previous parity bit
                    int x[LW];
#pragma unroll
                    for (int kk = 0; kk < LW; ++kk) x[kk] = post[a[kk]];
                    int min0 = 255, min1 = 255, sx = 0;
#pragma unroll
                    for (int kk = 0; kk < LW; ++kk) {
                        const int m = (int)__builtin_amdgcn_sbfe((int)rec, 8 * kk, 8);
                        int vv = clamp8(x[kk] - m);
                        int gg = mag_of(vv);
                        if ((kk + 1) * 8 > A.nl_min && ((absm >> kk) & 1u)) { vv = 0; gg = 127; }
                        if (kk == 0 && layer == 0 && l8 == 1 && j == 0) { vv = 0; gg = 127; }
                        v[kk] = vv; mg[kk] = gg;
                        min1 = min(min1, max(min0, gg));
                        min0 = min(min0, gg);
                        sx ^= vv;
                    }
                    // the row's totals over its 8 lanes: two smallest magnitudes (with multiplicity) and the sign
#define WJOIN(ctrl) do { const int o0 = WQUAD(min0, ctrl), o1 = WQUAD(min1, ctrl); min1 = min(max(min0, o0), min(min1, o1)); min0 = min(min0, o0); sx ^= WQUAD(sx, ctrl); } while (0)
                    WJOIN(0xB1);             // quad_perm [1,0,3,2]
                    WJOIN(0x4E);             // quad_perm [2,3,0,1]
                    WJOIN(0x141);            // row_half_mirror: the other quad of the 8-lane group
#undef WJOIN
                    uint32_t ro = 0;
                    if (valid) {
#pragma unroll
                        for (int kk = 0; kk < LW; ++kk) {
                            const int other = (mg[kk] == min0) ? min1 : min0;
                            const int neg = (sx ^ v[kk]) >> 31;
                            const int nm = med3i((other ^ neg) - neg, -32, 31);
                            post[a[kk]] = (int8_t)clamp8(v[kk] + nm);
                            ro |= ((uint32_t)nm & 0xffu) << (8 * kk);
                        }
                        msg[(size_t)rowid * 8 + l8] = (elem_t)ro;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { jn[u] = jm[u]; jm[u] = jq[u]; rn[u] = rq[u]; }
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
                if (nbytes == 8) *reinterpret_cast<uint2*>(hd + base / 8) = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
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

template <int LW, int MAXDEG_SYND>
static hipError_t launch_wave(const LdpcWaveArgs& A, int grid, size_t lds, hipStream_t stream) {
    auto kern = ldpc_wave_kernel<LW, MAXDEG_SYND>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, A);
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
    A.lanec = C.d_wave_lanec; A.steps = C.d_wave_steps; A.step_layer = C.d_wave_step_layer; A.ents = C.d_ents;
    A.nframes = nframes; A.N = C.N; A.K = C.K; A.R = C.R; A.q = C.q; A.nsteps = C.wave_nsteps; A.nl_min = C.wave_nl_min;
    A.absent_base = C.wave_absent_base; A.synd_base = C.synd_base; A.max_trials = max_trials; A.force = force; A.hard_stride = hard_stride;
    if (work_ctr) {
        hipError_t e = hipMemsetAsync(work_ctr, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
    }
    const size_t lds = ldpc_wave_lds_bytes(C);
    // (the syndrome check is instantiated per table width, like the lane-per-row kernels)
#define WAVE_CASE(LW_, MD_) if (C.wave_lw == LW_ && C.max_deg == MD_) return launch_wave<LW_, MD_>(A, grid, lds, stream)
    WAVE_CASE(1, 2); WAVE_CASE(1, 3); WAVE_CASE(1, 4); WAVE_CASE(1, 5);
    WAVE_CASE(2, 8); WAVE_CASE(2, 9); WAVE_CASE(2, 11); WAVE_CASE(2, 12);
    WAVE_CASE(3, 16); WAVE_CASE(3, 17); WAVE_CASE(3, 20);
    WAVE_CASE(4, 25); WAVE_CASE(4, 28);
#undef WAVE_CASE
    return hipErrorInvalidValue;
}

}  // namespace s2
