// Layered offset-min-sum LDPC decoder for DVB-S2, gfx950.
//
// Replaces BBFrameLDPC::decode (reference src/demod/dvbs2/codings/bbframe_ldpc.cpp:123-139) and the
// library under it (xdsopl-ldpc-pabr/layered_decoder.hh:23-133, algorithms.hh:206-277) -- bit-exact,
// including the sequential row order (see ldpc_plan.h) and the int8 saturation rules.
//
// Mapping (one workgroup = one frame, persistent over the batch):
//   * 384 threads; lane j < 360 owns row j of EVERY layer (a DVB-S2 layer = 360 rows, quasi-cyclic).
//   * the N int8 posteriors of the frame live in LDS for the whole decode (64.8 KB normal frame ->
//     2 frames per CU); information bits as [0,K), parity bits layer-major as K + 360*i + j.
//     Row j of a layer reads byte 360*r + (j - s) mod 360 of each linked group: consecutive lanes ->
//     consecutive bytes, conflict-free.
//   * check->bit messages: one fixed-size record per row (REC dwords, 1 byte per link) in a per-workgroup
//     global workspace.  Lane j re-reads only what lane j wrote, one coalesced vector load + store per
//     row per iteration, prefetched one layer ahead; the workspace of all resident workgroups
//     (<= 512 x ~260 KB) stays in L2 / Infinity Cache.
//   * rows of a layer that share a bit are ordered by the plan's levels (barrier per level).
// Arithmetic: int32 VALU emulating int8 saturating lanes.  Roofline: algorithmic bytes per frame =
// iters*4*edges + N + K/8 against HBM 8 TB/s; real HBM traffic is ~N + K/8 per
// frame because the state is on-chip -- the kernel is VALU/LDS-issue bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_plan.h"
#include "kernels.h"

namespace s2 {

struct LdpcKernelArgs {
    const LdpcLayerDesc* layers;
    const LdpcLinkEnt* ents;
    const LdpcRowInfo* rows;
    const int8_t* llr;     // [nframes][N]
    uint8_t* hard;         // [nframes][hard_stride] packed hard decisions of bits [0,K), MSB first
    int8_t* post;          // optional [nframes][N] posteriors (reference layout), may be null
    int32_t* trials;       // [nframes]
    uint32_t* msg_ws;      // [gridDim.x][R][REC]
    int nframes, N, K, R, q;
    int max_trials, force;
    int hard_stride;
};

__device__ __forceinline__ int med3i(int a, int lo, int hi) { return min(max(a, lo), hi); }  // folds to v_med3_i32 for lo <= hi
__device__ __forceinline__ int clamp8(int v) { return med3i(v, -128, 127); }

// |max(x,-127)| - 1 clamped at 0  == vqsub(vunsigned(vqabs(x)), 1)   (algorithms.hh:235-238)
__device__ __forceinline__ int mag_of(int in) {
    int a = in < 0 ? -in : in;
    return med3i(a - 1, 0, 126);
}

template <int REC>
__device__ __forceinline__ int rec_byte(const uint32_t (&rec)[REC], int k) {
    return (int)__builtin_amdgcn_sbfe((int)rec[k >> 2], (k & 3) * 8, 8);  // v_bfe_i32
}

template <int REC>
__device__ __forceinline__ void rec_load(uint32_t (&rec)[REC], const uint32_t* p) {
    if constexpr (REC == 1) {
        rec[0] = *p;
    } else if constexpr (REC == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        rec[0] = v.x; rec[1] = v.y;
    } else {
#pragma unroll
        for (int w = 0; w < REC; w += 4) {
            uint4 v = *reinterpret_cast<const uint4*>(p + w);
            rec[w] = v.x; rec[w + 1] = v.y; rec[w + 2] = v.z; rec[w + 3] = v.w;
        }
    }
}
template <int REC>
__device__ __forceinline__ void rec_store(const uint32_t (&rec)[REC], uint32_t* p) {
    if constexpr (REC == 1) {
        *p = rec[0];
    } else if constexpr (REC == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(rec[0], rec[1]);
    } else {
#pragma unroll
        for (int w = 0; w < REC; w += 4) *reinterpret_cast<uint4*>(p + w) = make_uint4(rec[w], rec[w + 1], rec[w + 2], rec[w + 3]);
    }
}

// new message for one link given the row totals (algorithms.hh:250-256 + clamp :275)
__device__ __forceinline__ int new_msg(int in, int mg, int min0, int min1, int sx) {
    int other = (mg == min0) ? min1 : min0;
    int neg = (sx ^ in) >> 31;             // 0 or -1
    int v = (other ^ neg) - neg;           // +-other
    return med3i(v, -32, 31);
}

// One sweep step for one layer.  CONF = layer has intra-layer shared bits.
template <int MAXDEG, int REC, bool CONF>
__device__ __forceinline__ void layer_update(int8_t* __restrict__ post, const LdpcKernelArgs& A, const LdpcLayerDesc L, int layer,
                                             int j, bool active, const uint32_t (&rec_in)[REC], uint32_t* __restrict__ rec_out_ptr) {
    constexpr int NL = MAXDEG + 2;
    int in[NL], mg[NL];
    int addr[MAXDEG];
    const int deg = L.deg;
    const LdpcLinkEnt* __restrict__ ents = A.ents + L.ent_off;
    uint32_t late = 0, early = 0, level = 1;
    if constexpr (CONF) {
        if (active) {
            LdpcRowInfo ri = A.rows[L.row_off + j];
            late = ri.late; early = ri.early; level = ri.level;
        }
    }
    int min0 = 255, min1 = 255, sx = 0;
    const int own = A.K + 360 * layer + j;
    const bool has_prev = (layer | j) != 0;
    const int prev = layer ? own - 360 : A.K + 360 * (A.q - 1) + j - 1;
    if (active) {
#pragma unroll
        for (int k = 0; k < MAXDEG; ++k) {
            if (k < deg) {
                LdpcLinkEnt e = ents[k];
                int t = j + (int)e.sb;
                if (t >= (int)e.thr) t -= 360;
                addr[k] = t;
                int x = post[t];
                int v = clamp8(x - rec_byte<REC>(rec_in, k));
                int m = mag_of(v);
                if constexpr (CONF) {
                    if ((late >> k) & 1) { v = 0; m = 255; }  // joins the totals at its level
                }
                in[k] = v; mg[k] = m;
                min1 = min(min1, max(min0, m));
                min0 = min(min0, m);
                sx ^= v;
            } else {
                in[k] = 0; mg[k] = 255; addr[k] = 0;
            }
        }
        {
            int x = post[own];
            int v = clamp8(x - rec_byte<REC>(rec_in, MAXDEG));
            int m = mag_of(v);
            in[MAXDEG] = v; mg[MAXDEG] = m;
            min1 = min(min1, max(min0, m));
            min0 = min(min0, m);
            sx ^= v;
        }
        if (has_prev) {
            int x = post[prev];
            int v = clamp8(x - rec_byte<REC>(rec_in, MAXDEG + 1));
            int m = mag_of(v);
            in[MAXDEG + 1] = v; mg[MAXDEG + 1] = m;
            min1 = min(min1, max(min0, m));
            min0 = min(min0, m);
            sx ^= v;
        } else {
            in[MAXDEG + 1] = 0; mg[MAXDEG + 1] = 255;
        }
    }
    if constexpr (CONF) {
        // level 1 rows have complete totals already: publish the links a later row waits for
        const int depth = L.depth;
        for (int lvl = 1; lvl <= depth; ++lvl) {
            if (lvl > 1) __syncthreads();
            if (active && level == (uint32_t)lvl) {
                if (lvl > 1) {
#pragma unroll
                    for (int k = 0; k < MAXDEG; ++k) {
                        if ((L.cmask >> k) & 1) {
                            if ((late >> k) & 1) {
                                int x = post[addr[k]];
                                int v = clamp8(x - rec_byte<REC>(rec_in, k));
                                int m = mag_of(v);
                                in[k] = v; mg[k] = m;
                                min1 = min(min1, max(min0, m));
                                min0 = min(min0, m);
                                sx ^= v;
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < MAXDEG; ++k) {
                    if ((L.cmask >> k) & 1) {
                        if ((early >> k) & 1) {
                            int nm = new_msg(in[k], mg[k], min0, min1, sx);
                            post[addr[k]] = (int8_t)clamp8(in[k] + nm);
                        }
                    }
                }
            }
        }
    }
    if (active) {
        uint32_t rec_out[REC];
#pragma unroll
        for (int w = 0; w < REC; ++w) rec_out[w] = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            bool present = (k < MAXDEG) ? (k < deg) : (k == MAXDEG ? true : has_prev);
            if (present) {
                int nm = new_msg(in[k], mg[k], min0, min1, sx);
                rec_out[k >> 2] |= ((uint32_t)nm & 0xffu) << ((k & 3) * 8);
                bool wr = true;
                if constexpr (CONF) {
                    if (k < MAXDEG) wr = !((early >> k) & 1);
                }
                if (wr) {
                    int a = (k < MAXDEG) ? addr[k] : (k == MAXDEG ? own : prev);
                    post[a] = (int8_t)clamp8(in[k] + nm);
                }
            }
        }
        rec_store<REC>(rec_out, rec_out_ptr);
    }
}

// LDPCDecoder::bad (layered_decoder.hh:28-45) for the rows owned by lane j; true if any is unsatisfied.
template <int MAXDEG>
__device__ __forceinline__ bool rows_bad(const int8_t* __restrict__ post, const LdpcKernelArgs& A, int j) {
    int badacc = 0;
    for (int layer = 0; layer < A.q; ++layer) {
        const LdpcLayerDesc L = A.layers[layer];
        const LdpcLinkEnt* __restrict__ ents = A.ents + L.ent_off;
        const int own = A.K + 360 * layer + j;
        int x = post[own];
        int sx = x;
        int zero = (x == 0);
        if (layer | j) {
            const int prev = layer ? own - 360 : A.K + 360 * (A.q - 1) + j - 1;
            x = post[prev];
            sx ^= x; zero |= (x == 0);
        }
#pragma unroll
        for (int k = 0; k < MAXDEG; ++k) {
            if (k < (int)L.deg) {
                LdpcLinkEnt e = ents[k];
                int t = j + (int)e.sb;
                if (t >= (int)e.thr) t -= 360;
                x = post[t];
                sx ^= x; zero |= (x == 0);
            }
        }
        badacc |= zero | ((sx >> 7) & 1);
    }
    return badacc != 0;
}

template <int MAXDEG, int REC>
__global__ __launch_bounds__(384) void ldpc_decode_kernel(LdpcKernelArgs A) {
    extern __shared__ __attribute__((aligned(16))) int8_t post[];
    const int j = threadIdx.x;
    const bool active = j < 360;
    const int N = A.N, K = A.K, R = A.R, q = A.q;
    uint32_t* __restrict__ msg = A.msg_ws + (size_t)blockIdx.x * (size_t)R * REC;

    for (int f = blockIdx.x; f < A.nframes; f += gridDim.x) {
        const int8_t* __restrict__ src = A.llr + (size_t)f * N;
        // information-bit LLRs: straight copy (K is a multiple of 8)
        for (int i = j; i < K / 8; i += 384) reinterpret_cast<uint2*>(post)[i] = reinterpret_cast<const uint2*>(src)[i];
        // parity LLRs: pty[360*i + jj] = llr[K + q*jj + i]   (layered_decoder.hh:124-126)
        for (int c = j; c < R; c += 384) {
            int jj = c / q, i = c - jj * q;
            post[K + 360 * i + jj] = src[K + c];
        }
        __syncthreads();

        int it = 0, ret = 0;
        while (true) {
            if (!A.force || it == A.max_trials) {
                bool bad = active ? rows_bad<MAXDEG>(post, A, j) : false;
                int any = __syncthreads_or(bad ? 1 : 0);
                if (A.force) { ret = any ? -1 : A.max_trials; break; }
                if (!any) { ret = it; break; }
                if (it == A.max_trials) { ret = -1; break; }
            }
            // ---- one layered sweep (LDPCDecoder::update)
            uint32_t rec_next[REC];
#pragma unroll
            for (int w = 0; w < REC; ++w) rec_next[w] = 0;
            const bool first = (it == 0);
            if (!first && active) rec_load<REC>(rec_next, msg + (size_t)j * REC);
            for (int layer = 0; layer < q; ++layer) {
                uint32_t rec[REC];
#pragma unroll
                for (int w = 0; w < REC; ++w) rec[w] = rec_next[w];
                uint32_t* rp = msg + ((size_t)layer * 360 + j) * REC;
                if (!first && active && layer + 1 < q) rec_load<REC>(rec_next, rp + 360 * REC);
                const LdpcLayerDesc L = A.layers[layer];
                if (L.depth == 1) layer_update<MAXDEG, REC, false>(post, A, L, layer, j, active, rec, rp);
                else layer_update<MAXDEG, REC, true>(post, A, L, layer, j, active, rec, rp);
                __syncthreads();
            }
            ++it;
        }

        // ---- outputs
        if (j == 0) A.trials[f] = ret;
        // hard decisions of [0,K): 64 bits per wave step via ballot, MSB-first bytes (module_dvbs2_demod.cpp:357-360)
        {
            uint8_t* __restrict__ hd = A.hard + (size_t)f * A.hard_stride;
            const int lane = j & 63, wave = j >> 6;
            for (int base = wave * 64; base < K; base += 6 * 64) {
                int idx = base + lane;
                int neg = (idx < K) ? (post[idx] < 0) : 0;
                unsigned long long b = __ballot(neg);
                b = __builtin_bswap64(__brevll(b));
                if (lane == 0) {
                    int nbytes = min(8, (K - base) / 8);
                    if (nbytes == 8) *reinterpret_cast<uint2*>(hd + base / 8) = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
                    else for (int n = 0; n < nbytes; ++n) hd[base / 8 + n] = (uint8_t)(b >> (8 * n));
                }
            }
        }
        if (A.post) {
            int8_t* __restrict__ dst = A.post + (size_t)f * N;
            for (int i = j; i < K / 8; i += 384) reinterpret_cast<uint2*>(dst)[i] = reinterpret_cast<const uint2*>(post)[i];
            for (int c = j; c < R; c += 384) {
                int jj = c / q, i = c - jj * q;
                dst[K + c] = post[K + 360 * i + jj];
            }
        }
        __syncthreads();
    }
}

template <int MAXDEG, int REC>
static hipError_t launch_ldpc(const LdpcKernelArgs& A, int grid, hipStream_t stream) {
    size_t lds = (size_t)((A.N + 15) / 16) * 16;
    auto kern = ldpc_decode_kernel<MAXDEG, REC>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(384), lds, stream, A);
    return hipGetLastError();
}

template <int MAXDEG, int REC>
static int occupancy_ldpc(int N) {
    int nb = 0;
    size_t lds = (size_t)((N + 15) / 16) * 16;
    auto kern = ldpc_decode_kernel<MAXDEG, REC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 384, lds) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}

#define LDPC_DISPATCH(FN, ...)                                                             \
    switch (max_deg) {                                                                     \
        case 2: return FN<2, 1>(__VA_ARGS__);                                              \
        case 3: return FN<3, 2>(__VA_ARGS__);                                              \
        case 4: return FN<4, 2>(__VA_ARGS__);                                              \
        case 5: return FN<5, 2>(__VA_ARGS__);                                              \
        case 8: return FN<8, 4>(__VA_ARGS__);                                              \
        case 9: return FN<9, 4>(__VA_ARGS__);                                              \
        case 11: return FN<11, 4>(__VA_ARGS__);                                            \
        case 12: return FN<12, 4>(__VA_ARGS__);                                            \
        case 16: return FN<16, 8>(__VA_ARGS__);                                            \
        case 17: return FN<17, 8>(__VA_ARGS__);                                            \
        case 20: return FN<20, 8>(__VA_ARGS__);                                            \
        case 25: return FN<25, 8>(__VA_ARGS__);                                            \
        case 28: return FN<28, 8>(__VA_ARGS__);                                            \
        default: break;                                                                    \
    }

int ldpc_blocks_per_cu(int max_deg, int N) {
    LDPC_DISPATCH(occupancy_ldpc, N)
    return 1;
}

hipError_t ldpc_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force,
                              uint8_t* hard, int hard_stride, int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid,
                              hipStream_t stream) {
    LdpcKernelArgs A;
    A.layers = C.d_layers; A.ents = C.d_ents; A.rows = C.d_rows;
    A.llr = llr; A.hard = hard; A.post = post; A.trials = trials; A.msg_ws = msg_ws;
    A.nframes = nframes; A.N = C.N; A.K = C.K; A.R = C.R; A.q = C.q;
    A.max_trials = max_trials; A.force = force; A.hard_stride = hard_stride;
    const int max_deg = C.max_deg;
    LDPC_DISPATCH(launch_ldpc, A, grid, stream)
    return hipErrorInvalidValue;
}

}  // namespace s2
