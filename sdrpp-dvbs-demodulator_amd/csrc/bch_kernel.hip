// Outer-code kernels for DVB-S2 BBFRAMEs on gfx950: BCH syndromes, BCH correction, BB de-scrambling.
//
// Replaces (bit-exact, including the "uncorrectable -> frame left as is, return -1" behaviour):
//   BBFrameBCH::decode                            dvbs2/codings/bbframe_bch.cpp:380-405
//   BoseChaudhuriHocquenghemDecoder::operator()   bch/bose_chaudhuri_hocquenghem_decoder.hh:83-143
//   compute_syndromes / update_syndromes          same file :41-71      (per-bit Horner, 2t log/exp fmas per bit)
//   ReedSolomonErrorCorrection (BM, Chien, deg-1/2 closed forms, Forney)  bch/reed_solomon_error_correction.hh:34-405
//   GF(2^m) log/exp arithmetic incl. uint16 wrap  bch/galois_field.hh:122-362
//   BBFrameDescrambler::work                      dvbs2/codings/bbframe_descramble.cpp:138-143
//
// Syndromes: S_i = r(alpha^i).  Only the t odd ones are evaluated from the data, byte-wise Horner
// P <- P*alpha^(8i) + T_i[byte] with three 256-entry LDS tables per root (constant multiply split into
// low/high byte); a frame is cut into 256 chunks, chunk partials are shifted by alpha^(i*bits_after) and
// XOR-reduced.  Even syndromes are squares (binary code).  Bytes read per frame: nbch/8, once -- HBM-bound,
// negligible next to the LDPC stage.  Correction (rare) runs one wave per frame: Berlekamp-Massey etc. on
// lane 0, Chien search across the 64 lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace s2 {

struct GfDev {
    const uint16_t* __restrict__ LOG;
    const uint16_t* __restrict__ EXP;
    int N;
    // Index arithmetic in the tables' uint16 element type, wrap-around included (galois_field.hh:237-268)
    __device__ __forceinline__ uint16_t imul(uint16_t a, uint16_t b) const {
        uint16_t tmp = (uint16_t)(a + b);
        return (N - (int)a <= (int)b) ? (uint16_t)(tmp - N) : tmp;
    }
    __device__ __forceinline__ uint16_t idiv(uint16_t a, uint16_t b) const {
        uint16_t tmp = (uint16_t)(a - b);
        return (a < b) ? (uint16_t)(tmp + N) : tmp;
    }
    __device__ __forceinline__ uint16_t vmul(uint16_t a, uint16_t b) const { return (!a || !b) ? (uint16_t)0 : EXP[imul(LOG[a], LOG[b])]; }
    __device__ __forceinline__ uint16_t vdiv(uint16_t a, uint16_t b) const { return !a ? (uint16_t)0 : EXP[idiv(LOG[a], LOG[b])]; }
    __device__ __forceinline__ uint16_t vmuli(uint16_t a, uint16_t idx) const { return !a ? (uint16_t)0 : EXP[imul(LOG[a], idx)]; }
};

__global__ __launch_bounds__(256) void bch_syndromes_kernel(BchDeviceCode C, const uint8_t* __restrict__ frames, int frame_stride,
                                                            int nbch, int nframes, uint16_t* __restrict__ syn_out) {
    __shared__ uint16_t tab[12 * 3 * 256];
    __shared__ uint16_t red[4][12];
    __shared__ uint16_t sfin[32];
    const int t = C.t;
    for (int i = threadIdx.x; i < t * 768; i += 256) tab[i] = C.d_syn_tab[i];
    __syncthreads();
    GfDev G{C.d_log, C.d_exp, C.N};
    const int nbytes = nbch / 8;
    const int chunk = (nbytes + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
        const uint8_t* __restrict__ fr = frames + (size_t)f * frame_stride;
        int b0 = tid * chunk, b1 = min(b0 + chunk, nbytes);
        uint32_t P[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) P[r] = 0;
        for (int b = b0; b < b1; ++b) {
            uint32_t byte = fr[b];
#pragma unroll
            for (int r = 0; r < 12; ++r)
                if (r < t) {
                    const uint16_t* T = tab + r * 768;
                    P[r] = (uint32_t)T[256 + (P[r] & 0xff)] ^ (uint32_t)T[512 + (P[r] >> 8)] ^ (uint32_t)T[byte];
                }
        }
        const long after = (b1 > b0) ? (long)nbch - 8L * b1 : 0;  // bits that follow this chunk
#pragma unroll
        for (int r = 0; r < 12; ++r)
            if (r < t) {
                uint32_t v = P[r];
                if (v) {
                    int sh = (int)(((long)(2 * r + 1) * after) % C.N);
                    int e = (int)G.LOG[v] + sh;
                    if (e >= C.N) e -= C.N;
                    v = G.EXP[e];
                }
                // XOR-reduce over the wave
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor((int)v, o);
                if (lane == 0) red[wave][r] = (uint16_t)v;
            }
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < 32; ++i) sfin[i] = 0;
            int nz = 0;
            for (int r = 0; r < t; ++r) { sfin[2 * r] = (uint16_t)(red[0][r] ^ red[1][r] ^ red[2][r] ^ red[3][r]); nz |= sfin[2 * r]; }  // S_(2r+1)
            if (nz)   // clean frame: the even syndromes are squares of zeros
                for (int k = 1; k <= t; ++k) sfin[2 * k - 1] = G.vmul(sfin[k - 1], sfin[k - 1]);                  // S_2k = S_k^2
            sfin[31] = nz ? 1 : 0;                                                                                 // "needs correction" flag
        }
        __syncthreads();
        if (tid < 32) syn_out[(size_t)f * 32 + tid] = sfin[tid];
        __syncthreads();
    }
}

__device__ __forceinline__ void xor_be_bit(uint8_t* buf, int pos) { buf[pos / 8] ^= (uint8_t)(1u << (7 - pos % 8)); }

__global__ __launch_bounds__(64) void bch_correct_kernel(BchDeviceCode C, uint8_t* __restrict__ frames, int frame_stride, int nbch,
                                                         int kbch, int nframes, const uint16_t* __restrict__ syn_in,
                                                         int32_t* __restrict__ corrections) {
    __shared__ uint16_t s_loc[32];      // locator
    __shared__ uint16_t s_pos[32];      // locations
    __shared__ int s_deg, s_count, s_state;
    GfDev G{C.d_log, C.d_exp, C.N};
    const int NR = 2 * C.t;
    const int lane = threadIdx.x;
    for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
        const uint16_t* __restrict__ syn = syn_in + (size_t)f * 32;
        // state: 0 = clean, 1 = need Chien, 2 = locations ready, -1 = failed
        // fast path: the syndrome kernel left a flag; a clean frame costs one load
        if (syn[31] == 0) {
            if (lane == 0 && corrections) corrections[f] = 0;
            continue;
        }
        if (lane == 0) {
            int nonzero = 0;
            for (int i = 0; i < NR; ++i) nonzero += !!syn[i];
            s_state = 0; s_count = 0; s_deg = 0;
            if (nonzero) {
                // Berlekamp-Massey (reed_solomon_error_correction.hh:226-276, count = 0)
                uint16_t Cc[33], B[33], T[33];
                for (int i = 0; i <= NR; ++i) Cc[i] = 0;
                Cc[0] = 1;
                for (int i = 0; i <= NR; ++i) B[i] = Cc[i];
                int L = 0;
                for (int n = 0, m = 1; n < NR; ++n) {
                    uint16_t d = syn[n];
                    for (int i = 1; i <= L; ++i) d ^= G.vmul(Cc[i], syn[n - i]);
                    if (!d) {
                        ++m;
                    } else {
                        for (int i = 0; i < m; ++i) T[i] = Cc[i];
                        for (int i = m; i <= NR; ++i) T[i] = (uint16_t)(G.vmul(d, B[i - m]) ^ Cc[i]);
                        if (2 * L <= n) {
                            L = n + 1 - L;
                            for (int i = 0; i <= NR; ++i) B[i] = G.vdiv(Cc[i], d);
                            m = 1;
                        } else {
                            ++m;
                        }
                        for (int i = 0; i <= NR; ++i) Cc[i] = T[i];
                    }
                }
                int deg = L;
                bool fail = false;
                while (!Cc[deg])
                    if (--deg < 0) { fail = true; break; }
                if (fail) {
                    s_state = -1;
                } else {
                    for (int i = 0; i <= NR; ++i) s_loc[i] = Cc[i];
                    s_deg = deg;
                    if (deg == 1) {
                        s_pos[0] = G.idiv(G.idiv(G.LOG[Cc[0]], G.LOG[Cc[1]]), 1);
                        s_count = 1; s_state = 2;
                    } else if (deg == 2) {
                        s_state = 2;
                        if (!Cc[1] || !Cc[0]) {
                            s_count = 0;
                        } else {
                            uint16_t a = Cc[2], b = Cc[1], c = Cc[0];
                            uint16_t ba = G.vdiv(b, a);
                            uint16_t Rr = C.d_imap[G.vdiv(G.vmul(a, c), G.vmul(b, b))];
                            if (!Rr) {
                                s_count = 0;
                            } else {
                                uint16_t v0 = G.vmul(ba, Rr);
                                s_pos[0] = G.idiv(G.LOG[v0], 1);
                                s_pos[1] = G.idiv(G.LOG[(uint16_t)(v0 ^ ba)], 1);
                                s_count = 2;
                            }
                        }
                    } else {
                        s_state = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (s_state == 1) {
            // Chien search: location i is a root iff sum_j locator[j]*alpha^(j*(i+1)) == 0  (:40-61)
            const int deg = s_deg;
            int lg[25];
            for (int jj = 0; jj <= deg && jj < 25; ++jj) lg[jj] = s_loc[jj] ? (int)G.LOG[s_loc[jj]] : -1;
            for (int i = lane; i < C.N; i += 64) {
                uint32_t sum = s_loc[0];
                for (int jj = 1; jj <= deg; ++jj) {
                    if (lg[jj] >= 0) {
                        long e = (long)lg[jj] + (long)jj * (long)(i + 1);
                        sum ^= G.EXP[(int)(e % C.N)];
                    }
                }
                if (!sum) {
                    int slot = atomicAdd(&s_count, 1);
                    if (slot < 32) s_pos[slot] = (uint16_t)i;
                }
            }
            __syncthreads();
            if (lane == 0) s_state = 2;
        }
        __syncthreads();
        if (lane == 0) {
            int result = 0;
            if (s_state == -1) {
                result = -1;
            } else if (s_state == 2) {
                const int deg = s_deg, count = s_count;
                if (count < deg || count > 32) {
                    result = -1;
                } else {
                    // Forney (:133-218); the evaluator uses `count` as locator degree
                    uint16_t evaluator[32], magnitudes[32];
                    int etmp = count < NR - 1 ? count : NR - 1;
                    int edeg = -1;
                    for (int i = 0; i <= etmp; ++i) {
                        evaluator[i] = G.vmul(syn[i], s_loc[0]);
                        for (int jj = 1; jj <= i; ++jj) evaluator[i] ^= G.vmul(syn[i - jj], s_loc[jj]);
                        if (evaluator[i]) edeg = i;
                    }
                    for (int i = 0; i < count; ++i) {
                        uint16_t root = G.imul(s_pos[i], 1), tmp = root;
                        uint16_t eval = evaluator[0];
                        for (int jj = 1; jj <= edeg; ++jj) {
                            eval ^= G.vmuli(evaluator[jj], tmp);
                            tmp = G.imul(tmp, root);
                        }
                        if (!eval) { magnitudes[i] = 0; continue; }
                        uint16_t deriv = s_loc[1];
                        uint16_t root2 = G.imul(root, root), tmp2 = root2;
                        for (int jj = 3; jj <= count; jj += 2) {
                            deriv ^= G.vmuli(s_loc[jj], tmp2);
                            tmp2 = G.imul(tmp2, root2);
                        }
                        magnitudes[i] = G.EXP[G.idiv(G.LOG[eval], G.LOG[deriv])];
                    }
                    const int short_by = C.K_full - kbch;
                    if (count <= 0) {
                        result = count;
                    } else {
                        bool bad = false;
                        for (int i = 0; i < count; ++i)
                            if ((int)s_pos[i] < short_by) bad = true;
                        if (!bad)
                            for (int i = 0; i < count; ++i)
                                if (1 < (int)magnitudes[i]) bad = true;
                        if (bad) {
                            result = -1;
                        } else {
                            uint8_t* fr = frames + (size_t)f * frame_stride;
                            for (int i = 0; i < count; ++i)
                                if (magnitudes[i]) { xor_be_bit(fr, (int)s_pos[i] - short_by); ++result; }
                        }
                    }
                }
            }
            if (corrections) corrections[f] = result;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bb_descramble_kernel(const uint8_t* __restrict__ frames, int frame_stride,
                                                            const uint8_t* __restrict__ prbs, int out_bytes, int nframes,
                                                            uint8_t* __restrict__ out) {
    for (int f = blockIdx.y; f < nframes; f += gridDim.y)
        for (int b = blockIdx.x * 256 + threadIdx.x; b < out_bytes; b += gridDim.x * 256)
            out[(size_t)f * out_bytes + b] = frames[(size_t)f * frame_stride + b] ^ prbs[b];
}

hipError_t bch_syndromes_launch(const BchDeviceCode& C, const uint8_t* frames, int frame_stride, int nbch, int nframes,
                                uint16_t* syn, hipStream_t stream) {
    int grid = nframes < 4096 ? nframes : 4096;
    hipLaunchKernelGGL(bch_syndromes_kernel, dim3(grid), dim3(256), 0, stream, C, frames, frame_stride, nbch, nframes, syn);
    return hipGetLastError();
}
hipError_t bch_correct_launch(const BchDeviceCode& C, uint8_t* frames, int frame_stride, int nbch, int kbch, int nframes,
                              const uint16_t* syn, int32_t* corrections, hipStream_t stream) {
    int grid = nframes < 2048 ? nframes : 2048;
    hipLaunchKernelGGL(bch_correct_kernel, dim3(grid), dim3(64), 0, stream, C, frames, frame_stride, nbch, kbch, nframes, syn, corrections);
    return hipGetLastError();
}
hipError_t bb_descramble_launch(const uint8_t* frames, int frame_stride, const uint8_t* prbs, int out_bytes, int nframes,
                                uint8_t* out, hipStream_t stream) {
    int gx = (out_bytes + 255) / 256;
    if (gx > 32) gx = 32;
    int gy = nframes < 2048 ? nframes : 2048;
    hipLaunchKernelGGL(bb_descramble_kernel, dim3(gx, gy), dim3(256), 0, stream, frames, frame_stride, prbs, out_bytes, nframes, out);
    return hipGetLastError();
}

}  // namespace s2
