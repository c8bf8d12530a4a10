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
// negligible next to the LDPC stage.  Correction runs one 256-thread workgroup per frame with non-zero syndromes: Berlekamp-
// Massey and Forney on thread 0 (short, table-lookup bound), the Chien search over the whole field across the 256 threads with
// per-term constant multipliers kept as split byte tables in LDS (a frame the LDPC decoder could not fix costs ~40 us here; the
// first version -- one wave, a 64-bit modulo and a global table lookup per term and position -- took milliseconds, which only showed
// once the benchmark input contained frames that do not decode).
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
                                                            int nbch, int nframes, uint16_t* __restrict__ syn_out, int32_t* __restrict__ todo,
                                                            int32_t* __restrict__ corrections) {
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
            // a frame with a syndrome joins the correction kernel's work list, a clean one is done
            if (nz) todo[2 + atomicAdd(&todo[0], 1)] = f;
            else if (corrections) corrections[f] = 0;
        }
        __syncthreads();
        if (tid < 32) syn_out[(size_t)f * 32 + tid] = sfin[tid];
        __syncthreads();
    }
}

__device__ __forceinline__ void xor_be_bit(uint8_t* buf, int pos) { buf[pos / 8] ^= (uint8_t)(1u << (7 - pos % 8)); }

constexpr int BCH_CT = 256;         // threads of a correction workgroup
constexpr int BCH_MAXDEG = 24;      // a locator found from 2t <= 24 syndromes has degree <= 24
__global__ __launch_bounds__(BCH_CT) void bch_correct_kernel(BchDeviceCode C, uint8_t* __restrict__ frames, int frame_stride, int nbch,
                                                             int kbch, int nframes, const uint16_t* __restrict__ syn_in, int32_t* __restrict__ todo,
                                                             int32_t* __restrict__ corrections) {
    __shared__ int s_item;
    __shared__ uint16_t s_loc[32];      // locator
    __shared__ uint16_t s_pos[32];      // locations
    __shared__ int s_deg, s_count, s_state;
    __shared__ uint16_t s_C[40], s_B[40], s_d;       // Berlekamp-Massey: connection polynomial, its last copy, the discrepancy
    __shared__ uint16_t s_ev[32], s_mag[32];         // Forney: error evaluator, magnitudes
    __shared__ int s_edeg;
    __shared__ uint16_t s_mul[BCH_MAXDEG][2][256];   // x * alpha^(BCH_CT * j): low-byte and high-byte tables of term j
    GfDev G{C.d_log, C.d_exp, C.N};
    const int NR = 2 * C.t;
    const int lane = threadIdx.x;
    // the frames with a syndrome, one by one off the list the syndrome kernel left (work counter in todo[1]): how many there are and where they sit is the
    // channel's business, and a frame costs a few hundred microseconds of dependent lookups -- a fixed frame-to-workgroup map ends with its unluckiest workgroup
    const int ntodo = todo[0];
    while (true) {
        if (lane == 0) s_item = atomicAdd(&todo[1], 1);
        __syncthreads();
        const int item = s_item;
        __syncthreads();
        if (item >= ntodo) break;
        const int f = todo[2 + item];
        const uint16_t* __restrict__ syn = syn_in + (size_t)f * 32;
        // state: 0 = clean, 1 = need Chien, 2 = locations ready, -1 = failed
        // Berlekamp-Massey (reed_solomon_error_correction.hh:226-276, count = 0) with the polynomials in LDS and the inner loops --
        // discrepancy, T = C + d x^m B, B = C / d -- spread over the threads (the serial form with per-thread arrays lived in scratch
        // memory: thousands of dependent memory round trips per frame)
        if (lane <= NR) { s_C[lane] = lane == 0 ? 1 : 0; s_B[lane] = lane == 0 ? 1 : 0; }
        if (lane == 0) { s_state = 0; s_count = 0; s_deg = 0; }
        __syncthreads();
        int L = 0;
        for (int n = 0, m = 1; n < NR; ++n) {
            // d = syn[n] + sum_{i=1..L} C[i] syn[n-i]: products by threads 1..L, XOR-reduced through LDS
            uint32_t prod = 0;
            if (lane >= 1 && lane <= L) prod = G.vmul(s_C[lane], syn[n - lane]);
            if (lane == 0) prod = syn[n];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) prod ^= (uint32_t)__shfl_xor((int)prod, o);     // (L <= NR <= 24: everything sits in wave 0)
            if (lane == 0) s_d = (uint16_t)prod;
            __syncthreads();
            const uint16_t d = s_d;
            if (!d) {
                ++m;
            } else {
                uint16_t tnew = 0;
                if (lane <= NR) tnew = lane < m ? s_C[lane] : (uint16_t)(G.vmul(d, s_B[lane - m]) ^ s_C[lane]);
                const bool grow = 2 * L <= n;
                uint16_t bnew = 0;
                if (grow && lane <= NR) bnew = G.vdiv(s_C[lane], d);
                __syncthreads();
                if (lane <= NR) { s_C[lane] = tnew; if (grow) s_B[lane] = bnew; }
                if (grow) { L = n + 1 - L; m = 1; } else ++m;
            }
            __syncthreads();
        }
        if (lane == 0) {
            const uint16_t* Cc = s_C;
            int nonzero = 1;            // (a flagged frame has a non-zero syndrome)
            if (nonzero) {
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
            // Chien search: location i is a root iff sum_j locator[j]*alpha^(j*(i+1)) == 0  (:40-61), i over the WHOLE field (the
            // reference counts every root; locations in the shortened part make the frame uncorrectable further down).
            // Thread t takes i = t, t + 256, ...: term j starts at locator[j]*alpha^(j*(t+1)) and is multiplied by the constant
            // alpha^(256 j) per step -- a GF(2)-linear map, applied as two byte-table lookups in LDS.
            const int deg = s_deg;
            for (int e = lane; e < deg * 512; e += BCH_CT) {
                const int j = e >> 9, h = (e >> 8) & 1, x = e & 255;
                const uint32_t v = h ? (uint32_t)x << 8 : (uint32_t)x;
                uint16_t r = 0;
                if (v && v <= (uint32_t)C.N) {            // (GF(2^14): the high-byte table only holds values below 2^14)
                    const int sh = (int)(((long)BCH_CT * (j + 1)) % C.N);
                    int lg2 = (int)G.LOG[v] + sh;
                    if (lg2 >= C.N) lg2 -= C.N;
                    r = G.EXP[lg2];
                }
                s_mul[j][h][x] = r;
            }
            uint32_t term[BCH_MAXDEG];
#pragma unroll
            for (int jj = 0; jj < BCH_MAXDEG; ++jj) {
                term[jj] = 0;
                if (jj < deg) {
                    const uint16_t c = s_loc[jj + 1];
                    if (c) {
                        const long e = (long)G.LOG[c] + (long)(jj + 1) * (long)(lane + 1);
                        term[jj] = G.EXP[(int)(e % C.N)];
                    }
                }
            }
            __syncthreads();
            const uint32_t c0 = s_loc[0];
            for (int i = lane; i < C.N; i += BCH_CT) {
                uint32_t sum = c0;
#pragma unroll
                for (int jj = 0; jj < BCH_MAXDEG; ++jj)
                    if (jj < deg) {
                        sum ^= term[jj];
                        term[jj] = (uint32_t)s_mul[jj][0][term[jj] & 255u] ^ (uint32_t)s_mul[jj][1][term[jj] >> 8];
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
        if (s_state == 2 && s_count >= s_deg && s_count <= 32 && s_count > 0) {
            // Forney (:133-218), the evaluator uses `count` as locator degree: coefficient i by thread i, then one location per thread
            const int count = s_count;
            const int etmp = count < NR - 1 ? count : NR - 1;
            if (lane <= etmp) {
                uint16_t e = G.vmul(syn[lane], s_loc[0]);
                for (int jj = 1; jj <= lane; ++jj) e ^= G.vmul(syn[lane - jj], s_loc[jj]);
                s_ev[lane] = e;
            }
            __syncthreads();
            if (lane == 0) {
                int edeg = -1;
                for (int i = 0; i <= etmp; ++i)
                    if (s_ev[i]) edeg = i;
                s_edeg = edeg;
            }
            __syncthreads();
            if (lane < count) {
                const int edeg = s_edeg;
                uint16_t root = G.imul(s_pos[lane], 1), tmp = root;
                uint16_t eval = s_ev[0];
                for (int jj = 1; jj <= edeg; ++jj) {
                    eval ^= G.vmuli(s_ev[jj], tmp);
                    tmp = G.imul(tmp, root);
                }
                uint16_t mag = 0;
                if (eval) {
                    uint16_t deriv = s_loc[1];
                    uint16_t root2 = G.imul(root, root), tmp2 = root2;
                    for (int jj = 3; jj <= count; jj += 2) {
                        deriv ^= G.vmuli(s_loc[jj], tmp2);
                        tmp2 = G.imul(tmp2, root2);
                    }
                    mag = G.EXP[G.idiv(G.LOG[eval], G.LOG[deriv])];
                }
                s_mag[lane] = mag;
            }
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
                    // (magnitudes: computed by the threads above, see the Forney block before this section)
                    const uint16_t* magnitudes = s_mag;
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
                                uint16_t* syn, int32_t* todo, int32_t* corrections, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(todo, 0, 2 * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    int grid = nframes < 4096 ? nframes : 4096;
    hipLaunchKernelGGL(bch_syndromes_kernel, dim3(grid), dim3(256), 0, stream, C, frames, frame_stride, nbch, nframes, syn, todo, corrections);
    return hipGetLastError();
}
hipError_t bch_correct_launch(const BchDeviceCode& C, uint8_t* frames, int frame_stride, int nbch, int kbch, int nframes,
                              const uint16_t* syn, int32_t* todo, int32_t* corrections, hipStream_t stream) {
    int grid = nframes < 2048 ? nframes : 2048;      // (as many workgroups as the device holds at once, 25 KB of LDS each; the idle ones leave at their first look at the list)
    hipLaunchKernelGGL(bch_correct_kernel, dim3(grid), dim3(BCH_CT), 0, stream, C, frames, frame_stride, nbch, kbch, nframes, syn, todo, corrections);
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
