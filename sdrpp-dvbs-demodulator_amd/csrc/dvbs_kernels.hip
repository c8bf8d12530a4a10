// DVB-S inner-code kernels for gfx950: soft slicer, self-locking punctured K=7 r=1/2 Viterbi decoder, Forney de-interleaver.
//
// Replaces (bit-exact integer work):
//   DVBSymToSoftBlock::process (conversion part)   dvbs/dvbs_syms_to_soft.cpp:7-13,28-31
//   Viterbi_DVBS::work                             dvbs/viterbi_all.cpp:74-276 (lock search, depuncturing, BER watchdog)
//   CCDecoder::work  = update_viterbi_blk (generic ACS kernel, viterbi/volk_k7_r2_generic_fixed.h:95-163) + find_endstate
//                      (viterbi/cc_decoder.cpp:192-209) + chainback_viterbi (:228-276) + init_viterbi (:159-175)
//   CCEncoder::work                                viterbi/cc_encoder.cpp:92-104
//   DVBSInterleaving::deinterleave                 dvbs/dvbs_interleaving.h:58-70
//
// Viterbi mapping: ONE WAVE PER STREAM, LANE = TRELLIS STATE (64 states).  A step is: fetch the two predecessor
// metrics (ds_bpermute), add the branch metrics in uint8 wrap-around arithmetic exactly like the reference's
// unsigned char sums, compare, keep the survivor, subtract the wave-wide minimum (4 DPP row rotations + 4 scalar
// reads) -- and the 64-bit decision word of the step is simply __ballot(decision): bit n = state n, the reference's
// decision_t layout.  Decision words are kept one per lane for 64 steps and stored coalesced; chain-back is a scalar
// walk (v_readlane + SALU) over 64 steps at a time.  Blocks of a stream are chained through the start state returned by
// the chain-back (the next block starts from metric 0 at that state, 63 elsewhere), so blocks are sequential per stream
// and streams fill the GPU.  The whole Viterbi_DVBS state machine (IDLE lock search over 2 phases x 26 rate/shift
// hypotheses in the reference's order, SYNCED decoding, watchdog) runs inside the kernel: no host round trip per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace s2 {

__global__ __launch_bounds__(256) void dvbs_slice_kernel(const float* __restrict__ iq, int n, int8_t* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += gridDim.x * 256) {
        float x = iq[i] * 100;
        out[i] = x < -127.0f ? (int8_t)-127 : (x > 127.0f ? (int8_t)127 : (int8_t)x);
    }
}

// wave-wide minimum, result uniform: 4 DPP row rotations (all lanes of a row hold the row minimum), then 4 scalar reads
__device__ __forceinline__ int wave_min_bcast(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));   // row_ror:1
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));   // row_ror:8
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One CCDecoder::work (cc_decoder.cpp:304-314) by one wave.  src: 2*(frame+6) unsigned softs; dst: frame bits, one per byte;
// dec: frame+6 decision words of scratch; (ss, biased) = the decoder object's chained start state.
//
// ROTATING STATE LAYOUT.  At step t (r = t mod 6) lane L holds the metric of state rotl6(L, r).  The new state n = ((p << 1) | bit) & 63 has
// rotr6(n, 1) equal to p except in bit 5, so in the layout of step t + 1 the lane of n IS the lane of one predecessor (p0 = n >> 1 for even n,
// p1 = p0 + 32 for odd n) and the other predecessor sits in the lane that differs in bit bp = 5 - r: the two fetches of a step (ds_bpermute
// before, two LDS round trips in the step's chain) are ONE lane exchange by a constant -- DPP quad_perm / row_ror / row_shl+shr for 1, 2, 8, 4,
// v_permlane16/32_swap for 16, 32 -- and the two lanes of a butterfly face each other.  With own = X[lane], partner = X[lane ^ 1 << bp]:
//   A = own + metric, B = partner + (63 - metric)   (uint8 wrap-around, volk_k7_r2_generic_fixed.h:27-50: for even n (m0, m1) = (A, B), for odd n (m2, m3) = (B, A))
//   survivor = min(A, B) either way; decision = A >= B in the even lanes, B >= A in the odd ones (two compares, combined on the scalar unit).
// The decision word of a step is in the layout of step t + 1 (bit L = the state in lane L); the chain-back walks in the same rotating layout
// (the predecessor of lane L is L with bit bp replaced by the decision bit), so nothing is ever permuted back.  Chunks of 60 steps = 10
// rotations; the wave-wide minimum (renormalize, :52-68) is a DPP chain (4 row rotations, row_bcast:15, row_bcast:31, one v_readlane) whose
// wait states are filled with the decision store and the NEXT step's branch metric.  ~27 issue slots per step instead of ~50 and two LDS
// round trips.
__device__ __forceinline__ int rotl6(int x, int r) { return ((x << r) | (x >> (6 - r))) & 63; }
__device__ __forceinline__ int rotr6(int x, int r) { return ((x >> r) | (x << (6 - r))) & 63; }

#define VIT_PART5 "v_mov_b32 %[T], %[X]\n\tv_mov_b32 %[Xp], %[X]\n\ts_nop 1\n\tv_permlane32_swap_b32 %[T], %[Xp]\n\tv_cndmask_b32_e64 %[Xp], %[T], %[Xp], %[e5]\n\t"
#define VIT_PART4 "v_mov_b32 %[T], %[X]\n\tv_mov_b32 %[Xp], %[X]\n\ts_nop 1\n\tv_permlane16_swap_b32 %[T], %[Xp]\n\tv_cndmask_b32_e64 %[Xp], %[T], %[Xp], %[e4]\n\t"
#define VIT_PART3 "v_mov_b32_dpp %[Xp], %[X] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define VIT_PART2 "v_mov_b32_dpp %[Xp], %[X] row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_mov_b32_dpp %[Xp], %[X] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
#define VIT_PART1 "v_mov_b32_dpp %[Xp], %[X] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
#define VIT_PART0 "v_mov_b32_dpp %[Xp], %[X] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
// one step; PART: the lane exchange, E: the even-state lanes of the NEW layout (bit bp clear), BN: the branch table of the NEXT step (b0 | b1 << 8: the
// branch metric sum 1 + (b0 ^ y0) + (b1 ^ y1) is ONE v_sad_u8 against the two softs -- b is 0 or 255, so b ^ y = |b - y|).
// Wait states: a DPP operand must have been written at least two instructions earlier, a v_readlane source one (tools/ubench/readlane_probe.hip: read right
// behind its write it returns the OLD value) -- filled with the decision store and the next step's metric, s_nop where nothing is left.
#define VIT_STEP(PART, E, BN)                                                                                                            \
    PART                                                                                                                                 \
    "v_add_u32 %[A], %[X], %[m]\n\t"                                                                                                     \
    "v_xad_u32 %[B], %[m], 63, %[Xp]\n\t"                                                                                                \
    "v_cmp_ge_u32_sdwa vcc, %[A], %[B] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"                                                              \
    "v_cmp_le_u32_sdwa s[20:21], %[A], %[B] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"                                                         \
    "v_min_u32_sdwa %[Y], %[A], %[B] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"                            \
    "s_and_b64 s[22:23], vcc, " E "\n\t"                                                                                                 \
    "s_andn2_b64 s[20:21], s[20:21], " E "\n\t"                                                                                          \
    "v_min_u32_dpp %[M], %[Y], %[Y] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                                            \
    "s_or_b64 s[22:23], s[22:23], s[20:21]\n\t"                                                                                          \
    "v_writelane_b32 %[wlo], s22, m0\n\t"                                                                                                \
    "v_min_u32_dpp %[M], %[M], %[M] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"                                                            \
    "v_writelane_b32 %[whi], s23, m0\n\t"                                                                                                \
    "s_add_u32 m0, m0, 1\n\t"                                                                                                            \
    "v_min_u32_dpp %[M], %[M], %[M] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"                                                            \
    "v_readlane_b32 s24, %[cur], m0\n\t"                                                                                                 \
    "v_sad_u8 %[m], " BN ", s24, 1\n\t"                                                                                                  \
    "v_min_u32_dpp %[M], %[M], %[M] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                                                            \
    "v_lshrrev_b32 %[m], 3, %[m]\n\t"                                                                                                    \
    "s_nop 0\n\t"                                                                                                                        \
    "v_min_u32_dpp %[M], %[M], %[M] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                                                         \
    "s_nop 1\n\t"                                                                                                                        \
    "v_min_u32_dpp %[M], %[M], %[M] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                                                         \
    "s_nop 0\n\t"                                                                                                                        \
    "v_readlane_b32 s27, %[M], 63\n\t"                                                                                                   \
    "v_subrev_u32 %[X], s27, %[Y]\n\t"                                                                                                   \
    "s_nop 1\n\t"
#define VIT_S0 VIT_STEP(VIT_PART5, "%[e5]", "%[b1]")
#define VIT_S1 VIT_STEP(VIT_PART4, "%[e4]", "%[b2]")
#define VIT_S2 VIT_STEP(VIT_PART3, "%[e3]", "%[b3]")
#define VIT_S3 VIT_STEP(VIT_PART2, "%[e2]", "%[b4]")
#define VIT_S4 VIT_STEP(VIT_PART1, "%[e1]", "%[b5]")
#define VIT_S5 VIT_STEP(VIT_PART0, "%[e0]", "%[b0]")

__device__ void cc_decode_wave(const uint8_t* src, int frame, int& ss, int& biased, unsigned long long* dec, uint8_t* dst, int lane) {
    const int veclen = frame + 6;
    // branch table bits of the butterfly in this lane, per rotation: the step with old layout r puts state n = rotl6(lane, r + 1) here; butterfly n >> 1 (cc_decoder.cpp:113-125: polys 79, 109)
    int bt[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int bi = rotl6(lane, (r + 1) % 6) >> 1;
        bt[r] = ((__popc((2 * bi) & 79) & 1) ? 255 : 0) | ((__popc((2 * bi) & 109) & 1) ? 255 << 8 : 0);
    }
    int X = biased ? ((lane == (ss & 63)) ? 0 : 63) : 31;   // init_viterbi :159-175 / first block unbiased :177-190 (layout 0: lane = state)
    for (int s0 = 0; s0 < veclen; s0 += 60) {
        const int cnt = min(60, veclen - s0);
        int cur = 0;
        if (lane < cnt) cur = src[2 * (s0 + lane)] | (src[2 * (s0 + lane) + 1] << 8);
        int wlo = 0, whi = 0, tA, tB, tY, tM, tm, tXp, tT;
        uint32_t ng = (uint32_t)__builtin_amdgcn_readfirstlane(cnt / 6), rem = (uint32_t)__builtin_amdgcn_readfirstlane(cnt % 6);
        asm volatile(
            "s_mov_b32 s26, m0\n\t"                  // (m0 = the lane the step reads its softs from and writes its decision word to; the caller's value comes back at the end)
            "s_mov_b32 m0, 0\n\t"
            "v_readlane_b32 s24, %[cur], m0\n\t"
            "v_sad_u8 %[m], %[b0], s24, 1\n\t"
            "v_lshrrev_b32 %[m], 3, %[m]\n\t"
            "s_cmp_eq_u32 %[ng], 0\n\t"
            "s_cbranch_scc1 2f\n\t"
            "1:\n\t"
            VIT_S0 VIT_S1 VIT_S2 VIT_S3 VIT_S4 VIT_S5
            "s_sub_u32 %[ng], %[ng], 1\n\t"
            "s_cmp_lg_u32 %[ng], 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            "s_cmp_lt_u32 %[rem], 1\n\t"
            "s_cbranch_scc1 9f\n\t"
            VIT_S0
            "s_cmp_lt_u32 %[rem], 2\n\t"
            "s_cbranch_scc1 9f\n\t"
            VIT_S1
            "s_cmp_lt_u32 %[rem], 3\n\t"
            "s_cbranch_scc1 9f\n\t"
            VIT_S2
            "s_cmp_lt_u32 %[rem], 4\n\t"
            "s_cbranch_scc1 9f\n\t"
            VIT_S3
            "s_cmp_lt_u32 %[rem], 5\n\t"
            "s_cbranch_scc1 9f\n\t"
            VIT_S4
            "9:\n\t"
            "s_mov_b32 m0, s26\n\t"
            : [X] "+v"(X), [wlo] "+v"(wlo), [whi] "+v"(whi), [ng] "+s"(ng), [A] "=&v"(tA), [B] "=&v"(tB), [Y] "=&v"(tY), [M] "=&v"(tM), [m] "=&v"(tm),
              [Xp] "=&v"(tXp), [T] "=&v"(tT)
            : [cur] "v"(cur), [rem] "s"(rem), [b0] "v"(bt[0]), [b1] "v"(bt[1]), [b2] "v"(bt[2]), [b3] "v"(bt[3]), [b4] "v"(bt[4]), [b5] "v"(bt[5]),
              [e0] "s"(0x5555555555555555ull), [e1] "s"(0x3333333333333333ull), [e2] "s"(0x0F0F0F0F0F0F0F0Full),
              [e3] "s"(0x00FF00FF00FF00FFull), [e4] "s"(0x0000FFFF0000FFFFull), [e5] "s"(0x00000000FFFFFFFFull)
            : "s20", "s21", "s22", "s23", "s24", "s26", "s27", "vcc", "scc");
        if (lane < cnt) dec[s0 + lane] = (unsigned long long)(unsigned)wlo | ((unsigned long long)(unsigned)whi << 32);
    }
    // find_endstate: first minimal metric in STATE order (cc_decoder.cpp:192-209); the layout after veclen steps is veclen mod 6
    const int rend = veclen % 6;
    const int endstate = wave_min_bcast((X << 8) | rotl6(lane, rend)) & 63;
    __syncthreads();
    // chainback (cc_decoder.cpp:228-276), tailsize 6, ADDSHIFT 2: a uniform (scalar) walk in the rotating layout.  L = the lane of the state at time
    // t + 1; bit L of decision word t says which predecessor survived, and the predecessor's lane is L with bit 5 - t mod 6 replaced by that bit.
    int L = rotr6(endstate, rend), retL = 0;
    for (int cbase = ((veclen - 1) / 60) * 60; cbase >= 0; cbase -= 60) {
        const int tlo = max(cbase, 6), thi = min(cbase + 60, veclen);          // decision words [tlo, thi) = output bits [tlo - 6, thi - 6)
        if (thi <= tlo) continue;
        unsigned long long myw = 0;
        if (lane < 60 && cbase + lane < veclen) myw = dec[cbase + lane];
        const int lo = (int)(unsigned)myw, hi = (int)(unsigned)(myw >> 32);
        unsigned long long bits = 0;
        int t = thi - 1;
        auto one = [&](int k, int bp) {
            const unsigned long long w =
                (unsigned)__builtin_amdgcn_readlane(lo, k) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, k) << 32);
            const int kb = (int)((w >> L) & 1ull);
            L = (L & ~(1 << bp)) | (kb << bp);
            bits |= (unsigned long long)kb << k;
            if (cbase + k == frame) retL = L;
        };
        for (; t >= tlo && (t + 1) % 6 != 0; --t) one(t - cbase, 5 - (t % 6));       // the partial rotation at the top of the last chunk
        for (; t >= tlo; t -= 6) {                                                    // whole rotations: the bit positions are constants
            const int k = t - cbase;
#pragma unroll
            for (int q = 0; q < 6; ++q) one(k - q, q);                                // t - q has (t - q) mod 6 = 5 - q
        }
        if (lane >= tlo - cbase && lane < thi - cbase) dst[cbase + lane - 6] = (uint8_t)((bits >> lane) & 1ull);
    }
    ss = rotl6(retL, frame % 6);          // the state at time `frame` (retval >> 2 of the reference's walk)
    biased = 1;
    __syncthreads();
}

// in: per stream `nblocks` blocks, block b at in + s*stream_stride + b*block_stride, 2*(frame+6) bytes each are read
// out: bits, one per byte, frame_size per block;  state: per stream {start_state, biased} (biased = 0: fresh decoder)
__global__ __launch_bounds__(64) void dvbs_cc_decode_kernel(const uint8_t* in, long stream_stride, int block_stride, int nblocks, int frame_size,
                                                            uint8_t* out, long out_stream_stride, unsigned long long* dec_ws, int* state) {
    const int lane = threadIdx.x, s = blockIdx.x;
    int ss = state[2 * s], biased = state[2 * s + 1];
    unsigned long long* dec = dec_ws + (size_t)s * (frame_size + 6);
    for (int blk = 0; blk < nblocks; ++blk)
        cc_decode_wave(in + (long)s * stream_stride + (long)blk * block_stride, frame_size, ss, biased, dec,
                       out + (long)s * out_stream_stride + (long)blk * frame_size, lane);
    if (lane == 0) { state[2 * s] = ss; state[2 * s + 1] = biased; }
}

// ------------------------------------------------------------------------------------------------ Viterbi_DVBS::work
// rotate_soft (rotation.cpp:4-63, phases 0 / 90 only: module_dvbs_demod.cpp:23) + signed_soft_to_unsigned (utils.cpp:11-20)
__device__ __forceinline__ int soft_rotate(const int8_t* in, int i, int phase) {   // rotate_soft, PHASE_0 / PHASE_90, no IQ swap
    int s;
    if (phase == 0) { s = in[i]; if (s == -128) s = -127; }
    else {
        int a = in[i & ~1], b = in[i | 1];
        if (a == -128) a = -127;
        if (b == -128) b = -127;
        s = (i & 1) ? -a : b;
    }
    return s;
}
__device__ __forceinline__ uint8_t soft_conv(const int8_t* in, int i, int phase) {
    const int u = (soft_rotate(in, i, phase) + 127) & 255;      // signed_soft_to_unsigned: 128 is reserved for erasures
    return (uint8_t)(u == 128 ? 127 : u);
}
__device__ __forceinline__ int depunc_e(int period, int p) { return period == 3 ? (p == 1) : (p == 1 || p == 3 || p == 4 || p == 5); }
// outputs emitted for inputs [0, i) of the 2/3 (period 3) or 5/6 (period 6) pattern started at phase p0 (depunc.h:8-190)
__device__ __forceinline__ int depunc_before(int period, int p0, int i) {
    const int full = i / period, rem = i - full * period;
    int ex = full * (period == 3 ? 1 : 4);
    for (int j = 0; j < rem; ++j) ex += depunc_e(period, (p0 + j) % period);
    return i + ex;
}
// lane-parallel Depunc23/Depunc56 body: returns the number of bytes written, `lead` bytes were placed before by the caller
__device__ int depunc_pattern(int period, int p0, int lead, const uint8_t* in, int size, uint8_t* out, int lane) {
    for (int i = lane; i < size; i += 64) {
        const int p = (p0 + i) % period;
        const int pos = lead + depunc_before(period, p0, i);
        const uint8_t x = in[i];
        if (period == 6 && p == 4) { out[pos] = 128; out[pos + 1] = x; }
        else if (depunc_e(period, p)) { out[pos] = x; out[pos + 1] = 128; }
        else out[pos] = x;
    }
    return lead + depunc_before(period, p0, size);
}
// Depunc23::depunc_cont / Depunc56::depunc_cont (depunc.h:46-80,139-185): one call over `size` inputs with the carried state
// (is_first, changing_shift, got_extra, buf); returns the even number of bytes the decoder may use
__device__ __forceinline__ int depunc_cont_wave(int period, int& first, int& shift, int& extra, int& buf, const uint8_t* in, int size,
                                                uint8_t* out, int lane) {
    int lead = 0;
    if (first || extra) { if (lane == 0) out[0] = (uint8_t)buf; lead = 1; first = 0; extra = 0; }
    const int p0 = shift % period;
    int oo = depunc_pattern(period, p0, lead, in, size, out, lane);
    shift = p0 + size;
    __syncthreads();
    if (oo & 1) { buf = out[oo - 1]; oo -= 1; extra = 1; }
    return oo;
}
// viterbi_all.h:92-113
__device__ int depunc34_wave(const uint8_t* in, uint8_t* out, int size, int shift, int lane) {
    const int np = size / 2;
    for (int i = lane; i < np; i += 64) {
        const int n2 = shift ? (i >> 1) : ((i + 1) >> 1);
        const int pos = 2 * n2 + 4 * (i - n2);
        const uint8_t a = in[2 * i], b = in[2 * i + 1];
        if ((shift != 0) ^ ((i & 1) == 0)) { out[pos] = a; out[pos + 1] = b; }
        else { out[pos] = 128; out[pos + 1] = a; out[pos + 2] = b; out[pos + 3] = 128; }
    }
    const int n2 = shift ? (np >> 1) : ((np + 1) >> 1);
    return 2 * n2 + 4 * (np - n2);
}
// viterbi_all.h:115-150
__device__ int depunc78_wave(const uint8_t* in, uint8_t* out, int size, int shift, int lane) {
    const int np = size / 2;
    const int c0 = (shift + 3) >> 2;
    for (int i = lane; i < np; i += 64) {
        const int n2 = ((i + shift + 3) >> 2) - c0;
        const int pos = 2 * n2 + 4 * (i - n2);
        const int m = (i + shift) & 3;
        const uint8_t a = in[2 * i], b = in[2 * i + 1];
        if (m == 0) { out[pos] = a; out[pos + 1] = b; }
        else if (m == 1) { out[pos] = 128; out[pos + 1] = a; out[pos + 2] = 128; out[pos + 3] = b; }
        else { out[pos] = 128; out[pos + 1] = a; out[pos + 2] = b; out[pos + 3] = 128; }
    }
    const int n2 = ((np + shift + 3) >> 2) - c0;
    return 2 * n2 + 4 * (np - n2);
}
// CCEncoder::work (cc_encoder.cpp:92-104) over `frame` bits into ber_enc, then get_ber (viterbi_all.cpp:59-72) of raw[0,len) against it
__device__ float reencode_ber(const uint8_t* bits, int frame, int& enc_state, uint8_t* ber_enc, const uint8_t* raw, int len, float ratio, int lane) {
    int last = enc_state;
    for (int i = lane; i < frame; i += 64) {
        unsigned st = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int q = i - j;
            const unsigned b = q >= 0 ? (bits[q] & 1u) : (((unsigned)enc_state >> (j - i - 1)) & 1u);
            st |= b << j;
        }
        ber_enc[2 * i] = (uint8_t)(__popc(st & 79) & 1);
        ber_enc[2 * i + 1] = (uint8_t)(__popc(st & 109) & 1);
        if (i == frame - 1) last = (int)st;
    }
    enc_state = __shfl(last, (frame - 1) & 63);
    __syncthreads();
    int err = 0, tot = 0;
    for (int i = lane; i < len; i += 64) {
        const int r = raw[i];
        if (r != 128) { err += ((r > 127) != (ber_enc[i] != 0)) ? 1 : 0; tot++; }
    }
    const float errors = (float)wave_sum(err), total = (float)wave_sum(tot);
    __syncthreads();
    return (errors / total) * ratio;
}

// stage entry for the parity tests (dvbs2gpu_dvbs_depuncture): the device functions above on caller-supplied bytes, one wave.
//   mode 0  Depunc23/56::depunc_static (lock search): state4[1] = shift
//   mode 1  depunc_cont with the carried state4 = {is_first, changing_shift, got_extra, buf}
//   mode 2  rotate_soft (state4[0] = phase 0 / 1 = 0 / 90 degrees), signed bytes out
__global__ __launch_bounds__(64) void dvbs_depunc_stage_kernel(int period, int mode, const uint8_t* in, int size, uint8_t* out, int* state4, int* n_out) {
    const int lane = threadIdx.x;
    if (mode == 2) {
        for (int i = lane; i < size; i += 64) out[i] = (uint8_t)(int8_t)soft_rotate((const int8_t*)in, i, state4[0]);
        if (lane == 0) *n_out = size;
        return;
    }
    int n;
    if (mode == 0) {
        const int shift = state4[1], lead = shift > period - 1;
        if (lead && lane == 0) out[0] = 128;
        n = depunc_pattern(period, shift % period, lead, in, size, out, lane);
    } else {
        int first = state4[0], shift = state4[1], extra = state4[2], buf = state4[3];
        n = depunc_cont_wave(period, first, shift, extra, buf, in, size, out, lane);
        __syncthreads();
        if (lane == 0) { state4[0] = first; state4[1] = shift; state4[2] = extra; state4[3] = buf; }
    }
    if (lane == 0) *n_out = n;
}
hipError_t dvbs_depunc_stage_launch(int period, int mode, const uint8_t* d_in, int size, uint8_t* d_out, int* d_state4, int* d_n, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_depunc_stage_kernel, dim3(1), dim3(64), 0, st, period, mode, d_in, size, d_out, d_state4, d_n);
    return hipGetLastError();
}

// in_ptrs / nblk (both optional): per-stream input base and block count (the demodulator's soft FIFOs); otherwise stream s reads
// in_all + s*nblocks*8192 and runs `nblocks` blocks.  Outputs are always laid out [stream][nblocks][...].
__global__ __launch_bounds__(64) void dvbs_viterbi_kernel(const int8_t* in_all, const int8_t* const* in_ptrs, const int* nblk, int nblocks,
                                                          uint8_t* out_all, int* out_n, DvbsVitStats* stats,
                                                          DvbsVitState* states, uint8_t* ws, float thr, int max_outsync, const int* blk0) {
    const int lane = threadIdx.x, s = blockIdx.x;
    DvbsVitState* sp = states + s;
    uint8_t* w = ws + (size_t)s * DVBS_VIT_WS_BYTES;
    uint8_t* ber_soft = w;                                    // 2048, directly followed by ber_depunc (viterbi_all.h:74-78)
    uint8_t* ber_depunc = w + 2048;                           // 8192 + 64
    uint8_t* ber_enc = w + DVBS_VIT_WS_BER_ENC;               // 8192
    uint8_t* ber_dec = w + DVBS_VIT_WS_BER_DEC;               // 2048
    uint8_t* soft = w + DVBS_VIT_WS_SOFT;                     // 8192 + 64
    uint8_t* depunc = w + DVBS_VIT_WS_DEPUNC;                 // 4 * 8192
    unsigned long long* dec = (unsigned long long*)(w + DVBS_VIT_WS_DEC);
    int state = sp->state, rate = sp->rate, d_phase = sp->phase, d_shift = sp->shift, invalid = sp->invalid;
    float ber = sp->ber;
    int ss[10], bs[10], enc[5], dfirst[2], dshift[2], dextra[2], dbuf[2];
    for (int i = 0; i < 10; ++i) { ss[i] = sp->dec_ss[i]; bs[i] = sp->dec_biased[i]; }
    for (int i = 0; i < 5; ++i) enc[i] = sp->enc_state[i];
    for (int i = 0; i < 2; ++i) { dfirst[i] = sp->dep_first[i]; dshift[i] = sp->dep_shift[i]; dextra[i] = sp->dep_extra[i]; dbuf[i] = sp->dep_buf[i]; }
    const int TEST = 2048, BUF = 8192;
    const int8_t* in_base = in_ptrs ? in_ptrs[s] : in_all + (size_t)s * nblocks * BUF;
    const int my_blocks = nblk ? min(nblk[s], nblocks) : nblocks;
    for (int blk = blk0 ? blk0[s] : 0; blk < my_blocks; ++blk) {       // (blk0: blocks an earlier time slice of the call has decoded)
        const int8_t* in = in_base + (size_t)blk * BUF;
        uint8_t* out = out_all + ((size_t)s * nblocks + blk) * BUF;
        if (state == 0) {                                     // ST_IDLE: viterbi_all.cpp:76-204
            ber = 10;
            bool locked = false;
            for (int phase = 0; phase < 2; ++phase) {
                for (int i = lane; i < TEST; i += 64) ber_soft[i] = soft_conv(in, i, phase);
                __syncthreads();
                for (int shift = 0; shift < 2; ++shift) {
                    cc_decode_wave(ber_soft + shift, TEST / 2, ss[0], bs[0], dec, ber_dec, lane);
                    const float b = reencode_ber(ber_dec, TEST / 2, enc[0], ber_enc, ber_soft + shift, TEST, 2.5f, lane);
                    if (b < thr) { ber = b; locked = true; d_phase = phase; d_shift = shift; rate = 0; }
                }
                for (int shift = 0; shift < 6; ++shift) {
                    const int lead = shift > 2;
                    if (lead && lane == 0) ber_depunc[0] = 128;
                    depunc_pattern(3, shift % 3, lead, ber_soft, TEST, ber_depunc, lane);
                    __syncthreads();
                    cc_decode_wave(ber_depunc, 1366, ss[1], bs[1], dec, ber_dec, lane);
                    const float b = reencode_ber(ber_dec, 1366, enc[1], ber_enc, ber_depunc, 2560, 3.5f, lane);
                    if (b < thr) { ber = b; locked = true; d_phase = phase; d_shift = shift; rate = 1; dshift[0] = shift; dfirst[0] = shift > 2; }
                }
                for (int shift = 0; shift < 2; ++shift) {
                    depunc34_wave(ber_soft, ber_depunc, TEST, shift, lane);
                    __syncthreads();
                    cc_decode_wave(ber_depunc, 1536, ss[2], bs[2], dec, ber_dec, lane);
                    const float b = reencode_ber(ber_dec, 1536, enc[2], ber_enc, ber_depunc, 3072, 5.f, lane);
                    if (b < thr) { ber = b; locked = true; d_phase = phase; d_shift = shift; rate = 2; }
                }
                for (int shift = 0; shift < 12; ++shift) {
                    const int lead = shift > 5;
                    if (lead && lane == 0) ber_depunc[0] = 128;
                    depunc_pattern(6, shift % 6, lead, ber_soft, TEST, ber_depunc, lane);
                    __syncthreads();
                    cc_decode_wave(ber_depunc, 1699, ss[3], bs[3], dec, ber_dec, lane);
                    const float b = reencode_ber(ber_dec, 1699, enc[3], ber_enc, ber_depunc, 3399, 8.f, lane);
                    if (b < thr) { ber = b; locked = true; d_phase = phase; d_shift = shift; rate = 3; dshift[1] = shift; dfirst[1] = shift > 5; }
                }
                for (int shift = 0; shift < 4; ++shift) {
                    depunc78_wave(ber_soft, ber_depunc, TEST, shift, lane);
                    __syncthreads();
                    cc_decode_wave(ber_depunc, 1792, ss[4], bs[4], dec, ber_dec, lane);
                    const float b = reencode_ber(ber_dec, 1792, enc[4], ber_enc, ber_depunc, 3584, 10.f, lane);
                    if (b < thr) { ber = b; locked = true; d_phase = phase; d_shift = shift; rate = 4; }
                }
            }
            if (locked) {                                     // :100-103 etc.: state, counters, both work buffers to erasures
                state = 1; invalid = 0;
                for (int i = lane; i < BUF + 64; i += 64) soft[i] = 128;
                for (int i = lane; i < 4 * BUF; i += 64) depunc[i] = 128;
                __syncthreads();
            }
        }
        int n_out = 0;
        if (state == 1) {                                     // ST_SYNCED: viterbi_all.cpp:206-273
            for (int i = lane; i < BUF; i += 64) soft[i] = soft_conv(in, i, d_phase);
            __syncthreads();
            if (rate == 0) {
                cc_decode_wave(soft + d_shift, BUF / 2, ss[5], bs[5], dec, out, lane);
                n_out = BUF / 2;
                ber = reencode_ber(out, TEST / 2, enc[0], ber_enc, soft + d_shift, TEST, 2.5f, lane);
            } else if (rate == 1 || rate == 3) {
                const int d = rate == 1 ? 0 : 1, period = rate == 1 ? 3 : 6;
                const int oo = depunc_cont_wave(period, dfirst[d], dshift[d], dextra[d], dbuf[d], soft, BUF, depunc, lane);
                if (rate == 1) {
                    cc_decode_wave(depunc, 5462, ss[6], bs[6], dec, out, lane);
                    ber = reencode_ber(out, 1366, enc[1], ber_enc, depunc, 2560, 3.5f, lane);
                } else {
                    cc_decode_wave(depunc, 6799, ss[8], bs[8], dec, out, lane);
                    ber = reencode_ber(out, 1699, enc[3], ber_enc, depunc, 3399, 8.f, lane);
                    // Rate 5/6: the decoder's frame is (int)(8192 * 1.66 / 2) = 6799 bits but the block advances the output by oo / 2 = 6826
                    // or 6827 (viterbi_all.cpp:246-249, cc_decoder.cpp:304-314): the reference never writes the 27-28 bits in between, they
                    // keep what its caller's buffer held.  Here they are ZERO -- what the oracle shows with a cleared buffer -- instead of
                    // whatever the block's slot of the workspace held from an earlier call (found by the bank test with carriers of differing
                    // block counts: the slot of a (stream, block) pair moves when the batch's block count changes).
                    for (int i = 6799 + lane; i < oo / 2; i += 64) out[i] = 0;
                }
                n_out = oo / 2;
            } else if (rate == 2) {
                const int sz = depunc34_wave(soft, depunc, BUF, d_shift, lane);
                __syncthreads();
                cc_decode_wave(depunc, 6144, ss[7], bs[7], dec, out, lane);
                n_out = sz / 2;
                ber = reencode_ber(out, 1536, enc[2], ber_enc, depunc, 3072, 5.f, lane);
            } else {
                const int sz = depunc78_wave(soft, depunc, BUF, d_shift, lane);
                __syncthreads();
                cc_decode_wave(depunc, 7168, ss[9], bs[9], dec, out, lane);
                n_out = sz / 2;
                ber = reencode_ber(out, 1792, enc[4], ber_enc, depunc, 3584, 10.f, lane);
            }
            if (ber > thr) { invalid++; if (invalid > max_outsync) state = 0; }
            else invalid = 0;
        }
        if (lane == 0) {
            out_n[(size_t)s * nblocks + blk] = n_out;
            if (stats) {
                DvbsVitStats& st = stats[(size_t)s * nblocks + blk];
                st.ber = ber; st.state = state; st.rate = rate; st.phase = d_phase; st.shift = d_shift;
            }
        }
    }
    if (lane == 0) {
        sp->state = state; sp->rate = rate; sp->phase = d_phase; sp->shift = d_shift; sp->invalid = invalid; sp->ber = ber;
        for (int i = 0; i < 10; ++i) { sp->dec_ss[i] = ss[i]; sp->dec_biased[i] = bs[i]; }
        for (int i = 0; i < 5; ++i) sp->enc_state[i] = enc[i];
        for (int i = 0; i < 2; ++i) { sp->dep_first[i] = dfirst[i]; sp->dep_shift[i] = dshift[i]; sp->dep_extra[i] = dextra[i]; sp->dep_buf[i] = dbuf[i]; }
    }
}

// ------------------------------------------------------------------------------------------------ Forney de-interleaver
// out[n] = in[n - 204*(11 - n%12)] over the continuous byte stream; hist = last 2244 bytes of the previous calls
__global__ __launch_bounds__(256) void dvbs_deinterleave_kernel(const uint8_t* __restrict__ in, long stream_stride, int nbytes_all,
                                                                const int* __restrict__ nframes, uint8_t* __restrict__ out,
                                                                const uint8_t* __restrict__ hist) {
    const int s = blockIdx.y;
    const int nbytes = nframes ? nframes[s] * 1632 : nbytes_all;      // per-stream length (tail pipeline) or uniform
    const uint8_t* __restrict__ src = in + (long)s * stream_stride;
    uint8_t* __restrict__ dst = out + (long)s * stream_stride;
    const uint8_t* __restrict__ h = hist + (long)s * DVBS_FORNEY_HIST;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < nbytes; n += gridDim.x * 256) {
        const int p = n - 204 * (11 - n % 12);
        dst[n] = p >= 0 ? src[p] : h[DVBS_FORNEY_HIST + p];
    }
}
// new history = last 2244 bytes of [hist ++ in]
__global__ __launch_bounds__(256) void dvbs_deinterleave_hist_kernel(const uint8_t* __restrict__ in, long stream_stride, int nbytes_all,
                                                                     const int* __restrict__ nframes, uint8_t* hist) {
    __shared__ uint8_t tmp[DVBS_FORNEY_HIST];
    const int s = blockIdx.x;
    const int nbytes = nframes ? nframes[s] * 1632 : nbytes_all;
    const uint8_t* __restrict__ src = in + (long)s * stream_stride;
    uint8_t* h = hist + (long)s * DVBS_FORNEY_HIST;
    for (int i = threadIdx.x; i < DVBS_FORNEY_HIST; i += 256) {
        const int p = nbytes - DVBS_FORNEY_HIST + i;
        tmp[i] = p >= 0 ? src[p] : h[DVBS_FORNEY_HIST + p];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DVBS_FORNEY_HIST; i += 256) h[i] = tmp[i];
}

// DVBSVitBlock::process output layout (dvbs_vit.cpp:6-12): the first nbits[b] bytes of every block, concatenated per stream
__global__ __launch_bounds__(256) void dvbs_pack_bits_kernel(const uint8_t* __restrict__ bits, const int* __restrict__ nbits, const int* __restrict__ nblk,
                                                             int nblocks, uint8_t* const* __restrict__ out_ptrs, int cap, int* __restrict__ out_count) {
    const int s = blockIdx.x;
    uint8_t* __restrict__ dst = out_ptrs[s];
    const int mb = min(nblk[s], nblocks);
    int off = 0;
    for (int b = 0; b < mb; ++b) {
        const int nb = nbits[(size_t)s * nblocks + b];
        if (off + nb > cap) break;
        const uint8_t* __restrict__ src = bits + ((size_t)s * nblocks + b) * 8192;
        for (int i = threadIdx.x; i < nb; i += 256) dst[off + i] = src[i];
        off += nb;
    }
    if (threadIdx.x == 0) out_count[s] = off;
}
hipError_t dvbs_pack_bits_launch(const uint8_t* d_bits, const int* d_nbits, const int* d_nblk, int nstreams, int nblocks, uint8_t* const* d_out_ptrs,
                                 int cap, int* d_out_count, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_pack_bits_kernel, dim3(nstreams), dim3(256), 0, st, d_bits, d_nbits, d_nblk, nblocks, d_out_ptrs, cap, d_out_count);
    return hipGetLastError();
}

hipError_t dvbs_slice_launch(const float* d_iq, int n, int8_t* d_out, hipStream_t st) {
    int grid = (2 * n + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(dvbs_slice_kernel, dim3(grid), dim3(256), 0, st, d_iq, n, d_out);
    return hipGetLastError();
}
hipError_t dvbs_cc_decode_launch(const uint8_t* d_in, long stream_stride, int block_stride, int nstreams, int nblocks, int frame_size,
                                 uint8_t* d_out, long out_stream_stride, unsigned long long* d_dec_ws, int* d_state, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_cc_decode_kernel, dim3(nstreams), dim3(64), 0, st, d_in, stream_stride, block_stride, nblocks, frame_size, d_out,
                       out_stream_stride, d_dec_ws, d_state);
    return hipGetLastError();
}
hipError_t dvbs_viterbi_launch(const int8_t* d_soft, const int8_t* const* d_soft_ptrs, const int* d_nblk, int nstreams, int nblocks,
                               uint8_t* d_bits, int* d_nbits, DvbsVitStats* d_stats, DvbsVitState* d_states, uint8_t* d_ws, float thr,
                               int max_outsync, hipStream_t st, const int* d_blk0) {
    hipLaunchKernelGGL(dvbs_viterbi_kernel, dim3(nstreams), dim3(64), 0, st, d_soft, d_soft_ptrs, d_nblk, nblocks, d_bits, d_nbits, d_stats,
                       d_states, d_ws, thr, max_outsync, d_blk0);
    return hipGetLastError();
}
hipError_t dvbs_deinterleave_launch(const uint8_t* d_in, long stream_stride, int nstreams, int nbytes, uint8_t* d_out, uint8_t* d_hist,
                                    hipStream_t st) {
    int gx = (nbytes + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(dvbs_deinterleave_kernel, dim3(gx, nstreams), dim3(256), 0, st, d_in, stream_stride, nbytes, (const int*)nullptr, d_out, d_hist);
    hipLaunchKernelGGL(dvbs_deinterleave_hist_kernel, dim3(nstreams), dim3(256), 0, st, d_in, stream_stride, nbytes, (const int*)nullptr, d_hist);
    return hipGetLastError();
}

// ================================================================================================ DVB-S tail (after the Viterbi)
// DVBS_TS_Deframer::work (dvbs/dvbs_ts_deframer.cpp:37-92): a 13 056-bit window slides over the bit stream one bit at a time; a
// frame (8 x 204 bytes) is emitted whenever the 8 sync bytes at 1632-bit spacing match B8 47 47 .. (or the inverted pattern)
// with at most 8 bit errors.  The reference shifts the whole window per bit; here every window position is tested in
// parallel on a byte-at-every-bit-offset view of the stream, hits are compacted in order, frames are gathered at stride 8.
//   state per stream: the last 13 055 bits (unpacked), errors_nor / errors_inv of the last hit
constexpr int TSD_WIN = 1632 * 8;

// v[p] = bits p .. p+7 of [history(13055) ++ input(n)] packed MSB first, for p = 0 .. n + 13055 - 8
__global__ __launch_bounds__(256) void dvbs_tsdef_pack_kernel(const uint8_t* const* __restrict__ in_ptrs, const int* __restrict__ counts,
                                                              const uint8_t* __restrict__ hist, uint8_t* __restrict__ v, long v_stride) {
    const int s = blockIdx.y, n = counts[s];
    const uint8_t* __restrict__ in = in_ptrs[s];
    const uint8_t* __restrict__ h = hist + (long)s * TSD_WIN;
    uint8_t* __restrict__ vo = v + (long)s * v_stride;
    const int total = n + (TSD_WIN - 1) - 7;          // number of byte positions
    for (int p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
        unsigned b = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = p + k;
            const unsigned bit = q < TSD_WIN - 1 ? h[q] : in[q - (TSD_WIN - 1)];
            b = (b << 1) | (bit & 1u);
        }
        vo[p] = (uint8_t)b;
    }
}
// one workgroup per stream: test every window end position t (window start = t in the v[] indexing), ordered compaction of hits
__global__ __launch_bounds__(256) void dvbs_tsdef_find_kernel(const int* __restrict__ counts, const uint8_t* __restrict__ v, long v_stride,
                                                              int max_frames, int* __restrict__ hit_pos, int* __restrict__ nframes,
                                                              int* __restrict__ errs) {
    __shared__ int s_base, s_wave[4], s_last[2];
    const int s = blockIdx.x, n = counts[s], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* __restrict__ vv = v + (long)s * v_stride;
    if (tid == 0) { s_base = 0; s_last[0] = errs[2 * s]; s_last[1] = errs[2 * s + 1]; }
    __syncthreads();
    for (int t0 = 0; t0 < n; t0 += 256) {
        const int t = t0 + tid;          // after pushing input bit t the window starts at stream position t (v index t)
        int en = 99, ei = 99;
        if (t < n) {
            en = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) en += __popc((unsigned)(vv[t + 1632 * i] ^ (i == 0 ? 0xB8u : 0x47u)));
            ei = 64 - en;
        }
        const bool hit = en <= 8 || ei <= 8;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (hit) {
            const int idx = off + __popcll(m & ((1ull << lane) - 1ull));
            if (idx < max_frames) hit_pos[(long)s * max_frames + idx] = en <= 8 ? t : (t | 0x40000000);
        }
        // errors_nor / errors_inv follow the LAST hit (dvbs_ts_deframer.cpp:71-72,82-83)
        const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (tot > 0) {
            // highest hit lane of the highest hitting wave
            int last_wave = s_wave[3] ? 3 : (s_wave[2] ? 2 : (s_wave[1] ? 1 : 0));
            if (wave == last_wave && hit && (m >> lane) == 1ull) { s_last[0] = en <= 8 ? en : 0; s_last[1] = en <= 8 ? 0 : ei; }
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) { nframes[s] = min(s_base, max_frames); errs[2 * s] = s_last[0]; errs[2 * s + 1] = s_last[1]; }
}
// frames out (byte k of a frame = v[start + 8k], inverted for an inverted hit) + new history; grid (max_frames + 1, nstreams)
__global__ __launch_bounds__(256) void dvbs_tsdef_emit_kernel(const uint8_t* const* __restrict__ in_ptrs, const int* __restrict__ counts,
                                                              const uint8_t* __restrict__ v, long v_stride, int max_frames,
                                                              const int* __restrict__ hit_pos, const int* __restrict__ nframes,
                                                              uint8_t* __restrict__ frames, long frames_stride, const uint8_t* __restrict__ hist_old,
                                                              uint8_t* __restrict__ hist_next) {
    const int s = blockIdx.y, f = blockIdx.x;
    if (f == max_frames) {                       // the extra block of each stream writes the new history: last 13055 bits
        const int n = counts[s];
        const uint8_t* __restrict__ in = in_ptrs[s];
        uint8_t* __restrict__ hn = hist_next + (long)s * TSD_WIN;
        for (int q = threadIdx.x; q < TSD_WIN - 1; q += 256) {
            // new history = last 13055 bits of [old history ++ input]
            const int p = n + q;                 // position in [old history ++ input]
            const int hq = p - (TSD_WIN - 1);    // input index, or < 0: still old history
            hn[q] = (uint8_t)((hq >= 0 ? in[hq] : hist_old[(long)s * TSD_WIN + p]) & 1u);
        }
        return;
    }
    if (f >= nframes[s]) return;
    const int hp = hit_pos[(long)s * max_frames + f];
    const int start = hp & 0x3fffffff;
    const unsigned inv = (hp & 0x40000000) ? 0xffu : 0u;
    const uint8_t* __restrict__ vv = v + (long)s * v_stride + start;
    uint8_t* __restrict__ o = frames + (long)s * frames_stride + (long)f * 1632;
    for (int k = threadIdx.x; k < 1632; k += 256) o[k] = (uint8_t)(vv[8 * k] ^ inv);
}

// ---- RS(204,188) = RS(255,239) shortened, GF(256) poly 0x11d, generator roots alpha^0..alpha^15.  ONE WAVE PER PACKET.
// correct_reed_solomon_decode (common/correct/reed-solomon/decode.c:299-380): syndromes, Berlekamp-Massey exactly as written there
// (:31-121: its result for more than 8 errors depends on the variant), Chien search, Forney.  Field arithmetic through
// log/exp tables (field.h); lanes hold 4 coefficients each.
struct Gf256Dev { const uint8_t* ex; const uint8_t* lg; };
__device__ __forceinline__ unsigned gf_mul(const uint8_t* ex, const uint8_t* lg, unsigned a, unsigned b) { return (a == 0 || b == 0) ? 0u : ex[lg[a] + lg[b]]; }
__device__ __forceinline__ unsigned gf_div(const uint8_t* ex, const uint8_t* lg, unsigned a, unsigned b) { return (a == 0 || b == 0) ? 0u : ex[255 + lg[a] - lg[b]]; }

// status per packet: 1 = message produced (possibly corrected), 0 = decoder gave up (output comes from the previous packet)
__global__ __launch_bounds__(64) void dvbs_rs_kernel(uint8_t* __restrict__ deint, long stream_stride, const int* __restrict__ nframes,
                                                     int max_packets, uint8_t* __restrict__ status, int* __restrict__ rs_err,
                                                     const uint8_t* __restrict__ gf_tab) {
    __shared__ uint8_t ex[512], lg[256];
    __shared__ uint8_t recv[256];
    __shared__ uint8_t syn[16];
    __shared__ uint8_t loc[64], last[64], ev[16], der[16], roots[16];
    __shared__ int s_order, s_nroots;
    const int lane = threadIdx.x, s = blockIdx.y, p = blockIdx.x;
    if (p >= nframes[s] * 8) return;
    for (int i = lane; i < 512; i += 64) ex[i] = gf_tab[i];
    for (int i = lane; i < 256; i += 64) lg[i] = gf_tab[512 + i];
    uint8_t* __restrict__ pkt = deint + (long)s * stream_stride + (long)p * 204;
    // received polynomial: coeff[i] = encoded[254 - i], encoded = 51 zeros ++ 204 bytes  ->  coeff[i] = pkt[203 - i] for i < 204, else 0
    for (int i = lane; i < 256; i += 64) recv[i] = i < 204 ? pkt[203 - i] : 0;
    __syncthreads();
    // syndromes S_r = sum_i coeff[i] * alpha^(r i): each lane its 4 coefficients (i = lane + 64 q; only i < 204 non-zero)
    unsigned myS[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) myS[r] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        const unsigned c = recv[i];
        if (c) {
            const unsigned lc = lg[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) myS[r] ^= ex[(lc + (unsigned)(r * i) % 255u) % 255u];
        }
    }
    unsigned anynz = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned x = myS[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o);
        if (lane == 0) syn[r] = (uint8_t)x;
        anynz |= x;
    }
    if (anynz == 0) { if (lane == 0) { status[(long)s * max_packets + p] = 1; rs_err[(long)s * max_packets + p] = 0; } return; }   // clean packet
    __syncthreads();
    if (lane == 0) {
        // Berlekamp-Massey, reed_solomon_find_error_locator with num_erasures = 0
        for (int i = 0; i < 64; ++i) { loc[i] = 0; last[i] = 0; }
        loc[0] = 1; last[0] = 1;
        unsigned loc_order = 0, last_order = 0, numerrors = 0, delay = 1, last_disc = 1;
        for (unsigned i = 0; i < 16; ++i) {
            unsigned disc = syn[i];
            for (unsigned j = 1; j <= numerrors; ++j) disc ^= gf_mul(ex, lg, loc[j], syn[i - j]);
            if (!disc) { delay++; continue; }
            if (2 * numerrors <= i) {
                for (int j = (int)last_order; j >= 0; --j) last[j + delay] = (uint8_t)gf_div(ex, lg, gf_mul(ex, lg, last[j], disc), last_disc);
                for (int j = (int)delay - 1; j >= 0; --j) last[j] = 0;
                for (unsigned j = 0; j <= last_order + delay; ++j) { uint8_t t = loc[j]; loc[j] ^= last[j]; last[j] = t; }
                const unsigned t = loc_order;
                loc_order = last_order + delay;
                last_order = t;
                numerrors = i + 1 - numerrors;
                last_disc = disc;
                delay = 1;
                continue;
            }
            for (int j = (int)last_order; j >= 0; --j) loc[j + delay] ^= (uint8_t)gf_div(ex, lg, gf_mul(ex, lg, last[j], disc), last_disc);
            loc_order = (last_order + delay > loc_order) ? last_order + delay : loc_order;
            delay++;
        }
        s_order = (int)loc_order;
        s_nroots = 0;
    }
    __syncthreads();
    const int order = s_order;
    // Chien search: Lambda(e) for every field element e = 1..255 (0 is never a root: Lambda(0) = 1), roots in increasing e
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        unsigned val = 1;                       // e == 0 -> coeff[0] = 1 (not a root); computed properly for e >= 1
        if (e >= 1) {
            val = 0;
            const unsigned le = lg[e] % 255u;
            for (int i = 0; i <= order && i < 64; ++i) {
                const unsigned c = loc[i];
                if (c) val ^= ex[(lg[c] % 255u + (le * (unsigned)i) % 255u) % 255u];
            }
        }
        const bool isroot = (e >= 1) && val == 0;
        const unsigned long long m = __ballot(isroot);
        const int base = s_nroots;
        if (isroot) {
            const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
            if (idx < 16) roots[idx] = (uint8_t)e;
        }
        __syncthreads();
        if (lane == 0) s_nroots = base + __popcll(m);
        __syncthreads();
    }
    if (s_nroots != order || order > 16) { if (lane == 0) status[(long)s * max_packets + p] = 0; return; }
    if (lane == 0) {
        // error evaluator = Lambda * S mod x^16 (polynomial.c:17-30), formal derivative (:74-87)
        for (int i = 0; i < 16; ++i) { ev[i] = 0; der[i] = 0; }
        for (int i = 0; i <= order && i <= 15; ++i)
            for (int j = 0; j <= 15 - i; ++j) ev[i + j] ^= (uint8_t)gf_mul(ex, lg, loc[i], syn[j]);
        for (int i = 0; i <= order - 1 && i < 16; ++i) der[i] = ((i + 1) & 1) ? loc[i + 1] : 0;
    }
    __syncthreads();
    int fixed = 0;
    if (lane < order) {
        const unsigned root = roots[lane];
        const unsigned lr = lg[root] % 255u;
        unsigned num = 0, den = 0;
        for (int i = 0; i < 16; ++i) {
            if (ev[i]) num ^= ex[(lg[ev[i]] % 255u + (lr * (unsigned)i) % 255u) % 255u];
            if (i <= order - 1 && der[i]) den ^= ex[(lg[der[i]] % 255u + (lr * (unsigned)i) % 255u) % 255u];
        }
        const unsigned rinv = ex[(255u - lr) % 255u];                        // field_pow(root, -1)
        const unsigned val = gf_mul(ex, lg, rinv, gf_div(ex, lg, num, den));
        const unsigned locn = lg[rinv] % 255u;                              // log(1/root), 0 for 1/root == 1
        if (locn < 204) pkt[203 - locn] ^= (uint8_t)val;                    // (locations >= 204 hit the zero padding: not part of the packet)
        fixed = (locn >= 16 && locn < 204 && val != 0) ? 1 : 0;             // the wrapper counts changed MESSAGE bytes (dvbs_reedsolomon.h:38-42)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fixed += __shfl_xor(fixed, o);
    if (lane == 0) { status[(long)s * max_packets + p] = 1; rs_err[(long)s * max_packets + p] = fixed; }
}

// ---- per stream, in packet order: the wrapper's stale-output rule for failed packets (dvbs_reedsolomon.h:26-47), energy
// dispersal removal (dvbs_scrambling.h:28-42) and the 188-byte TS packets (module_dvbs_demod.cpp:93-99).  ONE WAVE PER STREAM.
// DvbsTailState (kernels.h): prbs_pos = PRBS bytes consumed since the last reset, -1: never reset (register 0: all-zero sequence);
// last_msg = message part of the RS wrapper's obuffer (bytes 51..238)
__global__ __launch_bounds__(64) void dvbs_ts_finish_kernel(uint8_t* __restrict__ deint, long stream_stride, const int* __restrict__ nframes,
                                                            int max_packets, const uint8_t* __restrict__ status, const uint8_t* __restrict__ prbs_tab,
                                                            DvbsTailState* __restrict__ state, uint8_t* const* __restrict__ out_ptrs, int cap,
                                                            int* __restrict__ out_bytes, int* __restrict__ rs_err) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const int npk = nframes[s] * 8;
    DvbsTailState* st = state + s;
    int pos = st->prbs_pos;
    unsigned lastm[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) lastm[q] = (lane + 64 * q) < 188 ? st->last_msg[lane + 64 * q] : 0u;
    uint8_t* __restrict__ out = out_ptrs[s];
    int nout = 0;
    for (int p = 0; p < npk; ++p) {
        uint8_t* __restrict__ pkt = deint + (long)s * stream_stride + (long)p * 204;
        const int ok = status[(long)s * max_packets + p];
        unsigned d[3];
        int diff = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int i = lane + 64 * q;
            unsigned rx = i < 188 ? pkt[i] : 0u;
            if (!ok) { diff += (i < 188 && rx != lastm[q]) ? 1 : 0; rx = lastm[q]; }     // decoder gave up: previous message comes out
            d[q] = rx;
        }
        if (ok) {
#pragma unroll
            for (int q = 0; q < 3; ++q) lastm[q] = d[q];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) diff += __shfl_xor(diff, o);
        if (lane == 0 && !ok) rs_err[(long)s * max_packets + p] = diff;
        // energy dispersal: first byte B8 resets the generator, otherwise one PRBS byte is skipped
        const unsigned first = __shfl((int)d[0], 0);
        int base;                               // PRBS byte index of packet byte 1
        if (first == 0xB8u) { base = 0; pos = 187; }
        else if (pos >= 0) { base = pos + 1; pos += 188; }
        else base = -1;
        if (pos >= 32767 * 8) pos -= 32767 * 8;  // (sequence of PRBS bytes repeats every 32767 bytes)
        if (nout + 188 <= cap) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = lane + 64 * q;
                if (i < 188) {
                    unsigned x = d[q];
                    if (i == 0) x = 0x47;
                    else if (base >= 0) x ^= prbs_tab[(base + i - 1) % 32767];
                    out[nout + i] = (uint8_t)x;
                }
            }
            nout += 188;
        }
    }
    if (lane == 0) { st->prbs_pos = pos; out_bytes[s] = nout; }
#pragma unroll
    for (int q = 0; q < 3; ++q) if ((lane + 64 * q) < 188) st->last_msg[lane + 64 * q] = (uint8_t)lastm[q];
}

// stage entry for the parity tests: RS + stale-output rule + energy dispersal on packets the caller placed in d_deint (the deframer
// and the Forney de-interleaver are bypassed); skip_rs: every packet counts as decoded (d_status preset to 1 by the caller)
hipError_t dvbs_tail_rs_finish_launch(int nstreams, int max_frames, const int* d_nframes, uint8_t* d_deint, long frames_stride, uint8_t* d_status,
                                      const uint8_t* d_gf, const uint8_t* d_prbs, DvbsTailState* d_state, uint8_t* const* d_out_ptrs, int cap,
                                      int* d_out_bytes, int* d_rs_err, int skip_rs, hipStream_t st) {
    if (!skip_rs)
        hipLaunchKernelGGL(dvbs_rs_kernel, dim3(max_frames * 8, nstreams), dim3(64), 0, st, d_deint, frames_stride, d_nframes, max_frames * 8, d_status, d_rs_err, d_gf);
    hipLaunchKernelGGL(dvbs_ts_finish_kernel, dim3(nstreams), dim3(64), 0, st, d_deint, frames_stride, d_nframes, max_frames * 8, d_status, d_prbs,
                       d_state, d_out_ptrs, cap, d_out_bytes, d_rs_err);
    return hipGetLastError();
}

hipError_t dvbs_tail_launch(const uint8_t* const* d_in_ptrs, const int* d_counts, int nstreams, int max_bits, uint8_t* d_hist, uint8_t* d_hist_next,
                            uint8_t* d_v, long v_stride, int max_frames, int* d_hit_pos, int* d_nframes, int* d_errs, uint8_t* d_frames,
                            uint8_t* d_deint, long frames_stride, uint8_t* d_forney_hist, uint8_t* d_status, const uint8_t* d_gf, const uint8_t* d_prbs,
                            DvbsTailState* d_state, uint8_t* const* d_out_ptrs, int cap, int* d_out_bytes, int* d_rs_err, hipStream_t st) {
    int gx = (max_bits + TSD_WIN + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(dvbs_tsdef_pack_kernel, dim3(gx, nstreams), dim3(256), 0, st, d_in_ptrs, d_counts, d_hist, d_v, v_stride);
    hipLaunchKernelGGL(dvbs_tsdef_find_kernel, dim3(nstreams), dim3(256), 0, st, d_counts, d_v, v_stride, max_frames, d_hit_pos, d_nframes, d_errs);
    hipLaunchKernelGGL(dvbs_tsdef_emit_kernel, dim3(max_frames + 1, nstreams), dim3(256), 0, st, d_in_ptrs, d_counts, d_v, v_stride, max_frames,
                       d_hit_pos, d_nframes, d_frames, frames_stride, d_hist, d_hist_next);
    int gd = (max_frames * 1632 + 255) / 256;
    gd = gd < 1 ? 1 : (gd > 64 ? 64 : gd);
    hipLaunchKernelGGL(dvbs_deinterleave_kernel, dim3(gd, nstreams), dim3(256), 0, st, d_frames, frames_stride, 0, d_nframes, d_deint, d_forney_hist);
    hipLaunchKernelGGL(dvbs_deinterleave_hist_kernel, dim3(nstreams), dim3(256), 0, st, d_frames, frames_stride, 0, d_nframes, d_forney_hist);
    hipLaunchKernelGGL(dvbs_rs_kernel, dim3(max_frames * 8, nstreams), dim3(64), 0, st, d_deint, frames_stride, d_nframes, max_frames * 8, d_status, d_rs_err, d_gf);
    hipLaunchKernelGGL(dvbs_ts_finish_kernel, dim3(nstreams), dim3(64), 0, st, d_deint, frames_stride, d_nframes, max_frames * 8, d_status, d_prbs,
                       d_state, d_out_ptrs, cap, d_out_bytes, d_rs_err);
    return hipGetLastError();
}

}  // namespace s2
