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
__device__ void cc_decode_wave(const uint8_t* src, int frame, int& ss, int& biased, unsigned long long* dec, uint8_t* dst, int lane) {
    const int veclen = frame + 6;
    // branch table bits for butterfly i = lane>>1 (cc_decoder.cpp:113-125): polys 79, 109
    const int bi = lane >> 1;
    const int b0 = (__popc((2 * bi) & 79) & 1) ? 255 : 0, b1 = (__popc((2 * bi) & 109) & 1) ? 255 : 0;
    const int oddmask = (lane & 1) ? 63 : 0;
    int X = biased ? ((lane == (ss & 63)) ? 0 : 63) : 31;   // init_viterbi :159-175 / first block unbiased :177-190
    int sy = 0;
    if (lane < veclen) sy = src[2 * lane] | (src[2 * lane + 1] << 8);
    for (int s0 = 0; s0 < veclen; s0 += 64) {
        const int m = min(64, veclen - s0);
        const int cur = sy;
        if (s0 + 64 + lane < veclen) sy = src[2 * (s0 + 64 + lane)] | (src[2 * (s0 + 64 + lane) + 1] << 8);   // next chunk in flight
        unsigned long long myw = 0;
        for (int k = 0; k < m; ++k) {
            const int y = __builtin_amdgcn_readlane(cur, k);
            const int y0 = y & 255, y1 = y >> 8;
            const int metric = ((1 + (b0 ^ y0) + (b1 ^ y1)) >> 1) >> 2;   // BFLY: unsigned short sum, >>1, >>2 (volk_k7_r2_generic_fixed.h:27-50)
            const int xa = __shfl(X, bi), xb = __shfl(X, bi + 32);        // predecessors i and i + 32
            // even state 2i: m0 = X[i]+metric, m1 = X[i+32]+(63-metric); odd state 2i+1: m2 = X[i]+(63-metric), m3 = X[i+32]+metric
            const int ma = metric ^ oddmask;                              // 63 - m == m ^ 63 for m in 0..63
            const int c0 = (xa + ma) & 0xff, c1 = (xb + (ma ^ 63)) & 0xff;   // unsigned char wrap
            const bool decision = c0 >= c1;                               // (signed int)(m0 - m1) >= 0
            const unsigned long long w = __ballot(decision);              // bit n = state n: the reference's decision_t layout
            if (lane == k) myw = w;
            const int Y = decision ? c1 : c0;
            X = Y - wave_min_bcast(Y);                                    // renormalize (:52-68)
        }
        if (lane < m) dec[s0 + lane] = myw;
    }
    // find_endstate: first minimal metric (cc_decoder.cpp:192-209)
    const int endstate = wave_min_bcast((X << 8) | lane) & 63;
    __syncthreads();
    // chainback (cc_decoder.cpp:228-276), tailsize 6, ADDSHIFT 2: uniform (scalar) walk over 64 steps held one per lane
    unsigned es = (unsigned)endstate << 2;
    int retval = 0;
    for (int base = ((frame - 1) >> 6) << 6; base >= 0; base -= 64) {
        const int cnt = min(64, frame - base);
        unsigned long long myw = 0;
        if (lane < cnt) myw = dec[6 + base + lane];
        const int lo = (int)(unsigned)myw, hi = (int)(unsigned)(myw >> 32);
        unsigned long long bits = 0;
        for (int k = cnt - 1; k >= 0; --k) {
            const unsigned long long w =
                (unsigned)__builtin_amdgcn_readlane(lo, k) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, k) << 32);
            const unsigned kb = (unsigned)((w >> (es >> 2)) & 1ull);
            es = (es >> 1) | (kb << 7);
            bits |= (unsigned long long)kb << k;
            if (base + k == frame - 6) retval = (int)es;
        }
        if (lane < cnt) dst[base + lane] = (uint8_t)((bits >> lane) & 1ull);
    }
    ss = retval >> 2;
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
__device__ __forceinline__ uint8_t soft_conv(const int8_t* in, int i, int phase) {
    int s;
    if (phase == 0) { s = in[i]; if (s == -128) s = -127; }
    else {
        int a = in[i & ~1], b = in[i | 1];
        if (a == -128) a = -127;
        if (b == -128) b = -127;
        s = (i & 1) ? -a : b;
    }
    int u = (s + 127) & 255;
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

// in_ptrs / nblk (both optional): per-stream input base and block count (the demodulator's soft FIFOs); otherwise stream s reads
// in_all + s*nblocks*8192 and runs `nblocks` blocks.  Outputs are always laid out [stream][nblocks][...].
__global__ __launch_bounds__(64) void dvbs_viterbi_kernel(const int8_t* in_all, const int8_t* const* in_ptrs, const int* nblk, int nblocks,
                                                          uint8_t* out_all, int* out_n, DvbsVitStats* stats,
                                                          DvbsVitState* states, uint8_t* ws, float thr, int max_outsync) {
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
    for (int blk = 0; blk < my_blocks; ++blk) {
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
                int lead = 0;
                if (dfirst[d] || dextra[d]) { if (lane == 0) depunc[0] = (uint8_t)dbuf[d]; lead = 1; dfirst[d] = 0; dextra[d] = 0; }
                const int p0 = dshift[d] % period;
                int oo = depunc_pattern(period, p0, lead, soft, BUF, depunc, lane);
                dshift[d] = p0 + BUF;
                __syncthreads();
                if (oo & 1) { dbuf[d] = depunc[oo - 1]; oo -= 1; dextra[d] = 1; }
                if (rate == 1) {
                    cc_decode_wave(depunc, 5462, ss[6], bs[6], dec, out, lane);
                    ber = reencode_ber(out, 1366, enc[1], ber_enc, depunc, 2560, 3.5f, lane);
                } else {
                    cc_decode_wave(depunc, 6799, ss[8], bs[8], dec, out, lane);
                    ber = reencode_ber(out, 1699, enc[3], ber_enc, depunc, 3399, 8.f, lane);
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
__global__ __launch_bounds__(256) void dvbs_deinterleave_kernel(const uint8_t* __restrict__ in, long stream_stride, int nbytes,
                                                                uint8_t* __restrict__ out, const uint8_t* __restrict__ hist) {
    const int s = blockIdx.y;
    const uint8_t* __restrict__ src = in + (long)s * stream_stride;
    uint8_t* __restrict__ dst = out + (long)s * stream_stride;
    const uint8_t* __restrict__ h = hist + (long)s * DVBS_FORNEY_HIST;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < nbytes; n += gridDim.x * 256) {
        const int p = n - 204 * (11 - n % 12);
        dst[n] = p >= 0 ? src[p] : h[DVBS_FORNEY_HIST + p];
    }
}
// new history = last 2244 bytes of [hist ++ in]
__global__ __launch_bounds__(256) void dvbs_deinterleave_hist_kernel(const uint8_t* __restrict__ in, long stream_stride, int nbytes, uint8_t* hist) {
    __shared__ uint8_t tmp[DVBS_FORNEY_HIST];
    const int s = blockIdx.x;
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
                               int max_outsync, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_viterbi_kernel, dim3(nstreams), dim3(64), 0, st, d_soft, d_soft_ptrs, d_nblk, nblocks, d_bits, d_nbits, d_stats,
                       d_states, d_ws, thr, max_outsync);
    return hipGetLastError();
}
hipError_t dvbs_deinterleave_launch(const uint8_t* d_in, long stream_stride, int nstreams, int nbytes, uint8_t* d_out, uint8_t* d_hist,
                                    hipStream_t st) {
    int gx = (nbytes + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(dvbs_deinterleave_kernel, dim3(gx, nstreams), dim3(256), 0, st, d_in, stream_stride, nbytes, d_out, d_hist);
    hipLaunchKernelGGL(dvbs_deinterleave_hist_kernel, dim3(nstreams), dim3(256), 0, st, d_in, stream_stride, nbytes, d_hist);
    return hipGetLastError();
}

}  // namespace s2
