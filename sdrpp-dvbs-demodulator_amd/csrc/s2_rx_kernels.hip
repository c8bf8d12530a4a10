// DVB-S2 receive-chain kernels for gfx950 (compiled with -ffp-contract=off: every fp32 operation is a
// separately rounded IEEE op in the order written, which is what the CPU restatement does; only the
// libm calls -- cosf/sinf/atan2f/expf/logf -- may differ from the host's by ULPs).
//
// Replaces (reference file:line):
//   loop::FastAGC<complex_t>::process (SDR++)          call site module_dvbs2_demod.cpp:220
//   FreqShift::process                                  common/dsp/demod/freq_shift.cpp:4-17
//   clock_recovery::Gardner::process                    common/dsp/demod/gardner.cpp:89-152
//   filter::FIR<complex_t,float>::process (SDR++) + /2  module_dvbs2_demod.cpp:226,231-239
//   S2PLSyncBlock::internal_process correlation         dvbs2/dvbs2_pl_sync.cpp:102-143,167-193
//   dvbs2_pilot_coarse_fed + NCO feedback               dvbs2/dvbs2_fed.h:7-48, module_dvbs2_demod.cpp:319-331
//   S2PLLBlock::process                                 dvbs2/dvbs2_pll.cpp:34-86
//   S2PLHDRDemod::process                               dvbs2/dvbs2_plhdr_demod.cpp:33-79
//   S2BBToSoft::process + S2Deinterleaver::deinterleave dvbs2/dvbs2_bb_to_soft.cpp:7-33, codings/s2_deinterleaver.cpp:72-136
//
// Parallel decomposition: the AGC / NCO / Gardner recurrences and the PLL are serial per stream (each
// sample's gain/phase depends on the previous output), so those run ONE LANE PER STREAM and the GPU is filled
// by the number of transponders in the batch; RRC+decimation, the PL-header correlator and the demapper are
// data-parallel over symbols.  All of these are small next to the LDPC stage (~1.3 kflop and ~24 B per symbol).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s2_rx.h"
#include "s2_params.h"

namespace s2 {

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return cf32{a.re * b.re - a.im * b.im, a.im * b.re + a.re * b.im}; }
__device__ __forceinline__ cf32 cconj(cf32 a) { return cf32{a.re, -a.im}; }
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return cf32{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return cf32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf32 cscale(cf32 a, float s) { return cf32{a.re * s, a.im * s}; }
__device__ __forceinline__ float camp(cf32 a) { return sqrtf(a.re * a.re + a.im * a.im); }
__device__ __forceinline__ float cphase(cf32 a) { return atan2f(a.im, a.re); }
__device__ __forceinline__ cf32 phasor(float x) { return cf32{cosf(x), sinf(x)}; }

struct PclDev {
    float alpha, beta, phase, freq, minFreq, maxFreq;
    __device__ __forceinline__ void advance(float err) {
        freq += beta * err;
        if (freq > maxFreq) freq = maxFreq; else if (freq < minFreq) freq = minFreq;
        phase += freq + alpha * err;
    }
    __device__ __forceinline__ void wrap_pi() {   // CLAMP_PHASE with [-pi, pi]
        const float PI_F = 3.14159265358979323846f;
        const float delta = PI_F - (-PI_F);
        while (phase > PI_F) phase -= delta;
        while (phase < -PI_F) phase += delta;
    }
};

__device__ __forceinline__ cf32 dot8(const cf32* x, const float* t) {
    cf32 acc{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc.re += x[k].re * t[k]; acc.im += x[k].im * t[k]; }
    return acc;
}

// ------------------------------------------------------------------------------------------------ front end
// one lane = one stream.  fe_out holds [Gardner output: up to n + n/16 + 64][scratch: 7 history + n NCO samples];
// the host allocates 2n + n/16 + 256 complex values (fe_capacity in s2_demod.hip).
__global__ __launch_bounds__(64) void s2_frontend_kernel(const S2StreamWork* __restrict__ work, int nstreams, S2LoopCoefs co,
                                                         const float* __restrict__ bank_g) {
    __shared__ float bank[GARDNER_PHASES * GARDNER_TAPS];
    for (int i = threadIdx.x; i < GARDNER_PHASES * GARDNER_TAPS; i += 64) bank[i] = bank_g[i];
    __syncthreads();
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nstreams) return;
    const S2StreamWork w = work[s];
    S2StreamState* st = w.st;
    const int n = w.count;
    cf32* __restrict__ scratch = w.fe_out + (n + n / 16 + 64);   // [7 history + n]
    // ---- AGC + NCO
    float gain = st->agc_gain, nph = st->nco_phase;
    const float nfr = st->nco_freq;
    for (int i = 0; i < GARDNER_TAPS - 1; ++i) scratch[i] = st->g_hist[i];
    for (int i = 0; i < n; ++i) {
        cf32 x = w.in[i];
        cf32 y = cscale(x, gain);
        float a = camp(y);
        gain += (1.0f - a) * co.agc_rate;
        if (gain > 10e6f) gain = 10e6f;
        cf32 z = cmul(y, phasor(-nph));
        nph += nfr;
        while ((double)nph > 6.283185307179586) nph = (float)((double)nph - 6.283185307179586);
        while ((double)nph < -6.283185307179586) nph = (float)((double)nph + 6.283185307179586);
        scratch[GARDNER_TAPS - 1 + i] = z;
    }
    st->agc_gain = gain; st->nco_phase = nph;
    // ---- Gardner timing recovery (omega = 1 sample per output, TED on every 2nd output)
    PclDev pcl{co.g_alpha, co.g_beta, st->g_phase, st->g_freq, co.g_min_freq, co.g_max_freq};
    int offset = st->g_offset, spsctr = st->g_spsctr, outCount = 0;
    cf32* __restrict__ out = w.fe_out;
    while (offset < n) {
        int phase = (int)floorf(pcl.phase * 128.0f);
        phase = phase < 0 ? 0 : (phase > 127 ? 127 : phase);
        cf32 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = scratch[offset + k];
        cf32 outVal = dot8(x, &bank[phase * 8]);
        out[outCount++] = outVal;
        float error;
        if (spsctr == 0) {
            cf32 dfdt;
            if (phase == 0) {
                dfdt = csub(dot8(x, &bank[(phase + 1) * 8]), outVal);
            } else if (phase == 127) {
                dfdt = csub(outVal, dot8(x, &bank[(phase - 1) * 8]));
            } else {
                cf32 a = dot8(x, &bank[(phase + 1) * 8]), b = dot8(x, &bank[(phase - 1) * 8]);
                dfdt = cscale(csub(a, b), 0.5f);
            }
            error = -(((outVal.re > 0 ? 1.0f : -1.0f) * dfdt.re) + ((outVal.im > 0 ? 1.0f : -1.0f) * dfdt.im));
        } else {
            error = 0.f;
        }
        spsctr++;
        if (spsctr >= 2) spsctr = 0;
        if (error > 1.0f) error = 1.0f;
        if (error < -1.0f) error = -1.0f;
        pcl.advance(error);
        float delta = floorf(pcl.phase);
        offset = (int)((float)offset + delta);
        pcl.phase -= delta;
    }
    offset -= n;
    for (int i = 0; i < GARDNER_TAPS - 1; ++i) st->g_hist[i] = scratch[n + i];
    st->g_phase = pcl.phase; st->g_freq = pcl.freq; st->g_offset = offset; st->g_spsctr = spsctr;
    st->n_fe_out = outCount;
}

// ------------------------------------------------------------------------------------------------ RRC + /2
// grid (x: symbol tiles, y: stream).  Only the samples the decimator keeps are filtered.
__global__ __launch_bounds__(256) void s2_rrc_decim_kernel(const S2StreamWork* __restrict__ work, const float* __restrict__ taps_g, int ntaps) {
    __shared__ float taps[RRC_MAX_TAPS];
    for (int i = threadIdx.x; i < ntaps; i += 256) taps[i] = taps_g[i];
    __syncthreads();
    const S2StreamWork w = work[blockIdx.y];
    const S2StreamState* st = w.st;
    const int n = st->n_fe_out;
    const int first = st->cr_samp ? 0 : 1;            // first kept index (module_dvbs2_demod.cpp:231-239)
    const int nsym = n > first ? (n - first + 1) / 2 : 0;
    const int H = ntaps - 1;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < nsym; m += gridDim.x * 256) {
        const int i = 2 * m + first;
        cf32 acc{0.f, 0.f};
        for (int k = 0; k < ntaps; ++k) {
            int p = i + k;                            // index into [history(H) ++ fe_out]
            cf32 v = p < H ? st->rrc_hist[p] : w.fe_out[p - H];
            acc.re += v.re * taps[k];
            acc.im += v.im * taps[k];
        }
        w.fifo[w.fifo_fill + m] = acc;
    }
}
// state update after all symbols of the call are out: delay line, decimator phase, symbol count
__global__ __launch_bounds__(128) void s2_rrc_state_kernel(const S2StreamWork* __restrict__ work, int ntaps) {
    const S2StreamWork w = work[blockIdx.x];
    S2StreamState* st = w.st;
    const int n = st->n_fe_out, H = ntaps - 1;
    __shared__ cf32 nh[RRC_MAX_TAPS];
    for (int i = threadIdx.x; i < H; i += 128) {
        int p = n + i;
        nh[i] = p < H ? st->rrc_hist[p] : w.fe_out[p - H];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 128) st->rrc_hist[i] = nh[i];
    if (threadIdx.x == 0) {
        const int first = st->cr_samp ? 0 : 1;
        st->n_sym = n > first ? (n - first + 1) / 2 : 0;
        st->cr_samp = (st->cr_samp ^ (n & 1)) & 1;
    }
}

// ------------------------------------------------------------------------------------------------ PL sync
// One workgroup per candidate window: brute-force differential SOF+PLSC correlation at every offset
// (dvbs2_pl_sync.cpp:111-143), arg-max with strict '>' = lowest offset wins ties.
__global__ __launch_bounds__(256) void s2_plsync_kernel(const cf32* const* __restrict__ wins, int raw, int* __restrict__ best_pos,
                                                        float* __restrict__ best_match) {
    __shared__ cf32 d[256 + 96];
    __shared__ float r_val[256];
    __shared__ int r_idx[256];
    const cf32* __restrict__ s = wins[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t dsof = 0x18d2e82u ^ (0x18d2e82u >> 1);
    const unsigned long long SCR = 0x719d83c953422dfaull;
    const unsigned long long dscr = SCR ^ (SCR >> 1);
    const int noff = raw - 90;
    float bestv = 0.f;
    int besti = 0;
    for (int base = 0; base < noff; base += 256) {
        __syncthreads();
        // differential products d[k] = conj(s[base+k-1]) * s[base+k] for k = 1 .. 256+89
        for (int k = tid; k < 256 + 90; k += 256) {
            int a = base + k;
            cf32 v{0.f, 0.f};
            if (k >= 1 && a < raw) v = cmul(cconj(s[a - 1]), s[a]);
            d[k] = v;
        }
        __syncthreads();
        int ss = base + tid;
        if (ss < noff) {
            const cf32* dd = &d[tid];        // dd[i] = diffs[i] of this offset; diffs[0] is defined as 0
            cf32 csof{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 26; ++i) {
                cf32 v = (i == 0) ? cf32{0.f, 0.f} : dd[i];
                if (((dsof >> (25 - i)) ^ i) & 1) csof = cadd(csof, v);
                else csof = csub(csof, v);
            }
            cf32 cpl{0.f, 0.f};
#pragma unroll
            for (int i = 1; i < 64; i += 2) {
                if ((dscr >> (63 - i)) & 1) cpl = csub(cpl, dd[26 + i]);
                else cpl = cadd(cpl, dd[26 + i]);
            }
            cf32 c0 = cadd(csof, cpl), c1 = csub(csof, cpl);
            cf32 c = camp(c0) > camp(c1) ? c0 : c1;
            cf32 dv = cscale(c, 1.0f / (26 - 1 + 64 / 2));
            float diff = camp(dv);
            if (diff > bestv && dv.im > 0) { bestv = diff; besti = ss; }   // per-thread scan is in ascending ss
        }
    }
    r_val[tid] = bestv; r_idx[tid] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            float v2 = r_val[tid + o]; int i2 = r_idx[tid + o];
            float v1 = r_val[tid]; int i1 = r_idx[tid];
            if (v2 > v1 || (v2 == v1 && v2 > 0.f && i2 < i1)) { r_val[tid] = v2; r_idx[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) { best_pos[blockIdx.x] = r_val[0] > 0.f ? r_idx[0] : 0; best_match[blockIdx.x] = r_val[0]; }
}

// ------------------------------------------------------------------------------------------------ frame loops
__device__ __forceinline__ cf32 pl_descramble(cf32 p, int r) {
    switch (r) {
        case 3: return cf32{-p.im, p.re};
        case 2: return cf32{-p.re, -p.im};
        case 1: return cf32{p.im, -p.re};
        default: return p;
    }
}
__device__ __forceinline__ int lut_index(float v) {   // constellation.cpp:295-301: double math, truncation, clamp
    int x = (int)(((double)v / 1.5) * 256 + 128);
    return x < 0 ? 0 : (x > 255 ? 255 : x);
}
__device__ __forceinline__ int pilot_start(int b) { return 90 + (b + 1) * 1440 + b * 36; }

// constellation_t::demod_soft_calc (constellation.cpp:205-261) -- used directly for 32APSK
__device__ void soft_calc_dev(const S2ConstelDev& C, cf32 sample, int8_t* bits_out, float* phase_err) {
    float tmp[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) tmp[i] = 0.f;
    if (C.amp != 1) sample = cscale(sample, C.amp);
    if (C.prescale != 1) sample = cscale(sample, C.prescale);
    float min_dist = 3.402823466e+38f;
    cf32 closest{0.f, 0.f};
    for (int i = 0; i < C.states; i++) {
        float dist = camp(csub(sample, C.pts[i]));
        if (dist < min_dist) { min_dist = dist; closest = C.pts[i]; }
        // exp/log through double: within ~0.5 ULP of the exact float result and subnormal-safe, so the int8 LLRs agree
        // with the host libm version except on rare rounding ties (device expf also flushes subnormal results)
        float dd = (float)exp((double)(-dist / 1.0f));
        for (int j = 0; j < C.bits; j++) {
            if (((i >> j) & 1) == 0) tmp[2 * j + 0] += dd;
            else tmp[2 * j + 1] += dd;
        }
    }
    if (bits_out)
        for (int i = 0; i < C.bits; i++) {
            float x = ((float)log((double)tmp[2 * i + 1]) - (float)log((double)tmp[2 * i + 0])) * C.sca;
            while (x < -127 || x > 127) {
                x *= 0.5f;
                if (!isfinite(x)) break;
            }
            bits_out[C.bits - 1 - i] = (int8_t)x;
        }
    if (phase_err) *phase_err = cphase(cmul(sample, cconj(closest)));
}

__global__ __launch_bounds__(64) void s2_frame_loops_kernel(const S2StreamWork* __restrict__ work, int nstreams,
                                                            const S2FrameRef* __restrict__ frames, const int* __restrict__ first,
                                                            S2LoopCoefs co, S2PlTablesDev T, S2ConstelDev C, int pls_code, int slots,
                                                            int pilots, int pilot_blocks, int plframe, cf32* __restrict__ pllout,
                                                            S2FrameStats* __restrict__ stats) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nstreams) return;
    S2StreamState* st = work[s].st;
    PclDev pll{co.pll_alpha, co.pll_beta, st->pll_phase, st->pll_freq, co.pll_min_freq, co.pll_max_freq};
    PclDev hdr{co.hdr_alpha, co.hdr_beta, st->hdr_phase, st->hdr_freq, co.hdr_min_freq, co.hdr_max_freq};
    float nco_freq = st->nco_freq;
    const float PI_F = 3.14159265358979323846f;
    const cf32* __restrict__ plsc = T.plsc + (size_t)pls_code * 64;
    for (int f = first[s]; f < first[s + 1]; ++f) {
        const cf32* __restrict__ fr = frames[f].sym;
        cf32* __restrict__ out = pllout + (size_t)f * plframe;
        // ---- coarse frequency error detector (dvbs2_fed.h) and NCO feedback (module_dvbs2_demod.cpp:319-331)
        float err = 0.f, symcnt = 90 - 2;
        for (int i = 0; i < 88; ++i) {
            // term order of the reference: i = 0..23 SOF, 24, 25 (SOF/PLSC boundary), 26..87 PLSC
            cf32 r2 = (i + 2) < 26 ? T.sof[i + 2] : plsc[i + 2 - 26];
            cf32 r0 = i < 26 ? T.sof[i] : plsc[i - 26];
            err += cmul(cmul(cmul(fr[i + 2], cconj(r2)), cconj(fr[i])), r0).im;
        }
        if (pilots) {
            const cf32 p{0.707f, 0.707f};
            for (int b = 0; b < pilot_blocks; ++b) {
                int start = pilot_start(b);
                cf32 d1{0.f, 0.f}, d2{0.f, 0.f};
                for (int i = 0; i < 36; ++i) {
                    cf32 descr = pl_descramble(fr[start + i], T.rn[start - 90 + i]);
                    if (i >= 2) err += cmul(cmul(cmul(descr, cconj(p)), cconj(d2)), p).im;
                    d2 = d1; d1 = descr;
                }
                symcnt += 36 - 2;
            }
        }
        float est = err / symcnt;
        if (fabsf(est) < 0.02) nco_freq = nco_freq + est * (co.fll_bw / 100.0f);
        else nco_freq = nco_freq + est * co.fll_bw;
        if (nco_freq > 0.3f * PI_F) nco_freq = 0.3f * PI_F;
        if (nco_freq < -0.3f * PI_F) nco_freq = -0.3f * PI_F;
        // ---- PLL (dvbs2_pll.cpp:34-86)
        int next_pilot = (pilots && pilot_blocks > 0) ? pilot_start(0) : -1, pb = 0;
        for (int i = 0; i < plframe; ++i) {
            cf32 tmp_val = cmul(fr[i], phasor(-pll.phase));
            float error = 0.f;
            if (i >= 90) {
                cf32 descr = pl_descramble(tmp_val, T.rn[i - 90]);
                bool is_pilot = next_pilot >= 0 && i >= next_pilot && i < next_pilot + 36;
                if (!is_pilot) {
                    if (C.bits != 5) error = C.lut_err[lut_index(tmp_val.re) * 256 + lut_index(tmp_val.im)];
                    else soft_calc_dev(C, tmp_val, nullptr, &error);
                } else {
                    error = cphase(cmul(descr, cf32{0.707f, -0.707f}));
                    if (i == next_pilot + 35) { ++pb; next_pilot = pb < pilot_blocks ? pilot_start(pb) : -1; }
                }
                out[i] = descr;
            } else {
                if (i < 26) error = cphase(cmul(tmp_val, cconj(T.sof[i])));
                else error = cphase(cmul(tmp_val, cconj(plsc[i - 26])));
                // header symbols are overwritten by the PLHDR demod below (module_dvbs2_demod.cpp:332-333)
            }
            pll.advance(error);
            pll.wrap_pi();
        }
        // ---- PL header demod (dvbs2_plhdr_demod.cpp:33-67): own loop over the 90 header symbols, PLSC decode
        unsigned long long plheader = 0;
        const cf32 rot{(float)0.70710678118654757, (float)-0.70710678118654746};   // (cos(-pi/4), sin(-pi/4)) in double, cast
        for (int i = 0; i < 90; ++i) {
            cf32 tmp_val = cmul(fr[i], phasor(-hdr.phase));
            float error = ((tmp_val.re > 0 ? 1.0f : -1.0f) * tmp_val.im) - ((tmp_val.im > 0 ? 1.0f : -1.0f) * tmp_val.re);
            cf32 o = (i & 1) ? cf32{-tmp_val.re, tmp_val.im} : cf32{tmp_val.im, tmp_val.re};
            out[i] = o;
            if (i >= 26) {
                bool value = cmul(o, rot).re > 0;
                plheader = plheader << 1 | (unsigned long long)(!value);
            }
            hdr.advance(error);
            hdr.wrap_pi();
        }
        hdr.phase += hdr.freq * (plframe - 91);
        hdr.advance(0.f);
        hdr.wrap_pi();
        int best = 0, diffs = 64;
        for (int c = 0; c < 128; ++c) {
            int dd = __popcll((T.plsc_code[c] ^ plheader) & ((1ull << 60) - 1));
            if (dd < diffs) { best = c; diffs = dd; }
        }
        S2FrameStats stt;
        stt.best_match = 0.f; stt.ldpc_trials = 0; stt.bch_corr = 0;   // filled in by the host
        stt.detected_modcod = (best >> 2) & 31; stt.detected_short = (best & 2) >> 1; stt.detected_pilots = best & 1;
        stt.fed_err = est;
        stats[f] = stt;
    }
    st->pll_phase = pll.phase; st->pll_freq = pll.freq;
    st->hdr_phase = hdr.phase; st->hdr_freq = hdr.freq;
    st->nco_freq = nco_freq;
}

// ------------------------------------------------------------------------------------------------ demapper
// grid (x: symbol tiles, y: frame).  LUT fetch + bit de-interleave fused: LLR c of payload symbol j goes to
// column c (8PSK 3/5: columns reversed), QPSK just swaps the pair.
__global__ __launch_bounds__(256) void s2_demap_kernel(S2ConstelDev C, int rate, int slots, int pilots, int plframe,
                                                       const cf32* __restrict__ pllout, int8_t* __restrict__ llr, int N) {
    const int f = blockIdx.y;
    const cf32* __restrict__ fr = pllout + (size_t)f * plframe;
    int8_t* __restrict__ out = llr + (size_t)f * N;
    const int nsym = slots * 90;
    const int bits = C.bits;
    const int rows = N / bits;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nsym; j += gridDim.x * 256) {
        int pos = 90 + j;
        if (pilots) pos += 36 * (j / 1440);   // pilot blocks already passed (one after every 16 slots)
        cf32 v = fr[pos];
        int8_t b[5];
        if (bits != 5) {
            const int8_t* __restrict__ e = C.lut_bits + ((size_t)lut_index(v.re) * 256 + lut_index(v.im)) * bits;
            for (int c = 0; c < bits; ++c) b[c] = e[c];
        } else {
            soft_calc_dev(C, v, b, nullptr);
        }
        if (bits == 2) {
            out[2 * j + 1] = b[0]; out[2 * j] = b[1];
        } else {
            for (int c = 0; c < bits; ++c) {
                int col = (C.constel == C_8PSK && rate == R3_5) ? (2 - c) : c;
                out[col * rows + j] = b[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t s2_frontend_launch(const S2StreamWork* d_work, int nstreams, S2LoopCoefs coefs, const float* d_bank, hipStream_t st) {
    hipLaunchKernelGGL(s2_frontend_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, st, d_work, nstreams, coefs, d_bank);
    return hipGetLastError();
}
hipError_t s2_rrc_decim_launch(const S2StreamWork* d_work, int nstreams, int max_count, const float* d_taps, int ntaps, hipStream_t st) {
    int gx = (max_count / 2 + 2 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(s2_rrc_decim_kernel, dim3(gx, nstreams), dim3(256), 0, st, d_work, d_taps, ntaps);
    hipLaunchKernelGGL(s2_rrc_state_kernel, dim3(nstreams), dim3(128), 0, st, d_work, ntaps);
    return hipGetLastError();
}
hipError_t s2_plsync_launch(const cf32* const* d_win, int nwin, int raw, int* d_best_pos, float* d_best_match, hipStream_t st) {
    hipLaunchKernelGGL(s2_plsync_kernel, dim3(nwin), dim3(256), 0, st, d_win, raw, d_best_pos, d_best_match);
    return hipGetLastError();
}
hipError_t s2_frame_loops_launch(const S2StreamWork* d_work, int nstreams, const S2FrameRef* d_frames, const int* d_first,
                                 S2LoopCoefs coefs, S2PlTablesDev tabs, S2ConstelDev con, int pls_code, int slots, int pilots,
                                 int pilot_blocks, int plframe, cf32* d_pllout, S2FrameStats* d_stats, hipStream_t st) {
    hipLaunchKernelGGL(s2_frame_loops_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, st, d_work, nstreams, d_frames, d_first, coefs,
                       tabs, con, pls_code, slots, pilots, pilot_blocks, plframe, d_pllout, d_stats);
    return hipGetLastError();
}
hipError_t s2_demap_launch(S2ConstelDev con, int rate, int shortframe, int slots, int pilots, int plframe, const cf32* d_pllout,
                           int nframes, int8_t* d_llr, int N, hipStream_t st) {
    (void)shortframe;
    int gx = (slots * 90 + 255) / 256;
    hipLaunchKernelGGL(s2_demap_kernel, dim3(gx, nframes), dim3(256), 0, st, con, rate, slots, pilots, plframe, d_pllout, d_llr, N);
    return hipGetLastError();
}

}  // namespace s2
