// DVB-S2 receive-chain kernels for gfx950 (compiled with -ffp-contract=off: every fp32 operation is a
// separately rounded IEEE op in the order written, which is what the CPU restatement does).  sin/cos/atan2/exp/log
// are the engine's own straight-line definitions (include/dvbs2gpu_math.h) -- no device libm, no hardware
// v_sin/v_cos -- so these kernels are bit-identical to the CPU restatement that includes the same header.
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
#include <stdlib.h>
#include "s2_rx.h"
#include "../../include/dvbs2gpu_math.h"
#include "s2_params.h"

namespace s2 {

__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return cf32{a.re * b.re - a.im * b.im, a.im * b.re + a.re * b.im}; }
__device__ __forceinline__ cf32 cconj(cf32 a) { return cf32{a.re, -a.im}; }
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return cf32{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return cf32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf32 cscale(cf32 a, float s) { return cf32{a.re * s, a.im * s}; }
__device__ __forceinline__ float camp(cf32 a) { return sqrtf(a.re * a.re + a.im * a.im); }
__device__ __forceinline__ float cphase(cf32 a) { return dvbs2m::atan2f_det(a.im, a.re); }
__device__ __forceinline__ cf32 phasor(float x) { cf32 r; dvbs2m::sincosf_det(x, &r.im, &r.re); return r; }

// pointers read out of device structs are generic ("flat") to the compiler; flat accesses count against the LDS counter as well
// and serialise with the LDS reads of the serial loops.  These are known to be global memory.
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T* as_global(T* p) {
    return (__attribute__((address_space(1))) T*)p;
}

typedef float __attribute__((ext_vector_type(2))) f32x2;
typedef float __attribute__((ext_vector_type(4))) f32x4;
// (tried in round 3: these sample streams -- IQ in, per-sample scratch, timing-recovery output, ~57 GB per headline step -- as non-temporal
// accesses, so that they would not push the LDPC decoder's message records out of the Infinity Cache in the pipelined mode: no measurable change)
__device__ __forceinline__ cf32 ldg(const cf32* p) {
    const f32x2 v = *as_global(reinterpret_cast<const f32x2*>(p));
    return cf32{v.x, v.y};
}
__device__ __forceinline__ void stg(cf32* p, cf32 v) { *as_global(reinterpret_cast<f32x2*>(p)) = f32x2{v.re, v.im}; }

// branch-free select (the compiler turns short float ternaries of the serial loops into exec-mask branches otherwise)
__device__ __forceinline__ float fsel(bool c, float a, float b) {
    const int m = -(int)c;
    return __builtin_bit_cast(float, (__builtin_bit_cast(int, a) & m) | (__builtin_bit_cast(int, b) & ~m));
}

// x > hi ? hi : (x < lo ? lo : x) as ONE v_med3_f32 (the compare-select form costs a compare, a wait state and a select per bound in
// loops whose speed is their instruction count).  Same value for every x that is not a NaN (lo < hi, neither of them a zero: the
// hardware orders -0 < +0); a NaN -- which the reference would carry in its loop state for ever -- comes out as a bound instead.
__device__ __forceinline__ float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
struct PclDev {
    float alpha, beta, phase, freq, minFreq, maxFreq;
    __device__ __forceinline__ void advance(float err) {
        freq += beta * err;
        freq = clamp_med3(freq, minFreq, maxFreq);
        phase += freq + alpha * err;
    }
    // the same for a phase that moved by less than 2 pi since it was last wrapped (every per-symbol advance of the PLL, PLHDR and
    // Costas loops: |freq| <= pi, |alpha * err| << pi): the two while loops below run at most once each -> two selects, no divergent code
    __device__ __forceinline__ void wrap_pi_once() {
        const float PI_F = 3.14159265358979323846f;
        const float delta = PI_F - (-PI_F);
        phase = phase > PI_F ? phase - delta : phase;
        phase = phase < -PI_F ? phase + delta : phase;
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
// sin/cos on the serial per-sample paths: the shared straight-line definition (~35 instructions for both values, no slow-path
// branches).  The hardware's v_sin_f32 / v_cos_f32 would be 3 instructions (DVB-S bank +18 %, S2 headline +1.5 % when tried), but
// their ~1e-6 absolute error has no CPU restatement, and parity with the oracle is the gate: every loop uses the shared definition.
__device__ __forceinline__ cf32 phasor_fast(float x) { return phasor(x); }
__device__ __forceinline__ cf32 phasor_hw(float x) { return phasor(x); }

// The front end is split along its dependency structure:
//   agc_pc_kernel      LANE = STREAM.  The AGC gain and NCO phase recurrences depend only on the input, are strictly serial in
//                      time and ~40 instructions per sample: one lane runs one stream, a wave 64 streams, and the per-sample
//                      (gain, phase) pairs go to the stream's scratch area; a second wave of the workgroup moves the tiles.
//   s2_gardner_kernel  EIGHT LANES PER STREAM (polyphase arm x re/im), 8 streams per wave.  Tiles of 64 samples per stream are
//                      staged through LDS by all 64 lanes -- y = x*gain, z = y*phasor(-phase), the parallel part of
//                      FastAGC/FreqShift; the loads of the next tile stay in flight during the loop -- then every lane group
//                      runs its stream's Gardner loop: one 8-tap polyphase dot product per lane in the reference's
//                      accumulation order, DPP row shifts to bring the three arms together, a quad swap to add the re/im
//                      halves of the timing error, two DPP broadcasts to hand it to the whole group.
// Many streams = many lanes: the batch fills the GPU, and nothing here is redundant across lanes (the first version ran the
// serial chains once per wave, 64 lanes wide, and took 47 ms for 4096 x 43380 samples; this one ~4x less).
// Wave priorities of the serial front-end kernels while the LDPC decoder of the previous call shares the SIMDs (pipelined mode; the
// decoder's parallel phases run at 0, its serial ones at 3).  A/B on the bench (tools/ab_build.sh, 4096 streams x 8 frames): everything
// at 2: step 474 ms (decoder launch 422 ms, front end 330 ms, i.e. 100 ms of slack on the front-end stream); the Gardner kernel -- the
// largest consumer of issue slots among them -- at 0: decoder 393 ms, front end 391 ms, step 457 ms; the frame loops at 0 as well: the
// front end becomes the critical path (479 ms).  So: Gardner yields to the decoder, the others keep their latency.
#ifndef AGC_PRIO
#define AGC_PRIO 2
#endif
#ifndef G_PRIO
#define G_PRIO 1      // (round 6, with every front-end kernel above the decoder's parallel phases: the front end of the headline step is through in 129 ms instead of 256 and the
                      //  step is the decoder again: 283.9 -> 274.2 ms; at 0 the timing recovery shared the decoder's level for (8 - share) of every 8 tiles)
#endif
#ifndef G_PRIO_DUTY
#define G_PRIO_DUTY 0
#endif
#ifndef G_PRIO_HI
#define G_PRIO_HI (G_PRIO + 1)   // the timing recovery's priority for `share` of every 8 tiles ...
#endif
#ifndef G_PRIO_LO
#define G_PRIO_LO G_PRIO         // ... and for the rest
#endif
#ifndef FL_PRIO
#define FL_PRIO 2
#endif
#ifndef POST_PRIO
#define POST_PRIO 2   // wave priority of the data-parallel post stages (RRC, PL-sync walk, demapper) beside the decoder WHERE THE HOST ASKS FOR IT (S2LoopCoefs::post_prio: the balancer's share >= 5 = the front end is the critical path; a decoder-bound configuration -- config 5's stand-in -- loses 8 % to post stages above its decoder).  Round 6: at 0 -- the decoder's own -- an RRC slice took 35-44 ms
                      // beside the decoder (6 alone), a PL-sync walk 30 (5): they share the AGC's stream, so the AGC slices, and behind them the timing recovery, waited for them --
                      // the front end was the step (299.7 ms; 285.4 at 1, 285.9 at 2: same call, tools/ab.sh)
#endif
#ifndef FE_PRIO
#define FE_PRIO 2   // wave priority of the serial front-end loops (A/B switch; the decoder's parallel phases run at 0, its serial ones at 3)
#endif
// a uniform value stored by every lane (1) or by lane 0 behind a predicate (0): the predicate costs the single-carrier chains a taken branch
// per sample (A/B switches)
#ifndef FD_STORE_ALL
#define FD_STORE_ALL 0      // (the timing recovery of a 4096-carrier bank: 128 against 133 ms per call with the predicate; one carrier is paced by the FLL)
#endif
constexpr int G_TILE = 64;     // samples per stream per staging tile
constexpr int G_SPW = 8;       // streams per wave (8 lanes each)
constexpr int G_PITCH = G_TILE + 9;   // 7 history + tile, odd pitch spreads the rows over the LDS banks

__device__ __forceinline__ size_t fe_scratch_offset(int n) { return (size_t)n + n / 16 + 128; }

// ---- serial per-sample recurrences, LANE = STREAM, with the memory traffic on a second wave ------------------------------
// Workgroup = 2 waves for 64 streams.  Wave 0 runs the recurrences of its 64 streams out of LDS tiles (in place: the result
// overwrites the input slot); wave 1 moves the tiles: coalesced row loads of the next tile, coalesced row stores of the
// previous one, double-buffered.  The compute wave never issues a global access, so no memory latency and no store drain
// (loads and stores share one counter on this hardware) ever enters the serial chain.
#ifndef AG_T_N
#define AG_T_N 16
#endif
constexpr int AG_T = AG_T_N;             // samples per stream per tile (17 KB of LDS: fits beside a resident LDPC workgroup)
struct AgcS2Traits {                     // FastAGC gain + FreqShift phase recurrences; result = (gain, phase) per sample
    static constexpr bool DVBS_SLICES = false;
    typedef S2StreamWork Work;
    typedef S2LoopCoefs Coefs;
    struct Regs { float gain, nph, nfr; };
    static __device__ __forceinline__ const cf32* in_ptr(const Work& w) { return w.in; }
    static __device__ __forceinline__ cf32* out_ptr(const Work& w) { return w.fe_out + fe_scratch_offset(w.count); }
    static __device__ __forceinline__ Regs load(const Work& w) { return Regs{w.st->agc_gain, w.st->nco_phase, w.st->nco_agc}; }
    static __device__ __forceinline__ void store(const Work& w, const Regs& r) { w.st->agc_gain = r.gain; w.st->nco_phase = r.nph; }
    static __device__ __forceinline__ cf32 step(Regs& r, cf32 x, const Coefs& co) {
        const cf32 res{r.gain, r.nph};
        // FastAGC (SDR++ loop/fast_agc.h as used at module_dvbs2_demod.cpp:222)
        const cf32 y = cscale(x, r.gain);
        const float a = camp(y);
        r.gain += (1.0f - a) * co.agc_rate;
        r.gain = r.gain > 10e6f ? 10e6f : r.gain;
        // FreqShift phase accumulator (common/dsp/demod/freq_shift.cpp)
        r.nph += r.nfr;
        // (a binary32 value exceeds the double 2 pi exactly when it reaches the first binary32 above it: one float compare per sample, the
        //  double-precision wrap of the reference only when it acts -- once per 2 pi / |freq| samples)
        if (__builtin_expect(__any(__builtin_fabsf(r.nph) >= 6.2831854820251465f), 0)) {
            while ((double)r.nph > 6.283185307179586) r.nph = (float)((double)r.nph - 6.283185307179586);
            while ((double)r.nph < -6.283185307179586) r.nph = (float)((double)r.nph + 6.283185307179586);
        }
        return res;
    }
};
struct AgcDvbsTraits {                   // FastAGC only; result = scaled sample
    static constexpr bool DVBS_SLICES = true;
    typedef DvbsStreamWork Work;
    typedef DvbsLoopCoefs Coefs;
    struct Regs { float gain; };
    static __device__ __forceinline__ const cf32* in_ptr(const Work& w) { return w.in; }
    static __device__ __forceinline__ cf32* out_ptr(const Work& w) { return w.buf_a; }
    static __device__ __forceinline__ Regs load(const Work& w) { return Regs{w.st->agc_gain}; }
    static __device__ __forceinline__ void store(const Work& w, const Regs& r) { w.st->agc_gain = r.gain; }
    static __device__ __forceinline__ cf32 step(Regs& r, cf32 x, const Coefs& co) {
        const cf32 y = cscale(x, r.gain);
        const float a = camp(y);
        r.gain += (1.0f - a) * co.agc_rate;
        r.gain = r.gain > 10e6f ? 10e6f : r.gain;
        return y;
    }
};

// Sample sub-range `sub` of `nsub` of a call's n samples (the time-sliced front end: the AGC / NCO recurrences of slice c+1 run beside the
// timing loop of slice c; every stage keeps its state in the stream record, so slicing a call changes nothing in the results)
__device__ __forceinline__ void fe_sub_range(int n, int sub, int nsub, int& lo, int& hi) {
    lo = (int)((long long)n * sub / nsub);
    hi = (int)((long long)n * (sub + 1) / nsub);
}
// The DVB-S receiver's slices (12 or more): the first and the last two are a quarter and a half of the others -- what a call pays beyond its slowest
// stage is the stages before that one on the FIRST slice and the stages behind it on the LAST one (one carrier: 2.7 of 22 ms with equal slices).
// Weights 1 2 4 4 ... 4 2 1 in quarters of a full slice; every stage derives its range from here, and the data-parallel ones stride over whatever they get.
__device__ __forceinline__ int fe_taper_cum(int c, int nsub) { return c <= 0 ? 0 : c == 1 ? 1 : c <= nsub - 2 ? 4 * c - 5 : c == nsub - 1 ? 4 * nsub - 11 : 4 * nsub - 10; }
__device__ __forceinline__ void fe_sub_range_dvbs(int n, int sub, int nsub, int& lo, int& hi) {
    if (nsub < 12) { fe_sub_range(n, sub, nsub, lo, hi); return; }
    const int total = 4 * nsub - 10;
    lo = (int)((long long)n * fe_taper_cum(sub, nsub) / total);
    hi = (int)((long long)n * fe_taper_cum(sub + 1, nsub) / total);
}

template <class TR>
__global__ __launch_bounds__(128) void agc_pc_kernel(const typename TR::Work* __restrict__ work, int nstreams, typename TR::Coefs co, int sub, int nsub) {
    __shared__ cf32 buf[2][64][AG_T + 1];
    __shared__ const cf32* s_in[64];
    __shared__ cf32* s_out[64];
    __shared__ int s_n[64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, s = blockIdx.x * 64 + lane;
    const bool act = s < nstreams;
    const typename TR::Work w = work[act ? s : 0];
    int lo, hi;
    if constexpr (TR::DVBS_SLICES) fe_sub_range_dvbs(act ? w.count : 0, sub, nsub, lo, hi);
    else fe_sub_range(act ? w.count : 0, sub, nsub, lo, hi);
    const int n = hi - lo;
    typename TR::Regs regs = TR::load(w);
    // (a stream without samples in this slice may come with a null input pointer: the movers' clamped loads then read the work table)
    if (wave == 0) { s_in[lane] = n > 0 ? TR::in_ptr(w) + lo : reinterpret_cast<const cf32*>(work); s_out[lane] = TR::out_ptr(w) + lo; s_n[lane] = n; }
    int nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    const int ntiles = (nmax + AG_T - 1) / AG_T;
    __syncthreads();
    // tile movers (wave 1): each instruction moves 64/AG_T rows of a tile, AG_T*8 contiguous bytes per row.  The loads of a tile are issued
    // a whole period before their values go to LDS (unconditional, clamped indices: a load behind a branch is waited for on the spot), so
    // the memory latency of a 64-stream tile never enters a period -- with it inside, the mover, not the chains, set the pace of a full wave
    // of streams (160 ns per sample against 100 for a single stream)
    const int half = lane / AG_T, col = lane % AG_T;
    constexpr int RPI = 64 / AG_T;        // rows per load/store instruction
    constexpr int NQ = 64 / RPI;          // instructions per tile
    cf32 v[NQ];                           // (pointers and counts stay in LDS: this kernel has to fit into 128 registers, see s2_frame_loops_kernel)
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = ldg(s_in[RPI * q + half] + min(t * AG_T + col, max(s_n[RPI * q + half] - 1, 0)));
    };
    auto commit = [&](cf32 (*B)[AG_T + 1]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) B[RPI * q + half][col] = v[q];
    };
    auto store_tile = [&](int t, cf32 (*B)[AG_T + 1]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = RPI * q + half, idx = t * AG_T + col;
            if (idx < s_n[row]) stg(s_out[row] + idx, B[row][col]);
        }
    };
    if (wave == 1 && ntiles > 0) { issue(0); commit(buf[0]); if (ntiles > 1) issue(1); }
    __syncthreads();
    // the serial chains are latency-critical and issue little: win the issue arbitration against throughput kernels (the LDPC
    // decoder of the previous call shares the SIMDs in the pipelined mode)
    if (wave == 0) __builtin_amdgcn_s_setprio(AGC_PRIO);
    for (int t = 0; t < ntiles; ++t) {
        if (wave == 0) {
            cf32(*B)[AG_T + 1] = buf[t & 1];
            const int m = min(AG_T, n - t * AG_T);
            if (__all(m == AG_T || m <= 0)) {
                // whole tiles (all but a stream's last): the tile's samples fetched up front, no predicate inside the chain; streams that
                // are through (or absent) run along on stale data and get their state back
                const typename TR::Regs keep = regs;
                cf32 x[AG_T];
#pragma unroll
                for (int i = 0; i < AG_T; ++i) x[i] = B[lane][i];
#pragma unroll
                for (int i = 0; i < AG_T; ++i) B[lane][i] = TR::step(regs, x[i], co);
                if (m <= 0) regs = keep;
            } else {
#pragma unroll 4
                for (int i = 0; i < AG_T; ++i)
                    if (i < m) B[lane][i] = TR::step(regs, B[lane][i], co);
            }
        } else {
            if (t >= 1) store_tile(t - 1, buf[(t - 1) & 1]);          // (out of the buffer tile t+1 goes into next)
            if (t + 1 < ntiles) commit(buf[(t + 1) & 1]);
            if (t + 2 < ntiles) issue(t + 2);
        }
        __syncthreads();
    }
    if (wave == 1 && ntiles > 0) store_tile(ntiles - 1, buf[(ntiles - 1) & 1]);
    if (wave == 0 && act) TR::store(w, regs);
}

#define DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)(v)), (ctrl), 0xf, 0xf, false))
template <typename T>
__device__ __forceinline__ T* readlane_ptr(T* p, int srclane) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, srclane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), srclane);
    return (T*)(((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(64) void s2_gardner_kernel(const S2StreamWork* __restrict__ work, int nstreams, S2LoopCoefs co,
                                                        const float* __restrict__ bank_g, int sub, int nsub) {
    __shared__ __attribute__((aligned(16))) float bank[GARDNER_PHASES * GARDNER_TAPS];
    __shared__ float win[G_SPW * 2 * G_PITCH];          // [stream][re/im][7 history + tile]
    const int lane = threadIdx.x, g = lane >> 3, r = lane & 7, arm = r >> 1, c = r & 1;   // arm 0/1/2 = phase-1 / phase / phase+1, 3 = spare
    const int s0 = blockIdx.x * G_SPW, s = s0 + g;
    const bool act = s < nstreams;
    for (int i = lane; i < GARDNER_PHASES * GARDNER_TAPS; i += 64) bank[i] = bank_g[i];
    S2StreamWork w = work[act ? s : 0];
    int lo, hi;
    fe_sub_range(act ? w.count : 0, sub, nsub, lo, hi);
    const int n = hi - lo;
    S2StreamState* st = w.st;
    PclDev pcl{co.g_alpha, co.g_beta, st->g_phase, st->g_freq, co.g_min_freq, co.g_max_freq};
    int offset = st->g_offset, spsctr = st->g_spsctr, outCount = sub ? st->n_fe_out : 0;   // (later slices append to the call's output)
    float* row = &win[(g * 2 + c) * G_PITCH];           // aliases the staging writes below: no __restrict__
    auto outc = as_global(reinterpret_cast<float*>(w.fe_out) + c);
    const cf32* gpp = w.fe_out + fe_scratch_offset(w.count) + lo;
    w.in += lo;
    if (arm == 0)
        for (int k = 0; k < GARDNER_TAPS - 1; ++k) row[k] = c ? st->g_hist[k].im : st->g_hist[k].re;
    int nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    // staging registers: sample `lane` of the next tile of each of the 8 streams (loads stay in flight during the Gardner loop)
    cf32 px[G_SPW], pg[G_SPW];
    auto issue = [&](int base) {
#pragma unroll
        for (int jj = 0; jj < G_SPW; ++jj) {
            const cf32* inj = readlane_ptr(w.in, jj * 8);
            const cf32* gpj = readlane_ptr(gpp, jj * 8);
            const int cj = __builtin_amdgcn_readlane(n, jj * 8);
            if (base + lane < cj) { px[jj] = ldg(inj + base + lane); pg[jj] = ldg(gpj + base + lane); }
        }
    };
    auto commit = [&](int base) {
#pragma unroll
        for (int jj = 0; jj < G_SPW; ++jj) {
            const int cj = __builtin_amdgcn_readlane(n, jj * 8);
            if (base + lane < cj && lane < G_TILE) {
                const cf32 z = cmul(cscale(px[jj], pg[jj].re), phasor_fast(-pg[jj].im));   // FastAGC scaling, FreqShift rotation
                win[(jj * 2) * G_PITCH + GARDNER_TAPS - 1 + lane] = z.re;
                win[(jj * 2 + 1) * G_PITCH + GARDNER_TAPS - 1 + lane] = z.im;
            }
        }
    };
    issue(0);
    __syncthreads();
    __builtin_amdgcn_s_setprio(G_PRIO);       // latency-critical serial loop (see agc_pc_kernel)
    for (int base = 0; base < nmax; base += G_TILE) {
        // co.g_prio_duty (+ the build's G_PRIO_DUTY) of every 8 tiles run one priority level up: the balance point between "this kernel yields to
        // the decoder" (the front end becomes the critical path) and "it does not" (the decoder does) lies between two priority levels, and
        // where it lies depends on the MODCOD -- the host moves it from call to call (s2_demod.hip)
        if ((((unsigned)base / G_TILE) & 7u) < (unsigned)(co.g_prio_duty + G_PRIO_DUTY)) __builtin_amdgcn_s_setprio(G_PRIO_HI); else __builtin_amdgcn_s_setprio(G_PRIO_LO);
        commit(base);
        __syncthreads();
        issue(base + G_TILE);
        // ---- Gardner (common/dsp/demod/gardner.cpp:89-150): outputs whose 8-sample window starts inside this tile
        const int m = max(0, min(G_TILE, n - base));
        // (the trip count is bounded: a poisoned loop state -- NaN input -- must not hang the GPU)
        for (int guard = 0; guard < 4 * G_TILE && __any(offset < base + m && offset >= base); ++guard) {
            if (offset < base + m && offset >= base) {
                int phase = (int)floorf(pcl.phase * 128.0f);
                phase = phase < 0 ? 0 : (phase > 127 ? 127 : phase);
                int my = phase;
                if (arm == 0) my = phase > 0 ? phase - 1 : 0;
                if (arm == 2) my = phase < 127 ? phase + 1 : 127;
                const float* xw = row + (offset - base);
                const float* t = &bank[my * 8];
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += xw[k] * t[k];
                // arm-1 lanes (r = 2: re, 3: im) collect the neighbours' arms: row_shr:2 = lane-2 (phase-1), row_shl:2 = lane+2 (phase+1)
                const float xm = DPP_F(acc, 0x112), xp = DPP_F(acc, 0x102), xo = acc;
                if (arm == 1) outc[2 * outCount] = xo;
                ++outCount;
                // straight-line on purpose (streams of one wave differ in spsctr): the error of the off-symbol outputs is masked to 0
                const float d = fsel(phase == 0, xp - xo, fsel(phase == 127, xo - xm, (xp - xm) * 0.5f));
                const float e = (xo > 0 ? 1.0f : -1.0f) * d;                 // valid in lanes r = 2 (re half), 3 (im half)
                const float eo = DPP_F(e, 0xB1);                             // quad_perm [1,0,3,2]: the other half
                const float er = -(e + eo);                                  // -(re part + im part)
                const float eq = DPP_F(er, 0xAA);                            // quad_perm [2,2,2,2]: r = 0..3 <- r = 2
                const float eh = DPP_F(eq, 0x114);                           // row_shr:4: r = 4..7 <- r = 0..3
                float error = spsctr == 0 ? (r < 4 ? eq : eh) : 0.f;
                spsctr = spsctr >= 1 ? 0 : spsctr + 1;
                error = clamp_med3(error, -1.0f, 1.0f);
                pcl.advance(error);
                const float delta = floorf(pcl.phase);
                offset = (int)((float)offset + delta);
                pcl.phase -= delta;
            }
        }
        __syncthreads();
        // ---- slide the 7-sample history (arm-0 lanes own the rows)
        if (arm == 0) {
            float h[GARDNER_TAPS - 1];
#pragma unroll
            for (int k = 0; k < GARDNER_TAPS - 1; ++k) h[k] = row[m + k];
#pragma unroll
            for (int k = 0; k < GARDNER_TAPS - 1; ++k) row[k] = h[k];
        }
        __syncthreads();
    }
    if (act && arm == 0) {
        for (int k = 0; k < GARDNER_TAPS - 1; ++k) {
            if (c) st->g_hist[k].im = row[k]; else st->g_hist[k].re = row[k];
        }
        if (c == 0) {
            st->g_phase = pcl.phase; st->g_freq = pcl.freq; st->g_offset = offset - n; st->g_spsctr = spsctr;
            st->n_fe_out = outCount;
            st->n_fe_slice[sub & (S2_FE_MAX_SLICES - 1)] = outCount;
        }
    }
}

// ---- timing recovery, second form: the serial chain holds only what the recurrence needs ------------------------------------
// Of the reference loop's two outputs per symbol only the on-symbol one feeds the loop (gardner.cpp:100-131: the error of the
// other is 0, so PCL::advance(0) moves the phase by the loop frequency alone), and an output VALUE never feeds anything: the chain is
// "interpolate three arms at the on-symbol instant -> error -> advance; advance once more".  So the RESOLVER wave (8 lanes per stream,
// 8 streams, as above) walks symbol by symbol -- on-symbol output, then its follower without any interpolation -- and leaves one word
// (window slot, polyphase arm) per output in an LDS list; a PRODUCER wave of the same workgroup computes every output value from that
// list one period later (the same 8-tap dot product in the reference's accumulation order, lane = output: coalesced stores instead of
// a predicated store inside the chain), and stages the next samples (FastAGC scaling + FreqShift rotation).  Per symbol the chain is
// ~125 instructions of one wave instead of ~230.  Streams of a wave start a period aligned on an on-symbol output (one single step
// at the start of a slice where the state says otherwise); the last three sample positions of a slice go through single steps, so a
// slice ends in exactly the state the reference's loop has after the same samples.
// Samples live in a ring of 4 periods per stream and component (slot = buffer index mod ring; the first 8 slots are mirrored behind the
// ring so that an 8-sample window never wraps): period t is resolved while t+1 is being staged and the values of t-1 are produced.
#ifndef G2_EXP
#define G2_EXP 0          // development switches for TIMING experiments (results wrong): 1 no output values, 2 no staging
#endif
#ifndef S2_G2_ASM
#define S2_G2_ASM 1       // the resolver's symbol loop of s2_gardner2_kernel written out (A/B switch)
#endif
#ifndef G2_TILE_N
#define G2_TILE_N 16
#endif
#ifndef G2_EXTRA_VALU
#define G2_EXTRA_VALU 0     // sensitivity experiment: dead vector instructions per producer wave and period (how much does the decoder beside it pay per front-end instruction?)
#endif
#ifndef G2_FORCE_VGPRS
#define G2_FORCE_VGPRS 0
#endif
#ifndef G2_PROD_PRIO
#define G2_PROD_PRIO 1    // wave priority of the producer waves (A/B switch).  Round 6: at 0 -- the decoder's own level, one of seven waves of its SIMD -- the producer, not the resolver, paced a period beside the
                          // decoder; headline (driver's command, same call) 258.6 / 259.7 ms per step at 0, 255.1 / 252.6 at 1, 254.2 / 252.8 at 2; plugin's mode 126.9 / 124.7 / 123.9
#endif
#ifndef G2_PAIRS_N
#define G2_PAIRS_N 2      // resolver + producer pairs per workgroup (A/B switch: 1 = round 5's 128-thread workgroups)
#endif
constexpr int G2_PAIRS = G2_PAIRS_N;
constexpr int G2_TILE = G2_TILE_N;            // samples per stream and period
constexpr int G2_RING = 4 * G2_TILE;
constexpr int G2_PITCH = G2_RING + 8 + 1;     // ring + mirror of its first 8 slots; odd pitch
constexpr int G2_LIST = G2_TILE + 8;          // outputs of a stream per period: <= (G2_TILE + 3) / 0.96 (list entries beyond are refused by the loop bounds)
constexpr int G2_SPT = (G_SPW * G2_TILE) / 64; // samples each producer lane stages per period

__device__ __forceinline__ void lds_only_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(64 * 2 * G2_PAIRS) void s2_gardner2_kernel(const S2StreamWork* __restrict__ work, int nstreams, S2LoopCoefs co,
                                                         const float* __restrict__ bank_g, int sub, int nsub) {
    __shared__ __attribute__((aligned(16))) float bank[GARDNER_PHASES * GARDNER_TAPS];
    // G2_PAIRS resolver + producer pairs per workgroup, each with its own 8 streams, rings and lists; they share the tap bank and the period barriers.  Why two: the hardware
    // spreads the FOUR waves of a 256-thread workgroup over the four SIMDs of its compute unit, one each; of the two waves of a 128-thread workgroup it says nothing -- two such
    // workgroups per compute unit leave one SIMD with two of these waves and one with none in 5 of 8 compute units (tools/ubench/placement.hip), and a decoder workgroup beside
    // them (three waves per SIMD, a barrier per layer) runs at the pace of its most loaded SIMD.
    __shared__ float ring_all[G2_PAIRS][G_SPW * 2 * G2_PITCH];        // [stream][re/im][slot]
    __shared__ uint32_t list_all[G2_PAIRS][2][G_SPW][G2_LIST];        // per period parity: slot << 7 | arm of every output, in output order
    __shared__ int s_cnt_all[G2_PAIRS][2][G_SPW], s_ostart_all[G2_PAIRS][2][G_SPW];
    __shared__ const cf32* s_in_all[G2_PAIRS][G_SPW];
    __shared__ const cf32* s_gp_all[G2_PAIRS][G_SPW];
    __shared__ cf32* s_out_all[G2_PAIRS][G_SPW];
    __shared__ int s_n_all[G2_PAIRS][G_SPW];
    const int wave_wg = threadIdx.x >> 6, pair = wave_wg >> 1, wave = wave_wg & 1;
    const int lane = threadIdx.x & 63, g = lane >> 3, r = lane & 7, arm = r >> 1, c = r & 1;
    float* const ring = ring_all[pair];
    uint32_t (*const list)[G_SPW][G2_LIST] = list_all[pair];
    int (*const s_cnt)[G_SPW] = s_cnt_all[pair];
    int (*const s_ostart)[G_SPW] = s_ostart_all[pair];
    const cf32** const s_in = s_in_all[pair];
    const cf32** const s_gp = s_gp_all[pair];
    cf32** const s_out = s_out_all[pair];
    int* const s_n = s_n_all[pair];
    const int s0 = (blockIdx.x * G2_PAIRS + pair) * G_SPW, s = s0 + g;
    const bool act = s < nstreams;
    for (int i = threadIdx.x; i < GARDNER_PHASES * GARDNER_TAPS; i += 64 * 2 * G2_PAIRS) bank[i] = bank_g[i];
    S2StreamWork w = work[act ? s : 0];
    int lo, hi;
    fe_sub_range(act ? w.count : 0, sub, nsub, lo, hi);
    const int n = hi - lo;
    S2StreamState* st = w.st;
    float* row = &ring[(g * 2 + c) * G2_PITCH];
    if (wave == 0) {
        if (r == 0) { s_in[g] = w.in + lo; s_gp[g] = w.fe_out + fe_scratch_offset(w.count) + lo; s_out[g] = w.fe_out; s_n[g] = n; }
        if (arm == 0)
            for (int k = 0; k < GARDNER_TAPS - 1; ++k) { const float h = c ? st->g_hist[k].im : st->g_hist[k].re; row[k] = h; row[G2_RING + k] = h; }
    }
    int nmax = n;
    if constexpr (G2_PAIRS > 1) {
        // (the other pairs' streams: every wave of the workgroup runs the same number of periods -- the barriers are the workgroup's)
#pragma unroll
        for (int op = 1; op < G2_PAIRS; ++op) {
            const int so = (blockIdx.x * G2_PAIRS + ((pair + op) % G2_PAIRS)) * G_SPW + g;
            int lo2, hi2;
            fe_sub_range(so < nstreams ? work[so].count : 0, sub, nsub, lo2, hi2);
            nmax = max(nmax, hi2 - lo2);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    const int ntiles = (nmax + G2_TILE - 1) / G2_TILE;
    __syncthreads();

    if (wave == 1) {
        // ================= producer: staging + output values
        cf32 px[G2_SPT], pg[G2_SPT];
        const int si = lane % G2_TILE, sj0 = lane / G2_TILE;          // sample of the period, first stream of this lane
        constexpr int SSTEP = 64 / G2_TILE;                            // stream stride between a lane's samples
        auto issue = [&](int t) {
#pragma unroll
            for (int q = 0; q < G2_SPT; ++q) {
                const int sj = sj0 + SSTEP * q, idx = t * G2_TILE + si;
                if (idx < s_n[sj]) { px[q] = ldg(s_in[sj] + idx); pg[q] = ldg(s_gp[sj] + idx); }
            }
        };
        auto commit = [&](int t) {
#pragma unroll
            for (int q = 0; q < G2_SPT; ++q) {
                const int sj = sj0 + SSTEP * q, idx = t * G2_TILE + si;
                if (idx < s_n[sj]) {
                    const cf32 z = cmul(cscale(px[q], pg[q].re), phasor_fast(-pg[q].im));   // FastAGC scaling, FreqShift rotation
                    const int slot = (idx + GARDNER_TAPS - 1) & (G2_RING - 1);
                    float* rr = &ring[(sj * 2) * G2_PITCH + slot];
                    rr[0] = z.re; rr[G2_PITCH] = z.im;
                    if (slot < 8) { rr[G2_RING] = z.re; rr[G2_PITCH + G2_RING] = z.im; }
                }
            }
        };
        // values of the outputs the resolver listed in period p: two streams per pass (lanes 0..31 / 32..63), lane = output
        auto produce = [&](int p) {
            const int half = lane >> 5, li = lane & 31;
#pragma unroll 1
            for (int sp = 0; sp < G_SPW; sp += 2) {
                const int sj = sp + half;
                const int cnt = s_cnt[p & 1][sj];
                if (li < cnt) {
                    const uint32_t u = list[p & 1][sj][li];
                    const float* xr = &ring[(sj * 2) * G2_PITCH + (u >> 7)];
                    const float* t = &bank[(u & 127u) * 8];
                    float ar = 0.f, ai = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { ar += xr[k] * t[k]; ai += xr[G2_PITCH + k] * t[k]; }
                    stg(s_out[sj] + s_ostart[p & 1][sj] + li, cf32{ar, ai});
                }
            }
        };
        if (ntiles > 0) { issue(0); commit(0); }
        if (ntiles > 1) issue(1);
        lds_only_barrier();
        if (G2_PROD_PRIO) __builtin_amdgcn_s_setprio(G2_PROD_PRIO);
        for (int t = 0; t < ntiles; ++t) {
            if (!(G2_EXP & 2) && t + 1 < ntiles) commit(t + 1);
            if (t + 2 < ntiles) issue(t + 2);
            if (!(G2_EXP & 1) && t >= 1) produce(t - 1);
#if G2_EXTRA_VALU
            {   // (sensitivity experiment: dead vector instructions per period)
                float dead = (float)t;
#pragma unroll 8
                for (int i = 0; i < G2_EXTRA_VALU; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dead));
                asm volatile("" :: "v"(dead));
            }
#endif
            lds_only_barrier();
        }
        if (!(G2_EXP & 1) && ntiles > 0) produce(ntiles - 1);
        return;
    }

    // ================= resolver
    PclDev pcl{co.g_alpha, co.g_beta, st->g_phase, st->g_freq, co.g_min_freq, co.g_max_freq};
    int offset = st->g_offset, spsctr = st->g_spsctr, outCount = sub ? st->n_fe_out : 0;   // (later slices append to the call's output)
    int cnt = 0;
    uint32_t* lp = nullptr;
    // on-symbol output: interpolate at phase-1 / phase / phase+1, error, advance (gardner.cpp:100-140)
    auto err_step = [&]() {
        int phase = (int)floorf(pcl.phase * 128.0f);
        phase = phase < 0 ? 0 : (phase > 127 ? 127 : phase);
        int my = phase;
        if (arm == 0) my = phase > 0 ? phase - 1 : 0;
        if (arm == 2) my = phase < 127 ? phase + 1 : 127;
        const int slot = offset & (G2_RING - 1);
        const float* xw = row + slot;
        const float* t = &bank[my * 8];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += xw[k] * t[k];
        lp[cnt++] = ((uint32_t)slot << 7) | (uint32_t)phase;
        const float xm = DPP_F(acc, 0x112), xp = DPP_F(acc, 0x102), xo = acc;   // row_shr:2 = the phase-1 arm, row_shl:2 = the phase+1 arm
        float d = (xp - xm) * 0.5f;
        if (__any(phase == 0 || phase == 127)) d = fsel(phase == 0, xp - xo, fsel(phase == 127, xo - xm, d));   // (one-sided at the ends of the bank: rare)
        const float e = (xo > 0 ? 1.0f : -1.0f) * d;                 // valid in lanes r = 2 (re half), 3 (im half)
        const float eo = DPP_F(e, 0xB1);                             // quad_perm [1,0,3,2]: the other half
        const float er = -(e + eo);
        const float eq = DPP_F(er, 0xAA);                            // quad_perm [2,2,2,2]
        const float eh = DPP_F(eq, 0x114);                           // row_shr:4
        float error = r < 4 ? eq : eh;
        error = clamp_med3(error, -1.0f, 1.0f);
        pcl.advance(error);
        const float delta = floorf(pcl.phase);
        offset = (int)((float)offset + delta);
        pcl.phase -= delta;
    };
    // the other output of the symbol: error 0 (gardner.cpp:132-134)
    auto off_step = [&]() {
        int phase = (int)floorf(pcl.phase * 128.0f);
        phase = phase < 0 ? 0 : (phase > 127 ? 127 : phase);
        lp[cnt++] = ((uint32_t)(offset & (G2_RING - 1)) << 7) | (uint32_t)phase;
        pcl.advance(0.0f);
        const float delta = floorf(pcl.phase);
        offset = (int)((float)offset + delta);
        pcl.phase -= delta;
    };
#if G2_FORCE_VGPRS
    asm volatile("" ::: "v123");             // (A/B: the register count the kernel had while its written-out loop named v100..v123)
#endif
    lds_only_barrier();                       // period 0 is staged
    __builtin_amdgcn_s_setprio(G_PRIO);       // latency-critical serial loop (see agc_pc_kernel)
    for (int t = 0; t < ntiles; ++t) {
        if ((((unsigned)t * G2_TILE / G_TILE) & 7u) < (unsigned)(co.g_prio_duty + G_PRIO_DUTY)) __builtin_amdgcn_s_setprio(G_PRIO_HI); else __builtin_amdgcn_s_setprio(G_PRIO_LO);
        const int base = t * G2_TILE;
        const int lim = min(base + G2_TILE, n);            // outputs with offset < lim have their window staged
        lp = &list[t & 1][g][0];
        cnt = 0;
        const int ostart = outCount;
        // (the trip counts are bounded by the list: a poisoned loop state -- NaN input -- can neither hang the GPU nor overrun it)
        if (spsctr == 1 && offset < lim) { off_step(); spsctr = 0; }     // (only where a slice starts between the two outputs of a symbol)
#if S2_G2_ASM
        // THE SYMBOL LOOP WRITTEN OUT (the 4096-stream bank is 512 of these resolver waves: fewer than the GPU has SIMDs -- a wave's time is its instruction count,
        // its LDS round trips and its taken branches; tools/ubench/lone_wave.hip): ~90 instructions and one taken branch per symbol where the compiler's form of
        // err_step() + off_step() has ~140 and four.  Same operations, same order, same roundings: the arm, this lane's 8-tap dot product (products and sums
        // rounded one by one, from 0), the neighbours' arms by DPP, the sign error of the re and im halves, their negated sum handed to the stream's 8 lanes
        // (quad_perm, row_shr:4), clamp, PhaseControlLoop::advance, floor; the follower's list word and its advance by the loop frequency alone (advance(0): freq
        // + beta * 0 and freq + alpha * 0 are freq itself for finite gains and a loop frequency that is not -0, as in the candidate-table form).  A stream whose
        // condition no longer holds drops out of EXEC; a phase at the ends of the bank (one-sided derivative) leaves the loop BEFORE the symbol is touched and the
        // C++ loop underneath takes over.
        {
            static_assert(G2_RING == 64, "the written-out loop masks the slot with 63");
            const uint32_t rowa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)row;
            const uint32_t banka = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)bank;
            const uint32_t lista = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)lp;
            uint32_t lpa = lista + 4u * (uint32_t)cnt;
            const uint32_t lpa_lim = lista + 4u * (uint32_t)(G2_LIST - 2);
            const int lim3 = lim - 3;
            const int d_arm = arm == 0 ? -1 : (arm == 2 ? 1 : 0);
            const uint64_t inmask = __builtin_amdgcn_ballot_w64(spsctr == 0);
            const float alpha_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pcl.alpha)));
            const float beta_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pcl.beta)));
            const float minf_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pcl.minFreq)));
            const float maxf_v = pcl.maxFreq;
            asm volatile(
                "s_mov_b64 s[80:81], exec\n\t"
                "s_and_b64 exec, exec, %[inmask]\n\t"
                "s_mov_b32 s82, 0x43000000\n\t"                                            // 128.0f
                "s_movk_i32 s83, 0x7f\n\t"
                "s_movk_i32 s84, 0x7d\n\t"                                                 // 125
                "1:\n\t"
                "v_cmp_lt_i32 vcc, %[off], %[lim3]\n\t"
                "v_cmp_lt_u32 s[86:87], %[lpa], %[lpalim]\n\t"
                "s_and_b64 vcc, vcc, s[86:87]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 4f\n\t"
                "v_mul_f32 v36, s82, %[ph]\n\t"
                "v_floor_f32 v36, v36\n\t"
                "v_cvt_i32_f32 v36, v36\n\t"
                "v_med3_i32 v36, v36, 0, s83\n\t"                                        // phase
                "v_add_u32 v37, -1, v36\n\t"
                "v_cmp_lt_u32 vcc, s84, v37\n\t"                                          // phase 0 or 127: the one-sided derivative, the general way
                "s_cbranch_vccnz 4f\n\t"
                "v_add_u32 v37, v36, %[darm]\n\t"
                "v_med3_i32 v37, v37, 0, s83\n\t"                                        // this lane's arm
                "v_and_b32 v38, 63, %[off]\n\t"                                           // slot
                "v_lshl_add_u32 v39, v38, 2, %[rowa]\n\t"
                "v_lshl_add_u32 v37, v37, 5, %[banka]\n\t"
                "ds_read_b128 v[40:43], v37\n\t"
                "ds_read_b128 v[44:47], v37 offset:16\n\t"
                "ds_read2_b32 v[48:49], v39 offset1:1\n\t"
                "ds_read2_b32 v[50:51], v39 offset0:2 offset1:3\n\t"
                "ds_read2_b32 v[52:53], v39 offset0:4 offset1:5\n\t"
                "ds_read2_b32 v[54:55], v39 offset0:6 offset1:7\n\t"
                "v_lshl_or_b32 v38, v38, 7, v36\n\t"
                "ds_write_b32 %[lpa], v38\n\t"                                            // list: (slot, phase) of the on-symbol output
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_mul_f32 v56, v48, v40\n\t"
                "v_add_f32 v56, 0, v56\n\t"
                "v_mul_f32 v57, v49, v41\n\t"
                "v_add_f32 v56, v56, v57\n\t"
                "s_waitcnt lgkmcnt(3)\n\t"
                "v_mul_f32 v57, v50, v42\n\t"
                "v_add_f32 v56, v56, v57\n\t"
                "v_mul_f32 v57, v51, v43\n\t"
                "v_add_f32 v56, v56, v57\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "v_mul_f32 v57, v52, v44\n\t"
                "v_add_f32 v56, v56, v57\n\t"
                "v_mul_f32 v57, v53, v45\n\t"
                "v_add_f32 v56, v56, v57\n\t"
                "s_waitcnt lgkmcnt(1)\n\t"
                "v_mul_f32 v57, v54, v46\n\t"
                "v_add_f32 v56, v56, v57\n\t"
                "v_mul_f32 v57, v55, v47\n\t"
                "v_add_f32 v56, v56, v57\n\t"                                           // acc = this lane's interpolant (arm x re / im)
                "v_cvt_f32_i32 v58, %[off]\n\t"
                "s_nop 0\n\t"
                "v_mov_b32_dpp v57, v56 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          // xm: the phase - 1 arm
                "v_sub_f32_dpp v57, v56, v57 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    // xp - xm
                "v_mul_f32 v57, 0.5, v57\n\t"
                "v_cmp_lt_f32 vcc, 0, v56\n\t"
                "v_cndmask_b32_e64 v57, -v57, v57, vcc\n\t"                             // (xo > 0 ? 1 : -1) * d, valid in the lanes of arm 1
                "s_nop 1\n\t"
                "v_add_f32_dpp v57, v57, v57 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   // re half + im half
                "s_nop 1\n\t"
                "v_mov_b32_dpp v59, v57 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "s_nop 1\n\t"
                "v_mov_b32_dpp v57, v59 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_cndmask_b32_e64 v57, v57, v59, %[lo4]\n\t"
                "v_med3_f32 v57, -v57, -1.0, 1.0\n\t"                                    // error = clamp(-(e_re + e_im))
                "v_mul_f32 v59, %[beta], v57\n\t"
                "v_add_f32 %[fr], %[fr], v59\n\t"
                "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"
                "v_mul_f32 v59, %[alpha], v57\n\t"
                "v_add_f32 v59, %[fr], v59\n\t"
                "v_add_f32 %[ph], %[ph], v59\n\t"
                "v_floor_f32 v59, %[ph]\n\t"
                "v_sub_f32 %[ph], %[ph], v59\n\t"
                "v_add_f32 v58, v58, v59\n\t"
                "v_cvt_i32_f32 %[off], v58\n\t"
                // the follower
                "v_mul_f32 v36, s82, %[ph]\n\t"
                "v_floor_f32 v36, v36\n\t"
                "v_cvt_i32_f32 v36, v36\n\t"
                "v_med3_i32 v36, v36, 0, s83\n\t"
                "v_and_b32 v38, 63, %[off]\n\t"
                "v_lshl_or_b32 v38, v38, 7, v36\n\t"
                "ds_write_b32 %[lpa], v38 offset:4\n\t"
                "v_add_u32 %[lpa], 8, %[lpa]\n\t"
                "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"
                "v_add_f32 %[ph], %[ph], %[fr]\n\t"
                "v_floor_f32 v59, %[ph]\n\t"
                "v_cvt_f32_i32 v58, %[off]\n\t"
                "v_sub_f32 %[ph], %[ph], v59\n\t"
                "v_add_f32 v58, v58, v59\n\t"
                "v_cvt_i32_f32 %[off], v58\n\t"
                "s_branch 1b\n\t"
                "4:\n\t"
                "s_mov_b64 exec, s[80:81]\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [ph] "+v"(pcl.phase), [fr] "+v"(pcl.freq), [off] "+v"(offset), [lpa] "+v"(lpa)
                : [lim3] "v"(lim3), [lpalim] "v"(lpa_lim), [darm] "v"(d_arm), [rowa] "v"(rowa), [banka] "v"(banka), [inmask] "s"(inmask),
                  [alpha] "s"(alpha_s), [beta] "s"(beta_s), [minf] "s"(minf_s), [maxf] "v"(maxf_v), [lo4] "s"(0x0F0F0F0F0F0F0F0Full)
                : "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
                  "v54", "v55", "v56", "v57", "v58", "v59", "s80", "s81", "s82", "s83", "s84", "s86", "s87", "vcc", "scc", "memory");
            cnt = (int)((lpa - lista) >> 2);
        }
#endif
        while (__any(spsctr == 0 && offset < lim - 3 && cnt < G2_LIST - 2)) {
            if (spsctr == 0 && offset < lim - 3 && cnt < G2_LIST - 2) { err_step(); off_step(); }
        }
        if (base + G2_TILE >= n) {
            // the stream's last period of this slice: the remaining positions one output at a time
            while (__any(offset < n && cnt < G2_LIST)) {
                if (offset < n && cnt < G2_LIST) {
                    if (spsctr == 0) err_step(); else off_step();
                    spsctr ^= 1;
                }
            }
        }
        outCount += cnt;
        if (r == 0) { s_cnt[t & 1][g] = cnt; s_ostart[t & 1][g] = ostart; }
        lds_only_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    if (act && arm == 0) {
        for (int k = 0; k < GARDNER_TAPS - 1; ++k) {
            const float h = row[(n + k) & (G2_RING - 1)];
            if (c) st->g_hist[k].im = h; else st->g_hist[k].re = h;
        }
        if (c == 0) {
            st->g_phase = pcl.phase; st->g_freq = pcl.freq; st->g_offset = offset - n; st->g_spsctr = spsctr;
            st->n_fe_out = outCount;
            st->n_fe_slice[sub & (S2_FE_MAX_SLICES - 1)] = outCount;
        }
    }
}

// (a third form -- lane = stream, 16 or 64 streams per workgroup: 102 instead of 185 ms of timing recovery per headline step, but the co-resident decoder 342 -> 366 ms and
// no pipelined configuration where it beat the resolver + producer form -- was built in round 4 and deleted in round 5: profiles/r04_gardner_forms_ab.txt)

// ---- timing recovery, fourth form: CANDIDATE TABLES for small banks -------------------------------------------------------
// A small bank is bound by the length of ONE stream's chain, and ~100 of the ~135 instructions per symbol of the forms above are the
// four 8-tap dot products.  Which interpolants the loop will ask for is almost known in advance: an on-symbol output sits at a sample
// offset of the current period, and the polyphase arm moves by alpha * error + (freq - 1) ~ 0.05 arms per symbol.  So two helper waves
// compute, for the period AFTER the one being resolved, the interpolant of EVERY sample offset of that period with the 8 arms around the
// arm the resolver last reported -- the same dot product in the same order as everywhere else, hence the same bits -- into an LDS table.
// The resolver wave (lane = stream) is left with: arm -> three table fetches -> sign error -> advance -> floor, and, as in the second
// form, one 16-bit word (ring slot, arm) per output in an LDS list; it stores nothing itself.  A fetch off the table (the arm ran away,
// the ends of the tap bank) falls back to the dot products themselves, so no result ever depends on the prediction.  The output values
// come from the list one period later (helper wave 2: lane = output, the dot product again -- coalesced stores).
// Workgroup = resolver (GC_CS lanes used) + wave 1 (stages the samples: FastAGC scaling + FreqShift rotation; arms 0..3 of the tables)
// + wave 2 (arms 4..7; output values).  Periods, single steps at the ends of a slice and the state hand-over as in the other forms.
#ifndef S2_GCAND_ASM
#define S2_GCAND_ASM 1    // the resolver's symbol loop of s2_gardner_cand_kernel written out (A/B switch)
#endif
#ifndef GB_PRIO
#define GB_PRIO 2        // wave priority of the resolver (the value the deleted lane-per-stream form was tuned with)
#endif
constexpr int GC_T = 16;                      // samples per stream and period
constexpr int GC_CS = 4;                      // streams per workgroup (64 = GC_CS * GC_T: one staged sample per lane of wave 1)
constexpr int GC_RING = 8 * GC_T;             // ring slots per stream: t-1 (its outputs' values), t (resolved), t+1 (tables), t+2 (staged), slack
constexpr int GC_PITCH = GC_RING + 8;         // + mirror of the first 8 slots: a window never wraps
constexpr int GC_LB = 2;                      // a period's whole symbols start at offsets >= base - GC_LB (the period before ended at >= lim - 2)
constexpr int GC_NR = GC_T + GC_LB;           // table rows: offsets base-2 .. base+T-1 of the period (with 4 more rows the workgroup's LDS would
                                              // exceed the AGC kernel's 18.7 KB: beside a resident decoder workgroup it fits where that one fits)
constexpr int GC_WN = 8;                      // arms per row: (reported arm - 3) .. (reported arm + 4), kept inside 0 .. 127
constexpr int GC_LIST = GC_T + 8;             // outputs of a stream per period (bounded by the loops below)
static_assert(GC_CS * GC_T == 64 && GC_RING == 128, "stager layout; a list word holds slot << 7 | arm in 14 bits");

// pred_skew: added to the reported arm before the table's window is laid around it -- 0 in production; the tests skew the prediction so
// that the resolver leaves its tables (partly, or with every symbol) and has to get the same bits from the fall-back path
__global__ __launch_bounds__(192) void s2_gardner_cand_kernel(const S2StreamWork* __restrict__ work, int nstreams, S2LoopCoefs co,
                                                             const float* __restrict__ bank_g, int sub, int nsub, int pred_skew) {
    __shared__ __attribute__((aligned(16))) float bank[GARDNER_PHASES * GARDNER_TAPS];
    __shared__ __attribute__((aligned(8))) f32x2 ring[GC_CS * GC_PITCH];
    __shared__ __attribute__((aligned(8))) f32x2 tab[2][GC_CS][GC_NR][GC_WN];
    __shared__ __attribute__((aligned(4))) uint16_t list[2][GC_CS][GC_LIST];
    __shared__ int tab_alo[2][GC_CS], s_pred[2][GC_CS], s_cnt[2][GC_CS], s_ostart[2][GC_CS];
    __shared__ const cf32* s_in[GC_CS];
    __shared__ const cf32* s_gp[GC_CS];
    __shared__ cf32* s_out[GC_CS];
    __shared__ int s_n[GC_CS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < GARDNER_PHASES * GARDNER_TAPS; i += 192) bank[i] = bank_g[i];
    const int s = blockIdx.x * GC_CS + lane;
    const bool mine = lane < GC_CS, act = mine && s < nstreams;
    const S2StreamWork w = work[act ? s : 0];
    int lo, hi;
    fe_sub_range(act ? w.count : 0, sub, nsub, lo, hi);
    const int n = hi - lo;
    S2StreamState* st = w.st;
    auto arm_of_phase = [](float ph) {
        // (truncation instead of floor: the same arm for every phase once clamped -- negative products end at 0 either way)
        const int a = (int)(ph * 128.0f);
        return a < 0 ? 0 : (a > 127 ? 127 : a);
    };
    if (wave == 0 && mine) {
        s_in[lane] = n > 0 ? w.in + lo : reinterpret_cast<const cf32*>(bank_g);
        s_gp[lane] = n > 0 ? w.fe_out + fe_scratch_offset(w.count) + lo : reinterpret_cast<const cf32*>(bank_g);
        s_out[lane] = w.fe_out;
        s_n[lane] = n;
        const int a0 = act ? arm_of_phase(st->g_phase) : 0;
        s_pred[0][lane] = a0; s_pred[1][lane] = a0;
        s_cnt[0][lane] = 0; s_cnt[1][lane] = 0;
        if (act) {
            f32x2* row = &ring[lane * GC_PITCH];
            for (int k = 0; k < GARDNER_TAPS - 1; ++k) { const cf32 h = st->g_hist[k]; row[k] = f32x2{h.re, h.im}; row[GC_RING + k] = f32x2{h.re, h.im}; }
        }
    }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int j = 0; j < GC_CS; ++j) nmax = max(nmax, s_n[j]);
    const int ntiles = (nmax + GC_T - 1) / GC_T;
    // re and im of one arm: acc += x[k] * t[k] in tap order, every operation rounded (gardner.cpp / SDR++ polyphase dot product)
    auto dot_arm = [&](const f32x2* x, const float* t) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(t), t1 = *reinterpret_cast<const f32x4*>(t + 4);
        f32x2 acc = f32x2{0.f, 0.f};
        acc += x[0] * t0.x; acc += x[1] * t0.y; acc += x[2] * t0.z; acc += x[3] * t0.w;
        acc += x[4] * t1.x; acc += x[5] * t1.y; acc += x[6] * t1.z; acc += x[7] * t1.w;
        return acc;
    };

    if (wave >= 1) {
        // ================= helper waves
        const int sj = lane >> 4, smp = lane & 15, half = wave - 1;
        const cf32* q_in = s_in[sj];
        const cf32* q_gp = s_gp[sj];
        cf32* q_out = s_out[sj];
        const int q_n = s_n[sj];
        cf32 px, pg;
        auto issue = [&](int t) {
            const int idx = min(t * GC_T + smp, max(q_n - 1, 0));
            px = ldg(q_in + idx); pg = ldg(q_gp + idx);
        };
        auto commit = [&](int t) {
            const int idx = t * GC_T + smp;
            if (idx < q_n) {
                const cf32 z = cmul(cscale(px, pg.re), phasor_fast(-pg.im));   // FastAGC scaling, FreqShift rotation
                const int slot = (idx + GARDNER_TAPS - 1) & (GC_RING - 1);
                f32x2* rr = &ring[sj * GC_PITCH + slot];
                rr[0] = f32x2{z.re, z.im};
                if (slot < 8) rr[GC_RING] = f32x2{z.re, z.im};
            }
        };
        // the table of period t.  Lane = (stream, q): row q with this wave's four arms, then (q < 8) one (row, arm) of the rows 16, 17
        auto cand_row = [&](int t, int a_lo, int r, int j_first, int j_count) {
            const int o = t * GC_T - GC_LB + r;                               // the row's sample offset
            if (o >= 0 && r < GC_NR) {
                const f32x2* xr = &ring[sj * GC_PITCH + (o & (GC_RING - 1))];
                f32x2 x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = xr[i];
#pragma unroll
                for (int j = 0; j < GC_WN / 2; ++j)
                    if (j < j_count) tab[t & 1][sj][r][j_first + j] = dot_arm(x, &bank[(a_lo + j_first + j) * 8]);
            }
        };
        auto candidates = [&](int t, int pred) {
            const int a_lo = min(max(pred + pred_skew - 3, 0), GARDNER_PHASES - GC_WN);   // (no wrap: beside the ends of the bank the resolver computes itself)
            if (smp == 0 && half == 0) tab_alo[t & 1][sj] = a_lo;
            cand_row(t, a_lo, smp, half * (GC_WN / 2), GC_WN / 2);
            cand_row(t, a_lo, GC_T + (smp >> 2), half * (GC_WN / 2) + (smp & 3), 1);       // (rows >= GC_NR: nothing)
        };
        static_assert(GC_NR == GC_T + 2 && GC_WN == 8, "row / arm split of the helper waves");
        // values of the outputs the resolver listed in period p: lane = (stream, output), two passes
        auto produce = [&](int p) {
            const int cnt = s_cnt[p & 1][sj], ostart = s_ostart[p & 1][sj];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int li = smp + 16 * pass;
                if (li < cnt) {
                    const uint32_t u = list[p & 1][sj][li];
                    const f32x2* xr = &ring[sj * GC_PITCH + ((u >> 7) & (GC_RING - 1))];
                    f32x2 x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = xr[i];
                    const f32x2 v = dot_arm(x, &bank[(u & 127u) * 8]);
                    stg(q_out + ostart + li, cf32{v.x, v.y});
                }
            }
        };
        static_assert(GC_LIST <= 32, "two passes of 16 outputs");
        if (half == 0) {
            if (ntiles > 0) { issue(0); commit(0); }
            if (ntiles > 1) { issue(1); commit(1); }
            if (ntiles > 2) issue(2);
        }
        lds_only_barrier();                   // (period 0 is in the ring for both helper waves)
        if (ntiles > 0) candidates(0, s_pred[1][sj]);
        lds_only_barrier();
        for (int t = 0; t < ntiles; ++t) {
            if (half == 0) {
                if (t + 2 < ntiles) commit(t + 2);
                if (t + 3 < ntiles) issue(t + 3);
            }
            if (t + 1 < ntiles) candidates(t + 1, s_pred[(t + 1) & 1][sj]);   // (what the resolver reported at the end of period t-1)
            if (half == 1 && t >= 1) produce(t - 1);
            lds_only_barrier();
        }
        if (half == 1 && ntiles > 0) produce(ntiles - 1);
        return;
    }

    // ================= resolver (lane = stream)
    PclDev pcl{co.g_alpha, co.g_beta, st->g_phase, st->g_freq, co.g_min_freq, co.g_max_freq};
    asm volatile("" : "+v"(pcl.maxFreq));         // (v_med3_f32 takes one scalar operand: the other bound stays in a vector register)
    int offset = st->g_offset, spsctr = st->g_spsctr, outCount = sub ? st->n_fe_out : 0;   // (later slices append to the call's output)
    const f32x2* row = &ring[(mine ? lane : 0) * GC_PITCH];
    const f32x2* const tab0 = &tab[0][0][0][0];
    uint16_t* const list0 = &list[0][0][0];
    int tbi = 0, lbi = 0, a_lo = 0, rowbase = 0;     // the period's table: tab0 + tbi (arms a_lo .. a_lo + 7, rows from offset rowbase), list: list0 + lbi
    int cnt = 0;
    auto arm_of = [&]() { return arm_of_phase(pcl.phase); };
    auto note = [&](int phase) {                     // the output at (offset, phase) joins the period's list
        list0[lbi + cnt] = (uint16_t)((offset << 7) | phase);       // (bits 14, 15: stray offset bits, masked by the reader)
        ++cnt;
    };
    auto finish = [&](float error) {
        pcl.advance(error);
        const float delta = floorf(pcl.phase);
        offset = (int)((float)offset + delta);
        pcl.phase -= delta;
    };
    // PCL::advance(0): freq + beta * 0 and freq + alpha * 0 are freq itself for finite gains (configuration refuses others) and a loop
    // frequency that is not -0 (it lies within [1 - limit, 1 + limit] or is a NaN, which stays one)
    auto finish0 = [&]() {
        pcl.freq = clamp_med3(pcl.freq, pcl.minFreq, pcl.maxFreq);
        pcl.phase += pcl.freq;
        const float delta = floorf(pcl.phase);
        offset = (int)((float)offset + delta);
        pcl.phase -= delta;
    };
    // the dot products themselves (off the table, and the single steps at the ends of a slice): as in s2_gardner_bank_kernel
    auto err_direct = [&](int phase, f32x2& xo, f32x2& d) {
        const int base_arm = min(max(phase - 1, 0), GARDNER_PHASES - 3);       // three consecutive arms that hold what the phase needs
        const f32x2* xr = row + (offset & (GC_RING - 1));
        f32x2 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = xr[k];
        const float* t = &bank[base_arm * 8];
        const f32x2 a = dot_arm(x, t), b = dot_arm(x, t + 8), c = dot_arm(x, t + 16);
        xo = b; d = (c - a) * 0.5f;
        if (phase == 0) { xo = a; d = b - a; }                                  // one-sided at the ends of the bank
        if (phase == GARDNER_PHASES - 1) { xo = c; d = c - b; }
    };
    auto err_finish = [&](f32x2 xo, f32x2 d) {
        // -((xo.re > 0 ? 1 : -1) * d.re + (xo.im > 0 ? 1 : -1) * d.im): a product with +-1 is the operand or its negation
        const float e = (xo.x > 0 ? d.x : -d.x) + (xo.y > 0 ? d.y : -d.y);
        finish(clamp_med3(-e, -1.0f, 1.0f));
    };
    // one whole symbol: on-symbol output (error, advance), then its follower (error 0)
    auto symbol = [&]() {
        const int phase = arm_of();
        const int r = offset - rowbase;                    // (within the table's rows: checked once per period, see below)
        const unsigned j0 = (unsigned)(phase - 1 - a_lo);
        f32x2 xo, d;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(j0 > (unsigned)(GC_WN - 3)) == 0, 1)) {      // every active stream on its table
            const f32x2* e = tab0 + tbi + r * GC_WN + (int)j0;
            const f32x2 a = e[0], c = e[2];
            xo = e[1]; d = (c - a) * 0.5f;                 // (a_lo >= 0 and a_lo + 7 <= 127: phase is neither 0 nor 127 here)
        } else {
            err_direct(phase, xo, d);
        }
        note(phase);
        err_finish(xo, d);
        note(arm_of());
        finish0();
    };
    lds_only_barrier();
    lds_only_barrier();                       // periods 0 and 1 are in the ring, the table of period 0 is made
    __builtin_amdgcn_s_setprio(GB_PRIO);
    for (int t = 0; t < ntiles; ++t) {
        const int base = t * GC_T;
        const int lim = min(base + GC_T, n);              // outputs with offset < lim have their whole window in
        const bool last = base + GC_T >= n;               // this stream's last period of the slice
        a_lo = tab_alo[t & 1][mine ? lane : 0];
        tbi = ((t & 1) * GC_CS + (mine ? lane : 0)) * (GC_NR * GC_WN);
        lbi = ((t & 1) * GC_CS + (mine ? lane : 0)) * GC_LIST;
        rowbase = base - GC_LB;
        cnt = 0;
        // rows: a whole symbol starts at base - 2 <= offset < lim - 2 <= base + T - 2 -- the upper bound is the loop's own condition, the
        // lower one holds because a period ends at offset >= lim - 2; should it not (a period cut short by the list bound: poisoned state),
        // this period's arms are declared off the table
        if (offset < rowbase) a_lo = -1000;
        if (mine) {
            const int ostart = outCount;
            if (spsctr == 1 && offset < lim) { note(arm_of()); finish0(); spsctr = 0; }     // (a slice that starts between the two outputs of a symbol)
            // (trip counts bounded by the list: a poisoned loop state -- NaN input -- can neither hang the GPU nor overrun it)
            if (spsctr == 0) {
                int guard = 0;
#if S2_GCAND_ASM
                // THE SYMBOL LOOP WRITTEN OUT (the resolver is one wave per CU: its time is its instruction count + the table's LDS round trip + its taken
                // branches; tools/ubench/lone_wave.hip): ~52 instructions and one taken branch per symbol instead of ~68 and three.  Every stream (lane) still
                // inside the period takes symbol after symbol off its table -- arm -> three table fetches -> sign error -> advance -> floor; follower:
                // arm -> advance by the loop frequency -> floor; two list words -- with the operations and roundings of symbol() below.  A stream that has
                // finished the period drops out of EXEC; an arm off the table (ballot, as below) leaves the loop BEFORE that symbol is touched, and the
                // C++ loop underneath finishes the period the general way with what is left of the trip budget.
                {
                    const uint32_t tab_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) f32x2*)tab0;
                    const uint32_t list_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t*)list0;
                    const uint32_t tadr = tab_a + 8u * (uint32_t)tbi - 64u * (uint32_t)rowbase;          // + 64 offset + 8 (arm - 1 - a_lo): the entry of arm - 1
                    uint32_t la = list_a + 2u * (uint32_t)(lbi + cnt);
                    const int alo1 = a_lo + 1;
                    uint32_t budget = (GC_LIST - 4) / 2;
                    const int lim2 = lim - 2;                       // (per stream: a slice ends where the stream's samples end)
                    const float alpha_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pcl.alpha)));
                    const float beta_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pcl.beta)));
                    const float minf_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pcl.minFreq)));
                    asm volatile(
                        "s_mov_b64 s[80:81], exec\n\t"
                        "s_mov_b32 s82, 0x43000000\n\t"                                            // 128.0f
                        "s_movk_i32 s83, 0x7f\n\t"
                        "1:\n\t"
                        "v_cmp_lt_i32 vcc, %[off], %[lim2]\n\t"                                    // this stream's next symbol starts inside the period
                        "s_and_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execz 4f\n\t"
                        "v_mul_f32 v100, s82, %[ph]\n\t"
                        "v_cvt_i32_f32 v100, v100\n\t"
                        "v_med3_i32 v100, v100, 0, s83\n\t"                                        // the arm
                        "v_sub_u32 v101, v100, %[alo1]\n\t"
                        "v_cmp_lt_u32 vcc, 5, v101\n\t"                                            // arm - 1 .. arm + 1 not all on the table (GC_WN - 3)
                        "s_cbranch_vccnz 4f\n\t"
                        "v_lshl_add_u32 v102, %[off], 6, %[tadr]\n\t"
                        "v_lshl_add_u32 v102, v101, 3, v102\n\t"
                        "ds_read2_b64 v[104:107], v102 offset1:2\n\t"                              // arm - 1, arm + 1
                        "ds_read_b64 v[108:109], v102 offset:8\n\t"                                // arm
                        "v_lshl_or_b32 v103, %[off], 7, v100\n\t"
                        "ds_write_b16 %[la], v103\n\t"                                            // list: (offset, arm) of the on-symbol output
                        "s_waitcnt lgkmcnt(1)\n\t"
                        "v_pk_add_f32 v[104:105], v[106:107], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                        "v_pk_mul_f32 v[104:105], v[104:105], 0.5 op_sel_hi:[1,0]\n\t"            // d
                        "v_cmp_lt_f32 vcc, 0, v108\n\t"
                        "v_cndmask_b32_e64 v106, -v104, v104, vcc\n\t"
                        "v_cmp_lt_f32 vcc, 0, v109\n\t"
                        "v_cndmask_b32_e64 v107, -v105, v105, vcc\n\t"
                        "v_add_f32 v106, v106, v107\n\t"
                        "v_med3_f32 v106, -v106, -1.0, 1.0\n\t"                                   // error
                        "v_mul_f32 v107, %[beta], v106\n\t"
                        "v_add_f32 %[fr], %[fr], v107\n\t"
                        "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"
                        "v_mul_f32 v107, %[alpha], v106\n\t"
                        "v_add_f32 v107, %[fr], v107\n\t"
                        "v_add_f32 %[ph], %[ph], v107\n\t"
                        "v_floor_f32 v107, %[ph]\n\t"
                        "v_cvt_f32_i32 v106, %[off]\n\t"
                        "v_sub_f32 %[ph], %[ph], v107\n\t"
                        "v_add_f32 v106, v106, v107\n\t"
                        "v_cvt_i32_f32 %[off], v106\n\t"
                        // the follower: its list word, then the advance by the loop frequency alone
                        "v_mul_f32 v100, s82, %[ph]\n\t"
                        "v_cvt_i32_f32 v100, v100\n\t"
                        "v_med3_i32 v100, v100, 0, s83\n\t"
                        "v_lshl_or_b32 v103, %[off], 7, v100\n\t"
                        "ds_write_b16 %[la], v103 offset:2\n\t"
                        "v_add_u32 %[la], 4, %[la]\n\t"
                        "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"
                        "v_add_f32 %[ph], %[ph], %[fr]\n\t"
                        "v_floor_f32 v107, %[ph]\n\t"
                        "v_cvt_f32_i32 v106, %[off]\n\t"
                        "v_sub_f32 %[ph], %[ph], v107\n\t"
                        "v_add_f32 v106, v106, v107\n\t"
                        "v_cvt_i32_f32 %[off], v106\n\t"
                        "s_sub_u32 %[bud], %[bud], 1\n\t"
                        "s_cmp_lg_u32 %[bud], 0\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "4:\n\t"
                        "s_mov_b64 exec, s[80:81]\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : [ph] "+v"(pcl.phase), [fr] "+v"(pcl.freq), [off] "+v"(offset), [la] "+v"(la), [bud] "+s"(budget)
                        : [tadr] "v"(tadr), [alo1] "v"(alo1), [lim2] "v"(lim2), [alpha] "s"(alpha_s), [beta] "s"(beta_s), [minf] "s"(minf_s), [maxf] "v"(pcl.maxFreq)
                        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "s80", "s81", "s82", "s83", "vcc", "scc", "memory");
                    cnt = (int)((la - list_a) >> 1) - lbi;
                    guard = (GC_LIST - 4) / 2 - (int)budget;
                }
#endif
                for (; guard < (GC_LIST - 4) / 2; ++guard) {
                    const bool go = offset < lim - 2;
                    if (__builtin_amdgcn_ballot_w64(go) == 0) break;
                    if (go) symbol();
                }
            }
            if (last)
                for (int guard = 0; guard < 8 && offset < n && cnt < GC_LIST; ++guard) {
                    const int phase = arm_of();
                    note(phase);
                    if (spsctr == 0) {
                        f32x2 xo, d;
                        err_direct(phase, xo, d);
                        err_finish(xo, d);
                    } else {
                        finish0();
                    }
                    spsctr ^= 1;
                }
            outCount += cnt;
            s_cnt[t & 1][lane] = cnt; s_ostart[t & 1][lane] = ostart;
            s_pred[t & 1][lane] = arm_of();
        }
        lds_only_barrier();
    }
    __builtin_amdgcn_s_setprio(0);
    if (act) {
        for (int k = 0; k < GARDNER_TAPS - 1; ++k) { const f32x2 h = row[(n + k) & (GC_RING - 1)]; st->g_hist[k] = cf32{h.x, h.y}; }
        st->g_phase = pcl.phase; st->g_freq = pcl.freq; st->g_offset = offset - n; st->g_spsctr = spsctr;
        st->n_fe_out = outCount;
        st->n_fe_slice[sub & (S2_FE_MAX_SLICES - 1)] = outCount;
    }
}

// ------------------------------------------------------------------------------------------------ RRC + /2
// grid (x: symbol tiles, y: stream).  Only the samples the decimator keeps are filtered.
// sub / nsub: only the symbols whose windows the timing recovery's slice `sub` completed (the decimator phase and the delay line are
// the call's: they change in s2_rrc_state_kernel, after the last slice); nsub == 1: the whole call
__global__ __launch_bounds__(256) void s2_rrc_decim_kernel(const S2StreamWork* __restrict__ work, const float* __restrict__ taps_g, int ntaps, int sub, int nsub) {
    // (bits 16.. of nsub: the wave priority the host asks for -- POST_PRIO where its balancer has found the front end critical, s2_demod.hip)
    if (nsub >> 16) __builtin_amdgcn_s_setprio(POST_PRIO);
    nsub &= 0xffff;
    // a block = 256 consecutive kept symbols of one stream; their 2*256 + ntaps - 2 input samples go through LDS once (each is
    // used by up to (ntaps+1)/2 outputs: reading them from L2 per output cost 46 GB per step and competed with the LDPC messages).
    // The window lies DE-INTERLEAVED in LDS -- even samples in xe, odd ones in xo, re and im side by side: output t reads xe[t + k/2] / xo[t + k/2] for
    // tap k, 8 bytes per lane at consecutive addresses (one conflict-free ds_read_b64 per tap, where separate re / im arrays read at a stride of two
    // dwords cost two 2-way-conflicted reads), and re and im go through ONE packed multiplication and ONE packed addition per tap -- the products and
    // sums of the scalar form (tap order, every operation rounded).  The taps come through the scalar cache.  10 LDS cycles per tap and wave became 2;
    // alone the launch went from 1.29 to 1.10 ms per 4096 x 21 690 symbols -- 2.1 GB moved in that time: the rest is the memory system's.
    __shared__ f32x2 xe[256 + RRC_MAX_TAPS / 2 + 2], xo[256 + RRC_MAX_TAPS / 2 + 2];
    const S2StreamWork w = work[blockIdx.y];
    S2StreamState* st = w.st;
    const int n = nsub > 1 ? st->n_fe_slice[sub] : st->n_fe_out, n_before = (nsub > 1 && sub) ? st->n_fe_slice[sub - 1] : 0;
    const int first = st->cr_samp ? 0 : 1;            // first kept index (module_dvbs2_demod.cpp:231-239)
    const int nsym = n > first ? (n - first + 1) / 2 : 0;
    const int sym_before = n_before > first ? (n_before - first + 1) / 2 : 0;
    const int H = ntaps - 1;
    if (nsub > 1 && blockIdx.x == 0 && threadIdx.x == 0) st->n_sym_slice[sub] = nsym;
    for (int m0 = sym_before + blockIdx.x * 256; m0 < nsym; m0 += gridDim.x * 256) {
        const int i0 = 2 * m0 + first;                // index of the block's first window in [history(H) ++ fe_out]
        const int cnt = min(256, nsym - m0);
        const int need = 2 * (cnt - 1) + ntaps;
        __syncthreads();
        for (int q = threadIdx.x; q < need; q += 256) {
            const int p = i0 + q;
            const cf32 v = p < H ? st->rrc_hist[p] : w.fe_out[p - H];
            ((q & 1) ? xo : xe)[q >> 1] = f32x2{v.re, v.im};
        }
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const f32x2* pe = xe + threadIdx.x;
            const f32x2* po = xo + threadIdx.x;
            f32x2 acc{0.f, 0.f};
            int k = 0;
            for (; k + 1 < ntaps; k += 2) {
                acc += pe[k >> 1] * taps_g[k];
                acc += po[k >> 1] * taps_g[k + 1];
            }
            if (k < ntaps) acc += pe[k >> 1] * taps_g[k];
            w.fifo[w.fifo_fill + m0 + threadIdx.x] = cf32{acc.x, acc.y};
        }
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
    // (int)(((double)v / 1.5) * 256 + 128) exactly: the product form is within 1e-12 of it, so it truncates to the same integer
    // unless it lands that close to one -- only then pay for the double division
    const double y = (double)v * 170.66666666666666 + 128.0;
    int x = (int)y;
    if (__builtin_fabs(y - __builtin_rint(y)) < 1e-9) x = (int)(((double)v / 1.5) * 256 + 128);
    // (a value beyond the int range -- |v| > 1.2e7, +inf -- converts to INT_MIN on x86-64 and ends at cell 0 there; the device's conversion
    //  saturates: the x86 outcome is the reference's, tests/test_det_math.py)
    if (!(y < 2147483648.0)) x = 0;
    return x < 0 ? 0 : (x > 255 ? 255 : x);
}
// Both table indices of a sample at once, as re_index * 256 + im_index.  The binary32 product-sum y32 = fma(v, 256/1.5, 128) is within
// 2e-5 of the reference's double expression wherever the clamp does not decide anyway (|v| <= 0.75: 1.5e-5 of rounding at 256, 4e-6 from
// the constant), so it truncates to the same integer unless it lies that close to one: only then (2.5e-4 either side, 1 sample in 1000)
// the double form above is evaluated.  ~14 instructions for the pair instead of ~30 -- this sits in the per-symbol chain of the PLL.
__device__ __forceinline__ int lut_cell(float re, float im) {
    const float K = 170.66666666666666f;
    const float yr = __builtin_fmaf(re, K, 128.0f), yi = __builtin_fmaf(im, K, 128.0f);
    const float dr = yr - __builtin_rintf(yr), di = yi - __builtin_rintf(yi);
    int xr = (int)yr, xi = (int)yi;                    // (NaN -> 0, +-inf saturate: what the double form gives after its clamp)
    xr = max(0, min(255, xr)); xi = max(0, min(255, xi));
    const bool near = !(__builtin_fabsf(dr) >= 2.5e-4f && __builtin_fabsf(di) >= 2.5e-4f);      // (NaN differences -- infinite or NaN samples -- count as near)
    if (__builtin_expect(__any(near), 0)) {             // (uniform branch, its body out of line: a taken branch costs ~28 cycles here)
        if (near) { xr = lut_index(re); xi = lut_index(im); }
    }
    return xr * 256 + xi;
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
        float dd = dvbs2m::expf_det(-dist / 1.0f);   // the shared definition: fp64 evaluation, one rounding, subnormal results kept
        for (int j = 0; j < C.bits; j++) {
            if (((i >> j) & 1) == 0) tmp[2 * j + 0] += dd;
            else tmp[2 * j + 1] += dd;
        }
    }
    if (bits_out)
        for (int i = 0; i < C.bits; i++) {
            float x = (dvbs2m::logf_det(tmp[2 * i + 1]) - dvbs2m::logf_det(tmp[2 * i + 0])) * C.sca;
            bits_out[C.bits - 1 - i] = dvbs2m::llr_clamp_det(x);
        }
    if (phase_err) *phase_err = cphase(cmul(sample, cconj(closest)));
}

// Phase error only (constellation_t::demod_soft_calc's closest-point search + phase), for the PLL of the frame loops with 32APSK.  NOT inlined and
// fed with scalars + a copy of the points in LDS: the 32APSK path would otherwise set the register budget of the serial frame-loop kernel, which
// must stay under 128 VGPRs to share a SIMD with three resident LDPC waves in the pipelined mode (s2_demod.hip).
// The GSZ lanes of a stream's lane group all run the loop's chain with the same values, so they SHARE the search: lane gl takes the points
// gl, gl + GSZ, ...; the group then reduces (distance, index) lexicographically -- the reference's "first point of strictly smallest distance".
// (One lane scanning 32 points read from global memory cost 4.4 us per symbol: config 5's frame loops took 242 ms per step, now 45.)
typedef __attribute__((address_space(3))) const float lds_cf32;   // (interleaved re, im)
template <int GSZ>
__device__ __attribute__((noinline)) float soft_phase_err_group(lds_cf32* __restrict__ pts, int states, float amp, float prescale, cf32 sample, int gl) {
    if (amp != 1) sample = cscale(sample, amp);
    if (prescale != 1) sample = cscale(sample, prescale);
    float bd = 3.402823466e+38f;
    int bi = 64;
    for (int i = gl; i < states; i += GSZ) {
        const cf32 p{pts[2 * i], pts[2 * i + 1]};
        const float dist = camp(csub(sample, p));
        if (dist < bd) { bd = dist; bi = i; }                 // strict: the lowest index among this lane's equals stays
    }
    if constexpr (GSZ == 8) {
        // the (distance, index) minimum over the 8 lanes of the group by DPP exchanges (quad, quad, the other quad mirrored): a shuffle
        // through the LDS crossbar costs a round trip per step, and this sits in the per-symbol chain of the PLL
#define PE_JOIN(ctrl) do { const float od = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bd), (ctrl), 0xf, 0xf, false)); \
                           const int oi = __builtin_amdgcn_update_dpp(0, bi, (ctrl), 0xf, 0xf, false); \
                           if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; } } while (0)
        PE_JOIN(0xB1);      // quad_perm [1,0,3,2]
        PE_JOIN(0x4E);      // quad_perm [2,3,0,1]
        PE_JOIN(0x141);     // row_half_mirror
#undef PE_JOIN
    } else {
#pragma unroll
        for (int o = GSZ / 2; o > 0; o >>= 1) {
            const float od = __shfl_xor(bd, o);
            const int oi = __shfl_xor(bi, o);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
    }
    cf32 closest{0.f, 0.f};
    if (bi < 64) closest = cf32{pts[2 * bi], pts[2 * bi + 1]};
    return cphase(cmul(sample, cconj(closest)));
}

// ONE WAVE PER STREAM (see the front end): the PLL / PLHDR phase recurrences are serial per symbol and run as
// uniform code; lanes do the coalesced symbol loads/stores, the FED terms and the PLSC codeword search.
// EIGHT LANES PER STREAM, 8 streams per wave.  The FED / PLL / PLHDR recurrences are serial per stream and the phase-error
// LUT lookup sits in the PLL's chain, so a stream is latency-bound whatever the lane count; the first version (one wave per
// stream, 64 lanes running the same chain) was issue-bound with 4 waves per SIMD instead, and every redundant lane group also
// takes issue slots from the LDPC decoder that shares the SIMDs in the pipelined mode (16 lanes per stream: 28 ms beside it,
// LDPC 74 ms).  Groups whose stream has fewer frames in this call shadow a frame of another group (same code path, nothing
// stored, state restored afterwards).
// (tried in round 3 for small banks: the PLL's phase-error table folded onto one quadrant -- 8PSK's table is bit for bit invariant under
// 90-degree turns -- and kept in LDS, 66 KB: no gain.  The lookups of one stream mostly hit the CU's vector L1 (~130 cycles, tools/ubench/
// chase.hip), and the fold's index arithmetic costs what the LDS fetch saves: 459 against 463 cycles per symbol in the plain PLL loop.)
#ifndef FL_LPS_N
#define FL_LPS_N 8
#endif
#ifndef S2_PLL_TILES
#define S2_PLL_TILES 1   // one stream per workgroup: a payload tile of the PLL as a fixed point of (parallel phase-error evaluation, serial replay) instead of the serial loop (A/B switch)
#endif
#ifndef S2_PLL_ASM
#define S2_PLL_ASM 1     // the PLL's payload-tile loop of s2_frame_loops_kernel written out (A/B switch)
#endif
constexpr int FL_LPS = FL_LPS_N;
constexpr int FL_SPW = 64 / FL_LPS;
#ifndef FL_SMALL_BANK
#define FL_SMALL_BANK 256
#endif
static inline int frame_loops_spw(int nstreams) { return nstreams <= FL_SMALL_BANK ? 1 : FL_SPW; }      // streams per workgroup of s2_frame_loops_kernel (a CU each up to 256 streams)
constexpr int FL_TILE = 46;          // PLL symbols staged per round (>= 36: the FED reuses the tile for a pilot block; input + output
                                     // tile together hold the 90 header symbols for the PLHDR loop and the 88 FED terms); small: THREE
                                     // workgroups have to fit into the 24 KB of LDS an LDPC workgroup leaves free -- a mixed batch launches
                                     // one kernel per configuration group, and with two 11.7 KB workgroups per CU plus a stray third the
                                     // decoder workgroup of that CU could not start before the frame loops had finished (pipelined mode)

// A PAYLOAD TILE OF THE PLL AS A FIXED POINT (one stream per wave; s2_frame_loops_kernel with one stream per workgroup, s2_vcm_loops_kernel).
// THE TILE AS A FIXED POINT (oracle: S2Rx::pll_tile_study; profiles/r05_pll_tile_study.txt).  What a symbol contributes to the
// loop, e[k], depends on the loop phase at that symbol only (rotation, table cell); the recurrence proper (dvbs2_pll.cpp:81) is a dozen instructions.  So: lane k
// evaluates e[k] from a guessed phase (first the phase at the tile's start advanced by the frequency alone) -- all symbols of the tile side by side, one table
// round trip for the lot -- then every lane replays the recurrence over the tile with those e[k] in the reference's order (uniform values: the errors come out
// of the lanes through v_readlane, lane k keeps the phase of step k), the lanes evaluate again at the replayed phases ... until no symbol's table cell changes.
// By induction over k that fixed point IS the serial result: phase[0] is exact; an exact phase[k] gives the exact e[k], hence the exact phase[k + 1].  A replay
// starts at the first symbol whose cell changed (everything before it stands).  Same functions as the serial form below: bit-identical.
// `sym` = lane k's symbol (k < m), `lut` the constellation's phase-error table; returns lane k's rotated symbol (what the serial loop stores as tmp_val), the loop state in `pll` advanced by m symbols.
__device__ __forceinline__ cf32 pll_payload_tile(const cf32 sym, const bool mine_k, const int lane, const int m, PclDev& pll, const float* lut_err_v) {
    // (the loop state of lane 0: the lane groups without a stream are restored to their frame's start after every frame and would replay from there)
    const float ph0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pll.phase)));
    const float fq0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pll.freq)));
    float myph = ph0 + (float)lane * fq0, myfq = fq0, mye = 0.f;
    int mycell = -1;
    cf32 tmp_val{0.f, 0.f};
    PclDev r = pll;
    for (;;) {
        bool changed = false;
        if (mine_k) {
            tmp_val = cmul(sym, phasor_fast(-myph));
            const int cell = lut_cell(tmp_val.re, tmp_val.im);
            changed = cell != mycell;
            mycell = cell;
        }
        const unsigned long long chg = __ballot(changed);
        if (chg == 0) break;                                       // every error reproduced: the phases are the serial loop's
        if (changed) mye = as_global(lut_err_v)[mycell];
        const int c = __ffsll((long long)chg) - 1;                 // the first symbol whose error changed: the replay starts there
        if (c > 0) {
            r.phase = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myph), c));
            r.freq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myfq), c));
        } else { r.phase = ph0; r.freq = fq0; }
#if S2_PLL_ASM
        {
            // the replay written out (a lone wave's time is its instruction count, tools/ubench/lone_wave.hip): per symbol the error out of lane k (v_readlane), lane k
            // keeps the state the step starts from, PhaseControlLoop::advance and one wrap into [-pi, pi] -- the same operations in the same order as the serial loop's
            // tail below (multiplication and addition apart: no fused multiply-add); 11 vector instructions, four steps per trip of the loop (a taken branch costs ~28 cycles)
            uint32_t kq = (uint32_t)__builtin_amdgcn_readfirstlane(c), mq = (uint32_t)__builtin_amdgcn_readfirstlane(m);
            const float minf_q = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pll.minFreq)));
            float alpha_v = pll.alpha, beta_v = pll.beta, maxf_v = pll.maxFreq;
            asm volatile("" : "+v"(alpha_v), "+v"(beta_v), "+v"(maxf_v));
#define S2_REPLAY_STEP(W, B)                                                    \
                "v_readlane_b32 s60, %[mye], %[k]\n\t"            \
                "v_cmp_eq_u32 vcc, %[k], %[lane]\n\t"             \
                "v_cndmask_b32 %[myph], %[myph], %[ph], vcc\n\t"  \
                "v_cndmask_b32 %[myfq], %[myfq], %[fr], vcc\n\t"  \
                "v_mul_f32 v124, s60, %[beta]\n\t"                \
                "v_add_f32 %[fr], %[fr], v124\n\t"                \
                "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"   \
                "v_mul_f32 v124, s60, %[alpha]\n\t"               \
                "v_add_f32 v124, %[fr], v124\n\t"                 \
                "v_add_f32 %[ph], %[ph], v124\n\t"                \
                "v_cmp_gt_f32 vcc, |%[ph]|, s71\n\t"              \
                "s_cbranch_vccnz " W "f\n\t"                      \
                B ":\n\t"                                         \
                "s_add_u32 %[k], %[k], 1\n\t"                     \
                "s_cmp_lt_u32 %[k], %[m]\n\t"
            // (the wrap into [-pi, pi] out of line: |phase| <= pi is the rule, and a branch not taken is nearly free)
#define S2_REPLAY_WRAP(W, B)                                                    \
                W ":\n\t"                                         \
                "v_add_f32 v124, s74, %[ph]\n\t"                  \
                "v_cmp_lt_f32 vcc, s71, %[ph]\n\t"                \
                "v_cndmask_b32 %[ph], %[ph], v124, vcc\n\t"       \
                "v_add_f32 v124, s73, %[ph]\n\t"                  \
                "v_cmp_gt_f32 vcc, s72, %[ph]\n\t"                \
                "v_cndmask_b32 %[ph], %[ph], v124, vcc\n\t"       \
                "s_branch " B "b\n\t"
            asm volatile(
                "s_mov_b32 s71, 0x40490fdb\n\t"          // pi
                "s_mov_b32 s72, 0xc0490fdb\n\t"          // -pi
                "s_mov_b32 s73, 0x40c90fdb\n\t"          // 2 pi
                "s_mov_b32 s74, 0xc0c90fdb\n\t"          // -2 pi
                "s_cmp_lt_u32 %[k], %[m]\n\t"
                "s_cbranch_scc0 2f\n\t"
                "1:\n\t"
                S2_REPLAY_STEP("11", "21") "s_cbranch_scc0 2f\n\t"
                S2_REPLAY_STEP("12", "22") "s_cbranch_scc0 2f\n\t"
                S2_REPLAY_STEP("13", "23") "s_cbranch_scc0 2f\n\t"
                S2_REPLAY_STEP("14", "24") "s_cbranch_scc1 1b\n\t"
                "s_branch 2f\n\t"
                S2_REPLAY_WRAP("11", "21") S2_REPLAY_WRAP("12", "22") S2_REPLAY_WRAP("13", "23") S2_REPLAY_WRAP("14", "24")
                "2:"
                : [ph] "+v"(r.phase), [fr] "+v"(r.freq), [myph] "+v"(myph), [myfq] "+v"(myfq), [k] "+s"(kq)
                : [mye] "v"(mye), [lane] "v"(lane), [alpha] "v"(alpha_v), [beta] "v"(beta_v), [minf] "s"(minf_q), [maxf] "v"(maxf_v), [m] "s"(mq)
                : "v124", "s60", "s71", "s72", "s73", "s74", "vcc", "scc");
#undef S2_REPLAY_WRAP
#undef S2_REPLAY_STEP
        }
#else
        for (int kk = c; kk < m; ++kk) {
            const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mye), kk));
            const bool me = lane == kk;
            myph = me ? r.phase : myph;
            myfq = me ? r.freq : myfq;
            r.advance(e);
            r.wrap_pi_once();
        }
#endif
    }
    pll.phase = r.phase; pll.freq = r.freq;
    return tmp_val;
}

// SPEC: with the loops ahead of the PL sync (below) compiled in -- small banks only: the plain instantiation has to stay within 128 registers
// (it shares its SIMDs with three decoder waves in the pipelined mode), and the extra state costs it 14
#ifndef FL_WPB_MAX_N
#define FL_WPB_MAX_N 1      // waves per workgroup of a big bank's frame loops (A/B switch).  Round 6, same call: 2 -> decoder in the step 247.8 -> 245.8 ms but frame loops 108-115 -> 123-132 and the step 264.4 -> 271.7: a two-wave workgroup needs two free 128-register places on one compute unit at once, and beside the decoder every SIMD has ONE
#endif
constexpr int FL_WPB_MAX = FL_WPB_MAX_N;
// a bank of at least one single-wave workgroup per compute unit (256) goes out in workgroups of FL_WPB_MAX waves; smaller ones keep a workgroup -- a compute unit -- per wave
static inline int frame_loops_wpb(int nstreams, int spw) { const int nb = (nstreams + spw - 1) / spw; return nb >= 256 ? FL_WPB_MAX : 1; }
static inline int frame_loops_grid(int nstreams, int spw) { const int nb = (nstreams + spw - 1) / spw, w = frame_loops_wpb(nstreams, spw); return (nb + w - 1) / w; }
// what __syncthreads() is in a single-wave workgroup (the compiler drops its s_barrier there): a wave's LDS writes before its LDS reads -- the waves of these workgroups share nothing
#define FL_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
template <bool SPEC>
__global__ __launch_bounds__(64 * FL_WPB_MAX) __attribute__((amdgpu_waves_per_eu(4, 4))) void s2_frame_loops_kernel(const S2StreamWork* __restrict__ work, int nstreams,
                                                            const S2FrameRef* __restrict__ frames, const int* __restrict__ first,
                                                            S2LoopCoefs co, S2PlTablesDev T, S2ConstelDev C, int pls_code, int slots,
                                                            int pilots, int pilot_blocks, int plframe, cf32* __restrict__ pllout,
                                                            S2FrameStats* __restrict__ stats, const S2VcmFound* __restrict__ found, int maxf, int spw,
                                                            const S2StreamCfgDev* __restrict__ cfgs) {
    // spw: streams per workgroup, 1 .. FL_SPW.  A small bank gets a workgroup (and with it a CU's vector L1 for its phase-error table lookups)
    // per stream: 64 streams x 1 frame 6.1 -> 5.x ms; the lane groups without a stream shadow the others' code path as usual
    // found != nullptr (stage pipeline): no pooled frame table -- frame k of stream s is slot s * maxf + k of found / pllout / stats, and
    // this launch goes through the frames the PL-sync walk has found (walk_nf) beyond those an earlier slice's launch did (loops_done)
    static_assert(2 * FL_TILE >= 90 && 2 * FL_TILE >= 88 && FL_TILE >= 36, "input + output tile hold the 90 header symbols; the output tile alone the 88 FED terms");
    // mixed batch: this workgroup's streams share the configuration of its first one (s2_demod.hip sees to that); `plframe` as passed is then the
    // stride of the PLL-output slots (the longest PLFRAME of the batch)
    const int slot_stride = plframe;
    // WAVES of a workgroup are independent of one another (a big bank is launched with FL_WPB_MAX of them per workgroup for the sake of their PLACEMENT: the hardware spreads the
    // waves of one workgroup over the SIMDs of its compute unit, single-wave workgroups land where they land -- tools/ubench/placement.hip); wave `wv` of workgroup b is what
    // the single-wave workgroup b * waves + wv was: its own streams, its own tiles, no workgroup barrier anywhere (FL_SYNC orders a wave's own LDS traffic)
    const int wv = threadIdx.x >> 6, vb = (int)blockIdx.x * ((int)blockDim.x >> 6) + wv;
    if (vb * spw >= nstreams) return;
    if (cfgs) {
        const S2StreamCfgDev* __restrict__ q = cfgs + min(vb * spw, nstreams - 1);
        C = q->con; pls_code = q->pls_code; slots = q->slots; pilots = q->pilots; pilot_blocks = q->pilot_blocks; plframe = q->plframe;
    }
    __shared__ cf32 tiles_all[FL_WPB_MAX][FL_SPW][2 * FL_TILE]; // per wave and stream: [input tile | output tile]
    __shared__ uint8_t rnt_all[FL_WPB_MAX][FL_TILE];
    __shared__ cf32 s_pts_all[FL_WPB_MAX][32];                 // constellation points for the 32APSK phase-error search (the other constellations use the LUT)
    cf32 (*const tiles)[2 * FL_TILE] = tiles_all[wv];
    uint8_t* const rnt = rnt_all[wv];
    cf32* const s_pts = s_pts_all[wv];
    const int lane = threadIdx.x & 63, g = lane / FL_LPS, gl = lane % FL_LPS;
    if (C.bits == 5 && lane < 32) s_pts[lane] = lane < C.states ? C.pts_g[lane] : cf32{0.f, 0.f};
    cf32* const tl = &tiles[g][0];                      // input tile (also a 36-symbol pilot block for the FED; with the output tile: the 90 header symbols)
    cf32* const ot = &tiles[g][FL_TILE];                // output tile
    float* const fd = reinterpret_cast<float*>(ot);     // FED terms live in the (then unused) output tile
    const int s0 = vb * spw, s = s0 + g;
    const bool act = g < spw && s < nstreams;
    const int sc = act ? s : s0;
    S2StreamState* st = work[sc].st;
    PclDev pll{co.pll_alpha, co.pll_beta, st->pll_phase, st->pll_freq, co.pll_min_freq, co.pll_max_freq};
    PclDev hdr{co.hdr_alpha, co.hdr_beta, st->hdr_phase, st->hdr_freq, co.hdr_min_freq, co.hdr_max_freq};
    float nco_freq = st->nco_freq;
    const float PI_F = 3.14159265358979323846f;
    const cf32* __restrict__ plsc = T.plsc + (size_t)pls_code * 64;
    const int done = found ? st->loops_done : 0;
    const int f0 = found ? sc * maxf + done : first[sc], nf = !act ? 0 : (found ? st->walk_nf - done : first[sc + 1] - f0);
    // ---- AHEAD OF THE PL SYNC (spec; one stream per workgroup, stage pipeline).  A frame only exists once the PL sync has seen its whole
    // window -- until then a call of F frames has its frame loops waiting for symbols F - 1 times and working off a whole frame after the
    // last symbol has arrived.  In lock the next window IS the next frame, so this kernel (launched behind every time slice) runs the PLL
    // over the whole tiles of the window the FIFO holds so far, into the slot the frame will get; FED, PLHDR demod and statistics wait for
    // the confirmation.  The loop state saved at the window's start comes back if the walk decides otherwise, and at the next call.
    const int n_tiles = (plframe + FL_TILE - 1) / FL_TILE;
    const bool spec_on_here = SPEC && found && spw == 1 && !co.pilot_aided && work[sc].spec_out != nullptr;
    bool resume = false, sp_go = false, sp_keep = false;
    int sp_t0 = 0, sp_t1 = 0, sp_so = 0, sp_tiles = 0, sp_slot = 0, sp_carried = 0;
    float sp_ph0 = 0.f, sp_fr0 = 0.f;
    if (spec_on_here && act) {
        const int on = st->spec_on, off = st->spec_off;
        sp_tiles = st->spec_tiles; sp_ph0 = st->spec_phase0; sp_fr0 = st->spec_freq0; sp_carried = st->spec_carried;
        sp_so = st->walk_cur + st->pl_pending;
        sp_slot = st->walk_nf;
        bool cont = false;
        if (on) {
            if (nf > 0) { if (found[f0].offset == off) resume = true; else { pll.phase = sp_ph0; pll.freq = sp_fr0; } }
            else if (sp_so == off) cont = true;
            else { pll.phase = sp_ph0; pll.freq = sp_fr0; }
        }
        sp_t0 = cont ? sp_tiles : 0;
        sp_t1 = min((st->walk_avail - sp_so) / FL_TILE, n_tiles - 1);          // whole tiles, and never the frame's last one
        sp_go = sp_slot < maxf && sp_t1 > sp_t0;
        sp_keep = cont && !sp_go;                                              // nothing new to do: the speculation stands as it is
    } else if (!SPEC && act && st->spec_on) {
        // loops that do not run ahead (another bank size, the frames pooled by the host): what earlier ones did is given up
        pll.phase = st->spec_phase0; pll.freq = st->spec_freq0;
    }
    const bool sp_drop = !spec_on_here && act && st->spec_on;
    if (SPEC && sp_drop) { pll.phase = st->spec_phase0; pll.freq = st->spec_freq0; }
    // Wave priority: FL_PRIO (the serial loops' latency first) only where the balancer of the pipelined mode has found the FRONT END to be the
    // critical path (s2_demod.hip: g_prio_duty >= 2).  Beside a decoder that IS the critical path the loops run at the decoder's own base
    // priority: they take 88 instead of 60 ms per headline step -- there is slack for that on their stream -- and the decoder launch 342
    // instead of 352 (2 020 against 1 966 Msym/s).  At priority 1 nothing changes: what counts is being level with the decoder's parallel phases.
    if (co.g_prio_duty >= 2) __builtin_amdgcn_s_setprio(FL_PRIO); else __builtin_amdgcn_s_setprio(0);
    // the phase-error table's address lives in vector registers for the whole kernel (as a kernel argument it was re-fetched from the argument
    // segment -- a scalar-cache round trip -- in front of every symbol's lookup)
    const float* lut_err_v = C.lut_err;
    asm volatile("" : "+v"(lut_err_v));
    int nfmax = nf;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nfmax = max(nfmax, __shfl_xor(nfmax, o));
    for (int ff = 0; ff < nfmax + (SPEC ? 1 : 0); ++ff) {
        const bool spec_pass = SPEC && ff == nfmax;         // (after the confirmed frames: the window that is not one yet)
        if (spec_pass && !__any(sp_go)) break;
        const bool fact = spec_pass ? sp_go : ff < nf;
        const int fown = spec_pass ? sc * maxf + sp_slot : f0 + ff;
        const int donor = __ffsll((unsigned long long)__ballot(fact)) - 1;
        const int fdon = __shfl(fown, donor);
        const int f = fact ? fown : fdon;
        // tiles of this pass: all of them for a confirmed frame (from where the speculation stood, if this is its window), the whole tiles
        // at hand for the unconfirmed window -- uniform over the wave (with speculation a workgroup has ONE stream: the donor's values)
        const int t0 = SPEC ? __shfl(spec_pass ? sp_t0 : ((ff == 0 && resume) ? sp_tiles : 0), donor) : 0;
        const int t1 = SPEC ? __shfl(spec_pass ? sp_t1 : n_tiles, donor) : n_tiles;
        const bool fin = !spec_pass;
        const int foff = SPEC ? __shfl(sp_so, donor) : 0;
        if (spec_pass && fact && t0 == 0) { sp_ph0 = pll.phase; sp_fr0 = pll.freq; sp_carried = 0; }
        const PclDev pll_in = pll, hdr_in = hdr;
        const float nco_in = nco_freq;
        const cf32* __restrict__ fr = found ? work[f / maxf].fifo + (spec_pass ? foff : found[f].offset) : frames[f].sym;
        // (the unconfirmed window's output goes into the slot the frame will get AND into the stream's own buffer, which outlives the call: a
        //  frame that continues from tiles done in earlier calls first gets those out of there -- all 64 lanes copy, the workgroup has one stream)
        cf32* __restrict__ out = pllout + (size_t)f * slot_stride;
        cf32* __restrict__ keep = spec_pass ? work[f / maxf].spec_out : nullptr;
        if (SPEC && fin && t0 > 0) {
            const int carried = min(__shfl(sp_carried, donor), t0);
            const cf32* __restrict__ so = work[f / maxf].spec_out;
            for (int i = 90 + lane; i < carried * FL_TILE; i += 64) out[i] = so[i];
        }
        // ---- PLL (dvbs2_pll.cpp:34-86), 64-symbol tiles
        const int b0 = t0 * FL_TILE, b1 = fin ? plframe : t1 * FL_TILE;
        int next_pilot = -1, pb = 0;
        if (pilots && pilot_blocks > 0) {                   // (the pilot block at or ahead of the first symbol: a pass may start inside the frame)
            while (pb < pilot_blocks && pilot_start(pb) + 36 <= b0) ++pb;
            next_pilot = pb < pilot_blocks ? pilot_start(pb) : -1;
        }
        cf32 pacc{0.f, 0.f};
        // the next tile's symbols and Gold-sequence values are fetched into registers while the serial loop runs on the current one
        constexpr int NPF = (FL_TILE + FL_LPS - 1) / FL_LPS;
        cf32 pf[NPF];
        int prn = 0;
        auto fetch = [&](int base) {
#pragma unroll
            for (int t = 0; t < NPF; ++t) {
                const int i = gl + t * FL_LPS;
                if (i < FL_TILE && base + i < plframe) pf[t] = ldg(fr + base + i);
            }
            const int gi = base + lane;
            prn = (lane < FL_TILE && gi < plframe && gi >= 90) ? T.rn[gi - 90] : 0;
        };
        fetch(b0);
        for (int base = b0; base < b1; base += FL_TILE) {
            const int m = min(FL_TILE, plframe - base);
            FL_SYNC();
#pragma unroll
            for (int t = 0; t < NPF; ++t) {
                const int i = gl + t * FL_LPS;
                if (i < m) tl[i] = pf[t];
            }
            if (lane < m) rnt[lane] = (uint8_t)prn;
            FL_SYNC();
            if (base + FL_TILE < b1) fetch(base + FL_TILE);
            // A tile of payload symbols only (no header, no pilot symbol: all but a handful of the frame's tiles) takes the short loop: rotate, phase
            // error, advance -- the Gold-sequence rotation of the OUTPUT (descrambling acts on the rotated symbol and feeds nothing back into
            // the loop) is left to the lanes that copy the tile out.  Everything else goes through the general loop below.
            const bool plain = base >= 90 && !(next_pilot >= 0 && next_pilot < base + m && next_pilot + 36 > base);
            if (plain) {
                if (S2_PLL_TILES && C.bits != 5 && spw == 1) {
                    // one stream per workgroup: the tile as a fixed point over all 64 lanes (pll_payload_tile above)
                    const bool mine_k = lane < m;
                    const cf32 tmp_val = pll_payload_tile(mine_k ? tiles[0][lane] : cf32{0.f, 0.f}, mine_k, lane, m, pll, lut_err_v);
                    if (mine_k) tiles[0][FL_TILE + lane] = tmp_val;
                } else if (C.bits != 5) {
#if S2_PLL_ASM
                    // THE SHORT LOOP WRITTEN OUT (a small bank's frame loops are one wave per CU: its time is its instruction count + the table's round trip;
                    // tools/ubench/lone_wave.hip).  Per symbol, as in the C++ form below: tmp = tl[k] * phasor(-phase) (dvbs2m::sincosf_det; two packed
                    // multiplications + one packed addition), the table cell of tmp (lut_cell: y = fma(v, 256/1.5, 128), truncated and clamped; a symbol with
                    // a y within 2.5e-4 of an integer in ANY lane group leaves the loop untouched and goes through the C++ form), error = lut[cell],
                    // ot[k] = tmp, PhaseControlLoop::advance, one wrap into [-pi, pi].  ~60 instructions instead of ~85.
                    int k = 0;
                    const uint64_t lut_s = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)lut_err_v >> 32)) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)lut_err_v);
                    const float alpha_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pll.alpha)));
                    const float beta_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pll.beta)));
                    const float minf_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pll.minFreq)));
                    while (k < m) {
                        uint32_t ta = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) cf32*)tl + 8u * (uint32_t)k;
                        uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane(m - k), why = 0;
                        float tj, te;
                        int tji;
                        asm volatile(
                            "s_mov_b32 s60, 0xbf22f983\n\t"          // -2/pi
                            "s_mov_b32 s61, 0xbfc90fdb\n\t"          // -(pi/2 rounded to binary32)
                            "s_mov_b32 s62, 0x333bbd2e\n\t"          // -(pi/2 - that)
                            "s_mov_b32 s63, 0x80000000\n\t"
                            "s_mov_b32 s64, 0x37ccf5ce\n\t"          // polynomial coefficients (cos, sin): degree 2 ...
                            "s_mov_b32 s65, 0xb94ca1f9\n\t"
                            "v_mov_b32 v116, 0xbab6061a\n\t"         // ... degree 1 ...
                            "v_mov_b32 v117, 0x3c08839e\n\t"
                            "s_mov_b32 s66, 0x3d2aaaa5\n\t"          // ... degree 0
                            "s_mov_b32 s67, 0xbe2aaaa3\n\t"
                            "s_mov_b32 s68, 0x432aaaab\n\t"          // 256 / 1.5
                            "s_mov_b32 s69, 0x432aaaab\n\t"
                            "s_mov_b32 s70, 0x3983126f\n\t"          // 2.5e-4f
                            "s_mov_b32 s71, 0x40490fdb\n\t"          // pi
                            "s_mov_b32 s72, 0xc0490fdb\n\t"          // -pi
                            "s_mov_b32 s73, 0x40c90fdb\n\t"          // 2 pi
                            "s_mov_b32 s74, 0xc0c90fdb\n\t"          // -2 pi
                            "s_movk_i32 s75, 0xff\n\t"
                            "v_mov_b32 v126, 0x43000000\n\t"         // 128.0f
                            "v_mov_b32 v127, 0x43000000\n\t"
                            "1:\n\t"
                            "ds_read_b64 v[118:119], %[ta]\n\t"                                                       // tl[k]
                            "v_mul_f32 %[j], s60, %[ph]\n\t"
                            "v_rndne_f32 %[j], %[j]\n\t"
                            "v_cvt_i32_f32 %[ji], %[j]\n\t"
                            "v_fma_f32 v121, %[j], s61, -%[ph]\n\t"
                            "v_fmac_f32 v121, s62, %[j]\n\t"
                            "v_mul_f32 v120, v121, v121\n\t"                                                          // v[120:121] = (z, r)
                            "v_pk_fma_f32 v[122:123], v[120:121], s[64:65], v[116:117] op_sel_hi:[0,1,1]\n\t"
                            "v_pk_mul_f32 v[124:125], v[120:121], v[120:121] op_sel_hi:[1,0]\n\t"
                            "v_pk_fma_f32 v[122:123], v[122:123], v[120:121], s[66:67] op_sel_hi:[1,0,1]\n\t"
                            "v_fma_f32 v120, v120, -0.5, 1.0\n\t"
                            "v_pk_fma_f32 v[120:121], v[124:125], v[122:123], v[120:121]\n\t"                          // (pc, ps)
                            "v_and_b32 v122, 1, %[ji]\n\t"
                            "v_lshlrev_b32 v123, 30, %[ji]\n\t"
                            "v_sub_u32 v124, 0, v123\n\t"
                            "v_cmp_eq_u32 vcc, 0, v122\n\t"
                            "v_and_b32 v123, s63, v123\n\t"
                            "v_and_b32 v124, s63, v124\n\t"
                            "v_cndmask_b32 v125, v120, v121, vcc\n\t"
                            "v_cndmask_b32 v120, v121, v120, vcc\n\t"
                            "v_xor_b32 v121, v125, v123\n\t"
                            "v_xor_b32 v120, v120, v124\n\t"                                                          // v[120:121] = (cos, sin)
                            "s_waitcnt lgkmcnt(0)\n\t"
                            "v_pk_mul_f32 v[122:123], v[118:119], v[120:121] op_sel_hi:[0,1]\n\t"
                            "v_pk_mul_f32 v[124:125], v[118:119], v[120:121] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
                            "v_pk_add_f32 v[118:119], v[122:123], v[124:125] neg_lo:[0,1]\n\t"                         // tmp_val
                            // the table cell
                            "v_pk_fma_f32 v[122:123], v[118:119], s[68:69], v[126:127]\n\t"                            // y = fma(v, 256/1.5, 128)
                            "v_rndne_f32 v124, v122\n\t"
                            "v_rndne_f32 v125, v123\n\t"
                            "v_pk_add_f32 v[124:125], v[122:123], v[124:125] neg_lo:[0,1] neg_hi:[0,1]\n\t"            // y - rint(y)
                            "v_cvt_i32_f32 v122, v122\n\t"
                            "v_cvt_i32_f32 v123, v123\n\t"
                            "v_cmp_nge_f32 vcc, |v124|, s70\n\t"
                            "s_cbranch_vccnz 3f\n\t"
                            "v_cmp_nge_f32 vcc, |v125|, s70\n\t"
                            "s_cbranch_vccnz 3f\n\t"
                            "v_med3_i32 v122, v122, 0, s75\n\t"
                            "v_med3_i32 v123, v123, 0, s75\n\t"
                            "v_lshl_add_u32 v122, v122, 8, v123\n\t"
                            "v_lshlrev_b32 v122, 2, v122\n\t"
                            "global_load_dword %[e], v122, %[lut]\n\t"
                            "ds_write_b64 %[ta], v[118:119] offset:%[oto]\n\t"                                         // ot[k] = tmp_val
                            "v_add_u32 %[ta], 8, %[ta]\n\t"
                            "s_waitcnt vmcnt(0)\n\t"
                            "v_mul_f32 v124, %[beta], %[e]\n\t"
                            "v_add_f32 %[fr], %[fr], v124\n\t"
                            "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"
                            "v_mul_f32 v124, %[alpha], %[e]\n\t"
                            "v_add_f32 v124, %[fr], v124\n\t"
                            "v_add_f32 %[ph], %[ph], v124\n\t"
                            "v_add_f32 v124, s74, %[ph]\n\t"
                            "v_cmp_lt_f32 vcc, s71, %[ph]\n\t"
                            "v_cndmask_b32 %[ph], %[ph], v124, vcc\n\t"
                            "v_add_f32 v124, s73, %[ph]\n\t"
                            "v_cmp_gt_f32 vcc, s72, %[ph]\n\t"
                            "v_cndmask_b32 %[ph], %[ph], v124, vcc\n\t"
                            "s_sub_u32 %[left], %[left], 1\n\t"
                            "s_cmp_lg_u32 %[left], 0\n\t"
                            "s_cbranch_scc1 1b\n\t"
                            "s_branch 4f\n\t"
                            "3:\n\t"
                            "s_mov_b32 %[why], 1\n\t"
                            "4:\n\t"
                            "s_waitcnt lgkmcnt(0)"
                            : [ph] "+v"(pll.phase), [fr] "+v"(pll.freq), [ta] "+v"(ta), [left] "+s"(left), [why] "+s"(why), [j] "=&v"(tj), [ji] "=&v"(tji), [e] "=&v"(te)
                            : [alpha] "s"(alpha_s), [beta] "s"(beta_s), [minf] "s"(minf_s), [maxf] "v"(pll.maxFreq), [lut] "s"(lut_s), [oto] "n"(FL_TILE * 8)
                            : "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127",
                              "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "vcc", "scc", "memory");
                        k = m - (int)left;
                        if (!why) break;
                        const cf32 tmp_val = cmul(tl[k], phasor_fast(-pll.phase));           // (a cell index that needs the double form)
                        const float error = as_global(lut_err_v)[lut_cell(tmp_val.re, tmp_val.im)];
                        ot[k] = tmp_val;
                        pll.advance(error);
                        pll.wrap_pi_once();
                        ++k;
                    }
#else
                    for (int k = 0; k < m; ++k) {
                        const cf32 tmp_val = cmul(tl[k], phasor_fast(-pll.phase));
                        const float error = as_global(lut_err_v)[lut_cell(tmp_val.re, tmp_val.im)];
                        ot[k] = tmp_val;                     // (every lane of the group holds the same value: no predicate in the chain)
                        pll.advance(error);
                        pll.wrap_pi_once();
                    }
#endif
                } else {
                    for (int k = 0; k < m; ++k) {
                        const cf32 tmp_val = cmul(tl[k], phasor_fast(-pll.phase));
                        const float error = soft_phase_err_group<FL_LPS>((lds_cf32*)reinterpret_cast<const float*>(s_pts), C.states, C.amp, C.prescale, tmp_val, gl);
                        ot[k] = tmp_val;
                        pll.advance(error);
                        pll.wrap_pi_once();
                    }
                }
            } else
            for (int k = 0; k < m; ++k) {
                const int i = base + k;
                cf32 tmp_val = cmul(tl[k], phasor_fast(-pll.phase));
                float error = 0.f;
                cf32 o;
                bool block_end = false;
                if (i >= 90) {
                    cf32 descr = pl_descramble(tmp_val, rnt[k]);
                    bool is_pilot = next_pilot >= 0 && i >= next_pilot && i < next_pilot + 36;
                    if (!is_pilot) {
                        if (C.bits != 5) error = as_global(lut_err_v)[lut_cell(tmp_val.re, tmp_val.im)];
                        else error = soft_phase_err_group<FL_LPS>((lds_cf32*)reinterpret_cast<const float*>(s_pts), C.states, C.amp, C.prescale, tmp_val, gl);
                    } else {
                        if (co.pilot_aided) {
                            // own extension: data-aided on the known (1+j)/sqrt2 pilot, and the block estimate below
                            const cf32 pr = cmul(descr, cf32{0.707f, -0.707f});
                            error = cphase(pr);
                            pacc = cadd(pacc, pr);
                        } else {
                            // the reference's pilot error: decision-directed on the sign-sliced QPSK point, a tenth of the gain (dvbs2_pll.cpp:58)
                            const cf32 pt{descr.re > 0 ? 0.707f : -0.707f, descr.im > 0 ? 0.707f : -0.707f};
                            error = cphase(cmul(descr, cconj(pt))) / 10.0f;
                        }
                        if (i == next_pilot + 35) { ++pb; next_pilot = pb < pilot_blocks ? pilot_start(pb) : -1; block_end = true; }
                    }
                    o = descr;
                } else {
                    const cf32 pr = cmul(tmp_val, cconj(i < 26 ? T.sof[i] : plsc[i - 26]));
                    error = cphase(pr);
                    if (co.pilot_aided) pacc = cadd(pacc, pr);
                    block_end = i == 89;
                    o = cf32{0.f, 0.f};   // header symbols come from the PLHDR demod below
                }
                ot[k] = o;                                // (the same value in every lane of the group)
                pll.advance(error);
                pll.wrap_pi_once();
                if (co.pilot_aided && block_end) {
                    // pilot-aided mode (include/dvbs2gpu.h): the block estimate of the residual phase over the known symbols moves the loop phase
                    if (pacc.re != 0.f || pacc.im != 0.f) {
                        pll.phase += cphase(pacc);
                        pll.advance(0.f);
                        pll.wrap_pi();
                    }
                    pacc = cf32{0.f, 0.f};
                }
            }
            FL_SYNC();
            if (fact)
                for (int i = gl; i < m; i += FL_LPS)
                    if (base + i >= 90) {
                        const cf32 v = plain ? pl_descramble(ot[i], rnt[i]) : ot[i];
                        out[base + i] = v;
                        if (SPEC && keep) keep[base + i] = v;
                    }
        }
        if (fin) {
        FL_SYNC();
        // ---- coarse frequency error detector (dvbs2_fed.h): terms in parallel, summed in the reference's order
        #pragma unroll 1
        for (int i = gl; i < 88; i += FL_LPS) {
            cf32 r2 = (i + 2) < 26 ? T.sof[i + 2] : plsc[i + 2 - 26];
            cf32 r0 = i < 26 ? T.sof[i] : plsc[i - 26];
            fd[i] = cmul(cmul(cmul(fr[i + 2], cconj(r2)), cconj(fr[i])), r0).im;
        }
        FL_SYNC();
        float err = 0.f, symcnt = 90 - 2;
        for (int i = 0; i < 88; ++i) err += fd[i];
        if (pilots) {
            const cf32 p{0.707f, 0.707f};
            for (int b = 0; b < pilot_blocks; ++b) {
                int start = pilot_start(b);
                FL_SYNC();
                for (int i = gl; i < 36; i += FL_LPS) tl[i] = pl_descramble(fr[start + i], T.rn[start - 90 + i]);
                FL_SYNC();
                for (int i = gl; i < 36; i += FL_LPS)
                    if (i >= 2) fd[i] = cmul(cmul(cmul(tl[i], cconj(p)), cconj(tl[i - 2])), p).im;
                FL_SYNC();
                for (int i = 2; i < 36; ++i) err += fd[i];
                symcnt += 36 - 2;
            }
        }
        float est = err / symcnt;
        if (fabsf(est) < 0.02) nco_freq = nco_freq + est * (co.fll_bw / 100.0f);
        else nco_freq = nco_freq + est * co.fll_bw;
        if (nco_freq > 0.3f * PI_F) nco_freq = 0.3f * PI_F;
        if (nco_freq < -0.3f * PI_F) nco_freq = -0.3f * PI_F;
        // ---- PL header demod (dvbs2_plhdr_demod.cpp:33-67)
        // (the 90 header symbols are staged now, over the input + output tiles the PLL loop is through with)
        FL_SYNC();
        #pragma unroll 1
        for (int i = gl; i < 90; i += FL_LPS) tl[i] = fr[i];
        unsigned long long plheader = 0;
        const cf32 rot{(float)0.70710678118654757, (float)-0.70710678118654746};   // (cos(-pi/4), sin(-pi/4)) in double, cast
        FL_SYNC();
        for (int i = 0; i < 90; ++i) {
            cf32 tmp_val = cmul(tl[i], phasor_fast(-hdr.phase));
            float error = ((tmp_val.re > 0 ? 1.0f : -1.0f) * tmp_val.im) - ((tmp_val.im > 0 ? 1.0f : -1.0f) * tmp_val.re);
            cf32 o = (i & 1) ? cf32{-tmp_val.re, tmp_val.im} : cf32{tmp_val.im, tmp_val.re};
            if (gl == 0 && fact) out[i] = o;
            if (i >= 26) {
                const float sv = cmul(o, rot).re;
                plheader = plheader << 1 | (unsigned long long)(!(sv > 0));
                if (co.soft_plsc && gl == 0) tl[i - 26].re = sv;      // (slot i - 26 < i has been consumed: reuse it for the soft value)
            }
            hdr.advance(error);
            hdr.wrap_pi_once();
        }
        hdr.phase += hdr.freq * (plframe - 91);
        hdr.advance(0.f);
        hdr.wrap_pi();
        // codeword search: minimum distance, lowest index among the minima (the reference scans 0..127 with strict '<')
        int key = 0x7fffffff;
        if (!co.soft_plsc) {
            #pragma unroll 1
            for (int c = gl; c < 128; c += FL_LPS) {
                int dd = __popcll((T.plsc_code[c] ^ plheader) & ((1ull << 60) - 1));
                key = min(key, dd * 128 + c);
            }
#pragma unroll
            for (int o = FL_LPS / 2; o > 0; o >>= 1) key = min(key, __shfl_xor(key, o));      // all-reduce inside the lane group
        } else {
            // soft ML decode over all 64 code bits (include/dvbs2gpu.h): metric_c = sum_p +-soft[p] in index order, highest wins, lowest c on ties
            FL_SYNC();
            float bm = 0.f;
            int bc = 0x7fffffff;
            #pragma unroll 1
            for (int c = gl; c < 128; c += FL_LPS) {
                const unsigned long long code = T.plsc_code[c];
                float mtr = 0.f;
                for (int p = 0; p < 64; ++p) { const float sv = tl[p].re; mtr += ((code >> (63 - p)) & 1ull) ? -sv : sv; }
                if (bc == 0x7fffffff || mtr > bm) { bm = mtr; bc = c; }
            }
#pragma unroll
            for (int o = FL_LPS / 2; o > 0; o >>= 1) {
                const float om = __shfl_xor(bm, o);
                const int oc = __shfl_xor(bc, o);
                if (om > bm || (om == bm && oc < bc)) { bm = om; bc = oc; }
            }
            key = bc;
        }
        if (gl == 0 && fact) {
            const int best = key & 127;
            S2FrameStats stt;
            stt.best_match = 0.f; stt.ldpc_trials = 0; stt.bch_corr = 0;   // filled in by the host
            stt.detected_modcod = (best >> 2) & 31; stt.detected_short = (best & 2) >> 1; stt.detected_pilots = best & 1;
            stt.fed_err = est;
            stats[f] = stt;
        }
        }   // fin
        if (!fact) { pll = pll_in; hdr = hdr_in; nco_freq = nco_in; }
        FL_SYNC();
    }
    if (act && gl == 0) {
        st->pll_phase = pll.phase; st->pll_freq = pll.freq;
        st->hdr_phase = hdr.phase; st->hdr_freq = hdr.freq;
        st->nco_freq = nco_freq;
        if (found) st->loops_done = done + nf;
        if (spec_on_here) {
            if (sp_go) { st->spec_on = 1; st->spec_off = sp_so; st->spec_tiles = sp_t1; st->spec_phase0 = sp_ph0; st->spec_freq0 = sp_fr0; st->spec_carried = sp_carried; }
            else if (!sp_keep) st->spec_on = 0;
        }
        if (sp_drop) st->spec_on = 0;
    }
}

#undef FL_SYNC
// ------------------------------------------------------------------------------------------------ demapper
// where bit c (MSB first) of payload symbol j goes (s2_deinterleaver.cpp:72-136): QPSK swaps the pair (:80-84), the others are
// column-row de-interleaved, 8PSK 3/5 with the columns reversed (:26-31)
__device__ __forceinline__ int deint_pos(int constel, int rate, int bits, int rows, int j, int c) {
    if (bits == 2) return 2 * j + (1 - c);
    const int col = (constel == C_8PSK && rate == R3_5) ? (2 - c) : c;
    return col * rows + j;
}
// grid (x: symbol tiles, y: frame).  LUT fetch + bit de-interleave fused: LLR c of payload symbol j goes to
// column c (8PSK 3/5: columns reversed), QPSK just swaps the pair.
// Wide form (the LUT constellations, whenever a column is a whole number of words): a lane takes FOUR consecutive payload symbols -- two 16-byte
// loads, one table word per symbol (lut_bits4), one 32-bit store per bit column; the byte form (three byte loads and three byte stores per symbol)
// ran at 1.1 TB/s of its 7.8 GB per 32 768 frames.
// MIXED: a separate instantiation -- with the configuration a run-time choice between the arguments and a table the plain kernel lost its
// scalar operands (headline: 4.4 -> 18.8 ms per step)
template <bool MIXED>
__global__ __launch_bounds__(256) void s2_demap_kernel(S2ConstelDev C_arg, int rate, int slots, int pilots, int plframe,
                                                       const cf32* __restrict__ pllout, int8_t* __restrict__ llr, int N, const int* __restrict__ slot,
                                                       const S2StreamCfgDev* __restrict__ cfgs, int maxf, int8_t* const* __restrict__ llr_of) {
    if (!MIXED && maxf) __builtin_amdgcn_s_setprio(POST_PRIO);     // (one configuration: `maxf` carries the host's request, see s2_rrc_decim_kernel)
    const int f = blockIdx.y;
    const cf32* __restrict__ fr = pllout + (size_t)(slot ? slot[f] : f) * plframe;      // (stage pipeline: the loops wrote frame f to its stream's slot)
    int8_t* __restrict__ out = llr + (size_t)f * N;
    // mixed batch: the frame's configuration is its stream's (slot = stream * maxf + k; `plframe` as passed = the slot stride), its LLRs go where the table says.
    // The constellation record is read in place (a copy of its 32 points went to scratch)
    const S2StreamCfgDev* __restrict__ q = MIXED ? cfgs + slot[f] / maxf : nullptr;
    const S2ConstelDev& C = MIXED ? q->con : C_arg;
    if constexpr (MIXED) {
        rate = q->rate; slots = q->slots; pilots = q->pilots; N = q->N;
        out = llr_of[f];
    }
    const int nsym = slots * 90;
    const int bits = C.bits;
    const int rows = N / bits;
    int j0 = 0;
    if (C.lut_bits4 && (bits == 2 || (rows & 3) == 0)) {     // (PLFRAME lengths are even and the payload starts at symbol 90: fr + 90 + 4g is 16-byte aligned)
        const int nq = nsym >> 2;
        const bool rev = (C.constel == C_8PSK && rate == R3_5);
        const uint32_t* __restrict__ tab = C.lut_bits4;
        for (int g = blockIdx.x * 256 + threadIdx.x; g < nq; g += gridDim.x * 256) {
            const int j = 4 * g;
            int pos = 90 + j;
            if (pilots) pos += 36 * (j / 1440);   // (1440 is a multiple of 4: the four symbols lie in one run)
            const float4 a = *reinterpret_cast<const float4*>(fr + pos), b = *reinterpret_cast<const float4*>(fr + pos + 2);
            const uint32_t w0 = tab[lut_cell(a.x, a.y)], w1 = tab[lut_cell(a.z, a.w)], w2 = tab[lut_cell(b.x, b.y)], w3 = tab[lut_cell(b.z, b.w)];
            if (bits == 2) {
                // bytes {c1, c0} of each symbol, four symbols = eight consecutive bytes
                const uint32_t lo = __builtin_amdgcn_perm(w1, w0, 0x04050001u), hi = __builtin_amdgcn_perm(w3, w2, 0x04050001u);
                *reinterpret_cast<uint2*>(out + 2 * j) = make_uint2(lo, hi);
            } else {
                for (int c = 0; c < bits; ++c) {
                    // byte c of the four words
                    const uint32_t sel = 0x04000400u + 0x01010101u * (uint32_t)c;            // {w0.c, w1.c} per half
                    const uint32_t p01 = __builtin_amdgcn_perm(w1, w0, sel), p23 = __builtin_amdgcn_perm(w3, w2, sel);
                    const uint32_t word = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    const int col = rev ? 2 - c : c;
                    *reinterpret_cast<uint32_t*>(out + (size_t)col * rows + j) = word;
                }
            }
        }
        j0 = 4 * nq;
    }
    for (int j = j0 + blockIdx.x * 256 + threadIdx.x; j < nsym; j += gridDim.x * 256) {
        int pos = 90 + j;
        if (pilots) pos += 36 * (j / 1440);   // pilot blocks already passed (one after every 16 slots)
        cf32 v = fr[pos];
        int8_t b[5];
        if (bits != 5) {
            const int8_t* __restrict__ e = C.lut_bits + (size_t)lut_cell(v.re, v.im) * bits;
            for (int c = 0; c < bits; ++c) b[c] = e[c];
        } else {
            soft_calc_dev(C, v, b, nullptr);
        }
        for (int c = 0; c < bits; ++c) out[deint_pos(C.constel, rate, bits, rows, j, c)] = b[c];
    }
}
// S2Deinterleaver::deinterleave alone (s2_deinterleaver.cpp:72-136) on caller-supplied int8 frames: the same index function as the
// fused demapper; grid (x: tiles, y: frame)
__global__ __launch_bounds__(256) void s2_deinterleave_kernel(int constel, int rate, int bits, int N, const int8_t* __restrict__ in,
                                                              int8_t* __restrict__ out) {
    const int f = blockIdx.y, rows = N / bits;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
        out[(size_t)f * N + deint_pos(constel, rate, bits, rows, i / bits, i % bits)] = in[(size_t)f * N + i];
}

// ------------------------------------------------------------------------------------------------ PL sync walk (CCM)
// S2PLSyncBlock::process / internal_process (dvbs2_pl_sync.cpp:81-165) for one stream per workgroup, on the stream's symbol FIFO: every
// complete window of `raw` symbols from the FIFO head on is correlated -- brute force at every offset (dvbs2_pl_sync.cpp:111-143): 90
// differential products, signed sums over the SOF and the odd PLSC positions, c = max |csof +- cplsc|, arg-max with strict '>' and
// d.im > 0 = the lowest offset wins ties; best_pos == 0: the window is the frame; best_pos != 0 (state 0 -> 1): the frame is
// window[pos:] plus `pos` more symbols -- taken if they are in, else the window stays at the FIFO head and the stream waits in state 1
// (pl_pending) for the next call.  A frame carries the best_match of the correlation that placed it (the reference's member
// variable).  Per stream out: the frames' FIFO offsets, consumed symbols, symbols available, symbols this call added.
// sub / nsub: the walk of a call in time slices (stage pipeline): slice `sub` sees the symbols the RRC slices 0..sub have appended and resumes
// where slice sub - 1 stopped (walk_cur / walk_nf in the stream state); the frames come out as in one walk over the whole call, because a window
// is only ever looked at once it is complete and a realigned frame that is not all in yet waits (pl_pending) exactly as it does between calls.
__global__ __launch_bounds__(256) void s2_ccm_walk_kernel(const S2StreamWork* __restrict__ work, int raw, int maxf, S2VcmFound* __restrict__ found,
                                                          int* __restrict__ counts, int sub, int nsub, const S2StreamCfgDev* __restrict__ cfgs) {
    __shared__ cf32 d[256 + 96];
    __shared__ float r_val[256];
    __shared__ int r_idx[256];
    if (nsub >> 16) __builtin_amdgcn_s_setprio(POST_PRIO);         // (see s2_rrc_decim_kernel)
    nsub &= 0xffff;
    const int s = blockIdx.x, tid = threadIdx.x;
    if (cfgs) raw = cfgs[s].plframe;             // mixed batch: the PLFRAME length of THIS stream's MODCOD
    const S2StreamWork w = work[s];
    S2StreamState* st = w.st;
    const cf32* __restrict__ fifo = w.fifo;
    const int nsym = nsub > 1 ? st->n_sym_slice[sub] : st->n_sym;
    const int avail = w.fifo_fill + nsym;
    int pend = st->pl_pending, cur = (nsub > 1 && sub) ? st->walk_cur : 0, nf = (nsub > 1 && sub) ? st->walk_nf : 0;
    float lastbm = st->pl_last_bm;
    // a window the frame loops of the previous call were ahead in: the FIFO has lost the walk_cur symbols in front of it since
    if (sub == 0 && tid == 0 && st->spec_on) { st->spec_off -= st->walk_cur; st->spec_carried = st->spec_tiles; }
    const uint32_t dsof = 0x18d2e82u ^ (0x18d2e82u >> 1);
    const unsigned long long SCR = 0x719d83c953422dfaull;
    const unsigned long long dscr = SCR ^ (SCR >> 1);
    const int noff = raw - 90;
    bool waiting = false;
    if (pend > 0) {                 // (the window the realigned frame starts in lies at `cur`: the FIFO head at the start of a call)
        if (avail >= cur + raw + pend) {
            if (tid == 0) found[(size_t)s * maxf + nf] = S2VcmFound{cur + pend, 0, lastbm, 0};
            ++nf;
            cur += raw + pend;
            pend = 0;
        } else {
            waiting = true;
        }
    }
    while (!waiting && cur + raw <= avail && nf < maxf) {
        const cf32* __restrict__ x = fifo + cur;
        float bestv = 0.f;
        int besti = 0;
        for (int base = 0; base < noff; base += 256) {
            __syncthreads();
            for (int k = tid; k < 256 + 90; k += 256) {
                const int a = base + k;
                cf32 v{0.f, 0.f};
                if (k >= 1 && a < raw) v = cmul(cconj(x[a - 1]), x[a]);
                d[k] = v;
            }
            __syncthreads();
            const int ss = base + tid;
            if (ss < noff) {
                const cf32* dd = &d[tid];
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
                if (diff > bestv && dv.im > 0) { bestv = diff; besti = ss; }
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
        const float bm = r_val[0];
        const int pos = bm > 0.f ? r_idx[0] : 0;
        __syncthreads();
        lastbm = bm;
        if (pos == 0) {
            if (tid == 0) found[(size_t)s * maxf + nf] = S2VcmFound{cur, 0, bm, 0};
            ++nf;
            cur += raw;
        } else if (avail >= cur + raw + pos) {
            if (tid == 0) found[(size_t)s * maxf + nf] = S2VcmFound{cur + pos, 0, bm, 0};
            ++nf;
            cur += raw + pos;
        } else {
            pend = pos;
            waiting = true;
        }
    }
    if (tid == 0) {
        st->pl_pending = pend; st->pl_last_bm = lastbm;
        st->walk_cur = cur; st->walk_nf = nf; st->walk_avail = avail;
        if (sub == 0) st->loops_done = 0;
        counts[4 * s] = nf; counts[4 * s + 1] = cur; counts[4 * s + 2] = avail; counts[4 * s + 3] = nsym;
    }
}

// ================================================================================================ ACM/VCM path (acm_vcm)
// Own extension (include/dvbs2gpu.h): the reference's PL sync assumes ONE frame length (dvbs2_pl_sync.cpp:81-165)
// and its GUI re-configures the demodulator after 50 consistent PLS sightings (main.cpp:375-408).  Here the framing follows the PLS
// code of every frame.  The CPU oracle (oracle/s2chain.cpp, S2Rx::vcm_walk / pls_at / soft_plsc_decode) states the same rules.
//
// s2_vcm_walk_kernel -- ONE WORKGROUP PER STREAM walks the stream's symbol FIFO:
//   not locked: the reference's differential SOF + PLSC correlator (it does not depend on the MODCOD) over the next VCM_ACQ_WINDOW
//               offsets, same arg-max rule (strict '>', d.im > 0, lowest offset wins ties); the best offset becomes the frame start.
//   locked:     soft PLS decode at the frame start: phase reference z = sum x[k] conj(sof[k]) over the 26 SOF symbols, soft value of
//               PLSC symbol p = (w.re + w.im) for even p, (w.im - w.re) for odd p with w = x[26+p] conj(z) (pi/2-BPSK, s2_defs.h:60-80),
//               metric of codeword c = sum_p +-soft[p] in index order, highest metric wins (lowest c on ties), ratio = metric /
//               sum |soft|.  ratio >= VCM_MIN_RATIO, SOF quality |z| / sum |x| >= sof_threshold and a valid code: the frame (its
//               length follows from the code) is recorded once all its symbols are in the FIFO, the next header is expected right
//               behind it.  Otherwise the lock is dropped and the search resumes one symbol further.
__global__ __launch_bounds__(256) void s2_vcm_walk_kernel(const S2StreamWork* __restrict__ work, S2PlTablesDev T, const S2VcmMod* __restrict__ mods,
                                                          float sof_threshold, int maxf, S2VcmFound* __restrict__ found, int* __restrict__ counts) {
    __shared__ cf32 d[256 + 96];
    __shared__ float r_val[256];
    __shared__ int r_idx[256];
    __shared__ float soft[64];
    __shared__ cf32 hx[90];
    __shared__ float s_ratio, s_sofq;
    __shared__ int s_pls;
    const int s = blockIdx.x, tid = threadIdx.x;
    const S2StreamWork w = work[s];
    S2StreamState* st = w.st;
    const cf32* __restrict__ fifo = w.fifo;
    const int nsym = st->n_sym;
    const int avail_total = w.fifo_fill + nsym;
    int synced = st->vcm_synced;
    int cur = 0, nf = 0;
    const uint32_t dsof = 0x18d2e82u ^ (0x18d2e82u >> 1);
    const unsigned long long SCR = 0x719d83c953422dfaull;
    const unsigned long long dscr = SCR ^ (SCR >> 1);
    while (true) {
        const int avail = avail_total - cur;
        if (!synced) {
            if (avail < VCM_ACQ_WINDOW + 90) break;
            const cf32* __restrict__ x = fifo + cur;
            float bestv = 0.f;
            int besti = 0;
            for (int base = 0; base < VCM_ACQ_WINDOW; base += 256) {
                __syncthreads();
                for (int k = tid; k < 256 + 90; k += 256) {
                    const int a = base + k;
                    cf32 v{0.f, 0.f};
                    if (k >= 1 && a < VCM_ACQ_WINDOW + 90) v = cmul(cconj(x[a - 1]), x[a]);
                    d[k] = v;
                }
                __syncthreads();
                const int ss = base + tid;
                if (ss < VCM_ACQ_WINDOW) {
                    const cf32* dd = &d[tid];
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
                    if (diff > bestv && dv.im > 0) { bestv = diff; besti = ss; }
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
            const float bv = r_val[0];
            const int bi = r_idx[0];
            __syncthreads();
            if (bv > 0.f) { cur += bi; synced = 1; }
            else cur += VCM_ACQ_WINDOW;
            continue;
        }
        if (avail < 90) break;
        // ---- PLS decode at the expected frame start
        __syncthreads();
        if (tid < 90) hx[tid] = fifo[cur + tid];
        __syncthreads();
        cf32 z{0.f, 0.f};
        float amp = 0.f;
        for (int k = 0; k < 26; ++k) { z = cadd(z, cmul(hx[k], cconj(T.sof[k]))); amp += camp(hx[k]); }     // (every thread: the same sequential sums)
        if (tid < 64) {
            const cf32 wv = cmul(hx[26 + tid], cconj(z));
            soft[tid] = (tid & 1) ? (wv.im - wv.re) : (wv.re + wv.im);
        }
        __syncthreads();
        float mtr = 0.f;
        if (tid < 128) {
            const unsigned long long code = T.plsc_code[tid];
            for (int p = 0; p < 64; ++p) mtr += ((code >> (63 - p)) & 1ull) ? -soft[p] : soft[p];
        }
        r_val[tid] = mtr; r_idx[tid] = tid;
        __syncthreads();
        for (int o = 64; o > 0; o >>= 1) {        // over the 128 codewords: highest metric, lowest index on ties
            if (tid < o) {
                float v2 = r_val[tid + o]; int i2 = r_idx[tid + o];
                float v1 = r_val[tid]; int i1 = r_idx[tid];
                if (v2 > v1 || (v2 == v1 && i2 < i1)) { r_val[tid] = v2; r_idx[tid] = i2; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            float tot = 0.f;
            for (int p = 0; p < 64; ++p) tot += fabsf(soft[p]);
            s_ratio = tot > 0.f ? r_val[0] / tot : 0.f;
            s_sofq = amp > 0.f ? camp(z) / amp : 0.f;
            s_pls = r_idx[0];
        }
        __syncthreads();
        const int pls = s_pls;
        const float ratio = s_ratio, sofq = s_sofq;
        const S2VcmMod M = mods[pls];
        const bool ok = ratio >= VCM_MIN_RATIO && sofq >= sof_threshold && M.valid != 0;
        if (!ok) { synced = 0; cur += 1; continue; }
        if (avail < M.plframe || nf >= maxf) break;
        if (tid == 0) found[(size_t)s * maxf + nf] = S2VcmFound{cur, pls, sofq, 0};
        ++nf;
        cur += M.plframe;
    }
    if (tid == 0) {
        st->vcm_synced = synced;
        counts[4 * s] = nf; counts[4 * s + 1] = cur; counts[4 * s + 2] = avail_total; counts[4 * s + 3] = nsym;
    }
}

// s2_vcm_loops_kernel -- ONE WAVE PER STREAM, its frames in order, every frame with the parameters of ITS PLS code: coarse FED + NCO
// feedback, PLL, PLHDR demod exactly as in s2_frame_loops_kernel (same operation order), in the plain one-wave-per-stream form: the
// serial recurrences run uniformly in all lanes, the lanes share the loads / stores and the parallel parts.
__global__ __launch_bounds__(64) void s2_vcm_loops_kernel(const S2StreamWork* __restrict__ work, const S2VcmFrame* __restrict__ frames,
                                                          const int* __restrict__ first, S2LoopCoefs co, S2PlTablesDev T,
                                                          const S2VcmMod* __restrict__ mods, const S2ConstelDev* __restrict__ cons,
                                                          cf32* __restrict__ pllout, S2FrameStats* __restrict__ stats) {
    __shared__ cf32 tile[64];
    __shared__ cf32 otile[64];
    __shared__ uint8_t rnt[64];
    __shared__ float fedt[96];
    __shared__ cf32 hdr_sym[90];
    __shared__ float hsoft[64];
    __shared__ cf32 v_pts[32];
    const int lane = threadIdx.x, s = blockIdx.x;
    S2StreamState* st = work[s].st;
    PclDev pll{co.pll_alpha, co.pll_beta, st->pll_phase, st->pll_freq, co.pll_min_freq, co.pll_max_freq};
    PclDev hdr{co.hdr_alpha, co.hdr_beta, st->hdr_phase, st->hdr_freq, co.hdr_min_freq, co.hdr_max_freq};
    float nco_freq = st->nco_freq;
    const float PI_F = 3.14159265358979323846f;
    for (int f = first[s]; f < first[s + 1]; ++f) {
        const S2VcmFrame F = frames[f];
        const S2VcmMod M = mods[F.pls];
        S2FrameStats stt;
        stt.best_match = F.sofq; stt.ldpc_trials = 0; stt.bch_corr = 0; stt.fed_err = 0.f; stt.bbframe_bytes = M.valid == 1 ? M.kb : 0;
        stt.detected_modcod = F.pls >> 2; stt.detected_short = (F.pls >> 1) & 1; stt.detected_pilots = F.pls & 1;
        if (M.valid != 1) {                       // dummy PLFRAME: only the framing advances
            if (lane == 0) stats[f] = stt;
            continue;
        }
        const cf32* __restrict__ fr = F.sym;
        cf32* __restrict__ out = pllout + F.pll_off;
        const cf32* __restrict__ plsc = T.plsc + (size_t)F.pls * 64;
        const int plframe = M.plframe, pilots = M.pilots, pilot_blocks = M.pilot_blocks;
        const S2ConstelDev* __restrict__ C = cons + M.con;
        const int cbits = C->bits;
        // ---- coarse frequency error detector (dvbs2_fed.h): terms in parallel, summed in the reference's order
        __syncthreads();
        if (cbits == 5 && lane < 32) v_pts[lane] = lane < C->states ? C->pts_g[lane] : cf32{0.f, 0.f};   // (this frame's 32APSK points, see soft_phase_err_group)
        for (int i = lane; i < 88; i += 64) {
            cf32 r2 = (i + 2) < 26 ? T.sof[i + 2] : plsc[i + 2 - 26];
            cf32 r0 = i < 26 ? T.sof[i] : plsc[i - 26];
            fedt[i] = cmul(cmul(cmul(fr[i + 2], cconj(r2)), cconj(fr[i])), r0).im;
        }
        for (int i = lane; i < 90; i += 64) hdr_sym[i] = fr[i];
        __syncthreads();
        float err = 0.f, symcnt = 90 - 2;
        for (int i = 0; i < 88; ++i) err += fedt[i];
        if (pilots) {
            const cf32 p{0.707f, 0.707f};
            for (int b = 0; b < pilot_blocks; ++b) {
                const int start = pilot_start(b);
                __syncthreads();
                if (lane < 36) tile[lane] = pl_descramble(fr[start + lane], T.rn[start - 90 + lane]);
                __syncthreads();
                if (lane >= 2 && lane < 36) fedt[lane] = cmul(cmul(cmul(tile[lane], cconj(p)), cconj(tile[lane - 2])), p).im;
                __syncthreads();
                for (int i = 2; i < 36; ++i) err += fedt[i];
                symcnt += 36 - 2;
            }
        }
        const float est = err / symcnt;
        if (fabsf(est) < 0.02) nco_freq = nco_freq + est * (co.fll_bw / 100.0f);
        else nco_freq = nco_freq + est * co.fll_bw;
        if (nco_freq > 0.3f * PI_F) nco_freq = 0.3f * PI_F;
        if (nco_freq < -0.3f * PI_F) nco_freq = -0.3f * PI_F;
        stt.fed_err = est;
        // ---- PLL (dvbs2_pll.cpp:34-86)
        int next_pilot = (pilots && pilot_blocks > 0) ? pilot_start(0) : -1, pb = 0;
        cf32 pacc{0.f, 0.f};
        for (int base = 0; base < plframe; base += 64) {
            const int m = min(64, plframe - base);
            __syncthreads();
            if (lane < m) {
                tile[lane] = fr[base + lane];
                rnt[lane] = base + lane >= 90 ? T.rn[base + lane - 90] : 0;
            }
            __syncthreads();
            // a tile of payload symbols only (no header, no pilot symbol) of a LUT constellation: the fixed-point form (pll_payload_tile) -- the general loop below is
            // ~100 instructions and a table round trip per symbol
            const bool plain = S2_PLL_TILES && cbits != 5 && base >= 90 && !(next_pilot >= 0 && next_pilot < base + m && next_pilot + 36 > base);
            if (plain) {
                const bool mine_k = lane < m;
                const cf32 tmp_val = pll_payload_tile(mine_k ? tile[lane] : cf32{0.f, 0.f}, mine_k, lane, m, pll, C->lut_err);
                if (mine_k) otile[lane] = pl_descramble(tmp_val, rnt[lane]);
            } else
            for (int k = 0; k < m; ++k) {
                const int i = base + k;
                cf32 tmp_val = cmul(tile[k], phasor(-pll.phase));
                float error = 0.f;
                cf32 o;
                bool block_end = false;
                if (i >= 90) {
                    cf32 descr = pl_descramble(tmp_val, rnt[k]);
                    bool is_pilot = next_pilot >= 0 && i >= next_pilot && i < next_pilot + 36;
                    if (!is_pilot) {
                        if (cbits != 5) error = as_global(C->lut_err)[lut_cell(tmp_val.re, tmp_val.im)];
                        else error = soft_phase_err_group<64>((lds_cf32*)reinterpret_cast<const float*>(v_pts), C->states, C->amp, C->prescale, tmp_val, lane);
                    } else {
                        if (co.pilot_aided) {
                            // own extension: data-aided on the known (1+j)/sqrt2 pilot, and the block estimate below
                            const cf32 pr = cmul(descr, cf32{0.707f, -0.707f});
                            error = cphase(pr);
                            pacc = cadd(pacc, pr);
                        } else {
                            // the reference's pilot error: decision-directed on the sign-sliced QPSK point, a tenth of the gain (dvbs2_pll.cpp:58)
                            const cf32 pt{descr.re > 0 ? 0.707f : -0.707f, descr.im > 0 ? 0.707f : -0.707f};
                            error = cphase(cmul(descr, cconj(pt))) / 10.0f;
                        }
                        if (i == next_pilot + 35) { ++pb; next_pilot = pb < pilot_blocks ? pilot_start(pb) : -1; block_end = true; }
                    }
                    o = descr;
                } else {
                    const cf32 pr = cmul(tmp_val, cconj(i < 26 ? T.sof[i] : plsc[i - 26]));
                    error = cphase(pr);
                    if (co.pilot_aided) pacc = cadd(pacc, pr);
                    block_end = i == 89;
                    o = cf32{0.f, 0.f};
                }
                otile[k] = o;                             // (uniform value)
                pll.advance(error);
                pll.wrap_pi_once();
                if (co.pilot_aided && block_end) {
                    if (pacc.re != 0.f || pacc.im != 0.f) {
                        pll.phase += cphase(pacc);
                        pll.advance(0.f);
                        pll.wrap_pi();
                    }
                    pacc = cf32{0.f, 0.f};
                }
            }
            __syncthreads();
            if (lane < m && base + lane >= 90) out[base + lane] = otile[lane];
        }
        // ---- PL header demod (dvbs2_plhdr_demod.cpp:33-67): loop + header symbols; the frame's PLS code is the one the framing decoded
        const cf32 rot{(float)0.70710678118654757, (float)-0.70710678118654746};
        __syncthreads();
        for (int i = 0; i < 90; ++i) {
            cf32 tmp_val = cmul(hdr_sym[i], phasor(-hdr.phase));
            float error = ((tmp_val.re > 0 ? 1.0f : -1.0f) * tmp_val.im) - ((tmp_val.im > 0 ? 1.0f : -1.0f) * tmp_val.re);
            cf32 o = (i & 1) ? cf32{-tmp_val.re, tmp_val.im} : cf32{tmp_val.im, tmp_val.re};
            if (lane == 0) out[i] = o;
            hdr.advance(error);
            hdr.wrap_pi_once();
        }
        hdr.phase += hdr.freq * (plframe - 91);
        hdr.advance(0.f);
        hdr.wrap_pi();
        (void)hsoft; (void)rot;
        if (lane == 0) stats[f] = stt;
    }
    if (lane == 0) {
        st->pll_phase = pll.phase; st->pll_freq = pll.freq;
        st->hdr_phase = hdr.phase; st->hdr_freq = hdr.freq;
        st->nco_freq = nco_freq;
    }
}

// soft demap + bit de-interleave of pooled frames with per-frame MODCOD; grid (x: symbol tiles, y: frame); LLRs in frame order
__global__ __launch_bounds__(256) void s2_vcm_demap_kernel(const S2VcmFrame* __restrict__ frames, const S2VcmMod* __restrict__ mods,
                                                           const S2ConstelDev* __restrict__ cons, const cf32* __restrict__ pllout,
                                                           int8_t* __restrict__ llr) {
    const S2VcmFrame F = frames[blockIdx.y];
    const S2VcmMod M = mods[F.pls];
    if (M.valid != 1) return;
    const S2ConstelDev& C = cons[M.con];
    const cf32* __restrict__ fr = pllout + F.pll_off;
    int8_t* __restrict__ out = llr + F.llr_off;
    const int nsym = M.slots * 90, bits = M.bits, rows = M.N / bits;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nsym; j += gridDim.x * 256) {
        int pos = 90 + j;
        if (M.pilots) pos += 36 * (j / 1440);
        const cf32 v = fr[pos];
        int8_t b[5];
        if (bits != 5) {
            const int8_t* __restrict__ e = C.lut_bits + (size_t)lut_cell(v.re, v.im) * bits;
            for (int c = 0; c < bits; ++c) b[c] = e[c];
        } else {
            soft_calc_dev(C, v, b, nullptr);
        }
        for (int c = 0; c < bits; ++c) out[deint_pos(M.constel, M.rate, bits, rows, j, c)] = b[c];
    }
}
// frames of one FEC group -> contiguous LLR block (the decoder takes [frames][N]); grid (x: tiles, y: frame of the group)
__global__ __launch_bounds__(256) void s2_vcm_gather_kernel(const S2VcmFrame* __restrict__ frames, const int* __restrict__ idx, int N,
                                                            const int8_t* __restrict__ llr, int8_t* __restrict__ grp) {
    const int8_t* __restrict__ src = llr + frames[idx[blockIdx.y]].llr_off;
    int8_t* __restrict__ dst = grp + (size_t)blockIdx.y * N;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N / 4; i += gridDim.x * 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
}
// BBFRAMEs of one FEC group -> their places in the streams' output buffers
__global__ __launch_bounds__(256) void s2_vcm_scatter_kernel(const int* __restrict__ idx, int kb, const uint8_t* __restrict__ bb, uint8_t* const* __restrict__ dst) {
    const uint8_t* __restrict__ src = bb + (size_t)blockIdx.x * kb;
    uint8_t* __restrict__ d = dst[idx[blockIdx.x]];
    if (!d) return;                                 // (pipelined mode: the frame's stream has left the batch)
    for (int i = threadIdx.x; i < kb; i += 256) d[i] = src[i];
}

// ------------------------------------------------------------------------------------------------ call tails
__global__ void s2_collect_kernel(const S2StreamWork* __restrict__ work, int nstreams, int* __restrict__ nsym, float* __restrict__ nco) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nstreams) {
        S2StreamState* st = work[s].st;
        nsym[s] = st->n_sym; nco[s] = st->nco_freq;
        st->nco_agc = st->nco_freq;              // the FED's feedback of this call's frames steers the NCO from the next call on
    }
}
__global__ __launch_bounds__(256) void s2_scatter_out_kernel(const S2StreamWork* __restrict__ work, const S2FrameRef* __restrict__ frames,
                                                             const int* __restrict__ first, int kb, const uint8_t* __restrict__ bb) {
    const int f = blockIdx.x;
    const int s = frames[f].stream;
    uint8_t* __restrict__ dst = work[s].out + (size_t)(f - first[s]) * kb;
    const uint8_t* __restrict__ src = bb + (size_t)f * kb;
    for (int i = threadIdx.x; i < kb; i += 256) dst[i] = src[i];
}
// pipelined delivery: the BBFRAMEs of the PREVIOUS call go to the output buffers of the current one
__global__ __launch_bounds__(256) void s2_scatter_out2_kernel(uint8_t* const* __restrict__ outs, const S2FrameRef* __restrict__ frames,
                                                              const int* __restrict__ first, int kb, const uint8_t* __restrict__ bb) {
    const int f = blockIdx.x;
    const int s = frames[f].stream;
    if (!outs[s]) return;                           // (pipelined mode: the stream is not part of the call that collects this job -- its frames are dropped)
    uint8_t* __restrict__ dst = outs[s] + (size_t)(f - first[s]) * kb;
    const uint8_t* __restrict__ src = bb + (size_t)f * kb;
    for (int i = threadIdx.x; i < kb; i += 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void s2_fifo_compact_kernel(const S2StreamWork* __restrict__ work, const int* __restrict__ cur_fill) {
    const int s = blockIdx.y;
    const int cur = cur_fill[2 * s], fill = cur_fill[2 * s + 1];
    if (cur <= 0) return;
    const cf32* __restrict__ src = work[s].fifo + cur;
    cf32* __restrict__ dst = work[s].fifo_next;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < fill - cur; i += gridDim.x * 256) dst[i] = src[i];
}

// ================================================================================================ DVB-S front end (a17)
// demod::QPSK_ALT::process (common/dsp/demod/qpsk_alt.cpp:136-144): FastAGC -> FLL -> RRC FIR -> COMPLEX_FD -> Costas<4>,
// then DVBSymToSoftBlock's conversion into the 8192-soft block FIFO (dvbs_syms_to_soft.cpp:24-42).
__device__ __forceinline__ float fast_amplitude(cf32 v) {   // SDR++ complex_t::fastAmplitude: the larger of |re|, |im| plus 0.4 times the smaller one
    // (re_abs > im_abs ? re_abs + 0.4f * im_abs : im_abs + 0.4f * re_abs, as max / min: two instructions with |.| operand modifiers instead of two
    // ANDs, a compare, a wait state and a select -- the same value for every input that is not a NaN)
    const float re_abs = fabsf(v.re), im_abs = fabsf(v.im);
    return fmaxf(re_abs, im_abs) + 0.4f * fminf(re_abs, im_abs);
}
// CLAMP_PHASE with [-pi, pi] for loops whose phase moves by less than 2 pi per step (FLL: |freq| <= pi/2, Costas: pi/10 + alpha): the
// reference's two while loops then run at most once each, and two selects give the same value without the divergent loop code
__device__ __forceinline__ void pcl_wrap_pi(float& phase) {
    const float PI_F = 3.14159265358979323846f;
    const float delta = PI_F - (-PI_F);
    phase = phase > PI_F ? phase - delta : phase;
    phase = phase < -PI_F ? phase + delta : phase;
}

// complex product in THREE packed instructions -- (a.re b.re, a.re b.im), (a.im b.im, a.im b.re), then low halves subtracted and high halves added
// (v_pk_add_f32 neg_lo) -- where the compiler builds five (two packed adds, a register move to pick one half of each).  Same roundings as cmul.
__device__ __forceinline__ cf32 cmul_pk3(cf32 a, cf32 b) {
    const f32x2 av{a.re, a.im}, bv{b.re, b.im};
    f32x2 p, q, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(av), "v"(bv));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(av), "v"(bv));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));
    return cf32{r.x, r.y};
}
// ONE WAVE PER STREAM.  loop::FLL::process (fll.cpp:135-149): every sample is rotated by the loop phase and fed to the two
// band-edge FIRs whose amplitude difference steers the loop -- a feedback through two 65-tap complex dot products per sample.
// The dot products run as a SYSTOLIC ARRAY over the lanes: lane k holds tap k and a running sum; each new sample x[m] adds
// x[m]*t[k] to the running sum of output m+64-k and the sums move one lane up (DPP wave_shr:1), so every output accumulates its
// terms in tap order 0..64 (the order of VOLK's generic kernel) while only the LAST term (tap 64, the newest sample) sits in the
// loop's serial chain.  Lanes 0..63 = taps 0..63; tap 64 is applied by all lanes to the sum leaving lane 63.
// ONE SAMPLE of the FLL loop (dvbs_fll_kernel's inline assembly; operands as named there, v40.. / s40.. as set up there), in this order:
//   y = the sample (LDS read issued first, waited for behind the phasor)
//   phasor(-phase) = dvbs2m::sincosf_det: j = rint(-phase 2/pi), r = fma(j, -lo, fma(j, -hi, -phase)), z = r r, both minimax polynomials in packed FMAs
//     (v[46:47] = (cos part, sin part)), quadrant fix-ups on the sign bits -> v[46:47] = (cos, sin)
//   x = y * phasor (two packed multiplications, one packed addition with neg_lo) -> v[42:43], stored to the output tile
//   the two band-edge outputs of lane 63: x * tap 64 + that lane's running sums v[56:57], v[58:59] (taps 0..63); err = fastAmplitude(upper) - fastAmplitude(lower)
//     (max + 0.4 min with |.| operand modifiers); ONE v_readlane of lane 63 hands err to the wave, with the first systolic product between its source's write and it
//   the systolic products x * tap of every lane; freq += beta err, clamped (v_med3); phase += freq, wrapped into [-pi, pi] (compares through VCC)
//   every lane adds its product to the sum arriving from the lane below (v_add_f32_dpp wave_shr:1, lane 0 reads zero = starts the sum of output m + 64)
#define DVBS_FLL_SAMPLE \
    "ds_read_b64 v[42:43], %[xa] offset:512\n\t" \
    "v_mul_f32 v44, s40, %[ph]\n\t" \
    "v_rndne_f32 v44, v44\n\t" \
    "v_cvt_i32_f32 v45, v44\n\t" \
    "v_fma_f32 v47, v44, s41, -%[ph]\n\t" \
    "v_fmac_f32 v47, s42, v44\n\t" \
    "v_mul_f32 v46, v47, v47\n\t" \
    "v_pk_fma_f32 v[48:49], v[46:47], s[44:45], v[40:41] op_sel_hi:[0,1,1]\n\t" \
    "v_pk_mul_f32 v[50:51], v[46:47], v[46:47] op_sel_hi:[1,0]\n\t" \
    "v_pk_fma_f32 v[48:49], v[48:49], v[46:47], s[46:47] op_sel_hi:[1,0,1]\n\t" \
    "v_fma_f32 v46, v46, -0.5, 1.0\n\t" \
    "v_pk_fma_f32 v[46:47], v[50:51], v[48:49], v[46:47]\n\t" \
    "v_and_b32 v48, 1, v45\n\t" \
    "v_lshlrev_b32 v49, 30, v45\n\t" \
    "v_sub_u32 v50, 0, v49\n\t" \
    "v_cmp_eq_u32 vcc, 0, v48\n\t" \
    "v_and_b32 v49, s43, v49\n\t" \
    "v_and_b32 v50, s43, v50\n\t" \
    "v_cndmask_b32 v51, v46, v47, vcc\n\t" \
    "v_cndmask_b32 v46, v47, v46, vcc\n\t" \
    "v_xor_b32 v47, v51, v49\n\t" \
    "v_xor_b32 v46, v46, v50\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_pk_mul_f32 v[48:49], v[42:43], v[46:47] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 v[50:51], v[42:43], v[46:47] op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
    "v_pk_add_f32 v[42:43], v[48:49], v[50:51] neg_lo:[0,1]\n\t" \
    "ds_write_b64 %[xa], v[42:43]\n\t" \
    "v_pk_mul_f32 v[48:49], v[42:43], %[tlL] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 v[50:51], v[42:43], %[tlL] op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
    "v_pk_add_f32 v[48:49], v[48:49], v[50:51] neg_lo:[0,1]\n\t" \
    "v_pk_mul_f32 v[50:51], v[42:43], %[thL] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 v[52:53], v[42:43], %[thL] op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
    "v_pk_add_f32 v[50:51], v[50:51], v[52:53] neg_lo:[0,1]\n\t" \
    "v_pk_add_f32 v[48:49], v[48:49], v[56:57]\n\t" \
    "v_pk_add_f32 v[50:51], v[50:51], v[58:59]\n\t" \
    "v_max_f32 v52, |v50|, |v51|\n\t" \
    "v_max_f32 v53, |v48|, |v49|\n\t" \
    "v_min_f32 v54, |v50|, |v51|\n\t" \
    "v_min_f32 v55, |v48|, |v49|\n\t" \
    "v_pk_mul_f32 v[54:55], v[54:55], s[48:49] op_sel_hi:[1,0]\n\t" \
    "v_pk_add_f32 v[52:53], v[52:53], v[54:55]\n\t" \
    "v_sub_f32 v52, v52, v53\n\t" \
    "v_pk_mul_f32 v[48:49], v[42:43], %[tl] op_sel_hi:[0,1]\n\t" \
    "v_readlane_b32 s54, v52, 63\n\t" \
    "v_pk_mul_f32 v[50:51], v[42:43], %[tl] op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
    "v_pk_add_f32 v[48:49], v[48:49], v[50:51] neg_lo:[0,1]\n\t" \
    "v_pk_mul_f32 v[50:51], v[42:43], %[th] op_sel_hi:[0,1]\n\t" \
    "v_pk_mul_f32 v[52:53], v[42:43], %[th] op_sel:[1,1] op_sel_hi:[1,0]\n\t" \
    "v_pk_add_f32 v[50:51], v[50:51], v[52:53] neg_lo:[0,1]\n\t" \
    "v_mul_f32 v52, s54, %[beta]\n\t" \
    "v_add_f32 %[fr], %[fr], v52\n\t" \
    "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t" \
    "v_add_f32 %[ph], %[ph], %[fr]\n\t" \
    "v_add_f32 v52, s52, %[ph]\n\t" \
    "v_cmp_lt_f32 vcc, s49, %[ph]\n\t" \
    "v_cndmask_b32 %[ph], %[ph], v52, vcc\n\t" \
    "v_add_f32 v52, s51, %[ph]\n\t" \
    "v_cmp_gt_f32 vcc, s50, %[ph]\n\t" \
    "v_cndmask_b32 %[ph], %[ph], v52, vcc\n\t" \
    "v_add_u32 %[xa], 8, %[xa]\n\t" \
    "v_add_f32_dpp v56, v56, v48 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp v57, v57, v49 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp v58, v58, v50 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp v59, v59, v51 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__global__ __launch_bounds__(64) void dvbs_fll_kernel(const DvbsStreamWork* __restrict__ work, DvbsLoopCoefs co,
                                                      const cf32* __restrict__ bandedge, int sub, int nsub) {
    __shared__ cf32 tile[128];                   // [0, 64): rotated samples out (x), [64, 128): samples in (y) -- the loop below addresses both from one register
    cf32* const xtile = tile;
    cf32* const ytile = tile + 64;
    const int lane = threadIdx.x;
    DvbsStreamWork w = work[blockIdx.x];
    DvbsStreamState* st = w.st;
    int lo, hi;
    fe_sub_range_dvbs(w.count, sub, nsub, lo, hi);            // time slice of the call (dvbs_frontend_launch)
    w.buf_a += lo; w.buf_b += lo;
    const int n = hi - lo, T = co.ntaps, H = T - 1;     // the systolic layout needs T == 65 (checked on the host)
    const cf32 tl = bandedge[lane], th = bandedge[T + lane];
    const cf32 tl_last = bandedge[T - 1], th_last = bandedge[2 * T - 1];
    float phase = st->fll_phase, freq = st->fll_freq;
    // running sums from the delay line: lane k holds, for output n' = m0 + 63 - k, the terms of taps 0..k
    cf32 al{0.f, 0.f}, ah{0.f, 0.f};
    for (int jt = 0; jt <= lane; ++jt) {
        // term jt of output n' uses sample index n' - 64 + jt relative to the new data, i.e. history slot H + (n' - 64 + jt) with n' = 63 - lane
        const cf32 xs = st->fll_hist[H + (63 - lane) - 64 + jt];
        al = cadd(al, cmul(xs, bandedge[jt]));
        ah = cadd(ah, cmul(xs, bandedge[T + jt]));
    }
    const f32x2 tlv{tl.re, tl.im}, thv{th.re, th.im};
    const f32x2 tlL{__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tl_last.re))), __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tl_last.im)))};
    const f32x2 thL{__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, th_last.re))), __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, th_last.im)))};
    const float beta_v = co.fll_beta;
    const float minf_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, co.fll_min_freq)));
    const float maxf_v = co.fll_max_freq;
    for (int base = 0; base < n; base += 64) {
        const int m = min(64, n - base);
        __syncthreads();
        if (lane < m) ytile[lane] = w.buf_a[base + lane];
        __syncthreads();
        // ONE WAVE ALONE ON ITS SIMD issues one instruction per ~4.5 (4-byte encodings) to ~5.5 cycles (packed / VOP3 / literal operands), dependent or
        // not (tools/ubench/lone_wave.hip): this loop's time is its instruction count.  The compiler's form of it was 95 instructions per sample; written
        // out it is 66 -- a complex product as two packed multiplications and ONE packed addition with neg_lo (the compiler builds two additions and
        // register moves), fastAmplitude as max + 0.4 min with |.| operand modifiers, the systolic shift as the DPP operand of the addition that uses it,
        // compares through VCC (4-byte encodings, no wait states), constants in scalar registers, four samples per trip of the loop.  Every operation and
        // every rounding is the C++ form's (phasor = dvbs2m::sincosf_det(-phase), cmul, cadd, fast_amplitude, PhaseControlLoop::advance with alpha = 0, pcl_wrap_pi):
        // the DVB-S tests compare symbols, loop state and decoded bits with the oracle.
        uint32_t xa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) cf32*)xtile;
        const uint32_t mu = (uint32_t)__builtin_amdgcn_readfirstlane(m), cnt = mu & 3u;
        uint32_t n4 = mu >> 2;
        asm volatile(
            "s_mov_b32 s40, 0xbf22f983\n\t"          // -2/pi
            "s_mov_b32 s41, 0xbfc90fdb\n\t"          // -(pi/2 rounded to binary32)
            "s_mov_b32 s42, 0x333bbd2e\n\t"          // -(pi/2 - that)
            "s_mov_b32 s43, 0x80000000\n\t"
            "s_mov_b32 s44, 0x37ccf5ce\n\t"          // polynomial coefficients (cos, sin): degree 2 ...
            "s_mov_b32 s45, 0xb94ca1f9\n\t"
            "v_mov_b32 v40, 0xbab6061a\n\t"          // ... degree 1 ...
            "v_mov_b32 v41, 0x3c08839e\n\t"
            "s_mov_b32 s46, 0x3d2aaaa5\n\t"          // ... degree 0
            "s_mov_b32 s47, 0xbe2aaaa3\n\t"
            "s_mov_b32 s48, 0x3ecccccd\n\t"          // 0.4f
            "s_mov_b32 s49, 0x40490fdb\n\t"          // pi
            "s_mov_b32 s50, 0xc0490fdb\n\t"          // -pi
            "s_mov_b32 s51, 0x40c90fdb\n\t"          // 2 pi
            "s_mov_b32 s52, 0xc0c90fdb\n\t"          // -2 pi
            "v_mov_b32 v56, %[alr]\n\t"              // the running sums in register PAIRS (packed additions below), moved back behind the loop
            "v_mov_b32 v57, %[ali]\n\t"
            "v_mov_b32 v58, %[ahr]\n\t"
            "v_mov_b32 v59, %[ahi]\n\t"
            // four samples per trip of the loop (a taken branch costs a lone wave ~28 cycles: tools/ubench/branch.hip), then the tile's last 0..3
            "s_cmp_eq_u32 %[n4], 0\n\t"
            "s_cbranch_scc1 2f\n\t"
            "1:\n\t"
            DVBS_FLL_SAMPLE DVBS_FLL_SAMPLE DVBS_FLL_SAMPLE DVBS_FLL_SAMPLE
            "s_sub_u32 %[n4], %[n4], 1\n\t"
            "s_cmp_lg_u32 %[n4], 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            "s_cmp_lt_u32 %[cnt], 1\n\t"
            "s_cbranch_scc1 3f\n\t"
            DVBS_FLL_SAMPLE
            "s_cmp_lt_u32 %[cnt], 2\n\t"
            "s_cbranch_scc1 3f\n\t"
            DVBS_FLL_SAMPLE
            "s_cmp_lt_u32 %[cnt], 3\n\t"
            "s_cbranch_scc1 3f\n\t"
            DVBS_FLL_SAMPLE
            "3:\n\t"
            "v_mov_b32 %[alr], v56\n\t"
            "v_mov_b32 %[ali], v57\n\t"
            "v_mov_b32 %[ahr], v58\n\t"
            "v_mov_b32 %[ahi], v59\n\t"
            "s_waitcnt lgkmcnt(0)"
            : [ph] "+v"(phase), [fr] "+v"(freq), [alr] "+v"(al.re), [ali] "+v"(al.im), [ahr] "+v"(ah.re), [ahi] "+v"(ah.im), [xa] "+v"(xa), [n4] "+s"(n4)
            : [tl] "v"(tlv), [th] "v"(thv), [tlL] "s"(tlL), [thL] "s"(thL), [beta] "v"(beta_v), [minf] "s"(minf_s), [maxf] "v"(maxf_v), [cnt] "s"(cnt)
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
              "v56", "v57", "v58", "v59", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s54", "vcc", "scc", "memory");
        __syncthreads();
        if (lane < m) w.buf_b[base + lane] = xtile[lane];
    }
    __syncthreads();
    // new delay line = last H samples of [old delay line ++ rotated samples of this call]
    for (int i = lane; i < H; i += 64) {
        const int p = n - H + i;
        const cf32 v = p >= 0 ? w.buf_b[p] : st->fll_hist[H + p];
        __syncthreads();
        xtile[i & 63] = v;
        __syncthreads();
        st->fll_hist[i] = xtile[i & 63];
    }
    if (lane == 0) { st->fll_phase = phase; st->fll_freq = freq; }
}

// The same loop for a bank that fills the GPU: FOUR STREAMS PER WAVE, one row of 16 lanes each, 4 taps and 4 running sums per lane.  A
// wave per stream spends two thirds of its instructions on the loop's scalar part (phasor, amplitudes, frequency update), done 64 lanes
// wide for one stream; with thousands of carriers that is what bounds the kernel (VALU issue, 4 waves per SIMD).  Here the scalar part
// serves four streams at once.  The sums still move one TAP up per sample -- inside a lane from register to register, from a lane's last
// register to its neighbour's first (DPP row_shr:1: rows do not mix) -- so every output accumulates its 65 terms in tap order as before
// and the bits are the wave-per-stream kernel's; the finished sum leaves lane 15 of the row, which computes the error and hands it to
// its row (ds_swizzle).
// SPW = 4: a row of 16 lanes per stream (DPP row_shr:1 stays inside it); SPW = 2: 32 lanes per stream (wave_shr:1, lane 32 cleared by hand).
template <int SPW>
__global__ __launch_bounds__(64) void dvbs_fll4_kernel(const DvbsStreamWork* __restrict__ work, int nstreams, DvbsLoopCoefs co,
                                                       const cf32* __restrict__ bandedge, int sub, int nsub) {
    constexpr int LPS = 64 / SPW, TPL = 64 / LPS, NLD = 64 / LPS;      // lanes per stream, taps per lane, tile samples a lane moves
    __shared__ cf32 ytile[SPW][64];
    __shared__ cf32 xtile[SPW][64];
    const int lane = threadIdx.x, row = lane / LPS, j = lane % LPS;
    const int s = blockIdx.x * SPW + row;
    const bool act = s < nstreams;
    DvbsStreamWork w = work[act ? s : blockIdx.x * SPW];
    DvbsStreamState* st = w.st;
    int lo, hi;
    fe_sub_range_dvbs(w.count, sub, nsub, lo, hi);                       // time slice of the call (dvbs_frontend_launch)
    w.buf_a += lo; w.buf_b += lo;
    const int n = act ? hi - lo : 0, T = co.ntaps, H = T - 1;       // T == 65 (checked on the host)
    cf32 tl[TPL], th[TPL], al[TPL], ah[TPL];
#pragma unroll
    for (int q = 0; q < TPL; ++q) { tl[q] = bandedge[TPL * j + q]; th[q] = bandedge[T + TPL * j + q]; }
    const cf32 tl_last = bandedge[T - 1], th_last = bandedge[2 * T - 1];
    float phase = st->fll_phase, freq = st->fll_freq;
    // running sums from the delay line: tap position p = TPL j + q holds, for output 63 - p of the new data, the terms of taps 0..p
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
        const int p = TPL * j + q;
        cf32 a{0.f, 0.f}, b{0.f, 0.f};
        for (int jt = 0; jt <= p; ++jt) {
            const cf32 xs = st->fll_hist[H - 1 - p + jt];
            a = cadd(a, cmul(xs, bandedge[jt]));
            b = cadd(b, cmul(xs, bandedge[T + jt]));
        }
        al[q] = a; ah[q] = b;
    }
    int nmax = n;
#pragma unroll
    for (int o = LPS; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    // the finished sum leaves the stream's last lane: its error goes to the whole group (ds_swizzle: lane (l & ~(LPS - 1)) | (LPS - 1) of each 32)
    constexpr int SWZ = ((32 - 1) & ~(LPS - 1)) | ((LPS - 1) << 5);
    __builtin_amdgcn_s_setprio(FE_PRIO);       // the bank's longest latency chain: ahead of the kernels that fill the SIMDs beside it
    for (int base = 0; base < nmax; base += 64) {
        const int m = min(64, n - base), mmax = min(64, nmax - base);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NLD; ++t) { const int i = j + LPS * t; if (i < m) ytile[row][i] = w.buf_a[base + i]; }
        __syncthreads();
        for (int k = 0; k < mmax; ++k) {
            if (k < m) {                                     // (whole lane groups: the shifts and the swizzle stay inside a stream)
                const cf32 x = cmul_pk3(ytile[row][k], phasor_hw(-phase));
                // this sample's two outputs: the sum leaving the last tap position + newest sample * tap 64 (the stream's last lane)
                const cf32 lo = cadd(al[TPL - 1], cmul_pk3(x, tl_last)), hi = cadd(ah[TPL - 1], cmul_pk3(x, th_last));
                float err = fast_amplitude(hi) - fast_amplitude(lo);
                err = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, err), SWZ));
                freq += co.fll_beta * err;
                freq = clamp_med3(freq, co.fll_min_freq, co.fll_max_freq);
                phase += freq;
                pcl_wrap_pi(phase);
                if (j == 0) xtile[row][k] = x;
                cf32 sl, sh;
                if constexpr (LPS == 16) {                   // row_shr:1, 0 into the row's lane 0
                    sl.re = DPP_F(al[TPL - 1].re, 0x111); sl.im = DPP_F(al[TPL - 1].im, 0x111);
                    sh.re = DPP_F(ah[TPL - 1].re, 0x111); sh.im = DPP_F(ah[TPL - 1].im, 0x111);
                } else {                                     // wave_shr:1 (0 into lane 0), and 0 into the first lane of the other streams
                    sl.re = DPP_F(al[TPL - 1].re, 0x138); sl.im = DPP_F(al[TPL - 1].im, 0x138);
                    sh.re = DPP_F(ah[TPL - 1].re, 0x138); sh.im = DPP_F(ah[TPL - 1].im, 0x138);
                    if (j == 0) { sl = cf32{0.f, 0.f}; sh = cf32{0.f, 0.f}; }
                }
#pragma unroll
                for (int q = TPL - 1; q > 0; --q) { al[q] = cadd(al[q - 1], cmul_pk3(x, tl[q])); ah[q] = cadd(ah[q - 1], cmul_pk3(x, th[q])); }
                al[0] = cadd(sl, cmul_pk3(x, tl[0]));
                ah[0] = cadd(sh, cmul_pk3(x, th[0]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NLD; ++t) { const int i = j + LPS * t; if (i < m) w.buf_b[base + i] = xtile[row][i]; }
    }
    __syncthreads();
    // new delay line = last H samples of [old delay line ++ rotated samples of this call] (read everything, then write)
    cf32 nh[NLD];
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
        const int i = j + LPS * t, p = n - H + i;
        nh[t] = (act && i < H) ? (p >= 0 ? w.buf_b[p] : st->fll_hist[H + p]) : cf32{0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NLD; ++t) { const int i = j + LPS * t; if (act && i < H) st->fll_hist[i] = nh[t]; }
    if (act && j == 0) { st->fll_phase = phase; st->fll_freq = freq; }
}

// RRC FIR at the input rate (SDR++ filter::FIR, taps accumulated in order); grid (x: sample tiles, y: stream); in = buf_b, out = buf_a
__global__ __launch_bounds__(256) void dvbs_rrc_kernel(const DvbsStreamWork* __restrict__ work, const float* __restrict__ taps_g, int ntaps, int sub, int nsub) {
    __shared__ float taps[RRC_MAX_TAPS];
    for (int i = threadIdx.x; i < ntaps; i += 256) taps[i] = taps_g[i];
    __syncthreads();
    DvbsStreamWork w = work[blockIdx.y];
    const DvbsStreamState* st = w.st;
    int lo, hi;
    fe_sub_range_dvbs(w.count, sub, nsub, lo, hi);
    w.buf_a += lo; w.buf_b += lo;
    const int n = hi - lo, H = ntaps - 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        cf32 acc{0.f, 0.f};
        for (int k = 0; k < ntaps; ++k) {
            const int p = i + k;
            const cf32 v = p < H ? st->rrc_hist[p] : w.buf_b[p - H];
            acc.re += v.re * taps[k];
            acc.im += v.im * taps[k];
        }
        w.buf_a[i] = acc;
    }
}
__global__ __launch_bounds__(128) void dvbs_rrc_state_kernel(const DvbsStreamWork* __restrict__ work, int ntaps, int sub, int nsub) {
    DvbsStreamWork w = work[blockIdx.x];
    DvbsStreamState* st = w.st;
    int lo, hi;
    fe_sub_range_dvbs(w.count, sub, nsub, lo, hi);
    w.buf_b += lo;
    const int n = hi - lo, H = ntaps - 1;
    __shared__ cf32 nh[RRC_MAX_TAPS];
    for (int i = threadIdx.x; i < H; i += 128) {
        const int p = n + i;
        nh[i] = p < H ? st->rrc_hist[p] : w.buf_b[p - H];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 128) st->rrc_hist[i] = nh[i];
}

// Three 256-tap complex x real dot products over the wave (the interpolated sample and its two neighbours in phase: the same 256
// samples against three adjacent bank rows).  4 taps per lane in order, then the engine's documented tree for COMPLEX_FD
// (oracle/dvbs_fe.cpp fd_dot): inside each row of 16 lanes pairwise at distance 8, 4, 2, 1, then (row 0 + row 1) + (row 2 + row 3).
// The six sums (re / im of the three products) go through that tree TOGETHER: after the step at distance d only half of the lanes of a
// sum still matter, so the other half carries another sum -- re in lanes 0..7 and im in lanes 8..15 of every row after distance 8, the
// next product in the lanes with bit 2 set after distance 4, the third in the lanes with bit 1 set after distance 2: 17 DPP additions
// and selects instead of 24 additions, and ONE register goes through the cross-row steps (gfx950's v_permlane16_swap / v_permlane32_swap:
// no LDS crossbar, no v_readlane per row leader) instead of six.  Every pair of operands is the pair the plain tree adds, so the bits
// are the plain tree's.  (First version: distances 32 and 16 first, twelve ds_bpermute per symbol; second: one tree per sum, four
// v_readlane + three additions each, 63 instructions where this has 29.)
typedef float fd_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fd_rows_sum(float u) {
    const int ui = __builtin_bit_cast(int, u);
    const auto h = __builtin_amdgcn_permlane16_swap(ui, ui, false, false);      // {row 0, row 0, row 2, row 2}, {row 1, row 1, row 3, row 3}
    const float s = __builtin_bit_cast(float, (int)h[0]) + __builtin_bit_cast(float, (int)h[1]);
    const int si = __builtin_bit_cast(int, s);
    const auto g = __builtin_amdgcn_permlane32_swap(si, si, false, false);      // {rows 0 + 1} x 4, {rows 2 + 3} x 4
    return __builtin_bit_cast(float, (int)g[0]) + __builtin_bit_cast(float, (int)g[1]);
}
__device__ __forceinline__ void fd_dot3_wave(const cf32 (&x)[4], const float* t0, const float* tp, const float* tm, int lane,
                                             cf32& o, cf32& p, cf32& m) {
    fd_f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
    // (a row's taps lie lane-major in LDS: lane l finds its taps l, l + 64, l + 128, l + 192 side by side -- one ds_read_b128 per row)
    const float4 v0 = *reinterpret_cast<const float4*>(t0 + 4 * lane), vp = *reinterpret_cast<const float4*>(tp + 4 * lane),
                 vm = *reinterpret_cast<const float4*>(tm + 4 * lane);
    const float c0[4] = {v0.x, v0.y, v0.z, v0.w}, cp[4] = {vp.x, vp.y, vp.z, vp.w}, cm[4] = {vm.x, vm.y, vm.z, vm.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const fd_f2 xv = {x[q].re, x[q].im};
        a0 = a0 + xv * c0[q];
        a1 = a1 + xv * cp[q];
        a2 = a2 + xv * cm[q];
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    // (every DPP read is outside the selects: all lanes execute it)
    // distance 8 (row_ror:8 pairs lane l with lane l ^ 8): re sums in lanes 0..7, im sums in lanes 8..15
    const float x0 = a0.x + DPP_F(a0.x, 0x128), y0 = a0.y + DPP_F(a0.y, 0x128);
    const float x1 = a1.x + DPP_F(a1.x, 0x128), y1 = a1.y + DPP_F(a1.y, 0x128);
    const float x2 = a2.x + DPP_F(a2.x, 0x128), y2 = a2.y + DPP_F(a2.y, 0x128);
    const float r0 = b3 ? y0 : x0, r1 = b3 ? y1 : x1, r2 = b3 ? y2 : x2;
    // distance 4: o in the lanes with bit 2 clear (row_shl:4: lane l reads l + 4), p in those with bit 2 set (row_shr:4); m apart
    const float s0 = r0 + DPP_F(r0, 0x104), s1 = r1 + DPP_F(r1, 0x114);
    const float s = b2 ? s1 : s0;
    const float t = r2 + DPP_F(r2, 0x104);
    // distance 2: o / p in the lanes with bit 1 clear, m in those with bit 1 set; distance 1
    const float u0 = s + DPP_F(s, 0x102), u1 = t + DPP_F(t, 0x112);
    float u = b1 ? u1 : u0;
    u = u + DPP_F(u, 0x101);
    u = fd_rows_sum(u);
    o.re = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u), 0));
    o.im = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u), 8));
    p.re = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u), 4));
    p.im = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u), 12));
    m.re = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u), 2));
    m.im = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u), 10));
}
// phase moves slowly, so a window of FD_WROWS consecutive bank rows is kept in LDS and re-centred when the phase leaves it.
constexpr int FD_TILE = 256;
constexpr int FD_WROWS = 4;
__global__ __launch_bounds__(64) void dvbs_fd_kernel(const DvbsStreamWork* __restrict__ work, DvbsLoopCoefs co,
                                                            const float* __restrict__ bank, int sub, int nsub) {
    __shared__ cf32 win[FD_TILE + FD_TAPS];      // [255 history][tile]
    __shared__ cf32 ostage[FD_TILE / 2 + 72];
    __shared__ float odump[2];                    // where the lanes that hold no part of a symbol store (the written-out loop below)
    __shared__ __attribute__((aligned(16))) float brow[FD_WROWS * FD_TAPS];   // bank rows [wlo, wlo + FD_WROWS), each lane-major: [lane][tap lane + 64 q]
    const int lane = threadIdx.x;
    DvbsStreamWork w = work[blockIdx.x];
    DvbsStreamState* st = w.st;
    int lo, hi;
    fe_sub_range_dvbs(w.count, sub, nsub, lo, hi);
    w.buf_a += lo;
    const int n = hi - lo;
    PclDev pcl{co.fd_alpha, co.fd_beta, st->fd_phase, st->fd_freq, co.fd_min_freq, co.fd_max_freq};
    int offset = st->fd_offset, spsctr = st->fd_spsctr, outCount = sub ? st->n_sym : 0;   // (later slices append to the call's symbols)
    int wlo = -1000;                             // no window yet
    for (int i = lane; i < FD_TAPS - 1; i += 64) win[i] = st->fd_hist[i];
    __syncthreads();
    __builtin_amdgcn_s_setprio(FE_PRIO);               // latency-critical serial loop (see agc_pc_kernel)
    constexpr int OCAP = FD_TILE / 2 + 72;             // symbols a tile can produce at most (samples per symbol x (1 - limit) >= 1.5, checked on the host: <= 171)
    const uint32_t win_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) cf32*)win;
    const uint32_t ost_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) cf32*)ostage;
    const uint32_t brow_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)brow;
    const uint32_t dump_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)odump;
    const float alpha_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, co.fd_alpha)));
    const float beta_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, co.fd_beta)));
    const float minf_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, co.fd_min_freq)));
    const float maxf_v = co.fd_max_freq;
    for (int base = 0; base < n; base += FD_TILE) {
        const int m = min(FD_TILE, n - base);
        for (int i = lane; i < m; i += 64) win[FD_TAPS - 1 + i] = w.buf_a[base + i];
        __syncthreads();
        int nout = 0;
        // one symbol the general way: any phase (the one-sided derivative at the ends of the bank), the row window re-centred where the phase has left it
        auto symbol_general = [&]() {
            int phase = (int)floorf(pcl.phase * (float)FD_PHASES);
            phase = phase < 0 ? 0 : (phase > FD_PHASES - 1 ? FD_PHASES - 1 : phase);
            const int pm = phase > 0 ? phase - 1 : phase, pp = phase < FD_PHASES - 1 ? phase + 1 : phase;
            if (pm < wlo || pp > wlo + FD_WROWS - 1) {                       // re-centre the row window (uniform branch)
                wlo = phase - FD_WROWS / 2;
                wlo = wlo < 0 ? 0 : (wlo > FD_PHASES - FD_WROWS ? FD_PHASES - FD_WROWS : wlo);
                __syncthreads();
                for (int i = lane; i < FD_WROWS * FD_TAPS; i += 64) brow[(i & ~255) + 4 * lane + ((i >> 6) & 3)] = bank[(size_t)wlo * FD_TAPS + i];
                __syncthreads();
            }
            cf32 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = win[(offset - base) + lane + 64 * q];
            // the interpolated sample and, for the derivative of the signal, its neighbours in phase (complex_fd.cpp:103-120)
            cf32 outVal, fT1, fT_1;
            fd_dot3_wave(x, brow + (phase - wlo) * FD_TAPS, brow + (pp - wlo) * FD_TAPS, brow + (pm - wlo) * FD_TAPS, lane, outVal, fT1, fT_1);
            cf32 dfdt = cscale(csub(fT1, fT_1), 0.5f);
            if (__builtin_expect(phase == 0 || phase == FD_PHASES - 1, 0))         // (one-sided at the ends of the bank: rare, out of line)
                dfdt = phase == 0 ? csub(fT1, outVal) : csub(outVal, fT_1);
            float error = spsctr == 0 ? ((outVal.re * dfdt.re) + (outVal.im * dfdt.im)) : 0.f;
            spsctr++;
            if (spsctr >= 1) spsctr = 0;                 // outSps = 1 (qpsk_alt.cpp:22)
            error = clamp_med3(error, -1.0f, 1.0f);
            pcl.advance(error);
            const float delta = floorf(pcl.phase);
            offset = (int)((float)offset + delta);       // `offset += delta` with an int offset and a float delta
            pcl.phase -= delta;
            if (FD_STORE_ALL || lane == 0) ostage[nout] = outVal;       // (the Costas loop has its own kernel, below)
            ++nout;
        };
        while (offset < base + m && nout < OCAP) {
            if (spsctr == 0 && wlo >= 0) {
                // THE FAST PATH, written out (a wave alone on its SIMD pays per instruction, tools/ubench/lone_wave.hip; the compiler's form of this loop is ~140
                // instructions and three taken branches per symbol, this one 95): symbols whose phase row lies inside the window with both neighbours
                // and inside the bank (1 <= phase <= 254) -- everything else leaves the loop BEFORE the symbol is touched and goes through symbol_general.
                // Same operations, same order, same roundings as fd_dot3_wave + the loop body above: three 256-tap dot products (four taps per lane in
                // tap order, then the lane tree: l ^ 8, + 4, + 2, + 1, rows), o in lanes 0 / 8, fT1 in 4 / 12, fT_1 in 2 / 10; the derivative, the
                // error products and their sum stay in those lanes (DPP) and ONE v_readlane brings the error back; lanes 0 and 8 store the symbol.
                const int wlo_u = __builtin_amdgcn_readfirstlane(wlo);               // (wave-uniform by construction; the compiler sees a vector value)
                const int lo_ok = max(wlo_u + 1, 1), hi_ok = min(wlo_u + FD_WROWS - 2, FD_PHASES - 2);
                const uint32_t xb = win_a + 8u * (uint32_t)(lane - base);
                const uint32_t rb = brow_a + 16u * (uint32_t)lane - 1024u * (uint32_t)(wlo + 1);
                uint32_t oa = (lane == 0 ? ost_a + 8u * nout : lane == 8 ? ost_a + 8u * nout + 4u : dump_a) - (lane == 0 || lane == 8 ? 8u : 0u);
                const uint32_t incv = lane == 0 || lane == 8 ? 8u : 0u;
                uint32_t nneg = (uint32_t)__builtin_amdgcn_readfirstlane(nout - OCAP), why = 0;
                const int lim = __builtin_amdgcn_readfirstlane(base + m);
                asm volatile(
                    "s_mov_b32 s60, 0x43800000\n\t"                   // 256.0f
                    "s_movk_i32 s61, 0xff\n\t"
                    "1:\n\t"
                    "v_lshl_add_u32 v88, %[off], 3, %[xb]\n\t"
                    "ds_read2st64_b64 v[60:63], v88 offset1:1\n\t"                           // x[lane], x[lane + 64]
                    "ds_read2st64_b64 v[64:67], v88 offset0:2 offset1:3\n\t"                 // x[lane + 128], x[lane + 192]
                    "v_mul_f32 v89, s60, %[ph]\n\t"
                    "v_floor_f32 v89, v89\n\t"
                    "v_cvt_i32_f32 v89, v89\n\t"
                    "v_med3_i32 v89, v89, 0, s61\n\t"                                        // the phase row
                    "v_subrev_u32 v90, %[slo], v89\n\t"
                    "v_cmp_lt_u32 vcc, %[span], v90\n\t"
                    "s_cbranch_vccnz 3f\n\t"                                                 // outside [lo_ok, hi_ok]: the general way
                    "v_lshl_add_u32 v91, v89, 10, %[rb]\n\t"
                    "ds_read_b128 v[72:75], v91 offset:1024\n\t"                             // row phase
                    "ds_read_b128 v[76:79], v91 offset:2048\n\t"                             // row phase + 1
                    "ds_read_b128 v[68:71], v91\n\t"                                         // row phase - 1
                    "v_cvt_f32_i32 v92, %[off]\n\t"
                    "s_waitcnt lgkmcnt(2)\n\t"
                    "v_pk_mul_f32 v[80:81], v[60:61], v[72:73] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 v[80:81], v[80:81], 0 op_sel_hi:[1,0]\n\t"
                    "v_pk_mul_f32 v[86:87], v[62:63], v[72:73] op_sel:[0,1]\n\t"
                    "v_pk_add_f32 v[80:81], v[80:81], v[86:87]\n\t"
                    "v_pk_mul_f32 v[86:87], v[64:65], v[74:75] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 v[80:81], v[80:81], v[86:87]\n\t"
                    "v_pk_mul_f32 v[86:87], v[66:67], v[74:75] op_sel:[0,1]\n\t"
                    "v_pk_add_f32 v[80:81], v[80:81], v[86:87]\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_pk_mul_f32 v[82:83], v[60:61], v[76:77] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 v[82:83], v[82:83], 0 op_sel_hi:[1,0]\n\t"
                    "v_pk_mul_f32 v[86:87], v[62:63], v[76:77] op_sel:[0,1]\n\t"
                    "v_pk_add_f32 v[82:83], v[82:83], v[86:87]\n\t"
                    "v_pk_mul_f32 v[86:87], v[64:65], v[78:79] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 v[82:83], v[82:83], v[86:87]\n\t"
                    "v_pk_mul_f32 v[86:87], v[66:67], v[78:79] op_sel:[0,1]\n\t"
                    "v_pk_add_f32 v[82:83], v[82:83], v[86:87]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_pk_mul_f32 v[84:85], v[60:61], v[68:69] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 v[84:85], v[84:85], 0 op_sel_hi:[1,0]\n\t"
                    "v_pk_mul_f32 v[86:87], v[62:63], v[68:69] op_sel:[0,1]\n\t"
                    "v_pk_add_f32 v[84:85], v[84:85], v[86:87]\n\t"
                    "v_pk_mul_f32 v[86:87], v[64:65], v[70:71] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_f32 v[84:85], v[84:85], v[86:87]\n\t"
                    "v_pk_mul_f32 v[86:87], v[66:67], v[70:71] op_sel:[0,1]\n\t"
                    "v_pk_add_f32 v[84:85], v[84:85], v[86:87]\n\t"
                    // lane tree; a register a DPP operand reads was written at least two instructions earlier
                    "v_add_f32_dpp v80, v80, v80 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v81, v81, v81 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v82, v82, v82 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v83, v83, v83 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v84, v84, v84 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v85, v85, v85 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_cndmask_b32_e64 v80, v80, v81, %[b3]\n\t"
                    "v_cndmask_b32_e64 v82, v82, v83, %[b3]\n\t"
                    "v_cndmask_b32_e64 v84, v84, v85, %[b3]\n\t"
                    "v_add_f32_dpp v81, v80, v80 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v83, v82, v82 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_add_f32_dpp v85, v84, v84 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_cndmask_b32_e64 v81, v81, v83, %[b2]\n\t"
                    "v_add_u32 %[oa], %[oa], %[incv]\n\t"
                    "v_add_f32_dpp v85, v85, v85 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "s_nop 0\n\t"
                    "v_add_f32_dpp v81, v81, v81 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_cndmask_b32_e64 v81, v81, v85, %[b1]\n\t"
                    "s_nop 1\n\t"
                    "v_add_f32_dpp v81, v81, v81 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b32 v82, v81\n\t"
                    "s_nop 1\n\t"
                    "v_permlane16_swap_b32 v81, v82\n\t"
                    "v_add_f32 v81, v81, v82\n\t"
                    "v_mov_b32 v82, v81\n\t"
                    "s_nop 1\n\t"
                    "v_permlane32_swap_b32 v81, v82\n\t"
                    "v_add_f32 v81, v81, v82\n\t"                                            // lanes 0 / 8: o, 4 / 12: fT1, 2 / 10: fT_1
                    "ds_write_b32 %[oa], v81\n\t"
                    "s_nop 0\n\t"
                    "v_subrev_f32_dpp v82, v81, v81 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"     // lanes 4 / 12: fT1 - fT_1
                    "v_mul_f32 v82, 0.5, v82\n\t"
                    "s_nop 1\n\t"
                    "v_mul_f32_dpp v83, v82, v81 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"        // lanes 0 / 8: o.re dfdt.re, o.im dfdt.im
                    "s_nop 1\n\t"
                    "v_add_f32_dpp v83, v83, v83 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "s_add_u32 %[nneg], %[nneg], 1\n\t"                                      // (carry out: the tile's symbol store is full)
                    "v_readlane_b32 s62, v83, 0\n\t"
                    "v_med3_f32 v83, s62, -1.0, 1.0\n\t"
                    "v_mul_f32 v84, %[beta], v83\n\t"
                    "v_add_f32 %[fr], %[fr], v84\n\t"
                    "v_med3_f32 %[fr], %[fr], %[minf], %[maxf]\n\t"
                    "v_mul_f32 v84, %[alpha], v83\n\t"
                    "v_add_f32 v84, v84, %[fr]\n\t"
                    "v_add_f32 %[ph], %[ph], v84\n\t"
                    "v_floor_f32 v84, %[ph]\n\t"
                    "v_add_f32 v92, v92, v84\n\t"
                    "v_cvt_i32_f32 %[off], v92\n\t"
                    "v_sub_f32 %[ph], %[ph], v84\n\t"
                    "v_cmp_gt_i32 vcc, %[lim], %[off]\n\t"
                    "s_cbranch_scc1 4f\n\t"
                    "s_cbranch_vccnz 1b\n\t"
                    "s_branch 4f\n\t"
                    "3:\n\t"
                    "s_mov_b32 %[why], 1\n\t"
                    "4:\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : [ph] "+v"(pcl.phase), [fr] "+v"(pcl.freq), [off] "+v"(offset), [oa] "+v"(oa), [nneg] "+s"(nneg), [why] "+s"(why)
                    : [xb] "v"(xb), [rb] "v"(rb), [incv] "v"(incv), [maxf] "v"(maxf_v), [slo] "s"(lo_ok), [span] "s"(hi_ok - lo_ok), [lim] "s"(lim),
                      [alpha] "s"(alpha_s), [beta] "s"(beta_s), [minf] "s"(minf_s),
                      [b3] "s"(0xFF00FF00FF00FF00ull), [b2] "s"(0xF0F0F0F0F0F0F0F0ull), [b1] "s"(0xCCCCCCCCCCCCCCCCull)
                    : "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "s60", "s61", "s62", "vcc", "scc", "memory");
                nout = OCAP + (int)nneg;
                if (!why) continue;                      // (the tile is used up or its symbol store is full: the loop condition ends it)
            }
            symbol_general();
        }
        __syncthreads();
        for (int i = lane; i < nout; i += 64) w.sym[outCount + i] = ostage[i];
        outCount += nout;
        // slide the 255-sample history
        cf32 h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = lane + 64 * q; h[q] = i < FD_TAPS - 1 ? win[m + i] : cf32{0.f, 0.f}; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = lane + 64 * q; if (i < FD_TAPS - 1) win[i] = h[q]; }
        __syncthreads();
    }
    for (int i = lane; i < FD_TAPS - 1; i += 64) st->fd_hist[i] = win[i];
    if (lane == 0) {
        st->fd_phase = pcl.phase; st->fd_freq = pcl.freq; st->fd_offset = offset - n; st->fd_spsctr = spsctr;
        st->n_sym = outCount;
        st->n_sym_slice[sub] = outCount;
    }
}

// Costas<4> (SDR++ loop/costas.h: derotate, decision-directed QPSK error, clamp, advance) over the symbols the timing recovery's slice
// `sub` produced, in place.  LANE = STREAM: the loop only depends on the symbol sequence, not on the timing loop, so it does not belong in
// that kernel's wave-wide instruction stream (it was a third of it: 75 of 210 instructions per symbol, executed by 64 lanes for one result).
// Here a bank's 64 streams share a wave, and with few carriers the slice runs on the Viterbi stream beside the next timing-recovery slice.
// Symbols move in batches of 8 per lane (one 64-byte line), the next batch in flight while this one goes through the recurrence.
__global__ __launch_bounds__(64) void dvbs_costas_kernel(const DvbsStreamWork* __restrict__ work, int nstreams, DvbsLoopCoefs co, int sub, int nsub) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nstreams) return;
    const DvbsStreamWork w = work[s];
    DvbsStreamState* st = w.st;
    const int first = (nsub > 1 && sub) ? st->n_sym_slice[sub - 1] : 0, last = nsub > 1 ? st->n_sym_slice[sub] : st->n_sym;
    PclDev cos{co.cos_alpha, co.cos_beta, st->costas_phase, st->costas_freq, co.cos_min_freq, co.cos_max_freq};
    auto one = [&](cf32 x) -> cf32 {
        const cf32 v = cmul(x, phasor_hw(-cos.phase));
        float cerr = ((v.re > 0 ? 1.0f : -1.0f) * v.im) - ((v.im > 0 ? 1.0f : -1.0f) * v.re);
        cerr = clamp_med3(cerr, -1.0f, 1.0f);
        cos.advance(cerr);
        pcl_wrap_pi(cos.phase);
        return v;
    };
    constexpr int B = 8;
    __builtin_amdgcn_s_setprio(FE_PRIO);
    cf32 cur[B], nxt[B];
    int i = first;
    if (i + B <= last) {
#pragma unroll
        for (int k = 0; k < B; ++k) cur[k] = w.sym[i + k];
    }
    for (; i + B <= last; i += B) {
        if (i + 2 * B <= last) {
#pragma unroll
            for (int k = 0; k < B; ++k) nxt[k] = w.sym[i + B + k];
        }
#pragma unroll
        for (int k = 0; k < B; ++k) w.sym[i + k] = one(cur[k]);
#pragma unroll
        for (int k = 0; k < B; ++k) cur[k] = nxt[k];
    }
    for (; i < last; ++i) w.sym[i] = one(w.sym[i]);
    st->costas_phase = cos.phase; st->costas_freq = cos.freq;
}

// DVBSymToSoftBlock: symbols -> int8 soft pairs appended to the stream's block FIFO; grid (x: tiles, y: stream)
// sub / nsub: the symbols the timing recovery's slice `sub` produced (nsub == 1: the whole call)
__global__ __launch_bounds__(256) void dvbs_soft_fifo_kernel(const DvbsStreamWork* __restrict__ work, int sub, int nsub) {
    const DvbsStreamWork w = work[blockIdx.y];
    const DvbsStreamState* st = w.st;
    const int first = (nsub > 1 && sub) ? st->n_sym_slice[sub - 1] : 0, ns = nsub > 1 ? st->n_sym_slice[sub] : st->n_sym, fill = st->soft_fill;
    const float* __restrict__ sy = reinterpret_cast<const float*>(w.sym);
    for (int i = 2 * first + blockIdx.x * 256 + threadIdx.x; i < 2 * ns; i += gridDim.x * 256) {
        const float x = sy[i] * 100;
        w.soft[fill + i] = x < -127.0f ? (int8_t)-127 : (x > 127.0f ? (int8_t)127 : (int8_t)x);
    }
}
// whole blocks in the FIFO after slice `sub`, and the first one no earlier slice has decoded (the Viterbi launch of the slice takes [blk0, nblk))
__global__ __launch_bounds__(64) void dvbs_soft_avail_kernel(const DvbsStreamWork* __restrict__ work, int nstreams, int sub, int* __restrict__ blk0,
                                                            int* __restrict__ nblk) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nstreams) return;
    DvbsStreamState* st = work[s].st;
    const int avail = (st->soft_fill + 2 * st->n_sym_slice[sub]) / DVBS_SOFT_BLOCK;
    blk0[s] = sub ? st->vit_done : 0;
    nblk[s] = avail;
    st->vit_done = avail;
}
__global__ __launch_bounds__(64) void dvbs_soft_count_kernel(const DvbsStreamWork* __restrict__ work, int nstreams, int* __restrict__ nblocks_out) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nstreams) return;
    DvbsStreamState* st = work[s].st;
    const int fill = st->soft_fill + 2 * st->n_sym;
    st->n_blocks = fill / DVBS_SOFT_BLOCK;
    st->soft_fill = fill;                      // (reduced by the blocks consumed in dvbs_soft_compact_kernel, after the decoder ran)
    nblocks_out[s] = fill / DVBS_SOFT_BLOCK;
}
__global__ __launch_bounds__(256) void dvbs_soft_compact_kernel(const DvbsStreamWork* __restrict__ work) {
    const DvbsStreamWork w = work[blockIdx.x];
    DvbsStreamState* st = w.st;
    const int used = st->n_blocks * DVBS_SOFT_BLOCK, rest = st->soft_fill - used;
    __shared__ int8_t tmp[DVBS_SOFT_BLOCK];
    if (used > 0) {
        for (int i = threadIdx.x; i < rest; i += 256) tmp[i] = w.soft[used + i];
        __syncthreads();
        for (int i = threadIdx.x; i < rest; i += 256) w.soft[i] = tmp[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) st->soft_fill = rest;
}

// ------------------------------------------------------------------------------------------------ math self-test
// evaluates the shared definitions of include/dvbs2gpu_math.h on the device, one element per thread (dvbs2gpu_math_eval)
__global__ __launch_bounds__(256) void math_eval_kernel(int func, int n, const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ o0, float* __restrict__ o1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    switch (func) {
        case 0: { float sn, cs; dvbs2m::sincosf_det(a[i], &sn, &cs); o0[i] = sn; o1[i] = cs; break; }
        case 1: o0[i] = dvbs2m::atan2f_det(a[i], b[i]); break;
        case 2: o0[i] = dvbs2m::expf_det(a[i]); break;
        case 3: o0[i] = dvbs2m::logf_det(a[i]); break;
        case 5: o0[i] = (float)lut_cell(a[i], b[i]); break;       // the loops' and the demapper's table cell (binary32 fast path + double form)
        default: o0[i] = (float)dvbs2m::llr_clamp_det(a[i]); break;
    }
}
hipError_t math_eval_launch(int func, int n, const float* a, const float* b, float* o0, float* o1, hipStream_t st) {
    hipLaunchKernelGGL(math_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, st, func, n, a, b, o0, o1);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t s2_collect_launch(const S2StreamWork* d_work, int nstreams, int* d_nsym, float* d_nco, hipStream_t st) {
    hipLaunchKernelGGL(s2_collect_kernel, dim3((nstreams + 255) / 256), dim3(256), 0, st, d_work, nstreams, d_nsym, d_nco);
    return hipGetLastError();
}
hipError_t s2_scatter_out_launch(const S2StreamWork* d_work, const S2FrameRef* d_frames, const int* d_first, int nframes, int kb,
                                 const uint8_t* d_bb, hipStream_t st) {
    hipLaunchKernelGGL(s2_scatter_out_kernel, dim3(nframes), dim3(256), 0, st, d_work, d_frames, d_first, kb, d_bb);
    return hipGetLastError();
}
hipError_t s2_scatter_out2_launch(uint8_t* const* d_outs, const S2FrameRef* d_frames, const int* d_first, int nframes, int kb,
                                  const uint8_t* d_bb, hipStream_t st) {
    hipLaunchKernelGGL(s2_scatter_out2_kernel, dim3(nframes), dim3(256), 0, st, d_outs, d_frames, d_first, kb, d_bb);
    return hipGetLastError();
}
hipError_t s2_fifo_compact_launch(const S2StreamWork* d_work, int nstreams, const int* d_cur_fill, hipStream_t st) {
    hipLaunchKernelGGL(s2_fifo_compact_kernel, dim3(16, nstreams), dim3(256), 0, st, d_work, d_cur_fill);
    return hipGetLastError();
}
// The serial stages of the DVB-S receiver -- AGC and Costas (lane = stream), band-edge FLL (a wave per stream, or four streams per wave from
// `bank_min` carriers), RRC, timing recovery (a wave per stream) and the Viterbi decoder behind them -- keep their state in the stream record,
// so a call's samples can go through in `nsub` time slices with the stages on three streams: aux[0] runs the AGC slices ahead and later the
// Costas + soft FIFO + Viterbi slices (the hook), aux[1] the FLL and RRC slices, `st` the timing recovery; events between them.  Slice c of a
// stage runs beside slice c+1 of the stage before it: ONE carrier costs the slowest stage instead of the sum (116 -> 39 ms per 131 k samples),
// and in a bank the latency-bound stages (few waves) hide beside the ones that fill the SIMDs.  Three streams + the caller's: HIP's default of
// 4 hardware queues is enough.  nsub = 1: the stages back to back on `st`.  ev: 4 rows of nsub + 1 events.
hipError_t dvbs_frontend_launch(const DvbsStreamWork* d_work, int nstreams, int max_count, DvbsLoopCoefs coefs, const cf32* d_bandedge,
                                const float* d_rrc, const float* d_fd_bank, hipStream_t st, hipStream_t* aux, hipEvent_t (*ev)[DVBS_FE_MAX_SLICES + 1], int nsub, DvbsSliceHook* hook, int bank_min) {
    const dim3 ga((nstreams + 63) / 64);
    const bool sliced = nsub > 1 && aux && ev;
    if (!sliced) nsub = 1;
    int gx = (max_count / nsub + 1 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipError_t e;
    hipStream_t s0 = sliced ? (aux[2] ? aux[2] : aux[0]) : st, s1 = sliced ? aux[1] : st;       // (aux[2], optional: the AGC slices on a stream of their own)
    if (sliced) {
        if ((e = hipEventRecord(ev[0][nsub], st)) != hipSuccess) return e;          // the slices start behind what `st` holds now
        if ((e = hipStreamWaitEvent(s0, ev[0][nsub], 0)) != hipSuccess) return e;
    }
    // The AGC runs ahead (cheapest stage).  A bank: all its slices first.  A few carriers (aux[2]): AGC_AHEAD slices ahead of the FLL, enqueued slice by
    // slice -- the FLL's first launch then leaves the host behind four AGC launches instead of behind all of them (one carrier: 0.4 ms per call), and the
    // AGC never holds it up.
    constexpr int AGC_AHEAD = 4;
    const bool agc_interleaved = sliced && aux[2];
    auto agc_slice = [&](int c) -> hipError_t {
        hipLaunchKernelGGL(agc_pc_kernel<AgcDvbsTraits>, ga, dim3(128), 0, s0, d_work, nstreams, coefs, c, nsub);
        return sliced ? hipEventRecord(ev[0][c], s0) : hipSuccess;
    };
    for (int c = 0; c < (agc_interleaved ? (nsub < AGC_AHEAD ? nsub : AGC_AHEAD) : nsub); ++c)
        if ((e = agc_slice(c)) != hipSuccess) return e;
    for (int c = 0; c < nsub; ++c) {
        if (agc_interleaved && c + AGC_AHEAD < nsub && (e = agc_slice(c + AGC_AHEAD)) != hipSuccess) return e;
        if (sliced && (e = hipStreamWaitEvent(s1, ev[0][c], 0)) != hipSuccess) return e;
        // (two streams per wave, dvbs_fll4_kernel<2>: same bits, 9 % slower at 4096 carriers, 15 % at 8192)
        if (nstreams >= bank_min) hipLaunchKernelGGL(dvbs_fll4_kernel<4>, dim3((nstreams + 3) / 4), dim3(64), 0, s1, d_work, nstreams, coefs, d_bandedge, c, nsub);
        else hipLaunchKernelGGL(dvbs_fll_kernel, dim3(nstreams), dim3(64), 0, s1, d_work, coefs, d_bandedge, c, nsub);
        hipLaunchKernelGGL(dvbs_rrc_kernel, dim3(gx, nstreams), dim3(256), 0, s1, d_work, d_rrc, coefs.ntaps, c, nsub);
        hipLaunchKernelGGL(dvbs_rrc_state_kernel, dim3(nstreams), dim3(128), 0, s1, d_work, coefs.ntaps, c, nsub);
        if (sliced) { if ((e = hipEventRecord(ev[2][c], s1)) != hipSuccess) return e; if ((e = hipStreamWaitEvent(st, ev[2][c], 0)) != hipSuccess) return e; }
        hipLaunchKernelGGL(dvbs_fd_kernel, dim3(nstreams), dim3(64), 0, st, d_work, coefs, d_fd_bank, c, nsub);
        if (sliced && hook) { if ((e = hook->after_timing(c)) != hipSuccess) return e; }  // (Costas, soft FIFO + Viterbi of the slice, on aux[0])
        else hipLaunchKernelGGL(dvbs_costas_kernel, ga, dim3(64), 0, st, d_work, nstreams, coefs, c, nsub);
    }
    if (!(sliced && hook)) {
        int gs = (max_count + 255) / 256;
        gs = gs < 1 ? 1 : (gs > 64 ? 64 : gs);
        hipLaunchKernelGGL(dvbs_soft_fifo_kernel, dim3(gs, nstreams), dim3(256), 0, st, d_work, 0, 1);
    }
    return hipGetLastError();
}
hipError_t dvbs_costas_launch(const DvbsStreamWork* d_work, int nstreams, DvbsLoopCoefs coefs, int sub, int nsub, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_costas_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, st, d_work, nstreams, coefs, sub, nsub);
    return hipGetLastError();
}
hipError_t dvbs_soft_slice_launch(const DvbsStreamWork* d_work, int nstreams, int max_count, int sub, int nsub, int* d_blk0, int* d_nblk, hipStream_t st) {
    int gs = (max_count / nsub + 1 + 255) / 256;
    gs = gs < 1 ? 1 : (gs > 64 ? 64 : gs);
    hipLaunchKernelGGL(dvbs_soft_fifo_kernel, dim3(gs, nstreams), dim3(256), 0, st, d_work, sub, nsub);
    hipLaunchKernelGGL(dvbs_soft_avail_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, st, d_work, nstreams, sub, d_blk0, d_nblk);
    return hipGetLastError();
}
hipError_t dvbs_soft_count_launch(const DvbsStreamWork* d_work, int nstreams, int* d_nblocks, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_soft_count_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, st, d_work, nstreams, d_nblocks);
    return hipGetLastError();
}
hipError_t dvbs_soft_compact_launch(const DvbsStreamWork* d_work, int nstreams, hipStream_t st) {
    hipLaunchKernelGGL(dvbs_soft_compact_kernel, dim3(nstreams), dim3(256), 0, st, d_work);
    return hipGetLastError();
}
// The two serial front-end stages, time-sliced: the call's samples go through in `nsub` slices; the AGC / NCO recurrences (64 streams
// per wave, few waves) run on `aux` one slice ahead of the timing loop (8 streams per wave) on `st`, which waits for each slice's event.
// Both are latency chains that leave most of the GPU idle, so side by side they cost the longer of the two instead of the sum.
// ev: nsub + 1 events (the last one orders `aux` behind what `st` holds when the call starts).  nsub <= 1 or no aux stream: one slice on st.
//
// With `post` (CCM calls of one configuration) the rest of the front half joins the pipeline: behind every timing-recovery slice the auxiliary
// stream -- the AGC slices stay two ahead of the timing loop, then it is free -- appends the slice's symbols to the PL-sync FIFO (RRC + /2),
// walks the windows that are complete and runs the frame loops (FED, PLL, PLHDR: the call's other long latency chain) over the frames found
// so far, while the timing loop works on the next slice.  No host in between: frames stay in per-stream slots (S2PostStages), the host reads
// the frame tables once, after the last slice.  ev2: nsub + 1 more events (timing recovery of slice c done; the last: post stages done).
// which: 1 = RRC + decimation, 2 = PL-sync walk, 4 = frame loops (any combination; 7 = all of a slice on one stream)
static hipError_t post_stages_launch(const S2StreamWork* d_work, int nstreams, const S2LoopCoefs& coefs, const S2PostStages& p, int c, int nsub, hipStream_t s, int which = 7) {
    if (which & 1) {
        int gx = ((p.max_count / nsub) / 2 + 2 + 255) / 256;
        gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
        if (p.spans) p.spans->begin(1, s);
        hipLaunchKernelGGL(s2_rrc_decim_kernel, dim3(gx, nstreams), dim3(256), 0, s, d_work, p.d_taps, p.ntaps, c, nsub | (coefs.post_prio << 16));
        if (c == nsub - 1) hipLaunchKernelGGL(s2_rrc_state_kernel, dim3(nstreams), dim3(128), 0, s, d_work, p.ntaps);
        if (p.spans) p.spans->end(1, s);
    }
    if (which & 2) {
        if (p.spans) p.spans->begin(2, s);
        hipLaunchKernelGGL(s2_ccm_walk_kernel, dim3(nstreams), dim3(256), 0, s, d_work, p.raw, p.maxf, p.d_found, p.d_counts, c, nsub | (coefs.post_prio << 16), p.cfgs);
        if (p.spans) p.spans->end(2, s);
    }
    if (which & 4) {
        const int L = p.loops_launches < 1 ? 1 : (p.loops_launches > nsub ? nsub : p.loops_launches);
        if ((c + 1) * L / nsub > c * L / nsub) {
            if (p.spans) p.spans->begin(3, s);
            const int spw = frame_loops_spw(nstreams);
            if (p.spec && spw == 1)
                hipLaunchKernelGGL(s2_frame_loops_kernel<true>, dim3(nstreams), dim3(64), 0, s, d_work, nstreams, (const S2FrameRef*)nullptr,
                                   (const int*)nullptr, coefs, p.tabs, p.con, p.pls_code, p.slots, p.pilots, p.pilot_blocks, p.raw, p.d_pllout, p.d_stats,
                                   (const S2VcmFound*)p.d_found, p.maxf, spw, p.cfgs);
            else
            hipLaunchKernelGGL(s2_frame_loops_kernel<false>, dim3(frame_loops_grid(nstreams, spw)), dim3(64 * frame_loops_wpb(nstreams, spw)), 0, s, d_work, nstreams, (const S2FrameRef*)nullptr,
                               (const int*)nullptr, coefs, p.tabs, p.con, p.pls_code, p.slots, p.pilots, p.pilot_blocks, p.raw, p.d_pllout, p.d_stats,
                               (const S2VcmFound*)p.d_found, p.maxf, spw, p.cfgs);
            if (p.spans) p.spans->end(3, s);
        }
    }
    return hipGetLastError();
}
hipError_t s2_post_stages_launch(const S2StreamWork* d_work, int nstreams, const S2LoopCoefs& coefs, const S2PostStages& p, int c, int nsub, hipStream_t s) {
    return post_stages_launch(d_work, nstreams, coefs, p, c, nsub, s);
}
// Three forms of the timing recovery, all bit-identical (tests/test_gpu_gardner_forms.py runs every one): 1 = one wave, 8 lanes per stream
// (s2_gardner_kernel); 2 = resolver + producer waves, 8 lanes per stream (s2_gardner2_kernel); 4 = candidate tables (s2_gardner_cand_kernel: the
// shortest chain per stream -- what a small bank needs: one stream 3.68 -> 2.5 ms per 21 690-sample slice against form 2; 64 streams x 1 frame
// 18.6 -> 16.7 ms per call, 256 x 4 frames 50.3 -> 47.2).  (Form 3, lane = stream, is gone: see above.)  Who wins where (round 4, MI355X, ms per
// step, front end | decoder in the step | step):
//   4096 streams x 8 frames 8PSK 3/4 (decoder critical):  form 1  185 | 342 | 363     form 2  147 | 345 | 363
//   4096 streams x 8 frames QPSK 1/2 (front end critical): form 1  283 | 337 | 465     form 2  297 | 356 | 410
//   1024 streams x 4 frames 8PSK 3/4:                      form 1   71 |  48 |  76     form 2   60 |  49 |  65
//    384 streams x 4 frames:                               form 1   65 |  19 |  67     form 2   52 |  18 |  54     form 4  51 | 19 | 53
// Default: form 4 up to S2_GARDNER_CAND_MAX streams; form 2 above (rounds 4-5: form 1 for big banks beside a decoder that is the critical path -- see gardner_form()).
// The context option gardner_form = 1 | 2 | 4 forces one.
#ifndef S2_GARDNER_BANK_MIN
#define S2_GARDNER_BANK_MIN 512
#endif
#ifndef S2_GARDNER_LANE_MIN
#define S2_GARDNER_LANE_MIN 2048
#endif
#ifndef S2_GARDNER_CAND_MAX
#define S2_GARDNER_CAND_MAX 256
#endif
static int gardner_form(int nstreams, int prio_duty, int lane_form, int forced) {
    if (forced == 1 || forced == 2 || forced == 4) return forced;      // (context option gardner_form: the parity tests run every form)
    if (nstreams <= S2_GARDNER_CAND_MAX) return 4;
    if (nstreams < S2_GARDNER_BANK_MIN) return 2;
    // a big bank beside the decoder of the previous call: the one-wave form disturbs the decoder least; once the balancer of the pipelined
    // mode (s2_demod.hip) has found the FRONT END to be the critical path (it raises the timing loop's priority share), the shorter forms win
    // (round 6: form 2's workgroups of four waves spread evenly over a compute unit's SIMDs and its waves take 60 registers -- it now disturbs the decoder less than the one-wave
    //  form everywhere: mixed 64-entry batch, whose shared front-end pass has no balancer, 206 -> 195.5 ms per step; the headline's balancer ends at a share >= 2 anyway)
    (void)lane_form; (void)prio_duty;
    return 2;
}
static void gardner_launch(const S2StreamWork* d_work, int nstreams, const S2LoopCoefs& coefs, const float* d_bank, hipStream_t st, int c, int nsub) {
    switch (gardner_form(nstreams, coefs.g_prio_duty, coefs.g_lane_form, coefs.g_form)) {
        case 4:
            hipLaunchKernelGGL(s2_gardner_cand_kernel, dim3((nstreams + GC_CS - 1) / GC_CS), dim3(192), 0, st, d_work, nstreams, coefs, d_bank, c, nsub, coefs.g_cand_skew);
            break;
        case 2: hipLaunchKernelGGL(s2_gardner2_kernel, dim3((nstreams + G2_PAIRS * G_SPW - 1) / (G2_PAIRS * G_SPW)), dim3(64 * 2 * G2_PAIRS), 0, st, d_work, nstreams, coefs, d_bank, c, nsub); break;
        default: hipLaunchKernelGGL(s2_gardner_kernel, dim3((nstreams + G_SPW - 1) / G_SPW), dim3(64), 0, st, d_work, nstreams, coefs, d_bank, c, nsub); break;
    }
}
#define GARDNER_LAUNCH(c_, n_) gardner_launch(d_work, nstreams, coefs, d_bank, st, (c_), (n_))
hipError_t s2_frontend_launch(const S2StreamWork* d_work, int nstreams, S2LoopCoefs coefs, const float* d_bank, hipStream_t st, hipStream_t aux,
                              hipEvent_t* ev, int nsub, const S2PostStages* post, hipEvent_t* ev2, hipStream_t post_stream, hipStream_t loops_stream, hipEvent_t* ev3) {
    // loops_stream (+ ev3: 2 (nsub + 1) events): the frame loops of slice c on a stream of their own, beside the RRC of slice c + 1 -- a big bank's post stages are a pipeline
    // of their own (plugin's mode, r06 timeline: RRC 3 + PL-sync walk 2 + frame loops 10 ms per slice behind 10 ms of timing recovery: the post stages, not the timing
    // recovery, set the step).  The PL-sync walk of slice c + 1 still waits for the frame loops of slice c (they read its frame table); the RRC of slice c + 1 only appends
    // symbols behind what slice c's frames hold.
    // post_stream: a stream of their own for the post stages (synchronous mode: the FEC stream's hardware queue is free) -- on `aux` the
    // frame loops queue behind the AGC slices, and for a few streams that queue is the longest (AGC 4 x 7 + loops 4 x 10 ms against 47 ms of
    // timing recovery per 4-frame call)
    const dim3 ga((nstreams + 63) / 64), gg((nstreams + G_SPW - 1) / G_SPW);
    hipStream_t ps = post_stream ? post_stream : aux;
    if (nsub <= 1 || !aux || !ev || (post && !ev2)) {
        hipLaunchKernelGGL(agc_pc_kernel<AgcS2Traits>, ga, dim3(128), 0, st, d_work, nstreams, coefs, 0, 1);
        GARDNER_LAUNCH(0, 1);
        if (post) return post_stages_launch(d_work, nstreams, coefs, *post, 0, 1, st);
        return hipGetLastError();
    }
    hipError_t e;
    if ((e = hipEventRecord(ev[nsub], st)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(aux, ev[nsub], 0)) != hipSuccess) return e;
    int agc_next = 0;
    auto agc_upto = [&](int k) -> hipError_t {
        for (; agc_next <= k && agc_next < nsub; ++agc_next) {
            hipLaunchKernelGGL(agc_pc_kernel<AgcS2Traits>, ga, dim3(128), 0, aux, d_work, nstreams, coefs, agc_next, nsub);
            hipError_t e2 = hipEventRecord(ev[agc_next], aux);
            if (e2 != hipSuccess) return e2;
        }
        return hipSuccess;
    };
    for (int c = 0; c < nsub; ++c) {
        if ((e = agc_upto(post ? c + 1 : c)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st, ev[c], 0)) != hipSuccess) return e;
        GARDNER_LAUNCH(c, nsub);
        if (post) {
            if ((e = hipEventRecord(ev2[c], st)) != hipSuccess) return e;
            if ((e = agc_upto(c + 2)) != hipSuccess) return e;              // (the AGC stays ahead of the timing loop: its next slices go in before this slice's post stages)
            if ((e = hipStreamWaitEvent(ps, ev2[c], 0)) != hipSuccess) return e;
            if (loops_stream && ev3) {
                hipEvent_t* walked = ev3;                  // [c]: PL-sync walk of slice c done
                hipEvent_t* looped = ev3 + nsub + 1;       // [c]: frame loops of slice c done
                if ((e = post_stages_launch(d_work, nstreams, coefs, *post, c, nsub, ps, 1)) != hipSuccess) return e;
                if (c > 0 && (e = hipStreamWaitEvent(ps, looped[c - 1], 0)) != hipSuccess) return e;
                if ((e = post_stages_launch(d_work, nstreams, coefs, *post, c, nsub, ps, 2)) != hipSuccess) return e;
                if ((e = hipEventRecord(walked[c], ps)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(loops_stream, walked[c], 0)) != hipSuccess) return e;
                if ((e = post_stages_launch(d_work, nstreams, coefs, *post, c, nsub, loops_stream, 4)) != hipSuccess) return e;
                if ((e = hipEventRecord(looped[c], loops_stream)) != hipSuccess) return e;
            } else if ((e = post_stages_launch(d_work, nstreams, coefs, *post, c, nsub, ps)) != hipSuccess) return e;
        }
    }
    if (post) {
        if (loops_stream && ev3) {
            if ((e = hipStreamWaitEvent(ps, ev3[nsub + 1 + nsub - 1], 0)) != hipSuccess) return e;          // (the last slice's frame loops)
        }
        if ((e = hipEventRecord(ev2[nsub], ps)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st, ev2[nsub], 0)) != hipSuccess) return e;
    }
    return hipGetLastError();
}
hipError_t s2_rrc_decim_launch(const S2StreamWork* d_work, int nstreams, int max_count, const float* d_taps, int ntaps, hipStream_t st, int post_prio) {
    int gx = (max_count / 2 + 2 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(s2_rrc_decim_kernel, dim3(gx, nstreams), dim3(256), 0, st, d_work, d_taps, ntaps, 0, 1 | ((post_prio ? 1 : 0) << 16));
    hipLaunchKernelGGL(s2_rrc_state_kernel, dim3(nstreams), dim3(128), 0, st, d_work, ntaps);
    return hipGetLastError();
}
hipError_t s2_frame_loops_launch(const S2StreamWork* d_work, int nstreams, const S2FrameRef* d_frames, const int* d_first,
                                 S2LoopCoefs coefs, S2PlTablesDev tabs, S2ConstelDev con, int pls_code, int slots, int pilots,
                                 int pilot_blocks, int plframe, cf32* d_pllout, S2FrameStats* d_stats, hipStream_t st) {
    const int spw = frame_loops_spw(nstreams);
    hipLaunchKernelGGL(s2_frame_loops_kernel<false>, dim3(frame_loops_grid(nstreams, spw)), dim3(64 * frame_loops_wpb(nstreams, spw)), 0, st, d_work, nstreams, d_frames, d_first, coefs,
                       tabs, con, pls_code, slots, pilots, pilot_blocks, plframe, d_pllout, d_stats, (const S2VcmFound*)nullptr, 0, spw, (const S2StreamCfgDev*)nullptr);
    return hipGetLastError();
}
hipError_t s2_ccm_walk_launch(const S2StreamWork* d_work, int nstreams, int raw, int maxf, S2VcmFound* d_found, int* d_counts, hipStream_t st, int post_prio) {
    hipLaunchKernelGGL(s2_ccm_walk_kernel, dim3(nstreams), dim3(256), 0, st, d_work, raw, maxf, d_found, d_counts, 0, 1 | ((post_prio ? 1 : 0) << 16), (const S2StreamCfgDev*)nullptr);
    return hipGetLastError();
}
hipError_t s2_vcm_walk_launch(const S2StreamWork* d_work, int nstreams, S2PlTablesDev tabs, const S2VcmMod* d_mods, float sof_threshold, int maxf,
                              S2VcmFound* d_found, int* d_counts, hipStream_t st) {
    hipLaunchKernelGGL(s2_vcm_walk_kernel, dim3(nstreams), dim3(256), 0, st, d_work, tabs, d_mods, sof_threshold, maxf, d_found, d_counts);
    return hipGetLastError();
}
hipError_t s2_vcm_loops_launch(const S2StreamWork* d_work, int nstreams, const S2VcmFrame* d_frames, const int* d_first, S2LoopCoefs coefs,
                               S2PlTablesDev tabs, const S2VcmMod* d_mods, const S2ConstelDev* d_cons, cf32* d_pllout, S2FrameStats* d_stats,
                               hipStream_t st) {
    hipLaunchKernelGGL(s2_vcm_loops_kernel, dim3(nstreams), dim3(64), 0, st, d_work, d_frames, d_first, coefs, tabs, d_mods, d_cons, d_pllout, d_stats);
    return hipGetLastError();
}
hipError_t s2_vcm_demap_launch(const S2VcmFrame* d_frames, int nframes, const S2VcmMod* d_mods, const S2ConstelDev* d_cons, const cf32* d_pllout,
                               int8_t* d_llr, hipStream_t st) {
    hipLaunchKernelGGL(s2_vcm_demap_kernel, dim3((360 * 90 + 255) / 256, nframes), dim3(256), 0, st, d_frames, d_mods, d_cons, d_pllout, d_llr);
    return hipGetLastError();
}
hipError_t s2_vcm_gather_launch(const S2VcmFrame* d_frames, const int* d_idx, int count, int N, const int8_t* d_llr, int8_t* d_grp, hipStream_t st) {
    hipLaunchKernelGGL(s2_vcm_gather_kernel, dim3(16, count), dim3(256), 0, st, d_frames, d_idx, N, d_llr, d_grp);
    return hipGetLastError();
}
hipError_t s2_vcm_scatter_launch(const int* d_idx, int count, int kb, const uint8_t* d_bb, uint8_t* const* d_dst, hipStream_t st) {
    hipLaunchKernelGGL(s2_vcm_scatter_kernel, dim3(count), dim3(256), 0, st, d_idx, kb, d_bb, d_dst);
    return hipGetLastError();
}
hipError_t s2_deinterleave_launch(int constel, int rate, int bits, int N, const int8_t* d_in, int nframes, int8_t* d_out, hipStream_t st) {
    hipLaunchKernelGGL(s2_deinterleave_kernel, dim3((N + 255) / 256, nframes), dim3(256), 0, st, constel, rate, bits, N, d_in, d_out);
    return hipGetLastError();
}
hipError_t s2_demap_launch(S2ConstelDev con, int rate, int shortframe, int slots, int pilots, int plframe, const cf32* d_pllout,
                           int nframes, int8_t* d_llr, int N, hipStream_t st, const int* d_slot, int post_prio) {
    (void)shortframe;
    int gx = (slots * 90 + 255) / 256;
    // the wide form (four symbols per lane) wants 16-byte aligned symbols and word-aligned LLR columns: buffers of this library are, a caller's need
    // not be; 16APSK short frames have columns of 4050 bytes
    const bool wide = con.lut_bits4 && !((uintptr_t)d_pllout & 15u) && !((uintptr_t)d_llr & 7u) && !(N & 7) && !(plframe & 1) &&
                      (con.bits == 2 || ((N / con.bits) & 3) == 0);
    if (!wide) con.lut_bits4 = nullptr;
    else gx = (slots * 90 / 4 + 255) / 256;
    hipLaunchKernelGGL(s2_demap_kernel<false>, dim3(gx, nframes), dim3(256), 0, st, con, rate, slots, pilots, plframe, d_pllout, d_llr, N, d_slot,
                       (const S2StreamCfgDev*)nullptr, post_prio ? 1 : 0, (int8_t* const*)nullptr);
    return hipGetLastError();
}
hipError_t s2_demap_mixed_launch(const S2StreamCfgDev* cfgs, int max_slots, int maxf, int slot_stride, const cf32* d_pllout, int nframes,
                                 int8_t* const* d_llr_of, hipStream_t st, const int* d_slot) {
    // (every buffer here is the library's own: aligned for the wide form; the kernel falls back to the byte form per configuration where a column is not a whole number of words)
    const int gx = (max_slots * 90 / 4 + 255) / 256;
    hipLaunchKernelGGL(s2_demap_kernel<true>, dim3(gx, nframes), dim3(256), 0, st, S2ConstelDev{}, 0, 0, 0, slot_stride, d_pllout, (int8_t*)nullptr, 0, d_slot,
                       cfgs, maxf, d_llr_of);
    return hipGetLastError();
}

}  // namespace s2
