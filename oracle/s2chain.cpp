// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.  See s2chain.h for the
// pinning status of each stage ("parity unpinned" for the SDR++/VOLK based float stages).
//
// Restates, stage by stage:
//   s2_sof / s2_plscodes                         dvbs2/s2_defs.h:15-80
//   S2Scrambling (Gold sequence Rn, (de)scramble) dvbs2/codings/s2_scrambling.cpp:9-81, .h:17-25
//   constellation_t (points, soft calc, LUT)      common/dsp/demod/constellation.cpp:19-322
//   S2Deinterleaver::deinterleave                 dvbs2/codings/s2_deinterleaver.cpp:6-136
//   FastAGC (SDR++), FreqShift                    module_dvbs2_demod.cpp:220, common/dsp/demod/freq_shift.cpp:4-17
//   clock_recovery::Gardner                       common/dsp/demod/gardner.cpp:22,89-159
//   FIR + rootRaisedCosine (SDR++), /2 decimator  module_dvbs2_demod.cpp:42,226,231-239
//   S2PLSyncBlock                                 dvbs2/dvbs2_pl_sync.cpp:81-193
//   dvbs2_pilot_coarse_fed + NCO feedback         dvbs2/dvbs2_fed.h:7-48, module_dvbs2_demod.cpp:319-331
//   S2PLLBlock::process                           dvbs2/dvbs2_pll.cpp:10-11,34-86
//   S2PLHDRDemod::process                         dvbs2/dvbs2_plhdr_demod.cpp:9-10,33-79
//   S2BBToSoft::process                           dvbs2/dvbs2_bb_to_soft.cpp:7-33
//   DVBS2Demod::process                           dvbs2/module_dvbs2_demod.cpp:216-372
// Deliberate departures: every frame is LDPC-decoded (Q1); pilots follow the standard
// (one 36-symbol block after every 16 slots: Q3/Q4); BBFRAMEs are emitted in the call that completes them.
#include "s2chain.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

namespace orc {

void critically_damped(float bw, float* alpha, float* beta) {
    double damping = 0.70710678118654752440;
    double den = 1.0 + 2.0 * damping * bw + (double)bw * bw;
    *alpha = (float)((4.0 * damping * bw) / den);
    *beta = (float)((4.0 * (double)bw * bw) / den);
}

// ------------------------------------------------------------------------------------------ tables
PlTables::PlTables() {
    const uint32_t VALUE = 0x18d2e82;
    for (int s = 0; s < 26; ++s) {
        int bit = (VALUE >> (25 - s)) & 1;
        int angle = bit * 2 + (s & 1);
        sof[s] = phasor((float)(M_PI / 4 + 2 * M_PI * angle / 4));
    }
    const uint32_t G[6] = {0x55555555, 0x33333333, 0x0f0f0f0f, 0x00ff00ff, 0x0000ffff, 0xffffffff};
    const uint64_t SCR = 0x719d83c953422dfaull;
    for (int index = 0; index < 128; ++index) {
        uint32_t y = 0;
        for (int row = 0; row < 6; ++row)
            if ((index >> (6 - row)) & 1) y ^= G[row];
        uint64_t code = 0;
        for (int bit = 31; bit >= 0; --bit) {
            int yi = (y >> bit) & 1;
            if (index & 1) code = (code << 2) | ((uint64_t)yi << 1) | (uint64_t)(yi ^ 1);
            else code = (code << 2) | ((uint64_t)yi << 1) | (uint64_t)yi;
        }
        code ^= SCR;
        plsc_code[index] = code;
        for (int i = 0; i < 64; ++i) {
            int yi = (code >> (63 - i)) & 1;
            int nyi = yi ^ (i & 1);
            plsc_sym[index][i].re = (1 - 2 * nyi) / sqrtf(2);
            plsc_sym[index][i].im = (1 - 2 * yi) / sqrtf(2);
        }
    }
    Rn.assign(131072, 0);
    auto lfsr_x = [](uint32_t X) { int bit = ((X >> 7) ^ X) & 1; return ((uint32_t)(bit << 18) | X) >> 1; };
    auto lfsr_y = [](uint32_t Y) { int bit = ((Y >> 10) ^ (Y >> 7) ^ (Y >> 5) ^ Y) & 1; return ((uint32_t)(bit << 18) | Y) >> 1; };
    uint32_t stx = 0x00001, sty = 0x3ffff;
    for (int i = 0; i < 131072; ++i) { Rn[i] = (uint8_t)((stx ^ sty) & 1); stx = lfsr_x(stx); sty = lfsr_y(sty); }
    for (int i = 0; i < 131072; ++i) { Rn[i] |= (uint8_t)(((stx ^ sty) & 1) << 1); stx = lfsr_x(stx); sty = lfsr_y(sty); }
}
const PlTables& pl_tables() { static PlTables t; return t; }

static inline cf pl_descramble(cf p, int r) {   // s2_scrambling.cpp:37-58
    switch (r) {
        case 3: return cf{-p.im, p.re};
        case 2: return cf{-p.re, -p.im};
        case 1: return cf{p.im, -p.re};
        default: return p;
    }
}
static inline cf pl_scramble(cf p, int r) {     // s2_scrambling.cpp:60-81
    switch (r) {
        case 3: return cf{p.im, -p.re};
        case 2: return cf{-p.re, -p.im};
        case 1: return cf{-p.im, p.re};
        default: return p;
    }
}

// ------------------------------------------------------------------------------------------ constellation
static cf polar(float r, int n, float i) {
    float a = i * 2 * M_PI / n;
    cf u = phasor(a);
    return cf{r * u.re, r * u.im};
}

Constellation::Constellation(int type_, float g1, float g2) : type(type_) {
    amp = 1.0f; sca = 50.0f; prescale = 1.0f;
    const double SQ2 = 1.41421356237309504880;
    if (type == s2::C_QPSK) {
        states = 4; bits = 2; amp = 3;
        pts = {cf{(float)-SQ2, (float)-SQ2}, cf{(float)SQ2, (float)-SQ2}, cf{(float)-SQ2, (float)SQ2}, cf{(float)SQ2, (float)SQ2}};
    } else if (type == s2::C_8PSK) {
        states = 8; bits = 3;
        float r = 0.70710678118654752440;
        pts = {cf{0.0f, -1.0f}, cf{-r, r}, cf{r, -r}, cf{0.0f, 1.0f}, cf{-r, -r}, cf{-1.0f, 0.0f}, cf{1.0f, 0.0f}, cf{r, r}};
    } else if (type == s2::C_16APSK) {
        states = 16; bits = 4; amp = 100; sca = 1; prescale = 0.53;
        float gamma1 = g1;
        if (!gamma1) gamma1 = 2.57;
        float r1 = sqrtf(4 / (1 + 3 * gamma1 * gamma1));
        float r2 = gamma1 * r1;
        r1 *= 0.5; r2 *= 0.5;
        pts.assign(16, cf{0, 0});
        const float o12[12] = {8.5f, 3.5f, 9.5f, 2.5f, 6.5f, 5.5f, 11.5f, 0.5f, 7.5f, 4.5f, 10.5f, 1.5f};  // points 4..15
        const float o4[4] = {2.5f, 1.5f, 3.5f, 0.5f};                                                    // points 0..3
        for (int i = 0; i < 4; ++i) pts[i] = cscale(polar(r1, 4, o4[i]), amp);
        for (int i = 0; i < 12; ++i) pts[4 + i] = cscale(polar(r2, 12, o12[i]), amp);
    } else {
        states = 32; bits = 5; amp = 100; sca = 1; prescale = 0.54;
        float gamma1 = g1, gamma2 = g2;
        if (!gamma1) gamma1 = 2.53;
        if (!gamma2) gamma2 = 4.30;
        float r1 = sqrtf(8 / (1 + 3 * gamma1 * gamma1 + 4 * gamma2 * gamma2));
        float r2 = gamma1 * r1, r3 = gamma2 * r1;
        r1 *= 0.5; r2 *= 0.5; r3 *= 0.5;
        pts.assign(32, cf{0, 0});
        // (ring, points-on-ring, position) for constellation index 0..31 (constellation.cpp:117-148)
        struct P { int ring, n; float i; };
        const P tab[32] = {{3, 16, 10}, {3, 16, 8}, {3, 16, 5}, {3, 16, 7}, {3, 16, 13}, {3, 16, 15}, {3, 16, 2}, {3, 16, 0},
                           {1, 4, 2.5f}, {2, 12, 6.5f}, {1, 4, 1.5f}, {2, 12, 5.5f}, {1, 4, 3.5f}, {2, 12, 11.5f}, {1, 4, 0.5f}, {2, 12, 0.5f},
                           {3, 16, 11}, {3, 16, 9}, {3, 16, 4}, {3, 16, 6}, {3, 16, 12}, {3, 16, 14}, {3, 16, 3}, {3, 16, 1},
                           {2, 12, 8.5f}, {2, 12, 7.5f}, {2, 12, 3.5f}, {2, 12, 4.5f}, {2, 12, 9.5f}, {2, 12, 10.5f}, {2, 12, 2.5f}, {2, 12, 1.5f}};
        for (int i = 0; i < 32; ++i) {
            float r = tab[i].ring == 1 ? r1 : (tab[i].ring == 2 ? r2 : r3);
            pts[i] = cscale(polar(r, tab[i].n, tab[i].i), amp);
        }
    }
    if (bits != 5) {   // make_lut(256), constellation.cpp:272-291
        lut_bits.assign((size_t)256 * 256 * bits, 0);
        lut_err.assign((size_t)256 * 256, 0.f);
        for (int x = 0; x < 256; ++x)
            for (int y = 0; y < 256; ++y) {
                float xv = (float(x - 128) / float(256)) * 1.5f;
                float yv = (float(y - 128) / float(256)) * 1.5f;
                soft_calc(cf{xv, yv}, &lut_bits[((size_t)x * 256 + y) * bits], &lut_err[(size_t)x * 256 + y]);
            }
    }
}

cf Constellation::mod(int sym) const { return cscale(cscale(pts[sym], 1.0f / amp), 1.0f / prescale); }

static int8_t lut_clamp(float x) { return dvbs2m::llr_clamp_det(x); }   // constellation.cpp:263-270 (non-finite -> 0, see the header)

void Constellation::soft_calc(cf sample, int8_t* bits_out, float* phase_err) const {
    float tmp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (amp != 1) sample = cscale(sample, amp);
    if (prescale != 1) sample = cscale(sample, prescale);
    float min_dist = std::numeric_limits<float>::max();
    cf closest{0, 0};
    for (int i = 0; i < states; i++) {
        float dist = camp(csub(sample, pts[i]));
        if (dist < min_dist) { min_dist = dist; closest = pts[i]; }
        float d = dvbs2m::expf_det(-dist / 1.0f);
        for (int j = 0; j < bits; j++) {
            if (((i >> j) & 1) == 0) tmp[2 * j + 0] += d;
            else tmp[2 * j + 1] += d;
        }
    }
    if (bits_out)
        for (int i = 0; i < bits; i++) bits_out[bits - 1 - i] = lut_clamp((dvbs2m::logf_det(tmp[2 * i + 1]) - dvbs2m::logf_det(tmp[2 * i + 0])) * sca);
    if (phase_err) *phase_err = cphase(cmul(sample, cconj(closest)));
}

int Constellation::lut_index(float v) {   // constellation.cpp:295-301: double math, truncation, clamp
    int x = (int)((v / 1.5) * 256 + 128);
    if (x < 0) x = 0;
    if (x >= 256) x = 255;
    return x;
}

void Constellation::soft_lut(cf s, int8_t* bits_out, float* phase_err) const {
    if (bits != 5) {
        int x = lut_index(s.re), y = lut_index(s.im);
        if (bits_out)
            for (int i = 0; i < bits; ++i) bits_out[i] = lut_bits[((size_t)x * 256 + y) * bits + i];
        if (phase_err) *phase_err = lut_err[(size_t)x * 256 + y];
    } else {
        soft_calc(s, bits_out, phase_err);
    }
}

void s2_deinterleave(int constel, int rate, int shortframe, const int8_t* in, int8_t* out) {
    int N = shortframe ? 16200 : 64800;
    if (constel == s2::C_QPSK) {
        for (int i = 0; i < N / 2; i++) { out[i * 2 + 1] = in[i * 2 + 0]; out[i * 2 + 0] = in[i * 2 + 1]; }
        return;
    }
    int bits = constel == s2::C_8PSK ? 3 : (constel == s2::C_16APSK ? 4 : 5);
    int rows = N / bits;
    int col[5];
    for (int c = 0; c < bits; ++c) col[c] = rows * c;
    if (constel == s2::C_8PSK && rate == s2::R3_5) { col[0] = rows * 2; col[1] = rows; col[2] = 0; }
    int i = 0;
    for (int j = 0; j < rows; ++j)
        for (int c = 0; c < bits; ++c) out[col[c] + j] = in[i++];
}

// ------------------------------------------------------------------------------------------ taps
std::vector<float> rrc_taps(int count, double beta, double symbolrate, double samplerate) {
    double Ts = samplerate / symbolrate;
    const double PI = 3.14159265358979323846, SQ2 = 1.41421356237309504880;
    double limit = Ts / (4.0 * beta);
    std::vector<float> taps(count);
    double half = (double)count / 2.0;
    for (int i = 0; i < count; i++) {
        double t = (double)i - half + 0.5;
        double v;
        if (t == 0.0) v = (1.0 + beta * (4.0 / PI - 1.0)) / Ts;
        else if (t == limit || t == -limit)
            v = ((1.0 + 2.0 / PI) * sin(PI / (4.0 * beta)) + (1.0 - 2.0 / PI) * cos(PI / (4.0 * beta))) * beta / (Ts * SQ2);
        else
            v = ((sin((1.0 - beta) * PI * t / Ts) + cos((1.0 + beta) * PI * t / Ts) * 4.0 * beta * t / Ts) /
                 ((1.0 - (4.0 * beta * t / Ts) * (4.0 * beta * t / Ts)) * PI * t / Ts)) / Ts;
        taps[i] = (float)v;
    }
    return taps;
}

std::vector<float> gardner_bank(int phases, int tpp) {   // gardner.cpp:154-159 + SDR++ windowedSinc/nuttall/polyphase
    const double PI = 3.14159265358979323846;
    int count = phases * tpp;
    double bw = 0.5 / (double)phases;
    double omega = 2.0 * PI * bw;
    double half = (double)count / 2.0;
    double corr = (double)phases * omega / PI;
    const double coefs[4] = {0.355768, 0.487396, 0.144232, 0.012604};
    std::vector<float> lp(count);
    for (int i = 0; i < count; ++i) {
        double t = (double)i - half + 0.5;
        double x = t * omega;
        double sinc = (x == 0.0) ? 1.0 : sin(x) / x;
        double n = t - half, win = 0.0, sign = 1.0;
        for (int c = 0; c < 4; ++c) { win += sign * coefs[c] * cos((double)c * 2.0 * PI * n / (double)count); sign = -sign; }
        lp[i] = (float)(sinc * win * corr);
    }
    std::vector<float> bank((size_t)phases * tpp, 0.f);
    for (int i = 0; i < count; ++i) bank[(size_t)((phases - 1) - (i % phases)) * tpp + i / phases] = lp[i];
    return bank;
}

DemodCfg default_cfg(int modcod, int shortframes, int pilots) {   // main.cpp:64-73,134-140
    DemodCfg c;
    c.symbolrate = 27.5e6; c.samplerate = 55e6;
    c.agc_rate = 0.0001f; c.rrc_alpha = 0.35f; c.rrc_taps = 65; c.loop_bw = 0.00628f; c.fll_bw = 0.006f;
    float bw = 0.00628f, damp = 0.707f;
    float den = (1.0f + 2.0 * damp * bw + bw * bw);
    c.clock_mu_gain = (4.0f * damp * bw) / den;
    c.clock_omega_gain = (4.0f * bw * bw) / den;
    c.omega_rel_limit = 0.02f;
    c.modcod = modcod; c.shortframes = shortframes; c.pilots = pilots;
    c.sof_threshold = 0.6f; c.max_ldpc_trials = 16; c.force_ldpc_iters = 0;
    c.acm_vcm = 0; c.soft_plsc = 0; c.pilot_aided = 0;
    return c;
}

// ------------------------------------------------------------------------------------------ receiver
static std::mutex g_cmtx;
static const LdpcCode* get_ldpc(int ci) {
    static std::map<int, std::unique_ptr<LdpcCode>> cache;
    std::lock_guard<std::mutex> l(g_cmtx);
    auto& p = cache[ci];
    if (!p) p.reset(new LdpcCode(ci));
    return p.get();
}
static const BchCode* get_bch(const s2::FecParams& f) {
    static std::map<int, std::unique_ptr<BchCode>> cache;
    std::lock_guard<std::mutex> l(g_cmtx);
    auto& p = cache[f.code_index];
    if (!p) p.reset(new BchCode(f.bch_m, f.bch_t, f.K, f.kbch));
    return p.get();
}

static s2::ModcodParams mp_of(const DemodCfg& c) {
    s2::ModcodParams p;
    if (!s2::modcod_params(c.modcod, c.shortframes, c.pilots, &p)) throw std::runtime_error("bad MODCOD");
    return p;
}

S2Rx::S2Rx(const DemodCfg& c) : mp(mp_of(c)), cfg(c), constel(mp.constel, mp.g1, mp.g2) {
    rrc = rrc_taps(cfg.rrc_taps, cfg.rrc_alpha, cfg.symbolrate, cfg.samplerate);
    bank = gardner_bank(128, 8);
    pls_code = cfg.modcod << 2 | (cfg.shortframes ? 2 : 0) | (cfg.pilots ? 1 : 0);
    ldpc = get_ldpc(mp.fec.code_index);
    bch = get_bch(mp.fec);
    ccm = FrameCtx{mp, pls_code, &constel, ldpc, bch, false};
    vcm_ctx.resize(128);
    vcm_constel.resize(128);
    in_buffer.assign(mp.plframe, cf{0, 0});
    corr_buffer.assign(mp.plframe, cf{0, 0});
    reset();
}

void S2Rx::reset() {
    agc_gain = 1.0f;
    nco_phase = 0; nco_freq_ = 0;
    rrc_hist.assign(cfg.rrc_taps - 1, cf{0, 0});
    g_hist.assign(7, cf{0, 0});
    g_offset = 0; g_spsctr = 0;
    g_pcl.init(cfg.clock_mu_gain, cfg.clock_omega_gain, 0.0f, 0.0f, 1.0f, 1.0f, (float)(1.0 * (1.0 - cfg.omega_rel_limit)),
               (float)(1.0 * (1.0 + cfg.omega_rel_limit)), false);
    cr_samp = false;
    in_ptr = 0; in_lim = mp.plframe; in_state = 0; best_pos = 0; last_bm = 0;
    vfifo.clear(); vcm_synced = false;
    float a, b;
    critically_damped(cfg.loop_bw, &a, &b);
    pll_pcl.init(a, b, 0, -(float)M_PI, (float)M_PI, 0, -0.01f * (float)M_PI, 0.01f * (float)M_PI, true);
    critically_damped(cfg.loop_bw * 0.03f, &a, &b);
    hdr_pcl.init(a, b, 0, -(float)M_PI, (float)M_PI, 0, -1.0f * (float)M_PI, 1.0f * (float)M_PI, true);
}

// ---- round 6: the fixed-point question for FastAGC (test infrastructure, like pll_tile_study; nothing of the product runs this).  Per tile: the gain BEFORE every sample is guessed
// (first: the tile's starting gain everywhere), a[k] = |x[k] g[k]| is evaluated for all k at once, the recurrence g[k + 1] = min(g[k] + (1 - a[k]) rate, 10e6) is replayed in the
// reference's order with those a[k], and that is repeated until no gain changes: g[0] is exact, an exact g[k] gives the exact a[k] and so the exact g[k + 1] -- the fixed point is the
// serial loop's sequence.  Histogram: evaluation passes per tile (the last one only confirms).
void S2Rx::agc_tile_study(int n, const cf* in) {
    const int T = agc_study_tile;
    float g = agc_gain;
    std::vector<float> G(T + 1), Gn(T + 1), A(T);
    for (int base = 0; base < n; base += T) {
        const int m = std::min(T, n - base);
        std::vector<float> truth(m + 1);
        truth[0] = g;
        for (int k = 0; k < m; ++k) {
            float a = camp(cscale(in[base + k], truth[k]));
            float gn = truth[k] + (1.0f - a) * cfg.agc_rate;
            truth[k + 1] = gn > 10e6f ? 10e6f : gn;
        }
        for (int k = 0; k <= m; ++k) G[k] = g;
        int passes = 0;
        for (;;) {
            for (int k = 0; k < m; ++k) A[k] = camp(cscale(in[base + k], G[k]));
            Gn[0] = g;
            for (int k = 0; k < m; ++k) { float gn = Gn[k] + (1.0f - A[k]) * cfg.agc_rate; Gn[k + 1] = gn > 10e6f ? 10e6f : gn; }
            ++passes;
            bool same = true;
            for (int k = 0; k <= m; ++k) same = same && memcmp(&Gn[k], &G[k], 4) == 0;
            if (same || passes > T + 2) break;
            for (int k = 0; k <= m; ++k) G[k] = Gn[k];
        }
        agc_hist[passes < 65 ? passes : 65]++;
        for (int k = 0; k <= m; ++k) if (memcmp(&Gn[k], &truth[k], 4) != 0) { agc_mismatch++; break; }
        agc_samples += m;
        g = truth[m];
    }
}

void S2Rx::agc(int n, const cf* in, cf* out) {   // SDR++ loop::FastAGC<complex_t>: set point 1, max gain 10e6
    if (agc_study_tile > 0) agc_tile_study(n, in);
    for (int i = 0; i < n; ++i) {
        out[i] = cscale(in[i], agc_gain);
        float a = camp(out[i]);
        agc_gain += (1.0f - a) * cfg.agc_rate;
        if (agc_gain > 10e6f) agc_gain = 10e6f;
    }
}

void S2Rx::nco(int n, const cf* in, cf* out) {   // freq_shift.cpp:4-17 (phase kept in float, wrap at +-2pi in double compare)
    for (int i = 0; i < n; ++i) {
        cf x = cmul(in[i], phasor(-nco_phase));
        nco_phase += nco_freq_;
        while (nco_phase > (2 * M_PI)) nco_phase -= 2 * M_PI;
        while (nco_phase < (-2 * M_PI)) nco_phase += 2 * M_PI;
        out[i] = x;
    }
}

static inline cf dot8(const cf* x, const float* t) {
    cf acc{0, 0};
    for (int k = 0; k < 8; ++k) { acc.re += x[k].re * t[k]; acc.im += x[k].im * t[k]; }
    return acc;
}

// ---- round 6: the fixed-point question for the timing recovery (test infrastructure).  What an on-symbol output contributes to the loop -- the error e -- depends on WHERE it is taken
// only: the sample offset and the polyphase arm floor(128 phase), two integers.  Per tile of S symbols: the (offset, arm) of every on-symbol output is guessed (first: the loop run
// forward from the tile's start with e = 0), the errors are evaluated for all symbols at once (three interpolants each), the loop is replayed in the reference's order with those
// errors up to the first symbol whose (offset, arm) is not the guessed one; from there the guesses are renewed (the loop run on with e = 0) and the errors evaluated again.  A symbol
// whose place is right has the right error, so the replay's state behind it is the serial loop's: the result is the serial sequence.  Counted: evaluation passes per tile, replay
// steps per symbol.
void S2Rx::gardner_tile_study(int n, const cf* in) {
    const int S = gardner_study_tile;
    std::vector<cf> buf(n + 8);
    for (int i = 0; i < 7; ++i) buf[i] = g_hist[i];
    memcpy(&buf[7], in, sizeof(cf) * n);
    Pcl pcl = g_pcl;
    int offset = g_offset, sps = g_spsctr;
    auto arm_of = [](const Pcl& p) { int a = (int)floorf(p.phase * 128.0f); return a < 0 ? 0 : (a > 127 ? 127 : a); };
    auto step = [&](Pcl& p, int& off, float error) {
        if (error > 1.0f) error = 1.0f;
        if (error < -1.0f) error = -1.0f;
        p.advance(error);
        float delta = floorf(p.phase);
        off = (int)((float)off + delta);
        p.phase -= delta;
    };
    auto error_at = [&](int off, int phase) -> float {
        cf outVal = dot8(&buf[off], &bank[(size_t)phase * 8]);
        cf dfdt;
        if (phase == 0) dfdt = csub(dot8(&buf[off], &bank[(size_t)(phase + 1) * 8]), outVal);
        else if (phase == 127) dfdt = csub(outVal, dot8(&buf[off], &bank[(size_t)(phase - 1) * 8]));
        else dfdt = cscale(csub(dot8(&buf[off], &bank[(size_t)(phase + 1) * 8]), dot8(&buf[off], &bank[(size_t)(phase - 1) * 8])), 0.5f);
        return -(((outVal.re > 0 ? 1.0f : -1.0f) * dfdt.re) + ((outVal.im > 0 ? 1.0f : -1.0f) * dfdt.im));
    };
    // a slice that starts between the two outputs of a symbol: the follower first (as the engine's kernels do)
    if (sps == 1 && offset < n) { step(pcl, offset, 0.0f); sps = 0; }
    while (offset < n - 4 * S - 8) {          // whole tiles only (the ends of a slice go through single steps in the engine as well)
        // serial truth
        Pcl tp = pcl; int toff = offset;
        std::vector<int> t_off(S), t_arm(S);
        for (int k = 0; k < S; ++k) {
            t_off[k] = toff; t_arm[k] = arm_of(tp);
            step(tp, toff, error_at(toff, t_arm[k]));
            step(tp, toff, 0.0f);
        }
        // tile
        std::vector<int> g_off(S), g_arm(S);
        std::vector<float> e(S);
        Pcl rp = pcl; int roff = offset;      // the replay's state: exact up to symbol `done`
        int done = 0, evals = 0;
        while (done < S) {
            // guesses from `done` on: the loop run on with e = 0
            Pcl gp = rp; int goff = roff;
            for (int k = done; k < S; ++k) { g_off[k] = goff; g_arm[k] = arm_of(gp); step(gp, goff, 0.0f); step(gp, goff, 0.0f); }
            for (int k = done; k < S; ++k) e[k] = error_at(g_off[k], g_arm[k]);          // (in parallel on the device)
            ++evals;
            // replay up to the first symbol that is somewhere else
            for (; done < S; ++done) {
                if (roff != g_off[done] || arm_of(rp) != g_arm[done]) break;
                step(rp, roff, e[done]);
                step(rp, roff, 0.0f);
                gd_replay_steps++;
            }
            if (evals > S + 2) break;
        }
        gd_hist[evals < 33 ? evals : 33]++;
        gd_evals += evals; gd_tiles++; gd_syms += S;
        if (roff != toff || memcmp(&rp.phase, &tp.phase, 4) != 0 || memcmp(&rp.freq, &tp.freq, 4) != 0) gd_mismatch++;
        pcl = tp; offset = toff;
    }
}

int S2Rx::gardner(int n, const cf* in, cf* out) {   // gardner.cpp:89-152, omega 1, outSps 2
    if (gardner_study_tile > 0) gardner_tile_study(n, in);
    std::vector<cf> buf(n + 8);
    for (int i = 0; i < 7; ++i) buf[i] = g_hist[i];
    memcpy(&buf[7], in, sizeof(cf) * n);
    int outCount = 0;
    while (g_offset < n) {
        int phase = (int)floorf(g_pcl.phase * 128.0f);
        phase = phase < 0 ? 0 : (phase > 127 ? 127 : phase);
        cf outVal = dot8(&buf[g_offset], &bank[(size_t)phase * 8]);
        out[outCount++] = outVal;
        float error;
        if (g_spsctr == 0) {
            cf dfdt;
            if (phase == 0) {
                cf fT1 = dot8(&buf[g_offset], &bank[(size_t)(phase + 1) * 8]);
                dfdt = csub(fT1, outVal);
            } else if (phase == 127) {
                cf fT_1 = dot8(&buf[g_offset], &bank[(size_t)(phase - 1) * 8]);
                dfdt = csub(outVal, fT_1);
            } else {
                cf fT1 = dot8(&buf[g_offset], &bank[(size_t)(phase + 1) * 8]);
                cf fT_1 = dot8(&buf[g_offset], &bank[(size_t)(phase - 1) * 8]);
                dfdt = cscale(csub(fT1, fT_1), 0.5f);
            }
            error = -(((outVal.re > 0 ? 1.0f : -1.0f) * dfdt.re) + ((outVal.im > 0 ? 1.0f : -1.0f) * dfdt.im));
        } else {
            error = 0;
        }
        g_spsctr++;
        if (g_spsctr >= 2) g_spsctr = 0;
        if (error > 1.0f) error = 1.0f;
        if (error < -1.0f) error = -1.0f;
        g_pcl.advance(error);
        float delta = floorf(g_pcl.phase);
        g_offset = (int)((float)g_offset + delta);
        g_pcl.phase -= delta;
    }
    g_offset -= n;
    for (int i = 0; i < 7; ++i) g_hist[i] = buf[n + i];
    return outCount;
}

void S2Rx::rrc_filter(int n, const cf* in, cf* out) {   // SDR++ filter::FIR: out[i] = sum_k buf[i+k]*taps[k]
    int T = cfg.rrc_taps;
    std::vector<cf> buf(n + T - 1);
    for (int i = 0; i < T - 1; ++i) buf[i] = rrc_hist[i];
    if (n) memcpy(&buf[T - 1], in, sizeof(cf) * n);
    for (int i = 0; i < n; ++i) {
        cf acc{0, 0};
        for (int k = 0; k < T; ++k) { acc.re += buf[i + k].re * rrc[k]; acc.im += buf[i + k].im * rrc[k]; }
        out[i] = acc;
    }
    for (int i = 0; i < T - 1; ++i) rrc_hist[i] = buf[n + i];
}

// S2PLSyncBlock::internal_process: returns 1 when `out` received an aligned frame, else 0
int S2Rx::plsync_internal(std::vector<cf>& out, float* best_match_out) {
    const PlTables& T = pl_tables();
    const int raw = mp.plframe;
    if (in_state == 0) {
        corr_buffer = in_buffer;
        best_pos = 0;
        double best_match = 0;
        const uint32_t dsof = 0x18d2e82u ^ (0x18d2e82u >> 1);
        const uint64_t SCR = 0x719d83c953422dfaull;
        const uint64_t dscr = SCR ^ (SCR >> 1);
        (void)T;
        cf d90[90];
        for (int ss = 0; ss < raw - 90; ss++) {
            d90[0] = cf{0, 0};
            for (int k = 1; k < 90; ++k) d90[k] = cmul(cconj(corr_buffer[ss + k - 1]), corr_buffer[ss + k]);
            cf csof{0, 0};
            for (int i = 0; i < 26; ++i) {
                if (((dsof >> (25 - i)) ^ i) & 1) csof = cadd(csof, d90[i]);
                else csof = csub(csof, d90[i]);
            }
            cf cpl{0, 0};
            for (int i = 1; i < 64; i += 2) {
                if ((dscr >> (63 - i)) & 1) cpl = csub(cpl, d90[26 + i]);
                else cpl = cadd(cpl, d90[26 + i]);
            }
            cf c0 = cadd(csof, cpl), c1 = csub(csof, cpl);
            cf c = camp(c0) > camp(c1) ? c0 : c1;
            cf d = cscale(c, 1.0f / (26 - 1 + 64 / 2));
            double difference = camp(d);
            if (difference > best_match && d.im > 0) { best_match = difference; best_pos = ss; }
        }
        *best_match_out = (float)best_match;
        if (best_pos != 0 && best_pos < raw) {
            in_lim = best_pos;
            in_state = 1;
            return 0;
        }
    } else {
        if (best_pos != 0 && best_pos < raw) {
            int pos = best_pos;
            memmove(&corr_buffer[0], &corr_buffer[pos], (raw - pos) * sizeof(cf));
            memcpy(&corr_buffer[raw - pos], in_buffer.data(), pos * sizeof(cf));
            best_pos = 0;
        }
        in_lim = raw;
        in_state = 0;
    }
    out = corr_buffer;
    return 1;
}

// symbol index (inside the PLFRAME) of pilot block b, standard layout
static inline int pilot_start(int b) { return 90 + (b + 1) * 1440 + b * 36; }

float S2Rx::coarse_fed(const cf* frame, const FrameCtx& fc) const {   // dvbs2_fed.h:7-48 (pilot blocks at their standard positions)
    const PlTables& T = pl_tables();
    const s2::ModcodParams& mp = fc.mp;
    const int pls_code = fc.pls_code;
    float err = 0, symcnt = 90 - 2;
    auto term = [&](cf a2, cf r2, cf a0, cf r0) { return cmul(cmul(cmul(a2, cconj(r2)), cconj(a0)), r0).im; };
    auto refsym = [&](int i) { return i < 26 ? T.sof[i] : T.plsc_sym[pls_code][i - 26]; };
    for (int i = 0; i < 24; i++) err += term(frame[i + 2], refsym(i + 2), frame[i], refsym(i));
    err += term(frame[26], refsym(26), frame[24], refsym(24));
    err += term(frame[27], refsym(27), frame[25], refsym(25));
    for (int i = 26; i < 88; i++) err += term(frame[i + 2], refsym(i + 2), frame[i], refsym(i));
    if (mp.pilots) {
        const cf p{0.707f, 0.707f};
        for (int b = 0; b < mp.pilot_blocks; ++b) {
            int start = pilot_start(b);
            cf d1{0, 0}, d2{0, 0};
            for (int i = 0; i < 36; ++i) {
                cf descr = pl_descramble(frame[start + i], T.Rn[start - 90 + i]);
                if (i >= 2) err += cmul(cmul(cmul(descr, cconj(p)), cconj(d2)), p).im;
                d2 = d1; d1 = descr;
            }
            symcnt += 36 - 2;
        }
    }
    return err / symcnt;
}

void S2Rx::pll(const cf* in, cf* out, const FrameCtx& fc) {   // dvbs2_pll.cpp:34-86
    const PlTables& T = pl_tables();
    const s2::ModcodParams& mp = fc.mp;
    const int pls_code = fc.pls_code;
    const Constellation& constel = *fc.constel;
    const int total = mp.plframe;
    int next_pilot = mp.pilots && mp.pilot_blocks > 0 ? pilot_start(0) : -1, pb = 0;
    // pilot-aided mode (own extension; off = the reference's loop): the known symbols -- the 90 header symbols
    // and every 36-symbol pilot block -- additionally give a BLOCK estimate of the residual phase, the argument of the sum of
    // (derotated symbol x conj(known symbol)); at the end of the block the loop phase is moved by it.  The decision-directed loop
    // between the blocks is unchanged, but it can no longer stay in one of the constellation's rotational false locks.
    cf acc{0, 0};
    auto snap = [&]() {
        if (!cfg.pilot_aided) return;
        if (acc.re != 0.f || acc.im != 0.f) {
            pll_pcl.phase += cphase(acc);
            pll_pcl.advance(0.f);                       // (wraps the phase; the frequency moves by beta * 0)
        }
        acc = cf{0, 0};
    };
    Pcl pcl90;
    for (int i = 0; i < total; i++) {
        if (i == 90) pcl90 = pll_pcl;
        cf tmp_val = cmul(in[i], phasor(-pll_pcl.phase));
        float error = 0;
        bool block_end = false;
        if (i >= 90) {
            cf descr = pl_descramble(tmp_val, T.Rn[i - 90]);
            bool is_pilot = false;
            if (next_pilot >= 0 && i >= next_pilot && i < next_pilot + 36) is_pilot = true;
            if (!is_pilot) {
                constel.soft_lut(tmp_val, nullptr, &error);
            } else {
                if (cfg.pilot_aided) {
                    const cf pr = cmul(descr, cf{0.707f, -0.707f});
                    error = cphase(pr);   // own extension: data-aided on the known (1+j)/sqrt2 pilot
                    acc = cadd(acc, pr);
                } else {
                    // dvbs2_pll.cpp:58: decision-directed on the sign-sliced QPSK point, a tenth of the gain
                    const cf pt{descr.re > 0 ? 0.707f : -0.707f, descr.im > 0 ? 0.707f : -0.707f};
                    error = cphase(cmul(descr, cconj(pt))) / 10.0f;
                }
                if (i == next_pilot + 35) { ++pb; next_pilot = pb < mp.pilot_blocks ? pilot_start(pb) : -1; block_end = true; }
            }
            out[i] = descr;
        } else {
            const cf pr = cmul(tmp_val, cconj(i < 26 ? T.sof[i] : T.plsc_sym[pls_code][i - 26]));
            error = cphase(pr);
            acc = cadd(acc, pr);
            block_end = i == 89;
            out[i] = (i & 1) ? cf{-tmp_val.re, tmp_val.im} : cf{tmp_val.im, tmp_val.re};
        }
        pll_pcl.advance(error);
        if (block_end) snap();
    }
    if (study_tile > 0 && !(mp.pilots && mp.pilot_blocks > 0) && !cfg.pilot_aided) {
        const Pcl end = pll_pcl;
        pll_tile_study(in, fc, pcl90);
        pll_pcl = end;
    }
}

// The payload loop (dvbs2_pll.cpp:81: freq += beta e, clamp, phase += freq + alpha e, wrap) in tiles: what a symbol contributes, e[k], depends on the loop
// phase at that symbol only (rotation, LUT cell).  Per tile: evaluate e[k] for ALL symbols from guessed phases (pass 1: the phase at the tile's start advanced by
// the frequency alone), replay the recurrence over the tile with those e[k] in the reference's order, evaluate again from the replayed phases ... until a replay
// reproduces the phases its errors were evaluated at.  By induction over k that fixed point IS the serial result (phase[0] is exact; exact phase[k] gives exact
// e[k], hence exact phase[k + 1]).  This routine counts the evaluation passes per tile and checks the fixed point against the serial loop.
void S2Rx::pll_tile_study(const cf* in, const FrameCtx& fc, Pcl p0) {
    const Constellation& constel = *fc.constel;
    const int total = fc.mp.plframe, T = study_tile;
    Pcl truth = p0;
    std::vector<float> ph(T + 1), ph2(T + 1), e(T);
    for (int base = 90; base < total; base += T) {
        const int n = std::min(T, total - base);
        // serial truth over the tile
        Pcl st = truth;
        for (int k = 0; k < n; ++k) { float err = 0; constel.soft_lut(cmul(in[base + k], phasor(-st.phase)), nullptr, &err); st.advance(err); }
        // fixed point
        Pcl g = truth;
        for (int k = 0; k < n; ++k) { ph[k] = g.phase; g.advance(0.f); }
        int passes = 0;
        Pcl r;
        for (;;) {
            ++passes;
            for (int k = 0; k < n; ++k) { float err = 0; constel.soft_lut(cmul(in[base + k], phasor(-ph[k])), nullptr, &err); e[k] = err; }
            r = truth;
            bool same = true;
            for (int k = 0; k < n; ++k) { ph2[k] = r.phase; same &= (ph2[k] == ph[k]); r.advance(e[k]); }
            if (same || passes >= 33) break;
            std::swap(ph, ph2);
        }
        study_hist[passes < 33 ? passes : 33]++;
        if (r.phase != st.phase || r.freq != st.freq) study_mismatch++;
        {
            // the form the engine runs (s2_rx_kernels.hip, S2_PLL_TILES): guessed phases = start + k * freq; a pass evaluates every symbol's table cell, the replay restarts at
            // the first symbol whose cell changed (everything before it stands), done when no cell changes
            std::vector<int> cell(n, -1);
            std::vector<float> pk(n + 1), fk(n + 1);
            for (int k = 0; k < n; ++k) { pk[k] = truth.phase + (float)k * truth.freq; fk[k] = truth.freq; }
            Pcl q = truth;
            for (;;) {
                ++study_evals;
                int c = -1;
                for (int k = 0; k < n; ++k) {
                    float err = 0;
                    const cf rv = cmul(in[base + k], phasor(-pk[k]));
                    const int cl = Constellation::lut_index(rv.re) * 256 + Constellation::lut_index(rv.im);
                    constel.soft_lut(rv, nullptr, &err);
                    if (cl != cell[k]) { cell[k] = cl; e[k] = err; if (c < 0) c = k; }
                }
                if (c < 0) break;
                q = truth;
                if (c > 0) { q.phase = pk[c]; q.freq = fk[c]; }
                for (int k = c; k < n; ++k) { pk[k] = q.phase; fk[k] = q.freq; q.advance(e[k]); ++study_steps; }
            }
            study_syms += n;
            if (q.phase != st.phase || q.freq != st.freq) study_mismatch++;
        }
        truth = st;
    }
}

int soft_plsc_decode(const float soft[64], float* ratio) {
    const PlTables& T = pl_tables();
    float best = 0.f;
    int bi = 0;
    for (int c = 0; c < 128; ++c) {
        float m = 0.f;
        for (int p = 0; p < 64; ++p) m += ((T.plsc_code[c] >> (63 - p)) & 1) ? -soft[p] : soft[p];
        if (c == 0 || m > best) { best = m; bi = c; }
    }
    float tot = 0.f;
    for (int p = 0; p < 64; ++p) tot += fabsf(soft[p]);
    if (ratio) *ratio = tot > 0.f ? best / tot : 0.f;
    return bi;
}

int S2Rx::pls_at(const cf* x, float* ratio, float* sofq) {
    const PlTables& T = pl_tables();
    cf z{0, 0};
    float amp = 0.f;
    for (int k = 0; k < 26; ++k) { z = cadd(z, cmul(x[k], cconj(T.sof[k]))); amp += camp(x[k]); }
    *sofq = amp > 0.f ? camp(z) / amp : 0.f;
    float soft[64];
    for (int p = 0; p < 64; ++p) {
        // pi/2-BPSK: even positions sit on +-(1+j)/sqrt2, odd ones on +-(-1+j)/sqrt2 (s2_defs.h:60-80); bit 0 = the + sign
        const cf w = cmul(x[26 + p], cconj(z));
        soft[p] = (p & 1) ? (w.im - w.re) : (w.re + w.im);
    }
    return soft_plsc_decode(soft, ratio);
}

void S2Rx::plhdr(const cf* in, cf* out, int* modcod, int* sh, int* pil, int plframe) {   // dvbs2_plhdr_demod.cpp:33-79
    const PlTables& T = pl_tables();
    for (int i = 0; i < 90; i++) {
        cf tmp_val = cmul(in[i], phasor(-hdr_pcl.phase));
        float error = ((tmp_val.re > 0 ? 1.0f : -1.0f) * tmp_val.im) - ((tmp_val.im > 0 ? 1.0f : -1.0f) * tmp_val.re);
        out[i] = (i & 1) ? cf{-tmp_val.re, tmp_val.im} : cf{tmp_val.im, tmp_val.re};
        hdr_pcl.advance(error);
    }
    hdr_pcl.phase += hdr_pcl.freq * (plframe - 91);
    hdr_pcl.advance(0);
    uint64_t plheader = 0;
    const cf rot{(float)0.70710678118654757, (float)-0.70710678118654746};   // (cos, sin)(-pi/4) in double, cast (dvbs2_plhdr_demod.cpp:48)
    for (int y = 0; y < 64; y++) {
        bool value = cmul(out[26 + y], rot).re > 0;
        plheader = plheader << 1 | (uint64_t)(!value);
    }
    int best = 0, diffs = 64;
    if (cfg.soft_plsc) {
        // own extension: soft ML decode over all 64 bits of the demodulated header
        float soft[64];
        for (int y = 0; y < 64; y++) soft[y] = cmul(out[26 + y], rot).re;
        best = soft_plsc_decode(soft, nullptr);
    } else
    for (int c = 0; c < 128; c++) {
        uint64_t x = (T.plsc_code[c] ^ plheader) & ((1ull << 60) - 1);   // only bits 59..0 are compared (:71)
        int d = __builtin_popcountll(x);
        if (d < diffs) { best = c; diffs = d; }
    }
    *modcod = (best >> 2) & 31; *sh = (best & 2) >> 1; *pil = best & 1;
}

void S2Rx::to_soft(const cf* pllout, int8_t* llr, const FrameCtx& fc) const {   // dvbs2_bb_to_soft.cpp:7-33 with pilots skipped
    const s2::ModcodParams& mp = fc.mp;
    const Constellation& constel = *fc.constel;
    const int bits = mp.bits, N = mp.fec.N;
    std::vector<int8_t> soft(N);
    int sym = 90, idx = 0;
    for (int slot = 0; slot < mp.slots; ++slot) {
        for (int k = 0; k < 90; ++k) constel.soft_lut(pllout[sym++], &soft[(size_t)(idx++) * bits], nullptr);
        if (mp.pilots && ((slot + 1) % 16 == 0) && slot + 1 < mp.slots) sym += 36;
    }
    s2_deinterleave(mp.constel, mp.rate, mp.shortframe, soft.data(), llr);
}

const S2Rx::FrameCtx* S2Rx::ctx_for_pls(int pls) {
    if (pls < 0 || pls > 127) return nullptr;
    if (vcm_ctx[pls]) return vcm_ctx[pls].get();
    const int modcod = pls >> 2, sh = (pls >> 1) & 1, pil = pls & 1;
    auto fc = std::make_unique<FrameCtx>();
    if (modcod == 0) {                                   // dummy PLFRAME: header + 36 unmodulated slots, nothing to decode
        fc->mp = s2::ModcodParams{};
        fc->mp.modcod = 0; fc->mp.plframe = VCM_DUMMY_PLFRAME; fc->mp.slots = 36;
        fc->pls_code = pls; fc->constel = nullptr; fc->ldpc = nullptr; fc->bch = nullptr; fc->dummy = true;
    } else {
        s2::ModcodParams p;
        if (!s2::modcod_params(modcod, sh, pil, &p)) return nullptr;
        vcm_constel[pls] = std::make_unique<Constellation>(p.constel, p.g1, p.g2);
        fc->mp = p; fc->pls_code = pls; fc->constel = vcm_constel[pls].get();
        fc->ldpc = get_ldpc(p.fec.code_index); fc->bch = get_bch(p.fec); fc->dummy = false;
    }
    vcm_ctx[pls] = std::move(fc);
    return vcm_ctx[pls].get();
}

// one aligned PLFRAME through FED -> NCO feedback -> PLL -> PLHDR -> demap -> FEC (module_dvbs2_demod.cpp:317-366)
void S2Rx::process_frame(const cf* frame, const FrameCtx& fc, float best_match, uint8_t* out, int out_cap, int* outcnt) {
    const s2::ModcodParams& m = fc.mp;
    const int kb = m.fec.kbch / 8;
    std::vector<cf> pllout(m.plframe), hdr(90);
    std::vector<int8_t> llr(m.fec.N);
    FrameStats st{};
    st.best_match = best_match;
    st.bbframe_bytes = kb;
    float est = coarse_fed(frame, fc);   // module_dvbs2_demod.cpp:319-331
    st.fed_err = est;
    if (std::abs(est) < 0.02) nco_freq_ = nco_freq_ + est * (cfg.fll_bw / 100.0f);
    else nco_freq_ = nco_freq_ + est * cfg.fll_bw;
    if (nco_freq_ > 0.3f * (float)M_PI) nco_freq_ = 0.3f * (float)M_PI;
    if (nco_freq_ < -0.3f * (float)M_PI) nco_freq_ = -0.3f * (float)M_PI;
    pll(frame, pllout.data(), fc);
    plhdr(frame, hdr.data(), &st.detect_modcod, &st.detect_short, &st.detect_pilots, m.plframe);
    if (cfg.acm_vcm) { st.detect_modcod = fc.pls_code >> 2; st.detect_short = (fc.pls_code >> 1) & 1; st.detect_pilots = fc.pls_code & 1; }   // what the framing decoded
    for (int k = 0; k < 90; ++k) pllout[k] = hdr[k];
    to_soft(pllout.data(), llr.data(), fc);
    dbg_frames.insert(dbg_frames.end(), frame, frame + m.plframe);
    dbg_pll.insert(dbg_pll.end(), pllout.begin(), pllout.end());
    dbg_llr.insert(dbg_llr.end(), llr.begin(), llr.end());
    if (cfg.force_ldpc_iters < 0) { dbg_stats.push_back(st); return; }   // front-end timing only (bench cpu_baseline)
    int mt = cfg.force_ldpc_iters ? cfg.force_ldpc_iters : cfg.max_ldpc_trials;
    st.ldpc_trials = ldpc_decode(*fc.ldpc, llr.data(), mt, cfg.force_ldpc_iters ? 1 : 0);
    std::vector<uint8_t> fr(m.fec.K / 8);
    hard_pack(llr.data(), m.fec.K, fr.data());
    st.bch_corr = bch_decode(*fc.bch, fr.data());
    bb_descramble(fr.data(), kb);
    if (*outcnt + kb <= out_cap) { memcpy(out + *outcnt, fr.data(), kb); *outcnt += kb; }
    dbg_stats.push_back(st);
}

// ACM/VCM framing (own definition; the reference's PL sync assumes ONE frame length, dvbs2_pl_sync.cpp:81-165, and its GUI re-configures
// the whole demodulator after 50 consistent PLS sightings, main.cpp:375-408).  Works on the FIFO of 1-sps symbols:
//   not locked: the reference's differential SOF + PLSC correlator (dvbs2_pl_sync.cpp:111-143: it does not depend on the MODCOD) over the
//               next VCM_ACQ_WINDOW offsets; the best one (same arg-max rule) becomes the frame start.
//   locked:     soft PLS decode at the frame start (pls_at); a valid code with correlation ratio >= VCM_MIN_RATIO and SOF quality >=
//               sof_threshold names the frame's MODCOD / size / pilots and thereby its length; the frame is handed on when all its
//               symbols are in, the next header is expected right behind it.  Anything else drops the lock and the search resumes one
//               symbol further.
// Dummy PLFRAMEs (MODCOD 0) only advance the framing.  A frame's statistics report its PLS code and SOF quality.
void S2Rx::vcm_walk(uint8_t* out, int out_cap, int* outcnt) {
    size_t cur = 0;
    while (true) {
        const size_t avail = vfifo.size() - cur;
        if (!vcm_synced) {
            if (avail < (size_t)VCM_ACQ_WINDOW + 90) break;
            const cf* w = &vfifo[cur];
            const uint32_t dsof = 0x18d2e82u ^ (0x18d2e82u >> 1);
            const uint64_t SCR = 0x719d83c953422dfaull;
            const uint64_t dscr = SCR ^ (SCR >> 1);
            double best_match = 0;
            int bp = 0;
            cf d90[90];
            for (int ss = 0; ss < VCM_ACQ_WINDOW; ss++) {
                d90[0] = cf{0, 0};
                for (int k = 1; k < 90; ++k) d90[k] = cmul(cconj(w[ss + k - 1]), w[ss + k]);
                cf csof{0, 0};
                for (int i = 0; i < 26; ++i) {
                    if (((dsof >> (25 - i)) ^ i) & 1) csof = cadd(csof, d90[i]);
                    else csof = csub(csof, d90[i]);
                }
                cf cpl{0, 0};
                for (int i = 1; i < 64; i += 2) {
                    if ((dscr >> (63 - i)) & 1) cpl = csub(cpl, d90[26 + i]);
                    else cpl = cadd(cpl, d90[26 + i]);
                }
                cf c0 = cadd(csof, cpl), c1 = csub(csof, cpl);
                cf c = camp(c0) > camp(c1) ? c0 : c1;
                cf d = cscale(c, 1.0f / (26 - 1 + 64 / 2));
                double difference = camp(d);
                if (difference > best_match && d.im > 0) { best_match = difference; bp = ss; }
            }
            if (best_match > 0) { cur += bp; vcm_synced = true; }
            else cur += VCM_ACQ_WINDOW;
            continue;
        }
        if (avail < 90) break;
        float ratio, sofq;
        const int pls = pls_at(&vfifo[cur], &ratio, &sofq);
        const FrameCtx* fc = (ratio >= VCM_MIN_RATIO && sofq >= cfg.sof_threshold) ? ctx_for_pls(pls) : nullptr;
        if (!fc) { vcm_synced = false; cur += 1; continue; }
        if (avail < (size_t)fc->mp.plframe) break;
        if (fc->dummy) {
            FrameStats st{};
            st.best_match = sofq; st.detect_modcod = 0; st.detect_short = (pls >> 1) & 1; st.detect_pilots = pls & 1;
            st.ldpc_trials = 0; st.bch_corr = 0; st.bbframe_bytes = 0;
            dbg_stats.push_back(st);
            dbg_frames.insert(dbg_frames.end(), vfifo.begin() + cur, vfifo.begin() + cur + fc->mp.plframe);   // (tap 1 shows every frame the framing found)
        } else {
            process_frame(&vfifo[cur], *fc, sofq, out, out_cap, outcnt);
        }
        cur += fc->mp.plframe;
    }
    vfifo.erase(vfifo.begin(), vfifo.begin() + cur);
}

int S2Rx::process(int count, const cf* in, uint8_t* out, int out_cap) {
    dbg_symbols.clear(); dbg_frames.clear(); dbg_pll.clear(); dbg_llr.clear(); dbg_stats.clear();
    work1.resize(count + 16); work2.resize(count + 1024);
    agc(count, in, work1.data());
    nco(count, work1.data(), work1.data());
    int n = gardner(count, work1.data(), work2.data());
    rrc_filter(n, work2.data(), work2.data());
    std::vector<cf> syms;
    for (int i = 0; i < n; ++i) {   // module_dvbs2_demod.cpp:231-239
        if (cr_samp) syms.push_back(work2[i]);
        cr_samp = !cr_samp;
    }
    dbg_symbols = syms;
    int outcnt = 0;
    if (cfg.acm_vcm) {
        vfifo.insert(vfifo.end(), syms.begin(), syms.end());
        vcm_walk(out, out_cap, &outcnt);
        return outcnt;
    }
    std::vector<cf> frame;
    for (size_t i = 0; i < syms.size(); ++i) {   // dvbs2_pl_sync.cpp:81-100
        in_buffer[in_ptr++] = syms[i];
        if (in_ptr >= in_lim) {
            float bm = last_bm;   // S2PLSyncBlock::best_match is a member: state 1 reports the value of the last correlation
            int got = plsync_internal(frame, &bm);
            last_bm = bm;
            in_ptr = 0;
            if (!got) continue;
            process_frame(frame.data(), ccm, bm, out, out_cap, &outcnt);
        }
    }
    return outcnt;
}

// ------------------------------------------------------------------------------------------ transmitter
static inline uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rrc_cont(double t, double beta) {   // unit-energy RRC, t in symbols
    const double PI = 3.14159265358979323846;
    if (std::fabs(t) < 1e-9) return 1.0 + beta * (4.0 / PI - 1.0);
    if (std::fabs(std::fabs(t) - 1.0 / (4.0 * beta)) < 1e-9)
        return (beta / std::sqrt(2.0)) * ((1.0 + 2.0 / PI) * std::sin(PI / (4.0 * beta)) + (1.0 - 2.0 / PI) * std::cos(PI / (4.0 * beta)));
    double x = 4.0 * beta * t;
    return (std::sin(PI * t * (1.0 - beta)) + x * std::cos(PI * t * (1.0 + beta))) / (PI * t * (1.0 - x * x));
}

std::vector<cf> s2_transmit(const TxCfg& t, std::vector<uint8_t>* bbframes_out, std::vector<cf>* symbols_out) {
    const PlTables& T = pl_tables();
    // per-frame parameters: CCM = the configured MODCOD for every frame, ACM/VCM = the PLS code list, cycled
    struct Per { s2::ModcodParams mp; std::unique_ptr<Constellation> C; const LdpcCode* ldpc; const BchCode* bch; float norm; bool dummy; };
    std::map<int, Per> per;
    auto get = [&](int pls) -> Per& {
        auto it = per.find(pls);
        if (it != per.end()) return it->second;
        Per P;
        const int modcod = pls >> 2;
        if (modcod == 0) {
            P.mp = s2::ModcodParams{}; P.mp.plframe = VCM_DUMMY_PLFRAME; P.mp.slots = 36; P.dummy = true; P.norm = 1.f; P.ldpc = nullptr; P.bch = nullptr;
        } else {
            if (!s2::modcod_params(modcod, (pls >> 1) & 1, pls & 1, &P.mp)) throw std::runtime_error("bad MODCOD");
            P.C = std::make_unique<Constellation>(P.mp.constel, P.mp.g1, P.mp.g2);
            P.ldpc = get_ldpc(P.mp.fec.code_index);
            P.bch = get_bch(P.mp.fec);
            // normalise payload points to unit average power (the standard's convention); header is unit amplitude
            double pw = 0;
            for (int v = 0; v < P.C->states; ++v) { cf p = P.C->mod(v); pw += (double)p.re * p.re + (double)p.im * p.im; }
            P.norm = (float)(1.0 / std::sqrt(pw / P.C->states));
            P.dummy = false;
        }
        return per.emplace(pls, std::move(P)).first->second;
    };
    const int ccm_pls = t.modcod << 2 | (t.shortframes ? 2 : 0) | (t.pilots ? 1 : 0);
    std::vector<cf> syms;
    uint64_t rs = t.seed ^ 0xABCDEF12345ull;
    for (int i = 0; i < t.lead_symbols; ++i) {
        uint64_t r = sm64(rs);
        syms.push_back(cf{(r & 1) ? 0.70710678f : -0.70710678f, (r & 2) ? 0.70710678f : -0.70710678f});
    }
    if (bbframes_out) bbframes_out->clear();
    for (int f = 0; f < t.nframes; ++f) {
        const int pls = t.vcm_n > 0 ? t.vcm_pls[f % t.vcm_n] : ccm_pls;
        Per& P = get(pls);
        const s2::ModcodParams& mp = P.mp;
        for (int i = 0; i < 26; ++i) syms.push_back(T.sof[i]);
        for (int i = 0; i < 64; ++i) syms.push_back(T.plsc_sym[pls][i]);
        int scr = 0;
        if (P.dummy) {
            for (int k = 0; k < 36 * 90; ++k) syms.push_back(pl_scramble(cf{0.70710678f, 0.70710678f}, T.Rn[scr++]));
            continue;
        }
        const int kb = mp.fec.kbch / 8, N = mp.fec.N, bits = mp.bits;
        std::vector<uint8_t> code(N), inter(N), fr(mp.fec.K / 8);
        make_bbframe(fr.data(), mp.fec.kbch, t.seed * 1000003ull + f);
        if (bbframes_out) bbframes_out->insert(bbframes_out->end(), fr.begin(), fr.begin() + kb);
        bb_descramble(fr.data(), kb);
        bch_encode(*P.bch, fr.data());
        for (int i = 0; i < mp.fec.K; ++i) code[i] = (fr[i / 8] >> (7 - i % 8)) & 1;
        ldpc_encode(*P.ldpc, code.data());
        // bit interleaver = inverse of s2_deinterleave
        if (mp.constel == s2::C_QPSK) {
            for (int i = 0; i < N / 2; ++i) { inter[2 * i] = code[2 * i + 1]; inter[2 * i + 1] = code[2 * i]; }
        } else {
            int rows = N / bits, col[5];
            for (int c = 0; c < bits; ++c) col[c] = rows * c;
            if (mp.constel == s2::C_8PSK && mp.rate == s2::R3_5) { col[0] = rows * 2; col[1] = rows; col[2] = 0; }
            for (int j = 0; j < rows; ++j)
                for (int c = 0; c < bits; ++c) inter[bits * j + c] = code[col[c] + j];
        }
        int sidx = 0;
        for (int slot = 0; slot < mp.slots; ++slot) {
            for (int k = 0; k < 90; ++k) {
                int v = 0;
                for (int b = 0; b < bits; ++b) v = (v << 1) | (inter[(size_t)sidx * bits + b] ^ 1);   // labels are complemented
                ++sidx;
                syms.push_back(pl_scramble(cscale(P.C->mod(v), P.norm), T.Rn[scr++]));
            }
            if (mp.pilots && ((slot + 1) % 16 == 0) && slot + 1 < mp.slots)
                for (int k = 0; k < 36; ++k) syms.push_back(pl_scramble(cf{0.70710678f, 0.70710678f}, T.Rn[scr++]));
        }
    }
    if (symbols_out) *symbols_out = syms;
    // pulse shaping at 2 samples/symbol with a fractional timing offset, then CFO, phase and AWGN
    const int ns = (int)syms.size();
    const int nsamp = t.nsamples > 0 ? t.nsamples : 2 * ns;
    const int span = 16;
    const double beta = 0.35;
    std::vector<cf> iq(nsamp);
    double frac = t.timing;
    if (nsamp == 2 * ns) {
        // tabulate the two polyphase branches of h((n - timing)/2 - k)
        std::vector<double> h0(2 * span + 1), h1(2 * span + 1);
        for (int k = -span; k <= span; ++k) {
            h0[k + span] = rrc_cont((0.0 - frac) / 2.0 - k, beta);
            h1[k + span] = rrc_cont((1.0 - frac) / 2.0 - k, beta);
        }
        for (int n = 0; n < nsamp; ++n) {
            int m = n >> 1;
            const std::vector<double>& h = (n & 1) ? h1 : h0;
            double re = 0, im = 0;
            for (int k = -span; k <= span; ++k) {
                int si = m - k;
                if (t.circular) si = ((si % ns) + ns) % ns;
                else if (si < 0 || si >= ns) continue;
                // contribution of symbol si at time (n - frac)/2: h((n - frac)/2 - si) = h(((n&1) - frac)/2 + k) -> index -k
                double hv = h[-k + span];
                re += hv * syms[si].re; im += hv * syms[si].im;
            }
            iq[n] = cf{(float)re, (float)im};
        }
    } else {
        // sampling-clock error: output sample n sits at symbol time (n * 2ns/nsamp - frac) / 2; the pulse comes from a table of the
        // continuous RRC (1/4096 symbol steps, linear interpolation: error < 1e-7, far below the noise floor of any test)
        const int OS = 4096, half = (span + 2) * OS;
        std::vector<float> tab(2 * half + 2);
        for (int i = 0; i <= 2 * half + 1; ++i) tab[i] = (float)rrc_cont((double)(i - half) / OS, beta);
        const double step = (double)(2 * ns) / (double)nsamp;
        for (int n = 0; n < nsamp; ++n) {
            const double tau = ((double)n * step - frac) / 2.0;
            const int m = (int)std::floor(tau);
            double re = 0, im = 0;
            for (int k = -span; k <= span; ++k) {
                int si = m - k;
                if (t.circular) si = ((si % ns) + ns) % ns;
                else if (si < 0 || si >= ns) continue;
                const double x = (tau - (double)(m - k)) * OS + half;       // h(tau - si)
                const int xi = (int)x;
                const double fx = x - xi;
                const double hv = (1.0 - fx) * tab[xi] + fx * tab[xi + 1];
                re += hv * syms[si].re; im += hv * syms[si].im;
            }
            iq[n] = cf{(float)re, (float)im};
        }
    }
    double sigma = 0;
    if (t.esn0_db < 100) sigma = std::sqrt(0.5 * std::pow(10.0, -t.esn0_db / 10.0) * 1.0);   // per real dimension; Es = 1, 2 sps after MF
    uint64_t ns_state = t.seed * 7919ull + 17;
    auto gauss = [&](double& g0, double& g1) {
        double u1 = ((sm64(ns_state) >> 11) + 1.0) / 9007199254740993.0, u2 = (sm64(ns_state) >> 11) / 9007199254740992.0;
        double r = std::sqrt(-2.0 * std::log(u1)), a = 2.0 * 3.14159265358979323846 * u2;
        g0 = r * std::cos(a); g1 = r * std::sin(a);
    };
    for (int n = 0; n < nsamp; ++n) {
        double ph = t.phase0 + t.cfo * n;
        double c = std::cos(ph), s = std::sin(ph);
        double re = iq[n].re * c - iq[n].im * s, im = iq[n].re * s + iq[n].im * c;
        if (sigma > 0) {
            double g0, g1; gauss(g0, g1);
            // white noise at 2 sps with variance such that after the (unit-energy) matched filter Es/N0 holds:
            // noise PSD N0 -> per-sample variance N0 * fs/ (fs_sym) /2 per dim = sigma^2 * 2 (2 sps)
            re += g0 * sigma * std::sqrt(2.0); im += g1 * sigma * std::sqrt(2.0);
        }
        iq[n] = cf{(float)re, (float)im};
    }
    return iq;
}

}  // namespace orc
