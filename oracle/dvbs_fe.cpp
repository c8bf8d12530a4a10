// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// CPU restatement of the DVB-S front end demod::QPSK_ALT (reference src/demod/common/dsp/demod/qpsk_alt.cpp:11-30,136-144):
//   loop::FastAGC -> loop::FLL (fll.cpp:61-95,135-149) -> filter::FIR RRC -> clock_recovery::COMPLEX_FD (complex_fd.cpp:89-158)
//   -> loop::Costas<4> (SDR++ core).
// PARITY UNPINNED: SDR++ core (FastAGC, FIR, PhaseControlLoop, Costas, windowedSinc, fastAmplitude) and VOLK are not in
// /root/reference and the reference has no vectors for this path; the restatement follows the call sites above and the
// definitions listed in SURVEY.md Appendix C.  Summation orders: the band-edge and RRC FIRs accumulate tap 0..n-1 in order (VOLK's
// generic kernels); the three 256-tap interpolator dot products of COMPLEX_FD use 64 interleaved partial sums and a fixed
// tree (fd_dot: pairwise inside groups of 16, then the four group sums) -- VOLK picks a SIMD kernel at run time there, so the reference's own order is machine dependent.
#include "dvbs_fe.h"
#include <cmath>
#include <cstring>
#include <random>

namespace orc {

QpskAltCfg qpsk_alt_default_cfg() {   // main.cpp:64-73,134-139
    QpskAltCfg c;
    c.symbolrate = 2e6; c.samplerate = 4e6;
    c.rrc_taps = 65; c.rrc_alpha = 0.35f; c.agc_rate = 0.0001f; c.costas_bw = 0.00628f; c.fll_bw = 0.006f;
    float bw = 0.00628f, damp = 0.707f;
    float den = (1.0f + 2.0 * damp * bw + bw * bw);
    c.mu_gain = (4.0f * damp * bw) / den;
    c.omega_gain = (4.0f * bw * bw) / den;
    c.omega_rel_limit = 0.02f;
    return c;
}

static double sinc_d(double x) { return x == 0.0 ? 1.0 : sin(x) / x; }
static float fast_amplitude(cf v) {   // SDR++ complex_t::fastAmplitude
    float re_abs = fabsf(v.re), im_abs = fabsf(v.im);
    if (re_abs > im_abs) return re_abs + 0.4f * im_abs;
    return im_abs + 0.4f * re_abs;
}

QpskAlt::QpskAlt(const QpskAltCfg& c) : cfg(c) {
    const float PI_F = 3.14159265358979323846f;
    // FLL::init (fll.cpp:10-29): note sym_rate / samp_rate are int parameters there
    const int T = cfg.rrc_taps;
    {
        float sps = (float)((double)(int)cfg.samplerate / (double)(int)cfg.symbolrate);
        const int M = (int)(T / sps);
        float power = 0;
        std::vector<float> bb(T);
        for (int i = 0; i < T; i++) {
            float k = -M + i * 2.0f / sps;
            float tap = (float)(sinc_d(cfg.rrc_alpha * k - 0.5f) + sinc_d(cfg.rrc_alpha * k + 0.5f));
            power += tap;
            bb[i] = tap;
        }
        lbe.resize(T); hbe.resize(T);
        int N = (int)((T - 1.0f) / 2.0f);
        for (int i = 0; i < T; i++) {
            float tap = bb[i] / power;
            float k = (-N + (int)i) / (2.0f * sps);
            cf t1 = cscale(phasor(-2.0f * PI_F * (1.0f + cfg.rrc_alpha) * k), tap);
            cf t2 = cscale(phasor(2.0f * PI_F * (1.0f + cfg.rrc_alpha) * k), tap);
            lbe[T - i - 1] = t1;
            hbe[T - i - 1] = t2;
        }
        float a, b;
        critically_damped(cfg.fll_bw, &a, &b);
        a = 0;
        fll_pcl.init(a, b, 0, -PI_F, PI_F, 0, -PI_F / 2.0f, PI_F / 2.0f, true);
        fll_hist.assign(T - 1, cf{0, 0});
    }
    rrc = rrc_taps(T, cfg.rrc_alpha, cfg.symbolrate, cfg.samplerate);
    rrc_hist.assign(T - 1, cf{0, 0});
    bank = gardner_bank(FD_PHASES, FD_TAPS);
    float omega = (float)(cfg.samplerate / cfg.symbolrate);
    fd_pcl.init(cfg.mu_gain, cfg.omega_gain, 0.0f, 0.0f, 1.0f, omega, (float)(omega * (1.0 - cfg.omega_rel_limit)),
                (float)(omega * (1.0 + cfg.omega_rel_limit)), false);
    fd_hist.assign(FD_TAPS - 1, cf{0, 0});
    float a, b;
    critically_damped(cfg.costas_bw, &a, &b);
    costas_pcl.init(a, b, 0, -PI_F, PI_F, 0, -PI_F / 10.0f, PI_F / 10.0f, true);
}

void QpskAlt::agc(int n, const cf* in, cf* out) {   // SDR++ loop::FastAGC<complex_t>: set point 1, max gain 10e6, initial gain 1
    for (int i = 0; i < n; ++i) {
        out[i] = cscale(in[i], agc_gain);
        float a = sqrtf(out[i].re * out[i].re + out[i].im * out[i].im);
        agc_gain += (1.0f - a) * cfg.agc_rate;
        if (agc_gain > 10e6f) agc_gain = 10e6f;
    }
}

void QpskAlt::fll(int n, const cf* in, cf* out) {   // fll.cpp:135-149
    const int T = cfg.rrc_taps;
    std::vector<cf> win(fll_hist);   // T-1 older rotated samples
    win.resize(T);
    for (int i = 0; i < n; i++) {
        cf x = cmul(in[i], phasor(-fll_pcl.phase));
        win[T - 1] = x;
        cf lo{0, 0}, hi{0, 0};
        for (int k = 0; k < T; ++k) { lo = cadd(lo, cmul(win[k], lbe[k])); hi = cadd(hi, cmul(win[k], hbe[k])); }
        float freqError = fast_amplitude(hi) - fast_amplitude(lo);
        fll_pcl.advance(freqError);
        out[i] = x;
        memmove(win.data(), win.data() + 1, sizeof(cf) * (T - 1));
    }
    for (int k = 0; k < T - 1; ++k) fll_hist[k] = win[k];
}

void QpskAlt::rrc_filter(int n, const cf* in, cf* out) {
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

cf fd_dot(const cf* x, const float* t) {
    float pr[64], pi[64];
    for (int l = 0; l < 64; ++l) {
        float ar = 0.f, ai = 0.f;
        for (int q = 0; q < 4; ++q) { ar += x[l + 64 * q].re * t[l + 64 * q]; ai += x[l + 64 * q].im * t[l + 64 * q]; }
        pr[l] = ar; pi[l] = ai;
    }
    // fixed tree: inside each group of 16 partial sums pairwise at distance 8, 4, 2, 1, then (group 0 + group 1) + (group 2 + group 3)
    // (on the GPU: four DPP row shifts and three additions of the row leaders -- no cross-row permute)
    for (int r = 0; r < 4; ++r)
        for (int s = 8; s >= 1; s >>= 1)
            for (int l = 0; l < s; ++l) { pr[16 * r + l] = pr[16 * r + l] + pr[16 * r + l + s]; pi[16 * r + l] = pi[16 * r + l] + pi[16 * r + l + s]; }
    const float r01 = pr[0] + pr[16], r23 = pr[32] + pr[48], i01 = pi[0] + pi[16], i23 = pi[32] + pi[48];
    return cf{r01 + r23, i01 + i23};
}

int QpskAlt::complex_fd(int count, const cf* in, cf* out) {   // complex_fd.cpp:89-150
    std::vector<cf> buffer(count + FD_TAPS - 1);
    for (int i = 0; i < FD_TAPS - 1; ++i) buffer[i] = fd_hist[i];
    if (count) memcpy(&buffer[FD_TAPS - 1], in, sizeof(cf) * count);
    int outCount = 0;
    while (fd_offset < count) {
        float error;
        cf outVal, dfdt;
        int phase = (int)floorf(fd_pcl.phase * (float)FD_PHASES);
        phase = phase < 0 ? 0 : (phase > FD_PHASES - 1 ? FD_PHASES - 1 : phase);
        outVal = fd_dot(&buffer[fd_offset], &bank[(size_t)phase * FD_TAPS]);
        out[outCount++] = outVal;
        if (phase == 0) {
            cf fT1 = fd_dot(&buffer[fd_offset], &bank[(size_t)(phase + 1) * FD_TAPS]);
            dfdt = csub(fT1, outVal);
        } else if (phase == FD_PHASES - 1) {
            cf fT_1 = fd_dot(&buffer[fd_offset], &bank[(size_t)(phase - 1) * FD_TAPS]);
            dfdt = csub(outVal, fT_1);
        } else {
            cf fT1 = fd_dot(&buffer[fd_offset], &bank[(size_t)(phase + 1) * FD_TAPS]);
            cf fT_1 = fd_dot(&buffer[fd_offset], &bank[(size_t)(phase - 1) * FD_TAPS]);
            dfdt = cscale(csub(fT1, fT_1), 0.5f);
        }
        if (fd_spsctr == 0) error = ((outVal.re * dfdt.re) + (outVal.im * dfdt.im));
        else error = 0;
        fd_spsctr++;
        if (fd_spsctr >= 1) fd_spsctr = 0;   // outSps = 1 (qpsk_alt.cpp:22)
        if (error > 1.0f) error = 1.0f;
        if (error < -1.0f) error = -1.0f;
        fd_pcl.advance(error);
        float delta = floorf(fd_pcl.phase);
        fd_offset += delta;
        fd_pcl.phase -= delta;
    }
    fd_offset -= count;
    for (int i = 0; i < FD_TAPS - 1; ++i) fd_hist[i] = buffer[count + i];
    return outCount;
}

void QpskAlt::costas(int n, const cf* in, cf* out) {   // SDR++ loop::Costas<4>
    for (int i = 0; i < n; ++i) {
        cf v = cmul(in[i], phasor(-costas_pcl.phase));
        out[i] = v;
        float err = ((v.re > 0 ? 1.0f : -1.0f) * v.im) - ((v.im > 0 ? 1.0f : -1.0f) * v.re);
        err = err > 1.0f ? 1.0f : (err < -1.0f ? -1.0f : err);
        costas_pcl.advance(err);
    }
}

int QpskAlt::process(int n, const cf* in, cf* out) {
    std::vector<cf> a(n), b(n);
    agc(n, in, a.data());
    fll(n, a.data(), b.data());
    rrc_filter(n, b.data(), a.data());
    int m = complex_fd(n, a.data(), b.data());
    costas(m, b.data(), out);
    return m;
}

std::vector<cf> dvbs_modulate(const uint8_t* bits, int nsym, double esn0_db, double cfo, double timing, double phase0, uint64_t seed,
                              int ntaps, double alpha) {
    std::vector<float> taps = rrc_taps(ntaps, alpha, 1.0, 2.0);
    const int n = 2 * nsym;
    std::vector<cf> up(n + ntaps, cf{0, 0});
    const float A = 0.70710678f;
    for (int s = 0; s < nsym; ++s) up[2 * s + ntaps / 2] = cf{bits[2 * s] ? A : -A, bits[2 * s + 1] ? A : -A};
    // pulse shaping with a fractional timing offset by linear interpolation of the tap positions is not needed for tests:
    // `timing` shifts the sampling grid by interpolating the shaped waveform with a short windowed sinc
    std::vector<cf> shaped(n);
    for (int i = 0; i < n; ++i) {
        cf acc{0, 0};
        for (int k = 0; k < ntaps; ++k) {
            int p = i + k;
            acc.re += up[p].re * taps[k]; acc.im += up[p].im * taps[k];
        }
        shaped[i] = acc;
    }
    double pw = 0;
    for (auto& v : shaped) pw += (double)v.re * v.re + (double)v.im * v.im;
    pw /= n;
    const double g = 1.0 / sqrt(pw);
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> nd(0.0, 1.0);
    // Es = 2 samples x unit power; noise per sample per dimension
    const double sigma = sqrt(2.0 / pow(10.0, esn0_db / 10.0) / 2.0);
    std::vector<cf> out(n);
    for (int i = 0; i < n; ++i) {
        // fractional delay: 8-tap Hann-windowed sinc around position i + timing
        double re = 0, im = 0;
        if (timing == 0.0) { re = shaped[i].re; im = shaped[i].im; }
        else {
            for (int k = -3; k <= 4; ++k) {
                int p = i + k;
                if (p < 0 || p >= n) continue;
                double x = (double)k - timing;
                double w = 0.5 + 0.5 * cos(M_PI * x / 4.5);
                double sc = (fabs(x) < 1e-12 ? 1.0 : sin(M_PI * x) / (M_PI * x)) * w;
                re += shaped[p].re * sc; im += shaped[p].im * sc;
            }
        }
        double ph = phase0 + cfo * i;
        double c = cos(ph), s = sin(ph);
        double r2 = (re * c - im * s) * g + sigma * nd(rng), i2 = (re * s + im * c) * g + sigma * nd(rng);
        out[i] = cf{(float)r2, (float)i2};
    }
    return out;
}

}  // namespace orc
