// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.  See dvbs_fe.cpp.
#pragma once
#include <vector>
#include "s2chain.h"

namespace orc {

struct QpskAltCfg {          // arguments of demod::QPSK_ALT::init as DVBSDemod::init passes them (module_dvbs_demod.cpp:27)
    double symbolrate, samplerate;
    int rrc_taps;
    float rrc_alpha, agc_rate, costas_bw, fll_bw, omega_gain, mu_gain, omega_rel_limit;
};
QpskAltCfg qpsk_alt_default_cfg();

constexpr int FD_PHASES = 256, FD_TAPS = 256;   // complex_fd.h:33

struct QpskAlt {
    QpskAltCfg cfg;
    float agc_gain = 1.0f;
    // FLL
    Pcl fll_pcl;
    std::vector<cf> lbe, hbe;          // band-edge taps (already reversed, as the reference stores them)
    std::vector<cf> fll_hist;          // rrc_taps - 1 rotated samples
    // RRC
    std::vector<float> rrc;
    std::vector<cf> rrc_hist;
    // COMPLEX_FD
    std::vector<float> bank;           // [256][256]
    Pcl fd_pcl;
    int fd_offset = 0, fd_spsctr = 0;
    std::vector<cf> fd_hist;           // 255 samples
    // Costas
    Pcl costas_pcl;
    explicit QpskAlt(const QpskAltCfg& c);
    // stage functions (each keeps its own state); process = AGC -> FLL -> RRC -> COMPLEX_FD -> Costas (qpsk_alt.cpp:136-144)
    void agc(int n, const cf* in, cf* out);
    void fll(int n, const cf* in, cf* out);
    void rrc_filter(int n, const cf* in, cf* out);
    int complex_fd(int n, const cf* in, cf* out);
    void costas(int n, const cf* in, cf* out);
    int process(int n, const cf* in, cf* out);
};

// 256-tap complex x real dot product in the engine's documented order (see dvbs_fe.cpp): 64 interleaved partial sums, fixed tree
cf fd_dot(const cf* x, const float* t);

// DVB-S modulator for tests: soft-decision bit stream (0/1 per byte, I then Q) -> QPSK, 2 sps, RRC, AWGN, CFO, timing, phase
std::vector<cf> dvbs_modulate(const uint8_t* bits, int nsym, double esn0_db, double cfo, double timing, double phase0, uint64_t seed,
                              int rrc_taps, double alpha);

}  // namespace orc
