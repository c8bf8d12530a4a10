// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// DVB-S2 physical-layer receive chain (float DSP + PL framing) and a synthetic transmitter.
//
// PARITY UNPINNED for every float stage that rests on SDR++ core / VOLK (FastAGC, FIR, tap generators,
// PhaseControlLoop, phasor, the VOLK dot products): those libraries are not in /root/reference and the
// reference has no tests, so these restate the call-site semantics of the plugin plus the textbook
// definitions listed in SURVEY.md Appendix C.  sin/cos/atan2/exp/log are NOT the host libm's: they are the
// engine's own straight-line IEEE definitions (include/dvbs2gpu_math.h), evaluated identically by x86 and
// gfx950, so that the GPU path can be compared with this restatement bit for bit.  What IS pinned here against the reference's own code:
// the S2 deinterleaver (compiled reference, oracle/_ref) and everything integer downstream (LDPC/BCH/BB).
#pragma once
#include <cstdint>
#include <vector>
#include <cmath>
#include <memory>
#include "oracle.h"
#include "../include/dvbs2gpu_math.h"   // the engine's definition of sin/cos/atan2/exp/log (shared with the device code on purpose)

namespace orc {

struct cf { float re, im; };
static inline cf cmul(cf a, cf b) { return cf{a.re * b.re - a.im * b.im, a.im * b.re + a.re * b.im}; }
static inline cf cconj(cf a) { return cf{a.re, -a.im}; }
static inline cf cadd(cf a, cf b) { return cf{a.re + b.re, a.im + b.im}; }
static inline cf csub(cf a, cf b) { return cf{a.re - b.re, a.im - b.im}; }
static inline cf cscale(cf a, float s) { return cf{a.re * s, a.im * s}; }
static inline float camp(cf a) { return sqrtf(a.re * a.re + a.im * a.im); }
static inline float cphase(cf a) { return dvbs2m::atan2f_det(a.im, a.re); }
static inline cf phasor(float x) { cf r; dvbs2m::sincosf_det(x, &r.im, &r.re); return r; }

// SDR++ loop::PhaseControlLoop<float, CLAMP_PHASE> as used by the plugin
struct Pcl {
    float alpha = 0, beta = 0, phase = 0, freq = 0, minPhase = 0, maxPhase = 0, minFreq = 0, maxFreq = 0;
    bool clampPhase = true;
    void init(float a, float b, float ph, float minP, float maxP, float fr, float minF, float maxF, bool clamp) {
        alpha = a; beta = b; phase = ph; minPhase = minP; maxPhase = maxP; freq = fr; minFreq = minF; maxFreq = maxF; clampPhase = clamp;
    }
    void advance(float err) {
        freq += beta * err;
        if (freq > maxFreq) freq = maxFreq; else if (freq < minFreq) freq = minFreq;
        phase += freq + alpha * err;
        if (clampPhase) {
            float delta = maxPhase - minPhase;
            while (phase > maxPhase) phase -= delta;
            while (phase < minPhase) phase += delta;
        }
    }
};
void critically_damped(float bw, float* alpha, float* beta);

// ---- constants / tables
struct PlTables {
    cf sof[26];                 // s2_defs.h:15-30
    uint64_t plsc_code[128];    // s2_defs.h:32-80
    cf plsc_sym[128][64];
    std::vector<uint8_t> Rn;    // 131072 entries, s2_scrambling.cpp:9-28
    PlTables();
};
const PlTables& pl_tables();

// ---- soft demapper (constellation.cpp:19-322)
struct Constellation {
    int type;                   // s2::Constel
    int bits, states;
    float amp, sca, prescale;
    std::vector<cf> pts;
    std::vector<int8_t> lut_bits;   // [256][256][bits], x-major like the reference's lut[x][y]
    std::vector<float> lut_err;     // [256][256]
    Constellation(int type, float g1, float g2);
    cf mod(int sym) const;
    void soft_calc(cf s, int8_t* bits_out, float* phase_err) const;
    void soft_lut(cf s, int8_t* bits_out, float* phase_err) const;
    static int lut_index(float v);
};

void s2_deinterleave(int constel, int rate, int shortframe, const int8_t* in, int8_t* out);

// ---- stage functions
std::vector<float> rrc_taps(int count, double beta, double symbolrate, double samplerate);
std::vector<float> gardner_bank(int phases, int taps_per_phase);   // [phase][tap]

struct DemodCfg {           // same fields / meaning as dvbs2gpu_demod_cfg in include/dvbs2gpu.h
    double symbolrate, samplerate;
    float agc_rate, rrc_alpha;
    int rrc_taps;
    float loop_bw, fll_bw, clock_omega_gain, clock_mu_gain, omega_rel_limit;
    int modcod, shortframes, pilots;
    float sof_threshold;
    int max_ldpc_trials, force_ldpc_iters;
    int acm_vcm;       // 1: ACM/VCM receive mode -- the PL framing follows the PLS code of every frame (see S2Rx::vcm_walk)
    int soft_plsc;     // 1: PLHDR demod decodes the PLS code by soft correlation over all 64 bits instead of the reference's 60-bit hard compare
    int pilot_aided;   // 1: the PLL takes its error from the known header / pilot symbols only and coasts over the payload
};
DemodCfg default_cfg(int modcod, int shortframes, int pilots);

// soft ML decoder of the PLS code RM(64,7) (EN 302 307 5.5.2.4): argmax over the 128 scrambled codewords of sum_p soft[p] * (1 - 2 bit_p),
// soft[p] > 0 <=> bit 0; sums in index order, lowest index wins ties.  *ratio = best metric / sum |soft| (1 = noiseless).
int soft_plsc_decode(const float soft[64], float* ratio);

constexpr int VCM_ACQ_WINDOW = 33282;        // acquisition search span = the longest PLFRAME (QPSK normal with pilots)
constexpr float VCM_MIN_RATIO = 0.5f;        // PLS decodes below this correlation ratio count as "no header here"
constexpr int VCM_DUMMY_PLFRAME = 3330;      // dummy PLFRAME (MODCOD 0): 36 unmodulated slots

struct FrameStats { float best_match; int detect_modcod, detect_short, detect_pilots; float fed_err; int ldpc_trials, bch_corr; int bbframe_bytes; };

// Mirror of DVBS2Demod (module_dvbs2_demod.cpp) with every frame LDPC-decoded and BBFRAMEs
// emitted in the call that completes them.
class S2Rx {
public:
    explicit S2Rx(const DemodCfg& cfg);
    int process(int count, const cf* in, uint8_t* out, int out_cap);
    void reset();
    // stage taps for tests (filled by the last process() call)
    std::vector<cf> dbg_symbols;        // 1-sps symbols entering PL sync
    std::vector<cf> dbg_frames;         // aligned raw PL frames
    std::vector<cf> dbg_pll;            // PLL output per frame (header un-rotated + descrambled payload)
    // study of a parallel-in-time form of the payload PLL (tools/pll_tile_study.py): tile size (0 = off) and the histogram of evaluation
    // passes per tile the fixed-point scheme would need (index = passes, last = more); every tile's result is checked against the serial loop
    int study_tile = 0;
    long long study_hist[34] = {};
    long long study_mismatch = 0;
    long long study_steps = 0, study_syms = 0, study_evals = 0;
    // the same question for the other two serial chains of a stream (round 6, tools/g1_tile_study.py): FastAGC in tiles of agc_study_tile samples (gains guessed, |x g| evaluated in
    // parallel, the recurrence g += (1 - a) rate replayed, repeated until no gain changes) and the timing recovery in tiles of gardner_study_tile symbols ((sample offset, polyphase
    // arm) of every on-symbol output guessed, the three interpolants and the error evaluated in parallel, the loop replayed up to the first symbol that lands elsewhere)
    int agc_study_tile = 0, gardner_study_tile = 0;
    long long agc_hist[66] = {}, agc_mismatch = 0, agc_samples = 0;
    long long gd_hist[34] = {}, gd_mismatch = 0, gd_syms = 0, gd_evals = 0, gd_replay_steps = 0, gd_tiles = 0;
    void agc_tile_study(int n, const cf* in);
    void gardner_tile_study(int n, const cf* in);   // the same tiles with the replay restarted at the first symbol whose table cell changed: replay steps, symbols, evaluation passes
    std::vector<int8_t> dbg_llr;        // deinterleaved LLRs per frame
    std::vector<FrameStats> dbg_stats;
    float nco_freq() const { return nco_freq_; }
    float agc_gain_now() const { return agc_gain; }
    s2::ModcodParams mp;
    // what a frame is processed with: CCM = the configured MODCOD, ACM/VCM = the MODCOD its PLS code names
    struct FrameCtx { s2::ModcodParams mp; int pls_code; const Constellation* constel; const LdpcCode* ldpc; const BchCode* bch; bool dummy; };
    const FrameCtx* ctx_for_pls(int pls_code);       // nullptr: not a valid PLS code (reserved MODCODs, short 9/10)

private:
    DemodCfg cfg;
    // front end state
    float agc_gain;
    float nco_phase, nco_freq_;
    std::vector<float> rrc; std::vector<cf> rrc_hist;
    std::vector<float> bank; std::vector<cf> g_hist; int g_offset; int g_spsctr; Pcl g_pcl;
    bool cr_samp;
    // PL sync state (dvbs2_pl_sync.cpp)
    std::vector<cf> in_buffer, corr_buffer; int in_ptr, in_lim, in_state, best_pos; float last_bm = 0;
    // PLL / PLHDR
    Pcl pll_pcl, hdr_pcl;
    Constellation constel;
    int pls_code;
    std::vector<cf> work1, work2;
    const LdpcCode* ldpc; const BchCode* bch;
    FrameCtx ccm;                                     // the configured MODCOD
    std::vector<std::unique_ptr<FrameCtx>> vcm_ctx;  // by PLS code (lazily built)
    std::vector<std::unique_ptr<Constellation>> vcm_constel;
    // ACM/VCM framing state: symbols not yet consumed, locked?
    std::vector<cf> vfifo; bool vcm_synced = false;

    int plsync_internal(std::vector<cf>& out, float* best_match);
    void vcm_walk(uint8_t* out, int out_cap, int* outcnt);
    void process_frame(const cf* frame, const FrameCtx& fc, float best_match, uint8_t* out, int out_cap, int* outcnt);
public:
    // individual stages, exposed for stage-level parity tests
    void agc(int n, const cf* in, cf* out);
    void nco(int n, const cf* in, cf* out);
    int gardner(int n, const cf* in, cf* out);
    void rrc_filter(int n, const cf* in, cf* out);
    float coarse_fed(const cf* frame) const { return coarse_fed(frame, ccm); }
    void pll(const cf* frame, cf* out) { pll(frame, out, ccm); }
    void pll_tile_study(const cf* in, const FrameCtx& fc, Pcl pcl_at_payload_start);
    void plhdr(const cf* frame, cf* out, int* modcod, int* sh, int* pil) { plhdr(frame, out, modcod, sh, pil, ccm.mp.plframe); }
    void to_soft(const cf* pllout, int8_t* llr) const { to_soft(pllout, llr, ccm); }
    float coarse_fed(const cf* frame, const FrameCtx& fc) const;
    void pll(const cf* frame, cf* out, const FrameCtx& fc);
    void plhdr(const cf* frame, cf* out, int* modcod, int* sh, int* pil, int plframe);
    void to_soft(const cf* pllout, int8_t* llr, const FrameCtx& fc) const;
    // PLS decode at an assumed SOF (ACM/VCM framing): phase reference from the 26 SOF symbols, soft decode of the 64 PLSC symbols.
    // Returns the PLS code index 0..127; *ratio as soft_plsc_decode, *sofq = |sum x conj(sof)| / sum |x| over the SOF.
    static int pls_at(const cf* hdr90, float* ratio, float* sofq);
};

// ---- synthetic transmitter (SURVEY 8d "Synthetic inputs")
struct TxCfg {
    int modcod, shortframes, pilots;
    int nframes;
    uint64_t seed;
    double esn0_db;          // AWGN; >= 100 -> noiseless
    double cfo;              // carrier offset, rad/sample (2 sps)
    double timing;           // fractional timing offset, samples
    double phase0;           // initial carrier phase
    int lead_symbols;        // random QPSK symbols before the first frame (sync acquisition run-in)
    int circular;            // != 0: pulse shaping wraps around, so the block can be repeated as a seamless stream
    int nsamples;            // 0: exactly 2 samples per symbol; else the block is resampled to this many samples (sampling-clock error
                             // of 2*symbols/nsamples - 1, e.g. -10 ppm: nsamples = 2*symbols*(1 + 1e-5)); with `circular` still seamless
    int vcm_n;               // > 0: ACM/VCM stream -- frame f carries PLS code vcm_pls[f % vcm_n] (modcod << 2 | short << 1 | pilots; modcod 0 =
    int vcm_pls[64];         // dummy PLFRAME) instead of (modcod, shortframes, pilots); bbframes_out then holds kbch/8 bytes per non-dummy frame
};
// returns 2-sps IQ; bbframes_out gets nframes x kbch/8 bytes (what the receiver must output)
std::vector<cf> s2_transmit(const TxCfg& t, std::vector<uint8_t>* bbframes_out, std::vector<cf>* symbols_out = nullptr);

}  // namespace orc
