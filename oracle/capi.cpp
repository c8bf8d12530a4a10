// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// C entry points so tests/ and bench.py's cpu_baseline leg can drive the restatement through ctypes.
#include "oracle.h"
#include <map>
#include <memory>
#include <mutex>
#include <cstring>

using namespace orc;

static std::mutex g_mtx;
static const LdpcCode& ldpc_code(int code_index) {
    static std::map<int, std::unique_ptr<LdpcCode>> cache;
    std::lock_guard<std::mutex> l(g_mtx);
    auto& p = cache[code_index];
    if (!p) p.reset(new LdpcCode(code_index));
    return *p;
}
static const BchCode& bch_code(int rate, int shortframe) {
    static std::map<int, std::unique_ptr<BchCode>> cache;
    std::lock_guard<std::mutex> l(g_mtx);
    auto& p = cache[rate * 2 + shortframe];
    if (!p) {
        s2::FecParams f;
        s2::fec_params(rate, shortframe, &f);
        p.reset(new BchCode(f.bch_m, f.bch_t, f.K, f.kbch));
    }
    return *p;
}

extern "C" {

int orc_fec_params(int rate, int shortframe, int* out6) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -1;
    out6[0] = f.code_index; out6[1] = f.N; out6[2] = f.K; out6[3] = f.kbch; out6[4] = f.bch_m; out6[5] = f.bch_t;
    return 0;
}
int orc_modcod_params(int modcod, int shortframe, int pilots, int* out8, float* g) {
    s2::ModcodParams p;
    if (!s2::modcod_params(modcod, shortframe, pilots, &p)) return -1;
    out8[0] = p.constel; out8[1] = p.bits; out8[2] = p.rate; out8[3] = p.slots; out8[4] = p.pilot_blocks;
    out8[5] = p.plframe; out8[6] = p.fec.N; out8[7] = p.fec.kbch;
    g[0] = p.g1; g[1] = p.g2;
    return 0;
}
int orc_ldpc_edges(int rate, int shortframe) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -1;
    return QC_CODES[f.code_index].edges;
}

// the reference-order parity-check structure: for layer-major row rho = 360*i + j the ascending list of
// information bits (layered_decoder.hh:99-119).  out: R x CNL uint16, cnt: R counts.  Returns CNL.
int orc_ldpc_rows(int rate, int shortframe, uint16_t* out, uint8_t* cnt) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -2;
    const LdpcCode& C = ldpc_code(f.code_index);
    if (out) memcpy(out, C.pos.data(), C.pos.size() * sizeof(uint16_t));
    if (cnt)
        for (int i = 0; i < C.q; ++i)
            for (int j = 0; j < 360; ++j) cnt[360 * i + j] = C.cnc[i];
    return C.CNL;
}

int orc_ldpc_decode(int rate, int shortframe, int8_t* frame, int max_trials, int force) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -2;
    return ldpc_decode(ldpc_code(f.code_index), frame, max_trials, force);
}
// bits: N bytes of 0/1 with [0,K) filled
int orc_ldpc_encode(int rate, int shortframe, uint8_t* bits) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -2;
    ldpc_encode(ldpc_code(f.code_index), bits);
    return 0;
}
int orc_bch_decode(int rate, int shortframe, uint8_t* frame) { return bch_decode(bch_code(rate, shortframe), frame); }
int orc_bch_syndromes(int rate, int shortframe, const uint8_t* frame, uint16_t* syn) { return bch_syndromes(bch_code(rate, shortframe), frame, syn); }
void orc_bch_encode(int rate, int shortframe, uint8_t* frame) { bch_encode(bch_code(rate, shortframe), frame); }
void orc_bb_prbs(uint8_t* seq, int nbytes) { bb_prbs(seq, nbytes); }
void orc_bb_descramble(uint8_t* frame, int nbytes) { bb_descramble(frame, nbytes); }
void orc_hard_pack(const int8_t* post, int nbits, uint8_t* out) { hard_pack(post, nbits, out); }
void orc_make_bbframe(uint8_t* frame, int kbch, uint64_t seed) { make_bbframe(frame, kbch, seed); }

// Full transmit-side FEC for one frame: BBFRAME(seed) -> BB scramble -> BCH -> LDPC.
// bbframe_out: kbch/8 bytes (unscrambled, what the receiver must output); code_bits: N bytes 0/1.
int orc_fec_encode_frame(int rate, int shortframe, uint64_t seed, uint8_t* bbframe_out, uint8_t* code_bits) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -2;
    std::vector<uint8_t> fr(f.K / 8, 0);
    make_bbframe(fr.data(), f.kbch, seed);
    memcpy(bbframe_out, fr.data(), f.kbch / 8);
    bb_descramble(fr.data(), f.kbch / 8);  // XOR is its own inverse
    bch_encode(bch_code(rate, shortframe), fr.data());
    for (int i = 0; i < f.K; ++i) code_bits[i] = (fr[i / 8] >> (7 - i % 8)) & 1;
    ldpc_encode(ldpc_code(f.code_index), code_bits);
    return 0;
}

// Receive-side FEC for one frame, stage by stage as the plugin wires it
// (module_dvbs2_demod.cpp:349-366): LDPC -> hard pack -> BCH -> descramble -> first kbch/8 bytes.
// llr is modified in place (posteriors).  Returns LDPC trials; *bch_corr gets the BCH result.
int orc_fec_decode_frame(int rate, int shortframe, int8_t* llr, int max_trials, int force, uint8_t* bbframe_out, int* bch_corr) {
    s2::FecParams f;
    if (!s2::fec_params(rate, shortframe, &f)) return -2;
    int trials = ldpc_decode(ldpc_code(f.code_index), llr, max_trials, force);
    std::vector<uint8_t> fr(f.K / 8);
    hard_pack(llr, f.K, fr.data());
    int c = bch_decode(bch_code(rate, shortframe), fr.data());
    if (bch_corr) *bch_corr = c;
    bb_descramble(fr.data(), f.kbch / 8);
    memcpy(bbframe_out, fr.data(), f.kbch / 8);
    return trials;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- S2 chain (s2chain.cpp)
#include "s2chain.h"

extern "C" {

void orc_default_cfg(int modcod, int shortframes, int pilots, DemodCfg* out) { *out = default_cfg(modcod, shortframes, pilots); }

void* orc_s2rx_create(const DemodCfg* cfg) {
    try { return new S2Rx(*cfg); } catch (...) { return nullptr; }
}
void orc_s2rx_destroy(void* h) { delete (S2Rx*)h; }
void orc_s2rx_reset(void* h) { ((S2Rx*)h)->reset(); }
int orc_s2rx_process(void* h, int count, const float* iq, uint8_t* out, int out_cap) {
    return ((S2Rx*)h)->process(count, (const cf*)iq, out, out_cap);
}
// debug taps of the last process() call: which = 0 symbols, 1 aligned frames, 2 PLL output (complex float);
// 3 = LLRs (int8); 4 = stats (FrameStats).  Returns the element count; copies when dst != null.
int orc_s2rx_tap(void* h, int which, void* dst) {
    S2Rx* r = (S2Rx*)h;
    switch (which) {
        case 0: if (dst) memcpy(dst, r->dbg_symbols.data(), r->dbg_symbols.size() * sizeof(cf)); return (int)r->dbg_symbols.size();
        case 1: if (dst) memcpy(dst, r->dbg_frames.data(), r->dbg_frames.size() * sizeof(cf)); return (int)r->dbg_frames.size();
        case 2: if (dst) memcpy(dst, r->dbg_pll.data(), r->dbg_pll.size() * sizeof(cf)); return (int)r->dbg_pll.size();
        case 3: if (dst) memcpy(dst, r->dbg_llr.data(), r->dbg_llr.size()); return (int)r->dbg_llr.size();
        case 4: if (dst) memcpy(dst, r->dbg_stats.data(), r->dbg_stats.size() * sizeof(FrameStats)); return (int)r->dbg_stats.size();
    }
    return -1;
}
float orc_s2rx_nco_freq(void* h) { return ((S2Rx*)h)->nco_freq(); }
// tools/pll_tile_study.py: passes-per-tile histogram of the parallel-in-time form of the payload PLL (34 bins) + tiles whose fixed point differed from the serial loop
void orc_s2rx_pll_study(void* h, int tile, long long* hist34, long long* mismatch) {
    S2Rx* r = (S2Rx*)h;
    if (hist34) for (int i = 0; i < 34; ++i) hist34[i] = r->study_hist[i];
    if (mismatch) *mismatch = r->study_mismatch;
    r->study_tile = tile;
}
// tools/g1_tile_study.py: the same question for FastAGC and the timing recovery (S2Rx::agc_tile_study, gardner_tile_study).  set: tile sizes (0 = off); get (NULL to skip): agc66 =
// passes-per-tile histogram, agc2 = {tiles whose fixed point differs from the serial loop, samples}; gd34 = evaluation passes per tile, gd5 = {mismatching tiles, symbols, evaluation
// passes, replay steps, tiles}
void orc_s2rx_g1_study(void* h, int agc_tile, int gardner_tile, long long* agc66, long long* agc2, long long* gd34, long long* gd5) {
    S2Rx* r = (S2Rx*)h;
    if (agc66) for (int i = 0; i < 66; ++i) agc66[i] = r->agc_hist[i];
    if (agc2) { agc2[0] = r->agc_mismatch; agc2[1] = r->agc_samples; }
    if (gd34) for (int i = 0; i < 34; ++i) gd34[i] = r->gd_hist[i];
    if (gd5) { gd5[0] = r->gd_mismatch; gd5[1] = r->gd_syms; gd5[2] = r->gd_evals; gd5[3] = r->gd_replay_steps; gd5[4] = r->gd_tiles; }
    r->agc_study_tile = agc_tile; r->gardner_study_tile = gardner_tile;
    for (long long& x : r->agc_hist) x = 0;
    for (long long& x : r->gd_hist) x = 0;
    r->agc_mismatch = r->agc_samples = r->gd_mismatch = r->gd_syms = r->gd_evals = r->gd_replay_steps = r->gd_tiles = 0;
}
void orc_s2rx_pll_study2(void* h, long long* out3) { S2Rx* r = (S2Rx*)h; out3[0] = r->study_steps; out3[1] = r->study_syms; out3[2] = r->study_evals; }
float orc_s2rx_agc_gain(void* h) { return ((S2Rx*)h)->agc_gain_now(); }   // (tools/sensitivity.py: level at the AGC output = gain x input rms)

// stage-level entry points on a receiver object (state carried inside it)
void orc_s2rx_agc(void* h, int n, const float* in, float* out) { ((S2Rx*)h)->agc(n, (const cf*)in, (cf*)out); }
void orc_s2rx_nco(void* h, int n, const float* in, float* out) { ((S2Rx*)h)->nco(n, (const cf*)in, (cf*)out); }
int orc_s2rx_gardner(void* h, int n, const float* in, float* out) { return ((S2Rx*)h)->gardner(n, (const cf*)in, (cf*)out); }
void orc_s2rx_rrc(void* h, int n, const float* in, float* out) { ((S2Rx*)h)->rrc_filter(n, (const cf*)in, (cf*)out); }
float orc_s2rx_fed(void* h, const float* frame) { return ((S2Rx*)h)->coarse_fed((const cf*)frame); }
void orc_s2rx_pll(void* h, const float* frame, float* out) { ((S2Rx*)h)->pll((const cf*)frame, (cf*)out); }
void orc_s2rx_plhdr(void* h, const float* frame, float* out, int* det3) { ((S2Rx*)h)->plhdr((const cf*)frame, (cf*)out, det3, det3 + 1, det3 + 2); }
void orc_s2rx_to_soft(void* h, const float* pllout, int8_t* llr) { ((S2Rx*)h)->to_soft((const cf*)pllout, llr); }

// transmitter: returns number of complex samples; call with iq == null to get the size
int orc_s2_transmit(const TxCfg* t, float* iq, int iq_cap, uint8_t* bbframes, float* symbols, int sym_cap) {
    std::vector<uint8_t> bb;
    std::vector<cf> syms;
    std::vector<cf> v = s2_transmit(*t, &bb, &syms);
    if (iq && (int)v.size() <= iq_cap) memcpy(iq, v.data(), v.size() * sizeof(cf));
    if (bbframes && !bb.empty()) memcpy(bbframes, bb.data(), bb.size());
    if (symbols && (int)syms.size() <= sym_cap) memcpy(symbols, syms.data(), syms.size() * sizeof(cf));
    return (int)v.size();
}

// tables for fixtures / GPU table parity
int orc_constellation_lut(int constel, float g1, float g2, int8_t* bits_out, float* err_out) {
    Constellation C(constel, g1, g2);
    if (C.bits == 5) return 0;
    if (bits_out) memcpy(bits_out, C.lut_bits.data(), C.lut_bits.size());
    if (err_out) memcpy(err_out, C.lut_err.data(), C.lut_err.size() * sizeof(float));
    return C.bits;
}
void orc_constellation_soft_calc(int constel, float g1, float g2, int n, const float* samples, int8_t* bits_out, float* err_out) {
    Constellation C(constel, g1, g2);
    for (int i = 0; i < n; ++i) C.soft_calc(cf{samples[2 * i], samples[2 * i + 1]}, bits_out + (size_t)i * C.bits, err_out + i);
}
// host evaluation of include/dvbs2gpu_math.h (same func codes as dvbs2gpu_math_eval)
void orc_math_eval(int func, int n, const float* a, const float* b, float* o0, float* o1) {
    for (int i = 0; i < n; ++i) {
        switch (func) {
            case 0: dvbs2m::sincosf_det(a[i], &o0[i], &o1[i]); break;
            case 1: o0[i] = dvbs2m::atan2f_det(a[i], b[i]); break;
            case 2: o0[i] = dvbs2m::expf_det(a[i]); break;
            case 3: o0[i] = dvbs2m::logf_det(a[i]); break;
            case 5: o0[i] = (float)(Constellation::lut_index(a[i]) * 256 + Constellation::lut_index(b[i])); break;   // constellation.cpp:295-301, in double
            default: o0[i] = (float)dvbs2m::llr_clamp_det(a[i]); break;
        }
    }
}
void orc_constellation_points(int constel, float g1, float g2, float* pts_out) {
    Constellation C(constel, g1, g2);
    for (int i = 0; i < C.states; ++i) { cf p = C.mod(i); pts_out[2 * i] = p.re; pts_out[2 * i + 1] = p.im; }
}
void orc_s2_deinterleave(int constel, int rate, int shortframe, const int8_t* in, int8_t* out) { s2_deinterleave(constel, rate, shortframe, in, out); }
void orc_pl_tables(float* sof52, float* plsc_128x64x2, uint64_t* codes128, uint8_t* rn131072) {
    const PlTables& T = pl_tables();
    if (sof52) memcpy(sof52, T.sof, sizeof(T.sof));
    if (plsc_128x64x2) memcpy(plsc_128x64x2, T.plsc_sym, sizeof(T.plsc_sym));
    if (codes128) memcpy(codes128, T.plsc_code, sizeof(T.plsc_code));
    if (rn131072) memcpy(rn131072, T.Rn.data(), 131072);
}
void orc_rrc_taps(int count, double beta, double symrate, double samprate, float* out) {
    std::vector<float> t = rrc_taps(count, beta, symrate, samprate);
    memcpy(out, t.data(), t.size() * sizeof(float));
}
void orc_gardner_bank(float* out1024) {
    std::vector<float> b = gardner_bank(128, 8);
    memcpy(out1024, b.data(), b.size() * sizeof(float));
}

}  // extern "C"

// ---------------------------------------------------------------------------------- DVB-S (dvbs.cpp)
#include "dvbs.h"
extern "C" {
void* orc_dvbs_slicer_create() { return new DvbsSlicer(); }
void orc_dvbs_slicer_destroy(void* h) { delete (DvbsSlicer*)h; }
int orc_dvbs_slicer_process(void* h, int count, const float* iq, int8_t* out) { return ((DvbsSlicer*)h)->process(count, iq, out); }
void orc_rotate_soft(int8_t* soft, int size, int phase, int iqswap) { rotate_soft(soft, size, phase, iqswap != 0); }
void orc_signed_to_unsigned(const int8_t* in, uint8_t* out, int n) { signed_soft_to_unsigned(in, out, n); }
int orc_depuncture_34(const uint8_t* in, uint8_t* out, int size, int shift) { return depuncture_34(in, out, size, shift != 0); }
int orc_depuncture_78(const uint8_t* in, uint8_t* out, int size, int shift) { return depuncture_78(in, out, size, shift); }
void* orc_depunc_create(int period) { return new DepuncCont(period); }
void orc_depunc_destroy(void* h) { delete (DepuncCont*)h; }
int orc_depunc_static(void* h, const uint8_t* in, uint8_t* out, int size, int shift) { return ((DepuncCont*)h)->depunc_static(in, out, size, shift); }
void orc_depunc_set_shift(void* h, int shift) { ((DepuncCont*)h)->set_shift(shift); }
int orc_depunc_cont(void* h, const uint8_t* in, uint8_t* out, int size) { return ((DepuncCont*)h)->depunc_cont(in, out, size); }
void* orc_ccdec_create(int frame_size) { return new CcDecoder(frame_size); }
void orc_ccdec_destroy(void* h) { delete (CcDecoder*)h; }
// in must hold 2*(frame_size+6) bytes; returns the chained start state for the next block
int orc_ccdec_work(void* h, const uint8_t* in, uint8_t* out) { ((CcDecoder*)h)->work(in, out); return ((CcDecoder*)h)->start_state_chaining; }
void* orc_ccenc_create(int frame_size) { return new CcEncoder(frame_size); }
void orc_ccenc_destroy(void* h) { delete (CcEncoder*)h; }
void orc_ccenc_work(void* h, const uint8_t* in, uint8_t* out) { ((CcEncoder*)h)->work(in, out); }
void* orc_vitdvbs_create(float thr, int max_outsync, int bufsize) { return new ViterbiDvbs(thr, max_outsync, bufsize); }
void orc_vitdvbs_destroy(void* h) { delete (ViterbiDvbs*)h; }
int orc_vitdvbs_work(void* h, int8_t* input, int size, uint8_t* output, float* ber_state4) {
    ViterbiDvbs* v = (ViterbiDvbs*)h;
    int n = v->work(input, size, output);
    if (ber_state4) { ber_state4[0] = v->ber; ber_state4[1] = (float)v->state; ber_state4[2] = (float)v->rate; ber_state4[3] = (float)(v->d_phase * 16 + v->d_shift); }
    return n;
}
void* orc_forney_create() { return new ForneyDeint(); }
void orc_forney_destroy(void* h) { delete (ForneyDeint*)h; }
void orc_forney_deinterleave(void* h, const uint8_t* in, uint8_t* out) { ((ForneyDeint*)h)->deinterleave(in, out); }
}  // extern "C"

// ---------------------------------------------------------------------------------- DVB-S front end (dvbs_fe.cpp)
#include "dvbs_fe.h"
extern "C" {
void orc_qpskalt_default_cfg(QpskAltCfg* c) { *c = qpsk_alt_default_cfg(); }
void* orc_qpskalt_create(const QpskAltCfg* c) { return new QpskAlt(*c); }
void orc_qpskalt_destroy(void* h) { delete (QpskAlt*)h; }
int orc_qpskalt_process(void* h, int n, const float* iq, float* out) { return ((QpskAlt*)h)->process(n, (const cf*)iq, (cf*)out); }
// stage taps for parity tests: which 0 agc, 1 fll, 2 rrc (n -> n), 3 complex_fd (n -> m), 4 costas (n -> n)
int orc_qpskalt_stage(void* h, int which, int n, const float* in, float* out) {
    QpskAlt* q = (QpskAlt*)h;
    switch (which) {
        case 0: q->agc(n, (const cf*)in, (cf*)out); return n;
        case 1: q->fll(n, (const cf*)in, (cf*)out); return n;
        case 2: q->rrc_filter(n, (const cf*)in, (cf*)out); return n;
        case 3: return q->complex_fd(n, (const cf*)in, (cf*)out);
        default: q->costas(n, (const cf*)in, (cf*)out); return n;
    }
}
void orc_qpskalt_state(void* h, float* s8) {
    QpskAlt* q = (QpskAlt*)h;
    s8[0] = q->agc_gain; s8[1] = q->fll_pcl.phase; s8[2] = q->fll_pcl.freq; s8[3] = q->fd_pcl.phase; s8[4] = q->fd_pcl.freq;
    s8[5] = (float)q->fd_offset; s8[6] = q->costas_pcl.phase; s8[7] = q->costas_pcl.freq;
}
// bits: 2*nsym bytes (0/1); out: 2*nsym complex samples (interleaved floats)
void orc_dvbs_modulate(const uint8_t* bits, int nsym, double esn0_db, double cfo, double timing, double phase0, uint64_t seed, float* out) {
    std::vector<cf> v = dvbs_modulate(bits, nsym, esn0_db, cfo, timing, phase0, seed, 65, 0.35);
    memcpy(out, v.data(), v.size() * sizeof(cf));
}
}  // extern "C"

// ---------------------------------------------------------------------------------- DVB-S tail (dvbs_tail.cpp)
#include "dvbs_tail.h"
extern "C" {
void* orc_tsdef_create() { return new TsDeframer(); }
void orc_tsdef_destroy(void* h) { delete (TsDeframer*)h; }
int orc_tsdef_work(void* h, const uint8_t* bits, int size, uint8_t* out, int* errs2) {
    TsDeframer* d = (TsDeframer*)h;
    int n = d->work(bits, size, out);
    if (errs2) { errs2[0] = d->errors_nor; errs2[1] = d->errors_inv; }
    return n;
}
void* orc_dvbsrs_create() { return new DvbsRs(); }
void orc_dvbsrs_destroy(void* h) { delete (DvbsRs*)h; }
int orc_dvbsrs_decode(void* h, uint8_t* data204) { return ((DvbsRs*)h)->decode(data204); }
int orc_rs255_decode(const uint8_t* enc255, uint8_t* msg239) { return rs255_decode(enc255, msg239); }
void* orc_dvbsdescr_create() { return new DvbsDescrambler(); }
void orc_dvbsdescr_destroy(void* h) { delete (DvbsDescrambler*)h; }
void orc_dvbsdescr_work(void* h, uint8_t* frm1632) { ((DvbsDescrambler*)h)->descramble(frm1632); }
}  // extern "C"

// ---------------------------------------------------------------------------------- BBFRAME -> TS / GSE parser (bbframe_ts.cpp)
#include "bbframe_ts.h"
extern "C" {
void* orc_bbts_create(int kbch_bits) { auto p = new BbTsParser(); p->set_frame_size(kbch_bits); return p; }
void orc_bbts_destroy(void* h) { delete (BbTsParser*)h; }
void orc_bbts_set_frame_size(void* h, int kbch_bits) { ((BbTsParser*)h)->set_frame_size(kbch_bits); }
int orc_bbts_work(void* h, const uint8_t* bb, int cnt, uint8_t* out, int cap) { return ((BbTsParser*)h)->work(bb, cnt, out, cap); }
// {ts_gs, sis_mis, ccm_acm, issyi, npd, ro, isi, upl, dfl, sync, syncd, last_gse_crc_err, last_bb_cnt, last_bb_proc, last_ts_errs, synched, count}
void orc_bbts_get_stats(void* h, int32_t* o17) {
    const BbTsParser* p = (const BbTsParser*)h;
    const BbHeader& q = p->last_header;
    const int v[17] = {q.ts_gs, q.sis_mis, q.ccm_acm, q.issyi, q.npd, q.ro, q.isi, q.upl, q.dfl, q.sync, q.syncd,
                       p->last_gse_crc_err, p->last_bb_cnt, p->last_bb_proc, p->last_ts_errs, p->synched, p->count};
    for (int i = 0; i < 17; ++i) o17[i] = v[i];
}
unsigned orc_bbts_crc8_bits(const uint8_t* in, int nbits) { return bbts_crc8_bits(in, nbits); }
}  // extern "C"
