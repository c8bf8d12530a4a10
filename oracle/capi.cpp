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
