// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// Each function cites the reference file:line it restates.  Float stages that depend on SDR++ core /
// VOLK (absent from /root/reference) are marked "parity unpinned" where they are defined.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include "../sdrpp-dvbs-demodulator_amd/csrc/s2_params.h"
#include "../sdrpp-dvbs-demodulator_amd/csrc/ldpc_qc_tables.inc"

namespace orc {

// ---------------------------------------------------------------- LDPC (ldpc.cpp)
struct LdpcCode {
    int N, K, M, R, q, CNL, LT;
    std::vector<uint16_t> pos;  // R x CNL, layer-major rows (row M*i+j = check q*j+i), ascending bit order
    std::vector<uint8_t> cnc;   // data links per ORIGINAL check index (only [0,q) is read)
    explicit LdpcCode(int code_index);
};
int ldpc_decode(const LdpcCode& C, int8_t* frame, int max_trials, int force);
void ldpc_encode(const LdpcCode& C, uint8_t* bits);

// ---------------------------------------------------------------- BCH (bch.cpp)
struct BchCode {
    int m, t, NR, N;           // field width, correctable errors, roots = 2t, field order - 1
    uint32_t poly;
    int nbch, kbch, K_full;    // K_full = N - m*t (unshortened message length: 65343 / 65375 / 65407 / 16215)
    std::vector<uint16_t> LOG, EXP, IMAP;
    std::vector<uint8_t> gen;  // generator polynomial coefficients, gen[0] = x^0 ... gen[m*t] = 1
    BchCode(int m, int t, int nbch, int kbch);
    uint16_t imul(uint16_t a, uint16_t b) const;  // Index * Index
    uint16_t idiv(uint16_t a, uint16_t b) const;  // Index / Index
    uint16_t vmul(uint16_t a, uint16_t b) const;  // Value * Value
    uint16_t vdiv(uint16_t a, uint16_t b) const;  // Value / Value
};
int bch_syndromes(const BchCode& C, const uint8_t* frame, uint16_t* syn);
int bch_decode(const BchCode& C, uint8_t* frame);
void bch_encode(const BchCode& C, uint8_t* frame);

// ---------------------------------------------------------------- BB scrambler / packing (bbframe.cpp)
void bb_prbs(uint8_t* seq, int nbytes);
void bb_descramble(uint8_t* frame, int nbytes);
void hard_pack(const int8_t* post, int nbits, uint8_t* out);
uint8_t bbheader_crc8(const uint8_t* hdr9);
void make_bbframe(uint8_t* frame, int kbch, uint64_t seed);

}  // namespace orc
