// DVB-S2 MODCOD / FEC parameter tables (plain data, host side, no device code).
//
// Mirrors what the reference derives in
//   src/demod/dvbs2/codings/modcod_to_cfg.cpp:5-140   (MODCOD -> constellation/rate/slots/gamma)
//   src/demod/dvbs2/codings/bbframe_bch.cpp:39-193     (kbch/nbch/BCH family per rate)
//   src/demod/dvbs2/codings/bbframe_ldpc.cpp:28-107    (LDPC table per rate)
//   src/demod/dvbs2/dvbs2_pl_sync.cpp:14-31            (PLFRAME length incl. pilots)
// The numbers themselves are ETSI EN 302 307-1 tables 5a/5b/12/13.
#pragma once
#include <cstdint>

namespace s2 {

enum Rate { R1_4 = 0, R1_3, R2_5, R1_2, R3_5, R2_3, R3_4, R4_5, R5_6, R8_9, R9_10, RATE_COUNT };
enum Constel { C_QPSK = 0, C_8PSK = 1, C_16APSK = 2, C_32APSK = 3 };

// BCH families (GF width m, correction capability t). Primitive polynomials as in
// bbframe_bch.h:45-47: GF(2^16) 0x1002D, GF(2^14) 0x402B.
struct BchFamily { int m; int t; uint32_t prim_poly; };

struct FecParams {
    int code_index;   // index into QC_CODES (0..20), -1 if the combination does not exist
    int N;            // LDPC codeword bits (64800 / 16200)
    int K;            // LDPC information bits = nbch
    int kbch;         // BCH information bits
    int bch_m, bch_t; // field width, correctable errors
};

struct ModcodParams {
    int modcod;       // 1..28
    int constel;      // Constel
    int bits;         // bits per symbol
    int rate;         // Rate
    int shortframe;   // 0/1
    int pilots;       // 0/1
    int slots;        // 90-symbol payload slots per PLFRAME
    int pilot_blocks; // 36-symbol pilot blocks per PLFRAME
    int plframe;      // total PLFRAME symbols: 90 + 90*slots + 36*pilot_blocks
    float g1, g2;     // APSK ring ratios (0 for PSK)
    FecParams fec;
};

// kbch for normal frames, indexed by Rate (ETSI table 5a), and BCH t.
static const int KBCH_NORMAL[RATE_COUNT] = {16008, 21408, 25728, 32208, 38688, 43040, 48408, 51648, 53840, 57472, 58192};
static const int NBCH_NORMAL[RATE_COUNT] = {16200, 21600, 25920, 32400, 38880, 43200, 48600, 51840, 54000, 57600, 58320};
static const int BCH_T_NORMAL[RATE_COUNT] = {12, 12, 12, 12, 12, 10, 12, 12, 10, 8, 8};
// short frames (ETSI table 5b); 9/10 does not exist for short frames
static const int KBCH_SHORT[RATE_COUNT] = {3072, 5232, 6312, 7032, 9552, 10632, 11712, 12432, 13152, 14232, 0};
static const int NBCH_SHORT[RATE_COUNT] = {3240, 5400, 6480, 7200, 9720, 10800, 11880, 12600, 13320, 14400, 0};

inline bool fec_params(int rate, int shortframe, FecParams* out) {
    if (rate < 0 || rate >= RATE_COUNT) return false;
    if (shortframe && rate == R9_10) return false;
    FecParams f;
    f.code_index = (shortframe ? 11 : 0) + rate;
    f.N = shortframe ? 16200 : 64800;
    f.K = shortframe ? NBCH_SHORT[rate] : NBCH_NORMAL[rate];
    f.kbch = shortframe ? KBCH_SHORT[rate] : KBCH_NORMAL[rate];
    f.bch_m = shortframe ? 14 : 16;
    f.bch_t = shortframe ? 12 : BCH_T_NORMAL[rate];
    *out = f;
    return true;
}

inline int pilot_blocks_for_slots(int slots) {
    // dvbs2_pl_sync.cpp:18-28: one block after every 16 slots, none after the last slot
    return (slots - 1) / 16;
}

// Returns false for MODCOD <= 0 or >= 29 (the reference throws std::runtime_error there,
// modcod_to_cfg.cpp:11,135) and for short 9/10 (no table in the reference, SURVEY scope note).
inline bool modcod_params(int modcod, int shortframe, int pilots, ModcodParams* out) {
    if (modcod <= 0 || modcod >= 29) return false;
    ModcodParams p;
    p.modcod = modcod; p.shortframe = shortframe ? 1 : 0; p.pilots = pilots ? 1 : 0;
    p.g1 = 0.f; p.g2 = 0.f;
    static const int qpsk_rates[11] = {R1_4, R1_3, R2_5, R1_2, R3_5, R2_3, R3_4, R4_5, R5_6, R8_9, R9_10};
    static const int psk8_rates[6] = {R3_5, R2_3, R3_4, R5_6, R8_9, R9_10};
    static const int apsk16_rates[6] = {R2_3, R3_4, R4_5, R5_6, R8_9, R9_10};
    static const float apsk16_g1[6] = {3.15f, 2.85f, 2.75f, 2.70f, 2.60f, 2.57f};
    static const int apsk32_rates[5] = {R3_4, R4_5, R5_6, R8_9, R9_10};
    static const float apsk32_g1[5] = {2.84f, 2.72f, 2.64f, 2.54f, 2.53f};
    static const float apsk32_g2[5] = {5.27f, 4.87f, 4.64f, 4.33f, 4.30f};
    if (modcod < 12) {
        p.constel = C_QPSK; p.bits = 2; p.rate = qpsk_rates[modcod - 1]; p.slots = shortframe ? 90 : 360;
    } else if (modcod < 18) {
        p.constel = C_8PSK; p.bits = 3; p.rate = psk8_rates[modcod - 12]; p.slots = shortframe ? 60 : 240;
    } else if (modcod < 24) {
        p.constel = C_16APSK; p.bits = 4; p.rate = apsk16_rates[modcod - 18]; p.slots = shortframe ? 45 : 180;
        p.g1 = apsk16_g1[modcod - 18];
    } else {
        p.constel = C_32APSK; p.bits = 5; p.rate = apsk32_rates[modcod - 24]; p.slots = shortframe ? 36 : 144;
        p.g1 = apsk32_g1[modcod - 24]; p.g2 = apsk32_g2[modcod - 24];
    }
    if (!fec_params(p.rate, p.shortframe, &p.fec)) return false;
    p.pilot_blocks = p.pilots ? pilot_blocks_for_slots(p.slots) : 0;
    p.plframe = 90 + 90 * p.slots + 36 * p.pilot_blocks;
    *out = p;
    return true;
}

}  // namespace s2
