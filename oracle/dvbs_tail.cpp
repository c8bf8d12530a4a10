// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// CPU restatement of the DVB-S tail of DVBSDemod::process (reference src/demod/dvbs/module_dvbs_demod.cpp:82-99):
//   DVBS_TS_Deframer::work          dvbs/dvbs_ts_deframer.cpp:37-92
//   DVBSReedSolomon::decode         dvbs/dvbs_reedsolomon.h:26-47 over correct_reed_solomon_decode
//                                   (common/correct/reed-solomon/decode.c:299-380, BM :31-121, Chien :125-148, Forney :166-205,
//                                    locations :207-231; field.h; polynomial.c:17-30,74-87,113-171)
//   DVBSScrambling::descramble      dvbs/dvbs_scrambling.h:28-42
// PINNED: all three compile from the reference as they are and are compared in tests/test_oracle_dvbs_tail.py (oracle/_ref).
#include "dvbs_tail.h"
#include <cstring>

namespace orc {

static inline uint8_t pack_8(const uint8_t* bits) {
    return (uint8_t)(bits[0] << 7 | bits[1] << 6 | bits[2] << 5 | bits[3] << 4 | bits[4] << 3 | bits[5] << 2 | bits[6] << 1 | bits[7] << 0);
}
static inline int compare_8(uint8_t a, uint8_t b) { return __builtin_popcount((unsigned)(a ^ b)); }

int TsDeframer::work(const uint8_t* input, int size, uint8_t* output) {
    int frame_count = 0;
    for (int ibit = 0; ibit < size; ibit++) {
        memmove(&shifter[0], &shifter[1], TS_SIZE - 1);
        shifter[TS_SIZE - 1] = input[ibit];
        int t_nor = 0, t_inv = 0;
        for (int i = 0; i < 8; i++) {
            uint8_t sb = pack_8(&shifter[204 * 8 * i]);
            t_nor += compare_8(i == 0 ? 0xB8 : 0x47, sb);
            t_inv += compare_8(i == 0 ? 0x47 : 0xB8, sb);
        }
        if (t_nor <= 8) {
            uint8_t* o = &output[(size_t)frame_count * 204 * 8];
            memset(o, 0, 204 * 8);
            for (int i = 0; i < 204 * 8 * 8; i++) o[i / 8] = (uint8_t)(o[i / 8] << 1 | shifter[i]);
            frame_count++;
            errors_nor = t_nor; errors_inv = 0;
        }
        if (t_inv <= 8) {
            uint8_t* o = &output[(size_t)frame_count * 204 * 8];
            memset(o, 0, 204 * 8);
            for (int i = 0; i < 204 * 8 * 8; i++) o[i / 8] = (uint8_t)(o[i / 8] << 1 | !shifter[i]);
            frame_count++;
            errors_nor = 0; errors_inv = t_inv;
        }
    }
    return frame_count;
}

// ------------------------------------------------------------------ GF(256)
Gf256::Gf256() {
    unsigned element = 1;
    exp[0] = 1;
    memset(log, 0, sizeof(log));
    for (unsigned i = 1; i < 512; i++) {
        element = element * 2;
        element = (element > 255) ? (element ^ 0x11d) : element;
        exp[i] = (uint8_t)element;
        if (i < 256) log[element] = (uint8_t)i;      // note: log[1] ends up 255
    }
}
const Gf256& gf256() { static Gf256 g; return g; }

static inline uint8_t fmul(const Gf256& F, uint8_t l, uint8_t r) { return (l == 0 || r == 0) ? 0 : F.exp[(unsigned)F.log[l] + F.log[r]]; }
static inline uint8_t fdiv(const Gf256& F, uint8_t l, uint8_t r) { return (l == 0 || r == 0) ? 0 : F.exp[255u + F.log[l] - F.log[r]]; }
static inline uint8_t fpow(const Gf256& F, uint8_t e, int p) {
    int m = ((int)F.log[e] * p) % 255;
    if (m < 0) m += 255;
    return F.exp[m];
}
static inline uint8_t mul_log(unsigned l, unsigned r) { unsigned res = l + r; return (uint8_t)(res > 255 ? res - 255 : res); }
// polynomial_build_exp_lut: logs of val^0 .. val^order
static void build_exp_lut(const Gf256& F, uint8_t val, int order, uint8_t* out) {
    uint8_t ve = F.log[1], vl = F.log[val];
    for (int i = 0; i <= order; i++) {
        if (val == 0) out[i] = 0;
        else { out[i] = ve; ve = mul_log(ve, vl); }
    }
}
static uint8_t eval_lut(const Gf256& F, const uint8_t* coeff, int order, const uint8_t* val_exp) {
    if (val_exp[0] == 0) return coeff[0];
    uint8_t res = 0;
    for (int i = 0; i <= order; i++)
        if (coeff[i] != 0) res ^= F.exp[(unsigned)F.log[coeff[i]] + val_exp[i]];
    return res;
}
static uint8_t eval_log_lut(const Gf256& F, const uint8_t* coeff_log, int order, const uint8_t* val_exp) {
    if (val_exp[0] == 0) return coeff_log[0] == 0 ? 0 : F.exp[coeff_log[0]];
    uint8_t res = 0;
    for (int i = 0; i <= order; i++)
        if (coeff_log[i] != 0) res ^= F.exp[(unsigned)coeff_log[i] + val_exp[i]];
    return res;
}

int rs255_decode(const uint8_t* encoded, uint8_t* msg) {
    const Gf256& F = gf256();
    const int MD = 16, N = 255;
    uint8_t recv[256];
    for (int i = 0; i < N; i++) recv[i] = encoded[N - (i + 1)];
    // syndromes at the generator roots alpha^0 .. alpha^15
    uint8_t syn[MD];
    bool all_zero = true;
    for (int i = 0; i < MD; i++) {
        uint8_t lut[256];
        build_exp_lut(F, F.exp[i % 255], N - 1, lut);
        syn[i] = eval_lut(F, recv, N - 1, lut);
        if (syn[i]) all_zero = false;
    }
    if (all_zero) {
        for (int i = 0; i < N - MD; i++) msg[i] = recv[N - (i + 1)];
        return N - MD;
    }
    // Berlekamp-Massey exactly as reed_solomon_find_error_locator (num_erasures = 0)
    uint8_t loc[64], last[64];
    memset(loc, 0, sizeof(loc)); memset(last, 0, sizeof(last));
    loc[0] = 1; last[0] = 1;
    unsigned loc_order = 0, last_order = 0, numerrors = 0, delay = 1;
    uint8_t last_disc = 1;
    for (unsigned i = 0; i < (unsigned)MD; i++) {
        uint8_t disc = syn[i];
        for (unsigned j = 1; j <= numerrors; j++) disc ^= fmul(F, loc[j], syn[i - j]);
        if (!disc) { delay++; continue; }
        if (2 * numerrors <= i) {
            for (int j = (int)last_order; j >= 0; j--) last[j + delay] = fdiv(F, fmul(F, last[j], disc), last_disc);
            for (int j = (int)delay - 1; j >= 0; j--) last[j] = 0;
            for (unsigned j = 0; j <= last_order + delay; j++) { uint8_t t = loc[j]; loc[j] ^= last[j]; last[j] = t; }
            unsigned t = loc_order;
            loc_order = last_order + delay;
            last_order = t;
            numerrors = i + 1 - numerrors;
            last_disc = disc;
            delay = 1;
            continue;
        }
        for (int j = (int)last_order; j >= 0; j--) loc[j + delay] ^= fdiv(F, fmul(F, last[j], disc), last_disc);
        loc_order = (last_order + delay > loc_order) ? last_order + delay : loc_order;
        delay++;
    }
    const int order = (int)loc_order;
    uint8_t loc_log[64];
    for (int i = 0; i <= order; i++) loc_log[i] = F.log[loc[i]];
    // Chien search over every field element
    uint8_t roots[64];
    int nroots = 0;
    for (int e = 0; e < 256; e++) {
        uint8_t lut[MD];
        build_exp_lut(F, (uint8_t)e, MD - 1, lut);
        if (!eval_log_lut(F, loc_log, order, lut)) { if (nroots < 64) roots[nroots] = (uint8_t)e; nroots++; }
    }
    if (nroots != order) return -1;
    // locations
    int locs[64];
    for (int i = 0; i < order; i++) {
        uint8_t l = fdiv(F, 1, roots[i]);
        locs[i] = 0;
        for (int j = 0; j < 256; j++)
            if (fpow(F, (uint8_t)j, 1) == l) { locs[i] = F.log[j]; break; }
    }
    // Forney: evaluator = locator * S(x) mod x^16, derivative of the locator
    uint8_t ev[MD];
    memset(ev, 0, sizeof(ev));
    for (int i = 0; i <= order; i++) {
        if (i > MD - 1) continue;
        int jl = (MD - 1 > MD - 1 - i) ? MD - 1 - i : MD - 1;
        for (int j = 0; j <= jl; j++) ev[i + j] ^= fmul(F, loc[i], syn[j]);
    }
    uint8_t der[64];
    memset(der, 0, sizeof(der));
    for (int i = 0; i <= order - 1; i++) der[i] = ((i + 1) % 2) ? loc[i + 1] : 0;
    for (int i = 0; i < order; i++) {
        if (roots[i] == 0) continue;
        uint8_t lut[MD];
        build_exp_lut(F, roots[i], MD - 1, lut);
        uint8_t val = fmul(F, fpow(F, roots[i], -1), fdiv(F, eval_lut(F, ev, MD - 1, lut), eval_lut(F, der, order - 1, lut)));
        if (locs[i] < N) recv[locs[i]] ^= val;
    }
    for (int i = 0; i < N - MD; i++) msg[i] = recv[N - (i + 1)];
    return N - MD;
}

DvbsRs::DvbsRs() { memset(obuffer, 0, sizeof(obuffer)); }
int DvbsRs::decode(uint8_t* data) {
    uint8_t buffer[255];
    memset(buffer, 0, 51);
    memcpy(&buffer[51], &data[0], 188);
    memcpy(&buffer[239], &data[188], 16);
    int err = rs255_decode(buffer, obuffer);      // on failure obuffer keeps the previous packet's message (reference behaviour)
    if (err == 1) return -1;                      // (never true: the library returns 239 or -1; kept as written, dvbs_reedsolomon.h:33)
    err = 0;
    for (int i = 51; i < 239; i++)
        if ((buffer[i] ^ obuffer[i]) != 0) err++;
    memcpy(data, &obuffer[51], 188);
    return err;
}

int DvbsDescrambler::prbs(int clocks) {
    int res = 0;
    for (int i = 0; i < clocks; i++) {
        int feedback = ((reg >> (14 - 1)) ^ (reg >> (15 - 1))) & 0x1;
        reg = ((reg << 1) | feedback) & 0x7fff;
        res = (res << 1) | feedback;
    }
    return res;
}
void DvbsDescrambler::descramble(uint8_t* frm) {
    for (int pkt = 0; pkt < 8; pkt++) {
        int outc = 0;
        if (frm[pkt * 204 + outc] == 0xB8) reg = 0xa9;
        else prbs(8);
        frm[pkt * 204 + outc++] = 0x47;
        for (int k = 1; k < 188; k++) frm[pkt * 204 + outc++] ^= (uint8_t)prbs(8);
    }
}

}  // namespace orc
