// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// Restatement of the reference's binary BCH decoder (gr-dvbs2rx / aicodix code vendored by the plugin).
// Pinned bit-exact against the compiled reference (oracle/_ref) incl. the uncorrectable paths.
//
//   GaloisField / Tables / Index / Value ops   dvbs2/codings/bch/galois_field.hh:122-362
//   BoseChaudhuriHocquenghemDecoder            bch/bose_chaudhuri_hocquenghem_decoder.hh:29-144
//   Chien / ArtinSchreier / LocationFinder     bch/reed_solomon_error_correction.hh:34-130
//   Forney                                     reed_solomon_error_correction.hh:133-218
//   BerlekampMassey                            reed_solomon_error_correction.hh:221-277
//   ReedSolomonErrorCorrection::operator()     reed_solomon_error_correction.hh:281-405
//   BBFrameBCH (code selection, decode)        dvbs2/codings/bbframe_bch.cpp:39-193,380-405
//   bit helpers (big-endian bit order)         bch/bitman.cpp:5-27
// The encoder is this repo's own (generator = lcm of the minimal polynomials of alpha^1..alpha^2t,
// computed from the field; ETSI EN 302 307-1 5.3.1) -- it exists only to make test/bench input.
#include "oracle.h"
#include <cstring>

namespace orc {

static inline int get_be_bit(const uint8_t* b, int pos) { return (b[pos / 8] >> (7 - pos % 8)) & 1; }
static inline void xor_be_bit(uint8_t* b, int pos, int v) { b[pos / 8] ^= (uint8_t)(v << (7 - pos % 8)); }

// Index arithmetic is done in the table's uint16 element type, wrap-around included
// (galois_field.hh:237-243, 262-268): log(0) is stored as N, exp(N) as 0.
uint16_t BchCode::imul(uint16_t a, uint16_t b) const {
    uint16_t tmp = (uint16_t)(a + b);
    return (N - (int)a <= (int)b) ? (uint16_t)(tmp - N) : tmp;
}
uint16_t BchCode::idiv(uint16_t a, uint16_t b) const {
    uint16_t tmp = (uint16_t)(a - b);
    return (a < b) ? (uint16_t)(tmp + N) : tmp;
}
uint16_t BchCode::vmul(uint16_t a, uint16_t b) const { return (!a || !b) ? 0 : EXP[imul(LOG[a], LOG[b])]; }
uint16_t BchCode::vdiv(uint16_t a, uint16_t b) const { return !a ? 0 : EXP[idiv(LOG[a], LOG[b])]; }

BchCode::BchCode(int m_, int t_, int nbch_, int kbch_) : m(m_), t(t_), NR(2 * t_), N((1 << m_) - 1), nbch(nbch_), kbch(kbch_) {
    poly = (m == 16) ? 0x1002Du : 0x402Bu;  // bbframe_bch.h:45-47
    K_full = N - m * t;
    const int Q = 1 << m;
    LOG.assign(Q, 0); EXP.assign(Q, 0);
    EXP[N] = 0; LOG[0] = (uint16_t)N;  // galois_field.hh:158
    uint32_t a = 1;
    for (int i = 0; i < N; ++i) {
        EXP[i] = (uint16_t)a; LOG[a] = (uint16_t)i;
        a = (a & (uint32_t)(Q >> 1)) ? ((a << 1) ^ poly) : (a << 1);
        a &= (uint32_t)(Q - 1) | (uint32_t)Q;  // poly carries bit m, xor clears it
        a &= (uint32_t)(Q - 1);
    }
    // ArtinSchreier map (reed_solomon_error_correction.hh:67-96): imap[x*x+x] = x for even x
    IMAP.assign(Q, 0);
    for (int i = 2; i < N; i += 2) {
        uint16_t x = (uint16_t)i;
        uint16_t xxx = (uint16_t)(vmul(x, x) ^ x);
        if (xxx == (uint16_t)N) continue;
        IMAP[xxx] = x;
    }
    // generator polynomial over GF(2): product of distinct minimal polynomials of alpha^1..alpha^(2t)
    std::vector<uint8_t> g(1, 1);
    std::vector<char> done(N + 1, 0);
    for (int r = 1; r <= 2 * t; ++r) {
        if (done[r]) continue;
        // conjugacy class of r
        std::vector<int> cls;
        int e = r;
        do { cls.push_back(e); done[e] = 1; e = (int)(((int64_t)e * 2) % N); } while (e != r);
        // minimal polynomial = prod (x + alpha^e), coefficients in the field, end up in {0,1}
        std::vector<uint16_t> mp(1, 1);
        for (int ee : cls) {
            std::vector<uint16_t> nx(mp.size() + 1, 0);
            for (size_t k = 0; k < mp.size(); ++k) {
                nx[k + 1] ^= mp[k];
                nx[k] ^= vmul(mp[k], EXP[ee]);
            }
            mp.swap(nx);
        }
        std::vector<uint8_t> ng(g.size() + mp.size() - 1, 0);
        for (size_t i = 0; i < g.size(); ++i)
            if (g[i])
                for (size_t k = 0; k < mp.size(); ++k) ng[i + k] ^= (uint8_t)(mp[k] & 1);
        g.swap(ng);
    }
    gen = g;  // degree m*t
}

// bose_chaudhuri_hocquenghem_decoder.hh:41-71.  frame = nbch/8 bytes, data then parity, MSB first.
int bch_syndromes(const BchCode& C, const uint8_t* frame, uint16_t* syn) {
    const int NR = C.NR;
    int c0 = get_be_bit(frame, 0);
    for (int i = 0; i < NR; ++i) syn[i] = (uint16_t)c0;
    for (int j = 1; j < C.nbch; ++j) {
        int coeff = get_be_bit(frame, j);
        for (int i = 0; i < NR; ++i) {
            uint16_t root = (uint16_t)(i + 1);
            uint16_t s = syn[i];
            // fma(Index root, Value s, Value coeff) = !s ? coeff : value(root*index(s)) + coeff
            syn[i] = (uint16_t)((!s ? 0 : C.EXP[C.imul(root, C.LOG[s])]) ^ coeff);
        }
    }
    int nonzero = 0;
    for (int i = 0; i < NR; ++i) nonzero += !!syn[i];
    return nonzero;
}

static int berlekamp_massey(const BchCode& G, const uint16_t* s, uint16_t* Cc) {
    const int NR = G.NR;
    uint16_t B[64], T[64];
    for (int i = 0; i <= NR; ++i) B[i] = Cc[i];
    int L = 0;
    for (int n = 0, m = 1; n < NR; ++n) {
        uint16_t d = s[n];
        for (int i = 1; i <= L; ++i) d ^= G.vmul(Cc[i], s[n - i]);
        if (!d) {
            ++m;
        } else {
            for (int i = 0; i < m; ++i) T[i] = Cc[i];
            for (int i = m; i <= NR; ++i) T[i] = (uint16_t)(G.vmul(d, B[i - m]) ^ Cc[i]);
            if (2 * L <= n) {
                L = n + 1 - L;
                for (int i = 0; i <= NR; ++i) B[i] = G.vdiv(Cc[i], d);
                m = 1;
            } else {
                ++m;
            }
            for (int i = 0; i <= NR; ++i) Cc[i] = T[i];
        }
    }
    return L;
}

static int find_locations(const BchCode& G, const uint16_t* locator, int deg, uint16_t* locations) {
    if (deg == 1) {
        locations[0] = G.idiv(G.idiv(G.LOG[locator[0]], G.LOG[locator[1]]), 1);
        return 1;
    }
    if (deg == 2) {
        if (!locator[1] || !locator[0]) return 0;
        uint16_t a = locator[2], b = locator[1], c = locator[0];
        uint16_t ba = G.vdiv(b, a);
        uint16_t Rr = G.IMAP[G.vdiv(G.vmul(a, c), G.vmul(b, b))];
        if (!Rr) return 0;
        uint16_t v0 = G.vmul(ba, Rr);
        locations[0] = G.idiv(G.LOG[v0], 1);
        locations[1] = G.idiv(G.LOG[(uint16_t)(v0 ^ ba)], 1);
        return 2;
    }
    // Chien search (reed_solomon_error_correction.hh:40-61)
    uint16_t tmp[64];
    for (int i = 0; i <= deg; ++i) tmp[i] = locator[i];
    int count = 0;
    for (int i = 0; i < G.N; ++i) {
        uint16_t sum = tmp[0];
        for (int j = 1; j <= deg; ++j) {
            tmp[j] = !tmp[j] ? 0 : G.EXP[G.imul(G.LOG[tmp[j]], (uint16_t)j)];
            sum ^= tmp[j];
        }
        if (!sum) locations[count++] = (uint16_t)i;
    }
    return count;
}

// frame: nbch/8 bytes (data [0,kbch) then parity), corrected in place.
// Returns #corrected bits, 0 for a clean frame, -1 if uncorrectable (frame untouched).
int bch_decode(const BchCode& G, uint8_t* frame) {
    const int NR = G.NR;
    uint16_t syn[64];
    if (!bch_syndromes(G, frame, syn)) return 0;
    uint16_t locator[64];
    locator[0] = 1;
    for (int i = 1; i <= NR; ++i) locator[i] = 0;
    int deg = berlekamp_massey(G, syn, locator);
    while (!locator[deg])
        if (--deg < 0) return -1;
    uint16_t locations[64], magnitudes[64];
    int count = find_locations(G, locator, deg, locations);
    if (count < deg) return -1;
    // Forney (evaluator uses `count` as the locator degree, :205-207)
    uint16_t evaluator[64];
    int etmp = count < NR - 1 ? count : NR - 1;
    int edeg = -1;
    for (int i = 0; i <= etmp; ++i) {
        evaluator[i] = G.vmul(syn[i], locator[0]);
        for (int j = 1; j <= i; ++j) evaluator[i] ^= G.vmul(syn[i - j], locator[j]);
        if (evaluator[i]) edeg = i;
    }
    for (int i = 0; i < count; ++i) {
        uint16_t root = G.imul(locations[i], 1), tmp = root;
        uint16_t eval = evaluator[0];
        for (int j = 1; j <= edeg; ++j) {
            eval ^= !evaluator[j] ? 0 : G.EXP[G.imul(G.LOG[evaluator[j]], tmp)];
            tmp = G.imul(tmp, root);
        }
        if (!eval) { magnitudes[i] = 0; continue; }
        uint16_t deriv = locator[1];
        uint16_t root2 = G.imul(root, root), tmp2 = root2;
        for (int j = 3; j <= count; j += 2) {
            deriv ^= !locator[j] ? 0 : G.EXP[G.imul(G.LOG[locator[j]], tmp2)];
            tmp2 = G.imul(tmp2, root2);
        }
        uint16_t mag = G.idiv(G.LOG[eval], G.LOG[deriv]);
        magnitudes[i] = G.EXP[mag];
    }
    if (count <= 0) return count;
    const int short_by = G.K_full - G.kbch;
    for (int i = 0; i < count; ++i)
        if ((int)locations[i] < short_by) return -1;
    for (int i = 0; i < count; ++i)
        if (1 < (int)magnitudes[i]) return -1;
    for (int i = 0; i < count; ++i) {
        int idx = (int)locations[i] - short_by;
        xor_be_bit(frame, idx, magnitudes[i] ? 1 : 0);  // data and parity are contiguous in `frame`
    }
    int corr = 0;
    for (int i = 0; i < count; ++i) corr += !!magnitudes[i];
    return corr;
}

// frame: nbch/8 bytes, first kbch bits are data; parity bits are computed and written after them.
void bch_encode(const BchCode& G, uint8_t* frame) {
    const int np = G.m * G.t;
    std::vector<uint8_t> reg(np, 0);  // reg[k] = coefficient of x^k
    for (int j = 0; j < G.kbch; ++j) {
        int fb = get_be_bit(frame, j) ^ reg[np - 1];
        for (int k = np - 1; k > 0; --k) reg[k] = (uint8_t)(reg[k - 1] ^ (fb & G.gen[k]));
        reg[0] = (uint8_t)(fb & G.gen[0]);
    }
    for (int j = 0; j < np; ++j) {
        int pos = G.kbch + j;
        frame[pos / 8] = (uint8_t)((frame[pos / 8] & ~(1 << (7 - pos % 8))) | (reg[np - 1 - j] << (7 - pos % 8)));
    }
}

}  // namespace orc
