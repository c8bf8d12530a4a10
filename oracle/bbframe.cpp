// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
//   bb_prbs / bb_descramble   dvbs2/codings/bbframe_descramble.cpp:122-143 (PRBS 1+x^14+x^15, seed 0x4A80)
//   hard_pack                 dvbs2/module_dvbs2_demod.cpp:357-360        (bit = posterior < 0, MSB first)
//   bbheader_crc8             dvbs2/bbframe_ts_parser.cpp:44-83           (what the consumer checks over 80 bits)
//   make_bbframe              this repo's synthetic payload generator (SURVEY 8d "Synthetic inputs")
#include "oracle.h"
#include <cstring>

namespace orc {

void bb_prbs(uint8_t* seq, int nbytes) {
    memset(seq, 0, nbytes);
    int sr = 0x4A80;
    for (int i = 0; i < nbytes * 8; i++) {
        int b = ((sr) ^ (sr >> 1)) & 1;
        seq[i / 8] |= (uint8_t)(b << (7 - (i % 8)));
        sr >>= 1;
        if (b) sr |= 0x4000;
    }
}

void bb_descramble(uint8_t* frame, int nbytes) {
    std::vector<uint8_t> seq(nbytes);
    bb_prbs(seq.data(), nbytes);
    for (int j = 0; j < nbytes; ++j) frame[j] ^= seq[j];
}

void hard_pack(const int8_t* post, int nbits, uint8_t* out) {
    memset(out, 0, (nbits + 7) / 8);
    for (int i = 0; i < nbits; i++) out[i / 8] = (uint8_t)(out[i / 8] << 1 | (post[i] < 0));
}

// CRC-8, generator x^8+x^7+x^6+x^4+x^2+1 (0xD5), MSB first, zero init (ETSI EN 302 307-1 5.1.4).
uint8_t bbheader_crc8(const uint8_t* hdr9) {
    uint8_t crc = 0;
    for (int n = 0; n < 72; ++n) {
        int bit = (hdr9[n / 8] >> (7 - n % 8)) & 1;
        int fb = bit ^ (crc >> 7);
        crc = (uint8_t)(crc << 1);
        if (fb) crc ^= 0xD5;
    }
    return crc;
}

static inline uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Unscrambled BBFRAME of kbch bits: valid 10-byte BBHEADER (TS, single stream, CCM, no ISSY/NPD,
// RO 0.35, UPL 188*8, DFL kbch-80, SYNC 0x47, SYNCD 0) followed by seeded random payload.
void make_bbframe(uint8_t* frame, int kbch, uint64_t seed) {
    int nbytes = kbch / 8;
    uint64_t s = seed;
    for (int i = 0; i < nbytes; i += 8) {
        uint64_t v = splitmix(s);
        for (int k = 0; k < 8 && i + k < nbytes; ++k) frame[i + k] = (uint8_t)(v >> (8 * k));
    }
    int dfl = kbch - 80;
    frame[0] = 0xF0;  // TS/GS=11, SIS=1, CCM=1, ISSYI=0, NPD=0, RO=00 (0.35)
    frame[1] = 0x00;  // MATYPE-2
    frame[2] = (uint8_t)((188 * 8) >> 8); frame[3] = (uint8_t)((188 * 8) & 0xff);
    frame[4] = (uint8_t)(dfl >> 8); frame[5] = (uint8_t)(dfl & 0xff);
    frame[6] = 0x47;
    frame[7] = 0; frame[8] = 0;
    frame[9] = bbheader_crc8(frame);
}

}  // namespace orc
