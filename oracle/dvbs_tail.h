// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.  See dvbs_tail.cpp.
#pragma once
#include <cstdint>
#include <vector>

namespace orc {

struct TsDeframer {                 // deframing::DVBS_TS_Deframer (dvbs/dvbs_ts_deframer.cpp)
    static constexpr int TS_SIZE = 1632 * 8;
    std::vector<uint8_t> shifter;   // TS_SIZE unpacked bits (the reference's starts uninitialised; zero here)
    int errors_nor = 0, errors_inv = 0;
    TsDeframer() : shifter(TS_SIZE, 0) {}
    int work(const uint8_t* input, int size, uint8_t* output);   // returns frame count, 1632 bytes each
};

struct Gf256 {                      // common/correct/reed-solomon/field.h, primitive polynomial 0x11d
    uint8_t exp[512], log[256];
    Gf256();
};
const Gf256& gf256();

struct DvbsRs {                     // dsp::dvbs::DVBSReedSolomon (dvbs/dvbs_reedsolomon.h) over libcorrect's decoder
    uint8_t obuffer[255];
    DvbsRs();
    int decode(uint8_t* data204);   // in place on the first 188 bytes; returns the reference's error count
};
// correct_reed_solomon_decode for (255, 239), fcr 0, gap 1 (reed-solomon/decode.c:299-380): returns 239 or -1; msg untouched on -1
int rs255_decode(const uint8_t* encoded255, uint8_t* msg239);

struct DvbsDescrambler {            // dsp::dvbs::DVBSScrambling (dvbs/dvbs_scrambling.h)
    int reg = 0;
    int prbs(int clocks);
    void descramble(uint8_t* frm1632);
};

}  // namespace orc
