// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.  See bbframe_ts.cpp.
#pragma once
#include <cstdint>
#include <vector>

namespace orc {

// dsp::dvbs2::BBHeader (dvbs2/bbframe_ts_parser.h:37-66)
struct BbHeader {
    int ts_gs = 0, sis_mis = 0, ccm_acm = 0, issyi = 0, npd = 0, ro = 0, isi = 0, upl = 0, dfl = 0, sync = 0, syncd = 0;
};

// dsp::dvbs2::BBFrameTSParser (dvbs2/bbframe_ts_parser.h:68-112, .cpp:31-390)
struct BbTsParser {
    BbHeader last_header;
    int last_gse_crc_err = 0, last_bb_cnt = 0, last_bb_proc = 0, last_ts_errs = 0;

    BbTsParser();
    void set_frame_size(int kbch_bits);
    // returns bytes written to `out`, or -1 where the reference runs out of output space in its TS loop (it prints
    // "BUFF OVF!" and then copies an unbounded tail into a 188-byte array -- undefined; callers size `cap` to avoid it)
    int work(const uint8_t* bb, int cnt, uint8_t* out, int cap);

    // exposed for tests
    int synched = 0, count = 0;

private:
    int kbch = 0, max_dfl = 0;
    uint8_t partial[188];
    struct Slot {
        bool active = false;
        int id = 0, ctr = 0;
        uint16_t proto = 0;
        uint32_t crc = 0;
        std::vector<uint8_t> buf;
    } slot[3];
    uint32_t crc32_tab[256];
    uint32_t crc32(const uint8_t* p, int n, uint32_t c) const;
};

unsigned bbts_crc8_bits(const uint8_t* in, int nbits);   // check_crc8 (.cpp:70-83)

}  // namespace orc
