// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.  See dvbs.cpp.
#pragma once
#include <cstdint>
#include <deque>
#include <vector>

namespace orc {

constexpr int VIT_BUF = 8192;   // dvbs/dvbs_defines.h:3

int8_t dvbs_clamp(float x);
struct DvbsSlicer {            // DVBSymToSoftBlock
    int8_t sym_buffer[VIT_BUF + 16];
    int fill = 0;
    int process(int count, const float* iq, int8_t* out);
};
void rotate_soft(int8_t* soft, int size, int phase, bool iqswap);
void signed_soft_to_unsigned(const int8_t* in, uint8_t* out, int n);
int depuncture_34(const uint8_t* in, uint8_t* out, int size, bool shift);
int depuncture_78(const uint8_t* in, uint8_t* out, int size, int shift);
struct DepuncCont {            // Depunc23 (period 3) / Depunc56 (period 6)
    int period;
    bool is_first = false;
    int changing_shift = 0;
    int got_extra = false;
    uint8_t buf = 128;
    explicit DepuncCont(int p) : period(p) {}
    int depunc_static(const uint8_t* in, uint8_t* out, int size, int shift) const;
    void set_shift(int shift);
    int depunc_cont(const uint8_t* in, uint8_t* out, int size);
};
struct CcDecoder {
    int frame_size, veclen;
    uint8_t m1[64], m2[64];
    uint8_t branchtab[64];
    std::vector<uint8_t> decisions;
    int start_state_chaining;
    explicit CcDecoder(int frame_size);
    void init_viterbi(int starting_state);
    void work(const uint8_t* in, uint8_t* out);
};
struct CcEncoder {
    int frame_size;
    unsigned state = 0;
    explicit CcEncoder(int fs) : frame_size(fs) {}
    void work(const uint8_t* in, uint8_t* out);
};
float dvbs_get_ber(const uint8_t* raw, const uint8_t* rencoded, int len, float ratio);

struct ViterbiDvbs {
    float ber_thr; int max_outsync, bufsize;
    CcDecoder dec_ber_12; CcEncoder enc_ber_12; CcDecoder dec_ber_23; CcEncoder enc_ber_23; CcDecoder dec_ber_34; CcEncoder enc_ber_34;
    CcDecoder dec_ber_56; CcEncoder enc_ber_56; CcDecoder dec_ber_78; CcEncoder enc_ber_78;
    CcDecoder dec_12, dec_23, dec_34, dec_56, dec_78;
    DepuncCont dep23, dep56;
    std::vector<uint8_t> soft_buffer, depunc_buffer, ber_dec, ber_enc;
    // ber_soft_buffer[2048] and ber_depunc_buffer[8192] are adjacent members in the reference (viterbi_all.h:74-78) and the rate-1/2
    // test decoder reads 2060 bytes from ber_soft_buffer+shift, i.e. runs over into ber_depunc_buffer: keep them contiguous.
    std::vector<uint8_t> ber_area;
    uint8_t* ber_soft; uint8_t* ber_depunc;
    int state = 0, d_phase = 0, d_shift = 0, invalid = 0, rate = 0;
    float ber = 10;
    ViterbiDvbs(float ber_threshold, int max_outsync, int buffer_size);
    int work(int8_t* input, int size, uint8_t* output);
};

struct ForneyDeint {
    std::vector<std::deque<uint8_t>> fifo;
    ForneyDeint();
    void deinterleave(const uint8_t* in, uint8_t* out);   // 8 x 204 bytes
};

}  // namespace orc
