// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// DVB-S inner-code path: soft slicer, phase rotation, depuncturing, K=7 r=1/2 Viterbi block decoder with
// chained start state, re-encoder/BER watchdog, Forney de-interleaver.
//
// Restates (reference file:line):
//   DVBSymToSoftBlock::process / clamp        dvbs/dvbs_syms_to_soft.cpp:7-42
//   rotate_soft                               common/codings/rotation.cpp:4-63          (pinned via oracle/_ref)
//   signed_soft_to_unsigned                   common/utils.cpp:11-20
//   depuncture_34 / depuncture_78             dvbs/viterbi_all.h:92-150
//   Depunc23 / Depunc56                       dvbs/depunc.h:8-190                       (pinned via oracle/_ref)
//   CCDecoder (create/init/update/endstate/chainback/work)   dvbs/viterbi/cc_decoder.cpp:10-314
//   generic ACS kernel + renormalize          dvbs/viterbi/volk_k7_r2_generic_fixed.h:80-163
//   CCEncoder::work                           dvbs/viterbi/cc_encoder.cpp:92-104
//   Viterbi_DVBS::get_ber / work              dvbs/viterbi_all.cpp:59-276
//   DVBSInterleaving::deinterleave            dvbs/dvbs_interleaving.h:58-70            (pinned via oracle/_ref)
// PARITY UNPINNED for the Viterbi decoder itself: cc_decoder.cpp / volk_k7_r2_generic_fixed.h include VOLK headers,
// which are not in /root/reference, so they cannot be compiled here; on x86 with VOLK present the
// reference would even pick VOLK's "spiral" kernel instead of the bundled generic one (Q8).
#include "dvbs.h"
#include <cstring>
#include <cstdlib>

namespace orc {

int8_t dvbs_clamp(float x) {
    if (x < -127.0) return -127;
    if (x > 127.0) return 127;
    return (int8_t)x;
}

// ------------------------------------------------------------------ soft slicer (8192-byte blocks)
int DvbsSlicer::process(int count, const float* iq, int8_t* out) {
    int curroutidx = 0;
    for (int i = 0; i < count; i++) {
        sym_buffer[fill + 0] = dvbs_clamp(iq[2 * i] * 100);
        sym_buffer[fill + 1] = dvbs_clamp(iq[2 * i + 1] * 100);
        fill += 2;
        if (fill >= VIT_BUF) {
            memcpy(&out[curroutidx], sym_buffer, VIT_BUF);
            curroutidx += VIT_BUF;
            fill -= VIT_BUF;
        }
    }
    return curroutidx;
}

void rotate_soft(int8_t* soft, int size, int phase, bool iqswap) {
    for (int i = 0; i < size; i++)
        if (soft[i] == -128) soft[i] = -127;
    if (iqswap)
        for (int i = 0; i < size; i += 2) { int8_t x = soft[i + 1]; soft[i + 1] = soft[i]; soft[i] = x; }
    int8_t tmp;
    switch (phase) {
        case 1: for (; size > 0; size -= 2) { tmp = *soft; *soft = *(soft + 1); *(soft + 1) = (int8_t)-tmp; soft += 2; } break;
        case 2: for (; size > 0; size--) { *soft = (int8_t)-*soft; soft++; } break;
        case 3: for (; size > 0; size -= 2) { tmp = *soft; *soft = (int8_t)-*(soft + 1); *(soft + 1) = tmp; soft += 2; } break;
        default: break;
    }
}

void signed_soft_to_unsigned(const int8_t* in, uint8_t* out, int n) {
    for (int i = 0; i < n; i++) {
        out[i] = (uint8_t)(in[i] + 127);
        if (out[i] == 128) out[i] = 127;
    }
}

int depuncture_34(const uint8_t* in, uint8_t* out, int size, bool shift) {
    int oo = 0;
    for (int i = 0; i < size / 2; i++) {
        if (shift ^ (i % 2 == 0)) { out[oo++] = in[i * 2 + 0]; out[oo++] = in[i * 2 + 1]; }
        else { out[oo++] = 128; out[oo++] = in[i * 2 + 0]; out[oo++] = in[i * 2 + 1]; out[oo++] = 128; }
    }
    return oo;
}
int depuncture_78(const uint8_t* in, uint8_t* out, int size, int shift) {
    int oo = 0;
    for (int i = 0; i < size / 2; i++) {
        int m = (i + shift) % 4;
        if (m == 0) { out[oo++] = in[i * 2 + 0]; out[oo++] = in[i * 2 + 1]; }
        else if (m == 1) { out[oo++] = 128; out[oo++] = in[i * 2 + 0]; out[oo++] = 128; out[oo++] = in[i * 2 + 1]; }
        else { out[oo++] = 128; out[oo++] = in[i * 2 + 0]; out[oo++] = in[i * 2 + 1]; out[oo++] = 128; }
    }
    return oo;
}

// pattern of rate 2/3 (period 3) and 5/6 (period 6): for input i at phase p, emit {x}, {x,128} or {128,x}
static inline int depunc_emit(int period, int p, uint8_t x, uint8_t* out, int oo) {
    if (period == 3) {
        if (p == 1) { out[oo++] = x; out[oo++] = 128; } else out[oo++] = x;
    } else {
        if (p == 1 || p == 3 || p == 5) { out[oo++] = x; out[oo++] = 128; }
        else if (p == 4) { out[oo++] = 128; out[oo++] = x; }
        else out[oo++] = x;
    }
    return oo;
}
int DepuncCont::depunc_static(const uint8_t* in, uint8_t* out, int size, int shift) const {
    int oo = 0, actual = shift % period;
    if (shift > period - 1) out[oo++] = 128;
    for (int i = 0; i < size; i++) oo = depunc_emit(period, (i + actual) % period, in[i], out, oo);
    return oo;
}
void DepuncCont::set_shift(int shift) { changing_shift = shift; is_first = shift > period - 1; }
int DepuncCont::depunc_cont(const uint8_t* in, uint8_t* out, int size) {
    int oo = 0;
    if (is_first || got_extra) { out[oo++] = buf; is_first = false; got_extra = false; }
    changing_shift = changing_shift % period;
    for (int i = 0; i < size; i++) { oo = depunc_emit(period, changing_shift % period, in[i], out, oo); changing_shift++; }
    if (oo % 2 == 1) { buf = out[oo - 1]; oo -= 1; got_extra = true; }
    return oo;
}

// ------------------------------------------------------------------ convolutional decoder K=7, r=1/2, polys {79,109}
static int parity32(int x) { x ^= (x >> 16); x ^= (x >> 8); x &= 0xff; int c = 0; while (x) { c += x & 1; x >>= 1; } return c & 1; }

CcDecoder::CcDecoder(int frame_size_) : frame_size(frame_size_) {
    veclen = frame_size + 6;
    decisions.assign((size_t)veclen * 8, 0);
    const int polys[2] = {79, 109};
    for (int state = 0; state < 32; state++)
        for (int i = 0; i < 2; i++) branchtab[i * 32 + state] = (uint8_t)(((polys[i] < 0) ^ parity32((2 * state) & abs(polys[i]))) ? 255 : 0);
    for (int i = 0; i < 64; i++) m1[i] = 31;   // init_viterbi_unbiased (cc_decoder.cpp:177-190)
    start_state_chaining = 0;
}
void CcDecoder::init_viterbi(int starting_state) {
    for (int i = 0; i < 64; i++) m1[i] = 63;
    m1[starting_state & 63] = 0;
}
// in: 2*veclen unsigned softs (the caller's buffer must extend that far, like the reference's does)
void CcDecoder::work(const uint8_t* in, uint8_t* out) {
    // update_viterbi_blk + generic kernel
    memset(decisions.data(), 0, (size_t)8 * veclen);
    uint8_t* X = m1;   // old_metrics
    uint8_t* Y = m2;   // new_metrics
    for (int s = 0; s < veclen; s++) {
        for (int i = 0; i < 32; i++) {
            unsigned short metricsum = 1;
            for (int j = 0; j < 2; j++) metricsum += (branchtab[i + j * 32] ^ in[s * 2 + j]);
            uint8_t metric = (uint8_t)((metricsum >> 1) >> 2);
            const uint8_t max = ((2 * ((256 - 1) >> 1)) >> 2);
            uint8_t a0 = (uint8_t)(X[i] + metric), a1 = (uint8_t)(X[i + 32] + (max - metric));
            uint8_t a2 = (uint8_t)(X[i] + (max - metric)), a3 = (uint8_t)(X[i + 32] + metric);
            unsigned d0 = (signed int)(a0 - a1) >= 0, d1 = (signed int)(a2 - a3) >= 0;
            Y[2 * i] = d0 ? a1 : a0;
            Y[2 * i + 1] = d1 ? a3 : a2;
            uint32_t* w = reinterpret_cast<uint32_t*>(decisions.data());
            w[i / 16 + s * 2] |= (d0 | d1 << 1) << ((2 * i) & 31);
        }
        uint8_t mn = Y[0];
        for (int i = 0; i < 64; i++) if (mn > Y[i]) mn = Y[i];
        for (int i = 0; i < 64; i++) Y[i] -= mn;
        uint8_t* t = X; X = Y; Y = t;
    }
    // find_endstate: the struct's old/new pointers are not swapped by the kernel (cc_decoder.cpp:192-209)
    const uint8_t* met = ((7 + veclen) % 2 == 0) ? m2 : m1;
    uint8_t mn = met[0];
    int endstate = 0;
    for (int i = 1; i < 64; ++i) if (met[i] < mn) { mn = met[i]; endstate = i; }
    // chainback_viterbi(out, frame_size, endstate, tailsize = veclen - frame_size = 6)
    const int ADDSHIFT = 2, tailsize = 6;
    const uint8_t* d = decisions.data() + (size_t)tailsize * 8;
    unsigned es = (unsigned)(endstate % 64) << ADDSHIFT;
    unsigned nbits = (unsigned)frame_size;
    int retval = 0;
    const int dif = tailsize - 6;
    while (nbits-- > (unsigned)(frame_size - 6)) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&d[(size_t)nbits * 8]);
        int k = (w[(es >> ADDSHIFT) / 32] >> ((es >> ADDSHIFT) % 32)) & 1;
        es = (es >> 1) | ((unsigned)k << (7 - 2 + ADDSHIFT));
        out[(nbits + dif) % frame_size] = (uint8_t)k;
        retval = (int)es;
    }
    nbits += 1;
    while (nbits-- != 0) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&d[(size_t)nbits * 8]);
        int k = (w[(es >> ADDSHIFT) / 32] >> ((es >> ADDSHIFT) % 32)) & 1;
        es = (es >> 1) | ((unsigned)k << (7 - 2 + ADDSHIFT));
        out[(nbits + dif) % frame_size] = (uint8_t)k;
    }
    start_state_chaining = retval >> ADDSHIFT;
    init_viterbi(start_state_chaining);
}

void CcEncoder::work(const uint8_t* in, uint8_t* out) {
    const int polys[2] = {79, 109};
    unsigned my_state = state;
    for (int i = 0; i < frame_size; ++i) {
        my_state = (my_state << 1) | (in[i] & 1);
        for (int j = 0; j < 2; ++j) out[i * 2 + j] = (uint8_t)(((polys[j] < 0) ^ parity32((int)(my_state & (unsigned)abs(polys[j])))) ? 1 : 0);
    }
    state = my_state;
}

float dvbs_get_ber(const uint8_t* raw, const uint8_t* rencoded, int len, float ratio) {
    float errors = 0, total = 0;
    for (int i = 0; i < len; i++)
        if (raw[i] != 128) { errors += (raw[i] > 127) != rencoded[i]; total++; }
    return (errors / total) * ratio;
}

// ------------------------------------------------------------------ Viterbi_DVBS
static const int TEST_BITS = 2048;

ViterbiDvbs::ViterbiDvbs(float thr, int max_outsync, int buffer_size)
    : ber_thr(thr), max_outsync(max_outsync), bufsize(buffer_size),
      dec_ber_12(TEST_BITS / 2), enc_ber_12(TEST_BITS / 2), dec_ber_23((int)(TEST_BITS * 1.334 / 2)), enc_ber_23((int)(TEST_BITS * 1.334 / 2)),
      dec_ber_34((int)(TEST_BITS * 1.5 / 2)), enc_ber_34((int)(TEST_BITS * 1.5 / 2)), dec_ber_56((int)(TEST_BITS * 1.66 / 2)),
      enc_ber_56((int)(TEST_BITS * 1.66 / 2)), dec_ber_78((int)(TEST_BITS * 1.75 / 2)), enc_ber_78((int)(TEST_BITS * 1.75 / 2)),
      dec_12(buffer_size / 2), dec_23(10924 / 2), dec_34((int)(buffer_size * 1.5 / 2)), dec_56((int)(buffer_size * 1.66 / 2)),
      dec_78((int)(buffer_size * 1.75 / 2)), dep23(3), dep56(6) {
    soft_buffer.assign((size_t)bufsize * 4, 0); depunc_buffer.assign((size_t)bufsize * 4, 0);
    ber_area.assign(TEST_BITS * 5 + 64, 0); ber_soft = ber_area.data(); ber_depunc = ber_soft + TEST_BITS; ber_dec.assign(TEST_BITS * 4, 0); ber_enc.assign(TEST_BITS * 4, 0);
}

// `input` is modified in place (rotation), as in the reference (viterbi_all.cpp:211)
int ViterbiDvbs::work(int8_t* input, int size, uint8_t* output) {
    if (state == 0) {
        ber = 10;
        for (int phase = 0; phase < 2; ++phase) {   // module_dvbs_demod.cpp:23 passes {PHASE_0, PHASE_90}
            int8_t test[TEST_BITS];
            memcpy(test, input, TEST_BITS);
            rotate_soft(test, TEST_BITS, phase, false);
            signed_soft_to_unsigned(test, ber_soft, TEST_BITS);
            auto lock = [&](float b, int sh, int r) {
                ber = b; state = 1; d_phase = phase; d_shift = sh; invalid = 0; rate = r;
                memset(soft_buffer.data(), 128, soft_buffer.size()); memset(depunc_buffer.data(), 128, depunc_buffer.size());
            };
            for (int shift = 0; shift < 2; shift++) {
                dec_ber_12.work(ber_soft + shift, ber_dec.data());
                enc_ber_12.work(ber_dec.data(), ber_enc.data());
                float b = dvbs_get_ber(ber_soft + shift, ber_enc.data(), TEST_BITS, 2.5f);
                if (b < ber_thr) lock(b, shift, 0);
            }
            for (int shift = 0; shift < 6; shift++) {
                dep23.depunc_static(ber_soft, ber_depunc, TEST_BITS, shift);
                dec_ber_23.work(ber_depunc, ber_dec.data());
                enc_ber_23.work(ber_dec.data(), ber_enc.data());
                float b = dvbs_get_ber(ber_depunc, ber_enc.data(), (int)(TEST_BITS * 1.25), 3.5f);
                if (b < ber_thr) { lock(b, shift, 1); dep23.set_shift(shift); }
            }
            for (int shift = 0; shift < 2; shift++) {
                depuncture_34(ber_soft, ber_depunc, TEST_BITS, shift);
                dec_ber_34.work(ber_depunc, ber_dec.data());
                enc_ber_34.work(ber_dec.data(), ber_enc.data());
                float b = dvbs_get_ber(ber_depunc, ber_enc.data(), (int)(TEST_BITS * 1.5), 5.f);
                if (b < ber_thr) lock(b, shift, 2);
            }
            for (int shift = 0; shift < 12; shift++) {
                dep56.depunc_static(ber_soft, ber_depunc, TEST_BITS, shift);
                dec_ber_56.work(ber_depunc, ber_dec.data());
                enc_ber_56.work(ber_dec.data(), ber_enc.data());
                float b = dvbs_get_ber(ber_depunc, ber_enc.data(), (int)(TEST_BITS * 1.66), 8.f);
                if (b < ber_thr) { lock(b, shift, 3); dep56.set_shift(shift); }
            }
            for (int shift = 0; shift < 4; shift++) {
                depuncture_78(ber_soft, ber_depunc, TEST_BITS, shift);
                dec_ber_78.work(ber_depunc, ber_dec.data());
                enc_ber_78.work(ber_dec.data(), ber_enc.data());
                float b = dvbs_get_ber(ber_depunc, ber_enc.data(), (int)(TEST_BITS * 1.75), 10.f);
                if (b < ber_thr) lock(b, shift, 4);
            }
        }
    }
    int out_n = 0;
    if (state == 1) {
        rotate_soft(input, size, d_phase, false);
        signed_soft_to_unsigned(input, soft_buffer.data(), size);
        if (rate == 0) {
            dec_12.work(soft_buffer.data() + d_shift, output);
            out_n = size / 2;
            enc_ber_12.work(output, ber_enc.data());
            ber = dvbs_get_ber(soft_buffer.data() + d_shift, ber_enc.data(), TEST_BITS, 2.5f);
        } else if (rate == 1) {
            int sz = dep23.depunc_cont(soft_buffer.data(), depunc_buffer.data(), size);
            dec_23.work(depunc_buffer.data(), output);
            out_n = sz / 2;
            enc_ber_23.work(output, ber_enc.data());
            ber = dvbs_get_ber(depunc_buffer.data(), ber_enc.data(), (int)(TEST_BITS * 1.25), 3.5f);
        } else if (rate == 2) {
            int sz = depuncture_34(soft_buffer.data(), depunc_buffer.data(), size, d_shift);
            dec_34.work(depunc_buffer.data(), output);
            out_n = sz / 2;
            enc_ber_34.work(output, ber_enc.data());
            ber = dvbs_get_ber(depunc_buffer.data(), ber_enc.data(), (int)(TEST_BITS * 1.5), 5.f);
        } else if (rate == 3) {
            int sz = dep56.depunc_cont(soft_buffer.data(), depunc_buffer.data(), size);
            dec_56.work(depunc_buffer.data(), output);
            out_n = sz / 2;
            enc_ber_56.work(output, ber_enc.data());
            ber = dvbs_get_ber(depunc_buffer.data(), ber_enc.data(), (int)(TEST_BITS * 1.66), 8.f);
        } else {
            int sz = depuncture_78(soft_buffer.data(), depunc_buffer.data(), size, d_shift);
            dec_78.work(depunc_buffer.data(), output);
            out_n = sz / 2;
            enc_ber_78.work(output, ber_enc.data());
            ber = dvbs_get_ber(depunc_buffer.data(), ber_enc.data(), (int)(TEST_BITS * 1.75), 10.f);
        }
        if (ber > ber_thr) { invalid++; if (invalid > max_outsync) state = 0; }
        else invalid = 0;
    }
    return out_n;
}

// ------------------------------------------------------------------ Forney de-interleaver I=12, M=17
ForneyDeint::ForneyDeint() {
    for (int i = 11; i >= 0; i--) fifo.emplace_back((size_t)17 * i, 0);
}
void ForneyDeint::deinterleave(const uint8_t* in, uint8_t* out) {
    int count = 0;
    for (int mux_pkt = 0; mux_pkt < 8; mux_pkt++)
        for (int k = 0; k < 17 * 12; k++) {
            std::deque<uint8_t>& f = fifo[k % 12];
            f.push_back(in[count]);
            out[count++] = f.front();
            f.pop_front();
        }
}

}  // namespace orc
