// ORACLE support (test infrastructure, CPU only) -- not part of the shipped engine.
// extern "C" entry points around the reference's own, unmodified, self-contained FEC classes.
// This file is the only thing of ours in oracle/_ref/libdvbs2ref.so: the reference sources are compiled
// where they lie under /root/reference (see oracle/Makefile, target `ref`); nothing is copied.
// It exists so tests can (a) pin the restatement in oracle/*.cpp and (b) generate tests/golden/*.
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "dvbs2/codings/bbframe_ldpc.h"
#include "dvbs2/codings/bbframe_bch.h"
#include "dvbs2/codings/bbframe_descramble.h"
#include "dvbs2/codings/s2_deinterleaver.h"

using namespace dsp::dvbs2;

static dvbs2_code_rate_t to_rate(int r) {
    // our rate index 0..10 = 1/4 1/3 2/5 1/2 3/5 2/3 3/4 4/5 5/6 8/9 9/10; the reference enum has C7_8 at 9
    static const dvbs2_code_rate_t map[11] = {C1_4, C1_3, C2_5, C1_2, C3_5, C2_3, C3_4, C4_5, C5_6, C8_9, C9_10};
    return map[r];
}
static dvbs2_framesize_t to_fs(int s) { return s ? FECFRAME_SHORT : FECFRAME_NORMAL; }

// The reference's GF(2^m) LOG/EXP tables are static pointers that every BBFrameBCH constructor re-allocates and every destructor
// frees (bch/galois_field.hh:128, bbframe_bch.cpp:183-185,375-377): objects of different threads would pull the tables from under
// each other.  The plugin only ever has one; for the multi-threaded CPU baseline keep one object per (thread, code), construct
// under a lock and never destroy it.
static BBFrameBCH& bch_for(int rate, int shortframe) {
    static std::mutex mtx;
    thread_local std::map<int, BBFrameBCH*> cache;
    BBFrameBCH*& p = cache[rate * 2 + shortframe];
    if (!p) {
        std::lock_guard<std::mutex> lk(mtx);
        p = new BBFrameBCH(to_fs(shortframe), to_rate(rate));
    }
    return *p;
}

extern "C" {

// BBFrameLDPC::decode exactly as the plugin calls it (one frame, lane 0).  bbframe_ldpc.cpp:123-139
int ref_ldpc_decode(int rate, int shortframe, int8_t* frame, int max_trials) {
    BBFrameLDPC ldpc(to_fs(shortframe), to_rate(rate));
    return ldpc.decode(frame, max_trials);
}

// Decode `nframes` frames one after the other with one decoder object (amortises table set-up).
void ref_ldpc_decode_many(int rate, int shortframe, int8_t* frames, int nframes, int max_trials, int* trials_out) {
    BBFrameLDPC ldpc(to_fs(shortframe), to_rate(rate));
    int n = shortframe ? 16200 : 64800;
    for (int f = 0; f < nframes; ++f) trials_out[f] = ldpc.decode(frames + (size_t)f * n, max_trials);
}

// The library used the way its SIMD type was designed for (bbframe_ldpc.h:21-22): 16 frames, one per
// int8 lane, one decoder call.  Used only as the "library-intended" CPU baseline.
// frames: 16 x N, frame-major.  Returns the decoder's raw `trials` value (remaining, or -1).
int ref_ldpc_decode_simd16(int rate, int shortframe, int8_t* frames, int max_trials, int reps) {
    BBFrameLDPC holder(to_fs(shortframe), to_rate(rate));
    LDPCDecoder<simd_type, algorithm_type> dec;
    dec.init(holder.get_instance());
    int n = shortframe ? 16200 : 64800;
    int k = holder.dataSize();
    const int W = simd_type::SIZE;
    std::vector<simd_type> buf(n);
    int last = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int i = 0; i < n; ++i)
            for (int l = 0; l < W; ++l) reinterpret_cast<int8_t*>(&buf[i])[l] = frames[(size_t)l * n + i];
        last = dec(buf.data(), buf.data() + k, max_trials, W);
    }
    for (int i = 0; i < n; ++i)
        for (int l = 0; l < W; ++l) frames[(size_t)l * n + i] = reinterpret_cast<int8_t*>(&buf[i])[l];
    return last;
}

int ref_simd_width(void) { return simd_type::SIZE; }

// Persistent "library-intended" FEC object for the CPU baseline: tables built ONCE (BBFrameLDPC, LDPCDecoder::init, the thread's
// BBFrameBCH, BBFrameDescrambler), then every call = one 16-lane LDPC decode (input already lane-interleaved: lanes[i*16 + l] = LLR i
// of frame l, so no transposition is timed) + per frame the hard-decision repack of module_dvbs2_demod.cpp:357-360, BCH and descramble.
struct RefFec16 {
    BBFrameLDPC holder;
    LDPCDecoder<simd_type, algorithm_type> dec;
    BBFrameDescrambler descr;
    int rate, sh, n, k;
    std::vector<simd_type> buf;
    std::vector<uint8_t> frame;
    RefFec16(int r, int s) : holder(to_fs(s), to_rate(r)), descr(to_fs(s), to_rate(r)), rate(r), sh(s) {
        dec.init(holder.get_instance());
        n = s ? 16200 : 64800;
        k = holder.dataSize();
        buf.resize(n);
        frame.resize(n / 8);
    }
};
void* ref_fec16_create(int rate, int shortframe) { return new RefFec16(rate, shortframe); }
void ref_fec16_destroy(void* h) { delete (RefFec16*)h; }
// returns the decoder's raw trials value of the 16-lane call; bb_out (optional): 16 x kbch_bytes descrambled BBFRAMEs
int ref_fec16_run(void* h, const int8_t* lanes, int max_trials, uint8_t* bb_out, int kbch_bytes) {
    RefFec16* f = (RefFec16*)h;
    const int W = simd_type::SIZE;
    memcpy((void*)f->buf.data(), lanes, (size_t)f->n * W);
    const int ret = f->dec(f->buf.data(), f->buf.data() + f->k, max_trials, W);
    BBFrameBCH& bch = bch_for(f->rate, f->sh);
    for (int l = 0; l < W; ++l) {
        uint8_t* fr = f->frame.data();
        memset(fr, 0, f->frame.size());
        for (int i = 0; i < f->k; ++i) fr[i / 8] = (uint8_t)(fr[i / 8] << 1 | (reinterpret_cast<const int8_t*>(&f->buf[i])[l] < 0));
        bch.decode(fr);
        f->descr.work(fr);
        if (bb_out) memcpy(bb_out + (size_t)l * kbch_bytes, fr, kbch_bytes);
    }
    return ret;
}
const char* ref_build_flags(void) { return REF_BUILD_FLAGS; }

int ref_bch_decode(int rate, int shortframe, uint8_t* frame) { return bch_for(rate, shortframe).decode(frame); }

void ref_bch_decode_many(int rate, int shortframe, uint8_t* frames, int nframes, int nbch_bytes, int* corr_out) {
    BBFrameBCH& bch = bch_for(rate, shortframe);
    int nb = nbch_bytes;
    for (int f = 0; f < nframes; ++f) corr_out[f] = bch.decode(frames + (size_t)f * nb);
}

void ref_bch_encode(int rate, int shortframe, uint8_t* frame) {
    bch_for(rate, shortframe).encode(frame);
}

void ref_bb_descramble(int rate, int shortframe, uint8_t* frame) {
    BBFrameDescrambler d(to_fs(shortframe), to_rate(rate));
    d.work(frame);
}

void ref_deinterleave(int constel, int rate, int shortframe, int8_t* in, int8_t* out) {
    S2Deinterleaver d((dvbs2_constellation_t)constel, to_fs(shortframe), to_rate(rate));
    d.deinterleave(in, out);
}

}  // extern "C"

// ---------------------------------------------------------------------------------- DVB-S pieces that compile as is
#include "dvbs/depunc.h"
#include "dvbs/dvbs_interleaving.h"
#include "common/codings/rotation.h"

extern "C" {
void* ref_depunc23_create() { return new viterbi::Depunc23(); }
void* ref_depunc56_create() { return new viterbi::Depunc56(); }
int ref_depunc23_static(void* h, uint8_t* in, uint8_t* out, int size, int shift) { return ((viterbi::Depunc23*)h)->depunc_static(in, out, size, shift); }
int ref_depunc56_static(void* h, uint8_t* in, uint8_t* out, int size, int shift) { return ((viterbi::Depunc56*)h)->depunc_static(in, out, size, shift); }
void ref_depunc23_set_shift(void* h, int s) { ((viterbi::Depunc23*)h)->set_shift(s); }
void ref_depunc56_set_shift(void* h, int s) { ((viterbi::Depunc56*)h)->set_shift(s); }
int ref_depunc23_cont(void* h, uint8_t* in, uint8_t* out, int size) { return ((viterbi::Depunc23*)h)->depunc_cont(in, out, size); }
int ref_depunc56_cont(void* h, uint8_t* in, uint8_t* out, int size) { return ((viterbi::Depunc56*)h)->depunc_cont(in, out, size); }
void* ref_forney_create() { return new dsp::dvbs::DVBSInterleaving(); }
void ref_forney_deinterleave(void* h, uint8_t* in, uint8_t* out) { ((dsp::dvbs::DVBSInterleaving*)h)->deinterleave(in, out); }
void ref_rotate_soft(int8_t* soft, int size, int phase, int iqswap) { rotate_soft(soft, size, (phase_t)phase, iqswap != 0); }
}

// ---------------------------------------------------------------------------------- DVB-S tail: deframer, RS(204,188), descrambler
#include "dvbs/dvbs_ts_deframer.h"
#include "dvbs/dvbs_reedsolomon.h"
#include "dvbs/dvbs_scrambling.h"
extern "C" {
void* ref_tsdef_create() {
    auto* d = new deframing::DVBS_TS_Deframer();
    // the reference leaves its 13 056-bit shifter uninitialised; feed zeros so that runs are reproducible
    std::vector<uint8_t> z(1632 * 8, 0), o(16 * 1632);
    d->work(z.data(), (int)z.size(), o.data());
    return d;
}
int ref_tsdef_work(void* h, uint8_t* bits, int size, uint8_t* out) { return ((deframing::DVBS_TS_Deframer*)h)->work(bits, size, out); }
void* ref_dvbsrs_create() { return new dsp::dvbs::DVBSReedSolomon(); }
int ref_dvbsrs_decode(void* h, uint8_t* data204) { return ((dsp::dvbs::DVBSReedSolomon*)h)->decode(data204); }
void* ref_dvbsdescr_create() { return new dsp::dvbs::DVBSScrambling(); }
void ref_dvbsdescr_work(void* h, uint8_t* frm) { ((dsp::dvbs::DVBSScrambling*)h)->descramble(frm); }
}
