// Host side of the DVB-S2 demodulator handles: the mirror of dsp::dvbs2::DVBS2Demod
// (reference src/demod/dvbs2/module_dvbs2_demod.{h,cpp}) on top of the kernels in s2_rx_kernels.hip and
// the FEC launches of capi.hip.  One handle = one transponder stream; a batch call runs the same stage of
// all streams in one launch and pools their frames for the FEC kernels.
//
// Call anatomy (process / process_batch), all on one HIP stream:
//   1 front end (AGC+NCO, lane per stream; Gardner, 8 lanes per stream)    agc_pc_kernel, s2_gardner_kernel
//   2 RRC on the kept samples + /2, append to the symbol FIFO               s2_rrc_decim_kernel
//   3 PL sync: every stream's complete windows, correlation + the 2-state
//     realign machine of S2PLSyncBlock (dvbs2_pl_sync.cpp:102-165)          s2_ccm_walk_kernel       -> D2H frame tables (sync 1)
//   4 per-frame loops (FED -> NCO feedback, PLL, PLHDR)                     s2_frame_loops_kernel
//   5 soft demap + de-interleave                                            s2_demap_kernel
//   6 LDPC / BCH / descramble over the pooled frames                        ldpc_decode_kernel, bch_*, bb_descramble
//   7 D2H (or D2D for the batch entry point) of BBFRAMEs + stats; FIFO remainder moved to the spare buffer (sync 2)
// ACM/VCM streams (cfg.acm_vcm): 3 = s2_vcm_walk_kernel, 4-5 with per-frame MODCOD tables, 6 = one FEC job per LDPC code present.
#include "ctx.h"
#include "../../include/dvbs2gpu_math.h"
#include <list>
#include <unordered_map>
#include <tuple>
#include <thread>
#include <cmath>
#include <algorithm>
#include <memory>
#include <chrono>
#include <string>

using namespace s2;

namespace {

// ------------------------------------------------------------------------------- host-side table builders
// (own restatement of the SDR++ tap generators and the reference's constellation / PL constants; the CPU
//  oracle builds the same tables independently in oracle/s2chain.cpp)
}  // namespace
namespace s2 {
std::vector<float> make_rrc_taps(int count, double beta, double Ts) {
    const double PI = 3.14159265358979323846, SQ2 = 1.41421356237309504880;
    double limit = Ts / (4.0 * beta), half = (double)count / 2.0;
    std::vector<float> taps(count);
    for (int i = 0; i < count; i++) {
        double t = (double)i - half + 0.5, v;
        if (t == 0.0) v = (1.0 + beta * (4.0 / PI - 1.0)) / Ts;
        else if (t == limit || t == -limit)
            v = ((1.0 + 2.0 / PI) * sin(PI / (4.0 * beta)) + (1.0 - 2.0 / PI) * cos(PI / (4.0 * beta))) * beta / (Ts * SQ2);
        else
            v = ((sin((1.0 - beta) * PI * t / Ts) + cos((1.0 + beta) * PI * t / Ts) * 4.0 * beta * t / Ts) /
                 ((1.0 - (4.0 * beta * t / Ts) * (4.0 * beta * t / Ts)) * PI * t / Ts)) / Ts;
        taps[i] = (float)v;
    }
    return taps;
}

// gardner.cpp:154-159 (128 phases x 8 taps) and complex_fd.cpp:152-157 (256 x 256): Nuttall-windowed sinc, polyphase bank
std::vector<float> make_polyphase_bank(int phases, int tpp) {
    const double PI = 3.14159265358979323846;
    const int count = phases * tpp;
    double omega = 2.0 * PI * (0.5 / (double)phases), half = (double)count / 2.0, corr = (double)phases * omega / PI;
    const double coefs[4] = {0.355768, 0.487396, 0.144232, 0.012604};
    std::vector<float> bank((size_t)count, 0.f);
    for (int i = 0; i < count; ++i) {
        double t = (double)i - half + 0.5, x = t * omega;
        double sinc = (x == 0.0) ? 1.0 : sin(x) / x;
        double n = t - half, win = 0.0, sign = 1.0;
        for (int c = 0; c < 4; ++c) { win += sign * coefs[c] * cos((double)c * 2.0 * PI * n / (double)count); sign = -sign; }
        bank[(size_t)((phases - 1) - (i % phases)) * tpp + i / phases] = (float)(sinc * win * corr);
    }
    return bank;
}
}  // namespace s2
namespace {
std::vector<float> make_gardner_bank() { return make_polyphase_bank(GARDNER_PHASES, GARDNER_TAPS); }

struct HostConstel {
    int constel, bits, states;
    float amp = 1.0f, sca = 50.0f, prescale = 1.0f;
    cf32 pts[32];
    static cf32 polar(float r, int n, float i) {
        float a = i * 2 * M_PI / n, sn, cs;
        dvbs2m::sincosf_det(a, &sn, &cs);
        return cf32{r * cs, r * sn};
    }
    static cf32 scale(cf32 a, float s) { return cf32{a.re * s, a.im * s}; }
    HostConstel(int type, float g1, float g2) : constel(type) {   // constellation.cpp:19-150
        const double SQ2 = 1.41421356237309504880;
        if (type == C_QPSK) {
            states = 4; bits = 2; amp = 3;
            pts[0] = cf32{(float)-SQ2, (float)-SQ2}; pts[1] = cf32{(float)SQ2, (float)-SQ2};
            pts[2] = cf32{(float)-SQ2, (float)SQ2}; pts[3] = cf32{(float)SQ2, (float)SQ2};
        } else if (type == C_8PSK) {
            states = 8; bits = 3;
            float r = 0.70710678118654752440;
            const cf32 p[8] = {{0.0f, -1.0f}, {-r, r}, {r, -r}, {0.0f, 1.0f}, {-r, -r}, {-1.0f, 0.0f}, {1.0f, 0.0f}, {r, r}};
            for (int i = 0; i < 8; ++i) pts[i] = p[i];
        } else if (type == C_16APSK) {
            states = 16; bits = 4; amp = 100; sca = 1; prescale = 0.53;
            float gamma1 = g1 ? g1 : 2.57f;
            float r1 = sqrtf(4 / (1 + 3 * gamma1 * gamma1));
            float r2 = gamma1 * r1;
            r1 *= 0.5; r2 *= 0.5;
            const float inner[4] = {2.5f, 1.5f, 3.5f, 0.5f};
            const float outer[12] = {8.5f, 3.5f, 9.5f, 2.5f, 6.5f, 5.5f, 11.5f, 0.5f, 7.5f, 4.5f, 10.5f, 1.5f};
            for (int i = 0; i < 4; ++i) pts[i] = scale(polar(r1, 4, inner[i]), amp);
            for (int i = 0; i < 12; ++i) pts[4 + i] = scale(polar(r2, 12, outer[i]), amp);
        } else {
            states = 32; bits = 5; amp = 100; sca = 1; prescale = 0.54;
            float gamma1 = g1 ? g1 : 2.53f, gamma2 = g2 ? g2 : 4.30f;
            float r1 = sqrtf(8 / (1 + 3 * gamma1 * gamma1 + 4 * gamma2 * gamma2));
            float r2 = gamma1 * r1, r3 = gamma2 * r1;
            r1 *= 0.5; r2 *= 0.5; r3 *= 0.5;
            struct P { int ring, n; float i; };
            const P tab[32] = {{3, 16, 10}, {3, 16, 8}, {3, 16, 5}, {3, 16, 7}, {3, 16, 13}, {3, 16, 15}, {3, 16, 2}, {3, 16, 0},
                               {1, 4, 2.5f}, {2, 12, 6.5f}, {1, 4, 1.5f}, {2, 12, 5.5f}, {1, 4, 3.5f}, {2, 12, 11.5f}, {1, 4, 0.5f}, {2, 12, 0.5f},
                               {3, 16, 11}, {3, 16, 9}, {3, 16, 4}, {3, 16, 6}, {3, 16, 12}, {3, 16, 14}, {3, 16, 3}, {3, 16, 1},
                               {2, 12, 8.5f}, {2, 12, 7.5f}, {2, 12, 3.5f}, {2, 12, 4.5f}, {2, 12, 9.5f}, {2, 12, 10.5f}, {2, 12, 2.5f}, {2, 12, 1.5f}};
            for (int i = 0; i < 32; ++i) {
                float r = tab[i].ring == 1 ? r1 : (tab[i].ring == 2 ? r2 : r3);
                pts[i] = scale(polar(r, tab[i].n, tab[i].i), amp);
            }
        }
    }
    static int8_t clampv(float x) { return dvbs2m::llr_clamp_det(x); }   // constellation.cpp:263-270
    void soft_calc(cf32 sample, int8_t* bits_out, float* phase_err) const {   // constellation.cpp:205-261
        float tmp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (amp != 1) sample = scale(sample, amp);
        if (prescale != 1) sample = scale(sample, prescale);
        float min_dist = std::numeric_limits<float>::max();
        cf32 closest{0, 0};
        for (int i = 0; i < states; i++) {
            float dre = sample.re - pts[i].re, dim = sample.im - pts[i].im;
            float dist = sqrtf(dre * dre + dim * dim);
            if (dist < min_dist) { min_dist = dist; closest = pts[i]; }
            float d = dvbs2m::expf_det(-dist / 1.0f);
            for (int j = 0; j < bits; j++) {
                if (((i >> j) & 1) == 0) tmp[2 * j + 0] += d;
                else tmp[2 * j + 1] += d;
            }
        }
        for (int i = 0; i < bits; i++) bits_out[bits - 1 - i] = clampv((dvbs2m::logf_det(tmp[2 * i + 1]) - dvbs2m::logf_det(tmp[2 * i + 0])) * sca);
        // sample * conj(closest)
        float pre = sample.re * closest.re - sample.im * (-closest.im), pim = sample.im * closest.re + sample.re * (-closest.im);
        *phase_err = dvbs2m::atan2f_det(pim, pre);
    }
};

int get_rx_tables(dvbs2gpu_ctx* ctx) {
    std::lock_guard<std::mutex> l(ctx->mtx);
    if (ctx->d_gardner_bank) return 0;       // (set last: a failed upload below leaves the tables "not built")
    int rc;
    // SOF / PLSC / Gold sequence (s2_defs.h:15-80, s2_scrambling.cpp:9-28)
    std::vector<cf32> sof(26), plsc(128 * 64);
    std::vector<uint64_t> codes(128);
    const uint32_t VALUE = 0x18d2e82;
    for (int s = 0; s < 26; ++s) {
        int bit = (VALUE >> (25 - s)) & 1, angle = bit * 2 + (s & 1);
        dvbs2m::sincosf_det((float)(M_PI / 4 + 2 * M_PI * angle / 4), &sof[s].im, &sof[s].re);
    }
    const uint32_t G[6] = {0x55555555, 0x33333333, 0x0f0f0f0f, 0x00ff00ff, 0x0000ffff, 0xffffffff};
    const uint64_t SCR = 0x719d83c953422dfaull;
    for (int index = 0; index < 128; ++index) {
        uint32_t y = 0;
        for (int row = 0; row < 6; ++row)
            if ((index >> (6 - row)) & 1) y ^= G[row];
        uint64_t code = 0;
        for (int bit = 31; bit >= 0; --bit) {
            int yi = (y >> bit) & 1;
            code = (code << 2) | ((uint64_t)yi << 1) | (uint64_t)((index & 1) ? (yi ^ 1) : yi);
        }
        code ^= SCR;
        codes[index] = code;
        for (int i = 0; i < 64; ++i) {
            int yi = (code >> (63 - i)) & 1, nyi = yi ^ (i & 1);
            plsc[index * 64 + i].re = (1 - 2 * nyi) / sqrtf(2);
            plsc[index * 64 + i].im = (1 - 2 * yi) / sqrtf(2);
        }
    }
    std::vector<uint8_t> rn(131072, 0);
    auto lfsr_x = [](uint32_t X) { int bit = ((X >> 7) ^ X) & 1; return ((uint32_t)(bit << 18) | X) >> 1; };
    auto lfsr_y = [](uint32_t Y) { int bit = ((Y >> 10) ^ (Y >> 7) ^ (Y >> 5) ^ Y) & 1; return ((uint32_t)(bit << 18) | Y) >> 1; };
    uint32_t stx = 0x00001, sty = 0x3ffff;
    for (int i = 0; i < 131072; ++i) { rn[i] = (uint8_t)((stx ^ sty) & 1); stx = lfsr_x(stx); sty = lfsr_y(sty); }
    for (int i = 0; i < 131072; ++i) { rn[i] |= (uint8_t)(((stx ^ sty) & 1) << 1); stx = lfsr_x(stx); sty = lfsr_y(sty); }
    cf32 *d_sof = nullptr, *d_plsc = nullptr; uint64_t* d_codes = nullptr; uint8_t* d_rn = nullptr; float* d_bank = nullptr;
    if ((rc = upload(sof, &d_sof)) || (rc = upload(plsc, &d_plsc)) || (rc = upload(codes, &d_codes)) || (rc = upload(rn, &d_rn)) ||
        (rc = upload(make_gardner_bank(), &d_bank))) {
        (void)hipFree(d_sof); (void)hipFree(d_plsc); (void)hipFree(d_codes); (void)hipFree(d_rn); (void)hipFree(d_bank);
        return rc;
    }
    ctx->pl.sof = d_sof; ctx->pl.plsc = d_plsc; ctx->pl.plsc_code = d_codes; ctx->pl.rn = d_rn;
    ctx->d_gardner_bank = d_bank;
    return 0;
}

int get_constel(dvbs2gpu_ctx* ctx, const ModcodParams& mp, ConstelTables** out) {
    std::lock_guard<std::mutex> l(ctx->mtx);
    int key = mp.constel >= C_16APSK ? 100 + mp.modcod : mp.constel;   // PSK tables do not depend on the MODCOD
    auto it = ctx->constel.find(key);
    if (it == ctx->constel.end()) {
        HostConstel H(mp.constel, mp.g1, mp.g2);
        ConstelTables T;
        T.dev.constel = H.constel; T.dev.bits = H.bits; T.dev.states = H.states;
        T.dev.amp = H.amp; T.dev.sca = H.sca; T.dev.prescale = H.prescale;
        for (int i = 0; i < 32; ++i) T.dev.pts[i] = i < H.states ? H.pts[i] : cf32{0, 0};
        {
            std::vector<cf32> pv(T.dev.pts, T.dev.pts + 32);
            int rcp = upload(pv, &T.d_pts);
            if (rcp) return rcp;
            T.dev.pts_g = T.d_pts;
        }
        T.dev.lut_bits = nullptr; T.dev.lut_err = nullptr; T.dev.lut_bits4 = nullptr;
        if (H.bits != 5) {   // make_lut(256), constellation.cpp:272-291 -- built on the host with the shared math definitions, uploaded
            std::vector<int8_t> lb((size_t)65536 * H.bits);
            std::vector<float> le(65536);
            for (int x = 0; x < 256; ++x)
                for (int y = 0; y < 256; ++y) {
                    float xv = (float(x - 128) / float(256)) * 1.5f, yv = (float(y - 128) / float(256)) * 1.5f;
                    H.soft_calc(cf32{xv, yv}, &lb[((size_t)x * 256 + y) * H.bits], &le[(size_t)x * 256 + y]);
                }
            int rc;
            if ((rc = upload(lb, &T.d_bits))) return rc;
            if ((rc = upload(le, &T.d_err))) return rc;
            T.dev.lut_bits = T.d_bits; T.dev.lut_err = T.d_err;
            // the same soft values, one word per cell (the demapper fetches a cell with one load)
            std::vector<uint32_t> lb4(65536, 0u);
            for (size_t cell = 0; cell < 65536; ++cell)
                for (int c = 0; c < H.bits; ++c) lb4[cell] |= (uint32_t)(uint8_t)lb[cell * H.bits + c] << (8 * c);
            if ((rc = upload(lb4, &T.d_bits4))) return rc;
            T.dev.lut_bits4 = T.d_bits4;
        }
        it = ctx->constel.emplace(key, T).first;
    }
    *out = &it->second;
    return 0;
}

}  // namespace
namespace s2 {
int get_rrc(dvbs2gpu_ctx* ctx, int ntaps, float alpha, double Ts, float** out) {
    std::lock_guard<std::mutex> l(ctx->mtx);
    int key = ntaps * 100000 + (int)lround(alpha * 1000) * 10 + (int)lround(Ts);
    auto it = ctx->rrc.find(key);
    if (it == ctx->rrc.end()) {
        float* d;
        int rc = upload(make_rrc_taps(ntaps, alpha, Ts), &d);
        if (rc) return rc;
        it = ctx->rrc.emplace(key, d).first;
    }
    *out = it->second;
    return 0;
}

void critically_damped(float bw, float* alpha, float* beta) {   // SDR++ PhaseControlLoop::criticallyDamped
    double damping = 0.70710678118654752440;
    double den = 1.0 + 2.0 * damping * bw + (double)bw * bw;
    *alpha = (float)((4.0 * damping * bw) / den);
    *beta = (float)((4.0 * (double)bw * bw) / den);
}
}  // namespace s2
namespace {

inline size_t fe_capacity(int n) { return (size_t)2 * n + n / 16 + 256; }

}  // namespace

struct dvbs2gpu_demod {
    dvbs2gpu_ctx* ctx = nullptr;
    dvbs2gpu_demod_cfg cfg{};
    ModcodParams mp{};
    S2LoopCoefs co{};
    int pls_code = 0;
    int max_samples = 0;
    // device
    S2StreamState* d_state = nullptr;
    cf32* d_in = nullptr;        // staging for the host-pointer entry point
    cf32* d_fe = nullptr;        // timing-recovery output + scratch
    cf32* d_fifo[2] = {nullptr, nullptr};
    cf32* d_spec = nullptr;                  // PLL output of the window the frame loops are ahead of the PL sync in (small banks; allocated on first use)
    uint8_t* d_out = nullptr;    // staging for the host-pointer entry point
    int fifo_cap = 0, fifo_cur = 0, fifo_fill = 0;
    // (the PL-sync state machine's state -- pending realign offset, last best_match -- lives in the device-side S2StreamState)
    // results of the last call
    std::vector<S2FrameStats> stats;
    std::vector<const cf32*> frame_ptrs;     // aligned frames of the last call (device pointers into the old FIFO buffer)
    std::vector<long long> frame_pos;        // index of their first symbol in the stream's 1-sps symbol sequence since the last reset
    long long sym_base = 0;                  // symbols that have left the FIFO since the last reset
    int tap_sym_off = 0, tap_sym_cnt = 0, tap_fifo = 0;
    const cf32* tap_pll = nullptr;           // into ctx workspace, valid until the next call on this context
    int tap_pll_stride = 0;                  // elements between the frames behind tap_pll (0: one frame behind the other; mixed batches keep every stream's frames in slots of the batch's longest PLFRAME)
    const int8_t* tap_llr = nullptr;
    std::vector<int> frame_len;              // ACM/VCM: PLFRAME length of each frame of the last call (CCM: empty = mp.plframe)
    long long tap_pll_count = -1, tap_llr_count = -1;   // ACM/VCM: element counts of the taps (frames differ in size)
    float nco_freq_host = 0.f;
};

namespace {

constexpr int S2_SMALL_BANK = 256;       // streams up to which a batch is treated as a set of latency chains (stage pipeline always, groups of a mixed batch on their own)

int demod_configure(dvbs2gpu_demod* d) {
    const dvbs2gpu_demod_cfg& c = d->cfg;
    if (!modcod_params(c.modcod, c.shortframes, c.pilots, &d->mp)) { last_error() = "unsupported MODCOD"; return DVBS2GPU_ERR_MODCOD; }
    if (c.rrc_taps < 1 || c.rrc_taps > RRC_MAX_TAPS) { last_error() = "rrc_taps out of range"; return DVBS2GPU_ERR_ARG; }
    if (!(c.samplerate > 0) || !(c.symbolrate > 0)) { last_error() = "samplerate and symbolrate must be positive"; return DVBS2GPU_ERR_ARG; }
    // the Gardner loop (omega = 1 sample per output, module_dvbs2_demod.cpp:49) emits at most count / (1 - omega_rel_limit) samples; the
    // timing-recovery scratch (count + count/16 + 128) and the symbol FIFO are sized for a limit of a few percent (main.cpp:73: 0.02)
    if (!(c.omega_rel_limit >= 0.f) || !(c.omega_rel_limit <= 0.05f)) { last_error() = "omega_rel_limit must be within [0, 0.05]"; return DVBS2GPU_ERR_ARG; }
    // (the kernels evaluate PCL::advance(0) as "freq unchanged" -- exact for finite gains only)
    if (!std::isfinite(c.clock_mu_gain) || !std::isfinite(c.clock_omega_gain) || !std::isfinite(c.agc_rate) || !std::isfinite(c.loop_bw) || !std::isfinite(c.fll_bw)) {
        last_error() = "loop gains must be finite"; return DVBS2GPU_ERR_ARG;
    }
    S2LoopCoefs& co = d->co;
    co.g_prio_duty = 0; co.g_lane_form = 0; co.post_prio = 0;       // (scheduling hints: set per call by the pipelined mode's balancer)
    co.g_form = 0; co.g_cand_skew = 0;            // (context options, set per call)
    co.agc_rate = c.agc_rate;
    co.g_alpha = c.clock_mu_gain; co.g_beta = c.clock_omega_gain;
    co.g_min_freq = (float)(1.0 * (1.0 - c.omega_rel_limit)); co.g_max_freq = (float)(1.0 * (1.0 + c.omega_rel_limit));
    critically_damped(c.loop_bw, &co.pll_alpha, &co.pll_beta);
    co.pll_min_freq = -0.01f * (float)M_PI; co.pll_max_freq = 0.01f * (float)M_PI;
    critically_damped(c.loop_bw * 0.03f, &co.hdr_alpha, &co.hdr_beta);
    co.hdr_min_freq = -1.0f * (float)M_PI; co.hdr_max_freq = 1.0f * (float)M_PI;
    co.fll_bw = c.fll_bw;
    co.rrc_taps = c.rrc_taps;
    co.soft_plsc = c.soft_plsc ? 1 : 0; co.pilot_aided = c.pilot_aided ? 1 : 0;
    d->pls_code = c.modcod << 2 | (c.shortframes ? 2 : 0) | (c.pilots ? 1 : 0);
    return 0;
}

int demod_reset_state(dvbs2gpu_demod* d) {
    S2StreamState st;
    memset(&st, 0, sizeof(st));
    st.agc_gain = 1.0f;
    st.g_freq = 1.0f;
    HIP_TRY(hipMemcpy(d->d_state, &st, sizeof(st), hipMemcpyHostToDevice));
    d->fifo_fill = 0; d->nco_freq_host = 0.f; d->sym_base = 0; d->frame_pos.clear();
    return 0;
}

// Runs one group of streams that share (modcod, shortframes, pilots) and loop coefficients.
// FEC job launched by one pipelined call and delivered by the next
// The FEC job of a pipelined call, collected by the next call (or by a later one, or dropped when the mode is switched off).
// CCM groups: one job = one LDPC code, frames stream-major, every BBFRAME kb bytes.  ACM/VCM groups: one part per LDPC code present in the
// call (frames of a part = pooled frame indices idx[]), BBFRAMEs of different sizes at per-frame byte offsets inside their stream's output.
struct PendingFec {
    int n = 0, nf = 0, kb = 0;
    std::vector<dvbs2gpu_demod*> dm;    // the streams of the call that started the job (its stream indices are positions in this list)
    std::vector<int> first;             // [n + 1] pooled frames per stream
    std::vector<S2FrameStats> hstats;   // [nf]
    std::vector<std::vector<float>> frame_bm;
    const S2FrameRef* d_frames = nullptr;
    const int* d_first = nullptr;
    const uint8_t* d_bb = nullptr;
    const int8_t* d_llr = nullptr;      // (CCM jobs: what the decoder read; kept for dvbs2gpu_debug_last_fec_job)
    int N = 0, rate = -1, shortframe = 0, max_trials = 0, force = 0, slot = -1;
    const int32_t* d_trials = nullptr;  // CCM: [nf]; ACM/VCM: per part trials[cnt] ++ corrections[cnt] at part.to
    const int32_t* d_corr = nullptr;
    // ACM/VCM
    struct Part { int kb, cnt; const uint8_t* d_bb; const int* d_idx; size_t to; };
    bool vcm = false;
    std::vector<Part> parts;
    std::vector<int> frame_off;         // [nf] byte offset of the frame's BBFRAME inside its stream's output, -1: none (dummy PLFRAME)
    std::vector<int> frame_tr, frame_co;   // [nf] index of the frame's trial count / correction count in the job's result array, -1: none
    std::vector<int> stream_bytes;      // [n]
    uint8_t** d_dst = nullptr;          // [nf] device table of the frames' destinations, filled by the delivery
    size_t n_results = 0;               // int32 words in d_trials (ACM/VCM)
    hipEvent_t done = nullptr;          // recorded on the FEC stream behind the job
    hipEvent_t t0 = nullptr;            // ... and in front of it (big CCM jobs: the job's duration, for the partition rule)
    bool on_part = false;               // the job ran on the partition stream
    std::vector<hipEvent_t> done_more;  // (a job whose parts ran on several streams: one event per stream)
};


// FEC jobs of one context share its FEC workspaces: whichever FEC stream a job goes onto, it starts behind the job before it
#ifndef FEC_PART_CUS_N
#define FEC_PART_CUS_N 128
#endif
constexpr int FEC_PART_CUS = FEC_PART_CUS_N;
static int fec_stream_enter(dvbs2gpu_ctx* ctx, hipStream_t sf) {
    if (ctx->fec_last_done && ctx->fec_last_stream && ctx->fec_last_stream != sf) HIP_TRY(hipStreamWaitEvent(sf, ctx->fec_last_done, 0));
    return 0;
}
static void fec_stream_leave(dvbs2gpu_ctx* ctx, hipStream_t sf, hipEvent_t done) { ctx->fec_last_done = done; ctx->fec_last_stream = sf; }
// (hipExtStreamCreateWithCUMask has no flags argument: the stream it makes is a default-flag, BLOCKING stream -- work a host enqueues on the legacy null stream serialises with
// the decoder jobs on it; INTEGRATION.md tells hosts to use explicit non-blocking streams beside the engine.  A driver that refuses the mask must not take the call down -- the
// front end has advanced the streams' state by now: the rule is switched off for the context and the job goes onto the ordinary FEC stream.)
static bool fec_part_stream(dvbs2gpu_ctx* ctx, hipStream_t* out) {
    if (!ctx->fec_part_stream) {
        uint32_t mask[8] = {};
        for (int i = 0; i < FEC_PART_CUS && i < ctx->num_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
        if (hipExtStreamCreateWithCUMask(&ctx->fec_part_stream, 8, mask) != hipSuccess) {
            (void)hipGetLastError();
            ctx->fec_part_stream = nullptr;
            ctx->fec_part = 0; ctx->fec_part_on = false; ctx->fec_part_trend = 0;
            return false;
        }
    }
    *out = ctx->fec_part_stream;
    return true;
}

// the streams of the whole batch a pipelined call works on: a job is collected into the buffers of whichever of ITS streams are part of this batch
// -- in any order, in any configuration group; frames of a stream that has left are dropped (collect them with a zero-count call before it leaves)
struct BatchMap {
    std::unordered_map<const dvbs2gpu_demod*, int> pos;
    uint8_t* const* d_out = nullptr;
    int* out_bytes = nullptr;
    int out_cap = 0;
};

// results of a finished (or finishing) job -> the callers' output buffers and the per-frame stats of its streams.  `wo`: a scratch workspace of the caller's
static int deliver_job(dvbs2gpu_ctx* ctx, PendingFec* job, hipStream_t st, Workspace& wo, const BatchMap& bm) {
    if (!job->vcm && job->slot >= 0 && job->d_llr) {       // (bench.py's decoder self-check reads the job's buffers back through dvbs2gpu_debug_last_fec_job)
        dvbs2gpu_ctx::LastFecJob& lj = ctx->last_fec[job->slot];
        lj.d_llr = job->d_llr; lj.d_bb = job->d_bb; lj.nf = job->nf; lj.n = job->n; lj.N = job->N; lj.kb = job->kb;
        lj.rate = job->rate; lj.shortframe = job->shortframe; lj.max_trials = job->max_trials; lj.force = job->force;
        lj.first = job->first; lj.dm.assign(job->dm.begin(), job->dm.end());
    }
    if (job->done) HIP_TRY(hipStreamWaitEvent(st, job->done, 0));
    for (hipEvent_t e : job->done_more) HIP_TRY(hipStreamWaitEvent(st, e, 0));
    int rc2;
    std::vector<int> where(job->n, -1);
    for (int i = 0; i < job->n; ++i) {
        auto it = bm.pos.find(job->dm[i]);
        if (it != bm.pos.end()) where[i] = it->second;
    }
    std::vector<int32_t> res;
    if (!job->vcm) {
        std::vector<uint8_t*> outs(job->n, nullptr);
        for (int i = 0; i < job->n; ++i) {
            if (where[i] < 0) continue;
            const int bytes = (job->first[i + 1] - job->first[i]) * job->kb;
            if (bytes > bm.out_cap) { last_error() = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
            outs[i] = bm.d_out[where[i]];
            bm.out_bytes[where[i]] = bytes;
        }
        if ((rc2 = wo.ensure(sizeof(uint8_t*) * job->n))) return rc2;
        HIP_TRY(hipMemcpyAsync(wo.p, outs.data(), sizeof(uint8_t*) * job->n, hipMemcpyHostToDevice, st));
        res.resize(2 * (size_t)job->nf);
        HIP_TRY(hipMemcpyAsync(res.data(), job->d_trials, sizeof(int32_t) * job->nf, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(res.data() + job->nf, job->d_corr, sizeof(int32_t) * job->nf, hipMemcpyDeviceToHost, st));
        { StageSpan sp(ctx->timers, ST_DELIVER, st); HIP_TRY(s2_scatter_out2_launch((uint8_t* const*)wo.p, job->d_frames, job->d_first, job->nf, job->kb, job->d_bb, st)); }
        HIP_TRY(hipStreamSynchronize(st));      // (also: `outs` was copied from pageable memory)
        for (int i = 0; i < job->n; ++i) {
            if (where[i] < 0) continue;
            dvbs2gpu_demod* d = job->dm[i];
            d->stats.clear();
            for (int f = job->first[i]; f < job->first[i + 1]; ++f) {
                S2FrameStats s = job->hstats[f];
                s.best_match = job->frame_bm[i][f - job->first[i]];
                s.ldpc_trials = res[f]; s.bch_corr = res[job->nf + f]; s.bbframe_bytes = job->kb;
                d->stats.push_back(s);
            }
        }
        return 0;
    }
    // ACM/VCM: per-frame destinations
    std::vector<uint8_t*> dst(job->nf, nullptr);
    for (int i = 0; i < job->n; ++i) {
        if (where[i] < 0) continue;
        if (job->stream_bytes[i] > bm.out_cap) { last_error() = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
        bm.out_bytes[where[i]] = job->stream_bytes[i];
        for (int f = job->first[i]; f < job->first[i + 1]; ++f)
            if (job->frame_off[f] >= 0) dst[f] = bm.d_out[where[i]] + job->frame_off[f];
    }
    if (job->nf) HIP_TRY(hipMemcpyAsync(job->d_dst, dst.data(), sizeof(uint8_t*) * job->nf, hipMemcpyHostToDevice, st));
    res.resize(job->n_results);
    if (job->n_results) HIP_TRY(hipMemcpyAsync(res.data(), job->d_trials, sizeof(int32_t) * job->n_results, hipMemcpyDeviceToHost, st));
    for (const PendingFec::Part& P : job->parts) {
        StageSpan sp(ctx->timers, ST_DELIVER, st);
        HIP_TRY(s2_vcm_scatter_launch(P.d_idx, P.cnt, P.kb, P.d_bb, job->d_dst, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < job->n; ++i) {
        if (where[i] < 0) continue;
        dvbs2gpu_demod* d = job->dm[i];
        d->stats.clear();
        for (int f = job->first[i]; f < job->first[i + 1]; ++f) {
            S2FrameStats s = job->hstats[f];
            s.ldpc_trials = job->frame_tr[f] >= 0 ? res[job->frame_tr[f]] : 0;
            s.bch_corr = job->frame_co[f] >= 0 ? res[job->frame_co[f]] : 0;
            d->stats.push_back(s);
        }
    }
    return 0;
}

// development aid: DVBS2GPU_HOST_TIMING=1 prints where the HOST spends a call (ms since entry at each mark)
// AGC/NCO + timing recovery of a batch, time-sliced over the caller's stream and its auxiliary stream (ctx.h FeAux, created on first use)
// (round 6 tried HIP stream priorities -- front-end streams at the device's highest queue priority, the FEC stream at its lowest, or only the post-stage streams high:
//  302.8 / 303.2 / 304.0 ms per headline step, nothing: queue priority does not decide which resident kernel's workgroups get a compute unit's free wave slots)
// The data-parallel post stages (RRC, PL-sync walk, demapper) above the decoder's wave priority: where the pipelined mode's balancer has found the FRONT END to be the critical
// path (share >= 5: a decoder-bound configuration whose balancer strays to 4 in its first calls stays below).  At the decoder's own level they took 35-44 / 30 ms per slice beside it instead of 6 / 5, and the AGC slices -- same stream -- waited behind them (headline
// 299.7 -> 285.4 ms per step); a decoder-bound configuration loses to them (config 5's stand-in: 106.7 -> 115.0 ms per step with them always up).
#ifndef S2_POST_PRIO_MIN_DUTY
#define S2_POST_PRIO_MIN_DUTY 5
#endif
#ifndef S2_PRIO_START_DUTY
#define S2_PRIO_START_DUTY 3      // (third part of round 6, headline at --warmup 2 | 5: start 2 -> 274.8, 254.4 | 252.3; 3 -> 253.0, 252.8, 252.5 | 252.7, 252.6; 4 -> 254.8, 255.6 | 256.0: from 4 the first verdicts take it to 5, post stages up, and the dead band keeps it there)
#endif
static int post_prio_wanted(const dvbs2gpu_ctx* ctx) { return ctx->pipeline_fec && ctx->g_prio_duty >= S2_POST_PRIO_MIN_DUTY ? 1 : 0; }
static hipError_t create_stream(dvbs2gpu_ctx*, hipStream_t* out, int) { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }
#ifndef S2_INPUT_EVENT
#define S2_INPUT_EVENT 1      // (A/B switch, timing only: 0 = the throughput mode does not wait for the host's null-stream work -- the race of round 6)
#endif
#ifndef S2_MIN_SLICE_SAMPLES
#define S2_MIN_SLICE_SAMPLES 1024     // a time slice holds at least this many samples per stream (a call of a few thousand samples is not cut into 32 slices of 133 launches)
#endif
static hipError_t frontend_sliced(dvbs2gpu_ctx* ctx, const S2StreamWork* d_work, int n, const S2LoopCoefs& co, hipStream_t st, const S2PostStages* post = nullptr,
                                  bool own_post_stream = false, int* nsub_out = nullptr, int max_count = 0 /* the longest input of the call, samples (0: not known) */) {
    // (measured: beside the decoder of the previous call 4, alone 8; a small bank is a latency chain in either mode: the shorter pipeline fill wins)
    // (banks up to 256 streams: 16 -- one 8PSK stream 32.4 -> 31.6 ms per 4-frame call, 64 x 1 frame 16.1 -> 15.7)
    // (a big bank in the throughput mode whose FRONT END the balancer has found critical -- priority share 4 or more -- is sliced like a synchronous call: plugin's mode 139.9 -> 137.4 ms,
    //  config 2 187 -> 182; 6 or 12 slices cut the frames of a step unevenly: 162 ms)
    const bool fe_critical = ctx->pipeline_fec && ctx->g_prio_auto && ctx->g_prio_duty >= 4;
    int nsub = ctx->fe_slices > 0 ? ctx->fe_slices : (n <= 8 ? 32 : (n <= 256 ? 16 : (ctx->pipeline_fec && !fe_critical ? 4 : 8)));    // (a handful of streams: 32 -- 28.7 -> 27.9 ms per 4-frame call)
    if (nsub > S2_FE_MAX_SLICES) nsub = S2_FE_MAX_SLICES;
    // (SDR++-sized calls, third part of round 6: one stream, 8 192 samples per call, ms per call at 2 / 4 / 8 / 16 / 32 slices: 2.31 / 2.16 / 2.12 / 2.20 / 2.45; 32 768 samples:
    //  7.03 / 5.50 / 5.17 / 4.87 -- about a thousand samples per slice and stream)
    if (max_count > 0 && ctx->fe_slices <= 0) nsub = std::max(1, std::min(nsub, max_count / S2_MIN_SLICE_SAMPLES));
    if (nsub_out) *nsub_out = nsub > 1 ? nsub : 1;
    S2LoopCoefs cc = co;
    cc.g_prio_duty = ctx->pipeline_fec ? ctx->g_prio_duty : 0;
    cc.post_prio = post_prio_wanted(ctx);
    cc.g_lane_form = 0;
    cc.g_form = ctx->gardner_form; cc.g_cand_skew = ctx->gardner_cand_skew;
    dvbs2gpu_ctx::FeAux* fa = nullptr;
    if (nsub > 1) {
        std::lock_guard<std::mutex> l(ctx->mtx);
        fa = &ctx->fe_aux[st];
        if (!fa->aux) {
            hipError_t e = create_stream(ctx, &fa->aux, +1);
            if (e != hipSuccess) return e;
            for (int i = 0; i <= S2_FE_MAX_SLICES; ++i) {
                if ((e = hipEventCreateWithFlags(&fa->ev[i], hipEventDisableTiming)) != hipSuccess) return e;
                if ((e = hipEventCreateWithFlags(&fa->ev2[i], hipEventDisableTiming)) != hipSuccess) return e;
            }
        }
    }
    if (fa && post && own_post_stream && !fa->aux2) {
        std::lock_guard<std::mutex> l(ctx->mtx);
        hipError_t e = create_stream(ctx, &fa->aux2, +1);
        if (e != hipSuccess) return e;
    }
    // (big banks: the frame loops on a third auxiliary stream, s2_frontend_launch)
    const bool loops_own = fa && post && own_post_stream && ctx->stage_loops_stream && n > S2_SMALL_BANK && nsub > 1;
    if (loops_own && !fa->aux3) {
        std::lock_guard<std::mutex> l(ctx->mtx);
        hipError_t e = create_stream(ctx, &fa->aux3, +1);
        if (e != hipSuccess) return e;
        for (hipEvent_t& ev : fa->ev3) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
    }
    return s2_frontend_launch(d_work, n, cc, ctx->d_gardner_bank, st, fa ? fa->aux : nullptr, fa ? fa->ev : nullptr, nsub, post, fa ? fa->ev2 : nullptr,
                              fa && own_post_stream ? fa->aux2 : nullptr, loops_own ? fa->aux3 : nullptr, loops_own ? fa->ev3 : nullptr);
}

struct HostMarks {
    bool on; std::chrono::steady_clock::time_point t0; std::string line;
    explicit HostMarks(bool on_) : on(on_), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        char b[64];
        snprintf(b, sizeof(b), " %s=%.2f", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        line += b;
    }
    ~HostMarks() { if (on) fprintf(stderr, "[dvbs2gpu host]%s\n", line.c_str()); }
};

int process_group(dvbs2gpu_ctx* ctx, dvbs2gpu_demod* const* dm, int n, const cf32* const* d_iq, const int* counts,
                  uint8_t* const* d_out, int out_cap, int* out_bytes, hipStream_t st, bool pipelined, int slot, const int* pre_nsym, bool own_ws, bool deliver_now,
                  const BatchMap* bm = nullptr) {
    HostMarks hm(ctx->host_timing != 0);
    const auto t_entry = std::chrono::steady_clock::now();
    dvbs2gpu_demod* d0 = dm[0];
    Workspace* const W = own_ws ? ctx->ws_grp[slot] : ctx->ws_rx;      // per-call scratch: the group's own set when groups run side by side
    const hipEvent_t ev_llr = own_ws ? ctx->ev_llr_grp[slot] : ctx->ev_llr;
    const ModcodParams& mp = d0->mp;
    const int raw = mp.plframe, kb = mp.fec.kbch / 8, N = mp.fec.N;
    int rc;
    if ((rc = get_rx_tables(ctx))) return rc;
    ConstelTables* CT;
    if ((rc = get_constel(ctx, mp, &CT))) return rc;
    float* d_taps;
    if ((rc = get_rrc(ctx, d0->cfg.rrc_taps, d0->cfg.rrc_alpha, d0->cfg.samplerate / d0->cfg.symbolrate, &d_taps))) return rc;

    // ---- 1,2: front end + RRC
    // small banks (a workgroup per stream in the frame loops): the loops run behind EVERY slice and ahead of the PL sync (s2_frame_loops_kernel);
    // the window they are ahead in keeps its PLL output in a buffer of the stream's own (one PLFRAME of the longest kind)
    const bool loops_ahead = ctx->loops_ahead != 0 && n <= S2_SMALL_BANK && !d0->cfg.pilot_aided && ctx->stage_pipeline_launches <= 0;     // (S2_SMALL_BANK = FL_SMALL_BANK of the kernels)
    std::vector<S2StreamWork> work(n);
    int max_count = 0;
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        if (counts[i] < 0 || counts[i] > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
        if (loops_ahead && !d->d_spec) HIP_TRY(hipMalloc((void**)&d->d_spec, sizeof(cf32) * 33282));
        work[i].in = d_iq[i]; work[i].count = counts[i]; work[i].fe_out = d->d_fe;
        work[i].fifo = d->d_fifo[d->fifo_cur]; work[i].fifo_fill = d->fifo_fill; work[i].st = d->d_state;
        work[i].fifo_next = d->d_fifo[d->fifo_cur ^ 1]; work[i].out = d_out[i]; work[i].spec_out = d->d_spec;
        max_count = std::max(max_count, counts[i]);
        if (!pipelined) d->stats.clear();        // (pipelined: dvbs2gpu_demod_process_batch has cleared them -- another group's thread may be delivering this stream's frames)
        d->frame_ptrs.clear(); d->frame_pos.clear();
    }
    Workspace& ws_work = W[0];
    if ((rc = ws_work.ensure(sizeof(S2StreamWork) * n + sizeof(int) * (n + 1) + sizeof(int) * 4 * n + 64))) return rc;
    S2StreamWork* d_work = (S2StreamWork*)ws_work.p;
    int* d_nsym = (int*)((char*)ws_work.p + sizeof(S2StreamWork) * n + sizeof(int) * (n + 1));   // [n]
    float* d_nco = (float*)(d_nsym + n);                                                          // [n]
    int* d_curfill = (int*)(d_nco + n);                                                           // [2n]
    HIP_TRY(hipMemcpyAsync(d_work, work.data(), sizeof(S2StreamWork) * n, hipMemcpyHostToDevice, st));
    // ---- 3: PL sync.  The 2-state realign machine of S2PLSyncBlock runs on the device, one workgroup per stream walking its windows
    // in order (s2_ccm_walk_kernel); the host only pools the frame tables it gets back (ONE synchronisation for stages 1-3, or 1-4).
    int maxf = 0;
    for (int i = 0; i < n; ++i) maxf = std::max(maxf, dm[i]->fifo_cap / raw + 2);
    Workspace& ws_win = W[1];
    if ((rc = ws_win.ensure(sizeof(S2VcmFound) * (size_t)n * maxf + sizeof(int) * 4 * n + 64))) return rc;
    S2VcmFound* d_found = (S2VcmFound*)ws_win.p;
    int* d_counts = (int*)(d_found + (size_t)n * maxf);
    // Stage pipeline (calls of one configuration): RRC, walk and the frame loops run behind every timing-recovery slice on the auxiliary stream
    // (s2_frontend_launch); a stream's frames stay in its maxf slots of the PLL-output / statistics arrays until the host has pooled the tables.
    // Not while the decoder of the previous call is the critical path anyway (pipelined mode, the balancer has taken the timing loop's priority
    // share to its minimum): there the stages back to back leave the decoder more of the SIMDs (headline: 390 vs 394 ms per step).
    // (stage_pipeline == 2, for the tests: every other call, whatever the mode -- the two flows leave a stream in the same state)
    // (a small bank is a set of latency chains whatever the mode: always staged)
    const bool staged = !pre_nsym && (ctx->stage_pipeline == 2 ? (ctx->stage_calls++ & 1) != 0 :
        ctx->stage_pipeline && (n <= S2_SMALL_BANK || !(pipelined && ctx->g_prio_auto && ctx->g_prio_duty <= ctx->stage_pipeline_min_duty)));
    ctx->last_call_staged = staged;
    Workspace& ws_pll = W[3];
    Workspace& ws_slot = W[7];
    std::vector<S2FrameStats> slot_stats;
    if (staged) {
        const size_t nslot = (size_t)n * maxf;
        if ((rc = ws_pll.ensure(nslot * raw * sizeof(cf32)))) return rc;
        if ((rc = ws_slot.ensure(sizeof(S2FrameStats) * nslot + 64))) return rc;
        struct Spans : S2SliceSpans {
            StageTimers* T; std::unique_ptr<StageSpan> sp[4];
            void begin(int stage, hipStream_t s) override { sp[stage].reset(new StageSpan(*T, stage == 1 ? ST_RRC : (stage == 2 ? ST_PLSYNC : ST_LOOPS), s)); }
            void end(int stage, hipStream_t) override { sp[stage].reset(); }
        } spans;
        spans.T = &ctx->timers;
        // how often the frame loops run inside the call: about once per frame a stream gets per call (a launch costs its longest stream's chain,
        // 11 ms per frame, however few streams have a frame ready)
        int launches = std::min(S2_FE_MAX_SLICES, std::max(1, (max_count / 2) / raw));
        if (ctx->stage_pipeline_launches > 0) launches = ctx->stage_pipeline_launches;
        S2PostStages post{d_taps, d0->cfg.rrc_taps, max_count + max_count / 32 + 8, raw, maxf, d_found, d_counts, ctx->pl, CT->dev, d0->pls_code,
                          mp.slots, mp.pilots, mp.pilot_blocks, (cf32*)ws_pll.p, (S2FrameStats*)ws_slot.p, ctx->timers.on ? &spans : nullptr, launches};
        if (loops_ahead) {
            // behind (nearly) every slice -- but a launch wants a few thousand symbols per stream to chew on: a bank of 64 spends 0.4 ms per
            // launch on top of its symbols
            constexpr int sym_per_launch = 2700;
            post.spec = 1;
            post.loops_launches = std::min(S2_FE_MAX_SLICES, std::max(launches, (max_count / 2) / std::max(sym_per_launch, 1)));
        }
        { StageSpan sp(ctx->timers, ST_FRONTEND, st); HIP_TRY(frontend_sliced(ctx, d_work, n, d0->co, st, &post, ctx->stage_post_stream == 2 || ((!pipelined || n <= S2_SMALL_BANK || (ctx->g_prio_auto && ctx->g_prio_duty >= 4)) && ctx->stage_post_stream), nullptr, max_count)); }   // (a small bank is a set of latency chains in the throughput mode too: its post stages on the AGC's stream made one stream's 4-frame call 39.9 ms instead of 25;
        //  a big bank whose FRONT END the balancer has found critical -- priority share 4 or more: the plugin's mode, QPSK -- likewise: AGC + RRC + walk + frame loops on one
        //  stream were 127 ms of launches per 140 ms step; beside a decoder that is the critical path the shared stream stays: headline 347 vs 361 ms per step)
        slot_stats.resize(nslot);
        HIP_TRY(hipMemcpyAsync(slot_stats.data(), ws_slot.p, sizeof(S2FrameStats) * nslot, hipMemcpyDeviceToHost, st));
    } else {
        if (!pre_nsym) {
            // (with pre_nsym the MODCOD-independent stages already ran for the whole batch: frontend_prepass)
            { StageSpan sp(ctx->timers, ST_FRONTEND, st); HIP_TRY(frontend_sliced(ctx, d_work, n, d0->co, st, nullptr, false, nullptr, max_count)); }
            { StageSpan sp(ctx->timers, ST_RRC, st); HIP_TRY(s2_rrc_decim_launch(d_work, n, max_count + max_count / 32 + 8, d_taps, d0->cfg.rrc_taps, st, post_prio_wanted(ctx))); }
        }
        { StageSpan sp(ctx->timers, ST_PLSYNC, st); HIP_TRY(s2_ccm_walk_launch(d_work, n, raw, maxf, d_found, d_counts, st, post_prio_wanted(ctx))); }
    }
    std::vector<S2VcmFound> found((size_t)n * maxf);
    std::vector<int> cnts(4 * n);
    HIP_TRY(hipMemcpyAsync(cnts.data(), d_counts, sizeof(int) * 4 * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(found.data(), d_found, sizeof(S2VcmFound) * found.size(), hipMemcpyDeviceToHost, st));
    hm.mark("fe_enqueued");
    HIP_TRY(hipStreamSynchronize(st));
    hm.mark("fe_done");
    std::vector<int> cur(n, 0);
    std::vector<std::vector<int>> frame_start(n);
    std::vector<std::vector<float>> frame_bm(n);
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        const int nfi = cnts[4 * i], nsym_i = cnts[4 * i + 3];
        d->tap_sym_off = d->fifo_fill; d->tap_sym_cnt = nsym_i; d->tap_fifo = d->fifo_cur;
        d->fifo_fill = cnts[4 * i + 2];
        if (d->fifo_fill > d->fifo_cap) { last_error() = "symbol FIFO overflow"; return DVBS2GPU_ERR_CAPACITY; }
        cur[i] = cnts[4 * i + 1];
        for (int k = 0; k < nfi; ++k) { frame_start[i].push_back(found[(size_t)i * maxf + k].offset); frame_bm[i].push_back(found[(size_t)i * maxf + k].sofq); }
    }
    hm.mark("plsync_done");
    // ---- 4..6 on the pooled frames
    std::vector<S2FrameRef> frames;
    std::vector<int> first(n + 1, 0), fslot;
    for (int i = 0; i < n; ++i) {
        first[i] = (int)frames.size();
        const cf32* base = dm[i]->d_fifo[dm[i]->fifo_cur];
        int k = 0;
        for (int s : frame_start[i]) {
            frames.push_back(S2FrameRef{base + s, i, 0}); dm[i]->frame_ptrs.push_back(base + s); dm[i]->frame_pos.push_back(dm[i]->sym_base + s);
            fslot.push_back(i * maxf + k++);
        }
    }
    first[n] = (int)frames.size();
    const int nf = (int)frames.size();
    std::vector<S2FrameStats> hstats(nf);
    std::vector<int32_t> trials(nf), corr(nf);
    uint8_t* d_bb = nullptr;
    if (nf > 0) {
        Workspace& ws_fr = W[2];
        if ((rc = ws_fr.ensure(sizeof(S2FrameRef) * nf + sizeof(S2FrameStats) * nf + sizeof(int32_t) * 3 * nf + 64))) return rc;
        S2FrameRef* d_frames = (S2FrameRef*)ws_fr.p;
        S2FrameStats* d_stats = (S2FrameStats*)(d_frames + nf);
        int32_t* d_trials = (int32_t*)(d_stats + nf);
        int32_t* d_corr = d_trials + nf;
        int* d_slot = (int*)(d_corr + nf);
        int* d_first = (int*)((char*)ws_work.p + sizeof(S2StreamWork) * n);
        const int par = ctx->fec_parity[slot];
        Workspace &ws_llr = pipelined ? ctx->ws_fecbuf[slot][par][0] : W[4];
        Workspace &ws_bb = pipelined ? ctx->ws_fecbuf[slot][par][1] : W[5];
        if (pipelined) {
            // the FEC job keeps its own copy of the frame table and its result arrays (phase A of the next call reuses ws_rx[2])
            Workspace& wj = ctx->ws_fecbuf[slot][par][2];
            if ((rc = wj.ensure(sizeof(S2FrameRef) * nf + sizeof(int) * (n + 1) + sizeof(int32_t) * 2 * nf + 64))) return rc;
            d_trials = (int32_t*)((char*)wj.p + sizeof(S2FrameRef) * nf + sizeof(int) * (n + 1));
            d_corr = d_trials + nf;
            HIP_TRY(hipMemcpyAsync(wj.p, frames.data(), sizeof(S2FrameRef) * nf, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync((char*)wj.p + sizeof(S2FrameRef) * nf, first.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, st));
        }
        if (!staged && (rc = ws_pll.ensure((size_t)nf * raw * sizeof(cf32)))) return rc;
        if ((rc = ws_llr.ensure((size_t)nf * N))) return rc;
        if ((rc = ws_bb.ensure((size_t)nf * kb))) return rc;
        cf32* d_pll = (cf32*)ws_pll.p;
        int8_t* d_llr = (int8_t*)ws_llr.p;
        d_bb = (uint8_t*)ws_bb.p;
        HIP_TRY(hipMemcpyAsync(d_frames, frames.data(), sizeof(S2FrameRef) * nf, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_first, first.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, st));
        if (staged) {
            HIP_TRY(hipMemcpyAsync(d_slot, fslot.data(), sizeof(int) * nf, hipMemcpyHostToDevice, st));
            for (int f = 0; f < nf; ++f) hstats[f] = slot_stats[fslot[f]];
        } else {
            StageSpan sp(ctx->timers, ST_LOOPS, st);
            S2LoopCoefs lc = d0->co;
            lc.g_prio_duty = pipelined ? ctx->g_prio_duty : 0;      // (scheduling only: the loops' wave priority, see the kernel)
            HIP_TRY(s2_frame_loops_launch(d_work, n, d_frames, d_first, lc, ctx->pl, CT->dev, d0->pls_code, mp.slots, mp.pilots,
                                          mp.pilot_blocks, raw, d_pll, d_stats, st));
        }
        { StageSpan sp(ctx->timers, ST_DEMAP, st); HIP_TRY(s2_demap_launch(CT->dev, mp.rate, mp.shortframe, mp.slots, mp.pilots, raw, d_pll, nf, d_llr, N, st, staged ? d_slot : nullptr, post_prio_wanted(ctx))); }
        const int force = d0->cfg.force_ldpc_iters > 0;
        const int mt = force ? d0->cfg.force_ldpc_iters : d0->cfg.max_ldpc_trials;
        // keep a copy of the demapper output for the tap before LDPC consumes it? LDPC does not modify d_llr.
        if (!pipelined) {
            if ((rc = fec_run(ctx, mp.fec, d_llr, nf, mt, force, d_bb, d_trials, d_corr, st))) return rc;
            HIP_TRY(hipMemcpyAsync(trials.data(), d_trials, sizeof(int32_t) * nf, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(corr.data(), d_corr, sizeof(int32_t) * nf, hipMemcpyDeviceToHost, st));
        } else {
            HIP_TRY(hipEventRecord(ev_llr, st));
        }
        if (!staged) HIP_TRY(hipMemcpyAsync(hstats.data(), d_stats, sizeof(S2FrameStats) * nf, hipMemcpyDeviceToHost, st));
        for (int i = 0; i < n; ++i) {
            int cnt = first[i + 1] - first[i];
            dm[i]->tap_pll = d_pll + (size_t)(staged ? (size_t)i * maxf : first[i]) * raw;
            dm[i]->tap_pll_stride = 0;
            dm[i]->tap_llr = d_llr + (size_t)first[i] * N;
            int bytes = cnt * kb;
            if (bytes > out_cap) { last_error() = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
            out_bytes[i] = bytes;
        }
        if (!pipelined) { StageSpan sp(ctx->timers, ST_DELIVER, st); HIP_TRY(s2_scatter_out_launch(d_work, d_frames, d_first, nf, kb, d_bb, st)); }
    } else {
        for (int i = 0; i < n; ++i) { out_bytes[i] = 0; dm[i]->tap_pll = nullptr; dm[i]->tap_llr = nullptr; }
    }
    hm.mark("loops_enqueued");
    // ---- 7: FIFO remainder to the spare buffer, NCO frequency for the getter
    std::vector<int> curfill(2 * n);
    for (int i = 0; i < n; ++i) { curfill[2 * i] = cur[i]; curfill[2 * i + 1] = dm[i]->fifo_fill; }
    HIP_TRY(hipMemcpyAsync(d_curfill, curfill.data(), sizeof(int) * 2 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(s2_fifo_compact_launch(d_work, n, d_curfill, st));
    std::vector<float> nco(n);
    HIP_TRY(s2_collect_launch(d_work, n, d_nsym, d_nco, st));
    HIP_TRY(hipMemcpyAsync(nco.data(), d_nco, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    hm.mark("frontend_all_done");
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        if (cur[i] > 0) { d->fifo_fill -= cur[i]; d->fifo_cur ^= 1; d->sym_base += cur[i]; }
        d->nco_freq_host = nco[i];
    }
    if (!pipelined) {
        for (int i = 0; i < n; ++i) {
            for (int f = first[i]; f < first[i + 1]; ++f) {
                S2FrameStats s = hstats[f];
                s.best_match = frame_bm[i][f - first[i]];
                s.ldpc_trials = trials[f]; s.bch_corr = corr[f]; s.bbframe_bytes = kb;
                dm[i]->stats.push_back(s);
            }
        }
        return 0;
    }
    // ---- pipelined: deliver this group's job of the previous call into this call's output buffers, then start this call's job.
    // Jobs of all groups run in order on the FEC stream; each is followed by its own event, and the delivery (scatter into the
    // caller's buffers) runs on the front-end stream behind that event, so it never queues behind a later group's decoder.
    hipStream_t sf = ctx->fec_stream;
    // (out_bytes of a pipelined call are written by the deliveries alone, through the batch map: the job a slot holds may belong to other streams of the batch)
    BatchMap bm_local;
    if (!bm) {
        for (int i = 0; i < n; ++i) { bm_local.pos[dm[i]] = i; out_bytes[i] = 0; dm[i]->stats.clear(); }
        bm_local.d_out = d_out; bm_local.out_bytes = out_bytes; bm_local.out_cap = out_cap;
        bm = &bm_local;
    }
    auto deliver = [&](PendingFec* job) -> int { return deliver_job(ctx, job, st, W[6], *bm); };
    PendingFec* prev = (PendingFec*)ctx->pending_fec[slot];
    ctx->pending_fec[slot] = nullptr;
    std::unique_ptr<PendingFec> prev_guard(prev);
    // This call's job goes onto the FEC stream BEFORE the previous call's job is delivered: the stream then runs decoder launch after decoder
    // launch with no host in between (delivered first, a headline step left the FEC stream idle for ~5 ms: the scatter, its copies and the
    // synchronisation of the delivery).  The two jobs use the two halves of the double-buffered FEC workspaces and an event each.
    std::unique_ptr<PendingFec> started;
    // (a small job that runs on the group's own stream -- see below -- goes behind the previous job's delivery instead: that delivery ends with a
    //  synchronisation of this very stream)
    const bool beside = own_ws && st && nf <= ctx->num_cus;
    bool prev_delivered = false;
    if (beside && prev) {
        if ((rc = deliver(prev))) return rc;
        hm.mark("prev_delivered");
        prev_delivered = true;
    }
    if (nf > 0) {
        const int par = ctx->fec_parity[slot];
        Workspace& wj = ctx->ws_fecbuf[slot][par][2];
        S2FrameRef* j_frames = (S2FrameRef*)wj.p;
        int* j_first = (int*)((char*)wj.p + sizeof(S2FrameRef) * nf);
        int32_t* j_trials = (int32_t*)((char*)wj.p + sizeof(S2FrameRef) * nf + sizeof(int) * (n + 1));
        int32_t* j_corr = j_trials + nf;
        auto job = std::make_unique<PendingFec>();
        job->n = n; job->nf = nf; job->kb = kb;
        job->dm.assign(dm, dm + n);
        job->first = first; job->hstats = hstats; job->frame_bm = frame_bm;
        job->d_frames = j_frames; job->d_first = j_first; job->d_bb = (const uint8_t*)ctx->ws_fecbuf[slot][par][1].p;
        job->d_trials = j_trials; job->d_corr = j_corr;
        const int force = d0->cfg.force_ldpc_iters > 0;
        const int mt = force ? d0->cfg.force_ldpc_iters : d0->cfg.max_ldpc_trials;
        job->d_llr = (const int8_t*)ctx->ws_fecbuf[slot][par][0].p; job->N = N; job->rate = mp.rate; job->shortframe = mp.shortframe; job->max_trials = mt; job->force = force; job->slot = slot;
        // A group's job that cannot fill the device (fewer decoder workgroups than half the CUs: the groups of a 64-transponder batch -- a
        // handful of workgroups and 4-6 ms of decoder LATENCY each) runs beside the other groups' jobs: on the GROUP's stream, behind its
        // demapper, with the group's own FEC workspaces (the group's next call enqueues its stages a whole front-end pass later, so nothing
        // waits behind the job; more streams would only share the few hardware queues).  Big jobs queue up on the shared FEC stream as
        // before (two persistent decoders would only take turns).
        if (beside) {
            if ((rc = fec_run(ctx, mp.fec, (const int8_t*)ctx->ws_fecbuf[slot][par][0].p, nf, mt, force, (uint8_t*)ctx->ws_fecbuf[slot][par][1].p, j_trials, j_corr, st,
                              &ctx->fws_grp[slot])))
                return rc;
            if (!ctx->ev_fec[slot][par]) HIP_TRY(hipEventCreate(&ctx->ev_fec[slot][par]));
            HIP_TRY(hipEventRecord(ctx->ev_fec[slot][par], st));
            job->done = ctx->ev_fec[slot][par];
        } else {
            std::lock_guard<std::mutex> fl(ctx->fec_mtx);
            // (a single-configuration batch whose decoder has room to spare: onto the partition stream, ctx.h)
            if (ctx->fec_part_on && !own_ws && !pre_nsym) {
                job->on_part = fec_part_stream(ctx, &sf);
            }
            if ((rc = fec_stream_enter(ctx, sf))) return rc;
            HIP_TRY(hipStreamWaitEvent(sf, ev_llr, 0));
            if (!ctx->ev_fec_t0[slot][par]) HIP_TRY(hipEventCreate(&ctx->ev_fec_t0[slot][par]));
            HIP_TRY(hipEventRecord(ctx->ev_fec_t0[slot][par], sf));
            if ((rc = fec_run(ctx, mp.fec, (const int8_t*)ctx->ws_fecbuf[slot][par][0].p, nf, mt, force, (uint8_t*)ctx->ws_fecbuf[slot][par][1].p, j_trials, j_corr, sf)))
                return rc;
            if (!ctx->ev_fec[slot][par]) HIP_TRY(hipEventCreate(&ctx->ev_fec[slot][par]));
            HIP_TRY(hipEventRecord(ctx->ev_fec[slot][par], sf));
            job->done = ctx->ev_fec[slot][par];
            job->t0 = ctx->ev_fec_t0[slot][par];
            fec_stream_leave(ctx, sf, job->done);
        }
        ctx->fec_parity[slot] ^= 1;
        hm.mark("fec_enqueued");
        started = std::move(job);
    }
    if (prev && !prev_delivered) {
        // (asked BEFORE the delivery: had the previous call's FEC job already ended when this call's front end was through?  The delivery itself takes
        //  0.3-0.6 ms with the job long done -- copies, scatter, synchronisation --, which a fixed threshold on the waiting time mistook for "just in time")
        const bool job_was_done = prev->done && hipEventQuery(prev->done) == hipSuccess;
        const auto t_d0 = std::chrono::steady_clock::now();
        if ((rc = deliver(prev))) {               // FEC of the previous call (ran during this call's front end)
            // this call's job is already on the device: it stays parked for the next call (or reset) to collect -- dropping it here would free
            // buffers its kernels are still using and lose the call's frames behind the previous job's error
            if (started && !deliver_now) ctx->pending_fec[slot] = started.release();
            else if (started) (void)hipDeviceSynchronize();     // (a job that would have been collected by this very call: let it end before its buffers go)
            return rc;
        }
        hm.mark("prev_delivered");
        // Balance of the two streams: did this call have to wait for the previous call's FEC job (the decoder is the critical path: the timing
        // loop should yield more) or was the job long done (the front end is: it should yield less)?  One step of the priority duty per two
        // consistent calls; single-group batches only (the groups of a mixed batch share one front-end pass).
        if (ctx->g_prio_auto && !own_ws && !pre_nsym) {
            // (the balance belongs to a configuration: what one batch settled at says nothing about the next one's -- a front-end-bound configuration that inherited the
            //  headline's setting spent part of its steps in the wrong flow)
            const long long sig = ((long long)n << 32) ^ ((long long)d0->cfg.modcod << 20) ^ ((long long)d0->cfg.shortframes << 19) ^ ((long long)d0->cfg.pilots << 18) ^
                                  ((long long)(d0->cfg.force_ldpc_iters & 0xff) << 8) ^ (long long)(d0->cfg.max_ldpc_trials & 0xff);
            const bool sig_changed = sig != ctx->g_prio_sig;
            // (a new configuration starts at share S2_PRIO_START_DUTY -- from 0 the headline's first four calls ran 55 ms long each, the front end being their critical path, and the
            //  plugin's mode needed 14 calls to its share of 7; rounds 5-6 started at 2, where the decoder-bound configurations settled then; since the timing recovery's producers
            //  run above the decoder they settle at 4: headline 4, config 5's stand-in 4, config 2 5, plugin's mode 7)
            if (sig_changed) { ctx->g_prio_sig = sig; ctx->g_prio_duty = std::min(ctx->g_prio_cap, S2_PRIO_START_DUTY); ctx->g_prio_trend = 0; ctx->g_prio_hold = 0; ctx->g_prio_last_down = 0; }
            const auto t_d1 = std::chrono::steady_clock::now();
            const double wait_ms = std::chrono::duration<double, std::milli>(t_d1 - t_d0).count();
            const double call_ms = std::chrono::duration<double, std::milli>(t_d1 - t_entry).count();
            // the job just delivered: how long did it run (timing events around it), and how far apart do this batch's calls come?
            float job_ms = 0.f;
            const double period_ms = std::chrono::duration<double, std::milli>(t_entry - ctx->fec_last_entry).count();
            ctx->fec_last_entry = t_entry;
            const bool timed = prev->t0 && prev->done && hipEventElapsedTime(&job_ms, prev->t0, prev->done) == hipSuccess && period_ms > 0.0 && period_ms < 5000.0;
            const int verdict = job_was_done ? +1 : (wait_ms > 1.0 + 0.035 * call_ms ? -1 : 0);     // (3.5 %: the headline waits 2.2 % of its call at its best share, 2.9 % in the first calls of a run)
            // a clear case moves the share at once: the job took less than 0.8 of the call (the front end is the critical path by a wide margin) / the call waited more
            // than a twentieth of its time for the job; anything else needs two consistent calls
            const bool clear = verdict > 0 ? (timed && job_ms < 0.8 * call_ms) : (verdict < 0 && wait_ms > 0.05 * call_ms);
            // (no see-saw: a step down that the very next verdicts take back -- the lower share made the front end the critical path -- is not tried again for 64 calls;
            //  the headline sat at share 2 with a wait right at the threshold and dipped to 1 every dozen calls, one 380 ms call each time)
            if (ctx->g_prio_hold > 0) --ctx->g_prio_hold;
            const bool held = verdict < 0 && ctx->g_prio_hold > 0;
            if (verdict != 0 && !held && (clear || verdict == ctx->g_prio_trend)) {
                const int before = ctx->g_prio_duty;
                ctx->g_prio_duty = std::min(ctx->g_prio_cap, std::max(0, ctx->g_prio_duty + verdict));
                if (verdict > 0 && ctx->g_prio_last_down > 0 && ctx->g_prio_duty != before) ctx->g_prio_hold = 64;     // (up again right behind a step down)
                ctx->g_prio_last_down = (verdict < 0 && ctx->g_prio_duty != before) ? 4 : 0;
                ctx->g_prio_trend = 0;
            } else {
                ctx->g_prio_trend = held ? 0 : verdict;
                if (ctx->g_prio_last_down > 0) --ctx->g_prio_last_down;
            }
            // the partition rule (ctx.h)
            if (ctx->fec_part < 0 && timed) {
                if (sig_changed) { ctx->fec_part_on = false; ctx->fec_part_trend = 0; }
                const double scale = prev->on_part ? 1.0 : 256.0 / FEC_PART_CUS;
                const bool fits = job_was_done && job_ms * scale * 1.15 < period_ms;
                const int want = fits ? +1 : -1;
                if ((want > 0) != ctx->fec_part_on) {
                    if (ctx->fec_part_trend == want) { ctx->fec_part_on = want > 0; ctx->fec_part_trend = 0; }
                    else ctx->fec_part_trend = want;
                } else ctx->fec_part_trend = 0;
            }
            if (hm.on) { char b[128]; snprintf(b, sizeof(b), " wait=%.2f call=%.1f duty=%d job=%.1f period=%.1f part=%d", wait_ms, call_ms, ctx->g_prio_duty, job_ms, period_ms, (int)ctx->fec_part_on); hm.line += b; }
        }
    }
    if (started) {
        if (deliver_now) {
            // synchronous call with several groups side by side: the job is collected by the call that started it
            if ((rc = deliver(started.get()))) return rc;
        } else {
            ctx->pending_fec[slot] = started.release();
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------- ACM/VCM mode
// device tables: what each of the 128 PLS codes means and the constellation (LUTs) it uses
int get_vcm_tables(dvbs2gpu_ctx* ctx) {
    {
        std::lock_guard<std::mutex> l(ctx->mtx);
        if (ctx->d_vcm_mods) return 0;
    }
    std::vector<S2VcmMod> mods(128);
    std::vector<FecParams> fec(128);
    std::vector<S2ConstelDev> cons;
    std::map<const ConstelTables*, int> con_index;
    for (int pls = 0; pls < 128; ++pls) {
        S2VcmMod& M = mods[pls];
        memset(&M, 0, sizeof(M));
        const int modcod = pls >> 2;
        if (modcod == 0) { M.valid = 2; M.plframe = VCM_DUMMY_PLFRAME; M.slots = 36; continue; }
        ModcodParams mp;
        if (!modcod_params(modcod, (pls >> 1) & 1, pls & 1, &mp)) continue;
        ConstelTables* CT;
        int rc = get_constel(ctx, mp, &CT);
        if (rc) return rc;
        auto it = con_index.find(CT);
        if (it == con_index.end()) { it = con_index.emplace(CT, (int)cons.size()).first; cons.push_back(CT->dev); }
        M.valid = 1; M.plframe = mp.plframe; M.slots = mp.slots; M.pilots = mp.pilots; M.pilot_blocks = mp.pilot_blocks; M.bits = mp.bits;
        M.rate = mp.rate; M.constel = mp.constel; M.N = mp.fec.N; M.kb = mp.fec.kbch / 8; M.con = it->second; M.code_index = mp.fec.code_index;
        fec[pls] = mp.fec;
    }
    std::lock_guard<std::mutex> l(ctx->mtx);
    if (ctx->d_vcm_mods) return 0;
    S2VcmMod* dm = nullptr;
    S2ConstelDev* dc = nullptr;
    int rc;
    if ((rc = upload(mods, &dm)) || (rc = upload(cons, &dc))) { (void)hipFree(dm); (void)hipFree(dc); return rc; }
    ctx->h_vcm_mods = mods; ctx->h_vcm_fec = fec;
    ctx->d_vcm_cons = dc;
    ctx->d_vcm_mods = dm;
    return 0;
}

// One group of ACM/VCM streams (same loop settings), synchronous: front end -> framing walk on the device -> [frame tables to the host:
// pooling, grouping by LDPC code] -> per-frame-MODCOD loops + demapper -> one FEC job per code present -> BBFRAMEs of differing size
// into the streams' output buffers in frame order.
int process_vcm_group(dvbs2gpu_ctx* ctx, dvbs2gpu_demod* const* dm, int n, const cf32* const* d_iq, const int* counts, uint8_t* const* d_out,
                      int out_cap, int* out_bytes, hipStream_t st, bool pipelined = false, int slot = 0, const BatchMap* bm = nullptr) {
    dvbs2gpu_demod* d0 = dm[0];
    int rc;
    if ((rc = get_rx_tables(ctx)) || (rc = get_vcm_tables(ctx))) return rc;
    float* d_taps;
    if ((rc = get_rrc(ctx, d0->cfg.rrc_taps, d0->cfg.rrc_alpha, d0->cfg.samplerate / d0->cfg.symbolrate, &d_taps))) return rc;
    Workspace* const W = ctx->ws_vcm;
    std::vector<S2StreamWork> work(n);
    int max_count = 0, maxf = 0;
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        if (counts[i] < 0 || counts[i] > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
        work[i].in = d_iq[i]; work[i].count = counts[i]; work[i].fe_out = d->d_fe;
        work[i].fifo = d->d_fifo[d->fifo_cur]; work[i].fifo_fill = d->fifo_fill; work[i].st = d->d_state;
        work[i].fifo_next = d->d_fifo[d->fifo_cur ^ 1]; work[i].out = d_out[i]; work[i].spec_out = d->d_spec;
        max_count = std::max(max_count, counts[i]);
        maxf = std::max(maxf, d->fifo_cap / VCM_DUMMY_PLFRAME + 2);
        if (!pipelined) d->stats.clear();
        d->frame_ptrs.clear(); d->frame_pos.clear(); d->frame_len.clear();
        d->tap_pll = nullptr; d->tap_llr = nullptr; d->tap_pll_count = 0; d->tap_llr_count = 0; d->tap_pll_stride = 0;
    }
    if ((rc = W[0].ensure(sizeof(S2StreamWork) * n + sizeof(int) * (n + 1) + sizeof(int) * 8 * n + sizeof(float) * n + 64))) return rc;
    S2StreamWork* d_work = (S2StreamWork*)W[0].p;
    int* d_first = (int*)((char*)W[0].p + sizeof(S2StreamWork) * n);
    int* d_counts = d_first + (n + 1);            // [4n]
    int* d_curfill = d_counts + 4 * n;            // [2n]
    int* d_nsym = d_curfill + 2 * n;              // [n]
    float* d_nco = (float*)(d_nsym + n);          // [n]
    if ((rc = W[1].ensure(sizeof(S2VcmFound) * (size_t)n * maxf))) return rc;
    S2VcmFound* d_found = (S2VcmFound*)W[1].p;
    HIP_TRY(hipMemcpyAsync(d_work, work.data(), sizeof(S2StreamWork) * n, hipMemcpyHostToDevice, st));
    { StageSpan sp(ctx->timers, ST_FRONTEND, st); HIP_TRY(frontend_sliced(ctx, d_work, n, d0->co, st)); }
    { StageSpan sp(ctx->timers, ST_RRC, st); HIP_TRY(s2_rrc_decim_launch(d_work, n, max_count + max_count / 32 + 8, d_taps, d0->cfg.rrc_taps, st)); }
    { StageSpan sp(ctx->timers, ST_PLSYNC, st); HIP_TRY(s2_vcm_walk_launch(d_work, n, ctx->pl, ctx->d_vcm_mods, d0->cfg.sof_threshold, maxf, d_found, d_counts, st)); }
    std::vector<S2VcmFound> found((size_t)n * maxf);
    std::vector<int> cnts(4 * n);
    HIP_TRY(hipMemcpyAsync(cnts.data(), d_counts, sizeof(int) * 4 * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(found.data(), d_found, sizeof(S2VcmFound) * found.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // ---- pool the frames (stream-major, stream order), group them by LDPC code, lay out PLL output / LLRs / output bytes
    std::vector<S2VcmFrame> frames;
    std::vector<int> first(n + 1, 0);
    std::vector<uint8_t*> dst;                     // per pooled frame: where its BBFRAME goes
    std::vector<int> frame_off, stream_bytes(n, 0);   // (pipelined: the same as offsets -- the buffers are those of the call that collects the job)
    std::map<int, std::vector<int>> groups;        // PLS-independent FEC identity (code index * 2 + frame size is implied by the index) -> pooled frame indices
    std::map<int, int> group_pls;                  // a PLS code of the group (its FEC parameters)
    long long pll_off = 0, llr_off = 0;
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        const int nf = cnts[4 * i], avail = cnts[4 * i + 2], nsym = cnts[4 * i + 3];
        first[i] = (int)frames.size();
        d->tap_sym_off = d->fifo_fill; d->tap_sym_cnt = nsym; d->tap_fifo = d->fifo_cur;
        d->fifo_fill = avail;
        if (d->fifo_fill > d->fifo_cap) { last_error() = "symbol FIFO overflow"; return DVBS2GPU_ERR_CAPACITY; }
        const cf32* base = d->d_fifo[d->fifo_cur];
        int obytes = 0;
        for (int k = 0; k < nf; ++k) {
            const S2VcmFound& F = found[(size_t)i * maxf + k];
            const S2VcmMod& M = ctx->h_vcm_mods[F.pls];
            S2VcmFrame fr;
            fr.sym = base + F.offset; fr.stream = i; fr.pls = F.pls; fr.pll_off = pll_off; fr.llr_off = llr_off; fr.sofq = F.sofq; fr.dst_index = 0;
            if (M.valid == 1) {
                auto& g = groups[M.code_index];
                fr.dst_index = (int)g.size();
                g.push_back((int)frames.size());
                group_pls[M.code_index] = F.pls;
                if (obytes + M.kb > out_cap) { last_error() = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
                dst.push_back(d_out[i] + obytes);
                frame_off.push_back(obytes);
                obytes += M.kb;
                pll_off += M.plframe; llr_off += M.N;
                d->tap_pll_count += M.plframe; d->tap_llr_count += M.N;
            } else {
                dst.push_back(nullptr);
                frame_off.push_back(-1);
            }
            d->frame_ptrs.push_back(fr.sym); d->frame_len.push_back(M.plframe); d->frame_pos.push_back(d->sym_base + F.offset);
            frames.push_back(fr);
        }
        stream_bytes[i] = obytes;
        if (!pipelined) out_bytes[i] = obytes;
    }
    first[n] = (int)frames.size();
    const int nf = (int)frames.size();
    std::vector<S2FrameStats> hstats(nf);
    std::vector<int32_t> trials(nf, 0), corr(nf, 0);
    std::unique_ptr<PendingFec> job_started;
    if (nf > 0) {
        if ((rc = W[2].ensure(sizeof(S2VcmFrame) * nf + sizeof(S2FrameStats) * nf + sizeof(uint8_t*) * nf + sizeof(int) * nf + 64))) return rc;
        S2VcmFrame* d_frames = (S2VcmFrame*)W[2].p;
        S2FrameStats* d_stats = (S2FrameStats*)(d_frames + nf);
        uint8_t** d_dst = (uint8_t**)(d_stats + nf);
        int* d_idx = (int*)(d_dst + nf);
        if ((rc = W[3].ensure((size_t)std::max<long long>(pll_off, 1) * sizeof(cf32)))) return rc;
        if ((rc = W[4].ensure((size_t)std::max<long long>(llr_off, 4)))) return rc;
        cf32* d_pll = (cf32*)W[3].p;
        int8_t* d_llr = (int8_t*)W[4].p;
        HIP_TRY(hipMemcpyAsync(d_frames, frames.data(), sizeof(S2VcmFrame) * nf, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_first, first.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_dst, dst.data(), sizeof(uint8_t*) * nf, hipMemcpyHostToDevice, st));
        { StageSpan sp(ctx->timers, ST_LOOPS, st); HIP_TRY(s2_vcm_loops_launch(d_work, n, d_frames, d_first, d0->co, ctx->pl, ctx->d_vcm_mods, ctx->d_vcm_cons, d_pll, d_stats, st)); }
        { StageSpan sp(ctx->timers, ST_DEMAP, st); HIP_TRY(s2_vcm_demap_launch(d_frames, nf, ctx->d_vcm_mods, ctx->d_vcm_cons, d_pll, d_llr, st)); }
        const int force = d0->cfg.force_ldpc_iters > 0;
        const int mt = force ? d0->cfg.force_ldpc_iters : d0->cfg.max_ldpc_trials;
        // one FEC job per LDPC code present in the call; group buffers: LLRs | BBFRAMEs | trials + corrections
        std::vector<std::pair<int, std::vector<int32_t>>> results;     // (code, trials ++ corrections) filled after the sync
        size_t off_idx = 0;
        std::vector<int> all_idx;
        for (auto& kv : groups) all_idx.insert(all_idx.end(), kv.second.begin(), kv.second.end());
        HIP_TRY(hipMemcpyAsync(d_idx, all_idx.data(), sizeof(int) * all_idx.size(), hipMemcpyHostToDevice, st));
        size_t llr_need = 0, bb_need = 0;
        for (auto& kv : groups) {
            const FecParams& f = ctx->h_vcm_fec[group_pls[kv.first]];
            llr_need += (size_t)kv.second.size() * f.N; bb_need += (size_t)kv.second.size() * (f.kbch / 8 + 8);
        }
        // pipelined: the job's buffers are the slot's double-buffered FEC workspaces (the next call reuses ws_vcm while the job runs):
        // LLR groups | BBFRAME groups | results (trials ++ corrections per part) + the parts' frame index lists + the destination table
        const int par = pipelined ? ctx->fec_parity[slot] : 0;
        Workspace& ws_gl = pipelined ? ctx->ws_fecbuf[slot][par][0] : W[5];
        Workspace& ws_gb = pipelined ? ctx->ws_fecbuf[slot][par][1] : W[6];
        Workspace& ws_gt = pipelined ? ctx->ws_fecbuf[slot][par][2] : W[7];
        const size_t nall = all_idx.size();
        const size_t res_bytes = (sizeof(int32_t) * 2 * nall + 63) & ~(size_t)63, idx_bytes = (sizeof(int) * nall + 63) & ~(size_t)63;
        if ((rc = ws_gl.ensure(llr_need + 64)) || (rc = ws_gb.ensure(bb_need + 64)) || (rc = ws_gt.ensure(res_bytes + idx_bytes + sizeof(uint8_t*) * nf + 64))) return rc;
        int* j_idx = (int*)((char*)ws_gt.p + res_bytes);
        if (pipelined) {
            HIP_TRY(hipMemcpyAsync(j_idx, all_idx.data(), sizeof(int) * nall, hipMemcpyHostToDevice, st));
            if (!ctx->ev_llr) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_llr, hipEventDisableTiming));
        }
        size_t lo = 0, bo = 0, to = 0;
        struct Out { int code; size_t to; int cnt; };
        std::vector<Out> outs;
        std::vector<PendingFec::Part> parts;
        struct Run { FecParams f; int8_t* llr; int cnt; uint8_t* bb; int32_t* tr; };
        std::vector<Run> runs;
        for (auto& kv : groups) {
            const FecParams& f = ctx->h_vcm_fec[group_pls[kv.first]];
            const int cnt = (int)kv.second.size(), kb = f.kbch / 8;
            int8_t* g_llr = (int8_t*)ws_gl.p + lo;
            uint8_t* g_bb = (uint8_t*)ws_gb.p + bo;
            int32_t* g_tr = (int32_t*)ws_gt.p + to;
            HIP_TRY(s2_vcm_gather_launch(d_frames, d_idx + off_idx, cnt, f.N, d_llr, g_llr, st));
            if (!pipelined) {
                if ((rc = fec_run(ctx, f, g_llr, cnt, mt, force, g_bb, g_tr, g_tr + cnt, st))) return rc;
                // the group's index list, re-based onto the per-frame destination table
                { StageSpan sp(ctx->timers, ST_DELIVER, st); HIP_TRY(s2_vcm_scatter_launch(d_idx + off_idx, cnt, kb, g_bb, d_dst, st)); }
            } else {
                runs.push_back(Run{f, g_llr, cnt, g_bb, g_tr});
                parts.push_back(PendingFec::Part{kb, cnt, g_bb, j_idx + off_idx, to});
            }
            outs.push_back(Out{kv.first, to, cnt});
            off_idx += cnt; lo += (size_t)cnt * f.N; bo += (size_t)cnt * kb; bo = (bo + 7) & ~(size_t)7; to += 2 * (size_t)cnt;
        }
        if (pipelined) {
            // the decoders of this call's codes go onto the FEC stream, behind the gathers; the call that follows collects them
            hipStream_t sf = ctx->fec_stream;
            HIP_TRY(hipEventRecord(ctx->ev_llr, st));
            {
                std::lock_guard<std::mutex> fl(ctx->fec_mtx);
                if ((rc = fec_stream_enter(ctx, sf))) return rc;
                HIP_TRY(hipStreamWaitEvent(sf, ctx->ev_llr, 0));
                for (const Run& r : runs)
                    if ((rc = fec_run(ctx, r.f, r.llr, r.cnt, mt, force, r.bb, r.tr, r.tr + r.cnt, sf))) return rc;
                if (!ctx->ev_fec[slot][par]) HIP_TRY(hipEventCreate(&ctx->ev_fec[slot][par]));
                HIP_TRY(hipEventRecord(ctx->ev_fec[slot][par], sf));
                fec_stream_leave(ctx, sf, ctx->ev_fec[slot][par]);
            }
            ctx->fec_parity[slot] ^= 1;
            HIP_TRY(hipMemcpyAsync(hstats.data(), d_stats, sizeof(S2FrameStats) * nf, hipMemcpyDeviceToHost, st));
            job_started.reset(new PendingFec());
            PendingFec& J = *job_started;
            J.vcm = true; J.n = n; J.nf = nf; J.dm.assign(dm, dm + n); J.first = first;
            J.parts = std::move(parts); J.frame_off = frame_off; J.stream_bytes = stream_bytes;
            J.frame_tr.assign(nf, -1); J.frame_co.assign(nf, -1);
            for (const Out& o : outs) {
                const std::vector<int>& idx = groups[o.code];
                for (int k = 0; k < o.cnt; ++k) { J.frame_tr[idx[k]] = (int)(o.to + k); J.frame_co[idx[k]] = (int)(o.to + o.cnt + k); }
            }
            J.d_trials = (const int32_t*)ws_gt.p; J.n_results = 2 * nall;
            J.d_dst = (uint8_t**)((char*)ws_gt.p + res_bytes + idx_bytes);
            J.done = ctx->ev_fec[slot][par];
        } else {
        std::vector<int32_t> tc(2 * all_idx.size());
        HIP_TRY(hipMemcpyAsync(tc.data(), ws_gt.p, sizeof(int32_t) * tc.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hstats.data(), d_stats, sizeof(S2FrameStats) * nf, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (const Out& o : outs) {
            const std::vector<int>& idx = groups[o.code];
            for (int k = 0; k < o.cnt; ++k) { trials[idx[k]] = tc[o.to + k]; corr[idx[k]] = tc[o.to + o.cnt + k]; }
        }
        }
        for (int i = 0; i < n; ++i)
            if (first[i + 1] > first[i]) {
                // taps: PLL output and LLRs of this stream's data frames are contiguous, in frame order
                for (int f = first[i]; f < first[i + 1]; ++f)
                    if (ctx->h_vcm_mods[frames[f].pls].valid == 1) { dm[i]->tap_pll = d_pll + frames[f].pll_off; dm[i]->tap_llr = d_llr + frames[f].llr_off; break; }
            }
    }
    // ---- FIFO remainder to the spare buffer, NCO frequency for the getter
    std::vector<int> curfill(2 * n);
    for (int i = 0; i < n; ++i) { curfill[2 * i] = cnts[4 * i + 1]; curfill[2 * i + 1] = dm[i]->fifo_fill; }
    HIP_TRY(hipMemcpyAsync(d_curfill, curfill.data(), sizeof(int) * 2 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(s2_fifo_compact_launch(d_work, n, d_curfill, st));
    std::vector<float> nco(n);
    HIP_TRY(s2_collect_launch(d_work, n, d_nsym, d_nco, st));
    HIP_TRY(hipMemcpyAsync(nco.data(), d_nco, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (pipelined) {
        for (int i = 0; i < n; ++i) {
            dvbs2gpu_demod* d = dm[i];
            const int cur = cnts[4 * i + 1];
            if (cur > 0) { d->fifo_fill -= cur; d->fifo_cur ^= 1; d->sym_base += cur; }
            d->nco_freq_host = nco[i];
        }
        // the job of the previous call of this slot is collected AFTER this call's job has gone onto the FEC stream (process_group does the same)
        PendingFec* prev = (PendingFec*)ctx->pending_fec[slot];
        ctx->pending_fec[slot] = nullptr;
        std::unique_ptr<PendingFec> prev_guard(prev);
        if (job_started) { job_started->hstats = hstats; ctx->pending_fec[slot] = job_started.release(); }
        if (prev && (rc = deliver_job(ctx, prev, st, W[8], *bm))) return rc;
        return 0;
    }
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        const int cur = cnts[4 * i + 1];
        if (cur > 0) { d->fifo_fill -= cur; d->fifo_cur ^= 1; d->sym_base += cur; }
        d->nco_freq_host = nco[i];
        for (int f = first[i]; f < first[i + 1]; ++f) {
            S2FrameStats s = hstats[f];
            s.ldpc_trials = trials[f]; s.bch_corr = corr[f];
            d->stats.push_back(s);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------- mixed CCM batches, one launch per stage
// A batch of up to S2_SMALL_BANK streams with DIFFERENT MODCODs (BASELINE config 4 as named: 64 transponders in 8 groups) and one front-end
// configuration.  Round 3 ran the MODCOD-dependent stages per configuration group -- a host thread and a HIP stream (a hardware queue) per group
// behind a shared front-end pass.  Here the whole call is ONE stage pipeline: AGC / timing recovery / RRC are MODCOD-independent anyway, and the
// PL-sync walk, the frame loops (ahead of the PL sync, as for a small bank of one configuration) and the demapper take what they otherwise get as
// kernel arguments from a per-stream table (S2StreamCfgDev).  The FEC stays one job per LDPC code -- a handful of decoder workgroups each, a few
// milliseconds of latency: they run side by side on up to four side streams (context option mix_fec_streams) -- and the BBFRAMEs go out through the per-frame
// destination table of the ACM/VCM jobs.  No host thread per group; the streams: the caller's, the front end's two auxiliary ones, the side streams.
int process_mixed(dvbs2gpu_ctx* ctx, dvbs2gpu_demod* const* dm, int n, const cf32* const* d_iq, const int* counts, uint8_t* const* d_out,
                  int out_cap, int* out_bytes, hipStream_t st, bool pipelined, const BatchMap* bm) {
    dvbs2gpu_demod* d0 = dm[0];
    Workspace* const W = ctx->ws_mix;
    int rc;
    if ((rc = get_rx_tables(ctx))) return rc;
    float* d_taps;
    if ((rc = get_rrc(ctx, d0->cfg.rrc_taps, d0->cfg.rrc_alpha, d0->cfg.samplerate / d0->cfg.symbolrate, &d_taps))) return rc;
    const bool loops_ahead = ctx->loops_ahead != 0 && !d0->cfg.pilot_aided && ctx->stage_pipeline_launches <= 0;
    std::vector<S2StreamWork> work(n);
    std::vector<S2StreamCfgDev> cfgs(n);
    int max_count = 0, maxf = 0, raw_max = 0, raw_min = 1 << 30, max_slots = 0;
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        const ModcodParams& mp = d->mp;
        if (counts[i] < 0 || counts[i] > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
        if (loops_ahead && !d->d_spec) HIP_TRY(hipMalloc((void**)&d->d_spec, sizeof(cf32) * 33282));
        ConstelTables* CT;
        if ((rc = get_constel(ctx, mp, &CT))) return rc;
        work[i].in = d_iq[i]; work[i].count = counts[i]; work[i].fe_out = d->d_fe;
        work[i].fifo = d->d_fifo[d->fifo_cur]; work[i].fifo_fill = d->fifo_fill; work[i].st = d->d_state;
        work[i].fifo_next = d->d_fifo[d->fifo_cur ^ 1]; work[i].out = d_out[i]; work[i].spec_out = d->d_spec;
        S2StreamCfgDev& q = cfgs[i];
        q.con = CT->dev; q.pls_code = d->pls_code; q.slots = mp.slots; q.pilots = mp.pilots; q.pilot_blocks = mp.pilot_blocks; q.plframe = mp.plframe;
        q.rate = mp.rate; q.N = mp.fec.N;
        max_count = std::max(max_count, counts[i]);
        maxf = std::max(maxf, d->fifo_cap / mp.plframe + 2);
        raw_max = std::max(raw_max, mp.plframe); raw_min = std::min(raw_min, mp.plframe); max_slots = std::max(max_slots, mp.slots);
        if (!pipelined) d->stats.clear();
        d->frame_ptrs.clear(); d->frame_pos.clear();
    }
    if ((rc = W[0].ensure(sizeof(S2StreamWork) * n + sizeof(S2StreamCfgDev) * n + sizeof(int) * 4 * n + sizeof(float) * n + 256))) return rc;
    S2StreamWork* d_work = (S2StreamWork*)W[0].p;
    S2StreamCfgDev* d_cfgs = (S2StreamCfgDev*)(d_work + n);
    int* d_nsym = (int*)(d_cfgs + n);              // [n]
    float* d_nco = (float*)(d_nsym + n);           // [n]
    int* d_curfill = (int*)(d_nco + n);            // [2n]
    HIP_TRY(hipMemcpyAsync(d_work, work.data(), sizeof(S2StreamWork) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_cfgs, cfgs.data(), sizeof(S2StreamCfgDev) * n, hipMemcpyHostToDevice, st));
    const size_t nslot = (size_t)n * maxf;
    if ((rc = W[1].ensure(sizeof(S2VcmFound) * nslot + sizeof(int) * 4 * n + 64))) return rc;
    S2VcmFound* d_found = (S2VcmFound*)W[1].p;
    int* d_counts = (int*)(d_found + nslot);
    if ((rc = W[3].ensure(nslot * raw_max * sizeof(cf32)))) return rc;
    if ((rc = W[7].ensure(sizeof(S2FrameStats) * nslot + 64))) return rc;
    cf32* d_pll = (cf32*)W[3].p;
    // ---- front half: the stage pipeline of a small bank, every stage one launch per slice for ALL configurations
    {
        struct Spans : S2SliceSpans {
            StageTimers* T; std::unique_ptr<StageSpan> sp[4];
            void begin(int stage, hipStream_t s) override { sp[stage].reset(new StageSpan(*T, stage == 1 ? ST_RRC : (stage == 2 ? ST_PLSYNC : ST_LOOPS), s)); }
            void end(int stage, hipStream_t) override { sp[stage].reset(); }
        } spans;
        spans.T = &ctx->timers;
        int launches = std::min(S2_FE_MAX_SLICES, std::max(1, (max_count / 2) / raw_min));
        if (ctx->stage_pipeline_launches > 0) launches = ctx->stage_pipeline_launches;
        S2PostStages post{d_taps, d0->cfg.rrc_taps, max_count + max_count / 32 + 8, raw_max, maxf, d_found, d_counts, ctx->pl, S2ConstelDev{}, 0,
                          0, 0, 0, d_pll, (S2FrameStats*)W[7].p, ctx->timers.on ? &spans : nullptr, launches};
        post.cfgs = d_cfgs;
        if (loops_ahead) {
            constexpr int sym_per_launch = 2700;
            post.spec = 1;
            post.loops_launches = std::min(S2_FE_MAX_SLICES, std::max(launches, (max_count / 2) / std::max(sym_per_launch, 1)));
        }
        StageSpan sp(ctx->timers, ST_FRONTEND, st);
        // (the post stages on a stream of their own in either mode: on the AGC's stream every slice's frame loops queue in front of the AGC slice the
        //  timing recovery waits for next -- 4.3 instead of 2.4 ms per slice)
        HIP_TRY(frontend_sliced(ctx, d_work, n, d0->co, st, &post, true, nullptr, max_count));
    }
    std::vector<S2FrameStats> slot_stats(nslot);
    std::vector<S2VcmFound> found(nslot);
    std::vector<int> cnts(4 * n);
    HIP_TRY(hipMemcpyAsync(slot_stats.data(), W[7].p, sizeof(S2FrameStats) * nslot, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(cnts.data(), d_counts, sizeof(int) * 4 * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(found.data(), d_found, sizeof(S2VcmFound) * nslot, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // ---- pool the frames (stream-major), one FEC part per (LDPC code, iteration setting)
    struct PartH { FecParams f; int mt, force; std::vector<int> frames; size_t lo = 0, bo = 0, to = 0; };
    std::vector<PartH> parts;
    std::map<std::tuple<int, int, int>, int> part_of;
    std::vector<int> first(n + 1, 0), fslot, frame_off, cur(n, 0), stream_bytes(n, 0);
    std::vector<S2FrameStats> hstats;
    // (every capacity is checked before any handle's state is touched: a refused call leaves all receivers where they were)
    for (int i = 0; i < n; ++i) {
        if (cnts[4 * i + 2] > dm[i]->fifo_cap) { last_error() = "symbol FIFO overflow"; return DVBS2GPU_ERR_CAPACITY; }
        if (cnts[4 * i] * (dm[i]->mp.fec.kbch / 8) > out_cap) { last_error() = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
    }
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        const int nfi = cnts[4 * i], kb = d->mp.fec.kbch / 8;
        d->tap_sym_off = d->fifo_fill; d->tap_sym_cnt = cnts[4 * i + 3]; d->tap_fifo = d->fifo_cur;
        d->fifo_fill = cnts[4 * i + 2];
        cur[i] = cnts[4 * i + 1];
        first[i] = (int)fslot.size();
        const int force = d->cfg.force_ldpc_iters > 0, mt = force ? d->cfg.force_ldpc_iters : d->cfg.max_ldpc_trials;
        const auto key = std::make_tuple(d->mp.fec.code_index, mt, force);
        auto it = part_of.find(key);
        if (nfi > 0 && it == part_of.end()) { it = part_of.emplace(key, (int)parts.size()).first; parts.push_back(PartH{d->mp.fec, mt, force, {}}); }     // (a part exists only with frames in it)
        const cf32* base = d->d_fifo[d->fifo_cur];
        for (int k = 0; k < nfi; ++k) {
            const S2VcmFound& F = found[(size_t)i * maxf + k];
            d->frame_ptrs.push_back(base + F.offset); d->frame_pos.push_back(d->sym_base + F.offset);
            S2FrameStats s = slot_stats[(size_t)i * maxf + k];
            s.best_match = F.sofq; s.bbframe_bytes = kb;
            hstats.push_back(s);
            frame_off.push_back(k * kb);
            parts[it->second].frames.push_back((int)fslot.size());
            fslot.push_back(i * maxf + k);
        }
        stream_bytes[i] = nfi * kb;
        d->tap_pll = d_pll + (size_t)i * maxf * raw_max;
        d->tap_pll_stride = raw_max;
        d->tap_llr = nullptr;
    }
    first[n] = (int)fslot.size();
    const int nf = (int)fslot.size();
    std::unique_ptr<PendingFec> job;
    if (nf > 0) {
        const int par = pipelined ? ctx->fec_parity[0] : 0;
        Workspace& ws_gl = pipelined ? ctx->ws_fecbuf[0][par][0] : W[4];
        Workspace& ws_gb = pipelined ? ctx->ws_fecbuf[0][par][1] : W[5];
        Workspace& ws_gt = pipelined ? ctx->ws_fecbuf[0][par][2] : W[6];
        size_t lo = 0, bo = 0, to = 0;
        for (PartH& P : parts) {
            P.lo = lo; P.bo = bo; P.to = to;
            lo += (size_t)P.frames.size() * P.f.N; bo += ((size_t)P.frames.size() * (P.f.kbch / 8) + 63) & ~(size_t)63; to += 2 * P.frames.size();
        }
        const size_t res_bytes = (sizeof(int32_t) * to + 63) & ~(size_t)63, idx_bytes = (sizeof(int) * nf + 63) & ~(size_t)63;
        if ((rc = ws_gl.ensure(lo + 64)) || (rc = ws_gb.ensure(bo + 64)) || (rc = ws_gt.ensure(res_bytes + idx_bytes + sizeof(uint8_t*) * nf + 64))) return rc;
        if ((rc = W[2].ensure(sizeof(int) * nf + sizeof(int8_t*) * nf + 64))) return rc;
        int8_t** d_llr_of = (int8_t**)W[2].p;
        int* d_slot = (int*)(d_llr_of + nf);
        int* j_idx = (int*)((char*)ws_gt.p + res_bytes);
        std::vector<int8_t*> llr_of(nf);
        std::vector<int> all_idx;
        for (const PartH& P : parts)
            for (size_t k = 0; k < P.frames.size(); ++k) { llr_of[P.frames[k]] = (int8_t*)ws_gl.p + P.lo + k * P.f.N; all_idx.push_back(P.frames[k]); }
        for (int i = 0; i < n; ++i)
            if (first[i + 1] > first[i]) dm[i]->tap_llr = llr_of[first[i]];      // (a stream's frames are consecutive inside its part)
        HIP_TRY(hipMemcpyAsync(d_llr_of, llr_of.data(), sizeof(int8_t*) * nf, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_slot, fslot.data(), sizeof(int) * nf, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(j_idx, all_idx.data(), sizeof(int) * nf, hipMemcpyHostToDevice, st));
        { StageSpan sp(ctx->timers, ST_DEMAP, st); HIP_TRY(s2_demap_mixed_launch(d_cfgs, max_slots, maxf, raw_max, d_pll, nf, d_llr_of, st, d_slot)); }
        if (!ctx->ev_llr) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_llr, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->ev_llr, st));
        // the FEC jobs, side by side
        job.reset(new PendingFec());
        PendingFec& J = *job;
        J.vcm = true; J.n = n; J.nf = nf; J.dm.assign(dm, dm + n); J.first = first; J.hstats = hstats;
        J.frame_off = frame_off; J.stream_bytes = stream_bytes;
        J.frame_tr.assign(nf, -1); J.frame_co.assign(nf, -1);
        J.d_trials = (const int32_t*)ws_gt.p; J.n_results = to;
        J.d_dst = (uint8_t**)((char*)ws_gt.p + res_bytes + idx_bytes);
        const int nside = std::min<int>((int)parts.size(), std::min(8, std::max(1, ctx->mix_fec_streams)));
        size_t off_idx = 0;
        for (size_t pi = 0; pi < parts.size(); ++pi) {
            const PartH& P = parts[pi];
            const int k = (int)(pi % nside), cnt = (int)P.frames.size();
            if (!ctx->grp_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&ctx->grp_stream[k], hipStreamNonBlocking));
            hipStream_t sk = ctx->grp_stream[k];
            if ((int)pi < nside) HIP_TRY(hipStreamWaitEvent(sk, ctx->ev_llr, 0));
            int32_t* g_tr = (int32_t*)ws_gt.p + P.to;
            uint8_t* g_bb = (uint8_t*)ws_gb.p + P.bo;
            if ((rc = fec_run(ctx, P.f, (const int8_t*)ws_gl.p + P.lo, cnt, P.mt, P.force, g_bb, g_tr, g_tr + cnt, sk, &ctx->fws_grp[k]))) return rc;
            J.parts.push_back(PendingFec::Part{P.f.kbch / 8, cnt, g_bb, j_idx + off_idx, P.to});
            for (int q = 0; q < cnt; ++q) { J.frame_tr[P.frames[q]] = (int)(P.to + q); J.frame_co[P.frames[q]] = (int)(P.to + cnt + q); }
            off_idx += cnt;
        }
        for (int k = 0; k < nside; ++k) {
            if (!ctx->ev_mix[k][par]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_mix[k][par], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ctx->ev_mix[k][par], ctx->grp_stream[k]));
            J.done_more.push_back(ctx->ev_mix[k][par]);
        }
        if (pipelined) ctx->fec_parity[0] ^= 1;
    }
    // ---- FIFO remainder to the spare buffer, NCO frequency for the getter
    std::vector<int> curfill(2 * n);
    for (int i = 0; i < n; ++i) { curfill[2 * i] = cur[i]; curfill[2 * i + 1] = dm[i]->fifo_fill; }
    HIP_TRY(hipMemcpyAsync(d_curfill, curfill.data(), sizeof(int) * 2 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(s2_fifo_compact_launch(d_work, n, d_curfill, st));
    std::vector<float> nco(n);
    HIP_TRY(s2_collect_launch(d_work, n, d_nsym, d_nco, st));
    HIP_TRY(hipMemcpyAsync(nco.data(), d_nco, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        if (cur[i] > 0) { d->fifo_fill -= cur[i]; d->fifo_cur ^= 1; d->sym_base += cur[i]; }
        d->nco_freq_host = nco[i];
    }
    BatchMap bm_local;
    if (!bm) {
        for (int i = 0; i < n; ++i) { bm_local.pos[dm[i]] = i; out_bytes[i] = 0; dm[i]->stats.clear(); }
        bm_local.d_out = d_out; bm_local.out_bytes = out_bytes; bm_local.out_cap = out_cap;
        bm = &bm_local;
    }
    if (!pipelined) return job ? deliver_job(ctx, job.get(), st, W[8], *bm) : 0;
    PendingFec* prev = (PendingFec*)ctx->pending_fec[0];
    ctx->pending_fec[0] = job.release();
    std::unique_ptr<PendingFec> prev_guard(prev);
    return prev ? deliver_job(ctx, prev, st, W[8], *bm) : 0;
}

// AGC, NCO, Gardner, RRC + decimation do not depend on the MODCOD: for a batch of several configuration groups whose loop
// coefficients and matched filter agree they run ONCE over all streams (these kernels are latency-bound: eight groups of 512
// streams cost eight times one group of 4096).  Leaves the symbols in the streams' FIFOs exactly as process_group would.
int frontend_prepass(dvbs2gpu_ctx* ctx, dvbs2gpu_demod* const* dm, int n, const cf32* const* d_iq, const int* counts, uint8_t* const* d_out,
                     hipStream_t st, std::vector<int>* nsym_out) {
    int rc;
    if ((rc = get_rx_tables(ctx))) return rc;
    dvbs2gpu_demod* d0 = dm[0];
    float* d_taps;
    if ((rc = get_rrc(ctx, d0->cfg.rrc_taps, d0->cfg.rrc_alpha, d0->cfg.samplerate / d0->cfg.symbolrate, &d_taps))) return rc;
    std::vector<S2StreamWork> work(n);
    int max_count = 0;
    for (int i = 0; i < n; ++i) {
        dvbs2gpu_demod* d = dm[i];
        if (counts[i] < 0 || counts[i] > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
        work[i].in = d_iq[i]; work[i].count = counts[i]; work[i].fe_out = d->d_fe;
        work[i].fifo = d->d_fifo[d->fifo_cur]; work[i].fifo_fill = d->fifo_fill; work[i].st = d->d_state;
        work[i].fifo_next = d->d_fifo[d->fifo_cur ^ 1]; work[i].out = d_out[i]; work[i].spec_out = d->d_spec;
        max_count = std::max(max_count, counts[i]);
    }
    Workspace& ws = ctx->ws_rx[7];
    if ((rc = ws.ensure(sizeof(S2StreamWork) * n + sizeof(int) * n + sizeof(float) * n + 64))) return rc;
    S2StreamWork* d_work = (S2StreamWork*)ws.p;
    int* d_nsym = (int*)((char*)ws.p + sizeof(S2StreamWork) * n);
    float* d_nco = (float*)(d_nsym + n);
    HIP_TRY(hipMemcpyAsync(d_work, work.data(), sizeof(S2StreamWork) * n, hipMemcpyHostToDevice, st));
    { StageSpan sp(ctx->timers, ST_FRONTEND, st); HIP_TRY(frontend_sliced(ctx, d_work, n, d0->co, st)); }
    { StageSpan sp(ctx->timers, ST_RRC, st); HIP_TRY(s2_rrc_decim_launch(d_work, n, max_count + max_count / 32 + 8, d_taps, d0->cfg.rrc_taps, st)); }
    nsym_out->assign(n, 0);
    HIP_TRY(s2_collect_launch(d_work, n, d_nsym, d_nco, st));
    HIP_TRY(hipMemcpyAsync(nsym_out->data(), d_nsym, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// what the pre-pass stages read from a configuration
bool same_frontend(const dvbs2gpu_demod* a, const dvbs2gpu_demod* b) {
    const S2LoopCoefs &x = a->co, &y = b->co;
    return x.agc_rate == y.agc_rate && x.g_alpha == y.g_alpha && x.g_beta == y.g_beta && x.g_min_freq == y.g_min_freq && x.g_max_freq == y.g_max_freq &&
           a->cfg.rrc_taps == b->cfg.rrc_taps && a->cfg.rrc_alpha == b->cfg.rrc_alpha && a->cfg.samplerate == b->cfg.samplerate &&
           a->cfg.symbolrate == b->cfg.symbolrate;
}

}  // namespace

extern "C" {

void dvbs2gpu_demod_default_cfg(int modcod, int shortframes, int pilots, dvbs2gpu_demod_cfg* c) {   // main.cpp:64-73,134-140
    c->symbolrate = 27.5e6; c->samplerate = 55e6;
    c->agc_rate = 0.0001f; c->rrc_alpha = 0.35f; c->rrc_taps = 65; c->loop_bw = 0.00628f; c->fll_bw = 0.006f;
    float bw = 0.00628f, damp = 0.707f;
    float den = (1.0f + 2.0 * damp * bw + bw * bw);
    c->clock_mu_gain = (4.0f * damp * bw) / den;
    c->clock_omega_gain = (4.0f * bw * bw) / den;
    c->omega_rel_limit = 0.02f;
    c->modcod = modcod; c->shortframes = shortframes; c->pilots = pilots;
    c->sof_threshold = 0.6f; c->max_ldpc_trials = 16; c->force_ldpc_iters = 0;
    c->acm_vcm = 0; c->soft_plsc = 0; c->pilot_aided = 0;
}

int dvbs2gpu_demod_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_demod_cfg* cfg, int max_samples, dvbs2gpu_demod** out) {
    if (!ctx || !cfg || !out || max_samples < 16) return DVBS2GPU_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<dvbs2gpu_demod> d(new dvbs2gpu_demod());
    d->ctx = ctx; d->cfg = *cfg; d->max_samples = max_samples;
    int rc = demod_configure(d.get());
    if (rc) return rc;
    CallGuard guard(ctx);               // (one context may serve several blocks on several host threads: whole calls are serialised, ctx.h)
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc((void**)&d->d_state, sizeof(S2StreamState)));
    HIP_TRY(hipMalloc((void**)&d->d_fe, fe_capacity(max_samples) * sizeof(cf32)));
    // FIFO: leftover (< 2 PLFRAMEs, the largest one) + the symbols of one call
    d->fifo_cap = max_samples / 2 + max_samples / 32 + 2 * 33282 + 1024;
    for (int k = 0; k < 2; ++k) HIP_TRY(hipMalloc((void**)&d->d_fifo[k], (size_t)d->fifo_cap * sizeof(cf32)));
    if ((rc = demod_reset_state(d.get()))) return rc;
    *out = d.release();
    return DVBS2GPU_OK;
}

void dvbs2gpu_demod_destroy(dvbs2gpu_demod* d) {
    if (!d) return;
    CallGuard guard(d->ctx);
    (void)hipSetDevice(d->ctx->device);
    (void)hipDeviceSynchronize();
    // a pipelined FEC job still in flight names its streams by handle: this one leaves the lists, so that a handle created later at the same address is not taken
    // for it (its frames are dropped at the collecting call, like those of any stream that has left the batch)
    for (void* pj : d->ctx->pending_fec)
        if (pj) for (auto& h : ((PendingFec*)pj)->dm) if (h == d) h = nullptr;
    (void)hipFree(d->d_state); if (d->d_in) (void)hipFree(d->d_in); (void)hipFree(d->d_fe);
    (void)hipFree(d->d_fifo[0]); (void)hipFree(d->d_fifo[1]); if (d->d_out) (void)hipFree(d->d_out);
    if (d->d_spec) (void)hipFree(d->d_spec);
    delete d;
}

int dvbs2gpu_demod_reset(dvbs2gpu_demod* d) {
    if (!d) return DVBS2GPU_ERR_ARG;
    CallGuard guard(d->ctx);
    HIP_TRY(hipSetDevice(d->ctx->device));
    // the engine's streams are non-blocking: a null-stream copy does not order itself behind what another handle's call still has in flight
    HIP_TRY(hipDeviceSynchronize());
    return demod_reset_state(d);
}

int dvbs2gpu_demod_set_params(dvbs2gpu_demod* d, int modcod, int shortframes, int pilots, float sof_threshold, int max_ldpc_trials) {
    if (!d) return DVBS2GPU_ERR_ARG;
    CallGuard guard(d->ctx);
    dvbs2gpu_demod_cfg saved = d->cfg;
    d->cfg.modcod = modcod; d->cfg.shortframes = shortframes; d->cfg.pilots = pilots;
    d->cfg.sof_threshold = sof_threshold; d->cfg.max_ldpc_trials = max_ldpc_trials;
    int rc = demod_configure(d);
    if (rc) { d->cfg = saved; (void)demod_configure(d); return rc; }
    // setDemodParams restarts the PL sync buffer (dvbs2_pl_sync.cpp:51-79); loops keep running
    d->sym_base += d->fifo_fill;      // (the dropped symbols still count on the stream's symbol axis)
    d->fifo_fill = 0;
    {   // the whole PL-sync sub-state back to 0 on the device as well: pending realign + last best_match, and where the sliced walk / frame loops stood
        HIP_TRY(hipSetDevice(d->ctx->device));
        HIP_TRY(hipDeviceSynchronize());            // (non-blocking streams: see dvbs2gpu_demod_reset)
        const int zero_pl[2] = {0, 0};              // pl_pending, pl_last_bm (0.0f)
        const int zero_walk[3] = {0, 0, 0};         // walk_cur, walk_nf, loops_done
        static_assert(offsetof(S2StreamState, pl_last_bm) == offsetof(S2StreamState, pl_pending) + sizeof(int), "pl_pending and pl_last_bm are cleared together");
        static_assert(offsetof(S2StreamState, loops_done) == offsetof(S2StreamState, walk_cur) + 2 * sizeof(int), "walk_cur, walk_nf, loops_done are cleared together");
        {   // frame loops that ran ahead of the PL sync (s2_frame_loops_kernel): back to the loop state at that window's start -- the window is gone
            S2StreamState hs;
            HIP_TRY(hipMemcpy(&hs, d->d_state, sizeof(hs), hipMemcpyDeviceToHost));
            if (hs.spec_on) {
                const float back[2] = {hs.spec_phase0, hs.spec_freq0};
                static_assert(offsetof(S2StreamState, pll_freq) == offsetof(S2StreamState, pll_phase) + sizeof(float), "pll_phase, pll_freq are restored together");
                HIP_TRY(hipMemcpy((char*)d->d_state + offsetof(S2StreamState, pll_phase), back, sizeof(back), hipMemcpyHostToDevice));
                const int off = 0;
                HIP_TRY(hipMemcpy((char*)d->d_state + offsetof(S2StreamState, spec_on), &off, sizeof(off), hipMemcpyHostToDevice));
            }
        }
        HIP_TRY(hipMemcpy((char*)d->d_state + offsetof(S2StreamState, pl_pending), zero_pl, sizeof(zero_pl), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy((char*)d->d_state + offsetof(S2StreamState, walk_cur), zero_walk, sizeof(zero_walk), hipMemcpyHostToDevice));
    }
    return 0;
}

int dvbs2gpu_demod_get_kbch(dvbs2gpu_demod* d) { return d ? d->mp.fec.kbch : DVBS2GPU_ERR_ARG; }

// The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues as they are created, and streams that share a queue take turns: a pipelined run's streams should not find the
// queues half taken by streams an EARLIER run of another shape left behind (round 6: the 4096-carrier mixed batch 192 ms per step in an engine of its own, 205-214 ms behind the
// bench's other lines -- and config 4 as named 36 or 61 ms depending on who came first).  So a pipelined run starts by giving the context's streams back; everything here is created
// on demand again.  (Not at the END of a run: synchronous small calls are better off on the streams they inherit -- a wait between two streams of ONE hardware queue is cheap, and a
// call of a few thousand samples is ~130 launches tied together by such waits: with fresh streams the drop-in call of 8192 samples took 3.4 instead of 2.4 ms.)
static int release_streams(dvbs2gpu_ctx* ctx) {
    HIP_TRY(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> l(ctx->mtx);
        for (auto& kv : ctx->fe_aux) {
            dvbs2gpu_ctx::FeAux& a = kv.second;
            if (a.aux) (void)hipStreamDestroy(a.aux);
            if (a.aux2) (void)hipStreamDestroy(a.aux2);
            if (a.aux3) (void)hipStreamDestroy(a.aux3);
            for (hipEvent_t e : a.ev) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : a.ev2) if (e) (void)hipEventDestroy(e);
            for (hipEvent_t e : a.ev3) if (e) (void)hipEventDestroy(e);
            for (hipStream_t d : a.dvbs_aux) if (d) (void)hipStreamDestroy(d);
            for (auto& row : a.dvbs_ev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
        }
        ctx->fe_aux.clear();
    }
    for (hipStream_t& sg : ctx->grp_stream) if (sg) { (void)hipStreamDestroy(sg); sg = nullptr; }
    if (ctx->fec_part_stream) { (void)hipStreamDestroy(ctx->fec_part_stream); ctx->fec_part_stream = nullptr; }
    if (ctx->fec_stream) { (void)hipStreamDestroy(ctx->fec_stream); ctx->fec_stream = nullptr; }
    if (ctx->fe_stream) { (void)hipStreamDestroy(ctx->fe_stream); ctx->fe_stream = nullptr; }
    ctx->fec_last_done = nullptr; ctx->fec_last_stream = nullptr;
    return 0;
}
int dvbs2gpu_set_pipelined(dvbs2gpu_ctx* ctx, int on) {
    if (!ctx) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    if (on) {      // (the FEC stream may exist already: synchronous mixed batches use it too)
        if (!ctx->pipeline_fec) { int rr = release_streams(ctx); if (rr) return rr; }
        if (!ctx->fe_stream) HIP_TRY(create_stream(ctx, &ctx->fe_stream, +1));
        if (!ctx->fec_stream) HIP_TRY(create_stream(ctx, &ctx->fec_stream, -1));
        if (!ctx->ev_llr) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_llr, hipEventDisableTiming));
    }
    if (!on && ctx->fec_stream) {
        // frames of the last pipelined call that nobody collected are dropped (collect them with a zero-count call first)
        HIP_TRY(hipStreamSynchronize(ctx->fec_stream));
        if (ctx->fec_part_stream) HIP_TRY(hipStreamSynchronize(ctx->fec_part_stream));
        for (hipStream_t sg : ctx->grp_stream) if (sg) HIP_TRY(hipStreamSynchronize(sg));      // (small jobs of mixed batches run on their group's stream)
        for (auto& pj : ctx->pending_fec) { delete (PendingFec*)pj; pj = nullptr; }
    }
    ctx->pipeline_fec = on ? 1 : 0;
    return 0;
}

int dvbs2gpu_demod_process_batch(dvbs2gpu_demod* const* demods, int n, const float* const* d_iq, const int* counts,
                                 uint8_t* const* d_out, int out_cap, int* out_bytes) {
    if (!demods || n <= 0 || !d_iq || !counts || !d_out || !out_bytes) return DVBS2GPU_ERR_ARG;
    dvbs2gpu_ctx* ctx = demods[0]->ctx;
    CallGuard guard(ctx);                       // the per-call workspaces are context-wide: one call at a time per context
    HIP_TRY(hipSetDevice(ctx->device));
    { int rq = ws_quiesce(ctx); if (rq) return rq; }
    // group streams that share a configuration (order inside a group = caller's order)
    std::vector<std::vector<int>> groups;
    {
        std::vector<char> done(n, 0);
        for (int i = 0; i < n; ++i) {
            if (done[i]) continue;
            if (demods[i]->ctx != ctx) { last_error() = "all streams of a batch must belong to one context"; return DVBS2GPU_ERR_ARG; }
            std::vector<int> idx;
            for (int k = i; k < n; ++k) {
                if (done[k] || demods[k]->ctx != ctx) continue;
                const dvbs2gpu_demod_cfg &a = demods[i]->cfg, &b = demods[k]->cfg;
                if (memcmp(&a, &b, sizeof(a)) == 0) { idx.push_back(k); done[k] = 1; }
            }
            groups.push_back(std::move(idx));
        }
    }
    const bool pipe = ctx->pipeline_fec != 0;
    bool any_vcm = false;
    for (int i = 0; i < n; ++i) any_vcm = any_vcm || demods[i]->cfg.acm_vcm != 0;
    if (pipe && (int)groups.size() > dvbs2gpu_ctx::MAX_PIPE_GROUPS) { last_error() = "pipelined mode handles at most 16 configuration groups per batch"; return DVBS2GPU_ERR_ARG; }
    hipStream_t st = pipe ? ctx->fe_stream : nullptr;
    if (pipe && S2_INPUT_EVENT) {
        // The synchronous mode runs on the legacy null stream: whatever the host has put there before the call -- the kernels or copies that FILL its input buffers -- lies in
        // front of the demodulator by itself.  The throughput mode's own stream is non-blocking: without this it READ INPUT THAT WAS STILL BEING WRITTEN whenever the host's
        // producer had not finished (round 6, tools/stress_pipelined.py on fresh streams: a frame of a busy call came out with LDPC trials -1 and two BCH corrections where
        // the synchronous run had 0 / 0; with streams that happened to share a hardware queue with the null stream it never showed).  Same contract in both modes now: inputs
        // produced on the null stream are ordered; a host that fills them on a stream of its own synchronises that stream first.
        if (!ctx->ev_in) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->ev_in, nullptr));
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_in, 0));
    }
    // pipelined: every job is collected through this map -- the streams of a batch may come and go and change places between calls
    BatchMap bmap;
    if (pipe) {
        for (int i = 0; i < n; ++i) {
            if (!bmap.pos.emplace(demods[i], i).second) { last_error() = "a stream appears twice in the batch"; return DVBS2GPU_ERR_ARG; }
            out_bytes[i] = 0;
            demods[i]->stats.clear();
        }
        bmap.d_out = d_out; bmap.out_bytes = out_bytes; bmap.out_cap = out_cap;
    }
    // jobs left in slots this call has no group for (the batch has fewer configuration groups than the one before): collected at the end of the call
    auto collect_leftovers = [&](int first_free_slot) -> int {
        for (int sl = first_free_slot; sl < dvbs2gpu_ctx::MAX_PIPE_GROUPS; ++sl) {
            PendingFec* job = (PendingFec*)ctx->pending_fec[sl];
            if (!job) continue;
            ctx->pending_fec[sl] = nullptr;
            std::unique_ptr<PendingFec> guard(job);
            int rc = deliver_job(ctx, job, st, ctx->ws_rx[6], bmap);
            if (rc) return rc;
        }
        return 0;
    };
    if (any_vcm) {
        // ACM/VCM streams: group after group (the FEC jobs of a call depend on what its framing finds: one per LDPC code present); pipelined: the
        // jobs run on the FEC stream beside the next call's front end and are collected by that call, like those of the CCM groups
        int slot = 0;
        for (const std::vector<int>& idx : groups) {
            std::vector<dvbs2gpu_demod*> g;
            std::vector<const cf32*> gi;
            std::vector<int> gc, gb(idx.size());
            std::vector<uint8_t*> go;
            for (int k : idx) { g.push_back(demods[k]); gi.push_back((const cf32*)d_iq[k]); gc.push_back(counts[k]); go.push_back(d_out[k]); }
            int rc = g[0]->cfg.acm_vcm ? process_vcm_group(ctx, g.data(), (int)g.size(), gi.data(), gc.data(), go.data(), out_cap, gb.data(), st, pipe, slot, pipe ? &bmap : nullptr)
                                       : process_group(ctx, g.data(), (int)g.size(), gi.data(), gc.data(), go.data(), out_cap, gb.data(), st, pipe, slot, nullptr, false, false, pipe ? &bmap : nullptr);
            if (rc) return rc;
            if (!pipe) for (size_t k = 0; k < idx.size(); ++k) out_bytes[idx[k]] = gb[k];
            ++slot;
        }
        return pipe ? collect_leftovers(slot) : 0;
    }
    // a small mixed batch with one front-end and loop configuration: ONE stage pipeline for all MODCODs (process_mixed); DVBS2GPU_MIXED_GROUPS=1
    // keeps round 3's flow (a host thread and a HIP stream per configuration group behind a shared front-end pass), which bigger mixed batches use
    if (groups.size() > 1 && n <= S2_SMALL_BANK && ctx->stage_pipeline == 1) {
        const bool by_groups = ctx->mixed_groups != 0;
        bool one = !by_groups;
        for (int i = 1; one && i < n; ++i) {
            const S2LoopCoefs &x = demods[0]->co, &y = demods[i]->co;
            one = same_frontend(demods[0], demods[i]) && x.pll_alpha == y.pll_alpha && x.pll_beta == y.pll_beta && x.hdr_alpha == y.hdr_alpha && x.hdr_beta == y.hdr_beta &&
                  x.soft_plsc == y.soft_plsc && x.pilot_aided == y.pilot_aided;
        }
        if (one) {
            int rc = process_mixed(ctx, demods, n, (const cf32* const*)d_iq, counts, d_out, out_cap, out_bytes, st, pipe, pipe ? &bmap : nullptr);
            if (rc) return rc;
            return pipe ? collect_leftovers(1) : 0;
        }
    }
    // bigger mixed batches (and small ones whose loop settings differ): several groups with one front end -- the MODCOD-independent stages run once
    // for the whole batch, then the groups' MODCOD-dependent stages side by side
    std::vector<int> pre_nsym;
    bool merged = groups.size() > 1;
    for (int i = 1; merged && i < n; ++i) merged = same_frontend(demods[0], demods[i]);
    if (merged) {
        int rc = frontend_prepass(ctx, demods, n, (const cf32* const*)d_iq, counts, d_out, st, &pre_nsym);
        if (rc) return rc;
    }
    struct GroupJob {
        std::vector<int> idx, gc, gb, gn;
        std::vector<dvbs2gpu_demod*> g;
        std::vector<const cf32*> gi;
        std::vector<uint8_t*> go;
        int slot = 0, rc = 0;
        std::string err;
    };
    std::list<GroupJob> jobs;
    // (synchronous calls collect each group's FEC job before they return)
    const bool side_by_side = merged && (int)groups.size() <= dvbs2gpu_ctx::MAX_PIPE_GROUPS;
    int group_no = 0;
    for (const std::vector<int>& idx : groups) {
        std::vector<dvbs2gpu_demod*> g;
        std::vector<const cf32*> gi;
        std::vector<int> gc, gb(idx.size()), gn;
        std::vector<uint8_t*> go;
        for (int k : idx) {
            g.push_back(demods[k]); gi.push_back((const cf32*)d_iq[k]); gc.push_back(counts[k]); go.push_back(d_out[k]);
            if (merged) gn.push_back(pre_nsym[k]);
        }
        if (side_by_side) {
            jobs.emplace_back();
            GroupJob& J = jobs.back();
            J.idx = idx; J.g = std::move(g); J.gi = std::move(gi); J.gc = std::move(gc); J.go = std::move(go); J.gn = std::move(gn);
            J.gb.assign(idx.size(), 0); J.slot = group_no++;
            continue;
        }
        int rc = process_group(ctx, g.data(), (int)g.size(), gi.data(), gc.data(), go.data(), out_cap, gb.data(), st, pipe, pipe ? group_no : 0,
                               merged ? gn.data() : nullptr, false, false, pipe ? &bmap : nullptr);
        if (rc) return rc;
        if (!pipe) for (size_t k = 0; k < idx.size(); ++k) out_bytes[idx[k]] = gb[k];
        ++group_no;
    }
    if (side_by_side) {
        // the MODCOD-dependent stages (PL sync, frame loops, demapper, FEC hand-over, delivery) of the groups are independent and
        // latency-bound: one host thread and HIP stream per group (the pre-pass above has completed; every group ends synchronised)
        if (!ctx->fec_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fec_stream, hipStreamNonBlocking));
        // The groups' streams only run side by side while each has a hardware queue to itself (HIP maps streams onto GPU_MAX_HW_QUEUES
        // queues -- 4 by default -- and a stream created when all are taken shares one: 64 x 4 PLFRAMEs cost 75 ms per step with a queue per
        // group, 110 with four queues).  Auxiliary streams other flows of this context left behind (time-sliced front ends on other main
        // streams, the DVB-S receiver's stage streams) are idle now -- one call at a time per context -- and given back first.
        bool need_streams = false;
        for (GroupJob& J : jobs) need_streams = need_streams || !ctx->grp_stream[J.slot];
        if (need_streams) {
            std::lock_guard<std::mutex> l(ctx->mtx);
            for (auto it = ctx->fe_aux.begin(); it != ctx->fe_aux.end();) {
                if (it->first == st) { ++it; continue; }
                dvbs2gpu_ctx::FeAux& a = it->second;
                if (a.aux) (void)hipStreamDestroy(a.aux);
                if (a.aux2) (void)hipStreamDestroy(a.aux2);
                if (a.aux3) (void)hipStreamDestroy(a.aux3);
                for (hipEvent_t e : a.ev) if (e) (void)hipEventDestroy(e);
                for (hipEvent_t e : a.ev2) if (e) (void)hipEventDestroy(e);
                for (hipEvent_t e : a.ev3) if (e) (void)hipEventDestroy(e);
                for (hipStream_t d : a.dvbs_aux) if (d) (void)hipStreamDestroy(d);
                for (auto& row : a.dvbs_ev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
                it = ctx->fe_aux.erase(it);
            }
        }
        for (GroupJob& J : jobs) {
            if (!ctx->grp_stream[J.slot]) HIP_TRY(hipStreamCreateWithFlags(&ctx->grp_stream[J.slot], hipStreamNonBlocking));
            if (!ctx->ev_llr_grp[J.slot]) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_llr_grp[J.slot], hipEventDisableTiming));
        }
        std::vector<std::thread> th;
        for (GroupJob& J : jobs) {
            if (&J == &jobs.back()) break;            // (the last group runs on the calling thread, below)
            try {
                th.emplace_back([&J, ctx, out_cap, pipe, merged, &bmap]() {
                    if (hipSetDevice(ctx->device) != hipSuccess) { J.rc = DVBS2GPU_ERR_HIP; J.err = "hipSetDevice"; return; }
                    J.rc = process_group(ctx, J.g.data(), (int)J.g.size(), J.gi.data(), J.gc.data(), J.go.data(), out_cap, J.gb.data(),
                                         ctx->grp_stream[J.slot], true, J.slot, merged ? J.gn.data() : nullptr, true, !pipe, pipe ? &bmap : nullptr);
                    if (J.rc) J.err = last_error();
                });
            } catch (...) {                           // no thread to be had: run the group here (no exception leaves the C ABI)
                J.rc = process_group(ctx, J.g.data(), (int)J.g.size(), J.gi.data(), J.gc.data(), J.go.data(), out_cap, J.gb.data(),
                                     ctx->grp_stream[J.slot], true, J.slot, merged ? J.gn.data() : nullptr, true, !pipe, pipe ? &bmap : nullptr);
                if (J.rc) J.err = last_error();
            }
        }
        {
            GroupJob& J = jobs.back();
            J.rc = process_group(ctx, J.g.data(), (int)J.g.size(), J.gi.data(), J.gc.data(), J.go.data(), out_cap, J.gb.data(),
                                 ctx->grp_stream[J.slot], true, J.slot, merged ? J.gn.data() : nullptr, true, !pipe, pipe ? &bmap : nullptr);
            if (J.rc) J.err = last_error();
        }
        for (auto& t : th) t.join();
        for (GroupJob& J : jobs) {
            if (J.rc) { last_error() = J.err; return J.rc; }
            if (!pipe) for (size_t k = 0; k < J.idx.size(); ++k) out_bytes[J.idx[k]] = J.gb[k];
        }
    }
    return pipe ? collect_leftovers(group_no) : 0;
}

int dvbs2gpu_demod_process(dvbs2gpu_demod* d, int count, const float* h_iq, uint8_t* h_out, int out_cap) {
    if (!d || count < 0 || (count > 0 && !h_iq) || !h_out) return DVBS2GPU_ERR_ARG;
    if (count > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
    CallGuard guard(d->ctx);                    // the per-call workspaces are context-wide: one call at a time per context
    HIP_TRY(hipSetDevice(d->ctx->device));
    { int rq = ws_quiesce(d->ctx); if (rq) return rq; }
    if (!d->d_in) {   // staging buffers of the host-pointer entry point, allocated on first use
        HIP_TRY(hipMalloc((void**)&d->d_in, (size_t)d->max_samples * sizeof(cf32)));
        int max_frames = d->fifo_cap / 3330 + 2;
        HIP_TRY(hipMalloc((void**)&d->d_out, (size_t)max_frames * 8100));
    }
    if (count) HIP_TRY(hipMemcpy(d->d_in, h_iq, (size_t)count * sizeof(cf32), hipMemcpyHostToDevice));
    const cf32* in = d->d_in;
    uint8_t* dout = d->d_out;
    int bytes = 0;
    int cap = (d->fifo_cap / d->mp.plframe + 2) * (d->mp.fec.kbch / 8);
    dvbs2gpu_demod* dd = d;
    int rc;
    if (d->cfg.acm_vcm) {
        cap = (d->fifo_cap / 3330 + 2) * 8100;       // the staging buffer's size (allocated above)
        if (d->ctx->pipeline_fec) { last_error() = "ACM/VCM streams run in the synchronous mode (dvbs2gpu_set_pipelined(ctx, 0))"; return DVBS2GPU_ERR_ARG; }
        rc = process_vcm_group(d->ctx, &dd, 1, &in, &count, &dout, cap, &bytes, nullptr);
    } else
    rc = process_group(d->ctx, &dd, 1, &in, &count, &dout, cap, &bytes, nullptr, false, 0, nullptr, false, false);
    if (rc) return rc;
    if (bytes > out_cap) { last_error() = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
    if (bytes) HIP_TRY(hipMemcpy(h_out, d->d_out, bytes, hipMemcpyDeviceToHost));
    return bytes;
}

int dvbs2gpu_demod_get_stats(dvbs2gpu_demod* d, dvbs2gpu_frame_stats* h_out, int cap) {
    if (!d) return DVBS2GPU_ERR_ARG;
    int n = (int)d->stats.size();
    static_assert(sizeof(dvbs2gpu_frame_stats) == sizeof(S2FrameStats), "stats layout");
    if (h_out) memcpy(h_out, d->stats.data(), sizeof(S2FrameStats) * std::min(n, cap));
    return n;
}

float dvbs2gpu_demod_get_nco_freq(dvbs2gpu_demod* d) { return d ? d->nco_freq_host : 0.f; }

int dvbs2gpu_demod_get_frame_positions(dvbs2gpu_demod* d, int64_t* h_out, int cap) {
    if (!d) return DVBS2GPU_ERR_ARG;
    const int n = (int)d->frame_pos.size();
    for (int i = 0; h_out && i < std::min(n, cap); ++i) h_out[i] = d->frame_pos[i];
    return n;
}

int dvbs2gpu_demod_get_tap(dvbs2gpu_demod* d, int which, void* h_dst, int cap) {
    if (!d) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(d->ctx->device));
    const int raw = d->mp.plframe, N = d->mp.fec.N, nf = (int)d->frame_ptrs.size();
    switch (which) {
        case 0: {
            int n = d->tap_sym_cnt;
            if (h_dst && n) HIP_TRY(hipMemcpy(h_dst, d->d_fifo[d->tap_fifo] + d->tap_sym_off, sizeof(cf32) * std::min(n, cap), hipMemcpyDeviceToHost));
            return n;
        }
        case 1: {
            if (!d->frame_len.empty()) {           // ACM/VCM: frames of differing length, dummy PLFRAMEs included
                long long n = 0, pos = 0;
                for (int L : d->frame_len) n += L;
                if (h_dst)
                    for (int f = 0; f < nf && pos + d->frame_len[f] <= cap; pos += d->frame_len[f], ++f)
                        HIP_TRY(hipMemcpy((cf32*)h_dst + pos, d->frame_ptrs[f], sizeof(cf32) * d->frame_len[f], hipMemcpyDeviceToHost));
                return (int)n;
            }
            int n = nf * raw;
            if (h_dst)
                for (int f = 0; f < nf && (f + 1) * raw <= cap; ++f)
                    HIP_TRY(hipMemcpy((cf32*)h_dst + (size_t)f * raw, d->frame_ptrs[f], sizeof(cf32) * raw, hipMemcpyDeviceToHost));
            return n;
        }
        case 2: {
            if (d->tap_pll_count >= 0 && d->cfg.acm_vcm) {
                int n = (int)d->tap_pll_count;
                if (h_dst && n && d->tap_pll) HIP_TRY(hipMemcpy(h_dst, d->tap_pll, sizeof(cf32) * std::min(n, cap), hipMemcpyDeviceToHost));
                return n;
            }
            int n = nf * raw;
            if (h_dst && n && d->tap_pll) {
                if (d->tap_pll_stride > raw) {       // frames in slots wider than this stream's PLFRAME (mixed batches): frame by frame
                    const int whole = std::min(nf, cap / raw);
                    if (whole) HIP_TRY(hipMemcpy2D(h_dst, sizeof(cf32) * raw, d->tap_pll, sizeof(cf32) * d->tap_pll_stride, sizeof(cf32) * raw, whole, hipMemcpyDeviceToHost));
                } else {
                    HIP_TRY(hipMemcpy(h_dst, d->tap_pll, sizeof(cf32) * std::min(n, cap), hipMemcpyDeviceToHost));
                }
            }
            return n;
        }
        case 3: {
            if (d->tap_llr_count >= 0 && d->cfg.acm_vcm) {
                int n = (int)d->tap_llr_count;
                if (h_dst && n && d->tap_llr) HIP_TRY(hipMemcpy(h_dst, d->tap_llr, (size_t)std::min(n, cap), hipMemcpyDeviceToHost));
                return n;
            }
            int n = nf * N;
            if (h_dst && n && d->tap_llr) HIP_TRY(hipMemcpy(h_dst, d->tap_llr, (size_t)std::min(n, cap), hipMemcpyDeviceToHost));
            return n;
        }
    }
    return DVBS2GPU_ERR_ARG;
}

int dvbs2gpu_demap_batch(dvbs2gpu_ctx* ctx, int modcod, int shortframes, int pilots, const float* d_frames, int nframes,
                         int8_t* d_llr, void* stream) {
    if (!ctx || nframes < 0) return DVBS2GPU_ERR_ARG;
    ModcodParams mp;
    if (!modcod_params(modcod, shortframes, pilots, &mp)) { last_error() = "unsupported MODCOD"; return DVBS2GPU_ERR_MODCOD; }
    if (nframes == 0) return 0;
    if (!d_frames || !d_llr) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    ConstelTables* CT;
    int rc = get_constel(ctx, mp, &CT);
    if (rc) return rc;
    HIP_TRY(s2_demap_launch(CT->dev, mp.rate, mp.shortframe, mp.slots, mp.pilots, mp.plframe, (const cf32*)d_frames, nframes, d_llr,
                            mp.fec.N, (hipStream_t)stream));
    return 0;
}

int dvbs2gpu_deinterleave_batch(dvbs2gpu_ctx* ctx, int modcod, int shortframes, const int8_t* d_in, int nframes, int8_t* d_out, void* stream) {
    if (!ctx || nframes < 0) return DVBS2GPU_ERR_ARG;
    ModcodParams mp;
    if (!modcod_params(modcod, shortframes, 0, &mp)) { last_error() = "unsupported MODCOD"; return DVBS2GPU_ERR_MODCOD; }
    if (nframes == 0) return 0;
    if (!d_in || !d_out || d_in == d_out) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(s2_deinterleave_launch(mp.constel, mp.rate, mp.bits, mp.fec.N, d_in, nframes, d_out, (hipStream_t)stream));
    return 0;
}

int dvbs2gpu_math_eval(dvbs2gpu_ctx* ctx, int func, int n, const float* d_a, const float* d_b, float* d_out0, float* d_out1, void* stream) {
    if (!ctx || n < 0 || func < 0 || func > 5) return DVBS2GPU_ERR_ARG;
    if (n == 0) return 0;
    if (!d_a || !d_out0 || ((func == 1 || func == 5) && !d_b) || (func == 0 && !d_out1)) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(math_eval_launch(func, n, d_a, d_b, d_out0, d_out1, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
