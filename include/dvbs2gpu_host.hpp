/*
 * dvbs2gpu_host.hpp -- C++ host side above the C ABI (include/dvbs2gpu.h): the operator interface of the reference's hot path,
 * with the reference's names, argument meaning and error behaviour, for hosts that are C++ like the SDR++ plugin.
 *
 *   dvbs2gpu_host::dvbs2::DVBS2Demod       <->  dsp::dvbs2::DVBS2Demod        src/demod/dvbs2/module_dvbs2_demod.h:49-88
 *   dvbs2gpu_host::dvbs2::BBFrameTSParser  <->  dsp::dvbs2::BBFrameTSParser   src/demod/dvbs2/bbframe_ts_parser.h:68-80
 *   dvbs2gpu_host::dvbs::DVBSDemod         <->  dsp::dvbs::DVBSDemod          src/demod/dvbs/module_dvbs_demod.h:17-52
 *
 * What is NOT mirrored is SDR++'s block plumbing (Processor<>, stream<>, run(), tempStop/tempStart): those headers are not part of
 * the reference repository; INTEGRATION.md shows the same classes derived from Processor<> inside the plugin.  `process` has the
 * reference's signature and return value; the stream pointer argument of init() is dropped, everything else keeps its position.
 * Errors: the reference throws std::runtime_error for a bad MODCOD (modcod_to_cfg.cpp:11,135); every failing C-ABI call throws
 * std::runtime_error with dvbs2gpu_last_error() here.  There is no CPU fallback: without a gfx950 device init() throws.
 * Header-only; link with -ldvbs2gpu.
 */
#ifndef DVBS2GPU_HOST_HPP
#define DVBS2GPU_HOST_HPP

#include <dvbs2gpu.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace dvbs2gpu_host {

struct complex_t {   // layout of SDR++'s dsp::complex_t: two floats
    float re, im;
};

constexpr int STREAM_BUFFER_SIZE = 1000000;   // SDR++ core: the largest count a process() call is handed

inline void check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("dvbs2gpu: ") + dvbs2gpu_last_error() + " (" + std::to_string(rc) + ")");
}

/* One engine context per device, shared by every block of the process (the reference's blocks share nothing; here they share the code
 * tables, the decoder plans and the FEC workspaces). */
class Engine {
public:
    static std::shared_ptr<Engine> get(int device = 0) {
        static std::mutex m;
        static std::weak_ptr<Engine> cache[16];
        if (device < 0 || device >= 16) throw std::runtime_error("dvbs2gpu: device index out of range");
        std::lock_guard<std::mutex> l(m);
        auto e = cache[device].lock();
        if (!e) {
            e.reset(new Engine(device));
            cache[device] = e;
        }
        return e;
    }
    ~Engine() { dvbs2gpu_destroy(ctx); }
    dvbs2gpu_ctx* ctx = nullptr;

private:
    explicit Engine(int device) { check(dvbs2gpu_create(device, &ctx)); }
    Engine(const Engine&) = delete;
};

namespace dvbs2 {

class DVBS2Demod {
public:
    DVBS2Demod() {}
    ~DVBS2Demod() { release(); }
    DVBS2Demod(const DVBS2Demod&) = delete;
    DVBS2Demod& operator=(const DVBS2Demod&) = delete;

    /* DVBS2Demod::init (module_dvbs2_demod.cpp:7-30) without the input stream; handler is called once per PL frame a call completes, with that
     * frame's symbols behind the PLL (the constellation display, :337), and may be null */
    void init(double symbolrate, double samplerate, float agc_rate, float rrc_alpha, int rrc_taps, float loop_bw, float fll_bw, double omegaGain,
              double muGain, void (*handler)(complex_t* data, int count, void* ctx), void* ctx, int modcod, bool shortframes, bool pilots,
              float sof_thresold, int max_ldpc_trials, double omegaRelLimit = 0.01, int device = 0) {
        release();
        eng = Engine::get(device);
        cfg.symbolrate = symbolrate; cfg.samplerate = samplerate; cfg.agc_rate = agc_rate; cfg.rrc_alpha = rrc_alpha; cfg.rrc_taps = rrc_taps;
        cfg.loop_bw = loop_bw; cfg.fll_bw = fll_bw; cfg.clock_omega_gain = (float)omegaGain; cfg.clock_mu_gain = (float)muGain;
        cfg.omega_rel_limit = (float)omegaRelLimit; cfg.modcod = modcod; cfg.shortframes = shortframes; cfg.pilots = pilots;
        cfg.sof_threshold = sof_thresold; cfg.max_ldpc_trials = max_ldpc_trials; cfg.force_ldpc_iters = 0;
        d_handler = handler; d_ctx = ctx;
        check(dvbs2gpu_demod_create(eng->ctx, &cfg, STREAM_BUFFER_SIZE, &h));
    }
    void reset() { check(dvbs2gpu_demod_reset(need())); }
    void setDemodParams(int modcod, bool shortframes, bool pilots, float sof_thresold, int max_ldpc_trials) {
        check(dvbs2gpu_demod_set_params(need(), modcod, shortframes, pilots, sof_thresold, max_ldpc_trials));   // a bad MODCOD throws, old parameters stay
        cfg.modcod = modcod; cfg.shortframes = shortframes; cfg.pilots = pilots; cfg.sof_threshold = sof_thresold; cfg.max_ldpc_trials = max_ldpc_trials;
    }
    void setSymbolrate(double symbolrate) { cfg.symbolrate = symbolrate; rebuild(); }   // module_dvbs2_demod.cpp:153-166: new RRC taps, loops reset
    void setSamplerate(double samplerate) { cfg.samplerate = samplerate; rebuild(); }
    int getKBCH() { return dvbs2gpu_demod_get_kbch(need()); }

    /* module_dvbs2_demod.cpp:216: count samples in, the BBFRAMEs completed by this call out (getKBCH()/8 bytes each); returns bytes */
    int process(int count, const complex_t* in, uint8_t* out) {
        const int n = dvbs2gpu_demod_process(need(), count, reinterpret_cast<const float*>(in), out, STREAM_BUFFER_SIZE);
        check(n);
        // every PL frame this call completed (a call of STREAM_BUFFER_SIZE samples of short frames holds more than a fixed-size array would)
        const int k = dvbs2gpu_demod_get_stats(h, nullptr, 0);
        if (k > 0) {   // the public fields the plugin's menu polls (module_dvbs2_demod.h:82-87)
            stats.resize((size_t)k);
            dvbs2gpu_demod_get_stats(h, stats.data(), k);
            const dvbs2gpu_frame_stats& s = stats[(size_t)k - 1];
            detected_modcod = s.detected_modcod; detected_shortframes = s.detected_shortframes != 0; detected_pilots = s.detected_pilots != 0;
            pl_sync_best_match = s.pl_sync_best_match; ldpc_trials = (float)s.ldpc_trials; bch_corrections = (float)s.bch_corrections;
        }
        if (d_handler && k > 0) {
            // module_dvbs2_demod.cpp:337: the handler is called once per PL frame, with that frame's header + payload (+ pilot) symbols behind the PLL.
            // A CCM block's frames all have the configured MODCOD's PLFRAME length (dvbs2gpu_modcod_info): tap 2 holds k of them back to back.
            const int ns = dvbs2gpu_demod_get_tap(h, 2, nullptr, 0);
            dvbs2gpu_modcod_info mi;
            if (ns > 0 && dvbs2gpu_modcod_info_get(cfg.modcod, cfg.shortframes, cfg.pilots, &mi) == 0 && mi.plframe_symbols > 0 &&
                (long long)mi.plframe_symbols * k == ns) {
                tap.resize((size_t)ns);
                dvbs2gpu_demod_get_tap(h, 2, tap.data(), ns);
                for (int f = 0; f < k; ++f) d_handler(tap.data() + (size_t)f * mi.plframe_symbols, mi.plframe_symbols, d_ctx);
            }
        }
        return n;
    }

    int detected_modcod = -1;
    bool detected_shortframes = false;
    bool detected_pilots = false;
    float pl_sync_best_match = 0;
    float ldpc_trials = -1;
    float bch_corrections = -1;

private:
    dvbs2gpu_demod* need() {
        if (!h) throw std::runtime_error("dvbs2gpu: DVBS2Demod used before init()");
        return h;
    }
    void rebuild() {
        if (!h) return;
        dvbs2gpu_demod* n = nullptr;
        check(dvbs2gpu_demod_create(eng->ctx, &cfg, STREAM_BUFFER_SIZE, &n));
        dvbs2gpu_demod_destroy(h);
        h = n;
    }
    void release() {
        if (h) dvbs2gpu_demod_destroy(h);
        h = nullptr;
    }
    std::shared_ptr<Engine> eng;
    dvbs2gpu_demod_cfg cfg{};
    dvbs2gpu_demod* h = nullptr;
    void (*d_handler)(complex_t*, int, void*) = nullptr;
    void* d_ctx = nullptr;
    std::vector<complex_t> tap;
    std::vector<dvbs2gpu_frame_stats> stats;
};

struct BBHeader {   // bbframe_ts_parser.h:36-66
    int ts_gs = 0, sis_mis = 0, ccm_acm = 0, issyi = 0, npd = 0, ro = 0, isi = 0, upl = 0, dfl = 0, sync = 0, syncd = 0;
};

class BBFrameTSParser {
public:
    BBFrameTSParser() {}
    ~BBFrameTSParser() {
        if (h) dvbs2gpu_bbts_destroy(h);
    }
    BBFrameTSParser(const BBFrameTSParser&) = delete;
    BBFrameTSParser& operator=(const BBFrameTSParser&) = delete;

    void setFrameSize(int bbframe_size) {   // bbframe_ts_parser.cpp:31-42 (bits)
        if (!h) {
            eng = Engine::get(device);
            check(dvbs2gpu_bbts_create(eng->ctx, 1, bbframe_size, max_frames, &h));
        } else {
            check(dvbs2gpu_bbts_set_frame_size(h, bbframe_size));
        }
    }
    /* bbframe_ts_parser.cpp:104: cnt BBFRAMEs in, TS packets (or GSE PDUs) out; returns bytes.  buffer_outsize must be at least
     * cnt * kbch/8 + 376 (the reference's own "at least as big as bbframe"; it prints "BUFF OVF!" and stops otherwise: here the call throws) */
    int work(uint8_t* bbframe, int cnt, uint8_t* tsframes, int buffer_outsize) {
        if (!h) throw std::runtime_error("dvbs2gpu: BBFrameTSParser used before setFrameSize()");
        const int n = dvbs2gpu_bbts_work(h, bbframe, cnt, tsframes, buffer_outsize);
        check(n);
        int32_t s[15];
        check(dvbs2gpu_bbts_get_stats(h, 0, s, 15));
        last_header.ts_gs = s[0]; last_header.sis_mis = s[1]; last_header.ccm_acm = s[2]; last_header.issyi = s[3]; last_header.npd = s[4];
        last_header.ro = s[5]; last_header.isi = s[6]; last_header.upl = s[7]; last_header.dfl = s[8]; last_header.sync = s[9]; last_header.syncd = s[10];
        last_gse_crc_err = s[11] != 0; last_bb_cnt = s[12]; last_bb_proc = s[13]; last_ts_errs = s[14];
        return n;
    }

    BBHeader last_header;
    bool last_gse_crc_err = 0;
    int last_bb_cnt = 0;
    int last_bb_proc = 0;
    int last_ts_errs = 0;

    int device = 0;
    int max_frames = 64;   // most BBFRAMEs one work() call may bring (the plugin's sink handler passes what one process() call produced)

private:
    std::shared_ptr<Engine> eng;
    dvbs2gpu_bbts* h = nullptr;
};

}   // namespace dvbs2

namespace dvbs {

class DVBSDemod {
public:
    DVBSDemod() {}
    ~DVBSDemod() { release(); }
    DVBSDemod(const DVBSDemod&) = delete;
    DVBSDemod& operator=(const DVBSDemod&) = delete;

    /* DVBSDemod::init (module_dvbs_demod.cpp:9-31) without the input stream; handler receives the symbols after the Costas loop (:33-37) */
    void init(double symbolrate, double samplerate, float agc_rate, float rrc_alpha, int rrc_taps, float loop_bw, float fll_bw, double omegaGain,
              double muGain, void (*handler)(complex_t* data, int count, void* ctx), void* ctx, double omegaRelLimit = 0.01, int device = 0) {
        release();
        eng = Engine::get(device);
        dvbs2gpu_dvbs_demod_default_cfg(&cfg);
        cfg.symbolrate = symbolrate; cfg.samplerate = samplerate; cfg.agc_rate = agc_rate; cfg.rrc_alpha = rrc_alpha; cfg.rrc_taps = rrc_taps;
        cfg.loop_bw = loop_bw; cfg.fll_bw = fll_bw; cfg.clock_omega_gain = (float)omegaGain; cfg.clock_mu_gain = (float)muGain;
        cfg.omega_rel_limit = (float)omegaRelLimit;
        d_handler = handler; d_ctx = ctx;
        build();
    }
    void reset() {   // module_dvbs_demod.cpp:58-64
        check(dvbs2gpu_dvbs_demod_reset(need()));
        check(dvbs2gpu_dvbs_tail_reset(t));
    }
    void setSymbolrate(double symbolrate) { cfg.symbolrate = symbolrate; if (h) { release(); build(); } }
    void setSamplerate(double samplerate) { cfg.samplerate = samplerate; if (h) { release(); build(); } }

    /* module_dvbs_demod.cpp:78-117: count samples in, TS packets out (188 bytes each); returns bytes */
    int process(int count, const complex_t* in, uint8_t* out) {
        const int n = dvbs2gpu_dvbs_process_ts(need(), t, count, reinterpret_cast<const float*>(in), out, STREAM_BUFFER_SIZE);
        check(n);
        dvbs2gpu_viterbi_stats vs;
        check(dvbs2gpu_dvbs_demod_get_stats(h, &vs));
        stats_viterbi_ber = vs.ber; stats_viterbi_lock = vs.state;
        static const char* const names[5] = {"1/2", "2/3", "3/4", "5/6", "7/8"};
        stats_viterbi_rate = vs.rate >= 0 && vs.rate < 5 ? names[vs.rate] : "";
        int32_t ts[11];
        check(dvbs2gpu_dvbs_tail_get_stats(t, 0, ts));
        if (ts[0] > 0) {   // :115-116, from the last frame of the call
            int sum = 0;
            for (int i = 0; i < 8; ++i) sum += ts[3 + i];
            stats_rs_avg = (float)(sum / 8);   // (an integer mean in the reference too: int errors[8])
        }
        stats_deframer_err = std::min(ts[1], ts[2]);
        if (d_handler) {
            const int ns = dvbs2gpu_dvbs_demod_get_tap(h, 0, 0, nullptr, 0);
            if (ns > 0) {
                tap.resize((size_t)ns);
                dvbs2gpu_dvbs_demod_get_tap(h, 0, 0, tap.data(), ns);
                d_handler(tap.data(), ns, d_ctx);
            }
        }
        return n;
    }

    float stats_viterbi_ber = 0;
    int stats_viterbi_lock = 0;
    std::string stats_viterbi_rate = "";
    float stats_rs_avg = 0;
    int stats_deframer_err = 0;

private:
    dvbs2gpu_dvbs_demod* need() {
        if (!h) throw std::runtime_error("dvbs2gpu: DVBSDemod used before init()");
        return h;
    }
    void build() {
        check(dvbs2gpu_dvbs_demod_create(eng->ctx, &cfg, 1, STREAM_BUFFER_SIZE, &h));
        const int rc = dvbs2gpu_dvbs_tail_create(eng->ctx, 1, STREAM_BUFFER_SIZE + 4 * 8192, &t);
        if (rc < 0) { release(); check(rc); }
    }
    void release() {
        if (h) dvbs2gpu_dvbs_demod_destroy(h);
        if (t) dvbs2gpu_dvbs_tail_destroy(t);
        h = nullptr; t = nullptr;
    }
    std::shared_ptr<Engine> eng;
    dvbs2gpu_dvbs_cfg cfg{};
    dvbs2gpu_dvbs_demod* h = nullptr;
    dvbs2gpu_dvbs_tail* t = nullptr;
    void (*d_handler)(complex_t*, int, void*) = nullptr;
    void* d_ctx = nullptr;
    std::vector<complex_t> tap;
};

}   // namespace dvbs

/* The transponders of one host over several GPUs (dvbs2gpu_fleet_*, include/dvbs2gpu.h): the reference runs one independent DVBS2Demod per transponder
 * (src/main.cpp:588,595); a Fleet places a table of them on its members -- one engine context + worker thread per device -- by the MODCOD-grouped
 * longest-processing-time rule and returns every call's BBFRAMEs in table order.  Every failing call throws std::runtime_error, as above. */
class Fleet {
public:
    explicit Fleet(const std::vector<int>& devices) { check(dvbs2gpu_fleet_create(devices.data(), (int)devices.size(), &f)); }
    ~Fleet() { dvbs2gpu_fleet_destroy(f); }
    Fleet(const Fleet&) = delete;
    Fleet& operator=(const Fleet&) = delete;

    /* one table entry per transponder: DVBS2Demod::init's parameters as the plugin sets them (main.cpp:64-73,134-140), cfg.modcod etc. per transponder */
    static dvbs2gpu_fleet_entry entry(int modcod, bool shortframes, bool pilots, int max_samples, int max_ldpc_trials = 16, int force_ldpc_iters = 0, double weight = 0.0) {
        dvbs2gpu_fleet_entry e{};
        dvbs2gpu_demod_default_cfg(modcod, shortframes, pilots, &e.cfg);
        e.cfg.max_ldpc_trials = max_ldpc_trials; e.cfg.force_ldpc_iters = force_ldpc_iters;
        e.weight = weight; e.max_samples = max_samples;
        return e;
    }
    /* places the table; returns the member of every transponder */
    std::vector<int> assign(const std::vector<dvbs2gpu_fleet_entry>& table, int out_cap, double tolerance = 0.25) {
        std::vector<int32_t> m(table.size());
        check(dvbs2gpu_fleet_assign(f, table.data(), (int)table.size(), out_cap, tolerance, m.data()));
        nt = (int)table.size(); cap = out_cap;
        return std::vector<int>(m.begin(), m.end());
    }
    void setPipelined(bool on) { check(dvbs2gpu_fleet_set_pipelined(f, on ? 1 : 0)); }
    void reset() { check(dvbs2gpu_fleet_reset(f)); }
    /* one call for all transponders, in table order: in[i] / count[i] samples of transponder i, out[i] receives its BBFRAMEs (capacity: assign's out_cap); returns the byte counts */
    std::vector<int> process(const std::vector<const complex_t*>& in, const std::vector<int>& count, const std::vector<uint8_t*>& out) {
        if ((int)in.size() != nt || (int)count.size() != nt || (int)out.size() != nt) throw std::runtime_error("dvbs2gpu: Fleet::process needs one entry per transponder");
        std::vector<int> nb((size_t)nt, 0);
        check(dvbs2gpu_fleet_process_batch(f, reinterpret_cast<const float* const*>(in.data()), count.data(), out.data(), cap, nb.data()));
        return nb;
    }
    int size() const { return dvbs2gpu_fleet_size(f); }

private:
    dvbs2gpu_fleet* f = nullptr;
    int nt = 0, cap = 0;
};
}   // namespace dvbs2gpu_host
#endif
