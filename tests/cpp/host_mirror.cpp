// Test program for include/dvbs2gpu_host.hpp: drives the C++ mirrors of the reference's operators the way the plugin's worker threads
// do (process() per buffer of samples, work() on what came out) on files written by tests/test_host_cpp.py.
//   host_mirror s2   <iq.cf32> <out.bb> <modcod> <short> <pilots> <chunk>     DVBS2Demod::process per chunk      -> BBFRAMEs
//   host_mirror bbts <in.bb>   <out.ts> <kbch_bits> <frames_per_call>          BBFrameTSParser::work              -> TS packets
//   host_mirror dvbs <iq.cf32> <out.ts> <chunk>                                DVBSDemod::process per chunk       -> TS packets
//   host_mirror s2x2 <iqA> <outA> <modcodA> <shortA> <pilotsA> <iqB> <outB> <modcodB> <shortB> <pilotsB> <chunk>
//                    two DVBS2Demod blocks on two worker threads at once (two plugin instances share the engine context)
//   host_mirror fleet <table.txt> <egress.bin> <members> <chunk> <pipelined> <forced_iters>
//                    dvbs2gpu_host::Fleet: the table's transponders ("modcod short pilots iq-file" per line) over `members` logical devices (device = member index
//                    modulo the box's device count), fed `chunk` samples per transponder and call; egress.bin = per call, per transponder in TABLE order: int32 bytes + BBFRAMEs
// Prints one status line per mode; exit code 0 = ran, 2 = usage, 3 = exception (e.g. no GPU: there is no CPU fallback).
#include <dvbs2gpu_host.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

using namespace dvbs2gpu_host;

static std::vector<char> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int handler_calls = 0, handler_symbols = 0;
static void constellation_handler(complex_t*, int count, void* ctx) {
    ++handler_calls;
    handler_symbols += count;
    if (ctx != &handler_calls) std::abort();
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        const std::string mode = argv[1];
        std::vector<char> in = slurp(argv[2]);
        std::ofstream out(argv[3], std::ios::binary);
        std::vector<uint8_t> obuf(STREAM_BUFFER_SIZE);
        // the plugin's defaults (main.cpp:64-73,134-140): clock-recovery gains from bandwidth 0.00628 and damping 0.707
        const float bw = 0.00628f, damp = 0.707f, den = (1.0f + 2.0 * damp * bw + bw * bw);
        const double mu_gain = (4.0f * damp * bw) / den, omega_gain = (4.0f * bw * bw) / den;
        if (mode == "s2" && argc == 8) {
            const int modcod = atoi(argv[4]), shortframes = atoi(argv[5]), pilots = atoi(argv[6]), chunk = atoi(argv[7]);
            dvbs2::DVBS2Demod demod;
            demod.init(2e6, 4e6, 0.0001f, 0.35f, 65, 0.00628f, 0.006f, omega_gain, mu_gain, constellation_handler, &handler_calls, modcod, shortframes != 0,
                       pilots != 0, 0.6f, 16, 0.02);
            const complex_t* iq = reinterpret_cast<const complex_t*>(in.data());
            const long n = (long)(in.size() / sizeof(complex_t));
            long total = 0;
            for (long a = 0; a < n; a += chunk) {
                const int got = demod.process((int)std::min<long>(chunk, n - a), iq + a, obuf.data());
                out.write(reinterpret_cast<const char*>(obuf.data()), got);
                total += got;
            }
            std::printf("s2 bytes=%ld kbch=%d detected_modcod=%d short=%d pilots=%d match=%.1f trials=%.0f bch=%.0f handler_calls=%d handler_symbols=%d\n", total,
                        demod.getKBCH(), demod.detected_modcod, (int)demod.detected_shortframes, (int)demod.detected_pilots, demod.pl_sync_best_match,
                        demod.ldpc_trials, demod.bch_corrections, handler_calls, handler_symbols);
            // error behaviour of the reference: a MODCOD outside the table throws, the block keeps working with the old parameters
            bool threw = false;
            try { demod.setDemodParams(99, false, false, 0.6f, 25); } catch (const std::runtime_error&) { threw = true; }
            std::printf("bad_modcod_throws=%d kbch_after=%d\n", (int)threw, demod.getKBCH());
        } else if (mode == "s2x2" && argc == 13) {
            const int chunk = atoi(argv[12]);
            struct Job { const char* in; const char* out; int modcod, sh, pil; long total; std::string err; };
            Job jobs[2] = {{argv[2], argv[3], atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), 0, ""}, {argv[7], argv[8], atoi(argv[9]), atoi(argv[10]), atoi(argv[11]), 0, ""}};
            out.close();
            auto worker = [&](Job* j) {
                try {
                    std::vector<char> data = slurp(j->in);
                    std::ofstream o(j->out, std::ios::binary);
                    std::vector<uint8_t> buf(STREAM_BUFFER_SIZE);
                    dvbs2::DVBS2Demod demod;
                    demod.init(2e6, 4e6, 0.0001f, 0.35f, 65, 0.00628f, 0.006f, omega_gain, mu_gain, nullptr, nullptr, j->modcod, j->sh != 0, j->pil != 0, 0.6f, 16, 0.02);
                    const complex_t* iq = reinterpret_cast<const complex_t*>(data.data());
                    const long n = (long)(data.size() / sizeof(complex_t));
                    for (long a = 0; a < n; a += chunk) {
                        const int got = demod.process((int)std::min<long>(chunk, n - a), iq + a, buf.data());
                        o.write(reinterpret_cast<const char*>(buf.data()), got);
                        j->total += got;
                    }
                } catch (const std::exception& e) { j->err = e.what(); }
            };
            std::thread ta(worker, &jobs[0]), tb(worker, &jobs[1]);
            ta.join(); tb.join();
            if (!jobs[0].err.empty() || !jobs[1].err.empty()) throw std::runtime_error(jobs[0].err + " " + jobs[1].err);
            std::printf("s2x2 bytesA=%ld bytesB=%ld\n", jobs[0].total, jobs[1].total);
        } else if (mode == "fleet" && argc == 8) {
            const int members = atoi(argv[4]), chunk = atoi(argv[5]), pipelined = atoi(argv[6]), forced = atoi(argv[7]);
            const int ndev = dvbs2gpu_device_count();
            if (ndev < 1) throw std::runtime_error("no HIP device visible (this engine has no CPU fallback)");
            std::vector<int> devices;
            for (int i = 0; i < members; ++i) devices.push_back(i % ndev);
            std::vector<dvbs2gpu_fleet_entry> table;
            std::vector<std::vector<char>> sig;
            {
                std::string text(in.begin(), in.end());
                size_t pos = 0;
                while (pos < text.size()) {
                    size_t end = text.find('\n', pos);
                    if (end == std::string::npos) end = text.size();
                    const std::string ln = text.substr(pos, end - pos);
                    pos = end + 1;
                    int m, sh, pil; char path[1024];
                    if (std::sscanf(ln.c_str(), "%d %d %d %1023s", &m, &sh, &pil, path) != 4) continue;
                    table.push_back(Fleet::entry(m, sh != 0, pil != 0, chunk, 16, forced));
                    sig.push_back(slurp(path));
                }
            }
            const int nt = (int)table.size();
            const int cap = 1 << 20;
            Fleet fleet(devices);
            const std::vector<int> where = fleet.assign(table, cap);
            fleet.setPipelined(pipelined != 0);
            std::vector<std::vector<uint8_t>> obufs((size_t)nt, std::vector<uint8_t>((size_t)cap));
            std::vector<uint8_t*> outs;
            for (auto& b : obufs) outs.push_back(b.data());
            long longest = 0;
            for (auto& s : sig) longest = std::max<long>(longest, (long)(s.size() / sizeof(complex_t)));
            long total = 0, calls = 0;
            for (long a = 0; a < longest + (pipelined ? chunk : 0); a += chunk, ++calls) {       // (pipelined: one more call with all counts 0 collects the last frames)
                std::vector<const complex_t*> ins((size_t)nt);
                std::vector<int> cnt((size_t)nt);
                for (int t = 0; t < nt; ++t) {
                    const long n = (long)(sig[(size_t)t].size() / sizeof(complex_t));
                    ins[(size_t)t] = reinterpret_cast<const complex_t*>(sig[(size_t)t].data()) + std::min(a, n);
                    cnt[(size_t)t] = (int)std::max<long>(0, std::min<long>(chunk, n - a));
                }
                const std::vector<int> nb = fleet.process(ins, cnt, outs);
                for (int t = 0; t < nt; ++t) {
                    const int32_t b = nb[(size_t)t];
                    out.write(reinterpret_cast<const char*>(&b), sizeof b);
                    out.write(reinterpret_cast<const char*>(obufs[(size_t)t].data()), b);
                    total += b;
                }
            }
            std::printf("fleet members=%d devices=%d transponders=%d calls=%ld bytes=%ld placement=", fleet.size(), ndev, nt, calls, total);
            for (int t = 0; t < nt; ++t) std::printf("%d%s", where[(size_t)t], t + 1 < nt ? "," : "\n");
        } else if (mode == "bbts" && argc == 6) {
            const int kbch = atoi(argv[4]), per_call = atoi(argv[5]);
            dvbs2::BBFrameTSParser parser;
            parser.setFrameSize(kbch);
            const int fb = kbch / 8;
            const int nfr = (int)(in.size() / fb);
            long total = 0;
            for (int a = 0; a < nfr; a += per_call) {
                const int cnt = std::min(per_call, nfr - a);
                const int got = parser.work(reinterpret_cast<uint8_t*>(in.data()) + (size_t)a * fb, cnt, obuf.data(), (int)obuf.size());
                out.write(reinterpret_cast<const char*>(obuf.data()), got);
                total += got;
            }
            std::printf("bbts bytes=%ld ts_gs=%d upl=%d dfl=%d last_bb_cnt=%d last_bb_proc=%d gse_crc_err=%d\n", total, parser.last_header.ts_gs,
                        parser.last_header.upl, parser.last_header.dfl, parser.last_bb_cnt, parser.last_bb_proc, (int)parser.last_gse_crc_err);
        } else if (mode == "dvbs" && argc == 5) {
            const int chunk = atoi(argv[4]);
            dvbs::DVBSDemod demod;
            demod.init(2e6, 4e6, 0.0001f, 0.35f, 65, 0.00628f, 0.006f, omega_gain, mu_gain, constellation_handler, &handler_calls, 0.02);
            const complex_t* iq = reinterpret_cast<const complex_t*>(in.data());
            const long n = (long)(in.size() / sizeof(complex_t));
            long total = 0;
            for (long a = 0; a < n; a += chunk) {
                const int got = demod.process((int)std::min<long>(chunk, n - a), iq + a, obuf.data());
                out.write(reinterpret_cast<const char*>(obuf.data()), got);
                total += got;
            }
            std::printf("dvbs bytes=%ld lock=%d rate=%s ber=%.4f rs_avg=%.0f deframer_err=%d handler_calls=%d\n", total, demod.stats_viterbi_lock,
                        demod.stats_viterbi_rate.c_str(), demod.stats_viterbi_ber, demod.stats_rs_avg, demod.stats_deframer_err, handler_calls);
        } else {
            return 2;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_mirror: %s\n", e.what());
        return 3;
    }
    return 0;
}
