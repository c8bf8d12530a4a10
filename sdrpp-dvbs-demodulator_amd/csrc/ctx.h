// Engine context internals shared by capi.hip (FEC entry points) and s2_demod.hip (demodulator handles).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/dvbs2gpu.h"
#include "s2_params.h"
#include "ldpc_plan.h"
#include "ldpc_wave_plan.h"
#include "ldpc_split_plan.h"
#include <chrono>
#include "kernels.h"
#include "s2_rx.h"

namespace s2 {

std::string& last_error();
int fail_hip(hipError_t e, const char* what);
#define HIP_TRY(x)                                            \
    do {                                                      \
        hipError_t _e = (x);                                  \
        if (_e != hipSuccess) return s2::fail_hip(_e, #x);    \
    } while (0)

struct Workspace {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        size_t want = n + n / 4;   // grow with slack so alternating sizes do not thrash
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail_hip(e, "hipMalloc(workspace)");
        bytes = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

template <typename T>
inline int upload(const std::vector<T>& v, T** dptr) {
    *dptr = nullptr;
    size_t n = v.size() * sizeof(T);
    if (!n) n = sizeof(T);
    HIP_TRY(hipMalloc((void**)dptr, n));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// Per-stage device times (hipEvent pairs on the stream a stage is enqueued on), accumulated until dvbs2gpu_get_stage_times reads them.
enum StageId { ST_FRONTEND = 0, ST_RRC, ST_PLSYNC, ST_LOOPS, ST_DEMAP, ST_LDPC, ST_BCH, ST_DELIVER, ST_COUNT };
struct StageTimers {
    bool on = false;
    std::mutex mtx;
    struct Span { int stage; hipEvent_t a, b; long units; };
    std::vector<Span> pending;
    std::vector<hipEvent_t> pool;
    double ms[ST_COUNT] = {};
    long launches[ST_COUNT] = {}, units[ST_COUNT] = {};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
// RAII span: records an event on `st` now and another when it goes out of scope (no-op while timing is off)
struct StageSpan {
    StageTimers* T; int stage; hipStream_t st; hipEvent_t a = nullptr; long units;
    StageSpan(StageTimers& t, int stage_, hipStream_t st_, long units_ = 0) : T(&t), stage(stage_), st(st_), units(units_) {
        if (!T->on) { T = nullptr; return; }
        { std::lock_guard<std::mutex> l(T->mtx); a = T->get(); }
        if (a) (void)hipEventRecord(a, st);
    }
    ~StageSpan() {
        if (!T || !a) return;
        std::lock_guard<std::mutex> l(T->mtx);
        hipEvent_t b = T->get();
        if (b) { (void)hipEventRecord(b, st); T->pending.push_back({stage, a, b, units}); }
    }
};

// the FEC stage's scratch buffers: one set per context, and one per configuration group for jobs that run side by side (s2_demod.hip)
struct FecWs { Workspace msg, hard, syn, misc; void release() { msg.release(); hard.release(); syn.release(); misc.release(); } };

struct ConstelTables {          // device tables of one constellation (type, gamma1, gamma2)
    S2ConstelDev dev;
    int8_t* d_bits = nullptr;
    uint32_t* d_bits4 = nullptr;
    float* d_err = nullptr;
    cf32* d_pts = nullptr;
};

}  // namespace s2

struct dvbs2gpu_ctx {
    int device = 0;
    int num_cus = 256;
    std::mutex mtx;                           // table caches (creation of LDPC / BCH / constellation / tap tables)
    // One context may serve several handles and host threads (include/dvbs2gpu_host.hpp shares one per device between all blocks of
    // the process), and every entry point below works in the context-wide workspaces (ws_*): whole calls are serialised by
    // call_mtx (recursive: the segment receivers call process_batch from inside their own entry point).  The asynchronous stage
    // entry points only ENQUEUE under the lock; ev_ws, recorded behind the last enqueued user of the shared FEC workspaces, makes a
    // later user on a different stream wait for the earlier one (the message records and the LDPC work counter live there).
    s2::StageTimers timers;
    std::recursive_mutex call_mtx;
    hipEvent_t ev_ws = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;
    std::map<int, s2::LdpcDeviceCode> ldpc;   // by code_index
    std::map<int, s2::BchDeviceCode> bch;     // by m*100 + t
    uint8_t* d_prbs = nullptr;                // BB scrambler sequence, 8100 bytes
    s2::FecWs fws;                            // LDPC message records (+ work counter, sign scratch), hard decisions, BCH syndromes, trial counts
    // receive-chain tables (s2_demod.hip)
    float* d_gardner_bank = nullptr;
    s2::S2PlTablesDev pl{};
    std::map<int, s2::ConstelTables> constel; // by modcod (gammas depend on it)
    std::map<int, float*> rrc;                // by ntaps*1000 + round(alpha*100) (Ts = 2)
    s2::Workspace ws_rx[8];
    // ACM/VCM mode (s2_demod.hip): what every PLS code means + the constellations it points to, on the device; per-call scratch
    s2::S2VcmMod* d_vcm_mods = nullptr;
    s2::S2ConstelDev* d_vcm_cons = nullptr;
    std::vector<s2::S2VcmMod> h_vcm_mods;
    std::vector<s2::FecParams> h_vcm_fec;     // by PLS code
    s2::Workspace ws_vcm[12];
    // mixed CCM batches through one launch per stage (s2_demod.hip, process_mixed): per-call scratch, and per FEC side stream and job parity the event behind its jobs
    s2::Workspace ws_mix[12];
    hipEvent_t ev_mix[8][2] = {};
    // pipelined FEC (s2_demod.hip): with pipeline_fec set, dvbs2gpu_demod_process_batch runs the FEC of call k on fec_stream
    // while call k+1's front end runs on fe_stream; BBFRAMEs of call k are delivered by call k+1
    int pipeline_fec = 0;
    hipStream_t fe_stream = nullptr, fec_stream = nullptr;
    hipEvent_t ev_llr = nullptr;
    hipEvent_t ev_in = nullptr;               // throughput mode: the call's own stream starts behind what the host has put on the legacy null stream (its input buffers)
    // one slot per configuration group of the batch (groups are formed in order of first appearance, so a slot keeps its
    // streams from call to call): the job in flight, its buffers (double-buffered) and the event that marks its completion
    static constexpr int MAX_PIPE_GROUPS = 16;
    void* pending_fec[MAX_PIPE_GROUPS] = {};              // s2::PendingFec*
    s2::Workspace ws_fecbuf[MAX_PIPE_GROUPS][2][3];       // per parity: LLRs | BBFRAMEs | frame refs + first[] + trials + corrections
    int fec_parity[MAX_PIPE_GROUPS] = {};
    // what the bench's self-check reads back (dvbs2gpu_debug_last_fec_job): the pipelined CCM job a slot delivered last -- its LLR and BBFRAME buffers stay untouched until the
    // slot's next job of the same parity is started
    struct LastFecJob { const int8_t* d_llr = nullptr; const uint8_t* d_bb = nullptr; int nf = 0, n = 0, N = 0, kb = 0, rate = -1, shortframe = 0, max_trials = 0, force = 0; std::vector<int> first; std::vector<const void*> dm; };
    LastFecJob last_fec[MAX_PIPE_GROUPS];
    bool last_call_staged = false;            // the last CCM batch ran its post stages behind every front-end slice (the stage pipeline)
    hipEvent_t ev_fec[MAX_PIPE_GROUPS][2] = {};   // end of a group's FEC job, per job parity (the next job is enqueued before the previous one is delivered)
    hipEvent_t ev_fec_t0[MAX_PIPE_GROUPS][2] = {};   // ... and its start (timing events: the job's duration feeds the rule below)
    // THE PLUGIN'S MODE BY SPACE (s2_demod.hip, process_group): a second FEC stream confined to FEC_PART_CUS compute units.  A single-configuration batch whose decoder job is long
    // done when the next call's front end is through AND would still fit into a call period on that many units gets its jobs there: the front end, which is that batch's critical
    // path, then shares fewer units with decoder workgroups.  Jobs on the two streams never overlap (they share the FEC workspaces): fec_last_done / fec_last_stream.
    hipStream_t fec_part_stream = nullptr;
    hipEvent_t fec_last_done = nullptr;
    hipStream_t fec_last_stream = nullptr;
    // OFF BY DEFAULT since the third part of round 6: hipExtStreamCreateWithCUMask makes a BLOCKING stream, and every operation on the legacy null stream waits for the jobs on a
    // blocking stream -- the host's own null-stream work did, and so does the event that orders a throughput-mode call behind the host's input producers
    // (s2_demod.hip: ev_in): with the rule on, the front end of call k + 1 waited for the decoder job of call k (plugin's mode 118.7 -> 171.6 ms per step).  Without the rule
    // the plugin's mode takes 122.5 ms (it bought 3 %); fec_part = -1 / 1 still select it for hosts that keep everything off the null stream and synchronise their inputs.
    int fec_part = 0;                         // option fec_part: -1 by the rule, 0 never (default), 1 every big job of a single-configuration batch
    bool fec_part_on = false;
    int fec_part_trend = 0;
    std::chrono::steady_clock::time_point fec_last_entry{};
    // several groups of one pipelined batch run their MODCOD-dependent stages side by side (one host thread and HIP stream each)
    s2::Workspace ws_grp[MAX_PIPE_GROUPS][8];
    hipStream_t grp_stream[MAX_PIPE_GROUPS] = {};
    hipEvent_t ev_llr_grp[MAX_PIPE_GROUPS] = {};
    std::mutex fec_mtx;                                   // FEC jobs on the shared stream are enqueued whole, one at a time (shared FEC workspaces)
    // FEC jobs too small to fill the device (a group of a 64-transponder batch: a handful of decoder workgroups, 4-6 ms of latency each) run
    // SIDE BY SIDE instead: on the group's own stream (grp_stream) with a set of FEC workspaces per group
    s2::FecWs fws_grp[MAX_PIPE_GROUPS];
    // time-sliced front end (s2_rx_kernels.hip, s2_frontend_launch): per main stream one auxiliary stream + the slice events
    struct FeAux { hipStream_t aux = nullptr, aux2 = nullptr, aux3 = nullptr; hipEvent_t ev[s2::S2_FE_MAX_SLICES + 1] = {}, ev2[s2::S2_FE_MAX_SLICES + 1] = {}, ev3[2 * (s2::S2_FE_MAX_SLICES + 1)] = {}; hipStream_t dvbs_aux[4] = {}; hipEvent_t dvbs_ev[4][s2::DVBS_FE_MAX_SLICES + 1] = {}; };   // (dvbs_*: the DVB-S receiver's stage streams: AGC, FLL, RRC, soft FIFO + Viterbi)
    std::map<hipStream_t, FeAux> fe_aux;
    // development / test options (dvbs2gpu_set_option, or DVBS2GPU_OPTIONS="name=value,..." in the environment when the context is created)
    int loops_ahead = 1;                      // 0: frame loops only behind the PL sync (small banks)
    int mixed_groups = 0;                     // 1: small mixed batches through the per-group flow instead of process_mixed
    int mix_fec_streams = 4;                  // side streams for the FEC jobs of process_mixed (1..8)
    int gardner_form = 0;                     // 1 / 2 / 4: one form of the timing recovery (0: chosen by bank size and balance)
    int gardner_cand_skew = 0;                // tests only: skews the candidate form's arm prediction so that it leaves its tables
    int ldpc_wave = -1;                       // short frames: 0 / 1 = lane-per-row / wave-per-frame decoder (-1: per code)
    int ldpc_split_fail_attempts = 0;         // tests only: every attempt of the half-row decoder's layers with shared bits is made to fail (ldpc_split_kernel.hip: the long way must give the same bits)
    int ldpc_split = 1;                       // 1: the half-row decoder for the normal frames it takes (ldpc_split_kernel.hip); 0: the lane-per-row decoder for every code
    int host_timing = 0;                      // 1: print where the host spends a batch call
    int stage_pipeline_launches = 0;          // option stage_loops: frame-loop launches per call (0 = chosen per call, s2_demod.hip)
    int stage_pipeline_min_duty = 2;          // option stage_min_duty: pipelined mode uses the stage pipeline only above this balancer setting (-1: always).
                                              // 2: with the frame loops at the decoder's base priority the two streams of a headline step are 6 ms apart, a stray
                                              // verdict of the balancer must not tip the step into the other flow (which costs it 3 %)
    int stage_post_stream = 1;                // option stage_post_stream: synchronous mode runs the post stages on a stream of their own (0: on the AGC's)
    unsigned stage_calls = 0;
    int stage_pipeline = 1;                   // option stage_pipeline: RRC, PL-sync walk and frame loops of a CCM call behind every timing-recovery slice (0: after the last one, frames pooled by the host first)
    int fe_slices = 0;                        // option fe_slices (0 = by mode: 4 pipelined, 8 synchronous; 1 = both stages back to back on the caller's stream)
    // balance of the two streams of the pipelined mode (s2_demod.hip): share of the timing loop's tiles that run one priority level up
    int g_prio_duty = 0, g_prio_trend = 0;
    int g_prio_hold = 0, g_prio_last_down = 0;      // the balancer's damper: calls during which no step down is tried / calls since the last step down (s2_demod.hip)
    long long g_prio_sig = -1;                // what the balance was found for (streams, MODCOD, frame kind, iteration setting of the batch): another configuration starts from 0 again
    bool g_prio_auto = true;                  // option g_prio_duty fixes the value
    int stage_loops_stream = 1;               // option stage_loops_stream: big banks run the frame loops of a slice on a stream of their own, beside the next slice's RRC (0: all post stages of a slice on one stream)
    int g_prio_cap = 7;                       // option g_prio_cap: the highest share the balancer may reach (development aid)
    int dvbs_bank_min = 2048;                 // option dvbs_bank_min: carriers from which a bank uses the four-streams-per-wave FLL (measured crossover with the written-out wave-per-stream loop: 2048 carriers 74.2 vs 74.8 ms, 1024: 46.2 vs 54.8, 4096: 128.8 vs 111.4; tests: 1)
    int dvbs_agc_stream = 1;                  // option dvbs_agc_stream: the AGC slices of a bank below dvbs_bank_min carriers on a third auxiliary stream (0: on the Viterbi stream)
    int dvbs_fe_slices = 24;                 // option dvbs_fe_slices: time slices of a DVB-S call (dvbs_demod.hip)
    // DVB-S front end (dvbs_demod.hip)
    float* d_fd_bank = nullptr;               // COMPLEX_FD interpolator bank, 256 x 256
    std::map<int, s2::cf32*> bandedge;        // FLL band-edge taps [2][ntaps] by ntaps*100000 + round(alpha*1000)*10 + sps
    s2::Workspace ws_dvbs[4];
};

namespace s2 {
// serialises an entry point on its context; stage_* additionally order the shared FEC workspaces across streams
struct CallGuard {
    std::unique_lock<std::recursive_mutex> l;
    explicit CallGuard(dvbs2gpu_ctx* ctx) : l(ctx->call_mtx) {}
};
int ws_acquire(dvbs2gpu_ctx* ctx, hipStream_t st);   // before enqueuing work that uses the context's FEC workspaces (ctx->fws) on `st`
int ws_release(dvbs2gpu_ctx* ctx, hipStream_t st);   // after it
int ws_quiesce(dvbs2gpu_ctx* ctx);                   // host-side wait for the last asynchronous user (synchronous entry points)
bool fec_jobs_pending(dvbs2gpu_ctx* ctx);
int apply_option(dvbs2gpu_ctx* ctx, const char* name, int value);   // 0, or -1: unknown name / value out of range
int get_ldpc(dvbs2gpu_ctx* ctx, int code_index, LdpcDeviceCode** out);
int get_bch(dvbs2gpu_ctx* ctx, int m, int t, BchDeviceCode** out);
int get_prbs(dvbs2gpu_ctx* ctx);
// LLR -> BBFRAME for nframes frames of one code; all pointers device
std::vector<float> make_polyphase_bank(int phases, int taps_per_phase);
std::vector<float> make_rrc_taps(int count, double beta, double Ts);
int get_rrc(dvbs2gpu_ctx* ctx, int ntaps, float alpha, double Ts, float** out);
void critically_damped(float bw, float* alpha, float* beta);
int fec_run(dvbs2gpu_ctx* ctx, const FecParams& f, const int8_t* d_llr, int nframes, int max_trials, int force, uint8_t* d_bbframes,
            int32_t* d_trials, int32_t* d_corr, hipStream_t st, FecWs* ws = nullptr);     // ws: the FEC workspaces to use (null: the context's own set)
}  // namespace s2
