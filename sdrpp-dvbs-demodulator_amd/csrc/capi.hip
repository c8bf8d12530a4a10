// C ABI of the engine (include/dvbs2gpu.h): context, table upload, workspace management, stage launches.
// Host-side only logic here; kernels live in *_kernel.hip.  No CPU fallback: without a HIP device
// dvbs2gpu_create fails with DVBS2GPU_ERR_NODEVICE.
#include "ctx.h"
#include <climits>
#include <cstdlib>

using namespace s2;
namespace s2 { extern unsigned long long* g_ldpc_prof; }

namespace s2 {
std::atomic<long long> g_kernel_launches{0};
static thread_local std::string g_err;
std::string& last_error() { return g_err; }
int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return DVBS2GPU_ERR_HIP;
}
}  // namespace s2
#define g_err s2::last_error()

// ---------------------------------------------------------------------------------------------------
namespace s2 {
int ws_acquire(dvbs2gpu_ctx* ctx, hipStream_t st) {
    if (ctx->ws_used && ctx->ws_stream != st) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_ws, 0));
    return 0;
}
int ws_release(dvbs2gpu_ctx* ctx, hipStream_t st) {
    HIP_TRY(hipEventRecord(ctx->ev_ws, st));
    ctx->ws_stream = st; ctx->ws_used = true;
    return 0;
}
int ws_quiesce(dvbs2gpu_ctx* ctx) {
    if (ctx->ws_used) HIP_TRY(hipEventSynchronize(ctx->ev_ws));
    return 0;
}
bool fec_jobs_pending(dvbs2gpu_ctx* ctx) {
    for (void* p : ctx->pending_fec) if (p) return true;
    return false;
}
// the stage entry points share the context-wide FEC workspaces with the pipelined jobs on the FEC stream: refused while any is in flight
static int stage_enter(dvbs2gpu_ctx* ctx, hipStream_t st) {
    if (fec_jobs_pending(ctx)) { last_error() = "stage call while pipelined FEC jobs are in flight (collect them or leave the pipelined mode first)"; return DVBS2GPU_ERR_ARG; }
    return ws_acquire(ctx, st);
}

void free_ldpc_code(LdpcDeviceCode& D);
static int drop_ldpc_cache(dvbs2gpu_ctx* c) {
    if (fec_jobs_pending(c)) { last_error() = "option changes the decoder plan while pipelined FEC jobs are in flight"; return -1; }
    if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    std::lock_guard<std::mutex> l(c->mtx);
    for (auto& kv : c->ldpc) free_ldpc_code(kv.second);
    c->ldpc.clear();
    return 0;
}
// the development / test options of a context: one table for the environment variable and the entry point
int apply_option(dvbs2gpu_ctx* c, const char* name, int v) {
    const std::string n(name);
    auto in = [v](int lo, int hi) { return v >= lo && v <= hi; };
    if (n == "ldpc_wave") {
        // (which decoder serves a short code is fixed when the context first builds the code's tables: a change afterwards drops those tables -- refused while FEC jobs use them)
        if (!in(-1, 1)) return -1;
        if (v != c->ldpc_wave && !c->ldpc.empty()) { if (drop_ldpc_cache(c) != 0) return -1; }
        c->ldpc_wave = v;
    }
    else if (n == "ldpc_split") { if (!in(0, 1)) return -1; c->ldpc_split = v; }       // (read at every launch)
    else if (n == "ldpc_split_fail_attempts") { if (!in(0, 1)) return -1; c->ldpc_split_fail_attempts = v; }       // (tests: the fall-back path of the half-row decoder's speculative layers)
    else if (n == "gardner_form") { if (!(v == 0 || v == 1 || v == 2 || v == 4)) return -1; c->gardner_form = v; }
    else if (n == "gardner_cand_skew") { c->gardner_cand_skew = v; }
    else if (n == "fe_slices") { if (!in(0, s2::S2_FE_MAX_SLICES)) return -1; c->fe_slices = v; }
    else if (n == "stage_pipeline") { if (!in(0, 2)) return -1; c->stage_pipeline = v; }
    else if (n == "stage_post_stream") { if (!in(0, 2)) return -1; c->stage_post_stream = v; }
    else if (n == "stage_loops") { if (!in(0, s2::S2_FE_MAX_SLICES)) return -1; c->stage_pipeline_launches = v; }
    else if (n == "stage_min_duty") { if (!in(-1, 8)) return -1; c->stage_pipeline_min_duty = v; }
    else if (n == "loops_ahead") { if (!in(0, 1)) return -1; c->loops_ahead = v; }
    else if (n == "mixed_groups") { if (!in(0, 1)) return -1; c->mixed_groups = v; }
    else if (n == "fec_part") { if (!in(-1, 1)) return -1; c->fec_part = v; c->fec_part_on = v == 1; c->fec_part_trend = 0; }
    else if (n == "mix_fec_streams") { if (!in(1, 8)) return -1; c->mix_fec_streams = v; }
    else if (n == "g_prio_duty") { if (!in(-1, 8)) return -1; if (v < 0) c->g_prio_auto = true; else { c->g_prio_duty = v; c->g_prio_auto = false; } }
    else if (n == "stage_loops_stream") { if (!in(0, 1)) return -1; c->stage_loops_stream = v; }
    else if (n == "g_prio_cap") { if (!in(0, 7)) return -1; c->g_prio_cap = v; }
    else if (n == "dvbs_fe_slices") { if (!in(1, s2::DVBS_FE_MAX_SLICES)) return -1; c->dvbs_fe_slices = v; }
    else if (n == "dvbs_bank_min") { if (v < 1) return -1; c->dvbs_bank_min = v; }
    else if (n == "dvbs_agc_stream") { if (!in(0, 1)) return -1; c->dvbs_agc_stream = v != 0; }
    else if (n == "host_timing") { if (!in(0, 1)) return -1; c->host_timing = v; }
    else return -1;
    return 0;
}

// every device table of one code (hipFree(nullptr) is a no-op)
void free_ldpc_code(LdpcDeviceCode& D) {
    (void)hipFree(D.d_layers); (void)hipFree(D.d_ents); (void)hipFree(D.d_rows); (void)hipFree(D.d_atab);
    (void)hipFree(D.d_wave_lanec); (void)hipFree(D.d_wave_steps); (void)hipFree(D.d_wave_layer_end);
    (void)hipFree(D.d_split_layers); (void)hipFree(D.d_split_atab);
    D.d_split_layers = nullptr; D.d_split_atab = nullptr;
    D.d_layers = nullptr; D.d_ents = D.d_rows = D.d_atab = D.d_wave_lanec = D.d_wave_layer_end = nullptr; D.d_wave_steps = nullptr;
}
int get_ldpc(dvbs2gpu_ctx* ctx, int code_index, LdpcDeviceCode** out) {
    std::lock_guard<std::mutex> l(ctx->mtx);
    auto it = ctx->ldpc.find(code_index);
    if (it == ctx->ldpc.end()) {
        LdpcPlan P = build_ldpc_plan(code_index);
        LdpcDeviceCode D;
        D.code_index = code_index;
        D.N = P.N; D.K = P.K; D.R = P.R; D.q = P.q; D.max_deg = P.max_deg; D.irregular = (P.min_deg != P.max_deg); D.rec_dwords = P.rec_dwords; D.edges = P.edges; D.pent_base = P.pent_base; D.synd_base = P.synd_base;
        int rc;
        auto fail = [&D](int e) { free_ldpc_code(D); return e; };      // (a failed upload gives the earlier ones back)
        if ((rc = upload(P.layers, &D.d_layers))) return fail(rc);
        if ((rc = upload(P.ents, &D.d_ents))) return fail(rc);
        if ((rc = upload(P.rows, &D.d_rows))) return fail(rc);
        if (!P.atab.empty() && (rc = upload(P.atab, &D.d_atab))) return fail(rc);
        D.blocks_per_cu = ldpc_blocks_per_cu(P.max_deg, D.irregular, P.N);
        if (P.N > 16200 && ldpc_split_supported(P.max_deg)) {
            // the half-row decoder (ldpc_split_plan.h / ldpc_split_kernel.hip) for the normal frames it takes, and for them the default: 32 768 frames of rate 3/4 x 50 iterations
            // in 308 ms against the lane-per-row decoder's 336, the pipelined step 335 against 348 ms (r05).  The context option ldpc_split = 0 selects the
            // lane-per-row decoder (the parity tests run both).
            const LdpcSplitPlan SP = build_ldpc_split_plan(P);
            if (SP.ok && (!SP.noprev_shared || ldpc_split_noprev_shared(P.max_deg))) {
                if ((rc = upload(SP.layers, &D.d_split_layers))) return fail(rc);
                std::vector<uint32_t> tab(SP.atab);
                tab.insert(tab.end(), SP.side.begin(), SP.side.end());
                if ((rc = upload(tab, &D.d_split_atab))) return fail(rc);
                D.split_tab_words = (int)tab.size();
                D.split_npl = (int)SP.layers.size(); D.split_rec_total = SP.rec_total;
                D.split_blocks_per_cu = ldpc_split_blocks_per_cu(P.max_deg, P.N);
                D.use_split = true;
            }
        }
        if (P.N <= 16200) {
            // short frames also get the wave-per-frame plan; which decoder serves the code: ldpc_wave_default() (measured per code),
            // the context option ldpc_wave = 0 | 1 forces one (development aid / the parity tests run both)
            const LdpcWavePlan W = build_ldpc_wave_plan(P);
            D.wave_lw = W.lw; D.wave_nsteps = W.nsteps; D.wave_nl_min = W.nl_min; D.wave_absent_base = W.absent_base;
            if ((rc = upload(W.lanec, &D.d_wave_lanec))) return fail(rc);
            if ((rc = upload(W.steps, &D.d_wave_steps))) return fail(rc);
            if ((rc = upload(W.layer_end, &D.d_wave_layer_end))) return fail(rc);
            D.use_wave = ctx->ldpc_wave >= 0 ? ctx->ldpc_wave != 0 : ldpc_wave_default(code_index);
        }
        it = ctx->ldpc.emplace(code_index, D).first;
    }
    *out = &it->second;
    return 0;
}

// GF(2^m) tables exactly as the reference builds them (galois_field.hh:152-163) + Artin-Schreier map
// (reed_solomon_error_correction.hh:67-96) + this engine's byte-Horner syndrome tables.
int get_bch(dvbs2gpu_ctx* ctx, int m, int t, BchDeviceCode** out) {
    std::lock_guard<std::mutex> l(ctx->mtx);
    int key = m * 100 + t;
    auto it = ctx->bch.find(key);
    if (it == ctx->bch.end()) {
        const int Q = 1 << m, N = Q - 1;
        const uint32_t poly = (m == 16) ? 0x1002Du : 0x402Bu;
        std::vector<uint16_t> LOG(Q), EXP(Q), IMAP(Q, 0);
        EXP[N] = 0; LOG[0] = (uint16_t)N;
        uint32_t a = 1;
        for (int i = 0; i < N; ++i) {
            EXP[i] = (uint16_t)a; LOG[a] = (uint16_t)i;
            a = (a & (uint32_t)(Q >> 1)) ? ((a << 1) ^ poly) : (a << 1);
            a &= (uint32_t)(Q - 1);
        }
        auto vmul = [&](uint32_t x, uint32_t y) -> uint16_t {
            if (!x || !y) return 0;
            int e = LOG[x] + LOG[y];
            if (e >= N) e -= N;
            return EXP[e];
        };
        for (int i = 2; i < N; i += 2) {
            uint16_t x = (uint16_t)i;
            uint16_t xxx = (uint16_t)(vmul(x, x) ^ x);
            if (xxx == (uint16_t)N) continue;
            IMAP[xxx] = x;
        }
        // syndrome tables for the odd roots alpha^(2r+1), r < t
        std::vector<uint16_t> tab((size_t)t * 768, 0);
        for (int r = 0; r < t; ++r) {
            int i = 2 * r + 1;
            uint16_t* T = &tab[(size_t)r * 768];
            for (int byte = 0; byte < 256; ++byte) {
                uint16_t v = 0;
                for (int b = 0; b < 8; ++b)
                    if ((byte >> (7 - b)) & 1) v ^= EXP[(int)(((long)i * (7 - b)) % N)];
                T[byte] = v;
            }
            uint16_t c = EXP[(int)(((long)i * 8) % N)];  // alpha^(8i)
            for (int x = 0; x < 256; ++x) {
                T[256 + x] = vmul((uint32_t)x, c);
                uint32_t hi = (uint32_t)x << 8;
                T[512 + x] = (hi < (uint32_t)Q) ? vmul(hi, c) : 0;
            }
        }
        BchDeviceCode D;
        D.m = m; D.t = t; D.N = N; D.K_full = N - m * t;
        int rc;
        if ((rc = upload(LOG, &D.d_log))) return rc;
        if ((rc = upload(EXP, &D.d_exp))) return rc;
        if ((rc = upload(IMAP, &D.d_imap))) return rc;
        if ((rc = upload(tab, &D.d_syn_tab))) return rc;
        it = ctx->bch.emplace(key, D).first;
    }
    *out = &it->second;
    return 0;
}

int get_prbs(dvbs2gpu_ctx* ctx) {
    std::lock_guard<std::mutex> l(ctx->mtx);
    if (ctx->d_prbs) return 0;
    // PRBS 1 + x^14 + x^15, seed 0x4A80 as the reference loads it (bbframe_descramble.cpp:122-136)
    std::vector<uint8_t> seq(64800 / 8, 0);
    int sr = 0x4A80;
    for (int i = 0; i < 64800; i++) {
        int b = ((sr) ^ (sr >> 1)) & 1;
        seq[i / 8] |= (uint8_t)(b << (7 - (i % 8)));
        sr >>= 1;
        if (b) sr |= 0x4000;
    }
    return upload(seq, &ctx->d_prbs);
}
}  // namespace s2

namespace s2 {
static int ldpc_run(dvbs2gpu_ctx* ctx, const FecParams& f, const int8_t* d_llr, int nframes, int max_trials, int force,
                    uint8_t* d_hard, int hard_stride, int8_t* d_post, int32_t* d_trials, hipStream_t st, FecWs& W);
static int bch_run(dvbs2gpu_ctx* ctx, const FecParams& f, uint8_t* d_frames, int nframes, int32_t* d_corr, hipStream_t st, FecWs& W);

int fec_run(dvbs2gpu_ctx* ctx, const FecParams& f, const int8_t* d_llr, int nframes, int max_trials, int force, uint8_t* d_bbframes,
            int32_t* d_trials, int32_t* d_corr, hipStream_t st, FecWs* ws) {
    FecWs& W = ws ? *ws : ctx->fws;
    int rc;
    if ((rc = W.hard.ensure((size_t)nframes * (f.K / 8)))) return rc;
    uint8_t* hard = (uint8_t*)W.hard.p;
    {
        StageSpan sp(ctx->timers, ST_LDPC, st, nframes);
        if ((rc = ldpc_run(ctx, f, d_llr, nframes, max_trials, force, hard, f.K / 8, nullptr, d_trials, st, W))) return rc;
    }
    StageSpan sp(ctx->timers, ST_BCH, st, nframes);
    if ((rc = bch_run(ctx, f, hard, nframes, d_corr, st, W))) return rc;
    if ((rc = get_prbs(ctx))) return rc;
    HIP_TRY(bb_descramble_launch(hard, f.K / 8, ctx->d_prbs, f.kbch / 8, nframes, d_bbframes, st));
    return 0;
}
}  // namespace s2

namespace s2 {
static int ldpc_run(dvbs2gpu_ctx* ctx, const FecParams& f, const int8_t* d_llr, int nframes, int max_trials, int force,
                    uint8_t* d_hard, int hard_stride, int8_t* d_post, int32_t* d_trials, hipStream_t st, FecWs& W) {
    LdpcDeviceCode* C;
    int rc = get_ldpc(ctx, f.code_index, &C);
    if (rc) return rc;
    if (C->use_wave) {
        // wave-per-frame decoder: one 64-thread workgroup per frame slot, as many as the LDS of the device holds at once
        const size_t lds = ldpc_wave_lds_bytes(*C);
        int per_cu = (int)((size_t)(160 * 1024 - LDPC_WAVE_LDS_RESERVE) / lds);
        per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
        int grid = ctx->num_cus * per_cu;
        if (grid > nframes) grid = nframes;
        size_t need = (size_t)grid * ldpc_wave_msg_bytes_per_frame(*C);
        need = (need + 255) & ~(size_t)255;
        const size_t sgn_bytes = (size_t)grid * ldpc_sign_ws_bytes_per_slot();
        if ((rc = W.msg.ensure(need + 256 + sgn_bytes))) return rc;
        if (!d_trials) {
            if ((rc = W.misc.ensure((size_t)nframes * 2 * sizeof(int32_t)))) return rc;
            d_trials = (int32_t*)W.misc.p;
        }
        if (!d_hard) {
            if ((rc = W.hard.ensure((size_t)nframes * (f.K / 8)))) return rc;
            d_hard = (uint8_t*)W.hard.p; hard_stride = f.K / 8;
        }
        HIP_TRY(ldpc_wave_decode_launch(*C, d_llr, nframes, max_trials, force, d_hard, hard_stride, d_post, d_trials, (uint8_t*)W.msg.p, grid, st,
                                        (unsigned int*)((char*)W.msg.p + need), (uint32_t*)((char*)W.msg.p + need + 256)));
        return 0;
    }
    if (C->use_split && ctx->ldpc_split) {
        // half-row decoder: one frame per workgroup, frames beyond the first wave of workgroups claimed through the work counter
        int grid = ctx->num_cus * C->split_blocks_per_cu;
        if (grid > nframes) grid = nframes;
        size_t need = (size_t)grid * ldpc_split_msg_bytes_per_block(*C);
        need = (need + 255) & ~(size_t)255;
        const size_t sgn_bytes = (size_t)grid * ldpc_sign_ws_bytes_per_slot();
        if ((rc = W.msg.ensure(need + 256 + sgn_bytes))) return rc;
        if (!d_trials) {
            if ((rc = W.misc.ensure((size_t)nframes * 2 * sizeof(int32_t)))) return rc;
            d_trials = (int32_t*)W.misc.p;
        }
        if (!d_hard) {
            if ((rc = W.hard.ensure((size_t)nframes * (f.K / 8)))) return rc;
            d_hard = (uint8_t*)W.hard.p; hard_stride = f.K / 8;
        }
        if (nframes > 0)
            HIP_TRY(ldpc_split_decode_launch(*C, d_llr, nframes, max_trials, force, d_hard, hard_stride, d_post, d_trials, (uint32_t*)W.msg.p, grid, st,
                                             (unsigned int*)((char*)W.msg.p + need), (uint32_t*)((char*)W.msg.p + need + 256), ctx->ldpc_split_fail_attempts));
        return 0;
    }
    // workgroups hold 2 frame slots, or 1 for batches smaller than the device (ldpc_kernel.hip)
    const int fpb = ldpc_frames_per_block(nframes, ctx->num_cus);
    int grid = ctx->num_cus * C->blocks_per_cu;
    if (grid > (nframes + fpb - 1) / fpb) grid = (nframes + fpb - 1) / fpb;
    size_t need = (size_t)grid * fpb * C->R * C->rec_dwords * sizeof(uint32_t);
    need = (need + 255) & ~(size_t)255;
    const size_t sgn_bytes = (size_t)grid * fpb * ldpc_sign_ws_bytes_per_slot();   // bit-packed signs for the syndrome check
    if ((rc = W.msg.ensure(need + 256 + sgn_bytes))) return rc;   // + the dynamic work counter + the sign scratch
    if (!d_trials) {
        if ((rc = W.misc.ensure((size_t)nframes * 2 * sizeof(int32_t)))) return rc;
        d_trials = (int32_t*)W.misc.p;
    }
    if (!d_hard) {
        if ((rc = W.hard.ensure((size_t)nframes * (f.K / 8)))) return rc;
        d_hard = (uint8_t*)W.hard.p; hard_stride = f.K / 8;
    }
    HIP_TRY(ldpc_decode_launch(*C, d_llr, nframes, max_trials, force, d_hard, hard_stride, d_post, d_trials,
                               (uint32_t*)W.msg.p, grid, fpb, st, (unsigned int*)((char*)W.msg.p + need),
                               (uint32_t*)((char*)W.msg.p + need + 256)));
    return 0;
}
}  // namespace s2

namespace s2 {
static int bch_run(dvbs2gpu_ctx* ctx, const FecParams& f, uint8_t* d_frames, int nframes, int32_t* d_corr, hipStream_t st, FecWs& W) {
    BchDeviceCode* B;
    // GF(2^16): t = 8 and 10 use the first 2t syndromes of the same field; tables depend on (m, t) only via t rows
    int rc = get_bch(ctx, f.bch_m, f.bch_t, &B);
    if (rc) return rc;
    if ((rc = W.syn.ensure((size_t)nframes * 32 * sizeof(uint16_t) + (size_t)(nframes + 2) * sizeof(int32_t)))) return rc;
    uint16_t* syn = (uint16_t*)W.syn.p;
    int32_t* todo = (int32_t*)(syn + (size_t)nframes * 32);      // counters + list of the frames that need the correction kernel
    HIP_TRY(bch_syndromes_launch(*B, d_frames, f.K / 8, f.K, nframes, syn, todo, d_corr, st));
    HIP_TRY(bch_correct_launch(*B, d_frames, f.K / 8, f.K, f.kbch, nframes, syn, todo, d_corr, st));
    return 0;
}
}  // namespace s2

// ---------------------------------------------------------------------------------------------------
// The engine drives several HIP streams side by side (the caller's, the front end's, the FEC jobs'); HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
// streams that share a queue run one behind the other.  The runtime reads the variable when it initialises, i.e. at the first HIP call of the process.  The library does NOT touch the
// process environment on its own (round 5 did, from a load-time constructor: not thread-safe against a running host, and order-dependent): a host that wants more queues calls
// dvbs2gpu_preinit() before its first HIP call and before it starts threads, or exports the variable itself.  Results are the same on any number of queues.
extern "C" {

const char* dvbs2gpu_version(void) { return "dvbs2gpu 0.1 (gfx950)"; }
int dvbs2gpu_preinit(void) {
    // 0: the variable was already there (left alone); 1: set.  Only meaningful before the process's first HIP call.
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;
    return setenv("GPU_MAX_HW_QUEUES", "12", 0) == 0 ? 1 : DVBS2GPU_ERR_ARG;
}
const char* dvbs2gpu_last_error(void) { return g_err.c_str(); }

int dvbs2gpu_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

int dvbs2gpu_create(int device, dvbs2gpu_ctx** out) {
    if (!out) return DVBS2GPU_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_err = "no HIP device visible (this engine has no CPU fallback)";
        return DVBS2GPU_ERR_NODEVICE;
    }
    if (device < 0 || device >= n) { g_err = "bad device index"; return DVBS2GPU_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    dvbs2gpu_ctx* c = new dvbs2gpu_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // development / test options: DVBS2GPU_OPTIONS="name=value,name=value" in the environment -- the ONE variable the library reads -- or dvbs2gpu_set_option()
    if (const char* v = getenv("DVBS2GPU_OPTIONS")) {
        std::string all(v);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            const size_t eq = item.find('=');
            char* endp = nullptr;
            const long val = eq == std::string::npos ? 0 : strtol(item.c_str() + eq + 1, &endp, 10);
            const bool numeric = eq != std::string::npos && endp != item.c_str() + eq + 1 && *endp == '\0' && val >= INT_MIN && val <= INT_MAX;
            if (!numeric || s2::apply_option(c, item.substr(0, eq).c_str(), (int)val) != 0) {
                g_err = "DVBS2GPU_OPTIONS: unknown option or value out of range: " + item;
                delete c;
                return DVBS2GPU_ERR_ARG;
            }
            pos = end + 1;
        }
    }
    hipError_t ee = hipEventCreateWithFlags(&c->ev_ws, hipEventDisableTiming);
    if (ee != hipSuccess) { delete c; return fail_hip(ee, "hipEventCreate"); }
    *out = c;
    return DVBS2GPU_OK;
}

int dvbs2gpu_set_option(dvbs2gpu_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    if (s2::apply_option(ctx, name, value) != 0) { g_err = std::string("unknown option or value out of range: ") + name; return DVBS2GPU_ERR_ARG; }
    return DVBS2GPU_OK;
}

int dvbs2gpu_get_state(dvbs2gpu_ctx* ctx, const char* name, long long* value) {
    if (!ctx || !name || !value) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    const std::string n(name);
    if (n == "kernel_launches") *value = s2::g_kernel_launches.load(std::memory_order_relaxed);
    else if (n == "g_prio_duty") *value = ctx->g_prio_duty;
    else if (n == "g_prio_auto") *value = ctx->g_prio_auto ? 1 : 0;
    else if (n == "g_prio_hold") *value = ctx->g_prio_hold;
    else if (n == "stage_pipeline_on") *value = ctx->last_call_staged ? 1 : 0;
    else if (n == "fec_part_on") *value = ctx->fec_part_on ? 1 : 0;
    else if (n == "fec_part") *value = ctx->fec_part;
    else if (n == "pipelined") *value = ctx->pipeline_fec;
    else if (n == "num_cus") *value = ctx->num_cus;
    else { g_err = std::string("unknown state name: ") + name; return DVBS2GPU_ERR_ARG; }
    return DVBS2GPU_OK;
}

int dvbs2gpu_debug_last_fec_job(dvbs2gpu_ctx* ctx, int slot, long long* out10, int32_t* h_first, const void** h_handles, int cap) {
    if (!ctx || !out10 || slot < 0 || slot >= dvbs2gpu_ctx::MAX_PIPE_GROUPS) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    const dvbs2gpu_ctx::LastFecJob& lj = ctx->last_fec[slot];
    out10[0] = (long long)(uintptr_t)lj.d_llr; out10[1] = (long long)(uintptr_t)lj.d_bb; out10[2] = lj.nf; out10[3] = lj.n; out10[4] = lj.N; out10[5] = lj.kb;
    out10[6] = lj.rate; out10[7] = lj.shortframe; out10[8] = lj.max_trials; out10[9] = lj.force;
    for (int i = 0; h_first && i <= lj.n && i < cap && i < (int)lj.first.size(); ++i) h_first[i] = lj.first[i];
    for (int i = 0; h_handles && i < lj.n && i < cap; ++i) h_handles[i] = lj.dm[i];
    return DVBS2GPU_OK;
}

void dvbs2gpu_destroy(dvbs2gpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : ctx->ldpc) free_ldpc_code(kv.second);
    for (auto& kv : ctx->bch) {
        (void)hipFree(kv.second.d_log); (void)hipFree(kv.second.d_exp); (void)hipFree(kv.second.d_imap); (void)hipFree(kv.second.d_syn_tab);
    }
    if (ctx->d_prbs) (void)hipFree(ctx->d_prbs);
    // receive-chain tables
    if (ctx->d_gardner_bank) (void)hipFree(ctx->d_gardner_bank);
    if (ctx->pl.sof) (void)hipFree((void*)ctx->pl.sof);
    if (ctx->pl.plsc) (void)hipFree((void*)ctx->pl.plsc);
    if (ctx->pl.plsc_code) (void)hipFree((void*)ctx->pl.plsc_code);
    if (ctx->pl.rn) (void)hipFree((void*)ctx->pl.rn);
    for (auto& kv : ctx->constel) {
        if (kv.second.d_bits) (void)hipFree(kv.second.d_bits);
        if (kv.second.d_bits4) (void)hipFree(kv.second.d_bits4);
        if (kv.second.d_err) (void)hipFree(kv.second.d_err);
        if (kv.second.d_pts) (void)hipFree(kv.second.d_pts);
    }
    for (auto& kv : ctx->rrc) if (kv.second) (void)hipFree(kv.second);
    if (ctx->d_fd_bank) (void)hipFree(ctx->d_fd_bank);
    for (auto& kv : ctx->bandedge) if (kv.second) (void)hipFree(kv.second);
    if (ctx->d_vcm_mods) (void)hipFree(ctx->d_vcm_mods);
    if (ctx->d_vcm_cons) (void)hipFree(ctx->d_vcm_cons);
    for (auto& w : ctx->ws_vcm) w.release();
    for (auto& w : ctx->ws_mix) w.release();
    for (auto& row : ctx->ev_mix) for (hipEvent_t& e : row) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (ctx->ev_ws) (void)hipEventDestroy(ctx->ev_ws);
    for (auto& sp : ctx->timers.pending) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : ctx->timers.pool) (void)hipEventDestroy(e);
    ctx->fws.release();
    for (auto& kv : ctx->fe_aux) {
        if (kv.second.aux) (void)hipStreamDestroy(kv.second.aux);
        for (hipEvent_t e : kv.second.ev) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : kv.second.ev2) if (e) (void)hipEventDestroy(e);
        if (kv.second.aux2) (void)hipStreamDestroy(kv.second.aux2);
        if (kv.second.aux3) (void)hipStreamDestroy(kv.second.aux3);
        for (hipEvent_t e : kv.second.ev3) if (e) (void)hipEventDestroy(e);
        for (hipStream_t a : kv.second.dvbs_aux) if (a) (void)hipStreamDestroy(a);
        for (auto& row : kv.second.dvbs_ev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
    }
    if (ctx->fe_stream) (void)hipStreamDestroy(ctx->fe_stream);
    if (ctx->fec_stream) (void)hipStreamDestroy(ctx->fec_stream);
    if (ctx->fec_part_stream) (void)hipStreamDestroy(ctx->fec_part_stream);
    if (ctx->ev_llr) (void)hipEventDestroy(ctx->ev_llr);
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    for (int g = 0; g < dvbs2gpu_ctx::MAX_PIPE_GROUPS; ++g) {
        for (int k = 0; k < 2; ++k) if (ctx->ev_fec[g][k]) (void)hipEventDestroy(ctx->ev_fec[g][k]);
        for (int k = 0; k < 2; ++k) if (ctx->ev_fec_t0[g][k]) (void)hipEventDestroy(ctx->ev_fec_t0[g][k]);
        if (ctx->ev_llr_grp[g]) (void)hipEventDestroy(ctx->ev_llr_grp[g]);
        if (ctx->grp_stream[g]) (void)hipStreamDestroy(ctx->grp_stream[g]);
        ctx->fws_grp[g].release();
        for (auto& w : ctx->ws_grp[g]) w.release();
        for (auto& par : ctx->ws_fecbuf[g]) for (auto& w : par) w.release();
    }
    for (auto& w : ctx->ws_rx) w.release();
    for (auto& w : ctx->ws_dvbs) w.release();
    delete ctx;
}

int dvbs2gpu_set_stage_timing(dvbs2gpu_ctx* ctx, int on) {
    if (!ctx) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    std::lock_guard<std::mutex> l(ctx->timers.mtx);
    ctx->timers.on = on != 0;
    return 0;
}
int dvbs2gpu_get_stage_times(dvbs2gpu_ctx* ctx, dvbs2gpu_stage_times* out) {
    if (!ctx || !out) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    StageTimers& T = ctx->timers;
    std::lock_guard<std::mutex> l(T.mtx);
    for (auto& sp : T.pending) {
        HIP_TRY(hipEventSynchronize(sp.b));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { T.ms[sp.stage] += ms; T.launches[sp.stage]++; T.units[sp.stage] += sp.units; }
        T.pool.push_back(sp.a); T.pool.push_back(sp.b);
    }
    T.pending.clear();
    static_assert(DVBS2GPU_STAGE_COUNT == ST_COUNT, "stage list");
    for (int i = 0; i < ST_COUNT; ++i) {
        out->ms[i] = T.ms[i]; out->launches[i] = T.launches[i]; out->units[i] = T.units[i];
        T.ms[i] = 0; T.launches[i] = 0; T.units[i] = 0;
    }
    return 0;
}

// development aid: phase-cycle buffer (96 x u64, device memory) used by -DLDPC_PROF builds; not part of the public ABI
void dvbs2gpu_debug_set_prof(void* d_buf) { s2::g_ldpc_prof = (unsigned long long*)d_buf; }

static void fill_info(const ModcodParams& p, dvbs2gpu_modcod_info* o) {
    o->constellation = p.constel; o->bits_per_symbol = p.bits; o->rate = p.rate; o->slots = p.slots;
    o->pilot_blocks = p.pilot_blocks; o->plframe_symbols = p.plframe; o->ldpc_n = p.fec.N; o->ldpc_k = p.fec.K;
    o->kbch = p.fec.kbch; o->bch_t = p.fec.bch_t; o->ldpc_edges = QC_CODES[p.fec.code_index].edges;
    o->g1 = p.g1; o->g2 = p.g2;
}

int dvbs2gpu_modcod_info_get(int modcod, int shortframes, int pilots, dvbs2gpu_modcod_info* out) {
    if (!out) return DVBS2GPU_ERR_ARG;
    ModcodParams p;
    if (!modcod_params(modcod, shortframes, pilots, &p)) { g_err = "unsupported MODCOD"; return DVBS2GPU_ERR_MODCOD; }
    fill_info(p, out);
    return DVBS2GPU_OK;
}

int dvbs2gpu_fec_info_get(int rate, int shortframes, dvbs2gpu_modcod_info* out) {
    if (!out) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    memset(out, 0, sizeof(*out));
    out->constellation = -1; out->rate = rate; out->ldpc_n = f.N; out->ldpc_k = f.K; out->kbch = f.kbch; out->bch_t = f.bch_t;
    out->ldpc_edges = QC_CODES[f.code_index].edges;
    return DVBS2GPU_OK;
}


int dvbs2gpu_ldpc_plan_dump(int rate, int shortframes, uint32_t* layers4, uint32_t* ents, uint32_t* rows, int32_t* counts3) {
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (!counts3) return DVBS2GPU_ERR_ARG;
    LdpcPlan P = build_ldpc_plan(f.code_index);
    counts3[0] = (int32_t)P.layers.size(); counts3[1] = (int32_t)P.ents.size(); counts3[2] = (int32_t)P.rows.size();
    if (layers4) memcpy(layers4, P.layers.data(), P.layers.size() * sizeof(LdpcLayerDesc));
    if (ents) memcpy(ents, P.ents.data(), P.ents.size() * sizeof(uint32_t));
    if (rows) memcpy(rows, P.rows.data(), P.rows.size() * sizeof(uint32_t));
    return 0;
}

int dvbs2gpu_ldpc_wave_plan_dump(int rate, int shortframes, uint32_t* lanec, uint16_t* steps, uint32_t* layer_end, int32_t* counts6) {
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (!counts6) return DVBS2GPU_ERR_ARG;
    const LdpcPlan P = build_ldpc_plan(f.code_index);
    const LdpcWavePlan W = build_ldpc_wave_plan(P);
    counts6[0] = W.lw; counts6[1] = W.nl_min; counts6[2] = W.nsteps; counts6[3] = W.absent_base; counts6[4] = (int32_t)W.lanec.size(); counts6[5] = (int32_t)W.steps.size();
    if (lanec) memcpy(lanec, W.lanec.data(), W.lanec.size() * sizeof(uint32_t));
    if (steps) memcpy(steps, W.steps.data(), W.steps.size() * sizeof(uint16_t));
    if (layer_end) memcpy(layer_end, W.layer_end.data(), W.layer_end.size() * sizeof(uint32_t));
    return 0;
}

int dvbs2gpu_ldpc_addr_table_dump(int rate, int shortframes, uint32_t* table, int32_t* counts2) {
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (!counts2) return DVBS2GPU_ERR_ARG;
    const LdpcPlan P = build_ldpc_plan(f.code_index);
    const int npi = (P.max_deg + 1) / 2;
    counts2[0] = (int32_t)P.atab.size(); counts2[1] = npi <= 1 ? 1 : npi <= 2 ? 2 : npi <= 4 ? 4 : 8;
    if (table && !P.atab.empty()) memcpy(table, P.atab.data(), P.atab.size() * sizeof(uint32_t));
    return 0;
}

int dvbs2gpu_ldpc_split_plan_dump(int rate, int shortframes, uint32_t* layers4, uint32_t* table, int32_t* row_of, int32_t* layer_of, int32_t* counts6) {
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (!counts6) return DVBS2GPU_ERR_ARG;
    const LdpcPlan P = build_ldpc_plan(f.code_index);
    const LdpcSplitPlan S = build_ldpc_split_plan(P);
    counts6[0] = S.ok ? (int32_t)S.layers.size() : 0; counts6[1] = S.npw; counts6[2] = S.hs; counts6[3] = S.rec_total; counts6[4] = (int32_t)(S.atab.size() + S.side.size()); counts6[5] = S.rec_dwords;      // (table words: the pseudo-layers' tables, then the side entries of the kind-8 layers -- what the device holds)
    if (!S.ok) { g_err = std::string("the half-row decoder does not take this code: ") + S.why; return 0; }      // (not an error: counts6[0] = 0; the reason through dvbs2gpu_last_error)
    if (layers4) memcpy(layers4, S.layers.data(), S.layers.size() * sizeof(LdpcSplitLayer));
    if (table) { memcpy(table, S.atab.data(), S.atab.size() * sizeof(uint32_t)); if (!S.side.empty()) memcpy(table + S.atab.size(), S.side.data(), S.side.size() * sizeof(uint32_t)); }
    if (row_of) for (size_t i = 0; i < S.row_of.size(); ++i) row_of[i] = S.row_of[i];
    if (layer_of) for (size_t i = 0; i < S.layer_of.size(); ++i) layer_of[i] = S.layer_of[i];
    return 0;
}

int dvbs2gpu_ldpc_plan_info(dvbs2gpu_ctx* ctx, int rate, int shortframes, int32_t* out8) {
    if (!ctx || !out8) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    HIP_TRY(hipSetDevice(ctx->device));
    LdpcDeviceCode* C;
    int rc = get_ldpc(ctx, f.code_index, &C);
    if (rc) return rc;
    LdpcPlan P = build_ldpc_plan(f.code_index);
    out8[0] = C->q; out8[1] = C->max_deg; out8[2] = C->rec_dwords; out8[3] = P.sum_depth;
    out8[4] = C->blocks_per_cu; out8[5] = ctx->num_cus; out8[6] = C->edges; out8[7] = P.conflict_layers;
    if (C->use_split && ctx->ldpc_split) { out8[2] = ldpc_split_plan_rec_dwords(P.max_deg); out8[4] = C->split_blocks_per_cu; }
    return 0;
}

int dvbs2gpu_ldpc_decoder_form(dvbs2gpu_ctx* ctx, int rate, int shortframes) {
    if (!ctx) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    HIP_TRY(hipSetDevice(ctx->device));
    LdpcDeviceCode* C;
    int rc = get_ldpc(ctx, f.code_index, &C);
    if (rc) return rc;
    return C->use_wave ? 1 : (C->use_split && ctx->ldpc_split) ? 2 : 0;
}

int dvbs2gpu_ldpc_decode_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, const int8_t* d_llr, int nframes, int max_trials,
                               int force, uint8_t* d_hard, int8_t* d_post, int32_t* d_trials, void* stream) {
    if (!ctx || nframes < 0 || max_trials < 0) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (nframes == 0) return 0;
    if (!d_llr) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_enter(ctx, (hipStream_t)stream);
    if (rc) return rc;
    {
        StageSpan sp(ctx->timers, ST_LDPC, (hipStream_t)stream, nframes);
        rc = ldpc_run(ctx, f, d_llr, nframes, max_trials, force, d_hard, f.K / 8, d_post, d_trials, (hipStream_t)stream, ctx->fws);
    }
    int rc2 = ws_release(ctx, (hipStream_t)stream);
    return rc ? rc : rc2;
}


int dvbs2gpu_bch_decode_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, uint8_t* d_frames, int nframes, int32_t* d_corrections,
                              void* stream) {
    if (!ctx || nframes < 0) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (nframes == 0) return 0;
    if (!d_frames) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_enter(ctx, (hipStream_t)stream);
    if (rc) return rc;
    rc = bch_run(ctx, f, d_frames, nframes, d_corrections, (hipStream_t)stream, ctx->fws);
    int rc2 = ws_release(ctx, (hipStream_t)stream);
    return rc ? rc : rc2;
}

int dvbs2gpu_bb_descramble_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, const uint8_t* d_frames, int nframes, uint8_t* d_out,
                                 void* stream) {
    if (!ctx || nframes < 0) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (nframes == 0) return 0;
    if (!d_frames || !d_out) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = get_prbs(ctx);
    if (rc) return rc;
    HIP_TRY(bb_descramble_launch(d_frames, f.K / 8, ctx->d_prbs, f.kbch / 8, nframes, d_out, (hipStream_t)stream));
    return 0;
}

int dvbs2gpu_fec_decode_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, const int8_t* d_llr, int nframes, int max_trials,
                              int force, uint8_t* d_bbframes, int32_t* d_trials, int32_t* d_corrections, void* stream) {
    if (!ctx || nframes < 0 || max_trials < 0) return DVBS2GPU_ERR_ARG;
    FecParams f;
    if (!fec_params(rate, shortframes, &f)) { g_err = "unsupported code rate"; return DVBS2GPU_ERR_MODCOD; }
    if (nframes == 0) return 0;
    if (!d_llr || !d_bbframes) return DVBS2GPU_ERR_ARG;
    CallGuard guard(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = stage_enter(ctx, (hipStream_t)stream);
    if (rc) return rc;
    rc = fec_run(ctx, f, d_llr, nframes, max_trials, force, d_bbframes, d_trials, d_corrections, (hipStream_t)stream, nullptr);
    int rc2 = ws_release(ctx, (hipStream_t)stream);
    return rc ? rc : rc2;
}

}  // extern "C"
