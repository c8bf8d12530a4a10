// Host side of the DVB-S receive path: mirror of dsp::dvbs::DVBSDemod (module_dvbs_demod.cpp:9-117) from the input samples up to
// and including vit.process -- demod::QPSK_ALT (a17), DVBSymToSoftBlock (a18), Viterbi_DVBS (a19) -- for a BANK of independent
// streams.  The TS deframer, Forney de-interleaver call, Reed-Solomon and energy dispersal that follow in the reference are the
// next rows; the de-interleaver kernel itself is exposed separately (dvbs2gpu_forney_*).
#include "ctx.h"
#include "../../include/dvbs2gpu_math.h"
#include <cmath>
#include <algorithm>

using namespace s2;

struct dvbs2gpu_dvbs_demod {
    dvbs2gpu_ctx* ctx = nullptr;
    dvbs2gpu_dvbs_cfg cfg{};
    DvbsLoopCoefs co{};
    int nstreams = 0, max_samples = 0, max_blocks = 0;
    size_t sym_cap = 0, soft_cap = 0;
    DvbsStreamState* d_state = nullptr;
    cf32* d_buf_a = nullptr;        // [nstreams][max_samples]
    cf32* d_buf_b = nullptr;
    cf32* d_sym = nullptr;          // [nstreams][sym_cap]
    int8_t* d_soft = nullptr;       // [nstreams][soft_cap]
    cf32* d_in = nullptr;           // staging for the host entry point (stream 0)
    uint8_t* d_out = nullptr;
    uint8_t* d_ts = nullptr;        // TS staging of dvbs2gpu_dvbs_process_ts
    size_t ts_cap = 0;
    DvbsVitState* d_vstate = nullptr;
    uint8_t* d_vws = nullptr;
    cf32* d_bandedge = nullptr;
    float* d_rrc = nullptr;
    std::vector<DvbsStreamState> init_state;
};

namespace {

std::vector<cf32> make_bandedge(const dvbs2gpu_dvbs_cfg& c) {   // FLL::createBandedgeFilters, fll.cpp:61-95 (sym/samp rate are ints there)
    const int T = c.rrc_taps;
    const float PI_F = 3.14159265358979323846f;
    float sps = (float)((double)(int)c.samplerate / (double)(int)c.symbolrate);
    const int M = (int)(T / sps);
    float power = 0;
    std::vector<float> bb(T);
    auto sinc = [](double x) { return x == 0.0 ? 1.0 : sin(x) / x; };
    for (int i = 0; i < T; i++) {
        float k = -M + i * 2.0f / sps;
        float tap = (float)(sinc(c.rrc_alpha * k - 0.5f) + sinc(c.rrc_alpha * k + 0.5f));
        power += tap;
        bb[i] = tap;
    }
    std::vector<cf32> out(2 * T);
    int N = (int)((T - 1.0f) / 2.0f);
    for (int i = 0; i < T; i++) {
        float tap = bb[i] / power;
        float k = (-N + (int)i) / (2.0f * sps);
        float a1 = -2.0f * PI_F * (1.0f + c.rrc_alpha) * k, a2 = 2.0f * PI_F * (1.0f + c.rrc_alpha) * k;
        float s1, c1, s2, c2;   // the engine's own sin/cos (include/dvbs2gpu_math.h), like the oracle's tap builder
        dvbs2m::sincosf_det(a1, &s1, &c1);
        dvbs2m::sincosf_det(a2, &s2, &c2);
        out[T - i - 1] = cf32{c1 * tap, s1 * tap};
        out[T + T - i - 1] = cf32{c2 * tap, s2 * tap};
    }
    return out;
}

int vit_reset(dvbs2gpu_dvbs_demod* d) {
    std::vector<DvbsVitState> init(d->nstreams);
    memset(init.data(), 0, sizeof(DvbsVitState) * d->nstreams);
    for (auto& s : init) { s.ber = 10; s.dep_buf[0] = s.dep_buf[1] = 128; }
    HIP_TRY(hipMemcpy(d->d_vstate, init.data(), sizeof(DvbsVitState) * d->nstreams, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d->d_vws, 0, (size_t)DVBS_VIT_WS_BYTES * d->nstreams));
    return 0;
}

}  // namespace

extern "C" {

void dvbs2gpu_dvbs_demod_default_cfg(dvbs2gpu_dvbs_cfg* c) {   // main.cpp:64-73,134-139
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->symbolrate = 2e6; c->samplerate = 4e6;
    c->agc_rate = 0.0001f; c->rrc_alpha = 0.35f; c->rrc_taps = 65; c->loop_bw = 0.00628f; c->fll_bw = 0.006f;
    float bw = 0.00628f, damp = 0.707f;
    float den = (1.0f + 2.0 * damp * bw + bw * bw);
    c->clock_mu_gain = (4.0f * damp * bw) / den;
    c->clock_omega_gain = (4.0f * bw * bw) / den;
    c->omega_rel_limit = 0.02f;
    c->viterbi_ber_threshold = 0.15f; c->viterbi_max_outsync = 20;   // module_dvbs_demod.cpp:23
}

int dvbs2gpu_dvbs_demod_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_dvbs_cfg* cfg, int nstreams, int max_samples, dvbs2gpu_dvbs_demod** out) {
    if (!ctx || !cfg || !out || nstreams <= 0 || max_samples <= 0) return DVBS2GPU_ERR_ARG;
    if (cfg->rrc_taps != 65) { last_error() = "the DVB-S front end is built for the reference's 65-tap filters (RRC_TAP_COUNT)"; return DVBS2GPU_ERR_ARG; }
    if (!(cfg->samplerate > 0) || !(cfg->symbolrate > 0)) return DVBS2GPU_ERR_ARG;
    // the timing loop writes one symbol per `freq` input samples, freq >= omega * (1 - omega_rel_limit): the symbol / soft buffers are
    // sized from that bound, and the kernel stages at most FD_TILE/2 + 72 symbols per 256-sample tile (omega_min >= 1.5 keeps it at 171)
    {
        const double omega_min = (cfg->samplerate / cfg->symbolrate) * (1.0 - (double)cfg->omega_rel_limit);
        if (!(cfg->omega_rel_limit >= 0.f) || !(cfg->omega_rel_limit <= 0.25f) || !(omega_min >= 1.5) || !(cfg->samplerate / cfg->symbolrate <= 64.0)) {
            last_error() = "samplerate/symbolrate * (1 - omega_rel_limit) must be >= 1.5 (and the ratio <= 64, omega_rel_limit in [0, 0.25])";
            return DVBS2GPU_ERR_ARG;
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    auto d = new dvbs2gpu_dvbs_demod();
    d->ctx = ctx; d->cfg = *cfg; d->nstreams = nstreams; d->max_samples = max_samples;
    d->sym_cap = (size_t)((double)max_samples / ((cfg->samplerate / cfg->symbolrate) * (1.0 - (double)cfg->omega_rel_limit))) + 130;
    d->soft_cap = (size_t)2 * d->sym_cap + 2 * DVBS_SOFT_BLOCK + 128;
    d->max_blocks = (int)(d->soft_cap / DVBS_SOFT_BLOCK);
    const float PI_F = 3.14159265358979323846f;
    DvbsLoopCoefs& co = d->co;
    co.agc_rate = cfg->agc_rate;
    float a;
    critically_damped(cfg->fll_bw, &a, &co.fll_beta);
    co.fll_min_freq = -PI_F / 2.0f; co.fll_max_freq = PI_F / 2.0f;
    const float omega = (float)(cfg->samplerate / cfg->symbolrate);
    co.fd_alpha = cfg->clock_mu_gain; co.fd_beta = cfg->clock_omega_gain;
    co.fd_min_freq = (float)(omega * (1.0 - cfg->omega_rel_limit)); co.fd_max_freq = (float)(omega * (1.0 + cfg->omega_rel_limit));
    critically_damped(cfg->loop_bw, &co.cos_alpha, &co.cos_beta);
    co.cos_min_freq = -PI_F / 10.0f; co.cos_max_freq = PI_F / 10.0f;
    co.ntaps = cfg->rrc_taps;
    DvbsStreamState s0;
    memset(&s0, 0, sizeof(s0));
    s0.agc_gain = 1.0f; s0.fd_freq = omega;
    d->init_state.assign(nstreams, s0);
    int rc = 0;
    auto fail = [&](int code) { dvbs2gpu_dvbs_demod_destroy(d); return code; };
    {
        std::lock_guard<std::mutex> l(ctx->mtx);
        if (!ctx->d_fd_bank && (rc = upload(make_polyphase_bank(FD_PHASES, FD_TAPS), &ctx->d_fd_bank))) return fail(rc);
    }
    if ((rc = upload(make_bandedge(*cfg), &d->d_bandedge))) return fail(rc);
    if ((rc = get_rrc(ctx, cfg->rrc_taps, cfg->rrc_alpha, cfg->samplerate / cfg->symbolrate, &d->d_rrc))) return fail(rc);
    hipError_t e = hipMalloc((void**)&d->d_state, sizeof(DvbsStreamState) * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_buf_a, sizeof(cf32) * (size_t)max_samples * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_buf_b, sizeof(cf32) * (size_t)max_samples * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_sym, sizeof(cf32) * d->sym_cap * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_soft, d->soft_cap * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_vstate, sizeof(DvbsVitState) * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_vws, (size_t)DVBS_VIT_WS_BYTES * nstreams);
    if (e != hipSuccess) { fail_hip(e, "hipMalloc(dvbs demod)"); return fail(DVBS2GPU_ERR_HIP); }
    if ((rc = dvbs2gpu_dvbs_demod_reset(d))) return fail(rc);
    *out = d;
    return 0;
}

int dvbs2gpu_dvbs_demod_reset(dvbs2gpu_dvbs_demod* d) {
    if (!d) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(d->ctx->device));
    HIP_TRY(hipMemcpy(d->d_state, d->init_state.data(), sizeof(DvbsStreamState) * d->nstreams, hipMemcpyHostToDevice));
    return vit_reset(d);
}

void dvbs2gpu_dvbs_demod_destroy(dvbs2gpu_dvbs_demod* d) {
    if (!d) return;
    (void)hipFree(d->d_state); (void)hipFree(d->d_buf_a); (void)hipFree(d->d_buf_b); (void)hipFree(d->d_sym); (void)hipFree(d->d_soft);
    (void)hipFree(d->d_in); (void)hipFree(d->d_out); (void)hipFree(d->d_ts); (void)hipFree(d->d_vstate); (void)hipFree(d->d_vws); (void)hipFree(d->d_bandedge);
    delete d;
}


int dvbs2gpu_dvbs_demod_process_batch(dvbs2gpu_dvbs_demod* d, const float* const* d_iq, const int* counts, uint8_t* const* d_bits, int cap,
                                      int* out_counts) {
    if (!d || !d_iq || !counts || !d_bits || !out_counts || cap < 0) return DVBS2GPU_ERR_ARG;
    dvbs2gpu_ctx* ctx = d->ctx;
    CallGuard guard(ctx);                       // ws_dvbs[] are context-wide
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    const int n = d->nstreams;
    std::vector<DvbsStreamWork> work(n);
    int max_count = 0;
    for (int i = 0; i < n; ++i) {
        if (counts[i] < 0 || counts[i] > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
        if (counts[i] > 0 && !d_iq[i]) return DVBS2GPU_ERR_ARG;
        work[i].in = (const cf32*)d_iq[i]; work[i].count = counts[i];
        work[i].buf_a = d->d_buf_a + (size_t)i * d->max_samples; work[i].buf_b = d->d_buf_b + (size_t)i * d->max_samples;
        work[i].sym = d->d_sym + (size_t)i * d->sym_cap; work[i].soft = d->d_soft + (size_t)i * d->soft_cap; work[i].st = d->d_state + i;
        max_count = std::max(max_count, counts[i]);
    }
    const int mb = d->max_blocks;
    Workspace& ws = ctx->ws_dvbs[0];
    const size_t off_ptr_in = sizeof(DvbsStreamWork) * n, off_ptr_out = off_ptr_in + sizeof(void*) * n, off_nblk = off_ptr_out + sizeof(void*) * n;
    const size_t off_cnt = off_nblk + sizeof(int) * n, off_blk0 = off_cnt + sizeof(int) * n, off_nbits = off_blk0 + sizeof(int) * n, total = off_nbits + sizeof(int) * (size_t)n * mb;
    int rc;
    if ((rc = ws.ensure(total + 64))) return rc;
    Workspace& wsb = ctx->ws_dvbs[1];
    if ((rc = wsb.ensure((size_t)n * mb * DVBS_SOFT_BLOCK))) return rc;
    char* base = (char*)ws.p;
    DvbsStreamWork* d_work = (DvbsStreamWork*)base;
    std::vector<const int8_t*> pin(n);
    for (int i = 0; i < n; ++i) pin[i] = work[i].soft;
    HIP_TRY(hipMemcpyAsync(d_work, work.data(), sizeof(DvbsStreamWork) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_ptr_in, pin.data(), sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + off_ptr_out, d_bits, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    int* d_nblk = (int*)(base + off_nblk);
    int* d_cnt = (int*)(base + off_cnt);
    int* d_nbits = (int*)(base + off_nbits);
    int* d_blk0 = (int*)(base + off_blk0);
    // The serial stages time-sliced over their own streams (s2_rx_kernels.hip, dvbs_frontend_launch): AGC, FLL + RRC, timing recovery, and --
    // behind every timing-recovery slice -- the slice's Costas loop, soft FIFO append and the Viterbi decoding of the blocks it completed.
    // One carrier then costs its slowest stage instead of the sum; a bank of thousands gains too (4096 carriers: 206 -> 146 ms per call),
    // because its lane-per-stream and four-streams-per-wave stages (AGC, Costas, FLL) are latency chains that leave the SIMDs to the
    // wave-per-stream ones (timing recovery, Viterbi, RRC).
    const int nsub = ctx->dvbs_fe_slices;
    dvbs2gpu_ctx::FeAux* fa = nullptr;
    if (nsub > 1) {
        std::lock_guard<std::mutex> l(ctx->mtx);
        fa = &ctx->fe_aux[st];
        if (!fa->dvbs_aux[0]) {
            for (int a = 0; a < (ctx->dvbs_agc_stream ? 3 : 2); ++a) HIP_TRY(hipStreamCreateWithFlags(&fa->dvbs_aux[a], hipStreamNonBlocking));
            for (int a = 0; a < 4; ++a)
                for (int i = 0; i <= DVBS_FE_MAX_SLICES; ++i) HIP_TRY(hipEventCreateWithFlags(&fa->dvbs_ev[a][i], hipEventDisableTiming));
        }
    }
    struct Hook : DvbsSliceHook {
        dvbs2gpu_dvbs_demod* d; dvbs2gpu_ctx::FeAux* fa; const DvbsStreamWork* d_work; const int8_t* const* d_in_ptrs; int n, max_count, nsub, mb;
        int* d_blk0; int* d_nblk; int* d_nbits; uint8_t* d_bits; hipStream_t st;
        hipError_t after_timing(int c) override {
            hipStream_t sv = fa->dvbs_aux[0];        // a bank: behind the AGC slices, which were all enqueued before the first timing-recovery slice
            hipError_t e;
            if ((e = hipEventRecord(fa->dvbs_ev[3][c], st)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(sv, fa->dvbs_ev[3][c], 0)) != hipSuccess) return e;
            // (alternating the Costas slices between `st` and `sv`, which carry about the same load without them, was slower: every hand-over
            // between two streams costs tens of microseconds and ties the two chains together)
            if ((e = dvbs_costas_launch(d_work, n, d->co, c, nsub, sv)) != hipSuccess) return e;
            if ((e = dvbs_soft_slice_launch(d_work, n, max_count, c, nsub, d_blk0, d_nblk, sv)) != hipSuccess) return e;
            return dvbs_viterbi_launch(nullptr, d_in_ptrs, d_nblk, n, mb, d_bits, d_nbits, nullptr, d->d_vstate, d->d_vws, d->cfg.viterbi_ber_threshold,
                                       d->cfg.viterbi_max_outsync, sv, d_blk0);
        }
    } hook;
    hook.d = d; hook.fa = fa; hook.d_work = d_work; hook.d_in_ptrs = (const int8_t* const*)(base + off_ptr_in); hook.n = n; hook.max_count = max_count;
    hook.nsub = nsub; hook.mb = mb; hook.d_blk0 = d_blk0; hook.d_nblk = d_nblk; hook.d_nbits = d_nbits; hook.d_bits = (uint8_t*)wsb.p; hook.st = st;
    // (tried: the Costas slices on the AGC's stream and the decoder alone on `sv` -- AGC + Costas then carry 0.94 ms per slice beside the FLL's 0.81: nothing gained)
    // (a few carriers: the AGC slices on a stream of their own -- one carrier 39.4 -> 36.7 ms per call; a bank keeps them ahead on the Viterbi stream)
    hipStream_t aux3[3] = {fa ? fa->dvbs_aux[0] : nullptr, fa ? fa->dvbs_aux[1] : nullptr, fa && n < ctx->dvbs_bank_min ? fa->dvbs_aux[2] : nullptr};
    HIP_TRY(dvbs_frontend_launch(d_work, n, max_count, d->co, d->d_bandedge, d->d_rrc, ctx->d_fd_bank, st, fa ? aux3 : nullptr,
                                 fa ? fa->dvbs_ev : nullptr, nsub, fa ? &hook : nullptr, ctx->dvbs_bank_min));
    if (fa) {
        // the last slice's decoder run ends the Viterbi stream's work for this call
        HIP_TRY(hipEventRecord(fa->dvbs_ev[3][DVBS_FE_MAX_SLICES], fa->dvbs_aux[0]));
        HIP_TRY(hipStreamWaitEvent(st, fa->dvbs_ev[3][DVBS_FE_MAX_SLICES], 0));
        HIP_TRY(dvbs_soft_count_launch(d_work, n, d_nblk, st));          // (FIFO fill and block count of the whole call, for the packing and the compaction)
    } else {
        HIP_TRY(dvbs_soft_count_launch(d_work, n, d_nblk, st));
        HIP_TRY(dvbs_viterbi_launch(nullptr, (const int8_t* const*)(base + off_ptr_in), d_nblk, n, mb, (uint8_t*)wsb.p, d_nbits, nullptr, d->d_vstate,
                                    d->d_vws, d->cfg.viterbi_ber_threshold, d->cfg.viterbi_max_outsync, st));
    }
    HIP_TRY(dvbs_pack_bits_launch((const uint8_t*)wsb.p, d_nbits, d_nblk, n, mb, (uint8_t* const*)(base + off_ptr_out), cap, d_cnt, st));
    HIP_TRY(dvbs_soft_compact_launch(d_work, n, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int dvbs2gpu_dvbs_demod_process(dvbs2gpu_dvbs_demod* d, int count, const float* h_iq, uint8_t* h_bits, int cap) {
    if (!d || d->nstreams != 1 || count < 0 || cap < 0 || (count > 0 && !h_iq) || (cap > 0 && !h_bits)) return DVBS2GPU_ERR_ARG;
    if (count > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
    HIP_TRY(hipSetDevice(d->ctx->device));
    if (!d->d_in) HIP_TRY(hipMalloc((void**)&d->d_in, sizeof(cf32) * (size_t)d->max_samples));
    const size_t ocap = (size_t)d->max_blocks * DVBS_SOFT_BLOCK;
    if (!d->d_out) HIP_TRY(hipMalloc((void**)&d->d_out, ocap));
    if (count) HIP_TRY(hipMemcpy(d->d_in, h_iq, sizeof(cf32) * (size_t)count, hipMemcpyHostToDevice));
    const float* pi = (const float*)d->d_in;
    uint8_t* po = d->d_out;
    int nb = 0;
    const int c2 = (int)std::min<size_t>((size_t)cap, ocap);
    int rc = dvbs2gpu_dvbs_demod_process_batch(d, &pi, &count, &po, c2, &nb);
    if (rc) return rc;
    if (nb) HIP_TRY(hipMemcpy(h_bits, d->d_out, (size_t)nb, hipMemcpyDeviceToHost));
    return nb;
}

// The whole of DVBSDemod::process (module_dvbs_demod.cpp:78-99) on host buffers: samples in, TS packets out; the decoded bits stay in HBM
int dvbs2gpu_dvbs_process_ts(dvbs2gpu_dvbs_demod* d, dvbs2gpu_dvbs_tail* t, int count, const float* h_iq, uint8_t* h_ts, int cap) {
    if (!d || !t || d->nstreams != 1 || count < 0 || cap < 0 || (count > 0 && !h_iq) || (cap > 0 && !h_ts)) return DVBS2GPU_ERR_ARG;
    if (count > d->max_samples) { last_error() = "count exceeds max_samples"; return DVBS2GPU_ERR_ARG; }
    HIP_TRY(hipSetDevice(d->ctx->device));
    if (!d->d_in) HIP_TRY(hipMalloc((void**)&d->d_in, sizeof(cf32) * (size_t)d->max_samples));
    const size_t ocap = (size_t)d->max_blocks * DVBS_SOFT_BLOCK;
    if (!d->d_out) HIP_TRY(hipMalloc((void**)&d->d_out, ocap));
    if (!d->d_ts) {
        d->ts_cap = (ocap / 13056 + 2) * 8 * 188;                 // a deframer frame (8 packets) per 13056 bits, plus the carried ones
        HIP_TRY(hipMalloc((void**)&d->d_ts, d->ts_cap));
    }
    if (count) HIP_TRY(hipMemcpy(d->d_in, h_iq, sizeof(cf32) * (size_t)count, hipMemcpyHostToDevice));
    const float* pi = (const float*)d->d_in;
    uint8_t* po = d->d_out;
    int nb = 0;
    int rc = dvbs2gpu_dvbs_demod_process_batch(d, &pi, &count, &po, (int)ocap, &nb);
    if (rc) return rc;
    const uint8_t* pb = d->d_out;
    uint8_t* pt = d->d_ts;
    int nbytes = 0;
    if ((rc = dvbs2gpu_dvbs_tail_process_batch(t, &pb, &nb, &pt, (int)std::min<size_t>(d->ts_cap, (size_t)cap), &nbytes, nullptr))) return rc;
    if (nbytes) HIP_TRY(hipMemcpy(h_ts, d->d_ts, (size_t)nbytes, hipMemcpyDeviceToHost));
    return nbytes;
}

int dvbs2gpu_dvbs_demod_get_stats(dvbs2gpu_dvbs_demod* d, dvbs2gpu_viterbi_stats* h_out) {
    if (!d || !h_out) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(d->ctx->device));
    std::vector<DvbsVitState> v(d->nstreams);
    HIP_TRY(hipMemcpy(v.data(), d->d_vstate, sizeof(DvbsVitState) * d->nstreams, hipMemcpyDeviceToHost));
    for (int i = 0; i < d->nstreams; ++i) {
        h_out[i].ber = v[i].ber; h_out[i].state = v[i].state; h_out[i].rate = v[i].rate; h_out[i].phase = v[i].phase; h_out[i].shift = v[i].shift;
    }
    return d->nstreams;
}

int dvbs2gpu_dvbs_demod_get_tap(dvbs2gpu_dvbs_demod* d, int stream, int which, void* h_dst, int cap) {
    if (!d || stream < 0 || stream >= d->nstreams || which < 0 || which > 1) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(d->ctx->device));
    DvbsStreamState s;
    HIP_TRY(hipMemcpy(&s, d->d_state + stream, sizeof(s), hipMemcpyDeviceToHost));
    if (which == 1) {   // loop state: agc gain, fll phase/freq, fd phase/freq/offset, costas phase/freq
        float v[8] = {s.agc_gain, s.fll_phase, s.fll_freq, s.fd_phase, s.fd_freq, (float)s.fd_offset, s.costas_phase, s.costas_freq};
        if (h_dst && cap >= 8) memcpy(h_dst, v, sizeof(v));
        return 8;
    }
    const int n = s.n_sym;
    if (h_dst && cap > 0) HIP_TRY(hipMemcpy(h_dst, d->d_sym + (size_t)stream * d->sym_cap, sizeof(cf32) * (size_t)std::min(n, cap), hipMemcpyDeviceToHost));
    return n;
}

}  // extern "C"
