// C ABI of the DVB-S inner-code path (include/dvbs2gpu.h, section "DVB-S").
#include "ctx.h"

using namespace s2;
#define g_err last_error()

struct dvbs2gpu_viterbi {
    dvbs2gpu_ctx* ctx = nullptr;
    int nstreams = 0;
    float thr = 0.15f;
    int max_outsync = 20;
    DvbsVitState* d_states = nullptr;
    uint8_t* d_ws = nullptr;
};
struct dvbs2gpu_forney {
    dvbs2gpu_ctx* ctx = nullptr;
    int nstreams = 0;
    uint8_t* d_hist = nullptr;
};
struct dvbs2gpu_ccdec {
    dvbs2gpu_ctx* ctx = nullptr;
    int nstreams = 0, frame_size = 0;
    int* d_state = nullptr;
    unsigned long long* d_dec = nullptr;
};

extern "C" {

int dvbs2gpu_dvbs_slice(dvbs2gpu_ctx* ctx, const float* d_iq, int nsymbols, int8_t* d_soft, void* stream) {
    if (!ctx || nsymbols < 0) return DVBS2GPU_ERR_ARG;
    if (nsymbols == 0) return 0;
    if (!d_iq || !d_soft) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(dvbs_slice_launch(d_iq, nsymbols, d_soft, (hipStream_t)stream));
    return 0;
}

int dvbs2gpu_ccdec_create(dvbs2gpu_ctx* ctx, int nstreams, int frame_size, dvbs2gpu_ccdec** out) {
    if (!ctx || !out || nstreams <= 0 || frame_size < 6 || frame_size > 65536) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    dvbs2gpu_ccdec* h = new dvbs2gpu_ccdec();
    h->ctx = ctx; h->nstreams = nstreams; h->frame_size = frame_size;
    hipError_t e = hipMalloc((void**)&h->d_state, sizeof(int) * 2 * nstreams);
    if (e == hipSuccess) e = hipMemset(h->d_state, 0, sizeof(int) * 2 * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_dec, sizeof(unsigned long long) * (size_t)(frame_size + 6) * nstreams);
    if (e != hipSuccess) { if (h->d_state) (void)hipFree(h->d_state); delete h; return fail_hip(e, "hipMalloc(ccdec)"); }
    *out = h;
    return 0;
}
void dvbs2gpu_ccdec_destroy(dvbs2gpu_ccdec* h) {
    if (!h) return;
    (void)hipFree(h->d_state); (void)hipFree(h->d_dec);
    delete h;
}
int dvbs2gpu_ccdec_work_batch(dvbs2gpu_ccdec* h, const uint8_t* d_soft, int64_t stream_stride, int block_stride, int nblocks, uint8_t* d_bits,
                              void* stream) {
    if (!h || nblocks < 0 || block_stride < 0) return DVBS2GPU_ERR_ARG;
    if (nblocks == 0) return 0;
    if (!d_soft || !d_bits) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(h->ctx->device));
    HIP_TRY(dvbs_cc_decode_launch(d_soft, (long)stream_stride, block_stride, h->nstreams, nblocks, h->frame_size, d_bits,
                                  (long)nblocks * h->frame_size, h->d_dec, h->d_state, (hipStream_t)stream));
    return 0;
}

int dvbs2gpu_viterbi_create(dvbs2gpu_ctx* ctx, int nstreams, float ber_threshold, int max_outsync, dvbs2gpu_viterbi** out) {
    if (!ctx || !out || nstreams <= 0) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    dvbs2gpu_viterbi* h = new dvbs2gpu_viterbi();
    h->ctx = ctx; h->nstreams = nstreams; h->thr = ber_threshold; h->max_outsync = max_outsync;
    hipError_t e = hipMalloc((void**)&h->d_states, sizeof(DvbsVitState) * nstreams);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_ws, (size_t)DVBS_VIT_WS_BYTES * nstreams);
    if (e != hipSuccess) { if (h->d_states) (void)hipFree(h->d_states); delete h; return fail_hip(e, "hipMalloc(viterbi)"); }
    *out = h;
    int rc = dvbs2gpu_viterbi_reset(h);
    if (rc) { dvbs2gpu_viterbi_destroy(h); *out = nullptr; }
    return rc;
}
int dvbs2gpu_viterbi_reset(dvbs2gpu_viterbi* h) {
    if (!h) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(h->ctx->device));
    std::vector<DvbsVitState> init(h->nstreams);
    memset(init.data(), 0, sizeof(DvbsVitState) * h->nstreams);
    for (auto& s : init) { s.ber = 10; s.dep_buf[0] = s.dep_buf[1] = 128; }   // viterbi_all.cpp:10-50, depunc.h member initialisers
    HIP_TRY(hipMemcpy(h->d_states, init.data(), sizeof(DvbsVitState) * h->nstreams, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->d_ws, 0, (size_t)DVBS_VIT_WS_BYTES * h->nstreams));
    return 0;
}
void dvbs2gpu_viterbi_destroy(dvbs2gpu_viterbi* h) {
    if (!h) return;
    (void)hipFree(h->d_states); (void)hipFree(h->d_ws);
    delete h;
}
int dvbs2gpu_viterbi_work_batch(dvbs2gpu_viterbi* h, const int8_t* d_soft, int nblocks, uint8_t* d_bits, int32_t* d_nbits,
                                dvbs2gpu_viterbi_stats* d_stats, void* stream) {
    if (!h || nblocks < 0) return DVBS2GPU_ERR_ARG;
    if (nblocks == 0) return 0;
    if (!d_soft || !d_bits || !d_nbits) return DVBS2GPU_ERR_ARG;
    static_assert(sizeof(dvbs2gpu_viterbi_stats) == sizeof(DvbsVitStats), "stats POD mismatch");
    HIP_TRY(hipSetDevice(h->ctx->device));
    HIP_TRY(dvbs_viterbi_launch(d_soft, nullptr, nullptr, h->nstreams, nblocks, d_bits, d_nbits, (DvbsVitStats*)d_stats, h->d_states, h->d_ws,
                                h->thr, h->max_outsync, (hipStream_t)stream));
    return 0;
}

int dvbs2gpu_forney_create(dvbs2gpu_ctx* ctx, int nstreams, dvbs2gpu_forney** out) {
    if (!ctx || !out || nstreams <= 0) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    dvbs2gpu_forney* h = new dvbs2gpu_forney();
    h->ctx = ctx; h->nstreams = nstreams;
    hipError_t e = hipMalloc((void**)&h->d_hist, (size_t)DVBS_FORNEY_HIST * nstreams);
    if (e == hipSuccess) e = hipMemset(h->d_hist, 0, (size_t)DVBS_FORNEY_HIST * nstreams);   // FIFOs start zero-filled (dvbs_interleaving.h:27-43)
    if (e != hipSuccess) { delete h; return fail_hip(e, "hipMalloc(forney)"); }
    *out = h;
    return 0;
}
void dvbs2gpu_forney_destroy(dvbs2gpu_forney* h) {
    if (!h) return;
    (void)hipFree(h->d_hist);
    delete h;
}
int dvbs2gpu_forney_deinterleave_batch(dvbs2gpu_forney* h, const uint8_t* d_in, int nbytes, uint8_t* d_out, void* stream) {
    if (!h || nbytes < 0 || nbytes % 12 != 0) return DVBS2GPU_ERR_ARG;
    if (nbytes == 0) return 0;
    if (!d_in || !d_out || d_in == d_out) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(h->ctx->device));
    HIP_TRY(dvbs_deinterleave_launch(d_in, nbytes, h->nstreams, nbytes, d_out, h->d_hist, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------- DVB-S tail
// TS deframer -> Forney de-interleaver -> RS(204,188) -> energy dispersal removal -> 188-byte TS packets: the rest of
// DVBSDemod::process after vit.process (module_dvbs_demod.cpp:82-99), for a bank of streams.
struct dvbs2gpu_dvbs_tail {
    dvbs2gpu_ctx* ctx = nullptr;
    int nstreams = 0, max_bits = 0, max_frames = 0;
    long v_stride = 0, frames_stride = 0;
    uint8_t* d_hist[2] = {nullptr, nullptr};   // last 13055 bits, double-buffered
    int cur = 0;
    uint8_t* d_v = nullptr;
    int* d_hit = nullptr;
    int* d_nframes = nullptr;
    int* d_errs = nullptr;
    uint8_t* d_frames = nullptr;
    uint8_t* d_deint = nullptr;
    uint8_t* d_forney = nullptr;
    uint8_t* d_status = nullptr;
    int* d_rs_err = nullptr;
    uint8_t* d_gf = nullptr;
    uint8_t* d_prbs = nullptr;
    DvbsTailState* d_state = nullptr;
    void* d_args = nullptr;                    // [in ptrs][out ptrs][counts][out bytes]
};

extern "C" {

int dvbs2gpu_dvbs_tail_create(dvbs2gpu_ctx* ctx, int nstreams, int max_bits, dvbs2gpu_dvbs_tail** out) {
    if (!ctx || !out || nstreams <= 0 || max_bits <= 0) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    auto t = new dvbs2gpu_dvbs_tail();
    t->ctx = ctx; t->nstreams = nstreams; t->max_bits = max_bits;
    t->max_frames = max_bits / (1632 * 8) + 3;
    t->v_stride = ((long)max_bits + 1632 * 8 + 63) & ~63L;
    t->frames_stride = (long)t->max_frames * 1632;
    // tables: GF(256) exp[512] ++ log[256] (field.h, polynomial 0x11d); PRBS bytes after a reset (dvbs_scrambling.h)
    std::vector<uint8_t> gf(768, 0), prbs(32767);
    {
        unsigned e = 1;
        gf[0] = 1;
        for (unsigned i = 1; i < 512; i++) {
            e *= 2;
            e = e > 255 ? (e ^ 0x11d) : e;
            gf[i] = (uint8_t)e;
            if (i < 256) gf[512 + e] = (uint8_t)i;
        }
        int reg = 0xa9;
        for (int i = 0; i < 32767; ++i) {
            int v = 0;
            for (int k = 0; k < 8; ++k) {
                int fb = ((reg >> 13) ^ (reg >> 14)) & 1;
                reg = ((reg << 1) | fb) & 0x7fff;
                v = (v << 1) | fb;
            }
            prbs[i] = (uint8_t)v;
        }
    }
    const size_t n = (size_t)nstreams;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemset(*p, 0, bytes); } };
    A((void**)&t->d_hist[0], n * 1632 * 8); A((void**)&t->d_hist[1], n * 1632 * 8);
    A((void**)&t->d_v, n * (size_t)t->v_stride);
    A((void**)&t->d_hit, n * t->max_frames * sizeof(int)); A((void**)&t->d_nframes, n * sizeof(int)); A((void**)&t->d_errs, n * 2 * sizeof(int));
    A((void**)&t->d_frames, n * (size_t)t->frames_stride); A((void**)&t->d_deint, n * (size_t)t->frames_stride);
    A((void**)&t->d_forney, n * DVBS_FORNEY_HIST);
    A((void**)&t->d_status, n * t->max_frames * 8); A((void**)&t->d_rs_err, n * t->max_frames * 8 * sizeof(int));
    A((void**)&t->d_gf, 768); A((void**)&t->d_prbs, 32767);
    A((void**)&t->d_state, n * sizeof(DvbsTailState));
    A(&t->d_args, n * (2 * sizeof(void*) + 2 * sizeof(int)));
    if (e != hipSuccess) { dvbs2gpu_dvbs_tail_destroy(t); return fail_hip(e, "hipMalloc(dvbs tail)"); }
    HIP_TRY(hipMemcpy(t->d_gf, gf.data(), 768, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->d_prbs, prbs.data(), 32767, hipMemcpyHostToDevice));
    *out = t;
    int rc = dvbs2gpu_dvbs_tail_reset(t);
    if (rc) { dvbs2gpu_dvbs_tail_destroy(t); *out = nullptr; }
    return rc;
}
int dvbs2gpu_dvbs_tail_reset(dvbs2gpu_dvbs_tail* t) {
    if (!t) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(t->ctx->device));
    const size_t n = (size_t)t->nstreams;
    HIP_TRY(hipMemset(t->d_hist[0], 0, n * 1632 * 8)); HIP_TRY(hipMemset(t->d_hist[1], 0, n * 1632 * 8));
    HIP_TRY(hipMemset(t->d_forney, 0, n * DVBS_FORNEY_HIST));
    HIP_TRY(hipMemset(t->d_errs, 0, n * 2 * sizeof(int)));
    std::vector<DvbsTailState> st(n);
    memset(st.data(), 0, n * sizeof(DvbsTailState));
    for (auto& s : st) s.prbs_pos = -1;
    HIP_TRY(hipMemcpy(t->d_state, st.data(), n * sizeof(DvbsTailState), hipMemcpyHostToDevice));
    t->cur = 0;
    return 0;
}
void dvbs2gpu_dvbs_tail_destroy(dvbs2gpu_dvbs_tail* t) {
    if (!t) return;
    void* ps[] = {t->d_hist[0], t->d_hist[1], t->d_v, t->d_hit, t->d_nframes, t->d_errs, t->d_frames, t->d_deint, t->d_forney, t->d_status,
                  t->d_rs_err, t->d_gf, t->d_prbs, t->d_state, t->d_args};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete t;
}
int dvbs2gpu_dvbs_tail_process_batch(dvbs2gpu_dvbs_tail* t, const uint8_t* const* d_bits, const int* counts, uint8_t* const* d_ts, int cap,
                                     int* out_bytes, void* stream) {
    if (!t || !d_bits || !counts || !d_ts || !out_bytes || cap < 0) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(t->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int n = t->nstreams;
    for (int i = 0; i < n; ++i) {
        if (counts[i] < 0 || counts[i] > t->max_bits) { last_error() = "bit count exceeds max_bits"; return DVBS2GPU_ERR_ARG; }
        if (counts[i] > 0 && !d_bits[i]) return DVBS2GPU_ERR_ARG;
    }
    char* a = (char*)t->d_args;
    const uint8_t** d_in = (const uint8_t**)a;
    uint8_t** d_out = (uint8_t**)(a + sizeof(void*) * n);
    int* d_cnt = (int*)(a + 2 * sizeof(void*) * n);
    int* d_ob = d_cnt + n;
    HIP_TRY(hipMemcpyAsync(d_in, d_bits, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_out, d_ts, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_cnt, counts, sizeof(int) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(dvbs_tail_launch(d_in, d_cnt, n, t->max_bits, t->d_hist[t->cur], t->d_hist[t->cur ^ 1], t->d_v, t->v_stride, t->max_frames, t->d_hit,
                             t->d_nframes, t->d_errs, t->d_frames, t->d_deint, t->frames_stride, t->d_forney, t->d_status, t->d_gf, t->d_prbs,
                             t->d_state, d_out, cap, d_ob, t->d_rs_err, st));
    t->cur ^= 1;
    HIP_TRY(hipMemcpyAsync(out_bytes, d_ob, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}
/* per stream: frames found by the last call, errors_nor, errors_inv of the deframer (stats_deframer_err = min of the two,
 * module_dvbs_demod.cpp:116) and the RS error counts of the last frame's 8 packets (stats_rs_avg = their mean, :115) */
int dvbs2gpu_dvbs_tail_get_stats(dvbs2gpu_dvbs_tail* t, int stream, int32_t* h_out11) {
    if (!t || stream < 0 || stream >= t->nstreams || !h_out11) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(t->ctx->device));
    int nf = 0, er[2] = {0, 0};
    HIP_TRY(hipMemcpy(&nf, t->d_nframes + stream, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(er, t->d_errs + 2 * stream, 2 * sizeof(int), hipMemcpyDeviceToHost));
    h_out11[0] = nf; h_out11[1] = er[0]; h_out11[2] = er[1];
    for (int i = 0; i < 8; ++i) h_out11[3 + i] = 0;
    if (nf > 0) HIP_TRY(hipMemcpy(h_out11 + 3, t->d_rs_err + (size_t)stream * t->max_frames * 8 + (size_t)(nf - 1) * 8, 8 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

/* taps of the last process_batch / rs_stage call of one stream (see include/dvbs2gpu.h) */
int dvbs2gpu_dvbs_tail_get_tap(dvbs2gpu_dvbs_tail* t, int stream, int which, void* h_dst, int cap) {
    if (!t || stream < 0 || stream >= t->nstreams || which < 0 || which > 3 || cap < 0) return DVBS2GPU_ERR_ARG;
    CallGuard guard(t->ctx);
    HIP_TRY(hipSetDevice(t->ctx->device));
    int nf = 0;
    HIP_TRY(hipMemcpy(&nf, t->d_nframes + stream, sizeof(int), hipMemcpyDeviceToHost));
    const void* src;
    size_t bytes;
    switch (which) {
        case 0: src = t->d_frames + (size_t)stream * t->frames_stride; bytes = (size_t)nf * 1632; break;
        case 1: src = t->d_deint + (size_t)stream * t->frames_stride; bytes = (size_t)nf * 1632; break;
        case 2: src = t->d_status + (size_t)stream * t->max_frames * 8; bytes = (size_t)nf * 8; break;
        default: src = t->d_rs_err + (size_t)stream * t->max_frames * 8; bytes = (size_t)nf * 8 * sizeof(int); break;
    }
    if (h_dst && bytes) HIP_TRY(hipMemcpy(h_dst, src, std::min(bytes, (size_t)cap), hipMemcpyDeviceToHost));
    return (int)bytes;
}
int dvbs2gpu_dvbs_tail_rs_stage(dvbs2gpu_dvbs_tail* t, const uint8_t* h_packets, int npackets, int skip_rs, uint8_t* h_ts, int cap) {
    if (!t || !h_packets || npackets <= 0 || npackets % 8 || npackets > t->max_frames * 8 || !h_ts || cap < 0) return DVBS2GPU_ERR_ARG;
    // TEST HOOK ON A DEDICATED HANDLE (include/dvbs2gpu.h): it overwrites the handle's de-interleaved packets, status and frame counts and advances
    // stream 0's energy-dispersal phase and last RS message -- never call it on a tail that is receiving
    CallGuard guard(t->ctx);
    HIP_TRY(hipSetDevice(t->ctx->device));
    HIP_TRY(hipDeviceSynchronize());            // (the engine's streams are non-blocking: null-stream copies do not order themselves behind them)
    const int n = t->nstreams, nf = npackets / 8;
    std::vector<int> nfr(n, 0);
    nfr[0] = nf;
    HIP_TRY(hipMemcpy(t->d_nframes, nfr.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t->d_deint, h_packets, (size_t)npackets * 204, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(t->d_status, 1, (size_t)n * t->max_frames * 8));
    HIP_TRY(hipMemset(t->d_rs_err, 0, (size_t)n * t->max_frames * 8 * sizeof(int)));
    uint8_t* d_ts = nullptr;
    HIP_TRY(hipMalloc((void**)&d_ts, (size_t)std::max(cap, 1) * n));
    char* a = (char*)t->d_args;
    uint8_t** d_out = (uint8_t**)(a + sizeof(void*) * n);
    int* d_ob = (int*)(a + 2 * sizeof(void*) * n) + n;
    std::vector<uint8_t*> outs(n);
    for (int i = 0; i < n; ++i) outs[i] = d_ts + (size_t)i * std::max(cap, 1);
    hipError_t e = hipMemcpy(d_out, outs.data(), sizeof(void*) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dvbs_tail_rs_finish_launch(n, t->max_frames, t->d_nframes, t->d_deint, t->frames_stride, t->d_status, t->d_gf, t->d_prbs, t->d_state,
                                                        d_out, cap, d_ob, t->d_rs_err, skip_rs, 0);
    int nb = 0;
    if (e == hipSuccess) e = hipMemcpy(&nb, d_ob, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && nb > 0) e = hipMemcpy(h_ts, d_ts, (size_t)nb, hipMemcpyDeviceToHost);
    (void)hipFree(d_ts);
    if (e != hipSuccess) return fail_hip(e, "dvbs tail rs stage");
    return nb;
}
int dvbs2gpu_dvbs_depuncture(dvbs2gpu_ctx* ctx, int period, int mode, const uint8_t* h_in, int size, uint8_t* h_out, int out_cap, int32_t* h_state4) {
    if (!ctx || !h_in || !h_out || !h_state4 || size <= 0 || mode < 0 || mode > 2 || (mode != 2 && period != 3 && period != 6)) return DVBS2GPU_ERR_ARG;
    if (out_cap < 2 * size + 2) return DVBS2GPU_ERR_CAPACITY;
    CallGuard guard(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)size + out_cap + 64));
    uint8_t* d_in = d, *d_out = d + size;
    int* d_st = (int*)(d + (((size_t)size + out_cap + 3) & ~(size_t)3));
    int n = 0;
    hipError_t e = hipMemcpy(d_in, h_in, size, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_out, h_out, out_cap, hipMemcpyHostToDevice);       // (bytes the stage does not write keep the caller's fill)
    if (e == hipSuccess) e = hipMemcpy(d_st, h_state4, 4 * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dvbs_depunc_stage_launch(period, mode, d_in, size, d_out, d_st, d_st + 4, 0);
    if (e == hipSuccess) e = hipMemcpy(h_out, d_out, out_cap, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_state4, d_st, 4 * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&n, d_st + 4, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "dvbs depuncture stage");
    return n;
}

}  // extern "C"
