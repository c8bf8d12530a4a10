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
