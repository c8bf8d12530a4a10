// DVB-S segment receiver: ONE fast DVB-S carrier on the many-stream receiver bank (the DVB-S counterpart of segrx.hip).
// A DVB-S stream's loops (AGC, FLL, timing recovery, Costas) and its Viterbi decoder are serial per stream (one stream: 0.6 Msym/s on
// MI355X, below the 2 Msym/s of BASELINE config D).  Here a long chunk of one continuous IQ stream is cut into overlapping segments;
// every segment runs as an independent stream of one dvbs2gpu_dvbs_demod_process_batch call (fresh loops, fresh Viterbi lock search),
// and the segments' decoded bit streams are joined where they overlap: the last bits already handed out are searched for in the next
// segment's output (the decoded bits of a scrambled transport stream do not repeat), and the output continues behind the match.
// The result is the stream's decoded bit sequence in order (one bit per byte, like the bank's output), ready for the DVB-S tail
// (dvbs2gpu_dvbs_tail_*: deframer, de-interleaver, RS, energy dispersal), which is parallel over a long bit stream already.
// Host code above the C ABI; no counterpart in the reference (one DVBSDemod per carrier, serial).
#include "ctx.h"

#include <algorithm>

using namespace s2;
#define g_err last_error()

namespace {
constexpr int KEY_BITS = 64, WIN_BITS = 256, MAX_MISMATCH = 6, TAILWIN = 1024;
}

struct dvbs2gpu_dvbs_segrx {
    dvbs2gpu_ctx* ctx = nullptr;
    dvbs2gpu_dvbs_demod* bank = nullptr;
    int nseg = 0;
    long own_s = 0, warm_s = 0, tail_s = 0, seg_cap = 0, hist_cap = 0, bits_cap = 0;   // samples / bits
    float* d_hist = nullptr;
    float* d_hist2 = nullptr;
    float* d_seg0 = nullptr;
    uint8_t* d_bits = nullptr;          // [nseg][bits_cap]
    long hist_fill = 0;
    std::vector<uint8_t> tail;          // the last bits handed out (at most TAILWIN), in the polarity they were handed out
    bool first_call = true;
    int last_used = 0, last_matched = 0, last_unmatched = 0;
    long long last_bits = 0;
};

namespace {

// position in `b` (n bits, one per byte) right BEHIND the best match of the end of `tail`; -1 when nothing matches.
// QPSK leaves a 180-degree ambiguity that the inner decoder cannot see (the bit stream comes out inverted; the deframer behind accepts
// both, dvbs_ts_deframer.cpp): every segment locks with its own polarity, so the match is tried on the inverted bits as well.
long find_continuation(const std::vector<uint8_t>& tail, const uint8_t* b, long n, int* inverted) {
    const long tn = (long)tail.size();
    *inverted = 0;
    if (tn < WIN_BITS || n < WIN_BITS) return -1;
    // several 64-bit keys inside the last WIN_BITS of the tail: a bit error in one of them must not lose the match
    for (int key_no = 0; key_no < 3; ++key_no) {
        const long key_end = tn - key_no * KEY_BITS;             // key = tail[key_end - 64, key_end)
        unsigned long long key = 0;
        for (int i = 0; i < KEY_BITS; ++i) key = key << 1 | (tail[key_end - KEY_BITS + i] & 1);
        unsigned long long w = 0;
        for (long i = 0; i < n; ++i) {
            w = w << 1 | (b[i] & 1);
            if (i + 1 < KEY_BITS || (w != key && w != ~key)) continue;
            const int inv = w != key;
            // key ends at b[i]: the tail's end would sit at b[i + key_no*64]; compare the last WIN_BITS of the tail
            const long end_b = i + 1 + (long)key_no * KEY_BITS;
            if (end_b > n || end_b < WIN_BITS) continue;
            int mism = 0;
            for (int k = 0; k < WIN_BITS && mism <= MAX_MISMATCH; ++k) mism += (tail[tn - WIN_BITS + k] ^ b[end_b - WIN_BITS + k] ^ inv) & 1;
            if (mism <= MAX_MISMATCH) { *inverted = inv; return end_b; }
        }
    }
    return -1;
}

__global__ void dvbs_segrx_copy_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, long n, int flip) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i] ^ (uint8_t)flip;
}

}  // namespace

extern "C" {

void dvbs2gpu_dvbs_segrx_destroy(dvbs2gpu_dvbs_segrx* r) {
    if (!r) return;
    if (r->bank) dvbs2gpu_dvbs_demod_destroy(r->bank);
    void* ps[] = {r->d_hist, r->d_hist2, r->d_seg0, r->d_bits};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete r;
}

int dvbs2gpu_dvbs_segrx_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_dvbs_cfg* cfg, int nsegments, int own_symbols, int warm_symbols, dvbs2gpu_dvbs_segrx** out) {
    if (!ctx || !cfg || !out || nsegments < 1 || warm_symbols < 8192 || own_symbols < warm_symbols) {
        g_err = "DVB-S segment receiver: needs nsegments >= 1 and own_symbols >= warm_symbols >= 8192";
        return DVBS2GPU_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    auto r = new dvbs2gpu_dvbs_segrx();
    r->ctx = ctx; r->nseg = nsegments;
    r->own_s = 2L * own_symbols; r->warm_s = 2L * warm_symbols;
    r->tail_s = 2L * std::max(warm_symbols / 2, 12288);        // a segment runs on into its successor's part (the Viterbi hands out whole 4096-symbol blocks only)
    r->hist_cap = r->warm_s + r->tail_s;
    r->seg_cap = r->hist_cap + r->own_s + r->tail_s + 64;
    r->bits_cap = r->seg_cap + 4 * 8192;                        // at most 1.75 bits per symbol = 0.875 per sample
    if (r->bits_cap > 0x3fffffffL) {                             // per-segment counts are ints in the bank's entry
        delete r;
        g_err = "DVB-S segment receiver: a segment of this many symbols does not fit the bank's int counts";
        return DVBS2GPU_ERR_ARG;
    }
    int rc = dvbs2gpu_dvbs_demod_create(ctx, cfg, nsegments, (int)r->seg_cap, &r->bank);
    if (rc) { r->bank = nullptr; dvbs2gpu_dvbs_segrx_destroy(r); return rc; }
    hipError_t e = hipMalloc((void**)&r->d_hist, sizeof(float) * 2 * r->hist_cap);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_hist2, sizeof(float) * 2 * r->hist_cap);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_seg0, sizeof(float) * 2 * r->seg_cap);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_bits, (size_t)r->bits_cap * nsegments);
    if (e != hipSuccess) { dvbs2gpu_dvbs_segrx_destroy(r); return fail_hip(e, "hipMalloc(DVB-S segment receiver)"); }
    *out = r;
    return 0;
}

int dvbs2gpu_dvbs_segrx_reset(dvbs2gpu_dvbs_segrx* r) {
    if (!r) return DVBS2GPU_ERR_ARG;
    r->hist_fill = 0; r->tail.clear(); r->first_call = true;
    return 0;
}

long long dvbs2gpu_dvbs_segrx_chunk_samples(dvbs2gpu_dvbs_segrx* r) { return r ? (long long)r->nseg * r->own_s : DVBS2GPU_ERR_ARG; }

int dvbs2gpu_dvbs_segrx_process(dvbs2gpu_dvbs_segrx* r, const float* d_iq, long long count, uint8_t* d_out, long long out_cap) {
    if (!r || count < 0 || out_cap < 0 || (count > 0 && (!d_iq || !d_out))) return DVBS2GPU_ERR_ARG;
    if (count > (long long)r->nseg * r->own_s) { g_err = "DVB-S segment receiver: chunk longer than nsegments * own_symbols"; return DVBS2GPU_ERR_ARG; }
    HIP_TRY(hipSetDevice(r->ctx->device));
    r->last_used = r->last_matched = r->last_unmatched = 0; r->last_bits = 0;
    if (count == 0) return 0;
    const long n = (long)count;
    const int used = (int)((n + r->own_s - 1) / r->own_s);
    std::vector<const float*> in(r->nseg, nullptr);
    std::vector<int> cnt(r->nseg, 0), nbits(r->nseg, 0);
    std::vector<uint8_t*> outp(r->nseg);
    {
        const long head = std::min(n, r->own_s + r->tail_s);
        if (r->hist_fill) HIP_TRY(hipMemcpyAsync(r->d_seg0, r->d_hist, sizeof(float) * 2 * r->hist_fill, hipMemcpyDeviceToDevice, nullptr));
        HIP_TRY(hipMemcpyAsync(r->d_seg0 + 2 * r->hist_fill, d_iq, sizeof(float) * 2 * head, hipMemcpyDeviceToDevice, nullptr));
        in[0] = r->d_seg0; cnt[0] = (int)(r->hist_fill + head);
    }
    for (int g = 1; g < used; ++g) {
        const long a = (long)g * r->own_s - r->warm_s, b = std::min(n, (long)(g + 1) * r->own_s + r->tail_s);
        in[g] = d_iq + 2 * a; cnt[g] = (int)(b - a);
    }
    for (int g = 0; g < r->nseg; ++g) outp[g] = r->d_bits + (size_t)g * r->bits_cap;
    for (int g = used; g < r->nseg; ++g) in[g] = r->d_seg0;       // (unused streams of the bank: zero samples)
    int rc = dvbs2gpu_dvbs_demod_reset(r->bank);
    if (rc) return rc;
    if ((rc = dvbs2gpu_dvbs_demod_process_batch(r->bank, in.data(), cnt.data(), outp.data(), (int)std::min<long>(r->bits_cap, 0x7fffffff), nbits.data()))) return rc;
    r->last_used = used;
    // ---- join the segments' bit streams
    std::vector<uint8_t> hb;
    long long written = 0;
    auto emit = [&](int g, long from, long to, int flip) -> int {  // bits [from, to) of segment g (inverted when flip), device -> device
        if (to <= from) return 0;
        if (written + (to - from) > out_cap) { g_err = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
        const long cnt = to - from;
        hipLaunchKernelGGL(dvbs_segrx_copy_kernel, dim3((unsigned)std::min<long>((cnt + 255) / 256, 4096)), dim3(256), 0, nullptr, d_out + written, outp[g] + from, cnt, flip);
        HIP_TRY(hipGetLastError());
        written += cnt;
        // the window the next segment is matched against follows on the host, in the polarity handed out
        const long keep = std::min<long>(cnt, TAILWIN);
        if (keep == TAILWIN) r->tail.clear();
        for (long i = to - keep; i < to; ++i) r->tail.push_back((uint8_t)((hb[i] ^ flip) & 1));
        if ((long)r->tail.size() > TAILWIN) r->tail.erase(r->tail.begin(), r->tail.end() - TAILWIN);
        return 0;
    };
    for (int g = 0; g < used; ++g) {
        const long nb = nbits[g];
        if (nb <= 0) { ++r->last_unmatched; continue; }
        hb.resize(nb);
        HIP_TRY(hipMemcpy(hb.data(), outp[g], (size_t)nb, hipMemcpyDeviceToHost));
        // the decoder's last bits before the end of a segment are its least reliable (open trellis): they come from the next segment
        const long stop = std::max<long>(0, nb - 128);         // (also for the last segment of a call: the next call's first segment brings them)
        long from;
        int flip = 0;
        if (r->tail.empty()) {
            from = 0;                                              // start of the stream
        } else {
            from = find_continuation(r->tail, hb.data(), nb, &flip);
            if (from < 0) {
                // not found (the segment did not settle inside the overlap, or the previous one lost its end): restart in its second half;
                // the bit stream has a discontinuity here and the deframer behind it resynchronises
                ++r->last_unmatched;
                from = nb / 2;
                flip = 0;
            } else {
                ++r->last_matched;
            }
        }
        if ((rc = emit(g, from, stop, flip))) return rc;
    }
    // ---- history for the next call
    if (n >= r->hist_cap) {
        HIP_TRY(hipMemcpyAsync(r->d_hist, d_iq + 2 * (n - r->hist_cap), sizeof(float) * 2 * r->hist_cap, hipMemcpyDeviceToDevice, nullptr));
        r->hist_fill = r->hist_cap;
    } else {
        const long keep_old = std::min(r->hist_fill, r->hist_cap - n);
        if (keep_old) HIP_TRY(hipMemcpyAsync(r->d_hist2, r->d_hist + 2 * (r->hist_fill - keep_old), sizeof(float) * 2 * keep_old, hipMemcpyDeviceToDevice, nullptr));
        HIP_TRY(hipMemcpyAsync(r->d_hist2 + 2 * keep_old, d_iq, sizeof(float) * 2 * n, hipMemcpyDeviceToDevice, nullptr));
        std::swap(r->d_hist, r->d_hist2);
        r->hist_fill = keep_old + n;
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    r->first_call = false;
    r->last_bits = written;
    return (int)std::min<long long>(written, 0x7fffffff);
}

/* host-only: the join rule by itself (no device work), see include/dvbs2gpu.h */
long long dvbs2gpu_dvbs_segrx_find_join(const uint8_t* h_tail, long long ntail, const uint8_t* h_bits, long long nbits, int* inverted) {
    int inv = 0;
    if (!h_tail || !h_bits || ntail < 0 || nbits < 0) return DVBS2GPU_ERR_ARG;
    const long long keep = std::min<long long>(ntail, TAILWIN);
    const std::vector<uint8_t> tail(h_tail + (ntail - keep), h_tail + ntail);
    const long at = find_continuation(tail, h_bits, (long)nbits, &inv);
    if (inverted) *inverted = inv;
    return at;
}

/* h_out4 = {segments of the last call, joined by a match, without a match (discontinuity / no output), bits returned} */
int dvbs2gpu_dvbs_segrx_get_stats(dvbs2gpu_dvbs_segrx* r, int32_t* h_out4) {
    if (!r || !h_out4) return DVBS2GPU_ERR_ARG;
    h_out4[0] = r->last_used; h_out4[1] = r->last_matched; h_out4[2] = r->last_unmatched; h_out4[3] = (int32_t)std::min<long long>(r->last_bits, 0x7fffffff);
    return 0;
}

}  // extern "C"
