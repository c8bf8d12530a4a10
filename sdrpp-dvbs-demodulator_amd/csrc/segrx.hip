// Segment receiver: ONE fast transponder on the many-stream engine.
// A stream's AGC / NCO / Gardner / PLL recurrences are serial (0.69 Msym/s per stream on MI355X), so a single 27.5 Msym/s
// transponder cannot be followed sample by sample.  Here a long chunk of its IQ is cut into overlapping segments; every segment
// runs as an independent stream of one dvbs2gpu_demod_process_batch call with freshly reset loops, re-acquires on a warm-up prefix,
// and contributes the frames that start in its own part.  Frames are placed on the stream's time axis with the positions the engine
// reports (dvbs2gpu_demod_get_frame_positions), de-duplicated where neighbouring segments overlap and returned in order; a tail
// of the chunk is kept as the warm-up of the next call's first segment, so calls join without a gap.
// This is host code ABOVE the C ABI (it only uses the public entry points); it has no counterpart in the reference, whose one
// DVBS2Demod object per transponder runs the loops serially (main.cpp:588,595).  Where the reference decodes a frame the segment
// receiver returns the same BBFRAME; soft values are not bit-identical (each segment's loops start from reset).
#include "ctx.h"

#include <algorithm>

using namespace s2;
#define g_err last_error()

struct dvbs2gpu_segrx {
    dvbs2gpu_ctx* ctx = nullptr;
    dvbs2gpu_demod_cfg cfg{};
    int nseg = 0, own = 0, warm = 0, kb = 0;
    long spf = 0;                      // samples per PLFRAME at 2 samples per symbol
    long seg_cap = 0, hist_cap = 0, out_stride = 0;
    std::vector<dvbs2gpu_demod*> dm;
    float* d_hist = nullptr;           // the last hist_fill samples of the stream so far
    float* d_hist2 = nullptr;
    float* d_seg0 = nullptr;           // history ++ head of the chunk: the first segment's input
    uint8_t* d_segout = nullptr;
    long hist_fill = 0;
    long long abs_next = 0;            // stream index of the next call's first sample
    long long last_emit = 0;
    bool emitted_any = false, first_call = true;
    int last_found = 0, last_emitted = 0, last_dropped = 0;
};

extern "C" {

void dvbs2gpu_segrx_destroy(dvbs2gpu_segrx* r) {
    if (!r) return;
    for (auto* d : r->dm) if (d) dvbs2gpu_demod_destroy(d);
    void* ps[] = {r->d_hist, r->d_hist2, r->d_seg0, r->d_segout};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete r;
}

int dvbs2gpu_segrx_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_demod_cfg* cfg, int nsegments, int own_frames, int warm_frames, dvbs2gpu_segrx** out) {
    if (!ctx || !cfg || !out || nsegments < 1 || own_frames < 1 || warm_frames < 1 || own_frames < warm_frames) {
        g_err = "segment receiver: needs nsegments >= 1 and own_frames >= warm_frames >= 1";
        return DVBS2GPU_ERR_ARG;
    }
    dvbs2gpu_modcod_info mi;
    int rc = dvbs2gpu_modcod_info_get(cfg->modcod, cfg->shortframes, cfg->pilots, &mi);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    auto r = new dvbs2gpu_segrx();
    r->ctx = ctx; r->cfg = *cfg; r->nseg = nsegments; r->own = own_frames; r->warm = warm_frames;
    r->kb = mi.kbch / 8;
    r->spf = 2L * mi.plframe_symbols;
    r->hist_cap = (long)(warm_frames + 2) * r->spf;
    r->seg_cap = (long)(warm_frames + own_frames + warm_frames / 2 + 4) * r->spf + 64;
    r->out_stride = (long)(warm_frames + own_frames + warm_frames / 2 + 6) * r->kb;
    if (r->seg_cap > 0x3fffffffL || r->out_stride > 0x7fffffffL) {      // per-segment counts are ints in the batch entry
        delete r;
        g_err = "segment receiver: a segment of this many frames does not fit the batch entry's int counts";
        return DVBS2GPU_ERR_ARG;
    }
    r->dm.assign(nsegments, nullptr);
    for (int g = 0; g < nsegments; ++g) {
        if ((rc = dvbs2gpu_demod_create(ctx, cfg, (int)r->seg_cap, &r->dm[g]))) { dvbs2gpu_segrx_destroy(r); return rc; }
    }
    hipError_t e = hipMalloc((void**)&r->d_hist, sizeof(float) * 2 * r->hist_cap);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_hist2, sizeof(float) * 2 * r->hist_cap);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_seg0, sizeof(float) * 2 * r->seg_cap);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_segout, (size_t)r->out_stride * nsegments);
    if (e != hipSuccess) { dvbs2gpu_segrx_destroy(r); return fail_hip(e, "hipMalloc(segment receiver)"); }
    *out = r;
    return 0;
}

int dvbs2gpu_segrx_reset(dvbs2gpu_segrx* r) {
    if (!r) return DVBS2GPU_ERR_ARG;
    r->hist_fill = 0; r->abs_next = 0; r->last_emit = 0; r->emitted_any = false; r->first_call = true;
    return 0;
}

/* samples one call can take: nsegments * own_frames PLFRAMEs */
long long dvbs2gpu_segrx_chunk_samples(dvbs2gpu_segrx* r) { return r ? (long long)r->nseg * r->own * r->spf : DVBS2GPU_ERR_ARG; }

int dvbs2gpu_segrx_process(dvbs2gpu_segrx* r, const float* d_iq, long long count, uint8_t* d_out, long long out_cap) {
    if (!r || count < 0 || out_cap < 0 || (count > 0 && (!d_iq || !d_out))) return DVBS2GPU_ERR_ARG;
    if (count > (long long)r->nseg * r->own * r->spf) { g_err = "segment receiver: chunk longer than nsegments * own_frames frames"; return DVBS2GPU_ERR_ARG; }
    HIP_TRY(hipSetDevice(r->ctx->device));
    if (r->ctx->pipeline_fec) { g_err = "segment receiver: switch the context's throughput mode off (each call is a complete batch)"; return DVBS2GPU_ERR_ARG; }
    r->last_found = r->last_emitted = r->last_dropped = 0;
    if (count == 0) return 0;
    const long n = (long)count, spf = r->spf, own_s = (long)r->own * spf, warm_s = (long)r->warm * spf;
    const int used = (int)((n + own_s - 1) / own_s);
    const long tail_s = (long)(r->warm / 2 + 2) * spf;      // a segment runs on into its successor's part: frames the successor is still settling on come from here
    // ---- segment inputs: segment 0 = history ++ head of the chunk, segment g = [g*own - warm, (g+1)*own + warm/2 + 2) frames of the chunk
    std::vector<const float*> in(used);
    std::vector<int> cnt(used);
    std::vector<long long> seg_abs(used);
    std::vector<uint8_t*> outp(used);
    {
        const long head = std::min(n, own_s + tail_s);
        if (r->hist_fill) HIP_TRY(hipMemcpyAsync(r->d_seg0, r->d_hist, sizeof(float) * 2 * r->hist_fill, hipMemcpyDeviceToDevice, nullptr));
        HIP_TRY(hipMemcpyAsync(r->d_seg0 + 2 * r->hist_fill, d_iq, sizeof(float) * 2 * head, hipMemcpyDeviceToDevice, nullptr));
        in[0] = r->d_seg0; cnt[0] = (int)(r->hist_fill + head); seg_abs[0] = r->abs_next - r->hist_fill;
    }
    for (int g = 1; g < used; ++g) {
        const long a = (long)g * own_s - warm_s, b = std::min(n, (long)(g + 1) * own_s + tail_s);
        in[g] = d_iq + 2 * a; cnt[g] = (int)(b - a); seg_abs[g] = r->abs_next + a;
    }
    for (int g = 0; g < used; ++g) {
        outp[g] = r->d_segout + (size_t)g * r->out_stride;
        int rc = dvbs2gpu_demod_reset(r->dm[g]);
        if (rc) return rc;
    }
    std::vector<int> nb(used);
    int rc = dvbs2gpu_demod_process_batch(r->dm.data(), used, in.data(), cnt.data(), outp.data(), (int)r->out_stride, nb.data());
    if (rc) return rc;
    // ---- frames on the stream's time axis
    struct Found { long long pos; int seg, idx; bool trusted, decoded; long off; };
    std::vector<Found> found;
    std::vector<int64_t> pos;
    std::vector<dvbs2gpu_frame_stats> fst;
    for (int g = 0; g < used; ++g) {
        const int nf = nb[g] / r->kb;
        pos.assign(nf, 0);
        fst.assign(nf, dvbs2gpu_frame_stats{});
        const int np = dvbs2gpu_demod_get_frame_positions(r->dm[g], pos.data(), nf);
        const int ns = dvbs2gpu_demod_get_stats(r->dm[g], fst.data(), nf);
        for (int f = 0; f < std::min(nf, np); ++f) {
            const long off = 2 * (long)pos[f];
            const bool start_of_stream = r->first_call && g == 0;
            const bool decoded = f < ns && fst[f].bch_corrections >= 0 && fst[f].ldpc_trials >= 0;
            found.push_back(Found{seg_abs[g] + off, g, f, start_of_stream || off >= (warm_s * 3) / 4, decoded, off});
        }
    }
    r->last_found = (int)found.size();
    std::sort(found.begin(), found.end(), [](const Found& a, const Found& b) { return a.pos < b.pos; });
    // one frame per cluster of sightings closer than half a frame: a sighting whose FEC succeeded wins (a segment whose loops have not
    // settled yet delivers frames the BCH check rejects), then one behind its segment's warm-up, then the longer-settled one
    auto score = [](const Found& f) { return (f.decoded ? 2 : 0) + (f.trusted ? 1 : 0); };
    std::vector<Found> keep;
    for (size_t i = 0; i < found.size();) {
        size_t j = i, best = i;
        while (j < found.size() && found[j].pos - found[i].pos < spf / 2) {
            const Found &c = found[j], &bst = found[best];
            if (score(c) > score(bst) || (score(c) == score(bst) && c.off > bst.off)) best = j;
            ++j;
        }
        keep.push_back(found[best]);
        i = j;
    }
    long long bytes = 0;
    // frames of one segment sit back to back in its output buffer: consecutive ones leave in one copy
    long long run_dst = 0; size_t run_src = 0, run_len = 0;
    auto flush = [&]() -> int {
        if (run_len) HIP_TRY(hipMemcpyAsync(d_out + run_dst, r->d_segout + run_src, run_len, hipMemcpyDeviceToDevice, nullptr));
        run_len = 0;
        return 0;
    };
    for (const Found& k : keep) {
        if (!k.trusted && !k.decoded) { ++r->last_dropped; continue; }                                    // seen only inside a warm-up and not decodable there
        if (r->emitted_any && k.pos < r->last_emit + spf / 2) continue;                                   // the previous call returned it
        if (bytes + r->kb > out_cap) { g_err = "output buffer too small"; return DVBS2GPU_ERR_CAPACITY; }
        const size_t src = (size_t)k.seg * r->out_stride + (size_t)k.idx * r->kb;
        if (run_len && src == run_src + run_len) {
            run_len += (size_t)r->kb;
        } else {
            if ((rc = flush())) return rc;
            run_dst = bytes; run_src = src; run_len = (size_t)r->kb;
        }
        bytes += r->kb;
        r->last_emit = k.pos; r->emitted_any = true; ++r->last_emitted;
    }
    if ((rc = flush())) return rc;
    // ---- history for the next call: the last hist_cap samples of the stream
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
    r->abs_next += n;
    r->first_call = false;
    return (int)std::min<long long>(bytes, 0x7fffffff);
}

/* h_out3 = {frame sightings of the last call, frames returned, frames seen only inside a warm-up (dropped)} */
int dvbs2gpu_segrx_get_stats(dvbs2gpu_segrx* r, int32_t* h_out3) {
    if (!r || !h_out3) return DVBS2GPU_ERR_ARG;
    h_out3[0] = r->last_found; h_out3[1] = r->last_emitted; h_out3[2] = r->last_dropped;
    return 0;
}

}  // extern "C"
