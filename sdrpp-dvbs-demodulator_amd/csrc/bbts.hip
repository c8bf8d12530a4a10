// BBFRAME -> MPEG-TS / GSE parser bank: what the reference's sink handler runs on the demodulator's
// output (main.cpp:532-558 -> dsp::dvbs2::BBFrameTSParser::work, dvbs2/bbframe_ts_parser.cpp:104-390), for `nstreams`
// independent streams whose BBFRAMEs are already resident in HBM (they are the FEC's output).
//
// MPEG-TS frames (TS/GS = 11) never leave the device:
//   bbts_plan_kernel   one thread per stream walks its frames' 10-byte BBHEADERs in order -- CRC-8, DFL/SYNCD checks,
//                      resynchronisation, the carried partial packet -- and turns the reference's byte-by-byte loop into
//                      one copy descriptor per frame (the only serial part: a handful of integer operations per frame);
//   bbts_emit_kernel   one workgroup per (frame, stream) moves the bytes: 0x47 + 187 bytes per packet, the first packet of
//                      a frame completed from the tail of the previous one (HBM-bound byte movement, 2 x DFL/8 per frame).
// GSE frames (TS/GS = 01) are a byte-serial protocol parse with up to 64 KiB of reassembly state per fragment id; a stream
// that carries one in a call is handed, for that call, to the native host parser below (BbtsHostParser), which shares
// the synchronisation state with the device path.
#include "ctx.h"

#include <memory>

using namespace s2;
#define g_err last_error()

namespace s2 {

constexpr int TS = 188;
constexpr int REASM_STRIDE = 192;

struct BbtsDevState {              // per stream, device resident
    int synched, count;
    int hdr[11];                   // ts_gs, sis_mis, ccm_acm, issyi, npd, ro, isi, upl, dfl, sync, syncd (BBHeader, bbframe_ts_parser.h:37-66)
    int last_cnt, last_proc, pad;
};
struct BbtsFrameDesc {
    int src, npk, pre_len, pre_src, out_off, pad[3];   // pre_src < 0: the carried partial lives in the state buffer
};
struct BbtsStreamPlan {
    int needs_host, out_bytes, fin_len, fin_src;       // fin_src < 0: keep the state buffer's bytes
};

// check_crc8 (bbframe_ts_parser.cpp:70-83): LSB-first register, polynomial 0xAB (reflected 0xD5), over `nbits` MSB-first bits
__host__ __device__ inline unsigned crc8_bits(const uint8_t* in, int nbits) {
    unsigned crc = 0;
    for (int n = 0; n < nbits; ++n) {
        unsigned fb = ((in[n >> 3] >> (7 - (n & 7))) ^ crc) & 1u;
        crc >>= 1;
        if (fb) crc ^= 0xAB;
    }
    return crc;
}
struct HeaderFields { int v[11]; };
__host__ __device__ inline HeaderFields parse_bbheader(const uint8_t* b) {
    HeaderFields h;
    h.v[0] = b[0] >> 6; h.v[1] = (b[0] >> 5) & 1; h.v[2] = (b[0] >> 4) & 1; h.v[3] = (b[0] >> 3) & 1; h.v[4] = (b[0] >> 2) & 1;
    h.v[5] = b[0] & 3;
    h.v[6] = h.v[1] == 0 ? b[1] : 0;
    h.v[7] = b[2] << 8 | b[3];
    h.v[8] = b[4] << 8 | b[5];
    h.v[9] = b[6];
    h.v[10] = b[7] << 8 | b[8];
    return h;
}
// header validation of work() (.cpp:119-152): true when the frame is parsed at all
__host__ __device__ inline bool header_ok(const uint8_t* frame, int max_dfl, HeaderFields* h) {
    if (crc8_bits(frame, 80) != 0) return false;
    *h = parse_bbheader(frame);
    const int dfl = h->v[8], syncd = h->v[10];
    if ((unsigned)dfl > (unsigned)max_dfl || syncd >= dfl - 8) return false;
    return dfl % 8 == 0;
}

__global__ void bbts_plan_kernel(const uint8_t* const* __restrict__ in, const int* __restrict__ nframes, int nstreams, int fbytes, int max_dfl,
                                 int max_frames, BbtsDevState* __restrict__ state, BbtsFrameDesc* __restrict__ desc,
                                 BbtsStreamPlan* __restrict__ plan, int* __restrict__ out_bytes) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstreams) return;
    BbtsDevState st = state[s];
    const uint8_t* bb = in[s];
    const int nf = nframes[s];
    BbtsFrameDesc* d = desc + (size_t)s * max_frames;
    int synched = st.synched, pre_len = st.count, pre_src = -1, out_off = 0, proc = 0, needs_host = 0;
    for (int f = 0; f < nf; ++f) {
        BbtsFrameDesc e = {0, 0, 0, -1, 0, {0, 0, 0}};
        const int base = f * fbytes;
        uint8_t hb[10];
        for (int k = 0; k < 10; ++k) hb[k] = bb[base + k];
        HeaderFields h;
        if (!header_ok(hb, max_dfl, &h)) { synched = 0; d[f] = e; continue; }
        int df = h.v[8] / 8, pos = base + 10;
        if (!synched) {
            const int skip = h.v[10] / 8 + 1;
            pos += skip; df -= skip;
            pre_len = 0; synched = 1;
        }
        for (int k = 0; k < 11; ++k) st.hdr[k] = h.v[k];
        ++proc;
        if (h.v[0] == 1) { needs_host = 1; break; }
        if (h.v[0] == 3) {
            if (df >= TS) {
                // the reference's while loop (.cpp:178-199) in closed form: the first packet absorbs the carried partial
                const int d1 = pre_len > 0 ? df - (TS - pre_len) : df;
                const int n = (pre_len > 0 ? 1 : 0) + d1 / TS, rem = d1 % TS;
                e.src = pos; e.npk = n; e.pre_len = pre_len; e.pre_src = pre_src; e.out_off = out_off;
                out_off += n * TS;
                pre_len = rem; pre_src = pos + df - rem;
            } else if (df > 0) {
                pre_len = df; pre_src = pos;          // .cpp:201-205: a short data field REPLACES the partial
            }
        }
        d[f] = e;
    }
    BbtsStreamPlan p;
    p.needs_host = needs_host;
    if (needs_host) {
        for (int f = 0; f < nf; ++f) d[f].npk = 0;
        p.out_bytes = 0; p.fin_len = TS; p.fin_src = -1;
    } else {
        st.synched = synched; st.count = pre_len; st.last_cnt = nf; st.last_proc = proc;
        state[s] = st;
        p.out_bytes = out_off; p.fin_len = pre_len; p.fin_src = pre_src;
    }
    plan[s] = p;
    out_bytes[s] = p.out_bytes;
}

__global__ void __launch_bounds__(256) bbts_emit_kernel(const uint8_t* const* __restrict__ in, uint8_t* const* __restrict__ out,
                                                        const int* __restrict__ nframes, int max_frames,
                                                        const BbtsFrameDesc* __restrict__ desc, const BbtsStreamPlan* __restrict__ plan,
                                                        const uint8_t* __restrict__ reasm_old, uint8_t* __restrict__ reasm_new) {
    const int s = blockIdx.y, f = blockIdx.x;
    const uint8_t* bb = in[s];
    const uint8_t* old = reasm_old + (size_t)s * REASM_STRIDE;
    if (f == max_frames) {                      // the partial packet carried into the next call
        const BbtsStreamPlan p = plan[s];
        uint8_t* nw = reasm_new + (size_t)s * REASM_STRIDE;
        for (int i = threadIdx.x; i < p.fin_len; i += blockDim.x) nw[i] = p.fin_src < 0 ? old[i] : bb[p.fin_src + i];
        return;
    }
    if (f >= nframes[s]) return;
    const BbtsFrameDesc e = desc[(size_t)s * max_frames + f];
    if (e.npk == 0) return;
    const uint8_t* pre = e.pre_src < 0 ? old : bb + e.pre_src;
    const uint8_t* src = bb + e.src - e.pre_len;          // virtual stream = partial ++ data field
    uint8_t* o = out[s] + e.out_off;
    auto fetch = [&](int i) -> unsigned {                  // output byte i of this frame's packets
        const int b = i % TS;
        if (b == 0) return 0x47u;                          // TS_SYNC_BYTE in place of the CRC-8 of the previous packet
        const int u = i - 1;                               // packet p is bytes [188 p, 188 p + 187) of the virtual stream; its
        return u < e.pre_len ? pre[u] : src[u];            // 188th byte (the CRC-8 of this packet) is dropped
    };
    const int nbytes = e.npk * TS;
    if ((reinterpret_cast<uintptr_t>(o) & 3) == 0) {
        // 188 = 4 * 47: an output dword never straddles two packets, and its four source bytes are contiguous (one unaligned
        // dword load at src + i - 1; the byte under a packet's sync position is replaced)
        typedef unsigned __attribute__((aligned(1))) unaligned_u32;
        for (int w = threadIdx.x; w < nbytes / 4; w += blockDim.x) {
            const int i = 4 * w;
            unsigned v;
            if (i - 1 >= e.pre_len) {
                v = *reinterpret_cast<const unaligned_u32*>(src + i - 1);
                if (i % TS == 0) v = (v & ~0xffu) | 0x47u;
            } else {
                v = fetch(i) | fetch(i + 1) << 8 | fetch(i + 2) << 16 | fetch(i + 3) << 24;
            }
            reinterpret_cast<unsigned*>(o)[w] = v;
        }
    } else {
        for (int i = threadIdx.x; i < nbytes; i += blockDim.x) o[i] = (uint8_t)fetch(i);
    }
}

// ---------------------------------------------------------------------------------------------- host parser (GSE streams)
// Full BBFrameTSParser::work semantics for one stream.  Where the reference's behaviour is undefined (reads past the
// input buffer, writes past the output or the 64 KiB reassembly buffers, negative copy lengths) the rules stated in
// include/dvbs2gpu.h apply.
class BbtsHostParser {
public:
    int synched = 0, count = 0;
    uint8_t partial[TS] = {0};
    int hdr[11] = {0};
    int last_gse_crc_err = 0, last_cnt = 0, last_proc = 0;

    BbtsHostParser() {
        for (unsigned i = 0; i < 256; ++i) {
            uint32_t r = i << 24;
            for (int b = 0; b < 8; ++b) r = (r << 1) ^ ((r >> 31) ? 0x04c11db7u : 0u);
            tab_[i] = r;
        }
    }

    // returns bytes produced or DVBS2GPU_ERR_CAPACITY
    int run(const uint8_t* bb, int cnt, int fbytes, int max_dfl, uint8_t* out, int cap) {
        in_ = bb; in_end_ = (long)fbytes * cnt; out_ = out; cap_ = cap; w_ = 0;
        int proc = 0;
        bool stop = false;
        for (int f = 0; f < cnt && !stop; ++f) {
            const long base = (long)fbytes * f;
            HeaderFields h;
            if (!header_ok(bb + base, max_dfl, &h)) { synched = 0; continue; }
            long pos = base + 10;
            int df = h.v[8] / 8;
            if (!synched) {
                const int skip = h.v[10] / 8 + 1;
                pos += skip; df -= skip; count = 0; synched = 1;
            }
            memcpy(hdr, h.v, sizeof(hdr));
            ++proc;
            switch (h.v[0]) {
            case 3: {
                const int rc = ts_frame(pos, df);
                if (rc < 0) { synched = 0; return DVBS2GPU_ERR_CAPACITY; }
                stop = rc > 0;
                break;
            }
            case 1:
                if (!h.v[3] && !h.v[4] && h.v[7] == 0) gse_frame(pos, h.v[8] / 8);
                break;
            default: break;
            }
        }
        last_cnt = cnt; last_proc = proc;
        return w_;
    }

private:
    struct Reassembly {
        bool busy = false;
        int frag_id = 0;
        long fill = 0;
        unsigned proto = 0;
        uint32_t crc = 0;
        std::unique_ptr<uint8_t[]> data;      // 65536 bytes, allocated on first use
    };
    Reassembly slots_[3];
    uint32_t tab_[256];
    const uint8_t* in_ = nullptr;
    long in_end_ = 0;
    uint8_t* out_ = nullptr;
    int cap_ = 0, w_ = 0;

    uint32_t crc32(uint32_t c, long off, long n) const {
        for (long i = 0; i < n; ++i) c = (c << 8) ^ tab_[(c >> 24) ^ in_[off + i]];
        return c;
    }
    // 1: the output is nearly full, stop after this frame (.cpp:206-209); -1: undefined in the reference; 0 otherwise
    int ts_frame(long pos, int df) {
        while (df >= TS && cap_ - w_ > TS) {
            uint8_t* o = out_ + w_;
            o[0] = 0x47;
            if (count > 0) {
                const int need = TS - count;
                memcpy(partial + count, in_ + pos, need);
                memcpy(o + 1, partial, TS - 1);
                pos += need; df -= need; count = 0;
            } else {
                memcpy(o + 1, in_ + pos, TS - 1);
                pos += TS; df -= TS;
            }
            w_ += TS;
        }
        if (df >= TS) return -1;
        if (df > 0) { memcpy(partial, in_ + pos, df); count = df; }
        return cap_ - w_ <= TS ? 1 : 0;
    }
    void emit_gre(unsigned proto, const uint8_t* p, long n) {
        const bool known = proto == 0x0800 || proto == 0x86DD;
        const long total = 2 + (known ? 2 : 0) + n;
        if (n < 0 || w_ + total > cap_) return;
        uint8_t* o = out_ + w_;
        *o++ = 0; *o++ = 0;                    // GRE: no checksum, no key, no sequence number, version 0
        if (known) { *o++ = (uint8_t)(proto >> 8); *o++ = (uint8_t)proto; }
        memcpy(o, p, n);
        w_ += (int)total;
    }
    void gse_frame(long start, int dfl_bytes) {
        long at = start;
        const long end = start + dfl_bytes;
        while (at < end) {
            if (at + 2 > in_end_) return;
            const unsigned h1 = in_[at];
            const bool first = h1 & 0x80, last = h1 & 0x40;
            const bool label6 = (h1 & 0x30) == 0;          // the reference tests ((h1 & 0x30) >> 2) against 0 and 2: only 0 can match
            if (!first && !last && label6) return;          // padding
            const unsigned field = (h1 & 0x0f) << 8 | in_[at + 1];
            // bytes between the 2-byte GSE header and the payload; the length arithmetic is uint16 in the reference
            const int fixed = first && last ? 2 : first ? 5 : 1;
            const int label = first && label6 ? 6 : 0;
            const long plen = (field - fixed - label) & 0xffff;
            const long body = at + 2 + fixed + label;
            if (body + plen > in_end_) return;
            if (first && last) {
                emit_gre(in_[at + 2] << 8 | in_[at + 3], in_ + body, plen);
            } else {
                const int id = in_[at + 2];
                Reassembly* r = nullptr;
                for (auto& s : slots_) {
                    if (first ? (!s.busy || s.frag_id == id) : (s.busy && s.frag_id == id)) { r = &s; break; }
                }
                if (r && first) {
                    if (!r->data) r->data.reset(new uint8_t[65536]);
                    r->busy = true; r->frag_id = id;
                    r->proto = in_[at + 5] << 8 | in_[at + 6];
                    memcpy(r->data.get(), in_ + body, plen);
                    r->fill = plen;
                    // CRC-32 over total length, protocol type, label, payload
                    r->crc = crc32(crc32(crc32(0xffffffffu, at + 3, 4), at + 7, label), body, plen);
                } else if (r) {
                    if (r->fill + plen > 65536) {
                        r->busy = false;
                    } else if (!last) {
                        memcpy(r->data.get() + r->fill, in_ + body, plen);
                        r->fill += plen;
                        r->crc = crc32(r->crc, body, plen);
                    } else {
                        memcpy(r->data.get() + r->fill, in_ + body, plen);
                        r->busy = false;
                        r->fill += plen - 4;
                        r->crc = crc32(r->crc, body, plen - 4);
                        const uint32_t rx = (uint32_t)in_[body + plen - 4] << 24 | (uint32_t)in_[body + plen - 3] << 16 |
                                            (uint32_t)in_[body + plen - 2] << 8 | in_[body + plen - 1];
                        last_gse_crc_err = r->crc != rx;
                        if (!last_gse_crc_err) emit_gre(r->proto, r->data.get(), r->fill);
                    }
                }
            }
            at = body + plen;
        }
    }
};

}  // namespace s2

struct dvbs2gpu_bbts {
    dvbs2gpu_ctx* ctx = nullptr;
    int nstreams = 0, kbch = 0, max_frames = 0;
    BbtsDevState* d_state = nullptr;
    uint8_t* d_reasm[2] = {nullptr, nullptr};
    int cur = 0;
    BbtsFrameDesc* d_desc = nullptr;
    BbtsStreamPlan* d_plan = nullptr;
    void* d_args = nullptr;                    // [in ptrs][out ptrs][nframes][out bytes]
    uint8_t *d_in1 = nullptr, *d_out1 = nullptr;   // staging of the single-stream host-buffer entry point
    size_t out1_cap = 0;
    std::vector<std::unique_ptr<BbtsHostParser>> host;
    std::vector<BbtsStreamPlan> h_plan;
    std::vector<uint8_t> h_in, h_out;
};

extern "C" {

void dvbs2gpu_bbts_destroy(dvbs2gpu_bbts* b) {
    if (!b) return;
    void* ps[] = {b->d_state, b->d_reasm[0], b->d_reasm[1], b->d_desc, b->d_plan, b->d_args, b->d_in1, b->d_out1};
    for (void* p : ps) if (p) (void)hipFree(p);
    delete b;
}

int dvbs2gpu_bbts_set_frame_size(dvbs2gpu_bbts* b, int kbch_bits) {
    if (!b || kbch_bits < 88 || kbch_bits % 8 || kbch_bits > 65536) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ctx->device));
    // setFrameSize (.cpp:31-42) forgets the synchronisation; header copy and counters stay, as do the GSE slots
    std::vector<BbtsDevState> st(b->nstreams);
    HIP_TRY(hipMemcpy(st.data(), b->d_state, st.size() * sizeof(BbtsDevState), hipMemcpyDeviceToHost));
    for (auto& s : st) { s.synched = 0; s.count = 0; }
    HIP_TRY(hipMemcpy(b->d_state, st.data(), st.size() * sizeof(BbtsDevState), hipMemcpyHostToDevice));
    b->kbch = kbch_bits;
    return 0;
}

int dvbs2gpu_bbts_create(dvbs2gpu_ctx* ctx, int nstreams, int kbch_bits, int max_frames, dvbs2gpu_bbts** out) {
    if (!ctx || !out || nstreams <= 0 || max_frames <= 0 || kbch_bits < 88 || kbch_bits % 8 || kbch_bits > 65536) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    auto b = new dvbs2gpu_bbts();
    b->ctx = ctx; b->nstreams = nstreams; b->kbch = kbch_bits; b->max_frames = max_frames;
    b->host.resize(nstreams);
    b->h_plan.resize(nstreams);
    const size_t n = (size_t)nstreams;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemset(*p, 0, bytes); } };
    A((void**)&b->d_state, n * sizeof(BbtsDevState));
    A((void**)&b->d_reasm[0], n * REASM_STRIDE); A((void**)&b->d_reasm[1], n * REASM_STRIDE);
    A((void**)&b->d_desc, n * max_frames * sizeof(BbtsFrameDesc));
    A((void**)&b->d_plan, n * sizeof(BbtsStreamPlan));
    A(&b->d_args, n * (2 * sizeof(void*) + 2 * sizeof(int)));
    if (e != hipSuccess) { dvbs2gpu_bbts_destroy(b); return fail_hip(e, "hipMalloc(bbts)"); }
    *out = b;
    return 0;
}

int dvbs2gpu_bbts_process_batch(dvbs2gpu_bbts* b, const uint8_t* const* d_bb, const int* nframes, uint8_t* const* d_out, int cap,
                                int* out_bytes, void* stream) {
    if (!b || !d_bb || !nframes || !d_out || !out_bytes || cap < 0) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int n = b->nstreams, fbytes = b->kbch / 8;
    for (int i = 0; i < n; ++i) {
        if (nframes[i] < 0 || nframes[i] > b->max_frames) { g_err = "frame count exceeds max_frames"; return DVBS2GPU_ERR_ARG; }
        if (nframes[i] > 0 && (!d_bb[i] || !d_out[i])) return DVBS2GPU_ERR_ARG;
        // every TS packet is at most the bytes it was cut from + one carried partial; the reference additionally stops
        // when fewer than 189 bytes are left (.cpp:178,206): with this bound it never does
        if ((long)cap < (long)nframes[i] * fbytes + 2 * TS) { g_err = "cap must be >= nframes*kbch/8 + 376"; return DVBS2GPU_ERR_CAPACITY; }
    }
    char* a = (char*)b->d_args;
    const uint8_t** a_in = (const uint8_t**)a;
    uint8_t** a_out = (uint8_t**)(a + sizeof(void*) * n);
    int* a_nf = (int*)(a + 2 * sizeof(void*) * n);
    int* a_ob = a_nf + n;
    HIP_TRY(hipMemcpyAsync(a_in, d_bb, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(a_out, d_out, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(a_nf, nframes, sizeof(int) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bbts_plan_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a_in, a_nf, n, fbytes, b->kbch - 80, b->max_frames, b->d_state,
                       b->d_desc, b->d_plan, a_ob);
    hipLaunchKernelGGL(bbts_emit_kernel, dim3(b->max_frames + 1, n), dim3(256), 0, st, a_in, a_out, a_nf, b->max_frames, b->d_desc, b->d_plan,
                       b->d_reasm[b->cur], b->d_reasm[b->cur ^ 1]);
    HIP_TRY(hipGetLastError());
    b->cur ^= 1;
    HIP_TRY(hipMemcpyAsync(out_bytes, a_ob, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b->h_plan.data(), b->d_plan, sizeof(BbtsStreamPlan) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // streams that carried a GSE frame: the whole call of that stream goes through the host parser, synchronisation
    // state taken from and returned to the device
    int rc = 0;
    for (int i = 0; i < n; ++i) {
        if (!b->h_plan[i].needs_host) continue;
        if (!b->host[i]) b->host[i].reset(new BbtsHostParser());
        BbtsHostParser& hp = *b->host[i];
        BbtsDevState ds;
        HIP_TRY(hipMemcpy(&ds, b->d_state + i, sizeof(ds), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hp.partial, b->d_reasm[b->cur] + (size_t)i * REASM_STRIDE, TS, hipMemcpyDeviceToHost));
        hp.synched = ds.synched; hp.count = ds.count;
        memcpy(hp.hdr, ds.hdr, sizeof(ds.hdr));
        b->h_in.resize((size_t)nframes[i] * fbytes);
        b->h_out.resize((size_t)cap);
        HIP_TRY(hipMemcpy(b->h_in.data(), d_bb[i], b->h_in.size(), hipMemcpyDeviceToHost));
        const int got = hp.run(b->h_in.data(), nframes[i], fbytes, b->kbch - 80, b->h_out.data(), cap);
        ds.synched = hp.synched; ds.count = hp.count; ds.last_cnt = hp.last_cnt; ds.last_proc = hp.last_proc;
        memcpy(ds.hdr, hp.hdr, sizeof(ds.hdr));
        HIP_TRY(hipMemcpy(b->d_state + i, &ds, sizeof(ds), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_reasm[b->cur] + (size_t)i * REASM_STRIDE, hp.partial, TS, hipMemcpyHostToDevice));
        if (got < 0) { out_bytes[i] = 0; rc = got; g_err = "output buffer too small for the TS packets of a GSE-carrying call"; continue; }
        out_bytes[i] = got;
        if (got > 0) HIP_TRY(hipMemcpy(d_out[i], b->h_out.data(), got, hipMemcpyHostToDevice));
    }
    return rc;
}

int dvbs2gpu_bbts_work(dvbs2gpu_bbts* b, const uint8_t* h_bb, int cnt, uint8_t* h_ts, int cap) {
    if (!b || b->nstreams != 1 || cnt < 0 || cnt > b->max_frames || cap < 0 || (cnt > 0 && (!h_bb || !h_ts))) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ctx->device));
    const size_t fbytes = b->kbch / 8;
    if (!b->d_in1) {
        HIP_TRY(hipMalloc((void**)&b->d_in1, (size_t)b->max_frames * 8192 + 64));
    }
    HIP_TRY(hipMemcpy(b->d_in1, h_bb, cnt * fbytes, hipMemcpyHostToDevice));
    if (b->out1_cap < (size_t)cap + 64) {
        if (b->d_out1) (void)hipFree(b->d_out1);
        b->d_out1 = nullptr; b->out1_cap = 0;
        HIP_TRY(hipMalloc((void**)&b->d_out1, (size_t)cap + 64));
        b->out1_cap = (size_t)cap + 64;
    }
    uint8_t* d_out = b->d_out1;
    const uint8_t* in_p = b->d_in1;
    int got = 0;
    int rc = dvbs2gpu_bbts_process_batch(b, &in_p, &cnt, &d_out, cap, &got, nullptr);
    if (rc == 0 && got > 0) {
        hipError_t e = hipMemcpy(h_ts, d_out, got, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail_hip(e, "hipMemcpy(bbts out)");
    }
    return rc < 0 ? rc : got;
}

/* h_out15 = {ts_gs, sis_mis, ccm_acm, issyi, npd, ro, isi, upl, dfl, sync, syncd (last_header), last_gse_crc_err, last_bb_cnt,
 * last_bb_proc, last_ts_errs}; h_out15[15..16] = {synched, count} when 17 ints are asked for */
int dvbs2gpu_bbts_get_stats(dvbs2gpu_bbts* b, int stream, int32_t* h_out, int n_out) {
    if (!b || stream < 0 || stream >= b->nstreams || !h_out || n_out < 15) return DVBS2GPU_ERR_ARG;
    HIP_TRY(hipSetDevice(b->ctx->device));
    BbtsDevState ds;
    HIP_TRY(hipMemcpy(&ds, b->d_state + stream, sizeof(ds), hipMemcpyDeviceToHost));
    for (int i = 0; i < 11; ++i) h_out[i] = ds.hdr[i];
    h_out[11] = b->host[stream] ? b->host[stream]->last_gse_crc_err : 0;
    h_out[12] = ds.last_cnt; h_out[13] = ds.last_proc; h_out[14] = 0;
    if (n_out >= 17) { h_out[15] = ds.synched; h_out[16] = ds.count; }
    return 0;
}

}  // extern "C"
