// Device-side data structures + launch prototypes of the DVB-S2 receive chain (front end, PL sync,
// PLL/PLHDR/FED frame loops, soft demapper).  Kernels: s2_rx_kernels.hip; host orchestration: s2_demod.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "launch_count.h"

namespace s2 {

struct cf32 { float re, im; };

constexpr int RRC_MAX_TAPS = 129;
constexpr int GARDNER_PHASES = 128;
constexpr int GARDNER_TAPS = 8;

// Loop state of one stream, device resident.  Field-for-field the state the reference keeps in
// FastAGC / FreqShift / Gardner (+PCL) / FIR / cr_samp / S2PLLBlock::pcl / S2PLHDRDemod::pcl.
struct S2StreamState {
    float agc_gain;
    float nco_phase, nco_freq;
    float g_phase, g_freq;
    int g_offset, g_spsctr;
    cf32 g_hist[GARDNER_TAPS - 1];
    int cr_samp;
    cf32 rrc_hist[RRC_MAX_TAPS - 1];
    float pll_phase, pll_freq;
    float hdr_phase, hdr_freq;
    int n_fe_out;     // outputs of the timing-recovery stage in the last call
    int n_sym;        // symbols appended to the PL-sync FIFO in the last call
    int vcm_synced;   // ACM/VCM framing: locked to a frame start?
    int pl_pending;   // S2PLSyncBlock state 1 (dvbs2_pl_sync.cpp:145-164): offset of the realigned frame inside the window kept at the FIFO head
    float pl_last_bm; // S2PLSyncBlock::best_match of the last correlation
    float nco_agc;    // nco_freq as the NCO of the CURRENT call uses it (the frame loops of a call's early time slices already move nco_freq
                      // while its later AGC / NCO slices run; the reference applies the FED's feedback from the next process() call on)
    // stage pipeline of a call (s2_frontend_launch with post stages): per time slice what the timing recovery / the RRC decimator have
    // produced so far, where the PL-sync walk stands and how many of its frames the frame loops have been through
    int n_fe_slice[32], n_sym_slice[32];    // [S2_FE_MAX_SLICES]
    int walk_cur, walk_nf, loops_done;
    // frame loops ahead of the PL sync (s2_frame_loops_kernel, spec): the walk leaves how many symbols the FIFO holds (walk_avail); the loops
    // may have run the PLL over the first spec_tiles tiles of the window at FIFO offset spec_off that is not a confirmed frame yet -- their
    // state then IS ahead (pll_phase / pll_freq), and spec_phase0 / spec_freq0 hold what it was at that window's start (restored when the
    // window turns out not to be the next frame, or when frame loops that do not know about it come next).  It carries over from call to
    // call: the window's symbols stay in the FIFO, the PLL's output for them in S2StreamWork::spec_out
    int walk_avail, spec_on, spec_off, spec_tiles, spec_carried;     // (spec_carried: tiles of that window done in EARLIER calls -- only those have to come out of spec_out)
    float spec_phase0, spec_freq0;
};

// loop coefficients shared by all streams of one configuration
struct S2LoopCoefs {
    float agc_rate;
    float g_alpha, g_beta, g_min_freq, g_max_freq;    // Gardner PCL: alpha = mu gain, beta = omega gain
    float pll_alpha, pll_beta, pll_min_freq, pll_max_freq;
    float hdr_alpha, hdr_beta, hdr_min_freq, hdr_max_freq;
    float fll_bw;
    int rrc_taps;
    int soft_plsc, pilot_aided;      // extensions (include/dvbs2gpu.h), 0 = the reference's behaviour
    int g_form, g_cand_skew;         // scheduling / tests only: a forced form of the timing recovery (0: by bank size and balance), the candidate form's skew (context options)
    int g_lane_form;                 // scheduling only: a big bank's timing recovery runs in the lane-per-stream form (set by the balancer of s2_demod.hip, with hysteresis)
    int post_prio;                   // scheduling only: 1 = the data-parallel post stages (RRC, PL-sync walk, demapper) run above the decoder's wave priority (the balancer has found the front end critical)
    int g_prio_duty;                 // scheduling only: of every 8 tiles of the timing loop, this many run one wave-priority level up (s2_demod.hip balances the two streams of the pipelined mode with it)
};

// per-call work description of one stream (array in device memory, one entry per stream of the batch)
struct S2StreamWork {
    const cf32* in;          // 2-sps input of this call
    int count;
    cf32* fe_out;            // timing-recovery output (capacity count + 8)
    cf32* fifo;              // PL-sync symbol FIFO
    int fifo_fill;           // symbols already in the FIFO before this call
    S2StreamState* st;
    cf32* fifo_next;         // spare FIFO buffer (receives the unconsumed tail at the end of the call)
    uint8_t* out;            // BBFRAME output of this stream
    cf32* spec_out;          // PLL output of the window the frame loops are ahead of the PL sync in (one PLFRAME; null: no such loops for this stream)
};

// one aligned PLFRAME found by PL sync
struct S2FrameRef {
    const cf32* sym;         // plframe symbols
    int stream;              // index into the batch
    int pad;
};

struct S2FrameStats {        // == dvbs2gpu_frame_stats
    float best_match;
    int detected_modcod, detected_short, detected_pilots;
    float fed_err;
    int ldpc_trials, bch_corr;
    int bbframe_bytes;
};

// tables of one (constellation, gamma) pair
struct S2ConstelDev {
    int constel, bits, states;
    float amp, sca, prescale;
    const int8_t* lut_bits;    // [256][256][bits]   (null for 32APSK)
    const float* lut_err;      // [256][256]
    const uint32_t* lut_bits4; // [256][256]: the cell's `bits` soft values as bytes 0..bits-1 of one word (bits <= 4; null for 32APSK)
    cf32 pts[32];
    const cf32* pts_g;         // the same points in global memory
};

// Mixed CCM batches (streams of DIFFERENT MODCODs in one call): ONE launch per stage serves all configurations -- what the MODCOD-dependent
// kernels (PL-sync walk, frame loops, demapper) otherwise get as kernel arguments comes from a per-stream table in device memory.  A workgroup's
// streams share a configuration (the walk and the small-bank frame loops have one stream per workgroup), so the values stay scalar.
struct S2StreamCfgDev {
    S2ConstelDev con;
    int pls_code, slots, pilots, pilot_blocks, plframe;
    int rate, N;             // demapper
};

// ---- ACM/VCM mode (include/dvbs2gpu.h, acm_vcm): every frame carries its own MODCOD
constexpr int VCM_ACQ_WINDOW = 33282;        // acquisition search span = the longest PLFRAME (QPSK normal with pilots)
constexpr float VCM_MIN_RATIO = 0.5f;        // PLS decodes below this correlation ratio count as "no header here"
constexpr int VCM_DUMMY_PLFRAME = 3330;      // dummy PLFRAME (MODCOD 0): 36 unmodulated slots
struct S2VcmMod {            // what a PLS code (modcod << 2 | short << 1 | pilots) means; device table [128]
    int valid;               // 0 = not a valid code (reserved MODCODs, short 9/10), 1 = data frame, 2 = dummy PLFRAME
    int plframe, slots, pilots, pilot_blocks, bits, rate, constel, N, kb;
    int con;                 // index into the S2ConstelDev array
    int code_index;          // LDPC code
};
struct S2VcmFound {          // a frame the walker found in a stream's FIFO
    int offset;              // FIFO index of its first symbol
    int pls;
    float sofq;              // SOF quality at the frame start
    int pad;
};
struct S2VcmFrame {          // pooled frame of a call (stream-major, in stream order)
    const cf32* sym;
    int stream, pls;
    long long pll_off;       // element offset of this frame's PLL output
    long long llr_off;       // byte offset of this frame's LLRs (frame order)
    float sofq;
    int dst_index;           // index inside its FEC group
};

struct S2PlTablesDev {
    const cf32* sof;           // [26]
    const cf32* plsc;          // [128][64]
    const uint64_t* plsc_code; // [128]
    const uint8_t* rn;         // [131072]
};

// ---------------------------------------------------------------- DVB-S front end (demod::QPSK_ALT, qpsk_alt.cpp)
constexpr int FD_PHASES = 256, FD_TAPS = 256;   // complex_fd.h:33
constexpr int DVBS_SOFT_BLOCK = 8192;           // dvbs_defines.h:3

// Loop state of one DVB-S stream: FastAGC, FLL (+ band-edge FIR delay line), RRC FIR delay line, COMPLEX_FD (+ PCL, delay line),
// Costas<4>, and the soft-bit block FIFO fill of DVBSymToSoftBlock.
struct DvbsStreamState {
    float agc_gain;
    float fll_phase, fll_freq;
    cf32 fll_hist[RRC_MAX_TAPS - 1];
    cf32 rrc_hist[RRC_MAX_TAPS - 1];
    float fd_phase, fd_freq;
    int fd_offset, fd_spsctr;
    cf32 fd_hist[FD_TAPS - 1];
    float costas_phase, costas_freq;
    int n_sym;          // symbols produced by the last call
    int soft_fill;      // soft bits waiting in the block FIFO (after the last call: < 8192)
    int n_blocks;       // whole 8192-soft blocks handed to the Viterbi decoder in the last call
    int n_sym_slice[32]; // symbols of the call after each time slice (dvbs_frontend_launch); the soft-FIFO / Viterbi slices read these
    int vit_done;       // blocks of the call already decoded by earlier slices
};
struct DvbsLoopCoefs {
    float agc_rate;
    float fll_beta, fll_min_freq, fll_max_freq;
    float fd_alpha, fd_beta, fd_min_freq, fd_max_freq;
    float cos_alpha, cos_beta, cos_min_freq, cos_max_freq;
    int ntaps;          // RRC and band-edge filter length
};
struct DvbsStreamWork {
    const cf32* in;     // 2-sps input of this call
    int count;
    cf32* buf_a;        // [count]: AGC output, later RRC output
    cf32* buf_b;        // [count]: FLL output
    cf32* sym;          // [count/2 + 64]: symbols after COMPLEX_FD + Costas
    int8_t* soft;       // soft-bit FIFO (capacity count + 2*8192 + 128)
    DvbsStreamState* st;
};
// AGC -> FLL -> RRC -> COMPLEX_FD + Costas -> soft slicer into the per-stream block FIFO
constexpr int DVBS_FE_MAX_SLICES = 32;      // (n_sym_slice[] of the stream state, the event rows of the stage streams)
struct DvbsSliceHook { virtual hipError_t after_timing(int slice) = 0; virtual ~DvbsSliceHook() {} };   // called after the timing-recovery launch of every slice
hipError_t dvbs_frontend_launch(const DvbsStreamWork* d_work, int nstreams, int max_count, DvbsLoopCoefs coefs, const cf32* d_bandedge,
                                const float* d_rrc, const float* d_fd_bank, hipStream_t st, hipStream_t* aux = nullptr, hipEvent_t (*ev)[DVBS_FE_MAX_SLICES + 1] = nullptr, int nsub = 1,
                                DvbsSliceHook* hook = nullptr, int bank_min = 1 << 30);   // bank_min: carriers from which an unsliced bank uses the many-streams-per-wave kernels
hipError_t dvbs_costas_launch(const DvbsStreamWork* d_work, int nstreams, DvbsLoopCoefs coefs, int sub, int nsub, hipStream_t st);
hipError_t dvbs_soft_slice_launch(const DvbsStreamWork* d_work, int nstreams, int max_count, int sub, int nsub, int* d_blk0, int* d_nblk, hipStream_t st);

hipError_t dvbs_soft_count_launch(const DvbsStreamWork* d_work, int nstreams, int* d_nblocks, hipStream_t st);
hipError_t dvbs_soft_compact_launch(const DvbsStreamWork* d_work, int nstreams, hipStream_t st);

constexpr int S2_FE_MAX_SLICES = 32;     // (n_fe_slice[] / n_sym_slice[] of the stream state; a power of two)
// What runs behind every timing-recovery slice when the whole CCM front half of a call is pipelined (s2_frontend_launch): RRC + decimation of the
// slice's samples into the PL-sync FIFO, the PL-sync walk over the symbols that are in, the frame loops over the frames the walk has found so
// far.  Frame k of stream s lives in SLOT s * maxf + k of d_found / d_pllout / d_stats (the host pools them after the call's one read-back).
struct S2SliceSpans { virtual void begin(int stage, hipStream_t s) = 0; virtual void end(int stage, hipStream_t s) = 0; virtual ~S2SliceSpans() {} };
struct S2PostStages {
    const float* d_taps; int ntaps, max_count;                              // RRC + /2
    int raw, maxf; S2VcmFound* d_found; int* d_counts;                      // PL-sync walk
    S2PlTablesDev tabs; S2ConstelDev con; int pls_code, slots, pilots, pilot_blocks; cf32* d_pllout; S2FrameStats* d_stats;   // frame loops
    S2SliceSpans* spans;                                                    // per-stage timers (optional)
    int loops_launches;                                                     // the frame loops run behind this many of the slices, evenly spaced, the last one included (every
                                                                            // launch costs its longest stream's chain: it only pays with about a frame per stream and launch)
    int spec = 0;                                                           // frame loops ahead of the PL sync (small banks, see S2StreamState)
    const S2StreamCfgDev* cfgs = nullptr;                                   // mixed batch: per-stream configuration (then `raw` is the slot stride = the longest PLFRAME of the batch)
};
hipError_t s2_post_stages_launch(const S2StreamWork* d_work, int nstreams, const S2LoopCoefs& coefs, const S2PostStages& p, int c, int nsub, hipStream_t s);
hipError_t s2_frontend_launch(const S2StreamWork* d_work, int nstreams, S2LoopCoefs coefs, const float* d_bank, hipStream_t st, hipStream_t aux,
                              hipEvent_t* ev, int nsub, const S2PostStages* post = nullptr, hipEvent_t* ev2 = nullptr, hipStream_t post_stream = nullptr,
                              hipStream_t loops_stream = nullptr, hipEvent_t* ev3 = nullptr);
hipError_t s2_rrc_decim_launch(const S2StreamWork* d_work, int nstreams, int max_count, const float* d_taps, int ntaps, hipStream_t st, int post_prio = 0);
// per-stream frame loops: frames of stream s are d_frames[first[s] .. first[s+1])
hipError_t s2_frame_loops_launch(const S2StreamWork* d_work, int nstreams, const S2FrameRef* d_frames, const int* d_first,
                                 S2LoopCoefs coefs, S2PlTablesDev tabs, S2ConstelDev con, int pls_code, int slots, int pilots,
                                 int pilot_blocks, int plframe, cf32* d_pllout, S2FrameStats* d_stats, hipStream_t st);
// batched tails of a call: per-stream symbol counts / NCO frequency in contiguous arrays, BBFRAME scatter, FIFO compaction
hipError_t s2_collect_launch(const S2StreamWork* d_work, int nstreams, int* d_nsym, float* d_nco, hipStream_t st);
hipError_t s2_scatter_out_launch(const S2StreamWork* d_work, const S2FrameRef* d_frames, const int* d_first, int nframes, int kb,
                                 const uint8_t* d_bb, hipStream_t st);
hipError_t s2_scatter_out2_launch(uint8_t* const* d_outs, const S2FrameRef* d_frames, const int* d_first, int nframes, int kb,
                                  const uint8_t* d_bb, hipStream_t st);
hipError_t s2_fifo_compact_launch(const S2StreamWork* d_work, int nstreams, const int* d_cur_fill /*[2*nstreams]: cur, fill*/, hipStream_t st);
// PL sync with its 2-state realign machine on the device: one workgroup per stream walks the stream's complete windows in order
hipError_t s2_ccm_walk_launch(const S2StreamWork* d_work, int nstreams, int raw, int maxf, S2VcmFound* d_found, int* d_counts, hipStream_t st, int post_prio = 0);
// ACM/VCM path
hipError_t s2_vcm_walk_launch(const S2StreamWork* d_work, int nstreams, S2PlTablesDev tabs, const S2VcmMod* d_mods, float sof_threshold, int maxf,
                              S2VcmFound* d_found, int* d_counts /*[nstreams][4]: frames, consumed, avail, new symbols*/, hipStream_t st);
hipError_t s2_vcm_loops_launch(const S2StreamWork* d_work, int nstreams, const S2VcmFrame* d_frames, const int* d_first, S2LoopCoefs coefs,
                               S2PlTablesDev tabs, const S2VcmMod* d_mods, const S2ConstelDev* d_cons, cf32* d_pllout, S2FrameStats* d_stats,
                               hipStream_t st);
hipError_t s2_vcm_demap_launch(const S2VcmFrame* d_frames, int nframes, const S2VcmMod* d_mods, const S2ConstelDev* d_cons, const cf32* d_pllout,
                               int8_t* d_llr, hipStream_t st);
hipError_t s2_vcm_gather_launch(const S2VcmFrame* d_frames, const int* d_idx, int count, int N, const int8_t* d_llr, int8_t* d_grp, hipStream_t st);
hipError_t s2_vcm_scatter_launch(const int* d_idx, int count, int kb, const uint8_t* d_bb, uint8_t* const* d_dst, hipStream_t st);
hipError_t s2_deinterleave_launch(int constel, int rate, int bits, int N, const int8_t* d_in, int nframes, int8_t* d_out, hipStream_t st);
hipError_t math_eval_launch(int func, int n, const float* a, const float* b, float* o0, float* o1, hipStream_t st);
hipError_t s2_demap_launch(S2ConstelDev con, int rate, int shortframe, int slots, int pilots, int plframe, const cf32* d_pllout,
                           int nframes, int8_t* d_llr, int N, hipStream_t st, const int* d_slot = nullptr, int post_prio = 0);   // d_slot: frame f's symbols lie in slot d_slot[f] of d_pllout
// mixed batch: frame f belongs to stream d_slot[f] / maxf (its configuration: cfgs), its symbols lie in slot d_slot[f] (stride slot_stride), its LLRs go to d_llr_of[f]
hipError_t s2_demap_mixed_launch(const S2StreamCfgDev* cfgs, int max_slots, int maxf, int slot_stride, const cf32* d_pllout, int nframes,
                                 int8_t* const* d_llr_of, hipStream_t st, const int* d_slot);

}  // namespace s2
