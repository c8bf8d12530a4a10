/*
 * dvbs2gpu -- C ABI of the MI355X (gfx950) DVB-S / DVB-S2 demodulation + FEC engine.
 *
 * This is the drop-in boundary for the hot path of cropinghigh/sdrpp-dvbs-demodulator: the SDR++ plugin
 * shell (src/main.cpp) stays host C++; the bodies of
 *     int DVBS2Demod::process(int count, const complex_t* in, uint8_t* out)   src/demod/dvbs2/module_dvbs2_demod.cpp:216
 *     int DVBSDemod::process(int count, const complex_t* in, uint8_t* out)    src/demod/dvbs/module_dvbs_demod.cpp:78
 * and every stage they call are replaced by the entry points below.  Plain C types only, no exceptions
 * cross this boundary (the reference throws std::runtime_error for a bad MODCOD, modcod_to_cfg.cpp:11,135;
 * here that is DVBS2GPU_ERR_MODCOD).
 *
 * Pointer convention: arguments named d_* are DEVICE pointers (hipMalloc / torch CUDA tensors); h_* are
 * host pointers.  `stream` is a hipStream_t passed as void* (NULL = default stream).  Batch entry points
 * enqueue work on the stream and return without synchronising unless stated otherwise.
 *
 * Every function returns 0 (or a non-negative count) on success and a negative DVBS2GPU_ERR_* code on error.
 */
#ifndef DVBS2GPU_H
#define DVBS2GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVBS2GPU_OK 0
#define DVBS2GPU_ERR_ARG (-1)      /* null pointer / bad size */
#define DVBS2GPU_ERR_MODCOD (-2)   /* MODCOD <= 0, >= 29, or short-frame 9/10 (does not exist) */
#define DVBS2GPU_ERR_HIP (-3)      /* a HIP call failed; dvbs2gpu_last_error() has the text */
#define DVBS2GPU_ERR_NODEVICE (-4) /* no gfx950 device visible: the engine has no CPU fallback */
#define DVBS2GPU_ERR_CAPACITY (-5) /* output buffer too small */

/* code rate index used throughout: 0=1/4 1=1/3 2=2/5 3=1/2 4=3/5 5=2/3 6=3/4 7=4/5 8=5/6 9=8/9 10=9/10 */

typedef struct dvbs2gpu_ctx dvbs2gpu_ctx;   /* one per device / host thread; owns tables + workspaces */

const char* dvbs2gpu_version(void);
const char* dvbs2gpu_last_error(void);
/* Optional, for hosts that want the engine's HIP streams on hardware queues of their own: sets GPU_MAX_HW_QUEUES=12 in the process environment unless the
 * variable is already there.  The HIP runtime reads it at the process's first HIP call, so call this BEFORE that call and before starting threads (setenv is
 * not thread-safe); loading the library changes nothing by itself.  Returns 1 = set, 0 = already present.  No reference counterpart. */
int dvbs2gpu_preinit(void);

/* Create a context on HIP device `device`.  Fails with DVBS2GPU_ERR_NODEVICE when no GPU is present. */
int dvbs2gpu_create(int device, dvbs2gpu_ctx** out);
/* HIP devices visible to the process (0: none -- there is no CPU fallback); what a multi-GPU host sizes its fleet with */
int dvbs2gpu_device_count(void);
void dvbs2gpu_destroy(dvbs2gpu_ctx* ctx);
/* Development / test options of a context (which of several bit-identical flows runs, time slicing, side streams ...; DESIGN.md section 11 lists them).  The same
 * pairs can be given as DVBS2GPU_OPTIONS="name=value,name=value" in the environment when the context is created -- the one variable the library reads.  Options that
 * choose a decoder plan (ldpc_wave, ldpc_split) must be set before the first frame of the code is decoded.  The reference has no counterpart (its behaviour is the default). */
int dvbs2gpu_set_option(dvbs2gpu_ctx* ctx, const char* name, int value);
/* Read-only introspection (no reference counterpart; bench.py records the run-time balancer's final state and the launches a drop-in call costs with it).
 * Names: "kernel_launches" (kernel launches of the whole library in this process so far), "g_prio_duty" / "g_prio_auto" / "g_prio_hold" (the pipelined mode's priority
 * share of the timing loop, whether it is balanced at run time, calls the balancer still rests), "stage_pipeline_on" (the last CCM batch ran its post stages behind
 * every front-end slice), "fec_part_on" / "fec_part" (big decoder jobs on the 128-unit partition stream; the option's setting), "pipelined", "num_cus". */
int dvbs2gpu_get_state(dvbs2gpu_ctx* ctx, const char* name, long long* value);
/* Test / bench aid: the pipelined CCM decoder job that configuration group `slot` (0 for a single-configuration batch) DELIVERED last -- what the decoder read and what it
 * wrote, still in place until the group starts another job of the same parity (two calls later).  out10 = {device pointer of the LLRs [nf][N] int8, device pointer of the
 * BBFRAMEs [nf][kb], nf frames, n streams, N, kb bytes per BBFRAME, code rate index, short frames, max_trials, forced}; h_first[n + 1]: first pooled frame of every stream
 * of the job; h_handles[n]: the streams' dvbs2gpu_demod handles (as the call that started the job listed them).  bench.py decodes sampled LLR frames with the CPU oracle
 * and compares them with the BBFRAMEs the engine delivered.  No reference counterpart. */
int dvbs2gpu_debug_last_fec_job(dvbs2gpu_ctx* ctx, int slot, long long* out10, int32_t* h_first, const void** h_handles, int cap);

/* Static parameter queries (no GPU needed).  Mirrors get_dvbs2_cfg (modcod_to_cfg.cpp:5-140),
 * BBFrameBCH::BBFrameBCH (bbframe_bch.cpp:39-161) and the PLFRAME size of dvbs2_pl_sync.cpp:14-31. */
typedef struct dvbs2gpu_modcod_info {
    int32_t constellation;   /* 0 QPSK, 1 8PSK, 2 16APSK, 3 32APSK */
    int32_t bits_per_symbol;
    int32_t rate;            /* code rate index */
    int32_t slots;           /* 90-symbol payload slots */
    int32_t pilot_blocks;
    int32_t plframe_symbols; /* 90 + 90*slots + 36*pilot_blocks */
    int32_t ldpc_n, ldpc_k;  /* codeword / information bits (ldpc_k = nbch) */
    int32_t kbch;            /* BBFRAME bits; output is kbch/8 bytes per frame (DVBS2Demod::getKBCH) */
    int32_t bch_t;
    int32_t ldpc_edges;      /* Tanner-graph edges (LINKS_TOTAL) */
    float g1, g2;
} dvbs2gpu_modcod_info;
int dvbs2gpu_modcod_info_get(int modcod, int shortframes, int pilots, dvbs2gpu_modcod_info* out);
int dvbs2gpu_fec_info_get(int rate, int shortframes, dvbs2gpu_modcod_info* out);

/* ------------------------------------------------------------------ FEC stages (DVB-S2)
 *
 * dvbs2gpu_ldpc_decode_batch  replaces  BBFrameLDPC::decode  (bbframe_ldpc.cpp:123-139), one call per
 * frame in the reference, here `nframes` frames per launch.
 *   d_llr     [nframes][N] int8, bit 1 <-> negative (module_dvbs2_demod.cpp:360)
 *   d_hard    [nframes][K/8] hard decisions of the K information bits, MSB first (the repack loop of
 *             module_dvbs2_demod.cpp:357-360), may be NULL
 *   d_post    [nframes][N] int8 posteriors in the reference's layout, may be NULL
 *   d_trials  [nframes] int32: iterations used, or -1 if not converged after max_trials
 *   force != 0: benchmark mode, no early exit, exactly max_trials iterations; trials = max_trials or -1.
 */
int dvbs2gpu_ldpc_decode_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, const int8_t* d_llr, int nframes,
                               int max_trials, int force, uint8_t* d_hard, int8_t* d_post, int32_t* d_trials, void* stream);

/* Introspection of the decoder plan of one code (for DESIGN.md / bench reporting):
 * out8 = {layers q, max row degree, message record dwords, sum of layer depths, resident workgroups per CU (both of the
 *         decoder that serves the code),
 *         CUs, Tanner edges, layers with intra-layer shared bits}. */
int dvbs2gpu_ldpc_plan_info(dvbs2gpu_ctx* ctx, int rate, int shortframes, int32_t* out8);

/* Which of the three LDPC decoder kernels serves the code under the context's options (bench / profile reporting):
 * 0 lane per row (ldpc_kernel.hip), 1 wave per frame (short frames, ldpc_wave_kernel.hip), 2 half a row per lane
 * (ldpc_split_kernel.hip).  All three restate layered_decoder.hh:46-74 and are bit-exact with each other. */
int dvbs2gpu_ldpc_decoder_form(dvbs2gpu_ctx* ctx, int rate, int shortframes);

/* Host-only dump of the decoder plan (no GPU needed; used by the CPU test-suite to check the intra-layer
 * ordering against the reference's sequential row order).  Call with NULL arrays to get the counts:
 * counts3 = {layers, link entries, per-row words}.  layers4: 4 uint32 per layer {first entry, degree,
 * depth | nc<<16, first row word}; ents: sp | r<<16; rows: level | late<<8 | early<<20 (ldpc_plan.h). */
int dvbs2gpu_ldpc_plan_dump(int rate, int shortframes, uint32_t* layers4, uint32_t* ents, uint32_t* rows, int32_t* counts3);
/* The same for the wave-per-frame decoder's plan (csrc/ldpc_wave_plan.h) and for the lane-per-row decoder's address table
 * (csrc/ldpc_plan.h; regular codes of degree 2, 8, 12, else empty).  NULL arrays: counts only.
 * counts6 = {link slots per lane LW, smallest row degree incl. parity links, steps per sweep, index of the absent masks inside
 * lanec, words in lanec, entries in steps}; lanec: [q][8][LW] thr | cA << 16, then [q][8] absent masks; steps: [..][8] row ids
 * (360 * layer + j, 0xffff = empty slot); layer_end: [q] chunk (4 steps) index at which a layer's steps end.
 * counts2 = {words in the table, words per row}; table: [q][384][words per row], two 16-bit byte offsets per word. */
int dvbs2gpu_ldpc_wave_plan_dump(int rate, int shortframes, uint32_t* lanec, uint16_t* steps, uint32_t* layer_end, int32_t* counts6);
int dvbs2gpu_ldpc_addr_table_dump(int rate, int shortframes, uint32_t* table, int32_t* counts2);
/* The half-row decoder's plan (csrc/ldpc_split_plan.h; codes it does not take: counts6[0] = 0, return value 0, and dvbs2gpu_last_error() says why -- irregular rows, a layer
 * with more than four shared links, ...).  A plan is listed for every code the PLAN takes; the decoder serves the normal frames of rates 1/4, 2/5, 1/2, 3/5, 2/3, 3/4
 * (dvbs2gpu_ldpc_decoder_form).
 * counts6 = {pseudo-layers, table words per thread, slots per row half, message-workspace dwords per workgroup, words in the table, record dwords};
 * layers4: 4 uint32 per pseudo-layer {kind | waves << 8 | flags, kind 1: chain step | steps << 16, kind 8: levels << 16, record offset, kind 1: first link entry, kind 8: word offset
 * of the layer's side entries in `table`}; table: [pseudo-layer][768][words] (two 16-bit LDS byte offsets per word, the row word behind the last slot), then the side entries of the
 * kind-8 layers [384 rows][2 words] (per shared slot: distance from the row's output cell back to the slot's source cell in bits 0..10, 0 = the bit itself; bit 15: a later row
 * of the layer touches the slot) -- `words in the table` counts both; row_of: [pseudo-layer][384] the row a lane pair updates (-1: idle);
 * layer_of: [pseudo-layer] its layer.  NULL arrays: counts only. */
int dvbs2gpu_ldpc_split_plan_dump(int rate, int shortframes, uint32_t* layers4, uint32_t* table, int32_t* row_of, int32_t* layer_of, int32_t* counts6);

/* replaces BBFrameBCH::decode (bbframe_bch.cpp:380-405).  d_frames [nframes][K/8] corrected in place;
 * d_corrections [nframes] int32: #bits corrected, 0 clean, -1 uncorrectable (frame left untouched). */
int dvbs2gpu_bch_decode_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, uint8_t* d_frames, int nframes,
                              int32_t* d_corrections, void* stream);

/* replaces BBFrameDescrambler::work + the copy of module_dvbs2_demod.cpp:364-366.
 * d_frames [nframes][K/8] -> d_out [nframes][kbch/8] */
int dvbs2gpu_bb_descramble_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, const uint8_t* d_frames, int nframes,
                                 uint8_t* d_out, void* stream);

/* LDPC -> repack -> BCH -> descramble for a batch (module_dvbs2_demod.cpp:349-366, with every frame
 * LDPC-decoded -- the reference only decodes SIMD lane 0).
 * d_bbframes [nframes][kbch/8]; d_trials, d_corrections [nframes] (either may be NULL). */
int dvbs2gpu_fec_decode_batch(dvbs2gpu_ctx* ctx, int rate, int shortframes, const int8_t* d_llr, int nframes,
                              int max_trials, int force, uint8_t* d_bbframes, int32_t* d_trials, int32_t* d_corrections,
                              void* stream);

/* ------------------------------------------------------------------ soft demap + bit de-interleave
 * replaces S2BBToSoft::process (dvbs2_bb_to_soft.cpp:7-33): constellation_t::demod_soft_lut per payload
 * symbol (constellation.cpp:293-322) followed by S2Deinterleaver::deinterleave (s2_deinterleaver.cpp:72-136).
 *   d_frames  [nframes][plframe_symbols] complex64 PLL output (header in [0,90), payload after it, pilot
 *             blocks at their standard positions when pilots != 0)
 *   d_llr     [nframes][N] int8 */
int dvbs2gpu_demap_batch(dvbs2gpu_ctx* ctx, int modcod, int shortframes, int pilots, const float* d_frames, int nframes,
                         int8_t* d_llr, void* stream);

/* S2Deinterleaver::deinterleave alone (s2_deinterleaver.cpp:72-136) on caller-supplied LLR frames, d_in / d_out [nframes][N]
 * int8, out of place: the index function of the fused demapper above as a stage of its own (the parity tests push the
 * reference's golden index ramps through it). */
int dvbs2gpu_deinterleave_batch(dvbs2gpu_ctx* ctx, int modcod, int shortframes, const int8_t* d_in, int nframes, int8_t* d_out,
                                void* stream);

/* ------------------------------------------------------------------ shared math definitions, evaluated on the device
 * The engine's float stages take sin/cos/atan2/exp/log from include/dvbs2gpu_math.h (where the reference calls libm:
 * freq_shift.cpp:6, dvbs2_pll.cpp:39,50-75, dvbs2_plhdr_demod.cpp:35, fll.cpp:137, constellation.cpp:226,250,259).  This
 * entry point evaluates those definitions element-wise on the GPU so that a test can compare them bit for bit with the
 * host evaluation of the same header.  func: 0 sincos(a) -> out0 = sin, out1 = cos; 1 atan2(a, b) -> out0; 2 exp(a);
 * 3 log(a); 4 the LLR clamp of constellation.cpp:263-270 (as float); 5 the LUT cell of the sample (a, b) = re_index * 256 + im_index
 * (constellation.cpp:295-301; the kernels' binary32 fast path with the double form as its guard).  Device pointers, n elements. */
int dvbs2gpu_math_eval(dvbs2gpu_ctx* ctx, int func, int n, const float* d_a, const float* d_b, float* d_out0, float* d_out1,
                       void* stream);

/* ------------------------------------------------------------------ per-stage device times
 * The reference exposes its health through the stats fields the GUI polls (module_dvbs2_demod.h:82-87); a batched GPU engine also
 * needs to show where a call's time goes.  With timing on, every stage of dvbs2gpu_demod_process[_batch] (and the FEC stage entry
 * points) is bracketed by a hipEvent pair on the stream it is enqueued on -- also in the pipelined mode, where the FEC of call k
 * runs beside the front end of call k+1, so ms[LDPC] is the decoder's time AS IT RAN inside the calls.  get_stage_times waits for
 * the outstanding events, returns the sums since the previous get and resets them.  units: frames (LDPC, BCH) or 0. */
#define DVBS2GPU_STAGE_FRONTEND 0  /* FastAGC + FreqShift + Gardner */
#define DVBS2GPU_STAGE_RRC 1       /* RRC FIR + /2 */
#define DVBS2GPU_STAGE_PLSYNC 2
#define DVBS2GPU_STAGE_LOOPS 3     /* coarse FED, PLL, PLHDR demod */
#define DVBS2GPU_STAGE_DEMAP 4     /* soft demap + bit de-interleave */
#define DVBS2GPU_STAGE_LDPC 5
#define DVBS2GPU_STAGE_BCH 6       /* BCH + BB descramble */
#define DVBS2GPU_STAGE_DELIVER 7   /* BBFRAMEs into the caller's buffers */
#define DVBS2GPU_STAGE_COUNT 8
typedef struct dvbs2gpu_stage_times {
    double ms[DVBS2GPU_STAGE_COUNT];
    int64_t launches[DVBS2GPU_STAGE_COUNT];
    int64_t units[DVBS2GPU_STAGE_COUNT];
} dvbs2gpu_stage_times;
int dvbs2gpu_set_stage_timing(dvbs2gpu_ctx* ctx, int on);
int dvbs2gpu_get_stage_times(dvbs2gpu_ctx* ctx, dvbs2gpu_stage_times* out);

/* ------------------------------------------------------------------ full DVB-S2 demodulator (one stream)
 *
 * Mirror of dsp::dvbs2::DVBS2Demod (module_dvbs2_demod.h:51-160).  A handle is one transponder stream:
 * it owns the loop state that the reference keeps in its member blocks (AGC gain, NCO phase/frequency,
 * Gardner delay line + PCL, RRC delay line, /2 decimator phase, PL-sync buffer and state machine, PLL and
 * PLHDR loops).  Not re-entrant per handle (the reference guards process() with ctrlMtx the same way).
 *
 * cfg mirrors the arguments of DVBS2Demod::init (module_dvbs2_demod.cpp:7-30) in the same units. */
typedef struct dvbs2gpu_demod_cfg {
    double symbolrate, samplerate;       /* only their ratio matters (RRC taps); the plugin uses samplerate = 2*symbolrate */
    float agc_rate, rrc_alpha;
    int32_t rrc_taps;
    float loop_bw, fll_bw;
    float clock_omega_gain, clock_mu_gain, omega_rel_limit;   /* gains, agc_rate and the bandwidths must be finite (DVBS2GPU_ERR_ARG otherwise):
                                          * the timing-recovery kernels evaluate PCL::advance(0) as "frequency unchanged" */
    int32_t modcod, shortframes, pilots;
    float sof_threshold;                 /* stored, unused in CCM mode -- as in the reference (dvbs2_pl_sync.cpp:140-142); ACM/VCM mode:
                                          * minimum SOF quality of a frame start */
    int32_t max_ldpc_trials;
    int32_t force_ldpc_iters;            /* 0 = normal early exit; n > 0 = benchmark mode, exactly n iterations */
    /* Extensions beyond the reference's behaviour, all 0 by default (= reference-compatible, parity-tested against it):
     * acm_vcm      SURVEY 8(f) rank 3.  The PL framing follows the PLS code of EVERY frame (soft RM(64,7) decode at the frame start
     *              names MODCOD, frame size and pilots, hence the frame's length and its demapper / LDPC / BCH code); modcod /
     *              shortframes / pilots above are ignored.  BBFRAMEs then differ in size: dvbs2gpu_frame_stats.bbframe_bytes.  The
     *              reference only reports what it detected (dvbs2_plhdr_demod.cpp:43-64) and lets the GUI re-configure the whole
     *              demodulator after 50 consistent sightings (main.cpp:375-408).
     * soft_plsc    SURVEY 8(f) rank 4.  The PLHDR demodulator decodes the PLS code by soft correlation over all 64 code bits instead
     *              of the reference's hard decisions compared on 60 bits (dvbs2_plhdr_demod.cpp:45-58,69-79).
     * pilot_aided  SURVEY 8(f) rank 4.  The known symbols (PL header, pilot blocks) give the PLL a block phase estimate that is
     *              applied at the end of each block, on top of the reference's decision-directed loop (dvbs2_pll.cpp:34-86): no
     *              rotational false locks.  Pilot symbols then also drive the loop with the data-aided error phase(descrambled x
     *              conj((1+j)/sqrt2)) at full gain; with 0 they use the reference's error, decision-directed on the sign-sliced
     *              QPSK point and divided by 10 (dvbs2_pll.cpp:58). */
    int32_t acm_vcm, soft_plsc, pilot_aided;
} dvbs2gpu_demod_cfg;

typedef struct dvbs2gpu_demod dvbs2gpu_demod;

/* Defaults of the plugin shell (main.cpp:64-73,134-140). */
void dvbs2gpu_demod_default_cfg(int modcod, int shortframes, int pilots, dvbs2gpu_demod_cfg* out);

/* max_samples: largest `count` a single process call may pass (SDR++ STREAM_BUFFER_SIZE = 1000000). */
int dvbs2gpu_demod_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_demod_cfg* cfg, int max_samples, dvbs2gpu_demod** out);
void dvbs2gpu_demod_destroy(dvbs2gpu_demod* d);
int dvbs2gpu_demod_reset(dvbs2gpu_demod* d);                                            /* DVBS2Demod::reset */
int dvbs2gpu_demod_set_params(dvbs2gpu_demod* d, int modcod, int shortframes, int pilots, float sof_threshold,
                              int max_ldpc_trials);                                      /* DVBS2Demod::setDemodParams */
int dvbs2gpu_demod_get_kbch(dvbs2gpu_demod* d);                                          /* DVBS2Demod::getKBCH */

/* int DVBS2Demod::process(int count, const complex_t* in, uint8_t* out): h_iq = count interleaved (re,im)
 * float pairs at 2 samples/symbol, h_out receives the descrambled BBFRAMEs completed by this call
 * (kbch/8 bytes each, input order).  Returns bytes written (0 is legal) or a negative error.  Synchronous. */
int dvbs2gpu_demod_process(dvbs2gpu_demod* d, int count, const float* h_iq, uint8_t* h_out, int out_cap);

/* Same for `n` independent streams in one go (frames of all streams share the FEC launches).  d_iq[i] are
 * DEVICE pointers (inputs already resident in HBM), counts[i] samples each; d_out[i] DEVICE buffers of
 * out_cap bytes; out_bytes[i] (host) receives the byte count of stream i.  Synchronous. */
int dvbs2gpu_demod_process_batch(dvbs2gpu_demod* const* demods, int n, const float* const* d_iq, const int* counts,
                                 uint8_t* const* d_out, int out_cap, int* out_bytes);

/* Throughput mode for dvbs2gpu_demod_process_batch: with `on` != 0 the FEC (LDPC, BCH, descrambler) of call k runs on its own
 * HIP stream while call k+1 runs the front end, PL sync and frame loops of the next samples; the BBFRAMEs (and stats) of call
 * k are delivered by call k+1 into ITS output buffers (the reference delivers frames late as well: it holds them until 16 have
 * queued, module_dvbs2_demod.cpp:343-347).  The streams of a batch may change from call to call (transponders come and go, in any
 * order): a job is collected into the buffers of those of ITS streams that are part of the collecting call; a stream that joins has
 * nothing pending; the frames of a stream that is absent from the call after its own are dropped (pass it with count 0 once more
 * to collect them).  Streams of different configurations may share the batch (one FEC job per configuration group and call, at
 * most 16 groups); ACM/VCM streams too (one job per LDPC code present in the call, BBFRAMEs of differing size: per-frame sizes in
 * dvbs2gpu_frame_stats.bbframe_bytes).  A call with all counts 0 collects the last frames; switching the mode off drops
 * uncollected ones.
 * INPUT AND OUTPUT BUFFERS: as in the synchronous mode, device work the host has put on the legacy null stream before the call -- the copies or kernels that fill d_iq --
 * lies in front of the demodulator (the mode's own stream waits for it), and the buffers may be reused when the call returns; a host that fills them on a stream of
 * its own synchronises that stream before the call.  The delivered BBFRAMEs are complete when the call returns.  Switching the mode ON synchronises the device and
 * gives the context's internal streams back to the runtime (created again on demand). */
int dvbs2gpu_set_pipelined(dvbs2gpu_ctx* ctx, int on);

/* Stats of the frames completed by the last process call of this handle (the public fields the GUI polls,
 * module_dvbs2_demod.h:82-87, one record per frame). */
typedef struct dvbs2gpu_frame_stats {
    float pl_sync_best_match;
    int32_t detected_modcod, detected_shortframes, detected_pilots;
    float coarse_freq_err;
    int32_t ldpc_trials, bch_corrections;
    int32_t bbframe_bytes;               /* size of this frame's BBFRAME in the output (kbch/8; ACM/VCM: per frame, 0 for a dummy PLFRAME) */
} dvbs2gpu_frame_stats;
int dvbs2gpu_demod_get_stats(dvbs2gpu_demod* d, dvbs2gpu_frame_stats* h_out, int cap);
float dvbs2gpu_demod_get_nco_freq(dvbs2gpu_demod* d);
/* Where the frames FOUND by the last process call start: index of each frame's first symbol in the stream's symbol sequence
 * (after timing recovery, one per symbol) since the last reset.  Returns the frame count.  (In the synchronous mode these are the
 * frames the call delivered; in the throughput mode the ones the NEXT call will deliver.)  No reference counterpart: the segment
 * receiver below orders and de-duplicates frames with it. */
int dvbs2gpu_demod_get_frame_positions(dvbs2gpu_demod* d, int64_t* h_out, int cap);

/* Debug taps of the last process call (device->host copies; for parity tests and the constellation
 * display callback d_handler, module_dvbs2_demod.cpp:337).  which: 0 = 1-sps symbols entering PL sync,
 * 1 = aligned raw PLFRAMEs, 2 = PLL output, (complex64, count in complex samples); 3 = LLRs (int8).
 * Returns the element count; copies at most cap elements when h_dst != NULL. */
int dvbs2gpu_demod_get_tap(dvbs2gpu_demod* d, int which, void* h_dst, int cap);

/* ------------------------------------------------------------------ fleet: the transponders of one host over several GPUs
 * The reference runs one independent DVBS2Demod instance per transponder (src/main.cpp:588,595: every plugin instance owns its demodulator and worker thread); nothing is
 * exchanged inside a frame or between streams.  A FLEET is what a C++ plugin host with several GPUs calls: `n` members -- one engine context + one worker thread per entry
 * of devices[] (a HIP device index may appear more than once: "logical devices", how the tests run a 4-member fleet on a 1-GPU box) -- and a transponder table placed on them
 * by dvbs2gpu_fleet_plan's rule: whole MODCOD groups by longest-processing-time first (every member's LDPC batches stay homogeneous) when no member ends up more than
 * `tolerance` above the average load, else the MODCOD-sorted list cut into n pieces of near-equal weight -- the rule of the multi-process harness
 * (sdrpp-dvbs-demodulator_amd/distribute.py: assign_transponders; tests/test_cabi_host.py holds the two against each other).  No device talks to another.
 *   dvbs2gpu_fleet_plan           the placement alone (no GPU): member_of[i] for nt transponders of given MODCOD and weight over `world` members.
 *   dvbs2gpu_fleet_assign         places `table` (a demodulator configuration, a weight -- <= 0: edges x iterations + 40 x PLFRAME symbols -- and the largest call in samples per
 *                                 transponder), creates every transponder's demodulator on its member (an earlier table's are destroyed), out_cap = bytes of the largest
 *                                 delivery of one transponder and call; member_of (optional) receives the placement.
 *   dvbs2gpu_fleet_process_batch  one call for ALL transponders: h_iq[i] / counts[i] HOST samples of transponder i (2 sps, interleaved floats), h_out[i] HOST buffers of out_cap
 *                                 bytes, out_bytes[i] the bytes delivered -- EGRESS IN TABLE ORDER whichever member decoded them.  The members copy, run
 *                                 dvbs2gpu_demod_process_batch and copy back side by side on their worker threads; the call returns when all are through; the first member
 *                                 error is the return code (dvbs2gpu_last_error() names the device), every member is waited for either way.
 *   dvbs2gpu_fleet_set_pipelined  the throughput mode of dvbs2gpu_set_pipelined on every member (frames one call late; a call with all counts 0 collects the last ones).
 *   dvbs2gpu_fleet_get_stats      dvbs2gpu_demod_get_stats of transponder i.      dvbs2gpu_fleet_reset: DVBS2Demod::reset of every transponder. */
typedef struct dvbs2gpu_fleet dvbs2gpu_fleet;
typedef struct dvbs2gpu_fleet_entry {
    dvbs2gpu_demod_cfg cfg;
    double weight;
    int32_t max_samples;
    int32_t reserved;
} dvbs2gpu_fleet_entry;
int dvbs2gpu_fleet_plan(const int32_t* modcods, const double* weights, int nt, int world, double tolerance, int32_t* member_of);
int dvbs2gpu_fleet_create(const int* devices, int n, dvbs2gpu_fleet** out);
void dvbs2gpu_fleet_destroy(dvbs2gpu_fleet* f);
int dvbs2gpu_fleet_size(const dvbs2gpu_fleet* f);
int dvbs2gpu_fleet_assign(dvbs2gpu_fleet* f, const dvbs2gpu_fleet_entry* table, int nt, int out_cap, double tolerance, int32_t* member_of);
int dvbs2gpu_fleet_set_pipelined(dvbs2gpu_fleet* f, int on);
int dvbs2gpu_fleet_reset(dvbs2gpu_fleet* f);
int dvbs2gpu_fleet_process_batch(dvbs2gpu_fleet* f, const float* const* h_iq, const int* counts, uint8_t* const* h_out, int out_cap, int* out_bytes);
int dvbs2gpu_fleet_get_stats(dvbs2gpu_fleet* f, int transponder, dvbs2gpu_frame_stats* h_out, int cap);

/* ------------------------------------------------------------------ segment receiver: ONE fast transponder
 * A stream's loops are serial recurrences (one stream: 0.69 Msym/s on MI355X), so a single 27.5 Msym/s transponder cannot be
 * followed sample by sample.  The segment receiver cuts a long chunk of one continuous IQ stream into `nsegments` overlapping
 * segments of `own_frames` PLFRAMEs each (+ `warm_frames` of warm-up in front, own_frames >= warm_frames), runs them as
 * independent streams of one dvbs2gpu_demod_process_batch call with freshly reset loops, orders the frames by the positions of
 * dvbs2gpu_demod_get_frame_positions, drops duplicates where neighbours overlap and returns the BBFRAMEs in stream order; the
 * tail of a chunk is the warm-up of the next call's first segment, so calls join without a gap.  No counterpart in the reference
 * (one DVBS2Demod per transponder, serial): it returns the same BBFRAMEs wherever the reference would decode them, but it is a
 * high-latency mode (a chunk is nsegments * own_frames frames long) and soft values are not bit-identical to a serial run.
 * d_iq: DEVICE pointer to `count` complex samples (interleaved floats) continuing the stream, count <= dvbs2gpu_segrx_chunk_samples;
 * d_out: DEVICE buffer; returns bytes written (kbch/8 per frame) or a negative error.  The context must be in synchronous mode. */
typedef struct dvbs2gpu_segrx dvbs2gpu_segrx;
int dvbs2gpu_segrx_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_demod_cfg* cfg, int nsegments, int own_frames, int warm_frames, dvbs2gpu_segrx** out);
int dvbs2gpu_segrx_reset(dvbs2gpu_segrx* r);
void dvbs2gpu_segrx_destroy(dvbs2gpu_segrx* r);
long long dvbs2gpu_segrx_chunk_samples(dvbs2gpu_segrx* r);
int dvbs2gpu_segrx_process(dvbs2gpu_segrx* r, const float* d_iq, long long count, uint8_t* d_out, long long out_cap);
/* h_out3 = {frame sightings of the last call, frames returned, frames seen only inside a warm-up (not returned)} */
int dvbs2gpu_segrx_get_stats(dvbs2gpu_segrx* r, int32_t* h_out3);

/* ================================================================== DVB-S inner code (rows a18-a20)
 * All buffers are DEVICE pointers; calls are asynchronous on `stream`.  Handles keep per-stream state in HBM. */

/* replaces the conversion of DVBSymToSoftBlock::process (dvbs/dvbs_syms_to_soft.cpp:7-13,28-31):
 * d_soft[2i] = int8(clamp(re*100)), d_soft[2i+1] = int8(clamp(im*100)), truncation toward zero, clamp +-127.
 * The 8192-byte blocking of :33-39 is the caller's (blocks are contiguous in d_soft). */
int dvbs2gpu_dvbs_slice(dvbs2gpu_ctx* ctx, const float* d_iq, int nsymbols, int8_t* d_soft, void* stream);

/* replaces viterbi::CCDecoder (dvbs/viterbi/cc_decoder.cpp): `nstreams` independent K=7 r=1/2 (polys 79,109) block
 * decoders of `frame_size` bits with the reference's chaining (first block from all-31 metrics, every later block
 * biased to the start state returned by the previous chain-back).  work_batch runs CCDecoder::work (:304-314) on
 * `nblocks` consecutive blocks of every stream: block b of stream s is read at d_soft + s*stream_stride +
 * b*block_stride, 2*(frame_size+6) unsigned soft bytes (128 = erasure); d_bits [nstreams][nblocks][frame_size], one
 * bit per byte. */
typedef struct dvbs2gpu_ccdec dvbs2gpu_ccdec;
int dvbs2gpu_ccdec_create(dvbs2gpu_ctx* ctx, int nstreams, int frame_size, dvbs2gpu_ccdec** out);
void dvbs2gpu_ccdec_destroy(dvbs2gpu_ccdec* h);
int dvbs2gpu_ccdec_work_batch(dvbs2gpu_ccdec* h, const uint8_t* d_soft, int64_t stream_stride, int block_stride, int nblocks,
                              uint8_t* d_bits, void* stream);

/* replaces viterbi::Viterbi_DVBS (dvbs/viterbi_all.cpp:10-276) as created by DVBSDemod::init
 * (module_dvbs_demod.cpp:23: threshold 0.15, max_outsync 20, 8192-soft blocks, phases {0, 90}): one self-locking
 * punctured decoder per stream.  work_batch = DVBSVitBlock::process (dvbs_vit.cpp:6-12) for every stream: `nblocks`
 * consecutive Viterbi_DVBS::work calls.
 *   d_soft  [nstreams][nblocks][8192] int8 (not modified; the reference rotates it in place)
 *   d_bits  [nstreams][nblocks][8192] decoded bits, one per byte; the first d_nbits[s][b] of a block are that call's
 *           output (rate 5/6: bits [6799, nbits) are never written by the reference's decoder either -- stale bytes)
 *   d_nbits [nstreams][nblocks] return values of work (0 while IDLE)
 *   d_stats [nstreams][nblocks] or NULL: ber(), getState(), rate(), locked phase and shift after each call */
typedef struct dvbs2gpu_viterbi dvbs2gpu_viterbi;
typedef struct dvbs2gpu_viterbi_stats {
    float ber;
    int32_t state, rate, phase, shift;   /* state 0 IDLE / 1 SYNCED; rate 0..4 = 1/2,2/3,3/4,5/6,7/8 */
} dvbs2gpu_viterbi_stats;
int dvbs2gpu_viterbi_create(dvbs2gpu_ctx* ctx, int nstreams, float ber_threshold, int max_outsync, dvbs2gpu_viterbi** out);
int dvbs2gpu_viterbi_reset(dvbs2gpu_viterbi* h);
void dvbs2gpu_viterbi_destroy(dvbs2gpu_viterbi* h);
int dvbs2gpu_viterbi_work_batch(dvbs2gpu_viterbi* h, const int8_t* d_soft, int nblocks, uint8_t* d_bits, int32_t* d_nbits,
                                dvbs2gpu_viterbi_stats* d_stats, void* stream);

/* replaces DVBSInterleaving (dvbs/dvbs_interleaving.h:27-70): Forney de-interleaver I=12, M=17, one per stream.
 * d_in / d_out [nstreams][nbytes] (distinct buffers), nbytes a multiple of 12 (the reference handles 8*204 per call;
 * any number of such groups may be passed at once -- the FIFO state carries over exactly). */
typedef struct dvbs2gpu_forney dvbs2gpu_forney;
int dvbs2gpu_forney_create(dvbs2gpu_ctx* ctx, int nstreams, dvbs2gpu_forney** out);
void dvbs2gpu_forney_destroy(dvbs2gpu_forney* h);
int dvbs2gpu_forney_deinterleave_batch(dvbs2gpu_forney* h, const uint8_t* d_in, int nbytes, uint8_t* d_out, void* stream);

/* ------------------------------------------------------------------ DVB-S receiver bank (rows a17-a19)
 * Mirror of dsp::dvbs::DVBSDemod (module_dvbs_demod.h:14-60, module_dvbs_demod.cpp:9-117) from the input samples up to and
 * including vit.process: demod::QPSK_ALT (FastAGC, band-edge FLL, RRC, COMPLEX_FD timing recovery, Costas<4>;
 * common/dsp/demod/qpsk_alt.cpp:136-144), DVBSymToSoftBlock and Viterbi_DVBS, for `nstreams` independent streams.
 * The output of a call is what DVBSVitBlock::process hands to the TS deframer: decoded bits, one per byte (0 bytes while
 * the decoder is not locked).  cfg mirrors the arguments of DVBSDemod::init (module_dvbs_demod.cpp:9) in the same units. */
typedef struct dvbs2gpu_dvbs_cfg {
    double symbolrate, samplerate;       /* the plugin uses samplerate = 2*symbolrate (main.cpp:139) */
    float agc_rate, rrc_alpha;
    int32_t rrc_taps;                    /* 65 (RRC_TAP_COUNT); other lengths are rejected */
    float loop_bw, fll_bw;               /* Costas and FLL loop bandwidths */
    float clock_omega_gain, clock_mu_gain, omega_rel_limit;
    float viterbi_ber_threshold;         /* 0.15 */
    int32_t viterbi_max_outsync;         /* 20   (module_dvbs_demod.cpp:23) */
} dvbs2gpu_dvbs_cfg;
typedef struct dvbs2gpu_dvbs_demod dvbs2gpu_dvbs_demod;
void dvbs2gpu_dvbs_demod_default_cfg(dvbs2gpu_dvbs_cfg* cfg);                 /* main.cpp:64-73,134-139 */
int dvbs2gpu_dvbs_demod_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_dvbs_cfg* cfg, int nstreams, int max_samples, dvbs2gpu_dvbs_demod** out);
int dvbs2gpu_dvbs_demod_reset(dvbs2gpu_dvbs_demod* d);                        /* DVBSDemod::reset + a fresh Viterbi_DVBS */
void dvbs2gpu_dvbs_demod_destroy(dvbs2gpu_dvbs_demod* d);
/* DVBSDemod::process (module_dvbs_demod.cpp:78-81) for a bank with nstreams == 1: host buffers, returns the number of decoded
 * bits written to h_bits (one per byte) or a negative error.  Synchronous.  Rate 5/6 only: of the 6826-6827 bits a Viterbi block
 * advances the output by, the reference's decoder writes 6799 (viterbi_all.cpp:246-249, cc_decoder.cpp:304-314) and leaves the rest
 * of its caller's buffer as it was; here those 27-28 bits are written as 0. */
int dvbs2gpu_dvbs_demod_process(dvbs2gpu_dvbs_demod* d, int count, const float* h_iq, uint8_t* h_bits, int cap);
/* All streams of the bank in one go: d_iq[i] DEVICE pointers to counts[i] complex samples, d_bits[i] DEVICE buffers of cap
 * bytes; out_counts[i] (host) = bits written for stream i.  Synchronous. */
int dvbs2gpu_dvbs_demod_process_batch(dvbs2gpu_dvbs_demod* d, const float* const* d_iq, const int* counts, uint8_t* const* d_bits,
                                      int cap, int* out_counts);
/* stats_viterbi_ber / _lock / _rate of every stream (module_dvbs_demod.cpp:100-114); h_out [nstreams] */
int dvbs2gpu_dvbs_demod_get_stats(dvbs2gpu_dvbs_demod* d, dvbs2gpu_viterbi_stats* h_out);
/* which 0: symbols of the last call after the Costas loop (complex64; the constellation callback, module_dvbs_demod.cpp:35);
 * which 1: 8 floats of loop state (AGC gain, FLL phase/freq, timing phase/freq/offset, Costas phase/freq).  Returns the
 * element count; copies at most cap elements when h_dst != NULL. */
int dvbs2gpu_dvbs_demod_get_tap(dvbs2gpu_dvbs_demod* d, int stream, int which, void* h_dst, int cap);

/* ------------------------------------------------------------------ DVB-S segment receiver: ONE fast DVB-S carrier
 * The DVB-S counterpart of dvbs2gpu_segrx_*: one continuous IQ stream is cut into `nsegments` overlapping segments of `own_symbols`
 * (+ `warm_symbols` of warm-up in front, own >= warm >= 8192) that run as the streams of one receiver-bank call with fresh loops and
 * a fresh Viterbi lock search; their decoded bit streams are joined where they overlap (the bits already handed out are searched for
 * in the next segment's output) and returned as ONE bit stream in order (one bit per byte), ready for dvbs2gpu_dvbs_tail_*.  A segment
 * that does not lock inside its overlap leaves a discontinuity (the deframer behind resynchronises); get_stats counts those.
 * d_iq: DEVICE pointer to `count` complex samples continuing the stream (count <= chunk_samples); d_bits: DEVICE buffer of cap bytes;
 * returns the number of bits written or a negative error.  No counterpart in the reference (one DVBSDemod per carrier, serial). */
typedef struct dvbs2gpu_dvbs_segrx dvbs2gpu_dvbs_segrx;
int dvbs2gpu_dvbs_segrx_create(dvbs2gpu_ctx* ctx, const dvbs2gpu_dvbs_cfg* cfg, int nsegments, int own_symbols, int warm_symbols, dvbs2gpu_dvbs_segrx** out);
int dvbs2gpu_dvbs_segrx_reset(dvbs2gpu_dvbs_segrx* r);
void dvbs2gpu_dvbs_segrx_destroy(dvbs2gpu_dvbs_segrx* r);
long long dvbs2gpu_dvbs_segrx_chunk_samples(dvbs2gpu_dvbs_segrx* r);
int dvbs2gpu_dvbs_segrx_process(dvbs2gpu_dvbs_segrx* r, const float* d_iq, long long count, uint8_t* d_bits, long long cap);
/* h_out4 = {segments of the last call, joined by a match, without a match, bits returned} */
int dvbs2gpu_dvbs_segrx_get_stats(dvbs2gpu_dvbs_segrx* r, int32_t* h_out4);
/* The join rule by itself, on HOST buffers (one bit per byte), no device work: where in bits[0, nbits) the stream whose last ntail bits
 * are tail[] continues (the index of the first new bit), -1 when it is not found; *inverted = 1 when the match is on the complemented
 * bits.  The last 256 bits of the tail must occur with at most 6 mismatches, anchored on one of three 64-bit keys inside them. */
long long dvbs2gpu_dvbs_segrx_find_join(const uint8_t* h_tail, long long ntail, const uint8_t* h_bits, long long nbits, int* inverted);

/* ------------------------------------------------------------------ DVB-S tail (row f: after the Viterbi decoder)
 * Replaces DVBSDefra::process / DVBS_TS_Deframer::work (dvbs/dvbs_defra.cpp:5-9, dvbs_ts_deframer.cpp:37-92), the per-frame
 * loop of DVBSDemod::process (module_dvbs_demod.cpp:83-99): DVBSInterleaving::deinterleave, 8 x DVBSReedSolomon::decode
 * (dvbs_reedsolomon.h:26-47 over libcorrect, common/correct/reed-solomon/decode.c:299-380), DVBSScrambling::descramble
 * (dvbs_scrambling.h:28-42) and the 188-byte copies, for `nstreams` independent streams with persistent state.
 * d_bits[i]: DEVICE pointer to counts[i] decoded bits (one per byte, the Viterbi output); d_ts[i]: DEVICE buffer of cap bytes that
 * receives the 188-byte TS packets of every frame found; out_bytes[i] (host).  Frame k of a call is taken at byte offset
 * 1632*k of the deframer output (the reference indexes 204*k, SURVEY Q5: wrong for k >= 1).  On a failed RS decode the
 * reference's wrapper emits the previous packet's message; so does this.  Synchronous on `stream`. */
typedef struct dvbs2gpu_dvbs_tail dvbs2gpu_dvbs_tail;
int dvbs2gpu_dvbs_tail_create(dvbs2gpu_ctx* ctx, int nstreams, int max_bits, dvbs2gpu_dvbs_tail** out);
int dvbs2gpu_dvbs_tail_reset(dvbs2gpu_dvbs_tail* t);
void dvbs2gpu_dvbs_tail_destroy(dvbs2gpu_dvbs_tail* t);
int dvbs2gpu_dvbs_tail_process_batch(dvbs2gpu_dvbs_tail* t, const uint8_t* const* d_bits, const int* counts, uint8_t* const* d_ts, int cap,
                                     int* out_bytes, void* stream);
/* h_out11 = {frames of the last call, errors_nor, errors_inv, RS error counts of the last frame's 8 packets} */
int dvbs2gpu_dvbs_tail_get_stats(dvbs2gpu_dvbs_tail* t, int stream, int32_t* h_out11);
/* Stage taps of the last call for one stream (host copies; returns the byte count, h_dst may be NULL to query it):
 *   0  frames found by the deframer, 1632 bytes each (DVBS_TS_Deframer::work output, dvbs_ts_deframer.cpp:37-92)
 *   1  the same frames after the Forney de-interleaver and the in-place RS correction, 8 x 204 bytes each
 *   2  per packet: 1 = libcorrect produced a message, 0 = it gave up (decode.c:299-380)
 *   3  per packet (int32): message bytes changed, the value DVBSReedSolomon::decode returns (dvbs_reedsolomon.h:26-47) */
int dvbs2gpu_dvbs_tail_get_tap(dvbs2gpu_dvbs_tail* t, int stream, int which, void* h_dst, int cap);
/* Stage entry (parity tests): RS(204,188) + the wrapper's stale-output rule + energy dispersal removal on `npackets` (a multiple of
 * 8) 204-byte packets supplied by the caller as stream 0's de-interleaved frames -- DVBSReedSolomon::decode and
 * DVBSScrambling::descramble without the deframer and the de-interleaver in front.  skip_rs != 0: the packets are taken as decoded
 * (descrambler alone).  HOST pointers; returns the TS bytes written to h_ts (188 per packet); taps 1-3 above then hold the
 * corrected packets, the decoder status and the error counts.  Advances stream 0's dispersal / last-message state and overwrites the
 * handle's de-interleaved packets, status and frame counts: TEST HOOK, to be called on a handle of its own, never on one that is
 * receiving (calls are serialised per context like every other entry point). */
int dvbs2gpu_dvbs_tail_rs_stage(dvbs2gpu_dvbs_tail* t, const uint8_t* h_packets, int npackets, int skip_rs, uint8_t* h_ts, int cap);
/* Stage entry (parity tests): the de-puncturers and the soft rotation that run inside the Viterbi kernel, on HOST buffers.
 *   mode 0  Depunc23 / Depunc56 ::depunc_static (depunc.h:16-38,108-137), period 3 / 6, h_state4[1] = shift
 *   mode 1  ::depunc_cont (:46-80,139-185) with h_state4 = {is_first, changing_shift, got_extra, buf} carried across calls
 *           (set_shift(s): {s > period - 1, s, 0, 128})
 *   mode 2  rotate_soft (common/codings/rotation.cpp:4-63) for h_state4[0] = 0 / 1 (PHASE_0 / PHASE_90), signed bytes in and out
 * h_out (out_cap >= 2*size + 2 bytes) keeps the caller's fill where the stage does not write.  Returns the output count. */
int dvbs2gpu_dvbs_depuncture(dvbs2gpu_ctx* ctx, int period, int mode, const uint8_t* h_in, int size, uint8_t* h_out, int out_cap,
                             int32_t* h_state4);
/* int DVBSDemod::process(int count, const complex_t* in, uint8_t* out) as a whole (module_dvbs_demod.cpp:78-99), host buffers, for a
 * one-stream receiver bank `d` and a one-stream tail `t` (created with max_bits >= the bank's bit capacity: max_samples is enough):
 * count complex samples in, the TS packets completed by this call out (188 bytes each); the decoded bits stay in HBM between the
 * two halves.  Returns the bytes written or a negative error; packets that do not fit into cap are dropped, as by the tail.  Synchronous. */
int dvbs2gpu_dvbs_process_ts(dvbs2gpu_dvbs_demod* d, dvbs2gpu_dvbs_tail* t, int count, const float* h_iq, uint8_t* h_ts, int cap);

/* ------------------------------------------------------------------ BBFRAME -> MPEG-TS / GSE parser (row f, rank 1)
 * Replaces dsp::dvbs2::BBFrameTSParser (dvbs2/bbframe_ts_parser.h:68-112, .cpp:31-390), which the reference's sink handler runs
 * on DVBS2Demod's output (main.cpp:532-558), for `nstreams` independent streams with persistent state (synchronisation,
 * the TS packet cut by a frame boundary, three GSE reassembly slots).  BBFRAMEs are kbch/8 bytes each, as the engine emits them.
 * MPEG-TS frames (TS/GS = 11) are packetised on the GPU: header CRC-8 / DFL / SYNCD checks, resynchronisation at SYNCD, one
 * 0x47 + 187-byte packet per 188 bytes of data field.  A stream that carries a GSE frame (TS/GS = 01) in a call is parsed, for
 * that call, by the library's native host parser (GSE -> GRE, fragment reassembly with CRC-32), sharing the same state.
 * Where the reference is undefined the library does this: a GSE packet that would extend beyond the end of the input of the
 * call ends the parsing of its frame; a PDU that does not fit into the rest of the output buffer or whose reassembled length
 * is negative is dropped; a fragment overflowing the 64 KiB reassembly buffer frees its slot.
 * cap must be >= nframes*kbch/8 + 376, else DVBS2GPU_ERR_CAPACITY (the reference stops with "BUFF OVF!" when fewer than 189
 * bytes are left, .cpp:178,206; with this bound a TS-only call never gets there). */
typedef struct dvbs2gpu_bbts dvbs2gpu_bbts;
int dvbs2gpu_bbts_create(dvbs2gpu_ctx* ctx, int nstreams, int kbch_bits, int max_frames, dvbs2gpu_bbts** out);
/* BBFrameTSParser::setFrameSize (.cpp:31-42): new frame size, synchronisation of every stream forgotten */
int dvbs2gpu_bbts_set_frame_size(dvbs2gpu_bbts* b, int kbch_bits);
void dvbs2gpu_bbts_destroy(dvbs2gpu_bbts* b);
/* d_bb[i]: DEVICE pointer to nframes[i] BBFRAMEs; d_out[i]: DEVICE buffer of cap bytes; out_bytes[i] (host) = bytes produced
 * (work()'s return value per stream).  Synchronous on `stream`. */
int dvbs2gpu_bbts_process_batch(dvbs2gpu_bbts* b, const uint8_t* const* d_bb, const int* nframes, uint8_t* const* d_out, int cap,
                                int* out_bytes, void* stream);
/* BBFrameTSParser::work for a bank with nstreams == 1 and host buffers: returns the bytes written to h_ts or a negative error */
int dvbs2gpu_bbts_work(dvbs2gpu_bbts* b, const uint8_t* h_bb, int cnt, uint8_t* h_ts, int cap);
/* h_out[0..10] = last_header {ts_gs, sis_mis, ccm_acm, issyi, npd, ro, isi, upl, dfl, sync, syncd}, [11] last_gse_crc_err,
 * [12] last_bb_cnt, [13] last_bb_proc, [14] last_ts_errs (main.cpp reads these for its status lines); n_out >= 15;
 * with n_out >= 17 also [15] synched, [16] bytes of the carried partial packet */
int dvbs2gpu_bbts_get_stats(dvbs2gpu_bbts* b, int stream, int32_t* h_out, int n_out);

#ifdef __cplusplus
}
#endif
#endif
