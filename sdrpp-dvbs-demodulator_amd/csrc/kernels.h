// Internal (C++) interface between the C ABI layer (capi.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "launch_count.h"

namespace s2 {

struct LdpcLayerDesc;

// Device-resident plan of one LDPC code (built from ldpc_plan.h by the context, cached per code).
struct LdpcDeviceCode {
    int code_index = -1;
    int N = 0, K = 0, R = 0, q = 0, max_deg = 0, irregular = 0, rec_dwords = 0, edges = 0, pent_base = 0, synd_base = 0;
    LdpcLayerDesc* d_layers = nullptr;
    uint32_t* d_ents = nullptr;
    uint32_t* d_rows = nullptr;
    uint32_t* d_atab = nullptr;     // per-row link addresses (regular codes up to degree 12, ldpc_plan.h); null otherwise
    int blocks_per_cu = 1;
    // wave-per-frame form (ldpc_wave_plan.h / ldpc_wave_kernel.hip), built for short frames
    int wave_lw = 0, wave_nsteps = 0, wave_nl_min = 0, wave_absent_base = 0;
    uint32_t* d_wave_lanec = nullptr;
    uint16_t* d_wave_steps = nullptr;
    uint32_t* d_wave_layer_end = nullptr;
    bool use_wave = false;          // which of the two decoders a batch of this code goes to
    // half-row form (ldpc_split_plan.h / ldpc_split_kernel.hip): two lanes per row, one frame per workgroup
    struct LdpcSplitLayer* d_split_layers = nullptr;
    uint32_t* d_split_atab = nullptr;
    int split_npl = 0, split_rec_total = 0, split_blocks_per_cu = 1, split_tab_words = 0;
    bool use_split = false;
};

int ldpc_blocks_per_cu(int max_deg, int irregular, int N);
int ldpc_frames_per_block(int nframes, int num_cus);
hipError_t ldpc_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force,
                              uint8_t* hard, int hard_stride, int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid,
                              int fpb, hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws);
size_t ldpc_sign_ws_bytes_per_slot();
bool ldpc_split_supported(int max_deg);
bool ldpc_split_noprev_shared(int max_deg);      // kernels whose layers with shared bits handle the row without a previous parity bit
int ldpc_split_blocks_per_cu(int max_deg, int N);
size_t ldpc_split_msg_bytes_per_block(const LdpcDeviceCode& C);
hipError_t ldpc_split_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force, uint8_t* hard, int hard_stride,
                                    int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid, hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws, int dbg = 0);
size_t ldpc_wave_msg_bytes_per_frame(const LdpcDeviceCode& C);
size_t ldpc_wave_lds_bytes(const LdpcDeviceCode& C);
hipError_t ldpc_wave_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force, uint8_t* hard, int hard_stride,
                                   int8_t* post, int32_t* trials, uint8_t* msg_ws, int grid, hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws);

// Device-resident tables of one BCH family (GF(2^m), t).
struct BchDeviceCode {
    int m = 0, t = 0, N = 0;       // field width, correctable errors, 2^m - 1
    int K_full = 0;                // unshortened message length N - m*t
    uint16_t* d_log = nullptr;     // [2^m]  log(0) = N   (galois_field.hh:158)
    uint16_t* d_exp = nullptr;     // [2^m]  exp(N) = 0
    uint16_t* d_imap = nullptr;    // [2^m]  Artin-Schreier map for degree-2 locators
    uint16_t* d_syn_tab = nullptr; // [t][3][256] byte-Horner tables for the odd syndromes
};

// `todo` = int32 [2 + nframes]: {frames with a non-zero syndrome, the correction kernel's work counter} (zeroed by bch_syndromes_launch) and the list
// of those frames: the correction kernel's workgroups take them one by one -- a frame that needs correcting costs a few hundred microseconds of
// dependent table lookups, a clean one nothing, and which are which is the channel's business
hipError_t bch_syndromes_launch(const BchDeviceCode& C, const uint8_t* frames, int frame_stride, int nbch, int nframes,
                                uint16_t* syn /*[nframes][32]*/, int32_t* todo, int32_t* corrections, hipStream_t stream);
hipError_t bch_correct_launch(const BchDeviceCode& C, uint8_t* frames, int frame_stride, int nbch, int kbch, int nframes,
                              const uint16_t* syn, int32_t* todo, int32_t* corrections, hipStream_t stream);
hipError_t bb_descramble_launch(const uint8_t* frames, int frame_stride, const uint8_t* prbs, int out_bytes, int nframes,
                                uint8_t* out, hipStream_t stream);

// ------------------------------------------------------------------ DVB-S inner code (dvbs_kernels.hip)
// One Viterbi_DVBS object (viterbi_all.h:33-160) per stream; decoders 0-4 = BER-test decoders (rates 1/2,2/3,3/4,5/6,7/8),
// 5-9 = main decoders; dep[0] = Depunc23, dep[1] = Depunc56.
struct DvbsVitState {
    int state, rate, phase, shift, invalid;
    float ber;
    int dec_ss[10], dec_biased[10];
    int enc_state[5];
    int dep_first[2], dep_shift[2], dep_extra[2], dep_buf[2];
};
struct DvbsVitStats {   // == dvbs2gpu_viterbi_stats
    float ber;
    int state, rate, phase, shift;
};
// per-stream workspace layout (bytes): ber_soft[2048] ++ ber_depunc[8192+64] | ber_enc[8192] | ber_dec[2048] | soft[8192+64] |
// depunc[4*8192] | decision words [(7168+6) rounded up]
constexpr int DVBS_VIT_WS_BER_ENC = 2048 + 8192 + 64;
constexpr int DVBS_VIT_WS_BER_DEC = DVBS_VIT_WS_BER_ENC + 8192;
constexpr int DVBS_VIT_WS_SOFT = DVBS_VIT_WS_BER_DEC + 2048;
constexpr int DVBS_VIT_WS_DEPUNC = DVBS_VIT_WS_SOFT + 8192 + 64;
constexpr int DVBS_VIT_WS_DEC = DVBS_VIT_WS_DEPUNC + 4 * 8192;
constexpr int DVBS_VIT_WS_BYTES = DVBS_VIT_WS_DEC + 7232 * 8;
constexpr int DVBS_FORNEY_HIST = 204 * 11;
static_assert(DVBS_VIT_WS_DEC % 8 == 0 && DVBS_VIT_WS_BYTES % 8 == 0, "decision words must stay 8-byte aligned");

hipError_t dvbs_slice_launch(const float* d_iq, int n, int8_t* d_out, hipStream_t st);
hipError_t dvbs_cc_decode_launch(const uint8_t* d_in, long stream_stride, int block_stride, int nstreams, int nblocks, int frame_size,
                                 uint8_t* d_out, long out_stream_stride, unsigned long long* d_dec_ws, int* d_state, hipStream_t st);
hipError_t dvbs_viterbi_launch(const int8_t* d_soft, const int8_t* const* d_soft_ptrs, const int* d_nblk, int nstreams, int nblocks,
                               uint8_t* d_bits, int* d_nbits, DvbsVitStats* d_stats, DvbsVitState* d_states, uint8_t* d_ws, float thr,
                               int max_outsync, hipStream_t st, const int* d_blk0 = nullptr);
struct DvbsTailState {     // per stream: energy-dispersal phase + the RS wrapper's last decoded message (dvbs_kernels.hip)
    int prbs_pos;
    int pad;
    uint8_t last_msg[192];
};
hipError_t dvbs_tail_launch(const uint8_t* const* d_in_ptrs, const int* d_counts, int nstreams, int max_bits, uint8_t* d_hist, uint8_t* d_hist_next,
                            uint8_t* d_v, long v_stride, int max_frames, int* d_hit_pos, int* d_nframes, int* d_errs, uint8_t* d_frames,
                            uint8_t* d_deint, long frames_stride, uint8_t* d_forney_hist, uint8_t* d_status, const uint8_t* d_gf, const uint8_t* d_prbs,
                            DvbsTailState* d_state, uint8_t* const* d_out_ptrs, int cap, int* d_out_bytes, int* d_rs_err, hipStream_t st);
hipError_t dvbs_tail_rs_finish_launch(int nstreams, int max_frames, const int* d_nframes, uint8_t* d_deint, long frames_stride, uint8_t* d_status,
                                      const uint8_t* d_gf, const uint8_t* d_prbs, DvbsTailState* d_state, uint8_t* const* d_out_ptrs, int cap,
                                      int* d_out_bytes, int* d_rs_err, int skip_rs, hipStream_t st);
hipError_t dvbs_depunc_stage_launch(int period, int mode, const uint8_t* d_in, int size, uint8_t* d_out, int* d_state4, int* d_n, hipStream_t st);
hipError_t dvbs_pack_bits_launch(const uint8_t* d_bits, const int* d_nbits, const int* d_nblk, int nstreams, int nblocks, uint8_t* const* d_out_ptrs,
                                 int cap, int* d_out_count, hipStream_t st);
hipError_t dvbs_deinterleave_launch(const uint8_t* d_in, long stream_stride, int nstreams, int nbytes, uint8_t* d_out, uint8_t* d_hist,
                                    hipStream_t st);

}  // namespace s2
