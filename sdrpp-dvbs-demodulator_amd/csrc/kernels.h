// Internal (C++) interface between the C ABI layer (capi.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s2 {

struct LdpcLayerDesc;

// Device-resident plan of one LDPC code (built from ldpc_plan.h by the context, cached per code).
struct LdpcDeviceCode {
    int code_index = -1;
    int N = 0, K = 0, R = 0, q = 0, max_deg = 0, irregular = 0, rec_dwords = 0, edges = 0;
    LdpcLayerDesc* d_layers = nullptr;
    uint32_t* d_ents = nullptr;
    uint32_t* d_rows = nullptr;
    int blocks_per_cu = 1;
};

int ldpc_blocks_per_cu(int max_deg, int irregular, int N);
hipError_t ldpc_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force,
                              uint8_t* hard, int hard_stride, int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid,
                              hipStream_t stream);

// Device-resident tables of one BCH family (GF(2^m), t).
struct BchDeviceCode {
    int m = 0, t = 0, N = 0;       // field width, correctable errors, 2^m - 1
    int K_full = 0;                // unshortened message length N - m*t
    uint16_t* d_log = nullptr;     // [2^m]  log(0) = N   (galois_field.hh:158)
    uint16_t* d_exp = nullptr;     // [2^m]  exp(N) = 0
    uint16_t* d_imap = nullptr;    // [2^m]  Artin-Schreier map for degree-2 locators
    uint16_t* d_syn_tab = nullptr; // [t][3][256] byte-Horner tables for the odd syndromes
};

hipError_t bch_syndromes_launch(const BchDeviceCode& C, const uint8_t* frames, int frame_stride, int nbch, int nframes,
                                uint16_t* syn /*[nframes][32]*/, hipStream_t stream);
hipError_t bch_correct_launch(const BchDeviceCode& C, uint8_t* frames, int frame_stride, int nbch, int kbch, int nframes,
                              const uint16_t* syn, int32_t* corrections, hipStream_t stream);
hipError_t bb_descramble_launch(const uint8_t* frames, int frame_stride, const uint8_t* prbs, int out_bytes, int nframes,
                                uint8_t* out, hipStream_t stream);

}  // namespace s2
