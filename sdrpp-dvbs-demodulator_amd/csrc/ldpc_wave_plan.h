// Host-side plan of the WAVE-PER-FRAME LDPC decoder (ldpc_wave_kernel.hip) for one DVB-S2 code.
//
// The lane-per-row decoder (ldpc_kernel.hip) pays a workgroup barrier per dependency level of a layer; short codes of high degree
// (C10: five layers, 27 information links per row, 8-12 of them shared with other rows of the layer, 5-33 levels) spend their time
// there.  Here ONE WAVE decodes one frame: eight lanes share a row (link slot k = 8 * kk + lane % 8, kk < LW), eight rows make a
// step, and the steps of a layer are listed level by level -- rows of a step are independent, and a later step sees an earlier step's
// posteriors because the LDS executes one wave's instructions in order.  No barrier, no early / late bookkeeping: the reference's
// sequential row order (xdsopl-ldpc-pabr/layered_decoder.hh:46-74) is kept by construction wherever two rows share a bit.
// Link slots: 0 = the row's own parity bit, 1 = the previous parity bit, 2.. = the information links (absent ones of an irregular
// code at the tail).
#pragma once
#include <cstdint>
#include <vector>
#include "ldpc_plan.h"

namespace s2 {

constexpr int LDPC_WAVE_CHUNK = 4;          // steps per prefetch chunk (ldpc_wave_kernel.hip)
constexpr uint16_t LDPC_WAVE_NOROW = 0xffffu;
#ifndef LDPC_WAVE_LDS_RESERVE_KB
#define LDPC_WAVE_LDS_RESERVE_KB 28
#endif
constexpr int LDPC_WAVE_LDS_RESERVE = LDPC_WAVE_LDS_RESERVE_KB * 1024;   // LDS per CU left to the front-end kernels that run beside the decoder in the pipelined mode
// which codes (index into QC_CODES) go to the wave-per-frame decoder unless the context option ldpc_wave says otherwise (per-code timings: DESIGN.md)
// MI355X, 16384 frames x 50 forced iterations, lane-per-row vs wave-per-frame: 3/5 short 141 vs 119 ms, 4/5 short 65 vs 61, 5/6 short 155 vs 60,
// 8/9 short 106 vs 48; the other six short codes (few links per row, few levels) stay with the lane-per-row decoder (1/4: 53 vs 305 ms)
inline bool ldpc_wave_default(int code_index) { return code_index == 15 || code_index == 18 || code_index == 19 || code_index == 20; }

struct LdpcWavePlan {
    int lw = 0;                             // link slots per lane, ceil((max_deg + 2) / 8)
    int nl_min = 0;                         // smallest number of present links of a row over the layers (slots from there on may be absent)
    int nsteps = 0;                         // steps per sweep, a multiple of LDPC_WAVE_CHUNK (+ 2 empty chunks behind the list for the prefetch)
    std::vector<uint32_t> lanec;            // [q][8 lanes][lw] thr | cA << 16: lane reads byte (j >= thr ? cA - 360 : cA) + j of the posterior array; then [q][8] absent masks (bit kk)
    int absent_base = 0;
    std::vector<uint16_t> steps;            // [nsteps + 2 chunks][8] row ids (360 * layer + j), LDPC_WAVE_NOROW = empty slot; a layer's steps are padded to whole chunks
    std::vector<uint32_t> layer_end;        // [q] chunk index at which layer i's steps end
};

inline LdpcWavePlan build_ldpc_wave_plan(const LdpcPlan& P) {
    LdpcWavePlan W;
    const int NL = P.max_deg + 2;
    W.lw = (NL + 7) / 8;
    W.nl_min = P.min_deg + 2;
    const int q = P.q, K = P.K;
    W.lanec.assign((size_t)q * 8 * W.lw, 0);
    std::vector<uint32_t> absent((size_t)q * 8, 0);
    for (int i = 0; i < q; ++i) {
        const LdpcLayerDesc& L = P.layers[i];
        const int deg = (int)(L.deg & 0xffffu);
        for (int l8 = 0; l8 < 8; ++l8)
            for (int kk = 0; kk < W.lw; ++kk) {
                const int k = 8 * kk + l8;
                int sp = 0, base = 0;
                bool present = true;
                if (k == 0) { sp = 0; base = K + 360 * i; }                                   // own parity bit of row j: K + 360 i + j
                else if (k == 1) {                                                            // previous parity bit (layered_decoder.hh:56-60)
                    if (i) { sp = 0; base = K + 360 * (i - 1); }
                    else { sp = 359; base = K + 360 * (q - 1); }                              // row j of layer 0: bit (q-1, j-1); row 0 has none (masked in the kernel)
                } else if (k - 2 < deg) {
                    const uint32_t e = P.ents[L.ent_off + (k - 2)];
                    sp = (int)(e & 0xffffu); base = 360 * (int)(e >> 16);
                } else present = false;
                const int thr = sp ? 360 - sp : 0x7fff;                                       // j >= thr: j + sp wraps
                const int cA = present ? base + sp : 0;
                W.lanec[((size_t)i * 8 + l8) * W.lw + kk] = (uint32_t)thr | ((uint32_t)cA << 16);
                if (!present) absent[(size_t)i * 8 + l8] |= 1u << kk;
            }
    }
    W.absent_base = (int)W.lanec.size();
    W.lanec.insert(W.lanec.end(), absent.begin(), absent.end());
    // steps: per layer the rows level by level, eight per step; every layer ends on a chunk boundary
    for (int i = 0; i < q; ++i) {
        const LdpcLayerDesc& L = P.layers[i];
        const int depth = (int)(L.depth_nc & 0xffffu);
        std::vector<std::vector<int>> by_level(depth + 1);
        for (int j = 0; j < 360; ++j) by_level[depth > 1 ? (int)(P.rows[L.row_off + j] & 0xffu) : 1].push_back(j);
        for (int lv = 1; lv <= depth; ++lv)
            for (size_t o = 0; o < by_level[lv].size(); o += 8)
                for (size_t t = 0; t < 8; ++t) W.steps.push_back(o + t < by_level[lv].size() ? (uint16_t)(360 * i + by_level[lv][o + t]) : LDPC_WAVE_NOROW);
        while ((W.steps.size() / 8) % LDPC_WAVE_CHUNK) W.steps.insert(W.steps.end(), 8, LDPC_WAVE_NOROW);
        W.layer_end.push_back((uint32_t)(W.steps.size() / 8 / LDPC_WAVE_CHUNK));
    }
    W.nsteps = (int)(W.steps.size() / 8);
    W.steps.insert(W.steps.end(), 8 * 2 * LDPC_WAVE_CHUNK, LDPC_WAVE_NOROW);
    return W;
}

}  // namespace s2
