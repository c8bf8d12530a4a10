// Host-side tables of the HALF-ROW LDPC decoder (ldpc_split_kernel.hip): two lanes per parity-check row, one frame per workgroup.
//
// The reference sweeps rows in order (xdsopl-ldpc-pabr/layered_decoder.hh:46-74); which rows of a layer may run side by side, and which
// links must be read late / written early, is the plan of ldpc_plan.h -- unchanged.  What changes is who holds a row: thread t of the
// 768-thread workgroup owns HALF h = t & 1 of row j = t >> 1 of every layer.  A row has NL = max_deg + 2 links (table links, own parity
// bit, previous parity bit); half 0 holds table links [0, HS), half 1 table links [HS, max_deg), then the own and the previous parity
// bit, HS = ceil(NL / 2) slots each.  The plan puts a layer's shared ("conflict") links first, so they all sit in half 0 (codes whose
// layers share more than HS links keep the lane-per-row decoder).  min / xor are associative and commutative, so joining the two halves'
// (min0, min1, sign) with one cross-lane step leaves algorithms.hh:242-255 bit-exact.
//
// Tables (all indexed by the thread, stride LDPC_SPLIT_T, so that every fetch is one coalesced vector load):
//   atab  [q][768][NPW]  link addresses, two 16-bit LDS byte offsets per word: slot 2p | slot 2p+1 << 16.  Parity bits included (the kernel
//                        does no address arithmetic at all).  Absent slots and the missing previous parity bit of row 0 of layer 0 point
//                        at the scratch byte behind the posteriors (offset N): read, masked, written, never looked at.
//   rows  per conflict layer [768] row words (half 0: level | late << 8 | early << 20 of its row; half 1 and idle lanes: 0), followed for
//                        quad-walk layers by the step list of ldpc_plan.h
#pragma once
#include "ldpc_plan.h"

namespace s2 {

constexpr int LDPC_SPLIT_T = 768;              // threads per workgroup = 2 x 384 (rows 360..383 idle)

struct LdpcSplitPlan {
    bool ok = false;
    int hs = 0;                        // slots per half
    int npw = 0;                       // address words per thread and layer (power of two)
    int rec_dwords = 0;                // message record per thread and layer, dwords (1 byte per slot)
    std::vector<LdpcLayerDesc> layers; // the plan's descriptors with row_off pointing into rows below
    std::vector<uint32_t> atab;
    std::vector<uint32_t> rows;
};

// which codes the half-row decoder takes: regular, an even number of links per row (both halves then hold the same number of slots),
// every layer's shared links inside half 0 and of a kind the kernel implements (free, chain, quad walk, levels over <= 4 links)
inline LdpcSplitPlan build_ldpc_split_plan(const LdpcPlan& P) {
    LdpcSplitPlan S;
    const int NL = P.max_deg + 2;
    if (P.min_deg != P.max_deg || (NL & 1)) return S;
    S.hs = NL / 2;
    const int pairs = (S.hs + 1) / 2;
    S.npw = pairs <= 1 ? 1 : pairs <= 2 ? 2 : pairs <= 4 ? 4 : 8;
    S.rec_dwords = S.hs <= 4 ? 1 : S.hs <= 8 ? 2 : 4;
    if (S.hs > 16) return S;
    for (const LdpcLayerDesc& L : P.layers) {
        const int nc = (int)(L.depth_nc >> 16), depth = (int)(L.depth_nc & 0xffffu);
        if (depth > 1 && nc > 4) return S;
        if (nc > S.hs) return S;
    }
    const int q = P.q, K = P.K, N = P.N, T = LDPC_SPLIT_T;
    const uint32_t dummy = (uint32_t)N;
    S.atab.assign((size_t)q * T * S.npw, dummy | (dummy << 16));
    S.layers = P.layers;
    for (int i = 0; i < q; ++i) {
        const LdpcLayerDesc& L = P.layers[i];
        for (int t = 0; t < T; ++t) {
            const int j = t >> 1, h = t & 1;
            uint32_t* w = &S.atab[((size_t)i * T + t) * S.npw];
            if (j >= 360) continue;                       // idle lanes: the scratch byte
            for (int s = 0; s < S.hs; ++s) {
                uint32_t a = dummy;
                const int k = h * S.hs + s;               // link number in the row's order: table links, own parity, previous parity
                if (k < P.max_deg) {
                    const uint32_t e = P.ents[L.ent_off + k];
                    a = 360u * (e >> 16) + ((uint32_t)j + (e & 0xffffu)) % 360u;
                } else if (k == P.max_deg) {
                    a = (uint32_t)(K + 360 * i + j);
                } else if (k == P.max_deg + 1) {
                    if (i > 0) a = (uint32_t)(K + 360 * (i - 1) + j);
                    else if (j > 0) a = (uint32_t)(K + 360 * (q - 1) + j - 1);
                }
                w[s >> 1] = (w[s >> 1] & ~(0xffffu << (16 * (s & 1)))) | (a << (16 * (s & 1)));
            }
        }
        const int depth = (int)(L.depth_nc & 0xffffu);
        if (depth > 1) {
            S.layers[i].row_off = (uint32_t)S.rows.size();
            for (int t = 0; t < T; ++t) S.rows.push_back(((t & 1) == 0 && (t >> 1) < 360) ? P.rows[L.row_off + (t >> 1)] : 0u);
            if ((L.deg >> 16) == LDPC_WALK_MARK) {
                const uint32_t hd = P.rows[L.row_off + 360];
                const size_t nwords = 1 + (size_t)((hd & 0xffffu) + 3) * 16;       // header, the steps and the three empty ones behind them
                for (size_t n = 0; n < nwords; ++n) S.rows.push_back(P.rows[L.row_off + 360 + n]);
            }
        }
    }
    if (S.rows.empty()) S.rows.push_back(0);
    while (S.rows.size() < (size_t)T) S.rows.push_back(0);     // (conflict-free layers fetch "their" row word from offset 0: a word nobody looks at)
    S.ok = true;
    return S;
}

}  // namespace s2
