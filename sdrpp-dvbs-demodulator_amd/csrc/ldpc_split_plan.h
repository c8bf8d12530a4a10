// Host-side tables of the HALF-ROW LDPC decoder (ldpc_split_kernel.hip): two lanes per parity-check row, one frame per workgroup.
//
// The reference sweeps rows in order (xdsopl-ldpc-pabr/layered_decoder.hh:46-74); which rows of a layer may run side by side is the plan of
// ldpc_plan.h (levels, late / early links, chain steps) -- unchanged.  What changes is who holds a row and how dependent rows are ordered:
//   * thread t of the 768-thread workgroup owns HALF h = t & 1 of a row.  A row has NL = max_deg + 2 links (table links, own parity bit,
//     previous parity bit); half 0 holds links [0, HS), half 1 links [HS, NL), HS = ceil(NL / 2) slots each (an odd NL: half 1 has a link less and its last slot is a constant neutral link; an odd HS leaves the second
//     half of the last register pair to a constant neutral link, ldpc_split_kernel.hip).  min / xor are associative and commutative, so
//     joining the two halves' (min0, min1, sign) with one cross-lane step leaves algorithms.hh:242-255 bit-exact.
//   * a sweep is a list of PSEUDO-LAYERS, each ending in a workgroup barrier, one per layer of the code, all twelve waves at work in every one of them.  Everything a
//     thread needs for one -- the LDS byte offsets of its slots, parity bits included, and its row word -- comes from a per-thread table entry:
//       kind 0 (row update): a conflict-free layer, one row per lane pair, rows in lane order; kind 7: the same for layer 0, whose row 0 has no previous parity bit.
//       kinds 1 / 8 (chain walk / speculative passes): a layer with shared bits: all rows at once in lane order, the shared links (they all sit in half 0) resolved in a
//           middle section -- a single shared pair with short chains (< LDPC_SPLIT_SPEC_MIN_T steps): chains walked by d lanes of ONE wave through per-row hand-off records
//           (ldpc_kernel.hip's chain walk); two pairs, a triple or long chains: every row in its own lane, in passes that start from the posteriors as the previous layer left
//           them and end when no row reads anything new (ldpc_split_kernel.hip: spec_layer; rounds 5 / 6 walked these level by level with one wave -- barrier per level,
//           quad walk, level walk: kinds 3 / 6 / 5, all gone).
//     Round 5 also carried two alternatives for the layers with shared bits -- one packed conflict-free pseudo-layer per dependency level, and level PASSES under the lanes'
//     level mask -- as context options; both measured slower at every depth (rate 3/4, 4096 frames x 50 iterations: every chain layer walked 39.6 ms, chain layers of up to
//     5 levels as passes 41.9, layers of up to 3 levels packed 42.5: a barrier-separated pass costs ~1 000 cycles whatever it does, a packed level ~1 750, a walked row
//     ~135; profiles/r05_ldpc_split_layers.txt) and are gone: with every wave in every pseudo-layer the layer loop needs no per-wave bookkeeping.  (What makes kind 8's passes
//     pay where those did not: they do not follow the levels -- a frame that has converged needs TWO whatever the depth.)
//   * idle lanes (beyond the packed rows; rows 360..383 of a full layer) point every slot at scratch bytes behind the posteriors: they run
//     the same instructions and store to bytes nobody reads -- no exec masking in the row update.  The missing previous parity bit of row 0
//     of layer 0 points there too; its pseudo-layer carries a flag and the thread index.
//
// Tables: atab [pseudo-layer][768][NPW] words, two 16-bit LDS byte offsets per word (slot 2p | slot 2p+1 << 16), the row word (kind 1: level | late << 8 | early << 12; kind 8: the level)
// in the 16 bits behind the last slot; behind the tables the SIDE ENTRIES of the kind-8 layers, [layer][384 rows][2 words] (LdpcSplitPlan::side); message records [pseudo-layer][64 * nw][REC] per workgroup.
#pragma once
#include "ldpc_plan.h"

namespace s2 {

constexpr int LDPC_SPLIT_T = 768;              // threads per workgroup = 2 x 384 (rows 360..383 idle)
constexpr int LDPC_SPLIT_SCRATCH = 64;
#ifndef LDPC_SPLIT_SPEC_MIN_T
#define LDPC_SPLIT_SPEC_MIN_T 64              // a single shared pair whose chains are at least this long goes through the speculative passes (kind 8), shorter ones through the chain walk (kind 1)
#endif         // scratch bytes behind the posteriors (one per lane of a wave)

struct LdpcSplitLayer {   // 16 bytes = one s_load_dwordx4
    uint32_t kind_nw;     // bits 0..7 kind, 8..15 waves (always 12), 16..19 nc = shared links of the layer (slots 0..nc-1 of half 0), bit 20: holds row 0 of layer 0 (no previous parity bit)
    uint32_t aux;         // kind 7: the half-1 thread of row 0 of layer 0 (bit 20); kind 1: chain step d | (359 / d) << 16; kind 8: levels << 16
    uint32_t rec_off;     // dword offset of the pseudo-layer's records inside a workgroup's message workspace
    uint32_t ent_off;     // kind 1: index of the layer's link entries in LdpcPlan::ents (the walker reads link 1's); kind 8: word offset of the layer's side entries from the start of atab
};

struct LdpcSplitPlan {
    bool ok = false;
    const char* why = "";            // why the half-row decoder does not take the code (ok == false)
    int hs = 0;                        // slots per half
    int npw = 0;                       // table words per thread and pseudo-layer (power of two)
    int rec_dwords = 0;                // message record per thread and pseudo-layer, dwords (1 byte per slot)
    int rec_total = 0;                 // dwords of message workspace per workgroup
    int chain_layers = 0;
    bool noprev_shared = false;        // layer 0 -- it holds the row without a previous parity bit -- has shared bits: only kernels built for that take the code (SplitShape::NOPREV_SHARED)
    std::vector<LdpcSplitLayer> layers;
    std::vector<uint32_t> atab;
    std::vector<uint32_t> side;        // side entries of the kind-8 layers, [layer][384 rows][2]: uploaded BEHIND atab (LdpcSplitLayer::ent_off counts from the start of atab)
    std::vector<int> row_of;           // [pseudo-layer][768 / 2] original row of every lane pair (-1: idle) and
    std::vector<int> layer_of;         // [pseudo-layer] original layer: what the CPU-side plan test checks against the reference order
};

inline int ldpc_split_npw(int hs) {
    const int words = hs / 2 + 1;      // the 16 bits behind the last slot hold the row word
    return words <= 1 ? 1 : words <= 2 ? 2 : words <= 4 ? 4 : 8;
}

inline int ldpc_split_plan_rec_dwords(int max_deg) { const int hs = (max_deg + 3) / 2; return hs <= 4 ? 1 : hs <= 8 ? 2 : 4; }

// which codes the half-row decoder takes: regular ones with an even number of links per row (both halves then hold the same number of slots)
inline LdpcSplitPlan build_ldpc_split_plan(const LdpcPlan& P) {
    LdpcSplitPlan S;
    const int NL = P.max_deg + 2;
    if (P.min_deg != P.max_deg || (NL + 1) / 2 > 14 || P.N + LDPC_SPLIT_SCRATCH > 65536) { S.why = "irregular row degree, more than 14 slots per half or more than 65 472 bits"; return S; }     // (ceil(NL / 2) slots + the row word in at most 8 table words)
    for (const LdpcLayerDesc& L : P.layers) {
        const int nc = (int)(L.depth_nc >> 16), depth = (int)(L.depth_nc & 0xffffu);
        if (depth > 1 && (nc > 4 || nc > (NL + 1) / 2)) { S.why = "a layer with more than four shared links"; return S; }      // (the middle sections handle up to four shared links, all in half 0)
    }
    S.hs = (NL + 1) / 2;              // (an odd NL: half 1's last slot is a neutral link, its address a scratch byte)
    S.npw = ldpc_split_npw(S.hs);
    S.rec_dwords = ldpc_split_plan_rec_dwords(P.max_deg);
    const int q = P.q, K = P.K, N = P.N, T = LDPC_SPLIT_T, hs = S.hs, npw = S.npw;
    auto scratch = [&](int t) { return (uint32_t)(N + (t & (LDPC_SPLIT_SCRATCH - 1))); };
    auto slot_addr = [&](int i, int j, int k) -> uint32_t {     // link k of row j of layer i; ~0: none
        const LdpcLayerDesc& L = P.layers[i];
        if (k < P.max_deg) {
            const uint32_t e = P.ents[L.ent_off + k];
            return 360u * (e >> 16) + ((uint32_t)j + (e & 0xffffu)) % 360u;
        }
        if (k == P.max_deg) return (uint32_t)(K + 360 * i + j);
        if (i > 0) return (uint32_t)(K + 360 * (i - 1) + j);
        if (j > 0) return (uint32_t)(K + 360 * (q - 1) + j - 1);
        return ~0u;
    };
    std::vector<std::pair<int, std::vector<uint32_t>>> side_of;      // (pseudo-layer, its side entries): appended behind the tables once every layer is emitted
    bool bad_noprev = false;      // the row without a previous parity bit in a layer with shared bits: kinds 1 / 8 ask bit 20 of the descriptor in the kernels built for it (S.noprev_shared)
    // one pseudo-layer: rows[] = the original rows of lane pairs 0, 1, ...; info[] = their row words (kind 1)
    auto emit = [&](int kind, int i, const std::vector<int>& rows, const std::vector<uint32_t>* info, uint32_t aux) {
        LdpcSplitLayer D{};
        const int nw = T / 64;
        D.kind_nw = (uint32_t)kind | ((uint32_t)nw << 8) | ((kind != 0 ? (P.layers[i].depth_nc >> 16) : 0u) << 16);
        D.aux = aux;
        D.rec_off = (uint32_t)S.rec_total;
        D.ent_off = P.layers[i].ent_off;
        S.rec_total += nw * 64 * S.rec_dwords;
        const size_t base = S.atab.size();
        S.atab.resize(base + (size_t)T * npw, 0);
        const size_t rbase = S.row_of.size();
        S.row_of.resize(rbase + T / 2, -1);
        for (int t = 0; t < T; ++t) {
            uint32_t* w = &S.atab[base + (size_t)t * npw];
            const int pr = t >> 1, h = t & 1;
            const int j = pr < (int)rows.size() ? rows[pr] : -1;
            for (int s = 0; s < hs; ++s) {
                uint32_t a = scratch(t);
                if (j >= 0 && h * hs + s < NL) {
                    a = slot_addr(i, j, h * hs + s);
                    if (a == ~0u) { a = scratch(t); D.kind_nw |= 1u << 20; if (kind == 0) { D.aux = (uint32_t)t; D.kind_nw = (D.kind_nw & ~0xffu) | 7u; } else bad_noprev = true; }
                }
                w[s >> 1] |= a << (16 * (s & 1));
            }
            if (j >= 0 && h == 0) S.row_of[rbase + pr] = j;
            // the row word (idle lanes: 0, every condition on it false); half 1 gets the level alone
            if (j >= 0 && info && (h == 0 || kind == 1 || kind == 8)) w[hs >> 1] |= ((*info)[pr] & (h == 1 ? 0xffu : 0xffffu)) << (16 * (hs & 1));
        }
        S.layers.push_back(D);
        S.layer_of.push_back(i);
    };
    for (int i = 0; i < q; ++i) {
        const LdpcLayerDesc& L = P.layers[i];
        const int depth = (int)(L.depth_nc & 0xffffu);
        const uint32_t chain = L.deg >> 16;
        std::vector<int> rows;
        if (depth == 1) {
            for (int j = 0; j < 360; ++j) rows.push_back(j);
            emit(0, i, rows, nullptr, 0);
        } else {
            std::vector<uint32_t> info;
            for (int j = 0; j < 360; ++j) {
                const uint32_t rw = P.rows[L.row_off + j];
                rows.push_back(j);
                info.push_back((rw & 0xffu) | (((rw >> 8) & 15u) << 8) | (((rw >> 20) & 15u) << 12));
            }
            if (chain != 0 && chain != LDPC_WALK_MARK && 359u / chain < (uint32_t)LDPC_SPLIT_SPEC_MIN_T) emit(1, i, rows, &info, chain | ((359u / chain) << 16));
            else {
                // kind 8, the speculative passes (ldpc_split_kernel.hip: spec_layer): for every row and shared slot WHO touched the bit last before this row -- rows in
                // the reference's order, the shared links are the row's first ones (ldpc_plan.h) -- and whether a later row touches it.  A row without predecessor in the
                // layer has level 1; a predecessor of level 1 hands the bit over in place (it is finished before the passes start), one of a higher level through its
                // output cell, four bytes per row behind the posteriors: the side entry of (row, slot) holds the distance from the row's own cell back to that byte.
                std::map<uint32_t, std::pair<int, int>> last;       // bit -> (row, slot) of its latest toucher
                std::vector<int> lvl(360, 1);
                std::vector<uint32_t> side(2 * 384, 0);
                int depth8 = 1;
                for (int j = 0; j < 360; ++j) {
                    std::pair<int, int> pred[4] = {{-1, 0}, {-1, 0}, {-1, 0}, {-1, 0}};
                    for (int k = 0; k < NL; ++k) {
                        const uint32_t a = slot_addr(i, j, k);
                        if (a == ~0u) continue;
                        auto it = last.find(a);
                        if (it != last.end()) {
                            if (k >= 4 || k >= hs) { LdpcSplitPlan F; F.why = "a shared link outside slots 0..3 of half 0"; return F; }          // (a shared link outside slots 0..3 of half 0)
                            pred[k] = it->second;
                            lvl[j] = std::max(lvl[j], lvl[it->second.first] + 1);
                            side[2 * it->second.first + (it->second.second >> 1)] |= 0x8000u << (16 * (it->second.second & 1));       // that row's slot has a successor
                        }
                        last[a] = {j, k};
                    }
                    depth8 = std::max(depth8, lvl[j]);
                    for (int k = 0; k < 4; ++k)
                        if (pred[k].first >= 0 && lvl[pred[k].first] > 1) side[2 * j + (k >> 1)] |= (uint32_t)(4 * (j - pred[k].first) - pred[k].second) << (16 * (k & 1));
                }
                if (depth8 > 255 || hs < 4) { LdpcSplitPlan F; F.why = "a speculative layer deeper than 255 levels or fewer than 4 slots per half"; return F; }
                for (int j = 0; j < 360; ++j) info[j] = (uint32_t)lvl[j];
                emit(8, i, rows, &info, (uint32_t)depth8 << 16);
                side_of.push_back({(int)S.layers.size() - 1, side});
            }
            S.chain_layers++;
        }
    }
    for (auto& so : side_of) {
        S.layers[so.first].ent_off = (uint32_t)(S.atab.size() + S.side.size());
        S.side.insert(S.side.end(), so.second.begin(), so.second.end());
    }
    S.ok = true;
    S.noprev_shared = bad_noprev;
    return S;
}

}  // namespace s2
