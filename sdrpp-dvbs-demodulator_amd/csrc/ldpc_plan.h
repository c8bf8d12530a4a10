// Host-side construction of the device "plan" for one DVB-S2 LDPC code.
//
// The reference decoder (xdsopl-ldpc-pabr/layered_decoder.hh:46-74) sweeps the parity-check rows strictly
// in the order layer i = 0..q-1, row j = 0..359.  Rows of one layer are NOT always independent: when a
// table row has two (or more) addresses with the same residue mod q, rows j and j+d of that layer share
// an information bit, and the later row must see the earlier row's write (SURVEY section 7 hard part 1).
// To stay bit-exact while running the 360 rows of a layer on 360 lanes, the plan marks for every
// (layer, row, shared link):
//   late  : an earlier row of this layer also touches the bit  -> read only after that row has written
//   early : a later row of this layer also touches the bit     -> write before that row reads
// and assigns each row a level = 1 + max(level of the rows it waits for).  Rows of equal level are
// independent; levels are separated by a workgroup barrier (ldpc_kernel.hip).  Inside a layer the links
// are reordered so that shared ("conflict") links come first: the per-level work only walks those.
// Link order inside a row does not influence the result (min/xor are order-free, algorithms.hh:233-256).
#pragma once
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#include "ldpc_qc_tables.inc"

namespace s2 {

struct LdpcLayerDesc {   // 16 bytes = one s_load_dwordx4 (all fields 32-bit: sub-dword fields would force vector loads)
    uint32_t ent_off;    // index of the layer's first link entry
    uint32_t deg;        // bits 0..15: information-bit links per row in this layer; bits 16..31: chain step d (0 = none, LDPC_WALK_MARK = quad-walk layer):
                         // d != 0 marks a layer whose only shared links are one pair (links 0 = "E", 1 = "L") with row j's
                         // E-bit == row (j+d)'s L-bit: its dependency chains j, j+d, j+2d.. are walked by d lanes with the
                         // shared posterior forwarded in a register (ldpc_kernel.hip, chain walk)
    uint32_t depth_nc;   // bits 0..15 depth (1 = conflict-free, else number of levels), bits 16..31 nc = number of
                         // conflict links (they are links 0..nc-1)
    uint32_t row_off;    // index of the layer's first per-row word (conflict layers only)
};
// link entry: bits 0..15 = sp = (360 - s) % 360, bits 16..31 = r.
//   lane j reads byte 360*r + ((j + sp) mod 360) of the posterior array.
// per-row word (conflict layers): bits 0..7 level, 8..19 late mask, 20..31 early mask (over links 0..nc-1)

constexpr int LDPC_MAX_CONFLICT_LINKS = 12;
#ifndef LDPC_CHAIN_MAX_D
#define LDPC_CHAIN_MAX_D 180  // chain walk for every single shared pair (step d = 1..180, d lanes); 0 disables it
#endif

#ifndef LDPC_WALK_MIN_DEPTH
#define LDPC_WALK_MIN_DEPTH 4     // quad-walk layers: at least this many levels (and cheaper than the per-level barriers by the cost model below)
#endif
constexpr uint32_t LDPC_WALK_MARK = 0xfffeu;   // chain-step field of a quad-walk layer

// which kernels (by maximum row degree, regular codes only) take their link addresses from the address table (ldpc_kernel.hip: LDPC_ADDR_TABLE)
constexpr bool ldpc_atab_degree(int max_deg) { return max_deg == 2 || max_deg == 8 || max_deg == 12; }

struct LdpcPlan {
    int code_index = -1;
    int N = 0, K = 0, R = 0, q = 0, max_deg = 0, min_deg = 0, edges = 0;
    int rec_dwords = 0;  // message record size per row, dwords (power of two >= ceil((max_deg+2)/4))
    int sum_depth = 0;   // sum of layer depths (q when no layer has conflicts)
    int conflict_layers = 0;
    std::vector<LdpcLayerDesc> layers;
    std::vector<uint32_t> ents;   // per layer `deg` words sp | r<<16, then (from pent_base on) the pair-format table
    int pent_base = 0;            // pair table: per layer (max_deg+1)/2 pairs x 2 words {spA | spB<<16, 360*rA | 360*rB<<16}
    std::vector<uint32_t> rows;
    int synd_base = 0;            // syndrome-check table (in ents[]): [max_deg + 2][q * 6] words, one per (link, layer, 64-row word), see below
    std::vector<uint32_t> atab;   // address table [q][384 rows][(max_deg + 1) / 2, padded to 1 / 2 / 4 / 8 words]: byte offsets of the row's two links of a pair inside the posterior array,
                                  // a0 | a1 << 16 with a = 360 r + (j + sp) mod 360 (regular codes of the degrees ldpc_atab_degree names; ldpc_kernel.hip, LDPC_ADDR_TABLE)
};

inline LdpcPlan build_ldpc_plan(int code_index) {
    const QcCodeDesc& d = QC_CODES[code_index];
    LdpcPlan P;
    P.code_index = code_index;
    P.N = d.N; P.K = d.K; P.R = d.N - d.K; P.q = d.q; P.max_deg = d.max_deg; P.edges = d.edges;
    P.min_deg = d.max_deg;
    int rd = (d.max_deg + 2 + 3) / 4, pw = 1;
    while (pw < rd) pw <<= 1;
    P.rec_dwords = pw;
    for (int i = 0; i < d.q; ++i) {
        LdpcLayerDesc L;
        L.ent_off = (uint32_t)P.ents.size();
        L.deg = (uint32_t)(d.off[i + 1] - d.off[i]);
        L.row_off = 0;
        uint32_t l_depth = 1, l_nc = 0;
        P.min_deg = std::min(P.min_deg, (int)(L.deg & 0xffff));
        std::vector<int> rr, ss;
        for (int e = d.off[i]; e < d.off[i + 1]; ++e) {
            rr.push_back((int)(d.ent[e] >> 16));
            ss.push_back((int)(d.ent[e] & 0xffff));
        }
        // links sharing a table row r touch the same 360 bits -> conflict links, moved to the front
        std::map<int, int> cnt;
        for (int r : rr) cnt[r]++;
        std::vector<int> order;
        for (int k = 0; k < (int)rr.size(); ++k)
            if (cnt[rr[k]] > 1) order.push_back(k);
        l_nc = (uint32_t)order.size();
        for (int k = 0; k < (int)rr.size(); ++k)
            if (cnt[rr[k]] == 1) order.push_back(k);
        std::vector<int> r2, s2v;
        for (int k : order) { r2.push_back(rr[k]); s2v.push_back(ss[k]); }
        rr = r2; ss = s2v;
        uint32_t chain_d = 0;
        if (l_nc == 2 && rr[0] == rr[1]) {
            // row j's link-A bit (j + spA) equals row j' link-B bit (j' + spB)  <=>  j' = j + (spA - spB) mod 360
            int spA = (360 - ss[0]) % 360, spB = (360 - ss[1]) % 360;
            int DA = ((spA - spB) % 360 + 360) % 360;
            int d = std::min(DA, 360 - DA);
            if (d >= 1 && d <= LDPC_CHAIN_MAX_D) {
                chain_d = (uint32_t)d;
                if (DA != d) { std::swap(rr[0], rr[1]); std::swap(ss[0], ss[1]); }   // link 0 = E (partner j + d), link 1 = L (partner j - d)
            }
        }
        for (int k = 0; k < (int)rr.size(); ++k) P.ents.push_back((uint32_t)((360 - ss[k]) % 360) | ((uint32_t)rr[k] << 16));
        if (l_nc > 0) {
            std::vector<uint32_t> late(360, 0), early(360, 0), level(360, 1);
            std::vector<std::vector<int>> preds(360);
            std::map<int, std::vector<int>> byr;
            for (int k = 0; k < (int)l_nc; ++k) byr[rr[k]].push_back(k);
            for (auto& kv : byr)
                for (int m = 0; m < 360; ++m) {
                    // touchers of bit 360*r+m as (row, link); sequential order = ascending row
                    std::vector<std::pair<int, int>> t;
                    for (int k : kv.second) t.push_back({(ss[k] + m) % 360, k});
                    std::sort(t.begin(), t.end());
                    for (size_t a = 0; a < t.size(); ++a) {
                        if (a > 0) { late[t[a].first] |= 1u << t[a].second; preds[t[a].first].push_back(t[a - 1].first); }
                        if (a + 1 < t.size()) early[t[a].first] |= 1u << t[a].second;
                    }
                }
            uint32_t depth = 1;
            for (int j = 0; j < 360; ++j) {  // predecessors always have a smaller row index
                uint32_t lv = 1;
                for (int p : preds[j]) lv = std::max(lv, level[p] + 1);
                level[j] = lv;
                depth = std::max(depth, lv);
            }
            l_depth = depth;
            L.row_off = (uint32_t)P.rows.size();
            for (int j = 0; j < 360; ++j) P.rows.push_back(level[j] | (late[j] << 8) | (early[j] << 20));
            P.conflict_layers++;
        }
        // "quad walk" layers: at most 4 shared links and a deep, narrow level structure (few rows per level, e.g. B7 layer 42: 33 levels of
        // 11 rows).  A workgroup barrier per level costs ~900 cycles with one or two waves working; instead ONE wave walks the rows of
        // levels >= 2 in level order, four lanes per row (lane = shared link), 16 rows per step, ordered by the in-order LDS pipeline
        // (ldpc_kernel.hip, KIND 6).  The step list follows the layer's 360 row words: [number of steps | lanes per row << 16][steps x 16 row indices, ~0 = none].
        const int walk_lpr = l_nc <= 4 ? 4 : 8;       // lanes per row: the kernels for degree > 12 also walk layers with up to 8 shared links
        if (l_nc > 0 && (l_nc <= 4 || (l_nc <= 8 && d.max_deg > 12)) && chain_d == 0 && (int)l_depth >= LDPC_WALK_MIN_DEPTH) {
            const size_t per_step = 64 / walk_lpr;
            std::vector<std::vector<uint32_t>> by_level(l_depth + 1);
            for (int j = 0; j < 360; ++j) by_level[P.rows[L.row_off + j] & 0xffu].push_back((uint32_t)j);
            size_t widest = 0;
            for (uint32_t lv = 2; lv <= l_depth; ++lv) widest = std::max(widest, by_level[lv].size());
            // cost model in cycles (per-layer profiles on MI355X, tools/ldpc_prof.py): a level of the owner-lane code ~930 (<= 4 shared links) or
            // ~1600 (more), a walker step ~445 (4 lanes per row) or ~600 (8), ~1500 for the hand-off and its two barriers
            size_t steps = 0;
            for (uint32_t lv = 2; lv <= l_depth; ++lv) steps += (by_level[lv].size() + per_step - 1) / per_step;
            const double cost_walk = 1500.0 + (double)steps * (walk_lpr == 4 ? 445.0 : 600.0);
            const double cost_levels = (double)(l_depth - 1) * (l_nc <= 4 ? 930.0 : 1600.0);
            (void)widest;
            if (cost_walk < 0.9 * cost_levels) {
                std::vector<uint32_t> list;
                for (uint32_t lv = 2; lv <= l_depth; ++lv)
                    for (size_t o = 0; o < by_level[lv].size(); o += per_step)
                        for (size_t i = 0; i < 16; ++i) list.push_back(i < per_step && o + i < by_level[lv].size() ? by_level[lv][o + i] : 0xffffffffu);
                P.rows.push_back((uint32_t)(list.size() / 16) | ((uint32_t)walk_lpr << 16));
                list.insert(list.end(), 48, 0xffffffffu);        // three empty steps: the walker fetches ahead without a bound test
                P.rows.insert(P.rows.end(), list.begin(), list.end());
                chain_d = LDPC_WALK_MARK;
            }
        }
        L.deg |= chain_d << 16;
        L.depth_nc = l_depth | (l_nc << 16);
        P.sum_depth += (int)l_depth;
        P.layers.push_back(L);
    }
    // pair-format copy of the link table for the packed-uint16 address arithmetic of the kernel (absent links: zeros)
    P.pent_base = (int)P.ents.size();
    const int npt = (P.max_deg + 1) / 2;
    for (int i = 0; i < d.q; ++i) {
        const LdpcLayerDesc& L = P.layers[i];
        const int deg = (int)(L.deg & 0xffffu);
        for (int p = 0; p < npt; ++p) {
            uint32_t w0 = 0, w1 = 0;
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * p + h;
                if (k < deg) {
                    const uint32_t e = P.ents[L.ent_off + k];
                    w0 |= (e & 0xffffu) << (16 * h);
                    w1 |= ((e >> 16) * 360u) << (16 * h);
                }
            }
            P.ents.push_back(w0);
            P.ents.push_back(w1);
        }
    }
    // syndrome-check table for the bit-vector form of LDPCDecoder::bad (ldpc_kernel.hip, syndromes_bad): task t = 6*layer + w covers rows
    // 64w .. 64w+63 of a layer; entry [k][t] names the 64 sign bits link k contributes: bits 0..15 = dword index inside the packed sign
    // scratch (14 dwords per 360-bit group: 14*g + (a >> 5)), bits 16..20 = a & 31 with a = (64w + sp) mod 360, bit 30 = drop bit 0
    // (row 0 of layer 0 has no previous parity bit), bit 31 = present.  k < max_deg: information links, then own parity, previous parity.
    P.synd_base = (int)P.ents.size();
    {
        const int ntask = d.q * 6, pg0 = d.K / 360;
        auto entry = [](int g, int a, bool drop0) { return (uint32_t)(14 * g + (a >> 5)) | ((uint32_t)(a & 31) << 16) | (drop0 ? 1u << 30 : 0u) | (1u << 31); };
        for (int k = 0; k < d.max_deg + 2; ++k)
            for (int t = 0; t < ntask; ++t) {
                const int layer = t / 6, w = t % 6;
                const LdpcLayerDesc& L = P.layers[layer];
                const int deg = (int)(L.deg & 0xffffu);
                uint32_t e = 0;
                if (k < d.max_deg) {
                    if (k < deg) {
                        const uint32_t en = P.ents[L.ent_off + k];
                        e = entry((int)(en >> 16), (64 * w + (int)(en & 0xffffu)) % 360, false);
                    }
                } else if (k == d.max_deg) {
                    e = entry(pg0 + layer, 64 * w, false);
                } else {
                    if (layer) e = entry(pg0 + layer - 1, 64 * w, false);
                    else e = entry(pg0 + d.q - 1, (64 * w + 359) % 360, w == 0);
                }
                P.ents.push_back(e);
            }
    }
    if (ldpc_atab_degree(P.max_deg) && P.min_deg == P.max_deg) {
        const int npi = (P.max_deg + 1) / 2;
        const int stride = npi <= 1 ? 1 : npi <= 2 ? 2 : npi <= 4 ? 4 : 8;      // (ldpc_kernel.hip: ldpc_atab_stride)
        P.atab.assign((size_t)d.q * 384 * stride, 0);
        for (int i = 0; i < d.q; ++i) {
            const LdpcLayerDesc& L = P.layers[i];
            for (int j = 0; j < 360; ++j)
                for (int k = 0; k < P.max_deg; ++k) {
                    const uint32_t e = P.ents[L.ent_off + k];
                    const uint32_t a = 360u * (e >> 16) + ((uint32_t)j + (e & 0xffffu)) % 360u;
                    P.atab[((size_t)i * 384 + j) * stride + k / 2] |= a << (16 * (k & 1));
                }
        }
    }
    return P;
}

}  // namespace s2
