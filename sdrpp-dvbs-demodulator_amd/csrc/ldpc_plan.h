// Host-side construction of the device "plan" for one DVB-S2 LDPC code.
//
// The reference decoder (xdsopl-ldpc-pabr/layered_decoder.hh:46-74) sweeps the parity-check rows strictly
// in the order layer i = 0..q-1, row j = 0..359.  Rows of one layer are NOT always independent: when a
// table row has two (or three) addresses with the same residue mod q, the rows j and j+d of that layer
// share an information bit, and the later row must see the earlier row's write (SURVEY section 7 hard
// part 1).  To stay bit-exact while running the 360 rows of a layer on 360 lanes, the plan marks for
// every (layer, row, link):
//   late  : an earlier row of this layer also touches the bit  -> read only after that row has written
//   early : a later row of this layer also touches the bit     -> write before that row reads
// and assigns each row a level = 1 + max(level of the rows it waits for).  Rows of equal level are
// independent; levels are separated by a workgroup barrier (see ldpc_kernel.hip).
#pragma once
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#include "ldpc_qc_tables.inc"

namespace s2 {

struct LdpcLayerDesc {   // 16 bytes, wave-uniform, fetched with scalar loads
    uint32_t ent_off;    // index of the layer's first link entry
    uint16_t deg;        // information-bit links per row in this layer
    uint16_t depth;      // 1 = conflict-free, else number of levels
    uint32_t row_off;    // index (in rows) of the layer's RowInfo block, conflict layers only
    uint32_t cmask;      // bit k set: link k is shared between rows of this layer
};
struct LdpcLinkEnt {     // 8 bytes, wave-uniform
    uint32_t sb;         // 360*r + (360 - s) % 360 : lane j reads byte sb + j, minus 360 if >= thr
    uint32_t thr;        // 360*r + 360
};
struct LdpcRowInfo {     // 12 bytes per (conflict layer, row)
    uint32_t late;       // bit k: link k must be read after an earlier row's write
    uint32_t early;      // bit k: link k must be written before a later row's read
    uint32_t level;      // 1..depth
};

struct LdpcPlan {
    int code_index = -1;
    int N = 0, K = 0, R = 0, q = 0, max_deg = 0, edges = 0;
    int rec_dwords = 0;  // message record size per row, dwords (power of two >= ceil((max_deg+2)/4))
    int sum_depth = 0;   // sum of layer depths (q when no layer has conflicts)
    std::vector<LdpcLayerDesc> layers;
    std::vector<LdpcLinkEnt> ents;
    std::vector<LdpcRowInfo> rows;
};

inline LdpcPlan build_ldpc_plan(int code_index) {
    const QcCodeDesc& d = QC_CODES[code_index];
    LdpcPlan P;
    P.code_index = code_index;
    P.N = d.N; P.K = d.K; P.R = d.N - d.K; P.q = d.q; P.max_deg = d.max_deg; P.edges = d.edges;
    int rd = (d.max_deg + 2 + 3) / 4, pw = 1;
    while (pw < rd) pw <<= 1;
    P.rec_dwords = pw;
    for (int i = 0; i < d.q; ++i) {
        LdpcLayerDesc L;
        L.ent_off = (uint32_t)P.ents.size();
        L.deg = (uint16_t)(d.off[i + 1] - d.off[i]);
        L.depth = 1; L.row_off = 0; L.cmask = 0;
        std::vector<int> rr, ss;
        for (int e = d.off[i]; e < d.off[i + 1]; ++e) {
            int r = d.ent[e] >> 16, s = d.ent[e] & 0xffff;
            rr.push_back(r); ss.push_back(s);
            LdpcLinkEnt E;
            E.sb = (uint32_t)(360 * r + (360 - s) % 360);
            E.thr = (uint32_t)(360 * r + 360);
            P.ents.push_back(E);
        }
        // links sharing a table row r touch the same 360 bits
        std::map<int, std::vector<int>> byr;
        for (int k = 0; k < (int)rr.size(); ++k) byr[rr[k]].push_back(k);
        bool conflict = false;
        for (auto& kv : byr)
            if (kv.second.size() > 1) {
                conflict = true;
                for (int k : kv.second) L.cmask |= 1u << k;
            }
        if (conflict) {
            std::vector<LdpcRowInfo> ri(360);
            for (auto& x : ri) { x.late = 0; x.early = 0; x.level = 1; }
            // preds[j] = rows that must finish their shared-link write before row j reads
            std::vector<std::vector<int>> preds(360);
            for (auto& kv : byr) {
                if (kv.second.size() < 2) continue;
                for (int m = 0; m < 360; ++m) {
                    // touchers of bit 360*r+m: (row, link), sequential order = ascending row
                    std::vector<std::pair<int, int>> t;
                    for (int k : kv.second) t.push_back({(ss[k] + m) % 360, k});
                    std::sort(t.begin(), t.end());
                    for (size_t a = 0; a < t.size(); ++a) {
                        if (a > 0) { ri[t[a].first].late |= 1u << t[a].second; preds[t[a].first].push_back(t[a - 1].first); }
                        if (a + 1 < t.size()) ri[t[a].first].early |= 1u << t[a].second;
                    }
                }
            }
            int depth = 1;
            for (int j = 0; j < 360; ++j) {  // preds always have a smaller row index
                uint32_t lv = 1;
                for (int p : preds[j]) lv = std::max(lv, ri[p].level + 1);
                ri[j].level = lv;
                depth = std::max(depth, (int)lv);
            }
            L.depth = (uint16_t)depth;
            L.row_off = (uint32_t)P.rows.size();
            P.rows.insert(P.rows.end(), ri.begin(), ri.end());
        }
        P.sum_depth += L.depth;
        P.layers.push_back(L);
    }
    return P;
}

}  // namespace s2
