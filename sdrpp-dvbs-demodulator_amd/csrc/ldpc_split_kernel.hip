// Layered offset-min-sum LDPC decoder for DVB-S2, gfx950 -- HALF-ROW mapping: two lanes per parity-check row, one frame per workgroup.
//
// Replaces BBFrameLDPC::decode (reference src/demod/dvbs2/codings/bbframe_ldpc.cpp:123-139) and the library under it
// (xdsopl-ldpc-pabr/layered_decoder.hh:23-133, algorithms.hh:206-277), bit-exact, like ldpc_kernel.hip, whose schedule (ldpc_plan.h) and
// arithmetic (packed int16 "Q8" with the int8 saturation rules) it shares.  What differs is the mapping onto the machine:
//   * ldpc_kernel.hip gives a lane a whole row (up to 30 links) and a workgroup two frames: 12 waves per compute unit, each with a
//     ~300-instruction stream per layer between two barriers -- the decoder waits half of its resident cycles (profiles/r04_ldpc_pmc.txt).
//   * here thread t of a 768-thread workgroup holds half h = t & 1 of row j = t >> 1 (ldpc_split_plan.h: HS = (max_deg + 2) / 2 link slots
//     per half, shared links and the chain / walk / level machinery in half 0, the parity bits in half 1).  The halves join their
//     (min0, min1, sign) with one DPP quad_perm step -- min and xor are associative, algorithms.hh:242-255 stays exact.  One frame per
//     workgroup = 64.8 KB of posteriors in LDS, TWO workgroups per compute unit = 24 waves, each with half the stream; the two frames of a
//     compute unit are no longer in lockstep, so one frame's serial sections (chain walks, deep layers) run beside the other's wide ones.
//   * no address arithmetic in a layer: every slot's LDS byte offset, parity bits included, comes from the per-thread address table,
//     fetched a layer ahead together with the 8-byte message record and the row word.
// Roofline: algorithmic bytes per frame = iters*4*edges + N + K/8 against HBM 8 TB/s is the NOMINAL figure: the state is
// on-chip (LDS + Infinity Cache), what bounds the kernel is the latency of a layer's barrier-separated phases.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_lane_common.h"
#include "ldpc_split_plan.h"

namespace s2 {

#ifndef LDPC_SPLIT_EXP
#define LDPC_SPLIT_EXP 0          // development switch for TIMING experiments (results wrong): 1 = every layer runs as a conflict-free one, 2 = no table / record traffic in the layer loop, 4 = no layer barrier, 8 = no output phase
#endif
#ifndef LDPC_SPLIT_WPE
#define LDPC_SPLIT_WPE 6          // waves per SIMD the register allocation aims at (6 = 80 VGPRs: two workgroups per compute unit; 8 = 64: room for a 128-register front-end wave beside them)
#endif

template <int MAXDEG>
struct SplitShape {
    static constexpr int NL = MAXDEG + 2, HS = NL / 2, NP = (HS + 1) / 2;
    static constexpr int NPW = NP <= 1 ? 1 : NP <= 2 ? 2 : NP <= 4 ? 4 : 8;
    static constexpr int REC = HS <= 4 ? 1 : HS <= 8 ? 2 : 4;
    static_assert((NL & 1) == 0, "the half-row decoder takes rows with an even number of links");
};

__device__ __forceinline__ void lds_read_lo_i8(uint32_t a_lo, uint32_t& r_lo) {
    if (LDPC_EXP & 4) { r_lo = a_lo & 0xffu; return; }
    asm volatile("ds_read_u8_d16 %0, %1" : "=&v"(r_lo) : "v"(a_lo) : "memory");
}
#define LDS_READY_CASE(n) case n: asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r_lo), "+v"(r_hi) : : "memory"); break
__device__ __forceinline__ void lds_ready_n(int outstanding, uint32_t& r_lo, uint32_t& r_hi) {   // (constant after unrolling)
    switch (outstanding) {
        LDS_READY_CASE(0); LDS_READY_CASE(1); LDS_READY_CASE(2); LDS_READY_CASE(3); LDS_READY_CASE(4); LDS_READY_CASE(5); LDS_READY_CASE(6); LDS_READY_CASE(7);
        LDS_READY_CASE(8); LDS_READY_CASE(9); LDS_READY_CASE(10); LDS_READY_CASE(11); LDS_READY_CASE(12); LDS_READY_CASE(13); LDS_READY_CASE(14);
        default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(r_lo), "+v"(r_hi) : : "memory"); break;
    }
}
#undef LDS_READY_CASE

template <int NPW>
__device__ __forceinline__ void words_load(uint32_t (&w)[NPW], const uint32_t* __restrict__ p) {
    if constexpr (NPW == 1) { w[0] = p[0]; }
    else if constexpr (NPW == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y; }
    else {
#pragma unroll
        for (int i = 0; i < NPW; i += 4) { const uint4 v = *reinterpret_cast<const uint4*>(p + i); w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w; }
    }
}

#define QUAD_DPP(x_, ctrl) __builtin_amdgcn_update_dpp(0, (int)(x_), (ctrl), 0xf, 0xf, true)
constexpr int DPP_SWAP_HALVES = 0xB1;    // quad_perm [1,0,3,2]: the other half of the row
constexpr int DPP_FROM_HALF0 = 0xA0;     // quad_perm [0,0,2,2]: half 0's value in both lanes of a row

// One sweep step for one layer.  KIND 0: no shared bits in the layer; 1: one shared pair resolved by the chain walk; 3: levels (at most 4 shared
// links); 6: quad walk (ldpc_kernel.hip / ldpc_plan.h describe the kinds; the middle sections below are theirs, run by the half-0 lanes).
template <int MAXDEG, int KIND>
__device__ __forceinline__ void split_layer(const uint32_t lbase, int8_t* __restrict__ post, const uint32_t* __restrict__ ents, const uint32_t (&AD)[SplitShape<MAXDEG>::NPW],
                                            const LdpcLayerDesc L, const uint32_t rowword, const int layer, const int t, const bool active,
                                            const uint32_t (&rec_in)[SplitShape<MAXDEG>::REC], uint32_t (&rec_out)[SplitShape<MAXDEG>::REC],
                                            uint32_t* __restrict__ cw, uint8_t* __restrict__ cres, const uint32_t* __restrict__ walk) {
    using S = SplitShape<MAXDEG>;
    constexpr int HS = S::HS, NP = S::NP, REC = S::REC;
    constexpr bool CONF = KIND != 0;
    constexpr int MAXC0 = KIND == 1 ? 2 : 4;
    constexpr int MAXC = MAXC0 < HS ? MAXC0 : HS;
    const int j = t >> 1;
    const bool half1 = (t & 1) != 0;
    const bool act0 = active && !half1;
    s16x2 V[NP], G[NP];
    uint32_t addr[HS];
    const int nc = CONF ? (int)(L.depth_nc >> 16) : 0;
    const uint32_t level = rowword & 0xffu, late = (rowword >> 8) & 0xfffu, early = rowword >> 20;
    const bool noprev = (t == 1) && (layer == 0);         // row 0 of layer 0 has no previous parity bit (slot HS-1 of half 1)
#define LINK_IN(k) ((int)V[(k) >> 1][(k) & 1] >> 8)
#define LINK_MG(k) ((int)G[(k) >> 1][(k) & 1] >> 8)
#define LINK_SET(k, v, m) do { V[(k) >> 1][(k) & 1] = (short)((v) << 8); G[(k) >> 1][(k) & 1] = (short)((m) << 8); } while (0)
    // ---- input phase: every lane (idle lanes read the scratch byte and never store)
    uint32_t XR[NP], XH[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t a_lo = lbase + (AD[p] & 0xffffu), a_hi = lbase + (AD[p] >> 16);
        addr[2 * p] = a_lo;
        if (2 * p + 1 < HS) { addr[2 * p + 1] = a_hi; lds_read_pair_i8(a_lo, a_hi, XR[p], XH[p]); }
        else { lds_read_lo_i8(a_lo, XR[p]); XH[p] = 0; }
    }
    s16x2 MIN0 = splat2(Q8_NONE), MIN1 = splat2(Q8_NONE);
    uint32_t SX = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        // LDS returns in order: pair p is complete once at most HS - (reads up to and including it) operations are outstanding
        const int issued = 2 * p + 2 < HS ? 2 * p + 2 : HS;
        lds_ready_n(HS - issued, XR[p], XH[p]);
        // byte of the low load -> bits 15:8, byte of the high load (it sits in bits 23:16) -> bits 31:24
        const s16x2 X = from_bits2(__builtin_amdgcn_perm(XH[p], XR[p], 0x060c000cu));
        s16x2 v = sat_sub2(X, rec_pair<REC>(rec_in, 2 * p));                     // int8 saturation by the 16-bit clamp
        const s16x2 av = pmax2(v, sat_sub2(splat2(0), v));
        // |v| - 1 clamped at 0 (ldpc_kernel.hip: no upper clamp needed, only the high byte is ever consumed)
        s16x2 g = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, av), (u16x2){256, 256}));
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int k = 2 * p + hh;
            if (k >= HS) { v[hh] = 0; g[hh] = (short)Q8_NONE; }
            if (k == HS - 1 && noprev) { v[hh] = 0; g[hh] = (short)Q8_NONE; }
            if constexpr (CONF) {
                if (k < MAXC && k < nc && ((late >> k) & 1)) { v[hh] = 0; g[hh] = (short)Q8_NONE; }   // joins the totals at its level (half 1: no late links)
            }
        }
        V[p] = v; G[p] = g;
        if (p == 0) { MIN0 = g; }
        else if (p == 1) { MIN1 = pmax2(MIN0, g); MIN0 = pmin2(MIN0, g); }
        else { MIN1 = pmin2(MIN1, pmax2(MIN0, g)); MIN0 = pmin2(MIN0, g); }
        SX ^= bits2(v);
    }
    // even / odd slots joined: both halves of every word then hold the lane's value ...
    int M0, M1, SXs;
    {
        const s16x2 R0 = from_bits2(__builtin_amdgcn_alignbit(bits2(MIN0), bits2(MIN0), 16));
        const s16x2 R1 = from_bits2(__builtin_amdgcn_alignbit(bits2(MIN1), bits2(MIN1), 16));
        M0 = (int)bits2(pmin2(MIN0, R0));
        M1 = (int)bits2(pmin2(pmax2(MIN0, R0), pmin2(MIN1, R1)));
        SXs = (int)(SX ^ __builtin_amdgcn_alignbit(SX, SX, 16));
    }
    // ... and the two halves of the row: a word with equal halves orders like its 16-bit value under the 32-bit signed compare
    {
        const int o0 = QUAD_DPP(M0, DPP_SWAP_HALVES), o1 = QUAD_DPP(M1, DPP_SWAP_HALVES), os = QUAD_DPP(SXs, DPP_SWAP_HALVES);
        M1 = min(max(M0, o0), min(M1, o1));
        M0 = min(M0, o0);
        SXs ^= os;
    }
    s16x2 MIN1CB, SUMCB;
    uint32_t SXB;
    if constexpr (!CONF) {
        MIN1CB = pmin2(from_bits2((uint32_t)M1), q8(32));
        SUMCB = pmin2(from_bits2((uint32_t)M0), q8(32)) + MIN1CB;
        SXB = (uint32_t)SXs;
    } else {
        int min0 = M0 >> 24, min1 = M1 >> 24, sx = SXs;                 // (sx: the sign of the row's product sits in bit 31)
        const int chain_d = (int)(L.deg >> 16);
        if constexpr (KIND == 1) {
            // ---- chain walk (ldpc_kernel.hip: single shared pair, links 0 = E, 1 = L)
            if (act0) {
                if (level == 1u) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if ((early >> k) & 1) {
                            int nm = new_msg(LINK_IN(k), LINK_MG(k), min0, min1, sx);
                            LDS_I8(addr[k]) = (int8_t)clamp8(LINK_IN(k) + nm);
                        }
                    }
                }
                if ((late >> 1) & 1) {
                    const int qE = (late & 1u) ? 255 : ((LINK_MG(0) == min0) ? min1 : min0);
                    const ChainRec r = chain_record(rec_byte<REC>(rec_in, 1), qE, LINK_IN(0), (sx ^ LINK_IN(0)) >> 31);
                    reinterpret_cast<uint2*>(cw)[j] = make_uint2(r.lim, r.se);
                }
            }
            lds_barrier();
            if (t < chain_d) {
                // lane c walks rows c + k*d (ldpc_kernel.hip)
                __builtin_amdgcn_s_setprio(3);
                const uint32_t eL = ents[1];
                const int T = 359 / chain_d;
                int x = post[link_addr(eL, t + chain_d)];
                const uint2* c = reinterpret_cast<const uint2*>(cw) + t + chain_d;
                uint8_t* pr = reinterpret_cast<uint8_t*>(cres) + t + chain_d;
                uint2 ra = c[0], rb = c[chain_d];
                c += 2 * chain_d;
                int k = 1;
                for (; k + 2 <= T; k += 2) {
                    const uint2 na = c[0];
                    pr[0] = (uint8_t)x;
                    x = chain_step(x, ra.x, ra.y);
                    ra = na;
                    const uint2 nb = c[chain_d];
                    pr[chain_d] = (uint8_t)x;
                    x = chain_step(x, rb.x, rb.y);
                    rb = nb;
                    c += 2 * chain_d;
                    pr += 2 * chain_d;
                }
                if (k < T) {
                    pr[0] = (uint8_t)x;
                    x = chain_step(x, ra.x, ra.y);
                    pr += chain_d;
                }
                if (t + T * chain_d < 360) pr[0] = (uint8_t)x;
                __builtin_amdgcn_s_setprio(0);
            }
            lds_barrier();
            if (act0) {
                const int xL = (int)(int8_t)cres[j], xE = (int)LDS_I8(addr[0]);
                if ((late >> 1) & 1) {
                    int v = clamp8(xL - rec_byte<REC>(rec_in, 1));
                    int m = mag_of(v);
                    LINK_SET(1, v, m);
                    ROW_ACCUM(v, m);
                }
                if (late & 1u) {
                    int v = clamp8(xE - rec_byte<REC>(rec_in, 0));
                    int m = mag_of(v);
                    LINK_SET(0, v, m);
                    ROW_ACCUM(v, m);
                }
            }
        } else if constexpr (KIND == 6) {
            // ---- quad walk (ldpc_kernel.hip, ldpc_plan.h): four lanes of wave 0 per row of levels >= 2
            (void)chain_d;
            constexpr int CWD = 2;
            const uint32_t whd = walk[0];
            const int wk_steps = (int)(whd & 0xffffu);
            if (act0) {
                if (level == 1u) {
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        if (k < nc && ((early >> k) & 1)) {
                            int nm = new_msg(LINK_IN(k), LINK_MG(k), min0, min1, sx);
                            LDS_I8(addr[k]) = (int8_t)clamp8(LINK_IN(k) + nm);
                        }
                    }
                } else {
                    uint32_t lb = 0;
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        const int b = (k < nc && ((late >> k) & 1)) ? rec_byte<REC>(rec_in, k) : LINK_IN(k);
                        lb |= ((uint32_t)b & 0xffu) << (8 * k);
                    }
                    const uint32_t hd = (uint32_t)min(min0, 127) | ((uint32_t)min(min1, 127) << 7) | (((uint32_t)sx >> 31) << 14) | ((late & 0xffu) << 15) | ((early & 0xffu) << 23);
                    cw[CWD * j] = hd;
                    cw[CWD * j + 1] = lb;
                }
            }
            lds_barrier();
            if (t < 64) {
                __builtin_amdgcn_s_setprio(3);
                const int k = t & 3, qd = t >> 2;
                const uint32_t ek = ents[k < nc ? k : 0];
                const int spk = (int)(ek & 0xffffu);
                const uint32_t basek = lbase + 360u * (ek >> 16);
                const uint32_t scratch = lds_offset(reinterpret_cast<const int8_t*>(cres)) + (uint32_t)t;
                const uint32_t cwb = lds_offset(reinterpret_cast<const int8_t*>(cw));
                const int nsteps = wk_steps;
                const uint32_t* __restrict__ list = walk + 1 + qd;
                auto step = [&](const uint32_t e) {
                    const bool valid = e != 0xffffffffu && k < nc;
                    const int row = valid ? (int)e : 0;
                    const uint32_t ra = cwb + (uint32_t)(4 * CWD) * (uint32_t)row;
                    const uint2 r = reinterpret_cast<const uint2*>(cw)[row];
                    const uint32_t hd = r.x;
                    const int b = (int)__builtin_amdgcn_sbfe((int)r.y, 8 * k, 8);
                    int tt = row + spk;
                    tt = (int)min((uint32_t)tt, (uint32_t)(tt - 360));
                    const uint32_t a = basek + (uint32_t)tt;
                    const int x = (int)LDS_I8(a);
                    const bool lt = valid && ((hd >> (15 + k)) & 1u), er = valid && ((hd >> (23 + k)) & 1u);
                    const int v = lt ? clamp8(x - b) : b;
                    const int g = mag_of(v);
                    int m0 = lt ? g : 255, m1 = 255, sg = lt ? v : 0;
#define JOIN(ctrl) do { const int o0 = QUAD_DPP(m0, ctrl), o1 = QUAD_DPP(m1, ctrl); m1 = min(max(m0, o0), min(m1, o1)); m0 = min(m0, o0); sg ^= QUAD_DPP(sg, ctrl); } while (0)
                    JOIN(0xB1);                                                              // quad_perm [1,0,3,2]
                    JOIN(0x4E);                                                              // quad_perm [2,3,0,1]
#undef JOIN
                    const int q0 = (int)(hd & 0x7fu), q1 = (int)((hd >> 7) & 0x7fu);
                    const int t1 = min(max(m0, q0), min(m1, q1)), t0 = min(m0, q0);
                    const int ss = sg ^ (int)(hd << 17);                                     // bit 31 = sign of the row's totals
                    const int nm = new_msg(v, g, t0, t1, ss);
                    LDS_I8(er ? a : scratch) = (int8_t)clamp8(v + nm);
                    LDS_I8(valid ? ra + 4u + (uint32_t)k : scratch) = (int8_t)v;
                };
#define LIST_FETCH(r, p) asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory")
#define LIST_READY(r) asm volatile("s_waitcnt vmcnt(1)" : "+v"(r) : : "memory")
                uint32_t eA, eB;
                const uint32_t* lp = list;
                LIST_FETCH(eA, lp); LIST_FETCH(eB, lp + 16);
                lp += 32;
                for (int i = 0; i < nsteps; i += 2) {
                    LIST_READY(eA);
                    step(eA);
                    LIST_FETCH(eA, lp);
                    LIST_READY(eB);
                    if (i + 1 < nsteps) step(eB);
                    LIST_FETCH(eB, lp + 16);
                    lp += 32;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef LIST_FETCH
#undef LIST_READY
                __builtin_amdgcn_s_setprio(0);
            }
            lds_barrier();
            if (act0 && level != 1u) {
                const uint32_t lb = cw[CWD * j + 1];
#pragma unroll
                for (int k = 0; k < MAXC; ++k) {
                    if (k < nc && ((late >> k) & 1)) {
                        int v = (int)__builtin_amdgcn_sbfe((int)lb, 8 * k, 8);
                        int m = mag_of(v);
                        LINK_SET(k, v, m);
                        ROW_ACCUM(v, m);
                    }
                }
            }
        } else {
            // ---- levels: a barrier per dependency level (ldpc_kernel.hip, KIND 3)
            (void)chain_d;
            const int depth = (int)(L.depth_nc & 0xffffu);
            for (int lvl = 1; lvl <= depth; ++lvl) {
                if (lvl > 1) lds_barrier();
                if (act0 && level == (uint32_t)lvl) {
                    __builtin_amdgcn_s_setprio(3);
                    if (lvl > 1) {
                        int xs[MAXC];
#pragma unroll
                        for (int k = 0; k < MAXC; ++k) xs[k] = (int)LDS_I8(addr[k]);
#pragma unroll
                        for (int k = 0; k < MAXC; ++k) {
                            if (k < nc && ((late >> k) & 1)) {
                                int v = clamp8(xs[k] - rec_byte<REC>(rec_in, k));
                                int m = mag_of(v);
                                LINK_SET(k, v, m);
                                ROW_ACCUM(v, m);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        if (k < nc && ((early >> k) & 1)) {
                            int nm = new_msg(LINK_IN(k), LINK_MG(k), min0, min1, sx);
                            LDS_I8(addr[k]) = (int8_t)clamp8(LINK_IN(k) + nm);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
            }
        }
        // the complete totals sit in half 0: hand them to half 1
        min0 = QUAD_DPP(min0, DPP_FROM_HALF0);
        min1 = QUAD_DPP(min1, DPP_FROM_HALF0);
        sx = QUAD_DPP(sx, DPP_FROM_HALF0);
        const int min0c = min(min0, 32), min1c = min(min1, 32);
        MIN1CB = q8(min1c);
        SUMCB = q8(min0c + min1c);
        SXB = (uint32_t)(sx >> 31);
    }
    // ---- output phase
    if (LDPC_SPLIT_EXP & 8) { rec_out[0] = bits2(MIN1CB) ^ bits2(SUMCB) ^ SXB; return; }
    if (active) {
        s16x2 NM[NP + 1];
        NM[NP] = splat2(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // other = (mag == min0) ? min1c : min0c, as min0c + min1c - min(mag, min1c): a magnitude is the row minimum, where min(mag, min1c) = min0c, or at
            // least the second one, where it is min1c (algorithms.hh:250-256, the selected magnitude limited to 32 once per row)
            const s16x2 other = SUMCB - pmin2(G[p], MIN1CB);
            const s16x2 neg = from_bits2(SXB ^ bits2(V[p])) >> 15;                   // 0 or -1
            s16x2 nm = pmin2(from_bits2(bits2(other) ^ bits2(neg)) - neg, q8(31));
            // new posterior: 16-bit saturating add = int8 saturation; >> 8 brings the bytes to bits 7:0 / 23:16 for the stores
            const s16x2 pn = from_bits2(bits2(sat_add2(V[p], nm)) >> 8);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int k = 2 * p + hh;
                if (k >= HS) { nm[hh] = 0; continue; }
                bool wr = true;
                if constexpr (CONF) {
                    if (k < MAXC && k < nc) wr = !((early >> k) & 1);
                }
                if (wr) {
                    if (hh == 0) lds_write_lo_i8(addr[k], bits2(pn)); else lds_write_hi_i8(addr[k], bits2(pn));
                }
            }
            NM[p] = nm;
        }
#pragma unroll
        for (int w = 0; w < REC; ++w) {
            if (2 * w < NP) rec_out[w] = __builtin_amdgcn_perm(bits2(NM[2 * w + 1 <= NP ? 2 * w + 1 : NP]), bits2(NM[2 * w]), 0x07050301u);
            else rec_out[w] = 0;
        }
        lds_pairs_wait();
    }
#undef LINK_IN
#undef LINK_MG
#undef LINK_SET
}

template <int MAXDEG>
__global__ __launch_bounds__(LDPC_SPLIT_T) __attribute__((amdgpu_waves_per_eu(LDPC_SPLIT_WPE))) void ldpc_split_kernel(const LdpcLayerDesc* __restrict__ layers, const uint32_t* __restrict__ ents,
                                                                                                                     const uint32_t* __restrict__ rows, const uint32_t* __restrict__ atab, LdpcKernelArgs A) {
    using S = SplitShape<MAXDEG>;
    constexpr int T = LDPC_SPLIT_T, REC = S::REC, NPW = S::NPW;
    extern __shared__ __attribute__((aligned(16))) int8_t lds_all[];
    const int t = threadIdx.x;
    const int N = A.N, K = A.K, R = A.R, q = A.q;
    const int npad = (N + 16 + 15) & ~15;                 // posteriors + the scratch byte (ldpc_split_plan.h)
    int8_t* __restrict__ post = lds_all;
    uint32_t* __restrict__ cw = reinterpret_cast<uint32_t*>(lds_all + npad);      // hand-off records of the chain walk / quad walk: 8 bytes per row
    uint8_t* __restrict__ cres = reinterpret_cast<uint8_t*>(cw + 2 * 360);
    int* __restrict__ s_flag = reinterpret_cast<int*>(cres + 384);                // [12] + next frame
    const uint32_t lbase = lds_offset(post);
    uint32_t* __restrict__ msg = A.msg_ws + (size_t)blockIdx.x * (size_t)q * T * REC;
    uint32_t* __restrict__ sgn = A.sgn_ws + (size_t)blockIdx.x * SGN_WS_DWORDS;
    const bool row_ok = t < 720;

    int f = blockIdx.x;
    while (f < A.nframes) {
        {
            const int8_t* __restrict__ src = A.llr + (size_t)f * N;
            for (int i = t; i < K / 8; i += T) reinterpret_cast<uint2*>(post)[i] = reinterpret_cast<const uint2*>(src)[i];
            // parity LLRs: pty[360*i + jj] = llr[K + q*jj + i]   (layered_decoder.hh:124-126)
            for (int c = t; c < R; c += T) {
                int jj = c / q, i = c - jj * q;
                post[K + 360 * i + jj] = src[K + c];
            }
            // the first sweep reads all-zero messages: this thread's records are cleared here, so that a sweep fetches them without asking which sweep it is
            uint32_t z[REC];
#pragma unroll
            for (int w = 0; w < REC; ++w) z[w] = 0;
            for (int l = 0; l < q; ++l) rec_store<REC>(z, msg + (uint32_t)l * (T * REC) + (uint32_t)t * REC);
        }
        lds_barrier();

        int it = 0, ret = 0;
        while (true) {
            const bool check = !A.force || it == A.max_trials;
            if (check) {
                const uint32_t zflag = sign_pack(post, N, reinterpret_cast<uint8_t*>(sgn), t, T);
                __syncthreads();              // the sign bytes went to global memory: full barrier (drains vmcnt)
                const bool bad = zflag != 0 || syndromes_bad<MAXDEG>(q, A.synd_base, ents, sgn, t, T);
                const unsigned long long b = __ballot(bad);
                if ((t & 63) == 0) s_flag[t >> 6] = (b != 0);
                lds_barrier();
                int any = 0;
#pragma unroll
                for (int w = 0; w < T / 64; ++w) any |= s_flag[w];
                lds_barrier();                // (the flags are rewritten by the next check)
                if (A.force) { ret = any ? -1 : A.max_trials; break; }
                if (!any) { ret = it; break; }
                if (it == A.max_trials) { ret = -1; break; }
            }
            // ---- one layered sweep (LDPCDecoder::update); records, addresses and row words travel one layer ahead, descriptors two.
            // Every fetch is (uniform base of the layer) + (this thread's 32-bit offset): scalar base registers, no 64-bit address arithmetic per lane
            uint32_t rec_next[REC], pw_next[NPW];
            rec_load<REC>(rec_next, msg + (uint32_t)t * REC);
            words_load<NPW>(pw_next, atab + (uint32_t)t * NPW);
            LdpcLayerDesc Lnext = layers[0], Lnext2 = layers[q > 1 ? 1 : 0];
            uint32_t rw_next = (rows + Lnext.row_off)[(uint32_t)t];
            for (int layer = 0; layer < q; ++layer) {
                uint32_t rec[REC], pw[NPW];
#pragma unroll
                for (int w = 0; w < REC; ++w) rec[w] = rec_next[w];
#pragma unroll
                for (int w = 0; w < NPW; ++w) pw[w] = pw_next[w];
                const LdpcLayerDesc L = Lnext;
                const uint32_t rw = rw_next;
                uint32_t* __restrict__ rp = msg + (uint32_t)layer * (T * REC);
                const int ln = layer + 1 < q ? layer + 1 : q - 1;       // (behind the last layer: the last layer's once more -- nobody reads them)
                Lnext = Lnext2;
                Lnext2 = layers[layer + 2 < q ? layer + 2 : q - 1];
                if (!(LDPC_SPLIT_EXP & 2)) {
                    words_load<NPW>(pw_next, atab + (uint32_t)ln * (T * NPW) + (uint32_t)t * NPW);
                    rec_load<REC>(rec_next, msg + (uint32_t)ln * (T * REC) + (uint32_t)t * REC);
                    rw_next = (rows + Lnext.row_off)[(uint32_t)t];
                }
                uint32_t ro[REC];
#pragma unroll
                for (int w = 0; w < REC; ++w) ro[w] = 0;
                const uint32_t* le = ents + L.ent_off;
                if ((LDPC_SPLIT_EXP & 1) || (L.depth_nc & 0xffffu) == 1) split_layer<MAXDEG, 0>(lbase, post, le, pw, L, rw, layer, t, row_ok, rec, ro, cw, cres, nullptr);
                else if ((L.deg >> 16) == LDPC_WALK_MARK) split_layer<MAXDEG, 6>(lbase, post, le, pw, L, rw, layer, t, row_ok, rec, ro, cw, cres, rows + L.row_off + T);
                else if ((L.deg >> 16) > 0) split_layer<MAXDEG, 1>(lbase, post, le, pw, L, rw, layer, t, row_ok, rec, ro, cw, cres, nullptr);
                else split_layer<MAXDEG, 3>(lbase, post, le, pw, L, rw, layer, t, row_ok, rec, ro, cw, cres, nullptr);
                // the prefetched words are claimed here, in uniform control flow and before this layer's record store is issued (ldpc_kernel.hip)
#pragma unroll
                for (int w = 0; w < REC; ++w) asm volatile("" : "+v"(rec_next[w]));
#pragma unroll
                for (int w = 0; w < NPW; ++w) asm volatile("" : "+v"(pw_next[w]));
                asm volatile("" : "+v"(rw_next));
                asm volatile("" : "+s"(Lnext2.ent_off), "+s"(Lnext2.deg), "+s"(Lnext2.depth_nc), "+s"(Lnext2.row_off));
                if (LDPC_SPLIT_EXP & 2) { asm volatile("" :: "v"(ro[0]), "v"(ro[REC - 1])); }
                else if (row_ok) rec_store<REC>(ro, rp + (uint32_t)t * REC);
                if (!(LDPC_SPLIT_EXP & 4)) lds_barrier();
            }
            ++it;
        }

        // ---- outputs
        if (t == 0) A.trials[f] = ret;
        {
            // hard decisions of [0,K): 64 bits per wave step via ballot, MSB-first bytes (module_dvbs2_demod.cpp:357-360)
            uint8_t* __restrict__ hd = A.hard + (size_t)f * A.hard_stride;
            const int lane = t & 63, wave = t >> 6;
            for (int base = wave * 64; base < K; base += (T / 64) * 64) {
                int idx = base + lane;
                int neg = (idx < K) ? (post[idx] < 0) : 0;
                unsigned long long b = __ballot(neg);
                b = __builtin_bswap64(__brevll(b));
                if (lane == 0) {
                    int nbytes = min(8, (K - base) / 8);
                    if (nbytes == 8 && ((uintptr_t)(hd + base / 8) & 7u) == 0) *reinterpret_cast<uint2*>(hd + base / 8) = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
                    else for (int n = 0; n < nbytes; ++n) hd[base / 8 + n] = (uint8_t)(b >> (8 * n));
                }
            }
            if (A.post) {
                int8_t* __restrict__ dst = A.post + (size_t)f * N;
                for (int i = t; i < K / 8; i += T) reinterpret_cast<uint2*>(dst)[i] = reinterpret_cast<const uint2*>(post)[i];
                for (int c = t; c < R; c += T) {
                    int jj = c / q, i = c - jj * q;
                    dst[K + c] = post[K + 360 * i + jj];
                }
            }
        }
        if (A.work_ctr) {
            if (t == 0) s_flag[12] = (int)(gridDim.x + atomicAdd(A.work_ctr, 1u));
            lds_barrier();
            f = s_flag[12];
        } else {
            f += gridDim.x;
        }
        lds_barrier();
    }
}

size_t ldpc_split_lds_bytes(int N) { return (size_t)((N + 16 + 15) & ~15) + 2 * 360 * 4 + 384 + 16 * 4; }
size_t ldpc_split_msg_bytes_per_block(const LdpcDeviceCode& C) { return (size_t)C.q * LDPC_SPLIT_T * C.split_rec_dwords * sizeof(uint32_t); }

template <int MAXDEG>
static hipError_t launch_split(const LdpcDeviceCode& C, const LdpcKernelArgs& A, int grid, hipStream_t stream) {
    const size_t lds = ldpc_split_lds_bytes(A.N);
    auto kern = ldpc_split_kernel<MAXDEG>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_SPLIT_T), lds, stream, C.d_split_layers, C.d_ents, C.d_split_rows, C.d_split_atab, A);
    return hipGetLastError();
}
template <int MAXDEG>
static int occupancy_split(int N) {
    int nb = 0;
    const size_t lds = ldpc_split_lds_bytes(N);
    auto kern = ldpc_split_kernel<MAXDEG>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, LDPC_SPLIT_T, lds) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}

#define LDPC_SPLIT_DISPATCH(FN, ...)                    \
    switch (max_deg) {                                  \
        case 12: return FN<12>(__VA_ARGS__);            \
        default: break;                                 \
    }

bool ldpc_split_supported(int max_deg) { return max_deg == 12; }
int ldpc_split_blocks_per_cu(int max_deg, int N) {
    LDPC_SPLIT_DISPATCH(occupancy_split, N)
    return 1;
}

hipError_t ldpc_split_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force, uint8_t* hard, int hard_stride,
                                    int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid, hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws) {
    LdpcKernelArgs A;
    A.work_ctr = work_ctr;
    A.sgn_ws = sgn_ws;
    if (work_ctr) {
        hipError_t e = hipMemsetAsync(work_ctr, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
    }
    A.llr = llr; A.hard = hard; A.post = post; A.trials = trials; A.msg_ws = msg_ws;
    A.nframes = nframes; A.N = C.N; A.K = C.K; A.R = C.R; A.q = C.q; A.pent_base = C.pent_base; A.synd_base = C.synd_base;
    A.max_trials = max_trials; A.force = force; A.hard_stride = hard_stride;
    A.prof = nullptr;
    const int max_deg = C.max_deg;
    LDPC_SPLIT_DISPATCH(launch_split, C, A, grid, stream)
    return hipErrorInvalidValue;
}

}  // namespace s2
