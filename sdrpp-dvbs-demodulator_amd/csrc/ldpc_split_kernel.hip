// Layered offset-min-sum LDPC decoder for DVB-S2, gfx950 -- HALF-ROW mapping: two lanes per parity-check row, one frame per workgroup.
//
// Replaces BBFrameLDPC::decode (reference src/demod/dvbs2/codings/bbframe_ldpc.cpp:123-139) and the library under it
// (xdsopl-ldpc-pabr/layered_decoder.hh:23-133, algorithms.hh:206-277), bit-exact, like ldpc_kernel.hip, whose schedule (ldpc_plan.h) and
// arithmetic (packed int16 "Q8" with the int8 saturation rules) it shares.  What differs is the mapping onto the machine:
//   * ldpc_kernel.hip gives a lane a whole row and a workgroup two frames in lockstep: 12 waves per compute unit.
//   * here thread t of a 768-thread workgroup holds one half of a row (ldpc_split_plan.h), the halves join their (min0, min1, sign) with one
//     DPP quad_perm step.  One frame per workgroup = 64.8 KB of posteriors in LDS, TWO workgroups per compute unit = 24 waves; the two
//     frames of a compute unit are not in lockstep, so one frame's narrow phases run beside the other's wide ones.
//   * what a decoder of this shape is bound by is its VALU CYCLE count (profiles/r05_valu_rates.txt): packed 16-bit, VOP3-only, SDWA and DPP
//     instructions issue at half the rate of plain 32-bit VOP2 ones, and with 24 waves per compute unit the layer time follows the sum.  So:
//     no address arithmetic (every slot's LDS byte offset comes from a per-thread table, unpacked by one and / shift each), no exec masking
//     in the row update (idle lanes work on scratch bytes), the odd slot of a row half is a constant neutral link (posterior +127, message
//     0: magnitude 126 never displaces a real minimum, sign +), plain 32-bit operations wherever both halves of a register hold the same
//     value, and ONE code path for every layer without a deep dependency chain: a layer with shared bits is run level by level as packed
//     conflict-free row updates (ldpc_split_plan.h).
// Roofline: algorithmic bytes per frame = iters*4*edges + N + K/8 against HBM 8 TB/s is the NOMINAL figure: the state is
// on-chip (LDS + Infinity Cache); DESIGN.md section 5 prices the kernel against its VALU cycles as well.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "ldpc_lane_common.h"
#include "ldpc_split_plan.h"

namespace s2 {

#ifndef LDPC_SPLIT_EXP
#define LDPC_SPLIT_EXP 0          // development switch for TIMING experiments (results wrong): 2 = no table / record traffic in the layer loop (16 / 32 / 64: no table loads / record loads / record stores), 4 = no layer barrier, 8 = no output phase
#endif
#ifndef LDPC_SPLIT_SKIP
#define LDPC_SPLIT_SKIP 0           // development switch for counter experiments (results wrong): bit k set = the pseudo-layers of kind k do nothing
#endif
#ifndef LDPC_SPLIT_CHAIN_SPEC
#define LDPC_SPLIT_CHAIN_SPEC 1    // chain layers try the plain row update where the previous chain layer changed no shared posterior (A/B switch)
#endif
#ifndef LDPC_SPLIT_SPEC_ATTEMPT
#define LDPC_SPLIT_SPEC_ATTEMPT 1  // the same attempt ahead of the speculative passes (A/B switch)
#endif
#ifndef LDPC_SPLIT_DBG
#define LDPC_SPLIT_DBG 0
#endif
#ifndef LDPC_SPLIT_BASE_PRIO
#define LDPC_SPLIT_BASE_PRIO 0     // wave priority of the decoder outside its chain walks (A/B switch)
#endif
#ifndef LDPC_SPLIT_WPE
#define LDPC_SPLIT_WPE 8          // waves per SIMD the register allocation aims at (6 = 80 VGPRs: two workgroups per compute unit; 8 = 64: room for a 128-register front-end wave beside them)
#endif

template <int MAXDEG>
struct SplitShape {
    // a row has NL = MAXDEG + 2 links; half 0 holds the first HS = ceil(NL / 2), half 1 the rest.  ODD: half 1 has one link less -- its last slot (HS - 1) is a constant
    // neutral link in those lanes (posterior +127, message 0: magnitude 126 never displaces a real minimum, sign +), its table address a scratch byte
    static constexpr int NL = MAXDEG + 2, HS = (NL + 1) / 2, NP = (HS + 1) / 2;
    static constexpr bool ODD = (NL & 1) != 0;
    static constexpr int PREV_SLOT = ODD ? HS - 2 : HS - 1;      // the previous parity bit's slot in half 1 (the row's last link)
    static constexpr int NPW = (HS / 2 + 1) <= 1 ? 1 : (HS / 2 + 1) <= 2 ? 2 : (HS / 2 + 1) <= 4 ? 4 : 8;
    static constexpr int REC = HS <= 4 ? 1 : HS <= 8 ? 2 : 4;
    static constexpr int RI_WORD = HS / 2, RI_SHIFT = (HS & 1) ? 16 : 0;     // where the row word sits (ldpc_split_plan.h)
    // codes whose LAYER 0 -- it holds row 0, the row without a previous parity bit -- has shared bits (rates 2/5 and 2/3 normal): their layers with shared bits ask the
    // descriptor (bit 20) whether they hold that row; the other codes' instruction streams stay as they were (their layer 0 is conflict-free: kind 7)
    static constexpr int NOPREV_SHARED = (MAXDEG == 4 || MAXDEG == 8) ? 2 : 0;
    static_assert(HS / 2 + 1 <= NPW, "slots + row word must fit the thread's table entry (ldpc_split_plan.h)");
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// LDS of a workgroup = posteriors (N bytes) + LDPC_SPLIT_SCRATCH bytes (idle lanes' scratch during a sweep; the syndrome check's flags and the next frame's index outside of
// one: the two uses never overlap in time) + the hand-off area: 8 bytes per row (the chain walk's records; the speculative passes use 4 bytes per row of it + two flag words).
// gfx950 hands LDS out in GRANULES of 1 280 bytes: 67 744 bytes for N = 64 800 = 53 of a compute unit's 128, two decoder workgroups leave 22 -- room for TWO workgroups of the
// front end's timing recovery (s2_gardner2_kernel: 9 granules each) and a PL-sync walk.  Round 6 measured what a granule is worth: at 56 per decoder workgroup (16-byte level-walk
// records, since gone) only ONE timing-recovery workgroup fitted, the timing recovery beside the decoder ran in two rounds, and the pipelined step took 333.6 instead of 310.0 ms.

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

// Table / record traffic of the layer loop goes through buffer resources: (uniform base in the descriptor) + (uniform pseudo-layer offset in a scalar
// register) + (the thread's 32-bit byte offset) -- left to itself the compiler builds a 64-bit address per lane and load (v_lshl_add_u64, a half-rate
// instruction, and twice the address registers).  The loads stay visible to the compiler, which keeps vmcnt.
// (the words a thread fetches ahead live in ONE register tuple per fetch -- a vector value -- so that the empty asm statements that pin them in place keep them contiguous: as
// separate 32-bit values the allocator scatters them and the fetch lands in a second tuple, copied over behind a wait)
template <int NW> struct FetchWords { typedef uint32_t type __attribute__((ext_vector_type(NW))); };
template <> struct FetchWords<1> { typedef uint32_t type; };
template <int NW>
__device__ __forceinline__ typename FetchWords<NW>::type bload(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    static_assert(NW == 1 || NW == 2 || NW == 4, "one buffer load per fetch");
    if constexpr (NW == 1) return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
    else if constexpr (NW == 2) return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    else return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
template <int NW>
__device__ __forceinline__ uint32_t fetch_word(const typename FetchWords<NW>::type& v, int i) { if constexpr (NW == 1) return v; else return v[i]; }
template <int NW>
__device__ __forceinline__ void bstore(const uint32_t (&w)[NW], __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (NW == 1) { __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, voff, soff, 0); }
    else if constexpr (NW == 2) { const u32x2 v = {w[0], w[1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, soff, 0); }
    else {
#pragma unroll
        for (int i = 0; i < NW; i += 4) { const u32x4 v = {w[i], w[i + 1], w[i + 2], w[i + 3]}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + 4 * i, soff, 0); }
    }
}

#define QUAD_DPP(x_, ctrl) __builtin_amdgcn_update_dpp(0, (int)(x_), (ctrl), 0xf, 0xf, true)
constexpr int DPP_SWAP_HALVES = 0xB1;    // quad_perm [1,0,3,2]: the other half of the row
constexpr int DPP_FROM_HALF0 = 0xA0;     // quad_perm [0,0,2,2]: half 0's value in both lanes of a row

__device__ __forceinline__ s16x2 swap2(s16x2 a) { return __builtin_shufflevector(a, a, 1, 0); }     // (folds into the op_sel of the packed instruction that uses it)

// ---- the row update, in two parts so that a chain layer can put its walk between them.  Everything is per thread = per row half.
template <int MAXDEG>
struct RowState {
    using S = SplitShape<MAXDEG>;
    s16x2 V[S::NP], G[S::NP];        // extrinsic inputs and their offset magnitudes, two slots per register ("Q8": the int8 value in the high byte of each half)
    uint32_t addr[S::HS];            // LDS byte addresses of the slots' posteriors
    s16x2 RP[S::NP];                 // the slots' old messages, Q8 pairs
    uint32_t rw;                     // the row word (ldpc_split_plan.h)
    // everything the thread's table entry and message record hold, taken apart at the TOP of a pseudo-layer: the words fetched ahead are dead from here on, and the fetch for the
    // next pseudo-layer is issued into the very registers they arrived in -- the layer loop carries no copies of them (r05: 14 v_mov per wave and pseudo-layer)
    __device__ __forceinline__ void unpack(const typename FetchWords<S::NPW>::type& AD, const typename FetchWords<S::REC>::type& rec) {
#pragma unroll
        for (int p = 0; p < S::NP; ++p) {
            const uint32_t w = fetch_word<S::NPW>(AD, p);
            addr[2 * p] = w & 0xffffu;                                       // (the posteriors start at LDS offset 0)
            if (2 * p + 1 < S::HS) addr[2 * p + 1] = w >> 16;
            RP[p] = rec_pair_dw(fetch_word<S::REC>(rec, (2 * p) >> 2), 2 * p);
        }
        rw = (fetch_word<S::NPW>(AD, S::RI_WORD) >> S::RI_SHIFT) & 0xffffu;
    }
    __device__ __forceinline__ void pin() {
#pragma unroll
        for (int k = 0; k < S::HS; ++k) asm volatile("" : "+v"(addr[k]));
#pragma unroll
        for (int p = 0; p < S::NP; ++p) { uint32_t b = bits2(RP[p]); asm volatile("" : "+v"(b)); RP[p] = from_bits2(b); }
        asm volatile("" : "+v"(rw));
    }
    // old message of slot k as an int (kinds 1 / 3 / 6: the shared links are slots 0..3)
    __device__ __forceinline__ int msg(int k) const { return (k & 1) ? (int)bits2(RP[k >> 1]) >> 24 : (int)__builtin_amdgcn_sbfe((int)bits2(RP[k >> 1]), 8, 8); }
};

// input phase: posteriors in, extrinsic values, the row's two smallest magnitudes and sign -- M0 / M1 / SXs come back with BOTH halves of the
// word holding the whole row's value (a word with equal halves orders like its 16-bit value under 32-bit signed compares; SXs: sign in bits 15 and 31).
// LATE < 0: the whole first -LATE pairs are left out of
// the totals where `late` is not zero (chain walk: 1, speculative passes: 2) -- V / G keep the values read
// NOPREV: the pseudo-layer holds row 0 of layer 0, which has no previous parity bit (kind 7: a conflict-free layer; the plan refuses codes whose layer 0 has shared bits) --
// a kind of its own, so that the other 44 layers of a sweep do not carry the test (r05: 7 vector instructions per wave and layer)
template <int MAXDEG, int LATE, int NOPREV = 0>     // NOPREV 2: asked at run time (A/B builds, -DLDPC_SPLIT_NOPREV_RT)
__device__ __forceinline__ void row_input(RowState<MAXDEG>& R, const uint32_t late, const uint32_t noprev_t, const int t, int& M0, int& M1, int& SXs, const bool noprev_rt = false,
                                          uint32_t* x0 = nullptr /* [2]: the posteriors of slots 0 / 1 and 2 / 3 as read: bits 15:8 / 31:24 */, const int nx = 1) {
    using S = SplitShape<MAXDEG>;
    constexpr int HS = S::HS, NP = S::NP;
    uint32_t XR[NP], XH[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (2 * p + 1 < HS) lds_read_pair_i8(R.addr[2 * p], R.addr[2 * p + 1], XR[p], XH[p]);
        else { lds_read_lo_i8(R.addr[2 * p], XR[p]); XH[p] = 0x007f0000u; }          // the odd slot: a neutral link, posterior +127
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
        if (p < nx && x0) x0[p] = bits2(X);
        s16x2 v = sat_sub2(X, R.RP[p]);                     // int8 saturation by the 16-bit clamp
        const s16x2 av = pmax2(v, sat_sub2(splat2(0), v));
        // |v| - 1 clamped at 0 (ldpc_kernel.hip: no upper clamp needed, only the high byte is ever consumed)
        s16x2 g = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, av), (u16x2){256, 256}));
        if ((NOPREV == 1 || (NOPREV == 2 && noprev_rt)) && (S::PREV_SLOT >> 1) == p) {
            // row 0 of layer 0 has no previous parity bit (the last link of half 1): that thread's slot becomes the neutral link
            const bool me = (uint32_t)t == noprev_t;
            constexpr int hh = S::PREV_SLOT & 1;
            v[hh] = me ? (short)(127 << 8) : v[hh];
            g[hh] = me ? (short)(126 << 8) : g[hh];
        }
        if (S::ODD && ((HS - 1) >> 1) == p) {
            // an odd number of links per row: half 1's last slot is the neutral link (whatever its scratch byte and record byte hold)
            const bool me = (t & 1) != 0;
            constexpr int hh = (HS - 1) & 1;
            v[hh] = me ? (short)(127 << 8) : v[hh];
            g[hh] = me ? (short)(126 << 8) : g[hh];
        }
        R.V[p] = v; R.G[p] = g;
        if constexpr (LATE < 0) {
            if (p < -LATE) { v = late ? splat2(127 << 8) : v; g = late ? splat2(126 << 8) : g; }      // (the neutral link)
        }
        if (p == 0) { MIN0 = g; }
        else if (p == 1) { MIN1 = pmax2(MIN0, g); MIN0 = pmin2(MIN0, g); }
        else { MIN1 = pmin2(MIN1, pmax2(MIN0, g)); MIN0 = pmin2(MIN0, g); }
        SX ^= bits2(v);
    }
    // even / odd slots joined (op_sel swaps the halves inside the packed instructions) ...
    const int m0 = (int)bits2(pmin2(MIN0, swap2(MIN0)));
    const int m1 = (int)bits2(pmin2(pmax2(MIN0, swap2(MIN0)), pmin2(MIN1, swap2(MIN1))));
    const int sx = (int)(SX ^ __builtin_amdgcn_alignbit(SX, SX, 16));
    // ... and the two halves of the row
    const int o0 = QUAD_DPP(m0, DPP_SWAP_HALVES), o1 = QUAD_DPP(m1, DPP_SWAP_HALVES), os = QUAD_DPP(sx, DPP_SWAP_HALVES);
    M1 = min(max(m0, o0), min(m1, o1));
    M0 = min(m0, o0);
    SXs = sx ^ os;
}

// the totals again, from the values row_input left in the registers (a chain layer whose speculative attempt failed: now without the pair)
template <int MAXDEG, int LATE>
__device__ __forceinline__ void row_totals(const RowState<MAXDEG>& R, const uint32_t late, int& M0, int& M1, int& SXs) {
    using S = SplitShape<MAXDEG>;
    constexpr int NP = S::NP;
    s16x2 MIN0 = splat2(Q8_NONE), MIN1 = splat2(Q8_NONE);
    uint32_t SX = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        s16x2 v = R.V[p], g = R.G[p];
        if constexpr (LATE < 0) {
            if (p < -LATE) { v = late ? splat2(127 << 8) : v; g = late ? splat2(126 << 8) : g; }
        }
        if (p == 0) { MIN0 = g; }
        else if (p == 1) { MIN1 = pmax2(MIN0, g); MIN0 = pmin2(MIN0, g); }
        else { MIN1 = pmin2(MIN1, pmax2(MIN0, g)); MIN0 = pmin2(MIN0, g); }
        SX ^= bits2(v);
    }
    const int m0 = (int)bits2(pmin2(MIN0, swap2(MIN0)));
    const int m1 = (int)bits2(pmin2(pmax2(MIN0, swap2(MIN0)), pmin2(MIN1, swap2(MIN1))));
    const int sx = (int)(SX ^ __builtin_amdgcn_alignbit(SX, SX, 16));
    const int o0 = QUAD_DPP(m0, DPP_SWAP_HALVES), o1 = QUAD_DPP(m1, DPP_SWAP_HALVES), os = QUAD_DPP(sx, DPP_SWAP_HALVES);
    M1 = min(max(m0, o0), min(m1, o1));
    M0 = min(m0, o0);
    SXs = sx ^ os;
}

// output phase: new messages and posteriors of every slot from the row totals (both halves of M0 / M1 equal, sign of the row in bits 15 and 31 of SXs)
// SKIP: the first SKIP slots, where flagged in `early`, were written ahead (layers with shared links)
template <int MAXDEG, int SKIP>
__device__ __forceinline__ void row_output(const RowState<MAXDEG>& R, const int M0, const int M1, const int SXs, const uint32_t early, uint32_t (&rec_out)[SplitShape<MAXDEG>::REC],
                                           uint32_t* pn0 = nullptr /* [2]: the new posteriors of slots 0 / 1 and 2 / 3: bits 7:0 / 23:16 */, const int npn = 1) {
    using S = SplitShape<MAXDEG>;
    constexpr int HS = S::HS, NP = S::NP, REC = S::REC;
    // the selected magnitude is limited to 32 once per row (the per-link clamp to [-32, 31] then only needs its upper side); plain 32-bit operations on words with equal halves
    const uint32_t C32 = 0x20002000u;
    const uint32_t MIN1C = (uint32_t)min(M1, (int)C32);
    const uint32_t SUMC = (uint32_t)min(M0, (int)C32) + MIN1C;
    const uint32_t SXB = (uint32_t)SXs;
    s16x2 NM[NP + 1];
    NM[NP] = splat2(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        // other = (mag == min0) ? min1c : min0c, as min0c + min1c - min(mag, min1c): a magnitude is the row minimum, where min(mag, min1c) = min0c, or at
        // least the second one, where it is min1c (algorithms.hh:250-256).  No borrow between the halves: a 32-bit subtraction
        const s16x2 other = from_bits2(SUMC - bits2(pmin2(R.G[p], from_bits2(MIN1C))));
        const s16x2 neg = from_bits2(SXB ^ bits2(R.V[p])) >> 15;                   // 0 or -1
        s16x2 nm = pmin2(from_bits2(bits2(other) ^ bits2(neg)) - neg, q8(31));
        // new posterior: 16-bit saturating add = int8 saturation; >> 8 brings the bytes to bits 7:0 / 23:16 for the stores
        const uint32_t pn = bits2(sat_add2(R.V[p], nm)) >> 8;
        if (p < npn && pn0) pn0[p] = pn;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int k = 2 * p + hh;
            if (k >= HS) { nm[hh] = 0; continue; }
            if (k < SKIP) {
                if (!((early >> k) & 1)) { if (hh == 0) lds_write_lo_i8(R.addr[k], pn); else lds_write_hi_i8(R.addr[k], pn); }
            } else {
                if (hh == 0) lds_write_lo_i8(R.addr[k], pn); else lds_write_hi_i8(R.addr[k], pn);
            }
        }
        NM[p] = nm;
    }
#pragma unroll
    for (int w = 0; w < REC; ++w) {
        if (2 * w < NP) rec_out[w] = __builtin_amdgcn_perm(bits2(NM[2 * w + 1 <= NP ? 2 * w + 1 : NP]), bits2(NM[2 * w]), 0x07050301u);
        else rec_out[w] = 0;
    }
}

#if defined(LDPC_PROF) && LDPC_PROF == 3
// waypoints inside a chain layer (threads 0 and 384 of workgroup 0): prof[300 + 16 * (thread != 0) + i] = cycles since the previous waypoint
#define SPLIT_MARK(i) do { if (KIND == 1 && g_prof_dev && blockIdx.x == 0 && (t == 0 || t == 384)) { const unsigned long long t_now = clock64(); g_prof_dev[300 + (t ? 16 : 0) + (i)] += t_now - t_mark; t_mark = t_now; } } while (0)
#define SPLIT_MARK_DECL unsigned long long t_mark = clock64()
__device__ unsigned long long* g_prof_dev = nullptr;
#else
#define SPLIT_MARK(i) do { } while (0)
#define SPLIT_MARK_DECL do { } while (0)
#endif
#define LINK_IN(k) ((int)R.V[(k) >> 1][(k) & 1] >> 8)
#define LINK_MG(k) ((int)R.G[(k) >> 1][(k) & 1] >> 8)
#define LINK_SET(k, v, m) do { R.V[(k) >> 1][(k) & 1] = (short)((v) << 8); R.G[(k) >> 1][(k) & 1] = (short)((m) << 8); } while (0)

// The chain walk's hand-off (ldpc_lane_common.h: chain_record / chain_step), in the form the WALKER is cheapest in.  A/B of round 6: a record that took 8 instead of 29
// vector instructions per wave to build (17 chain layers x 12 waves per sweep) but cost the walker one instruction more per row made the kernel 1.2 % SLOWER -- the walk is the
// workgroup's serial section (349 rows per sweep), a row costs its instruction count at one instruction per ~5 cycles (lone wave), and nothing else of the workgroup runs
// meanwhile.  So: everything a row needs sits in ONE 8-byte record the walker reads with one load two rows ahead -- lim = the four clamp limits as bytes (SDWA operands),
// se = E'' (16 bits) | sneg (byte 2: 0 / -1) | byte 3 = the posterior the row's L link reads, WRITTEN BY THE WALKER into the record it has just consumed (one running
// pointer serves load and store) -- and the sign is applied as (t ^ sneg) + E'', E'' = E' - sneg: 4 clamps + add + xor + add + clamp = 8 vector instructions per row where the
// multiply-add form took 9 and two field extractions; four records in flight, no copies.
__device__ __forceinline__ uint2 chain_record_x(const int m, const int qE, const int vE, const int sneg /* 0 or -1 */) {
    const int sig = 1 | sneg;
    const int c31 = min(qE, 31), c32 = min(qE, 32);
    const int CA = (sneg ? c32 : c31) + 1, CB = (sneg ? c31 : c32) + 1;
    const uint32_t lim = ((uint32_t)(m + 1) & 0xffu) | (((uint32_t)(m + CA) & 0xffu) << 8) | (((uint32_t)(m - CB) & 0xffu) << 16) | ((uint32_t)(m - 1) << 24);
    const uint32_t se = ((uint32_t)(vE - 2 * sig * m - sneg) & 0xffffu) | (((uint32_t)sneg & 0xffu) << 16);
    return make_uint2(lim, se);
}
__device__ __forceinline__ int chain_step_x(const int x, const u32x2 r) {
    const int q1 = min(max(x, (int)(int8_t)r.x), (int)(int8_t)(r.x >> 8));
    const int q2 = min(max(x, (int)(int8_t)(r.x >> 16)), (int)r.x >> 24);
    int tx = (q1 + q2) ^ (int)(int8_t)(r.y >> 16);
    asm volatile("" : "+v"(tx));              // (left to itself the compiler fuses xor and add into a VOP3 v_xad_u32, which cannot take the SDWA byte / word operands: two extractions more)
    return clamp8(tx + (int)(short)r.y);
}
typedef __attribute__((address_space(3))) u32x2 lds_u2;
#define LDS_U2(a) (*(const lds_u2*)(uintptr_t)(a))
typedef u32x2 __attribute__((aligned(4))) u32x2_a4;
typedef __attribute__((address_space(3))) u32x2_a4 lds_u2_a4;
#define LDS_U2_A4(a) (*(const lds_u2_a4*)(uintptr_t)(a))          // (records at a 12-byte stride: two dwords, dword-aligned)

// A layer whose shared links are one pair (slots 0 = "E", 1 = "L" of half 0: row j's E bit is row (j + d)'s L bit) with a deep dependency chain: ldpc_kernel.hip's chain
// walk, its arithmetic kept packed.  Rows in lane order (row j = t >> 1).  The row word -- level | late << 8 | early << 12 in half 0, the level alone in half 1, zero in
// idle lanes -- rides in the table.
//   phase A  every row: the conflict-free input phase; rows of level > 1 keep their first pair out of the totals.  Rows of level 1 (no late link) finish here: their
//            output phase writes every slot, the early ones included, before the first barrier.  A row whose L link is late leaves the walk's record: what its E link
//            gives back as a function of what its L link reads (ldpc_lane_common.h: chain_record) -- the totals without the pair are exactly what that needs.
//   walk     lane c < d follows rows c + d, c + 2d, ...: the posterior one row's E link leaves is the next row's L input (cres[])
//   phase C  rows of level > 1: the pair's late slots re-read (L from cres[], E from its bit: a row whose E link is late waits for an EARLIER row's L link), joined into
//            the totals with packed operations, handed to half 1, then the output phase -- which leaves out the early slots: a later row's L link owns that bit's final value.
//   SPECULATION (round 6)  On a frame that has converged no row changes a posterior any more: what a row's late link reads from its bit at the start of the layer IS what
//            the earlier row will leave there.  A chain layer that finds this -- every slot a later row reads ("early") got back the value it held -- tells the next chain
//            layer (a flag word behind the records), and that one makes an ATTEMPT: the plain row update with every input taken from its bit, the pair's slots of the rows
//            of level > 1 held back, the same check; one barrier; where the check held for every row of the workgroup the layer is done after the held-back stores (the
//            guesses were the values the walk would have carried: induction along the chains) -- no records, no walk, no second output phase: 100 instead of 187 vector
//            instructions per wave, two barriers instead of three.  Where it did not, the layer goes the long way from the totals on (phase A's stores of the rows of
//            level 1 stand, everything else is computed again from the inputs in the registers) and the next chain layer does not try.
template <int MAXDEG>
__device__ __forceinline__ void chain_layer(RowState<MAXDEG>& R, uint32_t (&rec_out)[SplitShape<MAXDEG>::REC],
                                            const LdpcSplitLayer L, const uint32_t eL, const int t, int8_t* __restrict__ post, uint32_t* __restrict__ cw,
                                            const uint32_t flagb /* LDS address of the two verdict words */, const uint32_t cseq /* chain layers of this frame so far, this one included */,
                                            const bool fail_attempts /* tests: every attempt is made to fail */) {
    [[maybe_unused]] constexpr int KIND = 1;
    typedef __attribute__((address_space(3))) uint32_t lds_u1;
    const uint32_t rw = R.rw;
    const uint32_t level = rw & 0xffu, late = (rw >> 8) & 3u, early = (rw >> 12) & 3u;
    const int chain_d = (int)(L.aux & 0xffffu);
    const int j = t >> 1;
    const bool half1 = (t & 1) != 0;
    const uint32_t emask = ((early & 1u) ? 0xffu : 0u) | ((early & 2u) ? 0xff0000u : 0u);      // the bytes of the pair's posteriors a later row reads
    const uint32_t my_flag = flagb + 4u * (cseq & 1u);
    // the previous chain layer's verdict: its word holds its number where some early slot changed (the words are cleared with the frame: the first chain layer does not try).
    // (Tried: two consistent layers in a row before an attempt -- slower where frames converge slowly, no faster on noise; the word read behind the layer's barrier and
    //  claimed at the top of the next pseudo-layer -- the bookkeeping in the layer loop cost more than the round trip here.)
    const bool attempt = LDPC_SPLIT_CHAIN_SPEC && (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const lds_u1*)(uintptr_t)(flagb + 4u * ((cseq - 1u) & 1u))) != cseq - 1u;
    int M0, M1, SXs;
    uint32_t x0, pn0 = 0;
    SPLIT_MARK_DECL;
    row_input<MAXDEG, -1, SplitShape<MAXDEG>::NOPREV_SHARED>(R, (!attempt && level > 1u && !half1) ? 1u : 0u, 1u /* rows in lane order: row 0's half 1 */, t, M0, M1, SXs, ((L.kind_nw >> 20) & 1u) != 0, &x0);
    SPLIT_MARK(0);
    if (attempt) {
        row_output<MAXDEG, 2>(R, M0, M1, SXs, (level > 1u && !half1) ? 3u : 0u, rec_out, &pn0);
        const bool bad = fail_attempts || ((pn0 ^ (x0 >> 8)) & emask) != 0;
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && (t & 63) == 0) *(lds_u1*)(uintptr_t)my_flag = cseq;
        lds_pairs_wait();
        lds_barrier();
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)*(const lds_u1*)(uintptr_t)my_flag) != cseq) {
            // every guess was right: the held-back stores (a slot a later row touches stays that row's), and the layer is done
            if (level > 1u && !half1) {
                if (!(early & 1u)) lds_write_lo_i8(R.addr[0], pn0);
                if (!(early & 2u)) lds_write_hi_i8(R.addr[1], pn0);
            }
            return;
        }
        // the long way: the totals again, without the pair where the row has late links (the rows of level 1 are done: their inputs were no guesses)
        row_totals<MAXDEG, -1>(R, (level > 1u && !half1) ? 1u : 0u, M0, M1, SXs);
    } else if (level == 1u) {
        row_output<MAXDEG, 0>(R, M0, M1, SXs, 0u, rec_out, &pn0);
    }
    if ((late >> 1) & 1) {
        // totals without the pair: min0 = the smallest magnitude among the row's other links (what the E link's new message takes), sign = their product
        reinterpret_cast<uint2*>(cw)[j] = chain_record_x(R.msg(1), (late & 1u) ? 255 : (M0 >> 24), (int)R.V[0][0] >> 8, SXs >> 31);
    }
    SPLIT_MARK(1);
    lds_pairs_wait();
    lds_barrier();
    SPLIT_MARK(2);
    uint32_t cwb = lds_offset(reinterpret_cast<const int8_t*>(cw));
    asm volatile("" : "+s"(cwb));             // (an LDS address with the dynamic-LDS symbol still in it is re-added at every use)
    if (t < chain_d) {
        // lane c walks rows c + k d, k = 1 .. T - 1, and leaves row T its input
        __builtin_amdgcn_s_setprio(3);
        const int T = (int)(L.aux >> 16);                 // 359 / d, from the plan (a division here sits in the workgroup's serial section)
        int x = post[link_addr(eL, t + chain_d)];
        const uint32_t step = 8u * (uint32_t)chain_d, step4 = 4u * step;
        uint32_t p0 = cwb + 8u * (uint32_t)(t + chain_d), p1 = p0 + step, p2 = p1 + step, p3 = p2 + step;
        u32x2 r0 = LDS_U2(p0), r1 = LDS_U2(p1), r2 = LDS_U2(p2), r3 = LDS_U2(p3);       // (reads past row 359 land in LDS the workgroup owns and are never used)
        int k = 1;
        // (the asm statement orders the row's store, its arithmetic and the refill of its record register: the compiler otherwise hoists the refill and keeps a second set of pointers)
#define WALK_ROW(p, r) do { LDS_I8((p) + 7u) = (int8_t)x; x = chain_step_x(x, r); asm volatile("" : "+v"(p), "+v"(x) : : "memory"); } while (0)
        for (; k + 4 <= T; k += 4) {
            WALK_ROW(p0, r0); p0 += step4; r0 = LDS_U2(p0);
            WALK_ROW(p1, r1); p1 += step4; r1 = LDS_U2(p1);
            WALK_ROW(p2, r2); p2 += step4; r2 = LDS_U2(p2);
            WALK_ROW(p3, r3); p3 += step4; r3 = LDS_U2(p3);
        }
        if (k < T) {
            WALK_ROW(p0, r0); ++k;
            if (k < T) {
                WALK_ROW(p1, r1); ++k;
                if (k < T) { WALK_ROW(p2, r2); ++k; }
            }
        }
#undef WALK_ROW
        if (t + T * chain_d < 360) LDS_I8(cwb + 8u * (uint32_t)(t + T * chain_d) + 7u) = (int8_t)x;
        __builtin_amdgcn_s_setprio(LDPC_SPLIT_BASE_PRIO);
    }
    SPLIT_MARK(3);
    lds_barrier();
    SPLIT_MARK(4);
    if (level > 1u) {
        if (!half1) {
            // the pair again, from what the walk (L) and the earlier rows (E) left; only the late halves replace what phase A read
            uint32_t xr, xh;
            lds_read_pair_i8(R.addr[0], cwb + 8u * (uint32_t)j + 7u, xr, xh);        // (E from its bit, L from the byte the walker left in the row's record)
            lds_ready_n(0, xr, xh);
            const s16x2 X = from_bits2(__builtin_amdgcn_perm(xh, xr, 0x060c000cu));
            const s16x2 v = sat_sub2(X, R.RP[0]);
            const s16x2 av = pmax2(v, sat_sub2(splat2(0), v));
            const s16x2 g = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, av), (u16x2){256, 256}));
            // bits 8 / 9 of the row word -> 0xffff in the low / high half (v_perm: selectors 9 / 11 = the sign of byte 3 of either source)
            const uint32_t rws = rw << 22;            // bit 31 = late L (bit 9 of the row word), bit 30 = late E (bit 8)
            const uint32_t LM = __builtin_amdgcn_perm(rws << 1, rws, 0x09090b0bu);
            R.V[0] = from_bits2((bits2(v) & LM) | (bits2(R.V[0]) & ~LM));
            R.G[0] = from_bits2((bits2(g) & LM) | (bits2(R.G[0]) & ~LM));
            // the pair joins the totals: its smaller magnitude first, then the larger one (words with equal halves: 32-bit operations)
            const int a = (int)bits2(pmin2(R.G[0], swap2(R.G[0]))), b = (int)bits2(pmax2(R.G[0], swap2(R.G[0])));
            M1 = min(M1, max(M0, a));
            M0 = min(M0, a);
            M1 = min(M1, max(M0, b));
            const uint32_t sv = bits2(R.V[0]);
            SXs ^= (int)(sv ^ __builtin_amdgcn_alignbit(sv, sv, 16));
        }
        // the complete totals sit in half 0: hand them to half 1
        M0 = QUAD_DPP(M0, DPP_FROM_HALF0);
        M1 = QUAD_DPP(M1, DPP_FROM_HALF0);
        SXs = QUAD_DPP(SXs, DPP_FROM_HALF0);
        SPLIT_MARK(5);
        row_output<MAXDEG, 2>(R, M0, M1, SXs, early, rec_out, &pn0);
    }
    // the verdict for the next chain layer: did any slot a later row reads change?  (after a failed attempt the word is set already)
    if (LDPC_SPLIT_CHAIN_SPEC && !attempt) {
        const bool bad = ((pn0 ^ (x0 >> 8)) & emask) != 0;
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && (t & 63) == 0) *(lds_u1*)(uintptr_t)my_flag = cseq;
    }
    SPLIT_MARK(6);
}

// A layer with shared links (slots 0..nc-1 of half 0), rows in lane order (row j = t >> 1): ldpc_kernel.hip's three forms.  KIND 1: one shared pair (slots 0 = "E",
// 1 = "L"), the dependency chains walked by a few lanes; 6: quad walk (at most 4 shared links, deep and narrow level structure: one wave walks the rows of levels
// >= 2, four lanes per row); 3: a barrier per level (at most 4 shared links).  The row word -- level | late << 8 | early << 12, zero for half 1 and idle lanes, so
// every condition below is false there -- rides in the table.
#ifndef LDPC_SPLIT_FIX
#define LDPC_SPLIT_FIX 7
#endif
#if LDPC_SPLIT_FIX & 1
typedef const __attribute__((address_space(4))) uint32_t* const_u32_ptr;
typedef const __attribute__((address_space(4))) LdpcSplitLayer* const_layer_ptr;
#else
typedef const uint32_t* const_u32_ptr;
typedef const LdpcSplitLayer* const_layer_ptr;
#endif

__device__ __forceinline__ LdpcSplitLayer layer_at(const_layer_ptr layers, int i) {      // (field by field: a struct in the constant address space has no copy constructor on the host pass)
    LdpcSplitLayer L;
    L.kind_nw = layers[i].kind_nw; L.aux = layers[i].aux; L.rec_off = layers[i].rec_off; L.ent_off = layers[i].ent_off;
    return L;
}
// A layer with shared links handled SPECULATIVELY (kind 8; rate 3/4: layers 3, 5, 42 -- two shared pairs / a triple, 6 / 12 / 33 dependency levels -- and layer 18, one pair
// in two chains of 179 rows).  Rounds 5 / 6 walked such layers level by level with ONE wave (a barrier per level, a quad walk, the level walk: 7.8 / 11.5 / 18.4 / 21.8 k cycles
// of a 192 k sweep, the other eleven waves waiting).  Here every row of level > 1 runs in ITS OWN lane, all at once, in PASSES:
//   phase A  every row: the input phase; rows of level > 1 keep slots 0..3 out of the totals.  Rows of level 1 (no predecessor in the layer) finish here, shared slots included.
//   passes   a row of level > 1 reads the posteriors of its four slots from their SOURCES -- the bit itself where nobody of level > 1 touches it before this row, else the
//            output cell of the row that does (cw[row]: four bytes, one per slot; the plan's side entry names the cell) --, computes inputs, magnitudes, for every slot the
//            minimum over the other three and the rest of the row, new messages, new posteriors, and writes its own cell.  After pass p the rows of level <= p + 2 hold their
//            final values (induction over the levels), so depth - 1 passes always suffice; a pass in which NO row read anything new ends the layer earlier: the cells then
//            are a fixed point of "every row computed from its predecessors' outputs", and by the same induction there is only one -- the reference's sequential result
//            (layered_decoder.hh:46-74: a row reads a shared bit after exactly the rows before it).  What a row reads in its first pass is the bit as the previous layer
//            left it -- on a frame that has converged that IS what its predecessor will write, and every layer ends after two passes; an early, noisy sweep takes a few more
//            (a changed input reaches the next row's output only through the min, which mostly ignores it).  Reads race with the same pass's writes: a newer value is as good.
//   phase C  rows of level > 1: the four inputs of the last pass join the totals, then the output phase; a slot a LATER row touches stays unwritten (that row owns the bit).
// Bit-exact by construction; the time depends on the data (passes), the result does not.
#ifndef LDPC_SPLIT_SPEC_STATS
#define LDPC_SPLIT_SPEC_STATS 0      // development aid: prof[512 + 32 * (pseudo-layer % 11) + passes] counts the layers of workgroup 0 by passes taken (tools/ldpc_split_prof.py)
#endif
template <int MAXDEG>
__device__ __forceinline__ void spec_layer(RowState<MAXDEG>& R, uint32_t (&rec_out)[SplitShape<MAXDEG>::REC], const LdpcSplitLayer L, const uint32_t* __restrict__ tab, const int t,
                                           uint32_t* __restrict__ cw, [[maybe_unused]] unsigned long long* prof, [[maybe_unused]] const int pl,
                                           const uint32_t vflagb /* LDS address of the two verdict words */, const uint32_t cseq /* layers with shared bits of this frame so far, this one included */,
                                           const bool fail_attempts /* tests: every attempt is made to fail */) {
    static_assert(SplitShape<MAXDEG>::HS >= 4, "the shared links are slots 0..3 of half 0");
    typedef __attribute__((address_space(3))) uint32_t lds_u1;
    const uint32_t level = R.rw & 0xffu;               // (both halves carry it; idle lanes: 0)
    const int j = t >> 1;
    const bool half1 = (t & 1) != 0;
    const bool spec = level > 1u && !half1;
    const int depth = (int)(L.aux >> 16);
    // the row's side entry {f0 | f1 << 16, f2 | f3 << 16}: f & 0x7ff = distance from the row's own cell back to slot's source cell (0: the bit itself), bit 15: a later row touches the slot
    // (fetched behind the compiler's back and claimed by hand behind the first pass: a load the compiler sees inside ONE kind's branch makes it open the claims of the layer
    // loop -- every kind's -- with vmcnt(0), which also waits for the previous pseudo-layer's record store: +10 % on the whole kernel when tried)
    u32x2 side;
    {
        const uint64_t sbase = (uint64_t)(uintptr_t)(tab + L.ent_off);
        const uint32_t voff = (uint32_t)j * 8u;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(side) : "v"(voff), "s"(sbase) : "memory");
    }
    uint32_t cwb = lds_offset(reinterpret_cast<const int8_t*>(cw));
    asm volatile("" : "+s"(cwb));
    const uint32_t flagb = cwb + 4u * 384u;            // two words behind the cells: flag[p & 1] = p + 1 where a row read something new in pass p
    if (t == 0) { *(lds_u1*)(uintptr_t)flagb = 0u; *(lds_u1*)(uintptr_t)(flagb + 4u) = 0u; }
    // the verdict words the layers with shared bits hand on (chain_layer): where the layer before changed no posterior a later row reads, this one first tries the plain row
    // update -- every input from its bit, slots 0..3 of the rows of level > 1 held back, one barrier, done if every such posterior got its value back -- and only else the passes
    const uint32_t my_flag = vflagb + 4u * (cseq & 1u);
    const bool attempt = LDPC_SPLIT_CHAIN_SPEC && LDPC_SPLIT_SPEC_ATTEMPT && (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const lds_u1*)(uintptr_t)(vflagb + 4u * ((cseq - 1u) & 1u))) != cseq - 1u;
    // (a copy of the side entry of its own for the attempt: a value fetched behind the compiler's back must have ONE place where it is claimed -- with two, the compiler
    //  moved the registers between them while the load was still on its way)
    u32x2 side_a = {0u, 0u};
    if (attempt) {
        const uint64_t sbase = (uint64_t)(uintptr_t)(tab + L.ent_off);
        const uint32_t voff = (uint32_t)j * 8u;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(side_a) : "v"(voff), "s"(sbase) : "memory");
    }
    int M0, M1, SXs;
    uint32_t xr[2], pn[2] = {0u, 0u};
    row_input<MAXDEG, -2, SplitShape<MAXDEG>::NOPREV_SHARED>(R, (!attempt && spec) ? 1u : 0u, 1u, t, M0, M1, SXs, ((L.kind_nw >> 20) & 1u) != 0, xr, 2);
    uint32_t em0 = 0, em1 = 0;                         // the bytes of slots 0..3 a later row reads (known once the side entry is in)
#if LDPC_SPLIT_SPEC_STATS
    if (prof && blockIdx.x == 0 && t == 0) { prof[900] += 1; prof[901] += attempt; prof[910 + (cseq < 40 ? cseq : 40)] += attempt; }
#endif
    if (attempt) {
        if (!(LDPC_SPLIT_DBG & 2) || level == 1u) row_output<MAXDEG, 4>(R, M0, M1, SXs, spec ? 15u : 0u, rec_out, pn, 2);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(side_a) : : "memory");
        em0 = (((side_a.x >> 15) & 1u) ? 0xffu : 0u) | (((side_a.x >> 31) & 1u) ? 0xff0000u : 0u);
        em1 = (((side_a.y >> 15) & 1u) ? 0xffu : 0u) | (((side_a.y >> 31) & 1u) ? 0xff0000u : 0u);
        const bool bad = (LDPC_SPLIT_DBG & 1) || fail_attempts || (!half1 && ((((pn[0] ^ (xr[0] >> 8)) & em0) | ((pn[1] ^ (xr[1] >> 8)) & em1)) != 0));
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && (t & 63) == 0) *(lds_u1*)(uintptr_t)my_flag = cseq;
        if (!(LDPC_SPLIT_DBG & 8)) { lds_pairs_wait();
        lds_barrier(); }
        if (!(LDPC_SPLIT_DBG & 8) && (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const lds_u1*)(uintptr_t)my_flag) != cseq) {
#if LDPC_SPLIT_SPEC_STATS
            if (prof && blockIdx.x == 0 && t == 0) prof[902] += 1;
#endif
            if (spec) {
                if (!(em0 & 0xffu)) lds_write_lo_i8(R.addr[0], pn[0]);
                if (!(em0 >> 16)) lds_write_hi_i8(R.addr[1], pn[0]);
                if (!(em1 & 0xffu)) lds_write_lo_i8(R.addr[2], pn[1]);
                if (!(em1 >> 16)) lds_write_hi_i8(R.addr[3], pn[1]);
            }
            return;
        }
        if (LDPC_SPLIT_DBG & 4) row_input<MAXDEG, -2, SplitShape<MAXDEG>::NOPREV_SHARED>(R, spec ? 1u : 0u, 1u, t, M0, M1, SXs, ((L.kind_nw >> 20) & 1u) != 0);
        else row_totals<MAXDEG, -2>(R, spec ? 1u : 0u, M0, M1, SXs);          // the long way: the totals without slots 0..3 where the row has predecessors
    } else {
        if (level == 1u) row_output<MAXDEG, 0>(R, M0, M1, SXs, 0u, rec_out, pn, 2);
    }
    const uint32_t cell = cwb + 4u * (uint32_t)j;
    // (the FIRST pass reads every slot from its bit -- the guess: what the previous layer left there; the later ones from the sources)
    uint32_t cur[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cur[s] = R.addr[s];
    const s16x2 Q0 = from_bits2((uint32_t)min(M0 >> 24, 32) * 0x01000100u);      // min(min0 of the other links, 32) in both halves (Q8)
    const uint32_t s0w = (uint32_t)(SXs >> 31) & 0x80008000u;                     // their sign product in bits 15 and 31
    lds_pairs_wait();
    lds_barrier();
    s16x2 V0 = R.V[0], V1 = R.V[1];
    uint32_t X0p = 0, X1p = 0;
    int p = 0;
    bool voted = false;                                 // the layer ended on a vote: its barrier already lies behind the last pass
    const bool any_spec = __builtin_amdgcn_ballot_w64(spec) != 0;
    for (;;) {
        bool ch = false;
        if (any_spec) {
            if (spec) {
                uint32_t xr0, xh0, xr1, xh1;
                lds_read_pair_i8(cur[0], cur[1], xr0, xh0);
                lds_read_pair_i8(cur[2], cur[3], xr1, xh1);
                lds_ready_n(0, xr0, xh0);
                asm volatile("" : "+v"(xr1), "+v"(xh1));
                const uint32_t X0 = __builtin_amdgcn_perm(xh0, xr0, 0x060c000cu), X1 = __builtin_amdgcn_perm(xh1, xr1, 0x060c000cu);
                ch = ((X0 ^ X0p) | (X1 ^ X1p)) != 0;
                X0p = X0; X1p = X1;
                V0 = sat_sub2(from_bits2(X0), R.RP[0]);
                V1 = sat_sub2(from_bits2(X1), R.RP[1]);
                const s16x2 G0 = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, pmax2(V0, sat_sub2(splat2(0), V0))), (u16x2){256, 256}));
                const s16x2 G1 = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, pmax2(V1, sat_sub2(splat2(0), V1))), (u16x2){256, 256}));
                // for every slot the smallest magnitude among the row's OTHER links: the other slot of its pair, both slots of the other pair, min0 of the rest (limited to 32)
                const s16x2 P0 = pmin2(G0, swap2(G0)), P1 = pmin2(G1, swap2(G1));
                const s16x2 O0 = pmin2(pmin2(swap2(G0), P1), Q0), O1 = pmin2(pmin2(swap2(G1), P0), Q0);
                // sign of the row: the rest's product times the four slots' (bits 15 and 31 after the fold)
                uint32_t sx = bits2(V0) ^ bits2(V1);
                sx ^= __builtin_amdgcn_alignbit(sx, sx, 16);
                sx ^= s0w;
                const s16x2 N0 = from_bits2(sx ^ bits2(V0)) >> 15, N1 = from_bits2(sx ^ bits2(V1)) >> 15;
                const s16x2 NM0 = pmin2(from_bits2(bits2(O0) ^ bits2(N0)) - N0, q8(31)), NM1 = pmin2(from_bits2(bits2(O1) ^ bits2(N1)) - N1, q8(31));
                // the four new posteriors into the row's cell (byte s = slot s)
                *(lds_u1*)(uintptr_t)cell = __builtin_amdgcn_perm(bits2(sat_add2(V1, NM1)), bits2(sat_add2(V0, NM0)), 0x07050301u);
            }
        }
        if (p == 0) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(side) : : "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint32_t f = ((s < 2 ? side.x : side.y) >> (16 * (s & 1))) & 0x7ffu;
                cur[s] = f ? cell - f : R.addr[s];
            }
        }
        ++p;
        if (p >= depth - 1) break;                      // (the rows of every level have had their final inputs: no vote)
        // a row that read something new in this pass asks for another one (the first pass read guesses: no vote either)
        if (p > 1 && __builtin_amdgcn_ballot_w64(ch) != 0 && (t & 63) == 0) *(lds_u1*)(uintptr_t)(flagb + 4u * (uint32_t)(p & 1)) = (uint32_t)p + 1u;
        lds_pairs_wait();
        lds_barrier();
        if (p > 1) {
            const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const lds_u1*)(uintptr_t)(flagb + 4u * (uint32_t)(p & 1)));
            if (fl != (uint32_t)p + 1u) { voted = true; break; }
        }
    }
#if LDPC_SPLIT_SPEC_STATS
    if (prof && blockIdx.x == 0 && t == 0) prof[512 + 32 * (pl % 11) + (p < 31 ? p : 31)] += 1;
#endif
    // (phase C writes bits that rows of other waves read as sources in their last pass)
    if (!voted) { lds_pairs_wait(); lds_barrier(); }
    if (level > 1u) {
        if (!half1) {
            R.V[0] = V0; R.V[1] = V1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                R.G[q] = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, pmax2(R.V[q], sat_sub2(splat2(0), R.V[q]))), (u16x2){256, 256}));
                // the pair joins the totals: its smaller magnitude first, then the larger one (words with equal halves: 32-bit operations)
                const int a = (int)bits2(pmin2(R.G[q], swap2(R.G[q]))), b = (int)bits2(pmax2(R.G[q], swap2(R.G[q])));
                M1 = min(M1, max(M0, a));
                M0 = min(M0, a);
                M1 = min(M1, max(M0, b));
            }
            const uint32_t sv = bits2(R.V[0]) ^ bits2(R.V[1]);
            SXs ^= (int)(sv ^ __builtin_amdgcn_alignbit(sv, sv, 16));
        }
        M0 = QUAD_DPP(M0, DPP_FROM_HALF0);
        M1 = QUAD_DPP(M1, DPP_FROM_HALF0);
        SXs = QUAD_DPP(SXs, DPP_FROM_HALF0);
        const uint32_t early = ((side.x >> 15) & 1u) | ((side.x >> 30) & 2u) | ((side.y >> 13) & 4u) | ((side.y >> 28) & 8u);
        row_output<MAXDEG, 4>(R, M0, M1, SXs, half1 ? 0u : early, rec_out, pn, 2);          // (a slot a later row touches stays unwritten)
    }
    // the verdict for the next layer with shared bits (after a failed attempt the word is set already)
    if (LDPC_SPLIT_CHAIN_SPEC && !attempt) {
        em0 = (((side.x >> 15) & 1u) ? 0xffu : 0u) | (((side.x >> 31) & 1u) ? 0xff0000u : 0u);
        em1 = (((side.y >> 15) & 1u) ? 0xffu : 0u) | (((side.y >> 31) & 1u) ? 0xff0000u : 0u);
        const bool bad = !half1 && ((((pn[0] ^ (xr[0] >> 8)) & em0) | ((pn[1] ^ (xr[1] >> 8)) & em1)) != 0);
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && (t & 63) == 0) *(lds_u1*)(uintptr_t)my_flag = cseq;
    }
}
#undef LINK_IN
#undef LINK_MG
#undef LINK_SET

template <int MAXDEG>
__global__ __launch_bounds__(LDPC_SPLIT_T) __attribute__((amdgpu_waves_per_eu(LDPC_SPLIT_WPE))) void ldpc_split_kernel(LdpcKernelParams read_through_ldpc_params) {
    // (the arguments are read section by section through ldpc_params(), ldpc_lane_common.h: this by-value structure must stay the kernel's FIRST AND ONLY parameter --
    // ldpc_params() reads it at offset 0 of the kernel-argument segment)
    using S = SplitShape<MAXDEG>;
    constexpr int T = LDPC_SPLIT_T, REC = S::REC, NPW = S::NPW;
    extern __shared__ __attribute__((aligned(16))) int8_t lds_all[];     // (the kernel has no static LDS: the posteriors start at LDS offset 0, what the table's offsets count from)
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int8_t* __restrict__ post = lds_all;
    if (LDPC_SPLIT_BASE_PRIO) __builtin_amdgcn_s_setprio(LDPC_SPLIT_BASE_PRIO);

#if defined(LDPC_PROF) && LDPC_PROF == 3
    if (blockIdx.x == 0) g_prof_dev = ldpc_params()->A.prof;      // (threads 0 and 384 of workgroup 0 are the only readers)
    __syncthreads();
#endif
    int f = blockIdx.x;
    while (f < ldpc_params()->A.nframes) {
        {
            const LdpcKernelParamsPtr P = ldpc_params();
            const int N = P->A.N, K = P->A.K, R = P->A.R, q = P->A.q;
            uint32_t* __restrict__ msg = P->A.msg_ws + (size_t)blockIdx.x * (size_t)P->A.pent_base;
            const int8_t* __restrict__ src = P->A.llr + (size_t)f * N;
            int t = threadIdx.x;                  // (an opaque copy, as in the sweep: the strided indices of these loops, hoisted out of the frame loop, would stay live through every layer)
            asm volatile("" : "+v"(t));
            for (int i = t; i < K / 8; i += T) reinterpret_cast<uint2*>(post)[i] = reinterpret_cast<const uint2*>(src)[i];
            // parity LLRs: pty[360*i + jj] = llr[K + q*jj + i]   (layered_decoder.hh:124-126)
            for (int c = t; c < R; c += T) {
                int jj = c / q, i = c - jj * q;
                post[K + 360 * i + jj] = src[K + c];
            }
            // the first sweep reads all-zero messages: the records are cleared here, so that a sweep fetches them without asking which sweep it is
            for (uint32_t o = (uint32_t)t; o < (uint32_t)P->A.pent_base; o += T) msg[o] = 0;
            if (t < 2) reinterpret_cast<uint32_t*>(lds_all + ((N + LDPC_SPLIT_SCRATCH + 15) & ~15) + 8 * 360)[t] = 0;      // the chain layers' verdict words
        }
        __syncthreads();

        int it = 0, ret = 0;
        uint32_t cseq = 0;                  // chain layers of this frame so far (chain_layer: the verdict words)
        while (true) {
            const LdpcKernelParamsPtr P = ldpc_params();
            const LdpcKernelArgs A = ldpc_args(P);      // (what of it the check and the sweep use)
            const int N = A.N, q = A.q, npl = P->npl;
            // (the plan's tables through CONSTANT-address-space pointers: a uniform read is then a scalar load.  Through a plain pointer the compiler -- which sees the kernel
            // store to global memory -- issues a vector load, and the vmcnt(0) in front of its first use also waits for the table words just requested for the next pseudo-layer)
            const const_layer_ptr layers = (const_layer_ptr)P->layers;
            const const_u32_ptr ents = (const_u32_ptr)P->ents;
            const int npad = (N + LDPC_SPLIT_SCRATCH + 15) & ~15;                // posteriors + the scratch bytes (ldpc_split_plan.h)
            uint32_t* __restrict__ cw = reinterpret_cast<uint32_t*>(lds_all + npad);      // hand-off area: 8 bytes per row (chain walk) / 4 bytes per row + 2 flag words (speculative passes)
            int* __restrict__ s_flag = reinterpret_cast<int*>(lds_all + N);                // [12] + next frame, in the sweep's scratch bytes (LDPC_SPLIT_SCRATCH = 64)
            const uint32_t cflagb = (uint32_t)npad + 8u * 360u;                            // LDS address of the chain layers' two verdict words (the posteriors start at 0)
            uint32_t* __restrict__ msg = A.msg_ws + (size_t)blockIdx.x * (size_t)A.pent_base;
            uint32_t* __restrict__ sgn = A.sgn_ws + (size_t)blockIdx.x * SGN_WS_DWORDS;
            const __amdgpu_buffer_rsrc_t rs_tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(P->atab), 0, P->tab_words * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_msg = __builtin_amdgcn_make_buffer_rsrc(msg, 0, A.pent_base * 4, 0x00020000);
            const bool check = !A.force || it == A.max_trials;
            if (check) {
#if defined(LDPC_PROF) && LDPC_PROF == 5
                const unsigned long long tc0 = clock64();        // (where a check's time goes: prof[400..403], thread 0 of workgroup 0)
#endif
                const uint32_t zflag = sign_pack(post, N, reinterpret_cast<uint8_t*>(sgn), t, T);
#if defined(LDPC_PROF) && LDPC_PROF == 5
                const unsigned long long tc1 = clock64();
#endif
                __syncthreads();              // the sign bytes went to global memory: full barrier (drains vmcnt)
#if defined(LDPC_PROF) && LDPC_PROF == 5
                const unsigned long long tc2 = clock64();
#endif
                const bool bad = zflag != 0 || syndromes_bad<MAXDEG>(q, A.synd_base, P->ents, sgn, t, T);
                const unsigned long long b = __ballot(bad);
#if defined(LDPC_PROF) && LDPC_PROF == 5
                const unsigned long long tc3 = clock64();
#endif
                if ((t & 63) == 0) s_flag[t >> 6] = (b != 0);
                lds_barrier();
                int any = 0;
#pragma unroll
                for (int w = 0; w < T / 64; ++w) any |= s_flag[w];
                any = __builtin_amdgcn_readfirstlane(any);
                lds_barrier();                // (the flags are rewritten by the next check)
#if defined(LDPC_PROF) && LDPC_PROF == 5
                if (A.prof && blockIdx.x == 0 && t == 0) { const unsigned long long tc4 = clock64(); A.prof[400] += tc1 - tc0; A.prof[401] += tc2 - tc1; A.prof[402] += tc3 - tc2; A.prof[403] += tc4 - tc3; A.prof[404] += 1; }
#endif
                if (A.force) { ret = any ? -1 : A.max_trials; break; }
                if (!any) { ret = it; break; }
                if (it == A.max_trials) { ret = -1; break; }
            }
            // ---- one layered sweep (LDPCDecoder::update) = the plan's pseudo-layers in order, every wave in every one of them (ldpc_split_plan.h).  A thread's table entry and
            // message record are fetched one pseudo-layer ahead, descriptors travel two ahead.  The words fetched ahead are taken apart (RowState::unpack) at the top of their
            // pseudo-layer and the fetch for the one after it goes into the registers they arrived in: no copies ride round the loop.
            typename FetchWords<REC>::type rec_n;
            typename FetchWords<NPW>::type pw_n;
            LdpcSplitLayer Lnext = layer_at(layers, 0), Lnext2 = layer_at(layers, npl > 1 ? 1 : 0);
            uint32_t soff_tab = 0;
            pw_n = bload<NPW>(rs_tab, (uint32_t)t * (NPW * 4), soff_tab);
            rec_n = bload<REC>(rs_msg, (uint32_t)t * (REC * 4), Lnext.rec_off * 4u);
            // (claimed before the loop: a load still in flight on the loop's entry edge makes the compiler open EVERY pseudo-layer with vmcnt(0) -- which, coming round the
            // back edge, waits for the previous pseudo-layer's record store)
            asm volatile("" : "+v"(rec_n), "+v"(pw_n));
#if defined(LDPC_PROF)
            unsigned long long t_layer = clock64();       // development aid (-DLDPC_PROF builds): cycles of every pseudo-layer, thread 0 of workgroup 0 -> prof[128 + pl]; prof[127]: the rest of an iteration
            if (A.prof && blockIdx.x == 0 && t == 0) { A.prof[127] += t_layer - A.prof[126]; }
#endif
            for (int pl = 0; pl < npl; ++pl) {
                const LdpcSplitLayer L = Lnext;
                Lnext = Lnext2;
                Lnext2 = layer_at(layers, pl + 2 < npl ? pl + 2 : npl - 1);
                soff_tab += T * NPW * 4;
                // (everything a layer derives from the thread index is derived HERE, from a copy the compiler cannot see through: hoisted out of the loops these values
                // stay in registers across the whole kernel, and at 64 / 80 registers they are what gets spilled and reloaded inside the layers)
                int tt = t;
                asm volatile("" : "+v"(tt));
                const uint32_t voff_rec = (uint32_t)tt * (REC * 4), voff_tab = (uint32_t)tt * (NPW * 4);
                RowState<MAXDEG> RS;
                RS.unpack(pw_n, rec_n);
                // (the fetched words end HERE as far as the compiler can tell -- every value taken from them exists, and the registers are free for the fetch below: without
                // this the scheduler issues the fetch first, into a second register set, and copies it over at the bottom of the loop)
                RS.pin();
                asm volatile("" : "+v"(pw_n), "+v"(rec_n));
                if (pl + 1 < npl && !(LDPC_SPLIT_EXP & 2)) {
                    if (!(LDPC_SPLIT_EXP & 16)) pw_n = bload<NPW>(rs_tab, voff_tab, soff_tab);
                    if (!(LDPC_SPLIT_EXP & 32)) rec_n = bload<REC>(rs_msg, voff_rec, Lnext.rec_off * 4u);
                }
                uint32_t ro[REC];
                if (LDPC_SPLIT_SKIP && ((LDPC_SPLIT_SKIP >> (L.kind_nw & 0xffu)) & 1)) {      // (timing / counter experiments: the layers of these kinds do nothing)
#pragma unroll
                    for (int w = 0; w < REC; ++w) ro[w] = RS.rw;
#ifdef LDPC_SPLIT_NOPREV_RT
                } else if ((L.kind_nw & 0xffu) == 0 || (L.kind_nw & 0xffu) == 7) {
                    int M0, M1, SXs;
                    row_input<MAXDEG, 0, 2>(RS, 0u, L.aux, tt, M0, M1, SXs, (L.kind_nw >> 20) & 1u);
                    row_output<MAXDEG, 0>(RS, M0, M1, SXs, 0u, ro);
#endif
                } else if ((L.kind_nw & 0xffu) == 0) {
                    int M0, M1, SXs;
                    row_input<MAXDEG, 0>(RS, 0u, 0u, tt, M0, M1, SXs);
                    if (LDPC_SPLIT_EXP & 8) { ro[0] = (uint32_t)(M0 ^ M1 ^ SXs); ro[REC - 1] = ro[0]; }
                    else row_output<MAXDEG, 0>(RS, M0, M1, SXs, 0u, ro);
                } else if ((L.kind_nw & 0xffu) == 7) {
                    int M0, M1, SXs;
                    row_input<MAXDEG, 0, 1>(RS, 0u, L.aux, tt, M0, M1, SXs);
                    row_output<MAXDEG, 0>(RS, M0, M1, SXs, 0u, ro);
                } else if ((L.kind_nw & 0xffu) == 1) {
                    ++cseq;
                    chain_layer<MAXDEG>(RS, ro, L, ents[L.ent_off + 1], tt, post, cw, cflagb, cseq, A.dbg != 0);
                } else {
                    ++cseq;
                    // (a half of fewer than four slots has no speculative layers: the plan refuses them, ldpc_split_plan.h)
                    if constexpr (S::HS >= 4) spec_layer<MAXDEG>(RS, ro, L, P->atab, tt, cw, A.prof, pl, cflagb, cseq, A.dbg != 0);
                }
                // the next pseudo-layer's words are claimed HERE, before this one's record store is issued: the wait for them then sits where they have had a whole
                // pseudo-layer to arrive, and the top of the next pseudo-layer waits for nothing (ldpc_kernel.hip)
#if defined(LDPC_PROF) && LDPC_PROF == 4
                const unsigned long long t_claim = clock64();        // (how long the claim waits: prof[200 + pl])
#endif
                asm volatile("" : "+v"(rec_n), "+v"(pw_n));
#if defined(LDPC_PROF) && LDPC_PROF == 4
                if (A.prof && blockIdx.x == 0 && (t == 0 || t == 384)) A.prof[200 + pl + (t ? 50 : 0)] += clock64() - t_claim;
#endif
                if (LDPC_SPLIT_EXP & (2 | 64)) asm volatile("" :: "v"(ro[0]), "v"(ro[REC - 1]));
                else bstore<REC>(ro, rs_msg, voff_rec, L.rec_off * 4u);
                lds_pairs_wait();
                if (!(LDPC_SPLIT_EXP & 4)) lds_barrier();
#if defined(LDPC_PROF)
                if (A.prof && blockIdx.x == 0 && t == 0) { const unsigned long long t_now = clock64(); A.prof[128 + pl] += t_now - t_layer; t_layer = t_now; A.prof[126] = t_now; }
#endif
            }
            ++it;
        }

        // ---- outputs
        {
            const LdpcKernelParamsPtr P = ldpc_params();
            const LdpcKernelArgs A = ldpc_args(P);
            const int N = A.N, K = A.K, R = A.R, q = A.q;
            int* __restrict__ s_flag = reinterpret_cast<int*>(lds_all + N);
            int t = threadIdx.x;
            asm volatile("" : "+v"(t));
            if (t == 0) A.trials[f] = ret;
            // hard decisions of [0,K): 64 bits per wave step via ballot, MSB-first bytes (module_dvbs2_demod.cpp:357-360)
            uint8_t* __restrict__ hd = A.hard + (size_t)f * A.hard_stride;
            const int lane = t & 63;
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
            if (A.work_ctr) {
                if (t == 0) s_flag[12] = (int)(gridDim.x + atomicAdd(A.work_ctr, 1u));
                lds_barrier();
                f = __builtin_amdgcn_readfirstlane(s_flag[12]);
            } else {
                f += gridDim.x;
            }
        }
        __syncthreads();
    }
}

// posteriors + scratch, then the hand-off area: 8 bytes x 360 rows (spec_layer's cells and flags end at 4 * 384 + 8 bytes of it)
size_t ldpc_split_lds_bytes(int N) {
    const size_t npad = (size_t)((N + LDPC_SPLIT_SCRATCH + 15) & ~15);
    return npad + 8 * 360 + 16;                 // (+ the chain layers' two verdict words)
}
size_t ldpc_split_msg_bytes_per_block(const LdpcDeviceCode& C) { return (size_t)C.split_rec_total * sizeof(uint32_t); }

template <int MAXDEG>
static hipError_t launch_split(const LdpcDeviceCode& C, const LdpcKernelArgs& A, int grid, hipStream_t stream) {
    const size_t lds = ldpc_split_lds_bytes(A.N);
    auto kern = ldpc_split_kernel<MAXDEG>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    LdpcKernelParams P;
    P.A = A; P.layers = C.d_split_layers; P.ents = C.d_ents; P.atab = C.d_split_atab; P.rows = C.d_rows; P.npl = C.split_npl; P.tab_words = C.split_tab_words;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_SPLIT_T), lds, stream, P);
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
        case 2: return FN<2>(__VA_ARGS__);              \
        case 4: return FN<4>(__VA_ARGS__);              \
        case 5: return FN<5>(__VA_ARGS__);              \
        case 8: return FN<8>(__VA_ARGS__);              \
        case 9: return FN<9>(__VA_ARGS__);              \
        case 12: return FN<12>(__VA_ARGS__);            \
        default: break;                                 \
    }

extern unsigned long long* g_ldpc_prof;   // (ldpc_kernel.hip)
bool ldpc_split_supported(int max_deg) { return max_deg == 2 || max_deg == 4 || max_deg == 5 || max_deg == 8 || max_deg == 9 || max_deg == 12; }
bool ldpc_split_noprev_shared(int max_deg) { return max_deg == 4 || max_deg == 8; }      // SplitShape::NOPREV_SHARED
int ldpc_split_blocks_per_cu(int max_deg, int N) {
    LDPC_SPLIT_DISPATCH(occupancy_split, N)
    return 1;
}

hipError_t ldpc_split_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force, uint8_t* hard, int hard_stride,
                                    int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid, hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws, int dbg) {
    LdpcKernelArgs A;
    A.work_ctr = work_ctr;
    A.sgn_ws = sgn_ws;
    if (work_ctr) {
        hipError_t e = hipMemsetAsync(work_ctr, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
    }
    A.llr = llr; A.hard = hard; A.post = post; A.trials = trials; A.msg_ws = msg_ws;
    A.nframes = nframes; A.N = C.N; A.K = C.K; A.R = C.R; A.q = C.q; A.pent_base = C.split_rec_total; A.synd_base = C.synd_base;
    A.max_trials = max_trials; A.force = force; A.hard_stride = hard_stride; A.dbg = dbg;
    A.prof = g_ldpc_prof;
    const int max_deg = C.max_deg;
    LDPC_SPLIT_DISPATCH(launch_split, C, A, grid, stream)
    return hipErrorInvalidValue;
}

}  // namespace s2
