// Layered offset-min-sum LDPC decoder for DVB-S2, gfx950.
//
// Replaces BBFrameLDPC::decode (reference src/demod/dvbs2/codings/bbframe_ldpc.cpp:123-139) and the
// library under it (xdsopl-ldpc-pabr/layered_decoder.hh:23-133, algorithms.hh:206-277) -- bit-exact,
// including the sequential row order (see ldpc_plan.h) and the int8 saturation rules.
//
// Mapping (one workgroup = TWO frames in lockstep, persistent over the batch; see ldpc_decode_kernel below):
//   * 768 threads = 12 waves; threads [0,384) own frame slot 0, [384,768) slot 1; lane j < 360 of a slot owns row j of
//     EVERY layer (a DVB-S2 layer = 360 rows, quasi-cyclic).
//   * the N int8 posteriors of a frame live in LDS for the whole decode (64.8 KB per normal frame); information bits as
//     [0,K), parity bits layer-major as K + 360*i + j.  Row j of a layer reads byte 360*r + (j - s) mod 360 of each linked
//     group: consecutive lanes -> consecutive bytes, conflict-free.
//   * check->bit messages: one fixed-size record per row (REC dwords, 1 byte per link) in a per-slot global workspace.
//     Lane j re-reads only what lane j wrote, one coalesced vector load + store per row per iteration, prefetched one
//     layer ahead; the workspace of all resident workgroups (512 slots x ~260 KB) stays in the Infinity Cache.
//   * rows of a layer that share a bit follow the plan (ldpc_plan.h): chain walk for a single shared pair, levels otherwise.
//   * arithmetic: packed int16 ("Q8", two links per VALU instruction) with the int8 saturation rules of the reference's
//     SIMD lanes; the syndrome check before an iteration works on bit-packed sign vectors (quasi-cyclic: a layer's 360
//     syndromes are XORs of cyclic shifts of 360-bit groups).
// Roofline: algorithmic bytes per frame = iters*4*edges + N + K/8 against HBM 8 TB/s is the NOMINAL figure; the
// state is on-chip (LDS + Infinity Cache: 26 MB of fabric traffic per frame, 0.58 x the algorithmic bytes).  What the kernel
// spends is VECTOR-ALU time: its packed 16-bit / VOP3 / SDWA / DPP instructions issue at half the rate of plain 32-bit ones
// (4.3 against 2.3 cycles per wave instruction and SIMD, profiles/r05_valu_rates.txt): 1.53e10 of them per 4096 frames x 50
// iterations keep the SIMDs busy for 57 % of the launch; the rest is the latency of a layer's barrier-separated phases with
// twelve waves per compute unit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_lane_common.h"

namespace s2 {


#if defined(LDPC_PROF) && LDPC_PROF == 3
// waypoints inside a layer step, first lane of wave 0 and of wave 5 of slot 0 (workgroup 0): prof[200 + 64*kind + 16*(wave==5) + i] = cycles from the
// previous waypoint; kind 0 = free layer, 1 = chain layer.  i: 0 input phase, 1 hand-off, 2 barrier, 3 walk, 4 barrier, 5 late links, 6 output phase, 7 layer barrier
#define PROF_MARK(kind, i) do { if ((kind) < 2 && blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 320)) { unsigned long long t_now = clock64(); A.prof[200 + 64 * (kind) + (threadIdx.x ? 16 : 0) + (i)] += t_now - t_mark; t_mark = t_now; } } while (0)
#define PROF_MARK_DECL unsigned long long t_mark = clock64()
#define PROF_MARK_PTR (&t_mark)
#else
#define PROF_MARK_PTR nullptr
#define PROF_MARK(kind, i) do { } while (0)
#define PROF_MARK_DECL do { } while (0)
#endif
#if defined(LDPC_PROF) && (LDPC_PROF == 2 || LDPC_PROF == 3)
// one probe per layer (after its barrier, wave 0 of workgroup 0): prof[128 + layer] = cycles of that layer step, prof[127] = the rest of
// an iteration (check, loop top); the fine-grained probes below cost ~250 cycles each and distort what they measure
#define PROF_T(var) do { } while (0)
#define PROF_ADD(slot, t0, t1) do { } while (0)
#define PROF_LAYER(idx) do { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_now = clock64(); A.prof[128 + (idx)] += t_now - t_layer; t_layer = t_now; } } while (0)
#define PROF_LAYER_DECL unsigned long long t_layer = clock64()
#elif defined(LDPC_PROF)
#define PROF_T(var) unsigned long long var = clock64()
#define PROF_ADD(slot, t0, t1) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) A.prof[((threadIdx.x >> 6) % 6) * 16 + (slot)] += (t1) - (t0); } while (0)
#define PROF_LAYER(idx) do { } while (0)
#define PROF_LAYER_DECL do { } while (0)
#else
#define PROF_LAYER(idx) do { } while (0)
#define PROF_LAYER_DECL do { } while (0)
#define PROF_T(var) do { } while (0)
#define PROF_ADD(slot, t0, t1) do { } while (0)
#endif

#ifndef LDPC_ADDR_TABLE
#define LDPC_ADDR_TABLE 1   // regular codes of degree 2, 8 and 12 (ldpc_plan.h: ldpc_atab_degree): the LDS addresses of a row's links come from a per-code table [layer][row][pair] (two 16-bit
                            // addresses per word, fetched a layer ahead like the message record) instead of four packed VALU operations per pair:
                            // four half-rate vector instructions per pair (17 SIMD cycles) against 4 bytes per lane through the 64 B/clk vector-memory path (measured per kernel below)
#endif
// (measured per kernel, 4096 frames x 50 iterations: degree 12 (3/4) 44.1 -> 42.5 ms, degree 8 (2/3) 42.1 -> 39.6, degree 2 (1/4) 38.0 -> 36.1; the
// kernels of degree 3, 4, 5 and 9 -- many short layers: the fetch a layer ahead no longer hides behind the layer -- LOSE 8-23 % and keep the arithmetic)
template <int MAXDEG, bool IRREG> constexpr bool ldpc_use_atab() { return LDPC_ADDR_TABLE && !IRREG && ldpc_atab_degree(MAXDEG); }
// a row's words of the address table: NPI pairs, stored with a stride of LDPC_ATAB_STRIDE(NPI) words so that one or two aligned vector loads fetch them
__device__ __forceinline__ constexpr int ldpc_atab_stride(int npi) { return npi <= 1 ? 1 : npi <= 2 ? 2 : npi <= 4 ? 4 : 8; }
template <int NPI, int NPW>
__device__ __forceinline__ void atab_load(uint32_t (&pw)[NPW], const uint32_t* __restrict__ p) {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (NPI <= 1) { w[0] = p[0]; }
    else if constexpr (NPI <= 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y; }
    else {
        const uint4 v = *reinterpret_cast<const uint4*>(p); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        if constexpr (NPI > 4) { const uint4 u = *reinterpret_cast<const uint4*>(p + 4); w[4] = u.x; w[5] = u.y; w[6] = u.z; w[7] = u.w; }
    }
#pragma unroll
    for (int i = 0; i < NPI; ++i) pw[2 * i] = w[i];
}
#ifndef LDPC_DIET
#ifndef LDPC_OTHER_SUM
#define LDPC_OTHER_SUM 1   // the other-minimum selection of the output phase as a sum minus a minimum (2 packed operations per link pair instead of 3).  Frames/s, 1024 frames x 50
#endif                     // iterations, without -> with: normal 1/3 77 694 -> 82 151, 2/5 103 492 -> 104 048, 1/2 103 140 -> 106 015, 3/5 66 843 -> 67 977, 2/3 102 926 -> 106 170,
                           // 4/5 80 950 -> 82 588, 5/6 73 178 -> 74 543, 8/9 97 007 -> 99 090, 9/10 83 223 -> 83 865, 1/4 118 119 -> 117 120; short 1/4 314 901 -> 335 881, the other
                           // short codes within 1 %; the degree-12 kernel (3/4 normal, at the register cap: a different allocation eats it) 93 348 -> 90 917 -- it keeps the product form
template <int MAXDEG, bool IRREG> constexpr bool ldpc_other_sum() { return LDPC_OTHER_SUM && MAXDEG != 12; }
#define LDPC_DIET 1   // two cuts in a layer's per-wave instruction stream (what a layer costs): (1) the layer-ahead fetches -- message record, address / row words --
                      // are issued unconditionally by every lane (idle lanes with a clamped row index; the records are cleared at the start of a frame, so the first sweep needs no "is
                      // this the first sweep" either): no exec-mask juggling and, above all, no "fetched / not fetched" merge of the registers, which cost a copy of every prefetch register
                      // at the top AND the bottom of the layer loop (20 v_mov); (2) the parity bits' LDS addresses follow the layer by one add instead of being rebuilt from the layer number.
#endif
// Measured per kernel (1024 frames x 50 iterations, frames/s): degree 2 (1/4) 114 441 -> 118 196, degree 5 (1/2) 99 026 -> 102 814, degree 28 (9/10) 79 153 -> 83 527;
// degree 8 (2/3) 102 711 -> 100 674 and degree 12 (3/4) 92 541 -> 91 892: the two kernels that sit AT the 128-register cap lose what they gain to a different allocation and keep the old form
template <int MAXDEG, bool IRREG> constexpr bool ldpc_diet() { return LDPC_DIET && (IRREG || (MAXDEG != 8 && MAXDEG != 12)); }
#ifndef LDPC_WPE4_MAXDEG
#define LDPC_WPE4_MAXDEG 28    // kernels up to this degree are held to 128 VGPRs (4 waves per SIMD: room for a front-end wave beside three decoder waves)
#endif
#define LDPC_CW_DWORDS(maxdeg) ((maxdeg) > 12 ? 3 : 2)

// One sweep step for one layer.  CONF = layer has intra-layer shared bits (links 0..nc-1), IRREG = the
// code has layers of different degree (short tables C1, C4, C7, C8, C9).
// Links are processed in PAIRS held in packed int16 registers: pair p = links 2p, 2p+1 of the row, where links
// [0, MAXDEG) are the table links, MAXDEG the row's own parity bit and MAXDEG+1 the previous parity bit.
template <int MAXDEG, int REC, int KIND, bool IRREG>
__device__ __forceinline__ void layer_update(int8_t* __restrict__ post, const LdpcKernelArgs& A, const uint32_t* __restrict__ ents,
                                             const uint32_t (&pw)[2 * ((MAXDEG + 1) / 2)], const LdpcLayerDesc L, uint32_t rowword, int layer, int j, bool active, bool live,
                                             const uint32_t (&rec_in)[REC], uint32_t (&rec_out)[REC],
                                             uint32_t* __restrict__ cw, uint8_t* __restrict__ cres, uint32_t own_a, const uint32_t* __restrict__ walk = nullptr, unsigned long long* t_mark_p = nullptr) {
#if defined(LDPC_PROF) && LDPC_PROF == 3
    unsigned long long& t_mark = *t_mark_p;
#endif
    constexpr int NL = MAXDEG + 2;
    constexpr int NP = (NL + 1) / 2;
    // KIND 0: no shared bits in the layer; 1: one shared pair resolved by the chain walk (links 0, 1 only); 2: general levels;
    // 3: levels, at most 4 shared links
    constexpr bool CONF = KIND != 0;
    constexpr int MAXC_ALL = MAXDEG < LDPC_MAX_CONFLICT_LINKS ? MAXDEG : LDPC_MAX_CONFLICT_LINKS;
    // KIND 3 / 4 = KIND 2 for layers with at most 4 / 8 shared links: the per-level code only tests those (every tested link costs scalar
    // branches per level whether or not a row uses it)
    // KIND 6 = "quad walk" layers (at most 4 shared links, deep and narrow level structure): one wave walks the rows of levels >= 2 in
    // level order, four lanes per row, see below
    constexpr int MAXC = KIND == 1 ? (MAXDEG < 2 ? MAXDEG : 2) : KIND == 6 ? (MAXDEG > 12 ? 8 : (MAXDEG < 4 ? MAXDEG : 4)) : KIND == 3 ? (MAXDEG < 4 ? MAXDEG : 4) : KIND == 4 ? (MAXDEG < 8 ? MAXDEG : 8) : MAXC_ALL;
    s16x2 V[NP], G[NP];        // extrinsic inputs and their offset magnitudes
    uint32_t addr[MAXDEG];     // LDS byte addresses of the table links' posteriors
    const uint32_t lbase = lds_offset(post);
    const int deg = IRREG ? (int)(L.deg & 0xffffu) : MAXDEG;
    const int nc = CONF ? (int)(L.depth_nc >> 16) : 0;
    const uint32_t level = rowword & 0xffu, late = (rowword >> 8) & 0xfffu, early = rowword >> 20;
    int min0 = 255, min1 = 255, sx = 0;
    constexpr bool DIET = ldpc_diet<MAXDEG, IRREG>();
    const bool has_prev = (layer | j) != 0;
    // LDS addresses of the row's own parity bit and of the previous one (a layer back; layer 0: the last layer's, a row back -- row 0 of layer 0 has none: a valid byte, masked below).
    // DIET: own_a is kept by the caller (+ 360 per layer)
    const uint32_t own_l = DIET ? own_a : lbase + (uint32_t)(A.K + 360 * layer + j);
    const uint32_t prev_l = DIET ? own_a + (uint32_t)(layer ? -360 : 360 * (A.q - 1) - 1)
                                 : lbase + (uint32_t)(has_prev ? (layer ? A.K + 360 * layer + j - 360 : A.K + 360 * (A.q - 1) + j - 1) : A.K + 360 * layer + j);
#define LINK_IN(k) ((int)V[(k) >> 1][(k) & 1] >> 8)
#define LINK_MG(k) ((int)G[(k) >> 1][(k) & 1] >> 8)
#define LINK_SET(k, v, m) do { V[(k) >> 1][(k) & 1] = (short)((v) << 8); G[(k) >> 1][(k) & 1] = (short)((m) << 8); } while (0)
    PROF_T(t_a);
    // Conflict kinds: EVERY lane of a live slot runs the input phase (idle lanes read valid LDS and never store).  Its results live
    // across the barriers of the middle section; defined under a divergent condition they cost ~36 register initialisations per layer
    // and wave.  A slot that is not live (frame finished or absent) only keeps the barrier count of its partner.
    if constexpr (CONF) {
        if (!live) {
            if constexpr (KIND == 1 || KIND == 6) { lds_barrier(); lds_barrier(); }
            else { const int depth = (int)(L.depth_nc & 0xffffu); for (int lvl = 2; lvl <= depth; ++lvl) lds_barrier(); }
            return;
        }
    }
    if (CONF || active) {
        s16x2 MIN0 = splat2(Q8_NONE), MIN1 = splat2(Q8_NONE);
        uint32_t SX = 0;
        uint32_t XR[NP], XH[NP];
        const uint32_t JJ = (uint32_t)j * 0x10001u;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            bool absent[2] = {false, false};     // uniform (table) absence; the missing previous parity bit of row 0 is per lane
            uint32_t la[2] = {lbase, lbase};
            if (2 * p < MAXDEG) {
                uint32_t AD;
                if constexpr (ldpc_use_atab<MAXDEG, IRREG>()) {
                    AD = pw[2 * p];          // this row's two addresses, straight from the code's address table (ldpc_plan.h)
                } else {
                    // both table links of the pair at once: (j + sp) mod 360 + 360*r in packed uint16 (pair table, ldpc_plan.h)
                    u16x2 T = __builtin_bit_cast(u16x2, JJ) + __builtin_bit_cast(u16x2, pw[2 * p]);
                    T = __builtin_elementwise_min(T, (u16x2)(T - (u16x2){360, 360}));
                    AD = __builtin_bit_cast(uint32_t, (u16x2)(T + __builtin_bit_cast(u16x2, pw[2 * p + 1])));
                }
                la[0] = lbase + (AD & 0xffffu);
                la[1] = lbase + (AD >> 16);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * p + h;
                if (k < MAXDEG) {
                    addr[k] = la[h];
                    if (IRREG && k >= deg) absent[h] = true;
                } else if (k == MAXDEG) {
                    la[h] = own_l;
                } else if (k == MAXDEG + 1) {
                    la[h] = prev_l;
                } else {
                    absent[h] = true;
                }
            }
            lds_read_pair_i8(la[0], la[1], XR[p], XH[p]);
        }
        PROF_MARK(KIND, 8);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p == 1) PROF_MARK(KIND, 9);
            if (p == NP - 1) PROF_MARK(KIND, 10);
            // staged wait: LDS returns in order, so pair p is complete once at most 2*(NP-1-p) operations are outstanding (anything
            // else counted in lgkmcnt only makes this stricter) -- the arithmetic of the first pairs overlaps the later reads
            lds_pair_ready(2 * (NP - 1 - p), XR[p], XH[p]);
            // byte of the low load -> bits 15:8, byte of the high load (it sits in bits 23:16) -> bits 31:24
            const s16x2 X = from_bits2(__builtin_amdgcn_perm(XH[p], XR[p], 0x060c000cu));
            bool absent[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * p + h;
                absent[h] = (k < MAXDEG) ? (IRREG && k >= deg) : (k > MAXDEG + 1);
            }
            s16x2 v = sat_sub2(X, rec_pair<REC>(rec_in, 2 * p));                     // int8 saturation by the 16-bit clamp
            const s16x2 av = pmax2(v, sat_sub2(splat2(0), v));
            // mag_of: |v| - 1 clamped at 0.  No upper clamp: |v| <= 127 + 255/256 here, so the high byte -- all that is ever consumed -- is <= 126;
            // the low byte is 0xff instead of 0 only where |v| saturated, i.e. at magnitude 126, the largest there is: min / max / equality
            // against other halves can then only confuse 126 with 126 (and a row whose smallest magnitude is 126 has its second smallest there too)
            s16x2 g = __builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, av), (u16x2){256, 256}));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * p + h;
                if (absent[h]) { v[h] = 0; g[h] = (short)Q8_NONE; }
                if (k == MAXDEG + 1 && !has_prev) { v[h] = 0; g[h] = (short)Q8_NONE; }
                if constexpr (CONF) {
                    if (k < MAXC && k < nc && ((late >> k) & 1)) { v[h] = 0; g[h] = (short)Q8_NONE; }   // joins the totals at its level
                }
            }
            V[p] = v; G[p] = g;
            // two smallest magnitudes per half; every g is <= the "no link" sentinel the trackers start from, so the first two pairs need
            // no comparison against it
            if (p == 0) { MIN0 = g; }
            else if (p == 1) { MIN1 = pmax2(MIN0, g); MIN0 = pmin2(MIN0, g); }
            else { MIN1 = pmin2(MIN1, pmax2(MIN0, g)); MIN0 = pmin2(MIN0, g); }
            SX ^= bits2(v);
        }
        // merge the even-link and odd-link halves
        const int a0 = MIN0[0] >> 8, b0 = MIN0[1] >> 8, a1 = MIN1[0] >> 8, b1 = MIN1[1] >> 8;
        min0 = min(a0, b0);
        min1 = min(max(a0, b0), min(a1, b1));
        sx = (int)(short)(SX ^ (SX >> 16));
    }
    PROF_T(t_b);
    PROF_ADD(CONF ? 4 : 0, t_a, t_b);
    PROF_MARK(KIND, 0);
    if constexpr (CONF) {
        const int chain_d = (int)(L.deg >> 16);
        if constexpr (KIND == 1) {
            // ---- chain walk (single shared pair, links 0 = E, 1 = L; see ldpc_plan.h).  Level-1 rows (j < d) publish
            // their early links; every row whose L link is late leaves {exclusive min over its other links, in_E,
            // old L message, sign} in cw[]; lanes c < d of wave 0 then walk rows c+d, c+2d, ...: the posterior written
            // by one row's E link is the next row's L input and stays in a register.  new_msg() of a link equals
            // sign * (minimum magnitude over the row's OTHER links) -- that is what (mag==min0 ? min1 : min0) selects.
            PROF_T(t_m0);
            if (active) {
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
                    // hand-off record in the walk's own form (see chain_step)
                    const int qE = (late & 1u) ? 255 : ((LINK_MG(0) == min0) ? min1 : min0);
                    const ChainRec r = chain_record(rec_byte<REC>(rec_in, 1), qE, LINK_IN(0), (sx ^ LINK_IN(0)) >> 31);
                    reinterpret_cast<uint2*>(cw)[j] = make_uint2(r.lim, r.se);
                }
            }
            PROF_T(t_m1);
            PROF_MARK(KIND, 1);
            lds_barrier();
            PROF_T(t_m2);
            PROF_MARK(KIND, 2);
            if (j < chain_d && active) {
                // lane c walks rows c + k*d, k = 1..T with T = floor(359 / d): rows k < T exist for every lane, row T only
                // where c + T*d < 360.  Only this wave is running (the others wait at the barrier), so the walk is bound by
                // the instructions it issues: uniform trip count (scalar loop), two rows per trip, hand-off records fetched two
                // rows ahead, plain address increments.  (A hand-scheduled assembly version of this loop -- 11 instructions
                // per row -- was bit-exact on its own and 1 % faster, but produced wrong frames when the front-end kernels of
                // the pipelined mode shared the CUs; the cause was not found, so the loop is left to the compiler.)
                // the walker is the only wave of its slot that runs: co-resident front-end waves (priority 3) must not starve it
                __builtin_amdgcn_s_setprio(3);
                const uint32_t eL = ents[1];
                const int T = 359 / chain_d;
                int x = post[link_addr(eL, j + chain_d)];                 // written by the level-1 row j (its E link)
                const uint2* c = reinterpret_cast<const uint2*>(cw) + j + chain_d;
                uint8_t* pr = reinterpret_cast<uint8_t*>(cres) + j + chain_d;
                // records are fetched two rows ahead (those beyond row 359 are read but never used: LDS reads past the allocation
                // return zeros).  A four-row distance with hand-issued LDS traffic was tried: the long chain (d = 2) gained 7 %, every
                // short one lost 5 % to the longer prologue.  The row costs ~85 cycles because ONE wave pays ~18 cycles of issue per
                // LDS instruction and 5.4 per SDWA / VOP3 one (tools/ubench/valu.hip), not because it waits for its record.
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
                if (j + T * chain_d < 360) pr[0] = (uint8_t)x;
                __builtin_amdgcn_s_setprio(0);
            }
            PROF_T(t_m3);
            PROF_MARK(KIND, 3);
            lds_barrier();
            PROF_T(t_m4);
            PROF_MARK(KIND, 4);
            PROF_ADD(9, t_m0, t_m1); PROF_ADD(10, t_m1, t_m2); PROF_ADD(11, t_m2, t_m3); PROF_ADD(12, t_m3, t_m4);
            if (active) {
                // both late inputs are fetched before either is used (one LDS round trip instead of two behind the branches)
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
            (void)chain_d;
            // ---- quad walk (ldpc_plan.h).  Rows of level 1 publish their early links; every other row leaves a hand-off record in
            // cw[]: {min0, min1, sign of the totals over its links known so far, late mask, early mask} and one byte per shared link (its
            // old message if the link is late, else its input value).  Wave 0 of the slot then walks the plan's step list: 16 rows per
            // step, lane = (row, shared link).  A lane fetches its link's posterior, forms the link value (late links) and its magnitude;
            // the four lanes of a row join the late links into the row totals with quad permutes; lanes of early links write the new
            // posterior.  Steps of one level are independent, steps of consecutive levels are ordered by the in-order LDS pipeline of
            // the one wave -- no workgroup barrier per level.  Late link values go back into the record for the output phase.
            // (what the walker needs from global memory is fetched before the hand-off, so that its latency overlaps the barrier)
            // Kernels for degree > 12 also walk layers with up to 8 shared links, eight lanes per row (8 rows per step, 12-byte records).
            constexpr int WL = MAXDEG > 12 ? 8 : 4;                 // most lanes per row this kernel handles
            constexpr int CWD = LDPC_CW_DWORDS(MAXDEG);             // record size in dwords: header + WL link bytes
            const uint32_t whd = walk[0];
            const int wk_steps = (int)(whd & 0xffffu);
            const int lpr = WL == 8 ? (int)(whd >> 16) : 4;         // lanes per row of THIS layer (4 or 8)
            const int wk_k = j & (lpr - 1);
            const uint32_t wk_ent = ents[wk_k < nc ? wk_k : 0];
            if (active) {
                if (level == 1u) {
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        if (k < nc && ((early >> k) & 1)) {
                            int nm = new_msg(LINK_IN(k), LINK_MG(k), min0, min1, sx);
                            LDS_I8(addr[k]) = (int8_t)clamp8(LINK_IN(k) + nm);
                        }
                    }
                } else {
                    uint32_t lb[WL / 4];
#pragma unroll
                    for (int w = 0; w < WL / 4; ++w) lb[w] = 0;
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        const int b = (k < nc && ((late >> k) & 1)) ? rec_byte<REC>(rec_in, k) : LINK_IN(k);
                        lb[k >> 2] |= ((uint32_t)b & 0xffu) << (8 * (k & 3));
                    }
                    // header: min0, min1 (7 bits each: 127 stands for "none", every real magnitude is <= 126), sign of the totals, late mask, early mask
                    const uint32_t hd = (uint32_t)min(min0, 127) | ((uint32_t)min(min1, 127) << 7) | (((uint32_t)sx >> 31) << 14) | ((late & 0xffu) << 15) | ((early & 0xffu) << 23);
                    cw[CWD * j] = hd;
#pragma unroll
                    for (int w = 0; w < WL / 4; ++w) cw[CWD * j + 1 + w] = lb[w];
                }
            }
            lds_barrier();
            if (j < 64 && live) {
                __builtin_amdgcn_s_setprio(3);
                const int k = wk_k, q = WL == 8 ? (lpr == 8 ? j >> 3 : j >> 2) : j >> 2;
                const uint32_t ek = wk_ent;
                const int spk = (int)(ek & 0xffffu);
                const uint32_t basek = lbase + 360u * (ek >> 16);
                const uint32_t scratch = lds_offset(reinterpret_cast<const int8_t*>(cres)) + (uint32_t)j;
                const uint32_t cwb = lds_offset(reinterpret_cast<const int8_t*>(cw));
                const int nsteps = wk_steps;
                const uint32_t* __restrict__ list = walk + 1 + q;
                // Two steps per trip, each with its own list register fetched two steps ahead (the list is followed by three empty steps, so
                // the fetches need no bound test): the wait for the entry in use leaves the one younger fetch in flight.
                auto step = [&](const uint32_t e) {
                    const bool valid = e != 0xffffffffu && k < nc;
                    const int row = valid ? (int)e : 0;
                    const uint32_t ra = cwb + (uint32_t)(4 * CWD) * (uint32_t)row;
                    uint32_t hd;
                    int b;
                    if constexpr (CWD == 2) {        // header and link bytes in one 8-byte fetch
                        const uint2 r = reinterpret_cast<const uint2*>(cw)[row];
                        hd = r.x;
                        b = (int)__builtin_amdgcn_sbfe((int)r.y, 8 * k, 8);
                    } else {
                        hd = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)ra;
                        b = (int)LDS_I8(ra + 4u + (uint32_t)k);
                    }
                    int t = row + spk;
                    t = (int)min((uint32_t)t, (uint32_t)(t - 360));
                    const uint32_t a = basek + (uint32_t)t;
                    const int x = (int)LDS_I8(a);
                    const bool lt = valid && ((hd >> (15 + k)) & 1u), er = valid && ((hd >> (23 + k)) & 1u);
                    const int v = lt ? clamp8(x - b) : b;
                    const int g = mag_of(v);
                    // the row's late links joined across its lanes: two smallest magnitudes and the sign
                    int m0 = lt ? g : 255, m1 = 255, sg = lt ? v : 0;
#define QUAD(x_, ctrl) __builtin_amdgcn_update_dpp(0, (x_), (ctrl), 0xf, 0xf, true)
#define JOIN(ctrl) do { const int o0 = QUAD(m0, ctrl), o1 = QUAD(m1, ctrl); m1 = min(max(m0, o0), min(m1, o1)); m0 = min(m0, o0); sg ^= QUAD(sg, ctrl); } while (0)
                    JOIN(0xB1);                                                              // quad_perm [1,0,3,2]
                    JOIN(0x4E);                                                              // quad_perm [2,3,0,1]
                    if (WL == 8 && lpr == 8) JOIN(0x141);                                    // row_half_mirror: the other quad of the 8-lane group
#undef JOIN
#undef QUAD
                    const int q0 = (int)(hd & 0x7fu), q1 = (int)((hd >> 7) & 0x7fu);
                    const int t1 = min(max(m0, q0), min(m1, q1)), t0 = min(m0, q0);
                    const int ss = sg ^ (int)(hd << 17);                                     // bit 31 = sign of the row's totals
                    const int nm = new_msg(v, g, t0, t1, ss);
                    LDS_I8(er ? a : scratch) = (int8_t)clamp8(v + nm);
                    LDS_I8(valid ? ra + 4u + (uint32_t)k : scratch) = (int8_t)v;
                };
                // (issued by hand: the compiler sinks such a fetch to its use, or copies registers behind it, and then waits for it)
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
            if (active && level != 1u) {
                uint32_t lb[WL / 4];
#pragma unroll
                for (int w = 0; w < WL / 4; ++w) lb[w] = cw[CWD * j + 1 + w];
#pragma unroll
                for (int k = 0; k < MAXC; ++k) {
                    if (k < nc && ((late >> k) & 1)) {
                        int v = (int)__builtin_amdgcn_sbfe((int)lb[k >> 2], 8 * (k & 3), 8);
                        int m = mag_of(v);
                        LINK_SET(k, v, m);
                        ROW_ACCUM(v, m);
                    }
                }
            }
        } else {
        (void)chain_d;
        // rows of level 1 have complete totals: they publish the links a later row waits for right away
        const int depth = (int)(L.depth_nc & 0xffffu);
        for (int lvl = 1; lvl <= depth; ++lvl) {
            if (lvl > 1) lds_barrier();
            if (active && level == (uint32_t)lvl) {
                __builtin_amdgcn_s_setprio(3);       // few waves have rows at a level and everyone waits for them (see the chain walk)
                // (blocking LICM of the per-link mask tests with an empty asm was measured slower on MI355X: r01 A/B variant "e")
                const uint32_t late_l = late, early_l = early;
                if (lvl > 1) {
                    // (the <= 4 / <= 8-link variants fetch all their shared posteriors up front: one LDS round trip per level instead
                    // of one per late link behind the per-link branches)
                    int xs[MAXC];
                    if constexpr (KIND >= LDPC_LEVEL_PREFETCH_KIND) {
#pragma unroll
                        for (int k = 0; k < MAXC; ++k) xs[k] = (int)LDS_I8(addr[k]);
                    }
#pragma unroll
                    for (int k = 0; k < MAXC; ++k) {
                        if (k < nc && ((late_l >> k) & 1)) {
                            int v = clamp8((KIND >= LDPC_LEVEL_PREFETCH_KIND ? xs[k] : (int)LDS_I8(addr[k])) - rec_byte<REC>(rec_in, k));
                            int m = mag_of(v);
                            LINK_SET(k, v, m);
                            ROW_ACCUM(v, m);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < MAXC; ++k) {
                    if (k < nc && ((early_l >> k) & 1)) {
                        int nm = new_msg(LINK_IN(k), LINK_MG(k), min0, min1, sx);
                        LDS_I8(addr[k]) = (int8_t)clamp8(LINK_IN(k) + nm);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        }
    }
    PROF_T(t_c);
    PROF_ADD(KIND >= 2 ? 8 : (CONF ? 5 : 1), t_b, t_c);
    PROF_MARK(KIND, 5);
    if (active) {
        // equality test against the true minimum; the selected magnitude is limited to 32 once per row (the per-link clamp to
        // [-32, 31] then only needs its upper side)
        const int min0c = min(min0, 32), min1c = min(min1, 32);
        const s16x2 MIN0B = q8(min0), MIN1CB = q8(min1c), NDB = q8(min0c - min1c);
        const s16x2 SUMCB = q8(min0c + min1c);
        const uint32_t SXB = ((uint32_t)sx & 0xffffu) * 0x10001u;
        s16x2 NM[NP + 1];
        NM[NP] = splat2(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // new_msg for both links: other = (mag == min0) ? min1 : min0  ==  min1 + (mag != min0) * (min0 - min1)
            // other = (mag == min0) ? min1c : min0c -- as min1c + (mag != min0) * (min0c - min1c), or (the same value: a magnitude is the row minimum, where
            // min(mag, min1c) = min0c, or at least the second one, where it is min1c) as min0c + min1c - min(mag, min1c)
            s16x2 other;
            if constexpr (ldpc_other_sum<MAXDEG, IRREG>()) {
                other = SUMCB - pmin2(G[p], MIN1CB);
            } else {
                const s16x2 ne = pmin2(G[p] - MIN0B, splat2(1));
                other = ne * NDB + MIN1CB;
            }
            const s16x2 neg = from_bits2(SXB ^ bits2(V[p])) >> 15;                   // 0 or -1
            s16x2 nm = pmin2(from_bits2(bits2(other) ^ bits2(neg)) - neg, q8(31));
            // new posterior: 16-bit saturating add = int8 saturation; >> 8 brings the bytes to bits 7:0 / 23:16 for the stores
            const s16x2 pn = from_bits2(bits2(sat_add2(V[p], nm)) >> 8);   // (bits 15:8 then hold junk; the byte stores do not look at them)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * p + h;
                bool present_u = (k < MAXDEG) ? (!IRREG || k < deg) : (k <= MAXDEG + 1);   // uniform part
                if (!present_u) { nm[h] = 0; continue; }
                if (k == MAXDEG + 1 && !has_prev) nm[h] = 0;
                bool wr = (k == MAXDEG + 1) ? has_prev : true;
                if constexpr (CONF) {
                    if (k < MAXC && k < nc) wr = !((early >> k) & 1);
                }
                if (wr) {
                    const uint32_t a = (k < MAXDEG) ? addr[k] : (k == MAXDEG ? own_l : prev_l);
                    if (h == 0) lds_write_lo_i8(a, bits2(pn)); else lds_write_hi_i8(a, bits2(pn));
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
    PROF_T(t_d);
    PROF_ADD(CONF ? 6 : 2, t_c, t_d);
    PROF_MARK(KIND, 6);
}

// TWO FRAMES PER WORKGROUP, in lockstep: 768 threads = 12 waves; threads [0,384) decode frame slot 0, [384,768)
// slot 1 (lane j < 360 of a slot owns row j of every layer).  Both slots walk the same layer sequence, so the
// workgroup barriers are shared (half the barrier cost per frame), the serial chain / level phases of the two
// frames overlap, and one 12-wave workgroup per CU always gets 3 waves on each SIMD (two independent 6-wave
// workgroups only co-reside when the dispatcher happens to start them on complementary SIMDs).
// FPB = frame slots per workgroup.  2 is the throughput mapping above.  1 (a 384-thread workgroup = one frame) serves SMALL batches:
// with fewer frames than CUs every frame gets a compute unit of its own instead of sharing one with a second frame -- twice the CUs
// busy and no issue contention between the slots, which is what a few-transponder call (or one stream's frames) needs.
constexpr int LDPC_TPS = 384;        // threads per slot

template <int MAXDEG, int REC, bool IRREG, int LDPC_FPB>
__global__ __launch_bounds__(LDPC_FPB * LDPC_TPS) __attribute__((amdgpu_waves_per_eu(MAXDEG <= LDPC_WPE4_MAXDEG ? 4 : 3))) void ldpc_decode_kernel(const LdpcLayerDesc* __restrict__ layers, const uint32_t* __restrict__ ents,
                                                                               const uint32_t* __restrict__ rows, const uint32_t* __restrict__ atab, LdpcKernelArgs A) {
    extern __shared__ __attribute__((aligned(16))) int8_t post_all[];
    __shared__ int s_flag[LDPC_FPB][8];
    __shared__ int s_done[LDPC_FPB];
    __shared__ uint32_t s_cw[LDPC_FPB][LDPC_CW_DWORDS(MAXDEG) * 360];  // hand-off records of the chain walk (8 bytes per row) and of the quad walk (8 / 12)
    __shared__ uint8_t s_cres[LDPC_FPB][384];
    const int fs = threadIdx.x / LDPC_TPS;
    const int j = threadIdx.x - fs * LDPC_TPS;
    const int N = A.N, K = A.K, R = A.R, q = A.q;
    const int npad = (N + 15) & ~15;
    int8_t* __restrict__ post = post_all + (size_t)fs * npad;
    uint32_t* __restrict__ msg = A.msg_ws + ((size_t)blockIdx.x * LDPC_FPB + fs) * (size_t)R * REC;
    uint32_t* __restrict__ sgn = A.sgn_ws + ((size_t)blockIdx.x * LDPC_FPB + fs) * SGN_WS_DWORDS;
    uint32_t* __restrict__ cw = s_cw[fs];
    uint8_t* __restrict__ cres = s_cres[fs];

    __shared__ int s_next;
    int f0 = blockIdx.x * LDPC_FPB;
    while (f0 < A.nframes) {
        const int f = f0 + fs;
        const bool valid = f < A.nframes;
        const bool lane_ok = (j < 360) && valid;
        if (valid) {
            const int8_t* __restrict__ src = A.llr + (size_t)f * N;
            // information-bit LLRs: straight copy (K is a multiple of 8)
            for (int i = j; i < K / 8; i += LDPC_TPS) reinterpret_cast<uint2*>(post)[i] = reinterpret_cast<const uint2*>(src)[i];
            // parity LLRs: pty[360*i + jj] = llr[K + q*jj + i]   (layered_decoder.hh:124-126)
            for (int c = j; c < R; c += LDPC_TPS) {
                int jj = c / q, i = c - jj * q;
                post[K + 360 * i + jj] = src[K + c];
            }
            if constexpr (ldpc_diet<MAXDEG, IRREG>()) {
                // the first sweep reads all-zero messages: this lane's records are cleared here, so that a sweep fetches them without asking which sweep it is
                if (j < 360 && !(LDPC_EXP & 1)) {
                    uint32_t z[REC];
#pragma unroll
                    for (int w = 0; w < REC; ++w) z[w] = 0;
                    for (int l = 0; l < q; ++l) rec_store<REC>(z, msg + ((size_t)l * 360 + j) * REC);
                }
            }
        }
        lds_barrier();

        int it = 0, ret = 0, trip = 0;
        PROF_LAYER_DECL;
        PROF_MARK_DECL;
        bool done = !valid;
        while (true) {
            const bool check = !done && (!A.force || it == A.max_trials);
            // (uniform over the workgroup: both slots run the same trips; in forced mode only the trip after the last sweep checks)
            const bool any_check = !A.force || trip == A.max_trials;
            uint32_t zflag = 0;
            if (check && valid) zflag = sign_pack(post, N, reinterpret_cast<uint8_t*>(sgn), j, LDPC_TPS);
            if (any_check) __syncthreads();              // the sign bytes went to global memory: full barrier (drains vmcnt), once per iteration
            if (check) {
                bool bad = valid ? (zflag != 0 || syndromes_bad<MAXDEG>(A.q, A.synd_base, ents, sgn, j, LDPC_TPS)) : false;
                unsigned long long b = __ballot(bad);
                if ((j & 63) == 0) s_flag[fs][j >> 6] = (b != 0);
            }
            lds_barrier();
            if (check) {
                int any = s_flag[fs][0] | s_flag[fs][1] | s_flag[fs][2] | s_flag[fs][3] | s_flag[fs][4] | s_flag[fs][5];
                if (A.force) { ret = any ? -1 : A.max_trials; done = true; }
                else if (!any) { ret = it; done = true; }
                else if (it == A.max_trials) { ret = -1; done = true; }
            }
            if (j == 0) s_done[fs] = done;
            lds_barrier();
            if (s_done[0] && s_done[LDPC_FPB - 1]) break;
            PROF_LAYER(-1);
            // ---- one layered sweep (LDPCDecoder::update), descriptors / records / row words prefetched one layer ahead
            const bool active = lane_ok && !done;
            const bool live = __builtin_amdgcn_readfirstlane((int)(valid && !done)) != 0;   // uniform over the slot's waves
            const bool first = (it == 0);
            uint32_t rec_next[REC];
#pragma unroll
            for (int w = 0; w < REC; ++w) rec_next[w] = 0;
            constexpr bool DIET = ldpc_diet<MAXDEG, IRREG>();
            const int jc = min(j, 359);                                  // (the row index idle lanes fetch with)
            if constexpr (DIET) { if (!(LDPC_EXP & 1)) rec_load<REC>(rec_next, msg + (size_t)jc * REC); }
            else if (!first && active && !(LDPC_EXP & 1)) rec_load<REC>(rec_next, msg + (size_t)j * REC);
            // The layer descriptors travel TWO layers ahead: the one of layer + 1 decides, at the top of a layer, whether that layer's row words
            // are prefetched -- loaded only one layer ahead it was awaited right there, i.e. every layer began with a scalar-cache round trip.
            LdpcLayerDesc Lnext = layers[0], Lnext2 = layers[q > 1 ? 1 : 0];
            // the layer's pair table (link addresses) travels one layer ahead in scalar registers, like the descriptor: its scalar-cache
            // latency is then off the path between a layer barrier and the first LDS read
            constexpr int NPW = 2 * ((MAXDEG + 1) / 2);
            constexpr bool PW_AHEAD = MAXDEG <= 12;   // (wider tables do not fit the scalar registers twice: measured 14 % slower for degree 28)
            constexpr bool ATAB = ldpc_use_atab<MAXDEG, IRREG>();
            constexpr int NPI = (MAXDEG + 1) / 2;     // pairs of table links = words per row of the address table
            uint32_t pw_next[NPW];
            if constexpr (ATAB) {
#pragma unroll
                for (int i = 0; i < NPW; ++i) pw_next[i] = 0;
                if (DIET || active) atab_load<NPI>(pw_next, atab + (size_t)j * ldpc_atab_stride(NPI));      // layer 0
            } else if constexpr (PW_AHEAD) {
#pragma unroll
                for (int i = 0; i < NPW; ++i) pw_next[i] = ents[A.pent_base + i];
            }
            uint32_t rw_next = 1;
            if constexpr (DIET) rw_next = rows[Lnext.row_off + jc];      // (conflict-free layers have row_off 0: a word nobody looks at)
            else if ((Lnext.depth_nc & 0xffffu) > 1 && active) rw_next = rows[Lnext.row_off + j];
            uint32_t own_a = lds_offset(post) + (uint32_t)(K + j);       // LDS address of the row's own parity bit: + 360 per layer
            for (int layer = 0; layer < q; ++layer) {
                PROF_T(t_g);
                uint32_t rec[REC];
#pragma unroll
                for (int w = 0; w < REC; ++w) rec[w] = rec_next[w];
                const LdpcLayerDesc L = Lnext;
                uint32_t pw[NPW];
#pragma unroll
                for (int i = 0; i < NPW; ++i) pw[i] = (ATAB || PW_AHEAD) ? pw_next[i] : ents[A.pent_base + layer * NPW + i];
                const uint32_t rw = rw_next;
                uint32_t* rp = msg + ((size_t)layer * 360 + j) * REC;
                if constexpr (DIET) {
                    const int ln = layer + 1 < q ? layer + 1 : q - 1;       // (behind the last layer: the last layer's once more -- nobody reads them)
                    Lnext = Lnext2;
                    Lnext2 = layers[layer + 2 < q ? layer + 2 : q - 1];
                    if constexpr (ATAB) {
                        atab_load<NPI>(pw_next, atab + ((size_t)ln * LDPC_TPS + j) * ldpc_atab_stride(NPI));
                    } else if constexpr (PW_AHEAD) {
#pragma unroll
                        for (int i = 0; i < NPW; ++i) pw_next[i] = ents[A.pent_base + ln * NPW + i];
                    }
                    if (!(LDPC_EXP & 1)) rec_load<REC>(rec_next, msg + ((size_t)ln * 360 + jc) * REC);
                    rw_next = rows[Lnext.row_off + jc];
                } else if (layer + 1 < q) {
                    Lnext = Lnext2;
                    Lnext2 = layers[layer + 2 < q ? layer + 2 : q - 1];
                    if constexpr (ATAB) {
                        if (active) atab_load<NPI>(pw_next, atab + ((size_t)(layer + 1) * LDPC_TPS + j) * ldpc_atab_stride(NPI));
                    } else if constexpr (PW_AHEAD) {
#pragma unroll
                        for (int i = 0; i < NPW; ++i) pw_next[i] = ents[A.pent_base + (layer + 1) * NPW + i];
                    }
                    if (!first && active && !(LDPC_EXP & 1)) rec_load<REC>(rec_next, rp + 360 * REC);
                    if ((Lnext.depth_nc & 0xffffu) > 1 && active) rw_next = rows[Lnext.row_off + j];
                }
                PROF_T(t_h);
                PROF_ADD(7, t_g, t_h);
                uint32_t ro[REC];
#pragma unroll
                for (int w = 0; w < REC; ++w) ro[w] = 0;
                if ((L.depth_nc & 0xffffu) == 1) layer_update<MAXDEG, REC, 0, IRREG>(post, A, ents + L.ent_off, pw, L, 1u, layer, j, active, live, rec, ro, cw, cres, own_a, nullptr, PROF_MARK_PTR);
                else if ((L.deg >> 16) == LDPC_WALK_MARK) layer_update<MAXDEG, REC, 6, IRREG>(post, A, ents + L.ent_off, pw, L, rw, layer, j, active, live, rec, ro, cw, cres, own_a, rows + L.row_off + 360);
                else if ((L.deg >> 16) > 0) layer_update<MAXDEG, REC, 1, IRREG>(post, A, ents + L.ent_off, pw, L, rw, layer, j, active, live, rec, ro, cw, cres, own_a, nullptr, PROF_MARK_PTR);
                else if ((L.depth_nc >> 16) <= 4u) layer_update<MAXDEG, REC, 3, IRREG>(post, A, ents + L.ent_off, pw, L, rw, layer, j, active, live, rec, ro, cw, cres, own_a);
                else if (MAXDEG > 12 && (L.depth_nc >> 16) <= 8u) layer_update<MAXDEG, REC, (MAXDEG > 12 ? 4 : 2), IRREG>(post, A, ents + L.ent_off, pw, L, rw, layer, j, active, live, rec, ro, cw, cres, own_a);
                else layer_update<MAXDEG, REC, 2, IRREG>(post, A, ents + L.ent_off, pw, L, rw, layer, j, active, live, rec, ro, cw, cres, own_a);
                own_a += 360u;
                // The prefetched record / row word of the next layer are claimed HERE, in uniform control flow and before this layer's
                // record store is issued: the compiler's wait for those loads then sits where nothing recent is in flight.  Left to
                // itself it put an s_waitcnt vmcnt(0) behind the store (the loop-carried copy of the prefetch registers, merged over
                // paths with and without a store), i.e. every wave sat out the store's acknowledge before every layer barrier.
#pragma unroll
                for (int w = 0; w < REC; ++w) asm volatile("" : "+v"(rec_next[w]));
                asm volatile("" : "+v"(rw_next));
                // (the same for the scalar prefetches: claimed at the end of the layer, their wait does not open the next one)
                asm volatile("" : "+s"(Lnext2.ent_off), "+s"(Lnext2.deg), "+s"(Lnext2.depth_nc), "+s"(Lnext2.row_off));
                if constexpr (ATAB) {
#pragma unroll
                    for (int i = 0; i < NPI; ++i) asm volatile("" : "+v"(pw_next[2 * i]));
                } else if constexpr (PW_AHEAD) {
#pragma unroll
                    for (int i = 0; i < NPW; ++i) asm volatile("" : "+s"(pw_next[i]));
                }
                if (active && !(LDPC_EXP & 1)) rec_store<REC>(ro, rp);
                if (LDPC_EXP & 1) asm volatile("" :: "v"(ro[0]), "v"(ro[REC - 1]));
                PROF_T(t_e);
                lds_barrier();
                PROF_T(t_f);
                PROF_ADD(3, t_e, t_f);
                PROF_LAYER(layer);
                PROF_MARK(((L.depth_nc & 0xffffu) == 1) ? 0 : (((L.deg >> 16) > 0 && (L.deg >> 16) != LDPC_WALK_MARK) ? 1 : 2), 7);
            }
            if (!done) ++it;
            ++trip;
        }

        // ---- outputs
        if (valid) {
            if (j == 0) A.trials[f] = ret;
            // hard decisions of [0,K): 64 bits per wave step via ballot, MSB-first bytes (module_dvbs2_demod.cpp:357-360)
            uint8_t* __restrict__ hd = A.hard + (size_t)f * A.hard_stride;
            const int lane = j & 63, wave = j >> 6;
            for (int base = wave * 64; base < K; base += 6 * 64) {
                int idx = base + lane;
                int neg = (idx < K) ? (post[idx] < 0) : 0;
                unsigned long long b = __ballot(neg);
                b = __builtin_bswap64(__brevll(b));
                if (lane == 0) {
                    int nbytes = min(8, (K - base) / 8);
                    if (nbytes == 8 && ((uintptr_t)(hd + base / 8) & 7u) == 0) *reinterpret_cast<uint2*>(hd + base / 8) = make_uint2((uint32_t)b, (uint32_t)(b >> 32));   // (a caller's stride need not be a multiple of 8)
                    else for (int n = 0; n < nbytes; ++n) hd[base / 8 + n] = (uint8_t)(b >> (8 * n));
                }
            }
            if (A.post) {
                int8_t* __restrict__ dst = A.post + (size_t)f * N;
                for (int i = j; i < K / 8; i += LDPC_TPS) reinterpret_cast<uint2*>(dst)[i] = reinterpret_cast<const uint2*>(post)[i];
                for (int c = j; c < R; c += LDPC_TPS) {
                    int jj = c / q, i = c - jj * q;
                    dst[K + c] = post[K + 360 * i + jj];
                }
            }
        }
        if (A.work_ctr) {
            if (threadIdx.x == 0) s_next = (int)(gridDim.x * LDPC_FPB + atomicAdd(A.work_ctr, (unsigned int)LDPC_FPB));
            lds_barrier();
            f0 = s_next;
        } else {
            f0 += gridDim.x * LDPC_FPB;
        }
        lds_barrier();
    }
}

template <int MAXDEG, int REC, bool IRREG>
static hipError_t launch_ldpc(const LdpcDeviceCode& C, const LdpcKernelArgs& A, int grid, int fpb, hipStream_t stream) {
    size_t lds = (size_t)((A.N + 15) / 16) * 16 * fpb;
    if (fpb == 1) {
        auto kern = ldpc_decode_kernel<MAXDEG, REC, IRREG, 1>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_TPS), lds, stream, C.d_layers, C.d_ents, C.d_rows, C.d_atab, A);
    } else {
        auto kern = ldpc_decode_kernel<MAXDEG, REC, IRREG, 2>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * LDPC_TPS), lds, stream, C.d_layers, C.d_ents, C.d_rows, C.d_atab, A);
    }
    return hipGetLastError();
}

template <int MAXDEG, int REC, bool IRREG>
static int occupancy_ldpc(int N) {
    int nb = 0;
    size_t lds = (size_t)((N + 15) / 16) * 16 * 2;
    auto kern = ldpc_decode_kernel<MAXDEG, REC, IRREG, 2>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 2 * LDPC_TPS, lds) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}

// (max_deg, irregular) pairs that occur in DVB-S2: regular B1..B11, C2, C3, C5, C6, C10; irregular C1 C4 C7 C8 C9
#define LDPC_DISPATCH(FN, ...)                                                             \
    if (!irregular) {                                                                      \
        switch (max_deg) {                                                                 \
            case 2: return FN<2, 1, false>(__VA_ARGS__);                                   \
            case 3: return FN<3, 2, false>(__VA_ARGS__);                                   \
            case 4: return FN<4, 2, false>(__VA_ARGS__);                                   \
            case 5: return FN<5, 2, false>(__VA_ARGS__);                                   \
            case 8: return FN<8, 4, false>(__VA_ARGS__);                                   \
            case 9: return FN<9, 4, false>(__VA_ARGS__);                                   \
            case 12: return FN<12, 4, false>(__VA_ARGS__);                                 \
            case 16: return FN<16, 8, false>(__VA_ARGS__);                                 \
            case 20: return FN<20, 8, false>(__VA_ARGS__);                                 \
            case 25: return FN<25, 8, false>(__VA_ARGS__);                                 \
            case 28: return FN<28, 8, false>(__VA_ARGS__);                                 \
            default: break;                                                                \
        }                                                                                  \
    } else {                                                                               \
        switch (max_deg) {                                                                 \
            case 2: return FN<2, 1, true>(__VA_ARGS__);                                    \
            case 5: return FN<5, 2, true>(__VA_ARGS__);                                    \
            case 11: return FN<11, 4, true>(__VA_ARGS__);                                  \
            case 17: return FN<17, 8, true>(__VA_ARGS__);                                  \
            default: break;                                                                \
        }                                                                                  \
    }

// frame slots per workgroup for a batch of `nframes` on a device with `num_cus` compute units (see FPB above)
int ldpc_frames_per_block(int nframes, int num_cus) { return nframes <= num_cus ? 1 : 2; }
size_t ldpc_sign_ws_bytes_per_slot() { return (size_t)SGN_WS_DWORDS * sizeof(uint32_t); }

int ldpc_blocks_per_cu(int max_deg, int irregular, int N) {
    LDPC_DISPATCH(occupancy_ldpc, N)
    return 1;
}

unsigned long long* g_ldpc_prof = nullptr;   // set by tools/ldpc_prof.py through dvbs2gpu_debug_set_prof (PROF builds)

hipError_t ldpc_decode_launch(const LdpcDeviceCode& C, const int8_t* llr, int nframes, int max_trials, int force,
                              uint8_t* hard, int hard_stride, int8_t* post, int32_t* trials, uint32_t* msg_ws, int grid, int fpb,
                              hipStream_t stream, unsigned int* work_ctr, uint32_t* sgn_ws) {
    LdpcKernelArgs A;
    A.work_ctr = work_ctr;
    A.sgn_ws = sgn_ws;
    if (work_ctr) {
        hipError_t e = hipMemsetAsync(work_ctr, 0, sizeof(unsigned int), stream);
        if (e != hipSuccess) return e;
    }
    A.llr = llr; A.hard = hard; A.post = post; A.trials = trials; A.msg_ws = msg_ws;
    A.nframes = nframes; A.N = C.N; A.K = C.K; A.R = C.R; A.q = C.q; A.pent_base = C.pent_base; A.synd_base = C.synd_base;
    A.max_trials = max_trials; A.force = force; A.hard_stride = hard_stride; A.dbg = 0;
    A.prof = g_ldpc_prof;
    const int max_deg = C.max_deg, irregular = C.irregular;
    LDPC_DISPATCH(launch_ldpc, C, A, grid, fpb, stream)
    return hipErrorInvalidValue;
}

}  // namespace s2
