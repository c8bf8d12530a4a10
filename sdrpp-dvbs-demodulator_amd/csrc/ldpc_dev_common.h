// Device helpers shared by the LDPC decoder kernels (ldpc_kernel.hip: lane = row, workgroup = two frames; ldpc_wave_kernel.hip: wave =
// frame): the int8 rules of the reference's SIMD lanes and the bit-vector form of its syndrome check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s2 {

__device__ __forceinline__ int med3i(int a, int lo, int hi) { return min(max(a, lo), hi); }  // folds to v_med3_i32 for lo <= hi
__device__ __forceinline__ int clamp8(int v) { return med3i(v, -128, 127); }

// |max(x,-127)| - 1 clamped at 0  == vqsub(vunsigned(vqabs(x)), 1)   (algorithms.hh:235-238)
__device__ __forceinline__ int mag_of(int in) {
    int a = in < 0 ? -in : in;
    return med3i(a - 1, 0, 126);
}


typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// ---- packed int16 helpers: two links per VALU instruction (v_pk_*_i16), the int8 saturation rules emulated in 16 bits
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 splat2(int v) { return s16x2{(short)v, (short)v}; }
__device__ __forceinline__ s16x2 pmin2(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pmax2(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pclamp2(s16x2 v, int lo, int hi) { return pmin2(pmax2(v, splat2(lo)), splat2(hi)); }
__device__ __forceinline__ uint32_t bits2(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 from_bits2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
// The packed row arithmetic works on values scaled by 256 ("Q8": the int8 quantity sits in the HIGH byte of each 16-bit half):
// the 16-bit saturating add/subtract of the hardware (v_pk_add_i16 / v_pk_sub_i16 with clamp) then IS the int8 saturation of the
// reference's SIMD lanes, and message bytes drop into place with one v_perm, no sign extension.  Only the high byte of a half
// is ever consumed (a positive saturation leaves 0xff below it).  Magnitude sentinel ("no link"): 127 (real ones are <= 126).
constexpr int Q8_NONE = 127 << 8;
__device__ __forceinline__ s16x2 q8(int v) { return s16x2{(short)(v << 8), (short)(v << 8)}; }
__device__ __forceinline__ s16x2 sat_sub2(s16x2 a, s16x2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ s16x2 sat_add2(s16x2 a, s16x2 b) { return __builtin_elementwise_add_sat(a, b); }
// message bytes k, k+1 (k even) of a record into the high bytes of the two halves: one v_perm
__device__ __forceinline__ s16x2 rec_pair_dw(uint32_t w, int k) {
    return from_bits2((k & 2) ? __builtin_amdgcn_perm(0u, w, 0x030c020cu) : __builtin_amdgcn_perm(0u, w, 0x010c000cu));
}
template <int REC>
__device__ __forceinline__ s16x2 rec_pair(const uint32_t (&rec)[REC], int k) { return rec_pair_dw(rec[k >> 2], k); }


// LDPCDecoder::bad (layered_decoder.hh:28-45): true if any row is unsatisfied.  A row is bad when the sign product of its links'
// posteriors is negative or when one of them is 0.  Every posterior belongs to at least one row (each parity bit to its own row), so the
// second condition over all rows is "some posterior of the frame is 0": a dword scan shared by the slot's 384 threads.  The first one is
// evaluated on BIT VECTORS: the code is quasi-cyclic, so the 360 sign products of layer i are
//     Y_i = P_i ^ P'_i ^ XOR_k rot(S_{r_k}, sp_k)            (S_r: signs of information group r, P_i: of parity group i)
// -- XORs of cyclic shifts of 360-bit groups: ~15 shifted 64-bit fetches per (layer, 64-row word) instead of 15 byte gathers per row
// (227 k LDS byte reads per check before).  Step 1 (sign_pack): every thread turns 8 posteriors into one sign byte (and scans them for
// zeros); a group is stored as 360 bits followed by a copy of its first 64, so a shifted fetch never wraps; the 10 KB per frame go to a
// per-slot global scratch (L2-resident; LDS has no room for them beside two frames and the co-resident front-end kernels).  Step 2
// (syndromes_bad), after a workgroup barrier: lane t takes (layer, word) = (t / 6, t % 6).
constexpr int SGN_GROUP_DW = 14;                                  // 56 bytes: 360 bits + the first 64 again (424), padded to dwords
constexpr int SGN_WS_DWORDS = 180 * SGN_GROUP_DW + 16;            // per slot (N/360 <= 180 groups)

__device__ __forceinline__ uint32_t sign_pack(const int8_t* __restrict__ post, int N, uint8_t* __restrict__ sg, int j, int tps) {
    const uint2* __restrict__ p8 = reinterpret_cast<const uint2*>(post);
    uint32_t z = 0;
    for (int idx = j; idx < N / 8; idx += tps) {
        const uint2 v = p8[idx];
        z |= ((v.x - 0x01010101u) & ~v.x) | ((v.y - 0x01010101u) & ~v.y);   // bit 7 of a byte set <=> that byte is 0 (or a borrow from a zero byte below: still "a zero")
        // sign bits 7, 15, 23, 31 of a dword -> one nibble, LSB = lowest byte (the multiply adds four shifted copies; no two terms collide)
        const uint32_t lo = ((((v.x >> 7) & 0x01010101u) * 0x01020408u) >> 24);
        const uint32_t hi = ((((v.y >> 7) & 0x01010101u) * 0x01020408u) >> 24);
        const uint32_t sb = lo | (hi << 4);
        const int g = (int)((uint32_t)idx / 45u);
        const int o = idx - 45 * g;
        uint8_t* __restrict__ d = sg + (SGN_GROUP_DW * 4) * g + o;
        d[0] = (uint8_t)sb;
        if (o < 8) d[45] = (uint8_t)sb;                                       // cyclic extension: bits 360..423 = bits 0..63
    }
    return z & 0x80808080u;
}
// the syndromes of 64 rows of one layer per lane: XOR of the 64-bit windows the plan's syndrome table names (ldpc_plan.h), one table
// word and three sign dwords per link, all independent loads (two dependent global round trips per check in total)
template <int MAXDEG>
__device__ __forceinline__ bool syndromes_bad(int q, int synd_base, const uint32_t* __restrict__ ents_all, const uint32_t* __restrict__ S, int j, int tps) {
    const int ntask = q * 6;
    const uint32_t* __restrict__ tab = ents_all + synd_base;
    bool bad = false;
    for (int t = j; t < ntask; t += tps) {
        uint32_t e[MAXDEG + 2];
#pragma unroll
        for (int k = 0; k < MAXDEG + 2; ++k) e[k] = tab[k * ntask + t];
        // all sign windows are fetched before any is used: two dependent global round trips per check, not one per link
        uint32_t d0[MAXDEG + 2], d1[MAXDEG + 2], d2[MAXDEG + 2];
#pragma unroll
        for (int k = 0; k < MAXDEG + 2; ++k) {
            const uint32_t* __restrict__ p = S + (e[k] & 0xffffu);         // (absent links: entry 0 -> a harmless read of the first dwords)
            d0[k] = p[0]; d1[k] = p[1]; d2[k] = p[2];
        }
#pragma unroll
        for (int k = 0; k < MAXDEG + 2; ++k) asm volatile("" : "+v"(d0[k]), "+v"(d1[k]), "+v"(d2[k]));   // (keeps the loads ahead of the arithmetic)
        unsigned long long acc = 0;
#pragma unroll
        for (int k = 0; k < MAXDEG + 2; ++k) {
            const uint32_t sh = (e[k] >> 16) & 31u;
            uint32_t lo = __builtin_amdgcn_alignbit(d1[k], d0[k], sh), hi = __builtin_amdgcn_alignbit(d2[k], d1[k], sh);
            lo &= ~((e[k] >> 30) & 1u);                                      // row 0 of layer 0 has no previous parity bit
            const uint32_t m = (uint32_t)((int)e[k] >> 31);                  // present?
            acc ^= ((unsigned long long)(hi & m) << 32) | (lo & m);
        }
        if (t - 6 * (t / 6) == 5) acc &= (1ull << 40) - 1;                   // rows 320..359
        bad |= acc != 0;
    }
    return bad;
}


}  // namespace s2
