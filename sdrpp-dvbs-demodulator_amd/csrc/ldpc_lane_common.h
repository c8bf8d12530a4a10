// Device helpers shared by the lane-per-row LDPC decoders (ldpc_kernel.hip: one lane = one row, a workgroup = two frames;
// ldpc_split_kernel.hip: two lanes = one row, a workgroup = one frame): kernel arguments, LDS byte traffic issued by hand, the
// message records and the chain walk's hand-off records.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>
#include <type_traits>
#include "ldpc_plan.h"
#include "kernels.h"
#include "ldpc_dev_common.h"

namespace s2 {

// Code tables are separate `const T* __restrict__` kernel parameters (not struct members) so that hipcc can
// prove them invariant and fetch the wave-uniform layer/link descriptors with scalar loads (s_load_*).
struct LdpcKernelArgs {
    const int8_t* llr;     // [nframes][N]
    uint8_t* hard;         // [nframes][hard_stride] packed hard decisions of bits [0,K), MSB first
    int8_t* post;          // optional [nframes][N] posteriors (reference layout), may be null
    int32_t* trials;       // [nframes]
    uint32_t* msg_ws;      // [gridDim.x][R][REC]
    int nframes, N, K, R, q;
    int pent_base;         // offset of the pair-format link table inside ents[] (ldpc_plan.h)
    int synd_base;         // offset of the syndrome-check table inside ents[] (ldpc_plan.h)
    int max_trials, force;
    int hard_stride;
    int dbg;                    // tests only (context option ldpc_split_fail_attempts): 1 = every attempt of the half-row decoder's layers with shared bits is made to fail (the fall-back path)
    uint32_t* sgn_ws;           // [gridDim.x * slots][SGN_WS_DWORDS]: bit-packed posterior signs for the syndrome check
    unsigned int* work_ctr;     // optional: frames beyond the first gridDim.x*2 are claimed dynamically (workgroups slowed by
                                // co-resident kernels of the pipelined mode then simply take fewer frames)
    unsigned long long* prof;   // development aid (-DLDPC_PROF builds only): per-wave phase cycle sums of workgroup 0
};

// What a decoder kernel is told arrives in ONE structure, read from the kernel-argument segment through a pointer the compiler cannot see through, afresh in every section
// of the kernel (frame load | syndrome check + sweep | output): kept in scalar registers for the whole kernel, the two dozen values only the frame load and the output need
// are what the allocator spills -- into vector-register lanes, read back inside the layers (profiles/r05_ldpc_split_layers.txt).
struct LdpcKernelParams {
    LdpcKernelArgs A;
    const void* layers;                   // LdpcLayerDesc[] (ldpc_kernel.hip) / LdpcSplitLayer[] (ldpc_split_kernel.hip)
    const uint32_t* ents;
    const uint32_t* rows;
    const uint32_t* atab;
    int npl;                              // pseudo-layers (ldpc_split_kernel.hip)
    int tab_words;                        // words behind atab: the pseudo-layers' tables + the side entries of the kind-8 layers (ldpc_split_plan.h)
};
// ldpc_params() reads the structure at offset 0 of the kernel-argument segment: a kernel that uses it takes ONE by-value LdpcKernelParams as its first and only parameter
// (ldpc_split_kernel), and the launch arguments start with the LdpcKernelArgs member
static_assert(offsetof(LdpcKernelParams, A) == 0, "ldpc_params(): the launch arguments sit at the start of the kernel-argument segment");
static_assert(std::is_trivially_copyable<LdpcKernelParams>::value, "passed by value in the kernel-argument segment");
typedef const __attribute__((address_space(4))) LdpcKernelParams* LdpcKernelParamsPtr;
__device__ __forceinline__ LdpcKernelParamsPtr ldpc_params() {
    uint32_t off = 0;
    asm volatile("" : "+s"(off));        // (zero -- but not to the compiler: what is read through the pointer can be neither hoisted nor merged with another section's reads)
    return (LdpcKernelParamsPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}
__device__ __forceinline__ LdpcKernelArgs ldpc_args(LdpcKernelParamsPtr P) {
    LdpcKernelArgs A;
    A.llr = P->A.llr; A.hard = P->A.hard; A.post = P->A.post; A.trials = P->A.trials; A.msg_ws = P->A.msg_ws;
    A.nframes = P->A.nframes; A.N = P->A.N; A.K = P->A.K; A.R = P->A.R; A.q = P->A.q; A.pent_base = P->A.pent_base; A.synd_base = P->A.synd_base;
    A.max_trials = P->A.max_trials; A.force = P->A.force; A.hard_stride = P->A.hard_stride; A.dbg = P->A.dbg;
    A.sgn_ws = P->A.sgn_ws; A.work_ctr = P->A.work_ctr; A.prof = P->A.prof;
    return A;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. wait
// for the in-flight message-record prefetch and store of every layer (cdna guide, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int REC>
__device__ __forceinline__ int rec_byte(const uint32_t (&rec)[REC], int k) {
    return (int)__builtin_amdgcn_sbfe((int)rec[k >> 2], (k & 3) * 8, 8);  // v_bfe_i32
}

template <int REC>
__device__ __forceinline__ void rec_load(uint32_t (&rec)[REC], const uint32_t* p) {
    if constexpr (REC == 1) {
        rec[0] = *p;
    } else if constexpr (REC == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        rec[0] = v.x; rec[1] = v.y;
    } else {
#pragma unroll
        for (int w = 0; w < REC; w += 4) {
            uint4 v = *reinterpret_cast<const uint4*>(p + w);
            rec[w] = v.x; rec[w + 1] = v.y; rec[w + 2] = v.z; rec[w + 3] = v.w;
        }
    }
}
template <int REC>
__device__ __forceinline__ void rec_store(const uint32_t (&rec)[REC], uint32_t* p) {
    if constexpr (REC == 1) {
        *p = rec[0];
    } else if constexpr (REC == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(rec[0], rec[1]);
    } else {
#pragma unroll
        for (int w = 0; w < REC; w += 4) *reinterpret_cast<uint4*>(p + w) = make_uint4(rec[w], rec[w + 1], rec[w + 2], rec[w + 3]);
    }
}

// new message for one link given the row totals (algorithms.hh:250-256 + clamp :275)
__device__ __forceinline__ int new_msg(int in, int mg, int min0, int min1, int sx) {
    int other = (mg == min0) ? min1 : min0;
    int neg = (sx ^ in) >> 31;             // 0 or -1
    int v = (other ^ neg) - neg;           // +-other
    return med3i(v, -32, 31);
}

// byte address of link `ent` for row j: 360*r + (j + sp) mod 360
__device__ __forceinline__ int link_addr(uint32_t ent, int j) {
    int t = j + (int)(ent & 0xffffu);
    t = (int)min((uint32_t)t, (uint32_t)(t - 360));
    return t + 360 * (int)(ent >> 16);
}

#define ROW_ACCUM(v, m)                     \
    do {                                    \
        min1 = min(min1, max(min0, (m)));   \
        min0 = min(min0, (m));              \
        sx ^= (v);                          \
    } while (0)

typedef __attribute__((address_space(3))) int8_t lds_i8;
__device__ __forceinline__ uint32_t lds_offset(const int8_t* p) { return (uint32_t)(uintptr_t)(const lds_i8*)p; }
#define LDS_I8(a) (*(lds_i8*)(uintptr_t)(a))
// Two posterior bytes, sign-extended into the low / high half of a register by the LDS unit (ds_read_i8_d16 / _d16_hi).
// MI355X runs with SRAM ECC, where a d16 load ZEROES the other half instead of preserving it, so the two halves land in
// two registers and one v_or joins them (still 1 VALU op per pair instead of 2 sign extensions + a byte permute).
// Issue only; lds_pairs_wait() below orders the results.
#ifndef LDPC_LEVEL_PREFETCH_KIND
#define LDPC_LEVEL_PREFETCH_KIND 2   // per-level layers from this kind on fetch all their shared posteriors up front (2 = also the general form: +2-3 % on the codes with more than 8 shared links per row; 3 = only the <= 4 / <= 8-link forms; A/B switch)
#endif
#ifndef LDPC_EXP
#define LDPC_EXP 0   // development switches for TIMING experiments (results wrong): 1 no message records, 2 no posterior stores, 4 no posterior loads
#endif
__device__ __forceinline__ void lds_read_pair_i8(uint32_t a_lo, uint32_t a_hi, uint32_t& r_lo, uint32_t& r_hi) {
    if (LDPC_EXP & 4) { r_lo = a_lo & 0xffu; r_hi = (a_hi & 0xffu) << 16; return; }
    asm volatile("ds_read_u8_d16 %0, %2\n\tds_read_u8_d16_hi %1, %3" : "=&v"(r_lo), "=&v"(r_hi) : "v"(a_lo), "v"(a_hi) : "memory");
}
#define LDS_READY_CASE(n) case n: asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r_lo), "+v"(r_hi) : : "memory"); break
__device__ __forceinline__ void lds_pair_ready(int outstanding, uint32_t& r_lo, uint32_t& r_hi) {   // (constant after unrolling)
    switch (outstanding) {
        LDS_READY_CASE(0); LDS_READY_CASE(2); LDS_READY_CASE(4); LDS_READY_CASE(6); LDS_READY_CASE(8); LDS_READY_CASE(10); LDS_READY_CASE(12); LDS_READY_CASE(14);
        default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(r_lo), "+v"(r_hi) : : "memory"); break;
    }
}
#undef LDS_READY_CASE
__device__ __forceinline__ void lds_pairs_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Byte stores straight out of the halves of a packed register (ds_write_b8 takes bits 7:0, ds_write_b8_d16_hi bits 23:16): no
// VALU work to extract or narrow the value.  The compiler does not count these against lgkmcnt: lds_pairs_wait() before the
// next barrier.
__device__ __forceinline__ void lds_write_lo_i8(uint32_t addr, uint32_t packed) { if (LDPC_EXP & 2) { asm volatile("" :: "v"(addr), "v"(packed)); return; } asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(packed) : "memory"); }
__device__ __forceinline__ void lds_write_hi_i8(uint32_t addr, uint32_t packed) { if (LDPC_EXP & 2) { asm volatile("" :: "v"(addr), "v"(packed)); return; } asm volatile("ds_write_b8_d16_hi %0, %1" ::"v"(addr), "v"(packed) : "memory"); }

// One row of the chain walk (KIND 1 layers).  Per row the reference computes, from the posterior x the previous row's E link
// left:  vL = sat8(x - mL), mag = min(qE, max(|vL| - 1, 0)), nm = clamp(+-mag, -32, 31) with the sign of vL (flipped when the
// row's sign product is negative: s = -1), x' = sat8(in_E + nm).  As a function of x that is
//     x' = sat8(E' + s * (clamp(x, m + 1, m + CA) + clamp(x, m - CB, m - 1))),   E' = in_E - 2 s m,
// with (CA, CB) = (min(qE, 31) + 1, min(qE, 32) + 1) for s = +1 and swapped for s = -1: the dead zone |x - m| <= 1, the unit
// slope and the two saturation levels are the two clamps, checked exhaustively against the reference form over
// x, in_E in int8, m in [-32, 31], all qE, both signs.  Hand-off record of a row: two dwords, the four clamp limits as bytes and
// {s, E'} as halves (operands the compiler picks apart with SDWA selects): two clamps, an add and a multiply-add per row.
struct ChainRec { uint32_t lim, se; };   // lim = bytes {L1, H1, L2, H2} (each within [-66, 65]), se = s | E' << 16
__device__ __forceinline__ ChainRec chain_record(int m, int qE, int vE, int sneg /* 0 or -1 */) {
    const int sig = 1 | sneg;
    const int c31 = min(qE, 31), c32 = min(qE, 32);
    const int CA = (sneg ? c32 : c31) + 1, CB = (sneg ? c31 : c32) + 1;
    ChainRec r;
    r.lim = ((uint32_t)(m + 1) & 0xffu) | (((uint32_t)(m + CA) & 0xffu) << 8) | (((uint32_t)(m - CB) & 0xffu) << 16) | ((uint32_t)(m - 1) << 24);
    r.se = ((uint32_t)sig & 0xffffu) | ((uint32_t)(vE - 2 * sig * m) << 16);
    return r;
}
__device__ __forceinline__ int chain_step(int x, uint32_t lim, uint32_t se) {
    const int q1 = min(max(x, (int)(int8_t)lim), (int)(int8_t)(lim >> 8));
    const int q2 = min(max(x, (int)(int8_t)(lim >> 16)), (int)lim >> 24);
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"((int)(short)se), "v"(q1 + q2), "v"((int)se >> 16));   // (left alone the compiler picks a quarter-rate 64-bit multiply-add here)
    return clamp8(r);
}

}  // namespace s2
