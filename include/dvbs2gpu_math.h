/* dvbs2gpu_math.h -- the engine's OWN definition of the transcendental functions its float stages use.
 *
 * Why this exists: the reference calls libm (cosf/sinf through SDR++ math::phasor at freq_shift.cpp:6, dvbs2_pll.cpp:39,
 * dvbs2_plhdr_demod.cpp:35, fll.cpp:137; atan2f through complex_t::phase() at dvbs2_pll.cpp:50-75 and
 * constellation.cpp:259; expf/logf at constellation.cpp:226,250).  libm results are not specified bit for bit (glibc,
 * the device's OCML and the hardware v_sin_f32 all differ by an ULP here and there), and the receiver's loops are
 * sign-directed / LUT-quantised, so one ULP in a phasor becomes a different polyphase arm or LUT cell a few samples
 * later.  A bit-exact GPU engine therefore needs ONE definition that host and device evaluate identically.  This header
 * is that definition: straight-line IEEE-754 binary32/binary64 arithmetic (+, -, *, /, conversions, comparisons, and -- in sincosf_det
 * only, written out as fmaf -- the correctly rounded fused multiply-add; nothing is contracted implicitly: every translation unit that
 * includes it is compiled with -ffp-contract=off), so gfx950 and x86-64 produce the same bits.  The device kernels (csrc/s2_rx_kernels.hip), the host-side table builders of the library and
 * the CPU oracle (oracle/s2chain.cpp, oracle/dvbs_fe.cpp) all call these functions and nothing else for these values.
 *
 * Accuracy (checked in tests/test_det_math.py against the host libm in double): sincos <= 2 ULP for |x| <= 64,
 * atan2 <= 3 ULP, exp/log: correctly rounded double-rounding of a ~1e-15 accurate fp64 evaluation (<= 1 ULP binary32).
 * These are the build's own definitions of the SDR++/libm primitives: "parity unpinned" against
 * the reference's libm by necessity, bit-identical between the engine and its oracle by construction.
 */
#ifndef DVBS2GPU_MATH_H
#define DVBS2GPU_MATH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define DVBS2M_HD __host__ __device__ __forceinline__
#else
#define DVBS2M_HD static inline
#endif

namespace dvbs2m {

DVBS2M_HD float f_abs(float x) { return __builtin_fabsf(x); }
DVBS2M_HD float f_copysign(float mag, float sgn) { return __builtin_copysignf(mag, sgn); }
DVBS2M_HD double d_from_bits(uint64_t u) { return __builtin_bit_cast(double, u); }
DVBS2M_HD uint64_t d_to_bits(double d) { return __builtin_bit_cast(uint64_t, d); }
DVBS2M_HD float f_from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }

/* sin and cos of x, |x| up to a few thousand (the loops here keep |x| <= 2 pi; the band-edge taps go to ~70 rad).
 * Round 3 form: x = j pi/2 + r with j = rint(x 2/pi) and a two-term Cody-Waite reduction in fused multiply-adds (the products j * hi,
 * j * lo enter the sums unrounded), the classic single-precision minimax polynomials on |r| <= pi/4 in Horner form with fused
 * multiply-adds, quadrant fix-ups on the sign bits.  fmaf is ONE correctly rounded operation on both sides (v_fma_f32 on gfx950, vfmadd /
 * glibc's exact fmaf on x86-64), so host and device still agree bit for bit; what it buys is the length of the serial per-symbol chains
 * this sits in (PLL, PLHDR, Costas, FLL): ~27 device instructions instead of ~46 for the separately rounded three-term form of rounds
 * 1-2.  sincos(0) = (0, 1) exactly.  Accuracy unchanged (tests/test_det_math.py: |error| < 2.5 * 2^-24 for |x| <= 70). */
DVBS2M_HD void sincosf_det(float x, float* sn_out, float* cs_out) {
    const float j = __builtin_rintf(x * 0.636619772367581343f);
    float r = __builtin_fmaf(j, -1.57079637050628662109375f, x);            /* pi/2 = hi + lo, hi = the nearest binary32 */
    r = __builtin_fmaf(j, 4.37113900018624283e-8f, r);                      /* -lo */
    const float z = r * r;
    const float ps = __builtin_fmaf(r * z, __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), r);
    const float pc = __builtin_fmaf(z * z, __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                                    __builtin_fmaf(-0.5f, z, 1.0f));
    const int ji = (int)j;
    const bool swap = (ji & 1) != 0;
    const uint32_t s1 = (uint32_t)ji << 30;                                 /* bit 31 = bit 1 of j, bit 30 = bit 0 of j */
    const uint32_t sflip = s1 & 0x80000000u;                                /* sin changes sign in quadrants 2, 3 */
    const uint32_t cflip = (s1 + (s1 << 1)) & 0x80000000u;                  /* cos in quadrants 1, 2: bit 1 ^ bit 0 (no carry into bit 31) */
    *sn_out = f_from_bits(__builtin_bit_cast(uint32_t, swap ? pc : ps) ^ sflip);
    *cs_out = f_from_bits(__builtin_bit_cast(uint32_t, swap ? ps : pc) ^ cflip);
}

/* atan2(y, x) in (-pi, pi]: a = min/max of the magnitudes; beyond tan(pi/8) the identity
 * atan(a) = pi/4 + atan((a-1)/(a+1)) = pi/4 + atan((min-max)/(min+max)) keeps the polynomial's argument within
 * +-tan(pi/8) with ONE division either way; odd minimax polynomial (Cephes atanf coefficients), then the octant
 * fix-ups.  atan2(0, 0) = 0; the sign of the result is the sign bit of y. */
DVBS2M_HD float atan2f_det(float y, float x) {
    const float ax = f_abs(x), ay = f_abs(y);
    const bool swap = ay > ax;
    const float mx = swap ? ay : ax, mn = swap ? ax : ay;
    const bool hi = mn > 0.4142135679721832275390625f * mx;
    const float num = hi ? mn - mx : mn;
    const float den = hi ? mn + mx : mx;
    const float a = den > 0.f ? num / den : 0.f;
    const float z = a * a;
    float p = ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z) * a + a;
    if (hi) p = p + 0.785398163397448309616f;
    if (swap) p = 1.57079632679489661923f - p;
    if (x < 0.f) p = 3.14159265358979323846f - p;
    return f_copysign(p, y);
}

/* e^x for a binary32 argument, evaluated in binary64 and rounded once to binary32 (subnormal results included):
 * x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial (|error| < 5e-18 relative), exact scaling by 2^k.
 * Arguments below -104 give 0 (below half the smallest subnormal), above 89 give +inf. */
DVBS2M_HD float expf_det(float xf) {
    if (xf != xf) return xf;
    if (xf < -104.0f) return 0.0f;
    if (xf > 89.0f) return f_from_bits(0x7f800000u);
    const double x = (double)xf;
    const double t = x * 1.4426950408889634074;
    const int k = (int)(t < 0.0 ? t - 0.5 : t + 0.5);
    const double fk = (double)k;
    const double r = (x - fk * 6.93147180369123816490e-01) - fk * 1.90821492927058770002e-10;
    double p = 1.6059043836821613e-10;            /* 1/13! */
    p = p * r + 2.08767569878681e-09;             /* 1/12! */
    p = p * r + 2.505210838544172e-08;            /* 1/11! */
    p = p * r + 2.755731922398589e-07;            /* 1/10! */
    p = p * r + 2.7557319223985893e-06;           /* 1/9!  */
    p = p * r + 2.48015873015873e-05;             /* 1/8!  */
    p = p * r + 1.984126984126984e-04;            /* 1/7!  */
    p = p * r + 1.388888888888889e-03;            /* 1/6!  */
    p = p * r + 8.333333333333333e-03;            /* 1/5!  */
    p = p * r + 4.1666666666666664e-02;           /* 1/4!  */
    p = p * r + 1.6666666666666666e-01;           /* 1/3!  */
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const double scale = d_from_bits((uint64_t)(k + 1023) << 52);   /* k in [-151, 129]: a normal binary64 */
    return (float)(p * scale);
}

/* natural logarithm of a binary32 argument, evaluated in binary64 and rounded once to binary32:
 * v = 2^e f, f in (sqrt(1/2), sqrt(2)], log f = 2 atanh(s), s = (f-1)/(f+1), |s| <= 0.1716, series to s^19.
 * log(0) = -inf, log(negative) = NaN, log(inf) = inf. */
DVBS2M_HD float logf_det(float v) {
    if (v != v) return v;
    if (v == 0.0f) return f_from_bits(0xff800000u);
    if (v < 0.0f) return f_from_bits(0x7fc00000u);
    if (v == f_from_bits(0x7f800000u)) return v;
    const uint64_t b = d_to_bits((double)v);      /* every binary32 (subnormals too) is a normal binary64 */
    int e = (int)(b >> 52) - 1023;
    double f = d_from_bits((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (f > 1.4142135623730951) { f = f * 0.5; e += 1; }
    const double s = (f - 1.0) / (f + 1.0);
    const double z = s * s;
    double p = 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double r = (double)e * 0.693147180559945309417 + (2.0 * s) * p;
    return (float)r;
}

/* constellation_t::clamp (constellation.cpp:263-270): halve until within +-127, then truncate.  The reference converts a
 * non-finite float to int8 (undefined behaviour; x86's cvttss2si yields INT_MIN whose low byte is 0): defined as 0 here. */
DVBS2M_HD int8_t llr_clamp_det(float x) {
    if (!(x - x == 0.0f)) return 0;               /* NaN or +-inf */
    while (x < -127.0f || x > 127.0f) x = x * 0.5f;
    return (int8_t)(int)x;
}

}  /* namespace dvbs2m */
#endif
