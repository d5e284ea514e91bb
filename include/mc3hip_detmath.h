/*
 * mc3hip_detmath.h -- deterministic float64 elementary functions.
 *
 * The chain update decides accept/reject from exp(0.5*(chisq_cur-chisq_new)),
 * and the wavelet likelihood needs log(2*pi*sigma^2) (reference:
 * mc3/chain.py:257-260, src_c/_dwt.c:98-110).  libm on the host and ocml on the
 * device may round these differently in the last bit, which would break the
 * bit-exact accept-count contract between the HIP kernels and their CPU
 * checker.  These versions use only IEEE-754 +,-,*,/,sqrt,fma,rint and integer
 * bit moves, so one source gives bit-identical results when compiled by gcc
 * for x86-64 and by hipcc for gfx950, PROVIDED both are built with
 * -ffp-contract=off (every fused multiply-add below is spelled out).
 *
 * Accuracy (checked against libm in tests/test_detmath.py): <= 2 ulp.
 * This header is a numerical specification shared by the product kernels and
 * by oracle/; it contains no sampler logic.
 */
#ifndef MC3HIP_DETMATH_H
#define MC3HIP_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MC3_HD __host__ __device__ __forceinline__
#else
#define MC3_HD static inline
#endif

MC3_HD double mc3_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

MC3_HD uint64_t mc3_bits(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    return v.u;
}

MC3_HD double mc3_from_bits(uint64_t u)
{
    union { double d; uint64_t u; } v;
    v.u = u;
    return v.d;
}

/* 2^k for -1022 <= k <= 1023 */
MC3_HD double mc3_pow2i(int k) { return mc3_from_bits((uint64_t)(k + 1023) << 52); }

/* exp(x): Cody-Waite reduction x = k*ln2 + r, |r| <= ln2/2, degree-13 Taylor
 * polynomial (truncation < 5e-18 relative; Estrin order), result scaled by 2^k in two exact
 * halves so that overflow/underflow round once.
 * exp(NaN)=NaN, exp(x>709.78..)=+inf, exp(x<-745.2)=0. */
MC3_HD double mc3_det_exp(double x)
{
    if (x != x)
        return x;
    if (x > 709.782712893384)
        return mc3_from_bits(0x7ff0000000000000ull);
    if (x < -745.2)
        return 0.0;
    const double kd = __builtin_rint(x * 1.4426950408889634);
    double r = mc3_fma(-kd, 6.93147180369123816490e-01, x);
    r = mc3_fma(-kd, 1.90821492927058770002e-10, r);
    /* Estrin's scheme: four dependent fused steps instead of thirteen (a single
     * wavefront waits on this chain once per iteration) */
    const double r2 = r * r;
    const double a0 = mc3_fma(1.0, r, 1.0);                                   /* 1/1!, 1/0! */
    const double a1 = mc3_fma(0.16666666666666666, r, 0.5);                   /* 1/3!, 1/2! */
    const double a2 = mc3_fma(0.008333333333333333, r, 0.041666666666666664); /* 1/5!, 1/4! */
    const double a3 = mc3_fma(0.0001984126984126984, r, 0.001388888888888889);
    const double a4 = mc3_fma(2.7557319223985893e-06, r, 2.48015873015873e-05);
    const double a5 = mc3_fma(2.505210838544172e-08, r, 2.755731922398589e-07);
    const double a6 = mc3_fma(1.6059043836821613e-10, r, 2.08767569878681e-09); /* 1/13!, 1/12! */
    const double r4 = r2 * r2;
    const double b0 = mc3_fma(a1, r2, a0);
    const double b1 = mc3_fma(a3, r2, a2);
    const double b2 = mc3_fma(a5, r2, a4);
    const double r8 = r4 * r4;
    const double d0 = mc3_fma(b1, r4, b0);
    const double d1 = mc3_fma(a6, r4, b2);
    const double p = mc3_fma(d1, r8, d0);
    const int k = (int)kd;
    const int k1 = k / 2;
    const int k2 = k - k1;
    return (p * mc3_pow2i(k1)) * mc3_pow2i(k2);
}

/* 2^y: y = k + f with k = rint(y), |f| <= 1/2; exact for integer y
 * (the wavelet noise model uses 2^(-gamma*m), src_c/_dwt.c:96,103). */
MC3_HD double mc3_det_exp2(double y)
{
    if (y != y)
        return y;
    if (y > 1024.0)
        return mc3_from_bits(0x7ff0000000000000ull);
    if (y < -1080.0)
        return 0.0;
    const double kd = __builtin_rint(y);
    const int k = (int)kd;
    const int k1 = k / 2;
    const int k2 = k - k1;
    /* integer y (the wavelet noise model with gamma = 1, the only value the reference
     * supports, src_c/_dwt.c:42): mc3_det_exp(0) is exactly 1, so the product below is the
     * same number without the polynomial (tests/test_detmath_philox.py checks every k) */
    if (y == kd)
        return mc3_pow2i(k1) * mc3_pow2i(k2);
    const double p = mc3_det_exp((y - kd) * 0.6931471805599453);
    return (p * mc3_pow2i(k1)) * mc3_pow2i(k2);
}

/* log(x): the published fdlibm/musl algorithm (x = 2^k (1+f),
 * sqrt(1/2) <= 1+f < sqrt(2); s = f/(2+f); minimax polynomial in s^2).
 * log(0) = -inf, log(x<0) = NaN, log(inf) = inf. */
MC3_HD double mc3_det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01;
    const double Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01;
    const double Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01;
    const double Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    uint64_t ix = mc3_bits(x);
    int k = 0;
    if (x != x)
        return x;
    if ((ix << 1) == 0)
        return mc3_from_bits(0xfff0000000000000ull);          /* -inf */
    if (ix >> 63)
        return mc3_from_bits(0x7ff8000000000000ull);          /* NaN  */
    if (ix >= 0x7ff0000000000000ull)
        return x;                                             /* +inf */
    if (ix < 0x0010000000000000ull) {                         /* subnormal */
        x = x * 18014398509481984.0;                          /* 2^54 */
        ix = mc3_bits(x);
        k -= 54;
    }
    uint32_t hx = (uint32_t)(ix >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    ix = ((uint64_t)hx << 32) | (ix & 0xffffffffull);
    const double f = mc3_from_bits(ix) - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* sin(2*pi*u), cos(2*pi*u) for 0 <= u < 1: exact quadrant split
 * (4u = q + f, |f| <= 1/2), Taylor series in y = f*pi/2, |y| <= pi/4. */
MC3_HD void mc3_det_sincos2pi(double u, double *sn, double *cs)
{
    const double t = 4.0 * u;
    const double qd = __builtin_rint(t);
    const double y = (t - qd) * 1.5707963267948966;
    const double y2 = y * y;
    double ps = 2.8114572543455206e-15;           /* 1/17! */
    ps = mc3_fma(ps, y2, -7.647163731819816e-13); /* 1/15! */
    ps = mc3_fma(ps, y2, 1.6059043836821613e-10); /* 1/13! */
    ps = mc3_fma(ps, y2, -2.505210838544172e-08); /* 1/11! */
    ps = mc3_fma(ps, y2, 2.7557319223985893e-06); /* 1/9!  */
    ps = mc3_fma(ps, y2, -0.0001984126984126984); /* 1/7!  */
    ps = mc3_fma(ps, y2, 0.008333333333333333);   /* 1/5!  */
    ps = mc3_fma(ps, y2, -0.16666666666666666);   /* 1/3!  */
    const double s = mc3_fma(ps * y2, y, y);
    double pc = 1.5619206968586225e-16;           /* 1/18! */
    pc = mc3_fma(pc, y2, -4.779477332387385e-14); /* 1/16! */
    pc = mc3_fma(pc, y2, 1.1470745597729725e-11); /* 1/14! */
    pc = mc3_fma(pc, y2, -2.08767569878681e-09);  /* 1/12! */
    pc = mc3_fma(pc, y2, 2.755731922398589e-07);  /* 1/10! */
    pc = mc3_fma(pc, y2, -2.48015873015873e-05);  /* 1/8!  */
    pc = mc3_fma(pc, y2, 0.001388888888888889);   /* 1/6!  */
    pc = mc3_fma(pc, y2, -0.041666666666666664);  /* 1/4!  */
    pc = mc3_fma(pc, y2, 0.5);
    const double c = mc3_fma(-pc, y2, 1.0);
    const int q = ((int)qd) & 3;
    *sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    *cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

/* base^(e2/2) for a non-negative integer e2: the snooker Hastings factor
 * (nnorm/cnorm)^((nfree-1)/2), mc3/chain.py:255.  Integer powers by repeated
 * multiplication in a fixed order, half-integer ones through one IEEE sqrt. */
MC3_HD double mc3_det_pow_half(double base, int e2)
{
    const double b = (e2 & 1) ? __builtin_sqrt(base) : base;
    const int n = (e2 & 1) ? e2 : e2 / 2;
    double r = 1.0;
    for (int i = 0; i < n; i++)
        r = r * b;
    return r;
}

#endif /* MC3HIP_DETMATH_H */
