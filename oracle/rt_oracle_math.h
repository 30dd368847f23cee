/*
 * oracle/rt_oracle_math.h -- TEST INFRASTRUCTURE (CPU oracle only).
 *
 * Transcendentals used by the CPU restatement of the reference hot path
 * (cosf / sinf / acosf / atan2f call sites: /root/reference/kernel.cu:252-255,
 * 1157-1158, 1267-1277, 1402-1403, 1451, 1462-1463, 1466).
 *
 * The reference calls CUDA libdevice (built with FastMath=true,
 * "Ray Tracer engine.vcxproj":68,97), which is not bit-specified. The oracle
 * therefore fixes ONE definition: evaluate in binary64 with plain + - * / sqrt
 * (no FMA, no libm) and round once to binary32. The result is the correctly
 * rounded float in all but ~1e-9 of cases, i.e. it agrees with any
 * correctly-rounded libm, and it is bit-reproducible on every IEEE-754 machine
 * -- which is what lets the HIP kernel be compared bit-for-bit.
 *
 * Build with -DRT_ORACLE_LIBM to swap in the host libm instead (used by the
 * tests to bound how much the choice of libm matters: max relative error on
 * non-flipped pixels + flipped-pixel fraction).
 *
 * Constants are derived by tools/gen_math_consts.py from exact rationals.
 */
#ifndef RT_ORACLE_MATH_H
#define RT_ORACLE_MATH_H

#include <math.h>

#ifdef RT_ORACLE_LIBM

static inline float o_cosf(float x) { return cosf(x); }
static inline float o_sinf(float x) { return sinf(x); }
static inline float o_acosf(float x) { return acosf(x); }
static inline float o_atan2f(float y, float x) { return atan2f(y, x); }

#else

#define O_PIO2_HEAD 0x1.921fb54400000p+0   /* first 33 bits of pi/2 */
#define O_PIO2_TAIL 0x1.0b4611a626331p-34  /* pi/2 - head            */
#define O_TWO_OVER_PI 0x1.45f306dc9c883p-1
#define O_PI 0x1.921fb54442d18p+1
#define O_PIO2 0x1.921fb54442d18p+0
#define O_PIO4 0x1.921fb54442d18p-1
#define O_3PIO4 0x1.2d97c7f3321d2p+1

/* sin(r), |r| <= pi/4 : Taylor through r^15 (truncation < 1e-16 relative). */
static inline double o_ksin(double r)
{
    double z = r * r;
    double p = -0x1.ae7f3e733b81fp-41;
    p = p * z + 0x1.6124613a86d09p-33;
    p = p * z + -0x1.ae64567f544e4p-26;
    p = p * z + 0x1.71de3a556c734p-19;
    p = p * z + -0x1.a01a01a01a01ap-13;
    p = p * z + 0x1.1111111111111p-7;
    p = p * z + -0x1.5555555555555p-3;
    return r + r * (z * p);
}

/* cos(r), |r| <= pi/4 : Taylor through r^16. */
static inline double o_kcos(double r)
{
    double z = r * r;
    double p = 0x1.ae7f3e733b81fp-45;
    p = p * z + -0x1.93974a8c07c9dp-37;
    p = p * z + 0x1.1eed8eff8d898p-29;
    p = p * z + -0x1.27e4fb7789f5cp-22;
    p = p * z + 0x1.a01a01a01a01ap-16;
    p = p * z + -0x1.6c16c16c16c17p-10;
    p = p * z + 0x1.5555555555555p-5;
    p = p * z + -0x1.0000000000000p-1;
    return 1.0 + z * p;
}

/* Cody-Waite reduction in binary64; exact k*head for |k| < 2^20, which covers
 * every argument the hot path produces (|x| <= 2*pi). Larger |x| still works
 * but loses the correctly-rounded guarantee above ~1e5. */
static inline double o_reduce(float x, int *quad)
{
    double xd = (double)x;
    double k = rint(xd * O_TWO_OVER_PI);
    double r = (xd - k * O_PIO2_HEAD) - k * O_PIO2_TAIL;
    *quad = (int)((long long)k & 3);
    return r;
}

static inline float o_cosf(float x)
{
    if (!(fabsf(x) < 1.0e9f)) return x - x; /* inf/NaN -> NaN; absurd range -> 0 */
    int q;
    double r = o_reduce(x, &q);
    double v;
    switch (q) {
    case 0: v = o_kcos(r); break;
    case 1: v = -o_ksin(r); break;
    case 2: v = -o_kcos(r); break;
    default: v = o_ksin(r); break;
    }
    return (float)v;
}

static inline float o_sinf(float x)
{
    if (!(fabsf(x) < 1.0e9f)) return x - x;
    int q;
    double r = o_reduce(x, &q);
    double v;
    switch (q) {
    case 0: v = o_ksin(r); break;
    case 1: v = o_kcos(r); break;
    case 2: v = -o_ksin(r); break;
    default: v = -o_kcos(r); break;
    }
    return (float)v;
}

static const double o_atan_tab[9] = {
    0x0.0p+0,
    0x1.fd5ba9aac2f6ep-4,
    0x1.f5b75f92c80ddp-3,
    0x1.6f61941e4def1p-2,
    0x1.dac670561bb4fp-2,
    0x1.1e00babdefeb4p-1,
    0x1.4978fa3269ee1p-1,
    0x1.700a7c5784634p-1,
    0x1.921fb54442d18p-1,
};

/* atan(num/den) for num >= 0, den >= 0, not both zero, result in [0, pi/2]. */
static inline double o_atan_pos(double num, double den)
{
    int swap = num > den;
    double a = swap ? den / num : num / den; /* in [0,1] */
    int idx = (int)(a * 8.0 + 0.5);
    double c = (double)idx * 0.125;
    double z = (a - c) / (1.0 + a * c); /* |z| <= 1/16 */
    double w = z * z;
    double p = -0x1.1111111111111p-4;
    p = p * w + 0x1.3b13b13b13b14p-4;
    p = p * w + -0x1.745d1745d1746p-4;
    p = p * w + 0x1.c71c71c71c71cp-4;
    p = p * w + -0x1.2492492492492p-3;
    p = p * w + 0x1.999999999999ap-3;
    p = p * w + -0x1.5555555555555p-2;
    double t = o_atan_tab[idx] + (z + z * (w * p));
    return swap ? O_PIO2 - t : t;
}

static inline double o_atan2d(double y, double x)
{
    double ay = fabs(y), ax = fabs(x);
    double r;
    if (ay == 0.0 && ax == 0.0)
        r = 0.0;
    else if (isinf(ax) && isinf(ay))
        r = O_PIO4;
    else
        r = o_atan_pos(ay, ax);
    if (signbit(x)) r = O_PI - r;
    return signbit(y) ? -r : r;
}

static inline float o_atan2f(float y, float x)
{
    if (isnan(x) || isnan(y)) return x + y;
    return (float)o_atan2d((double)y, (double)x);
}

static inline float o_acosf(float x)
{
    if (!(fabsf(x) <= 1.0f)) return (x - x) / (x - x); /* NaN for |x|>1 and NaN */
    double xd = (double)x;
    double s = sqrt((1.0 - xd) * (1.0 + xd));
    return (float)o_atan2d(s, xd);
}

#endif /* RT_ORACLE_LIBM */

#endif /* RT_ORACLE_MATH_H */
