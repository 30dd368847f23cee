// rt_math.h -- the library's own transcendentals (host + device).
//
// The reference calls CUDA libdevice cosf/sinf/acosf/atan2f
// (/root/reference/kernel.cu:252-255, 1157-1158, 1267-1277, 1402-1403, 1451,
// 1462-1463, 1466), compiled with FastMath, so no bit-level definition exists.
// This library fixes one: evaluate in binary64 using only + - * / sqrt (never
// FMA, never a vendor libm), round once to binary32. That is the correctly
// rounded float except for ~1e-9 of inputs, identical on host and gfx950, and
// it is what makes host-hoisted uniforms (yaw/pitch, per-sample phi) agree
// bit-for-bit with values the device would have computed per pixel.
//
// Constants: tools/gen_math_consts.py (exact rationals / 60-digit decimals).
// This translation unit must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#define RT_HD __host__ __device__ __forceinline__

namespace rtm {

constexpr double kPio2Head = 0x1.921fb54400000p+0;   // top 33 bits of pi/2
constexpr double kPio2Tail = 0x1.0b4611a626331p-34;  // pi/2 - head
constexpr double kTwoOverPi = 0x1.45f306dc9c883p-1;
constexpr double kPi = 0x1.921fb54442d18p+1;
constexpr double kPio2 = 0x1.921fb54442d18p+0;
constexpr double kPio4 = 0x1.921fb54442d18p-1;

// The polynomial coefficients. On the host they are literals. On the DEVICE they are read from a table in memory
// through the constant address space, behind an opaque copy of its address: a binary64 literal is no inline operand
// of the instruction set, so every one of them is materialised in a pair of scalar registers, and the compiler hoists
// all forty pairs out of the frame kernel's loops and holds them across the whole light loop -- the scalar file spilled
// into vector lanes around every loop for it. A scalar load where a polynomial is evaluated costs one instruction per
// eight coefficients and no register outside that evaluation. (Same values, same operations: same bits.)
#define RTM_SIN_COEF {-0x1.ae7f3e733b81fp-41, 0x1.6124613a86d09p-33, -0x1.ae64567f544e4p-26, 0x1.71de3a556c734p-19, \
                      -0x1.a01a01a01a01ap-13, 0x1.1111111111111p-7, -0x1.5555555555555p-3, 0.0}
#define RTM_COS_COEF {0x1.ae7f3e733b81fp-45, -0x1.93974a8c07c9dp-37, 0x1.1eed8eff8d898p-29, -0x1.27e4fb7789f5cp-22, \
                      0x1.a01a01a01a01ap-16, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5, -0x1.0000000000000p-1}
#define RTM_ATAN_COEF {-0x1.1111111111111p-4, 0x1.3b13b13b13b14p-4, -0x1.745d1745d1746p-4, 0x1.c71c71c71c71cp-4, \
                       -0x1.2492492492492p-3, 0x1.999999999999ap-3, -0x1.5555555555555p-2, 0.0}
// {2/pi, head and tail of pi/2 (Cody-Waite), pi, pi/2, 3.1415, RN(1/3.1415), 1/8}
#define RTM_MISC_COEF {0x1.45f306dc9c883p-1, 0x1.921fb54400000p+0, 0x1.0b4611a626331p-34, 0x1.921fb54442d18p+1, \
                       0x1.921fb54442d18p+0, 3.1415, 0x1.45f57ce20d722p-2, 0.125}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const double kMiscCoefDev[8] = RTM_MISC_COEF;
__device__ const double kSinCoefDev[8] = RTM_SIN_COEF;
__device__ const double kCosCoefDev[8] = RTM_COS_COEF;
__device__ const double kAtanCoefDev[8] = RTM_ATAN_COEF;
typedef const double __attribute__((address_space(4))) *CoefPtr;
__device__ __forceinline__ CoefPtr coef_table(const double *table)
{
    CoefPtr p = (CoefPtr)(unsigned long long)table;
    asm volatile("" : "+s"(p));   // opaque: the loads stay where the polynomial is
    return p;
}
#define RTM_COEF(NAME, DEV) const CoefPtr NAME = coef_table(DEV)
#else
#define RTM_COEF(NAME, DEV) static const double NAME##_host[8] = RTM_##NAME; const double *const NAME = NAME##_host
#endif

// sin on [-pi/4, pi/4], odd Taylor polynomial through r^15.
RT_HD double sin_core(double r)
{
    RTM_COEF(SIN_COEF, kSinCoefDev);
    const double z = r * r;
    double p = SIN_COEF[0];
    p = p * z + SIN_COEF[1];
    p = p * z + SIN_COEF[2];
    p = p * z + SIN_COEF[3];
    p = p * z + SIN_COEF[4];
    p = p * z + SIN_COEF[5];
    p = p * z + SIN_COEF[6];
    return r + r * (z * p);
}

// cos on [-pi/4, pi/4], even Taylor polynomial through r^16.
RT_HD double cos_core(double r)
{
    RTM_COEF(COS_COEF, kCosCoefDev);
    const double z = r * r;
    double p = COS_COEF[0];
    p = p * z + COS_COEF[1];
    p = p * z + COS_COEF[2];
    p = p * z + COS_COEF[3];
    p = p * z + COS_COEF[4];
    p = p * z + COS_COEF[5];
    p = p * z + COS_COEF[6];
    p = p * z + COS_COEF[7];
    return 1.0 + z * p;
}

struct Reduced {
    double r;
    int quadrant;
};

// Two-constant Cody-Waite reduction; k*head is exact for |k| < 2^20.
RT_HD Reduced reduce_pio2(float x)
{
    RTM_COEF(MISC_COEF, kMiscCoefDev);
    const double xd = (double)x;
    const double k = __builtin_rint(xd * MISC_COEF[0]);
    Reduced o;
    o.r = (xd - k * MISC_COEF[1]) - k * MISC_COEF[2];
    o.quadrant = (int)((long long)k & 3);
    return o;
}

RT_HD bool in_trig_range(float x) { return __builtin_fabsf(x) < 1.0e9f; }

// cos and sin of the same argument share the reduction (rotate() needs both).
RT_HD void sincosf_rt(float x, float &s, float &c)
{
    if (!in_trig_range(x)) {
        s = c = x - x;
        return;
    }
    const Reduced q = reduce_pio2(x);
    const double sv = sin_core(q.r), cv = cos_core(q.r);
    double so, co;
    switch (q.quadrant) {
    case 0: so = sv; co = cv; break;
    case 1: so = cv; co = -sv; break;
    case 2: so = -sv; co = -cv; break;
    default: so = -cv; co = sv; break;
    }
    s = (float)so;
    c = (float)co;
}

RT_HD float cosf_rt(float x)
{
    if (!in_trig_range(x)) return x - x;
    const Reduced q = reduce_pio2(x);
    double v;
    switch (q.quadrant) {
    case 0: v = cos_core(q.r); break;
    case 1: v = -sin_core(q.r); break;
    case 2: v = -cos_core(q.r); break;
    default: v = sin_core(q.r); break;
    }
    return (float)v;
}

RT_HD float sinf_rt(float x)
{
    if (!in_trig_range(x)) return x - x;
    const Reduced q = reduce_pio2(x);
    double v;
    switch (q.quadrant) {
    case 0: v = sin_core(q.r); break;
    case 1: v = cos_core(q.r); break;
    case 2: v = -sin_core(q.r); break;
    default: v = -cos_core(q.r); break;
    }
    return (float)v;
}

// atan(k/8), k = 0..8, selected without a memory table (keeps the device
// version free of divergent constant loads).
RT_HD double atan_eighth(int k)
{
    switch (k) {
    case 0: return 0x0.0p+0;
    case 1: return 0x1.fd5ba9aac2f6ep-4;
    case 2: return 0x1.f5b75f92c80ddp-3;
    case 3: return 0x1.6f61941e4def1p-2;
    case 4: return 0x1.dac670561bb4fp-2;
    case 5: return 0x1.1e00babdefeb4p-1;
    case 6: return 0x1.4978fa3269ee1p-1;
    case 7: return 0x1.700a7c5784634p-1;
    default: return 0x1.921fb54442d18p-1;
    }
}

// atan(num/den), num >= 0, den >= 0, not both zero; result in [0, pi/2]. `tab`, when
// given, holds atan_eighth(0..8) (the device keeps a copy in LDS: one read instead of a
// nine-way select of 64-bit constants).
RT_HD double atan_first_quadrant(double num, double den, const double *tab = nullptr)
{
    RTM_COEF(MISC_COEF, kMiscCoefDev);
    const bool swap = num > den;
    const double a = swap ? den / num : num / den;
    const int idx = (int)(a * 8.0 + 0.5);
    const double c = (double)idx * MISC_COEF[7];
    const double z = (a - c) / (1.0 + a * c);
    const double w = z * z;
    RTM_COEF(ATAN_COEF, kAtanCoefDev);
    double p = ATAN_COEF[0];
    p = p * w + ATAN_COEF[1];
    p = p * w + ATAN_COEF[2];
    p = p * w + ATAN_COEF[3];
    p = p * w + ATAN_COEF[4];
    p = p * w + ATAN_COEF[5];
    p = p * w + ATAN_COEF[6];
    const double t = (tab ? tab[idx] : atan_eighth(idx)) + (z + z * (w * p));
    return swap ? MISC_COEF[4] - t : t;
}

RT_HD double atan2_d(double y, double x, const double *tab = nullptr)
{
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    double r;
    if (ay == 0.0 && ax == 0.0) {
        r = 0.0;
    } else {
        // inf/inf is pi/4: evaluated as atan(1/1), which is atan_eighth(8) + 0 = kPio4 exactly
        // (a separate `r = kPio4` makes the device compiler keep that literal in registers
        // across the whole kernel)
        const bool both_inf = __builtin_isinf(ax) && __builtin_isinf(ay);
        r = atan_first_quadrant(both_inf ? 1.0 : ay, both_inf ? 1.0 : ax, tab);
    }
    {
        RTM_COEF(MISC_COEF, kMiscCoefDev);
        if (__builtin_signbit(x)) r = MISC_COEF[3] - r;
    }
    return __builtin_signbit(y) ? -r : r;
}

RT_HD float atan2f_rt(float y, float x, const double *tab = nullptr)
{
    if (x != x || y != y) return x + y;
    return (float)atan2_d((double)y, (double)x, tab);
}

RT_HD float acosf_rt(float x, const double *tab = nullptr)
{
    if (!(__builtin_fabsf(x) <= 1.0f)) return (x - x) / (x - x);
    const double xd = (double)x;
    const double s = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
    return (float)atan2_d(s, xd, tab);
}

// x / 3.1415 in binary64 for x = (double)(a float), without a division: q = x*RC, one
// residual correction. Verified EXHAUSTIVELY (all 2^32 floats, tests/test_const_div.py) to
// equal the IEEE quotient bit for bit, except that -0 gives +0 -- callers add 1.0 to the
// result or never pass -0. The oracle keeps the plain division (kernel.cu:1402-1403).
RT_HD double div_by_3p1415(double x)
{
    RTM_COEF(MISC_COEF, kMiscCoefDev);
    const double c = MISC_COEF[5], rc = MISC_COEF[6];   // 3.1415, RN(1/3.1415)
    const double q = x * rc;
    const double r = __builtin_fma(-c, q, x);
    return __builtin_fma(r, rc, q);
}

}  // namespace rtm
