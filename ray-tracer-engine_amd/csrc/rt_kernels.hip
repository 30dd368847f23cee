// rt_kernels.hip -- gfx950 (MI355X) kernels for the ray-tracing hot path.
//
// What is computed is the reference's `rayTrace` kernel restricted to its
// sphere path (/root/reference/kernel.cu:1614-1690 and callees: castRay
// :1287-1431, castLightRay :1432-1544, sphere::intersect :292-354,
// skybox::getFColor :1146-1166, rgbToInt :546-556). How it is computed is not
// the reference's one-thread-per-pixel brute force:
//
//   * one wave64 owns a TW x (64/TW) pixel tile; a 256-thread workgroup stages
//     the sphere table {cx,cy,cz,radius^2} into LDS once;
//   * per tile the wave cooperatively culls the table against a conservative
//     bound of the tile's rays (a cone for the primary rays, a cone-capped beam
//     for each light's shadow rays), ballot-compacts the survivors IN LIST
//     ORDER into a per-wave LDS list, and only those are tested; every lane
//     reads the same list entry (LDS broadcast);
//   * every value that decides a pixel (the quadratic, sqrt, divisions, the
//     shadow-sample construction) is evaluated with exactly the reference's
//     IEEE binary32/binary64 operations -- this file is compiled with
//     -ffp-contract=off and correctly rounded divide/sqrt; only the culling
//     bounds use fast approximate math, and they are padded so that a culled
//     sphere is one whose exact test would have returned false;
//   * shadow rays leave their loop through a wave-wide "all lanes occluded".
//
// A sphere skipped by culling can never change the closest hit (it is not hit)
// nor an any-hit result, and survivors keep their list order, so first-index-
// wins ties (kernel.cu:1335) resolve identically: the output is bit-identical
// to the brute-force loops (template CULL=false), which tests check.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"
#include "rt_math.h"

// Timing experiments (RT_ABLATE) exist in tuning builds only (make EXTRA=-DRT_TUNING, tools/variants.sh):
// the product kernel has no such switch and the product library reads no environment.
#ifdef RT_TUNING
#define RT_ABL(bits) ((fc.ablate & (bits)) != 0)
#else
#define RT_ABL(bits) false
#endif

namespace {

// RtFrameAux is read through the CONSTANT address space: the memory does not change while a
// kernel runs, and a load from a wave-uniform address there is a scalar load (s_load into
// SGPRs) wherever it stands -- as a plain global pointer the compiler has to assume the
// kernel's own stores may alias it and uses per-lane vector loads into VGPRs.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const RtFrameAux __attribute__((address_space(4))) *AuxPtr;
typedef const RtFrameConsts __attribute__((address_space(4))) *FcPtr;
#else
typedef const RtFrameAux *AuxPtr;   // host pass over this translation unit (device functions are only parsed there)
typedef const RtFrameConsts *FcPtr;
#endif

struct V3 {
    float x, y, z;
};

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
// Wave-wide reductions on the VALU's DPP path (no LDS round trips): butterfly
// inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row
// results upward so that lane 63 holds the reduction, which is broadcast back
// through an SGPR (only lane 63 is meaningful after the broadcast steps, which run on all
// rows: lanes without a source get `old`). All 64 lanes must be active (callers are in
// uniform control flow). `old` is the operation's identity, so that the compiler folds each
// move into the operation (one v_max_u32_dpp / v_add_f32_dpp per step, nothing to canonicalise).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move0(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// max over the wave of NON-NEGATIVE floats, compared as unsigned integers (same order);
// a NaN compares above +inf and therefore survives into the result.
__device__ __forceinline__ float wave_max(float f)
{
    unsigned v = __builtin_bit_cast(unsigned, f);
#define RT_STEP(CTRL, MASK) { const unsigned m = (unsigned)dpp_move0<CTRL, MASK>((int)v); v = v > m ? v : m; }
    RT_STEP(0xB1, 0xf)    /* quad_perm [1,0,3,2]  */
    RT_STEP(0x4E, 0xf)    /* quad_perm [2,3,0,1]  */
    RT_STEP(0x141, 0xf)   /* row_half_mirror      */
    RT_STEP(0x140, 0xf)   /* row_mirror           */
    RT_STEP(0x142, 0xf)   /* row_bcast:15 -> rows 1,3 */
    RT_STEP(0x143, 0xf)   /* row_bcast:31 -> rows 2,3 */
#undef RT_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ float wave_sum(float v)
{
#define RT_STEP(CTRL, MASK) v = v + __builtin_bit_cast(float, dpp_move0<CTRL, MASK>(__builtin_bit_cast(int, v)));
    RT_STEP(0xB1, 0xf) RT_STEP(0x4E, 0xf) RT_STEP(0x141, 0xf) RT_STEP(0x140, 0xf) RT_STEP(0x142, 0xf) RT_STEP(0x143, 0xf)
#undef RT_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// three sums at once, their steps interleaved (a DPP read needs two wait states after the
// write of its source: with three chains in flight no s_nop is needed)
__device__ __forceinline__ void wave_sum3(float &a, float &b, float &c)
{
#define RT_STEP(CTRL, MASK)                                                                    \
    {                                                                                          \
        const float ta = __builtin_bit_cast(float, dpp_move0<CTRL, MASK>(__builtin_bit_cast(int, a))); \
        const float tb = __builtin_bit_cast(float, dpp_move0<CTRL, MASK>(__builtin_bit_cast(int, b))); \
        const float tc = __builtin_bit_cast(float, dpp_move0<CTRL, MASK>(__builtin_bit_cast(int, c))); \
        a = a + ta; b = b + tb; c = c + tc;                                                    \
    }
    RT_STEP(0xB1, 0xf) RT_STEP(0x4E, 0xf) RT_STEP(0x141, 0xf) RT_STEP(0x140, 0xf) RT_STEP(0x142, 0xf) RT_STEP(0x143, 0xf)
#undef RT_STEP
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
    c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
}
__device__ __forceinline__ float uniform(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
// Lanes of one wave exchange data through LDS without a workgroup barrier.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// exact (reference-order) vector helpers, kernel.cu:46-108
// ---------------------------------------------------------------------------
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// normalise(vec3d&): l = sqrtf(dot); if (l != 0) v /= l (in place) and return it,
// else return (0,0,0) leaving v alone. The reference divides in binary64 and
// narrows; for binary32 operands that equals the correctly rounded binary32
// quotient (53 >= 2*24+2), so a float division reproduces it bit for bit.
__device__ __forceinline__ V3 normalise_inplace(V3 &v)
{
    const float l = __builtin_sqrtf(dot3(v, v));
    if (l != 0.f) {
        v.x = v.x / l;
        v.y = v.y / l;
        v.z = v.z / l;
        return v;
    }
    return V3{0.f, 0.f, 0.f};
}

// ---------------------------------------------------------------------------
// The same normalise() -- and the same correctly rounded binary32 square root --
// without the instructions that only matter outside the ordinary range.
//
// hipcc expands an IEEE `a / b` into v_div_scale (x2), v_rcp, four FMAs, a multiply,
// v_div_fmas and v_div_fixup, and an IEEE sqrtf into a pre-scaling select, v_sqrt, the
// one-ulp-down / one-ulp-up residual test and a class fix-up (see the disassembly of
// normalise_inplace above: 55 instructions). Pre-scaling and fix-up only act on
// denormal, huge, zero, infinite or NaN operands; for operands in an ordinary range
// v_div_scale returns its operand unchanged and clears VCC, v_div_fmas is then a plain
// FMA and v_div_fixup returns its first operand. lean_sqrt() and the division below are
// those expansions with exactly these no-ops left out -- every remaining instruction is
// the one the full expansion executes, on the same operands -- and the reciprocal
// refinement (which depends on the divisor only) shared by the three quotients: 35
// instructions. Lanes outside the safe range take the full IEEE expansion:
//   every component >= 2^-48 in magnitude (so dot >= 2^-96, above sqrt's pre-scaling
//   threshold, no quotient below 2^-68, no numerator the scaling would touch) and
//   dot <= 2^40.
// A zero component is "unsafe" (the lean sequence would lose the sign of -0).
// tests/test_gpu_parity.py compares both functions with the IEEE ones exhaustively
// (sqrt: every float of the range) and on 2^28 random vectors; the brute-force and
// force-slow kernels keep the IEEE forms, so every cull-vs-brute comparison checks
// them against each other as well.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float lean_sqrt(float x)   // x in [2^-96, 2^126]
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rd = __builtin_fmaf(-sd, s, x);
    const float ru = __builtin_fmaf(-su, s, x);
    float r = (rd <= 0.f) ? sd : s;
    r = (ru > 0.f) ? su : r;
    return r;
}

struct LeanRcp {   // the divisor-only part of the division expansion
    float d, r;
    __device__ __forceinline__ explicit LeanRcp(float den) : d(den)
    {
        const float r0 = __builtin_amdgcn_rcpf(den);
        const float e0 = __builtin_fmaf(-den, r0, 1.f);
        r = __builtin_fmaf(e0, r0, r0);
    }
    __device__ __forceinline__ float divide(float n) const
    {
        const float q0 = n * r;
        const float e1 = __builtin_fmaf(-d, q0, n);
        const float q1 = __builtin_fmaf(e1, r, q0);
        const float e2 = __builtin_fmaf(-d, q1, n);
        return __builtin_fmaf(e2, r, q1);
    }
};

// PREC: 0 = the IEEE forms as hipcc expands them, 1 = lean (the same bits, see above), 2 = FAST: the opt-in
// approximate mode of rt_launch_opts.fast (hardware reciprocal square root / reciprocal / square root,
// binary32 trigonometry, FMA contraction) -- NOT bit-exact; north_star's tolerance is 1e-5 relative per
// channel away from the discrete decisions (a shadow sample, a texel, a silhouette), see DESIGN.md section 4c.
template <int PREC>
__device__ __forceinline__ V3 normalise_t(V3 &v)
{
    if constexpr (PREC == 0) {
        return normalise_inplace(v);
    } else if constexpr (PREC == 2) {
        const float d2 = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z));
        if (d2 > 0.f) {   // a zero vector stays as it is and (0,0,0) is returned, as normalise() does
            const float inv = __builtin_amdgcn_rsqf(d2);
            v.x *= inv; v.y *= inv; v.z *= inv;
            return v;
        }
        return V3{0.f, 0.f, 0.f};
    } else {
        const float d2 = dot3(v, v);
        const float amin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z));
        const bool safe = (amin >= 0x1.0p-48f) & (d2 <= 0x1.0p40f);   // false for a NaN anywhere
        if (__builtin_expect(!safe, 0)) return normalise_inplace(v);
        const LeanRcp rl(lean_sqrt(d2));
        v.x = rl.divide(v.x);
        v.y = rl.divide(v.y);
        v.z = rl.divide(v.z);
        return v;
    }
}

// ---------------------------------------------------------------------------
// Texel index of a unit normal without binary64: castRay's (tx, ty) (kernel.cu:1402-1403)
// only ever select a texel, c_index = (int)(ty*maxY)*maxX + (int)(tx*maxX) (kernel.cu:1653).
// approx_sphere_uv() evaluates tx = (1 + atan2(n.z, n.x)/3.1415)/2 and ty = acos(n.y)/3.1415
// in binary32 with absolute error below RT_UV_DELTA (budget: quotient after one Newton step
// 0.5 ulp, degree-8 polynomial 1.2e-8 + ~1 ulp of evaluation, all quadrant reconstruction done
// in units of the RESULT so that no step rounds at the magnitude of pi: 1.7e-7 for tx, 2.6e-7
// for ty; rt_debug_uv measures it on the device and tests/test_gpu_parity.py asserts half of
// RT_UV_DELTA). A lane is "sure" when both products tx*W, ty*H stay at least
// mu = size * (RT_UV_DELTA + 2^-22) away from every integer (the second term covers the
// rounding of the two float products): then truncation gives the same column and row as the
// exact value would. Unsure lanes (about 5 % of the 8x8 tiles contain one at 512x512) -- and
// anything negative or NaN -- take the exact binary64 path.
// ---------------------------------------------------------------------------
#define RT_UV_DELTA 5.0e-7f

// atan(a) for a in [0, 1], absolute error < 1.0e-7 (Chebyshev fit of atan(sqrt(s))/sqrt(s), degree 8 in s)
__device__ __forceinline__ float atan01(float a)
{
    const float s = a * a;
    float p = 0x1.73776ap-9f;
    p = __builtin_fmaf(p, s, -0x1.0639f6p-6f);
    p = __builtin_fmaf(p, s, 0x1.5ce0b0p-5f);
    p = __builtin_fmaf(p, s, -0x1.330372p-4f);
    p = __builtin_fmaf(p, s, 0x1.b3ae74p-4f);
    p = __builtin_fmaf(p, s, -0x1.22de60p-3f);
    p = __builtin_fmaf(p, s, 0x1.997232p-3f);
    p = __builtin_fmaf(p, s, -0x1.5554a2p-2f);
    p = __builtin_fmaf(p, s, 1.0f);
    return a * p;
}

// angle(y, x) * scale for y >= 0 given as magnitudes: returns scale * atan2(ay, x) in [0, scale*pi],
// reconstructed in units of the result. k_half = scale*pi/2, k_full = scale*pi (both rounded once).
__device__ __forceinline__ float scaled_angle(float ay, float x, float scale, float k_half, float k_full)
{
    const float ax = __builtin_fabsf(x);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float r = __builtin_amdgcn_rcpf(mx);
    const float a0 = mn * r;
    const float a = __builtin_fmaf(__builtin_fmaf(-mx, a0, mn), r, a0);   // mn / mx to half an ulp (NaN for 0/0)
    float u = atan01(a) * scale;
    u = (ay > ax) ? k_half - u : u;
    u = (x < 0.f) ? k_full - u : u;
    return u;
}

__device__ __forceinline__ void approx_sphere_uv(V3 n, float &tx, float &ty)
{
    constexpr double kC = 1.0 / 3.1415, kPiD = 3.14159265358979323846;
    // tx = 0.5 + sign(n.z) * atan2(|n.z|, n.x) / (2 * 3.1415)
    const float u = scaled_angle(__builtin_fabsf(n.z), n.x, (float)(0.5 * kC), (float)(0.25 * kPiD * kC), (float)(0.5 * kPiD * kC));
    tx = __builtin_signbit(n.z) ? 0.5f - u : 0.5f + u;
    // ty = atan2(sqrt((1 - y)(1 + y)), y) / 3.1415; the root to about an ulp (one Newton step)
    const float q = (1.f - n.y) * (1.f + n.y);
    const float s0 = __builtin_amdgcn_sqrtf(q);
    const float rs = __builtin_amdgcn_rsqf(q);
    const float sy = (q > 0.f) ? __builtin_fmaf(__builtin_fmaf(-s0, s0, q), 0.5f * rs, s0) : 0.f;
    ty = scaled_angle(sy, n.y, (float)kC, (float)(0.5 * kPiD * kC), (float)(kPiD * kC));
    if (!(__builtin_fabsf(n.y) <= 1.f)) ty = __builtin_nanf("");   // the exact function returns NaN there
}

// Column/row selection with certainty: returns the linear index row*w + col when both products are
// clear of every integer by the margins, else -1 (the caller evaluates the exact expressions).
__device__ __forceinline__ int sure_texel(float tx, float ty, int w, int h, float mu_x, float mu_y)
{
    const float px = tx * (float)w, py = ty * (float)h;
    const float fx = __builtin_floorf(px), fy = __builtin_floorf(py);
    const float rx = px - fx, ry = py - fy;
    const bool sure = (rx > mu_x) & (rx < 1.f - mu_x) & (ry > mu_y) & (ry < 1.f - mu_y) & (fx >= 0.f) & (fy >= 0.f) &
                      (fx <= 16384.f) & (fy <= 16384.f);
    return sure ? (int)fy * w + (int)fx : -1;
}

// float -> int of the implicit conversions at kernel.cu:1653, 1157-1158, 1682:
// truncation toward zero, NaN -> 0, saturating (v_cvt_i32_f32 semantics, which
// are also CUDA's cvt.rzi.s32.f32).
__device__ __forceinline__ int f2i(float v) { return (int)v; }

// b after n executions of `b += 0.1` (float += double literal, kernel.cu:1538),
// starting from 0: depends only on how many samples were unshadowed. Literals
// instead of a table so that a per-lane n needs no memory (a kernarg array
// indexed per lane would live in scratch); the host re-derives the sequence and
// refuses to launch if it ever disagreed (rt_build_frame_consts).
__device__ __forceinline__ float brightness_steps(int n)
{
    float b = 0x0.0p+0f;
    b = n >= 1 ? 0x1.99999ap-4f : b;
    b = n >= 2 ? 0x1.99999ap-3f : b;
    b = n >= 3 ? 0x1.333334p-2f : b;
    b = n >= 4 ? 0x1.99999ap-2f : b;
    b = n >= 5 ? 0x1.000000p-1f : b;
    b = n >= 6 ? 0x1.333334p-1f : b;
    b = n >= 7 ? 0x1.666668p-1f : b;
    b = n >= 8 ? 0x1.99999cp-1f : b;
    b = n >= 9 ? 0x1.ccccd0p-1f : b;
    b = n >= 10 ? 0x1.000002p+0f : b;
    return b;
}

// The same eleven values for per-lane lookups (copied into LDS once per wave).
__device__ const float kBrightnessSteps[16] = {0x0.0p+0f,      0x1.99999ap-4f, 0x1.99999ap-3f, 0x1.333334p-2f,
                                               0x1.99999ap-2f, 0x1.000000p-1f, 0x1.333334p-1f, 0x1.666668p-1f,
                                               0x1.99999cp-1f, 0x1.ccccd0p-1f, 0x1.000002p+0f, 0.f, 0.f, 0.f, 0.f, 0.f};

// rtm::atan_eighth(0..8), for the same purpose (values checked against the function by
// tests/test_gpu_parity.py through the math debug ops, which use the LDS copy as well).
__device__ const double kAtanEighth[16] = {0x0.0p+0,
                                           0x1.fd5ba9aac2f6ep-4,
                                           0x1.f5b75f92c80ddp-3,
                                           0x1.6f61941e4def1p-2,
                                           0x1.dac670561bb4fp-2,
                                           0x1.1e00babdefeb4p-1,
                                           0x1.4978fa3269ee1p-1,
                                           0x1.700a7c5784634p-1,
                                           0x1.921fb54442d18p-1,
                                           0, 0, 0, 0, 0, 0, 0};

// rgbToInt, kernel.cu:547-556
__device__ __forceinline__ unsigned rgb_to_int(int r, int g, int b)
{
    if (r > 255) r = 255;
    if (g > 255) g = 255;
    if (b > 255) b = 255;
    return (unsigned)(((r & 0xff) << 16) + ((g & 0xff) << 8) + (b & 0xff));
}

// ---------------------------------------------------------------------------
// sphere::intersect, kernel.cu:293-354, on a table entry {cx,cy,cz,radius*radius}
// ---------------------------------------------------------------------------
struct RayK {          // a ray plus the one per-ray constant of the quadratic that every test needs
    float ox, oy, oz;
    float dx, dy, dz;
    float a4;          // 4*A (kernel.cu:334: B*B - 4*A*C)
    // 2*A, the divisor of both roots, only where a root is actually formed (the same expression on the same operands)
    __device__ __forceinline__ float a2() const { return 2.f * ((dx * dx + dy * dy) + dz * dz); }
};

__device__ __forceinline__ RayK make_ray(V3 o, V3 d)
{
    RayK r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    const float A = (d.x * d.x + d.y * d.y) + d.z * d.z;
    r.a4 = 4.f * A;
    return r;
}

struct Quad {  // B, B*B and the discriminant, in the reference's evaluation order
    float h, B, BB, disc;
};

__device__ __forceinline__ Quad quadratic(const RayK &r, float4 s)
{
    const float ocx = r.ox - s.x, ocy = r.oy - s.y, ocz = r.oz - s.z;
    Quad q;
    q.h = (r.dx * ocx + r.dy * ocy) + r.dz * ocz;
    q.B = 2.f * q.h;
    const float C = ((ocx * ocx + ocy * ocy) + ocz * ocz) - s.w;
    q.BB = q.B * q.B;
    q.disc = q.BB - r.a4 * C;
    return q;
}

// FAST mode: the same quadratic with fused multiply-adds (11 instructions instead of 17)
__device__ __forceinline__ Quad quadratic_fast(const RayK &r, float4 s)
{
    const float ocx = r.ox - s.x, ocy = r.oy - s.y, ocz = r.oz - s.z;
    Quad q;
    q.h = __builtin_fmaf(r.dx, ocx, __builtin_fmaf(r.dy, ocy, r.dz * ocz));
    q.B = 2.f * q.h;
    const float C = __builtin_fmaf(ocx, ocx, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocz, ocz, -s.w)));
    q.BB = q.B * q.B;
    q.disc = __builtin_fmaf(-r.a4, C, q.BB);
    return q;
}

// The tail of intersect once B and the discriminant are known.
__device__ __forceinline__ bool intersect_tail(const RayK &r, const Quad &q, float &t)
{
    const float sq = __builtin_sqrtf(q.disc);
    const float a2 = r.a2();
    t = (-q.B + sq) / a2;
    if (t == 0.f) return true;
    if (t >= RT_T_MIN) {
        const float t2 = (-q.B - sq) / a2;
        if (t > t2) t = t2;
        return true;
    }
    return false;
}

// The same tail with the lean square root and the lean divisions (see normalise_t above; both roots
// divide by the same 2A, so the divisor's part of the expansion is shared). Safe range, per lane:
// disc in [2^-96, 2^60], 2A in [2^-20, 2^20], both numerators between 2^-40 and 2^40 in magnitude --
// anything else (a zero numerator in particular: the reference's `t == 0` clause) takes the IEEE forms.
template <int PREC>
__device__ __forceinline__ bool intersect_tail_t(const RayK &r, const Quad &q, float &t)
{
    if constexpr (PREC == 0) {
        return intersect_tail(r, q, t);
    } else if constexpr (PREC == 2) {
        const float sq = __builtin_amdgcn_sqrtf(q.disc);
        const float inv = __builtin_amdgcn_rcpf(r.a2());
        t = (-q.B + sq) * inv;
        if (t == 0.f) return true;
        if (t >= RT_T_MIN) {
            const float t2 = (-q.B - sq) * inv;
            if (t > t2) t = t2;
            return true;
        }
        return false;
    } else {
        const float nb = -q.B, a2 = r.a2();
        const bool pre = (q.disc >= 0x1.0p-96f) & (q.disc <= 0x1.0p60f) & (a2 >= 0x1.0p-20f) & (a2 <= 0x1.0p20f);
        if (__builtin_expect(!pre, 0)) return intersect_tail(r, q, t);
        const float sq = lean_sqrt(q.disc);
        const float n1 = nb + sq, n2 = nb - sq;
        const bool safe = (__builtin_fminf(__builtin_fabsf(n1), __builtin_fabsf(n2)) >= 0x1.0p-40f) &
                          (__builtin_fmaxf(__builtin_fabsf(n1), __builtin_fabsf(n2)) <= 0x1.0p40f);
        if (__builtin_expect(!safe, 0)) return intersect_tail(r, q, t);
        const LeanRcp ra(a2);
        t = ra.divide(n1);
        if (t >= RT_T_MIN) {           // t == 0 cannot happen here: |n1| >= 2^-40 and 2A <= 2^20
            const float t2 = ra.divide(n2);
            if (t > t2) t = t2;
            return true;
        }
        return false;
    }
}


// plane::intersect, kernel.cu:370-380
__device__ __forceinline__ bool plane_intersect(const RtPlaneDev &p, V3 o, V3 d, float &t)
{
    const float denom = (p.nx * d.x + p.ny * d.y) + p.nz * d.z;
    if (denom < 0.f) {
        const V3 pl0{p.ox - o.x, p.oy - o.y, p.oz - o.z};
        t = ((pl0.x * p.nx + pl0.y * p.ny) + pl0.z * p.nz) / denom;
        return t >= 0.f;
    }
    return false;
}

// cube::intersect, kernel.cu:457-485. min/max are the reference's macros
// (kernel.cu:16-26), whose NaN behaviour differs from fminf/fmaxf. `inv` is
// 1.f / Dir per component, hoisted out of the per-cube call.
#define RT_MAXM(a, b) (((a) > (b)) ? (a) : (b))
#define RT_MINM(a, b) (((a) < (b)) ? (a) : (b))
__device__ __forceinline__ bool cube_intersect(const RtCubeDev &c, V3 o, V3 inv, float &t)
{
    const float t1 = (c.ax - o.x) * inv.x, t2 = (c.bx - o.x) * inv.x;
    const float t3 = (c.ay - o.y) * inv.y, t4 = (c.by - o.y) * inv.y;
    const float t5 = (c.az - o.z) * inv.z, t6 = (c.bz - o.z) * inv.z;
    const float tmin = RT_MAXM(RT_MAXM(RT_MINM(t1, t2), RT_MINM(t3, t4)), RT_MINM(t5, t6));
    const float tmax = RT_MINM(RT_MINM(RT_MAXM(t1, t2), RT_MAXM(t3, t4)), RT_MAXM(t5, t6));
    if (tmax < 0.f) { t = tmax; return false; }
    if (tmax < tmin) { t = tmax; return false; }
    t = tmin;
    return true;
}

// mesh::rayIntersect (Moller-Trumbore), kernel.cu:1024-1059. The two double
// literals there (`a < 0.0000001`, `t > 0.0000001`) compare the widened float
// with 1e-7; 1e-7f is the smallest binary32 >= 1e-7, so `a < 1e-7f` and
// `t >= 1e-7f` are the same predicates.
__device__ __forceinline__ bool tri_intersect(V3 o, V3 d, const float *p0, const float *p1, const float *p2,
                                              float &t, float &u, float &v)
{
    const V3 e1{p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    const V3 e2{p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const V3 h{d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x};
    const float a = dot3(e1, h);
    if (a > -0.0000001f && a < 0.0000001f) return false;
    const float f = 1.f / a;
    const V3 s{o.x - p0[0], o.y - p0[1], o.z - p0[2]};
    u = f * dot3(s, h);
    if (u < 0.f || u > 1.f) return false;
    const V3 q{s.y * e1.z - s.z * e1.y, s.z * e1.x - s.x * e1.z, s.x * e1.y - s.y * e1.x};
    v = f * dot3(d, q);
    if (v < 0.f || u + v > 1.f) return false;
    t = f * dot3(e2, q);
    return t >= 0.0000001f;
}

__device__ __forceinline__ bool box_intersect(const RtBoxDev &b, V3 o, V3 inv)
{
    RtCubeDev c;
    c.ax = b.lo[0]; c.ay = b.lo[1]; c.az = b.lo[2];
    c.bx = b.hi[0]; c.by = b.hi[1]; c.bz = b.hi[2];
    float t;
    return cube_intersect(c, o, inv, t);
}

// A ray that starts outside a sphere whose centre lies behind it has B > 0 and
// disc < B*B; then sqrt(disc) < B, t < 0 strictly and intersect() is false. The
// factor keeps sqrt(disc) below B even after rounding, so the `t == 0` clause
// (kernel.cu:338) cannot fire. Purely a shortcut: when in doubt the full tail runs.
#define RT_BEHIND_FACTOR 0.99999f

// ---------------------------------------------------------------------------
// conservative culling
// ---------------------------------------------------------------------------
struct Beam {        // all members wave-uniform
    float ax, ay, az;   // a point on the axis
    float ux, uy, uz;   // unit axis
    float smin;         // rays start at axial coordinate >= smin
    float smax;         // ... and <= smax (used only by the full-occluder test)
    float r0;           // ... within r0 of the axis
    float k;            // and spread with slope k = tan(theta)
};

// Keep sphere s unless no ray inside the beam can make intersect() return true.
// intersect() is true only if the float discriminant is >= 0, which (rounding
// included) needs the ray's line within sqrt(R^2 + eps*(1+|oc|^2)) of the centre
// (R^2 = s.w is the squared effective radius), and a far root >= 0.
__device__ __forceinline__ bool beam_keeps(const Beam &b, float4 s)
{
    const float vx = s.x - b.ax, vy = s.y - b.ay, vz = s.z - b.az;
    const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float sa = __builtin_fmaf(vx, b.ux, __builtin_fmaf(vy, b.uy, vz * b.uz));
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-sa, sa, vv), 0.f);
    // padded radius: rounding noise of the exact test + slack of this test
    const float rc2 = s.w + __builtin_fmaf(4.0e-5f, vv, 1.0e-3f);
    const float rc = __builtin_amdgcn_sqrtf(rc2) * 1.0001f;
    const float reach = sa + rc - b.smin;
    const float rad = __builtin_fmaf(b.k, __builtin_fmaxf(reach, 0.f), b.r0) + rc;
    return (reach >= 0.f) && (d2 <= rad * rad * 1.0005f);
}

// A block of the Morton-ordered table: {centre, radius} of a sphere containing all
// of its members (host side, rounded up). A member passes beam_keeps() only if
// the block passes this test: the block's padded radius covers the member's
// centre offset, its radius and its own padding sqrt(4e-5*|v|^2 + 1e-3)
// (<= 6.4e-3*|v| + 0.032 with |v| <= |v_block| + r_block).
__device__ __forceinline__ bool beam_keeps_block(const Beam &b, float4 blk)
{
    const float vx = blk.x - b.ax, vy = blk.y - b.ay, vz = blk.z - b.az;
    const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float sa = __builtin_fmaf(vx, b.ux, __builtin_fmaf(vy, b.uy, vz * b.uz));
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-sa, sa, vv), 0.f);
    const float dist = __builtin_amdgcn_sqrtf(vv) * 1.0001f;
    const float rc = __builtin_fmaf(6.5e-3f, dist + blk.w, blk.w) + 0.04f;
    const float reach = sa + rc - b.smin;
    const float rad = __builtin_fmaf(b.k, __builtin_fmaxf(reach, 0.f), b.r0) + rc;
    return !(reach < 0.f) && !(d2 > rad * rad * 1.0005f);   // NaN / inf bounds keep the block
}

// A column block of a light's table (RtFrameConsts::lsorted/lblocks): blkA = {point c on the
// column's axis, lateral radius rho}, blkB = {s_hi, r3d, -, -}. A member j passes
// beam_member_test() only if its block passes this test: its axial coordinate plus radius is
// at most (c - a).u + s_hi, its centre lies within rho - R_j of the column's axis, and its
// padding sqrt(4e-5 |v_j|^2 + 1e-3) <= 6.4e-3 |v_j| + 0.032 with |v_j| <= |c - a| + r3d.
// Only valid for beams whose axis is the light's u (all shadow beams are).
__device__ __forceinline__ bool beam_keeps_column(const Beam &b, float4 blkA, float4 blkB)
{
    const float vx = blkA.x - b.ax, vy = blkA.y - b.ay, vz = blkA.z - b.az;
    const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float sa = __builtin_fmaf(vx, b.ux, __builtin_fmaf(vy, b.uy, vz * b.uz));
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-sa, sa, vv), 0.f);
    const float dist = __builtin_amdgcn_sqrtf(vv) * 1.0001f;
    const float pad = __builtin_fmaf(6.5e-3f, dist + blkB.y, 0.04f);
    const float reach = sa + blkB.x + pad - b.smin;
    const float rad = __builtin_fmaf(b.k, __builtin_fmaxf(reach, 0.f), b.r0) + blkA.w + pad;
    return !(reach < 0.f) & !(d2 > rad * rad * 1.001f);   // NaN / inf bounds keep the block
}

// The sphere table is either the workgroup's LDS copy (TABLDS, up to a few
// thousand spheres) or read straight from global memory (any N; coalesced 16 B
// per lane, L2-resident), in which case LDS only holds the survivor lists.
template <bool TABLDS>
__device__ __forceinline__ float4 table_at(const float4 *lds_tab, const float4 *__restrict__ gl_tab, int i)
{
    if constexpr (TABLDS) return lds_tab[i];
    else return gl_tab[i];
}
// An entry of the list being walked: the wave's survivor list, or the whole table.
template <bool TABLDS>
__device__ __forceinline__ float4 entry_at(bool use_list, const float4 *list, const float4 *lds_tab,
                                           const float4 *__restrict__ gl_tab, int e)
{
    if constexpr (TABLDS) {
        const float4 *p = use_list ? list : lds_tab;
        return p[e];
    } else {
        if (use_list) return list[e];
        return gl_tab[e];
    }
}

// The member test of the culling loop: beam_keeps(), and with OCCL the question whether the
// sphere occludes the whole beam, in one straight line -- the two share |v|^2, the axial
// coordinate and the distance from the axis, and written with short-circuit conditions the
// compiler wraps every clause in its own exec-mask region (a third of the loop's instructions).
// `blocked`, only meaningful for kept entries:
// Sphere s certainly occludes EVERY ray of the beam: it lies entirely ahead of all
// ray origins, and the beam's cross-section at the centre's axial coordinate --
// radius r0 + k*(sa - smin) around the axis, both already padded -- sits inside
// the sphere shrunk by the same rounding allowance the cull test adds. Each ray
// then passes within that shrunken radius of the centre in its forward direction,
// so the exact float test has disc > 0 and h far below -h_sure: it returns true.
template <bool OCCL>
__device__ __forceinline__ bool beam_member_test(const Beam &b, float4 s, bool enable, bool &blocked)
{
    const float vx = s.x - b.ax, vy = s.y - b.ay, vz = s.z - b.az;
    const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float sa = __builtin_fmaf(vx, b.ux, __builtin_fmaf(vy, b.uy, vz * b.uz));
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-sa, sa, vv), 0.f);
    const float pad = __builtin_fmaf(4.0e-5f, vv, 1.0e-3f);
    const float rc = __builtin_amdgcn_sqrtf(s.w + pad) * 1.0001f;
    const float reach = sa + rc - b.smin;
    const float rad = __builtin_fmaf(b.k, __builtin_fmaxf(reach, 0.f), b.r0) + rc;
    const bool keep = enable & (reach >= 0.f) & (d2 <= rad * rad * 1.0005f);
    if (OCCL) {
        const float r2b = s.w - pad;                                       // shrunken radius^2
        const float rr = __builtin_amdgcn_sqrtf(s.w);
        const bool ahead = (sa - b.smax) >= __builtin_fmaf(rr, 1.001f, 0.01f);
        const float rho = __builtin_fmaf(b.k, sa - b.smin, b.r0);
        const float lhs = __builtin_fmaf(__builtin_amdgcn_sqrtf(d2) + rho, 1.001f, 1.0e-4f);
        blocked = blocked | (keep & ahead & (r2b > 0.f) & (lhs * lhs <= r2b));
    }
    return keep;
}

// Two-level cull over an ordered copy of the table: the blocks of RT_BLOCK the beam can touch,
// then their members (64/RT_BLOCK blocks per step). Survivors come out in the table's order,
// which is fine for an any-hit; for the primary rays (ORDERED) their list positions are
// carried along in keys[] and the short list is re-ordered front to back (see below), ties
// between equal t being settled by those positions (kernel.cu:1335). Returns the survivor
// count (with OCCL, bit 30 flags "one sphere occludes the whole beam"); a count above
// RT_LIST_CAP tells the caller to walk the whole table instead.
// BLOCKS selects the first level: 0 = cubes of the 3-D order (fc.sorted/fc.blocks), 1 = a
// light's columns, 2 = eye cones (the last two: csorted/cblocks/corig, read from global memory).
template <int STATS, bool TABLDS, bool OCCL, bool ORDERED, int BLOCKS = 0>
__device__ __forceinline__ int build_list2(const float4 *tab, const RtFrameConsts &fc, int n, float4 *list, int *keys,
                                           int *blist, const Beam &b, int lane, unsigned long long &n_cull,
                                           const float4 *__restrict__ csorted = nullptr,
                                           const float4 *__restrict__ cblocks = nullptr,
                                           const int *__restrict__ corig = nullptr, bool stop_when_blocked = false)
{
    constexpr bool COLUMNS = BLOCKS != 0;   // two float4 per block, table in global memory
    const float4 *__restrict__ gsorted = COLUMNS ? csorted : reinterpret_cast<const float4 *>(fc.sorted);
    const float4 *__restrict__ gblocks = COLUMNS ? cblocks : reinterpret_cast<const float4 *>(fc.blocks);
    const int *__restrict__ gorig = COLUMNS ? corig : fc.orig_idx;
    // eye cones: cos and sin of the beam's own half-angle (slope padded as in the host's bound)
    float cone_cw = 1.f, cone_sw = 0.f;
    if (BLOCKS == 2) {
        const float kw = b.k * 1.001f;
        cone_cw = __builtin_amdgcn_rsqf(__builtin_fmaf(kw, kw, 1.f));
        cone_sw = kw * cone_cw;
    }
    const int nb = fc.n_blocks;
    int count = 0;
    bool blk = false;
    constexpr int G = 64 / RT_BLOCK;           // blocks examined side by side in one step
    const int grp = lane / RT_BLOCK, sub = lane % RT_BLOCK;
    for (int bbase = 0; bbase < nb; bbase += 64) {
        const int bi = bbase + lane;
        const int bc = bi < nb ? bi : nb - 1;
        bool kb;
        if (BLOCKS == 2) {
            // angle(axis, beam axis) <= theta + theta_beam, both below pi (build_eye_cones)
            const float4 ba = gblocks[2 * bc], bbx = gblocks[2 * bc + 1];
            const float dotp = __builtin_fmaf(ba.x, b.ux, __builtin_fmaf(ba.y, b.uy, ba.z * b.uz));
            const float rhs = __builtin_fmaf(ba.w, cone_cw, -bbx.x * cone_sw) - 2.0e-5f;   // cos(theta + theta_beam)
            const bool inside = !(dotp < rhs);                                                 // NaN keeps the block
            kb = (bi < nb) & (bbx.y >= 0.f) & ((bbx.y > 0.f) | inside);
        } else if (BLOCKS == 1) {
            const float4 ba = gblocks[2 * bc], bbx = gblocks[2 * bc + 1];
            kb = (bi < nb) & (ba.w >= 0.f || ba.w != ba.w) & beam_keeps_column(b, ba, bbx);
        } else {
            const float4 bb = gblocks[bc];
            kb = (bi < nb) && (bb.w >= 0.f || bb.w != bb.w) && beam_keeps_block(b, bb);   // w < 0: padding block
        }
        const unsigned long long bm = __ballot(kb);
        if (STATS == 1) n_cull += 64;
        // the marked blocks, compacted into the wave's block list; then G of them per step,
        // one per group of RT_BLOCK lanes (next step's block number read one step ahead)
        const int marked = __popcll(bm);
        if (kb) blist[lane_prefix(bm)] = bi;
        wave_lds_sync();
        int cur = (grp < marked) ? blist[grp] : -1;
        for (int t = 0; t < marked; t += G) {
            const int nslot = t + G + grp;
            const int nxt = (nslot < marked) ? blist[nslot] : -1;
            const int i = (cur < 0 ? 0 : cur) * RT_BLOCK + sub;   // inside the padded table
            const float4 s = table_at<TABLDS && !COLUMNS>(tab, gsorted, i);
            const bool keep = beam_member_test<OCCL>(b, s, (cur >= 0) & (i < n), blk);
            const unsigned long long m = __ballot(keep);
            const int pos = count + lane_prefix(m);
            if (keep && pos < RT_LIST_CAP) {
                list[pos] = s;
                if (ORDERED) keys[pos] = gorig[i];
            }
            count += __popcll(m);
            if (STATS == 1) n_cull += 64;
            cur = nxt;
            // the caller only wants to know whether one sphere occludes the whole beam, and one does:
            // the rest of the list is of no interest (the light adds nothing)
            if (OCCL && stop_when_blocked && __any(blk)) {
                wave_lds_sync();
                return count | 0x40000000;
            }
        }
        if (bbase + 64 < nb) wave_lds_sync();   // the next 64 blocks reuse the block list
    }
    wave_lds_sync();
    if (ORDERED) {
        if (count > 64) {
            count = RT_LIST_CAP + 1;     // too long to reorder in one step: caller walks the table
        } else if (count > 1) {
            // Front to back: the list is ordered by a LOWER BOUND of any t the entry can return
            // to a ray from the apex (|D| = 1): dist - R outside the sphere, -(dist + R) inside
            // (the near root is what intersect() returns), less an allowance for the float
            // evaluation. That allowance is set by grazing rays: B^2 - 4AC is formed with an
            // absolute error of about 12 ulp(dist^2) = 3e-6 dist^2, so sqrt(disc) -- and with it
            // the near root -- can be off by sqrt(7.5e-7) dist = 8.7e-4 dist where the exact
            // discriminant vanishes (where it does not, the error is far smaller and the exact
            // root exceeds dist - R by up to R): 1e-3 + 1.5e-3 (dist + R) covers both cases.
            // The caller stops as soon as every lane's nearest hit lies
            // strictly below the next entry's bound; ties between equal t are resolved by the
            // list positions in keys[] (first index wins, kernel.cu:1335), so the order of
            // evaluation does not matter. The bounds go where the block list was (<= 64 entries).
            float *lbs = reinterpret_cast<float *>(blist);
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            int key = 0, rank = 0;
            float lb = 0.f;
            if (lane < count) {
                e = list[lane];
                key = keys[lane];
                const float vx = e.x - b.ax, vy = e.y - b.ay, vz = e.z - b.az;
                const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
                const float dist = __builtin_amdgcn_sqrtf(vv), rr = __builtin_amdgcn_sqrtf(e.w);
                const float slack = __builtin_fmaf(1.5e-3f, dist + rr, 1.0e-3f);
                lb = (vv > e.w * 1.001f + 1.0e-6f) ? (dist - rr) - slack : -(dist + rr) - slack;
                lb = (lb == lb) ? lb : -__builtin_inff();   // non-finite entries first: they never end the walk early
                lbs[lane] = lb;
            }
            wave_lds_sync();
            for (int j = 0; j < count; ++j) {
                const float lj = lbs[j];
                rank += (lj < lb || (lj == lb && j < lane)) ? 1 : 0;
            }
            wave_lds_sync();
            if (lane < count) {
                list[rank] = e;
                keys[rank] = key;
                lbs[rank] = lb;
            }
            wave_lds_sync();
        }
    }
    if (OCCL && __any(blk)) count |= 0x40000000;
    return count;
}

// One triangle of a leaf against the beam, by its bounding sphere ts = {centre, radius} and tn = {unit normal, kappa}
// (host side: rt_scene_set_mesh). Moller-Trumbore (kernel.cu:1024-1059) accepts a ray when its computed barycentrics
// lie in the unit simplex. Their rounding errors are those of two 3x3 determinants (absolute error <= gamma |s| |e|,
// gamma ~ 10 eps, |s| the distance from the ray origin to the first vertex) divided by a = D . (e2 x e1) =
// |e1| |e2| sin(phi0) cos(theta_n), theta_n the angle between the ray and the normal: in the triangle's plane they
// displace the hit point by at most 2 gamma (|s| + |e|) / (sigma |cos(theta_n)|), sigma the smallest corner sine. For a
// ray with |cos(theta_n)| >= kappa = 3e-3 / sigma that is below 4e-4 (|s| + |e|), inside pad = 2e-3 + 1e-3 (dist + r)
// (|s| <= dist + r, |e| <= 2 r): such a ray can only be accepted if its line comes within pad of the triangle, hence
// of its bounding sphere. A ray that GRAZES the triangle's plane is another matter: the error grows like
// 1 / cos(theta_n) (float Moller-Trumbore does accept rays that pass the sphere at several radii once cos(theta_n)
// drops below 3e-4: tools/mt_grazing.py), so a triangle is only culled when every ray of the beam keeps
// |cos(theta_n)| >= kappa, i.e. |n . u| >= (kappa + k)(1 + k) for the beam's axis u and slope k. A well-shaped
// triangle is "edge-on" to 0.5 % of the directions, a sliver (sigma -> 0, kappa >= 1; also anything degenerate or
// non-finite: normal 0) to all of them: never culled.
__device__ __forceinline__ bool beam_keeps_triangle(const Beam &b, float4 ts, float4 tn)
{
    const float vx = ts.x - b.ax, vy = ts.y - b.ay, vz = ts.z - b.az;
    const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
    const float sa = __builtin_fmaf(vx, b.ux, __builtin_fmaf(vy, b.uy, vz * b.uz));
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-sa, sa, vv), 0.f);
    const float dist = __builtin_amdgcn_sqrtf(vv) * 1.0001f;
    const float rc = __builtin_fmaf(1.0e-3f, dist + ts.w, ts.w + 2.0e-3f) * 1.0001f;
    const float reach = sa + rc - b.smin;
    const float rad = __builtin_fmaf(b.k, __builtin_fmaxf(reach, 0.f), b.r0) + rc;
    const float nu = __builtin_fabsf(__builtin_fmaf(tn.x, b.ux, __builtin_fmaf(tn.y, b.uy, tn.z * b.uz)));
    const bool facing = nu >= (tn.w + b.k) * (1.f + b.k);              // false for a NaN anywhere
    return !facing | (!(reach < 0.f) & !(d2 > rad * rad * 1.0005f));   // NaN / inf keep the triangle
}

// Leaf boxes of the mesh that the beam can touch, as indices in leaf order. A ray
// tests a leaf's triangles only after passing the leaf's slab test, i.e. only if
// it crosses the box, hence its bounding sphere: the sphere test with the usual
// padding is conservative. Order is kept (first triangle wins ties, kernel.cu:1309):
// two levels as for the spheres, but over blocks of RT_BLOCK CONSECUTIVE leaves (the host
// appends their bounding spheres after the leaf spheres), marked blocks and their members
// both taken in increasing order.
__device__ __forceinline__ int build_box_list(const float4 *__restrict__ bsph, int nb, int *list, int *blist, const Beam &b,
                                              int lane)
{
    const int nb_pad = (nb + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK, nblk = nb_pad / RT_BLOCK;
    const float4 *__restrict__ blocks = bsph + nb_pad;
    constexpr int G = 64 / RT_BLOCK;
    const int grp = lane / RT_BLOCK, sub = lane % RT_BLOCK;
    int count = 0;
    for (int bbase = 0; bbase < nblk; bbase += 64) {
        const int bi = bbase + lane;
        const float4 bb = blocks[bi < nblk ? bi : nblk - 1];
        const bool kb = (bi < nblk) && beam_keeps_block(b, bb);
        const unsigned long long bm = __ballot(kb);
        const int marked = __popcll(bm);
        if (kb) blist[lane_prefix(bm)] = bi;
        wave_lds_sync();
        for (int t = 0; t < marked; t += G) {
            const int slot = t + grp;
            const int blk = (slot < marked) ? blist[slot] : -1;
            const int i = (blk < 0 ? 0 : blk) * RT_BLOCK + sub;
            const float4 s = bsph[i];
            const bool keep = (blk >= 0) && (i < nb) && beam_keeps(b, s);
            const unsigned long long m = __ballot(keep);
            const int pos = count + lane_prefix(m);
            if (keep && pos < RT_BOX_CAP) list[pos] = i;
            count += __popcll(m);
        }
        if (bbase + 64 < nblk) wave_lds_sync();
    }
    wave_lds_sync();
    return count;
}

// ---------------------------------------------------------------------------
// castLightRay's sample construction, kernel.cu:1438-1468 (exact)
// ---------------------------------------------------------------------------
template <int LEAN>
struct ShadowChain {
    V3 toL;            // keeps being re-normalised in place by the reference
    bool stable;       // an iteration leaves toL as it found it: every later iteration repeats this one
    bool fixed1;       // normalise() maps toL to itself: later iterations can only differ in `angle`
    float angle;
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;

    // FAST mode: toL itself is exact -- begin() and settle() run as in the exact kernel, because toL
    // multiplies the brightness continuously (kernel.cu:1541) -- but everything that only shapes the
    // sample directions (the angle to the light's edge, the rotation axis, angle and matrix of
    // kernel.cu:1444-1466) is computed ONCE per light from the settled toL, in binary32 with hardware
    // rsq / sqrt and fused multiply-adds, instead of being followed through the ten iterations.
    __device__ __forceinline__ void setup_fast(const RtLightDev &L, V3 start)
    {
        const V3 lpos{L.px, L.py, L.pz};
        const V3 P{-toL.z, 0.f, toL.x};                                     // cross(toL, (0,1,0))
        V3 e0{__builtin_fmaf(P.x, L.size, lpos.x) - start.x, lpos.y - start.y, __builtin_fmaf(P.z, L.size, lpos.z) - start.z};
        const V3 toEdge = normalise_t<2>(e0);
        angle = __builtin_cosf(2.f * __builtin_fmaf(toL.x, toEdge.x, __builtin_fmaf(toL.y, toEdge.y, toL.z * toEdge.z)));
        V3 ax0{-toL.y, toL.x, 0.f};                                         // cross((0,0,1), toL)
        const V3 axis = normalise_t<2>(ax0);
        const float cs = toL.z;                                             // cos(acos(toL.z))
        const float sn = __builtin_amdgcn_sqrtf(__builtin_fmaxf(__builtin_fmaf(-cs, cs, 1.f), 0.f));
        const float omc = 1.f - cs;
        m00 = __builtin_fmaf(axis.x, axis.x, cs);
        m01 = axis.x * axis.y * omc;
        m02 = -axis.y * sn;
        m10 = m01;
        m11 = __builtin_fmaf(axis.y * axis.y, omc, cs);
        m12 = -axis.x * sn;
        m20 = -axis.y * sn;
        m21 = axis.x * sn;
        m22 = cs;
        stable = true;
        fixed1 = true;
    }
    __device__ __forceinline__ V3 direction_fast(AuxPtr ax, const RtLightDev &L, int j)
    {
        const float z = __builtin_fmaf(ax->jf[j], 1.0f - angle, angle);
        const float sq = __builtin_amdgcn_sqrtf(__builtin_fmaf(-z, z, 1.f));   // NaN beyond |z| = 1, as the exact form
        const float x = sq * ax->jcos[j], y = sq * ax->jsin[j];
        V3 nd{L.px - __builtin_fmaf(x, m00, __builtin_fmaf(y, m10, z * m20)),
              L.py - __builtin_fmaf(x, m01, __builtin_fmaf(y, m11, z * m21)),
              L.pz - __builtin_fmaf(x, m02, __builtin_fmaf(y, m12, z * m22))};
        return normalise_t<2>(nd);
    }

    __device__ __forceinline__ void begin(V3 lpos, V3 start)
    {
        // toL = normalise(l.pos - start), kernel.cu:1438
        toL = V3{lpos.x - start.x, lpos.y - start.y, lpos.z - start.z};
        normalise_t<LEAN>(toL);
        stable = false;
        fixed1 = false;
        angle = 0.f;
        m00 = m01 = m02 = m10 = m11 = m12 = m20 = m21 = m22 = 0.f;
    }

    // Only the evolution of toL over the ten iterations (two in-place
    // normalisations each, kernel.cu:1465-1466), for callers that need the final
    // toL of kernel.cu:1541 but none of the sample directions.
    __device__ __forceinline__ void settle()
    {
#pragma unroll 1
        for (int j = 0; j < RT_SHADOW_SAMPLES; ++j) {
            if (__all(stable)) break;
            if (!stable) {
                const V3 tin = toL;
                normalise_t<LEAN>(toL);
                const V3 mid = toL;
                normalise_t<LEAN>(toL);
                // unchanged by the pair, or a fixed point of normalise(): toL has its final value
                stable = ((toL.x == tin.x) && (toL.y == tin.y) && (toL.z == tin.z)) ||
                         ((toL.x == mid.x) && (toL.y == mid.y) && (toL.z == mid.z));
            }
        }
    }

    // Direction of sample j. Everything up to the rotation matrix depends on j
    // only through toL, which normalise() keeps re-normalising in place
    // (kernel.cu:1465-1466). Once two more normalisations leave toL
    // bit-identical, every later iteration reproduces the same values, so the
    // block is skipped (70 % of lanes are stable after j = 0, 99.6 % after j = 1).
    __device__ __forceinline__ V3 direction(AuxPtr ax, bool force_slow, const RtLightDev &L,
                                            V3 start, int j, const double *atab = nullptr)
    {
        const V3 lpos{L.px, L.py, L.pz};
        if (!stable || force_slow) {
            const V3 tin = toL;
            // P = cross(toL, (0,1,0)), kernel.cu:1444
            const V3 P{toL.y * 0.f - toL.z * 1.f, toL.z * 0.f - toL.x * 0.f, toL.x * 1.f - toL.y * 0.f};
            V3 e0{(lpos.x + P.x * L.size) - start.x, (lpos.y + P.y * L.size) - start.y,
                  (lpos.z + P.z * L.size) - start.z};
            const V3 toEdge = normalise_t<LEAN>(e0);                       // kernel.cu:1450
            angle = rtm::cosf_rt(dot3(toL, toEdge) * 2.f);                 // kernel.cu:1451
            // The rest depends on toL only through its next two in-place normalisations. Once
            // normalise() maps toL to itself (`fixed1`, 96 % of the lanes that are not `stable`
            // after the first iteration) those leave it -- and the axis, nAngle and the matrix
            // -- as they are: this iteration only had a new `angle` to compute, the next ones
            // repeat it.
            if (!fixed1 || force_slow) {
                // axis = normalise(cross((0,0,1), normalise(toL))), kernel.cu:1465
                const V3 n1 = normalise_t<LEAN>(toL);
                const V3 mid = toL;
                V3 ax0{0.f * n1.z - 1.f * n1.y, 1.f * n1.x - 0.f * n1.z, 0.f * n1.y - 0.f * n1.x};
                const V3 axis = normalise_t<LEAN>(ax0);
                // nAngle = acosf(dot(normalise(toL), (0,0,1))), kernel.cu:1466
                const V3 n2 = normalise_t<LEAN>(toL);
                const float nAngle = rtm::acosf_rt((n2.x * 0.f + n2.y * 0.f) + n2.z * 1.f, atab);
                float sn, cs;
                rtm::sincosf_rt(nAngle, sn, cs);
                const float omc = 1.f - cs;
                // rotate(nAngle, axis), kernel.cu:1267-1277 (non-standard on purpose)
                m00 = cs + axis.x * axis.x;
                m01 = axis.x * axis.y * omc - axis.z * sn;
                m02 = axis.x * axis.z * omc - axis.y * sn;
                m10 = axis.y * axis.x * omc + axis.z * sn;
                m11 = cs + axis.y * axis.y * omc;
                m12 = axis.y * axis.z * omc - axis.x * sn;
                m20 = axis.z * axis.x * omc - axis.y * sn;
                m21 = axis.z * axis.y * omc + axis.x * sn;
                m22 = cs + axis.z * axis.z * omc;
                fixed1 = (toL.x == mid.x) && (toL.y == mid.y) && (toL.z == mid.z) &&
                         (n1.x == n2.x) && (n1.y == n2.y) && (n1.z == n2.z);
                stable = (toL.x == tin.x) && (toL.y == tin.y) && (toL.z == tin.z);
            } else {
                stable = true;
            }
        }
        const float z = ax->jf[j] * (1.0f - angle) + angle;                // kernel.cu:1453
        const float zz = 1.f - z * z;
        float sq;                                                          // kernel.cu:1462-1463
        if (LEAN && __builtin_expect(zz >= 0x1.0p-96f, 1)) sq = lean_sqrt(zz);   // zz <= 1
        else sq = __builtin_sqrtf(zz);
        const float x = sq * ax->jcos[j];
        const float y = sq * ax->jsin[j];
        // multiply(rot, {x,y,z}), kernel.cu:123-125
        const V3 rv{(x * m00 + y * m10) + z * m20, (x * m01 + y * m11) + z * m21,
                    (x * m02 + y * m12) + z * m22};
        V3 nd{lpos.x - rv.x, lpos.y - rv.y, lpos.z - rv.z};
        return normalise_t<LEAN>(nd);                                      // kernel.cu:1468
    }
};

// One shadow ray against one table entry: updates `shadowed` (any-hit,
// kernel.cu:1504-1508). Two shortcuts decide most entries without sqrt/div:
// h < -h_sure puts t above RT_T_MIN (-B >= 2*h_sure and sqrt(disc) >= 0), and
// "behind" (see RT_BEHIND_FACTOR) makes t strictly negative.
template <int LEAN = 0>
__device__ __forceinline__ void shadow_test(const RayK &sr, float4 s, bool &shadowed, bool force_slow)
{
    const Quad q = (LEAN == 2) ? quadratic_fast(sr, s) : quadratic(sr, s);
    bool need;
    if (force_slow) {
        need = !shadowed;
    } else {
        const bool cand = (q.disc >= 0.f) && !shadowed;
        // h < -2e-4*A puts t above RT_T_MIN (-B >= 4e-4*A, sqrt(disc) >= 0, divided by 2A); 5e-5 * (4A) is that
        // bound up to a rounding the strict margins of the argument swallow many times over
        const bool sure = cand && (q.h < -5.0e-5f * sr.a4);
        const bool behind = (q.h > 0.f) && (q.disc < q.BB * RT_BEHIND_FACTOR);
        shadowed = shadowed || sure;
        need = cand && !sure && !behind;
    }
    if (__any(need)) {
        if (need) {
            float t;
            if (intersect_tail_t<LEAN>(sr, q, t)) shadowed = true;
        }
    }
}

// ---------------------------------------------------------------------------
// the frame kernel
// ---------------------------------------------------------------------------
// MODE: 0 = product kernel, 1 = work counters, 2 = every exactness-preserving shortcut off
// (rt_launch_opts.force_slow_path: tests), 3 = per-phase cycle stamps (s_memtime; RT_TUNING
// builds only, run time never quoted).
// FEAT: 0 = spheres only (the reference's default scene and every BASELINE config), 1 = with
// the cube / plane branches of castRay and castLightRay, 2 = with those and the triangle mesh.
// Each is its own instantiation so that the sphere-only kernel carries neither the code nor
// the live scalars of primitives that are not there.
// MULTI: the launch may take several samples per pixel (rt_launch_opts.spp > 1). The
// one-sample kernel has no sample loop: 74 -> 22 spilled scalars and 15 fewer vector registers.
template <int TW, bool CULL, int MODE, bool TABLDS, int FEAT = 0, bool MULTI = true>
__global__ __launch_bounds__(64 * (TABLDS ? RT_WAVES_PER_WG : 1), (FEAT == 2) ? (MODE == 1 ? 3 : (MULTI || TABLDS) ? RT_MIN_WAVES_MESH_MULTI : RT_MIN_WAVES_MESH) : (MODE == 1) ? 4 : (!MULTI && !TABLDS) ? RT_MIN_WAVES_ONE_SAMPLE : RT_MIN_WAVES_PER_SIMD) void rt_trace_tiles(const RtFrameConsts fc,
                                                                     const float4 *__restrict__ spheres)
{
    constexpr int STATS = (MODE == 1) ? 1 : (MODE == 3) ? 2 : 0;
    constexpr bool force_slow = (MODE == 2);
    constexpr bool FAST = (MODE == 4);    // rt_launch_opts.fast: approximate arithmetic, never the default
    constexpr bool MESH = (FEAT == 2);
    constexpr bool PRIMS = (FEAT >= 1);   // cubes and planes may be present
    // the culling kernels take every exactness-preserving shortcut (lean normalise/sqrt, fast
    // texel index); the brute-force and force-slow ones evaluate everything the long way
#ifdef RT_NO_LEAN
    constexpr int LEAN = 0;
#else
    constexpr int LEAN = (CULL && !force_slow) ? 1 : 0;   // precision class of normalise_t & co. (FAST: primary rays, normal and toL stay exact)
#endif
    // the lean tail of intersect(): for the primary rays; for the shadow rays it costs the one-sample
    // kernel two spilled registers at its 6-waves-per-SIMD budget (measured: profiles/r02_variants.txt)
#ifdef RT_NO_LEAN_PRIMARY_TAIL
    constexpr int LEAN_PRIMARY_TAIL = 0;
#else
    constexpr int LEAN_PRIMARY_TAIL = LEAN;
#endif
#ifdef RT_LEAN_SHADOW_TAIL
    constexpr int LEAN_SHADOW_TAIL = LEAN;
#else
    constexpr int LEAN_SHADOW_TAIL = FAST ? 2 : 0;
#endif
    constexpr int TH = 64 / TW;
    // Waves per workgroup: RT_WAVES_PER_WG share one staged table (TABLDS); with the table left
    // in global memory nothing is shared, and one-wave workgroups fill the SIMDs best
    // (0.68 vs 0.71 ms at C3) and need no barrier.
    constexpr int WPW = TABLDS ? RT_WAVES_PER_WG : 1;
    constexpr int WGX = (TW <= 16 && WPW >= 2) ? 2 : 1;   // wave tiles per workgroup in x
    extern __shared__ float4 lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = fc.n_spheres;
    const int n_pad = (n + 63) & ~63;
    const AuxPtr ax = (AuxPtr)(uintptr_t)fc.aux;   // lights, sample constants, sky, planes/cubes, mesh (device memory)

    // ---- stage the sphere table into LDS (coalesced 16 B/lane) ----
    float4 *tab = lds;
    if constexpr (TABLDS) {
        // the culling kernels stage the Morton-ordered copy (whole blocks, n_pad entries);
        // the brute-force kernels stage the list as it is
        const float4 *src = CULL ? reinterpret_cast<const float4 *>(fc.sorted) : spheres;
        for (int i = tid; i < (CULL ? n_pad : n); i += 64 * WPW) tab[i] = src[i];
        __syncthreads();
    }
    float4 *mylist = lds + (TABLDS ? n_pad : 0) + wave * RT_LIST_CAP;
    int *mykeys = reinterpret_cast<int *>(lds + (TABLDS ? n_pad : 0) + WPW * RT_LIST_CAP) + wave * RT_LIST_CAP;
    // b after n float+=double steps of 0.1 (brightness_steps), one 16-entry copy per wave: a
    // per-lane n then costs one LDS read instead of a ten-deep select chain per light
    float *mybtab = reinterpret_cast<float *>(lds + (TABLDS ? n_pad : 0) + WPW * RT_LIST_CAP) +
                    WPW * RT_LIST_CAP + wave * 16;
    // marked blocks of one culling pass (at most 64 at a time)
    int *myblks = reinterpret_cast<int *>(lds + (TABLDS ? n_pad : 0) + WPW * RT_LIST_CAP) +
                  WPW * (RT_LIST_CAP + 16) + wave * 64;
    // rtm::atan_eighth(0..8) for the binary64 arctangent (one LDS read instead of a select chain)
    double *myatan = reinterpret_cast<double *>(reinterpret_cast<int *>(lds + (TABLDS ? n_pad : 0) + WPW * RT_LIST_CAP) +
                                                WPW * (RT_LIST_CAP + 16 + 64)) + wave * 16;
    int *myboxes = reinterpret_cast<int *>(lds + (TABLDS ? n_pad : 0) + WPW * RT_LIST_CAP) +
                   WPW * (RT_LIST_CAP + 16 + 64 + 32) + wave * (RT_BOX_CAP + 128);   // + marked leaf blocks + 64 staged floats
    float *mytri = reinterpret_cast<float *>(myboxes + RT_BOX_CAP + 64);   // staged vertices of a leaf (MESH launches only)
    if (lane < 16) {
        mybtab[lane] = kBrightnessSteps[lane];   // same values as brightness_steps()
        myatan[lane] = kAtanEighth[lane];
    }
    wave_lds_sync();

    // which tile: the workgroup's own, or (one-wave workgroups) the one the frame's tile order puts at this place
    unsigned blk_x = blockIdx.x, blk_y = blockIdx.y;
    unsigned t_start = 0;
    const unsigned tiles_x = (unsigned)(fc.width + TW - 1) / (unsigned)TW;   // = gridDim.x of a one-wave-workgroup launch
    if (!TABLDS) {
        if (fc.tile_perm) {
            const unsigned p = fc.tile_perm[blockIdx.y * tiles_x + blockIdx.x];
            blk_x = p & 0xffffu;
            blk_y = p >> 16;
        }
        if (fc.tile_cost) t_start = (unsigned)__builtin_amdgcn_s_memtime();
    }
    const int tile_x = (blk_x * WGX + (wave % WGX)) * TW;
    // local row -> global row: a contiguous band, or row blocks dealt round-robin
    // to the ranks of a multi-GPU frame (il_rows is a multiple of the tile rows a
    // workgroup covers, so a tile never straddles two blocks)
    const int ly = (blk_y * (WPW / WGX) + (wave / WGX)) * TH + (lane / TW);
    const int px = tile_x + (lane % TW);
    const int py = (fc.il_count > 1) ? ((ly / fc.il_rows) * fc.il_count + fc.il_index) * fc.il_rows + (ly % fc.il_rows)
                                     : fc.y0 + ly;
    const bool valid = (px < fc.width) && (ly < fc.local_rows) && (py < fc.y1);
    const unsigned out_idx = (unsigned)ly * (unsigned)fc.width + (unsigned)px;   // band-local pixel index
    if (!__any(valid)) return;   // wave-uniform; after the only workgroup barrier

    unsigned long long st_primary = 0, st_shadow = 0, st_cull = 0, st_slots = 0, st_entries = 0,
                       st_overflow = 0, st_hits = 0, st_unshadowed = 0, st_clusters = 0;

    unsigned long long hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_walks = 0, st_walks_no_penumbra = 0, st_pen_lanes = 0, st_walk_lanes = 0;
    unsigned long long st_walks_all_dark = 0, st_walks_all_lit = 0;
    unsigned long long sm_plisted = 0, sm_pleaf = 0, sm_ptri = 0, sm_slisted = 0, sm_sslab = 0, sm_stri = 0;   // mesh work (STATS, MESH)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    const bool span_only = RT_ABL(512);   // no inner stamps: near-real wave durations
    const unsigned long long rt_begin = (STATS == 2) ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz
    auto phase = [&](int k, bool last = false) {   // charge the cycles since the previous stamp to phase k
        if (STATS == 2 && (k < 0 || last || !span_only)) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): s_memtime returns through it
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 0) ph[k] += now - t_prev;
            t_prev = now;
        }
    };
    phase(-1);

    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f;

    const int n_samples = MULTI ? fc.spp : 1;
    for (int sample = 0; sample < n_samples; ++sample) {
        // ================= primary ray, kernel.cu:1624-1631 =================
        // dx, dy of kernel.cu:1624-1625 (binary64 expressions of the column resp. the row) come
        // from the frame's tables, evaluated by the host with the reference's operations; lanes
        // beyond the frame edge read the last column / row (their pixels are never stored)
        const int sidx = fc.sample_base + sample;
        const float dx = fc.dx_tab[sidx * fc.width + (px < fc.width ? px : fc.width - 1)];
        const float dy = fc.dy_tab[sidx * fc.height + (py < fc.height ? py : fc.height - 1)];
        V3 dir{dx, dy, fc.eye_nz};   // (dx,dy,0) - (0,0,-1/aspect)
        normalise_t<LEAN>(dir);
        // camera::rotateDir, kernel.cu:252-257 (cos/sin hoisted to the host)
        V3 D;
        {
            const float y = dir.y * fc.cos_pitch - dir.z * fc.sin_pitch;
            float z = dir.y * fc.sin_pitch + dir.z * fc.cos_pitch;
            const float x = dir.x * fc.cos_yaw + z * fc.sin_yaw;
            z = -dir.x * fc.sin_yaw + z * fc.cos_yaw;
            D = V3{x, y, z};
        }
        const V3 O{fc.org_x, fc.org_y, fc.org_z};
        const RayK pr = make_ray(O, D);

        phase(0);
        // ================= castRay, sphere branch =================
        bool p_use_list = false;   // false: walk the whole table
        int pcount = n;
        bool pb_use_list = false;  // leaf boxes of the mesh: false = all of them
        int pbcount = MESH ? fc.n_boxes : 0;
        Beam pbeam;                // the tile's primary beam, kept for the per-triangle cull of the mesh leaves
        bool pbeam_ok = false;
        if (CULL) {
            // cone around the tile's mean direction, apex at the (shared) origin
            float sx = D.x, sy = D.y, sz = D.z;
            wave_sum3(sx, sy, sz);
            const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(sx, sx, __builtin_fmaf(sy, sy, sz * sz)));
            Beam b;
            b.ux = uniform(sx * inv); b.uy = uniform(sy * inv); b.uz = uniform(sz * inv);
            const float cx = D.y * b.uz - D.z * b.uy, cy = D.z * b.ux - D.x * b.uz, cz = D.x * b.uy - D.y * b.ux;
            float s2 = wave_max(__builtin_fmaf(cx, cx, __builtin_fmaf(cy, cy, cz * cz)));
            s2 = uniform(s2);
            // s2 is sin^2 of the largest deviation (|D| = 1 up to rounding)
            const bool ok = (s2 < 0.25f);   // NaN or a degenerate tile: do not cull
            const float sn = __builtin_amdgcn_sqrtf(s2) * 1.01f + 1.0e-5f;
            b.k = sn * __builtin_amdgcn_rsqf(1.f - sn * sn);
            b.ax = O.x; b.ay = O.y; b.az = O.z;
            b.smin = 0.f;
            b.smax = 0.f;
            b.r0 = 1.0e-4f;
            if (ok) {
                // eye cones when the scene has them and the tile's beam is within their slope limit
                const float4 *csorted = reinterpret_cast<const float4 *>(fc.csorted);
                const int c = (csorted && b.k <= fc.cone_kcap)
                                  ? build_list2<STATS, TABLDS, false, true, 2>(tab, fc, n, mylist, mykeys, myblks, b, lane, st_cull,
                                                                              csorted, reinterpret_cast<const float4 *>(fc.cblocks),
                                                                              fc.corig)
                                  : build_list2<STATS, TABLDS, false, true>(tab, fc, n, mylist, mykeys, myblks, b, lane, st_cull);
                if (c <= RT_LIST_CAP) {
                    p_use_list = true;
                    pcount = c;
                } else if (STATS == 1) {
                    st_overflow += 1;
                }
                if (STATS == 1) st_entries += (unsigned long long)(c <= RT_LIST_CAP ? c : n);
                if (MESH) {
                    const int cb = build_box_list(reinterpret_cast<const float4 *>(ax->box_spheres), fc.n_boxes, myboxes, myboxes + RT_BOX_CAP, b, lane);
                    if (cb <= RT_BOX_CAP) {
                        pb_use_list = true;
                        pbcount = cb;
                    }
                    pbeam = b;
                    pbeam_ok = !force_slow;
                    if (STATS == 1) sm_plisted += (unsigned long long)pbcount;
                }
            }
        }

        phase(1);
        float nt = __builtin_inff();
        float hcx = 0.f, hcy = 0.f, hcz = 0.f;   // centre of the closest sphere
        int hkind = 1;                           // 0 triangle, 1 sphere, 2 plane, 3 cube (kernel.cu:1376)
        int htri = 0;
        if (MESH) {
            // triangles through the flat list of leaf boxes, kernel.cu:1293-1328 (before
            // the spheres, as there): a lane tests a leaf's triangles iff its ray hits the box
            const V3 inv{1.f / D.x, 1.f / D.y, 1.f / D.z};
            for (int jj = 0; jj < (RT_ABL(2048) ? 0 : pbcount); ++jj) {
                const int j = pb_use_list ? myboxes[jj] : jj;
                const RtBoxDev bx = ax->boxes[j];
                const bool bh = box_intersect(bx, O, inv);
                if (__any(bh) && !RT_ABL(1024)) {
                    if (STATS == 1) sm_pleaf += 1;
                    // which of the leaf's triangles the tile's beam can touch at all (a leaf of the reference's
                    // ten-pass split holds triangles far larger than a tile: about a third survive), 63 triangles
                    // -- nine loads of seven -- at a time: the split leaves a few leaves of a hundred and more
                    for (int c0 = 0; c0 < bx.len; c0 += 63) {
                        const int clen = bx.len - c0 < 63 ? bx.len - c0 : 63;
                        unsigned long long tmask = ~0ull;
                        if (CULL && pbeam_ok) {
                            const int ti = bx.start + c0 + (lane < clen ? lane : 0);
                            const float4 ts = reinterpret_cast<const float4 *>(ax->tri_bs)[ti];
                            const float4 tn = reinterpret_cast<const float4 *>(ax->tri_nrm)[ti];
                            tmask = __ballot(lane < clen && beam_keeps_triangle(pbeam, ts, tn));
                            if (tmask == 0) continue;
                        }
                        // the leaf's vertices, seven triangles (63 floats) per coalesced load, staged in LDS
                        // and broadcast from there: one memory round trip per seven triangles instead of
                        // two dependent scalar loads per triangle
                        for (int base = 0; base < clen; base += 7) {
                            const int cnt = clen - base < 7 ? clen - base : 7;
                            if (((tmask >> base) & 0x7full) == 0) continue;
                            mytri[lane] = ax->tri9[(size_t)(bx.start + c0 + base) * 9 + lane];   // the array is padded by 64 floats
                            wave_lds_sync();
                            for (int i = 0; i < cnt; ++i) {
                                if (!((tmask >> (base + i)) & 1ull)) continue;
                                const float *tv = mytri + 9 * i;
                                float t, u, v;
                                if (STATS == 1) sm_ptri += 1;
                                if (bh && tri_intersect(O, D, tv, tv + 3, tv + 6, t, u, v) && t < nt) {
                                    nt = t;
                                    htri = bx.start + c0 + base + i;   // position in tri_idx; resolved when shading (u, v too)
                                    hkind = 0;
                                }
                            }
                            wave_lds_sync();
                        }
                    }
                }
            }
        }
        // without a list the table is walked in list order: from global memory in the culling
        // kernels (their LDS copy is Morton-ordered), from LDS in the brute-force ones
        auto primary_entry = [&](int e) -> float4 {
            if (CULL) return p_use_list ? mylist[e] : spheres[e];
            return entry_at<TABLDS>(false, mylist, tab, spheres, e);
        };
        // A culled list comes front to back (build_list2): entry e carries its list position in
        // mykeys[e] and a lower bound of its t in the block-list slot e. `holder` is the list
        // position of the sphere that currently holds nt (-1: none, e.g. a triangle does), so
        // that "first index wins ties" (kernel.cu:1335) holds in any order of evaluation.
        const float *plbs = reinterpret_cast<const float *>(myblks);
        const bool p_front_to_back = CULL && p_use_list && pcount > 1;
        int holder = -1;
        float4 pcur = pcount > 0 ? primary_entry(0) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = 0; e < pcount; ++e) {
            const float4 s = pcur;
            const int pos = (CULL && p_use_list) ? mykeys[e] : e;
            pcur = primary_entry(e + 1 < pcount ? e + 1 : e);   // one entry in flight
            const Quad q = quadratic(pr, s);
            bool need = (q.disc >= 0.f);
            if (!force_slow) need = need && !(q.h > 0.f && q.disc < q.BB * RT_BEHIND_FACTOR);
            if (force_slow) need = true;
            if (__any(need)) {
                if (need) {
                    float t;
                    if (intersect_tail_t<LEAN_PRIMARY_TAIL>(pr, q, t)) {
                        // strict, and among equal t the lower list position: first index wins ties
                        if (t < nt || (t == nt && holder >= 0 && pos < holder)) {
                            nt = t;
                            holder = pos;
                            hcx = s.x; hcy = s.y; hcz = s.z;
                            if (MESH) hkind = 1;
                        }
                    }
                }
            }
            if (STATS == 1) { st_primary += __popcll(__ballot(valid)); st_slots += 64; }
            // every remaining entry returns t >= its bound >= the next entry's bound
            if (p_front_to_back && e + 1 < pcount && !force_slow) {
                const float lb_next = plbs[e + 1];
                if (__all(!valid || nt < lb_next)) break;
            }
        }
        if (CULL) wave_lds_sync();   // the list is rebuilt below
        // cubes (kernel.cu:1344-1356) then planes (:1359-1372): few, tested exhaustively;
        // for a plane hit hc* carries the plane's normal instead of a centre
        if (PRIMS && fc.n_cubes > 0) {
            const V3 inv{1.f / D.x, 1.f / D.y, 1.f / D.z};
            for (int i = 0; i < fc.n_cubes; ++i) {
                const RtCubeDev c = ax->cubes[i];
                float t;
                if (cube_intersect(c, O, inv, t) && t < nt) {
                    nt = t;
                    hkind = 3;
                    hcx = c.cx; hcy = c.cy; hcz = c.cz;
                }
            }
        }
        for (int i = 0; i < (PRIMS ? fc.n_planes : 0); ++i) {
            const RtPlaneDev p = ax->planes[i];
            float t;
            if (plane_intersect(p, O, D, t) && t < nt) {
                nt = t;
                hkind = 2;
                hcx = p.nx; hcy = p.ny; hcz = p.nz;
            }
        }
        phase(2);

        const bool hit = valid && (nt != __builtin_inff());   // kernel.cu:1374

        // ================= miss: skybox::getFColor, kernel.cu:1147-1166 =================
        int sky_idx = -1;
        if (valid && !hit) {
            const float4 sk = make_float4(ax->sky_cx, ax->sky_cy, ax->sky_cz, ax->sky_r2);
            const Quad q = quadratic(pr, sk);
            float t;
            intersect_tail_t<LEAN_PRIMARY_TAIL>(pr, q, t);   // the boolean is ignored there, t is used as left
            const V3 hp{O.x + D.x * t, O.y + D.y * t, O.z + D.z * t};
            V3 nrm{hp.x - sk.x, hp.y - sk.y, hp.z - sk.z};
            normalise_t<LEAN>(nrm);
            const int sky_w = ax->sky_w, sky_h = ax->sky_h;
            int ix, iy;
            int sky_fast = -1;
            if (FAST) {
                float ux, uy;
                approx_sphere_uv(nrm, ux, uy);
                ix = f2i(ux * (float)sky_w);
                iy = f2i(uy * (float)sky_h);
            } else {
                if (LEAN) {
                    // the same certainty test as for the object texture (approx_sphere_uv / sure_texel above); here
                    // the exact expressions are binary32 chains of their own -- atan2f and acosf rounded to float, a
                    // float division, 1.f + ..., two products -- which stay within 1.7e-7 of the real value, so the
                    // margin is size * (1e-6 + 2^-22): 5e-7 approximation budget + that + both product roundings
                    float ux, uy;
                    approx_sphere_uv(nrm, ux, uy);
                    sky_fast = sure_texel(ux, uy, sky_w, sky_h, ax->sky_mu_x, ax->sky_mu_y);
                }
                ix = iy = 0;
                if (__builtin_expect(sky_fast < 0, 0)) {
                    ix = f2i((1.f + rtm::atan2f_rt(nrm.z, nrm.x, myatan) / 3.1415f) * 0.5f * (float)sky_w);
                    iy = f2i(rtm::acosf_rt(nrm.y, myatan) / 3.1415f * (float)sky_h);
                }
            }
            int idx = sky_fast >= 0 ? sky_fast : iy * sky_w + ix;
            const int last = sky_w * sky_h - 1;
            idx = idx < 0 ? 0 : (idx > last ? last : idx);   // documented clamp (reference is UB there)
            sky_idx = idx;
        }

        // ================= hit: shade, kernel.cu:1396-1405, 1643-1677 =================
        V3 start{0.f, 0.f, 0.f}, normal{0.f, 1.f, 0.f};
        float tr = 0.f, tg = 0.f, tb = 0.f;
        if (hit) {
            // the texture's uniforms: read from the kernel-argument segment here (see the write-back), not carried through
            // the primary cull and tests
            FcPtr kt = (FcPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kt));
            const int t_w = kt->tex_w, t_h = kt->tex_h;
            const V3 new_org{O.x + D.x * nt, O.y + D.y * nt, O.z + D.z * nt};
            float tx = 0.5f, ty = 0.5f;   // plane, kernel.cu:1413-1414
            int ci_fast = -1;             // texel index already known for sure (sphere / cube hits of the culling kernels)
            V3 hp = new_org;              // what start_O is offset from
            if (MESH && hkind == 0) {     // triangle, kernel.cu:1378-1393
                const RtTriDev *tp = ax->tris + ax->tri_idx[htri];
                // the barycentrics of the winning triangle: the same operations on the same operands as
                // in the loop above (which does not carry them along)
                float hnt, hnu = 0.f, hnv = 0.f;
                (void)tri_intersect(O, D, tp->p0, tp->p1, tp->p2, hnt, hnu, hnv);
                const float w0 = 1 - hnu - hnv;
                if (fc.flags & RT_FLAG_MESH_NORMALS) {
                    normal = V3{(tp->vn[0] * w0 + tp->vn[3] * hnu) + tp->vn[6] * hnv,
                                (tp->vn[1] * w0 + tp->vn[4] * hnu) + tp->vn[7] * hnv,
                                (tp->vn[2] * w0 + tp->vn[5] * hnu) + tp->vn[8] * hnv};
                    normalise_t<LEAN>(normal);
                } else {
                    normal = V3{tp->n[0], tp->n[1], tp->n[2]};
                }
                tx = (w0 * tp->vt[0]) + (hnu * tp->vt[2]) + (hnv * tp->vt[4]);
                ty = (w0 * tp->vt[1]) + (hnu * tp->vt[3]) + (hnv * tp->vt[5]);
                // new_org = add(normal, add(Org, Dir * nt)): displaced by the whole normal
                hp = V3{normal.x + new_org.x, normal.y + new_org.y, normal.z + new_org.z};
                hcx = hcy = hcz = 0.f;    // one group for all triangle hits of the tile
            } else if (PRIMS && hkind == 2) {      // plane, kernel.cu:1407-1416: the normal as stored
                normal = V3{hcx, hcy, hcz};
            } else {                      // sphere / cube, kernel.cu:1396-1405, 1418-1425
                normal = V3{new_org.x - hcx, new_org.y - hcy, new_org.z - hcz};
                normalise_t<LEAN>(normal);
                if (RT_ABL(8)) {
                    tx = normal.x; ty = normal.y;
                } else if (FAST) {
                    approx_sphere_uv(normal, tx, ty);   // no certainty test: a texel now and then is the neighbour
                } else if (LEAN && !RT_ABL(4096)) {
                    float ux, uy;
                    approx_sphere_uv(normal, ux, uy);
                    ci_fast = sure_texel(ux, uy, t_w, t_h, kt->tex_mu_x, kt->tex_mu_y);
                    if (__builtin_expect(ci_fast < 0, 0)) {
                        tx = (float)((1.0 + rtm::div_by_3p1415((double)rtm::atan2f_rt(normal.z, normal.x, myatan))) * 0.5);
                        ty = (float)rtm::div_by_3p1415((double)rtm::acosf_rt(normal.y, myatan));
                    }
                } else {
                    // the literals 1, 3.1415, 0.5 make these binary64 expressions
                    // kernel.cu:1402-1403; "/ 3.1415" in binary64 through div_by_3p1415 (same bits)
                    tx = (float)((1.0 + rtm::div_by_3p1415((double)rtm::atan2f_rt(normal.z, normal.x, myatan))) * 0.5);
                    ty = (float)rtm::div_by_3p1415((double)rtm::acosf_rt(normal.y, myatan));
                }
            }
            int ci = f2i(ty * (float)t_h) * t_w + f2i(tx * (float)t_w);
            if (ci_fast >= 0) ci = ci_fast;
            const int last = t_w * t_h - 1;
            ci = ci < 0 ? 0 : (ci > last ? last : ci);       // documented clamp
            tr = kt->tex_r[ci];
            tg = kt->tex_g[ci];
            tb = kt->tex_b[ci];
            // start_O = normal * 0.00001 + new_org, kernel.cu:1647
            start = V3{normal.x * 0.00001f + hp.x, normal.y * 0.00001f + hp.y, normal.z * 0.00001f + hp.z};
            if (STATS == 1) st_hits += 1;
        }

        phase(3);
        const bool tex_fin = (__builtin_fabsf(tr) < __builtin_inff()) & (__builtin_fabsf(tg) < __builtin_inff()) &
                             (__builtin_fabsf(tb) < __builtin_inff());
        float fr = 0.f, fg = 0.f, fb = 0.f;   // this sample's colour (sky texel on a miss)
        if (sky_idx >= 0) {
            fr = ax->sky_r[sky_idx];
            fg = ax->sky_g[sky_idx];
            fb = ax->sky_b[sky_idx];
        }
        if (__any(hit) && !RT_ABL(16)) {
            // A tile that straddles a silhouette sees several spheres at different
            // depths; one beam around all of their shadow rays would be fat and its
            // survivor list long. So the hit lanes are processed in groups that share
            // the closest sphere (3.5 % of the 8x8 tiles at 4K see more than one):
            // each group's origins lie on one small patch, its beam is thin, its
            // list short -- and no wave runs orders of magnitude longer than the rest.
            // group ids first (the sphere centres / kinds they are derived from are
            // not needed afterwards, which frees four registers for the light loop)
            int gid = -1, n_groups = 0;
            {
                unsigned long long rem = __ballot(hit);
                while (rem) {
                    const int first = __builtin_ctzll(rem);
                    const float gx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hcx), first));
                    const float gy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hcy), first));
                    const float gz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hcz), first));
                    const int gk = __builtin_amdgcn_readlane(hkind, first);
                    const bool in = hit && ((rem >> lane) & 1ull) &&
                                    (lane == first || (hkind == gk && hcx == gx && hcy == gy && hcz == gz));
                    if (in) gid = n_groups;
                    rem &= ~__ballot(in);
                    ++n_groups;
                }
            }
            for (int g = 0; g < n_groups; ++g) {
            const bool inc = hit && (gid == g);
            if (STATS == 1) st_clusters += 1;
            // The group's ray origins, bounded once for all lights: a ball around the first
            // member's `start`. (Per-light axial and perpendicular extents would be a little
            // tighter, for three more wave reductions per light; the patch a tile sees of one
            // primitive is small against the spheres it is culled against.)
            float g_ax = 0.f, g_ay = 0.f, g_az = 0.f, g_r2 = 0.f;
            if (CULL) {
                const int glead = __builtin_ctzll(__ballot(inc));
                g_ax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, start.x), glead));
                g_ay = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, start.y), glead));
                g_az = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, start.z), glead));
                const float ox = start.x - g_ax, oy = start.y - g_ay, oz = start.z - g_az;
                g_r2 = uniform(wave_max(inc ? __builtin_fmaf(ox, ox, __builtin_fmaf(oy, oy, oz * oz)) : 0.f));
            }
            for (int li = 0; li < fc.n_lights; ++li) {
                const RtLightDev L = ax->lights[li];
                const V3 lpos{L.px, L.py, L.pz};

                // toL = normalise(l.pos - start), kernel.cu:1438 -- here only to ~1e-6 (fast
                // reciprocal square root): it feeds the facing test and the beam bound, both of
                // which carry margins; the exact value is formed by ShadowChain::begin when the
                // samples are actually constructed.
                V3 toL{lpos.x - start.x, lpos.y - start.y, lpos.z - start.z};
                {
                    const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(toL.x, toL.x, __builtin_fmaf(toL.y, toL.y, toL.z * toL.z)));
                    toL.x *= inv; toL.y *= inv; toL.z *= inv;
                }

                // A surface that faces away from the light gets b *= 0 at kernel.cu:1542
                // whatever its ten shadow samples say, and fr + (0*l.r)*r leaves fr
                // untouched. The reference's `a` uses toL after its in-place
                // re-normalisations, which move it by an ulp or two, so only a clearly
                // negative normal.toL (and finite factors, so that 0*x is 0) counts.
                // Such lanes take no part in this light; if the whole group faces away
                // the light is skipped. (Not applied in the brute-force build, which
                // runs the reference's loops as written and is what tests compare with.)
                bool lit = inc;
                bool zero_ok = false;   // brightness 0 adds exactly nothing for this lane
                if (CULL && !force_slow) {
                    const float a0 = dot3(normal, toL);
                    // (0 * l.r) * r is 0 iff the light's colour (frame constant, checked by the host) and the
                    // texel (checked once per pixel) are finite
                    zero_ok = tex_fin && (L.fin != 0.f) && (__builtin_fabsf(a0) < 1.0e30f);
                    const bool away = (a0 < -1.0e-4f) && zero_ok;
                    lit = inc && !away;
                    if (!__any(lit)) continue;
                }

                // ---------- conservative beam for this light's 10 x 64 rays ----------
                bool s_use_list = false;
                int scount = n;
                bool sb_use_list = false;
                int sbcount = MESH ? fc.n_boxes : 0;
                float beam_k = 0.f;   // slope of the light's beam (valid when s_use_list)
                if (CULL) {
                    bool ok = true;
                    Beam b;
                    b.ux = L.ux; b.uy = L.uy; b.uz = L.uz;
                    // approximate sample directions from toL (fast math; padded below)
                    float smax2 = 0.f;
                    if (RT_ABL(32)) smax2 = 0.01f;
                    else {
                        const float c = toL.z;
                        const float sn = __builtin_amdgcn_sqrtf(__builtin_fmaxf(1.f - c * c, 0.f));
                        const float q2 = toL.x * toL.x + toL.y * toL.y;
                        const float rq = q2 > 0.f ? __builtin_amdgcn_rsqf(q2) : 0.f;
                        const float ax = -toL.y * rq, ay = toL.x * rq;   // axis = (0,0,1) x toL, az = 0
                        const float omc = 1.f - c;
                        // rotate(nAngle, axis), kernel.cu:1267-1277, with az = 0
                        const float m00 = c + ax * ax, m01 = ax * ay * omc, m02 = -ay * sn;
                        const float m10 = m01, m11 = c + ay * ay * omc, m12 = -ax * sn;
                        const float m20 = -ay * sn, m21 = ax * sn, m22 = c;
                        // A sample direction is w_j = l.pos - r_j with r_j = x_j R0 + y_j R1 + z_j R2
                        // (rows of the matrix), (x_j, y_j) = sqrt(1 - z_j^2) (cos phi_j, sin phi_j) and
                        // |z_j| <= 1. Its deviation from u = l.pos/|l.pos|:
                        // sin = |w x u| / |w| = |r x u| / |w| (l.pos x u = 0), with
                        // r x u = x A + y B + z C, A = R0 x u, B = R1 x u, C = R2 x u, and
                        // |w| >= |l.pos| - ||M||_F. For ANY phi and any |z| <= 1:
                        // |x A + y B + z C| <= sqrt(1 - z^2) sigma + |z| |C| <= sqrt(sigma^2 + |C|^2),
                        // sigma^2 the larger eigenvalue of the Gram matrix [[A.A, A.B], [A.B, B.B]].
                        // (The ten samples themselves come within a few per cent of this bound; walking
                        // them cost 250 instructions per light.) A NaN or inf anywhere makes kmax2 a NaN,
                        // which switches culling off below.
                        // |Ri x u|^2 = |Ri|^2 - (Ri.u)^2 and (R0 x u).(R1 x u) = R0.R1 - (R0.u)(R1.u) for the unit
                        // u (Lagrange; |u|^2 is 1 to 2e-7, inside the padding): the Gram entries without
                        // forming the cross products, and ||M||_F^2 is the sum of the row norms. No clamping:
                        // a NaN (a light at the origin has a NaN axis) must reach kmax2.
                        const float ru0 = __builtin_fmaf(m00, b.ux, __builtin_fmaf(m01, b.uy, m02 * b.uz));
                        const float ru1 = __builtin_fmaf(m10, b.ux, __builtin_fmaf(m11, b.uy, m12 * b.uz));
                        const float ru2 = __builtin_fmaf(m20, b.ux, __builtin_fmaf(m21, b.uy, m22 * b.uz));
                        const float n0 = __builtin_fmaf(m00, m00, __builtin_fmaf(m01, m01, m02 * m02));
                        const float n1 = __builtin_fmaf(m10, m10, __builtin_fmaf(m11, m11, m12 * m12));
                        const float n2 = __builtin_fmaf(m20, m20, __builtin_fmaf(m21, m21, m22 * m22));
                        const float r01 = __builtin_fmaf(m00, m10, __builtin_fmaf(m01, m11, m02 * m12));
                        const float frob2 = n0 + n1 + n2;
                        const float den = L.pos_len - __builtin_amdgcn_sqrtf(frob2) * 1.001f;
                        const float gaa = __builtin_fmaf(-ru0, ru0, n0);
                        const float gbb = __builtin_fmaf(-ru1, ru1, n1);
                        const float gab = __builtin_fmaf(-ru0, ru1, r01);
                        const float gcc = __builtin_fmaf(-ru2, ru2, n2);
                        const float hd = 0.5f * (gaa - gbb);
                        const float kmax2 = (0.5f * (gaa + gbb) + __builtin_amdgcn_sqrtf(hd * hd + gab * gab)) * 1.001f + gcc;
                        // a light closer to the origin than the matrix can reach has no usable bound
                        const float rden = __builtin_amdgcn_rcpf(den);
                        smax2 = (den > 0.05f * L.pos_len) ? kmax2 * rden * rden * 1.0001f : __builtin_nanf("");
                    }
                    if (!lit || smax2 < 0.f) smax2 = 0.f;   // (the wave maximum compares bit patterns; a NaN passes through)
                    const bool lane_bad = lit && !(smax2 < 0.25f);
                    ok = !__any(lane_bad);
                    const float s2w = uniform(wave_max(smax2));
                    const float snw = __builtin_amdgcn_sqrtf(s2w) * 1.02f + 2.0e-3f;
                    b.k = snw * __builtin_amdgcn_rsqf(__builtin_fmaxf(1.f - snw * snw, 0.05f));
                    // origins: the group's ball (see above), centred on the axis
                    b.ax = g_ax; b.ay = g_ay; b.az = g_az;
                    ok = ok && (g_r2 < 1.0e30f);
                    b.r0 = __builtin_amdgcn_sqrtf(g_r2) * 1.001f + 1.0e-3f;
                    b.smin = -b.r0;
                    b.smax = b.r0;
                    phase(4);
                    if (RT_ABL(4)) { ok = false; scount = 0; }
                    if (ok) {
                        // One sphere in front of the whole beam shadows all 10 samples of
                        // every lit lane: unshadowed = 0, b = 0, and the light adds exactly
                        // nothing -- the sample construction and the tests are skipped.
                        const bool may_skip = !force_slow && !RT_ABL(64) && __all(!lit || zero_ok);
                        const float4 *lsorted = reinterpret_cast<const float4 *>(ax->lsorted[li]);
                        const int cb = lsorted ? build_list2<STATS, TABLDS, true, false, 1>(
                                                     tab, fc, n, mylist, mykeys, myblks, b, lane, st_cull, lsorted,
                                                     reinterpret_cast<const float4 *>(ax->lblocks[li]), nullptr, may_skip)
                                               : build_list2<STATS, TABLDS, true, false>(tab, fc, n, mylist, mykeys, myblks, b,
                                                                                         lane, st_cull, nullptr, nullptr, nullptr, may_skip);
                        const int c = cb & 0x3fffffff;
                        if (may_skip && (cb & 0x40000000)) {
                            if (STATS == 1) hist[7] += 1;
                            wave_lds_sync();
                            continue;
                        }
                        if (MESH && RT_ABL(32768)) {
                            sb_use_list = true;
                            sbcount = 0;
                        } else if (MESH) {
                            const int cbx = build_box_list(reinterpret_cast<const float4 *>(ax->box_spheres), fc.n_boxes, myboxes, myboxes + RT_BOX_CAP, b, lane);
                            if (cbx <= RT_BOX_CAP) {
                                sb_use_list = true;
                                sbcount = cbx;
                            }
                        }
                        if (c <= RT_LIST_CAP) {
                            s_use_list = true;
                            scount = c;
                            beam_k = b.k;
                        } else if (STATS == 1) {
                            st_overflow += 1;
                        }
                        if (STATS == 1) st_entries += (unsigned long long)(c <= RT_LIST_CAP ? c : n);
                        if (STATS == 1) {
                            const int bin = c <= 1 ? 0 : c <= 2 ? 1 : c <= 4 ? 2 : c <= 8 ? 3 : c <= 16 ? 4 : c <= RT_LIST_CAP ? 5 : 6;
                            hist[bin] += 1;
                        }
                    }
                }

                phase(5);
                // ---------- the 10 samples, kernel.cu:1442-1540 (exact) ----------
                ShadowChain<LEAN> chain;
                chain.begin(lpos, start);
                int unshadowed = 0;
                // A short list whose every sphere lies behind every ray of the beam (for
                // each lit lane: start outside the sphere by more than the `behind`
                // shortcut needs, C > 2e-5*|oc|^2, and the whole cone of directions on the
                // far side, cos(u,oc) > sin(u,oc)*tan(theta) + 0.02) leaves all ten samples
                // unshadowed without constructing one of them: only toL's evolution is
                // needed for kernel.cu:1541. Typically the list is just the sphere the
                // tile itself lies on.
                bool all_clear = false;
                if (CULL && (!MESH || (sb_use_list && sbcount == 0)) && s_use_list && scount <= 4 && !force_slow &&
                    !RT_ABL(128) && (!PRIMS || (fc.n_planes | fc.n_cubes) == 0)) {
                    bool clear = true;
                    for (int e = 0; e < scount; ++e) {
                        const float4 sp = mylist[e];
                        const float ocx = start.x - sp.x, ocy = start.y - sp.y, ocz = start.z - sp.z;
                        const float q = ocx * ocx + ocy * ocy + ocz * ocz;
                        const float C = q - sp.w;
                        const float cu = (ocx * L.ux + ocy * L.uy + ocz * L.uz) * __builtin_amdgcn_rsqf(q);
                        const float su = __builtin_amdgcn_sqrtf(__builtin_fmaxf(1.f - cu * cu, 0.f));
                        clear = clear && (C > 2.0e-5f * q) && (cu > su * beam_k + 0.02f);
                    }
                    all_clear = __all(!lit || clear);
                }
                if (all_clear) {
                    chain.settle();
                    unshadowed = RT_SHADOW_SAMPLES;
                    if (STATS == 1) hist[6] += 1;
                }
                // The samples will be walked: put the likeliest occluders first -- the spheres that
                // reach farthest across the beam's axis (distance from the axis minus radius) -- so
                // that the any-hit loops end sooner. An any-hit does not depend on the order.
                if (CULL && s_use_list && !all_clear && scount > 2 && scount <= 64 && !force_slow &&
                    !RT_ABL(256)) {
                    float *kbuf = reinterpret_cast<float *>(myblks);
                    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                    float key = 0.f;
                    int rank = 0;
                    if (lane < scount) {
                        e = mylist[lane];
                        const float vx = e.x - g_ax, vy = e.y - g_ay, vz = e.z - g_az;
                        const float vv = __builtin_fmaf(vx, vx, __builtin_fmaf(vy, vy, vz * vz));
                        const float sa = __builtin_fmaf(vx, L.ux, __builtin_fmaf(vy, L.uy, vz * L.uz));
                        key = __builtin_amdgcn_sqrtf(__builtin_fmaxf(__builtin_fmaf(-sa, sa, vv), 0.f)) - __builtin_amdgcn_sqrtf(e.w);
                        key = (key == key) ? key : __builtin_inff();
                        kbuf[lane] = key;
                    }
                    wave_lds_sync();
                    for (int jj = 0; jj < scount; ++jj) {
                        const float kj = kbuf[jj];
                        rank += (kj < key || (kj == key && jj < lane)) ? 1 : 0;
                    }
                    wave_lds_sync();
                    if (lane < scount) mylist[rank] = e;
                    wave_lds_sync();
                }
                if (FAST && !all_clear) {   // toL as the reference leaves it (exact), then the sample frame once, approximately
                    chain.settle();
                    chain.setup_fast(L, start);
                }
#pragma unroll 1
                for (int j = 0; j < (all_clear ? 0 : RT_SHADOW_SAMPLES); ++j) {
                    const V3 new_dir = RT_ABL(2) ? chain.toL
                                       : FAST   ? chain.direction_fast(ax, L, j)
                                                : chain.direction(ax, force_slow, L, start, j, myatan);
                    const RayK sr = make_ray(start, new_dir);
                    phase(6);
                    // any-hit over the list, kernel.cu:1501-1510
                    bool shadowed = !lit;   // lanes outside the group (or unlit) are simply done
                    const int scount_j = RT_ABL(1) ? 0 : scount;
                    if (scount_j > 0) {
                        const float4 *gtab = CULL ? reinterpret_cast<const float4 *>(fc.sorted) : spheres;
                        // (no entry kept in flight: at 7 waves per SIMD the LDS latency is covered by the other
                        // waves, and the four registers are what lets the kernel run at 7)
                        for (int e = 0; e < scount_j; ++e) {
                            const float4 cur = entry_at<TABLDS>(s_use_list, mylist, tab, gtab, e);
                            shadow_test<LEAN_SHADOW_TAIL>(sr, cur, shadowed, force_slow);
                            if (STATS == 1) { st_shadow += __popcll(__ballot(lit)); st_slots += 64; }
                            if (__all(shadowed)) break;
                        }
                    }
                    // triangles, kernel.cu:1475-1497 (the reference tests them first; an
                    // any-hit does not depend on the order)
                    if (MESH && !__all(shadowed) && !RT_ABL(8192)) {
                        const V3 inv{1.f / new_dir.x, 1.f / new_dir.y, 1.f / new_dir.z};
                        for (int bjj = 0; bjj < sbcount; ++bjj) {
                            const int bj = sb_use_list ? myboxes[bjj] : bjj;
                            const RtBoxDev bx = ax->boxes[bj];
                            const bool bh = !shadowed && box_intersect(bx, start, inv);
                            if (STATS == 1) sm_sslab += 1;
                            if (__any(bh) && !RT_ABL(16384)) {
                                // (a per-triangle cull against the light's beam, as for the primary rays, was measured:
                                // -1.7 % at 4K, +2.8 % at 1080p, one more spilled register -- not kept)
                                for (int base = 0; base < bx.len; base += 7) {
                                    const int cnt = bx.len - base < 7 ? bx.len - base : 7;
                                    mytri[lane] = ax->tri9[(size_t)(bx.start + base) * 9 + lane];
                                    wave_lds_sync();
                                    for (int i = 0; i < cnt; ++i) {
                                        const float *tv = mytri + 9 * i;
                                        float t, u, v;
                                        if (STATS == 1) sm_stri += 1;
                                        if (bh && !shadowed && tri_intersect(start, new_dir, tv, tv + 3, tv + 6, t, u, v))
                                            shadowed = true;
                                    }
                                    wave_lds_sync();
                                }
                                if (__all(shadowed)) break;
                            }
                        }
                    }
                    // planes (kernel.cu:1511-1523) then cubes (:1524-1536), any-hit
                    if (PRIMS && (fc.n_planes | fc.n_cubes) != 0 && !__all(shadowed)) {
                        for (int i = 0; i < fc.n_planes; ++i) {
                            float t;
                            if (!shadowed && plane_intersect(ax->planes[i], start, new_dir, t)) shadowed = true;
                            if (__all(shadowed)) break;
                        }
                        if (fc.n_cubes > 0 && !__all(shadowed)) {
                            const V3 inv{1.f / new_dir.x, 1.f / new_dir.y, 1.f / new_dir.z};
                            for (int i = 0; i < fc.n_cubes; ++i) {
                                float t;
                                if (!shadowed && cube_intersect(ax->cubes[i], start, inv, t)) shadowed = true;
                                if (__all(shadowed)) break;
                            }
                        }
                    }
                    if (!shadowed) unshadowed += 1;   // b += 0.1, kernel.cu:1537-1539
                    phase(7);
                }
                if (CULL) wave_lds_sync();
                if (STATS == 1 && MESH && !all_clear) sm_slisted += (unsigned long long)sbcount;
                if (STATS == 1 && !all_clear) {   // how many of the walked lights had a lane in a penumbra at all
                    const bool pen = lit && unshadowed != 0 && unshadowed != RT_SHADOW_SAMPLES;
                    st_walks += 1;
                    st_walks_no_penumbra += __any(pen) ? 0 : 1;
                    st_walks_all_dark += __any(lit && unshadowed != 0) ? 0 : 1;
                    st_walks_all_lit += __any(lit && unshadowed != RT_SHADOW_SAMPLES) ? 0 : 1;
                    st_pen_lanes += (unsigned long long)__popcll(__ballot(pen));
                    st_walk_lanes += (unsigned long long)__popcll(__ballot(lit));
                }

                if (lit) {   // unlit lanes would add (0 * l.r) * r = +0
                    // b after `unshadowed` float+=double steps, then b *= max(normal.toL, 0)
                    float bsum = mybtab[unshadowed];
                    const float a = dot3(normal, chain.toL);                    // kernel.cu:1541
                    bsum = bsum * (a > 0.f ? a : 0.f);
                    fr = fr + bsum * L.r * tr;                                  // kernel.cu:1673-1675
                    fg = fg + bsum * L.g * tg;
                    fb = fb + bsum * L.b * tb;
                    if (STATS == 1) st_unshadowed += (unsigned long long)unshadowed;
                }
            }
            }   // groups
        }
        if (valid) {
            acc_r = acc_r + fr;
            acc_g = acc_g + fg;
            acc_b = acc_b + fb;
        }
    }

    // ================= write-back =================
    // The frame uniforms of this part -- output pointers, flags, the divisor -- are read from the kernel-argument
    // segment HERE: taken from `fc` they are loaded at the kernel's entry and held in scalar registers across the whole
    // kernel, whose scalar file is full (each one more is a v_writelane / v_readlane pair in the loops above).
    FcPtr kargs = (FcPtr)__builtin_amdgcn_kernarg_segment_ptr();   // `fc` is the first kernel argument
    asm volatile("" : "+s"(kargs));                                // opaque: not merged with the loads at the entry
    float *const o_rgba = kargs->rgba;
    uint32_t *const o_packed = kargs->packed, *const o_packed24 = kargs->packed24;
    unsigned *const o_cost = kargs->tile_cost;
    const int o_flags = kargs->flags;
    const float o_total = kargs->sample_total;
    if (valid) {
        const size_t o = out_idx;
        float w = (float)n_samples;
        if (o_rgba) {
            float4 *dst = reinterpret_cast<float4 *>(o_rgba) + o;
            if (o_flags & RT_FLAG_ACCUMULATE) {
                const float4 old = *dst;
                acc_r = old.x + acc_r;
                acc_g = old.y + acc_g;
                acc_b = old.z + acc_b;
                w = old.w + w;
            }
            *dst = make_float4(acc_r, acc_g, acc_b, w);
        }
        if (o_packed && (o_flags & RT_FLAG_RESOLVE)) {
            // mean over the frame's samples (x/1.0f is exact, so 1 spp is the
            // reference's rgbToInt(fr*254, fg*254, fb*254), kernel.cu:1682/1688)
            float mr = acc_r, mg = acc_g, mb = acc_b;
            if (o_total != 1.f) {   // wave-uniform; three IEEE divisions saved at 1 spp
                mr = acc_r / o_total;
                mg = acc_g / o_total;
                mb = acc_b / o_total;
            }
            o_packed[o] = rgb_to_int(f2i(mr * 254.f), f2i(mg * 254.f), f2i(mb * 254.f));
        }
    }
    if (o_packed24 && (o_flags & RT_FLAG_RESOLVE)) {   // wave-uniform; all lanes take part in the quad exchange
        float mr = acc_r, mg = acc_g, mb = acc_b;
        if (o_total != 1.f) {
            mr = acc_r / o_total;
            mg = acc_g / o_total;
            mb = acc_b / o_total;
        }
        const unsigned p = rgb_to_int(f2i(mr * 254.f), f2i(mg * 254.f), f2i(mb * 254.f));
        // the next pixel of the quad (lanes 4q..4q+3 hold four consecutive pixels of a row)
        const unsigned pn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0xF9 /* quad_perm [1,2,3,3] */, 0xf, 0xf, true);
        const int i = lane & 3;
        // bytes B,G,R of pixel k at 3k..3k+2: dword i of the quad's three
        const unsigned w24 = (p >> (8 * i)) | (pn << (24 - 8 * i));
        if (valid && i < 3) o_packed24[(size_t)(out_idx >> 2) * 3 + (size_t)i] = w24;
    }

    if (!TABLDS && o_cost) {   // this tile's wave duration, for the order of later frames (a plain store: an atomic maximum
        // per block of tiles here, 256 waves ending together on one address, slowed the whole launch down by 3 %)
        const unsigned dt = (unsigned)__builtin_amdgcn_s_memtime() - t_start;
        if (lane == 0) o_cost[blk_y * tiles_x + blk_x] = dt;
    }

    phase(3, true);
    if (STATS == 2 && fc.stats) {
        if (lane == 0) {
            unsigned long long tot = 0;
            for (int k = 0; k < 8; ++k) {
                atomicAdd(&fc.stats[8 + k], ph[k]);
                tot += ph[k];
            }
            // wave durations (constant 100 MHz clock), octaves: [<5, <10, <20, <40, <80, <160, >=160] us
            const unsigned long long span = __builtin_amdgcn_s_memrealtime() - rt_begin;
            int bin = 0;
            while (bin < 6 && span >= (500ull << bin)) ++bin;
            atomicAdd(&fc.stats[17 + bin], 1ull);
            (void)tot;
        }
    }
    if (STATS == 1 && fc.stats) {
        // per-lane counters were kept wave-uniform except hits/unshadowed
        const unsigned long long h = (unsigned long long)wave_sum((float)st_hits);
        const unsigned long long u = (unsigned long long)wave_sum((float)st_unshadowed);
        if (lane == 0) {
            atomicAdd(&fc.stats[0], st_primary);
            atomicAdd(&fc.stats[1], st_shadow);
            atomicAdd(&fc.stats[2], st_cull);
            atomicAdd(&fc.stats[3], h);
            atomicAdd(&fc.stats[4], u);
            atomicAdd(&fc.stats[5], st_slots);
            atomicAdd(&fc.stats[6], st_entries);
            atomicAdd(&fc.stats[7], st_overflow);
            for (int k = 0; k < 8; ++k) atomicAdd(&fc.stats[8 + k], hist[k]);
            atomicAdd(&fc.stats[17], st_walks);               // (slots 17.. hold the wave-duration histogram in MODE 3)
            if (MESH) {   // mesh launches: the mesh's work instead of the penumbra counters (tools/mesh_stats.py)
                atomicAdd(&fc.stats[18], sm_plisted);
                atomicAdd(&fc.stats[19], sm_pleaf);
                atomicAdd(&fc.stats[20], sm_ptri);
                atomicAdd(&fc.stats[21], sm_slisted);
                atomicAdd(&fc.stats[22], sm_sslab);
                atomicAdd(&fc.stats[23], sm_stri);
            } else {
                atomicAdd(&fc.stats[18], st_walks_no_penumbra);
                atomicAdd(&fc.stats[19], st_pen_lanes);
                atomicAdd(&fc.stats[20], st_walk_lanes);
                atomicAdd(&fc.stats[21], st_walks_all_dark);
                atomicAdd(&fc.stats[22], st_walks_all_lit);
            }
            atomicAdd(&fc.stats[16], st_clusters);
        }
    }
}

// ---------------------------------------------------------------------------
// diagnostics: scalar building blocks evaluated on the device (tests only)
// ---------------------------------------------------------------------------
__global__ void rt_dbg_math(int op, const float *a, const float *b, float *out, int n)
{
    __shared__ double atab[16];   // as the frame kernel: atan(k/8) read from LDS
    if (threadIdx.x < 16) atab[threadIdx.x] = kAtanEighth[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
    case 0: r = rtm::cosf_rt(a[i]); break;
    case 1: r = rtm::sinf_rt(a[i]); break;
    case 2: r = rtm::acosf_rt(a[i], atab); break;
    case 4: r = (float)((1.0 + rtm::div_by_3p1415((double)a[i])) * 0.5); break;   // kernel.cu:1402
    case 5: r = (float)rtm::div_by_3p1415((double)a[i]); break;                  // kernel.cu:1403
    default: r = rtm::atan2f_rt(a[i], b[i], atab); break;
    }
    out[i] = r;
}

__global__ void rt_dbg_intersect(const float4 *tab, const float *rays, int n, int *hit, float *t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]};
    const V3 d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
    const RayK r = make_ray(o, d);
    const Quad q = quadratic(r, tab[i]);
    float tt;
    hit[i] = intersect_tail(r, q, tt) ? 1 : 0;
    t[i] = tt;
}


// castLightRay for n independent (start, normal) pairs against the whole table
// (brute force, no culling): the 10 sample directions and the returned brightness.
__global__ void rt_dbg_light(const RtFrameConsts fc, const float4 *tab, const float *starts,
                             const float *normals, int light_index, int n, float *dirs, float *bright)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int k = live ? i : 0;
    const V3 start{starts[3 * k + 0], starts[3 * k + 1], starts[3 * k + 2]};
    const V3 normal{normals[3 * k + 0], normals[3 * k + 1], normals[3 * k + 2]};
    const AuxPtr ax = (AuxPtr)(uintptr_t)fc.aux;
    const bool force_slow = (fc.flags & RT_FLAG_FORCE_SLOW) != 0;
    const RtLightDev L = ax->lights[light_index];
    ShadowChain<false> chain;
    chain.begin(V3{L.px, L.py, L.pz}, start);
    int unshadowed = 0;
    for (int j = 0; j < RT_SHADOW_SAMPLES; ++j) {
        const V3 d = chain.direction(ax, force_slow, L, start, j);
        if (live) {
            dirs[30 * i + 3 * j + 0] = d.x;
            dirs[30 * i + 3 * j + 1] = d.y;
            dirs[30 * i + 3 * j + 2] = d.z;
        }
        const RayK sr = make_ray(start, d);
        bool shadowed = !live;
        for (int e = 0; e < fc.n_spheres; ++e) {
            shadow_test(sr, tab[e], shadowed, force_slow);
            if (__all(shadowed)) break;
        }
        if (!shadowed) unshadowed += 1;
    }
    if (live) {
        float b = brightness_steps(unshadowed);
        const float a = dot3(normal, chain.toL);
        bright[i] = b * (a > 0.f ? a : 0.f);
    }
}

// The shortcuts of the culling kernels against the long forms (rt_debug_shortcuts, tests only).
__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float unit_float(unsigned h) { return (float)(h >> 8) * 0x1.0p-24f; }   // [0, 1)

__global__ void rt_dbg_shortcuts(int what, unsigned seed, long long n, unsigned long long *out)
{
    __shared__ double atab[16];
    if (threadIdx.x < 16) atab[threadIdx.x] = kAtanEighth[threadIdx.x];
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, accepted = 0, wrong = 0;
    float emax_x = 0.f, emax_y = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned h0 = hash32((unsigned)i * 3u + seed), h1 = hash32((unsigned)i * 3u + 1u + seed * 7919u),
                       h2 = hash32((unsigned)i * 3u + 2u + seed * 104729u), h3 = hash32(h0 ^ (unsigned)(i >> 32) ^ 0x9e3779b9u);
        if (what == 0) {
            // vectors of every scale: unit-ish, tiny, huge, with zero / denormal components now and then
            const int e = (int)(h3 % 61u) - 30 + ((h3 >> 8) % 7u == 0 ? ((h3 >> 12) % 2u ? 60 : -60) : 0);
            const float sc = __builtin_ldexpf(1.f, e);
            V3 a{(unit_float(h0) * 2.f - 1.f) * sc, (unit_float(h1) * 2.f - 1.f) * sc, (unit_float(h2) * 2.f - 1.f) * sc};
            if ((h3 >> 20) % 13u == 0) a.x = 0.f;
            if ((h3 >> 24) % 17u == 0) a.y = -0.f;
            if ((h3 >> 28) % 5u == 0) a.z = a.z * 0x1.0p-100f;
            V3 b = a;
            const V3 ra = normalise_inplace(a);
            const V3 rb = normalise_t<true>(b);
            const bool same = __builtin_bit_cast(unsigned, a.x) == __builtin_bit_cast(unsigned, b.x) &&
                              __builtin_bit_cast(unsigned, a.y) == __builtin_bit_cast(unsigned, b.y) &&
                              __builtin_bit_cast(unsigned, a.z) == __builtin_bit_cast(unsigned, b.z) &&
                              __builtin_bit_cast(unsigned, ra.x) == __builtin_bit_cast(unsigned, rb.x) &&
                              __builtin_bit_cast(unsigned, ra.y) == __builtin_bit_cast(unsigned, rb.y) &&
                              __builtin_bit_cast(unsigned, ra.z) == __builtin_bit_cast(unsigned, rb.z);
            if (!same) bad += 1;
        } else if (what == 2) {
            // unit normals: uniform on the sphere, plus clusters at the poles and the seams
            float z = unit_float(h0) * 2.f - 1.f, phi = unit_float(h1) * 6.2831853f;
            if ((h3 & 15u) == 0) z = 1.f - unit_float(h2) * 1.0e-6f;
            if ((h3 & 15u) == 1) z = -1.f + unit_float(h2) * 1.0e-6f;
            if ((h3 & 15u) == 2) phi = (float)((h2 >> 4) & 7u) * 0.78539816f + (unit_float(h2) - 0.5f) * 1.0e-6f;
            const float r = __builtin_sqrtf(__builtin_fmaxf(1.f - z * z, 0.f));
            V3 nrm{r * __builtin_cosf(phi), z, r * __builtin_sinf(phi)};
            normalise_inplace(nrm);
            const float tx = (float)((1.0 + rtm::div_by_3p1415((double)rtm::atan2f_rt(nrm.z, nrm.x, atab))) * 0.5);
            const float ty = (float)rtm::div_by_3p1415((double)rtm::acosf_rt(nrm.y, atab));
            float ux, uy;
            approx_sphere_uv(nrm, ux, uy);
            const float ex = __builtin_fabsf(ux - tx), ey = __builtin_fabsf(uy - ty);
            // a NaN approximation (0/0 at the poles, where the exact atan2 is defined) is simply never "sure"
            if (ex == ex) emax_x = __builtin_fmaxf(emax_x, ex);
            if (ey == ey) emax_y = __builtin_fmaxf(emax_y, ey);
            const int w = 512, hh = 512;
            const float mu = (float)w * (RT_UV_DELTA + 0x1.0p-22f) * 1.01f;
            const int cf = sure_texel(ux, uy, w, hh, mu, mu);
            if (cf >= 0) {
                accepted += 1;
                if (cf != f2i(ty * (float)hh) * w + f2i(tx * (float)w)) wrong += 1;
            }
            // the sky's own (binary32) expressions, kernel.cu:1157-1158, at 2048 x 1024 with its wider margin
            const int sw = 2048, sh = 1024;
            const int cs = sure_texel(ux, uy, sw, sh, (float)sw * (1.0e-6f + 0x1.0p-22f) * 1.01f, (float)sh * (1.0e-6f + 0x1.0p-22f) * 1.01f);
            if (cs >= 0) {
                const int ix = f2i((1.f + rtm::atan2f_rt(nrm.z, nrm.x, atab) / 3.1415f) * 0.5f * (float)sw);
                const int iy = f2i(rtm::acosf_rt(nrm.y, atab) / 3.1415f * (float)sh);
                if (cs != iy * sw + ix) wrong += 1;
            }
        }
    }
    if (what == 1) {   // every float of [2^-96, 2^40]: bit patterns 0x0F800000 .. 0x53800000
        const unsigned lo = 0x0F800000u, hi = 0x53800000u;
        for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (unsigned long long)stride) {
            const float x = __builtin_bit_cast(float, (unsigned)b);
            if (__builtin_bit_cast(unsigned, lean_sqrt(x)) != __builtin_bit_cast(unsigned, __builtin_sqrtf(x))) bad += 1;
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (what == 2) {
        atomicMax(&out[0], (unsigned long long)__builtin_bit_cast(unsigned, emax_x));
        atomicMax(&out[1], (unsigned long long)__builtin_bit_cast(unsigned, emax_y));
        if (accepted) atomicAdd(&out[2], accepted);
        if (wrong) atomicAdd(&out[3], wrong);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host-side launchers (called from rt_engine.cpp / rt_graph.cpp)
// ---------------------------------------------------------------------------
// The instantiations that exist. Tile widths other than 8 and whole-table LDS staging are
// tuning / test dimensions: TABLDS exists for the default tile only (other tiles read the
// table from global memory whatever was asked), mesh scenes (FEAT 2) render with the default
// tile in modes 0 and 2, and MODE 3 (phase stamps) exists in RT_TUNING builds only.
typedef void (*RtTraceFn)(const RtFrameConsts, const float4 *);

template <int TW, bool CULL, bool TABLDS, bool MULTI>
static RtTraceFn trace_fn_mode_feat(int mode, int feat)
{
    if (feat == 2) {   // whole-table LDS staging and a sample loop together are not instantiated for mesh scenes
        if constexpr (!(TABLDS && MULTI)) {
            if (mode == 0) return rt_trace_tiles<TW, CULL, 0, TABLDS, 2, MULTI>;
            if (mode == 1) return rt_trace_tiles<TW, CULL, 1, TABLDS, 2, MULTI>;
            if (mode == 2) return rt_trace_tiles<TW, CULL, 2, TABLDS, 2, MULTI>;
        }
        return nullptr;
    }
    switch (mode) {
    case 0: return feat ? rt_trace_tiles<TW, CULL, 0, TABLDS, 1, MULTI> : rt_trace_tiles<TW, CULL, 0, TABLDS, 0, MULTI>;
    case 1: return feat ? rt_trace_tiles<TW, CULL, 1, TABLDS, 1, MULTI> : rt_trace_tiles<TW, CULL, 1, TABLDS, 0, MULTI>;
    case 2: return feat ? rt_trace_tiles<TW, CULL, 2, TABLDS, 1, MULTI> : rt_trace_tiles<TW, CULL, 2, TABLDS, 0, MULTI>;
    case 4:   // fast mode: the default tile, culling on, table in global memory
        if constexpr (TW == 8 && CULL && !TABLDS)
            return feat ? rt_trace_tiles<TW, CULL, 4, TABLDS, 1, MULTI> : rt_trace_tiles<TW, CULL, 4, TABLDS, 0, MULTI>;
        return nullptr;
#ifdef RT_TUNING
    case 3: return feat ? rt_trace_tiles<TW, CULL, 3, TABLDS, 1, MULTI> : rt_trace_tiles<TW, CULL, 3, TABLDS, 0, MULTI>;
#endif
    default: return nullptr;
    }
}

template <int TW>
static RtTraceFn trace_fn_tw(int cull, int mode, int table_in_lds, int feat, int multi)
{
    if constexpr (TW == 8) {   // the default tile: every combination, with and without the sample loop
        if (table_in_lds) {
            if (multi) return cull ? trace_fn_mode_feat<TW, true, true, true>(mode, feat) : trace_fn_mode_feat<TW, false, true, true>(mode, feat);
            return cull ? trace_fn_mode_feat<TW, true, true, false>(mode, feat) : trace_fn_mode_feat<TW, false, true, false>(mode, feat);
        }
        if (!multi) return cull ? trace_fn_mode_feat<TW, true, false, false>(mode, feat) : trace_fn_mode_feat<TW, false, false, false>(mode, feat);
    }
    return cull ? trace_fn_mode_feat<TW, true, false, true>(mode, feat) : trace_fn_mode_feat<TW, false, false, true>(mode, feat);
}

static RtTraceFn trace_fn(int tile_w, int cull, int mode, int table_in_lds, int feat, int multi)
{
    switch (tile_w) {
    case 8: return trace_fn_tw<8>(cull, mode, table_in_lds, feat, multi);
    case 16: return trace_fn_tw<16>(cull, mode, table_in_lds, feat, multi);
    case 32: return trace_fn_tw<32>(cull, mode, table_in_lds, feat, multi);
    case 64: return trace_fn_tw<64>(cull, mode, table_in_lds, feat, multi);
    default: return nullptr;
    }
}

// Raise the dynamic-LDS limit of the instantiations that stage the whole table (a single
// workgroup may use the whole 160 KiB). Not a stream operation: it runs once, outside any
// stream capture or graph construction.
extern "C" hipError_t rt_dev_prepare(void)
{
    static unsigned long long done = 0;   // one bit per device: function attributes belong to a device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorNoDevice;
    if (dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    for (int cull = 0; cull < 2; ++cull)
        for (int mode = 0; mode < 5; ++mode)
            for (int feat = 0; feat < 3; ++feat)
                for (int multi = 0; multi < 2; ++multi) {
                    const RtTraceFn fn = trace_fn(8, cull, mode, 1, feat, multi);
                    if (!fn) continue;
                    const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return e;
                }
    if (dev < 64) done |= 1ull << dev;
    return hipSuccess;
}

// Everything a launch of the frame kernel needs besides its two arguments. table_in_lds is
// honoured for the default tile only. hipErrorNotSupported: no such instantiation.
extern "C" hipError_t rt_dev_trace_config(const RtFrameConsts *fc, int tile_w, int cull, int mode, int table_in_lds, int feat,
                                          const void **func, dim3 *grid, dim3 *block, unsigned *lds_bytes)
{
    if (tile_w != 8 || (feat == 2 && fc->spp > 1)) table_in_lds = 0;
    const RtTraceFn fn = trace_fn(tile_w, cull, mode, table_in_lds, feat, fc->spp > 1 ? 1 : 0);
    if (!fn) return hipErrorNotSupported;
    const int n_pad = (fc->n_spheres + 63) & ~63;
    const int wpw = table_in_lds ? RT_WAVES_PER_WG : 1;   // as WPW in the kernel
    *lds_bytes = (unsigned)((size_t)((table_in_lds ? n_pad : 0) + wpw * RT_LIST_CAP) * sizeof(float4) +
                            (size_t)wpw * RT_LIST_CAP * sizeof(int) +   // list positions (primary order)
                            (size_t)wpw * 16 * sizeof(float) +          // brightness table per wave
                            (size_t)wpw * 64 * sizeof(int) +            // marked blocks of a culling pass
                            (size_t)wpw * 16 * sizeof(double) +         // atan(k/8) per wave
                            (feat == 2 ? (size_t)wpw * (RT_BOX_CAP + 128) * sizeof(int) : 0));
    const int th = 64 / tile_w;
    const int wgx = (tile_w <= 16 && wpw >= 2) ? 2 : 1;
    const int wgy = wpw / wgx;
    *grid = dim3((fc->width + tile_w * wgx - 1) / (tile_w * wgx), (fc->local_rows + th * wgy - 1) / (th * wgy));
    *block = dim3(64 * wpw);
    *func = (const void *)fn;
    return rt_dev_prepare();
}

extern "C" hipError_t rt_dev_launch_trace(const RtFrameConsts *fc, const float4 *spheres, int tile_w, int cull, int mode,
                                          int table_in_lds, int feat, hipStream_t stream)
{
    const void *func = nullptr;
    dim3 grid, block;
    unsigned lds_bytes = 0;
    const hipError_t ce = rt_dev_trace_config(fc, tile_w, cull, mode, table_in_lds, feat, &func, &grid, &block, &lds_bytes);
    if (ce != hipSuccess) return ce;
    hipLaunchKernelGGL((RtTraceFn)func, grid, block, lds_bytes, stream, *fc, spheres);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_shortcuts(int what, unsigned seed, long long n, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_shortcuts, dim3(4096), dim3(256), 0, stream, what, seed, n, out);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_math(int op, const float *a, const float *b, float *out, int n,
                                             hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_intersect(const float4 *tab, const float *rays, int n, int *hit,
                                                  float *t, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_intersect, dim3((n + 255) / 256), dim3(256), 0, stream, tab, rays, n, hit, t);
    return hipGetLastError();
}

extern "C" hipError_t rt_dev_launch_dbg_light(const RtFrameConsts *fc, const float4 *tab, const float *starts,
                                              const float *normals, int light_index, int n, float *dirs,
                                              float *bright, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_dbg_light, dim3((n + 63) / 64), dim3(64), 0, stream, *fc, tab, starts, normals,
                       light_index, n, dirs, bright);
    return hipGetLastError();
}
