// rt_tables.hip -- the culling tables of a device-resident scene: orderings of the sphere
// table {cx,cy,cz,radius*radius} with conservative block bounds (see rt_device.h and DESIGN.md
// section 4). None of them can change a pixel: a table only decides which spheres the exact
// tests of rt_kernels.hip get to see, and every bound is padded so that a sphere left out is
// one whose exact test (/root/reference/kernel.cu:293-354) would have returned false.
//
//   build_sorted_blocks   3-D Morton order, bounding spheres      host, when the list changes
//   build_light_columns   per light: order ACROSS its axis        host, when a light or the list changes
//   eye cones             order by direction as seen from the     DEVICE (rt_eye_cones_launch), every time
//                         ray origin, cones from the origin       the camera moves: one 1024-thread
//                                                                 workgroup, bitonic sort in LDS; the
//                                                                 host version is the fallback for
//                                                                 lists beyond RT_EYE_DEVICE_MAX
// The reference moves its camera every frame (checkKey, kernel.cu:1716-1759): that must not
// cost a host-side sort, a blocking upload and a device synchronisation per frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "rt_device.h"
#include "rt_tables.h"

// ---------------------------------------------------------------------------
// shared host / device pieces
// ---------------------------------------------------------------------------
#define RT_HDI __host__ __device__ inline

RT_HDI unsigned morton16(unsigned v)
{
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

static unsigned morton10(unsigned v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

RT_HDI void oct_map(const double d[3], double *u, double *v)
{
    const double s = fabs(d[0]) + fabs(d[1]) + fabs(d[2]);
    double x = d[0] / s, y = d[1] / s;
    if (d[2] < 0) {
        const double ox = (1 - fabs(y)) * (x >= 0 ? 1 : -1), oy = (1 - fabs(x)) * (y >= 0 ? 1 : -1);
        x = ox;
        y = oy;
    }
    *u = x;
    *v = y;
}

// One sphere as seen from the ray origin. For a beam with apex O, r0 = 1e-4, smin = 0 and slope
// k <= kcap the member test of the kernel (beam_member_test) passes only if the angle alpha
// between the beam axis and the direction to the centre satisfies
// sin(alpha - atan(1.00025 k)) <= (rc (1 + k) + r0) 1.00025 / |v|, rc = sqrt(R^2 + 4e-5 |v|^2 + 1e-3) 1.0001
// -- all known here because the apex is. `ext` is the asin of that bound; a sphere around or
// next to O is unbounded.
struct ConeEnt {
    double dir[3];
    double ext;
    bool bounded;
};

RT_HDI ConeEnt cone_entry(float4 sph, const float org[3])
{
    const double kcap = RT_CONE_KCAP, r0 = 1.0e-4;
    ConeEnt e;
    const double v[3] = {(double)sph.x - org[0], (double)sph.y - org[1], (double)sph.z - org[2]};
    const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dist = sqrt(vv), w = sph.w;
    e.bounded = (vv - vv == 0.0) && (w - w == 0.0) && dist > 0;   // finite, and not at the origin
    e.ext = 0;
    e.dir[0] = e.dir[1] = e.dir[2] = 0;
    if (e.bounded) {
        const double rc = sqrt((w > 0 ? w : 0.0) + (double)RT_PAD_REL * 1.0001 * vv + (double)RT_PAD_ABS * 1.0001) * 1.0001;
        const double q = (rc * (1.0 + kcap) + r0) * 1.00025 * 1.001 / dist;
        if (!(q < 0.99)) e.bounded = false;   // the origin is inside or next to the (padded) sphere
        else e.ext = asin(q);
        for (int k = 0; k < 3; ++k) e.dir[k] = v[k] / dist;
    }
    return e;
}

// Sort key: 2-D Morton code of the octahedral map of the direction; unbounded entries last.
RT_HDI unsigned cone_key(const ConeEnt &e)
{
    if (!e.bounded) return 0xffffffffu;
    double ou, ov;
    oct_map(e.dir, &ou, &ov);
    const double a = (ou * 0.5 + 0.5) * 65535.0, b = (ov * 0.5 + 0.5) * 65535.0;
    const unsigned q1 = (unsigned)(a < 0.0 ? 0.0 : (a > 65535.0 ? 65535.0 : a));
    const unsigned q2 = (unsigned)(b < 0.0 ? 0.0 : (b > 65535.0 ? 65535.0 : b));
    return morton16(q1) | (morton16(q2) << 1);
}

// ---------------------------------------------------------------------------
// 3-D Morton order of the centres (10 bits per axis over the scene's bounds), blocks of
// RT_BLOCK consecutive spheres, and for each block a sphere that contains every member
// (centre = mean of the members' centres, radius = max |c_i - centre| + R_i, rounded
// up). Spheres with non-finite data make their block unbounded (always examined).
// ---------------------------------------------------------------------------
void rt_build_sorted_blocks(const float4 *tab, int n, float4 *sorted, float4 *blocks, int *orig)
{
    const int n_pad = (n + 63) & ~63;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; ++i) {
        const float c[3] = {tab[i].x, tab[i].y, tab[i].z};
        for (int k = 0; k < 3; ++k)
            if (std::isfinite(c[k])) {
                lo[k] = std::min(lo[k], c[k]);
                hi[k] = std::max(hi[k], c[k]);
            }
    }
    std::vector<std::pair<unsigned, int>> keys((size_t)n);
    for (int i = 0; i < n; ++i) {
        const float c[3] = {tab[i].x, tab[i].y, tab[i].z};
        unsigned q[3];
        for (int k = 0; k < 3; ++k) {
            const float span = hi[k] - lo[k];
            const float t = (std::isfinite(c[k]) && span > 0) ? (c[k] - lo[k]) / span : 0.f;
            q[k] = (unsigned)std::min(1023.f, std::max(0.f, t * 1023.f));
        }
        keys[i] = {morton10(q[0]) | (morton10(q[1]) << 1) | (morton10(q[2]) << 2), i};
    }
    std::sort(keys.begin(), keys.end());
    for (int i = 0; i < n_pad; ++i) {
        sorted[i] = i < n ? tab[keys[i].second] : make_float4(0.f, 0.f, 0.f, 0.f);
        orig[i] = i < n ? keys[i].second : 0x7fffffff;
    }
    for (int b = 0; b < n_pad / RT_BLOCK; ++b) {
        const int i0 = b * RT_BLOCK, i1 = std::min(n, i0 + RT_BLOCK);
        if (i0 >= n) {   // padding block: nothing in it, never examined
            blocks[b] = make_float4(0.f, 0.f, 0.f, -1.f);
            continue;
        }
        double cx = 0, cy = 0, cz = 0;
        for (int i = i0; i < i1; ++i) { cx += sorted[i].x; cy += sorted[i].y; cz += sorted[i].z; }
        const double inv = 1.0 / std::max(1, i1 - i0);
        cx *= inv; cy *= inv; cz *= inv;
        double r = 0;
        for (int i = i0; i < i1; ++i) {
            const double dx = sorted[i].x - cx, dy = sorted[i].y - cy, dz = sorted[i].z - cz;
            const double ri = std::sqrt(std::max(0.0, (double)sorted[i].w));
            const double d = std::sqrt(dx * dx + dy * dy + dz * dz) + ri;
            r = (d > r || d != d) ? d : r;   // a NaN sticks
        }
        float rf = (float)(r * 1.001 + 1e-3);
        if (!(rf == rf) || !std::isfinite(cx + cy + cz)) { rf = INFINITY; cx = cy = cz = 0; }
        blocks[b] = make_float4((float)cx, (float)cy, (float)cz, rf);
    }
}

// ---------------------------------------------------------------------------
// Per-light column blocks. All shadow rays of a light run within a few degrees of
// u = l.pos/|l.pos| (kernel.cu:1468 builds them relative to the world origin), so the table
// is ordered by where the centres fall ACROSS u and cut into blocks of RT_BLOCK: columns
// along u, which a beam along u touches far less often than the cubes of the 3-D order.
// Block record, two float4: {cx, cy, cz, rho} and {s_hi, r3d, 0, 0} -- c the mean centre,
// rho >= |(c_j - c) across u| + R_j, s_hi >= (c_j - c).u + R_j, r3d >= |c_j - c| + R_j for
// every member j (R_j = sqrt of the table's squared effective radius), all rounded up.
// ---------------------------------------------------------------------------
void rt_build_light_columns(const float4 *tab, int n, const float u_f[3], float4 *sorted, float4 *blocks)
{
    const int n_pad = (n + 63) & ~63;
    const double u[3] = {u_f[0], u_f[1], u_f[2]};
    // two directions across u
    double e1[3] = {0, 0, 0};
    {
        const int k = (std::fabs(u[0]) <= std::fabs(u[1]) && std::fabs(u[0]) <= std::fabs(u[2])) ? 0
                      : (std::fabs(u[1]) <= std::fabs(u[2]) ? 1 : 2);
        double t[3] = {0, 0, 0};
        t[k] = 1;
        const double d = t[0] * u[0] + t[1] * u[1] + t[2] * u[2];
        for (int i = 0; i < 3; ++i) e1[i] = t[i] - d * u[i];
        const double l = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
        for (int i = 0; i < 3; ++i) e1[i] /= l;
    }
    const double e2[3] = {u[1] * e1[2] - u[2] * e1[1], u[2] * e1[0] - u[0] * e1[2], u[0] * e1[1] - u[1] * e1[0]};
    std::vector<double> p1((size_t)n), p2((size_t)n);
    double lo1 = INFINITY, hi1 = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
    std::vector<char> fin((size_t)n);
    for (int i = 0; i < n; ++i) {
        const double c[3] = {tab[i].x, tab[i].y, tab[i].z};
        p1[i] = c[0] * e1[0] + c[1] * e1[1] + c[2] * e1[2];
        p2[i] = c[0] * e2[0] + c[1] * e2[1] + c[2] * e2[2];
        fin[i] = std::isfinite(p1[i]) && std::isfinite(p2[i]) && std::isfinite((double)tab[i].w);
        if (fin[i]) {
            lo1 = std::min(lo1, p1[i]); hi1 = std::max(hi1, p1[i]);
            lo2 = std::min(lo2, p2[i]); hi2 = std::max(hi2, p2[i]);
        }
    }
    std::vector<std::pair<unsigned long long, int>> keys((size_t)n);
    for (int i = 0; i < n; ++i) {
        unsigned long long key = ~0ull;   // non-finite entries go last (their blocks are unbounded)
        if (fin[i]) {
            const double s1 = hi1 - lo1, s2 = hi2 - lo2;
            const unsigned q1 = (unsigned)std::min(65535.0, std::max(0.0, s1 > 0 ? (p1[i] - lo1) / s1 * 65535.0 : 0.0));
            const unsigned q2 = (unsigned)std::min(65535.0, std::max(0.0, s2 > 0 ? (p2[i] - lo2) / s2 * 65535.0 : 0.0));
            key = morton16(q1) | ((unsigned long long)morton16(q2) << 1);
        }
        keys[i] = {key, i};
    }
    std::sort(keys.begin(), keys.end());
    for (int i = 0; i < n_pad; ++i) sorted[i] = i < n ? tab[keys[i].second] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < n_pad / RT_BLOCK; ++b) {
        const int i0 = b * RT_BLOCK, i1 = std::min(n, i0 + RT_BLOCK);
        if (i0 >= n) {   // padding block: nothing in it, never examined
            blocks[2 * b] = make_float4(0.f, 0.f, 0.f, -1.f);
            blocks[2 * b + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        double cx = 0, cy = 0, cz = 0;
        for (int i = i0; i < i1; ++i) { cx += sorted[i].x; cy += sorted[i].y; cz += sorted[i].z; }
        const double inv = 1.0 / std::max(1, i1 - i0);
        // the bounds below are taken around the ROUNDED centre the device will use
        const float cf[3] = {(float)(cx * inv), (float)(cy * inv), (float)(cz * inv)};
        double rho = 0, s_hi = -INFINITY, r3d = 0;
        bool bad = !(std::isfinite(cf[0]) && std::isfinite(cf[1]) && std::isfinite(cf[2]));
        for (int i = i0; i < i1 && !bad; ++i) {
            const double d[3] = {sorted[i].x - (double)cf[0], sorted[i].y - (double)cf[1], sorted[i].z - (double)cf[2]};
            const double w = sorted[i].w;
            if (!(w == w) || !std::isfinite(d[0] + d[1] + d[2]) || !std::isfinite(w)) { bad = true; break; }
            const double R = std::sqrt(std::max(0.0, w));
            const double ax = d[0] * u[0] + d[1] * u[1] + d[2] * u[2];
            const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const double lat = std::sqrt(std::max(0.0, dd - ax * ax));
            rho = std::max(rho, lat + R);
            s_hi = std::max(s_hi, ax + R);
            r3d = std::max(r3d, std::sqrt(dd) + R);
        }
        if (bad) {   // always examined
            blocks[2 * b] = make_float4(0.f, 0.f, 0.f, INFINITY);
            blocks[2 * b + 1] = make_float4(INFINITY, INFINITY, 0.f, 0.f);
            continue;
        }
        // rounded up; |u| differs from 1 by a few 1e-8, which the factors cover as well
        blocks[2 * b] = make_float4(cf[0], cf[1], cf[2], (float)(rho * 1.001 + 1e-3));
        blocks[2 * b + 1] = make_float4((float)(s_hi + std::fabs(s_hi) * 1e-3 + 1e-3), (float)(r3d * 1.001 + 1e-3), 0.f, 0.f);
    }
}

// ---------------------------------------------------------------------------
// Occluder lists. A shadow ray of castLightRay (kernel.cu:1438-1510) starts a hair above the surface it shades and
// runs within a few degrees of u = l.pos/|l.pos| WHATEVER its start (kernel.cu:1468 forms the direction relative to the
// world origin). So the spheres such a ray can hit from anywhere on sphere S are those that reach into the cone of
// slope kcap around u from a ball around S: a short list per (S, light), built here once per scene, of which a tile
// then only has to cull the members against its own, much thinner beam (rt_trace.inc: build_list_cand) instead of
// walking the light's column blocks. Conservative by the same member test the kernel applies (beam_member_test) with
// the ball of S (radius R_S 1.001 + 1e-3: the starts lie 1e-5 above the surface, give or take the rounding of the hit
// point) as the ray origins; kcap = the slope every group on S uses (kbeam, rt_sphere_beam_slope below) plus 0.1 %, or,
// for a sphere without one, 1.15 x the largest slope the kernel's own bound yields at 14 points of S; the kernel checks
// its beam's slope against kcap before using the list. Built on the device (rt_occluder_lists_kernel); the host builder
// is the same member test in a plain loop, for the tests.
// ---------------------------------------------------------------------------
RT_HDI bool fin_d(double x) { return x - x == 0.0; }

RT_HDI double light_beam_slope_hd(const double lpos[3], const double start[3])
{
    const double L = sqrt(lpos[0] * lpos[0] + lpos[1] * lpos[1] + lpos[2] * lpos[2]);
    double t[3] = {lpos[0] - start[0], lpos[1] - start[1], lpos[2] - start[2]};
    const double tl = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (!(L > 0) || !(tl > 0) || !fin_d(L + tl)) return NAN;
    for (int i = 0; i < 3; ++i) t[i] /= tl;
    const double u[3] = {lpos[0] / L, lpos[1] / L, lpos[2] / L};
    // an orthonormal pair across u (any will do: the singular value does not depend on it)
    const int k = (fabs(u[0]) <= fabs(u[1]) && fabs(u[0]) <= fabs(u[2])) ? 0 : (fabs(u[1]) <= fabs(u[2]) ? 1 : 2);
    double e1[3] = {-u[k] * u[0], -u[k] * u[1], -u[k] * u[2]};
    e1[0] += k == 0 ? 1.0 : 0.0;
    e1[1] += k == 1 ? 1.0 : 0.0;
    e1[2] += k == 2 ? 1.0 : 0.0;
    const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int i = 0; i < 3; ++i) e1[i] /= l1;
    const double e2[3] = {u[1] * e1[2] - u[2] * e1[1], u[2] * e1[0] - u[0] * e1[2], u[0] * e1[1] - u[1] * e1[0]};
    const double c = t[2], sn = sqrt(fmax(1.0 - c * c, 0.0)), q2 = t[0] * t[0] + t[1] * t[1];
    double kmax2, frob2;
    if (q2 < 1.0e-4) {
        kmax2 = frob2 = 8.0;     // as the kernel
    } else {
        const double rq = 1.0 / sqrt(q2), ax = -t[1] * rq, ay = t[0] * rq, omc = 1.0 - c;
        const double M[3][3] = {{c + ax * ax, ax * ay * omc, -ay * sn}, {ax * ay * omc, c + ay * ay * omc, -ax * sn}, {-ay * sn, ax * sn, c}};
        double p[3], q[3];
        frob2 = 0;
        for (int i = 0; i < 3; ++i) {
            p[i] = M[i][0] * e1[0] + M[i][1] * e1[1] + M[i][2] * e1[2];
            q[i] = M[i][0] * e2[0] + M[i][1] * e2[1] + M[i][2] * e2[2];
            frob2 += M[i][0] * M[i][0] + M[i][1] * M[i][1] + M[i][2] * M[i][2];
        }
        const double h11 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], h22 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
        const double h12 = p[0] * q[0] + p[1] * q[1] + p[2] * q[2], hd = 0.5 * (h11 - h22);
        kmax2 = (0.5 * (h11 + h22) + sqrt(hd * hd + h12 * h12)) * 1.001;
    }
    const double den = L - sqrt(frob2) * 1.001;
    if (!(den > 0.05 * L)) return NAN;
    const double s2 = kmax2 / (den * den) * 1.0001;
    if (!(s2 < 0.25)) return NAN;
    const double snw = sqrt(s2) * (double)RT_SPREAD_MUL + (double)RT_SPREAD_ADD;
    return snw / sqrt(fmax(1.0 - snw * snw, 0.05));
}

double rt_light_beam_slope(const double lpos[3], const double start[3]) { return light_beam_slope_hd(lpos, start); }

// ---------------------------------------------------------------------------
// The same bound once per SPHERE instead of once per tile and light: the supremum of the spread over every start in the
// ball B(c, r0) around a sphere -- which the frame kernel then takes as its beam's slope for any group of pixels on that
// sphere, without computing anything (the per-tile evaluation is a hundred instructions per lit light).
//   The spread depends on the start only through t = normalise(l.pos - start): s(t) = sigma_max(M(t) E) / (|l.pos| -
// ||M(t)||_F), M the (non-standard) rotation of kernel.cu:1267-1277 about (0,0,1) x t, E an orthonormal pair across u.
// The directions t of all starts in the ball form the cone of half-angle theta = asin(r0 / |l.pos - c|) around
// t_c = normalise(l.pos - c). s is sampled at 217 directions of that cone (the axis and rings of 6k points at
// alpha = theta k/8, k = 1..8: every direction of the cone is within 0.091 theta of a sample) and a Lipschitz term covers
// the rest: sigma_max and ||.||_F are 1-Lipschitz in the Frobenius norm and ||E||_2 = 1, so |ds| <= ||dM||_F (1/den +
// sigma/den^2), and along the unit sphere, with t = (q cos phi, q sin phi, t_z): M00 = t_z + sin^2 phi,
// M01 = M10 = -sin(2 phi)(1 - t_z)/2, M11 = t_z + cos^2 phi (1 - t_z), M02 = M20 = -t_x, M12 = -M21 = t_y, M22 = t_z, whose
// derivatives along an arc ds (|d phi| <= ds/q, |d t_z| <= ds) give ||dM||_F <= (4/q + 3) ds (checked numerically in
// tests/test_occluder_lists.py). sigma <= ||M||_F <= sqrt(6) everywhere (same test). Where the cone comes within 0.1 of
// the pole q = 0 (the light within six degrees of +-z as seen from the sphere) the bound for ANY direction is taken:
// sqrt(6) / (|l.pos| - sqrt(6)). Returns the slope as the kernel forms it from the spread (RT_SPREAD_MUL / _ADD: the
// allowances for the chain's own float arithmetic), or -1 where there is no usable bound (the kernel then evaluates its
// own, per tile).
RT_HDI void light_frame_hd(const double lpos[3], double *Lout, double u[3], double e1[3], double e2[3])
{
    const double L = sqrt(lpos[0] * lpos[0] + lpos[1] * lpos[1] + lpos[2] * lpos[2]);
    *Lout = L;
    for (int i = 0; i < 3; ++i) u[i] = lpos[i] / L;
    const int k = (fabs(u[0]) <= fabs(u[1]) && fabs(u[0]) <= fabs(u[2])) ? 0 : (fabs(u[1]) <= fabs(u[2]) ? 1 : 2);
    for (int i = 0; i < 3; ++i) e1[i] = -u[k] * u[i] + (i == k ? 1.0 : 0.0);
    const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int i = 0; i < 3; ++i) e1[i] /= l1;
    e2[0] = u[1] * e1[2] - u[2] * e1[1];
    e2[1] = u[2] * e1[0] - u[0] * e1[2];
    e2[2] = u[0] * e1[1] - u[1] * e1[0];
}

// s(t) for a unit t away from the pole; *fro = ||M(t)||_F, *sig = sigma_max(M(t) E)
RT_HDI double beam_sine_hd(const double t[3], const double e1[3], const double e2[3], double L, double *sig, double *fro)
{
    const double c = t[2], q2 = t[0] * t[0] + t[1] * t[1], sn = sqrt(q2);
    const double rq = 1.0 / sn, ax = -t[1] * rq, ay = t[0] * rq, omc = 1.0 - c;
    const double M[3][3] = {{c + ax * ax, ax * ay * omc, -ay * sn}, {ax * ay * omc, c + ay * ay * omc, -ax * sn}, {-ay * sn, ax * sn, c}};
    double p[3], q[3], frob2 = 0;
    for (int i = 0; i < 3; ++i) {
        p[i] = M[i][0] * e1[0] + M[i][1] * e1[1] + M[i][2] * e1[2];
        q[i] = M[i][0] * e2[0] + M[i][1] * e2[1] + M[i][2] * e2[2];
        frob2 += M[i][0] * M[i][0] + M[i][1] * M[i][1] + M[i][2] * M[i][2];
    }
    const double h11 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], h22 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double h12 = p[0] * q[0] + p[1] * q[1] + p[2] * q[2], hd = 0.5 * (h11 - h22);
    *sig = sqrt(0.5 * (h11 + h22) + sqrt(hd * hd + h12 * h12));
    *fro = sqrt(frob2);
    return *sig / (L - *fro);
}

#define RT_CONE_SAMPLES 217   // 1 + 6 (1 + 2 + ... + 8)
// the i-th sample direction of the cone of half-angle theta around tc (a1, a2: an orthonormal pair across tc)
RT_HDI void cone_sample_hd(int i, const double tc[3], const double a1[3], const double a2[3], double theta, double t[3])
{
    int k = 0;
    while (i >= 1 + 3 * k * (k + 1)) ++k;                   // ring k holds the indices 1 + 3k(k-1) .. 3k(k+1)
    const double alpha = theta * (double)k / 8.0;
    const double psi = k ? 6.283185307179586 * (double)(i - 1 - 3 * k * (k - 1)) / (double)(6 * k) : 0.0;
    const double ca = cos(alpha), sa = sin(alpha), cp = cos(psi), sp = sin(psi);
    for (int j = 0; j < 3; ++j) t[j] = ca * tc[j] + sa * (cp * a1[j] + sp * a2[j]);
    const double l = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int j = 0; j < 3; ++j) t[j] /= l;
}

struct ConeOfStarts {       // what every evaluation of one (sphere, light) pair shares
    double L, e1[3], e2[3], tc[3], a1[3], a2[3], theta, qmin, den_min;
    bool usable, polar;
};
RT_HDI ConeOfStarts cone_of_starts_hd(const double lpos[3], const double c[3], double r0)
{
    ConeOfStarts k;
    double u[3];
    light_frame_hd(lpos, &k.L, u, k.e1, k.e2);
    const double w[3] = {lpos[0] - c[0], lpos[1] - c[1], lpos[2] - c[2]};
    const double D = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    k.den_min = k.L - 2.46;                                  // ||M||_F <= sqrt(6) = 2.4495
    k.usable = (k.L > 0) && fin_d(k.L + D + r0) && (r0 >= 0) && (D > 0) && (r0 <= 0.3 * D) && (k.den_min > 0.05 * k.L);
    k.theta = 0; k.qmin = 0; k.polar = true;
    for (int i = 0; i < 3; ++i) { k.tc[i] = 0; k.a1[i] = 0; k.a2[i] = 0; }
    if (!k.usable) return k;
    for (int i = 0; i < 3; ++i) k.tc[i] = w[i] / D;
    k.theta = asin(r0 / D) * 1.001 + 2.0e-6;                // (+ the float chain's toL against the exact direction)
    k.qmin = sqrt(k.tc[0] * k.tc[0] + k.tc[1] * k.tc[1]) - k.theta * 1.01;
    k.polar = !(k.qmin >= 0.1);
    // a1, a2 across tc
    const int m = (fabs(k.tc[0]) <= fabs(k.tc[1]) && fabs(k.tc[0]) <= fabs(k.tc[2])) ? 0 : (fabs(k.tc[1]) <= fabs(k.tc[2]) ? 1 : 2);
    for (int i = 0; i < 3; ++i) k.a1[i] = -k.tc[m] * k.tc[i] + (i == m ? 1.0 : 0.0);
    const double l1 = sqrt(k.a1[0] * k.a1[0] + k.a1[1] * k.a1[1] + k.a1[2] * k.a1[2]);
    for (int i = 0; i < 3; ++i) k.a1[i] /= l1;
    k.a2[0] = k.tc[1] * k.a1[2] - k.tc[2] * k.a1[1];
    k.a2[1] = k.tc[2] * k.a1[0] - k.tc[0] * k.a1[2];
    k.a2[2] = k.tc[0] * k.a1[1] - k.tc[1] * k.a1[0];
    return k;
}
// from the largest sampled s (or anything, for a polar cone) to the slope; -1: no bound
RT_HDI double cone_slope_hd(const ConeOfStarts &k, double s_max_sampled)
{
    if (!k.usable) return -1.0;
    double s_sup;
    if (k.polar) {
        s_sup = 2.46 / k.den_min;
    } else {
        if (!(s_max_sampled == s_max_sampled)) return -1.0;
        const double G = (4.0 / k.qmin + 3.0) * 1.05 * (1.0 / k.den_min + 2.46 / (k.den_min * k.den_min));
        s_sup = s_max_sampled + G * (0.12 * k.theta);
    }
    s_sup *= 1.00001;
    if (!(s_sup < 0.45)) return -1.0;
    const double snw = s_sup * (double)RT_SPREAD_MUL + (double)RT_SPREAD_ADD;
    return snw / sqrt(fmax(1.0 - snw * snw, 0.05)) * 1.000001;
}

// s, sigma_max(M E), ||M||_F and M itself at t = normalise(l.pos - start): for the tests of the bound
double rt_beam_sine_at_start(const double lpos[3], const double start[3], double *sigma, double *frob, double m9[9])
{
    double L, u[3], e1[3], e2[3];
    light_frame_hd(lpos, &L, u, e1, e2);
    double t[3] = {lpos[0] - start[0], lpos[1] - start[1], lpos[2] - start[2]};
    const double tl = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int i = 0; i < 3; ++i) t[i] /= tl;
    if (m9) {
        const double c = t[2], sn = sqrt(t[0] * t[0] + t[1] * t[1]), ax = -t[1] / sn, ay = t[0] / sn, omc = 1.0 - c;
        const double M[9] = {c + ax * ax, ax * ay * omc, -ay * sn, ax * ay * omc, c + ay * ay * omc, -ax * sn, -ay * sn, ax * sn, c};
        for (int i = 0; i < 9; ++i) m9[i] = M[i];
    }
    return beam_sine_hd(t, e1, e2, L, sigma, frob);
}

double rt_sphere_beam_slope(const double lpos[3], const double c[3], double r0)
{
    const ConeOfStarts k = cone_of_starts_hd(lpos, c, r0);
    double smax = 0;
    if (k.usable && !k.polar) {
        for (int i = 0; i < RT_CONE_SAMPLES; ++i) {
            double t[3], sig, fro;
            cone_sample_hd(i, k.tc, k.a1, k.a2, k.theta, t);
            const double s = beam_sine_hd(t, k.e1, k.e2, k.L, &sig, &fro);
            if (!(s == s)) return -1.0;
            smax = s > smax ? s : smax;
        }
    }
    return cone_slope_hd(k, smax);
}

// the k-th of the 14 points of a sphere at which the slope is sampled: the six axis points, the eight diagonals
RT_HDI void occluder_probe_dir(int k, double d[3])
{
    if (k < 6) {
        d[0] = d[1] = d[2] = 0.0;
        d[k >> 1] = (k & 1) ? -1.0 : 1.0;
    } else {
        const int b = k - 6;
        const double s = 0.57735026918962576;
        d[0] = (b & 4) ? -s : s;
        d[1] = (b & 2) ? -s : s;
        d[2] = (b & 1) ? -s : s;
    }
}

RT_HDI bool table_entry_finite(float4 e) { return (e.x - e.x == 0.f) && (e.y - e.y == 0.f) && (e.z - e.z == 0.f) && (e.w - e.w == 0.f) && e.w >= 0.f; }

// Can a shadow ray leaving sphere S (centre c, ball radius r0 around it) within slope kcap of u hit entry T at all: the
// kernel's member test (beam_member_test) in binary64, margins a little wider. `key`: how far T reaches across S's axis.
RT_HDI bool occluder_member_hd(float4 T, const double c[3], const double u[3], double r0, double kcap, double *key)
{
    const double v[3] = {T.x - c[0], T.y - c[1], T.z - c[2]};
    const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], sa = v[0] * u[0] + v[1] * u[1] + v[2] * u[2];
    const double d2 = fmax(vv - sa * sa, 0.0);
    const double pad = (double)RT_PAD_REL * (vv + r0 * r0) * 1.001 + (double)RT_PAD_ABS * 1.001;
    const double rc = sqrt((double)T.w + pad) * 1.0002;
    const double reach = sa + rc + r0;
    const double rad = kcap * fmax(reach, 0.0) + r0 + rc;
    *key = sqrt(d2) - sqrt((double)T.w);
    return reach >= 0 && d2 <= rad * rad * 1.001;
}

void rt_build_occluder_lists(const float4 *tab, int n, const float lpos_f[3], std::vector<RtCandHdr> &hdr, std::vector<float4> &ent)
{
    hdr.assign((size_t)n, RtCandHdr{0, -1, 0.f, -1.f});
    ent.clear();
    const double lpos[3] = {lpos_f[0], lpos_f[1], lpos_f[2]};
    const double L = std::sqrt(lpos[0] * lpos[0] + lpos[1] * lpos[1] + lpos[2] * lpos[2]);
    bool all_finite = L > 0 && std::isfinite(L);
    for (int i = 0; i < n && all_finite; ++i) all_finite = table_entry_finite(tab[i]);   // a non-finite entry can return anything to the exact test: no lists
    if (all_finite) {
        const double u[3] = {lpos[0] / L, lpos[1] / L, lpos[2] / L};
        std::vector<std::pair<double, int>> keep;
        for (int si = 0; si < n; ++si) {
            const double c[3] = {tab[si].x, tab[si].y, tab[si].z}, R = std::sqrt((double)tab[si].w);
            double kmax = 0;
            bool usable = true;
            for (int k = 0; k < 14 && usable; ++k) {
                double d[3];
                occluder_probe_dir(k, d);
                const double st[3] = {c[0] + R * d[0], c[1] + R * d[1], c[2] + R * d[2]};
                const double kk = light_beam_slope_hd(lpos, st);
                if (!(kk == kk)) usable = false;
                kmax = std::max(kmax, kk);
            }
            const double r0 = R * 1.001 + 1.0e-3;
            const double kbeam = rt_sphere_beam_slope(lpos, c, r0);   // the slope the kernel will use for groups on this sphere
            hdr[si].kbeam = kbeam > 0 ? (float)kbeam : -1.f;           // (-1: it forms its own, and checks it against kcap)
            if (!usable && !(kbeam > 0)) continue;
            const double kcap = kbeam > 0 ? (double)(float)kbeam * 1.001 : kmax * 1.15 + 1.0e-4;
            keep.clear();
            for (int ti = 0; ti < n; ++ti) {
                double key;
                if (occluder_member_hd(tab[ti], c, u, r0, kcap, &key)) keep.push_back({key, ti});
            }
            if ((int)keep.size() > RT_CAND_CAP) continue;
            std::sort(keep.begin(), keep.end());   // likeliest occluder first: reaches farthest across S's own axis
            hdr[si].offset = (int)ent.size();
            hdr[si].count = (int)keep.size();
            hdr[si].kcap = (float)(kcap * 0.9999);
            for (const auto &kv : keep) ent.push_back(tab[kv.second]);
        }
    }
    // A wave reads whole steps of 64 entries from a list's offset on (build_list_cand): up to 63 entries past the end of
    // the list -- for the last list, past the end of the array. Pad by a whole step (a first version rounded the TOTAL up
    // to a multiple of 64, which pads nothing when it already is one: a read past the allocation, caught by the soak).
    ent.resize(ent.size() + 64, make_float4(0.f, 0.f, 0.f, 0.f));
}

// The same lists built on the DEVICE, one wave per sphere: hdr[s] = {s * RT_CAND_CAP, count, kcap}, the entries in a slot
// of RT_CAND_CAP per sphere (whole-step reads stay inside a slot), in table order -- the kernel re-orders a tile's
// survivors itself. A change of the sphere list or of a light then costs a launch of n waves instead of an O(n^2) loop
// on the host (20 ms at 1024 spheres, 0.3 s at 4096).
__global__ __launch_bounds__(64) void rt_occluder_lists_kernel(const float4 *__restrict__ tab, int n, float lx, float ly, float lz,
                                                               RtCandHdr *__restrict__ hdr, float4 *__restrict__ ent)
{
    const int si = blockIdx.x, lane = threadIdx.x;
    const double lpos[3] = {lx, ly, lz};
    const double L = sqrt(lpos[0] * lpos[0] + lpos[1] * lpos[1] + lpos[2] * lpos[2]);
    const float4 S = tab[si];
    const double c[3] = {S.x, S.y, S.z}, R = sqrt((double)S.w), r0 = R * 1.001 + 1.0e-3;
    bool fin_all = (L > 0) && fin_d(L);                 // every entry of the table finite (a non-finite one: no lists at all)
    const bool s_fin = table_entry_finite(S);
    // the slope of every group on this sphere: the cone of starts, its 217 sample directions four to a lane
    const ConeOfStarts cone = cone_of_starts_hd(lpos, c, s_fin ? r0 : NAN);
    double smax = 0;
    bool s_nan = false;
    if (cone.usable && !cone.polar) {
        for (int i = lane; i < RT_CONE_SAMPLES; i += 64) {
            double t[3], sig, fro;
            cone_sample_hd(i, cone.tc, cone.a1, cone.a2, cone.theta, t);
            const double s = beam_sine_hd(t, cone.e1, cone.e2, cone.L, &sig, &fro);
            s_nan = s_nan || !(s == s);
            smax = s > smax ? s : smax;
        }
        s_nan = __any(s_nan);
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(smax, off);
            smax = (o > smax) ? o : smax;
        }
    }
    const double kbeam = cone_slope_hd(cone, s_nan ? NAN : smax);
    // without one: 1.15 x the largest slope the kernel's own bound yields at 14 points of the sphere (it checks)
    double k = 0;
    if (lane < 14) {
        double d[3];
        occluder_probe_dir(lane, d);
        const double st[3] = {c[0] + R * d[0], c[1] + R * d[1], c[2] + R * d[2]};
        k = light_beam_slope_hd(lpos, st);
    }
    const bool probes_ok = !__any(k != k);
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(k, off);
        k = (o > k) ? o : k;
    }
    const bool usable = s_fin && (kbeam > 0 || probes_ok);
    const double kcap = kbeam > 0 ? (double)(float)kbeam * 1.001 : k * 1.15 + 1.0e-4;
    const double u[3] = {lpos[0] / L, lpos[1] / L, lpos[2] / L};
    int count = 0;
    for (int base = 0; base < n; base += 64) {
        const int ti = base + lane;
        const float4 T = tab[ti < n ? ti : n - 1];
        const bool fin = table_entry_finite(T);
        if (__any(ti < n && !fin)) fin_all = false;
        double key;
        const bool keep = usable && ti < n && fin && occluder_member_hd(T, c, u, r0, kcap, &key);
        const unsigned long long m = __ballot(keep);
        const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < RT_CAND_CAP) ent[(size_t)si * RT_CAND_CAP + pos] = T;
        count += __popcll(m);
    }
    if (lane == 0) {
        RtCandHdr h;
        h.offset = si * RT_CAND_CAP;
        h.count = (fin_all && usable && count <= RT_CAND_CAP) ? count : -1;
        h.kcap = (float)(kcap * 0.9999);
        h.kbeam = (fin_all && kbeam > 0) ? (float)kbeam : -1.f;
        hdr[si] = h;
    }
}

hipError_t rt_occluder_lists_launch(const float4 *tab, int n, const float lpos[3], RtCandHdr *hdr, float4 *ent, hipStream_t stream)
{
    if (n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rt_occluder_lists_kernel, dim3((unsigned)n), dim3(64), 0, stream, tab, n, lpos[0], lpos[1], lpos[2], hdr, ent);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Eye cones. Every primary ray starts at the same point O, so the table is ordered by the
// direction of the centres as seen from O (octahedral map, 2-D Morton) and cut into blocks of
// RT_BLOCK: cones from O that a tile's thin beam meets far less often than the cubes of the
// 3-D order. theta of a block: max over members of (angle(axis, dir_j) + ext_j), plus margin,
// the angles being taken from the ROUNDED axis the kernel will use.
// Layout of the result: [n_pad float4 entries][2 float4 per block][n_pad ints: list positions].
// ---------------------------------------------------------------------------
// What a block's members reduce to: sum of the bounded members' directions, whether all are
// bounded; then, given the axis, the largest angle + extension. Host: plain loops.
void rt_build_eye_cones_host(const float4 *tab, int n, const float org[3], float4 *sorted, float4 *blocks, int *orig)
{
    const int n_pad = (n + 63) & ~63;
    std::vector<ConeEnt> ent((size_t)n);
    std::vector<std::pair<unsigned long long, int>> keys((size_t)n);
    for (int i = 0; i < n; ++i) {
        ent[i] = cone_entry(tab[i], org);
        keys[i] = {((unsigned long long)cone_key(ent[i]) << 32) | (unsigned)i, i};
    }
    std::sort(keys.begin(), keys.end());
    for (int i = 0; i < n_pad; ++i) {
        sorted[i] = i < n ? tab[keys[i].second] : make_float4(0.f, 0.f, 0.f, 0.f);
        orig[i] = i < n ? keys[i].second : 0x7fffffff;
    }
    for (int b = 0; b < n_pad / RT_BLOCK; ++b) {
        const int i0 = b * RT_BLOCK, i1 = std::min(n, i0 + RT_BLOCK);
        if (i0 >= n) {   // padding block: nothing in it, never examined
            blocks[2 * b] = make_float4(0.f, 0.f, 0.f, 0.f);
            blocks[2 * b + 1] = make_float4(0.f, -1.f, 0.f, 0.f);
            continue;
        }
        bool bounded = true;
        double m[3] = {0, 0, 0};
        for (int i = i0; i < i1; ++i) {
            const ConeEnt &e = ent[keys[i].second];
            bounded = bounded && e.bounded;
            if (e.bounded) for (int k = 0; k < 3; ++k) m[k] += e.dir[k];
        }
        const double ml = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        bounded = bounded && ml > 1e-6;
        float af[3] = {0, 0, 0};
        double theta = 0;
        if (bounded) {
            for (int k = 0; k < 3; ++k) af[k] = (float)(m[k] / ml);
            const double al = std::sqrt((double)af[0] * af[0] + (double)af[1] * af[1] + (double)af[2] * af[2]);
            for (int i = i0; i < i1; ++i) {
                const ConeEnt &e = ent[keys[i].second];
                double c = (af[0] * e.dir[0] + af[1] * e.dir[1] + af[2] * e.dir[2]) / al;
                c = std::min(1.0, std::max(-1.0, c));
                theta = std::max(theta, std::acos(c) + e.ext);
            }
            theta += 2.0e-3;
            if (!(theta < 2.9)) bounded = false;   // theta + the beam's own angle must stay below pi
        }
        if (!bounded) {
            blocks[2 * b] = make_float4(0.f, 0.f, 0.f, 0.f);
            blocks[2 * b + 1] = make_float4(0.f, 1.f, 0.f, 0.f);
            continue;
        }
        blocks[2 * b] = make_float4(af[0], af[1], af[2], (float)std::cos(theta));
        blocks[2 * b + 1] = make_float4((float)std::sin(theta), 0.f, 0.f, 0.f);
    }
}

// The same table built by ONE workgroup (any multiple of 64 threads up to 1024): keys into LDS,
// bitonic sort, entries out, then 16 lanes per block reduce their members' directions and
// angles with shuffles. 1024 threads when the build is a node in front of a graph's passes;
// 256 when it runs on the scene's table stream beside the previous frame's kernel (four waves
// find room on a CU that the frame kernel has filled to 24 of its 32 wave slots).
__global__ __launch_bounds__(1024) void rt_eye_cones_kernel(const float4 *__restrict__ tab, int n, float ox, float oy, float oz,
                                                            float4 *__restrict__ out)
{
    extern __shared__ unsigned long long keys[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n_pad = (n + 63) & ~63, nb = n_pad / RT_BLOCK;
    int P = 64;
    while (P < n_pad) P <<= 1;
    const float org[3] = {ox, oy, oz};
    for (int i = tid; i < P; i += nt) {
        unsigned long long key = ~0ull;   // padding sorts last
        if (i < n) key = ((unsigned long long)cone_key(cone_entry(tab[i], org)) << 32) | (unsigned)i;
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += nt) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
    }
    float4 *sorted = out, *blocks = out + n_pad;
    int *orig = reinterpret_cast<int *>(out + n_pad + 2 * nb);
    for (int i = tid; i < n_pad; i += nt) {
        const unsigned long long key = keys[i];
        const bool have = key != ~0ull;
        const int idx = (int)(unsigned)(key & 0xffffffffu);
        sorted[i] = have ? tab[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        orig[i] = have ? idx : 0x7fffffff;
    }
    // blocks: RT_BLOCK consecutive lanes per block (RT_BLOCK divides 64: a group never straddles a wave)
    const int sub = tid % RT_BLOCK, grp = tid / RT_BLOCK;
    for (int b0 = 0; b0 < nb; b0 += nt / RT_BLOCK) {
        const int b = b0 + grp;
        const bool live = b < nb;                      // whole groups leave together; shuffles stay inside a group
        const int i = (live ? b : 0) * RT_BLOCK + sub;
        const unsigned long long key = keys[i];
        const bool have = live && key != ~0ull;
        ConeEnt e;
        e.bounded = false; e.ext = 0; e.dir[0] = e.dir[1] = e.dir[2] = 0;
        if (have) e = cone_entry(tab[(int)(unsigned)(key & 0xffffffffu)], org);
        int all_bounded = (!have || e.bounded) ? 1 : 0, any = have ? 1 : 0;
        double m0 = (have && e.bounded) ? e.dir[0] : 0.0, m1 = (have && e.bounded) ? e.dir[1] : 0.0,
               m2 = (have && e.bounded) ? e.dir[2] : 0.0;
        for (int o = RT_BLOCK / 2; o > 0; o >>= 1) {
            m0 += __shfl_xor(m0, o, RT_BLOCK);
            m1 += __shfl_xor(m1, o, RT_BLOCK);
            m2 += __shfl_xor(m2, o, RT_BLOCK);
            all_bounded &= __shfl_xor(all_bounded, o, RT_BLOCK);
            any |= __shfl_xor(any, o, RT_BLOCK);
        }
        const double ml = sqrt(m0 * m0 + m1 * m1 + m2 * m2);
        bool bounded = all_bounded && ml > 1e-6;
        float af[3] = {0.f, 0.f, 0.f};
        double theta = 0;
        if (bounded) {
            af[0] = (float)(m0 / ml); af[1] = (float)(m1 / ml); af[2] = (float)(m2 / ml);
            const double al = sqrt((double)af[0] * af[0] + (double)af[1] * af[1] + (double)af[2] * af[2]);
            if (have) {
                double c = (af[0] * e.dir[0] + af[1] * e.dir[1] + af[2] * e.dir[2]) / al;
                c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
                theta = acos(c) + e.ext;
            }
        }
        for (int o = RT_BLOCK / 2; o > 0; o >>= 1) {
            const double t = __shfl_xor(theta, o, RT_BLOCK);
            theta = (t > theta || t != t) ? t : theta;
        }
        theta += 2.0e-3;
        if (!(theta < 2.9)) bounded = false;
        if (live && sub == 0) {
            if (!any) {                       // padding block: never examined
                blocks[2 * b] = make_float4(0.f, 0.f, 0.f, 0.f);
                blocks[2 * b + 1] = make_float4(0.f, -1.f, 0.f, 0.f);
            } else if (!bounded) {            // always examined
                blocks[2 * b] = make_float4(0.f, 0.f, 0.f, 0.f);
                blocks[2 * b + 1] = make_float4(0.f, 1.f, 0.f, 0.f);
            } else {
                blocks[2 * b] = make_float4(af[0], af[1], af[2], (float)cos(theta));
                blocks[2 * b + 1] = make_float4((float)sin(theta), 0.f, 0.f, 0.f);
            }
        }
    }
}

// float4 units of an eye-cone table for n spheres
size_t rt_eye_cones_size(int n)
{
    const size_t n_pad = ((size_t)n + 63) & ~(size_t)63, nb = n_pad / RT_BLOCK;
    return n_pad + 2 * nb + (n_pad + 3) / 4;
}

void rt_eye_cones_kernel_config(int n, int threads, const void **func, dim3 *grid, dim3 *block, unsigned *lds_bytes)
{
    const int n_pad = (n + 63) & ~63;
    int P = 64;
    while (P < n_pad) P <<= 1;
    *func = (const void *)rt_eye_cones_kernel;
    *grid = dim3(1);
    *block = dim3(threads);
    *lds_bytes = (unsigned)P * (unsigned)sizeof(unsigned long long);
}

hipError_t rt_eye_cones_launch(const float4 *tab, int n, const float org[3], float4 *out, int threads, hipStream_t stream)
{
    if (n < 1 || n > RT_EYE_DEVICE_MAX || threads < 64 || threads > 1024 || threads % 64) return hipErrorInvalidValue;
    const void *func;
    dim3 grid, block;
    unsigned lds;
    rt_eye_cones_kernel_config(n, threads, &func, &grid, &block, &lds);
    hipLaunchKernelGGL(rt_eye_cones_kernel, grid, block, lds, stream, tab, n, org[0], org[1], org[2], out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// tile order of a launch: the blocks with the longest tiles first
// ---------------------------------------------------------------------------
// A launch ends when its slowest wave does. Tiles take 6 ... 90 us (a tile that walks the shadow samples of three
// lights for several groups of pixels against one that is fully occluded), and in row-major order the expensive ones
// of the last rows start last: the SIMDs drain for tens of microseconds -- 45 us per launch at C3, 12 % of a frame
// and half of an eighth of it. So the frame is cut into blocks of 16 x 16 tiles, the blocks are started in the order
// of their longest tile (wave durations of the previous launch, RtFrameConsts::tile_cost), tiles row-major inside a
// block. Blocks rather than single tiles: neighbouring tiles keep running together (textures, tables in cache) --
// on the GPU the block order beats the per-tile order even with durations of the very same view, 0.340 against 0.356 ms
// at C3 -- and a camera step, which moves the expensive tiles by several tiles and makes a per-tile order worthless,
// costs a block order 1 % (tools/tile_order_probe.py).
#define RT_ORDER_BUCKETS 256
#define RT_ORDER_BLOCK 16
// eight buckets per octave of the duration (exponent and three mantissa bits of the value as a float)
__device__ __forceinline__ unsigned tile_cost_bucket(unsigned c)
{
    const unsigned b = __float_as_uint((float)c) >> 20;      // 0 for c = 0; 127 << 3 for c = 1
    return b >= (127u << 3) ? min(b - (127u << 3), (unsigned)RT_ORDER_BUCKETS - 1u) : 0u;
}

// The longest tile of every block: a workgroup per block, a thread per tile.
__global__ __launch_bounds__(256) void rt_tile_order_keys_kernel(const unsigned *__restrict__ cost, unsigned *__restrict__ key, int nbx,
                                                                 int tiles_x, int tiles_y)
{
    __shared__ unsigned m;
    if (threadIdx.x == 0) m = 0;
    __syncthreads();
    const int b = blockIdx.x;
    const int tx = (b % nbx) * RT_ORDER_BLOCK + (int)(threadIdx.x % RT_ORDER_BLOCK), ty = (b / nbx) * RT_ORDER_BLOCK + (int)(threadIdx.x / RT_ORDER_BLOCK);
    unsigned c = (tx < tiles_x && ty < tiles_y) ? cost[ty * tiles_x + tx] : 0u;
    for (int o = 32; o > 0; o >>= 1) c = max(c, (unsigned)__shfl_xor((int)c, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&m, c);
    __syncthreads();
    if (threadIdx.x == 0) key[b] = m;
}

// Inclusive prefix sums of a[0..n) in LDS (n <= 4 * 1024), all 1024 threads of the workgroup; tmp: as large as a.
__device__ void scan_inclusive_lds(unsigned *a, unsigned *tmp, int n)
{
    unsigned *src = a, *dst = tmp;
    for (int o = 1; o < n; o <<= 1) {
        for (int i = threadIdx.x; i < n; i += 1024) dst[i] = src[i] + (i >= o ? src[i - o] : 0u);
        __syncthreads();
        unsigned *t = src; src = dst; dst = t;
    }
    if (src != a) {
        for (int i = threadIdx.x; i < n; i += 1024) a[i] = src[i];
        __syncthreads();
    }
}

// Sorts the blocks by their keys, longest first (counting sort, one workgroup), and writes where each block's tiles
// start in the order.
__global__ __launch_bounds__(1024) void rt_tile_order_sort_kernel(const unsigned *__restrict__ key, unsigned *__restrict__ start, int nbx,
                                                                  int nby, int tiles_x, int tiles_y)
{
    __shared__ unsigned at_rank[RT_TILE_ORDER_MAX_BLOCKS];  // the block at each place of the order
    __shared__ unsigned sizes[RT_TILE_ORDER_MAX_BLOCKS];    // tiles of the block at each place, then their prefix sums
    __shared__ unsigned tmp[RT_TILE_ORDER_MAX_BLOCKS];
    __shared__ unsigned hist[RT_ORDER_BUCKETS], hsum[RT_ORDER_BUCKETS], htmp[RT_ORDER_BUCKETS];
    const int tid = threadIdx.x;
    const int nb = nbx * nby;
    for (int b = tid; b < RT_ORDER_BUCKETS; b += 1024) hist[b] = 0;
    __syncthreads();
    for (int b = tid; b < nb; b += 1024) atomicAdd(&hist[tile_cost_bucket(key[b])], 1u);
    __syncthreads();
    // where each bucket starts, the longest durations first: prefix sums over the buckets in descending order
    for (int k = tid; k < RT_ORDER_BUCKETS; k += 1024) hsum[k] = hist[RT_ORDER_BUCKETS - 1 - k];
    __syncthreads();
    scan_inclusive_lds(hsum, htmp, RT_ORDER_BUCKETS);
    for (int k = tid; k < RT_ORDER_BUCKETS; k += 1024) hist[RT_ORDER_BUCKETS - 1 - k] = hsum[k] - hist[RT_ORDER_BUCKETS - 1 - k];
    __syncthreads();
    for (int b = tid; b < nb; b += 1024) at_rank[atomicAdd(&hist[tile_cost_bucket(key[b])], 1u)] = (unsigned)b;
    __syncthreads();
    // where each block's tiles start (blocks at the right and lower edge are smaller)
    for (int r = tid; r < nb; r += 1024) {
        const unsigned b = at_rank[r];
        const int bx = (int)(b % (unsigned)nbx), by = (int)(b / (unsigned)nbx);
        sizes[r] = (unsigned)(min(RT_ORDER_BLOCK, tiles_x - bx * RT_ORDER_BLOCK) * min(RT_ORDER_BLOCK, tiles_y - by * RT_ORDER_BLOCK));
    }
    __syncthreads();
    scan_inclusive_lds(sizes, tmp, nb);
    for (int r = tid; r < nb; r += 1024) start[at_rank[r]] = r ? sizes[r - 1] : 0u;
}

// perm[] from the blocks' starting places: a thread per tile, tiles row-major inside their block
__global__ __launch_bounds__(256) void rt_tile_order_expand_kernel(const unsigned *__restrict__ start, unsigned *__restrict__ perm, int n,
                                                                   int tiles_x, int nbx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned ty = (unsigned)i / (unsigned)tiles_x, tx = (unsigned)i - ty * (unsigned)tiles_x;
    const unsigned bx = tx / RT_ORDER_BLOCK, by = ty / RT_ORDER_BLOCK;
    const unsigned bw = (unsigned)min(RT_ORDER_BLOCK, tiles_x - (int)bx * RT_ORDER_BLOCK);
    perm[start[by * nbx + bx] + (ty % RT_ORDER_BLOCK) * bw + (tx % RT_ORDER_BLOCK)] = (ty << 16) | tx;
}

hipError_t rt_tile_order_launch(const unsigned *cost, unsigned *key, unsigned *start, unsigned *perm, int tiles_x, int tiles_y,
                                hipStream_t stream)
{
    const int nbx = (tiles_x + RT_ORDER_BLOCK - 1) / RT_ORDER_BLOCK, nby = (tiles_y + RT_ORDER_BLOCK - 1) / RT_ORDER_BLOCK;
    const int n = tiles_x * tiles_y;
    hipLaunchKernelGGL(rt_tile_order_keys_kernel, dim3(nbx * nby), dim3(256), 0, stream, cost, key, nbx, tiles_x, tiles_y);
    hipLaunchKernelGGL(rt_tile_order_sort_kernel, dim3(1), dim3(1024), 0, stream, key, start, nbx, nby, tiles_x, tiles_y);
    hipLaunchKernelGGL(rt_tile_order_expand_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, start, perm, n, tiles_x, nbx);
    return hipGetLastError();
}

// The same three launches as graph kernel nodes (rt_graph.cpp): functions and geometries; the arguments are
// keys: (const unsigned *cost, unsigned *key, int nbx, int tiles_x, int tiles_y)
// sort: (const unsigned *key, unsigned *start, int nbx, int nby, int tiles_x, int tiles_y)
// expand: (const unsigned *start, unsigned *perm, int n, int tiles_x, int nbx)
void rt_tile_order_kernel_configs(int tiles_x, int tiles_y, const void *func[3], dim3 grid[3], dim3 block[3])
{
    const int nbx = (tiles_x + RT_ORDER_BLOCK - 1) / RT_ORDER_BLOCK, nby = (tiles_y + RT_ORDER_BLOCK - 1) / RT_ORDER_BLOCK;
    func[0] = (const void *)rt_tile_order_keys_kernel;   grid[0] = dim3(nbx * nby); block[0] = dim3(256);
    func[1] = (const void *)rt_tile_order_sort_kernel;   grid[1] = dim3(1); block[1] = dim3(1024);
    func[2] = (const void *)rt_tile_order_expand_kernel; grid[2] = dim3((tiles_x * tiles_y + 255) / 256); block[2] = dim3(256);
}
