// rt_engine.cpp -- host side of the C ABI (include/rt_engine.h): device-resident
// scene, frame-uniform hoisting, kernel launch, the rayTrace launch shim and the
// memManager surface.
//
// Reference interfaces replaced here:
//   memManager / check_cuda      /root/reference/memManager.h:11-18, memManager.cpp:3-22
//   rayTrace<<<...>>> launch     /root/reference/kernel.cu:1615, 1780-1783
//   object / sprite / skybox     /root/reference/kernel.cu:1116-1244, sprite.h:11-47
//
// There is no CPU fallback: every render entry point needs a gfx950 device and
// fails with RT_ERR_NO_DEVICE / RT_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/rt_engine.h"
#include "rt_device.h"
#include "rt_internal.h"
#include "rt_math.h"

// launchers defined in rt_kernels.hip
extern "C" hipError_t rt_dev_launch_trace(const RtFrameConsts *fc, const float4 *spheres, int tile_w,
                                          int cull, int stats, int table_in_lds, hipStream_t stream);
extern "C" hipError_t rt_dev_launch_dbg_math(int op, const float *a, const float *b, float *out, int n,
                                             hipStream_t stream);
extern "C" hipError_t rt_dev_launch_dbg_intersect(const float4 *tab, const float *rays, int n, int *hit,
                                                  float *t, hipStream_t stream);
extern "C" hipError_t rt_dev_launch_dbg_light(const RtFrameConsts *fc, const float4 *tab, const float *starts,
                                              const float *normals, int light_index, int n, float *dirs,
                                              float *bright, hipStream_t stream);

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static char g_last_error[512] = "";
static int g_soft_errors = 0;

void rt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
}

int rt_hip_fail(hipError_t e, const char *expr, const char *file, int line)
{
    rt_set_error("HIP error = %u (%s) at %s:%d '%s'", (unsigned)e, hipGetErrorString(e), file, line, expr);
    (void)hipGetLastError();   // clear the sticky error so later calls can proceed
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE : RT_ERR_HIP;
}

extern "C" const char *rt_last_error(void) { return g_last_error; }
extern "C" int rt_abi_version(void) { return RT_ABI_VERSION; }
extern "C" int rt_set_soft_errors(int on)
{
    const int old = g_soft_errors;
    g_soft_errors = on ? 1 : 0;
    return old;
}
extern "C" int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// check_cuda, memManager.cpp:3-11
extern "C" void rt_check(int err, const char *expr, const char *file, int line)
{
    if (err) {
        fprintf(stderr, "HIP error = %u at %s:%d '%s' \n", (unsigned)err, file, line, expr);
        rt_set_error("HIP error = %u at %s:%d '%s'", (unsigned)err, file, line, expr);
        if (g_soft_errors) return;
        (void)hipDeviceReset();
        exit(99);
    }
}
#define checkHipErrors(val) rt_check((int)(val), #val, __FILE__, __LINE__)

// memManager::operator new / delete, memManager.cpp:12-22
extern "C" void *rt_managed_alloc(size_t len)
{
    void *ptr = nullptr;
    checkHipErrors(hipMallocManaged(&ptr, len ? len : 1));
    checkHipErrors(hipDeviceSynchronize());
    return ptr;
}
extern "C" void rt_managed_free(void *ptr)
{
    if (!ptr) return;
    checkHipErrors(hipDeviceSynchronize());
    (void)hipFree(ptr);
}

// ---------------------------------------------------------------------------
// device-resident scene
// ---------------------------------------------------------------------------
struct rt_scene {
    float4 *d_spheres = nullptr;     // [n] list order | [n_pad] Morton order | [n_blocks] block bounds | [n_pad] ints
    int n_spheres = 0, cap_spheres = 0;
    int n_blocks = 0;
    std::vector<float4> h_prev;      // what was uploaded last (skip identical re-mirrors)
    float4 *h_stage = nullptr;   // pinned staging for asynchronous re-uploads
    int cap_stage = 0;
    hipEvent_t stage_done = nullptr;   // the last upload out of h_stage
    bool stage_busy = false;
    float *d_tex[3] = {nullptr, nullptr, nullptr};
    int tex_w = 0, tex_h = 0;
    float *d_sky[3] = {nullptr, nullptr, nullptr};
    int sky_w = 0, sky_h = 0;
    float sky_c[3] = {0, 0, 0};
    float sky_radius = 0;        // the sphere's `radius` field (already r*r)
    bool have_sky = false;
    rt_light lights[RT_MAX_LIGHTS];
    int n_lights = 0;
    RtPlaneDev *d_planes = nullptr;
    RtCubeDev *d_cubes = nullptr;
    int n_planes = 0, n_cubes = 0;
    RtTriDev *d_tris = nullptr;
    RtBoxDev *d_boxes = nullptr;
    int *d_tri_idx = nullptr;
    float *d_box_spheres = nullptr;
    float *d_tri9 = nullptr;
    int n_boxes = 0, n_tris = 0, mesh_has_normals = 0;
    // per-light column blocks (see RtFrameConsts::lsorted): one allocation, rebuilt when the
    // sphere list or a light's position changes
    float4 *d_light_tabs = nullptr;
    size_t cap_light_tabs = 0;           // float4 units
    unsigned long long sphere_gen = 0;   // bumped whenever the mirrored sphere list changes
    unsigned long long ltab_gen = ~0ull; // sphere_gen the light tables were built from
    int ltab_n_lights = 0;
    float ltab_axis[RT_MAX_LIGHTS][3];   // axis each table was built for
    bool ltab_valid[RT_MAX_LIGHTS] = {};
    // eye cones for the primary rays (see RtFrameConsts::csorted): rebuilt when the sphere
    // list or the ray origin changes
    float4 *d_cone_tab = nullptr;
    size_t cap_cone_tab = 0;             // float4 units
    unsigned long long cone_gen = ~0ull;
    float cone_org[3] = {0, 0, 0};
    bool cone_valid = false;
};

// Where the kernel reads the sphere table from. Default: global memory (L2-resident), LDS
// holding only each wave's survivor lists -- the tile's spheres, broadcast across lanes.
// RT_TABLE_LDS=1 stages the whole table in LDS per workgroup of RT_WAVES_PER_WG waves when
// it fits (the round's first design; measured slower: 0.71 vs 0.68 ms at 1024 spheres,
// 4.5 vs 2.6 ms at 4096), RT_TABLE_LDS=0 forces the default.
static const int kMaxSpheresLds = (160 * 1024 - RT_WAVES_PER_WG * (RT_LIST_CAP * 20 + 16 * 4 + 64 * 4 + RT_BOX_CAP * 4)) / 16;
static const int kMaxSpheres = 1 << 22;
static int table_in_lds_for(int n)
{
    const char *e = getenv("RT_TABLE_LDS");   // tuning/testing override
    if (e && *e == '1') return n <= kMaxSpheresLds ? 1 : 0;
    return 0;
}

extern "C" rt_scene *rt_scene_create(void) { return new rt_scene(); }

static void free_planes(float *p[3])
{
    for (int i = 0; i < 3; ++i) {
        if (p[i]) (void)hipFree(p[i]);
        p[i] = nullptr;
    }
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    if (s->d_spheres) (void)hipFree(s->d_spheres);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->stage_done) (void)hipEventDestroy(s->stage_done);
    free_planes(s->d_tex);
    free_planes(s->d_sky);
    if (s->d_planes) (void)hipFree(s->d_planes);
    if (s->d_cubes) (void)hipFree(s->d_cubes);
    if (s->d_tris) (void)hipFree(s->d_tris);
    if (s->d_boxes) (void)hipFree(s->d_boxes);
    if (s->d_tri_idx) (void)hipFree(s->d_tri_idx);
    if (s->d_box_spheres) (void)hipFree(s->d_box_spheres);
    if (s->d_tri9) (void)hipFree(s->d_tri9);
    if (s->d_light_tabs) (void)hipFree(s->d_light_tabs);
    if (s->d_cone_tab) (void)hipFree(s->d_cone_tab);
    delete s;
}

// {cx, cy, cz, radius*radius}: the only four numbers sphere::intersect reads
// (kernel.cu:332-334); radius*radius is the same binary32 product either way.
static void pack_spheres(const rt_sphere *src, int n, float4 *dst)
{
    for (int i = 0; i < n; ++i)
        dst[i] = make_float4(src[i].orgin.x, src[i].orgin.y, src[i].orgin.z, src[i].radius * src[i].radius);
}

// Morton order of the centres (10 bits per axis over the scene's bounds), blocks of
// RT_BLOCK consecutive spheres, and for each block a sphere that contains every member
// (centre = mean of the members' centres, radius = max |c_i - centre| + R_i, rounded
// up). Spheres with non-finite data make their block unbounded (always examined).
static unsigned morton10(unsigned v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

static void build_sorted_blocks(const float4 *tab, int n, float4 *sorted, float4 *blocks, int *orig)
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
static unsigned morton16(unsigned v)
{
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

static void build_light_columns(const float4 *tab, int n, const float u_f[3], float4 *sorted, float4 *blocks)
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
// Eye cones. Every primary ray starts at the same point O, so the table is ordered by the
// direction of the centres as seen from O (octahedral map, 2-D Morton) and cut into blocks of
// RT_BLOCK: cones from O that a tile's thin beam meets far less often than the cubes of the
// 3-D order. For a beam with apex O, r0 = 1e-4, smin = 0 and slope k <= kcap the member test
// of the kernel (beam_member_test) passes only if the angle alpha between the beam axis and
// the direction to the centre satisfies sin(alpha - atan(1.00025 k)) <= (rc (1 + k) + r0) 1.00025 / |v|,
// rc = sqrt(R^2 + 4e-5 |v|^2 + 1e-3) 1.0001 -- all known here because the apex is. theta of
// a block: max over members of (angle(axis, dir_j) + asin of that bound), plus margin.
// ---------------------------------------------------------------------------
static const float kConeKcap = 0.1f;

static void oct_map(const double d[3], double *u, double *v)
{
    const double s = std::fabs(d[0]) + std::fabs(d[1]) + std::fabs(d[2]);
    double x = d[0] / s, y = d[1] / s;
    if (d[2] < 0) {
        const double ox = (1 - std::fabs(y)) * (x >= 0 ? 1 : -1), oy = (1 - std::fabs(x)) * (y >= 0 ? 1 : -1);
        x = ox; y = oy;
    }
    *u = x; *v = y;
}

static void build_eye_cones(const float4 *tab, int n, const float org[3], float4 *sorted, float4 *blocks, int *orig)
{
    const int n_pad = (n + 63) & ~63;
    const double kcap = kConeKcap, r0 = 1.0e-4;
    struct Ent { double dir[3]; double ext; bool bounded; };
    std::vector<Ent> ent((size_t)n);
    std::vector<std::pair<unsigned long long, int>> keys((size_t)n);
    for (int i = 0; i < n; ++i) {
        Ent &e = ent[i];
        const double v[3] = {(double)tab[i].x - org[0], (double)tab[i].y - org[1], (double)tab[i].z - org[2]};
        const double vv = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dist = std::sqrt(vv), w = tab[i].w;
        e.bounded = std::isfinite(vv) && std::isfinite(w) && dist > 0;
        e.ext = 0;
        if (e.bounded) {
            const double rc = std::sqrt(std::max(0.0, w) + 4.0e-5 * vv + 1.0e-3) * 1.0001;
            const double q = (rc * (1.0 + kcap) + r0) * 1.00025 * 1.001 / dist;
            if (!(q < 0.99)) e.bounded = false;   // the origin is inside or next to the (padded) sphere
            else e.ext = std::asin(q);
            for (int k = 0; k < 3; ++k) e.dir[k] = v[k] / dist;
        }
        unsigned long long key = ~0ull;   // unbounded entries go last
        if (e.bounded) {
            double ou, ov;
            oct_map(e.dir, &ou, &ov);
            const unsigned q1 = (unsigned)std::min(65535.0, std::max(0.0, (ou * 0.5 + 0.5) * 65535.0));
            const unsigned q2 = (unsigned)std::min(65535.0, std::max(0.0, (ov * 0.5 + 0.5) * 65535.0));
            key = morton16(q1) | ((unsigned long long)morton16(q2) << 1);
        }
        keys[i] = {key, i};
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
            const Ent &e = ent[keys[i].second];
            bounded = bounded && e.bounded;
            if (e.bounded) for (int k = 0; k < 3; ++k) m[k] += e.dir[k];
        }
        const double ml = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        bounded = bounded && ml > 1e-6;
        float af[3] = {0, 0, 0};
        double theta = 0;
        if (bounded) {
            for (int k = 0; k < 3; ++k) af[k] = (float)(m[k] / ml);
            // angles are taken from the ROUNDED axis the device will use
            const double al = std::sqrt((double)af[0] * af[0] + (double)af[1] * af[1] + (double)af[2] * af[2]);
            for (int i = i0; i < i1; ++i) {
                const Ent &e = ent[keys[i].second];
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

// Bring the eye cones up to date with the mirrored sphere list and the ray origin `org`
// (= RtFrameConsts::org_*). Not inside a stream capture.
int rt_scene_prepare_eye(rt_scene *s, const float org[3], hipStream_t stream)
{
    const int n = s->n_spheres;
    const int n_pad = (n + 63) & ~63, nb = n_pad / RT_BLOCK;
    const bool want = n >= 64 && s->h_prev.size() == (size_t)n && !getenv("RT_NO_EYE_CONES") &&
                      std::isfinite(org[0]) && std::isfinite(org[1]) && std::isfinite(org[2]);
    if (!want) {
        s->cone_valid = false;
        s->cone_gen = ~0ull;
        return RT_OK;
    }
    if (s->cone_valid && s->cone_gen == s->sphere_gen && memcmp(org, s->cone_org, sizeof s->cone_org) == 0) return RT_OK;
    const size_t total = (size_t)n_pad + 2 * (size_t)nb + ((size_t)n_pad + 3) / 4;   // float4 units
    if (total > s->cap_cone_tab) {
        if (s->d_cone_tab) RT_HIP(hipFree(s->d_cone_tab));
        s->d_cone_tab = nullptr;
        s->cap_cone_tab = 0;
        RT_HIP(hipMalloc((void **)&s->d_cone_tab, sizeof(float4) * total));
        s->cap_cone_tab = total;
    }
    // frames still in flight (another stream, a replaying graph) may be reading the old table
    RT_HIP(hipDeviceSynchronize());
    std::vector<float4> h(total);
    build_eye_cones(s->h_prev.data(), n, org, h.data(), h.data() + n_pad, reinterpret_cast<int *>(h.data() + n_pad + 2 * nb));
    RT_HIP(hipMemcpyAsync(s->d_cone_tab, h.data(), sizeof(float4) * total, hipMemcpyHostToDevice, stream));
    RT_HIP(hipStreamSynchronize(stream));   // `h` goes out of scope; only when the eye or the scene moved
    s->cone_gen = s->sphere_gen;
    memcpy(s->cone_org, org, sizeof s->cone_org);
    s->cone_valid = true;
    return RT_OK;
}

// Bring the per-light tables up to date with the mirrored sphere list and the lights'
// positions. Not inside a stream capture (rt_graph_capture calls it first).
int rt_scene_prepare_lights(rt_scene *s, hipStream_t stream)
{
    const int n = s->n_spheres;
    const int n_pad = (n + 63) & ~63, nb = n_pad / RT_BLOCK;
    const bool want = n >= 64 && s->h_prev.size() == (size_t)n && !getenv("RT_NO_LIGHT_COLUMNS");
    if (!want) {
        for (int i = 0; i < RT_MAX_LIGHTS; ++i) s->ltab_valid[i] = false;
        s->ltab_gen = ~0ull;
        return RT_OK;
    }
    const size_t per_light = (size_t)n_pad + 2 * (size_t)nb;   // float4 units
    float axis[RT_MAX_LIGHTS][3];
    bool usable[RT_MAX_LIGHTS];
    bool same = (s->ltab_gen == s->sphere_gen) && (s->ltab_n_lights == s->n_lights);
    for (int i = 0; i < s->n_lights; ++i) {
        const rt_light &l = s->lights[i];
        const float len = std::sqrt(l.pos.x * l.pos.x + l.pos.y * l.pos.y + l.pos.z * l.pos.z);   // as rt_build_frame_consts
        usable[i] = len > 0 && std::isfinite(len);
        axis[i][0] = usable[i] ? l.pos.x / len : 0.f;
        axis[i][1] = usable[i] ? l.pos.y / len : 0.f;
        axis[i][2] = usable[i] ? l.pos.z / len : 0.f;
        usable[i] = usable[i] && std::isfinite(axis[i][0]) && std::isfinite(axis[i][1]) && std::isfinite(axis[i][2]);
        same = same && (usable[i] == s->ltab_valid[i]) &&
               (!usable[i] || memcmp(axis[i], s->ltab_axis[i], sizeof axis[i]) == 0);
    }
    if (same) return RT_OK;
    const size_t total = per_light * (size_t)std::max(1, s->n_lights);
    if (total > s->cap_light_tabs) {
        if (s->d_light_tabs) RT_HIP(hipFree(s->d_light_tabs));
        s->d_light_tabs = nullptr;
        s->cap_light_tabs = 0;
        RT_HIP(hipMalloc((void **)&s->d_light_tabs, sizeof(float4) * total));
        s->cap_light_tabs = total;
    }
    // frames still in flight (another stream, a replaying graph) may be reading the old tables
    RT_HIP(hipDeviceSynchronize());
    std::vector<float4> h(total);
    for (int i = 0; i < s->n_lights; ++i) {
        s->ltab_valid[i] = usable[i];
        memcpy(s->ltab_axis[i], axis[i], sizeof axis[i]);
        if (usable[i])
            build_light_columns(s->h_prev.data(), n, axis[i], h.data() + per_light * i, h.data() + per_light * i + n_pad);
    }
    for (int i = s->n_lights; i < RT_MAX_LIGHTS; ++i) s->ltab_valid[i] = false;
    // pageable source: the copy has left `h` when the call returns
    RT_HIP(hipMemcpyAsync(s->d_light_tabs, h.data(), sizeof(float4) * total, hipMemcpyHostToDevice, stream));
    RT_HIP(hipStreamSynchronize(stream));   // rare (scene or light change): `h` goes out of scope
    s->ltab_gen = s->sphere_gen;
    s->ltab_n_lights = s->n_lights;
    return RT_OK;
}

int rt_scene_set_spheres_async(rt_scene *s, const rt_sphere *host_spheres, int n, hipStream_t stream)
{
    if (!s || n < 0 || (n > 0 && !host_spheres)) {
        rt_set_error("rt_scene_set_spheres: invalid argument");
        return RT_ERR_INVALID;
    }
    if (n > kMaxSpheres) {
        rt_set_error("rt_scene_set_spheres: %d spheres exceed the limit of %d", n, kMaxSpheres);
        return RT_ERR_CAPACITY;
    }
    const int n_pad = (n + 63) & ~63, nb = n_pad / RT_BLOCK;
    const size_t total = (size_t)n + (size_t)n_pad + (size_t)nb + ((size_t)n_pad + 3) / 4;   // in float4 units
    if ((int)total > s->cap_spheres) {
        if (s->d_spheres) RT_HIP(hipFree(s->d_spheres));
        s->d_spheres = nullptr;
        s->cap_spheres = 0;
        RT_HIP(hipMalloc((void **)&s->d_spheres, sizeof(float4) * total));
        s->cap_spheres = (int)total;
        s->h_prev.clear();
    }
    if ((int)total > s->cap_stage) {
        if (s->stage_busy) RT_HIP(hipEventSynchronize(s->stage_done));
        if (s->h_stage) RT_HIP(hipHostFree(s->h_stage));
        s->h_stage = nullptr;
        s->cap_stage = 0;
        RT_HIP(hipHostMalloc((void **)&s->h_stage, sizeof(float4) * total, hipHostMallocDefault));
        s->cap_stage = (int)total;
    }
    if (n > 0) {
        std::vector<float4> packed((size_t)n);
        pack_spheres(host_spheres, n, packed.data());
        if (s->n_spheres == n && s->h_prev.size() == (size_t)n &&
            memcmp(s->h_prev.data(), packed.data(), sizeof(float4) * (size_t)n) == 0)
            return RT_OK;   // unchanged since the last mirror: the device copy is current
        // the staging buffer is reused every frame: wait for the previous upload to have left it
        if (!s->stage_done) RT_HIP(hipEventCreateWithFlags(&s->stage_done, hipEventDisableTiming));
        if (s->stage_busy) RT_HIP(hipEventSynchronize(s->stage_done));
        float4 *h_orig = s->h_stage, *h_sorted = h_orig + n, *h_blocks = h_sorted + n_pad;
        int *h_idx = reinterpret_cast<int *>(h_blocks + nb);
        memcpy(h_orig, packed.data(), sizeof(float4) * (size_t)n);
        build_sorted_blocks(packed.data(), n, h_sorted, h_blocks, h_idx);
        RT_HIP(hipMemcpyAsync(s->d_spheres, s->h_stage, sizeof(float4) * total, hipMemcpyHostToDevice, stream));
        RT_HIP(hipEventRecord(s->stage_done, stream));
        s->stage_busy = true;
        s->h_prev.swap(packed);
        s->sphere_gen++;
    } else {
        s->h_prev.clear();
        s->sphere_gen++;
    }
    s->n_blocks = nb;
    s->n_spheres = n;
    return RT_OK;
}

extern "C" int rt_scene_set_spheres(rt_scene *s, const rt_sphere *host_spheres, int n)
{
    const int rc = rt_scene_set_spheres_async(s, host_spheres, n, nullptr);
    if (rc != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(nullptr));
    return RT_OK;
}

extern "C" int rt_scene_set_planes(rt_scene *s, const rt_plane *host_planes, int n)
{
    if (!s || n < 0 || (n > 0 && !host_planes)) {
        rt_set_error("rt_scene_set_planes: invalid argument");
        return RT_ERR_INVALID;
    }
    if (n > RT_MAX_PLANES) {
        rt_set_error("rt_scene_set_planes: %d planes > RT_MAX_PLANES %d", n, RT_MAX_PLANES);
        return RT_ERR_CAPACITY;
    }
    if (!s->d_planes) RT_HIP(hipMalloc((void **)&s->d_planes, sizeof(RtPlaneDev) * RT_MAX_PLANES));
    std::vector<RtPlaneDev> tmp(n ? n : 1);
    for (int i = 0; i < n; ++i)
        tmp[i] = RtPlaneDev{host_planes[i].orgin.x, host_planes[i].orgin.y, host_planes[i].orgin.z,
                            host_planes[i].normal.x, host_planes[i].normal.y, host_planes[i].normal.z, 0.f, 0.f};
    if (n) RT_HIP(hipMemcpy(s->d_planes, tmp.data(), sizeof(RtPlaneDev) * n, hipMemcpyHostToDevice));
    s->n_planes = n;
    return RT_OK;
}

extern "C" int rt_scene_set_cubes(rt_scene *s, const rt_cube *host_cubes, int n)
{
    if (!s || n < 0 || (n > 0 && !host_cubes)) {
        rt_set_error("rt_scene_set_cubes: invalid argument");
        return RT_ERR_INVALID;
    }
    if (n > RT_MAX_CUBES) {
        rt_set_error("rt_scene_set_cubes: %d cubes > RT_MAX_CUBES %d", n, RT_MAX_CUBES);
        return RT_ERR_CAPACITY;
    }
    if (!s->d_cubes) RT_HIP(hipMalloc((void **)&s->d_cubes, sizeof(RtCubeDev) * RT_MAX_CUBES));
    std::vector<RtCubeDev> tmp(n ? n : 1);
    for (int i = 0; i < n; ++i) {
        const rt_cube &c = host_cubes[i];
        tmp[i] = RtCubeDev{c.bounds[0].x, c.bounds[0].y, c.bounds[0].z, c.bounds[1].x, c.bounds[1].y, c.bounds[1].z,
                           c.orgin.x, c.orgin.y, c.orgin.z, 0.f, 0.f, 0.f};
    }
    if (n) RT_HIP(hipMemcpy(s->d_cubes, tmp.data(), sizeof(RtCubeDev) * n, hipMemcpyHostToDevice));
    s->n_cubes = n;
    return RT_OK;
}

// Flatten the reference-layout mesh (triangles, leaf boxes with their own index
// arrays) into three device arrays: triangles, boxes {bounds, start, len}, indices.
extern "C" int rt_scene_set_mesh(rt_scene *s, const rt_mesh *mesh)
{
    if (!s) {
        rt_set_error("rt_scene_set_mesh: null scene");
        return RT_ERR_INVALID;
    }
    if (s->d_tris) RT_HIP(hipFree(s->d_tris));
    if (s->d_boxes) RT_HIP(hipFree(s->d_boxes));
    if (s->d_tri_idx) RT_HIP(hipFree(s->d_tri_idx));
    if (s->d_box_spheres) RT_HIP(hipFree(s->d_box_spheres));
    if (s->d_tri9) RT_HIP(hipFree(s->d_tri9));
    s->d_tri9 = nullptr;
    s->d_tris = nullptr; s->d_boxes = nullptr; s->d_tri_idx = nullptr; s->d_box_spheres = nullptr;
    s->n_boxes = s->n_tris = 0;
    if (!mesh || mesh->bvhbox_count == 0) return RT_OK;
    if (mesh->poly_count <= 0 || mesh->bvhbox_count < 0 || !mesh->d_tri_arr || !mesh->d_box) {
        rt_set_error("rt_scene_set_mesh: malformed mesh (poly_count=%d bvhbox_count=%d)", mesh->poly_count,
                     mesh->bvhbox_count);
        return RT_ERR_INVALID;
    }
    std::vector<RtTriDev> tris((size_t)mesh->poly_count);
    for (int i = 0; i < mesh->poly_count; ++i) {
        const rt_triangle &t = mesh->d_tri_arr[i];
        RtTriDev &d = tris[i];
        memset(&d, 0, sizeof d);
        memcpy(d.p0, &t.points[0], 12); memcpy(d.p1, &t.points[1], 12); memcpy(d.p2, &t.points[2], 12);
        memcpy(d.n, &t.normal, 12);
        memcpy(d.vn, t.vecNormal, 36);
        memcpy(d.vt, t.vt, 24);
    }
    std::vector<RtBoxDev> boxes((size_t)mesh->bvhbox_count);
    std::vector<float> bsph((size_t)mesh->bvhbox_count * 4);   // bounding sphere of each leaf (for beam culling)
    std::vector<int> idx;
    for (int j = 0; j < mesh->bvhbox_count; ++j) {
        const rt_bvhbox &b = mesh->d_box[j];
        const rt_cube *c = b.d_bvhbox ? b.d_bvhbox : b.bvhbox;
        if (!c || !b.d_indexes || b.length < 0) {
            rt_set_error("rt_scene_set_mesh: leaf %d is incomplete", j);
            return RT_ERR_INVALID;
        }
        RtBoxDev &d = boxes[j];
        d.lo[0] = c->bounds[0].x; d.lo[1] = c->bounds[0].y; d.lo[2] = c->bounds[0].z;
        d.hi[0] = c->bounds[1].x; d.hi[1] = c->bounds[1].y; d.hi[2] = c->bounds[1].z;
        {
            const double cx = 0.5 * ((double)d.lo[0] + d.hi[0]), cy = 0.5 * ((double)d.lo[1] + d.hi[1]),
                         cz = 0.5 * ((double)d.lo[2] + d.hi[2]);
            const double hx = 0.5 * std::fabs((double)d.hi[0] - d.lo[0]), hy = 0.5 * std::fabs((double)d.hi[1] - d.lo[1]),
                         hz = 0.5 * std::fabs((double)d.hi[2] - d.lo[2]);
            bsph[4 * j + 0] = (float)cx; bsph[4 * j + 1] = (float)cy; bsph[4 * j + 2] = (float)cz;
            bsph[4 * j + 3] = (float)((hx * hx + hy * hy + hz * hz) * 1.001 + 1e-6);   // radius^2, rounded up
        }
        d.start = (int)idx.size();
        d.len = b.length;
        for (int i = 0; i < b.length; ++i) {
            if (b.d_indexes[i] < 0 || b.d_indexes[i] >= mesh->poly_count) {
                rt_set_error("rt_scene_set_mesh: leaf %d references triangle %d of %d", j, b.d_indexes[i], mesh->poly_count);
                return RT_ERR_INVALID;
            }
            idx.push_back(b.d_indexes[i]);
        }
    }
    RT_HIP(hipMalloc((void **)&s->d_tris, sizeof(RtTriDev) * tris.size()));
    RT_HIP(hipMalloc((void **)&s->d_boxes, sizeof(RtBoxDev) * boxes.size()));
    RT_HIP(hipMalloc((void **)&s->d_tri_idx, sizeof(int) * (idx.size() ? idx.size() : 1)));
    {   // blocks of RT_BLOCK consecutive leaves (leaf order is kept: it decides ties between triangles),
        // each with a sphere around its members' spheres, appended after the (padded) leaf spheres
        const int nb = mesh->bvhbox_count, nb_pad = (nb + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK, nblk = nb_pad / RT_BLOCK;
        bsph.resize((size_t)(nb_pad + nblk) * 4, 0.f);
        for (int j = nb; j < nb_pad; ++j) bsph[4 * (size_t)j + 3] = -1.f;
        for (int k = 0; k < nblk; ++k) {
            const int j0 = k * RT_BLOCK, j1 = std::min(nb, j0 + RT_BLOCK);
            double cx = 0, cy = 0, cz = 0;
            for (int j = j0; j < j1; ++j) { cx += bsph[4 * (size_t)j]; cy += bsph[4 * (size_t)j + 1]; cz += bsph[4 * (size_t)j + 2]; }
            const double inv = 1.0 / std::max(1, j1 - j0);
            const float cf[3] = {(float)(cx * inv), (float)(cy * inv), (float)(cz * inv)};
            double r = 0;
            for (int j = j0; j < j1; ++j) {
                const double dx = bsph[4 * (size_t)j] - (double)cf[0], dy = bsph[4 * (size_t)j + 1] - (double)cf[1],
                             dz = bsph[4 * (size_t)j + 2] - (double)cf[2];
                const double d = std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt(std::max(0.0, (double)bsph[4 * (size_t)j + 3]));
                r = (d > r || d != d) ? d : r;   // a NaN sticks
            }
            float rf = (float)(r * 1.001 + 1e-3);
            const bool fin = std::isfinite(cf[0]) && std::isfinite(cf[1]) && std::isfinite(cf[2]) && rf == rf;
            float *o = &bsph[4 * (size_t)(nb_pad + k)];
            o[0] = fin ? cf[0] : 0.f; o[1] = fin ? cf[1] : 0.f; o[2] = fin ? cf[2] : 0.f;
            o[3] = fin ? rf : INFINITY;
        }
    }
    RT_HIP(hipMalloc((void **)&s->d_box_spheres, sizeof(float) * bsph.size()));
    RT_HIP(hipMemcpy(s->d_box_spheres, bsph.data(), sizeof(float) * bsph.size(), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(s->d_tris, tris.data(), sizeof(RtTriDev) * tris.size(), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(s->d_boxes, boxes.data(), sizeof(RtBoxDev) * boxes.size(), hipMemcpyHostToDevice));
    if (!idx.empty()) RT_HIP(hipMemcpy(s->d_tri_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    {   // vertices per (leaf, triangle) pair, de-indexed and padded by 64 floats so that a full-wave load stays inside
        std::vector<float> t9(idx.size() * 9 + 64, 0.f);
        for (size_t k = 0; k < idx.size(); ++k) {
            memcpy(&t9[9 * k + 0], tris[idx[k]].p0, 12);
            memcpy(&t9[9 * k + 3], tris[idx[k]].p1, 12);
            memcpy(&t9[9 * k + 6], tris[idx[k]].p2, 12);
        }
        RT_HIP(hipMalloc((void **)&s->d_tri9, sizeof(float) * t9.size()));
        RT_HIP(hipMemcpy(s->d_tri9, t9.data(), sizeof(float) * t9.size(), hipMemcpyHostToDevice));
    }
    s->n_boxes = mesh->bvhbox_count;
    s->n_tris = mesh->poly_count;
    s->mesh_has_normals = mesh->has_normals ? 1 : 0;
    return RT_OK;
}

static int upload_planes(float *dst[3], const float *r, const float *g, const float *b, int w, int h)
{
    const float *src[3] = {r, g, b};
    const size_t bytes = sizeof(float) * (size_t)w * (size_t)h;
    free_planes(dst);
    for (int i = 0; i < 3; ++i) {
        RT_HIP(hipMalloc((void **)&dst[i], bytes));
        RT_HIP(hipMemcpy(dst[i], src[i], bytes, hipMemcpyDefault));
    }
    return RT_OK;
}

extern "C" int rt_scene_set_texture(rt_scene *s, const float *r, const float *g, const float *b, int w, int h)
{
    if (!s || !r || !g || !b || w <= 0 || h <= 0) {
        rt_set_error("rt_scene_set_texture: invalid argument");
        return RT_ERR_INVALID;
    }
    const int rc = upload_planes(s->d_tex, r, g, b, w, h);
    if (rc != RT_OK) return rc;
    s->tex_w = w;
    s->tex_h = h;
    return RT_OK;
}

extern "C" int rt_scene_set_sky(rt_scene *s, const rt_sphere *box, const float *r, const float *g,
                                const float *b, int w, int h)
{
    if (!s || !box || !r || !g || !b || w <= 0 || h <= 0) {
        rt_set_error("rt_scene_set_sky: invalid argument");
        return RT_ERR_INVALID;
    }
    const int rc = upload_planes(s->d_sky, r, g, b, w, h);
    if (rc != RT_OK) return rc;
    s->sky_w = w;
    s->sky_h = h;
    s->sky_c[0] = box->orgin.x;
    s->sky_c[1] = box->orgin.y;
    s->sky_c[2] = box->orgin.z;
    s->sky_radius = box->radius;
    s->have_sky = true;
    return RT_OK;
}

extern "C" int rt_scene_set_lights(rt_scene *s, const rt_light *lights, int n)
{
    if (!s || n < 0 || (n > 0 && !lights)) {
        rt_set_error("rt_scene_set_lights: invalid argument");
        return RT_ERR_INVALID;
    }
    if (n > RT_MAX_LIGHTS) {
        rt_set_error("rt_scene_set_lights: light_size %d > RT_MAX_LIGHTS %d", n, RT_MAX_LIGHTS);
        return RT_ERR_CAPACITY;
    }
    for (int i = 0; i < n; ++i) s->lights[i] = lights[i];
    s->n_lights = n;
    return RT_OK;
}

// ---------------------------------------------------------------------------
// sample positions (build-defined extension; n = 1 is the reference's +0.5)
// ---------------------------------------------------------------------------
extern "C" int rt_sample_offset(int k, int n, double *ox, double *oy)
{
    if (n < 1 || k < 0 || k >= n || !ox || !oy) return RT_ERR_INVALID;
    int g = 1;
    while (g * g < n) ++g;   // stratified g x g grid, cell centres
    *ox = ((double)(k % g) + 0.5) / (double)g;
    *oy = ((double)(k / g) + 0.5) / (double)g;
    return RT_OK;
}

extern "C" float rt_default_aspect(void)
{
    return (float)std::tan((90 * 0.5 * 3.1415) / 180);   // kernel.cu:1701
}

// ---------------------------------------------------------------------------
// frame uniforms: everything the reference recomputes per pixel from
// frame-constant inputs, evaluated once with the same operations.
// ---------------------------------------------------------------------------
// eyePos + cam.Org, kernel.cu:1629-1631: the origin of every primary ray of the frame
void rt_ray_origin(const rt_frame_desc *fd, float org[3])
{
    const float ez = -1.f / fd->aspect;
    org[0] = 0.f + fd->cam.Org.x;
    org[1] = 0.f + fd->cam.Org.y;
    org[2] = ez + fd->cam.Org.z;
}

int rt_build_frame_consts(const rt_scene *s, const rt_frame_desc *fd, RtFrameConsts *fc)
{
    if (!s || !fd) {
        rt_set_error("rt_scene_render: null scene or frame");
        return RT_ERR_INVALID;
    }
    if (fd->width <= 0 || fd->height <= 0) {
        rt_set_error("rt_scene_render: width/height must be positive (%d x %d)", fd->width, fd->height);
        return RT_ERR_INVALID;
    }
    const rt_launch_opts &o = fd->opts;
    int y0 = o.y0, y1 = o.y1;
    if (y0 == 0 && y1 == 0) y1 = fd->height;
    if (y0 < 0 || y1 > fd->height || y0 >= y1) {
        rt_set_error("rt_scene_render: bad row band [%d,%d) for height %d", y0, y1, fd->height);
        return RT_ERR_INVALID;
    }
    const int spp = o.spp > 0 ? o.spp : 1;
    const int total = o.sample_total > 0 ? o.sample_total : spp;
    if (spp > RT_MAX_SPP || total > RT_MAX_SPP || o.sample_base < 0 || o.sample_base + spp > total) {
        rt_set_error("rt_scene_render: bad sample range base=%d spp=%d total=%d (max %d)", o.sample_base, spp,
                     total, RT_MAX_SPP);
        return RT_ERR_INVALID;
    }
    if ((s->n_spheres > 0 || s->n_planes > 0 || s->n_cubes > 0 || s->n_boxes > 0) && (!s->d_tex[0] || s->tex_w <= 0)) {
        rt_set_error("rt_scene_render: scene has primitives but no object texture");
        return RT_ERR_INVALID;
    }
    if (!s->have_sky) {
        rt_set_error("rt_scene_render: scene has no skybox");
        return RT_ERR_INVALID;
    }
    {
        bool owns_rows = true;
        if (o.interleave_count > 1) {
            const int b = o.interleave_rows > 0 ? o.interleave_rows : 16;
            owns_rows = b > 0 && (long long)o.interleave_index * b < fd->height;
        }
        if (owns_rows && !fd->pixels && !o.rgba) {
            rt_set_error("rt_scene_render: no output buffer (pixels and opts.rgba are both null)");
            return RT_ERR_INVALID;
        }
    }

    memset(fc, 0, sizeof *fc);
    fc->width = fd->width;
    fc->height = fd->height;
    fc->y0 = y0;
    fc->y1 = y1;
    fc->n_spheres = s->n_spheres;
    fc->n_lights = s->n_lights;
    fc->spp = spp;
    fc->sample_base = o.sample_base;
    fc->sample_total = (float)total;
    fc->accumulate = o.accumulate ? 1 : 0;
    fc->resolve = ((fd->pixels || o.packed24) && o.resolve >= 0) ? 1 : 0;
    fc->local_rows = y1 - y0;
    if (o.interleave_count > 1) {
        const int b = o.interleave_rows > 0 ? o.interleave_rows : 16;
        if (o.y0 != 0 || (o.y1 != 0 && o.y1 != fd->height) || b % 16 != 0 || o.interleave_index < 0 ||
            o.interleave_index >= o.interleave_count) {
            rt_set_error("rt_scene_render: bad interleave (count=%d index=%d rows=%d; y0/y1 must be 0)",
                         o.interleave_count, o.interleave_index, b);
            return RT_ERR_INVALID;
        }
        fc->il_count = o.interleave_count;
        fc->il_index = o.interleave_index;
        fc->il_rows = b;
        int rows = 0;   // rows of the blocks this rank owns
        for (int k = o.interleave_index; k * b < fd->height; k += o.interleave_count)
            rows += (fd->height - k * b < b) ? fd->height - k * b : b;
        fc->local_rows = rows;   // may be 0 (more ranks than row blocks): the launch is then skipped
    }
    fc->force_slow = o.force_slow_path ? 1 : 0;
    {   // timing experiments only: output is wrong when set
        const char *ab = getenv("RT_ABLATE");
        fc->ablate = ab ? atoi(ab) : 0;
    }

    // kernel.cu:1624-1625
    const float aspect = fd->aspect;
    fc->aspect_d = (double)aspect;
    fc->width_d = (double)(float)fd->width;
    fc->height_d = (double)(float)fd->height;
    fc->hw_d = (double)((float)fd->height / (float)fd->width);
    // kernel.cu:1629-1631: eyePos = (0,0,-1/aspect); dir - eyePos; eyePos + cam.Org
    const float ez = -1.f / aspect;
    fc->eye_nz = 0.f - ez;
    {
        float org[3];
        rt_ray_origin(fd, org);
        fc->org_x = org[0];
        fc->org_y = org[1];
        fc->org_z = org[2];
    }
    // camera::rotateDir, kernel.cu:249-250
    const float yawRad = (float)(fd->cam.Camyaw * (3.1415 / 180));
    const float pitchRad = (float)(fd->cam.Campitch * (3.1415 / 180));
    fc->cos_pitch = rtm::cosf_rt(pitchRad);
    fc->sin_pitch = rtm::sinf_rt(pitchRad);
    fc->cos_yaw = rtm::cosf_rt(yawRad);
    fc->sin_yaw = rtm::sinf_rt(yawRad);
    for (int k = 0; k < total; ++k) rt_sample_offset(k, total, &fc->off_x[k], &fc->off_y[k]);

    // castLightRay sample constants, kernel.cu:1453-1454, 1462-1463, 1538
    {   // the device uses literals for this sequence (brightness_steps); re-derive and compare
        static const float kSteps[RT_SHADOW_SAMPLES + 1] = {
            0x0.0p+0f, 0x1.99999ap-4f, 0x1.99999ap-3f, 0x1.333334p-2f, 0x1.99999ap-2f, 0x1.000000p-1f,
            0x1.333334p-1f, 0x1.666668p-1f, 0x1.99999cp-1f, 0x1.ccccd0p-1f, 0x1.000002p+0f};
        float b = 0;
        for (int j = 0; j <= RT_SHADOW_SAMPLES; ++j) {
            fc->btab[j] = b;
            if (b != kSteps[j]) {
                rt_set_error("internal: brightness step table mismatch at %d", j);
                return RT_ERR_INVALID;
            }
            b = (float)(b + 0.1);
        }
    }
    for (int j = 0; j < RT_SHADOW_SAMPLES; ++j) {
        const float jf = (float)j / 10;
        const float phi = jf * 2.f * 3.1415f;
        fc->jf[j] = jf;
        fc->jcos[j] = rtm::cosf_rt(phi);
        fc->jsin[j] = rtm::sinf_rt(phi);
    }
    for (int i = 0; i < s->n_lights; ++i) {
        const rt_light &l = s->lights[i];
        RtLightDev &d = fc->lights[i];
        d.px = l.pos.x; d.py = l.pos.y; d.pz = l.pos.z;
        d.size = l.size;
        d.r = l.r; d.g = l.g; d.b = l.b;
        const float len = std::sqrt(l.pos.x * l.pos.x + l.pos.y * l.pos.y + l.pos.z * l.pos.z);
        d.pos_len = len;
        // a light at the origin has no beam axis: NaN makes the kernel skip culling
        d.ux = len > 0 ? l.pos.x / len : NAN;
        d.uy = len > 0 ? l.pos.y / len : NAN;
        d.uz = len > 0 ? l.pos.z / len : NAN;
    }
    fc->tex_r = s->d_tex[0]; fc->tex_g = s->d_tex[1]; fc->tex_b = s->d_tex[2];
    fc->tex_w = s->tex_w; fc->tex_h = s->tex_h;
    fc->sky_r = s->d_sky[0]; fc->sky_g = s->d_sky[1]; fc->sky_b = s->d_sky[2];
    fc->sky_w = s->sky_w; fc->sky_h = s->sky_h;
    fc->sky_cx = s->sky_c[0]; fc->sky_cy = s->sky_c[1]; fc->sky_cz = s->sky_c[2];
    fc->sky_r2 = s->sky_radius * s->sky_radius;
    {   // as quadratic() evaluates them on the device (no contraction: this file is built with -ffp-contract=off)
        const float ocx = fc->org_x - fc->sky_cx, ocy = fc->org_y - fc->sky_cy, ocz = fc->org_z - fc->sky_cz;
        fc->sky_ocx = ocx; fc->sky_ocy = ocy; fc->sky_ocz = ocz;
        fc->sky_C = ((ocx * ocx + ocy * ocy) + ocz * ocz) - fc->sky_r2;
    }
    fc->planes = s->d_planes;
    fc->cubes = s->d_cubes;
    fc->n_planes = s->n_planes;
    fc->n_cubes = s->n_cubes;
    {
        const int n_pad = (s->n_spheres + 63) & ~63;
        const float4 *base = s->d_spheres;
        fc->sorted = base ? reinterpret_cast<const float *>(base + s->n_spheres) : nullptr;
        fc->blocks = base ? reinterpret_cast<const float *>(base + s->n_spheres + n_pad) : nullptr;
        fc->orig_idx = base ? reinterpret_cast<const int *>(base + s->n_spheres + n_pad + s->n_blocks) : nullptr;
        fc->n_blocks = s->n_blocks;
        const size_t per_light = (size_t)n_pad + 2 * (size_t)s->n_blocks;
        const bool current = s->d_light_tabs && s->ltab_gen == s->sphere_gen && s->ltab_n_lights == s->n_lights;
        for (int i = 0; i < RT_DEV_MAX_LIGHTS; ++i) {
            const bool on = current && i < s->n_lights && s->ltab_valid[i];
            fc->lsorted[i] = on ? reinterpret_cast<const float *>(s->d_light_tabs + per_light * i) : nullptr;
            fc->lblocks[i] = on ? reinterpret_cast<const float *>(s->d_light_tabs + per_light * i + n_pad) : nullptr;
        }
        const float org[3] = {fc->org_x, fc->org_y, fc->org_z};
        const bool cones = s->d_cone_tab && s->cone_valid && s->cone_gen == s->sphere_gen &&
                           memcmp(org, s->cone_org, sizeof org) == 0;
        fc->csorted = cones ? reinterpret_cast<const float *>(s->d_cone_tab) : nullptr;
        fc->cblocks = cones ? reinterpret_cast<const float *>(s->d_cone_tab + n_pad) : nullptr;
        fc->corig = cones ? reinterpret_cast<const int *>(s->d_cone_tab + n_pad + 2 * (size_t)s->n_blocks) : nullptr;
        fc->cone_kcap = kConeKcap;
    }
    fc->tris = s->d_tris;
    fc->boxes = s->d_boxes;
    fc->tri_idx = s->d_tri_idx;
    fc->box_spheres = s->d_box_spheres;
    fc->tri9 = s->d_tri9;
    fc->n_boxes = s->n_boxes;
    fc->mesh_has_normals = s->mesh_has_normals;
    fc->rgba = o.rgba;
    fc->packed = fd->pixels;
    fc->packed24 = (uint32_t *)o.packed24;
    if (o.packed24 && fd->width % 4 != 0) {
        rt_set_error("rt_scene_render: packed24 needs a frame width that is a multiple of 4 (got %d)", fd->width);
        return RT_ERR_INVALID;
    }
    fc->stats = (unsigned long long *)o.stats;
    return RT_OK;
}

static int tile_from_opts(const rt_launch_opts &o, int *tile)
{
    const int t = o.tile ? o.tile : 8;
    if (t != 8 && t != 16 && t != 32 && t != 64) {
        rt_set_error("rt_scene_render: tile width %d not in {8,16,32,64}", t);
        return RT_ERR_INVALID;
    }
    *tile = t;
    return RT_OK;
}

extern "C" int rt_scene_render(rt_scene *s, const rt_frame_desc *fd, void *stream)
{
    RtFrameConsts fc;
    int rc = RT_OK;
    if (s && fd && fd->opts.cull != 0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream) (void)hipStreamIsCapturing((hipStream_t)stream, &cs);
        if (cs == hipStreamCaptureStatusNone) {   // a capture re-uses what rt_graph_capture prepared
            rc = rt_scene_prepare_lights(s, (hipStream_t)stream);
            if (rc != RT_OK) return rc;
            float org[3];
            rt_ray_origin(fd, org);
            rc = rt_scene_prepare_eye(s, org, (hipStream_t)stream);
            if (rc != RT_OK) return rc;
        }
    }
    rc = rt_build_frame_consts(s, fd, &fc);
    if (rc != RT_OK) return rc;
    int tile = 8;
    rc = tile_from_opts(fd->opts, &tile);
    if (rc != RT_OK) return rc;
    if (fc.local_rows == 0) return RT_OK;   // this rank owns no rows of the frame
    if (fc.n_boxes > 0 && (tile != 8 || fd->opts.stats)) {
        rt_set_error("rt_scene_render: scenes with a triangle mesh render with the default tile and without stats");
        return RT_ERR_UNSUPPORTED;
    }
    const int cull = (fd->opts.cull == 0) ? 0 : 1;
    const int stats = fd->opts.stats ? (fd->opts.profile ? 2 : 1) : 0;
    RT_HIP(rt_dev_launch_trace(&fc, s->d_spheres, tile, cull, stats, table_in_lds_for(s->n_spheres), (hipStream_t)stream));
    return RT_OK;
}

// ---------------------------------------------------------------------------
// rayTrace launch shim (kernel.cu:1615, 1780-1783): same argument list, the
// object / skybox graphs are read on the host and mirrored to the device.
// ---------------------------------------------------------------------------
struct ShimCache {
    rt_scene *scene = nullptr;
    const rt_mesh *mesh_key = nullptr;
    int mesh_polys = -1, mesh_boxes = -1;
    const float *tex_key[3] = {nullptr, nullptr, nullptr};
    int tex_w = 0, tex_h = 0;
    const float *sky_key[3] = {nullptr, nullptr, nullptr};
    int sky_w = 0, sky_h = 0;
    float sky_c[3] = {0, 0, 0};
    float sky_radius = -1;
};
static ShimCache g_shim;

extern "C" void rt_invalidate_textures(void)
{
    g_shim.mesh_key = nullptr;
    g_shim.mesh_polys = g_shim.mesh_boxes = -1;
    g_shim.tex_key[0] = g_shim.tex_key[1] = g_shim.tex_key[2] = nullptr;
    g_shim.sky_key[0] = g_shim.sky_key[1] = g_shim.sky_key[2] = nullptr;
}

static bool sprite_ok(const rt_sprite *t)
{
    return t && t->rBuff && t->gBuff && t->bBuff && t->rBuff->data && t->gBuff->data && t->bBuff->data &&
           t->width > 0 && t->height > 0;
}

extern "C" int rt_launch_raytrace_ex(uint32_t *pixels, int width, int height, float aspect,
                                     const rt_object *objs, const rt_light *lights, int light_size,
                                     rt_camera cam, const rt_skybox *sky, void *stream,
                                     const rt_launch_opts *opts)
{
    if (!objs || !sky || (!lights && light_size > 0)) {
        rt_set_error("rt_launch_raytrace: null objs/lights/sky");
        return RT_ERR_INVALID;
    }
    if (objs->cube_count < 0 || objs->plane_count < 0 || (objs->cube_count > 0 && !objs->d_cubes) ||
        (objs->plane_count > 0 && !objs->d_planes)) {
        rt_set_error("rt_launch_raytrace: bad cube/plane list");
        return RT_ERR_INVALID;
    }
    if (objs->sphere_count < 0 || (objs->sphere_count > 0 && !objs->d_spheres)) {
        rt_set_error("rt_launch_raytrace: bad sphere list");
        return RT_ERR_INVALID;
    }
    if (!sky->box || !sprite_ok(sky->skyboxTex)) {
        rt_set_error("rt_launch_raytrace: skybox needs a box sphere and a texture");
        return RT_ERR_INVALID;
    }
    if ((objs->sphere_count > 0 || objs->cube_count > 0 || objs->plane_count > 0 || objs->mesh1) &&
        !sprite_ok(objs->texture)) {
        rt_set_error("rt_launch_raytrace: object texture missing");
        return RT_ERR_INVALID;
    }
    if (!g_shim.scene) g_shim.scene = rt_scene_create();
    rt_scene *s = g_shim.scene;
    int rc;
    // textures: uploaded once per (planes, size); see rt_invalidate_textures()
    if (objs->sphere_count > 0 || objs->cube_count > 0 || objs->plane_count > 0 || objs->mesh1) {
        const rt_sprite *t = objs->texture;
        if (t->rBuff->data != g_shim.tex_key[0] || t->gBuff->data != g_shim.tex_key[1] ||
            t->bBuff->data != g_shim.tex_key[2] || t->width != g_shim.tex_w || t->height != g_shim.tex_h) {
            rc = rt_scene_set_texture(s, t->rBuff->data, t->gBuff->data, t->bBuff->data, t->width, t->height);
            if (rc != RT_OK) return rc;
            g_shim.tex_key[0] = t->rBuff->data; g_shim.tex_key[1] = t->gBuff->data; g_shim.tex_key[2] = t->bBuff->data;
            g_shim.tex_w = t->width; g_shim.tex_h = t->height;
        }
    }
    {
        const rt_sprite *t = sky->skyboxTex;
        if (t->rBuff->data != g_shim.sky_key[0] || t->gBuff->data != g_shim.sky_key[1] ||
            t->bBuff->data != g_shim.sky_key[2] || t->width != g_shim.sky_w || t->height != g_shim.sky_h ||
            sky->box->orgin.x != g_shim.sky_c[0] || sky->box->orgin.y != g_shim.sky_c[1] ||
            sky->box->orgin.z != g_shim.sky_c[2] || sky->box->radius != g_shim.sky_radius) {
            rc = rt_scene_set_sky(s, sky->box, t->rBuff->data, t->gBuff->data, t->bBuff->data, t->width, t->height);
            if (rc != RT_OK) return rc;
            g_shim.sky_key[0] = t->rBuff->data; g_shim.sky_key[1] = t->gBuff->data; g_shim.sky_key[2] = t->bBuff->data;
            g_shim.sky_w = t->width; g_shim.sky_h = t->height;
            g_shim.sky_c[0] = sky->box->orgin.x; g_shim.sky_c[1] = sky->box->orgin.y; g_shim.sky_c[2] = sky->box->orgin.z;
            g_shim.sky_radius = sky->box->radius;
        }
    }
    // the mesh is uploaded once per (pointer, counts), like the textures
    {
        const rt_mesh *m = (objs->mesh1 && objs->mesh1->bvhbox_count > 0) ? objs->mesh1 : nullptr;
        const int polys = m ? m->poly_count : 0, boxes = m ? m->bvhbox_count : 0;
        if (m != g_shim.mesh_key || polys != g_shim.mesh_polys || boxes != g_shim.mesh_boxes) {
            rc = rt_scene_set_mesh(s, m);
            if (rc != RT_OK) return rc;
            g_shim.mesh_key = m;
            g_shim.mesh_polys = polys;
            g_shim.mesh_boxes = boxes;
        }
    }
    // spheres and lights are small and may change every frame: re-mirror them
    rc = rt_scene_set_spheres_async(s, objs->d_spheres, objs->sphere_count, (hipStream_t)stream);
    if (rc != RT_OK) return rc;
    rc = rt_scene_set_lights(s, lights, light_size);
    if (rc != RT_OK) return rc;
    if (objs->plane_count > 0 || s->n_planes > 0) {
        rc = rt_scene_set_planes(s, objs->d_planes, objs->plane_count);
        if (rc != RT_OK) return rc;
    }
    if (objs->cube_count > 0 || s->n_cubes > 0) {
        rc = rt_scene_set_cubes(s, objs->d_cubes, objs->cube_count);
        if (rc != RT_OK) return rc;
    }

    rt_frame_desc fd;
    memset(&fd, 0, sizeof fd);
    fd.struct_size = sizeof fd;
    fd.width = width;
    fd.height = height;
    fd.aspect = aspect;
    fd.cam = cam;
    fd.pixels = pixels;
    if (opts) {
        const size_t nbytes = opts->struct_size < sizeof fd.opts ? opts->struct_size : sizeof fd.opts;
        memcpy(&fd.opts, opts, nbytes);
        fd.opts.struct_size = (uint32_t)sizeof fd.opts;
    } else {
        fd.opts.cull = -1;
    }
    return rt_scene_render(s, &fd, stream);
}

extern "C" int rt_launch_raytrace(uint32_t *pixels, int width, int height, float aspect,
                                  const rt_object *objs, const rt_light *lights, int light_size,
                                  rt_camera cam, const rt_skybox *sky, void *stream)
{
    return rt_launch_raytrace_ex(pixels, width, height, aspect, objs, lights, light_size, cam, sky, stream, nullptr);
}

// ---------------------------------------------------------------------------
// diagnostics: device evaluation of scalar building blocks (host arrays in/out)
// ---------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { RT_HIP(hipMalloc((void **)&p, sizeof(T) * (n ? n : 1))); return RT_OK; }
};

extern "C" int rt_debug_math(int op, const float *a, const float *b, float *out, int n)
{
    if (n <= 0 || !a || !out || op < 0 || op > 5 || (op == 3 && !b)) return RT_ERR_INVALID;
    DevBuf<float> da, db, dout;
    int rc;
    if ((rc = da.alloc(n)) || (rc = db.alloc(n)) || (rc = dout.alloc(n))) return rc;
    RT_HIP(hipMemcpy(da.p, a, sizeof(float) * n, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(db.p, b ? b : a, sizeof(float) * n, hipMemcpyHostToDevice));
    RT_HIP(rt_dev_launch_dbg_math(op, da.p, db.p, dout.p, n, nullptr));
    RT_HIP(hipMemcpy(out, dout.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_debug_intersect(const rt_sphere *spheres, const rt_ray *rays, int n, int *hit, float *t)
{
    if (n <= 0 || !spheres || !rays || !hit || !t) return RT_ERR_INVALID;
    std::vector<float4> tab(n);
    pack_spheres(spheres, n, tab.data());
    DevBuf<float4> dtab;
    DevBuf<float> drays, dt;
    DevBuf<int> dhit;
    int rc;
    if ((rc = dtab.alloc(n)) || (rc = drays.alloc(6 * (size_t)n)) || (rc = dt.alloc(n)) || (rc = dhit.alloc(n)))
        return rc;
    RT_HIP(hipMemcpy(dtab.p, tab.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(drays.p, rays, sizeof(float) * 6 * n, hipMemcpyHostToDevice));
    RT_HIP(rt_dev_launch_dbg_intersect(dtab.p, drays.p, n, dhit.p, dt.p, nullptr));
    RT_HIP(hipMemcpy(hit, dhit.p, sizeof(int) * n, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(t, dt.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_debug_light(const rt_sphere *spheres, int n_spheres, const rt_vec3 *start,
                              const rt_vec3 *normal, const rt_light *light, int n, float *dirs,
                              float *brightness)
{
    if (n <= 0 || n_spheres < 0 || !start || !normal || !light || !dirs || !brightness) return RT_ERR_INVALID;
    rt_scene sc;
    sc.lights[0] = *light;
    sc.n_lights = 1;
    sc.have_sky = true;
    rt_frame_desc fd;
    memset(&fd, 0, sizeof fd);
    fd.width = fd.height = 1;
    fd.aspect = 1.f;
    uint32_t dummy;
    fd.pixels = &dummy;
    RtFrameConsts fc;
    int rc = rt_build_frame_consts(&sc, &fd, &fc);
    if (rc != RT_OK) return rc;
    fc.n_spheres = n_spheres;
    std::vector<float4> tab(n_spheres ? n_spheres : 1);
    if (n_spheres) pack_spheres(spheres, n_spheres, tab.data());
    DevBuf<float4> dtab;
    DevBuf<float> dstart, dnormal, ddirs, dbright;
    if ((rc = dtab.alloc(tab.size())) || (rc = dstart.alloc(3 * (size_t)n)) || (rc = dnormal.alloc(3 * (size_t)n)) ||
        (rc = ddirs.alloc(30 * (size_t)n)) || (rc = dbright.alloc(n)))
        return rc;
    RT_HIP(hipMemcpy(dtab.p, tab.data(), sizeof(float4) * tab.size(), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(dstart.p, start, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(dnormal.p, normal, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    RT_HIP(rt_dev_launch_dbg_light(&fc, dtab.p, dstart.p, dnormal.p, 0, n, ddirs.p, dbright.p, nullptr));
    RT_HIP(hipMemcpy(dirs, ddirs.p, sizeof(float) * 30 * n, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(brightness, dbright.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return RT_OK;
}
