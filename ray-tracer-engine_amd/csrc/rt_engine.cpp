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
#include "rt_tables.h"

// launchers defined in rt_kernels.hip
extern "C" hipError_t rt_dev_launch_dbg_shortcuts(int what, unsigned seed, long long n, unsigned long long *out, hipStream_t stream);
extern "C" hipError_t rt_dev_launch_dbg_math(int op, const float *a, const float *b, float *out, int n,
                                             hipStream_t stream);
extern "C" hipError_t rt_dev_launch_dbg_intersect(const float4 *tab, const float *rays, int n, int *hit,
                                                  float *t, hipStream_t stream);
extern "C" hipError_t rt_dev_launch_dbg_light(const RtFrameConsts *fc, const float4 *tab, const float *starts,
                                              const float *normals, int light_index, int n, float *dirs,
                                              float *bright, float *adirs, int *aok, hipStream_t stream);

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
static void shim_forget(const void *ptr);
extern "C" void rt_managed_free(void *ptr)
{
    if (!ptr) return;
    shim_forget(ptr);   // a sprite / mesh re-created at the same address must be uploaded again
    checkHipErrors(hipDeviceSynchronize());
    (void)hipFree(ptr);
}

// ---------------------------------------------------------------------------
// device-resident scene
// ---------------------------------------------------------------------------
// Frames in flight and the buffers they read. A frame is asynchronous work on the caller's
// stream; the scene's device buffers may be read by frames on several streams at once (two
// frames in flight, a replaying graph). Whoever is about to overwrite or free such a buffer
// first waits for the frames launched so far -- not for the whole device:
//   * every launch records an event into a ring of RT_RING slots; before a slot is re-used the
//     launching stream waits on the event it held, so "the ring's events are done" implies
//     "every earlier frame is done";
//   * rare mutations (sphere list, lights, textures, resolution) wait on the host for the ring
//     (rt_scene_quiesce) and then change the buffers in place;
//   * the one per-frame mutation, the eye-cone table of a moving camera, never waits on the
//     host: it rotates through RT_CONE_SLOTS device buffers, the build kernel is ordered after
//     the slot's last reader with hipStreamWaitEvent, and runs on the frame's own stream.
#define RT_RING 4
#define RT_CONE_SLOTS 3

struct ConeSlot {
    float4 *buf = nullptr;
    size_t cap = 0;                       // float4 units
    float org[3] = {0, 0, 0};
    unsigned long long gen = ~0ull;       // sphere_gen the table was built from
    bool valid = false;
    bool used = false;                    // read by some frame since it was built
    unsigned long long last_use = 0;      // ring sequence number of the last frame that read it
    hipEvent_t built = nullptr;           // the build on the scene's table stream
    bool build_pending = false;           // `built` not yet seen complete: readers wait on it (on the device)
};

#define RT_ORDER_SLOTS 4
#define RT_ORDER_EVERY 32            // an unchanged view: the order is sorted again from fresh durations every so many launches
#ifndef RT_ORDER_MOVING
#define RT_ORDER_MOVING 3            // a view that keeps changing: every so many
#endif
struct TileOrder {
    int key[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // tile width, frame width / height, y0, y1, local rows, interleave
                                                  // count / index / rows, and which kernel: cull, mode, samples
    unsigned *cost = nullptr, *perm = nullptr, *bkey = nullptr, *start = nullptr;   // one allocation: [cap] durations, [cap] order,
                                                                                      // [nb_cap] block keys, [nb_cap] block starts
    size_t cap = 0, nb_cap = 0;
    int n = 0, tiles_x = 0, tiles_y = 0, nb = 0;
    float view[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // camera and sphere list the last launch saw
    int same_view = 0;                           // consecutive launches of that view so far
    int since_sort = 0;                          // launches (all of which recorded durations) since the order was sorted / reset
    bool have_perm = false;
    unsigned long long last_use = 0;
};

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
    float *d_tri_bs = nullptr;
    float *d_tri_nrm = nullptr;
    int n_boxes = 0, n_tris = 0, mesh_has_normals = 0;
    // per-light column blocks (see RtFrameAux::lsorted): one allocation, rebuilt when the
    // sphere list or a light's position changes
    float4 *d_light_tabs = nullptr;
    size_t cap_light_tabs = 0;           // float4 units
    // per-light occluder lists (rt_build_occluder_lists): [n_lights][n] headers, then the lights' entry arrays
    char *d_cand = nullptr;
    size_t cap_cand = 0;                 // bytes
    size_t cand_ent_off[RT_MAX_LIGHTS] = {};   // byte offset of light i's entries in d_cand (headers: i * n * 16)
    bool cand_valid[RT_MAX_LIGHTS] = {};
    float cand_pos[RT_MAX_LIGHTS][3];    // light position each list set was built for
    unsigned long long cand_gen = ~0ull;
    int cand_n_lights = 0;
    unsigned long long sphere_gen = 0;   // bumped whenever the mirrored sphere list changes
    unsigned long long ltab_gen = ~0ull; // sphere_gen the light tables were built from
    int ltab_n_lights = 0;
    float ltab_axis[RT_MAX_LIGHTS][3];   // axis each table was built for
    bool ltab_valid[RT_MAX_LIGHTS] = {};
    // eye cones for the primary rays (see RtFrameConsts::csorted), one table per recent ray origin
    ConeSlot cones[RT_CONE_SLOTS];
    // dx / dy of the primary rays per column / row and sample (RtFrameConsts::dx_tab)
    float *d_raygen = nullptr;
    size_t cap_raygen = 0;               // floats
    int rg_w = 0, rg_h = 0, rg_total = 0;
    float rg_aspect = 0.f;
    // RtFrameAux as uploaded last
    RtFrameAux h_aux;
    RtFrameAux *d_aux = nullptr;
    bool aux_valid = false;
    // where the eye-cone builds run: beside the frames, not in front of them
    hipStream_t table_stream = nullptr;
    // frames in flight
    hipEvent_t ring[RT_RING] = {};
    bool ring_used[RT_RING] = {};
    unsigned long long ring_seq = 0;     // sequence number of the next launch
    // bumped whenever a buffer a recorded graph may point into is rewritten or re-allocated
    unsigned long long epoch = 0;
    // order of the tiles within a launch (RtFrameConsts::tile_perm / tile_cost, rt_tables.hip): per launch
    // layout (which rows of which frame, tile shape) the tiles' wave durations as the frame kernel records
    // them and, rebuilt from those every RT_ORDER_EVERY launches, the order that starts the longest first
    TileOrder orders[RT_ORDER_SLOTS];
    unsigned long long order_clock = 0;      // for least-recently-used replacement
    int tile_order_mode = 1;                 // rt_scene_set_tile_order
    hipEvent_t order_built = nullptr;        // the last rebuild; launches on other streams wait for it on the device
    bool order_pending = false;
#ifdef RT_TUNING
    int tune_no_eye_cones = 0, tune_no_light_columns = 0, tune_table_lds = 0, tune_ablate = 0;
#endif
};

static const int kMaxSpheresLds = (160 * 1024 - RT_WAVES_PER_WG * (RT_LIST_CAP * 20 + 16 * 4 + 64 * 4 + RT_BOX_CAP * 4)) / 16;
static const int kMaxSpheres = 1 << 22;
static const int kMaxSpheresOccluders = 8192;    // the per-sphere occluder lists take n * 128 entries (2 KiB per sphere) per light

extern "C" rt_scene *rt_scene_create(void)
{
    rt_scene *s = new rt_scene();
    memset(&s->h_aux, 0, sizeof s->h_aux);
#ifdef RT_TUNING
    // tuning builds (make EXTRA=-DRT_TUNING, tools/variants.sh) read their switches once per scene;
    // the product library reads no environment
    if (const char *e = getenv("RT_NO_EYE_CONES")) s->tune_no_eye_cones = atoi(e);
    if (const char *e = getenv("RT_NO_LIGHT_COLUMNS")) s->tune_no_light_columns = atoi(e);
    if (const char *e = getenv("RT_TABLE_LDS")) s->tune_table_lds = atoi(e);
    if (const char *e = getenv("RT_ABLATE")) s->tune_ablate = atoi(e);
#endif
    return s;
}

static void free_planes(float *p[3])
{
    for (int i = 0; i < 3; ++i) {
        if (p[i]) (void)hipFree(p[i]);
        p[i] = nullptr;
    }
}

// Wait (on the host) for every frame launched on this scene so far.
int rt_scene_quiesce(rt_scene *s)
{
    for (int i = 0; i < RT_RING; ++i)
        if (s->ring_used[i]) RT_HIP(hipEventSynchronize(s->ring[i]));
    if (s->table_stream) RT_HIP(hipStreamSynchronize(s->table_stream));   // a table build still reading the list
    return RT_OK;
}

// Make `stream` wait for every frame launched on this scene so far (no host wait).
static int stream_wait_all_frames(rt_scene *s, hipStream_t stream)
{
    for (int i = 0; i < RT_RING; ++i)
        if (s->ring_used[i]) RT_HIP(hipStreamWaitEvent(stream, s->ring[i], 0));
    return RT_OK;
}

// A frame has just been enqueued on `stream`: give it the next ring slot. `cone_slot` >= 0:
// the eye-cone table the frame reads.
int rt_scene_note_launch(rt_scene *s, hipStream_t stream, int cone_slot)
{
    const int k = (int)(s->ring_seq % RT_RING);
    if (!s->ring[k]) RT_HIP(hipEventCreateWithFlags(&s->ring[k], hipEventDisableTiming));
    // chain: whoever sees this slot's new event done has also seen the one it replaces
    if (s->ring_used[k]) RT_HIP(hipStreamWaitEvent(stream, s->ring[k], 0));
    RT_HIP(hipEventRecord(s->ring[k], stream));
    s->ring_used[k] = true;
    if (cone_slot >= 0) {
        s->cones[cone_slot].used = true;
        s->cones[cone_slot].last_use = s->ring_seq;
    }
    s->ring_seq++;
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    (void)rt_scene_quiesce(s);
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
    if (s->d_tri_bs) (void)hipFree(s->d_tri_bs);
    if (s->d_tri_nrm) (void)hipFree(s->d_tri_nrm);
    if (s->d_light_tabs) (void)hipFree(s->d_light_tabs);
    if (s->d_cand) (void)hipFree(s->d_cand);
    for (TileOrder &o : s->orders)
        if (o.cost) (void)hipFree(o.cost);
    if (s->order_built) (void)hipEventDestroy(s->order_built);
    for (ConeSlot &c : s->cones) {
        if (c.buf) (void)hipFree(c.buf);
        if (c.built) (void)hipEventDestroy(c.built);
    }
    if (s->table_stream) (void)hipStreamDestroy(s->table_stream);
    if (s->d_raygen) (void)hipFree(s->d_raygen);
    if (s->d_aux) (void)hipFree(s->d_aux);
    for (hipEvent_t e : s->ring)
        if (e) (void)hipEventDestroy(e);
    delete s;
}

// {cx, cy, cz, radius*radius}: the only four numbers sphere::intersect reads
// (kernel.cu:332-334); radius*radius is the same binary32 product either way.
static void pack_spheres(const rt_sphere *src, int n, float4 *dst)
{
    for (int i = 0; i < n; ++i)
        dst[i] = make_float4(src[i].orgin.x, src[i].orgin.y, src[i].orgin.z, src[i].radius * src[i].radius);
}

// ---------------------------------------------------------------------------
// eye cones: which table a frame with ray origin `org` reads, building it if need be
// ---------------------------------------------------------------------------
static bool eye_cones_wanted(const rt_scene *s, const float org[3])
{
#ifdef RT_TUNING
    if (s->tune_no_eye_cones) return false;
#endif
    return s->n_spheres >= 64 && s->h_prev.size() == (size_t)s->n_spheres && std::isfinite(org[0]) &&
           std::isfinite(org[1]) && std::isfinite(org[2]);
}

// Fill `buf` (rt_eye_cones_size(n) float4) for `org`: on the device, on `stream`, when the list
// fits the one-workgroup builder; else on the host with a blocking upload (the caller has made
// sure nothing reads `buf`).
static int build_eye_cones_into(rt_scene *s, const float org[3], float4 *buf, hipStream_t stream)
{
    const int n = s->n_spheres;
    if (((n + 63) & ~63) <= RT_EYE_DEVICE_MAX) {
        RT_HIP(rt_eye_cones_launch(s->d_spheres, n, org, buf, 1024, stream));
        return RT_OK;
    }
    const int n_pad = (n + 63) & ~63, nb = n_pad / RT_BLOCK;
    std::vector<float4> h(rt_eye_cones_size(n));
    rt_build_eye_cones_host(s->h_prev.data(), n, org, h.data(), h.data() + n_pad, reinterpret_cast<int *>(h.data() + n_pad + 2 * nb));
    RT_HIP(hipMemcpyAsync(buf, h.data(), sizeof(float4) * h.size(), hipMemcpyHostToDevice, stream));
    RT_HIP(hipStreamSynchronize(stream));   // `h` goes out of scope
    return RT_OK;
}

// Returns the slot whose table is current for `org` (building it on `stream` if none is), or
// -1 when the scene renders without eye cones. Not inside a stream capture.
static int rt_scene_prepare_eye(rt_scene *s, const float org[3], hipStream_t stream, int *slot_out)
{
    *slot_out = -1;
    if (!eye_cones_wanted(s, org)) return RT_OK;
    for (int i = 0; i < RT_CONE_SLOTS; ++i) {
        const ConeSlot &c = s->cones[i];
        if (c.valid && c.gen == s->sphere_gen && memcmp(org, c.org, sizeof c.org) == 0) {
            *slot_out = i;
            return RT_OK;
        }
    }
    // victim: an unused or stale slot, else the one read longest ago
    int v = 0;
    for (int i = 0; i < RT_CONE_SLOTS; ++i) {
        const ConeSlot &c = s->cones[i], &b = s->cones[v];
        const bool c_free = !c.valid || c.gen != s->sphere_gen, b_free = !b.valid || b.gen != s->sphere_gen;
        if ((c_free && !b_free) || (c_free == b_free && (!c.used || (b.used && c.last_use < b.last_use)))) v = i;
    }
    ConeSlot &c = s->cones[v];
    const size_t total = rt_eye_cones_size(s->n_spheres);
    const bool host_build = ((s->n_spheres + 63) & ~63) > RT_EYE_DEVICE_MAX;
    if (total > c.cap || host_build) {
        const int rc = rt_scene_quiesce(s);   // nothing may still read the buffer that is freed / overwritten from the host
        if (rc != RT_OK) return rc;
    }
    if (total > c.cap) {
        if (c.buf) RT_HIP(hipFree(c.buf));
        c.buf = nullptr;
        c.cap = 0;
        c.valid = false;
        RT_HIP(hipMalloc((void **)&c.buf, sizeof(float4) * total));
        c.cap = total;
        s->epoch++;
    }
    c.valid = false;
    if (host_build) {
        const int rc = build_eye_cones_into(s, org, c.buf, stream);
        if (rc != RT_OK) return rc;
        c.build_pending = false;
    } else {
        // On the scene's table stream, so that the build of frame k+1's table runs beside frame
        // k's kernel instead of in front of frame k+1's (one small workgroup; a moving camera
        // then costs the frames nothing but an event wait). Ordered, on the device only, after
        // the last frame that read this slot and after a sphere-table upload still in flight.
        if (!s->table_stream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            RT_HIP(hipStreamCreateWithPriority(&s->table_stream, hipStreamNonBlocking, hi));
        }
        if (!c.built) RT_HIP(hipEventCreateWithFlags(&c.built, hipEventDisableTiming));
        if (c.used) {
            if (s->ring_seq - c.last_use <= RT_RING) RT_HIP(hipStreamWaitEvent(s->table_stream, s->ring[c.last_use % RT_RING], 0));
            else {
                const int rc = stream_wait_all_frames(s, s->table_stream);
                if (rc != RT_OK) return rc;
            }
        }
        if (s->stage_busy) RT_HIP(hipStreamWaitEvent(s->table_stream, s->stage_done, 0));
        RT_HIP(rt_eye_cones_launch(s->d_spheres, s->n_spheres, org, c.buf, 256, s->table_stream));
        RT_HIP(hipEventRecord(c.built, s->table_stream));
        c.build_pending = true;
    }
    memcpy(c.org, org, sizeof c.org);
    c.gen = s->sphere_gen;
    c.valid = true;
    c.used = false;
    *slot_out = v;
    return RT_OK;
}

// ---------------------------------------------------------------------------
// per-light column tables (host build: lights and the list rarely change)
// ---------------------------------------------------------------------------
static int rt_scene_prepare_lights(rt_scene *s, hipStream_t stream)
{
    const int n = s->n_spheres;
    const int n_pad = (n + 63) & ~63, nb = n_pad / RT_BLOCK;
    bool want = n >= 64 && s->h_prev.size() == (size_t)n;
#ifdef RT_TUNING
    if (s->tune_no_light_columns) want = false;
#endif
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
    // frames still in flight (another stream, a replaying graph) may be reading the old tables
    int rc = rt_scene_quiesce(s);
    if (rc != RT_OK) return rc;
    const size_t total = per_light * (size_t)std::max(1, s->n_lights);
    if (total > s->cap_light_tabs) {
        if (s->d_light_tabs) RT_HIP(hipFree(s->d_light_tabs));
        s->d_light_tabs = nullptr;
        s->cap_light_tabs = 0;
        RT_HIP(hipMalloc((void **)&s->d_light_tabs, sizeof(float4) * total));
        s->cap_light_tabs = total;
    }
    std::vector<float4> h(total);
    for (int i = 0; i < s->n_lights; ++i) {
        s->ltab_valid[i] = usable[i];
        memcpy(s->ltab_axis[i], axis[i], sizeof axis[i]);
        if (usable[i])
            rt_build_light_columns(s->h_prev.data(), n, axis[i], h.data() + per_light * i, h.data() + per_light * i + n_pad);
    }
    for (int i = s->n_lights; i < RT_MAX_LIGHTS; ++i) s->ltab_valid[i] = false;
    RT_HIP(hipMemcpyAsync(s->d_light_tabs, h.data(), sizeof(float4) * total, hipMemcpyHostToDevice, stream));
    RT_HIP(hipStreamSynchronize(stream));   // rare (scene or light change): `h` goes out of scope
    s->ltab_gen = s->sphere_gen;
    s->ltab_n_lights = s->n_lights;
    s->epoch++;
    return RT_OK;
}

// Per-light occluder lists (rt_tables.hip): which spheres a shadow ray from each sphere's surface can hit at all. Built
// on the DEVICE (one wave per sphere and light, from the list-order table that is already there) when the list or a
// light's position changes; the host waits for the build (rare, ~0.1 ms) so that frames on any stream may follow.
static int rt_scene_prepare_occluders(rt_scene *s, hipStream_t stream)
{
    const int n = s->n_spheres;
    bool want = n >= 64 && n <= kMaxSpheresOccluders && s->h_prev.size() == (size_t)n && s->d_spheres;
#ifdef RT_TUNING
    if (s->tune_no_light_columns) want = false;
#endif
    if (!want) {
        for (int i = 0; i < RT_MAX_LIGHTS; ++i) s->cand_valid[i] = false;
        s->cand_gen = ~0ull;
        return RT_OK;
    }
    bool same = (s->cand_gen == s->sphere_gen) && (s->cand_n_lights == s->n_lights);
    for (int i = 0; i < s->n_lights && same; ++i) {
        const float p[3] = {s->lights[i].pos.x, s->lights[i].pos.y, s->lights[i].pos.z};
        same = memcmp(p, s->cand_pos[i], sizeof p) == 0;
    }
    if (same) return RT_OK;
    int rc = rt_scene_quiesce(s);   // frames in flight may be reading the old lists
    if (rc != RT_OK) return rc;
    const size_t hdr_bytes = sizeof(RtCandHdr) * (size_t)n, ent_bytes = sizeof(float4) * (size_t)n * RT_CAND_CAP;
    const size_t bytes = (hdr_bytes + ent_bytes) * (size_t)s->n_lights;
    if (bytes > s->cap_cand) {
        if (s->d_cand) RT_HIP(hipFree(s->d_cand));
        s->d_cand = nullptr;
        s->cap_cand = 0;
        RT_HIP(hipMalloc((void **)&s->d_cand, bytes));
        s->cap_cand = bytes;
        // a wave reads whole steps of 64 from a slot and masks what lies past the count: let that be zeros, once
        RT_HIP(hipMemsetAsync(s->d_cand, 0, bytes, stream));
    }
    if (s->stage_busy) RT_HIP(hipStreamWaitEvent(stream, s->stage_done, 0));   // the table the build reads may still be on its way
    for (int i = 0; i < s->n_lights; ++i) {
        const float p[3] = {s->lights[i].pos.x, s->lights[i].pos.y, s->lights[i].pos.z};
        memcpy(s->cand_pos[i], p, sizeof p);
        s->cand_ent_off[i] = hdr_bytes * (size_t)s->n_lights + ent_bytes * (size_t)i;
        RT_HIP(rt_occluder_lists_launch(s->d_spheres, n, p, reinterpret_cast<RtCandHdr *>(s->d_cand + hdr_bytes * (size_t)i),
                                        reinterpret_cast<float4 *>(s->d_cand + s->cand_ent_off[i]), stream));
        s->cand_valid[i] = true;
    }
    for (int i = s->n_lights; i < RT_MAX_LIGHTS; ++i) s->cand_valid[i] = false;
    RT_HIP(hipStreamSynchronize(stream));
    s->cand_gen = s->sphere_gen;
    s->cand_n_lights = s->n_lights;
    s->epoch++;
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
    std::vector<float4> packed((size_t)n);
    if (n > 0) {
        pack_spheres(host_spheres, n, packed.data());
        if (s->n_spheres == n && s->h_prev.size() == (size_t)n && (int)total <= s->cap_spheres &&
            memcmp(s->h_prev.data(), packed.data(), sizeof(float4) * (size_t)n) == 0)
            return RT_OK;   // unchanged since the last mirror: the device copy is current
    }
    // the table changes: frames in flight on ANY stream may still be reading the device copy
    // (two frames in flight, a replaying graph), so wait for them before it is overwritten or freed
    {
        const int rc = rt_scene_quiesce(s);
        if (rc != RT_OK) return rc;
    }
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
        // the staging buffer is reused: wait for the previous upload to have left it
        if (!s->stage_done) RT_HIP(hipEventCreateWithFlags(&s->stage_done, hipEventDisableTiming));
        if (s->stage_busy) RT_HIP(hipEventSynchronize(s->stage_done));
        float4 *h_orig = s->h_stage, *h_sorted = h_orig + n, *h_blocks = h_sorted + n_pad;
        int *h_idx = reinterpret_cast<int *>(h_blocks + nb);
        memcpy(h_orig, packed.data(), sizeof(float4) * (size_t)n);
        rt_build_sorted_blocks(packed.data(), n, h_sorted, h_blocks, h_idx);
        RT_HIP(hipMemcpyAsync(s->d_spheres, s->h_stage, sizeof(float4) * total, hipMemcpyHostToDevice, stream));
        RT_HIP(hipEventRecord(s->stage_done, stream));
        s->stage_busy = true;
        s->h_prev.swap(packed);
    } else {
        s->h_prev.clear();
    }
    s->sphere_gen++;
    s->epoch++;
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
    { const int rc = rt_scene_quiesce(s); if (rc != RT_OK) return rc; }
    if (!s->d_planes) RT_HIP(hipMalloc((void **)&s->d_planes, sizeof(RtPlaneDev) * RT_MAX_PLANES));
    std::vector<RtPlaneDev> tmp(n ? n : 1);
    for (int i = 0; i < n; ++i)
        tmp[i] = RtPlaneDev{host_planes[i].orgin.x, host_planes[i].orgin.y, host_planes[i].orgin.z,
                            host_planes[i].normal.x, host_planes[i].normal.y, host_planes[i].normal.z, 0.f, 0.f};
    if (n) RT_HIP(hipMemcpy(s->d_planes, tmp.data(), sizeof(RtPlaneDev) * n, hipMemcpyHostToDevice));
    s->n_planes = n;
    s->epoch++;
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
    { const int rc = rt_scene_quiesce(s); if (rc != RT_OK) return rc; }
    if (!s->d_cubes) RT_HIP(hipMalloc((void **)&s->d_cubes, sizeof(RtCubeDev) * RT_MAX_CUBES));
    std::vector<RtCubeDev> tmp(n ? n : 1);
    for (int i = 0; i < n; ++i) {
        const rt_cube &c = host_cubes[i];
        tmp[i] = RtCubeDev{c.bounds[0].x, c.bounds[0].y, c.bounds[0].z, c.bounds[1].x, c.bounds[1].y, c.bounds[1].z,
                           c.orgin.x, c.orgin.y, c.orgin.z, 0.f, 0.f, 0.f};
    }
    if (n) RT_HIP(hipMemcpy(s->d_cubes, tmp.data(), sizeof(RtCubeDev) * n, hipMemcpyHostToDevice));
    s->n_cubes = n;
    s->epoch++;
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
    { const int rc = rt_scene_quiesce(s); if (rc != RT_OK) return rc; }
    s->epoch++;
    if (s->d_tris) RT_HIP(hipFree(s->d_tris));
    if (s->d_boxes) RT_HIP(hipFree(s->d_boxes));
    if (s->d_tri_idx) RT_HIP(hipFree(s->d_tri_idx));
    if (s->d_box_spheres) RT_HIP(hipFree(s->d_box_spheres));
    if (s->d_tri9) RT_HIP(hipFree(s->d_tri9));
    if (s->d_tri_bs) RT_HIP(hipFree(s->d_tri_bs));
    if (s->d_tri_nrm) RT_HIP(hipFree(s->d_tri_nrm));
    s->d_tri_nrm = nullptr;
    s->d_tri9 = nullptr;
    s->d_tri_bs = nullptr;
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
        // bounding sphere of every (leaf, triangle) pair for the per-triangle beam cull (beam_keeps_triangle):
        // centre = centroid, radius = farthest vertex, rounded up; with it the unit normal and kappa, the least
        // |cos| between a ray and the normal for which the cull is valid (slivers and anything non-finite: radius
        // +inf, normal 0, kappa 2 -- never culled)
        std::vector<float> bs(idx.size() * 4 + 4 * 64, 0.f);
        std::vector<float> bn(idx.size() * 4 + 4 * 64, 0.f);   // unit normals (zero = "always edge-on" for degenerate ones)
        for (size_t k = 0; k < idx.size(); ++k) {
            const float *p = &t9[9 * k];
            double c[3], r = 0, e[3][3], len[3];
            for (int a = 0; a < 3; ++a) c[a] = ((double)p[a] + p[3 + a] + p[6 + a]) / 3.0;
            for (int v = 0; v < 3; ++v) {
                double d2 = 0;
                for (int a = 0; a < 3; ++a) d2 += ((double)p[3 * v + a] - c[a]) * ((double)p[3 * v + a] - c[a]);
                r = std::max(r, std::sqrt(d2));
            }
            for (int v = 0; v < 3; ++v) {   // edge v: from vertex v to vertex (v+1)%3
                len[v] = 0;
                for (int a = 0; a < 3; ++a) {
                    e[v][a] = (double)p[3 * ((v + 1) % 3) + a] - p[3 * v + a];
                    len[v] += e[v][a] * e[v][a];
                }
                len[v] = std::sqrt(len[v]);
            }
            const double cx = e[0][1] * e[1][2] - e[0][2] * e[1][1], cy = e[0][2] * e[1][0] - e[0][0] * e[1][2],
                         cz = e[0][0] * e[1][1] - e[0][1] * e[1][0];
            const double area2 = std::sqrt(cx * cx + cy * cy + cz * cz);   // |e0 x e1| = twice the area
            double min_sin = INFINITY;
            for (int v = 0; v < 3; ++v) min_sin = std::min(min_sin, area2 / (len[v] * len[(v + 2) % 3]));
            // kappa = 3e-3 / (smallest corner sine), see beam_keeps_triangle; >= 1 means "never culled"
            const bool good = std::isfinite(r) && std::isfinite(c[0] + c[1] + c[2]) && min_sin > 3.0e-3 && min_sin == min_sin &&
                              area2 > 0 && std::isfinite(area2);
            bs[4 * k + 0] = (float)c[0]; bs[4 * k + 1] = (float)c[1]; bs[4 * k + 2] = (float)c[2];
            bs[4 * k + 3] = good ? (float)(r * 1.001 + 1e-6) : INFINITY;
            if (good) {
                bn[4 * k + 0] = (float)(cx / area2); bn[4 * k + 1] = (float)(cy / area2); bn[4 * k + 2] = (float)(cz / area2);
                bn[4 * k + 3] = (float)(3.0e-3 / min_sin * 1.001);
            } else {
                bn[4 * k + 3] = 2.f;
            }
        }
        RT_HIP(hipMalloc((void **)&s->d_tri_bs, sizeof(float) * bs.size()));
        RT_HIP(hipMemcpy(s->d_tri_bs, bs.data(), sizeof(float) * bs.size(), hipMemcpyHostToDevice));
        RT_HIP(hipMalloc((void **)&s->d_tri_nrm, sizeof(float) * bn.size()));
        RT_HIP(hipMemcpy(s->d_tri_nrm, bn.data(), sizeof(float) * bn.size(), hipMemcpyHostToDevice));
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
    int rc = rt_scene_quiesce(s);
    if (rc != RT_OK) return rc;
    s->epoch++;
    rc = upload_planes(s->d_tex, r, g, b, w, h);
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
    int rc = rt_scene_quiesce(s);
    if (rc != RT_OK) return rc;
    s->epoch++;
    rc = upload_planes(s->d_sky, r, g, b, w, h);
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

// dx and dy of kernel.cu:1624-1625 for every column, row and sample of a frame:
//   dx = aspect*(2*(x+0.5)/(float)width) - 1,  dy = aspect*(2*(y+0.5)/(float)height)*((float)height/width) - 1
// binary64 expressions (the literal 0.5) narrowed to float on assignment. They depend on the
// camera in no way, so a moving camera re-uses them; a new size, aspect or sample count
// rebuilds them (host, W + H divisions per sample) after waiting for the frames in flight.
static int rt_scene_prepare_raygen(rt_scene *s, int width, int height, float aspect, int total)
{
    if (s->d_raygen && s->rg_w == width && s->rg_h == height && s->rg_total == total &&
        memcmp(&s->rg_aspect, &aspect, sizeof aspect) == 0)
        return RT_OK;
    const int rc = rt_scene_quiesce(s);
    if (rc != RT_OK) return rc;
    const size_t need = (size_t)total * ((size_t)width + (size_t)height);
    if (need > s->cap_raygen) {
        if (s->d_raygen) RT_HIP(hipFree(s->d_raygen));
        s->d_raygen = nullptr;
        s->cap_raygen = 0;
        RT_HIP(hipMalloc((void **)&s->d_raygen, sizeof(float) * need));
        s->cap_raygen = need;
    }
    std::vector<float> h(need);
    const double aspect_d = (double)aspect;
    const double width_d = (double)(float)width, height_d = (double)(float)height;
    const double hw_d = (double)((float)height / (float)width);
    for (int k = 0; k < total; ++k) {
        double ox, oy;
        rt_sample_offset(k, total, &ox, &oy);
        float *dx = h.data() + (size_t)k * width, *dy = h.data() + (size_t)total * width + (size_t)k * height;
        for (int x = 0; x < width; ++x) {
            const double tx_d = (2.0 * ((double)x + ox)) / width_d;
            dx[x] = (float)(aspect_d * tx_d - 1.0);
        }
        for (int y = 0; y < height; ++y) {
            const double ty_d = (2.0 * ((double)y + oy)) / height_d;
            dy[y] = (float)((aspect_d * ty_d) * hw_d - 1.0);
        }
    }
    RT_HIP(hipMemcpy(s->d_raygen, h.data(), sizeof(float) * need, hipMemcpyHostToDevice));
    s->rg_w = width;
    s->rg_h = height;
    s->rg_total = total;
    s->rg_aspect = aspect;
    s->epoch++;
    return RT_OK;
}

// margin of the fast texel-index path for a texture dimension of `size` texels (rt_kernels.hip:
// sure_texel): approximation error RT_UV_DELTA plus the rounding of the two float products
static float texel_margin(int size, float delta)
{
    return (float)size * (delta + 0x1.0p-22f) * 1.01f;
}

// Everything of the frame that lives in RtFrameAux (device memory): pure host computation.
static void rt_build_frame_aux(const rt_scene *s, RtFrameAux *ax)
{
    memset(ax, 0, sizeof *ax);
    // castLightRay sample constants, kernel.cu:1453-1454, 1462-1463
    for (int j = 0; j < RT_SHADOW_SAMPLES; ++j) {
        const float jf = (float)j / 10;
        const float phi = jf * 2.f * 3.1415f;
        ax->jf[j] = jf;
        ax->jcos[j] = rtm::cosf_rt(phi);
        ax->jsin[j] = rtm::sinf_rt(phi);
    }
    for (int i = 0; i < s->n_lights; ++i) {
        const rt_light &l = s->lights[i];
        RtLightDev &d = ax->lights[i];
        d.px = l.pos.x; d.py = l.pos.y; d.pz = l.pos.z;
        d.size = l.size;
        d.r = l.r; d.g = l.g; d.b = l.b;
        const float len = std::sqrt(l.pos.x * l.pos.x + l.pos.y * l.pos.y + l.pos.z * l.pos.z);
        d.pos_len = len;
        d.fin = (std::isfinite(l.r) && std::isfinite(l.g) && std::isfinite(l.b)) ? 1.f : 0.f;
        // a light at the origin has no beam axis: NaN makes the kernel skip culling
        d.ux = len > 0 ? l.pos.x / len : NAN;
        d.uy = len > 0 ? l.pos.y / len : NAN;
        d.uz = len > 0 ? l.pos.z / len : NAN;
        // e1 = the coordinate axis least aligned with u, made orthogonal to it; e2 = u x e1 (binary64, rounded once)
        {
            const double u[3] = {d.ux, d.uy, d.uz};
            const int k = (std::fabs(u[0]) <= std::fabs(u[1]) && std::fabs(u[0]) <= std::fabs(u[2])) ? 0 : (std::fabs(u[1]) <= std::fabs(u[2]) ? 1 : 2);
            double t[3] = {0, 0, 0};
            t[k] = 1;
            const double dt = u[k];
            double e1[3] = {t[0] - dt * u[0], t[1] - dt * u[1], t[2] - dt * u[2]};
            const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
            for (double &v : e1) v /= l1;   // NaN for a light at the origin: culling is off for it anyway
            const double e2[3] = {u[1] * e1[2] - u[2] * e1[1], u[2] * e1[0] - u[0] * e1[2], u[0] * e1[1] - u[1] * e1[0]};
            d.e1x = (float)e1[0]; d.e1y = (float)e1[1]; d.e1z = (float)e1[2];
            d.e2x = (float)e2[0]; d.e2y = (float)e2[1]; d.e2z = (float)e2[2];
            d.pad0_ = d.pad1_ = 0.f;
        }
    }
    {
        const int n_pad = (s->n_spheres + 63) & ~63;
        const size_t per_light = (size_t)n_pad + 2 * (size_t)s->n_blocks;
        const bool current = s->d_light_tabs && s->ltab_gen == s->sphere_gen && s->ltab_n_lights == s->n_lights;
        for (int i = 0; i < RT_DEV_MAX_LIGHTS; ++i) {
            const bool on = current && i < s->n_lights && s->ltab_valid[i];
            ax->lsorted[i] = on ? reinterpret_cast<const float *>(s->d_light_tabs + per_light * i) : nullptr;
            ax->lblocks[i] = on ? reinterpret_cast<const float *>(s->d_light_tabs + per_light * i + n_pad) : nullptr;
            const bool con = s->d_cand && s->cand_gen == s->sphere_gen && s->cand_n_lights == s->n_lights && i < s->n_lights && s->cand_valid[i];
            ax->cand_hdr[i] = con ? reinterpret_cast<const RtCandHdr *>(s->d_cand + sizeof(RtCandHdr) * (size_t)s->n_spheres * (size_t)i) : nullptr;
            ax->cand_ent[i] = con ? reinterpret_cast<const float *>(s->d_cand + s->cand_ent_off[i]) : nullptr;
        }
    }
    ax->sky_r = s->d_sky[0]; ax->sky_g = s->d_sky[1]; ax->sky_b = s->d_sky[2];
    ax->sky_w = s->sky_w; ax->sky_h = s->sky_h;
    ax->sky_cx = s->sky_c[0]; ax->sky_cy = s->sky_c[1]; ax->sky_cz = s->sky_c[2];
    ax->sky_r2 = s->sky_radius * s->sky_radius;
    ax->sky_mu_x = texel_margin(s->sky_w, 1.0e-6f);
    ax->sky_mu_y = texel_margin(s->sky_h, 1.0e-6f);
    ax->planes = s->d_planes;
    ax->cubes = s->d_cubes;
    ax->tris = s->d_tris;
    ax->boxes = s->d_boxes;
    ax->tri_idx = s->d_tri_idx;
    ax->box_spheres = s->d_box_spheres;
    ax->tri9 = s->d_tri9;
    ax->tri_bs = s->d_tri_bs;
    ax->tri_nrm = s->d_tri_nrm;
}

// Bring the device copy of RtFrameAux up to date (a camera move never changes it).
static int rt_scene_sync_aux(rt_scene *s)
{
    RtFrameAux ax;
    rt_build_frame_aux(s, &ax);
    if (s->aux_valid && memcmp(&ax, &s->h_aux, sizeof ax) == 0) return RT_OK;
    const int rc = rt_scene_quiesce(s);
    if (rc != RT_OK) return rc;
    if (!s->d_aux) RT_HIP(hipMalloc((void **)&s->d_aux, sizeof(RtFrameAux)));
    RT_HIP(hipMemcpy(s->d_aux, &ax, sizeof ax, hipMemcpyHostToDevice));
    s->h_aux = ax;
    s->aux_valid = true;
    s->epoch++;
    return RT_OK;
}

// The by-value frame uniforms. Pure host computation: no device call, the scene is not
// changed. `cones`: the eye-cone table the frame reads (its org must be the frame's), or null.
int rt_build_frame_consts(const rt_scene *s, const rt_frame_desc *fd, const float4 *cones, RtFrameConsts *fc)
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
            owns_rows = b > 0 && (long long)o.interleave_index * b < (y1 - y0);
        }
        if (owns_rows && !fd->pixels && !o.rgba && !o.packed24) {
            rt_set_error("rt_scene_render: no output buffer (pixels, opts.rgba and opts.packed24 are all null)");
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
    fc->flags = (o.accumulate ? RT_FLAG_ACCUMULATE : 0) |
                (((fd->pixels || o.packed24) && o.resolve >= 0) ? RT_FLAG_RESOLVE : 0) |
                (o.force_slow_path ? RT_FLAG_FORCE_SLOW : 0) | (s->mesh_has_normals ? RT_FLAG_MESH_NORMALS : 0);
    fc->local_rows = y1 - y0;
    fc->il_count = 1;      // a contiguous band is the interleave of one rank (the kernel has one row formula)
    fc->il_index = 0;
    fc->il_rows = 16;
    if (o.interleave_count > 1) {
        const int b = o.interleave_rows > 0 ? o.interleave_rows : 16;
        // with a row band the blocks are dealt from the band's first row (which must start a block)
        if (b < 16 || (b & (b - 1)) != 0 || y0 % b != 0 || o.interleave_index < 0 || o.interleave_index >= o.interleave_count) {
            rt_set_error("rt_scene_render: bad interleave (count=%d index=%d rows=%d: a power of two >= 16; y0=%d must be a multiple of rows)",
                         o.interleave_count, o.interleave_index, b, y0);
            return RT_ERR_INVALID;
        }
        fc->il_count = o.interleave_count;
        fc->il_index = o.interleave_index;
        fc->il_rows = b;
        const int band = y1 - y0;
        int rows = 0;   // rows of the blocks this rank owns
        for (int k = o.interleave_index; k * b < band; k += o.interleave_count)
            rows += (band - k * b < b) ? band - k * b : b;
        fc->local_rows = rows;   // may be 0 (more ranks than row blocks): the launch is then skipped
    }
    fc->n_planes = s->n_planes;
    fc->n_cubes = s->n_cubes;
    fc->n_boxes = s->n_boxes;
#ifdef RT_TUNING
    fc->ablate = s->tune_ablate;   // timing experiments only: output is wrong when set
#endif

    // kernel.cu:1624-1625 through the raygen tables; :1629-1631: eyePos = (0,0,-1/aspect); dir - eyePos; eyePos + cam.Org
    const bool rg = s->d_raygen && s->rg_w == fd->width && s->rg_h == fd->height && s->rg_total == total &&
                    memcmp(&s->rg_aspect, &fd->aspect, sizeof(float)) == 0;
    fc->dx_tab = rg ? s->d_raygen : nullptr;
    fc->dy_tab = rg ? s->d_raygen + (size_t)total * fd->width : nullptr;
    const float ez = -1.f / fd->aspect;
    fc->eye_nz = 0.f - ez;
    float org[3];
    rt_ray_origin(fd, org);
    fc->org_x = org[0];
    fc->org_y = org[1];
    fc->org_z = org[2];
    // camera::rotateDir, kernel.cu:249-250
    const float yawRad = (float)(fd->cam.Camyaw * (3.1415 / 180));
    const float pitchRad = (float)(fd->cam.Campitch * (3.1415 / 180));
    fc->cos_pitch = rtm::cosf_rt(pitchRad);
    fc->sin_pitch = rtm::sinf_rt(pitchRad);
    fc->cos_yaw = rtm::cosf_rt(yawRad);
    fc->sin_yaw = rtm::sinf_rt(yawRad);

    fc->tex_r = s->d_tex[0]; fc->tex_g = s->d_tex[1]; fc->tex_b = s->d_tex[2];
    fc->tex_w = s->tex_w; fc->tex_h = s->tex_h;
    fc->tex_mu_x = texel_margin(s->tex_w, 5.0e-7f);   // RT_UV_DELTA of rt_kernels.hip
    fc->tex_mu_y = texel_margin(s->tex_h, 5.0e-7f);
    {
        const int n_pad = (s->n_spheres + 63) & ~63;
        const float4 *base = s->d_spheres;
        fc->sorted = base ? reinterpret_cast<const float *>(base + s->n_spheres) : nullptr;
        fc->blocks = base ? reinterpret_cast<const float *>(base + s->n_spheres + n_pad) : nullptr;
        fc->orig_idx = base ? reinterpret_cast<const int *>(base + s->n_spheres + n_pad + s->n_blocks) : nullptr;
        fc->n_blocks = s->n_blocks;
        fc->csorted = cones ? reinterpret_cast<const float *>(cones) : nullptr;
        fc->cblocks = cones ? reinterpret_cast<const float *>(cones + n_pad) : nullptr;
        fc->corig = cones ? reinterpret_cast<const int *>(cones + n_pad + 2 * (size_t)s->n_blocks) : nullptr;
        fc->cone_kcap = (float)RT_CONE_KCAP;
    }
    fc->aux = s->d_aux;
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

// Which instantiation renders this frame (rt_kernels.hip: MODE, FEAT, TABLDS).
int rt_frame_kernel_choice(const rt_scene *s, const rt_frame_desc *fd, RtKernelChoice *kc)
{
    int rc = tile_from_opts(fd->opts, &kc->tile);
    if (rc != RT_OK) return rc;
    kc->cull = (fd->opts.cull == 0) ? 0 : 1;
    kc->mode = fd->opts.stats ? (fd->opts.profile ? 3 : 1) : (fd->opts.force_slow_path ? 2 : 0);
    kc->feat = s->n_boxes > 0 ? 2 : ((s->n_planes > 0 || s->n_cubes > 0) ? 1 : 0);
    // the opt-in approximate mode exists for the product configuration only; anything else renders exactly
    if (fd->opts.fast == 1 && kc->mode == 0 && kc->cull && kc->tile == 8 && kc->feat < 2 && fd->opts.table_lds != 1) kc->mode = 4;
    // whole-table LDS staging (north_star's first design, measured slower: DESIGN.md section 3)
    // is opt-in per launch and only when the table fits next to the survivor lists
    kc->table_lds = (fd->opts.table_lds == 1 && s->n_spheres <= kMaxSpheresLds) ? 1 : 0;
#ifdef RT_TUNING
    if (s->tune_table_lds && s->n_spheres <= kMaxSpheresLds) kc->table_lds = 1;
#else
    if (kc->mode == 3) {
        rt_set_error("rt_scene_render: opts.profile (phase stamps) needs a tuning build of the library (make EXTRA=-DRT_TUNING)");
        return RT_ERR_UNSUPPORTED;
    }
#endif
    if (fd->opts.stats && fd->opts.force_slow_path) {
        rt_set_error("rt_scene_render: stats and force_slow_path exclude each other");
        return RT_ERR_UNSUPPORTED;
    }
    return RT_OK;
}

// Everything a frame needs on the device that is NOT the eye-cone table: per-light tables,
// raygen tables, RtFrameAux. Host waits happen here, and only when something changed.
int rt_scene_prepare_static(rt_scene *s, const rt_frame_desc *fd, hipStream_t stream)
{
    if (!s || !fd || fd->width <= 0 || fd->height <= 0) {
        rt_set_error("rt_scene_render: null scene or bad frame");
        return RT_ERR_INVALID;
    }
    int rc = RT_OK;
    if (fd->opts.cull != 0) {
        rc = rt_scene_prepare_lights(s, stream);
        if (rc != RT_OK) return rc;
        rc = rt_scene_prepare_occluders(s, stream);
        if (rc != RT_OK) return rc;
    }
    const int spp = fd->opts.spp > 0 ? fd->opts.spp : 1;
    const int total = fd->opts.sample_total > 0 ? fd->opts.sample_total : spp;
    if (total < 1 || total > RT_MAX_SPP) {
        rt_set_error("rt_scene_render: bad sample total %d (max %d)", total, RT_MAX_SPP);
        return RT_ERR_INVALID;
    }
    rc = rt_scene_prepare_raygen(s, fd->width, fd->height, fd->aspect, total);
    if (rc != RT_OK) return rc;
    return rt_scene_sync_aux(s);
}

// The frame kernel records every tile's wave duration (two s_memtime and one store per wave: free). From the
// previous launch's durations the blocks of 16 x 16 tiles are sorted "longest tile first" (one workgroup on the
// launching stream, rt_tables.hip) and the launches start their tiles in that order: sorted again after 2, 4, 8, 16,
// 32, 64, 96, ... launches of an unchanged view (camera, sphere list), every RT_ORDER_MOVING launches while the view
// keeps changing. Scheduling only -- every tile is rendered once, by the same instructions. Ordering against frames
// in flight: the sort waits (on the device) for every frame launched so far, which read the old order; frames launched
// afterwards on other streams wait for the sort's event.
static int rt_scene_prepare_tile_order(rt_scene *s, const RtKernelChoice &kc, RtFrameConsts *fc, hipStream_t stream)
{
    const int tile_w = kc.tile, th = 64 / tile_w;
    const int tiles_x = (fc->width + tile_w - 1) / tile_w, tiles_y = (fc->local_rows + th - 1) / th;
    const int nbx = (tiles_x + RT_TILE_ORDER_BLOCK - 1) / RT_TILE_ORDER_BLOCK, nby = (tiles_y + RT_TILE_ORDER_BLOCK - 1) / RT_TILE_ORDER_BLOCK;
    if (tiles_x > 0xffff || tiles_y > 0xffff || (long long)nbx * nby > RT_TILE_ORDER_MAX_BLOCKS) return RT_OK;   // grid order
    const int n = tiles_x * tiles_y, nb = nbx * nby;
    const int key[12] = {tile_w, fc->width, fc->height, fc->y0, fc->y1, fc->local_rows, fc->il_count, fc->il_index, fc->il_rows,
                         kc.cull, kc.mode, fc->spp};
    // what the durations depend on from frame to frame: the view and the sphere list
    const float view[8] = {fc->org_x, fc->org_y, fc->org_z, fc->cos_pitch, fc->sin_pitch, fc->cos_yaw, fc->sin_yaw,
                           (float)(s->sphere_gen & 0xffffff)};
    TileOrder *t = nullptr, *lru = &s->orders[0];
    for (TileOrder &o : s->orders) {
        if (o.cap && memcmp(o.key, key, sizeof key) == 0) t = &o;
        if (o.last_use < lru->last_use) lru = &o;
    }
    if (s->order_pending) {   // an order being sorted (any layout: one event) precedes this launch
        if (hipEventQuery(s->order_built) == hipSuccess) s->order_pending = false;
        else RT_HIP(hipStreamWaitEvent(stream, s->order_built, 0));
        (void)hipGetLastError();
    }
    if (!t) {                 // a new layout takes the least recently used slot
        t = lru;
        int rc = stream_wait_all_frames(s, stream);   // frames that still write into the slot's old arrays
        if (rc != RT_OK) return rc;
        if ((size_t)n > t->cap || (size_t)nb > t->nb_cap) {
            rc = rt_scene_quiesce(s);                  // re-allocation: nothing may still use the old arrays
            if (rc != RT_OK) return rc;
            if (t->cost) RT_HIP(hipFree(t->cost));
            t->cost = t->perm = t->bkey = t->start = nullptr;
            t->cap = t->nb_cap = 0;
            RT_HIP(hipMalloc((void **)&t->cost, sizeof(unsigned) * (2 * (size_t)n + 2 * (size_t)nb)));
            t->cap = (size_t)n;
            t->nb_cap = (size_t)nb;
        }
        t->perm = t->cost + t->cap;
        t->bkey = t->perm + t->cap;
        t->start = t->bkey + t->nb_cap;
        RT_HIP(hipMemsetAsync(t->cost, 0, sizeof(unsigned) * t->cap, stream));
        memcpy(t->key, key, sizeof key);
        memcpy(t->view, view, sizeof view);
        t->n = n;
        t->nb = nb;
        t->tiles_x = tiles_x;
        t->tiles_y = tiles_y;
        t->same_view = 0;
        t->since_sort = 0;
        t->have_perm = false;
        if (!s->order_built) RT_HIP(hipEventCreateWithFlags(&s->order_built, hipEventDisableTiming));
        RT_HIP(hipEventRecord(s->order_built, stream));   // launches on other streams: after the reset
        s->order_pending = true;
    } else {
        if (memcmp(t->view, view, sizeof view) != 0) {
            memcpy(t->view, view, sizeof view);
            t->same_view = 0;
        }
        // launches of this view so far: t->same_view; launches since the last sort: t->since_sort (all recorded durations)
        const int k = t->same_view;
        const bool due = k == 0 ? t->since_sort >= (t->have_perm ? RT_ORDER_MOVING : 1)                 // a view that changes
                                : (k >= 2 && ((k & (k - 1)) == 0 || k % RT_ORDER_EVERY == 0)) || !t->have_perm;
        if (due && t->since_sort >= 1) {
            int rc = stream_wait_all_frames(s, stream);
            if (rc != RT_OK) return rc;
            RT_HIP(rt_tile_order_launch(t->cost, t->bkey, t->start, t->perm, t->tiles_x, t->tiles_y, stream));
            if (!s->order_built) RT_HIP(hipEventCreateWithFlags(&s->order_built, hipEventDisableTiming));
            RT_HIP(hipEventRecord(s->order_built, stream));
            s->order_pending = true;
            t->have_perm = true;
            t->since_sort = 0;
        }
    }
    t->since_sort++;
    t->same_view++;
    t->last_use = ++s->order_clock;
    fc->tile_cost = t->cost;
    fc->tile_perm = t->have_perm ? t->perm : nullptr;
    return RT_OK;
}

extern "C" int rt_scene_render(rt_scene *s, const rt_frame_desc *fd, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!s || !fd) {
        rt_set_error("rt_scene_render: null scene or frame");
        return RT_ERR_INVALID;
    }
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream) (void)hipStreamIsCapturing(stream, &cs);
        if (cs != hipStreamCaptureStatusNone) {
            rt_set_error("rt_scene_render: the stream is being captured; use rt_graph_capture, which records the frame as graph nodes");
            return RT_ERR_UNSUPPORTED;
        }
    }
    if (s->stage_busy) {   // a sphere-table upload enqueued on some stream: order this frame after it
        if (hipEventQuery(s->stage_done) == hipSuccess) s->stage_busy = false;
        else RT_HIP(hipStreamWaitEvent(stream, s->stage_done, 0));
        (void)hipGetLastError();   // hipEventQuery reports "not ready" as an error
    }
    int rc = rt_scene_prepare_static(s, fd, stream);
    if (rc != RT_OK) return rc;
    int slot = -1;
    if (fd->opts.cull != 0) {
        float org[3];
        rt_ray_origin(fd, org);
        rc = rt_scene_prepare_eye(s, org, stream, &slot);
        if (rc != RT_OK) return rc;
    }
    if (slot >= 0 && s->cones[slot].build_pending) {   // the table's build (table stream) precedes its readers
        ConeSlot &c = s->cones[slot];
        if (hipEventQuery(c.built) == hipSuccess) c.build_pending = false;
        else RT_HIP(hipStreamWaitEvent(stream, c.built, 0));
        (void)hipGetLastError();
    }
    RtFrameConsts fc;
    rc = rt_build_frame_consts(s, fd, slot >= 0 ? s->cones[slot].buf : nullptr, &fc);
    if (rc != RT_OK) return rc;
    RtKernelChoice kc;
    rc = rt_frame_kernel_choice(s, fd, &kc);
    if (rc != RT_OK) return rc;
    if (fc.local_rows == 0) return RT_OK;   // this rank owns no rows of the frame
    if (s->tile_order_mode != 0 && !kc.table_lds) {
        rc = rt_scene_prepare_tile_order(s, kc, &fc, stream);
        if (rc != RT_OK) return rc;
    }
    RT_HIP(rt_dev_launch_trace(&fc, s->d_spheres, kc.tile, kc.cull, kc.mode, kc.table_lds, kc.feat, stream));
    return rt_scene_note_launch(s, stream, slot);
}

int rt_scene_tile_order_mode(const rt_scene *s) { return s->tile_order_mode; }

extern "C" int rt_scene_set_tile_order(rt_scene *s, int mode)
{
    if (!s || (mode != 0 && mode != 1)) {
        rt_set_error("rt_scene_set_tile_order: null scene or mode %d not in {0, 1}", mode);
        return RT_ERR_INVALID;
    }
    s->tile_order_mode = mode;
    return RT_OK;
}

// For rt_graph.cpp: the scene's buffers a graph node needs.
const float4 *rt_scene_sphere_table(const rt_scene *s) { return s->d_spheres; }
int rt_scene_sphere_count(const rt_scene *s) { return s->n_spheres; }
unsigned long long rt_scene_epoch(const rt_scene *s) { return s->epoch; }
bool rt_scene_wants_eye_cones(const rt_scene *s, const float org[3]) { return eye_cones_wanted(s, org); }
int rt_scene_build_eye_cones_host(rt_scene *s, const float org[3], float4 *buf, hipStream_t stream)
{
    return build_eye_cones_into(s, org, buf, stream);
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

// memManager::operator delete on something the shim has mirrored: the next launch re-uploads.
static void shim_forget(const void *ptr)
{
    for (int i = 0; i < 3; ++i) {
        if (ptr == g_shim.tex_key[i]) g_shim.tex_key[0] = g_shim.tex_key[1] = g_shim.tex_key[2] = nullptr;
        if (ptr == g_shim.sky_key[i]) g_shim.sky_key[0] = g_shim.sky_key[1] = g_shim.sky_key[2] = nullptr;
    }
    if (ptr == g_shim.mesh_key) {
        g_shim.mesh_key = nullptr;
        g_shim.mesh_polys = g_shim.mesh_boxes = -1;
    }
}

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

extern "C" int rt_debug_shortcuts(int what, unsigned seed, long long n, unsigned long long out[4])
{
    if (what < 0 || what > 2 || !out || n < 0) return RT_ERR_INVALID;
    DevBuf<unsigned long long> d;
    int rc = d.alloc(4);
    if (rc != RT_OK) return rc;
    RT_HIP(hipMemset(d.p, 0, sizeof(unsigned long long) * 4));
    RT_HIP(rt_dev_launch_dbg_shortcuts(what, seed, n, d.p, nullptr));
    RT_HIP(hipMemcpy(out, d.p, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

// The occluder lists of one light (rt_build_occluder_lists; host only, no GPU): counts[i] = entries of sphere i's list
// (-1: none), kcaps[i] = the beam slope it holds for, members: n x cap ints, the list positions of the first `cap`
// members of every list (an entry is identified by its four floats: the first sphere of the table with those).
extern "C" int rt_debug_occluder_lists_ex(const rt_sphere *spheres, int n, const rt_light *light, int *counts, float *kcaps, int *members, int cap,
                                          int *offsets, int *entries_allocated)
{
    if (n <= 0 || !spheres || !light || !counts || !kcaps || (cap > 0 && !members)) return RT_ERR_INVALID;
    std::vector<float4> tab((size_t)n);
    pack_spheres(spheres, n, tab.data());
    std::vector<RtCandHdr> hdr;
    std::vector<float4> ent;
    const float p[3] = {light->pos.x, light->pos.y, light->pos.z};
    rt_build_occluder_lists(tab.data(), n, p, hdr, ent);
    if (entries_allocated) *entries_allocated = (int)ent.size();
    for (int i = 0; i < n; ++i) {
        counts[i] = hdr[(size_t)i].count;
        kcaps[i] = hdr[(size_t)i].kcap;
        if (offsets) offsets[i] = hdr[(size_t)i].offset;
        for (int k = 0; k < cap; ++k) members[(size_t)i * cap + k] = -1;
        for (int k = 0; k < hdr[(size_t)i].count && k < cap; ++k) {
            const float4 e = ent[(size_t)hdr[(size_t)i].offset + k];
            for (int j = 0; j < n; ++j)
                if (memcmp(&tab[(size_t)j], &e, sizeof e) == 0) { members[(size_t)i * cap + k] = j; break; }
        }
    }
    return RT_OK;
}

// The per-sphere beam slopes (RtCandHdr::kbeam; -1: none) as the host builder (host_kbeam, or NULL) and the device builder
// (device_kbeam, or NULL: no GPU needed then) compute them, and the pieces of the bound for tests: the spread at ONE start.
extern "C" int rt_debug_sphere_beam_slopes(const rt_sphere *spheres, int n, const rt_light *light, float *host_kbeam, float *device_kbeam)
{
    if (n <= 0 || !spheres || !light) return RT_ERR_INVALID;
    std::vector<float4> tab((size_t)n);
    pack_spheres(spheres, n, tab.data());
    const float p[3] = {light->pos.x, light->pos.y, light->pos.z};
    if (host_kbeam) {
        std::vector<RtCandHdr> hdr;
        std::vector<float4> ent;
        rt_build_occluder_lists(tab.data(), n, p, hdr, ent);
        for (int i = 0; i < n; ++i) host_kbeam[i] = hdr[(size_t)i].kbeam;
    }
    if (device_kbeam) {
        DevBuf<float4> dtab, dent;
        DevBuf<RtCandHdr> dhdr;
        int rc;
        if ((rc = dtab.alloc((size_t)n)) || (rc = dent.alloc((size_t)n * RT_CAND_CAP)) || (rc = dhdr.alloc((size_t)n))) return rc;
        RT_HIP(hipMemcpy(dtab.p, tab.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
        RT_HIP(rt_occluder_lists_launch(dtab.p, n, p, dhdr.p, dent.p, nullptr));
        std::vector<RtCandHdr> hdr((size_t)n);
        RT_HIP(hipMemcpy(hdr.data(), dhdr.p, sizeof(RtCandHdr) * (size_t)n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) device_kbeam[i] = hdr[(size_t)i].kbeam;
    }
    return RT_OK;
}
extern "C" double rt_debug_sphere_beam_slope(const double lpos[3], const double centre[3], double r0)
{
    return rt_sphere_beam_slope(lpos, centre, r0);
}
extern "C" double rt_debug_beam_sine(const double lpos[3], const double start[3], double *sigma, double *frob, double m9[9])
{
    return rt_beam_sine_at_start(lpos, start, sigma, frob, m9);
}

// The same lists as the DEVICE builds them (rt_occluder_lists_launch: what the scene uses), downloaded: counts, kcaps and
// the first `cap` members of every list as list positions (device order = table order).
extern "C" int rt_debug_occluder_lists_device(const rt_sphere *spheres, int n, const rt_light *light, int *counts, float *kcaps, int *members, int cap)
{
    if (n <= 0 || !spheres || !light || !counts || !kcaps || (cap > 0 && !members)) return RT_ERR_INVALID;
    std::vector<float4> tab((size_t)n);
    pack_spheres(spheres, n, tab.data());
    DevBuf<float4> dtab, dent;
    DevBuf<RtCandHdr> dhdr;
    int rc;
    if ((rc = dtab.alloc((size_t)n)) || (rc = dent.alloc((size_t)n * RT_CAND_CAP)) || (rc = dhdr.alloc((size_t)n))) return rc;
    RT_HIP(hipMemcpy(dtab.p, tab.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
    RT_HIP(hipMemset(dent.p, 0, sizeof(float4) * (size_t)n * RT_CAND_CAP));
    const float p[3] = {light->pos.x, light->pos.y, light->pos.z};
    RT_HIP(rt_occluder_lists_launch(dtab.p, n, p, dhdr.p, dent.p, nullptr));
    std::vector<RtCandHdr> hdr((size_t)n);
    std::vector<float4> ent((size_t)n * RT_CAND_CAP);
    RT_HIP(hipMemcpy(hdr.data(), dhdr.p, sizeof(RtCandHdr) * (size_t)n, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(ent.data(), dent.p, sizeof(float4) * ent.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        counts[i] = hdr[(size_t)i].count;
        kcaps[i] = hdr[(size_t)i].kcap;
        for (int k = 0; k < cap; ++k) members[(size_t)i * cap + k] = -1;
        for (int k = 0; k < hdr[(size_t)i].count && k < cap; ++k) {
            const float4 e = ent[(size_t)hdr[(size_t)i].offset + k];
            for (int j = 0; j < n; ++j)
                if (memcmp(&tab[(size_t)j], &e, sizeof e) == 0) { members[(size_t)i * cap + k] = j; break; }
        }
    }
    return RT_OK;
}

extern "C" int rt_debug_occluder_lists(const rt_sphere *spheres, int n, const rt_light *light, int *counts, float *kcaps, int *members, int cap)
{
    return rt_debug_occluder_lists_ex(spheres, n, light, counts, kcaps, members, cap, nullptr, nullptr);
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

static int debug_light_impl(const rt_sphere *spheres, int n_spheres, const rt_vec3 *start,
                            const rt_vec3 *normal, const rt_light *light, int n, float *dirs,
                            float *brightness, float *approx_dirs, int *approx_ok)
{
    if (n <= 0 || n_spheres < 0 || !start || !normal || !light || !dirs || !brightness) return RT_ERR_INVALID;
    rt_scene sc;
    sc.lights[0] = *light;
    sc.n_lights = 1;
    RtFrameAux ax;
    rt_build_frame_aux(&sc, &ax);
    DevBuf<RtFrameAux> dax;
    int rc = dax.alloc(1);
    if (rc != RT_OK) return rc;
    RT_HIP(hipMemcpy(dax.p, &ax, sizeof ax, hipMemcpyHostToDevice));
    RtFrameConsts fc;
    memset(&fc, 0, sizeof fc);
    fc.n_lights = 1;
    fc.aux = dax.p;
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
    DevBuf<float> dadirs;
    DevBuf<int> daok;
    if (approx_dirs && ((rc = dadirs.alloc(30 * (size_t)n)) || (rc = daok.alloc(10 * (size_t)n)))) return rc;
    RT_HIP(rt_dev_launch_dbg_light(&fc, dtab.p, dstart.p, dnormal.p, 0, n, ddirs.p, dbright.p, approx_dirs ? dadirs.p : nullptr,
                                   approx_dirs ? daok.p : nullptr, nullptr));
    RT_HIP(hipMemcpy(dirs, ddirs.p, sizeof(float) * 30 * n, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(brightness, dbright.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (approx_dirs) {
        RT_HIP(hipMemcpy(approx_dirs, dadirs.p, sizeof(float) * 30 * n, hipMemcpyDeviceToHost));
        RT_HIP(hipMemcpy(approx_ok, daok.p, sizeof(int) * 10 * n, hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

extern "C" int rt_debug_light(const rt_sphere *spheres, int n_spheres, const rt_vec3 *start,
                              const rt_vec3 *normal, const rt_light *light, int n, float *dirs,
                              float *brightness)
{
    return debug_light_impl(spheres, n_spheres, start, normal, light, n, dirs, brightness, nullptr, nullptr);
}

// The exact sample directions next to the pre-pass's approximate ones (frame kernel: setup_approx / direction_approx)
// and the pre-pass's guard flags, for the error bound RT_PRE_DELTA (tests only).
extern "C" int rt_debug_light_prepass(const rt_vec3 *start, const rt_light *light, int n, float *dirs, float *approx_dirs, int *approx_ok)
{
    if (!approx_dirs || !approx_ok || n <= 0) return RT_ERR_INVALID;
    std::vector<rt_vec3> normal((size_t)n, rt_vec3{0.f, 1.f, 0.f});
    std::vector<float> bright((size_t)n);
    return debug_light_impl(nullptr, 0, start, normal.data(), light, n, dirs, bright.data(), approx_dirs, approx_ok);
}
