// rt_frame.cpp -- frame driver: the host mirror of the reference's
// `onStart()` / `update()` pair and the globals around them
// (/root/reference/kernel.cuh:3-4, kernel.cu:1692-1714, 1762-1792), plus the
// memManager-derived texture classes (sprite.h:11-47, Sprite.cpp:13-65).
//
// What changed relative to the reference's update(): the framebuffer and the
// pinned present staging are allocated once per resolution instead of
// malloc/free every frame (kernel.cu:1775-1776, 1789-1790), the kernel writes
// device memory and the frame reaches setPixelBuff() through one asynchronous
// D2H copy instead of managed-page migration (kernel.cu:1788), and the lights
// travel as kernel arguments instead of a per-frame cudaMalloc+cudaMemcpy
// (kernel.cu:1776-1778). Behaviour at the boundary is unchanged: update()
// returns after the frame has been handed to setPixelBuff().
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_engine.h"
#include "../../include/rt_kernel.h"
#include "../../include/rt_memmanager.h"
#include "../../include/rt_window.h"
#include "rt_internal.h"

// ---- buffer / sprite (Sprite.cpp:13-52) ----
buffer::buffer(float *pixels, int length)
{
    size = length * (int)sizeof(float);
    data = (float *)rt_managed_alloc((size_t)size);
    if (data && pixels) memcpy(data, pixels, (size_t)size);
}

sprite::sprite(std::string file)
{
    rBuff = gBuff = bBuff = nullptr;
    width = height = 0;
    float *r = nullptr, *g = nullptr, *b = nullptr;
    int kind = -1;
    if (file == "synthetic:object") kind = 0;
    if (file == "synthetic:sky") kind = 1;
    if (kind >= 0) {
        rt_synth_texture_size(kind, &width, &height);
        const size_t n = (size_t)width * height;
        r = (float *)malloc(n * sizeof(float));
        g = (float *)malloc(n * sizeof(float));
        b = (float *)malloc(n * sizeof(float));
        rt_synth_texture(kind, r, g, b);
    } else if (rt_load_ppm(file.c_str(), &r, &g, &b, &width, &height) != RT_OK) {
        fprintf(stderr, "sprite: %s\n", rt_last_error());
        rt_check(1, "sprite(file)", __FILE__, __LINE__);
        return;
    }
    rBuff = new buffer(r, width * height);
    gBuff = new buffer(g, width * height);
    bBuff = new buffer(b, width * height);
    rt_free_planes(r, g, b);
}

int sprite::getBytes() { return (int)sizeof(float) * width * height * 3; }

static_assert(sizeof(buffer) == sizeof(rt_buffer), "buffer must match rt_buffer");
static_assert(sizeof(sprite) == sizeof(rt_sprite), "sprite must match rt_sprite");

// ---- globals, kernel.cu:1692-1702 ----
static int light_size = 3;
static rt_light lights[RT_MAX_LIGHTS];
static rt_camera cam = {{4, 3, 10}, {0, 0, 1}, 0.f, 180.f, -20.f};   // kernel.cu:1695, :261
static rt_object *objs = nullptr;
static rt_skybox *Skybox = nullptr;
static float aspect = 0.f;

// asset files (kernel.cu:1700, 1706, 1181: hard-coded C:\\ paths in the reference): set through
// rt_config_set_assets(), else the environment of the APPLICATION SHELL (RT_OBJECT_TEXTURE,
// RT_SKY_TEXTURE, RT_MESH_OBJ, read once in onStart), else the synthetic stand-ins
static std::string cfg_object_texture, cfg_sky_texture, cfg_mesh_obj;
static int cfg_gpus = 0;                 // 0: not set (RT_GPUS of the application's environment when onStart() ran, else 1)
static int env_gpus = 0;                 // RT_GPUS as onStart() found it (the environment is not read per frame)
static int cfg_sphere_count = 1024;      // the reference ships 0 (kernel.cu:1231); BASELINE configs set it
static unsigned int cfg_seed = 1;        // un-seeded MSVC rand() starts from state 1

// ---- persistent per-resolution frame resources ----
static struct FrameRes {
    uint32_t *d_pixels = nullptr;
    uint32_t *h_pixels = nullptr;   // pinned
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;          // device-to-host copies of finished row bands
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t band_done[4] = {nullptr, nullptr, nullptr, nullptr};
    double last_ms = 0.0;
} fr;

extern "C" int rt_config_set_sphere_count(int n)
{
    if (n < 0) return RT_ERR_INVALID;
    cfg_sphere_count = n;
    return RT_OK;
}
extern "C" int rt_config_set_seed(unsigned int seed)
{
    cfg_seed = seed;
    return RT_OK;
}
extern "C" int rt_config_set_assets(const char *object_texture, const char *sky_texture, const char *mesh_obj)
{
    cfg_object_texture = object_texture ? object_texture : "";
    cfg_sky_texture = sky_texture ? sky_texture : "";
    cfg_mesh_obj = mesh_obj ? mesh_obj : "";
    return RT_OK;
}
extern "C" int rt_config_set_gpus(int n)
{
    if (n < -64 || n > 64) return RT_ERR_INVALID;
    cfg_gpus = n;   // n < 0: rehearsal, |n| shares of the frame all rendered on device 0 (peer-copy transport)
    return RT_OK;
}
extern "C" rt_camera *rt_config_camera(void) { return &cam; }
extern "C" rt_object *rt_config_object(void) { return objs; }   // the global `objs` (kernel.cu:1699); null before onStart()
extern "C" rt_light *rt_config_lights(int *count)
{
    if (count) *count = light_size;
    return lights;
}
extern "C" double rt_last_frame_ms(void) { return fr.last_ms; }

static std::string asset(const std::string &configured, const char *env_name, const char *dflt)
{
    if (!configured.empty()) return configured;
    const char *v = getenv(env_name);
    return (v && *v) ? v : dflt;
}

// onStart, kernel.cu:1704-1714 (+ object::loadMesh :1181-1207, skybox ctor :1120-1123)
void onStart()
{
    aspect = rt_default_aspect();                                  // kernel.cu:1701
    {   // the application shell's environment, read here once (like the asset paths), never by update()
        const char *e = getenv("RT_GPUS");
        env_gpus = (e && *e) ? atoi(e) : 0;
        if (env_gpus < -64 || env_gpus > 64) env_gpus = 0;
    }

    objs = (rt_object *)rt_managed_alloc(sizeof(rt_object));       // `new object()` through memManager
    if (!objs) return;
    memset(objs, 0, sizeof *objs);
    objs->depth = 3;
    objs->sphere_count = cfg_sphere_count;
    // loadMesh (kernel.cu:1181-1207): `mesh1 = new mesh(file)` -- here only when RT_MESH_OBJ names an
    // OBJ file (the reference's skull2.obj is not in its repository); cubes/planes stay empty as shipped
    {
        const std::string obj_path = asset(cfg_mesh_obj, "RT_MESH_OBJ", "");
        if (!obj_path.empty()) {
            objs->mesh1 = rt_mesh_load_obj(obj_path.c_str());
            if (!objs->mesh1) {
                fprintf(stderr, "onStart: %s\n", rt_last_error());
                rt_check(1, "mesh(file)", __FILE__, __LINE__);
            }
        }
    }
    // spheres from the rand() replay
    objs->s1 = (rt_sphere *)calloc((size_t)(cfg_sphere_count > 0 ? cfg_sphere_count : 1), sizeof(rt_sphere));
    rt_generate_spheres(objs->s1, cfg_sphere_count, cfg_seed);
    objs->texture = (rt_sprite *)new sprite(asset(cfg_object_texture, "RT_OBJECT_TEXTURE", "synthetic:object"));
    // sphereAllocMem, kernel.cu:1208-1212 (32 bytes per sphere, :1218-1220)
    objs->d_spheres = (rt_sphere *)rt_managed_alloc(sizeof(float) * 8 * (size_t)(cfg_sphere_count > 0 ? cfg_sphere_count : 1));
    if (objs->d_spheres && cfg_sphere_count > 0)
        memcpy(objs->d_spheres, objs->s1, sizeof(float) * 8 * (size_t)cfg_sphere_count);

    // skybox(img, 10000), kernel.cu:1120-1123, 1700
    Skybox = (rt_skybox *)rt_managed_alloc(sizeof(rt_skybox));
    if (!Skybox) return;
    Skybox->skyboxTex = (rt_sprite *)new sprite(asset(cfg_sky_texture, "RT_SKY_TEXTURE", "synthetic:sky"));
    Skybox->box = (rt_sphere *)rt_managed_alloc(sizeof(rt_sphere));
    if (Skybox->box) rt_sphere_init(Skybox->box, 0, 0, 0, 10000);

    // kernel.cu:1708-1712
    const rt_light m_light = {{20, 20, 20}, 20, 1, 0, 0};
    const rt_light b_light = {{0, 20, -20}, 20, 0, 0, 1};
    const rt_light c_light = {{0, 20, 0}, 20, 0, 1, 0};
    lights[0] = m_light;
    lights[1] = b_light;
    lights[2] = c_light;
    light_size = 3;
}

static void release_frame_buffers()
{
    if (fr.d_pixels) checkHipErrors(hipFree(fr.d_pixels));
    if (fr.h_pixels) checkHipErrors(hipHostFree(fr.h_pixels));
    fr.d_pixels = fr.h_pixels = nullptr;
    fr.width = fr.height = 0;
}

// The frame on several GPUs of the node (rt_multi.hip): the scene graph is mirrored to every
// device when it changes, each device renders its 16-row blocks, one RCCL gather, rows scattered
// home on the first device; then the same D2H + setPixelBuff as the single-GPU path.
static struct MultiRes {
    rt_multi *m = nullptr;
    int gpus = 0;
    const void *tex_key = nullptr, *sky_key = nullptr, *mesh_key = nullptr;
    std::vector<char> spheres_prev, lights_prev, planes_prev, cubes_prev;
    bool planes_set = false, cubes_set = false;
} mr;

static int update_multi(int gpus, int width, int height)
{
    if (!mr.m || mr.gpus != gpus) {
        if (mr.m) rt_multi_destroy(mr.m);
        mr = MultiRes();
        if (gpus > 0) {
            mr.m = rt_multi_create(gpus);
        } else {
            std::vector<int> same((size_t)-gpus, 0);
            if (rt_multi_create_ex(same.data(), -gpus, RT_MULTI_PEER_COPY, &mr.m) != RT_OK) mr.m = nullptr;
        }
        if (!mr.m) return RT_ERR_HIP;
        mr.gpus = gpus;
    }
    int rc = RT_OK;
    const size_t sbytes = sizeof(rt_sphere) * (size_t)(objs->sphere_count > 0 ? objs->sphere_count : 0);
    if (mr.spheres_prev.size() != sbytes || (sbytes && memcmp(mr.spheres_prev.data(), objs->d_spheres, sbytes) != 0)) {
        if ((rc = rt_multi_set_spheres(mr.m, objs->d_spheres, objs->sphere_count)) != RT_OK) return rc;
        mr.spheres_prev.assign((const char *)objs->d_spheres, (const char *)objs->d_spheres + sbytes);
    }
    const size_t lbytes = sizeof(rt_light) * (size_t)light_size;
    if (mr.lights_prev.size() != lbytes || memcmp(mr.lights_prev.data(), lights, lbytes) != 0) {
        if ((rc = rt_multi_set_lights(mr.m, lights, light_size)) != RT_OK) return rc;
        mr.lights_prev.assign((const char *)lights, (const char *)lights + lbytes);
    }
    const rt_sprite *t = objs->texture;
    if (t && t->rBuff->data != mr.tex_key) {
        if ((rc = rt_multi_set_texture(mr.m, t->rBuff->data, t->gBuff->data, t->bBuff->data, t->width, t->height)) != RT_OK) return rc;
        mr.tex_key = t->rBuff->data;
    }
    const rt_sprite *k = Skybox->skyboxTex;
    if (k->rBuff->data != mr.sky_key) {
        if ((rc = rt_multi_set_sky(mr.m, Skybox->box, k->rBuff->data, k->gBuff->data, k->bBuff->data, k->width, k->height)) != RT_OK) return rc;
        mr.sky_key = k->rBuff->data;
    }
    if (objs->mesh1 != mr.mesh_key) {
        if ((rc = rt_multi_set_mesh(mr.m, (objs->mesh1 && objs->mesh1->bvhbox_count > 0) ? objs->mesh1 : nullptr)) != RT_OK) return rc;
        mr.mesh_key = objs->mesh1;
    }
    // planes and cubes by CONTENT, as the single-GPU path passes them every frame (a plane that moves must move)
    const size_t pbytes = sizeof(rt_plane) * (size_t)(objs->plane_count > 0 && objs->d_planes ? objs->plane_count : 0);
    if (!mr.planes_set || mr.planes_prev.size() != pbytes || (pbytes && memcmp(mr.planes_prev.data(), objs->d_planes, pbytes) != 0)) {
        if ((rc = rt_multi_set_planes(mr.m, objs->d_planes, pbytes ? objs->plane_count : 0)) != RT_OK) return rc;
        mr.planes_prev.assign((const char *)objs->d_planes, (const char *)objs->d_planes + pbytes);
        mr.planes_set = true;
    }
    const size_t cbytes = sizeof(rt_cube) * (size_t)(objs->cube_count > 0 && objs->d_cubes ? objs->cube_count : 0);
    if (!mr.cubes_set || mr.cubes_prev.size() != cbytes || (cbytes && memcmp(mr.cubes_prev.data(), objs->d_cubes, cbytes) != 0)) {
        if ((rc = rt_multi_set_cubes(mr.m, objs->d_cubes, cbytes ? objs->cube_count : 0)) != RT_OK) return rc;
        mr.cubes_prev.assign((const char *)objs->d_cubes, (const char *)objs->d_cubes + cbytes);
        mr.cubes_set = true;
    }
    rt_frame_desc fd;
    memset(&fd, 0, sizeof fd);
    fd.struct_size = sizeof fd;
    fd.width = width;
    fd.height = height;
    fd.aspect = aspect;
    fd.cam = cam;
    fd.opts.struct_size = sizeof fd.opts;
    fd.opts.cull = -1;
    // Row bands, as the single-GPU path below: every device renders its blocks of band k+1 while band k is
    // gathered, scattered home and copied to the pinned buffer (the copy stream waits for the band's assembly
    // on the device; the host waits once, at the end).
    const int bands = (height >= 512) ? 3 : 1;
    for (int k = 0; k < bands; ++k) {
        const int y0 = (int)((long long)height * k / bands) & ~15, y1 = (k + 1 == bands) ? height : ((int)((long long)height * (k + 1) / bands) & ~15);
        fd.opts.y0 = y0;
        fd.opts.y1 = y1;
        if ((rc = rt_multi_render(mr.m, &fd, fr.d_pixels)) != RT_OK) return rc;
        if ((rc = rt_multi_stream_wait(mr.m, fr.copy_stream)) != RT_OK) return rc;
        RT_HIP(hipMemcpyAsync(fr.h_pixels + (size_t)y0 * width, fr.d_pixels + (size_t)y0 * width,
                              (size_t)width * (size_t)(y1 - y0) * sizeof(uint32_t), hipMemcpyDeviceToHost, fr.copy_stream));
    }
    RT_HIP(hipStreamSynchronize(fr.copy_stream));
    return RT_OK;
}

// update, kernel.cu:1762-1792
void update()
{
    if (!objs || !Skybox) {
        rt_check(1, "update() before onStart()", __FILE__, __LINE__);
        return;
    }
    // checkKey() (kernel.cu:1716-1759) is keyboard input: out of scope; move the
    // camera through rt_config_camera().

    const int width = getScreenWidth(), height = getScreenHeight();   // kernel.cu:1771
    if (width <= 0 || height <= 0) return;
    cam.aspect = (float)height / width;                               // kernel.cu:1773

    if (!fr.stream) {
        checkHipErrors(hipStreamCreateWithFlags(&fr.stream, hipStreamNonBlocking));
        checkHipErrors(hipStreamCreateWithFlags(&fr.copy_stream, hipStreamNonBlocking));
        checkHipErrors(hipEventCreate(&fr.ev0));
        checkHipErrors(hipEventCreate(&fr.ev1));
        for (hipEvent_t &e : fr.band_done) checkHipErrors(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (width != fr.width || height != fr.height) {   // resize between frames keeps working
        release_frame_buffers();
        const size_t pixelSize = (size_t)width * (size_t)height * sizeof(unsigned int);
        checkHipErrors(hipMalloc((void **)&fr.d_pixels, pixelSize));
        checkHipErrors(hipHostMalloc((void **)&fr.h_pixels, pixelSize, hipHostMallocDefault));
        fr.width = width;
        fr.height = height;
    }

    {   // several GPUs: rt_config_set_gpus(n), else what onStart() found in the application's environment.
        // A frame whose width is no multiple of 4 has no 24-bit rows: it is rendered by the first device alone.
        const int gpus = cfg_gpus != 0 ? cfg_gpus : (env_gpus != 0 ? env_gpus : 1);
        if ((gpus > 1 || gpus < -1) && width % 4 == 0) {
            (void)hipSetDevice(0);
            const int rc = update_multi(gpus, width, height);
            if (rc != RT_OK) {
                fprintf(stderr, "update: %s\n", rt_last_error());
                rt_check(rc, "rt_multi_render", __FILE__, __LINE__);
                return;
            }
            setPixelBuff(fr.h_pixels);                                    // kernel.cu:1788
            return;
        }
    }

    // The frame is rendered in row bands (kernel.cu:1783 is one launch; the bands are the same
    // pixels) so that the copy of a finished band to the pinned buffer -- a DMA engine's work --
    // runs while the next band is being rendered: kernel + copy/bands instead of kernel + copy.
    const int bands = (height >= 512) ? 3 : 1;
    checkHipErrors(hipEventRecord(fr.ev0, fr.stream));
    for (int k = 0; k < bands; ++k) {
        const int y0 = (int)((long long)height * k / bands) & ~15, y1 = (k + 1 == bands) ? height : ((int)((long long)height * (k + 1) / bands) & ~15);
        rt_launch_opts o;
        memset(&o, 0, sizeof o);
        o.struct_size = sizeof o;
        o.cull = -1;
        o.y0 = y0;
        o.y1 = y1;
        uint32_t *band = fr.d_pixels + (size_t)y0 * width;
        const int rc = rt_launch_raytrace_ex(band, width, height, aspect, objs, lights, light_size, cam, Skybox, fr.stream, &o);
        if (rc != RT_OK) {
            fprintf(stderr, "update: %s\n", rt_last_error());
            rt_check(rc, "rt_launch_raytrace", __FILE__, __LINE__);
            return;
        }
        checkHipErrors(hipGetLastError());                            // kernel.cu:1785
        checkHipErrors(hipEventRecord(fr.band_done[k], fr.stream));
        checkHipErrors(hipStreamWaitEvent(fr.copy_stream, fr.band_done[k], 0));
        checkHipErrors(hipMemcpyAsync(fr.h_pixels + (size_t)y0 * width, band, (size_t)width * (y1 - y0) * sizeof(unsigned int),
                                      hipMemcpyDeviceToHost, fr.copy_stream));
    }
    checkHipErrors(hipEventRecord(fr.ev1, fr.stream));
    checkHipErrors(hipStreamSynchronize(fr.copy_stream));             // kernel.cu:1786 (the stream's kernels precede its copies)
    checkHipErrors(hipStreamSynchronize(fr.stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, fr.ev0, fr.ev1) == hipSuccess) fr.last_ms = ms;

    setPixelBuff(fr.h_pixels);                                        // kernel.cu:1788
}

extern "C" void rt_on_start(void) { onStart(); }
extern "C" void rt_update(void) { update(); }
