/*
 * include/rt_engine.h -- C ABI of the MI355X-native ray-tracing hot path.
 *
 * Drop-in boundary for ONE path of leonZtiger/Ray-Tracer-engine: the per-pixel
 * ray-generation -> sphere-scene intersection -> shading -> framebuffer-write
 * kernel (`rayTrace`, /root/reference/kernel.cu:1614-1690) together with the
 * host surface around it (kernel.cuh:3-4, memManager.h:11-18, window.h:7-16,
 * sprite.h:11-47). Plain C types only: no HIP, torch or C++ types appear in any
 * signature, so the library can be bound from C, C++, ctypes, cgo, JNI, ...
 *
 * Every entry point names the reference interface it replaces (file:line).
 * Entry points return 0 on success and a non-zero rt_status on failure, unless
 * they mirror a reference function that is void (those keep the reference's
 * fatal convention: print, reset the device, exit(99) -- memManager.cpp:3-11).
 *
 * Threading: like the reference (single Win32 UI thread) the entry points are
 * not thread-safe; use one caller thread per process. A process drives one GPU
 * through rt_scene_* / rt_launch_raytrace / update(), or several GPUs of the node
 * through rt_multi_* (one process, one scene per device; update() after
 * rt_config_set_gpus(n)); hosts that run one process per GPU (bench.py under
 * torch.distributed) render their rows with rt_scene_render and run the gather
 * themselves (rt_assemble_rows24 is its root side). See INTEGRATION.md.
 */
#ifndef RT_ENGINE_H
#define RT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1
#define RT_MAX_LIGHTS 8      /* reference uses light_size = 3 (kernel.cu:1692) */
#define RT_MAX_SPP 16
#define RT_MAX_PLANES 64    /* planes and cubes are tested exhaustively (no culling) */
#define RT_MAX_CUBES 256

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument (null pointer, size <= 0, ...)       */
    RT_ERR_UNSUPPORTED = 2,  /* scene uses a primitive outside the sphere path    */
    RT_ERR_HIP = 3,          /* a HIP runtime call failed (see rt_last_error())   */
    RT_ERR_NO_DEVICE = 4,    /* no gfx950 device / kernel image not loadable      */
    RT_ERR_CAPACITY = 5      /* sphere_count or light_size above the build limit  */
} rt_status;

/* ------------------------------------------------------------------ *
 * POD mirrors of the reference's kernel-argument types. Layouts are   *
 * byte-compatible with the MSVC x64 / nvcc layouts the reference      *
 * hard-codes (kernel.cu:1214,1219: 40 and 32 bytes).                  *
 * ------------------------------------------------------------------ */
typedef struct rt_vec3 { float x, y, z; } rt_vec3;          /* vec3d, kernel.cu:38-40   */
typedef struct rt_ray { rt_vec3 Org, Dir; } rt_ray;          /* ray,   kernel.cu:225-236 */

typedef struct rt_camera {                                   /* camera, kernel.cu:237-262 */
    rt_vec3 Org, Dir;
    float aspect;            /* set per frame (kernel.cu:1773), unused by the kernel */
    float Camyaw, Campitch;  /* degrees; defaults 180 / -20 (kernel.cu:261)          */
} rt_camera;                                                 /* 36 bytes */

typedef struct rt_light {                                    /* light, kernel.cu:1246-1261 */
    rt_vec3 pos;
    float size, r, g, b;
} rt_light;                                                  /* 28 bytes */

typedef struct rt_sphere {                                   /* sphere : shape, kernel.cu:265-358 */
    void *vptr_slot;         /* shape has a virtual method; ignored by this library   */
    rt_vec3 orgin;           /* (sic) centre                                          */
    uint8_t reflective;      /* unused by the kernel                                  */
    uint8_t pad_[3];
    float radius;            /* holds r*r (ctor, kernel.cu:287); intersect squares it AGAIN (:334) */
    uint32_t tail_pad_;
} rt_sphere;                                                 /* 32 bytes */

typedef struct rt_plane {                                    /* plane : shape, kernel.cu:360-384 */
    void *vptr_slot;
    rt_vec3 orgin;           /* a point on the plane                                  */
    uint8_t reflective;
    uint8_t pad_[3];
    rt_vec3 normal;          /* used as given (not normalised by the kernel)          */
    uint32_t tail_pad_;
} rt_plane;                                                  /* 40 bytes (kernel.cu:1214) */

typedef struct rt_cube {                                     /* cube : shape, kernel.cu:387-509 */
    void *vptr_slot;
    rt_vec3 orgin;           /* (c1 + c2) / 2 (ctor, kernel.cu:395)                   */
    rt_vec3 normals[3];      /* unused by the kernel                                  */
    rt_vec3 bounds[2];       /* the two corners of the slab test (kernel.cu:457-485)  */
} rt_cube;                                                   /* 80 bytes */

typedef struct rt_vec2 { float u, v; } rt_vec2;              /* vec2d, kernel.cu:32-34 */

typedef struct rt_triangle {                                 /* triangle, kernel.cu:206-212 */
    rt_vec3 points[3];
    rt_vec3 normal;          /* face normal                                           */
    rt_vec3 vecNormal[3];    /* vertex normals (used when mesh.has_normals)           */
    rt_vec2 vt[3];           /* texture coordinates                                   */
} rt_triangle;                                               /* 108 bytes (kernel.cu:1018-1020) */

typedef struct rt_bvhbox {                                   /* Bvhbox, kernel.cu:512-543 */
    rt_cube *bvhbox;         /* bounds of this leaf (read by the shadow path, :1479)  */
    rt_cube *d_bvhbox;       /* same cube (read by castRay, :1297)                    */
    int *indexes;
    int *d_indexes;          /* triangle indices of this leaf                         */
    int length;
} rt_bvhbox;

typedef struct rt_mesh {                                     /* mesh, kernel.cu:559-575 (field order kept) */
    rt_triangle *d_tri_arr;
    rt_triangle *h_tri_arr;
    int poly_count;
    int bvhbox_count;        /* flat list of leaves after 10 median splits            */
    int bvhLayer_count;      /* 10                                                    */
    uint8_t has_normals;
    rt_bvhbox *h_box;
    rt_bvhbox *d_box;
    int *indexes;
} rt_mesh;

typedef struct rt_buffer {                                   /* buffer, sprite.h:11-19 */
    float *data;             /* planar floats in [0,1]                                */
    int size;                /* bytes (Sprite.cpp:14)                                 */
} rt_buffer;

typedef struct rt_sprite {                                   /* sprite, sprite.h:25-47 */
    rt_buffer *rBuff, *gBuff, *bBuff;
    int width, height;
} rt_sprite;

typedef struct rt_skybox {                                   /* skybox, kernel.cu:1116-1173 */
    rt_sphere *box;          /* centre (0,0,0), ctor radius 10000 (kernel.cu:1122,1700) */
    rt_sprite *skyboxTex;
} rt_skybox;

typedef struct rt_object {                                   /* object, kernel.cu:1176-1244 (field order kept) */
    int sphere_count, plane_count, cube_count;               /* :1231 */
    int depth;                                               /* :1232 */
    rt_sphere *s1;           /* host staging copy                                     */
    rt_sphere *d_spheres;    /* what the kernel reads (:1333)                         */
    rt_cube *c1, *d_cubes;   /* cubes read by the kernel at kernel.cu:1344-1356, 1526-1536 */
    rt_plane *planes, *d_planes; /* planes, kernel.cu:1359-1372, 1513-1523            */
    rt_mesh *mesh1;          /* triangle mesh + flat BVH (kernel.cu:1293-1328, 1475-1497);
                                NULL = no mesh (the reference's bvhbox_count = 0)       */
    rt_sprite *texture;      /* :1240, read at :1643-1655                             */
    void *mat;               /* unused                                                */
    void **tot_mesh;
    int meshes;
} rt_object;

/* ------------------------------------------------------------------ *
 * memManager (memManager.h:11-18, memManager.cpp:3-22)                *
 * ------------------------------------------------------------------ */
/* check_cuda(result, func, file, line): on non-zero `err` print
 * "HIP error = <n> at <file>:<line> '<expr>'", reset the device, exit(99). */
void rt_check(int err, const char *expr, const char *file, int line);
/* memManager::operator new : managed allocation + device synchronise. */
void *rt_managed_alloc(size_t len);
/* memManager::operator delete : device synchronise + free. */
void rt_managed_free(void *ptr);

/* ------------------------------------------------------------------ *
 * Kernel launch (rayTrace<<<blocks,threads>>>, kernel.cu:1615,1780-1783) *
 * ------------------------------------------------------------------ */
typedef struct rt_launch_opts {
    uint32_t struct_size;    /* = sizeof(rt_launch_opts); for ABI growth              */
    float *rgba;             /* optional device float4 buffer, width*(y1-y0) texels:
                                linear colour BEFORE the *254 pack        */
    int y0, y1;              /* row band rendered by this call; 0,0 = whole frame.
                                `pixels`/`rgba` point at row y0 (band-local buffers)  */
    int spp;                 /* samples per pixel taken by this call (1..RT_MAX_SPP);
                                0 = 1. Sample k uses the fixed stratified offset
                                table (rt_sample_offset); 1 spp = pixel centre, as
                                the reference (kernel.cu:1624-1625)                   */
    int sample_base;         /* index of the first sample of this call               */
    int sample_total;        /* total samples of the frame (divisor at resolve);
                                0 = spp                                               */
    int accumulate;          /* 1: add into rgba (progressive); 0: overwrite          */
    int resolve;             /* 0 default: write packed words when pixels != NULL;
                                -1: leave `pixels` untouched (intermediate progressive pass) */
    int cull;                /* -1 default (on); 0 = brute force over the whole list,
                                exactly the reference's loops; 1 = conservative tile
                                culling (same output, fewer tests)                    */
    int tile;                /* 0 default; else tile width in {8,16,32,64} (64 px/wave) */
    uint64_t *stats;         /* optional device array of RT_STATS_COUNT counters      */
    int force_slow_path;     /* testing: disable every exactness-preserving shortcut  */
    int profile;             /* diagnostics: with `stats`, fill the per-phase cycle
                                counters (RT_STAT_PHASE0..) instead of work counters;
                                tuning builds of the library only (-DRT_TUNING)        */
    int interleave_count;    /* multi-GPU load balance: when > 1 this call renders the
                                row blocks k = interleave_index, +count, +2*count, ... of
                                `interleave_rows` rows each, counted from the first row
                                of the band [y0, y1) (y0 a multiple of `rows`; 0/0 = the
                                whole frame). Output buffers are compact: local row L holds
                                global row y0 + ((L / rows) * count + index) * rows + L % rows */
    int interleave_index;
    int interleave_rows;     /* block height, a power of two >= 16; 0 = 16             */
    void *packed24;          /* optional device buffer, 3 bytes per pixel (B,G,R: the packed
                                word without its zero top byte), width*rows*3 bytes, band-local
                                like `pixels`; needs width % 4 == 0. What a multi-GPU rank
                                sends to the root: a quarter less than the 32-bit words      */
    int table_lds;           /* 1: each 256-thread workgroup stages the WHOLE sphere table in LDS
                                (north_star's first design) instead of reading it from global
                                memory / L2 and keeping only the tiles' survivor lists in LDS.
                                Same pixels; measured slower, so opt-in.
                                Default tile only; ignored when the table does not fit          */
    int fast;                /* 1: opt-in APPROXIMATE mode. Everything that enters a pixel continuously
                                (primary hit, normal, toL) stays exact; the ten shadow-sample
                                directions of a light are built once per light in binary32 with
                                hardware rsq / sqrt and FMAs (instead of following the reference's
                                in-place re-normalisations in binary64 trigonometry), the shadow
                                tests use FMAs and approximate roots, the texel index comes from the
                                binary32 (tx, ty) without the certainty test. NOT bit-exact: a pixel
                                is either identical or one where a discrete decision flipped (a
                                shadow sample = 0.1 of a light's brightness, a neighbouring texel):
                                about 2 pixels in 10^5 at C3. Never the
                                default; culling kernels with the default tile and no mesh only --
                                otherwise ignored (the launch is exact)                            */
} rt_launch_opts;

enum { RT_STAT_PRIMARY_TESTS = 0, /* sphere tests issued for primary rays (per lane) */
       RT_STAT_SHADOW_TESTS = 1,  /* sphere tests issued for shadow rays (per lane)  */
       RT_STAT_CULL_TESTS = 2,    /* sphere-vs-beam tests (per lane)                 */
       RT_STAT_HIT_PIXELS = 3,
       RT_STAT_UNSHADOWED = 4,
       RT_STAT_WAVE_TEST_SLOTS = 5, /* 64 x wave-level test iterations (issue slots) */
       RT_STAT_LIST_ENTRIES = 6,  /* sum of survivor-list lengths                     */
       RT_STAT_LIST_OVERFLOWS = 7,
       RT_STAT_PHASE0 = 8,        /* 8 per-phase cycle sums (profile builds only):
                                     ray setup, primary cull, primary tests, shading+sky,
                                     beam bound, shadow cull, sample construction, shadow tests */
       RT_STAT_CLUSTERS = 16,     /* sum over waves of distinct hit spheres per tile  */
       RT_STATS_COUNT = 24 };

/* Same argument order and meaning as the reference kernel; references become
 * pointers; `stream` is a hipStream_t (NULL = default stream). `pixels` is a
 * device-accessible buffer of width*height packed 0x00RRGGBB words. objs, lights
 * and sky are read on the HOST at call time (they live in managed/host memory
 * in the reference) and mirrored into device-resident tables; texture planes
 * are uploaded once per (pointer, size) and cached -- call
 * rt_invalidate_textures() after modifying texel data in place. */
int rt_launch_raytrace(uint32_t *pixels, int width, int height, float aspect,
                       const rt_object *objs, const rt_light *lights, int light_size,
                       rt_camera cam, const rt_skybox *sky, void *stream);
int rt_launch_raytrace_ex(uint32_t *pixels, int width, int height, float aspect,
                          const rt_object *objs, const rt_light *lights, int light_size,
                          rt_camera cam, const rt_skybox *sky, void *stream,
                          const rt_launch_opts *opts);
void rt_invalidate_textures(void);

/* ------------------------------------------------------------------ *
 * Frame driver (kernel.cuh:3-4; kernel.cu:1692-1714, 1762-1792)       *
 * The C++ symbols `void onStart(); void update();` are exported too.  *
 * ------------------------------------------------------------------ */
void rt_on_start(void);      /* builds the default scene (spheres from the MSVC rand()
                                replay, 3 lights, synthetic textures)                 */
void rt_update(void);        /* one frame: size query -> launch -> sync -> setPixelBuff */
/* Scene knobs the reference keeps as compile-time globals (kernel.cu:1231,1695-1702). */
int rt_config_set_sphere_count(int n);       /* before rt_on_start(); default 1024    */
int rt_config_set_seed(unsigned int seed);   /* MSVC rand() seed; default 1           */
/* Asset files of onStart() (kernel.cu:1700,1706 and loadMesh :1181 hard-code C:\ paths): binary
 * PPM (P6) textures and an OBJ mesh. NULL / "" = not set: onStart() then looks at the
 * application's environment (RT_OBJECT_TEXTURE, RT_SKY_TEXTURE, RT_MESH_OBJ), and without
 * those uses the synthetic textures and no mesh. Call before rt_on_start().                  */
int rt_config_set_assets(const char *object_texture, const char *sky_texture, const char *mesh_obj);
rt_camera *rt_config_camera(void);           /* the global `cam` (kernel.cu:1695)     */
rt_object *rt_config_object(void);           /* the global `objs` (kernel.cu:1699); NULL before onStart(). update()
                                                reads its sphere / plane / cube / mesh fields every frame */
rt_light *rt_config_lights(int *count);      /* the global `lights` (kernel.cu:1694)  */
float rt_default_aspect(void);               /* (float)tan(90*0.5*3.1415/180), :1701  */
double rt_last_frame_ms(void);               /* device time of the last rt_update()   */

/* ------------------------------------------------------------------ *
 * Offscreen stand-in for window.cpp (window.h:7-16). The C++ symbols   *
 * getScreenWidth/getScreenHeight/setPixelBuff/... are exported (weak)  *
 * so an application's own window.cpp overrides them.                   *
 * ------------------------------------------------------------------ */
int rt_offscreen_resize(int width, int height);   /* WM_SIZE equivalent (window.cpp:29-46),
                                                      takes the RENDER size directly  */
const uint32_t *rt_offscreen_pixels(void);        /* render.buffmemory                */
int rt_offscreen_width(void);
int rt_offscreen_height(void);
int rt_offscreen_write_ppm(const char *path);     /* dump the presented frame         */

/* ------------------------------------------------------------------ *
 * Scene construction helpers (host side, no GPU needed)               *
 * ------------------------------------------------------------------ */
/* plane(pos, normal) (kernel.cu:364-367) and cube(c1, c2) (kernel.cu:391-396). */
void rt_plane_init(rt_plane *p, float px, float py, float pz, float nx, float ny, float nz);
void rt_cube_init(rt_cube *c, float ax, float ay, float az, float bx, float by, float bz);
/* sphere::sphere(org, r) (kernel.cu:285-288): stores radius = r*r.     */
void rt_sphere_init(rt_sphere *s, float x, float y, float z, float r);
/* The scene of object::loadMesh (kernel.cu:1189-1192) with the MSVC rand()
 * LCG replayed from `seed`; draw order x, y, z, r.      */
int rt_generate_spheres(rt_sphere *out, int n, unsigned int seed);
int rt_msvc_rand_sequence(unsigned int seed, int *out, int n);
/* Deterministic synthetic textures standing in for wood.jpg / sky_box.jpg
 * (kernel.cu:1700,1706; the assets are not in the reference repo). Planes are
 * k/255 with integer k, like the 8-bit decode of Sprite.cpp:35-51.
 * kind 0 = object texture (512x512), kind 1 = sky (2048x1024).           */
int rt_synth_texture_size(int kind, int *width, int *height);
int rt_synth_texture(int kind, float *r, float *g, float *b);
/* mesh(filename) (kernel.cu:575-747) + createBvhMesh (:752-937): OBJ text (v / vt / vn /
 * f with a, a//c or a/b/c tokens, triangles and quads) -> triangles -> flat list of
 * leaf boxes. Host memory only (d_* alias h_*); no GPU needed. Blank lines, which
 * corrupt the reference's parser state, are skipped. NULL on failure.            */
rt_mesh *rt_mesh_from_obj_text(const char *text);
rt_mesh *rt_mesh_load_obj(const char *path);
void rt_mesh_free(rt_mesh *m);
/* sprite(file) without OpenCV: binary PPM (P6) -> planar float planes.    */
int rt_load_ppm(const char *path, float **r, float **g, float **b, int *width, int *height);
void rt_free_planes(float *r, float *g, float *b);
/* Sub-pixel position of sample k of n (n=1 -> 0.5,0.5: the reference).    */
int rt_sample_offset(int k, int n, double *ox, double *oy);

/* ------------------------------------------------------------------ *
 * Device-resident frame pipeline (what update() uses internally;      *
 * exposed so a host can render into its own device buffers, capture   *
 * the frame into a hipGraph, or render a row band for multi-GPU).     *
 * ------------------------------------------------------------------ */
typedef struct rt_scene rt_scene;    /* opaque, device-resident copy of one scene */

rt_scene *rt_scene_create(void);
void rt_scene_destroy(rt_scene *s);
int rt_scene_set_spheres(rt_scene *s, const rt_sphere *host_spheres, int n);
int rt_scene_set_planes(rt_scene *s, const rt_plane *host_planes, int n);   /* SURVEY.md 8(f) row 2 */
int rt_scene_set_cubes(rt_scene *s, const rt_cube *host_cubes, int n);
int rt_scene_set_mesh(rt_scene *s, const rt_mesh *mesh);    /* NULL removes it; SURVEY.md 8(f) row 4 */
int rt_scene_set_texture(rt_scene *s, const float *r, const float *g, const float *b, int w, int h);
int rt_scene_set_sky(rt_scene *s, const rt_sphere *box, const float *r, const float *g,
                     const float *b, int w, int h);
int rt_scene_set_lights(rt_scene *s, const rt_light *lights, int n);

typedef struct rt_frame_desc {
    uint32_t struct_size;
    int width, height;
    float aspect;
    rt_camera cam;
    uint32_t *pixels;        /* device, band-local (row y0 first); may be NULL        */
    rt_launch_opts opts;     /* rgba / band / spp / cull / stats                       */
} rt_frame_desc;

int rt_scene_render(rt_scene *s, const rt_frame_desc *fd, void *stream);

/* Order in which a launch starts its tiles. 1 (default): in blocks of 16 x 16 tiles, the block with the longest
 * tile first -- the frame kernel records every tile's wave duration, and from the previous launch's durations the
 * blocks are sorted on the device (three small kernels, ~15 us): after 1, 2, 4, 8, 16, 32, 64, 96, ... launches of an
 * unchanged view (camera, sphere list) and layout (frame size, rows, tile shape), every third launch while the view
 * keeps changing. A launch then ends with its cheap tiles instead of draining the SIMDs behind a few expensive ones
 * (C3: 0.38 -> 0.35 ms per frame, an eighth of the frame 0.085 -> 0.071 ms; a moving camera 0.398 -> 0.381 ms).
 * 0: grid order. Scheduling only: the pixels are the same bits either way. A frame graph sorts its own order at the
 * head of every replay (from the durations of the previous one); table_lds launches always run in grid order.     */
int rt_scene_set_tile_order(rt_scene *s, int mode);

/* hipGraph-captured frame loop (config C4): `passes` samples per pixel + resolve + optional async copy of the packed
 * frame to pinned host memory, recorded once and replayed per frame. passes > 0: the samples are taken by ONE kernel
 * node (the sample loop runs inside the kernel); passes < 0: |passes| progressive one-sample nodes, each adding into
 * opts.rgba (needs opts.rgba) -- the same bits in the end, 9 % slower at C4, for callers that present between passes. */
typedef struct rt_frame_graph rt_frame_graph;
rt_frame_graph *rt_graph_capture(rt_scene *s, const rt_frame_desc *fd, int passes,
                                 uint32_t *host_pixels /* pinned, may be NULL */, void *stream);
/* Replays the frame. If the scene's tables were rewritten since the graph was built (another
 * sphere list, lights, textures, or a direct render at another resolution) the graph is
 * rebuilt first -- it never replays against tables it was not built for.                     */
int rt_graph_launch(rt_frame_graph *g, void *stream);
/* A camera move (kernel.cu:1716-1759 moves `cam` every frame): the kernel nodes' by-value
 * uniforms are replaced with hipGraphExecKernelNodeSetParams and the eye-cone table is rebuilt
 * by a kernel node of the graph itself -- no re-capture, no synchronisation.                  */
int rt_graph_set_camera(rt_frame_graph *g, const rt_camera *cam);
void rt_graph_destroy(rt_frame_graph *g);

/* ------------------------------------------------------------------ *
 * Several GPUs of one node, one process (BASELINE C5): *
 * the frame's rows are dealt to the devices in 16-row blocks           *
 * round-robin, every device renders its rows as 3 bytes per pixel,     *
 * ONE RCCL gather over xGMI brings them to the first device, and one   *
 * small kernel there scatters the rows home and widens them to the     *
 * 0x00RRGGBB words setPixelBuff() consumes (kernel.cu:1788). update()  *
 * takes this path when rt_config_set_gpus(n > 1) was called (or the    *
 * application's environment said RT_GPUS=n when onStart() ran).        *
 * ------------------------------------------------------------------ */
typedef struct rt_multi rt_multi;
enum { RT_MULTI_AUTO = 0,       /* RCCL for distinct devices IF its gather passes a self-test at creation
                                   (every rank's 256 known bytes arrive in its slot of the root's buffer),
                                   else peer copies; rt_multi_note() says which and why              */
       RT_MULTI_RCCL = 1,       /* ncclCommInitAll + one ncclGather per frame (librccl loaded with dlopen);
                                   creation fails if the self-test does. With ONE device the whole exchange
                                   still runs (24-bit rows, one-rank in-place gather, scatter kernel)  */
       RT_MULTI_PEER_COPY = 2   /* the first device pulls the rows with hipMemcpyPeerAsync (SDMA over xGMI);
                                   accepts the same device several times (one-GPU rehearsal of the path) */ };
rt_multi *rt_multi_create(int n_gpus);                       /* devices 0 .. n_gpus-1, RT_MULTI_AUTO; NULL on failure */
int rt_multi_create_ex(const int *devices, int n, int transport, rt_multi **out);
void rt_multi_destroy(rt_multi *m);
int rt_multi_device_count(const rt_multi *m);
int rt_multi_transport(const rt_multi *m);
rt_scene *rt_multi_scene(rt_multi *m, int i, int *device);   /* the i-th device's scene (hipSetDevice(*device) before using it) */
/* the same scene on every device (rt_scene_set_* per device) */
int rt_multi_set_spheres(rt_multi *m, const rt_sphere *host_spheres, int n);
int rt_multi_set_planes(rt_multi *m, const rt_plane *host_planes, int n);
int rt_multi_set_cubes(rt_multi *m, const rt_cube *host_cubes, int n);
int rt_multi_set_mesh(rt_multi *m, const rt_mesh *mesh);
int rt_multi_set_texture(rt_multi *m, const float *r, const float *g, const float *b, int w, int h);
int rt_multi_set_sky(rt_multi *m, const rt_sphere *box, const float *r, const float *g, const float *b, int w, int h);
int rt_multi_set_lights(rt_multi *m, const rt_light *lights, int n);
/* One frame, or one row band of it (width, height, aspect, cam and opts.spp / cull / tile of `fd`;
 * opts.y0 / y1 select a band of whole 16-row blocks, dealt to the devices from the band's first row;
 * its output pointers and interleave fields are ignored). The assembled rows land at their place in
 * `pixels_dev0` (the WHOLE frame's buffer in memory of the first device) or, if that is NULL, in an
 * internal buffer (rt_multi_frame). Asynchronous; two frames (or bands) may be in flight. */
int rt_multi_render(rt_multi *m, const rt_frame_desc *fd, uint32_t *pixels_dev0);
int rt_multi_sync(rt_multi *m);                              /* wait for every frame enqueued so far */
int rt_multi_stream_wait(rt_multi *m, void *stream);         /* `stream` (first device) waits ON THE DEVICE for the frame /
                                                                band enqueued last: copies of it need no host wait */
const char *rt_multi_note(const rt_multi *m);                /* transport chosen at creation, and why */
unsigned long long rt_multi_gathers(const rt_multi *m);      /* ncclGather groups issued so far (RCCL transport) */
const uint32_t *rt_multi_frame(const rt_multi *m);           /* device pointer of the last assembled frame */
int rt_multi_download(rt_multi *m, uint32_t *host);          /* sync + copy the last frame to host memory */
/* The root side of the exchange alone, for hosts that run the gather themselves (one process per GPU:
 * bench.py under torch.distributed): `recv` = n slots of slot_rows rows of width*3 bytes, slot r being
 * what rank r rendered with interleave (n, r, 16) into opts.packed24; `frame` = width*height words. */
int rt_assemble_rows24(const void *recv, uint32_t *frame, int width, int height, int n, int slot_rows, void *stream);
int rt_config_set_gpus(int n);                               /* update(): devices used per frame; default 1. n < 0:
                                                                rehearsal of the path on one GPU -- |n| shares of the
                                                                frame, all on device 0, RT_MULTI_PEER_COPY */

/* ------------------------------------------------------------------ *
 * Introspection / diagnostics                                         *
 * ------------------------------------------------------------------ */
int rt_abi_version(void);
const char *rt_last_error(void);
int rt_device_count(void);
int rt_set_soft_errors(int on);  /* 1: rt_check() records + returns instead of exit(99) */
/* Device evaluation of the library's scalar building blocks, for bit-for-bit
 * comparison with the CPU oracle (tests only; all pointers are HOST arrays).
 * op: 0 cosf, 1 sinf, 2 acosf, 3 atan2f(a,b)                                    */
int rt_debug_math(int op, const float *a, const float *b, float *out, int n);
/* sphere::intersect on the device: hit[i], t[i] for rays[i] vs spheres[i].      */
int rt_debug_intersect(const rt_sphere *spheres, const rt_ray *rays, int n, int *hit, float *t);
/* The 10 shadow-sample directions of castLightRay (kernel.cu:1442-1468) and its
 * brightness result for (start, normal, light) against `spheres`.               */
int rt_debug_light(const rt_sphere *spheres, int n_spheres, const rt_vec3 *start,
                   const rt_vec3 *normal, const rt_light *light, int n,
                   float *dirs /* n*30 */, float *brightness /* n */);
/* The exact sample directions (n*30) next to the approximate ones of the frame kernel's sample pre-pass (n*30) and the
 * pre-pass's guard flags (n*10: 1 = the approximate direction may be used), for its error bound (tests only).   */
int rt_debug_light_prepass(const rt_vec3 *start, const rt_light *light, int n, float *dirs, float *approx_dirs, int *approx_ok);
/* The culling kernels' shortcuts against the long forms they stand for, evaluated on the
 * device for n pseudo-random inputs derived from `seed` (tests only):
 *  what 0: lean normalise vs the IEEE one on vectors of every scale -> out[0] = differing results
 *  what 1: lean sqrt vs IEEE sqrtf on EVERY float in [2^-96, 2^40] (n, seed ignored) -> out[0] = differing results
 *  what 2: approximate (tx, ty) of a unit normal vs the exact binary64 expressions of
 *          kernel.cu:1402-1403 -> out[0], out[1] = max |error| of tx, ty (as float bits in the
 *          low word), out[2] = lanes the 512x512 certainty test accepted, out[3] = accepted
 *          lanes whose texel index differs from the exact one (must be 0)                   */
int rt_debug_shortcuts(int what, unsigned seed, long long n, unsigned long long out[4]);
/* The per-sphere occluder lists the culling kernels use for one light (host computation, no GPU): for every sphere i
 * of the list the number of spheres a shadow ray leaving i's surface towards `light` can hit at all (counts[i]; -1: no
 * list), the beam slope the list holds for (kcaps[i]) and the list positions of its first `cap` members
 * (members[i*cap ..]; likeliest occluder first). Tests only.                                                      */
int rt_debug_occluder_lists(const rt_sphere *spheres, int n, const rt_light *light, int *counts, float *kcaps, int *members, int cap);
/* ... plus every list's offset into the light's entry array (offsets[i], or NULL) and the number of entries that array is
 * allocated with (the kernel reads whole steps of 64 entries from a list's offset on)                                     */
/* The beam slope of every group of pixels on a sphere (what the frame kernel takes instead of bounding the sample spread
 * per tile; -1: none), host builder and device builder (either pointer may be NULL; the device one needs a GPU), the
 * same for one ball, and the spread s = sigma / (|l.pos| - frob) at ONE start with the 3x3 matrix it comes from */
int rt_debug_sphere_beam_slopes(const rt_sphere *spheres, int n, const rt_light *light, float *host_kbeam, float *device_kbeam);
double rt_debug_sphere_beam_slope(const double lpos[3], const double centre[3], double r0);
double rt_debug_beam_sine(const double lpos[3], const double start[3], double *sigma, double *frob, double m9[9]);
/* ... and as the DEVICE builds them (what a scene uses: one wave per sphere, members in list order); needs a GPU */
int rt_debug_occluder_lists_device(const rt_sphere *spheres, int n, const rt_light *light, int *counts, float *kcaps, int *members, int cap);
int rt_debug_occluder_lists_ex(const rt_sphere *spheres, int n, const rt_light *light, int *counts, float *kcaps, int *members, int cap,
                               int *offsets, int *entries_allocated);

#ifdef __cplusplus
}
#endif
#endif /* RT_ENGINE_H */
