/*
 * oracle/rt_oracle.h -- TEST INFRASTRUCTURE. NOT part of the product.
 *
 * CPU restatement of the reference's per-pixel hot path
 * (/root/reference/kernel.cu:1614-1690 `rayTrace` and its callees, sphere path
 * only: SURVEY.md section 8(a) rows a1-a12). Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (ray-tracer-engine_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden
 * vectors or fixtures, and cannot be built here (CUDA + Win32 + OpenCV). This
 * oracle is pinned only by (1) the hand-derived known-answer tests of
 * SURVEY.md section 4 (tests/test_oracle_kat.py) and (2) a line-by-line
 * reading of the cited source ranges.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* POD mirrors of the reference types (sizes pinned in tests: SURVEY section 4). */
typedef struct { float x, y, z; } o_vec3d;                 /* kernel.cu:38-40   */
typedef struct { o_vec3d Org, Dir; } o_ray;                 /* kernel.cu:225-236 */
typedef struct {                                            /* kernel.cu:237-262 */
    o_vec3d Org, Dir;
    float aspect;
    float Camyaw, Campitch;
} o_camera;                                                 /* 36 bytes */
typedef struct { o_vec3d pos; float size, r, g, b; } o_light; /* kernel.cu:1246-1261, 28 B */
typedef struct {                                            /* kernel.cu:265-358 */
    void *vptr;          /* shape has a virtual function -> vtable pointer */
    o_vec3d orgin;       /* @8  (sic) */
    unsigned char reflective; /* @20 */
    float radius;        /* @24: ALREADY r*r (ctor, kernel.cu:287) */
} o_sphere;                                                 /* 32 bytes */
typedef struct {                                            /* plane : shape, kernel.cu:360-384 */
    void *vptr;
    o_vec3d orgin;            /* @8  */
    unsigned char reflective; /* @20 */
    o_vec3d normal;           /* @24 */
} o_plane;                                                  /* 40 bytes (kernel.cu:1214: sizeof(float)*10) */
typedef struct {                                            /* cube : shape, kernel.cu:387-509 */
    void *vptr;
    o_vec3d orgin;            /* @8: (c1+c2)/2 */
    o_vec3d normals[3];       /* @20 */
    o_vec3d bounds[2];        /* @56 */
} o_cube;                                                   /* 80 bytes */
typedef struct {                                            /* triangle, kernel.cu:206-212 */
    o_vec3d points[3];
    o_vec3d normal;           /* face normal, normalise(cross(p1-p0, p2-p0)) */
    o_vec3d vecNormal[3];
    float vt[3][2];           /* vec2d {u, v} */
} o_triangle;                                               /* 108 bytes (mesh::getBytes, kernel.cu:1018-1020) */
typedef struct o_mesh o_mesh;                               /* mesh, kernel.cu:559-1113 */
typedef struct {                                            /* sprite.h:25-47 */
    const float *r, *g, *b; /* rBuff/gBuff/bBuff ->data : planar floats in [0,1] */
    int width, height;
} o_sprite;

typedef struct {
    int width, height;        /* full frame size used by ray generation      */
    float aspect;             /* kernel.cu:1701                               */
    const o_sphere *spheres;  /* objs.d_spheres                               */
    int sphere_count;         /* objs.sphere_count                            */
    const o_sprite *texture;  /* objs.texture                                 */
    const o_light *lights;    /* kernel.cu:1708-1712                          */
    int light_size;           /* kernel.cu:1692                               */
    o_camera cam;             /* by value, kernel.cu:1615                     */
    const o_sphere *sky_box;  /* sky.box  (kernel.cu:1122)                    */
    const o_sprite *sky_tex;  /* sky.skyboxTex                                */
    int y0, y1;               /* rows [y0,y1) to render (full frame: 0,height) */
    double off_x, off_y;      /* sub-pixel sample position; reference = 0.5,0.5
                                 (kernel.cu:1624-1625). Build-defined extension
                                 for the 4-spp config. */
    const o_cube *cubes;      /* objs.d_cubes  (kernel.cu:1344-1356); SURVEY 8(f) row 2 */
    int cube_count;
    const o_plane *planes;    /* objs.d_planes (kernel.cu:1359-1372) */
    int plane_count;
    const o_mesh *mesh;       /* objs.mesh1 (kernel.cu:1293-1328, 1475-1497); NULL = bvhbox_count 0 */
} o_frame;

/* counters[0]=primary sphere tests, [1]=shadow sphere tests,
 * [2]=hit pixels, [3]=unshadowed shadow samples. May be NULL. */
int oracle_render(const o_frame *f, float *rgba, uint32_t *packed,
                  uint64_t counters[4], int nthreads);

/* Unit entry points for the known-answer tests. */
int oracle_sphere_intersect(const o_sphere *s, const o_ray *r, float *t); /* kernel.cu:293-354 */
uint32_t oracle_rgb_to_int(int r, int g, int b);                         /* kernel.cu:547-556 */
int oracle_f2i(float v);                 /* float->int as CUDA cvt.rzi (NaN->0, saturating) */
void oracle_make_sphere(o_sphere *s, float x, float y, float z, float r); /* ctor kernel.cu:285-288 */
float oracle_default_aspect(void);                                       /* kernel.cu:1701 */
void oracle_primary_ray(int x, int y, int width, int height, float aspect,
                        const o_camera *cam, double off_x, double off_y, o_ray *out); /* kernel.cu:1624-1631 */
void oracle_rotate_dir(const o_camera *cam, const o_vec3d *v, float yaw, float pitch, o_vec3d *out); /* :248-258 */
float oracle_cast_light_ray(const o_sphere *spheres, int n, const o_vec3d *start,
                            const o_light *l, const o_vec3d *normal);     /* kernel.cu:1433-1544 */
void oracle_light_dirs(const o_vec3d *start, const o_light *l, float *dirs30);   /* kernel.cu:1442-1468 */
int oracle_plane_intersect(const o_plane *p, const o_ray *r, float *t);  /* kernel.cu:370-380 */
int oracle_cube_intersect(const o_cube *c, const o_ray *r, float *t);    /* kernel.cu:400-485 */
void oracle_make_plane(o_plane *p, float px, float py, float pz, float nx, float ny, float nz); /* :364-367 */
void oracle_make_cube(o_cube *c, float ax, float ay, float az, float bx, float by, float bz);   /* :391-396 */
/* mesh: OBJ text -> triangles (loader, kernel.cu:575-747) -> flat BVH (createBvhMesh, :752-937).
 * Restated for well-formed OBJ text; lines the reference mis-parses (blank lines leave its
 * stream state stale) are skipped instead -- documented deviation. */
o_mesh *oracle_mesh_from_obj(const char *text);
void oracle_mesh_free(o_mesh *m);
int oracle_mesh_counts(const o_mesh *m, int *poly_count, int *bvhbox_count, int *has_normals);
const o_triangle *oracle_mesh_triangles(const o_mesh *m);
int oracle_mesh_box(const o_mesh *m, int j, float bounds[6], float orgin[3], const int **indexes, int *length);
int oracle_triangle_intersect(const o_triangle *tri, const o_ray *r, float *t, float *u, float *v); /* :1024-1059 */
/* MSVC rand() replay + scene generator (kernel.cu:1189-1192). */
void oracle_msvc_srand(unsigned int seed);
int oracle_msvc_rand(void);
void oracle_generate_spheres(o_sphere *out, int n, unsigned int seed);
/* Portable math, exported for direct comparison with the device copy. */
float oracle_cosf(float x);
float oracle_sinf(float x);
float oracle_acosf(float x);
float oracle_atan2f(float y, float x);
int oracle_uses_libm(void);
/* 4-spp combine: acc/spp then pack, as the product does (build-defined). */
uint32_t oracle_pack_color(float r, float g, float b);                   /* kernel.cu:1682 */

#ifdef __cplusplus
}
#endif
#endif
