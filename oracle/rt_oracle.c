/*
 * oracle/rt_oracle.c -- TEST INFRASTRUCTURE. NOT part of the product.
 *
 * Scalar CPU restatement of the reference hot path, written to follow
 * /root/reference/kernel.cu expression by expression (operand types, evaluation
 * order, float/double promotions, NaN-as-miss, the r*r*r*r radius quirk, the
 * negative near root). Each function cites the range it follows.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile). FMA
 * contraction and fast-math MUST stay off: the restatement is defined as the
 * IEEE-754 evaluation of the reference's source text.
 *
 * PARITY UNPINNED BY THE REFERENCE (no fixtures exist there) -- see rt_oracle.h.
 */
#include "rt_oracle.h"
#include "rt_oracle_math.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef o_vec3d vec3d;
typedef o_ray ray;

typedef struct { float mat[4][4]; } matrix; /* kernel.cu:35-37 */

typedef struct {
    uint64_t primary_tests, shadow_tests, hit_pixels, unshadowed;
} counters_t;

/* ---- float -> int as the CUDA device does it (cvt.rzi.s32.f32):
 * truncation, NaN -> 0, saturating. Plain C casts are UB for those inputs. */
int oracle_f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

/* ---- vec3d helpers, kernel.cu:46-108 ---- */
static vec3d sub(const vec3d *a, const vec3d *b)
{
    vec3d r = { a->x - b->x, a->y - b->y, a->z - b->z };
    return r;
}
static vec3d add(const vec3d *a, const vec3d *b)
{
    vec3d r = { a->x + b->x, a->y + b->y, a->z + b->z };
    return r;
}
static vec3d multiplyf(vec3d a, float b) /* kernel.cu:76-79 */
{
    vec3d r = { a.x * b, a.y * b, a.z * b };
    return r;
}
static vec3d cross(const vec3d *a, const vec3d *b) /* kernel.cu:81-86 */
{
    vec3d r = { a->y * b->z - a->z * b->y,
                a->z * b->x - a->x * b->z,
                a->x * b->y - a->y * b->x };
    return r;
}
static float dotproduct(const vec3d *a, const vec3d *b) /* kernel.cu:93-96 */
{
    return (a->x * b->x + a->y * b->y + a->z * b->z);
}
static float length(const vec3d *v) /* kernel.cu:98-100 */
{
    return sqrtf(dotproduct(v, v));
}
/* kernel.cu:102-108: length is widened to double, each component is divided in
 * double and narrowed back, AND the argument itself is modified. */
static vec3d normalise(vec3d *v)
{
    double l = length(v);
    if (l != 0) {
        v->x = (float)((double)v->x / l);
        v->y = (float)((double)v->y / l);
        v->z = (float)((double)v->z / l);
        return *v;
    } else {
        vec3d z = { 0, 0, 0 };
        return z;
    }
}
/* kernel.cu:120-128 */
static vec3d multiplym(const matrix *m, vec3d v1)
{
    vec3d v2;
    v2.x = v1.x * m->mat[0][0] + v1.y * m->mat[1][0] + v1.z * m->mat[2][0];
    v2.y = v1.x * m->mat[0][1] + v1.y * m->mat[1][1] + v1.z * m->mat[2][1];
    v2.z = v1.x * m->mat[0][2] + v1.y * m->mat[1][2] + v1.z * m->mat[2][2];
    return v2;
}

/* ---- camera::rotateDir, kernel.cu:248-258 ---- */
static vec3d rotateDir(const vec3d *vec, float yaw, float pitch)
{
    float yawRad = (float)(yaw * (3.1415 / 180));
    float pitchRad = (float)(pitch * (3.1415 / 180));

    float y = vec->y * o_cosf(pitchRad) - vec->z * o_sinf(pitchRad);
    float z = vec->y * o_sinf(pitchRad) + vec->z * o_cosf(pitchRad);
    float x = vec->x * o_cosf(yawRad) + z * o_sinf(yawRad);
    z = -vec->x * o_sinf(yawRad) + z * o_cosf(yawRad);

    vec3d r = { x, y, z };
    return r;
}

/* ---- sphere::intersect, kernel.cu:293-354 ---- */
static int sphere_intersect(const o_sphere *s, const ray *cam_ray, float *t_out)
{
    const vec3d *D = &cam_ray->Dir, *O = &cam_ray->Org, *c = &s->orgin;
    float radius = s->radius;

    float A = (D->x * (D->x) + D->y * (D->y) + D->z * (D->z));
    float B = 2 * (D->x * (O->x - c->x) + D->y * (O->y - c->y) + D->z * (O->z - c->z));
    float C = (O->x - c->x) * (O->x - c->x) + (O->y - c->y) * (O->y - c->y)
            + (O->z - c->z) * (O->z - c->z) - radius * radius;

    float t = (-B + sqrtf(B * B - 4 * A * C)) / (2 * A);
    *t_out = t;

    if (t == 0.f)
        return 1;

    if ((double)t >= 0.0001) {
        float t2 = (-B - sqrtf(B * B - 4 * A * C)) / (2 * A);
        if (t > t2) {
            t = t2;
            *t_out = t;
            return 1;
        }
        return 1;
    }
    return 0;
}

int oracle_sphere_intersect(const o_sphere *s, const o_ray *r, float *t)
{
    return sphere_intersect(s, r, t);
}

void oracle_make_sphere(o_sphere *s, float x, float y, float z, float r)
{
    memset(s, 0, sizeof *s);
    s->orgin.x = x; s->orgin.y = y; s->orgin.z = z;
    s->radius = r * r; /* kernel.cu:287 */
}

/* ---- plane::intersect, kernel.cu:370-380 ---- */
static int plane_intersect(const o_plane *p, const ray *cam_ray, float *t)
{
    float denom = dotproduct(&p->normal, &cam_ray->Dir);
    if (denom < 0) {
        vec3d pl0 = sub(&p->orgin, &cam_ray->Org);
        *t = dotproduct(&pl0, &p->normal) / denom;
        return *t >= 0;
    }
    return 0;
}

/* the reference's min/max are macros (kernel.cu:16-26): (a) > (b) ? (a) : (b) */
#define O_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define O_MIN(a, b) (((a) < (b)) ? (a) : (b))

/* ---- cube::intersect, kernel.cu:457-485 (slab test; tmin may be negative) ---- */
static int cube_intersect(const o_cube *c, const ray *cam_ray, float *t)
{
    float dirx = 1.f / cam_ray->Dir.x;
    float diry = 1.f / cam_ray->Dir.y;
    float dirz = 1.f / cam_ray->Dir.z;

    float t1 = (c->bounds[0].x - cam_ray->Org.x) * dirx;
    float t2 = (c->bounds[1].x - cam_ray->Org.x) * dirx;

    float t3 = (c->bounds[0].y - cam_ray->Org.y) * diry;
    float t4 = (c->bounds[1].y - cam_ray->Org.y) * diry;

    float t5 = (c->bounds[0].z - cam_ray->Org.z) * dirz;
    float t6 = (c->bounds[1].z - cam_ray->Org.z) * dirz;

    float tmin = O_MAX(O_MAX(O_MIN(t1, t2), O_MIN(t3, t4)), O_MIN(t5, t6));
    float tmax = O_MIN(O_MIN(O_MAX(t1, t2), O_MAX(t3, t4)), O_MAX(t5, t6));

    if (tmax < 0) {
        *t = tmax;
        return 0;
    }
    if (tmax < tmin) {
        *t = tmax;
        return 0;
    }
    *t = tmin;
    return 1;
}

int oracle_plane_intersect(const o_plane *p, const o_ray *r, float *t) { return plane_intersect(p, r, t); }
int oracle_cube_intersect(const o_cube *c, const o_ray *r, float *t) { return cube_intersect(c, r, t); }

void oracle_make_plane(o_plane *p, float px, float py, float pz, float nx, float ny, float nz)
{
    memset(p, 0, sizeof *p);                 /* plane(pos, normal), kernel.cu:364-367 */
    p->orgin.x = px; p->orgin.y = py; p->orgin.z = pz;
    p->normal.x = nx; p->normal.y = ny; p->normal.z = nz;
}

void oracle_make_cube(o_cube *c, float ax, float ay, float az, float bx, float by, float bz)
{
    memset(c, 0, sizeof *c);                 /* cube(c1, c2), kernel.cu:391-396 */
    c->bounds[0].x = ax; c->bounds[0].y = ay; c->bounds[0].z = az;
    c->bounds[1].x = bx; c->bounds[1].y = by; c->bounds[1].z = bz;
    c->orgin.x = (ax + bx) / 2;              /* divide(add(c1, c2), 2) */
    c->orgin.y = (ay + by) / 2;
    c->orgin.z = (az + bz) / 2;
    c->normals[0].x = 1; c->normals[1].y = 1; c->normals[2].z = 1;   /* kernel.cu:507 */
}

#include "rt_oracle_mesh.inc"

/* ---- rgbToInt, kernel.cu:547-556 ---- */
uint32_t oracle_rgb_to_int(int r, int g, int b)
{
    if (r > 255) r = 255;
    if (g > 255) g = 255;
    if (b > 255) b = 255;
    return (uint32_t)(((r & 0xff) << 16) + ((g & 0xff) << 8) + (b & 0xff));
}

/* kernel.cu:1682/1688: rgbToInt(fr * 254, fg * 254, fb * 254) -- the float
 * products are implicitly converted to the int parameters. */
uint32_t oracle_pack_color(float r, float g, float b)
{
    return oracle_rgb_to_int(oracle_f2i(r * 254), oracle_f2i(g * 254), oracle_f2i(b * 254));
}

/* ---- rotate(angle, vec), kernel.cu:1263-1280 (non-standard on purpose) ---- */
static matrix rotate(float angle, vec3d vec)
{
    matrix rot;
    memset(&rot, 0, sizeof rot);

    rot.mat[0][0] = o_cosf(angle) + vec.x * vec.x;
    rot.mat[0][1] = vec.x * vec.y * (1.f - o_cosf(angle)) - vec.z * o_sinf(angle);
    rot.mat[0][2] = vec.x * vec.z * (1.f - o_cosf(angle)) - vec.y * o_sinf(angle);

    rot.mat[1][0] = vec.y * vec.x * (1.f - o_cosf(angle)) + vec.z * o_sinf(angle);
    rot.mat[1][1] = o_cosf(angle) + vec.y * vec.y * (1.f - o_cosf(angle));
    rot.mat[1][2] = vec.y * vec.z * (1.f - o_cosf(angle)) - vec.x * o_sinf(angle);

    rot.mat[2][0] = vec.z * vec.x * (1.f - o_cosf(angle)) - vec.y * o_sinf(angle);
    rot.mat[2][1] = vec.z * vec.y * (1.f - o_cosf(angle)) + vec.x * o_sinf(angle);
    rot.mat[2][2] = o_cosf(angle) + vec.z * vec.z * (1.f - o_cosf(angle));

    return rot;
}

typedef struct {
    const o_frame *f;
    counters_t *cnt;
} ctx_t;

/* ---- castRay, sphere branch: kernel.cu:1288-1292, 1330-1342, 1374,
 *      1396-1405, 1427-1431 (mesh/cube/plane loops run zero times) ---- */
static int castRay(const ctx_t *cx, const ray *cam_ray, int *hit_index, float *nt,
                   vec3d *new_org, vec3d *normal, float *tx, float *ty)
{
    const o_frame *f = cx->f;
    int hit_type = 1;
    float nu = 0, nv = 0;
    *nt = INFINITY;

    /* triangles through the flat box list, kernel.cu:1293-1328 */
    if (f->mesh)
        for (int j = 0; j < f->mesh->bvhbox_count; j++) {
            float temp;
            if (cube_intersect(&f->mesh->boxes[j].box, cam_ray, &temp)) {
                for (int i = 0; i < f->mesh->boxes[j].length; i++) {
                    float t, u, v;
                    int idx = f->mesh->boxes[j].indexes[i];
                    if (rayIntersect(cam_ray, &f->mesh->tris[idx], &t, &u, &v)) {
                        if (t < *nt) {
                            *nt = t;
                            nv = v;
                            nu = u;
                            *hit_index = idx;
                            hit_type = 0;
                        }
                    }
                }
            }
        }

    for (int i = 0; i < f->sphere_count; i++) {
        float t;
        cx->cnt->primary_tests++;
        if (sphere_intersect(&f->spheres[i], cam_ray, &t)) {
            if (t < *nt) {
                *nt = t;
                *hit_index = i;
                hit_type = 1;
            }
        }
    }
    /* cubes, kernel.cu:1344-1356 */
    for (int i = 0; i < f->cube_count; i++) {
        float t;
        if (cube_intersect(&f->cubes[i], cam_ray, &t)) {
            if (t < *nt) {
                *nt = t;
                *hit_index = i;
                hit_type = 3;
            }
        }
    }
    /* planes, kernel.cu:1359-1372 */
    for (int i = 0; i < f->plane_count; i++) {
        float t;
        if (plane_intersect(&f->planes[i], cam_ray, &t)) {
            if (t < *nt) {
                *nt = t;
                *hit_index = i;
                hit_type = 2;
            }
        }
    }

    if (*nt != INFINITY) {
        if (hit_type == 0) {            /* kernel.cu:1378-1393 */
            const o_triangle *tr = &f->mesh->tris[*hit_index];
            if (f->mesh->has_normals) {
                vec3d a = multiplyf(tr->vecNormal[0], (1 - nu - nv));
                vec3d b = multiplyf(tr->vecNormal[1], nu);
                vec3d c = multiplyf(tr->vecNormal[2], nv);
                vec3d ab = add(&a, &b);
                *normal = add(&ab, &c);
                *normal = normalise(normal);
            } else {
                *normal = tr->normal;
            }
            *tx = ((1 - nu - nv) * tr->vt[0][0]) + (nu * tr->vt[1][0]) + (nv * tr->vt[2][0]);
            *ty = ((1 - nu - nv) * tr->vt[0][1]) + (nu * tr->vt[1][1]) + (nv * tr->vt[2][1]);
            /* new_org = add(normal, add(Org, Dir*nt)): offset by the WHOLE normal */
            vec3d step = multiplyf(cam_ray->Dir, *nt);
            vec3d hp = add(&cam_ray->Org, &step);
            *new_org = add(normal, &hp);
        }
        if (hit_type == 1) {            /* kernel.cu:1396-1405 */
            vec3d step = multiplyf(cam_ray->Dir, *nt);
            *new_org = add(&cam_ray->Org, &step);
            *normal = sub(new_org, &f->spheres[*hit_index].orgin);
            *normal = normalise(normal);
            /* double arithmetic: the literals 1, 3.1415, 0.5 are int/double */
            *tx = (float)((1 + o_atan2f(normal->z, normal->x) / 3.1415) * 0.5);
            *ty = (float)(o_acosf(normal->y) / 3.1415);
        }
        if (hit_type == 2) {            /* kernel.cu:1407-1416 */
            vec3d step = multiplyf(cam_ray->Dir, *nt);
            *new_org = add(&cam_ray->Org, &step);
            *normal = f->planes[*hit_index].normal;
            *tx = (float)0.5;
            *ty = (float)0.5;
        }
        if (hit_type == 3) {            /* kernel.cu:1418-1425 */
            vec3d step = multiplyf(cam_ray->Dir, *nt);
            *new_org = add(&cam_ray->Org, &step);
            *normal = sub(new_org, &f->cubes[*hit_index].orgin);
            *normal = normalise(normal);
            *tx = (float)((1 + o_atan2f(normal->z, normal->x) / 3.1415) * 0.5);
            *ty = (float)(o_acosf(normal->y) / 3.1415);
        }
        return 1;
    }
    return 0;
}

/* ---- castLightRay, sphere branch: kernel.cu:1433-1471, 1499-1510, 1537-1544 ---- */
static float castLightRay_impl3(const o_mesh *mesh, const o_sphere *spheres, int sphere_count, const o_plane *planes,
                                int plane_count, const o_cube *cubes, int cube_count, const vec3d *start,
                                const o_light *l, const vec3d *normal, counters_t *cnt, float *dirs_out);
static float castLightRay_impl2(const o_sphere *spheres, int sphere_count, const o_plane *planes, int plane_count,
                                const o_cube *cubes, int cube_count, const vec3d *start,
                                const o_light *l, const vec3d *normal, counters_t *cnt, float *dirs_out)
{
    return castLightRay_impl3(NULL, spheres, sphere_count, planes, plane_count, cubes, cube_count, start, l, normal,
                              cnt, dirs_out);
}
static float castLightRay_impl3(const o_mesh *mesh, const o_sphere *spheres, int sphere_count, const o_plane *planes,
                                int plane_count, const o_cube *cubes, int cube_count, const vec3d *start,
                                const o_light *l, const vec3d *normal, counters_t *cnt, float *dirs_out)
{
    float b = 0;
    int shadow = 0;

    vec3d d0 = sub(&l->pos, start);
    vec3d toL = normalise(&d0);

    for (int j = 0; j < 10; j++) {
        vec3d up = { 0, 1, 0 };
        vec3d P = cross(&toL, &up);

        vec3d ps = multiplyf(P, l->size);
        vec3d edge = add(&l->pos, &ps);
        vec3d e0 = sub(&edge, start);
        vec3d toEdge = normalise(&e0);
        float angle = o_cosf((dotproduct(&toL, &toEdge)) * 2);

        float _z = (float)j / 10 * (1.0f - angle) + angle;
        float phi = (float)j / 10 * 2.f * 3.1415f;

        float x = sqrtf(1.f - _z * _z) * o_cosf(phi);
        float y = sqrtf(1.f - _z * _z) * o_sinf(phi);

        vec3d zaxis = { 0, 0, 1 };
        vec3d n1 = normalise(&toL); /* modifies toL */
        vec3d ax0 = cross(&zaxis, &n1);
        vec3d axis = normalise(&ax0);
        vec3d n2 = normalise(&toL); /* modifies toL again */
        float nAngle = o_acosf(dotproduct(&n2, &zaxis));

        matrix rot = rotate(nAngle, axis);
        vec3d v = { x, y, _z };
        vec3d rv = multiplym(&rot, v);
        vec3d nd0 = sub(&l->pos, &rv);
        vec3d new_dir = normalise(&nd0);
        ray light_ray;
        light_ray.Org = *start;
        light_ray.Dir = new_dir;
        if (dirs_out) { dirs_out[3 * j] = new_dir.x; dirs_out[3 * j + 1] = new_dir.y; dirs_out[3 * j + 2] = new_dir.z; }

        shadow = 0;

        /* triangles, kernel.cu:1475-1497 (the reference reads Bvhbox::bvhbox there, a
         * managed copy of the same cube as d_bvhbox) */
        if (mesh)
            for (int bj = 0; bj < mesh->bvhbox_count; bj++) {
                float temp;
                if (cube_intersect(&mesh->boxes[bj].box, &light_ray, &temp)) {
                    for (int i = 0; i < mesh->boxes[bj].length; i++) {
                        float t, u, v;
                        if (rayIntersect(&light_ray, &mesh->tris[mesh->boxes[bj].indexes[i]], &t, &u, &v)) {
                            shadow = 1;
                            break;
                        }
                    }
                    if (shadow) break;
                }
            }

        if (!shadow)
            for (int i = 0; i < sphere_count; i++) {
                float t;
                if (cnt) cnt->shadow_tests++;
                if (sphere_intersect(&spheres[i], &light_ray, &t)) {
                    shadow = 1;
                    break;
                }
            }
        if (!shadow) /* planes, kernel.cu:1511-1523 */
            for (int i = 0; i < plane_count; i++) {
                float t;
                if (plane_intersect(&planes[i], &light_ray, &t)) {
                    shadow = 1;
                    break;
                }
            }
        if (!shadow) /* cubes, kernel.cu:1524-1536 */
            for (int i = 0; i < cube_count; i++) {
                float t;
                if (cube_intersect(&cubes[i], &light_ray, &t)) {
                    shadow = 1;
                    break;
                }
            }
        if (!shadow) {
            b = (float)(b + 0.1); /* float += double literal */
            if (cnt) cnt->unshadowed++;
        }
    }
    float a = dotproduct(normal, &toL);
    b *= a > 0 ? a : 0;
    return b;
}

static float castLightRay_impl(const o_sphere *spheres, int sphere_count, const vec3d *start,
                               const o_light *l, const vec3d *normal, counters_t *cnt, float *dirs_out)
{
    return castLightRay_impl2(spheres, sphere_count, NULL, 0, NULL, 0, start, l, normal, cnt, dirs_out);
}

float oracle_cast_light_ray(const o_sphere *spheres, int n, const o_vec3d *start,
                            const o_light *l, const o_vec3d *normal)
{
    return castLightRay_impl(spheres, n, start, l, normal, NULL, NULL);
}

/* test aid: the 10 sample directions new_dir (kernel.cu:1468) for one start point */
void oracle_light_dirs(const o_vec3d *start, const o_light *l, float *dirs30)
{
    vec3d n = { 0, 1, 0 };
    castLightRay_impl(NULL, 0, start, l, &n, NULL, dirs30);
}

/* Reference indexes the planar texture with an unchecked linear index
 * (kernel.cu:1653, 1160); it runs past the end when ty >= 1 (or tx >= 1 on the
 * last row), which is undefined there. Documented deviation: the LINEAR index
 * is clamped into the plane; every in-bounds index is used exactly as computed. */
static int clamp_index(int idx, const o_sprite *s)
{
    int last = s->width * s->height - 1;
    if (idx < 0) return 0;
    if (idx > last) return last;
    return idx;
}

/* ---- skybox::getFColor, kernel.cu:1147-1166 ---- */
static void getFColor(const o_frame *f, const ray *in_ray, float *r, float *g, float *b)
{
    float t;
    sphere_intersect(f->sky_box, in_ray, &t);

    vec3d step = multiplyf(in_ray->Dir, t);
    vec3d hit_point = add(&in_ray->Org, &step);
    vec3d normal = sub(&hit_point, &f->sky_box->orgin);
    normal = normalise(&normal);

    int x = oracle_f2i((1.f + o_atan2f(normal.z, normal.x) / 3.1415f) * 0.5f * f->sky_tex->width);
    int y = oracle_f2i(o_acosf(normal.y) / 3.1415f * f->sky_tex->height);

    int index = clamp_index(y * f->sky_tex->width + x, f->sky_tex);

    *r = f->sky_tex->r[index];
    *g = f->sky_tex->g[index];
    *b = f->sky_tex->b[index];
}

/* ---- primary ray, kernel.cu:1624-1631 ---- */
static ray primary_ray(int x, int y, int width, int height, float aspect,
                       const o_camera *cam, double off_x, double off_y)
{
    /* (x + 0.5) is double, so the whole product is evaluated in double */
    float dx = (float)(aspect * (2 * (x + off_x) / (float)width) - 1);
    float dy = (float)(aspect * (2 * (y + off_y) / (float)height) * ((float)height / width) - 1);

    vec3d eyePos = { 0, 0, (-1 / aspect) };
    vec3d dir = { dx, dy, 0 };
    vec3d d0 = sub(&dir, &eyePos);
    vec3d nd = normalise(&d0);
    ray cam_ray;
    cam_ray.Org = add(&eyePos, &cam->Org);
    cam_ray.Dir = rotateDir(&nd, cam->Camyaw, cam->Campitch);
    return cam_ray;
}

void oracle_primary_ray(int x, int y, int width, int height, float aspect,
                        const o_camera *cam, double off_x, double off_y, o_ray *out)
{
    *out = primary_ray(x, y, width, height, aspect, cam, off_x, off_y);
}

void oracle_rotate_dir(const o_camera *cam, const o_vec3d *v, float yaw, float pitch, o_vec3d *out)
{
    (void)cam;
    *out = rotateDir(v, yaw, pitch);
}

float oracle_default_aspect(void)
{
    return (float)tan((90 * 0.5 * 3.1415) / 180); /* kernel.cu:1701 (host, double tan) */
}

/* ---- rayTrace, one pixel: kernel.cu:1615-1690 ---- */
static void trace_pixel(const ctx_t *cx, int x, int y, float *rgba, uint32_t *packed)
{
    const o_frame *f = cx->f;
    ray cam_ray = primary_ray(x, y, f->width, f->height, f->aspect, &f->cam, f->off_x, f->off_y);

    int hit_index = 0;
    float n_t;
    vec3d new_org, normal;
    float tx, ty;

    if (castRay(cx, &cam_ray, &hit_index, &n_t, &new_org, &normal, &tx, &ty)) {
        cx->cnt->hit_pixels++;
        int maxX = f->texture->width;
        int maxY = f->texture->height;

        /* multiply(normal, 0.00001): the double literal narrows to float b */
        vec3d eps = multiplyf(normal, (float)0.00001);
        vec3d start_O = add(&eps, &new_org);
        vec3d obj_normal = normal;

        int c_index = oracle_f2i(ty * maxY) * maxX + oracle_f2i(tx * maxX);
        c_index = clamp_index(c_index, f->texture);

        float r = f->texture->r[c_index], g = f->texture->g[c_index], b = f->texture->b[c_index];

        float fr = 0, fg = 0, fb = 0;
        for (int i = 0; i < f->light_size; i++) {
            float brightness = castLightRay_impl3(f->mesh, f->spheres, f->sphere_count, f->planes, f->plane_count,
                                                  f->cubes, f->cube_count, &start_O,
                                                  &f->lights[i], &obj_normal, cx->cnt, NULL);
            fr += brightness * f->lights[i].r * r;
            fg += brightness * f->lights[i].g * g;
            fb += brightness * f->lights[i].b * b;
        }
        if (rgba) { rgba[0] = fr; rgba[1] = fg; rgba[2] = fb; rgba[3] = 1.0f; }
        if (packed) *packed = oracle_pack_color(fr, fg, fb);
        return;
    }
    float r, g, b;
    getFColor(f, &cam_ray, &r, &g, &b);
    if (rgba) { rgba[0] = r; rgba[1] = g; rgba[2] = b; rgba[3] = 1.0f; }
    if (packed) *packed = oracle_pack_color(r, g, b);
}

/* ---- frame loop: the grid of kernel.cu:1780-1783 flattened to row-parallel
 * host threads (pixels are independent: one store per thread, :1682/:1688). ---- */
typedef struct {
    const o_frame *f;
    float *rgba;
    uint32_t *packed;
    int *next_row;
    pthread_mutex_t *mu;
    counters_t cnt;
} job_t;

static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    const o_frame *f = jb->f;
    ctx_t cx = { f, &jb->cnt };
    for (;;) {
        int y = __atomic_fetch_add(jb->next_row, 1, __ATOMIC_RELAXED);
        if (y >= f->y1) break;
        for (int x = 0; x < f->width; x++) {
            size_t o = (size_t)(y - f->y0) * (size_t)f->width + (size_t)x;
            trace_pixel(&cx, x, y, jb->rgba ? jb->rgba + 4 * o : NULL,
                        jb->packed ? jb->packed + o : NULL);
        }
    }
    return NULL;
}

int oracle_render(const o_frame *f, float *rgba, uint32_t *packed,
                  uint64_t counters[4], int nthreads)
{
    if (!f || f->width <= 0 || f->height <= 0 || f->y0 < 0 || f->y1 > f->height || f->y0 > f->y1)
        return 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    int next_row = f->y0;
    job_t jobs[256];
    pthread_t th[256];
    for (int i = 0; i < nthreads; i++) {
        jobs[i].f = f; jobs[i].rgba = rgba; jobs[i].packed = packed;
        jobs[i].next_row = &next_row; jobs[i].mu = NULL;
        memset(&jobs[i].cnt, 0, sizeof jobs[i].cnt);
    }
    if (nthreads == 1) {
        worker(&jobs[0]);
    } else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &jobs[i]);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    if (counters) {
        counters[0] = counters[1] = counters[2] = counters[3] = 0;
        for (int i = 0; i < nthreads; i++) {
            counters[0] += jobs[i].cnt.primary_tests;
            counters[1] += jobs[i].cnt.shadow_tests;
            counters[2] += jobs[i].cnt.hit_pixels;
            counters[3] += jobs[i].cnt.unshadowed;
        }
    }
    return 0;
}

/* ---- MSVC rand() replay (ucrt: state*214013+2531011, (state>>16)&0x7fff) and
 * the sphere scene of kernel.cu:1189-1192. Evaluation order fixed as x,y,z,r
 * (documented convention). ---- */
static unsigned int msvc_state = 1;
void oracle_msvc_srand(unsigned int seed) { msvc_state = seed; }
int oracle_msvc_rand(void)
{
    msvc_state = msvc_state * 214013u + 2531011u;
    return (int)((msvc_state >> 16) & 0x7fff);
}
void oracle_generate_spheres(o_sphere *out, int n, unsigned int seed)
{
    oracle_msvc_srand(seed);
    for (int i = 0; i < n; i++) {
        float x = (float)(oracle_msvc_rand() % 100) / 10;
        float y = (float)(oracle_msvc_rand() % 100) / 10;
        float z = (float)(oracle_msvc_rand() % 100) / 10;
        float r = (float)(oracle_msvc_rand() % 100) / 100;
        oracle_make_sphere(&out[i], x, y, z, r);
    }
}

float oracle_cosf(float x) { return o_cosf(x); }
float oracle_sinf(float x) { return o_sinf(x); }
float oracle_acosf(float x) { return o_acosf(x); }
float oracle_atan2f(float y, float x) { return o_atan2f(y, x); }
int oracle_uses_libm(void)
{
#ifdef RT_ORACLE_LIBM
    return 1;
#else
    return 0;
#endif
}
