// rt_scenegen.cpp -- host-side scene construction helpers (no GPU involved).
//
//   rt_sphere_init          sphere::sphere(org, r)        /root/reference/kernel.cu:285-288
//   rt_generate_spheres     object::loadMesh sphere fill  /root/reference/kernel.cu:1189-1192
//   rt_synth_texture        stands in for sprite(file)    /root/reference/Sprite.cpp:28-52
//   rt_load_ppm             sprite(file) without OpenCV   /root/reference/Sprite.cpp:28-52
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/rt_engine.h"

void rt_set_error(const char *fmt, ...);

extern "C" void rt_sphere_init(rt_sphere *s, float x, float y, float z, float r)
{
    memset(s, 0, sizeof *s);
    s->orgin.x = x;
    s->orgin.y = y;
    s->orgin.z = z;
    s->radius = r * r;   // kernel.cu:287 -- intersect() squares it once more (:334)
}

// plane(pos, normal), kernel.cu:364-367
extern "C" void rt_plane_init(rt_plane *p, float px, float py, float pz, float nx, float ny, float nz)
{
    memset(p, 0, sizeof *p);
    p->orgin.x = px; p->orgin.y = py; p->orgin.z = pz;
    p->normal.x = nx; p->normal.y = ny; p->normal.z = nz;
}

// cube(c1, c2), kernel.cu:391-396: orgin = divide(add(c1, c2), 2)
extern "C" void rt_cube_init(rt_cube *c, float ax, float ay, float az, float bx, float by, float bz)
{
    memset(c, 0, sizeof *c);
    c->bounds[0].x = ax; c->bounds[0].y = ay; c->bounds[0].z = az;
    c->bounds[1].x = bx; c->bounds[1].y = by; c->bounds[1].z = bz;
    c->orgin.x = (ax + bx) / 2;
    c->orgin.y = (ay + by) / 2;
    c->orgin.z = (az + bz) / 2;
    c->normals[0].x = 1; c->normals[1].y = 1; c->normals[2].z = 1;   // kernel.cu:507
}

// The reference fills the scene from un-seeded C rand() under MSVC, i.e. the
// ucrt LCG from state 1: state = state*214013 + 2531011; return (state>>16)&0x7fff.
struct MsvcRand {
    uint32_t state;
    explicit MsvcRand(uint32_t seed) : state(seed) {}
    int next()
    {
        state = state * 214013u + 2531011u;
        return (int)((state >> 16) & 0x7fffu);
    }
};

extern "C" int rt_msvc_rand_sequence(unsigned int seed, int *out, int n)
{
    if (!out || n < 0) return RT_ERR_INVALID;
    MsvcRand r(seed);
    for (int i = 0; i < n; ++i) out[i] = r.next();
    return RT_OK;
}

// sphere({rand()%100/10, rand()%100/10, rand()%100/10}, rand()%100/100), with the
// draws consumed in the order x, y, z, r (documented convention, SURVEY.md 8(c):
// the braced list is left-to-right; MSVC's order between the two constructor
// arguments is unspecified, and the scene is an explicit input everywhere else).
extern "C" int rt_generate_spheres(rt_sphere *out, int n, unsigned int seed)
{
    if (n < 0 || (n > 0 && !out)) {
        rt_set_error("rt_generate_spheres: invalid argument");
        return RT_ERR_INVALID;
    }
    MsvcRand rng(seed);
    for (int i = 0; i < n; ++i) {
        const float x = (float)(rng.next() % 100) / 10;
        const float y = (float)(rng.next() % 100) / 10;
        const float z = (float)(rng.next() % 100) / 10;
        const float r = (float)(rng.next() % 100) / 100;
        rt_sphere_init(&out[i], x, y, z, r);
    }
    return RT_OK;
}

// ---------------------------------------------------------------------------
// synthetic textures: 8-bit patterns k(x,y) turned into planes k/255 exactly as
// Sprite.cpp:43-45 turns decoded bytes into floats.
// ---------------------------------------------------------------------------
static const int kSynthW[2] = {512, 2048};
static const int kSynthH[2] = {512, 1024};

extern "C" int rt_synth_texture_size(int kind, int *width, int *height)
{
    if (kind < 0 || kind > 1 || !width || !height) return RT_ERR_INVALID;
    *width = kSynthW[kind];
    *height = kSynthH[kind];
    return RT_OK;
}

extern "C" int rt_synth_texture(int kind, float *r, float *g, float *b)
{
    if (kind < 0 || kind > 1 || !r || !g || !b) return RT_ERR_INVALID;
    const int w = kSynthW[kind], h = kSynthH[kind];
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int kr, kg, kb;
            if (kind == 0) {   // object texture: two ramps + a 32-texel checker
                kr = 64 + ((x + 2 * y) & 127);
                kg = 48 + (((3 * x + y) >> 1) & 127);
                kb = (((x >> 5) + (y >> 5)) & 1) ? 200 : 90;
            } else {           // sky: vertical gradient with faint meridian bands
                kr = 40 + (y >> 3);
                kg = 80 + (y >> 3);
                kb = 255 - (y >> 4) - 16 * ((x >> 7) & 1);
            }
            const size_t i = (size_t)y * w + x;
            r[i] = (float)kr / 255;
            g[i] = (float)kg / 255;
            b[i] = (float)kb / 255;
        }
    return RT_OK;
}

// ---------------------------------------------------------------------------
// binary PPM (P6, maxval 255) -> planar float planes, r = byte/255 ...
// ---------------------------------------------------------------------------
static int ppm_token(FILE *f, int *out)
{
    int c = fgetc(f);
    for (;;) {
        while (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = fgetc(f);
        if (c == '#') {
            while (c != '\n' && c != EOF) c = fgetc(f);
            continue;
        }
        break;
    }
    if (c < '0' || c > '9') return 0;
    long v = 0;
    while (c >= '0' && c <= '9') {
        v = v * 10 + (c - '0');
        if (v > 1000000) return 0;
        c = fgetc(f);
    }
    *out = (int)v;
    return 1;   // exactly one whitespace byte after the token has been consumed
}

extern "C" int rt_load_ppm(const char *path, float **r, float **g, float **b, int *width, int *height)
{
    if (!path || !r || !g || !b || !width || !height) return RT_ERR_INVALID;
    FILE *f = fopen(path, "rb");
    if (!f) {
        rt_set_error("rt_load_ppm: cannot open '%s'", path);
        return RT_ERR_INVALID;
    }
    int w = 0, h = 0, maxv = 0;
    const bool magic = (fgetc(f) == 'P' && fgetc(f) == '6');
    if (!magic || !ppm_token(f, &w) || !ppm_token(f, &h) || !ppm_token(f, &maxv) || w <= 0 || h <= 0 ||
        maxv != 255) {
        fclose(f);
        rt_set_error("rt_load_ppm: '%s' is not a binary P6 PPM with maxval 255", path);
        return RT_ERR_INVALID;
    }
    const size_t n = (size_t)w * (size_t)h;
    unsigned char *raw = (unsigned char *)malloc(n * 3);
    float *pr = (float *)malloc(n * sizeof(float)), *pg = (float *)malloc(n * sizeof(float)),
          *pb = (float *)malloc(n * sizeof(float));
    if (!raw || !pr || !pg || !pb || fread(raw, 3, n, f) != n) {
        fclose(f);
        free(raw); free(pr); free(pg); free(pb);
        rt_set_error("rt_load_ppm: '%s' is truncated", path);
        return RT_ERR_INVALID;
    }
    fclose(f);
    for (size_t i = 0; i < n; ++i) {   // Sprite.cpp:43-45 (there BGR order from OpenCV)
        pr[i] = (float)raw[3 * i + 0] / 255;
        pg[i] = (float)raw[3 * i + 1] / 255;
        pb[i] = (float)raw[3 * i + 2] / 255;
    }
    free(raw);
    *r = pr; *g = pg; *b = pb;
    *width = w;
    *height = h;
    return RT_OK;
}

extern "C" void rt_free_planes(float *r, float *g, float *b)
{
    free(r);
    free(g);
    free(b);
}
