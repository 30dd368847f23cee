// rt_device.h -- data shared between the host launcher and the gfx950 kernels.
//
// Everything that is uniform over a frame is computed ONCE on the host with the
// same IEEE operations (and the same rt_math.h transcendentals) the reference
// evaluates per pixel on the device (/root/reference/kernel.cu:248-258,
// 1454, 1462-1463, 1538, 1629) and handed to the kernel by value, so it sits in
// SGPRs instead of being recomputed by 64 lanes.
#pragma once
#include <stdint.h>

#define RT_DEV_MAX_LIGHTS 8

// Paddings of the conservative culling tests (rt_trace.inc: beam_keeps, beam_member_test, beam_keeps_block,
// beam_keeps_column; rt_tables.hip: cone_entry). What they must cover is the rounding of the EXACT test. intersect()
// (kernel.cu:332-336, `quadratic()` in rt_trace.inc) can only return true if its binary32 discriminant is >= 0. With
// eps = 2^-24, d = |o - c| and |D| = 1 (to 2e-7): the dot product h carries at most 4 eps d (three products, two sums, the
// rounded o - c), B*B at most 20 eps d^2, 4A*C at most 40 eps d^2 (C: 6 eps d^2; 4A: 4 eps relative on |C| <= d^2 ... with
// R <= d, else read d as R), the final subtraction 4 eps d^2: |disc_float - disc| <= 64 eps d^2 = 3.8e-6 d^2, i.e. 9.5e-7 d^2
// in units of (distance of the ray's line from the centre)^2, 1.2e-6 d^2 with |D|^2 - 1. So the float test can return true only
// for a line within sqrt(R^2 + 1.2e-6 d^2) of the centre. A beam's rays start within r0 <= RT_R0_CAP = 4 of the point a the
// tests measure v = c - a from (a tile's patch of a surface; the whole-normal displacement of triangle hits, kernel.cu:1393,
// can spread them over a couple of units): d^2 <= 2 (|v|^2 + r0^2), needed <= 2.4e-6 |v|^2 + 3.9e-5; the tests use
// REL |v|^2 + ABS = 4e-6 |v|^2 + 8e-5 (the surplus, 1.6e-6 |v|^2, also covers the cancellation error 3 eps |v|^2 of the
// tests' own |v|^2 - (v.u)^2). Round 2 used 4e-5 |v|^2 + 1e-3, fifty times the noise: a third of the BASELINE spheres have
// R < 0.1 and were tested as if R were 0.07 more.
// Block-level tests must cover their members' paddings: sqrt(REL |v|^2 + ABS) <= sqrt(REL) |v| + sqrt(ABS) <= 2e-3 |v| + 9e-3
// with |v| <= dist + r_block.
#ifndef RT_PAD_REL
#define RT_PAD_REL 4.0e-6f
#define RT_PAD_ABS 8.0e-5f
#define RT_BLK_PAD_REL 2.1e-3f
#define RT_BLK_PAD_ABS 0.01f
#endif
#define RT_R0_CAP 4.0f           // a group whose ray origins do not fit a ball of this radius is not culled for
// Allowance on the bound of a light's sample directions (sine of the deviation from the light's axis), for what separates
// the bound's inputs from the exact chain of kernel.cu:1438-1468: toL from a reciprocal square root against the chain's
// re-normalised toL (|dt| <= 5e-7, which the matrix amplifies by K = 4/q + 3 <= 403 for q >= 0.01: 3e-4 of S >= 1),
// cosf/sinf(acosf(z)) against z and sqrt(1 - z^2), the products' roundings, |v_j| <= 1 + 2e-7, the rounding of l.pos - r and
// of the final normalise (2e-7 rad), and this bound's own binary32 evaluation (1e-5): together below 5e-4 relative.
#ifndef RT_SPREAD_MUL
#define RT_SPREAD_MUL 1.002f
#define RT_SPREAD_ADD 5.0e-5f
#endif
#ifndef RT_ORIGIN_ADD
#define RT_ORIGIN_ADD 1.0e-5f    // allowance on the radius of the ball around a group's ray origins (its float evaluation: 1e-6 relative)
#endif
#define RT_DEV_MAX_SPP 16
#define RT_SHADOW_SAMPLES 10   // kernel.cu:1442 `for (int j = 0; j < 10; j++)`
#ifndef RT_LIST_CAP
#define RT_LIST_CAP 128        // survivor-list capacity per wave (float4 entries in LDS);
                               // lists beyond it fall back to the whole table (never seen at C2-C5)
#endif
#ifndef RT_MIN_WAVES_PER_SIMD
#define RT_MIN_WAVES_PER_SIMD 5 // __launch_bounds__ second argument: register budget 512/this.
                               // Measured at C3: 4 -> 1.59 ms; 5 -> 1.45 ms with no spill traffic
                               // (HBM writes = the 166 MB of pixels); 6 -> 1.43 ms but spills
                               // that add 0.6 GB of scratch writes per frame.
#endif
#ifndef RT_MIN_WAVES_ONE_SAMPLE
#define RT_MIN_WAVES_ONE_SAMPLE 7 // the one-sample kernels without a mesh (every BASELINE config at 1 spp, and each pass of the
                               // hipGraph frame) at 72 registers = 7 waves per SIMD (compiled without the SLP vectoriser and
                               // without machine LICM, see the Makefile). With round 3's sample pre-pass the sphere-only
                               // kernel wants 75: at 7 waves two registers live in scratch (one store, two loads per tile),
                               // which measures 4.5 % faster than the clean allocation at 6 waves (0.2695 against 0.2818 ms).
                               // (Their work-counter builds get RT_MIN_WAVES_PER_SIMD.)
#endif
#ifndef RT_MIN_WAVES_MESH
#define RT_MIN_WAVES_MESH 7      // one-sample kernels with the triangle-mesh branches: 71 of 72 registers, no vector spill, 24 scalars in lanes (6 waves: 12; the 4K mesh frame 1.04 -> 0.975 ms; round 2: 89 registers at 5 waves)
#endif
#ifndef RT_MIN_WAVES_MESH_MULTI
#define RT_MIN_WAVES_MESH_MULTI 4 // ... with a sample loop or work counters on top: 128 registers, no spill
#endif
#ifndef RT_WAVES_PER_WG
#define RT_WAVES_PER_WG 4         // wave tiles per workgroup when the table is staged in LDS (1, 2 or 4)
#endif
#ifndef RT_BLOCK
#define RT_BLOCK 16             // spheres per block of the Morton-ordered table (divides 64)
#endif
#define RT_BOX_CAP 128          // leaf-box list capacity per wave (ints in LDS, mesh scenes only)

// Smallest binary32 >= 0.0001 (binary64): `t >= 0.0001` (kernel.cu:342) compares
// the widened float with the double literal, which is equivalent to a float
// compare against this value.
#define RT_T_MIN 1.00000004749745130538940429688e-4f

struct RtLightDev {
    float px, py, pz;   // light.pos
    float size;
    float r, g, b;
    float ux, uy, uz;   // pos / |pos| (beam axis for conservative shadow culling)
    float pos_len;      // |pos|
    float fin;          // 1 if r, g and b are all finite, else 0
    float e1x, e1y, e1z;   // e1, e2: an orthonormal pair across u (the plane a sample direction's deviation is measured in)
    float e2x, e2y, e2z;
    float pad0_, pad1_;
};

struct RtCandHdr {      // one sphere's occluder list for one light (RtFrameAux::cand_hdr)
    int offset, count;  // entries [offset, offset + count) of the light's entry array; count < 0: no list
    float kcap;         // largest beam slope the list was built for
    float kbeam;        // > 0: the beam slope of ANY group of pixels on this sphere for this light (rt_tables.hip:
                        // rt_sphere_beam_slope) -- the kernel takes it instead of bounding the spread per tile; else -1
};
#define RT_CAND_CAP 128         // longest occluder list kept (= RT_LIST_CAP: what a wave's LDS list holds)

struct RtPlaneDev {     // plane: a point and the normal as given (kernel.cu:364-367)
    float ox, oy, oz, nx, ny, nz, pad0_, pad1_;
};
struct RtCubeDev {      // cube: the two corners and (c1+c2)/2 (kernel.cu:391-396)
    float ax, ay, az, bx, by, bz, cx, cy, cz, pad0_, pad1_, pad2_;
};

struct RtTriDev {       // triangle, kernel.cu:206-212, padded to 28 floats
    float p0[3], p1[3], p2[3];
    float n[3];
    float vn[9];
    float vt[6];
    float pad_;
};
struct RtBoxDev {       // one leaf of the flat BVH: bounds + its slice of the index array
    float lo[3], hi[3];
    int start, len;
};

// Frame data the kernel reads rarely or from inside its loops only: lights, shadow-sample
// constants, sky, the few planes/cubes, the mesh. It lives in DEVICE memory behind
// RtFrameConsts::aux (re-uploaded only when its content changes -- a camera move does not
// touch it), so that none of it competes for the SGPR file with the frame uniforms the whole
// kernel uses (by value, the 70 SGPR spills of round 1 came from there).
struct RtFrameAux {
    // shadow-sample uniforms (kernel.cu:1453-1454, 1462-1463)
    float jf[RT_SHADOW_SAMPLES];      // (float)j / 10
    float jcos[RT_SHADOW_SAMPLES];    // cosf(phi_j), phi_j = (float)j/10 * 2.f * 3.1415f
    float jsin[RT_SHADOW_SAMPLES];    // sinf(phi_j)
    float pad0_[2];

    RtLightDev lights[RT_DEV_MAX_LIGHTS];

    // per light, the table once more: ordered by 2-D Morton code of the centres' coordinates
    // ACROSS the light's axis u = l.pos/|l.pos| and cut into blocks of RT_BLOCK, i.e. columns
    // along u. Every shadow ray of a light runs within a few degrees of u, so a beam touches
    // few columns. Two float4 per block: {point on the column axis, lateral radius} and
    // {highest axial extent above that point, 3-D radius, -, -}. Null: use `sorted`/`blocks`.
    const float *lsorted[RT_DEV_MAX_LIGHTS];
    const float *lblocks[RT_DEV_MAX_LIGHTS];

    // per light and per SPHERE S, the spheres that a shadow ray leaving S's surface towards that light can hit at all
    // (host: rt_build_occluder_lists): cand_hdr[light][S] = {offset, count, kcap}, cand_ent[light] + offset = `count`
    // float4 table entries, likeliest occluder first. A group of pixels whose closest hit is S culls those `count`
    // entries against its beam in ONE step instead of walking the light's column blocks. count < 0: no list (too long,
    // non-finite data); valid for beams of slope <= kcap only. Null: no such table.
    const RtCandHdr *cand_hdr[RT_DEV_MAX_LIGHTS];
    const float *cand_ent[RT_DEV_MAX_LIGHTS];

    // sky texture and skybox sphere (kernel.cu:1116-1166)
    const float *sky_r, *sky_g, *sky_b;
    int sky_w, sky_h;
    float sky_cx, sky_cy, sky_cz, sky_r2;   // centre, radius*radius
    float sky_mu_x, sky_mu_y;               // certainty margins of the fast texel-index path, in texels

    // cubes and planes: few, tested exhaustively
    const RtPlaneDev *planes;
    const RtCubeDev *cubes;

    // triangle mesh behind a flat list of leaf boxes
    const RtTriDev *tris;
    const RtBoxDev *boxes;
    const int *tri_idx;
    const float *box_spheres;   // float4 per leaf: bounding sphere {cx,cy,cz,r^2} for beam culling (+ blocks of leaves)
    const float *tri9;          // the three vertices (9 floats) of every (leaf, triangle) pair in the order of
                                // tri_idx: a leaf's triangles are contiguous, one coalesced load stages 7 of them
    const float *tri_bs;        // float4 per (leaf, triangle) pair, same order: bounding sphere {centre, radius} of
                                // the triangle for the per-triangle beam cull; radius +inf = never culled
    const float *tri_nrm;       // float4 per pair, same order: the triangle's unit normal (a beam that grazes the
                                // triangle's plane never culls it, see beam_keeps_triangle)
};

enum {                          // RtFrameConsts::flags
    RT_FLAG_ACCUMULATE = 1,
    RT_FLAG_RESOLVE = 2,
    RT_FLAG_FORCE_SLOW = 4,
    RT_FLAG_MESH_NORMALS = 8,
};

// Frame uniforms used all over the kernel, by value (kernel argument -> SGPRs).
struct RtFrameConsts {
    // frame / band geometry
    int width, height;          // full frame (ray generation uses these)
    int y0, y1;                 // rows rendered by this launch
    int n_spheres, n_lights;
    int spp, sample_base;       // samples taken by this launch, index of the first
    float sample_total;         // divisor at resolve time, as float
    int flags;                  // RT_FLAG_*
    int il_count, il_index, il_rows;   // interleaved row blocks (multi-GPU); il_count <= 1: contiguous band
    int local_rows;             // rows rendered by this launch (= y1 - y0 for a contiguous band)
    int n_planes, n_cubes, n_boxes;
    int ablate;                 // RT_TUNING builds only (RT_ABLATE): skip parts of the kernel to price them; 0 otherwise

    // primary rays (kernel.cu:1624-1631, 248-258). dx and dy of kernel.cu:1624-1625 depend on
    // the column (resp. row) and the sample only: the host evaluates the reference's binary64
    // expressions once per column and row (rt_scene raygen tables) instead of every pixel
    // dividing in binary64 twice. dx_tab[s * width + px], dy_tab[s * height + py].
    const float *dx_tab, *dy_tab;
    float eye_nz;               // -( -1/aspect ) : z component of (dir - eyePos)
    float org_x, org_y, org_z;  // eyePos + cam.Org
    float cos_pitch, sin_pitch, cos_yaw, sin_yaw;

    // object texture (sprite planes)
    const float *tex_r, *tex_g, *tex_b;
    int tex_w, tex_h;
    float tex_mu_x, tex_mu_y;   // certainty margins of the fast texel-index path, in texels

    // the sphere table once more in Morton order of the centres, cut into blocks of RT_BLOCK
    // with a bounding sphere each: culling first tests the blocks, then only the spheres
    // of the blocks a beam can touch. orig_idx[i] is the list position of sorted[i]
    // (primary hits must be examined in list order: first index wins ties).
    const float *sorted;       // float4 per sphere, n_pad entries
    const float *blocks;       // float4 per block: centre and radius (already padded for its members)
    const int *orig_idx;
    int n_blocks;

    // for the primary rays, the table ordered by the DIRECTION of the centres as seen from the
    // ray origin and cut into blocks of RT_BLOCK, i.e. cones from the eye. Two float4 per
    // block: {unit axis, cos(theta)} and {sin(theta), flag, -, -}: theta bounds, for every
    // member, the angle between the axis and any ray from the origin that can pass the
    // member test of a beam with slope <= cone_kcap (flag 1: unbounded, always examined;
    // -1: padding). corig: list position of every entry. Null: use `sorted`/`blocks`.
    float cone_kcap;
    const float *csorted;
    const float *cblocks;
    const int *corig;

    const RtFrameAux *aux;      // device memory, see above

    // outputs
    float *rgba;                // float4 per pixel, band-local, may be null
    uint32_t *packed;           // 0x00RRGGBB per pixel, band-local, may be null
    uint32_t *packed24;         // the same without the zero byte: 3 dwords per 4 pixels, may be null (width % 4 == 0)
    unsigned long long *stats;  // RT_STATS_COUNT counters, may be null

    // Order of the tiles within the launch (one-wave workgroups only). A launch ends when its slowest wave does:
    // waves run 6 ... 90 us, and in grid order the expensive tiles of the last rows start last, so the SIMDs drain
    // for tens of microseconds (a fixed ~45 us per launch, 12 % of a C3 frame, half of an eighth of it).
    // tile_cost[tile] receives every tile's wave duration (shader clocks); tile_perm[block] = (tile_y << 16) | tile_x,
    // sorted from the durations of an earlier frame of the SAME view, starts the longest tiles first. Either may be
    // null. Scheduling only: every tile is rendered exactly once, by the same instructions.
    const unsigned *tile_perm;
    unsigned *tile_cost;
};
