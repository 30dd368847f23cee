// rt_mesh.cpp -- host side of the triangle-mesh path:
//   mesh(string filename)   OBJ loader            /root/reference/kernel.cu:575-747
//   createBvhMesh()         flat BVH, 10 layers   /root/reference/kernel.cu:752-937
//   getMinMaxP()                                  /root/reference/kernel.cu:971-995
// Produces an object with the reference's `mesh` / `Bvhbox` layout so that
// `objs->mesh1` can be handed to rt_launch_raytrace() unchanged. Pure host code.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rt_engine.h"

void rt_set_error(const char *fmt, ...);

namespace {

struct V3d { float x, y, z; };
inline rt_vec3 to3(const V3d &v) { return rt_vec3{v.x, v.y, v.z}; }

// triangle with its face normal normalise(cross(p1-p0, p2-p0)) (kernel.cu:653-654);
// normalise divides by the length widened to double (kernel.cu:102-108).
rt_triangle triangle_of(const V3d &p0, const V3d &p1, const V3d &p2)
{
    rt_triangle t;
    memset(&t, 0, sizeof t);
    const V3d e1{p1.x - p0.x, p1.y - p0.y, p1.z - p0.z}, e2{p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
    V3d n{e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
    const double l = (double)sqrtf((n.x * n.x + n.y * n.y) + n.z * n.z);
    if (l != 0) {
        n.x = (float)(n.x / l);
        n.y = (float)(n.y / l);
        n.z = (float)(n.z / l);
    } else {
        n = V3d{0, 0, 0};
    }
    t.points[0] = to3(p0);
    t.points[1] = to3(p1);
    t.points[2] = to3(p2);
    t.normal = to3(n);
    return t;
}

void set_normals(rt_triangle &t, const V3d &a, const V3d &b, const V3d &c)
{
    t.vecNormal[0] = to3(a);
    t.vecNormal[1] = to3(b);
    t.vecNormal[2] = to3(c);
}
void set_uv(rt_triangle &t, rt_vec2 a, rt_vec2 b, rt_vec2 c)
{
    t.vt[0] = a;
    t.vt[1] = b;
    t.vt[2] = c;
}

int slash_count(const std::string &s)   // SlashCount, kernel.cu:1063-1070
{
    int n = 0;
    for (char ch : s) n += (ch == '/');
    return n;
}

// splitString + extraction: "7/3/5" -> {7,3,5}, "7//5" -> {7,5} (kernel.cu:1072-1108)
std::vector<int> face_ints(const std::string &tok)
{
    std::string spaced = tok;
    for (char &ch : spaced)
        if (ch == '/') ch = ' ';
    std::istringstream ss(spaced);
    std::vector<int> out;
    int v;
    while (ss >> v) out.push_back(v);
    return out;
}

struct Leaf {
    V3d lo, hi;
    std::vector<int> idx;
};

void grow(const rt_triangle &t, V3d &lo, V3d &hi)   // getMinMaxP
{
    for (int j = 0; j < 3; ++j) {
        const rt_vec3 &p = t.points[j];
        if (p.x > hi.x) hi.x = p.x;
        if (p.y > hi.y) hi.y = p.y;
        if (p.z > hi.z) hi.z = p.z;
        if (p.x < lo.x) lo.x = p.x;
        if (p.y < lo.y) lo.y = p.y;
        if (p.z < lo.z) lo.z = p.z;
    }
}

V3d first_point(const rt_triangle &t) { return V3d{t.points[0].x, t.points[0].y, t.points[0].z}; }

Leaf leaf_of(const std::vector<rt_triangle> &T, std::vector<int> idx)
{
    Leaf l;
    l.lo = first_point(T[idx.front()]);   // bounds start from the first and last triangle's
    l.hi = first_point(T[idx.back()]);    // first vertex (kernel.cu:885-886)
    for (int i : idx) grow(T[i], l.lo, l.hi);
    l.idx = std::move(idx);
    return l;
}

// createBvhMesh: ten passes over the current leaves; a leaf with more than five
// triangles is cut at the middle of its bounds; the cut axis advances y -> x -> z
// after every cut (not every pass); triangles go by their first vertex.
std::vector<Leaf> build_leaves(const std::vector<rt_triangle> &T)
{
    std::vector<int> all(T.size());
    for (size_t i = 0; i < T.size(); ++i) all[i] = (int)i;
    std::vector<Leaf> cur{leaf_of(T, all)};
    int axis = 0;   // 0: y, 1: x, 2: z
    for (int pass = 0; pass < 10; ++pass) {
        std::vector<Leaf> next;
        for (Leaf &lf : cur) {
            if (lf.idx.size() <= 5) {
                next.push_back(std::move(lf));
                continue;
            }
            const float mid = axis == 0 ? (lf.hi.y + lf.lo.y) / 2 : axis == 1 ? (lf.hi.x + lf.lo.x) / 2 : (lf.hi.z + lf.lo.z) / 2;
            std::vector<int> below, above;
            for (int i : lf.idx) {
                const rt_vec3 &p = T[i].points[0];
                const float key = axis == 0 ? p.y : axis == 1 ? p.x : p.z;
                (key <= mid ? below : above).push_back(i);
            }
            if (!below.empty()) next.push_back(leaf_of(T, std::move(below)));
            if (!above.empty()) next.push_back(leaf_of(T, std::move(above)));
            axis = (axis + 1) % 3;
        }
        cur = std::move(next);
    }
    return cur;
}

rt_mesh *mesh_from_stream(std::istream &in)
{
    std::vector<V3d> p, vn;
    std::vector<rt_vec2> vt;
    std::vector<rt_triangle> tris;
    bool has_normals = false;
    std::string line;
    const V3d zero{0, 0, 0};
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        char c = 0;
        if (!(ss >> c)) continue;                 // blank line (the reference would mis-parse it)
        const char type = line.size() > 1 ? line[1] : 0;
        if (c == 'v' || c == 'V') {
            char junk;
            if (type == 'n' || type == 'N') {
                V3d n{0, 0, 0};
                ss >> junk >> n.x >> n.y >> n.z;
                vn.push_back(n);
            } else if (type == 't' || type == 'T') {
                rt_vec2 uv{0, 0};
                ss >> junk >> uv.u >> uv.v;
                vt.push_back(uv);
            } else {
                V3d v{0, 0, 0};
                ss >> v.x >> v.y >> v.z;
                p.push_back(v);
            }
        } else if (c == 'f' || c == 'F') {
            std::string tok[4];
            int ntok = 0;
            while (ntok < 4 && (ss >> tok[ntok])) ++ntok;
            if (ntok < 3) continue;
            const int slashes = slash_count(line);
            int f[4] = {0, 0, 0, 0}, it[4] = {0, 0, 0, 0}, in_[4] = {0, 0, 0, 0};
            auto P = [&](int k) -> const V3d & { return p.at((size_t)f[k] - 1); };
            if (!vn.empty() && !vt.empty()) {     // "a/b/c": position / texture / normal
                for (int k = 0; k < ntok; ++k) {
                    const std::vector<int> v = face_ints(tok[k]);
                    f[k] = v.size() > 0 ? v[0] : 0;
                    it[k] = v.size() > 1 ? v[1] : 0;
                    in_[k] = v.size() > 2 ? v[2] : 0;
                }
                auto N = [&](int k) -> const V3d & { return vn.at((size_t)in_[k] - 1); };
                auto UV = [&](int k) { return vt.at((size_t)it[k] - 1); };
                if (slashes <= 6) {
                    rt_triangle t = triangle_of(P(0), P(1), P(2));
                    set_normals(t, N(0), N(1), N(2));
                    set_uv(t, UV(0), UV(1), UV(2));
                    tris.push_back(t);
                }
                if (slashes >= 8) {               // quad: both halves carry the first half's face normal
                    rt_triangle t1 = triangle_of(P(0), P(1), P(2));
                    set_normals(t1, N(0), N(1), N(2));
                    set_uv(t1, UV(0), UV(1), UV(2));
                    rt_triangle t2 = triangle_of(P(0), P(2), P(3));
                    t2.normal = t1.normal;
                    set_normals(t2, N(0), N(2), N(3));
                    set_uv(t2, UV(0), UV(2), UV(2));   // sic: vt[pvt[2]] twice (kernel.cu:672)
                    tris.push_back(t1);
                    tris.push_back(t2);
                }
                has_normals = true;
            } else if (!vn.empty()) {             // "a//c": position // normal
                for (int k = 0; k < ntok; ++k) {
                    const std::vector<int> v = face_ints(tok[k]);
                    f[k] = v.size() > 0 ? v[0] : 0;
                    in_[k] = v.size() > 1 ? v[1] : 0;
                }
                auto N = [&](int k) -> const V3d & { return vn.at((size_t)in_[k] - 1); };
                const rt_vec2 a{0, 0}, b{0, 1}, cc{1, 0};
                if (slashes <= 6) {
                    rt_triangle t = triangle_of(P(0), P(1), P(2));
                    set_normals(t, N(0), N(1), N(2));
                    set_uv(t, a, b, cc);
                    tris.push_back(t);
                }
                if (slashes >= 8) {
                    rt_triangle t1 = triangle_of(P(0), P(1), P(2));
                    set_normals(t1, N(0), N(1), N(2));
                    set_uv(t1, a, b, cc);
                    rt_triangle t2 = triangle_of(P(0), P(2), P(3));
                    t2.normal = t1.normal;
                    set_normals(t2, N(0), N(1), N(2));   // sic: kernel.cu:710
                    set_uv(t2, a, b, cc);
                    tris.push_back(t1);
                    tris.push_back(t2);
                }
                has_normals = true;
            } else {                              // bare indices
                for (int k = 0; k < ntok; ++k) f[k] = atoi(tok[k].c_str());
                if (slashes <= 6) {
                    rt_triangle t = triangle_of(P(0), P(1), P(2));
                    set_normals(t, zero, zero, zero);
                    set_uv(t, rt_vec2{(float)0.666413, (float)0.250594}, rt_vec2{(float)0.333587, (float)0.250594},
                           rt_vec2{(float)0.333587, (float)0.000975});
                    tris.push_back(t);
                }
                if (slashes == 8) {
                    rt_triangle t1 = triangle_of(P(0), P(1), P(2));
                    rt_triangle t2 = triangle_of(P(0), P(2), P(3));
                    t2.normal = t1.normal;
                    set_uv(t1, rt_vec2{0, 0}, rt_vec2{0, 1}, rt_vec2{1, 0});
                    set_uv(t2, rt_vec2{0, 0}, rt_vec2{0, 1}, rt_vec2{1, 0});
                    tris.push_back(t1);
                    tris.push_back(t2);
                }
                has_normals = false;
            }
        }
    }
    if (tris.empty()) {
        rt_set_error("rt_mesh: no triangles in the OBJ data");
        return nullptr;
    }
    const std::vector<Leaf> leaves = build_leaves(tris);

    rt_mesh *m = (rt_mesh *)calloc(1, sizeof(rt_mesh));
    m->poly_count = (int)tris.size();
    m->bvhLayer_count = 10;
    m->has_normals = has_normals ? 1 : 0;
    m->h_tri_arr = (rt_triangle *)malloc(sizeof(rt_triangle) * tris.size());
    memcpy(m->h_tri_arr, tris.data(), sizeof(rt_triangle) * tris.size());
    m->d_tri_arr = m->h_tri_arr;
    m->indexes = (int *)malloc(sizeof(int) * tris.size());
    for (size_t i = 0; i < tris.size(); ++i) m->indexes[i] = (int)i;
    m->bvhbox_count = (int)leaves.size();
    m->h_box = (rt_bvhbox *)calloc(leaves.size(), sizeof(rt_bvhbox));
    m->d_box = m->h_box;
    for (size_t j = 0; j < leaves.size(); ++j) {
        rt_bvhbox &b = m->h_box[j];
        b.bvhbox = (rt_cube *)malloc(sizeof(rt_cube));
        rt_cube_init(b.bvhbox, leaves[j].lo.x, leaves[j].lo.y, leaves[j].lo.z, leaves[j].hi.x, leaves[j].hi.y,
                     leaves[j].hi.z);
        b.d_bvhbox = b.bvhbox;
        b.length = (int)leaves[j].idx.size();
        b.indexes = (int *)malloc(sizeof(int) * leaves[j].idx.size());
        memcpy(b.indexes, leaves[j].idx.data(), sizeof(int) * leaves[j].idx.size());
        b.d_indexes = b.indexes;
    }
    return m;
}

}  // namespace

extern "C" rt_mesh *rt_mesh_from_obj_text(const char *text)
{
    if (!text) {
        rt_set_error("rt_mesh_from_obj_text: null text");
        return nullptr;
    }
    std::istringstream in{std::string(text)};
    try {
        return mesh_from_stream(in);
    } catch (const std::exception &e) {   // index out of range in a face
        rt_set_error("rt_mesh: malformed OBJ data (%s)", e.what());
        return nullptr;
    }
}

extern "C" rt_mesh *rt_mesh_load_obj(const char *path)
{
    std::ifstream file(path ? path : "");
    if (!file.is_open()) {   // the reference returns a half-constructed mesh here (kernel.cu:583-585)
        rt_set_error("rt_mesh_load_obj: cannot open '%s'", path ? path : "(null)");
        return nullptr;
    }
    try {
        return mesh_from_stream(file);
    } catch (const std::exception &e) {
        rt_set_error("rt_mesh: malformed OBJ data (%s)", e.what());
        return nullptr;
    }
}

extern "C" void rt_mesh_free(rt_mesh *m)
{
    if (!m) return;
    for (int j = 0; j < m->bvhbox_count; ++j) {
        free(m->h_box[j].bvhbox);
        free(m->h_box[j].indexes);
    }
    free(m->h_box);
    free(m->h_tri_arr);
    free(m->indexes);
    free(m);
}
