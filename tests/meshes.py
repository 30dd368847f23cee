"""Synthetic OBJ text for the mesh tests (the reference's skull2.obj lives on the
author's disk only, kernel.cu:1706)."""
import math


def uv_sphere_obj(cx=4.0, cy=2.0, cz=5.0, r=1.6, n_lat=10, n_lon=16, quads=True):
    """Lat/long sphere with v / vt / vn and f a/b/c faces: triangles at the poles,
    quads (or two triangles) elsewhere."""
    lines = ["# synthetic uv sphere"]
    idx = {}
    k = 0
    for i in range(n_lat + 1):
        th = math.pi * i / n_lat
        for j in range(n_lon + 1):
            ph = 2 * math.pi * j / n_lon
            nx, ny, nz = math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)
            lines.append("v %.6f %.6f %.6f" % (cx + r * nx, cy + r * ny, cz + r * nz))
            lines.append("vt %.6f %.6f" % (j / n_lon, i / n_lat))
            lines.append("vn %.6f %.6f %.6f" % (nx, ny, nz))
            k += 1
            idx[(i, j)] = k
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = idx[(i, j)], idx[(i + 1, j)], idx[(i + 1, j + 1)], idx[(i, j + 1)]
            t = lambda v: "%d/%d/%d" % (v, v, v)
            if i == 0:
                lines.append("f %s %s %s" % (t(a), t(b), t(c)))
            elif i == n_lat - 1:
                lines.append("f %s %s %s" % (t(a), t(b), t(d)))
            elif quads:
                lines.append("f %s %s %s %s" % (t(a), t(b), t(c), t(d)))
            else:
                lines.append("f %s %s %s" % (t(a), t(b), t(c)))
                lines.append("f %s %s %s" % (t(a), t(c), t(d)))
    return "\n".join(lines) + "\n"


def box_obj_no_normals(x0=1.0, y0=0.0, z0=6.0, s=1.5):
    """A box as 12 bare-index triangles (no vn / vt): the loader's third branch."""
    v = [(x0, y0, z0), (x0 + s, y0, z0), (x0 + s, y0 + s, z0), (x0, y0 + s, z0),
         (x0, y0, z0 + s), (x0 + s, y0, z0 + s), (x0 + s, y0 + s, z0 + s), (x0, y0 + s, z0 + s)]
    f = [(1, 3, 2), (1, 4, 3), (5, 6, 7), (5, 7, 8), (1, 2, 6), (1, 6, 5), (4, 7, 3), (4, 8, 7), (1, 5, 8), (1, 8, 4),
         (2, 3, 7), (2, 7, 6)]
    lines = ["v %.6f %.6f %.6f" % p for p in v] + ["f %d %d %d" % t for t in f]
    return "\n".join(lines) + "\n"


def normals_only_obj():
    """vn but no vt, faces a//c, one quad: the loader's second branch."""
    lines = ["v 2 1 4", "v 5 1 4", "v 5 4 4.5", "v 2 4 4.5", "v 3.5 5.5 4.2",
             "vn 0 0 1", "vn 0 0.1 1", "vn 0.1 0 1",
             "f 1//1 2//2 3//3 4//1", "f 4//1 3//2 5//3"]
    return "\n".join(lines) + "\n"
