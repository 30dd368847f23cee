"""Shared test inputs: the reference's default scene at the BASELINE.json
configs, as inputs for both the HIP path and the oracle."""
import numpy as np


class Inputs:
    def __init__(self, rt, n_spheres, seed=1):
        self.rt = rt
        self.n = n_spheres
        self.spheres = rt.generate_spheres(n_spheres, seed)
        self.tex = rt.synth_texture(0)
        self.sky = rt.synth_texture(1)
        self.sky_box = rt.sky_sphere()
        self.lights = rt.default_lights()
        self.n_lights = 3
        self.cam = rt.default_camera()
        self.aspect = rt.default_aspect()

    def oracle_render(self, oracle, width, height, y0=0, y1=None, off=(0.5, 0.5), nthreads=8, libm=False):
        return oracle.render(self.spheres, self.n, self.tex, self.sky, self.sky_box, self.lights, self.n_lights,
                             self.cam, width, height, self.aspect, y0=y0, y1=y1, off=off, nthreads=nthreads,
                             libm=libm)

    def oracle_render_spp(self, oracle, rt, width, height, spp, y0=0, y1=None, nthreads=8):
        """Mean of `spp` samples exactly as the product defines it: float32 adds in
        sample order, divide by float(spp), then the reference's pack."""
        import ctypes as C
        lib = rt.load_library()
        acc = None
        for k in range(spp):
            ox, oy = C.c_double(), C.c_double()
            assert lib.rt_sample_offset(k, spp, C.byref(ox), C.byref(oy)) == 0
            rgba, _, _ = self.oracle_render(oracle, width, height, y0=y0, y1=y1, off=(ox.value, oy.value),
                                            nthreads=nthreads)
            acc = rgba.copy() if acc is None else (acc + rgba).astype(np.float32)
        mean = (acc[..., :3] / np.float32(spp)).astype(np.float32)
        olib = oracle.load()
        packed = np.empty(mean.shape[:2], dtype=np.uint32)
        flat = mean.reshape(-1, 3)
        pk = packed.reshape(-1)
        for i in range(flat.shape[0]):
            pk[i] = olib.oracle_pack_color(float(flat[i, 0]), float(flat[i, 1]), float(flat[i, 2]))
        return acc, packed

    def scene(self):
        s = self.rt.Scene()
        s.set_spheres(self.spheres, self.n)
        s.set_texture(self.tex)
        s.set_sky(self.sky_box, self.sky)
        s.set_lights(self.lights, self.n_lights)
        return s


# name -> (width, height, n_spheres, y0, y1): small enough for the oracle to
# finish in seconds; the 3840x2160 entries are row bands of the full C3 frame.
GOLDEN_CASES = {
    "c1_256x256_n8": (256, 256, 8, 0, 256),
    "c2_160x90_n256": (160, 90, 256, 0, 90),
    "c3_160x90_n1024": (160, 90, 1024, 0, 90),
    "c3_3840x2160_rows1080": (3840, 2160, 1024, 1080, 1082),
    "c3_3840x2160_rows300": (3840, 2160, 1024, 300, 302),
    "c5_128x72_n4096": (128, 72, 4096, 0, 72),
    # C2 at its BASELINE size: row bands of the 1920x1080 / 256-sphere frame (sky above, spheres below)
    "c2_1920x1080_rows100": (1920, 1080, 256, 100, 104),
    "c2_1920x1080_rows700": (1920, 1080, 256, 700, 704),
}

# C4 at its BASELINE size (3840x2160, 1024 spheres, 4 spp): name -> (width, height, n, spp, y0, y1);
# the fixtures hold the accumulated float sums and the resolved words of those rows
GOLDEN_SPP_CASES = {
    "c4_3840x2160_spp4_rows1080": (3840, 2160, 1024, 4, 1080, 1082),
    "c4_3840x2160_spp4_rows300": (3840, 2160, 1024, 4, 300, 302),
}


def mixed_scene(rt):
    """Spheres + cubes + planes: 200 spheres of the rand()
    replay, the reference's own plane (kernel.cu:1187) and a few boxes."""
    import ctypes as C
    lib = rt.load_library()
    inp = Inputs(rt, 200)
    planes = (rt.Plane * 2)()
    lib.rt_plane_init(C.byref(planes[0]), 0.0, -4.0, 0.0, 0.0, 1.0, 0.0)
    lib.rt_plane_init(C.byref(planes[1]), 0.0, 0.0, -3.0, 0.0, 0.25, 1.0)
    boxes = [(1, 0, 1, 3, 2, 3), (6, 1, 2, 7.5, 2.5, 3.5), (4, 4, 4, 5, 6, 5), (8, -3, 8, 9.5, -1, 9.5), (2.5, 2, 8, 3.5, 3, 9)]
    cubes = (rt.Cube * len(boxes))()
    for i, b in enumerate(boxes):
        lib.rt_cube_init(C.byref(cubes[i]), *[float(v) for v in b])
    inp.planes, inp.n_planes, inp.cubes, inp.n_cubes = planes, 2, cubes, len(boxes)
    return inp


def mixed_oracle_render(inp, oracle, width, height, **kw):
    return oracle.render(inp.spheres, inp.n, inp.tex, inp.sky, inp.sky_box, inp.lights, inp.n_lights, inp.cam,
                         width, height, inp.aspect, cubes=inp.cubes, n_cubes=inp.n_cubes, planes=inp.planes,
                         n_planes=inp.n_planes, nthreads=kw.get("nthreads", 8))
