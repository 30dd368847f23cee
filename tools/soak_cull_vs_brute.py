#!/usr/bin/env python3
"""Soak test on the GPU: many random scenes, culled kernel (all shortcuts: beams, eye cones built on
the device, lean normalise / sqrt, fast texel index) against the brute-force instantiation (the
reference's loops as written, IEEE forms everywhere), bit for bit. No oracle involved, so frames can
be large. Every scene is rendered from two cameras in a row (the second one re-uses the scene's
tables and rebuilds its eye cones), now and then with 4 spp or with the table staged in LDS.
Prints the seeds that differ (none expected); --json writes a summary for profiles/."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rt_amd

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=150)
ap.add_argument("--width", type=int, default=960)
ap.add_argument("--height", type=int, default=540)
ap.add_argument("--seed0", type=int, default=0)
ap.add_argument("--json", default="")
a = ap.parse_args()
rt = rt_amd.load()
lib = rt.load_library()
tex, sky = rt.synth_texture(0), rt.synth_texture(1)
bad = []
t0 = time.time()
frames = pixels = 0
kinds = {"tile": {}, "spp4": 0, "table_lds": 0, "mesh": 0, "prims": 0}
for k in range(a.scenes):
    seed = a.seed0 + k
    rng = np.random.default_rng(seed)
    n = int(rng.choice([0, 3, 40, 200, 700, 1500]))
    ext = float(rng.choice([4.0, 10.0, 25.0]))
    sph = (rt.Sphere * max(n, 1))()
    for i in range(n):
        r = float(rng.choice([rng.uniform(0, 1), rng.uniform(0.9, 1.6), 0.03]))
        lib.rt_sphere_init(C.byref(sph[i]), *[float(v) for v in rng.uniform(-0.1 * ext, ext, 3)], r)
    nl = int(rng.integers(1, 5))
    lights = (rt.Light * nl)()
    for i in range(nl):
        pos = rng.uniform(-40, 40, 3) if rng.random() < 0.75 else rng.uniform(0, ext, 3)
        lights[i] = rt.Light(rt.Vec3(*[float(v) for v in pos]), float(rng.uniform(0.5, 30)), *[float(v) for v in rng.uniform(0, 1, 3)])
    cam = rt.Camera(rt.Vec3(*[float(v) for v in rng.uniform(-3, ext + 6, 3)]), rt.Vec3(0, 0, 1), 0.0,
                    float(rng.uniform(0, 360)), float(rng.uniform(-50, 50)))
    sc = rt.Scene()
    sc.set_spheres(sph, n)
    sc.set_texture(tex)
    sc.set_sky(rt.sky_sphere(), sky)
    sc.set_lights(lights, nl)
    has_mesh = False
    if rng.random() < 0.3:       # cubes and planes
        kinds["prims"] += 1
        npl, ncu = int(rng.integers(0, 3)), int(rng.integers(0, 8))
        pl = (rt.Plane * max(npl, 1))()
        for i in range(npl):
            lib.rt_plane_init(C.byref(pl[i]), *[float(v) for v in rng.uniform(-5, ext, 3)], *[float(v) for v in rng.normal(size=3)])
        cu = (rt.Cube * max(ncu, 1))()
        for i in range(ncu):
            p0 = rng.uniform(0, ext, 3)
            lib.rt_cube_init(C.byref(cu[i]), *[float(v) for v in p0], *[float(v) for v in p0 + rng.uniform(0.1, 2.5, 3)])
        sc.set_planes(pl, npl)
        sc.set_cubes(cu, ncu)
    if rng.random() < 0.25:      # a random triangle soup
        nt = int(rng.integers(4, 400))
        lines = []
        for _ in range(nt):
            c = rng.uniform(0, ext, 3)
            for _ in range(3):
                lines.append("v %.5f %.5f %.5f" % tuple(c + rng.normal(size=3) * float(rng.choice([0.05, 0.5, 2.0]))))
        lines += ["f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3) for i in range(nt)]
        sc.set_mesh(rt.mesh_from_obj_text("\n".join(lines) + "\n"))
        has_mesh = True
    tile = 8 if has_mesh else int(rng.choice([8, 8, 8, 16, 32, 64]))
    spp = 4 if rng.random() < 0.15 else 1
    tlds = bool(tile == 8 and rng.random() < 0.1)
    kinds["tile"][tile] = kinds["tile"].get(tile, 0) + 1
    kinds["spp4"] += spp == 4
    kinds["table_lds"] += tlds
    kinds["mesh"] += has_mesh
    for view in range(2):
        if view == 1:   # the reference moves its camera every frame: nudge it, as checkKey does
            cam.Org.z += 0.1
            cam.Org.x -= 0.1
            cam.Camyaw += 1.0
        x = sc.render(a.width, a.height, cam=cam, cull=True, tile=tile, spp=spp, table_lds=tlds)
        y = sc.render(a.width, a.height, cam=cam, cull=False, spp=spp)
        torch.cuda.synchronize()
        frames += 1
        pixels += a.width * a.height
        if not (torch.equal(x["rgba"].view(torch.int32), y["rgba"].view(torch.int32)) and torch.equal(x["packed"], y["packed"])):
            nbad = int((x["rgba"].view(torch.int32) != y["rgba"].view(torch.int32)).any(dim=2).sum())
            bad.append((seed, n, tile, view, nbad))
            print("MISMATCH seed", seed, "spheres", n, "tile", tile, "view", view, "pixels", nbad, flush=True)
    sc.close()
    if k % 25 == 24:
        print(f"{k + 1} scenes, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
print("soak done:", a.scenes, "scenes,", len(bad), "mismatches", bad[:10])
if a.json:
    with open(a.json, "w") as f:
        json.dump({"tool": "tools/soak_cull_vs_brute.py", "what": "culled kernel (every shortcut) == brute-force kernel (reference loops, IEEE forms), "
                   "float4 and packed frames bit for bit", "scenes": a.scenes, "seed0": a.seed0, "frames_compared": frames,
                   "width": a.width, "height": a.height, "pixels_compared": pixels, "mismatching_frames": len(bad),
                   "mismatches": bad[:50], "mix": kinds, "seconds": round(time.time() - t0, 1)}, f, indent=1)
sys.exit(1 if bad else 0)
