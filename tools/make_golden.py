#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/librt_oracle.so).

The reference ships no golden vectors and cannot be built here, so these are
outputs of the CPU restatement on the default scene; they pin the oracle
against drift and give the GPU tests a fixed target. Re-run after any
deliberate change to the oracle:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_py  # noqa: E402
import rt_amd  # noqa: E402
from scenes import GOLDEN_CASES, GOLDEN_SPP_CASES, Inputs, mixed_oracle_render, mixed_scene  # noqa: E402


def main():
    rt = rt_amd.load()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])     # names to (re)generate; none = all
    for name, (w, h, n, spp, y0, y1) in GOLDEN_SPP_CASES.items():
        if only and name not in only:
            continue
        acc, packed = Inputs(rt, n).oracle_render_spp(oracle_py, rt, w, h, spp, y0=y0, y1=y1)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), acc=acc, packed=packed)
        print(name, acc.shape)
    for name, (w, h, n, y0, y1) in GOLDEN_CASES.items():
        if only and name not in only:
            continue
        inp = Inputs(rt, n)
        rgba, packed, cnt = inp.oracle_render(oracle_py, w, h, y0=y0, y1=y1)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), rgb=rgba[..., :3].copy(), packed=packed,
                            counters=np.array([cnt["primary_tests"], cnt["shadow_tests"], cnt["hit_pixels"],
                                               cnt["unshadowed"]], dtype=np.uint64))
        print(name, rgba.shape, {k: v for k, v in cnt.items()})
    if only:
        return
    # 4-spp extension (build-defined): accumulated float sums + resolved words
    inp = Inputs(rt, 256)
    acc, packed = inp.oracle_render_spp(oracle_py, rt, 96, 54, 4)
    np.savez_compressed(os.path.join(out_dir, "spp4_96x54_n256.npz"), acc=acc, packed=packed)
    print("spp4_96x54_n256", acc.shape)
    # spheres + cubes + planes
    rgba, packed, cnt = mixed_oracle_render(mixed_scene(rt), oracle_py, 160, 96)
    np.savez_compressed(os.path.join(out_dir, "mixed_160x96.npz"), rgb=rgba[..., :3].copy(), packed=packed,
                        counters=np.array([cnt["primary_tests"], cnt["shadow_tests"], cnt["hit_pixels"],
                                           cnt["unshadowed"]], dtype=np.uint64))
    print("mixed_160x96", cnt)
    # triangle mesh (uv sphere, quads + triangles, vn + vt) among 64 spheres
    import meshes
    inp = Inputs(rt, 64)
    om = oracle_py.Mesh(meshes.uv_sphere_obj())
    rgba, packed, cnt = oracle_py.render(inp.spheres, inp.n, inp.tex, inp.sky, inp.sky_box, inp.lights, 3, inp.cam,
                                         160, 90, inp.aspect, nthreads=8, mesh=om.handle)
    np.savez_compressed(os.path.join(out_dir, "mesh_160x90.npz"), rgb=rgba[..., :3].copy(), packed=packed,
                        counters=np.array([cnt["primary_tests"], cnt["shadow_tests"], cnt["hit_pixels"],
                                           cnt["unshadowed"]], dtype=np.uint64))
    print("mesh_160x90", cnt)


if __name__ == "__main__":
    main()
