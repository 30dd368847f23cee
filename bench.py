#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ray-tracing hot path on MI355X.

Metric (BASELINE.json): primary Mrays/s (+ frames/s) at 3840x2160, 1024 spheres,
1 spp, on 1/2/4/8 GPUs. One "step" = one frame: every rank renders its
contiguous row band with one HIP kernel launch (float4 linear-colour buffer +
packed 0x00RRGGBB buffer, both resident in HBM), then -- for N > 1 -- a single
RCCL gather of the packed bands to rank 0 reassembles the image.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      HBM-write roofline of the frame kernel (the roof north_star names),
  roofline_valu the roof that actually binds (fp32 VALU),
  cpu_baseline  the CPU oracle timed on this box's cores on a bounded row sample
                (rank 0, N=1 only), with GPU-vs-CPU parity checked on those rows.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rt_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak fp32 vector
FLOP_PER_TEST = 17           # the 17-flop discriminant


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spheres", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tile", type=int, default=0, help="tile width 8/16/32/64 (0 = library default)")
    ap.add_argument("--no-cull", action="store_true", help="brute-force loops exactly as the reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-band-stride", type=int, default=8,
                    help="CPU baseline renders every k-th 8-row band of the frame")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) for real runs; gloo + --share-gpu only to rehearse the "
                         "N>1 code path on a box with fewer GPUs than ranks")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--gather-words", type=int, default=24, choices=[24, 32],
                    help="N > 1: bits per pixel a rank sends to the root (24: packed word without its zero byte)")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=[1, 2],
                    help="consecutive frames alternate between this many HIP streams, each with its own "
                         "framebuffers (2: one frame's last waves overlap the next frame's first)")
    return ap.parse_args()


def cpu_baseline(rt, scene, args, gpu_rgba, gpu_packed):
    """Time the CPU oracle (kind 'port': the C restatement of the reference's
    kernel) on a deterministic row sample of the SAME workload with every host
    core, and check the GPU frame against it on those rows. bench.py's
    cpu_baseline leg is one of the three places allowed to touch oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, args.cpu_threads)      # a 1-GPU box's CPU share is 16 cores
    w, h = args.width, args.height
    bands = list(range(0, h, 8 * args.cpu_band_stride))
    rows = 0
    tests = 0
    mismatched = 0
    t_total = 0.0
    for y0 in bands:
        y1 = min(y0 + 8, h)
        t0 = time.perf_counter()
        rgba, packed, cnt = oracle_py.render(scene.spheres, scene.n_spheres, scene.texture, scene.sky, scene.sky_box,
                                             scene.lights, scene.n_lights, rt.default_camera(), w, h,
                                             rt.default_aspect(), y0=y0, y1=y1, nthreads=cores)
        t_total += time.perf_counter() - t0
        rows += y1 - y0
        tests += cnt["primary_tests"] + cnt["shadow_tests"]
        if gpu_rgba is not None:
            mismatched += int((rgba.view(np.uint32) != gpu_rgba[y0:y1].view(np.uint32)).any(axis=2).sum())
            mismatched += int((packed != gpu_packed[y0:y1]).sum())
    rays = rows * w
    return {
        "value": rays / t_total / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"every {args.cpu_band_stride}th 8-row band of the {w}x{h}/{args.spheres}-sphere frame "
                  f"({rows} rows, {rays} rays, {t_total:.1f} s)",
        "tests_per_ray": tests / rays,
        "gpu_vs_cpu_mismatched_pixels": mismatched if gpu_rgba is not None else None,
    }


def valu_issue(args, world, ms_per_step):
    """VALU instruction issue rate against the plain-op issue rate measured on this chip
    (tools/ubench/valu_rate.hip: 1.05 ns per wave64 v_mul/v_add per SIMD, profiles/r01_valu_issue_rates.txt).
    Instructions per wave come from the committed PMC pass (SQ_INSTS_VALU), the time is this run's."""
    if world != 1 or (args.width, args.height, args.spheres, args.spp) != (3840, 2160, 1024, 1) or args.no_cull:
        return {}
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_frame_kernel_pmc_summary.json")))
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        per_wave = d["derived"]["valu_insts_per_wave"]
        waves = d["pmc_avg_per_dispatch"]["SQ_WAVES"]
    except Exception:
        return {}
    simds = 256 * 4
    rate = per_wave * waves / simds / (ms_per_step * 1e-3)          # VALU instructions per second per SIMD
    peak = 1.0 / 1.05e-9
    return {"valu_insts_per_wave_pmc": per_wave, "valu_issue_rate_per_simd_Ginst_s": rate / 1e9,
            "valu_issue_peak_per_simd_Ginst_s": peak / 1e9, "valu_issue_frac": rate / peak,
            "valu_issue_note": "plain fp32 ops issue at 0.95 G/s per SIMD; compares, selects, binary64, division "
                               "helpers take 1.7x and transcendentals 3.3x that slot, so the pipe is busier than "
                               "this fraction (DESIGN.md section 4, roofline)"}


def pmc_traffic(args, world):
    """HBM bytes per launch of the frame kernel from the committed rocprofv3 PMC
    passes (profiles/*_pmc_summary.json; FETCH_SIZE x2 + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes). bench.py cannot collect counters itself;
    the number applies to the default C3 single-GPU configuration only."""
    if world != 1 or (args.width, args.height, args.spheres, args.spp) != (3840, 2160, 1024, 1) or args.no_cull:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_frame_kernel_pmc_summary.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f)["derived"].get("hbm_traffic_bytes")
    except Exception:
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched through torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the ray-tracing path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"   # gloo rehearsal stages through host memory

    rt = rt_amd.load()
    scene = rt.Scene.default(args.spheres, args.seed)
    w, h = args.width, args.height
    from ray_tracer_engine_amd import distributed as rd
    BLOCK = 16
    if world > 1:
        # balanced split: 16-row blocks dealt round-robin (contiguous bands leave the
        # sky-heavy ranks idle: 0.36 vs 0.28 ms at N=8), compact band-local buffers
        my_rows = rt.interleaved_rows(h, rank, world, BLOCK)
        rows = len(my_rows)
        max_rows = rd.max_interleaved_rows(h, world, BLOCK)
        interleave = (world, rank, BLOCK)
        y0 = y1 = 0
    else:
        y0, y1 = 0, h
        rows = max_rows = h
        interleave = None

    # outputs resident in HBM: float4 linear colour + packed words, band-local. Two sets:
    # frames alternate between them (and between two streams when --frames-in-flight 2), so
    # that frame i's last waves -- and, for N > 1, its gather -- overlap frame i+1's kernel.
    nfl = args.frames_in_flight
    rgb24 = world > 1 and args.gather_words == 24 and w % 4 == 0
    rgba2 = [torch.empty((rows, w, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    packed2 = [torch.zeros((max_rows, w), dtype=torch.int32, device="cuda") for _ in range(2)]
    # what a rank sends: the packed rows, or (default) their 3-bytes-per-pixel form written by the same launch
    send2 = [torch.zeros((max_rows, rd.rgb24_row_words(w)), dtype=torch.int32, device="cuda") for _ in range(2)] \
        if rgb24 else packed2
    roots = [rd.InterleavedGather(h, w, world, coll_dev, BLOCK, rgb24=rgb24) for _ in range(2)] \
        if (world > 1 and rank == 0) else None
    gathered2 = [roots[0].views, roots[1].views] if roots else [None, None]
    frame = None
    main_stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(2)] if nfl == 2 else [main_stream, main_stream]
    fds = [scene.frame_desc(w, h, pixels=packed2[b].data_ptr(), rgba=rgba2[b].data_ptr(), y0=y0, y1=y1, spp=args.spp,
                            cull=not args.no_cull, tile=args.tile, interleave=interleave,
                            packed24=send2[b].data_ptr() if rgb24 else 0) for b in range(2)]

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    pending = [None, None]

    def finish(b):
        """Complete frame b's gather and, on the root, put the rows in place (on b's stream)."""
        if pending[b] is not None:
            with torch.cuda.stream(streams[b]):
                pending[b].wait()
                pending[b] = None
                if rank == 0:
                    nonlocal frame
                    frame = roots[b].assemble()       # one index_select puts every row in place

    def step(k, i=None):
        b = k & 1
        if world > 1:
            finish(b)                      # buffer set b is free again
        with torch.cuda.stream(streams[b]):
            if i is not None:
                ev[i][0].record(streams[b])
            scene.render_raw(fds[b], streams[b].cuda_stream)
            if i is not None:
                ev[i][1].record(streams[b])
            if world > 1:                  # the frame's single collective, asynchronous, ordered after b's kernel
                src = send2[b] if coll_dev == "cuda" else send2[b].cpu()
                if rank == 0:
                    pending[b] = dist.gather(src, gathered2[b], dst=0, async_op=True)
                else:
                    pending[b] = dist.gather(src, None, dst=0, async_op=True)

    k = 0
    for _ in range(args.warmup):
        step(k)
        k += 1
    if world > 1:
        finish(0)
        finish(1)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(k, i)
        k += 1
    if world > 1:
        finish(k & 1)
        finish((k + 1) & 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    packed = packed2[(k - 1) & 1]
    rgba = rgba2[(k - 1) & 1]

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(t[0]), float(t[1])

    # executed-work statistics from the instrumented kernel variant (untimed)
    stats = scene.render(w, h, y0=y0, y1=y1, want_stats=True, spp=args.spp, cull=not args.no_cull, tile=args.tile,
                         interleave=interleave)["stats"]
    st = torch.tensor([stats[k] for k in rt.STAT_NAMES], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
    stats = dict(zip(rt.STAT_NAMES, [float(v) for v in st.cpu()]))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = w * h * args.spp
        value = rays / (ms_per_step * 1e-3) / 1e6
        band_bytes = rows * w * (16 + 4 + (3 if rgb24 else 0)) * 1.0   # algorithmic bytes of one launch: float4 + packed word (+ 24-bit copy for the gather) per pixel
        achieved = band_bytes / (kernel_ms * 1e-3) / 1e9
        slots = stats["wave_test_slots"] + stats["cull_tests"]     # lane slots issued for sphere/beam tests
        valu_tflops = slots * FLOP_PER_TEST / (ms_per_step * 1e-3) / 1e12
        out = {
            "metric": "primary_Mrays_per_s", "value": value, "unit": "Mrays/s",
            "frames_per_s": 1e3 / ms_per_step,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: {w}x{h}, {args.spheres} spheres (MSVC rand() replay seed {args.seed}), "
                                   f"{args.spp} spp, 3 lights x 10 shadow samples, LDS sphere-tile staging on "
                                   f"(per-tile survivor lists; RT_TABLE_LDS=1 stages the whole table instead)",
                       "cull": not args.no_cull, "tile": args.tile or 8,
                       "parallelism": f"{BLOCK}-row blocks round-robin x{world} + 1 RCCL gather per frame of "
                                      f"{24 if rgb24 else 32}-bit pixels (overlapped with the next frame's kernel)"
                                      if world > 1 else "single GPU",
                       "frames_in_flight": nfl,
                       "outputs": "float4 RGBA + packed 0x00RRGGBB in HBM"},
            "kernel_ms": kernel_ms,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, world),
                         "algorithmic_bytes_per_launch": band_bytes,
                         "achieved_chip": nfl * achieved,
                         "note": "HBM is NOT the binding roof for this path (SURVEY.md F2); see roofline_valu. "
                                 + ("Two frames are in flight on two streams: a launch's duration (kernel_ms, HIP "
                                    "events on its stream) spans two overlapped kernels, so the chip moves "
                                    "achieved_chip = 2 x achieved; per-frame time is ms_per_step."
                                    if nfl == 2 else "One frame in flight.")},
            "roofline_valu": {"bound": "valu_fp32", "achieved": valu_tflops, "peak": VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": valu_tflops / VALU_PEAK_TFLOPS,
                              "executed_test_slots": slots,
                              "note": "64-lane issue slots of sphere + beam tests x 17 flop / frame time; "
                                      "excludes ray setup and shading ALU work (most of the instructions)",
                              **valu_issue(args, world, ms_per_step)},
            "work": {"hit_fraction": stats["hit_pixels"] / rays, "primary_lane_tests_per_ray": stats["primary_tests"] / rays,
                     "shadow_lane_tests_per_ray": stats["shadow_tests"] / rays,
                     "cull_tests_per_ray": stats["cull_tests"] / rays,
                     "list_overflows": stats["list_overflows"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            g_rgba = rgba.cpu().numpy() if args.spp == 1 else None
            g_packed = packed[:rows].cpu().numpy().view(np.uint32)
            cb = cpu_baseline(rt, scene, args, g_rgba, g_packed)
            brute_tests = cb["tests_per_ray"] * rays
            out["cpu_baseline"] = cb
            out["work"]["brute_force_tests_per_ray"] = cb["tests_per_ray"]
            out["work"]["algorithmic_rate_TFLOPs"] = brute_tests * FLOP_PER_TEST / (ms_per_step * 1e-3) / 1e12
            out["work"]["executed_over_brute_force"] = slots / brute_tests
        if world > 1:
            idx = torch.as_tensor(my_rows, device=frame.device)
            out["gathered_frame_shape"] = list(frame.shape)
            out["gathered_rank0_rows_ok"] = bool(torch.equal(frame[idx], packed[:rows].to(frame.device)))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
